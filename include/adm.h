/* adm.h — C-ABI of the MI355X-native audio-diffusion hot path (libadm_hip.so).
 *
 * The reference (teticio/audio-diffusion) is pure Python: its hot path sits behind a duck-typed
 * component API (`AudioDiffusionPipeline(vqvae, unet, mel, scheduler)`,
 * audiodiffusion/pipeline_audio_diffusion.py:53-61), not behind an FFI. This header is the boundary a
 * maintainer binds with ctypes to replace the calls cited per function.
 *
 * Conventions: every function returns 0 on success, non-zero on error (message via
 * adm_last_error()); nothing throws across the ABI; the CALLER owns every I/O buffer (device
 * pointers, fp32 NCHW contiguous unless stated); the library owns only opaque handles;
 * `stream` is a hipStream_t passed as void* (NULL = default stream). One handle = one device = one stream
 * at a time (the reference pipeline is not re-entrant either).
 */
#ifndef ADM_H
#define ADM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int adm_version(void);
const char* adm_last_error(void);
/* 1 if built for the device (hipcc, gfx950), 0 for the CPU-emulation test build. */
int adm_is_device_build(void);
/* Always 0 since round 6 (kept for ABI stability): rounds 1-5 had -DADM_EXPERIMENTS builds carrying superseded Winograd kernel generations
 * ("conv_wino" = 1 / 2 / 3) and ablation / cycle-accounting instantiations; they were retired with that code. */
int adm_has_experiments(void);
/* Runtime options (process-wide; a training net re-learns which weight images it reads after any change):
 * "conv_wino" = 0 (direct MFMA kernel only) | 4 (default: the Winograd kernels wherever they tile the layer; they have their own filter image,
 *   so set the mode BEFORE weights are packed) | -1 (back to the default / ADM_CONV_WINO environment variable); 1 / 2 / 3 (earlier kernel
 *   generations, retired in round 6) are an error;
 * "wino5" = 1 (default, round 5) layers with 128 | Cout whose 128-cout workgroup tiles fill the chip run on conv_wino5_kernel (every input patch
 *   transformed once per 128 output channels; all eight waves are MFMA waves and share the staging work, placed between their MFMA groups) |
 *   0 conv_wino4_kernel everywhere (same filter image, bit-identical results) | bit 1 (2): conv_wino5_kernel for every layer with 128 | Cout,
 *   however few tiles (tests) | bit 3 (8): the two halves of the workgroup run MFMA block and staging block in antiphase instead (the first
 *   schedule built; same results, same speed) | -1 (ADM_WINO5);
 * "wino6" = 1 (default, round 5) 3x3 stride-1 convolutions with 128 | Cout, 32 | Cin, 16 | H, 16 | W on planes of at least 64x64 pixels whose
 *   16x16x128 tiles give one sample at least 32 workgroups run on conv_wino6_kernel: Winograd F(4x4,3x3), 1.78x fewer MFMAs than F(2x2,3x3) at
 *   0.6-1.7e-5 of max|out| per layer (F(2x2): 0.7-1.5e-6; the per-layer bar is 1e-4; profiles/r05_accuracy.md). The choice depends on the layer
 *   only (the two transforms are not bit-identical) | 0 F(2x2,3x3) kernels everywhere | 2 every layer the kernel tiles (tests) | n >= 16: planes
 *   of at least n x n pixels (256: the latency setting — single-sample sampling at 256x256 is 18 % faster with it than with the default, a B = 32
 *   forward 10 % slower; 128: in between) | -1 (ADM_WINO6);
 * "single_sample" = 0 (default) | 1: the partition rules for models that are sampled ONE spectrogram at a time (where a layer's tiles cannot
 *   fill 256 CUs and its time is a serial chain per workgroup). (a) conv_wino4_kernel: layers whose 64-cout x 8x16-pixel tiles give one sample
 *   fewer than 256 workgroups (planes of 16x16 .. 64x64 pixels) and that conv_wino6_kernel does not take split their input channels over
 *   2 / 4 / 8 workgroups per tile (every part a multiple of 32 channels); the partial sums are added in order by a finish launch (which also
 *   leaves the GroupNorm partial sums the unsplit kernel's epilogue would). (b) the split-K 3x3 kernel of the <= 8x8-pixel planes takes 16
 *   parts instead of 4 / 8. The partitions depend on the LAYER only, never on the batch — a sample's bits do not depend on its batch — so with
 *   the rule on a model pays the slab traffic at every batch size: it is a per-model opt-in (adm_unet_set_option; AudioDiffusion selects it).
 *   Results differ from the default rules' in the last bits (another fp32 summation order) | -1 (ADM_SINGLE_SAMPLE);
 * "wino_pair": accepted and ignored since round 6 (conv_wino4_kernel keeps one cadence: one workgroup barrier per two chunks);
 * "wgrad_max_split" = n caps the split-K factor of adm_conv2d_wgrad (0 = heuristic; tests use it to put several pixel tiles on
 *   one workgroup);
 * "conv_bf16" = 1 runs eligible 3x3 stride-1 convolutions (forward, data gradient and weight gradient) on 16-bit MFMA operands
 *   with fp32 accumulation (`--mixed_precision bf16`, scripts/train_unet.py:391-401), 2 = additionally the eligible 1x1 convolutions
 *   and stride-2 data gradients, 0 = fp32 everywhere (default), -1 = back to the ADM_CONV_BF16 environment variable;
 *   3 (round 4; what `enable_training(mixed_precision=...)` selects) = 2, with the 3x3 stride-1 convolutions of all three passes on
 *   blocked 16-bit operand images (adm_blocked_apply / adm_conv2d_bf16_blocked / adm_conv2d_wgrad_bf16_blocked below);
 * "conv_op16_f16" = 1 makes those kernels' operand format IEEE binary16 instead of bf16 (`--mixed_precision fp16`);
 * "blk_direct_dy" = 1 (default; level 3, read when a training plan is made) the backward of a GroupNorm whose input is read by nothing
 *   else writes the producing convolution's dy image directly | 0 fp32 dx tensor + image pass (bit-identical results) | -1 ADM_BLK_DIRECT_DY.
 * "gn_fuse_finish" = 1 (default) a split-K convolution (output planes of <= 8x8 pixels) whose output a GroupNorm reads next leaves
 *   that GroupNorm's scale / shift from its finish pass (one launch instead of finish + statistics; the tensor is bit-identical) |
 *   0 separate launches | -1 ADM_GN_FUSE_FINISH.
 * The dispatch epoch moves only when a value really changes; set options BEFORE adm_unet_refresh_weights / the next train step.
 * adm_version() = 104 since round 6 (adm_conv_args.single_sample, option "single_sample"; 103: adm_conv_args.wino6_rule, adm_unet_set_option, adm_release_stream); 102 since round 5 (Winograd filter buffers hold two images: adm_winograd_packed_floats); 101 since round 4 (adm_slerp_grid takes DOUBLE weights since round 3; blocked-image entry points). */
int adm_set_option(const char* name, int value);
/* Per-(device, stream) scratch the library keeps for a stream (the split-K slab buffer of the small-plane convolutions, >= 1 MiB, at most
 * 1024 streams per device): give it back BEFORE destroying a stream that has run library calls. Drains the stream first (a captured graph of
 * that stream holds the buffer's address: destroy or stop replaying such graphs before). No-op for a stream the library holds nothing for. */
int adm_release_stream(void* stream);
/* Kernel variant the last adm_conv2d on this thread dispatched to (see adm_op_profile.variant; 4311 = Winograd). */
int adm_last_conv_variant(void);

/* ---------------------------------------------------------------- scheduler epilogue (rows S2,S3,P4,P5)
 * One fused elementwise kernel replacing DDIMScheduler.step / DDPMScheduler.step
 * (pipeline_audio_diffusion.py:165-179), the mask overwrite (:181-185) and, on the last step, the
 * dequantisation (images/2+0.5).clamp(0,1)*255 -> round-half-even -> uint8 (:192-194).
 *   x0   = clamp((x - sqrt_beta*eps) / sqrt_alpha, -clip, clip)        (clip < 0: no clamp)
 *   prev = k_x0*x0 + k_x*x + k_eps*eps + k_noise*noise
 * DDIM: k_x0=sqrt(a_prev), k_x=0, k_eps=sqrt(1-a_prev-std^2), k_noise=std (eta>0).
 * DDPM: k_x0=sqrt(a_prev)*cur_beta/beta_t, k_x=sqrt(cur_alpha)*beta_prev/beta_t, k_eps=0, k_noise=sqrt(var) (t>0).
 * The host computes the scalars in the same fp32 arithmetic diffusers uses. */
typedef struct adm_sched_coef {
  float sqrt_beta, sqrt_alpha, clip, k_x0, k_x, k_eps, k_noise, timestep;
} adm_sched_coef;

/* coef_table: device array of adm_sched_coef; entry used = step_dev ? *step_dev : step.
 * noise: NULL or (B,C,H,W). mask: NULL or (B,n_steps,H,W) (requires C==1); columns [0,mask_start) and
 * [W-mask_end,W) of `out` are overwritten with mask[:,step]. u8_out: NULL or (B,H*W*C) uint8 image of `out`.
 * out may alias x. */
int adm_sched_step(const float* x, const float* eps, const float* noise, float* out, uint8_t* u8_out,
                   const adm_sched_coef* coef_table, const int* step_dev, int step,
                   const float* mask, int n_mask_steps, int mask_start, int mask_end,
                   int B, int C, int H, int W, void* stream);

/* scheduler.add_noise (rows S4,P3,T3): out[b][n][p] = sa[b*cb+n*cn]*x0[b*x0_bstride+p] + sb[..]*noise[b*P+p];
 * sa/sb are device arrays (sqrt(acp[t]), sqrt(1-acp[t])). */
int adm_add_noise(const float* x0, long x0_bstride, const float* noise, const float* sa, const float* sb,
                  int cb, int cn, float* out, int B, int N, long P, void* stream);

/* (x/2+0.5).clamp(0,1)*255 round-half-even -> u8 (pipeline_audio_diffusion.py:192-194). */
int adm_dequant_u8(const float* x, uint8_t* out, long n, void* stream);
/* AudioDiffusionPipeline.slerp (pipeline_audio_diffusion.py:244-258) for a whole grid of interpolation weights at once:
 * out (n_alpha, n) = sin((1-alpha) theta) x0 / sin(theta) + sin(alpha theta) x1 / sin(theta), theta the angle between x0 and
 * x1 (n floats each); alphas_dev: device double[n_alpha] (the reference evaluates sin((1 - alpha) * theta) on Python doubles); scratch3: device double[3]. */
int adm_slerp_grid(const float* x0, const float* x1, long n, const double* alphas_dev, int n_alpha, float* out,
                   double* scratch3, void* stream);

/* ---------------------------------------------------------------- op-level entry points (parity tests)
 * GroupNorm statistics over a (virtually concatenated) NCHW input: writes per-(n,c) scale/shift with
 *   scale = rstd*gamma[c], shift = beta[c] - mean*scale   (torch.nn.GroupNorm, biased variance). */
int adm_groupnorm_stats(const float* x1, int C1, const float* x2, int C2, int N, int HW, int groups, float eps,
                        const float* gamma, const float* beta, float* scale, float* shift, void* stream);

/* Fused 2-D convolution (rows U2-U5,U7,U8): implicit GEMM on v_mfma_f32_32x32x2_f32.
 *   in  = concat(x1[C1], x2[C2]) (x2 may be NULL), optional nearest x2 upsample (up=1) or zero-insertion x2 (up=2,
 *         the transposed stride-2 conv of the backward pass) of the input,
 *   optional per-(n,c) affine (gn_scale/gn_shift from adm_groupnorm_stats) and SiLU (act=1) applied on load,
 *   out = conv(in, w, stride, pad) + bias[co] + chan_add[n][co] (NULL ok) + residual[n][co][y][x] (NULL ok).
 * wpacked: weights repacked by adm_pack_conv_weight to [Cin][ks*ks][Cout]. H,W are the *source* dims of x1/x2.
 * pad_lo is the top/left zero padding (1 for ks=3 "same"; 0 with stride 2 gives diffusers' (0,1,0,1) pad). */
typedef struct adm_conv_args {
  const float* x1; int C1;
  const float* x2; int C2;
  int N, H, W;
  int up, stride, ks, pad_lo;
  const float* gn_scale; const float* gn_shift; int act;
  const float* wpacked; const float* bias; int Cout;
  const float* chan_add; int chan_add_stride;
  const float* residual;
  float* out;
  /* optional (0 = default): batch strides in elements of x1/x2 when they are channel slices of wider tensors, and
   * of `wpacked` when every sample has its own weights (activation x activation products, e.g. Q K^T). */
  long x1_bstride, x2_bstride, w_bstride;
  /* optional: the same 3x3 weights pre-transformed by adm_pack_winograd_weight ([Cin][16][Cout]); when present and the
   * shape is eligible (stride 1, output >= 8x16) the Winograd F(2x2,3x3) kernel may be used (ADM_CONV_WINO=1). */
  const float* wino_packed;
  /* optional: the same 3x3 weights as bf16 MFMA operands (adm_pack_bf16_weight: [tap][Cin/8][Cout][8] bf16); used when
   * option "conv_bf16" is on and the shape is eligible (stride 1, output a multiple of 16x16, Cout % 128 == 0). */
  const void* bf16_packed;
  /* optional: GroupNorm statistics of the OUTPUT, produced by the convolution's own epilogue instead of a separate read
   * pass: per (sample, output channel, pixel tile) the fp64 pair (sum, sum of squares) of the final output values,
   * stats_out[((n * Cout + c) * stats_tiles + tile) * 2 + {0, 1}]. stats_tiles must equal adm_conv_stats_tiles(args) (> 0
   * only for the kernels that can do it: the Winograd v4 kernel and the conv_in class kernel (Cin <= 4, W % 4 == 0));
   * adm_groupnorm_finalize turns them into scale / shift. */
  double* stats_out; int stats_tiles;
  /* optional (0 = the process-wide "wino6" option decides): this call's own F(4x4) layer rule, same values as the option (1 default rule,
   * 2 every layer the kernel tiles, n >= 16 planes of at least n x n pixels). A model handle carries one (adm_unet_set_option), so that a
   * single-sample front end can run the latency rule on ITS model without touching other models in the process. Since adm_version() 103. */
  int wino6_rule;
  /* optional (0 = the process-wide "single_sample" option decides; 1 on; -1 off): this call's (its model's) single-sample partition rules —
   * see the option. Since adm_version() 104. */
  int single_sample;
} adm_conv_args;
/* number of statistic tiles per (sample, channel) the kernel chosen for these arguments would emit, 0 = it cannot. */
int adm_conv_stats_tiles(const adm_conv_args* a);
/* GroupNorm scale / shift (as adm_groupnorm_stats) from the per-tile partial sums of one tensor or of a virtual concat
 * (x1 | x2) whose parts were produced by different convolutions (stats2 NULL / C2 0: one part); HW = pixels per channel. */
int adm_groupnorm_finalize(const double* stats1, int C1, int tiles1, const double* stats2, int C2, int tiles2, int N, int HW,
                           int groups, float eps, const float* gamma, const float* beta, float* scale, float* shift,
                           void* stream);
int adm_conv2d(const adm_conv_args* a, void* stream);
/* (Cout,Cin,ks,ks) -> [Cin][ks*ks][Cout]; both device pointers. */
int adm_pack_conv_weight(const float* w, float* wpacked, int Cout, int Cin, int ks, void* stream);
/* Backward-data form: (Cout,Cin,ks,ks) -> [Cout][ks*ks][Cin] (channel-transposed, spatially flipped). Feeding it to
 * adm_conv2d with x1 = dy (Cout channels) yields d(input): stride-1 convs directly; stride-2 convs with up = 2
 * (zero-insertion); nearest-upsampled convs give the gradient at the upsampled resolution (then adm_sumpool2x2). */
int adm_pack_conv_weight_T(const float* w, float* wpT, int Cout, int Cin, int ks, void* stream);
/* Number of floats a Winograd filter buffer (wu: transposed = 0, wuT: transposed = 1) must hold: the F(2x2,3x3) image (16 per filter) and,
 * where the channel counts allow conv_wino6_kernel (128 | output channels, 32 | input channels of the packed convolution), the F(4x4,3x3)
 * image (36 per filter) behind it. adm_version() >= 102. */
long adm_winograd_packed_floats(int Cout, int Cin, int transposed);
/* (Cout,Cin,3,3) -> Winograd-domain weights U = G g G^T into a buffer of adm_winograd_packed_floats(Cout, Cin, 0) floats (kernel-specific
 * layouts: [Cin][16][Cout], or the MFMA-fragment images of conv_wino4/5/6_kernel); both device pointers. */
int adm_pack_winograd_weight(const float* w, float* wu, int Cout, int Cin, void* stream);
/* ... of the data-gradient convolution (input channels = Cout, output channels = Cin, taps flipped): [Cout][16][Cin].
 * Pass it as `wino_packed` together with adm_pack_conv_weight_T's packing to run the backward-data pass of a 3x3
 * stride-1 Conv2d (what torch autograd does for scripts/train_unet.py:262 `accelerator.backward(loss)`). */
int adm_pack_winograd_weight_T(const float* w, float* wuT, int Cout, int Cin, void* stream);
/* (Cout,Cin,3,3) fp32 -> bf16 (round-to-nearest-even) MFMA operand layout [tap][Cin/8][Cout][8]; transposed != 0 packs
 * the data-gradient filters instead ([flipped tap][Cout/8][Cin][8]). 2 bytes per weight; both device pointers. */
int adm_pack_bf16_weight(const float* w, void* wb, int Cout, int Cin, int transposed, void* stream);
/* the same for (Cout,Cin,ks,ks), ks = 3 or 1 (1x1: [Cin/8][Cout][8]; used when option "conv_bf16" = 2 also puts the
 * 1x1 convolutions on bf16 operands — opt-in, see k_conv1x1_bf16.hip). */
int adm_pack_bf16_weight_ks(const float* w, void* wb, int Cout, int Cin, int ks, int transposed, void* stream);
void adm_conv_out_dims(int H, int W, int up, int stride, int ks, int pad_lo, int* Ho, int* Wo);

/* Self-attention core (row U6): qkv is (N, 3*C, T) with channels [q | k | v], head h = channels
 * [h*d, (h+1)*d); out (N, C, T) = softmax(q^T k * d^-0.5) v per head, fp32 softmax. */
int adm_attention(const float* qkv, float* out, int N, int C, int T, int head_dim, void* stream);

/* Transformer2DModel pieces of UNet2DConditionModel (scripts/train_unet.py:139-159; diffusers transformer_2d.py /
 * attention.py), all on (N, C, T) activations:
 *   adm_layernorm_nct: LayerNorm over the channel axis for every token (norm1/2/3 of BasicTransformerBlock).
 *   adm_geglu: (N, 2*C4, T) = [h | gate] -> (N, C4, T) = h * gelu(gate) (exact erf GELU).
 *   adm_cross_attention: tokens attend to the encoding ctx (N, S, Dc) through to_k / to_v weights (C, Dc), q given.
 *   adm_attention_blocked: adm_attention with the keys in blocks of `key_block` (0 = 64 KiB of K+V) and an online
 *   softmax; adm_attention switches to it by itself when a head's K/V exceed 64 KiB.
 *   Round 5: for head_dim 16 / 32 / 64 and T % 128 == 0 (T % 256 for head_dim <= 32) — the Transformer2DModel blocks of the conditional UNet —
 *   adm_attention runs the flash form on the f32 matrix pipe (attention_mfma_kernel: both products on v_mfma_f32_16x16x4_f32, online softmax;
 *   the choice depends on the layer's shape only). */
int adm_layernorm_nct(const float* x, const float* gamma, const float* beta, float* y, int N, int C, long T, float eps,
                      void* stream);
int adm_geglu(const float* in, float* out, int N, int C4, long T, void* stream);
int adm_cross_attention(const float* q, const float* ctx, const float* Wk, const float* Wv, float* out, int N, int C, int T,
                        int S, int Dc, int head_dim, void* stream);
int adm_attention_blocked(const float* qkv, float* out, int N, int C, int T, int head_dim, int key_block, void* stream);
/* 1 if adm_attention runs this shape on the f32 matrix pipe (attention_mfma_kernel), 0 if on the VALU kernels. */
int adm_attention_mfma_eligible(int C, int T, int head_dim);
/* Backward passes of the three (training of the conditional UNet, scripts/train_unet.py:254-259 with --encodings):
 *   adm_layernorm_nct_backward: dx (accumulate != 0: +=), dgamma += , dbeta += ; stats: 2*N*T floats of scratch.
 *   adm_geglu_backward: d(in) (N, 2*C4, T) from dy (N, C4, T).
 *   adm_cross_attention_backward: dq (N, C, T); dWk, dWv (C, Dc) += (atomics). The encoding receives no gradient. */
int adm_layernorm_nct_backward(const float* x, const float* dy, const float* gamma, float* dx, int accumulate, float* stats,
                               float* dgamma, float* dbeta, int N, int C, long T, float eps, void* stream);
int adm_geglu_backward(const float* in, const float* dy, float* din, int N, int C4, long T, void* stream);
/* adm_attention_backward for any token count: keys / queries in LDS blocks of `block` (0 = 64 KiB), probabilities
 * recomputed; stats: 3 * N * (C / head_dim) * T floats of scratch. */
int adm_attention_backward_blocked(const float* qkv, const float* dout, float* dqkv, float* stats, int N, int C, int T,
                                   int head_dim, int block, void* stream);
int adm_cross_attention_backward(const float* q, const float* ctx, const float* Wk, const float* Wv, const float* dy,
                                 float* dq, float* dWk, float* dWv, int N, int C, int T, int S, int Dc, int head_dim,
                                 void* stream);

/* AudioEncoder (audiodiffusion/audio_encoder.py:62-84; produces the `encoding` of the conditional models), eval mode:
 *   adm_sepconv_block: ConvBlock = depthwise 3x3 (no bias, dw (Ci,1,3,3)) -> pointwise 1x1 (pw (Co,Ci), pb) ->
 *     LeakyReLU(slope) -> BatchNorm2d folded to bn_scale/bn_shift -> MaxPool 2x2; x (N,Ci,H,W) -> y (N,Co,H/2,W/2);
 *     tmp: N*Ci*H*W floats of scratch.
 *   adm_dense_act: y (N,J) = post(b + W x), W (J,K); leaky != 0 applies LeakyReLU(slope), post_scale/shift (NULL ok) a
 *     folded BatchNorm1d; hwc_C > 0 reads x (N, hwc_C, K/hwc_C) as if flattened in NHWC order (DenseBlock, :55). */
int adm_sepconv_block(const float* x, const float* dw, const float* pw, const float* pb, const float* bn_scale,
                      const float* bn_shift, float slope, float* tmp, float* y, int N, int Ci, int Co, int H, int W,
                      void* stream);
int adm_dense_act(const float* x, const float* W, const float* b, const float* post_scale, const float* post_shift,
                  float slope, int leaky, float* y, int N, int K, int J, int hwc_C, void* stream);

/* ---------------------------------------------------------------- UNet2DModel executor (rows U1-U8)
 * Replaces `self.unet(images, t)["sample"]` (pipeline_audio_diffusion.py:163,237; train_unet.py:257). */
typedef struct adm_unet adm_unet_t;
typedef struct adm_unet_config {
  int in_channels, out_channels, layers_per_block, n_blocks;
  int block_out_channels[8];
  int down_attn[8]; /* 1 = AttnDownBlock2D, 2 = CrossAttnDownBlock2D (UNet2DConditionModel) */
  int up_attn[8];   /* 1 = AttnUpBlock2D,   2 = CrossAttnUpBlock2D */
  int attention_head_dim, norm_num_groups; /* conditional model: attention_head_dim is the NUMBER of heads (diffusers 0.24) */
  float norm_eps;
  int flip_sin_to_cos;
  float freq_shift;
  int sample_h, sample_w;
  /* > 0: UNet2DConditionModel (scripts/train_unet.py:139-159): width of the encoding; the mid block is then
   * UNetMidBlock2DCrossAttn and adm_unet_set_encoding must be called before a forward / sample loop. */
  int cross_attention_dim;
} adm_unet_config;

int adm_unet_create(const adm_unet_config* cfg, adm_unet_t** out);
/* Per-MODEL options (adm_version() >= 103; "single_sample" >= 104: 0 = follow the process-wide option, 1 = on, -1 = off for this model —
 * AudioDiffusion sets 1: one 256x256 sample per call 6.6 -> 4.2 ms per step, profiles/r06_single_sample.md). "wino6": this model's F(4x4) layer rule — 0 = follow the process-wide option (default), 1 / 2 /
 * n >= 16 as adm_set_option("wino6", .). `audiodiffusion.AudioDiffusion` (the reference's single-sample facade, audiodiffusion/__init__.py:58-68:
 * batch_size is forced to 1) sets 256 on its own model: planes whose 16x16x128 tiles fill the chip with ONE sample keep F(4x4), the
 * levels below run the 64-cout F(2x2) kernel, whose smaller tiles are 4x as many workgroups (256x256, one sample: 6.8 instead of 8.3 ms per
 * step). The rule stays a function of the layer and the model — never of the batch — so rows of one model's batches remain bit-identical to
 * single-sample runs of THAT model. Re-plans the model (drops a captured loop). */
int adm_unet_set_option(adm_unet_t* h, const char* name, int value);
void adm_unet_destroy(adm_unet_t* h);
/* Upload one parameter by its diffusers state-dict key (host pointer, fp32, `numel` elements).
 * Deprecated attention names (query/key/value/proj_attn, audiodiffusion/utils.py:41-54) are accepted. */
int adm_unet_set_param(adm_unet_t* h, const char* key, const float* host_data, size_t numel);
/* Conditional model: encoder_hidden_states for the following forwards / sample loops — device pointer (B, seq_len,
 * cross_attention_dim) fp32, owned by the caller and kept alive until replaced
 * (`self.unet(images, t, encoding)`, pipeline_audio_diffusion.py:160-161). */
int adm_unet_set_encoding(adm_unet_t* h, const float* encoding_dev, int seq_len);
/* Number of parameters still missing after the set_param calls (0 = ready); names via adm_last_error(). */
int adm_unet_missing_params(adm_unet_t* h);
/* eps = unet(x, t): x,out (B,Cin,H,W)/(B,Cout,H,W) device; timesteps: B floats on the HOST (or 1 broadcast). */
int adm_unet_forward(adm_unet_t* h, const float* x, const float* timesteps_host, int n_timesteps, float* out,
                     int B, void* stream);
size_t adm_unet_workspace_bytes(adm_unet_t* h);
/* Measurement aid (bench.py roofline leg): one eager forward with a HIP-event pair around every launch on `stream`.
 * kind: 0 GroupNorm stats, 1 MFMA conv, 2 attention core, 3 direct small-channel conv, 4 time-embedding projection.
 * variant (MFMA conv): ks*100 + stride*10 + cout_tile/32, +2000 for the software-pipelined kernel
 * (conv_mfma_pf_kernel); 1001/1002 = direct small-Cin / small-Cout kernels. flops/bytes are ALGORITHMIC. */
typedef struct adm_op_profile { int kind, variant; float ms; double flops, bytes; } adm_op_profile;
int adm_unet_profile(adm_unet_t* h, const float* x, float timestep, float* out, int B, adm_op_profile* recs, int cap,
                     int* n_out, void* stream);

/* ---- training (rows T4,T5; scripts/train_unet.py:257-259): the master parameters live in ONE caller-owned flat fp32
 * device buffer (so the fused optimizer kernel can update them in a single launch); call order:
 *   create -> adm_unet_bind_param for every key -> adm_unet_enable_training -> adm_unet_forward_backward ...
 *   -> (optimizer step on the flat buffer) -> adm_unet_refresh_weights -> next step. */
int adm_unet_bind_param(adm_unet_t* h, const char* key, float* dev_ptr);
int adm_unet_enable_training(adm_unet_t* h, const float* params_base, long numel);
int adm_unet_refresh_weights(adm_unet_t* h, void* stream);
/* loss_dev[0] = mean((unet(x,t) - target)^2); grads_base (flat, same offsets as the parameter buffer) = d loss/d params.
 * Every activation is kept for the reverse pass; GroupNorm/SiLU are recomputed in the weight-gradient load path. */
int adm_unet_forward_backward(adm_unet_t* h, const float* x, const float* timesteps_host, int n_timesteps,
                              const float* target, float* loss_dev, float* grads_base, int B, void* stream);
/* `--mixed_precision fp16`: the factor the loss gradient is multiplied with before the reverse pass (GradScaler's scale;
 * 1 = off). The gradients adm_unet_forward_backward writes then carry it: un-scale with adm_grad_norm_clip_scaled. */
int adm_unet_set_loss_scale(adm_unet_t* h, float scale);
/* Data-parallel overlap (DistributedDataParallel's bucketed all-reduce running under autograd, train_unet.py:259): cut the
 * flat gradient buffer into n_buckets ranges [bounds[b], bounds[b+1]) (elements, ascending, n_buckets + 1 values);
 * fn(user, b) is called on the calling thread during adm_unet_forward_backward as soon as the last kernel writing into
 * bucket b has been ENQUEUED on the stream — queue the bucket's all-reduce behind it (RCCL orders after the stream) while
 * the rest of the reverse pass runs. Buckets holding a parameter the pass never touches do not fire: reduce those after
 * the call returns. n_buckets = 0 removes the hook. */
typedef void (*adm_bucket_fn)(void* user, int bucket);
int adm_unet_set_grad_bucket_hook(adm_unet_t* h, int n_buckets, const long* bounds, adm_bucket_fn fn, void* user);

/* ---------------------------------------------------------------- whole denoising loop (row P4; hipGraph)
 * Runs n_steps x {UNet forward, scheduler epilogue, mask} on `x` in place and (optionally) the final u8 image.
 * coef_host: n_steps adm_sched_coef (host) including .timestep; step_noise: NULL or device
 * (n_steps,B,C,H,W) noise consumed where k_noise != 0. With use_graph=1 one step is captured into a hipGraph
 * and replayed n_steps times (step-dependent scalars come from a device table indexed by a device counter). */
int adm_sample_loop(adm_unet_t* h, float* x, int B, const adm_sched_coef* coef_host, int n_steps,
                    const float* step_noise, const float* mask, int mask_start, int mask_end,
                    uint8_t* u8_out, int use_graph, void* stream);
/* DDIM inversion loop (row P6, pipeline_audio_diffusion.py:228-240): per step
 *   x = (x - c_dir*eps) * c_inv * c_fwd + c_eps*eps  with coef {sqrt_beta=c_dir, sqrt_alpha=c_inv, k_x0=c_fwd, k_eps=c_eps}. */
int adm_encode_loop(adm_unet_t* h, float* x, int B, const adm_sched_coef* coef_host, int n_steps, int use_graph,
                    void* stream);

/* ---------------------------------------------------------------- AutoencoderKL executor (rows V1-V3, config 4)
 * Replaces `vqvae.encode(x).latent_dist.sample(generator)` (pipeline_audio_diffusion.py:144; train_unet.py:104,233)
 * and `vqvae.decode(z)["sample"]` (pipeline_audio_diffusion.py:190). Architecture per audiodiffusion/utils.py:132-153:
 * Down/UpDecoderBlock2D stacks, GroupNorm eps 1e-6, single-head mid-block attention. */
typedef struct adm_vae adm_vae_t;
typedef struct adm_vae_config {
  int in_channels, out_channels, latent_channels, layers_per_block, n_blocks;
  int block_out_channels[8];
  int norm_num_groups;
  int sample_h, sample_w;
} adm_vae_config;
int adm_vae_create(const adm_vae_config* cfg, adm_vae_t** out);
void adm_vae_destroy(adm_vae_t* h);
int adm_vae_set_param(adm_vae_t* h, const char* key, const float* host_data, size_t numel);
int adm_vae_latent_dims(adm_vae_t* h, int* lat_h, int* lat_w);
/* z_out (B,Cz,h,w) = (mean + exp(0.5*clamp(logvar,-30,20)) * noise) * out_scale, noise (B,Cz,h,w) or NULL (= mode);
 * moments_out: NULL or (B,2*Cz,h,w) receives quant_conv(encoder(x)). out_scale carries the reference's 0.18215. */
int adm_vae_encode(adm_vae_t* h, const float* x, const float* noise, float out_scale, float* z_out, float* moments_out,
                   int B, void* stream);
/* out (B,Cout,H,W) = decoder(post_quant_conv(in_scale * z)); in_scale carries the reference's 1/0.18215. */
int adm_vae_decode(adm_vae_t* h, const float* z, float in_scale, float* out, int B, void* stream);

/* ---------------------------------------------------------------- training-step optimizer side (rows T4,T6,T7,T9)
 * scripts/train_unet.py:258-267 over FLAT fp32 buffers (all parameters / grads / moments / EMA shadow contiguous).
 * Nothing here synchronises with the host: scalars (loss, norm, clip factor) stay on the device. */
/* F.mse_loss (:258): loss_out[0] = mean((pred-target)^2); grad_out (NULL ok) = 2(pred-target)/n; scratch: double[1]. */
int adm_mse_loss(const float* pred, const float* target, long n, float* loss_out, float* grad_out, double* scratch,
                 void* stream);
/* clip_grad_norm_ (:261-262): norm_clip_out = {||g||_2, min(1, max_norm/(||g||_2+1e-6))}; scratch: double[1]. */
int adm_grad_norm_clip(const float* grads, long n, float max_norm, float* norm_clip_out, double* scratch, void* stream);
/* the same on loss-scaled gradients (`--mixed_precision fp16`: GradScaler.unscale_ + clip_grad_norm_ in one pass):
 * norm_clip_out = {||g||_2 * inv_scale, min(1, max_norm/(norm + 1e-6)) * inv_scale}; inf / nan gradients give a non-finite
 * norm, on which the caller skips the optimizer step and backs the scale off. */
int adm_grad_norm_clip_scaled(const float* grads, long n, float max_norm, float inv_scale, float* norm_clip_out, double* scratch,
                              void* stream);
/* torch.optim.AdamW step (:263; lr/betas/wd/eps :166-172) fused with the clip factor (device scalar, NULL = 1) and
 * diffusers EMAModel.step (:265-266; ema == NULL skips): shadow -= (1-ema_decay)*(shadow-param). step is 1-based. */
int adm_adamw_ema_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema, long n, float lr,
                       float beta1, float beta2, float eps, float weight_decay, int step, const float* clip_coef_dev,
                       float ema_decay, void* stream);
/* the remaining optimizer-side passes over a flat buffer, one launch each: op 0: y += x (accelerator.accumulate micro-step
 * sum, :252); 1: y = x*a (mean of the accumulated micro-gradients); 2: y /= a (DDP's 1/world, :259); 3: y -= a*(y-x)
 * (EMAModel.step on its own, a = 1-decay, :265-266). Buffers 16-byte aligned. */
int adm_flat_op(float* y, const float* x, long n, int op, float a, void* stream);

/* ---------------------------------------------------------------- backward ops of the training step (rows T5/T8)
 * What `accelerator.backward(loss)` (scripts/train_unet.py:259-262) asks torch autograd to run for the ops of
 * UNet2DModel; adm_unet_forward_backward chains them natively, these entry points expose each op for parity tests and
 * for a host that wants to drive the backward pass itself. All pointers are device pointers. */
/* adm_groupnorm_stats that also keeps (N, groups, 2) {mean, rstd} for the backward pass (mean_rstd may be NULL). */
int adm_groupnorm_stats_ex(const float* x1, int C1, const float* x2, int C2, int N, int HW, int groups, float eps,
                           const float* gamma, const float* beta, float* scale, float* shift, float* mean_rstd,
                           void* stream);
/* Backward of GroupNorm(+SiLU) applied on a conv's load path: da = gradient w.r.t. the activated tensor (virtual concat
 * of x1|x2); writes/accumulates dx1, dx2 (acc1/acc2 != 0: add) and ACCUMULATES dgamma, dbeta. s12_scratch: N*groups*2. */
int adm_groupnorm_backward(const float* x1, int C1, const float* x2, int C2, const float* da, int N, int HW, int groups,
                           const float* mean_rstd, const float* gamma, const float* beta, int act, float* s12_scratch,
                           float* dgamma, float* dbeta, float* dx1, int acc1, float* dx2, int acc2, void* stream);
/* Conv2d weight gradient dW (Cout,Cin,ks,ks) of the fused convolution described by `a` (same load-path fusions) for the
 * output gradient dy; split-K partial slabs live in `workspace` (adm_conv_wgrad_workspace floats). */
long adm_conv_wgrad_workspace(const adm_conv_args* a);
int adm_conv2d_wgrad(const adm_conv_args* a, const float* dy, float* dW, int accumulate, float* workspace, void* stream);
/* ---- blocked 16-bit operand images (option "conv_bf16" = 3; `--mixed_precision bf16`, scripts/train_unet.py:391-401) ----
 * A blocked image is  img[n][C/8][H+2][W+2] x 16 B : one 16-byte unit = 8 consecutive channels of one pixel as bf16 (or IEEE
 * binary16 under option "conv_op16_f16"), with a one-pixel halo that must be ZERO (zero the buffer once after allocating it:
 * adm_blocked_apply writes the interior only).  adm_blocked_image_bytes gives its size.
 *   adm_blocked_apply: img = round16(act(scale[n][c] * concat(x1, x2)[n][c] + shift[n][c])), scale / shift as produced by
 *     adm_groupnorm_stats (both NULL: identity), act != 0: SiLU — the activated conv input, written ONCE per layer; with
 *     scale = NULL, act = 0 on dy it is the operand image of the backward kernels.  C1, C2 multiples of 8, W of 4.
 *     sum_nc (N, nc_stride): NULL or receives the per-(n, c) sums of the fp32 INPUT; sum_c (C): NULL or ACCUMULATES the per-c
 *     sums (the time-embedding-bias and bias gradients when the input is dy, as adm_chan_sums); both need sum_scratch
 *     (adm_blocked_sums_scratch floats: one partial per workgroup, added in a fixed order).
 *   adm_conv2d_bf16_blocked: out (N,Cout,H,W) fp32 = conv3x3(img, stride 1, pad 1) + bias[co] + chan_add[n][co] + residual
 *     (each NULL ok), filters `wb` from adm_pack_bf16_weight (transposed = 1 and img = image of dy: the data gradient).
 *     Cin % 16 == 0, Cin >= 32, Cout % 128 == 0, H % 8 == 0, W % 32 == 0 (adm_conv2d_bf16_blocked_eligible) — or W == 16 / W == 8 with
 *     Cin >= 64 and N % 2 / N % 4 == 0: two / four images side by side in one 32-column tile, K split over workgroups and finished in
 *     slab order through the library's per-(device, stream) scratch (no stats_out, no stride-2 variant; up = 1 at W == 16 only).
 *   adm_conv2d_wgrad_bf16_blocked: dW (Cout,Cin,3,3) (+)= sum over pixels of dy x activated input, both as blocked images;
 *     Cin % 64 == 0, Cout % 128 == 0, H % 4 == 0, W % 32 == 0 (or W == 16 / 8 with N % 2 / N % 4 == 0, as above); workspace:
 *     adm_conv_wgrad_blocked_workspace floats (0 = shape not eligible for this N).
 *     stats_out: NULL or (N, Cout, (H/8)*(W/32), 2) fp64 — per 8x32-pixel tile the (sum, sum of squares) of the final output
 *     values, the input adm_groupnorm_finalize needs (as adm_conv_args.stats_out).
 *   up = 1 (both): the convolution of Upsample2D — H, W are the OUTPUT dims and the input image is the half-resolution tensor
 *     (N, Cin, H/2, W/2); the nearest x2 is folded into the patch addresses.
 *   Stride 2 (Downsample2D.conv) = every other pixel of the stride-1 "same" convolution o of the same image: adm_conv2d_bf16_blocked
 *     with up = 2 for pad (0, 1, 0, 1) + 3x3 stride 2 without padding (AutoencoderKL encoder; out(y, x) = o(2y + 1, 2x + 1)) or up = 3
 *     for 3x3 stride 2 padding 1 (UNet2DModel; out(y, x) = o(2y, 2x)) — H, W are the INPUT dims, out is (N, Cout, H/2, W/2), bias only.
 *     Its backward passes take the ZERO-INSERTED image of dy: adm_blocked_apply(zero_insert = 1 / 2 for up = 2 / 3) on dy
 *     (N, Cout, H/2, W/2) writes an image of (H, W) pixels (adm_blocked_image_bytes(N, Cout, H, W), zeroed once) with dy(y, x) on pixel
 *     (2y + 1, 2x + 1) / (2y, 2x); the data gradient is then the plain stride-1 call with the transposed filters, the weight gradient
 *     the plain call with the input's image. */
size_t adm_blocked_image_bytes(int N, int C, int H, int W);
long adm_blocked_sums_scratch(int N, int C, int H, int W);
int adm_blocked_apply(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, const float* scale,
                      const float* shift, int act, int zero_insert, void* img, float* sum_scratch, float* sum_nc, int nc_stride,
                      float* sum_c, void* stream);
int adm_conv2d_bf16_blocked_eligible(int Cin, int Cout, int H, int W);
int adm_conv2d_bf16_blocked(const void* img, int Cin, int N, int H, int W, const void* wb, int Cout, const float* bias,
                            const float* chan_add, int chan_add_stride, const float* residual, float* out, int up,
                            double* stats_out, void* stream);
int adm_conv2d_wgrad_bf16_blocked_eligible(int Cin, int Cout, int H, int W);
long adm_conv_wgrad_blocked_workspace(int Cin, int Cout, int N, int H, int W);
int adm_conv2d_wgrad_bf16_blocked(const void* x_img, int Cin, const void* dy_img, int Cout, int N, int H, int W, float* dW,
                                  int accumulate, float* workspace, int up, void* stream);
/* gradient of the nearest-x2 upsample folded into a conv: out (planes,H/2,W/2) (+)= 2x2 sums of in (planes,H,W). */
int adm_sumpool2x2(const float* in, float* out, int H, int W, long planes, int accumulate, void* stream);
/* dst[n][0:per_sample] (+)= src[n][0:per_sample] with independent batch strides (gradient fan-in of channel slices). */
int adm_accumulate(float* dst, long dst_bs, const float* src, long src_bs, long per_sample, int N, int accumulate,
                   void* stream);
/* per-(n,c) sums of dy over HW into out_nc (time-embedding bias gradient, NULL ok) and per-c sums into out_c (bias
 * gradient, accumulated, NULL ok). */
int adm_chan_sums(const float* dy, int N, int C, int HW, float* out_nc, int nc_stride, int nc_accumulate, float* out_c,
                  void* stream);
/* backward of adm_attention: dqkv (N,3C,T) from dout (N,C,T) and the saved qkv. */
int adm_attention_backward(const float* qkv, const float* dout, float* dqkv, int N, int C, int T, int head_dim,
                           void* stream);
/* Linear Y = act(X) W^T + b: dW (J,K) += dY^T act(X), db += colsum(dY), dX = (dY W) * act'(X); x_silu: act = SiLU. */
int adm_linear_backward(const float* dY, int ldy, const float* X, const float* W, int B, int J, int K, int x_silu,
                        float* dW, float* db, float* dX, void* stream);
/* conv_in (Cin <= 4) weight gradient and conv_out (Cout <= 4) weight + data gradient (the direct small-channel kernels). */
int adm_conv_small_cin_wgrad(const float* x, int Cin, int N, int H, int W, const float* dy, int Cout, float* dW,
                             void* stream);
int adm_conv_small_cout_backward(const float* x, int Cin, int N, int H, int W, const float* gn_scale,
                                 const float* gn_shift, int act, const float* w, const float* dy, int Cout, float* da,
                                 float* dW, void* stream);

/* ---------------------------------------------------------------- Mel codec (rows M3-M8)
 * Replaces Mel.audio_slice_to_image (audiodiffusion/mel.py:135-151: librosa melspectrogram + power_to_db + u8) and
 * Mel.image_to_audio (mel.py:153-168: db_to_power + mel_to_stft NNLS + Griffin-Lim), batched over slices/images.
 * The constant tables of a configuration are computed by the host binding exactly as librosa/scipy do:
 * window (n_fft fp64, periodic Hann), twiddle (n_fft/2 complex fp64 exp(-2 pi i q/n)), the Slaney filterbank as CSR
 * (fb_start/fb_count per mel row, fb_w32/fb_w64 taps) and CSC (fbt_off/fbt_idx/fbt_w64), pinv (n_bins x n_mels fp64),
 * wss (window sum-square envelope, fp32, trimmed to hop*(x_res-1)), nnls_cols (librosa.util.nnls column blocking). */
typedef struct adm_mel adm_mel_t;
typedef struct adm_mel_config { int x_res, y_res, sample_rate, n_fft, hop_length, top_db, n_iter; } adm_mel_config;
int adm_mel_create(const adm_mel_config* cfg, const double* window, const double* twiddle, const int* fb_start,
                   const int* fb_count, const float* fb_w32, const double* fb_w64, int nnz, const int* fbt_off,
                   const int* fbt_idx, const double* fbt_w64, const double* pinv, const float* wss, int nnls_cols,
                   adm_mel_t** out);
void adm_mel_destroy(adm_mel_t* h);
/* audio: device, B slices of n_samples fp32 (is_f64=0) or fp64 (=1), slice_stride elements apart;
 * image_out: device (B, y_res, 1 + n_samples/hop) uint8. */
int adm_mel_forward(adm_mel_t* h, const void* audio, int is_f64, int B, long slice_stride, int n_samples,
                    uint8_t* image_out, void* stream);
/* the mel power spectrogram before the dB conversion (librosa.feature.melspectrogram, audiodiffusion/mel.py:140-147):
 * melspec_out: device (B, y_res, 1 + n_samples/hop) in the audio's precision (fp32 or fp64). */
int adm_mel_forward_power(adm_mel_t* h, const void* audio, int is_f64, int B, long slice_stride, int n_samples,
                          void* melspec_out, void* stream);
/* images: device (B, y_res, n_frames) uint8; init_phase: device (B, n_bins, n_frames) fp64 in [0,1) (Griffin-Lim start
 * phase / 2 pi); audio_out: device (B, hop*(n_frames-1)) fp32; stft_mag_out: NULL or device (B, n_frames, n_bins) fp64;
 * pg_max_host: NULL or receives max |projected gradient| of the NNLS solution returned (librosa's rule: <= 1e-5). */
int adm_mel_inverse(adm_mel_t* h, const uint8_t* images, const double* init_phase, int B, int n_frames,
                    float* audio_out, double* stft_mag_out, float* pg_max_host, void* stream);
/* librosa.util.nnls behind mel_to_stft (audiodiffusion/mel.py:165): column blocks whose start point fails L-BFGS-B's rule
 * (max |projected gradient| > pgtol = 1e-5) are solved on the device (projected accelerated gradient, step 1/lipschitz,
 * lipschitz = lambda_max(A A^T) of the fp64 filterbank, at most max_iter iterations per column); blocks that satisfy it are
 * returned as they are, as scipy does (nit = 0). Without this call adm_mel_inverse returns the start point everywhere. */
int adm_mel_set_nnls_solver(adm_mel_t* h, double lipschitz, int max_iter);
/* after an adm_mel_inverse with pg_max_host != NULL (which then holds the projected gradient of the RETURNED point):
 * the start point's max |projected gradient| and the largest iteration count any column needed. */
int adm_mel_last_nnls(adm_mel_t* h, float* pg_start, int* iterations);

#ifdef __cplusplus
}
#endif
#endif /* ADM_H */
