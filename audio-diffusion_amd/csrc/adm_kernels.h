// adm_kernels.h — internal launcher declarations shared by the C-ABI layer and the UNet executor.
#pragma once
#include "../../include/adm.h"
#include "adm_rt.h"

namespace adm {

// k_sched.hip
int launch_sched_step(const float* x, const float* eps, const float* noise, float* out, uint8_t* u8,
                      const adm_sched_coef* table, const int* step_dev, int step, const float* mask,
                      int n_mask_steps, int mask_start, int mask_end, int B, int C, int H, int W, hipStream_t st);
int launch_sched_step_loop(const float* x, const float* eps, const float* noise, long noise_step_stride, float* out,
                           uint8_t* u8, int u8_step, const adm_sched_coef* table, const int* step_dev, int step,
                           const float* mask, int n_mask_steps, int mask_start, int mask_end, int B, int C, int H,
                           int W, hipStream_t st);
int launch_step_advance(int* step_dev, hipStream_t st);
int launch_encode_step(float* x, const float* eps, const adm_sched_coef* table, const int* step_dev, int step, long n,
                       hipStream_t st);
int launch_add_noise(const float* x0, long x0_bstride, const float* noise, const float* sa, const float* sb, int cb,
                     int cn, float* out, int B, int N, long P, hipStream_t st);
int launch_slerp_grid(const float* x0, const float* x1, long n, const double* alphas_dev, int n_alpha, float* out,
                      double* scratch3, hipStream_t st);
int launch_dequant(const float* x, uint8_t* out, long n, hipStream_t st);

// k_groupnorm.hip
int launch_groupnorm_finalize(const double* st1, int C1, int tiles1, const double* st2, int C2, int tiles2, int N, int HW,
                              int groups, float eps, const float* gamma, const float* beta, float* scale, float* shift,
                              hipStream_t st, float* mean_rstd = nullptr);
int conv_stats_tiles(const adm_conv_args& a);        // k_conv_mfma.hip: statistic tiles the dispatched kernel would emit (0: none)
int winograd_stats_tiles(const adm_conv_args& a);    // k_conv_wino.hip
int launch_groupnorm_stats(const float* x1, int C1, const float* x2, int C2, int N, int HW, int groups, float eps,
                           const float* gamma, const float* beta, float* scale, float* shift, hipStream_t st,
                           float* mean_rstd = nullptr);

// k_conv_mfma.hip / k_conv_small.hip
int launch_conv2d(const adm_conv_args& a, hipStream_t st);
// One derived weight image to (re)write: the batched launchers below take a DEVICE array of these (blockIdx.y = item), so that
// Net::refresh_weights after an optimizer step is a handful of launches instead of ~500 tiny ones (2.3 ms of an 85 ms step).
// flag: transposed (data-gradient form) for the conv / Winograd / 16-bit images; element count for PACK_COPY.
struct PackItem { const float* src; void* dst; int Cout, Cin, ks, flag; };
int launch_pack_conv_weight_batch(const PackItem* items_dev, int n, hipStream_t st);     // k_conv_mfma.hip: wp / wpT
int launch_pack_winograd_batch(const PackItem* items_dev, int n, hipStream_t st);        // k_conv_wino.hip: wu / wuT (v3 or v4 image)
int winograd_pack_flag(int Cout, int Cin, int transposed);                                // PackItem::flag of a Winograd image
int launch_pack_bf16_batch(const PackItem* items_dev, int n, hipStream_t st);            // k_conv_bf16.hip: wb / wbT
int launch_copy_batch(const PackItem* items_dev, int n, hipStream_t st);                 // k_conv_mfma.hip: dst[0..Cout) = src[0..Cout)
void conv_ksplit_release(hipStream_t st);                    // give the stream's slab buffer back (before the stream is destroyed)
float* conv_ksplit_scratch(size_t floats, hipStream_t st);   // per-(device, stream) split-K slab buffer; nullptr: take the unsplit path
int launch_ksplit_finish(const float* part, int S, long total, const float* bias, const float* chan_add, int chan_add_stride,
                         const float* residual, float* out, int Cout, int HW, hipStream_t st);   // out = bias + ... + sum of S slabs
int launch_ksplit_finish_stats(const float* part, int S, long total, const float* bias, const float* chan_add, int chan_add_stride,
                               const float* residual, float* out, int Cout, int HW, double* stats, hipStream_t st);   // + (sum, sum of squares) per 256-pixel strip
int launch_pack_conv_weight(const float* w, float* wp, int Cout, int Cin, int ks, hipStream_t st);
int launch_pack_conv_weight_T(const float* w, float* wpT, int Cout, int Cin, int ks, hipStream_t st);
void conv_out_dims(int H, int W, int up, int stride, int ks, int pad_lo, int* Ho, int* Wo);
// kernel variant chosen by the last launch_conv2d on this thread: ks*100 + stride*10 + (bm/32) for the MFMA kernel,
// 1000 + ... for the direct small-channel kernels (profiling only).
int last_conv_variant();
void set_last_conv_variant(int v);
// GroupNorm of a convolution's OUTPUT folded into its split-K finish pass (k_groupnorm.hip: ksplit_finish_gn_kernel). The executor
// announces the GroupNorm that reads the tensor before launch_conv2d (request), the split-K launchers honour it when they can
// (pending -> launch_ksplit_finish_gn) and the executor then skips its statistics launch (taken). Thread-local like the variant.
struct GnFuse { const float* gamma; const float* beta; float eps; int groups; float* scale; float* shift; float* mean_rstd; };
void conv_gn_fuse_request(const GnFuse* f);            // nullptr clears
bool conv_gn_fuse_taken();
void set_gn_fuse_finish(int v);                         // option "gn_fuse_finish" (c_api.hip)
const GnFuse* conv_gn_fuse_pending(int Cout);          // the request if this output (Cout channels) can take it, else nullptr
int launch_ksplit_finish_gn(const float* part, int S, long part_stride, const float* bias, const float* chan_add, int chan_add_stride,
                            const float* residual, float* out, int N, int Cout, int HW, const GnFuse& f, hipStream_t st);

// k_conv_wino.hip
const float* conv_zero_bias(int n);  // shared all-zero device buffer of >= n floats (k_conv_mfma.hip)
const float* conv_const_ones(int n); // shared all-ones device buffer of >= n floats
int conv_dev_slot();                 // current HIP device as an index 0..15 (per-device static state: constant buffers, LDS attributes)
bool winograd_mode_available(int m); // k_conv_wino.hip: 0 and 4 (modes 1-3 were the kernel generations retired in round 6)
int launch_pack_winograd_weight(const float* w, float* wu, int Cout, int Cin, hipStream_t st);
int launch_pack_winograd_weight_T(const float* w, float* wu, int Cout, int Cin, hipStream_t st);  // data-gradient filters
bool winograd_enabled();
void set_wgrad_max_split(int v);  // k_conv_wgrad.hip
void bump_dispatch_epoch();        // net_exec.hip: a process-wide option changed -> training nets re-learn which packings they read
unsigned dispatch_epoch();
void set_blk_direct_dy(int v);     // net_exec.hip: option "blk_direct_dy" (read when a training plan is made)
long winograd_packed_floats(int Cout, int Cin, int transposed);   // size of a wu / wuT buffer: the F(2x2) image (+ the F(4x4) image where conv_wino6_kernel may run)
void set_single_sample(int v);     // "single_sample" (k_conv_mfma.hip): 0 (default) off | 1 the single-sample partition rules (conv_wino4_kernel split K, 16-part 3x3 split on <= 8x8 planes)
bool conv_single_sample(const adm_conv_args& a);   // the call's (= its model's) rule, else the option
bool single_sample_rule(int model_value);          // the same for a bare model value (1 on, -1 off, 0 = the process-wide option)
void set_winograd_v6(int v);    // conv_wino6_kernel (F(4x4,3x3)): 1 (default) planes >= 64x64 with >= 32 workgroups per sample | 0 off | 2 every plane the kernel tiles | n >= 16: planes >= n x n
void set_winograd_v5(int v);    // conv_wino5_kernel (128-cout tiles) where eligible: 1 (default) / 0 = conv_wino4_kernel everywhere (bit-identical)
void set_winograd_pair(int v);  // conv_wino4_kernel: 1 (default) = one workgroup barrier per two chunks, 0 = one per chunk (bit-identical)
void set_winograd_mode(int m);  // 0 off, 1 v1, 2 wave-specialised v2, 3 persistent v3 (default), -1 = default
bool winograd_eligible(const adm_conv_args& a);
int launch_conv_winograd(const adm_conv_args& a, hipStream_t st);

// k_conv_bf16.hip (mixed-precision training: bf16 MFMA operands, fp32 accumulate)
int launch_pack_bf16_weight(const float* w, void* wb, int Cout, int Cin, int transposed, hipStream_t st, int ks = 3);
void set_conv_bf16(int m);   // 0 off (default), 1 = 3x3 convs, 2 = 3x3, 1x1 and stride-2 data-gradient convs, -1 = ADM_CONV_BF16
bool conv_bf16_enabled();
int conv_bf16_mode();
void set_conv_op16_f16(int v);   // operand format of the 16-bit-operand kernels: 0 = bf16 (default), 1 = IEEE binary16
bool conv_op16_f16();
// k_conv1x1_bf16.hip (mode 2)
bool conv1x1_bf16_eligible(const adm_conv_args& a);
int launch_conv1x1_bf16(const adm_conv_args& a, hipStream_t st);
int launch_conv1x1_bf16_split(const adm_conv_args& a, int split_c, float* out2, const float* residual2, hipStream_t st);
bool conv1x1_wgrad_bf16_eligible(const adm_conv_args& a);
int launch_conv1x1_wgrad_bf16(const adm_conv_args& a, const float* dy, float* workspace, int split, hipStream_t st);
bool conv_bf16_eligible(const adm_conv_args& a);
int launch_conv_bf16(const adm_conv_args& a, hipStream_t st);
bool conv_wgrad_bf16_eligible(const adm_conv_args& a);
int launch_conv_wgrad_bf16(const adm_conv_args& a, const float* dy, float* dW, int accumulate, float* workspace, int split,
                           hipStream_t st);

// k_conv_bf16b.hip (round 4, level 3: blocked 16-bit operand images [n][C/8][H+2][W+2] x 16 B, zero halo; LDS-DMA fed kernels)
size_t blk_image_bytes(int N, int C, int H, int W);
bool blk_apply_eligible(int C1, int C2, int H, int W);
// img = round16(act(scale * concat(x1, x2) + shift)) (scale NULL: identity); sum_scratch (blk_sums_scratch floats, NULL ok):
// per-workgroup sums of the INPUT, turned into per-(n, c) / per-c sums by launch_blk_sums_finalize (bias gradients when x1 = dy)
long blk_sums_scratch(int N, int C, int H, int W);
int launch_blk_apply(const float* x1, int C1, long x1_bs, const float* x2, int C2, long x2_bs, int N, int H, int W,
                     const float* scale, const float* shift, int act, void* out, float* sum_scratch, hipStream_t st, int zins = 0);
int launch_blk_gn_bwd_image(const float* x, int C, const float* da, int N, int H, int W, int groups, const float* mean_rstd,
                            const float* gamma, const float* beta, int act, const float* s12, void* out, float* sum_scratch,
                            hipStream_t st);
bool gn_bwd_streaming(int N, int C, int HW);   // k_backward.hip: elementwise passes over >= 64 MB use streaming (non-temporal) accesses
int launch_gn_backward_stats(const float* x1, int C1, const float* x2, int C2, const float* da, int N, int HW, int groups,
                             const float* mean_rstd, const float* gamma, const float* beta, int act, float* s12_scratch,
                             float* dgamma, float* dbeta, hipStream_t st);   // k_backward.hip: pass 1 of launch_gn_backward alone
int launch_blk_sums_finalize(const float* sum_scratch, int N, int C, int H, int W, float* out_nc, int nc_stride, int nc_accumulate,
                             float* out_c, hipStream_t st, float* out_c2 = nullptr);
bool conv_bf16b_eligible(int Cin, int Cout, int H, int W, int N = 0);   // N > 0: also rows of 16 / 8 pixels when N % 2 / 4 == 0
int launch_conv_bf16b(const void* img, int Cin, int N, int H, int W, const void* wb, int Cout, const float* bias,
                      const float* chan_add, int chan_add_stride, const float* residual, float* out, hipStream_t st, int mode = 0,
                      double* stats_out = nullptr);   // stats_out: GroupNorm partial sums of the OUTPUT, [n][cout][(H/8)*(W/32)][2] fp64
int conv_bf16b_stats_tiles(int H, int W);
bool conv_wgradb_eligible(int Ct, int Cout, int H, int W, int N = 0);
long conv_wgradb_workspace(int Ct, int Cout, int N, int H, int W, int* split_out);
int launch_conv_wgradb(const void* xa, int Ct, const void* dyb, int Cout, int N, int H, int W, float* dW, int accumulate,
                       float* workspace, hipStream_t st, int up = 0);
int launch_wgrad_reduce(const float* workspace, int split, long numel, float* dW, int accumulate, int taps, hipStream_t st);

// k_backward.hip / k_conv_wgrad.hip (training)
int launch_sumpool2x2(const float* in, float* out, int H, int W, long planes, int accumulate, hipStream_t st);
int launch_accumulate(float* dst, long dst_bs, const float* src, long src_bs, long per_sample, int N, int accumulate,
                      hipStream_t st);
int launch_chan_sums(const float* dy, int N, int C, int HW, float* out_nc, int nc_stride, int nc_accumulate, float* out_c,
                     hipStream_t st);
int launch_gn_backward(const float* x1, int C1, const float* x2, int C2, const float* da, int N, int HW, int groups,
                       const float* mean_rstd, const float* gamma, const float* beta, int act, float* s12_scratch,
                       float* dgamma, float* dbeta, float* dx1, int acc1, float* dx2, int acc2, hipStream_t st);
int launch_attention_bwd(const float* qkv, const float* dout, float* dqkv, int N, int C, int T, int head_dim,
                         hipStream_t st);
int launch_linear_bwd(const float* dY, int ldy, const float* X, const float* W, int B, int J, int K, int x_silu, float* dW,
                      float* db, float* dX, hipStream_t st);
int launch_conv_small_cin_wgrad(const float* x, int Cin, int N, int H, int W, const float* dy, int Cout, float* dW,
                                hipStream_t st);
int launch_conv_small_cout_bwd(const float* x, int Cin, int N, int H, int W, const float* gn_scale, const float* gn_shift,
                               int act, const float* w, const float* dy, int Cout, float* da, float* dW, hipStream_t st);
long conv_wgrad_workspace(const adm_conv_args& a, int* split_out);
int launch_conv_wgrad(const adm_conv_args& a, const float* dy, float* dW, int accumulate, float* workspace, hipStream_t st);

// k_attention.hip
int launch_attention(const float* qkv, float* out, int N, int C, int T, int head_dim, hipStream_t st, int single_sample = 0);   // single_sample: the model's rule (0 = the option)

// k_transformer.hip (UNet2DConditionModel: Transformer2DModel blocks)
int launch_layernorm_nct(const float* x, const float* gamma, const float* beta, float* y, int N, int C, long T, float eps,
                         hipStream_t st);
int launch_geglu(const float* in, float* out, int N, int C4, long T, hipStream_t st);
int launch_cross_attention(const float* q, const float* ctx, const float* Wk, const float* Wv, float* out, int N, int C,
                           int T, int S, int Dc, int head_dim, hipStream_t st);
int launch_layernorm_nct_bwd(const float* x, const float* dy, const float* gamma, float* dx, int accumulate, float* stats,
                             float* dgamma, float* dbeta, int N, int C, long T, float eps, hipStream_t st);
int launch_geglu_bwd(const float* in, const float* dy, float* din, int N, int C4, long T, hipStream_t st);
int launch_cross_attention_bwd(const float* q, const float* ctx, const float* Wk, const float* Wv, const float* dy, float* dq,
                               float* dWk, float* dWv, int N, int C, int T, int S, int Dc, int head_dim, hipStream_t st);
int launch_attention_bwd_blocked(const float* qkv, const float* dout, float* dqkv, float* stats, int N, int C, int T,
                                 int head_dim, int block, hipStream_t st);
int launch_attention_blocked(const float* qkv, float* out, int N, int C, int T, int head_dim, int key_block, hipStream_t st);
bool attention_mfma_eligible(int C, int T, int head_dim);                // k_transformer.hip: flash self-attention on the f32 MFMAs (head_dim 16 / 32 / 64)
int launch_attention_mfma(const float* qkv, float* out, int N, int C, int T, int head_dim, hipStream_t st);

// k_audio_encoder.hip (AudioEncoder: audiodiffusion/audio_encoder.py:62-84)
int launch_sepconv_block(const float* x, const float* dw, const float* pw, const float* pb, const float* bn_scale,
                         const float* bn_shift, float slope, float* tmp, float* y, int N, int Ci, int Co, int H, int W,
                         hipStream_t st);
int launch_dense_act(const float* x, const float* W, const float* b, const float* post_scale, const float* post_shift,
                     float slope, int leaky, float* y, int N, int K, int J, int hwc_C, hipStream_t st);

// k_vae.hip
int launch_softmax_channels(float* s, int N, int J, int T, float scale, hipStream_t st);
int launch_transpose_ct(const float* in, long in_bs, float* out, int N, int C, int T, hipStream_t st);
int launch_gaussian_sample(const float* moments, const float* noise, float* out, int N, int Cz, long HW, float out_scale,
                           hipStream_t st);
int launch_scale(const float* x, float* out, float s, long n, hipStream_t st);

// k_temb.hip
// emb[b][:] = linear_2(silu(linear_1(sinusoid(t_b)))); t from host-provided device array or coef table.
int launch_time_embedding(const float* t_dev, int t_stride, const adm_sched_coef* table, const int* step_dev,
                          const float* freqs, int half_dim, int flip, const float* w1, const float* b1, const float* w2,
                          const float* b2, int dim_in, int dim_emb, float* emb, int B, hipStream_t st,
                          float* save_sinus = nullptr, float* save_z = nullptr, float* emb_act = nullptr);
// out[b][r] = bias[r] + sum_k W[r][k] * silu(emb[b][k])   for all resnets' time_emb_proj rows at once.
// emb_is_activated: `emb` already holds silu(emb) (launch_time_embedding's emb_act output)
int launch_temb_proj(const float* emb, const float* w, const float* bias, float* out, int B, int K, int R,
                     hipStream_t st, int emb_is_activated = 0);

}  // namespace adm
