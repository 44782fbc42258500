// net_exec.h — generic flat-op-list executor shared by the UNet2DModel and AutoencoderKL executors.
// A `Net` is built once from a diffusers config: tensors (NCHW fp32, batch dimension added at plan time), per-(n,c)
// GroupNorm scale/shift buffers, and ops (GroupNorm statistics, fused MFMA convolution, attention cores, and the two
// helper ops of the single-head VAE attention). Activations are planned into an arena by liveness.
#pragma once
#include <deque>
#include <map>
#include <string>
#include <vector>

#include "adm_kernels.h"

namespace adm {

struct ParamSlot {
  std::vector<long> shape;
  size_t numel = 0;
  float* dev = nullptr;
  bool set = false;
  bool external = false;  // bound into a caller-owned flat buffer (training): never freed here
};

struct ParamStore {
  std::map<std::string, ParamSlot> params;
  void declare(const std::string& key, std::vector<long> shape);
  void declare_conv(const std::string& p, int co, int ci, int ks);
  void declare_lin(const std::string& p, int co, int ci);
  void declare_gn(const std::string& p, int c);
  void declare_resnet(const std::string& p, int ci, int co, int temb);  // temb <= 0: no time_emb_proj
  void declare_attn(const std::string& p, int c);
  void declare_transformer(const std::string& p, int c, int cross_dim);   // Transformer2DModel with one BasicTransformerBlock
  int set(const char* key, const float* host_data, size_t numel);       // accepts deprecated attention names
  int bind(const char* key, float* dev_ptr);                            // training: parameter lives in a flat buffer
  int missing(std::string* names) const;
  float* P(const std::string& k) const { return params.at(k).dev; }
  void free_all();
};

struct ConvW {
  float* wp = nullptr;    // forward packing [Cin][tap][Cout]
  float* wpT = nullptr;   // backward-data packing [Cout][tap][Cin] (training only)
  float* wu = nullptr;    // Winograd-domain weights [Cin][16][Cout] (3x3 convs, when a Winograd mode is selected)
  float* wuT = nullptr;   // ... of the data-gradient convolution [Cout][16][Cin] (training only)
  void* wb = nullptr;     // bf16 MFMA operands [tap][Cin/8][Cout][8] (training nets under option conv_bf16)
  void* wbT = nullptr;    // ... of the data-gradient convolution
  float* bias = nullptr;  // master bias (for q|k|v: the stacked copy)
  int Cin = 0, Cout = 0, ks = 3;
  std::string key;        // diffusers prefix of the master parameter ("" for derived weights)
  std::string qkv_prefix; // non-empty: q|k|v stacked from <prefix>.to_q/.to_k/.to_v
  bool has_bias = true;   // false: Linear(bias=False) (Transformer2DModel's to_q/to_k/to_v); qkv: the stacked bias stays zero
  float* stacked = nullptr;  // (3C, C) stacked master copy of q|k|v
  // training: which derived packings the convolution dispatch has actually read since the last plan (bit 0 wp, 1 wpT,
  // 2 wu, 3 wuT, 4 wb, 5 wbT). Once a whole step has run (Net::use_known), refresh_weights re-packs only those.
  mutable unsigned used = 0;
};
enum { PK_WP = 1, PK_WPT = 2, PK_WU = 4, PK_WUT = 8, PK_WB = 16, PK_WBT = 32 };
struct GNW {
  float* gamma = nullptr;
  float* beta = nullptr;
  int C = 0;
};
struct Tensor {
  int C = 0, H = 0, W = 0;
  float* ptr = nullptr;
  int last_use = -1;
  bool external = false;
  float* grad = nullptr;  // training: gradient buffer (same shape), ginit = already holds a contribution
  bool ginit = false;
  // inference: GroupNorm partial sums written by the producing convolution's epilogue ([n][c][tile][2] fp64), when some
  // GroupNorm reads this tensor and the producing kernel can emit them (stat_tiles > 0)
  bool want_stats = false;
  double* stats = nullptr;
  int stat_tiles = 0;
};
struct GnBuf {
  int C = 0;
  float* scale = nullptr;
  float* shift = nullptr;
  float* mean_rstd = nullptr;  // training: (B, groups, 2)
  const GNW* g = nullptr;      // the affine parameters this buffer was computed with
};
struct Op {
  enum Kind { GN, CONV, ATTN, SOFTMAXC, TRANSP, LN, GEGLU, XATTN } kind = CONV;
  int in1 = -1, in2 = -1, out = -1, res = -1, gn = -1;
  int in1_coff = 0, in1_C = 0;      // channel-slice view of in1 (in1_C == 0: whole tensor)
  int wt = -1, wt_coff = 0;         // CONV with per-sample weights taken from tensor `wt` (channel offset wt_coff)
  int dyn_cout = 0;                 // ... producing this many output channels
  int up = 0, stride = 1, ks = 3, pad_lo = 1, act = 0;
  const ConvW* w = nullptr;
  const GNW* g = nullptr;
  int temb_off = -1;
  int head_dim = 0;
  float scale = 1.f;
  float eps = 0.f;                  // GN / LN: epsilon of this op (0: the net's norm_eps)
  const float* wk = nullptr;        // XATTN: to_k / to_v master weights (C, cross_dim), no bias
  const float* wv = nullptr;
};

struct OpTimer;  // unet_exec.hip (profiling aid)

struct Net {
  ParamStore* ps = nullptr;
  int groups = 32;
  float eps = 1e-5f;
  int head_dim_cfg = 8;
  std::deque<ConvW> convs;
  std::deque<GNW> gns;
  std::vector<Tensor> tensors;
  std::vector<GnBuf> gnbufs;
  std::vector<Op> ops;
  std::vector<void*> owned;
  std::vector<std::pair<std::string, int>> temb_rows;  // (time_emb_proj prefix, Cout) in op order (UNet only)
  int t_in = -1, t_out = -1;
  int planned_B = 0;
  int single_sample = 0;             // this model's single-sample partition rules (adm_unet_set_option; 0 = the process-wide option, 1 on, -1 off)
  int wino6_rule = 0;                // this model's F(4x4) layer rule (adm_unet_set_option; 0 = the process-wide option): copied into every adm_conv_args
  unsigned plan_epoch = 0;           // dispatch_epoch() the plan was made under: adm_set_option moves kernel choices, and with them the statistic-tile counts
  bool plan_current(int B) const;    // planned for exactly B under the current options
  // conditional UNet: encoder_hidden_states of the current call, device (B, ctx_S, ctx_D) (set by the owner before run)
  const float* ctx = nullptr;
  int ctx_S = 0, ctx_D = 0;
  std::vector<void*> arena;
  size_t arena_bytes = 0;
  // training
  bool training = false;             // keep every activation, allocate gradient buffers, maintain wpT
  bool use_known = false;            // a forward + backward pass has run on the current plan: ConvW::used is complete
  bool stale_packings = false;       // a refresh has skipped packings (see begin_inference)
  std::vector<int> learned_B;        // batch sizes that have completed a forward + backward on this plan (masks cover them)
  // a training pass at batch B <= planned_B is about to run inside the current plan (the partial last batch of an epoch does
  // not re-plan): a batch size the masks have not seen yet may dispatch kernels whose packings a refresh skipped -> bring
  // every packing up to date first, once per new size
  int begin_training_batch(int B, hipStream_t st);
  // Batched re-pack (training, masks learned): device tables of PackItems, one per kernel family, rebuilt whenever a mask changes
  std::vector<PackItem> pk_host[4];  // 0 copies (q|k|v stacks), 1 wp / wpT, 2 Winograd images, 3 16-bit images
  PackItem* pk_dev = nullptr;
  size_t pk_dev_cap = 0;
  bool pk_valid = false;
  int build_pack_tables(hipStream_t st);
  unsigned known_epoch = 0;          // dispatch_epoch() when use_known was set: adm_set_option invalidates the masks
  // a pass read packing bit `pk` of `w`: record it; if a refresh has skipped that packing (the dispatch changed under a
  // learned mask: option / environment / pointer alignment), the launch just made read weights from before the optimizer
  // steps — fail instead of training on them
  int note_packing(const ConvW& w, unsigned pk);
  int begin_inference(hipStream_t st, std::vector<unsigned>* saved);
  void end_inference(const std::vector<unsigned>& saved);
  const float* params_base = nullptr;  // flat master parameter buffer and the matching flat gradient buffer
  float* grads_base = nullptr;
  // level 3 of option conv_bf16 (training nets; k_conv_bf16b.hip): per 3x3 stride-1 convolution the blocked 16-bit image of
  // its activated input (written once by the forward pass, read by the forward and the weight-gradient kernel) and the
  // blocked image of its output gradient (one buffer per distinct (Cout, H, W): the zero halo belongs to the geometry)
  // s2: Downsample2D.conv; img_for: this op's GroupNorm backward writes the dy image of op `img_for` directly (its input is that
  // convolution's output and nothing else reads it), -1: none; img_done: set by that pass for the producer during a reverse walk
  struct BlkOp { void* xa = nullptr; void* dyb = nullptr; bool fwd = false, wg = false, dg = false, s2 = false; int img_for = -1; bool img_done = false; };
  std::vector<BlkOp> blk;            // indexed like ops; empty below level 3 / for inference
  float* blk_part = nullptr;         // per-workgroup channel sums of the dy image pass (bias gradients)
  std::vector<int> reader_count;     // tensor -> number of (non-statistics) ops that read it, level-3 training plans
  std::vector<int> producer_of;      // tensor -> index of the op that writes it (-1: none), level-3 training plans
  std::vector<int> gn_fuse_of;       // per op: the GroupNorm op whose statistics this convolution's split-K finish may leave (-1: none)
  std::vector<char> gn_skip;         // per op, during a forward walk: this GroupNorm's scale / shift have been written already
  // Side-stream overlap (inference, round 6): a convolution whose inputs are ready several ops before its place in the list — the resnets'
  // 1x1 conv_shortcut, which reads the block input — is launched on a second stream at hoist_from[j] and joined at its own position, so that
  // it runs beside norm1 / conv1 / norm2 instead of between them. Scheduling only (no arithmetic moves): taken where the launches do not
  // fill the chip (small planes / batches: the latency regimes), planned per batch size, checked against the arena's buffer reuse (plan()).
  std::vector<int> hoist_from;       // per op j: the op index in front of which it is launched on the side stream, -1 = in place
  std::vector<std::vector<int>> hoist_at;   // per op i: the ops launched on the side stream in front of it
  hipStream_t side = nullptr;
  void* ev_fork = nullptr;           // hipEvent_t (opaque here: the emulator has no events)
  void* ev_join = nullptr;
  int plan_side_overlap(int B);
  int launch_side_conv(const Op& o, int B, const float* temb_all, int temb_stride, hipStream_t st);
  std::vector<char> bias_done;       // per op, during a reverse walk: its bias gradient came with another convolution's channel sums
  float *tmp_da = nullptr, *wgrad_ws = nullptr, *s12 = nullptr, *tmp_w = nullptr;
  size_t tmp_da_floats = 0, wgrad_ws_floats = 0, tmp_w_floats = 0;

  // ---- construction -----------------------------------------------------------------------------------
  int dalloc(void** p, size_t bytes);
  int make_conv(const std::string& p, int co, int ci, int ks, const ConvW** out, bool bias = true);
  const GNW* make_gn(const std::string& p, int c);
  int new_tensor(int C, int H, int W, bool ext = false);
  int gn_op(int in1, int in2, const GNW* g);
  int conv_op(int in1, int in2, const ConvW* w, int gn, int act, int up, int stride, int pad_lo, int res, int temb_off,
              int out_ext = -1);
  int resnet(const std::string& p, int x1, int x2, int ci, int co, bool temb, int* rc);
  int attention(const std::string& p, int x, int C, int head_dim, int* rc);  // head_dim <= 64: fused small-head kernel
  // diffusers Transformer2DModel (GroupNorm eps 1e-6, 1x1 proj_in/out, one BasicTransformerBlock: self-attention,
  // cross-attention on `ctx`, GEGLU feed-forward; `heads` heads of C / heads channels)
  int transformer(const std::string& p, int x, int C, int heads, int cross_dim, int* rc);
  int stacked_qkv(const std::string& prefix, int C, bool bias, const ConvW** out);
  void finish_liveness();
  void fill_conv_args(const Op& o, int B, const float* temb_all, int temb_stride, adm_conv_args* a) const;

  // ---- execution ----------------------------------------------------------------------------------------
  int arena_alloc(void** p, size_t bytes);
  void free_plan();
  int plan(int B);
  int run(const float* x, float* out, int B, const float* temb_all, int temb_stride, hipStream_t st, OpTimer* tm);
  // training: re-pack every derived weight from the (updated) master parameters
  int refresh_weights(hipStream_t st);
  // training: reverse pass. grad of t_out must be in tensors[t_out].grad (set by the caller); writes parameter
  // gradients into grads_base (+ offset of the master parameter) and dtemb_all (B, temb_stride) when non-NULL.
  int run_backward(int B, float* dtemb_all, int temb_stride, hipStream_t st);
  float* grad_of(const float* master_param) const { return grads_base + (master_param - params_base); }
  // training, data-parallel overlap: the flat gradient buffer is cut into buckets [bk_lo[b], bk_lo[b+1]); as soon as the
  // reverse pass has ENQUEUED the last kernel that writes into a bucket, bk_fn(bk_user, b) is called on the host thread so
  // the caller can queue that bucket's all-reduce behind it while the rest of the backward pass is still running
  std::vector<long> bk_lo;
  std::vector<int> bk_total, bk_pending;
  void (*bk_fn)(void*, int) = nullptr;
  void* bk_user = nullptr;
  int set_bucket_hook(int n_buckets, const long* bounds, void (*fn)(void*, int), void* user);
  void bucket_reset() { bk_pending = bk_total; }
  void mark_ready(const float* master_param, size_t numel);
  void destroy();
};

struct OpTimer {  // optional per-op HIP-event timing; disabled (recs == nullptr) on the product path
  std::vector<adm_op_profile>* recs = nullptr;
#if !defined(ADM_EMU)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;
#endif
  hipStream_t st = nullptr;
  void begin();
  void end(int kind, int variant, double flops, double bytes);
  void finish();
};

}  // namespace adm
