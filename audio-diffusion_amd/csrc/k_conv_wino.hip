// k_conv_wino.hip — Winograd convolution of the fused 3x3 stride-1 layers: filter packing, options, eligibility and the dispatch.
// The kernels are in k_conv_wino_f4.hip (F(4x4,3x3): conv_wino6_kernel) and k_conv_wino_f2.hip (F(2x2,3x3): conv_wino5_kernel,
// conv_wino4_kernel); k_conv_wino.h holds what they share.
#include "k_conv_wino.h"

namespace adm {

// Filter image of conv_wino6_kernel: U = G g G^T (6x6) as [Cin/8][Cout/16][k step 2][point group 9][lane = 16 k4 + l15][4 points] holding
// U[point = 4 pg + e][cout = 16 cblk + l15][cin = 8 chunk + 4 ks + k4]; transposed: the data-gradient filters (as pack_winograd4_body).
__device__ __forceinline__ void pack_winograd6_body(const float* __restrict__ w, float* __restrict__ wu, int Cout, int Cin,
                                                    int transposed, long first, long step) {
  const int PCo = transposed ? Cin : Cout, PCi = transposed ? Cout : Cin;
  const long total = (long)PCo * PCi;
  const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                         {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
  for (long i = first; i < total; i += step) {
    const int l15 = (int)(i & 15);
    long r = i >> 4;
    const int k4 = (int)(r & 3); r >>= 2;
    const int ks = (int)(r & 1); r >>= 1;
    const int cblk = (int)(r % (PCo >> 4));
    const int chunk = (int)(r / (PCo >> 4));
    const int po = cblk * 16 + l15, pi = chunk * 8 + 4 * ks + k4;
    const int co = transposed ? pi : po, c = transposed ? po : pi;
    const float* g = w + ((long)co * Cin + c) * 9;
    float gg[3][3];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) gg[a][b] = transposed ? g[(2 - a) * 3 + (2 - b)] : g[a * 3 + b];
    float t[6][3];
    for (int a = 0; a < 6; ++a)
      for (int b = 0; b < 3; ++b) t[a][b] = G[a][0] * gg[0][b] + G[a][1] * gg[1][b] + G[a][2] * gg[2][b];
    float* blk = wu + ((long)chunk * (PCo >> 4) + cblk) * W6ABLK + (long)ks * 9 * 256 + (k4 * 16 + l15) * 4;
    for (int a = 0; a < 6; ++a)
      for (int b = 0; b < 6; ++b) {
        const int pt = a * 6 + b;
        blk[(pt >> 2) * 256 + (pt & 3)] = t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2];
      }
  }
}

__global__ void pack_winograd6_weight_kernel(const float* __restrict__ w, float* __restrict__ wu, int Cout, int Cin, int transposed) {
  pack_winograd6_body(w, wu, Cout, Cin, transposed, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

// Filter image of conv_wino4_kernel: U = G g G^T as [Cin/8][Cout/16][point group q][k step ks][lane = 16 k4 + li][e]
// holding U[xi = 4 q + e][cout = 16 cblk + li][cin = 8 chunk + 4 ks + k4]. transposed: the data-gradient filters (roles of
// Cout / Cin swapped, taps flipped), as pack_winograd_weight_kernel.
__device__ __forceinline__ void pack_winograd4_body(const float* __restrict__ w, float* __restrict__ wu, int Cout, int Cin,
                                                    int transposed, long first, long step) {
  const int PCo = transposed ? Cin : Cout, PCi = transposed ? Cout : Cin;      // channel counts of the packed convolution
  const long total = (long)PCo * PCi;
  for (long i = first; i < total; i += step) {
    const int li = (int)(i & 15);
    long r = i >> 4;
    const int k4 = (int)(r & 3); r >>= 2;
    const int ks = (int)(r & 1); r >>= 1;
    const int cblk = (int)(r % (PCo >> 4));
    const int chunk = (int)(r / (PCo >> 4));
    const int po = cblk * 16 + li, pi = chunk * 8 + 4 * ks + k4;
    const int co = transposed ? pi : po, c = transposed ? po : pi;              // indices into w (Cout, Cin, 3, 3)
    const float* g = w + ((long)co * Cin + c) * 9;
    float t[4][3];
    for (int j = 0; j < 3; ++j) {
      const float g0 = transposed ? g[2 * 3 + (2 - j)] : g[0 * 3 + j];
      const float g1 = transposed ? g[1 * 3 + (2 - j)] : g[1 * 3 + j];
      const float g2 = transposed ? g[0 * 3 + (2 - j)] : g[2 * 3 + j];
      t[0][j] = g0;
      t[1][j] = 0.5f * (g0 + g1 + g2);
      t[2][j] = 0.5f * (g0 - g1 + g2);
      t[3][j] = g2;
    }
    float* blk = wu + ((long)chunk * (PCo >> 4) + cblk) * W4ABLK + ((long)ks * 64 + k4 * 16 + li) * 4;
    for (int a = 0; a < 4; ++a) {          // a = point group q (row of U), e = column
      f32x4 u;
      u[0] = t[a][0]; u[1] = 0.5f * (t[a][0] + t[a][1] + t[a][2]); u[2] = 0.5f * (t[a][0] - t[a][1] + t[a][2]); u[3] = t[a][2];
      *reinterpret_cast<f32x4*>(blk + a * 512) = u;
    }
  }
}

__global__ void pack_winograd4_weight_kernel(const float* __restrict__ w, float* __restrict__ wu, int Cout, int Cin,
                                             int transposed) {
  pack_winograd4_body(w, wu, Cout, Cin, transposed, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

// (Cout,Cin,3,3) -> U = G g G^T laid out [Cin][16][Cout]; G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]].
// transposed != 0: filters of the DATA-GRADIENT convolution (input channels = Cout, output channels = Cin, taps flipped):
// U' = G flip(g) G^T laid out [Cout][16][Cin].
__device__ __forceinline__ void pack_winograd3_body(const float* __restrict__ w, float* __restrict__ wu, int Cout, int Cin,
                                                    int transposed, long first, long step) {
  const long total = (long)Cout * Cin;
  for (long i = first; i < total; i += step) {
    int co, c;
    if (transposed) { c = (int)(i % Cin); co = (int)(i / Cin); }     // consecutive threads -> consecutive destination words
    else { co = (int)(i % Cout); c = (int)(i / Cout); }
    const float* g = w + ((long)co * Cin + c) * 9;
    float t[4][3];
    for (int j = 0; j < 3; ++j) {
      const float g0 = transposed ? g[2 * 3 + (2 - j)] : g[0 * 3 + j];
      const float g1 = transposed ? g[1 * 3 + (2 - j)] : g[1 * 3 + j];
      const float g2 = transposed ? g[0 * 3 + (2 - j)] : g[2 * 3 + j];
      t[0][j] = g0;
      t[1][j] = 0.5f * (g0 + g1 + g2);
      t[2][j] = 0.5f * (g0 - g1 + g2);
      t[3][j] = g2;
    }
    const long ostride = transposed ? Cin : Cout;
    for (int a = 0; a < 4; ++a) {
      const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]),
                  u3 = t[a][2];
      float* dst = transposed ? wu + ((long)co * 16 + a * 4) * Cin + c : wu + ((long)c * 16 + a * 4) * Cout + co;
      dst[0] = u0; dst[ostride] = u1; dst[2 * ostride] = u2; dst[3 * ostride] = u3;
    }
  }
}

__global__ void pack_winograd_weight_kernel(const float* __restrict__ w, float* __restrict__ wu, int Cout, int Cin,
                                            int transposed) {
  pack_winograd3_body(w, wu, Cout, Cin, transposed, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}
// blockIdx.y = item of a device table; flag bit 0 = transposed (data-gradient filters), bit 1 = the conv_wino4_kernel image
__global__ void __launch_bounds__(256) pack_winograd_batch_kernel(const PackItem* __restrict__ items) {
  const PackItem it = items[blockIdx.y];
  const long first = (long)blockIdx.x * blockDim.x + threadIdx.x, step = (long)gridDim.x * blockDim.x;
  if (it.flag & 2) pack_winograd4_body(it.src, (float*)it.dst, it.Cout, it.Cin, it.flag & 1, first, step);
  else pack_winograd3_body(it.src, (float*)it.dst, it.Cout, it.Cin, it.flag & 1, first, step);
  if (it.flag & 4) pack_winograd6_body(it.src, (float*)it.dst + (long)it.Cout * it.Cin * 16, it.Cout, it.Cin, it.flag & 1, first, step);
}

static int wino_mode();
// Which filter image a convolution with these PACKED channel counts uses — decided by the mode and the channel counts
// alone, so that the packing (done once per layer) and every later launch agree: mode 4 and 64 | couts, 32 | cins -> the
// conv_wino4_kernel image (such a layer then runs on conv_wino4_kernel or, for arguments that kernel cannot take, on the
// direct kernel). The option must be set before the weights are packed.
// (mode 0 — no Winograd kernel runs — packs the mode-4 image as well: an image packed under "conv_wino" = 0 and convolved under the default
// would otherwise be read in the wrong layout and past its 16 floats per filter)
static bool wino4_layout(int couts, int cins) {
  const int m = wino_mode();
  return (m == 4 || m == 0) && couts % W3BM == 0 && cins % (4 * WCK) == 0;
}

// The F(4x4,3x3) image of conv_wino6_kernel FOLLOWS the F(2x2,3x3) image in the same buffer (16 + 36 transformed values per filter) whenever
// the packed channel counts allow the kernel (128 | couts on top of the v4 rule) — whatever the "wino6" option says at packing time, so that
// the option may change between packing and launch.
static bool wino6_layout(int couts, int cins) { return wino4_layout(couts, cins) && couts % W5BM == 0; }
// (the SIZE follows the channel counts alone — not the "conv_wino" mode at the time of the call: a buffer sized under one mode and re-packed
//  under another must hold whichever images that mode writes)
long winograd_packed_floats(int Cout, int Cin, int transposed) {
  const int couts = transposed ? Cin : Cout, cins = transposed ? Cout : Cin;
  const bool room6 = couts % W3BM == 0 && cins % (4 * WCK) == 0 && couts % W5BM == 0;
  return (long)Cout * Cin * (room6 ? 52 : 16);
}
static int pack_winograd(const float* w, float* wu, int Cout, int Cin, int transposed, hipStream_t st) {
  long g = ((long)Cout * Cin + 255) / 256;
  if (g > 4096) g = 4096;
  if (wino6_layout(transposed ? Cin : Cout, transposed ? Cout : Cin))
    ADM_LAUNCH(pack_winograd6_weight_kernel, dim3((unsigned)g), dim3(256), 0, st, w, wu + (long)Cout * Cin * 16, Cout, Cin, transposed);
  if (wino4_layout(transposed ? Cin : Cout, transposed ? Cout : Cin))
    ADM_LAUNCH(pack_winograd4_weight_kernel, dim3((unsigned)g), dim3(256), 0, st, w, wu, Cout, Cin, transposed);
  else
    ADM_LAUNCH(pack_winograd_weight_kernel, dim3((unsigned)g), dim3(256), 0, st, w, wu, Cout, Cin, transposed);
  return ADM_CHECK_LAUNCH();
}
int launch_pack_winograd_weight(const float* w, float* wu, int Cout, int Cin, hipStream_t st) {
  return pack_winograd(w, wu, Cout, Cin, 0, st);
}
// PackItem::flag of a Winograd image: bit 0 = transposed, bit 1 = the conv_wino4_kernel layout — decided exactly as pack_winograd
// decides it, so that the batched re-pack writes the image the kernels read
int winograd_pack_flag(int Cout, int Cin, int transposed) {
  return (transposed ? 1 : 0) | (wino4_layout(transposed ? Cin : Cout, transposed ? Cout : Cin) ? 2 : 0) |
         (wino6_layout(transposed ? Cin : Cout, transposed ? Cout : Cin) ? 4 : 0);
}
int launch_pack_winograd_batch(const PackItem* items_dev, int n, hipStream_t st) {
  if (n <= 0) return 0;
  ADM_LAUNCH(pack_winograd_batch_kernel, dim3(32, (unsigned)n), dim3(256), 0, st, items_dev);
  return ADM_CHECK_LAUNCH();
}
int launch_pack_winograd_weight_T(const float* w, float* wu, int Cout, int Cin, hipStream_t st) {
  return pack_winograd(w, wu, Cout, Cin, 1, st);
}

// "conv_wino": 4 (default) = the Winograd kernels wherever they tile the layer, 0 = the direct MFMA kernel only (k_conv_mfma.hip). Shapes the
// Winograd kernels cannot take run on the direct kernel: the only fallback. ("wino_pair" is accepted and ignored since round 6:
// conv_wino4_kernel keeps one cadence, one workgroup barrier per two chunks.)
void set_winograd_pair(int) {}
// conv_wino5_kernel (128-cout workgroup tiles, all eight waves MFMA + staging): 1 (default) = used wherever the layer has 128 | Cout
// and its 128-cout tiles fill the chip; 0 = conv_wino4_kernel everywhere; bit 1 (2) = also when the tiles do not fill the chip (tests);
// bit 3 (8) = the two-halves-in-antiphase schedule instead of the interleaved one. Bit-identical results in every case (same filter image).
static int g_wino5 = -1;       // -1: take ADM_WINO5 from the environment (default 1) on first use
void set_winograd_v5(int v) { g_wino5 = v; }
static int wino5_on() {
  if (g_wino5 < 0) { const char* e = getenv("ADM_WINO5"); g_wino5 = e ? atoi(e) : 1; }
  return g_wino5;
}
// conv_wino6_kernel (F(4x4,3x3)): 1 (default) = every layer the kernel tiles (128 | Cout, 32 | Cin, 16 | H, 16 | W) on a plane of at least
// 64x64 pixels whose tiles give ONE SAMPLE at least W6_MIN_WGS workgroups — a function of the LAYER only: F(4x4) is not bit-identical to
// F(2x2), so the choice must not depend on the batch (a sample's bits must not depend on the batch it is sampled in); 0 = F(2x2,3x3) kernels
// everywhere; 2 = every layer the kernel tiles (tests); n >= 16 = planes of at least n x n pixels, whatever their workgroup count.
static int g_wino6 = -1;
void set_winograd_v6(int v) { g_wino6 = v; }
static int wino6_on() {
  if (g_wino6 < 0) { const char* e = getenv("ADM_WINO6"); g_wino6 = e ? atoi(e) : 1; }
  return g_wino6;
}
// The rule, measured (profiles/r05_wino.md §5; captured loop, ms per step). One 16x16x128 tile is 2.25x the work of a 64-cout F(2x2)
// workgroup, so a plane whose tiles do not fill the chip pays for it at small batches — and the choice cannot follow the batch:
//                         B = 32 forward   256x256 B = 16   256x256 B = 4   256x256 B = 1   64x64 model B = 1
//   F(2x2) only                73.8             —                —              7.12             3.43
//   planes >= 256              63.5             —                —              6.79             3.42        ("wino6" = 256: the latency setting)
//   planes >= 128              59.7            30.8             11.3            7.29             3.41
//   default (below)            57.4            29.8             11.7            8.30             3.41
//   planes >= 64               57.7             —                —              8.33             4.00
// The default takes the 64x64 level of the 256x256 model (Cout = 256: 32 workgroups per sample) and leaves the 64x64 model's own top level
// (Cout = 128: 16) on F(2x2): the batched configurations gain 3.6 - 3.8 % over the 128 floor, single-sample sampling at 256x256 loses 14 %.
constexpr int W6_MIN_PLANE = 64, W6_MIN_WGS = 32;
static bool wino6_eligible(const adm_conv_args& a) {
  const int rule = a.wino6_rule != 0 ? a.wino6_rule : wino6_on();       // the call's (= its model's) own rule, else the process-wide option
  if (!rule) return false;
  const int C2 = a.x2 ? a.C2 : 0;
  const int Ho = a.up ? 2 * a.H : a.H, Wo = a.up ? 2 * a.W : a.W;
  if (!wino6_layout(a.Cout, a.C1 + C2) || a.C1 % 16 != 0 || Ho % 16 != 0 || Wo % 16 != 0) return false;
  if (rule == 2) return true;
  if (rule >= 16) return Ho >= rule && Wo >= rule;
  return Ho >= W6_MIN_PLANE && Wo >= W6_MIN_PLANE && (Ho / 16) * (Wo / 16) * (a.Cout / W5BM) >= W6_MIN_WGS;
}
// Split K for conv_wino4_kernel — part (a) of the single-sample rules ("single_sample", adm_conv_args.single_sample; off by default). A 64-cout x 8x16-pixel
// tile walks every input channel in one workgroup: at one sample per launch a 16x16 .. 64x64 plane gives the chip 8 .. 128 workgroups, each
// a serial chain of Cin / 8 chunks (40 - 80 us for 0.3 - 0.6 GFLOP; profiles/r06_small_regime.txt). With the rule on, S workgroups share a
// tile, each walks Cin / (8 S) chunks and writes its partial sums to slab s of the split-K scratch; ksplit_finish_kernel adds the slabs in
// order with bias / per-sample term / residual (deterministic). S is a function of the LAYER (and of the rule its model carries) only —
// the partition fixes the fp32 summation order, and a sample's bits must not depend on the batch it is sampled in: a model with the rule on
// pays the slab traffic at every batch size (which is why it is a per-model opt-in: AudioDiffusion, the single-sample front end, selects it).
constexpr int WKS_FILL = 256, WKS_MAX = 8;      // split until one sample's workgroups reach WKS_FILL, at most WKS_MAX parts (measured: 512 / 16 no better, 128 worse)
static int wino_ksplit_parts(const adm_conv_args& a) {
  if (!conv_single_sample(a) || wino6_eligible(a)) return 1;
  const int C2 = a.x2 ? a.C2 : 0;
  const int Ho = a.up ? 2 * a.H : a.H, Wo = a.up ? 2 * a.W : a.W;
  const int wgs1 = (Wo / 16) * (Ho / 8) * (a.Cout / W3BM), nch = (a.C1 + C2) / WCK;
  if (wgs1 >= WKS_FILL) return 1;
  int S = 2;
  while (S < WKS_MAX && S * wgs1 < WKS_FILL) S *= 2;
  while (S > 1 && nch % (4 * S) != 0) S >>= 1;      // the kernel's pipeline stages four chunks per round: every part a multiple of four
  return S;
}
static int g_wino_mode = -1;   // -1: take ADM_CONV_WINO from the environment (default 4) on first use
bool winograd_mode_available(int m) {
  return m == -1 || m == 0 || m == 4;      // (modes 1-3 were the kernel generations retired in round 6)
}
void set_winograd_mode(int m) { g_wino_mode = winograd_mode_available(m) ? m : 4; }
static int wino_mode() {
  if (g_wino_mode < 0) {
    const char* e = getenv("ADM_CONV_WINO");
    g_wino_mode = e ? atoi(e) : 4;
    if (!winograd_mode_available(g_wino_mode) || g_wino_mode < 0) g_wino_mode = 4;
  }
  return g_wino_mode;
}
bool winograd_enabled() { return wino_mode() != 0; }

// Eligibility: 3x3 stride 1 "same", output at least 8x16 with Wo % 16 == 0 and Ho % 8 == 0, Cin % 8, Cout % 32.
static bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
// what the persistent kernels need beyond the shape: float4 row loads of the activations, identity GroupNorm rows
// only without SiLU
static bool wino_persistent_args_ok(const adm_conv_args& a) {
  const int C2 = a.x2 ? a.C2 : 0;
  const long x1_bs = a.x1_bstride ? a.x1_bstride : (long)a.C1 * a.H * a.W, x2_bs = a.x2_bstride ? a.x2_bstride : (long)C2 * a.H * a.W;
  return (a.gn_scale != nullptr || !a.act) && aligned16(a.x1) && (a.x2 == nullptr || aligned16(a.x2)) && x1_bs % 4 == 0 &&
         x2_bs % 4 == 0;
}
bool winograd_eligible(const adm_conv_args& a) {
  if (a.ks != 3 || a.stride != 1 || a.pad_lo != 1 || a.w_bstride != 0 || a.wino_packed == nullptr || a.up > 1) return false;
  const int C2 = a.x2 ? a.C2 : 0;
  const int Hi = a.up ? 2 * a.H : a.H, Wi = a.up ? 2 * a.W : a.W;
  if (!(Wi % 16 == 0 && Hi % 8 == 0 && (a.C1 + C2) % 8 == 0 && a.C1 % 8 == 0 && a.Cout % 32 == 0)) return false;
  if (wino4_layout(a.Cout, a.C1 + C2))     // filters are in the v4 image: conv_wino4_kernel or nothing (-> direct kernel)
    return wino_persistent_args_ok(a) && aligned16(a.out) && (a.residual == nullptr || aligned16(a.residual));
  return false;                            // shapes conv_wino4_kernel cannot tile take the direct MFMA kernel: the only fallback
}

const float* conv_zero_bias(int n);  // k_conv_mfma.hip

// GroupNorm statistic tiles of the output (one per 8 x 16-pixel tile) — only conv_wino4_kernel has the epilogue
int winograd_stats_tiles(const adm_conv_args& a) {
  const int C2 = a.x2 ? a.C2 : 0;
  if (!wino4_layout(a.Cout, a.C1 + C2)) return 0;
  const int Ho = a.up ? 2 * a.H : a.H, Wo = a.up ? 2 * a.W : a.W;
  if (wino6_eligible(a)) return (Wo / 16) * (Ho / 16);          // conv_wino6_kernel: one (sum, sum of squares) per 16x16-pixel tile
  if (wino_ksplit_parts(a) > 1) return (Ho * Wo) % 256 == 0 ? Ho * Wo / 256 : 0;   // split K: the finish pass's 256-pixel strips (or none)
  return (Wo / 16) * (Ho / 8);
}

// Which kernel takes a layer:
//   conv_wino6_kernel  F(4x4): chosen by the LAYER alone (wino6_eligible) — it rounds differently from the F(2x2) kernels, and a sample's bits
//                      must not depend on the batch it is sampled in;
//   conv_wino5_kernel  F(2x2), 128-cout tiles: where those tiles fill the chip;
//   conv_wino4_kernel  F(2x2), 64-cout tiles: Cout = 64 * odd, or fewer 128-cout tiles than CUs (twice as many workgroups) — bit-identical to
//                      conv_wino5_kernel (same filter image, same summation order), so this batch-dependent choice cannot move a sample's bits.
int launch_conv_winograd(const adm_conv_args& a, hipStream_t st) {
  const int C2 = a.x2 ? a.C2 : 0;
  ADM_REQUIRE(wino4_layout(a.Cout, a.C1 + C2), "conv_winograd: shape outside the kernels' tiling (winograd_eligible should have said no)");
  WinoParams p;
  p.tune = 0;
  p.ksplit = 1; p.cps = (a.C1 + C2) / WCK; p.part_stride = 0;
  p.x1 = a.x1; p.x2 = a.x2; p.C1 = a.C1; p.C2 = C2;
  p.N = a.N; p.Hs = a.H; p.Ws = a.W;
  p.Hi = a.up ? 2 * a.H : a.H; p.Wi = a.up ? 2 * a.W : a.W;
  p.Ho = p.Hi; p.Wo = p.Wi; p.up = a.up;
  p.gn_scale = a.gn_scale; p.gn_shift = a.gn_shift; p.act = a.act;
  p.wu = a.wino_packed; p.bias = a.bias ? a.bias : conv_zero_bias(a.Cout); p.Cout = a.Cout;
  ADM_REQUIRE(p.bias != nullptr, "conv_winograd: zero-bias buffer");
  p.chan_add = a.chan_add; p.chan_add_stride = a.chan_add_stride;
  p.residual = a.residual; p.out = a.out;
  p.stats = a.stats_out;
  p.x1_bs = a.x1_bstride ? a.x1_bstride : (long)a.C1 * a.H * a.W;
  p.x2_bs = a.x2_bstride ? a.x2_bstride : (long)C2 * a.H * a.W;
  p.gn_nstride = a.C1 + C2;
  if (p.gn_scale == nullptr) {                                 // no GroupNorm on the load path: identity affine rows
    p.gn_scale = conv_const_ones(a.C1 + C2); p.gn_shift = conv_zero_bias(a.C1 + C2); p.gn_nstride = 0;
    ADM_REQUIRE(p.gn_scale != nullptr && p.gn_shift != nullptr, "conv_winograd: constant buffers");
  }
  if (p.chan_add == nullptr) { p.chan_add = conv_zero_bias(a.Cout); p.chan_add_stride = 0; }
  ADM_REQUIRE(p.chan_add != nullptr, "conv_winograd: zero-bias buffer");
#if !defined(ADM_EMU)
  static int cu_count[16] = {};                                // per device: persistent grids = #CUs
  static std::mutex cu_mu;
  int n_cu;
  {
    std::lock_guard<std::mutex> lk(cu_mu);
    int& c = cu_count[conv_dev_slot() & 15];
    if (c == 0) {
      int dev = 0, n = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
      c = n > 0 ? n : 256;
    }
    n_cu = c;
  }
#else
  const int n_cu = 3;                                          // exercise persistence (several tiles per block) on the emulator
#endif
  if (wino6_eligible(a)) {
    p.tiles_x = p.Wo / 16; p.tiles_y = p.Ho / 16;
    p.n_ct = a.Cout / W5BM;
    p.nblk = p.tiles_x * p.tiles_y * a.N * p.n_ct;
    p.wu = a.wino_packed + (long)a.Cout * (a.C1 + C2) * 16;    // the F(4x4) image follows the F(2x2) image
    set_last_conv_variant(4000 + 316);
    return launch_wino6(p, a.up != 0, a.act != 0, p.nblk < n_cu ? p.nblk : n_cu, st);
  }
  p.tiles_x = p.Wo / 16; p.tiles_y = p.Ho / 8;
  if (const int S = wino_ksplit_parts(a); S > 1) {             // the single-sample rule: by layer and model, before any batch-dependent choice
    const long total = (long)a.N * a.Cout * p.Ho * p.Wo;
    float* scratch = conv_ksplit_scratch((size_t)S * total, st);
    if (scratch == nullptr) return -1;
    WinoParams q = p;
    q.n_ct = a.Cout / W3BM;
    q.ksplit = S; q.cps = (a.C1 + C2) / WCK / S; q.part_stride = total;
    q.nblk = p.tiles_x * p.tiles_y * a.N * q.n_ct * S;
    q.out = scratch; q.stats = nullptr; q.residual = nullptr;
    q.bias = conv_zero_bias(a.Cout); q.chan_add = q.bias; q.chan_add_stride = 0;
    ADM_REQUIRE(q.bias != nullptr, "conv_winograd: zero-bias buffer");
    ADM_TRY(launch_wino4(q, a.up != 0, a.act != 0, q.nblk < n_cu ? q.nblk : n_cu, st));
    set_last_conv_variant(4000 + 317);
    const int HW = p.Ho * p.Wo;
    if (a.stats_out != nullptr)
      return launch_ksplit_finish_stats(scratch, S, total, p.bias, a.chan_add, a.chan_add_stride, a.residual, a.out, a.Cout, HW, a.stats_out, st);
    if (const GnFuse* f = conv_gn_fuse_pending(a.Cout))
      return launch_ksplit_finish_gn(scratch, S, total, p.bias, a.chan_add, a.chan_add_stride, a.residual, a.out, a.N, a.Cout, HW, *f, st);
    return launch_ksplit_finish(scratch, S, total, p.bias, a.chan_add, a.chan_add_stride, a.residual, a.out, a.Cout, HW, st);
  }
  if (wino5_on() && a.Cout % W5BM == 0) {
    const int nblk5 = p.tiles_x * p.tiles_y * a.N * (a.Cout / W5BM);
    if (nblk5 >= n_cu || (wino5_on() & 2)) {                   // ("wino5" bit 1: wherever the shape allows — tests on small tensors)
      p.n_ct = a.Cout / W5BM;
      p.nblk = nblk5;
#if !defined(ADM_EMU)
      static const int tune = [] { const char* e = getenv("ADM_WINO5_TUNE"); return e ? atoi(e) : 1; }();
      p.tune = tune;
#endif
      set_last_conv_variant(4000 + 315);
      return launch_wino5(p, a.up != 0, a.act != 0, nblk5 < n_cu ? nblk5 : n_cu, (wino5_on() & 8) != 0, st);
    }
  }
  p.n_ct = a.Cout / W3BM;
  p.nblk = p.tiles_x * p.tiles_y * a.N * p.n_ct;
  set_last_conv_variant(4000 + 314);
  return launch_wino4(p, a.up != 0, a.act != 0, p.nblk < n_cu ? p.nblk : n_cu, st);
}

}  // namespace adm
