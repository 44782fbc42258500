// k_conv_bf16b.hip — `--mixed_precision bf16` training convolutions on BLOCKED 16-bit operand images (round 4; level 3 of
// option "conv_bf16").  scripts/train_unet.py:391-401 hands the choice to accelerate -> torch.autocast: Conv2d runs on 16-bit
// operands with fp32 accumulation.  What round 2's phase accounting asked for (profiles/r02_bf16_training.md):
//
//   * the ACTIVATED convolution input (GroupNorm affine + SiLU + rounding, virtual concat resolved) is written ONCE per layer
//     by a streaming pass (blk_apply_kernel) as  img[n][C/8][H+2][W+2] x 16 B  — one 16-byte unit = 8 consecutive channels of
//     one pixel, a zero halo of one pixel all around (written once at allocation, never touched again) — and read by the
//     forward kernel AND, kept like every training activation, by the weight-gradient kernel; the output gradient dy gets the
//     same image once per layer and feeds the data-gradient and the weight-gradient kernel;
//   * every operand byte reaches the MFMAs by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB per instruction, whole
//     cache lines, no VGPRs, no conversion or staging VALU, no bounds logic — the halo is in memory), issued as inline asm so
//     that the kernels count the vector-memory queue themselves (ADM_WAIT_VMEM) and nothing drains the prefetch;
//   * forward / data gradient (conv_bf16b_kernel): 128 couts x 8x32 pixels per workgroup, 4 waves x (32 couts x 256 pixels),
//     K in chunks of 16 channels; the 10x34-pixel patch of a chunk is shared through LDS (three buffers, requested two chunks
//     ahead), a wave's filter fragment of one tap is ONE 1-KiB DMA into its private nine-slot ring (requested a whole chunk
//     ahead); 70.5 KiB of LDS and <= 256 registers: TWO workgroups per CU, so one's prologue / epilogue runs under the
//     other's MFMAs;
//   * weight gradient (conv_wgradb_kernel): k = pixels; the channel-blocked units are transposed on the way out of LDS by
//     ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group), so the nine taps are nine base addresses into one
//     haloed patch — no funnel shifts; 8 waves = 128 couts x 64 cins x 9 taps per workgroup, 4x32-pixel tiles double-buffered.
//
// Operand bits are the same as in k_conv_bf16.hip (same affine / SiLU / rounding expressions), accumulation order differs.
#include <type_traits>
#include <vector>

#include "adm_kernels.h"

namespace adm {


__device__ __forceinline__ float silu_bb(float v) { return v * ADM_RCP(1.0f + __expf(-v)); }

// ---------------------------------------------------------------------------------------------- blocked image writer
struct BlkApplyParams {
  const float* x1; const float* x2; int C1, C2;
  long x1_bs, x2_bs;                 // batch strides (floats)
  int N, H, W;
  const float* scale; const float* shift; int nstride;   // per-(n, channel) affine rows, NULL = identity
  int act;
  u32x4* out; int Hp, Wp;
  float* part;                       // optional: per-(n, c, workgroup) sums of the INPUT ([n][c][gridDim.x], blk_sums_finalize_kernel)
  // GNBWD kernels: the "input" of the pass is computed, not loaded — dx of GroupNorm (+ SiLU) backward, pass 2 (k_backward.hip
  // gn_bwd_apply_kernel): x1 = the GroupNorm input, gda = dL/d(activated tensor), per-(n, group) mean / rstd and s1 / s2
  const float* gda; const float* g_mean_rstd; const float* g_gamma; const float* g_beta; const float* g_s12; int g_groups, g_act;
  int zins;                          // 1 / 2: the image has (2H, 2W) pixels and source pixel (y, x) lands on pixel (2y + 1, 2x + 1) / (2y, 2x), the
                                     // rest stays zero (the zero-inserted dy of a stride-2 convolution with pad (0,1,0,1) / pad 1, see below)
};

// One thread = 8 channels x 4 consecutive pixels: eight float4 row loads (a wave reads 1 KiB of ONE channel row per
// instruction), affine + SiLU + rounding, four 16-byte units stored back to back (a wave writes 4 KiB contiguous).
// GNBWD: the image of dy for a convolution whose OUTPUT is read by nothing but a GroupNorm (+ SiLU) + convolution (conv1 of a resnet):
// its dy IS the dx of that GroupNorm's backward, so pass 2 of the backward writes the 16-bit image (and the channel sums) directly —
// the fp32 dx tensor and the image pass over it disappear.  Same arithmetic as gn_bwd_apply_kernel: rstd * (g * gamma - s1 - xhat * s2).
// NT: streaming loads of the fp32 tensors and streaming stores of the image (passes over >= 64 MB; adm_ld_nt)
template <bool F16, bool GNBWD = false, bool NT = false>
__global__ void __launch_bounds__(256) blk_apply_kernel(const BlkApplyParams p) {
  const int cg = blockIdx.y, n = blockIdx.z;
  const int W4 = p.W >> 2;
  const int q = blockIdx.x * 256 + threadIdx.x;
  const int c0 = cg * 8;
  const long HW = (long)p.H * p.W;
  const bool live = q < p.H * W4;
  const int y = live ? q / W4 : 0, x4 = live ? q - y * W4 : 0;
  const float* src = c0 < p.C1 ? p.x1 + (long)n * p.x1_bs + (long)c0 * HW : p.x2 + (long)n * p.x2_bs + (long)(c0 - p.C1) * HW;
  float4 v[8];
  ADM_UNROLL
  for (int e = 0; e < 8; ++e) {
    const float4* q4 = reinterpret_cast<const float4*>(src + (long)e * HW + (long)y * p.W + 4 * x4);
    v[e] = !live ? make_float4(0.f, 0.f, 0.f, 0.f) : (NT ? adm_ld_nt(q4) : *q4);
  }
  if (GNBWD) {
    const float* dsrc = p.gda + ((long)n * p.C1 + c0) * HW;
    const int cpg = p.C1 / p.g_groups;
    ADM_UNROLL
    for (int e = 0; e < 8; ++e) {
      const float4* q4 = reinterpret_cast<const float4*>(dsrc + (long)e * HW + (long)y * p.W + 4 * x4);
      const float4 dv = !live ? make_float4(0.f, 0.f, 0.f, 0.f) : (NT ? adm_ld_nt(q4) : *q4);
      const int g = (c0 + e) / cpg;
      const float mean = p.g_mean_rstd[((long)n * p.g_groups + g) * 2], rstd = p.g_mean_rstd[((long)n * p.g_groups + g) * 2 + 1];
      const float s1 = p.g_s12[((long)n * p.g_groups + g) * 2], s2 = p.g_s12[((long)n * p.g_groups + g) * 2 + 1];
      const float gm = p.g_gamma[c0 + e], bt = p.g_beta[c0 + e];
      const float xe[4] = {v[e].x, v[e].y, v[e].z, v[e].w}, de[4] = {dv.x, dv.y, dv.z, dv.w};
      float r[4];
      ADM_UNROLL
      for (int k = 0; k < 4; ++k) {
        const float xh = (xe[k] - mean) * rstd;
        float gy = de[k];
        if (p.g_act) { const float yv = xh * gm + bt, sg = ADM_RCP(1.0f + __expf(-yv)); gy *= sg * (1.0f + yv * (1.0f - sg)); }
        r[k] = live ? rstd * (gy * gm - s1 - xh * s2) : 0.f;
      }
      v[e] = make_float4(r[0], r[1], r[2], r[3]);
    }
  }
  if (p.part != nullptr) {
    // bias gradient of the producing layer folded into the pass that reads dy anyway: per-thread sums of the four pixels, a
    // fixed 64-lane shuffle tree, the four waves in order — one partial per (n, channel, workgroup); blk_sums_finalize_kernel
    // adds the partials in order (fp64), so the sums are as reproducible as adm_chan_sums'
    __shared__ float red[8][4];
    ADM_UNROLL
    for (int e = 0; e < 8; ++e) {
      float sm = (v[e].x + v[e].y) + (v[e].z + v[e].w);
      ADM_UNROLL
      for (int m = 32; m >= 1; m >>= 1) sm += __shfl_xor(sm, m);
      if ((threadIdx.x & 63) == 0) red[e][threadIdx.x >> 6] = sm;
    }
    __syncthreads();
    if (threadIdx.x < 8)
      p.part[((long)n * (p.C1 + p.C2) + c0 + threadIdx.x) * gridDim.x + blockIdx.x] =
          (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
  }
  float o[8][4];
  ADM_UNROLL
  for (int e = 0; e < 8; ++e) {
    float sc = 1.f, sh = 0.f;
    if (p.scale) { sc = p.scale[(long)n * p.nstride + c0 + e]; sh = p.shift[(long)n * p.nstride + c0 + e]; }
    const float in[4] = {v[e].x, v[e].y, v[e].z, v[e].w};
    ADM_UNROLL
    for (int j = 0; j < 4; ++j) {
      float t = in[j] * sc + sh;
      if (p.act) t = silu_bb(t);
      o[e][j] = t;
    }
  }
  // A lane owns 4 consecutive units (64 B): stored straight from here, one instruction would touch 64 separate 16-byte pieces
  // of 32 cache lines.  The wave's 256 units go through 4 KiB of LDS instead and leave as four runs of 64 consecutive units
  // (1 KiB per store instruction where the image row is long enough).  LDS position of unit 4 l + j: 4 l + (j ^ ((l >> 1) & 3)) —
  // conflict-free for the 8-lane groups of ds_write_b128 and the 16-lane groups of ds_read_b128.
  if (p.zins) {      // scattered units (every other pixel of every other row): stored straight from the lane
    if (!live) return;
    const int zo = p.zins == 1 ? 2 : 1;                  // haloed coordinate of source pixel 0: odd pixels (pad 0) / even pixels (pad 1)
    u32x4* const zrow = p.out + (((long)n * ((p.C1 + p.C2) >> 3) + cg) * p.Hp + (2 * y + zo)) * p.Wp + (8 * x4 + zo);
    ADM_UNROLL
    for (int j = 0; j < 4; ++j) {
      u32x4 w;
      w[0] = ADM_PK16(F16, o[0][j], o[1][j]); w[1] = ADM_PK16(F16, o[2][j], o[3][j]);
      w[2] = ADM_PK16(F16, o[4][j], o[5][j]); w[3] = ADM_PK16(F16, o[6][j], o[7][j]);
      zrow[2 * j] = w;
    }
    return;
  }
  __shared__ u32x4 stg[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  ADM_UNROLL
  for (int j = 0; j < 4; ++j) {
    u32x4 w;
    w[0] = ADM_PK16(F16, o[0][j], o[1][j]); w[1] = ADM_PK16(F16, o[2][j], o[3][j]);
    w[2] = ADM_PK16(F16, o[4][j], o[5][j]); w[3] = ADM_PK16(F16, o[6][j], o[7][j]);
    stg[wave][4 * lane + (j ^ ((lane >> 1) & 3))] = w;
  }
  // destination (in units, relative to this (n, channel group) plane) of the lane's first unit; -1: thread past the end
  const int my_dst = live ? (y + 1) * p.Wp + (4 * x4 + 1) : -1;
  ADM_WAVE_LDS_ORDER();
  u32x4* const plane = p.out + ((long)n * ((p.C1 + p.C2) >> 3) + cg) * p.Hp * p.Wp;
  ADM_UNROLL
  for (int j = 0; j < 4; ++j) {
    const int owner = 16 * j + (lane >> 2);                       // lane that produced unit 64 j + lane
    const int od = __shfl(my_dst, owner);
    const u32x4 w = stg[wave][64 * j + 4 * (lane >> 2) + ((lane & 3) ^ ((lane >> 3) & 3))];
    if (od >= 0) { if (NT) adm_st_nt(plane + od + (lane & 3), w); else plane[od + (lane & 3)] = w; }
  }
}

size_t blk_image_bytes(int N, int C, int H, int W) { return (size_t)16 * N * (C / 8) * (H + 2) * (W + 2); }

bool blk_apply_eligible(int C1, int C2, int H, int W) { return C1 % 8 == 0 && C2 % 8 == 0 && W % 4 == 0 && H > 0; }

// out_nc[n * nc_stride + c] (= | +=) sum over the workgroup partials (NULL ok); out_c[c] += the same over n (atomic; NULL ok)
__global__ void __launch_bounds__(256) blk_sums_finalize_kernel(const float* __restrict__ part, int C, int nbx, float* out_nc,
                                                               int nc_stride, int nc_accumulate, float* out_c, float* out_c2) {
  const int c = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (c >= C) return;
  const float* q = part + ((long)n * C + c) * nbx;
  double sm = 0.0;
  for (int i = 0; i < nbx; ++i) sm += (double)q[i];
  if (out_nc) {
    float* o = out_nc + (long)n * nc_stride + c;
    *o = nc_accumulate ? *o + (float)sm : (float)sm;
  }
  if (out_c) atomicAdd(out_c + c, (float)sm);
  if (out_c2) atomicAdd(out_c2 + c, (float)sm);      // a second bias that sees the same dy (the shortcut convolution of a resnet)
}

long blk_sums_scratch(int N, int C, int H, int W) { return (long)N * C * ceil_div(H * (W / 4), 256); }

int launch_blk_apply(const float* x1, int C1, long x1_bs, const float* x2, int C2, long x2_bs, int N, int H, int W,
                     const float* scale, const float* shift, int act, void* out, float* sum_scratch, hipStream_t st, int zins) {
  ADM_REQUIRE(blk_apply_eligible(C1, x2 ? C2 : 0, H, W), "blk_apply: channel counts must be multiples of 8 and W of 4");
  BlkApplyParams p;
  p.x1 = x1; p.x2 = x2; p.C1 = C1; p.C2 = x2 ? C2 : 0;
  p.x1_bs = x1_bs ? x1_bs : (long)C1 * H * W; p.x2_bs = x2_bs ? x2_bs : (long)p.C2 * H * W;
  p.N = N; p.H = H; p.W = W; p.scale = scale; p.shift = shift; p.nstride = C1 + p.C2; p.act = act;
  p.out = reinterpret_cast<u32x4*>(out); p.Hp = (zins ? 2 * H : H) + 2; p.Wp = (zins ? 2 * W : W) + 2;
  p.part = sum_scratch; p.zins = zins;
  p.gda = nullptr; p.g_mean_rstd = p.g_gamma = p.g_beta = p.g_s12 = nullptr; p.g_groups = 1; p.g_act = 0;
  ADM_REQUIRE((scale != nullptr) == (shift != nullptr), "blk_apply: scale and shift come together");
  ADM_REQUIRE((reinterpret_cast<uintptr_t>(x1) & 15) == 0 && (x2 == nullptr || (reinterpret_cast<uintptr_t>(x2) & 15) == 0),
              "blk_apply: inputs must be 16-byte aligned");
  const dim3 grid((unsigned)ceil_div(H * (W / 4), 256), (unsigned)((C1 + p.C2) / 8), (unsigned)N);
  const bool f16 = conv_op16_f16();
  if (gn_bwd_streaming(N, C1 + p.C2, H * W) && !zins) {
    if (f16) ADM_LAUNCH((blk_apply_kernel<true, false, true>), grid, dim3(256), 0, st, p);
    else ADM_LAUNCH((blk_apply_kernel<false, false, true>), grid, dim3(256), 0, st, p);
  } else {
    if (f16) ADM_LAUNCH((blk_apply_kernel<true, false>), grid, dim3(256), 0, st, p);
    else ADM_LAUNCH((blk_apply_kernel<false, false>), grid, dim3(256), 0, st, p);
  }
  return ADM_CHECK_LAUNCH();
}

// pass 2 of GroupNorm (+ SiLU) backward writing the blocked 16-bit image of dx (and its channel sums) instead of the fp32 tensor
int launch_blk_gn_bwd_image(const float* x, int C, const float* da, int N, int H, int W, int groups, const float* mean_rstd,
                            const float* gamma, const float* beta, int act, const float* s12, void* out, float* sum_scratch,
                            hipStream_t st) {
  ADM_REQUIRE(blk_apply_eligible(C, 0, H, W) && C % groups == 0, "blk_gn_bwd_image: C % 8, W % 4, C % groups");
  ADM_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(da)) & 15) == 0, "blk_gn_bwd_image: inputs must be 16-byte aligned");
  BlkApplyParams p;
  p.x1 = x; p.x2 = nullptr; p.C1 = C; p.C2 = 0; p.x1_bs = (long)C * H * W; p.x2_bs = 0;
  p.N = N; p.H = H; p.W = W; p.scale = p.shift = nullptr; p.nstride = C; p.act = 0;
  p.out = reinterpret_cast<u32x4*>(out); p.Hp = H + 2; p.Wp = W + 2; p.part = sum_scratch; p.zins = 0;
  p.gda = da; p.g_mean_rstd = mean_rstd; p.g_gamma = gamma; p.g_beta = beta; p.g_s12 = s12; p.g_groups = groups; p.g_act = act;
  const dim3 grid((unsigned)ceil_div(H * (W / 4), 256), (unsigned)(C / 8), (unsigned)N);
  const bool f16 = conv_op16_f16();
  if (gn_bwd_streaming(N, C, H * W)) {
    if (f16) ADM_LAUNCH((blk_apply_kernel<true, true, true>), grid, dim3(256), 0, st, p);
    else ADM_LAUNCH((blk_apply_kernel<false, true, true>), grid, dim3(256), 0, st, p);
  } else {
    if (f16) ADM_LAUNCH((blk_apply_kernel<true, true>), grid, dim3(256), 0, st, p);
    else ADM_LAUNCH((blk_apply_kernel<false, true>), grid, dim3(256), 0, st, p);
  }
  return ADM_CHECK_LAUNCH();
}

// the sums blk_apply left in sum_scratch -> per-(n, c) sums (written, or added with nc_accumulate) and per-c sums (ADDED)
int launch_blk_sums_finalize(const float* sum_scratch, int N, int C, int H, int W, float* out_nc, int nc_stride, int nc_accumulate,
                             float* out_c, hipStream_t st, float* out_c2) {
  if (out_nc == nullptr && out_c == nullptr && out_c2 == nullptr) return 0;
  ADM_LAUNCH(blk_sums_finalize_kernel, dim3((unsigned)ceil_div(C, 256), (unsigned)N), dim3(256), 0, st, sum_scratch, C,
             ceil_div(H * (W / 4), 256), out_nc, nc_stride, nc_accumulate, out_c, out_c2);
  return ADM_CHECK_LAUNCH();
}

// ---------------------------------------------------------------------------------------------- forward / data gradient
struct Bf16BConvParams {
  const u32x4* img; int Cg, Hp, Wp;     // blocked input, Cg = Cin / 8
  int N, H, W;
  const u32x4* wb; int Cout;            // filters [tap][Cin/8][Cout] x 16 B (adm_pack_bf16_weight)
  const float* bias; const float* chan_add; int chan_add_stride;
  const float* residual; float* out;
  double* stats;                        // NULL or GroupNorm partial sums of the output: [n][cout][tile][2] (sum, sum of squares)
  int tiles_x, tiles_y, n_ct, nblk;
  int nt;                               // streaming stores of the output / loads of the residual (tensors of >= 64 MB)
  int ksplit; float* part; long part_stride;   // > 1: K split over `ksplit` workgroups per tile, raw partial sums to slab kpart of `part`
  unsigned long long* prof;             // developer aid (ADM_BF16B_PROF=1): per-phase cycle counters, else NULL
};

#if defined(ADM_EMU)
#define BB_CLK() 0ull
#else
#define BB_CLK() ((unsigned long long)__builtin_readcyclecounter())
#endif
#define BB_LAP(slot) do { if (PROF) { const unsigned long long tn_ = BB_CLK(); pr[slot] += tn_ - tq; tq = tn_; } } while (0)

constexpr int FB_PR = 10;                                        // patch of an 8x32-pixel tile: 10 rows x 34 pixels per channel group (680 units per chunk)
constexpr int FB_BUF = 768;                                      // units per patch buffer: 12 DMA instructions of 64 units
constexpr int FB_NBUF = 3;
constexpr int FB_LDS_UNITS = FB_NBUF * FB_BUF + 4 * 9 * 64;      // + the four waves' nine-slot filter rings

// UP: the image is the HALF-resolution input of an Upsample2D convolution (nearest x2 folded into the patch addresses: patch
// pixel (r, c) of upsampled pixel (8 ty - 1 + r, 32 tx - 1 + c) is source unit (4 ty + (r + 1) / 2, 16 tx + (c + 1) / 2) of the
// haloed image — the halo doubles as the zero padding of the UPSAMPLED tensor); H, W are the output dims, Hp, Wp the image's.
// S2: the stride-2 3x3 convolutions of Downsample2D as every other pixel of the stride-1 "same" convolution o of the same image (the
// halo is the zero padding): S2 = 1, padding 1 (UNet2DModel's DownBlock2D): out(y, x) = o(2y, 2x); S2 = 2, pad (0, 1, 0, 1) then no
// padding (the AutoencoderKL encoder): out(y, x) = o(2y + 1, 2x + 1).  The tile is still 8 x 32 pixels of o, but only its four rows of
// that parity get accumulators (64 registers, half the MFMAs), and the epilogue stores the columns of that parity: H, W are the
// dims of the INPUT image, out is (N, Cout, H / 2, W / 2).
// LW < 5: planes of 16 (LW = 4) / 8 (LW = 3) pixels per row — the 16x16 and 8x8 levels of the 256x256 model.  The 32 columns of a tile
// are GI = 2 / 4 IMAGES side by side (tile "n" = a group of GI consecutive samples); neighbouring images share one zero column of
// the patch (the right halo of one is the left halo of the next: both are zero), so a patch row is 1 + GI (W + 1) = 35 / 37 units.
// K may be split over p.ksplit workgroups per tile (16..64 tiles cannot fill 256 CUs): raw partial sums to slab kpart, finished by
// ksplit_finish_kernel in slab order.
template <bool F16, bool UP = false, bool PROF = false, int S2 = 0, int LW = 5>
__global__ void __launch_bounds__(256, 2) conv_bf16b_kernel(const Bf16BConvParams p) {
  static_assert(LW == 5 || S2 == 0, "the stride-2 variant tiles 32-pixel rows only");
  constexpr int WI = 1 << LW, GI = 32 >> LW;             // image width the tile is cut for, images per tile
  constexpr int FB_PW = LW == 5 ? 34 : 1 + GI * (WI + 1);
  constexpr int FB_CGP = FB_PW * FB_PR, FB_UNITS = 2 * FB_CGP;
  static_assert(FB_UNITS <= FB_BUF, "patch does not fit its buffer");
  constexpr int NPT = S2 ? 4 : 8;                        // pixel rows (32-pixel N tiles) per wave
#define FB_ROW(q) (S2 ? 2 * (q) + (S2 - 1) : (q))
  unsigned long long pr[6] = {0, 0, 0, 0, 0, 0}, tq = 0;
  const unsigned long long t_start = BB_CLK();
  if (PROF) tq = t_start;
  ADM_DYN_SMEM(u32x4, lds);
  u32x4* const ldsP = lds;                               // [3][768] patch units: [channel group 2][row 10][pixel 34]
  const int tid = threadIdx.x, lane = tid & 63, wave = ADM_UNIFORM(tid >> 6);
  u32x4* const ldsA = lds + FB_NBUF * FB_BUF + wave * (9 * 64);   // this wave's ring: slot t = the fragment of tap t
  const int l31 = lane & 31, h = lane >> 5;
  int lid;
  {   // all cout tiles of a pixel tile, then the neighbouring pixel tile: neighbours on one XCD share the patch in its L2
    const int b = blockIdx.x, q = p.nblk >> 3, r = p.nblk & 7, xcd = b & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int ct = lid % p.n_ct; lid /= p.n_ct;
  int kpart = 0;
  if (p.ksplit > 1) { kpart = lid % p.ksplit; lid /= p.ksplit; }
  const int tx = lid % p.tiles_x; lid /= p.tiles_x;
  const int ty = lid % p.tiles_y, n = lid / p.tiles_y;   // LW < 5: n = group of GI images
  const int m0 = ct * 128 + wave * 32;
  const int c_begin = (int)((long)kpart * (p.Cg >> 1) / p.ksplit);
  const int n_chunks = (int)((long)(kpart + 1) * (p.Cg >> 1) / p.ksplit) - c_begin;     // >= 2 (launcher)
  const long planeU = (long)p.Hp * p.Wp;

  // patch DMA roles (tile- and chunk-invariant): instruction i of this wave fills units 64 (wave + 4 i) .. + 63 of a buffer
  unsigned poff[3];
  ADM_UNROLL
  for (int i = 0; i < 3; ++i) {
    int u = 64 * (wave + 4 * i) + lane;
    if (u > FB_UNITS - 1) u = FB_UNITS - 1;            // tail lanes re-request the last unit (they land in the buffer's pad)
    const int cgp = u / FB_CGP, rem = u - cgp * FB_CGP;
    const int r = rem / FB_PW;
    int c = rem - r * FB_PW;
    long img_off = 0;
    if (LW < 5) {      // patch column c = image j, haloed column c - j (W + 1); the last column is the right halo of the last image
      int j = c / (WI + 1);
      if (j > GI - 1) j = GI - 1;
      c -= j * (WI + 1);
      img_off = (long)j * p.Cg * planeU;
    }
    poff[i] = UP ? (unsigned)((img_off + (long)cgp * planeU + (long)((r + 1) >> 1) * p.Wp + ((c + 1) >> 1)) * 16)
                 : (unsigned)((img_off + (long)cgp * planeU + (long)r * p.Wp + c) * 16);
  }
  // haloed coordinates: the patch of output tile (ty, tx) starts at image pixel (8 ty - 1, 32 tx - 1) = unit (8 ty, 32 tx)
  const u32x4* const img_t = p.img + ((long)(LW < 5 ? n * GI : n) * p.Cg * p.Hp + (long)ty * (UP ? 4 : 8)) * p.Wp + (long)tx * (UP ? 16 : 32) +
                             (long)(2 * c_begin) * planeU;      // (chunk indices below are relative to this workgroup's first)
  const unsigned ldsP_a = ADM_LDS_ADDR(lds) + 1024u * wave;               // LDS byte addresses (wave-uniform integers)
  const unsigned ldsA_a = ADM_LDS_ADDR(lds) + 16u * (FB_NBUF * FB_BUF + 9 * 64 * wave);
  auto issue_patch = [&](int ch) __attribute__((always_inline)) {
    const u32x4* src = img_t + (long)(2 * ch) * planeU;                    // wave-uniform
    const unsigned dst = ldsP_a + 16u * FB_BUF * (unsigned)(ch % FB_NBUF);
    ADM_UNROLL
    for (int i = 0; i < 3; ++i) ADM_GLDS16_ASM(src, poff[i], dst + 4096u * i);
  };
  // filter fragment of (chunk, tap): lane (cout l31, k half h) = unit ((tap KG + 2 chunk + h) Cout + m0 + l31): two 512-byte runs
  const unsigned aoff = (unsigned)(((long)h * p.Cout + l31) * 16);
  auto issue_filt = [&](int ch, int t) __attribute__((always_inline)) {
    const u32x4* src = p.wb + ((long)t * p.Cg + 2 * (c_begin + ch)) * p.Cout + m0;     // wave-uniform
    ADM_GLDS16_ASM(src, aoff, ldsA_a + 1024u * t);
  };

  f32x16 acc[NPT];
  ADM_UNROLL
  for (int t = 0; t < NPT; ++t)
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // B fragment of pixel row pt (32 pixels), tap (dy, dx): unit h * 340 + (pt + dy) * 34 + dx + l31 — 32 consecutive units per half
  const int bbase = h * FB_CGP + l31 + (LW < 5 ? (l31 >> LW) : 0);      // LW < 5: image j = l31 / W starts at patch column j (W + 1)

  // ---- vector-memory queue of one wave (every entry is a DMA, issued in this order):
  //   prologue: P(0) x3, P(1) x3, A(0, 0..8)
  //   chunk c : [P(c+2) x3 if c + 2 < n] then after tap t's MFMAs A(c+1, t)            (nothing in the last chunk)
  // vmcnt(N) = "at most N entries outstanding"; entries complete in order.
  issue_patch(0);
  issue_patch(1);                                        // n_chunks >= 2 (launcher)
  ADM_UNROLL
  for (int t = 0; t < 9; ++t) issue_filt(0, t);
  ADM_WAIT_VMEM(12);                                     // own pieces of P(0) landed (younger: P(1) x3 + A x9)
  BB_LAP(1);                                             // prologue: address set-up + the first patch from HBM
  ADM_BARRIER_LGKM();
  BB_LAP(3);

  // MODE 0: a chunk with P(c+2) behind it; 1: the last but one (nothing left to request but the last chunk's filters);
  // 2: the last chunk (no requests at all) — three copies of the body so that every wait count is an immediate
  auto chunk = [&](int c, auto mode_tag) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode_tag)::value;
    constexpr bool LAST = MODE == 2, HASP = MODE == 0;
    if (HASP) issue_patch(c + 2);
    const u32x4* cur = ldsP + (c % FB_NBUF) * FB_BUF + bbase;
    u32x4 Ac, An, Bc[NPT], Bn[NPT];
    if (HASP) ADM_WAIT_VMEM(11); else ADM_WAIT_VMEM(8);  // A(c, 0): younger = A(c, 1..8) [+ P(c+2) x3]
    Ac = ldsA[lane];
    ADM_UNROLL
    for (int pt = 0; pt < NPT; ++pt) Bc[pt] = cur[FB_ROW(pt) * FB_PW];
    ADM_UNROLL
    for (int t = 0; t < 9; ++t) {
      if (t < 8) {
        // A(c, t+1): younger = A(c, t+2..8) [7 - t] [+ P(c+2) x3] + A(c+1, 0..t-1) [t, not in the last chunk]
        if (LAST) {
          switch (t) {
            case 0: ADM_WAIT_VMEM(7); break; case 1: ADM_WAIT_VMEM(6); break; case 2: ADM_WAIT_VMEM(5); break;
            case 3: ADM_WAIT_VMEM(4); break; case 4: ADM_WAIT_VMEM(3); break; case 5: ADM_WAIT_VMEM(2); break;
            case 6: ADM_WAIT_VMEM(1); break; default: ADM_WAIT_VMEM(0); break;
          }
        } else if (HASP) {
          ADM_WAIT_VMEM(10);
        } else {
          ADM_WAIT_VMEM(7);
        }
        An = ldsA[64 * (t + 1) + lane];
        ADM_UNROLL
        for (int pt = 0; pt < NPT; ++pt) Bn[pt] = cur[(FB_ROW(pt) + (t + 1) / 3) * FB_PW + (t + 1) % 3];
      }
      ADM_SCHED_FENCE();
      ADM_UNROLL
      for (int pt = 0; pt < NPT; ++pt) acc[pt] = ADM_MFMA16(F16, Ac, Bc[pt], acc[pt]);
      ADM_SCHED_FENCE();
      if (!LAST) issue_filt(c + 1, t);                   // slot t has been read into registers: refill it for the next chunk
      if (t < 8) {
        Ac = An;
        ADM_UNROLL
        for (int pt = 0; pt < NPT; ++pt) Bc[pt] = Bn[pt];
      }
    }
    // P(c+1) is older than A(c, 8), which has landed: this wave's pieces of the next patch are in LDS; the barrier makes the
    // other waves' pieces visible and tells everybody that buffer c % 3 (refilled next by P(c+3)) is no longer read
    BB_LAP(2);
    ADM_BARRIER_LGKM();
    BB_LAP(3);
  };
  for (int c = 0; c + 2 < n_chunks; ++c) chunk(c, std::integral_constant<int, 0>{});
  chunk(n_chunks - 2, std::integral_constant<int, 1>{});
  chunk(n_chunks - 1, std::integral_constant<int, 2>{});

  // epilogue: D row = output channel, column = pixel.  A wave's 32 couts x 32 pixels of one pixel row go through 4 KiB of its
  // own in LDS (the patch buffers are dead behind the last barrier) and come back as float4 per lane, so that bias, per-(n,
  // channel) term, residual and the store are 16-byte operations: one store instruction = 8 couts x 128 contiguous bytes,
  // 32 stores per wave instead of 128 dword stores (the dword version spent ~40 % of a workgroup's life ISSUING its stores).
  // LDS executes a wave's operations in order: no barrier, the next row's writes queue behind this row's reads.
  const int planeO = S2 ? (p.H >> 1) * (p.W >> 1) : p.H * p.W;
  const int n_img = LW < 5 ? n * GI : n;                                    // first image of the tile
  const bool split = p.ksplit > 1;                                          // raw partial sums to slab kpart (no bias / residual)
  float* const out_n = (split ? p.part + (long)kpart * p.part_stride : p.out) + (long)n_img * p.Cout * planeO;   // wave-uniform bases, 32-bit lane offsets
  const float* const res_n = (p.residual && !split) ? p.residual + (long)n_img * p.Cout * planeO : nullptr;
  float* const stage = reinterpret_cast<float*>(lds) + 1024 * wave;
  const int srow = lane >> 3, scol = 4 * (lane & 7);
  const int jimg = LW < 5 ? scol >> LW : 0;                                 // LW < 5: the lane's four columns lie in image jimg of the tile
  // S2: row 2 q (+ 1) of the tile is output row 4 ty + q, the lane's columns scol (+ 1), scol + 2 (+ 1) are output columns 16 tx + scol / 2 + {0, 1}
  const int lane_off = S2 ? (m0 + srow) * planeO + (ty * 4) * (p.W >> 1) + tx * 16 + (scol >> 1)
                     : LW < 5 ? (jimg * p.Cout + m0 + srow) * planeO + (ty * 8) * p.W + (scol & (WI - 1))
                              : (m0 + srow) * planeO + (ty * 8) * p.W + tx * 32 + scol;
  float bv[4], gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
  ADM_UNROLL
  for (int k = 0; k < 4; ++k) {
    const int co = m0 + srow + 8 * k;
    bv[k] = split ? 0.f : p.bias[co] + p.chan_add[(long)(n_img + jimg) * p.chan_add_stride + co];
  }
  ADM_UNROLL
  for (int pt = 0; pt < NPT; ++pt) {
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = acc[pt][r];
    ADM_WAVE_LDS_ORDER();
    float4 sv[4];
    ADM_UNROLL
    for (int k = 0; k < 4; ++k) sv[k] = *reinterpret_cast<const float4*>(stage + (srow + 8 * k) * 32 + scol);
    ADM_WAVE_LDS_ORDER();
    if (S2) {          // bias only (Downsample2D.conv has no per-sample term; the launcher refuses a residual): two odd columns per lane
      ADM_UNROLL
      for (int k = 0; k < 4; ++k) {
        const int o = lane_off + 8 * k * planeO + pt * (p.W >> 1);
        *reinterpret_cast<float2*>(out_n + o) = S2 == 2 ? make_float2(sv[k].y + bv[k], sv[k].w + bv[k])
                                                         : make_float2(sv[k].x + bv[k], sv[k].z + bv[k]);
      }
      continue;
    }
    ADM_UNROLL
    for (int k = 0; k < 4; ++k) {
      float4 v = sv[k];
      const int o = lane_off + 8 * k * planeO + pt * p.W;
      v.x += bv[k]; v.y += bv[k]; v.z += bv[k]; v.w += bv[k];
      if (res_n) {
        const float4 rr = p.nt ? adm_ld_nt(reinterpret_cast<const float4*>(res_n + o)) : *reinterpret_cast<const float4*>(res_n + o);
        v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
      }
      if (p.nt) adm_st_nt(reinterpret_cast<float4*>(out_n + o), v); else *reinterpret_cast<float4*>(out_n + o) = v;
      if (p.stats) {        // wave-uniform: statistics of the FINAL output values for the GroupNorm that reads this tensor
        gs[k] += (v.x + v.y) + (v.z + v.w);
        gq[k] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
    }
  }
  if (p.stats) {
    // the 8 lanes that share a cout row hold its 256 pixels: fp32 inside a lane (32 values), fp64 across the lanes
    ADM_UNROLL
    for (int k = 0; k < 4; ++k) {
      double a = (double)gs[k], b = (double)gq[k];
      ADM_UNROLL
      for (int m = 1; m <= 4; m <<= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
      if ((lane & 7) == 0) {
        double* d = p.stats + ((((long)n * p.Cout + m0 + srow + 8 * k) * p.tiles_y + ty) * p.tiles_x + tx) * 2;
        d[0] = a; d[1] = b;
      }
    }
  }
  if (PROF) {
    BB_LAP(4);                            // epilogue (issue only: the stores drain after the wave has left)
    pr[0] = BB_CLK() - t_start;
    if (lane == 0)
      for (int i = 0; i < 6; ++i) p.prof[((long)blockIdx.x * 4 + wave) * 8 + i] = pr[i];
  }
}

#undef FB_ROW

// N = 0: the 32-pixel-row tiling only. N > 0 additionally admits rows of 16 / 8 pixels when the batch fills whole tiles of 2 / 4 images.
bool conv_bf16b_eligible(int Cin, int Cout, int H, int W, int N) {
  if (!(Cin % 16 == 0 && Cin >= 32 && Cout % 128 == 0 && H % 8 == 0)) return false;
  if (W % 32 == 0) return true;
  static const int narrow = [] { const char* e = getenv("ADM_BF16B_NARROW"); return e ? atoi(e) : 1; }();     // developer A/B
  return narrow && N > 0 && (W == 16 || W == 8) && N % (32 / W) == 0 && Cin >= 64;
}

int conv_bf16b_stats_tiles(int H, int W) { return (H / 8) * (W / 32); }

// parts of the K split of the narrow-row tilings: a function of the layer only (see k_conv_mfma.hip launch_ksplit_generic), at least two
// 16-channel chunks each. 16-pixel rows: 4 (B = 16: 64 tiles -> 256 workgroups of 8 chunks at 512 channels; the slabs are 4 x the
// output), 8-pixel rows: 16 (16 tiles -> 256 workgroups of 2 chunks).
static int conv_bf16b_parts(int Cin, int W) {
  int S = W == 16 ? 4 : (W == 8 ? 16 : 1);
  while (S > 1 && (Cin / 16) / S < 2) S >>= 1;
  return S;
}

int launch_conv_bf16b(const void* img, int Cin, int N, int H, int W, const void* wb, int Cout, const float* bias,
                      const float* chan_add, int chan_add_stride, const float* residual, float* out, hipStream_t st, int mode,
                      double* stats_out) {
  // 1: nearest x2 of the image folded in; 2 / 3: stride-2 output (H, W = INPUT dims) of a pad-(0,1,0,1) / padding-1 convolution
  const int up = mode == 1, s2 = mode == 2 || mode == 3;
  ADM_REQUIRE(mode >= 0 && mode <= 3, "conv_bf16b: mode is 0 (stride 1), 1 (nearest x2 folded), 2 (stride 2, pad (0,1,0,1)) or 3 (stride 2, padding 1)");
  ADM_REQUIRE(!s2 || (residual == nullptr && chan_add == nullptr && stats_out == nullptr), "conv_bf16b: the stride-2 variant has a bias only");
  ADM_REQUIRE(conv_bf16b_eligible(Cin, Cout, H, W, N), "conv_bf16b: shape not eligible (Cin % 16, Cin >= 32, Cout % 128, H % 8, W % 32 — or W = 16 / 8 with N % 2 / 4 == 0)");
  const int lw = W % 32 == 0 ? 5 : (W == 16 ? 4 : 3), gi = 32 >> lw;
  ADM_REQUIRE(lw == 5 || (!s2 && stats_out == nullptr), "conv_bf16b: the 16- / 8-pixel-row tilings have no stride-2 variant and no statistics epilogue");
  Bf16BConvParams p;
  p.img = reinterpret_cast<const u32x4*>(img); p.Cg = Cin / 8;
  p.Hp = (up ? H / 2 : H) + 2; p.Wp = (up ? W / 2 : W) + 2;     // H, W: OUTPUT dims; the image of an up-convolution is half-size
  p.N = N; p.H = H; p.W = W;
  p.wb = reinterpret_cast<const u32x4*>(wb); p.Cout = Cout;
  p.bias = bias ? bias : conv_zero_bias(Cout);
  p.chan_add = chan_add; p.chan_add_stride = chan_add_stride;
  if (p.chan_add == nullptr) { p.chan_add = conv_zero_bias(Cout); p.chan_add_stride = 0; }
  ADM_REQUIRE(p.bias && p.chan_add, "conv_bf16b: constant buffers");
  p.residual = residual; p.out = out; p.stats = stats_out;
  {
    static const int nt_on = [] { const char* e = getenv("ADM_NT_CONV"); return e ? atoi(e) : 1; }();     // developer A/B
    p.nt = nt_on && !s2 && gn_bwd_streaming(N, Cout, H * W);
  }
  ADM_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0 && (reinterpret_cast<uintptr_t>(residual) & 15) == 0,
              "conv_bf16b: out / residual must be 16-byte aligned");
  p.tiles_x = lw == 5 ? W / 32 : 1; p.tiles_y = H / 8; p.n_ct = Cout / 128;
  p.ksplit = lw == 5 ? 1 : conv_bf16b_parts(Cin, W);
  p.part = nullptr; p.part_stride = (long)N * Cout * H * W;
  if (p.ksplit > 1) {
    p.part = conv_ksplit_scratch((size_t)p.ksplit * p.part_stride, st);
    if (p.part == nullptr) return -1;
  }
  p.nblk = p.tiles_x * p.tiles_y * (N / gi) * p.n_ct * p.ksplit;
  const size_t smem = sizeof(u32x4) * FB_LDS_UNITS;
#if !defined(ADM_EMU)
  static bool once = [] {
    (void)hipFuncSetAttribute((const void*)conv_bf16b_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_bf16b_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_bf16b_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_bf16b_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_bf16b_kernel<false, false, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_bf16b_kernel<true, false, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_bf16b_kernel<false, false, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_bf16b_kernel<true, false, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_bf16b_kernel<false, false, false, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_bf16b_kernel<true, false, false, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_bf16b_kernel<false, true, false, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_bf16b_kernel<true, true, false, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_bf16b_kernel<false, false, false, 0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_bf16b_kernel<true, false, false, 0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    return true;
  }();
  (void)once;
#endif
  set_last_conv_variant(5000 + 332);
  p.prof = nullptr;
#if !defined(ADM_EMU)
  static const bool want_prof = getenv("ADM_BF16B_PROF") != nullptr;
  if (want_prof && !conv_op16_f16() && mode == 0 && lw == 5) {   // developer aid: per-phase cycle accounting, printed after the launch (synchronous)
    const size_t pn = (size_t)p.nblk * 4 * 8;
    static unsigned long long* dprof = nullptr;
    static size_t dcap = 0;
    if (pn > dcap) { if (dprof) (void)hipFree(dprof); void* q = nullptr; (void)hipMalloc(&q, pn * sizeof(unsigned long long)); dprof = (unsigned long long*)q; dcap = pn; }
    static bool once_p = [] { (void)hipFuncSetAttribute((const void*)conv_bf16b_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); return true; }();
    (void)once_p;
    p.prof = dprof;
    ADM_LAUNCH((conv_bf16b_kernel<false, false, true>), dim3(p.nblk), dim3(256), smem, st, p);
    std::vector<unsigned long long> hv(pn);
    (void)hipMemcpyAsync(hv.data(), dprof, pn * sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    double h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < pn; ++i) h[i & 7] += (double)hv[i];
    const double w = 4.0 * p.nblk, nch = Cin / 16.0;
    fprintf(stderr, "[bf16b prof] %d->%d @%dx%d N=%d (%d workgroups): per wave: total %.0f | prologue %.0f | per chunk: mfma+issue %.0f barrier %.0f "
            "| epilogue issue %.0f cycles (%d chunks; 72 MFMAs = 2304)\n", Cin, Cout, H, W, N, p.nblk, h[0] / w, h[1] / w, h[2] / w / nch,
            h[3] / w / nch, h[4] / w, (int)nch);
    return ADM_CHECK_LAUNCH();
  }
#endif
  if (lw < 5) {
    ADM_REQUIRE(lw == 4 || !up, "conv_bf16b: no nearest-x2 variant for 8-pixel rows");
    const bool f16 = conv_op16_f16();
    if (lw == 3) {
      if (f16) ADM_LAUNCH((conv_bf16b_kernel<true, false, false, 0, 3>), dim3(p.nblk), dim3(256), smem, st, p);
      else ADM_LAUNCH((conv_bf16b_kernel<false, false, false, 0, 3>), dim3(p.nblk), dim3(256), smem, st, p);
    } else if (up) {
      if (f16) ADM_LAUNCH((conv_bf16b_kernel<true, true, false, 0, 4>), dim3(p.nblk), dim3(256), smem, st, p);
      else ADM_LAUNCH((conv_bf16b_kernel<false, true, false, 0, 4>), dim3(p.nblk), dim3(256), smem, st, p);
    } else {
      if (f16) ADM_LAUNCH((conv_bf16b_kernel<true, false, false, 0, 4>), dim3(p.nblk), dim3(256), smem, st, p);
      else ADM_LAUNCH((conv_bf16b_kernel<false, false, false, 0, 4>), dim3(p.nblk), dim3(256), smem, st, p);
    }
    if (ADM_CHECK_LAUNCH() != 0) return -1;
    if (p.ksplit > 1) {
      if (const GnFuse* f = conv_gn_fuse_pending(Cout))      // the executor announced the GroupNorm that reads `out` next (k_groupnorm.hip)
        return launch_ksplit_finish_gn(p.part, p.ksplit, p.part_stride, bias ? bias : conv_zero_bias(Cout), chan_add, chan_add_stride, residual,
                                       out, N, Cout, H * W, *f, st);
      return launch_ksplit_finish(p.part, p.ksplit, p.part_stride, bias, chan_add, chan_add_stride, residual, out, Cout, H * W, st);
    }
    return 0;
  }
  if (mode == 2) {
    if (conv_op16_f16()) ADM_LAUNCH((conv_bf16b_kernel<true, false, false, 2>), dim3(p.nblk), dim3(256), smem, st, p);
    else ADM_LAUNCH((conv_bf16b_kernel<false, false, false, 2>), dim3(p.nblk), dim3(256), smem, st, p);
  } else if (mode == 3) {
    if (conv_op16_f16()) ADM_LAUNCH((conv_bf16b_kernel<true, false, false, 1>), dim3(p.nblk), dim3(256), smem, st, p);
    else ADM_LAUNCH((conv_bf16b_kernel<false, false, false, 1>), dim3(p.nblk), dim3(256), smem, st, p);
  } else if (up) {
    if (conv_op16_f16()) ADM_LAUNCH((conv_bf16b_kernel<true, true>), dim3(p.nblk), dim3(256), smem, st, p);
    else ADM_LAUNCH((conv_bf16b_kernel<false, true>), dim3(p.nblk), dim3(256), smem, st, p);
  } else {
    if (conv_op16_f16()) ADM_LAUNCH((conv_bf16b_kernel<true, false>), dim3(p.nblk), dim3(256), smem, st, p);
    else ADM_LAUNCH((conv_bf16b_kernel<false, false>), dim3(p.nblk), dim3(256), smem, st, p);
  }
  return ADM_CHECK_LAUNCH();
}

// ---------------------------------------------------------------------------------------------- weight gradient
// dW[co][ci][tap] = sum over (n, y, x) of dy[n][co][y][x] * a[n][ci][y + dy - 1][x + dx - 1]; GEMM with k = pixels, one MFMA
// k-step = 16 consecutive pixels of a row.  Both operands arrive channel-blocked; LDS holds them as slabs of 16 channels,
// [slab][pixel][16 channels] = 32 B per pixel, and ds_read_b64_tr_b16 hands a lane 4 pixels of ONE channel (two reads = the
// 8 k-values of an MFMA operand).  A 16-lane group reads 128 consecutive bytes (4 pixels); the two groups of a 32-lane half
// read the same pixels of two slabs whose bases differ by 128 (mod 256) bytes: all 64 banks, conflict-free.
struct Bf16BWgradParams {
  const u32x4* xa; int CgI;             // activated input image, CgI = Ct / 8
  const u32x4* dyb; int CgO;            // output-gradient image, CgO = Cout / 8
  int Hp, Wp, N, H, W, Cout, Ct;       // Hp, Wp: the dy image (H + 2, W + 2)
  int HpX, WpX;                         // the input image: the same, or (H / 2 + 2, W / 2 + 2) for an up-convolution
  float* part;
  int tiles_x, tiles_y, n_ptiles, n_ct, n_ci, split, tiles_per_block, nblk;
  unsigned mTX, mTXY;
};

constexpr int WB_DSLAB = 264;                   // units per dy slab: 128 pixels x 2 + 8 (128 bytes) so that slab bases alternate halves
constexpr int WB_DY = 8 * WB_DSLAB;             // 2112 units: 128 couts
// patch slab of 16 cins: 6 rows x PW pixels x 2 units, padded so that slab bases alternate 128-byte halves (stride = 128 mod 256 bytes):
// rows of 32 pixels (LW = 5): PW = 34, 408 units = 6528 bytes, no pad; LW = 4 / 3 (2 / 4 images side by side, see conv_bf16b_kernel):
// PW = 35 / 37, 420 -> 424 / 444 -> 456 units
constexpr int wb_pw(int LW) { return LW == 5 ? 34 : 1 + (32 >> LW) * ((1 << LW) + 1); }
constexpr int wb_xslab(int LW) { return LW == 5 ? 408 : (LW == 4 ? 424 : 456); }
constexpr int wb_buf(int LW) { return WB_DY + 4 * wb_xslab(LW) + 32; }     // + 32 units of slack behind the half-used last DMA instruction
constexpr int wb_ndma(int LW) { return 32 + (4 * wb_xslab(LW) + 63) / 64; }   // DMA instructions per tile: 32 (dy) + 26 / 27 / 29 (patch)

__device__ __forceinline__ int bdivb(int n, int d, unsigned magic) {   // n / d; exact via umulhi for n, d < 2^16
  return magic ? (int)(((unsigned long long)(unsigned)n * magic) >> 32) : n / d;
}

template <bool F16, bool UP = false, int LW = 5>
__global__ void __launch_bounds__(512, 2) conv_wgradb_kernel(const Bf16BWgradParams p) {
  constexpr int WI = 1 << LW, GI = 32 >> LW;
  constexpr int PW = wb_pw(LW), WB_XSLAB = wb_xslab(LW), WB_XA = 4 * WB_XSLAB, WB_BUF = wb_buf(LW), WB_NDMA = wb_ndma(LW);
  static_assert((WB_XSLAB * 16) % 256 == 128 && WB_XSLAB >= 6 * PW * 2 && WB_NDMA <= 64, "patch slab geometry");
  ADM_DYN_SMEM(u32x4, lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = ADM_UNIFORM(tid >> 6);
  const int wc = wave & 3, wi = wave >> 2;          // 32-cout block, 32-cin block of this wave
  int lid;
  {   // the n_ci workgroups that read the same dy tiles (and the n_ct that read the same patches) are neighbours on one XCD
    const int b = blockIdx.x, q = p.nblk >> 3, r = p.nblk & 7, xcd = b & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int cic = lid % p.n_ci; lid /= p.n_ci;
  const int ct = lid % p.n_ct, sp = lid / p.n_ct;
  const long planeU = (long)p.Hp * p.Wp, planeX = (long)p.HpX * p.WpX;
  const int t_begin = sp * p.tiles_per_block;
  int t_end = t_begin + p.tiles_per_block;
  if (t_end > p.n_ptiles) t_end = p.n_ptiles;

  // DMA roles: instruction j = wave + 8 i (i = 0..7, j < 58); i < 4 -> dy image, i >= 4 -> patch
  unsigned goff[8];
  int ldst[8];
  ADM_UNROLL
  for (int i = 0; i < 8; ++i) {
    const int j = wave + 8 * i;
    if (i < 4) {
      const int slab = j >> 2, quarter = j & 3;
      const int u = quarter * 64 + lane, px = u >> 1, cgs = u & 1;
      // LW < 5: column l of the tile = image l / W of the group, column l % W
      const long img_off = LW < 5 ? (long)((px & 31) >> LW) * p.CgO * planeU : 0;
      goff[i] = (unsigned)((img_off + (long)(slab * 2 + cgs) * planeU + (long)(px >> 5) * p.Wp + (LW < 5 ? (px & (WI - 1)) : (px & 31))) * 16);
      ldst[i] = slab * WB_DSLAB + quarter * 64;
    } else {
      const int k = j - 32;
      int U = 64 * k + lane;
      if (U > WB_XA - 1) U = WB_XA - 1;
      const int slab = U / WB_XSLAB;
      int rem = U - slab * WB_XSLAB;
      if (rem > 6 * PW * 2 - 1) rem = 6 * PW * 2 - 1;      // the slab's pad units re-request its last unit
      const int pp = rem >> 1, cgs = rem & 1, prow = pp / PW;
      int pcol = pp - prow * PW;
      long img_off = 0;
      if (LW < 5) {       // patch column = image j, haloed column pcol - j (W + 1); neighbours share their zero halo column
        int j = pcol / (WI + 1);
        if (j > GI - 1) j = GI - 1;
        pcol -= j * (WI + 1);
        img_off = (long)j * p.CgI * planeX;
      }
      goff[i] = UP ? (unsigned)((img_off + (long)(slab * 2 + cgs) * planeX + (long)((prow + 1) >> 1) * p.WpX + ((pcol + 1) >> 1)) * 16)
                   : (unsigned)((img_off + (long)(slab * 2 + cgs) * planeX + (long)prow * p.WpX + pcol) * 16);
      ldst[i] = WB_DY + 64 * k;
    }
  }
  const unsigned lds_a = ADM_LDS_ADDR(lds);
  const u32x4* const dy_c = p.dyb + (long)(ct * 16) * planeU;
  const u32x4* const xa_c = p.xa + (long)(cic * 8) * planeX;
  auto issue_tile = [&](int pt, int buf) __attribute__((always_inline)) {
    const int nimg = bdivb(pt, p.tiles_x * p.tiles_y, p.mTXY);
    const int rem = pt - nimg * (p.tiles_x * p.tiles_y);
    const int ty = bdivb(rem, p.tiles_x, p.mTX), tx = rem - ty * p.tiles_x;
    // haloed coordinates: dy pixel (4 ty, 32 tx) = unit (4 ty + 1, 32 tx + 1); the patch starts one up / left = unit (4 ty, 32 tx)
    const int n0 = LW < 5 ? nimg * GI : nimg;            // LW < 5: "image" of the tile index = group of GI samples
    const u32x4* dsrc = dy_c + ((long)n0 * p.CgO * p.Hp + (long)(ty * 4 + 1)) * p.Wp + (long)(tx * 32 + 1);
    const u32x4* xsrc = xa_c + ((long)n0 * p.CgI * p.HpX + (long)(ty * (UP ? 2 : 4))) * p.WpX + (long)(tx * (UP ? 16 : 32));
    const unsigned base = lds_a + 16u * WB_BUF * (unsigned)buf;
    ADM_UNROLL
    for (int i = 0; i < 8; ++i) {
      if (i < 7 || wave < WB_NDMA - 56) {                // j = wave + 56 exists for waves 0 and 1 only (wave-uniform)
        if (i < 4) ADM_GLDS16_ASM(dsrc, goff[i], base + 16u * ldst[i]);
        else ADM_GLDS16_ASM(xsrc, goff[i], base + 16u * ldst[i]);
      }
    }
  };

  f32x16 acc[9];
  ADM_UNROLL
  for (int t = 0; t < 9; ++t)
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // per-lane read bases (bytes inside a buffer): slab of the lane's 16-lane group, pixel 8 (lane >> 5) + ((lane & 15) >> 2),
  // 8-byte piece lane & 3
  const int lgrp = (lane >> 4) & 1, lpx = 8 * (lane >> 5) + ((lane & 15) >> 2), lpc = lane & 3;
  const int abase = (2 * wc + lgrp) * (WB_DSLAB * 16) + lpx * 32 + lpc * 8;
  // 8-pixel rows: the two 8-pixel halves of a k-step are two images, one shared halo column apart in the patch
  const int xbase = WB_DY * 16 + (2 * wi + lgrp) * (WB_XSLAB * 16) + (lpx + (LW == 3 ? (lane >> 5) : 0)) * 32 + lpc * 8;
  constexpr int XC1 = LW == 5 ? 16 : (LW == 4 ? 17 : 18);      // patch column of the tile's pixel 16 (the second k-step of a row)

  auto read_frag = [&](const unsigned char* b, int byte_off) __attribute__((always_inline)) -> u32x4 {
    const adm_u32x2 lo = ADM_DS_READ_TR16_B64(b + byte_off), hi = ADM_DS_READ_TR16_B64(b + byte_off + 128);
    u32x4 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = hi[0]; r[3] = hi[1];
    return r;
  };

  issue_tile(t_begin, 0);
  int it = 0;
  for (int pt = t_begin; pt < t_end; ++pt, ++it) {
    ADM_WAIT_VMEM(0);                                    // this wave's pieces of tile pt (nothing younger is in flight)
    ADM_BARRIER_LGKM();                                  // everybody's pieces; everybody is done with the other buffer
    if (pt + 1 < t_end) issue_tile(pt + 1, (it + 1) & 1);
    const unsigned char* buf = reinterpret_cast<const unsigned char*>(lds + (it & 1) * WB_BUF);
    const unsigned char* bA = buf + abase;
    const unsigned char* bX = buf + xbase;
    u32x4 Ac, Bc[9], An, Bn[9];
    Ac = read_frag(bA, 0);
    ADM_UNROLL
    for (int t = 0; t < 9; ++t) Bc[t] = read_frag(bX, ((t / 3) * PW + (t % 3)) * 32);
    ADM_UNROLL
    for (int s = 0; s < 8; ++s) {
      if (s < 7) {
        const int row = (s + 1) >> 1, col0 = ((s + 1) & 1) * 16;
        An = read_frag(bA, (row * 32 + col0) * 32);
        ADM_UNROLL
        for (int t = 0; t < 9; ++t) Bn[t] = read_frag(bX, ((row + t / 3) * PW + ((s + 1) & 1) * XC1 + (t % 3)) * 32);
      }
      ADM_SCHED_FENCE();
      ADM_UNROLL
      for (int t = 0; t < 9; ++t) acc[t] = ADM_MFMA16(F16, Ac, Bc[t], acc[t]);
      ADM_SCHED_FENCE();
      if (s < 7) {
        Ac = An;
        ADM_UNROLL
        for (int t = 0; t < 9; ++t) Bc[t] = Bn[t];
      }
    }
  }

  // partial slab [split][tap][cout][cin] (wgrad_reduce9_kernel sums the slabs and transposes to (cout, cin, tap))
  float* out = p.part + (long)sp * p.Cout * p.Ct * 9;
  const int cc = cic * 64 + wi * 32 + (lane & 31);
  ADM_UNROLL
  for (int t = 0; t < 9; ++t) {
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int co = ct * 128 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      out[((long)t * p.Cout + co) * p.Ct + cc] = acc[t][r];
    }
  }
}

// N > 0 additionally admits rows of 16 / 8 pixels when the batch fills whole tiles of 2 / 4 images (as conv_bf16b_eligible)
bool conv_wgradb_eligible(int Ct, int Cout, int H, int W, int N) {
  if (!(Ct % 64 == 0 && Cout % 128 == 0 && H % 4 == 0)) return false;
  if (W % 32 == 0) return true;
  static const int narrow = [] { const char* e = getenv("ADM_BF16B_NARROW"); return e ? atoi(e) : 1; }();     // developer A/B
  return narrow && N > 0 && (W == 16 || W == 8) && N % (32 / W) == 0;
}

// split-K factor and workspace floats ([split][Cout * Ct * 9]) of the blocked weight-gradient kernel
long conv_wgradb_workspace(int Ct, int Cout, int N, int H, int W, int* split_out) {
  const int n_ptiles = W % 32 == 0 ? (W / 32) * (H / 4) * N : (H / 4) * (N / (32 / W));
  const int pairs = (Cout / 128) * (Ct / 64);
  static const int target = [] { const char* e = getenv("ADM_WGRADB_WGS"); return e ? atoi(e) : 256; }();
  int split = ceil_div(target, pairs);           // one workgroup per CU (120 KiB of LDS each): 256 measured 68.3 ms per step, 384: 70.5, 512: 69.7
  if (split > n_ptiles) split = n_ptiles;
  if (split < 1) split = 1;
  const int tpb = ceil_div(n_ptiles, split);
  split = ceil_div(n_ptiles, tpb);               // no empty workgroups
  if (split_out) *split_out = split;
  return (long)split * Cout * Ct * 9;
}

int launch_conv_wgradb(const void* xa, int Ct, const void* dyb, int Cout, int N, int H, int W, float* dW, int accumulate,
                       float* workspace, hipStream_t st, int up) {
  ADM_REQUIRE(conv_wgradb_eligible(Ct, Cout, H, W, N), "conv_wgradb: shape not eligible (Ct % 64, Cout % 128, H % 4, W % 32 — or W = 16 / 8 with N % 2 / 4 == 0)");
  const int lw = W % 32 == 0 ? 5 : (W == 16 ? 4 : 3);
  ADM_REQUIRE(lw >= 4 || !up, "conv_wgradb: no nearest-x2 variant for 8-pixel rows");
  Bf16BWgradParams p;
  p.xa = reinterpret_cast<const u32x4*>(xa); p.CgI = Ct / 8;
  p.dyb = reinterpret_cast<const u32x4*>(dyb); p.CgO = Cout / 8;
  p.Hp = H + 2; p.Wp = W + 2; p.N = N; p.H = H; p.W = W; p.Cout = Cout; p.Ct = Ct;   // H, W: output (= dy) dims
  p.HpX = (up ? H / 2 : H) + 2; p.WpX = (up ? W / 2 : W) + 2;
  p.part = workspace;
  p.tiles_x = lw == 5 ? W / 32 : 1; p.tiles_y = H / 4;
  p.n_ptiles = p.tiles_x * p.tiles_y * (N >> (5 - lw));
  p.n_ct = Cout / 128; p.n_ci = Ct / 64;
  conv_wgradb_workspace(Ct, Cout, N, H, W, &p.split);
  p.tiles_per_block = ceil_div(p.n_ptiles, p.split);
  p.nblk = p.n_ct * p.n_ci * p.split;
  auto magic = [&](long d) { return (d <= 1 || p.n_ptiles >= 65536) ? 0u : (unsigned)((1ULL << 32) / (unsigned long long)d + 1ULL); };
  p.mTX = magic(p.tiles_x); p.mTXY = magic((long)p.tiles_x * p.tiles_y);
  const size_t smem = sizeof(u32x4) * 2 * wb_buf(lw);
#if !defined(ADM_EMU)
  static bool once = [] {
    (void)hipFuncSetAttribute((const void*)conv_wgradb_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgradb_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgradb_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgradb_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgradb_kernel<false, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgradb_kernel<true, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgradb_kernel<false, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgradb_kernel<true, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgradb_kernel<false, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgradb_kernel<true, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    return true;
  }();
  (void)once;
#endif
  const bool f16 = conv_op16_f16();
  if (lw == 3) {
    if (f16) ADM_LAUNCH((conv_wgradb_kernel<true, false, 3>), dim3(p.nblk), dim3(512), smem, st, p);
    else ADM_LAUNCH((conv_wgradb_kernel<false, false, 3>), dim3(p.nblk), dim3(512), smem, st, p);
  } else if (lw == 4 && up) {
    if (f16) ADM_LAUNCH((conv_wgradb_kernel<true, true, 4>), dim3(p.nblk), dim3(512), smem, st, p);
    else ADM_LAUNCH((conv_wgradb_kernel<false, true, 4>), dim3(p.nblk), dim3(512), smem, st, p);
  } else if (lw == 4) {
    if (f16) ADM_LAUNCH((conv_wgradb_kernel<true, false, 4>), dim3(p.nblk), dim3(512), smem, st, p);
    else ADM_LAUNCH((conv_wgradb_kernel<false, false, 4>), dim3(p.nblk), dim3(512), smem, st, p);
  } else if (up) {
    if (conv_op16_f16()) ADM_LAUNCH((conv_wgradb_kernel<true, true>), dim3(p.nblk), dim3(512), smem, st, p);
    else ADM_LAUNCH((conv_wgradb_kernel<false, true>), dim3(p.nblk), dim3(512), smem, st, p);
  } else {
    if (conv_op16_f16()) ADM_LAUNCH((conv_wgradb_kernel<true, false>), dim3(p.nblk), dim3(512), smem, st, p);
    else ADM_LAUNCH((conv_wgradb_kernel<false, false>), dim3(p.nblk), dim3(512), smem, st, p);
  }
  ADM_TRY(ADM_CHECK_LAUNCH());
  return launch_wgrad_reduce(workspace, p.split, (long)Cout * Ct * 9, dW, accumulate, 9, st);
}

}  // namespace adm
