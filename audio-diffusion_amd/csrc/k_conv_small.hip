// k_conv_small.hip — direct 3x3 convolutions for the two degenerate GEMM shapes of the UNet/VAE:
//   conv_in  class (Cin <= 4, e.g. 1 -> 128 @256^2): K = 9 — write-bound, one thread per output pixel;
//   conv_out class (Cout <= 4, e.g. GN+SiLU -> 128 -> 1 @256^2): read-bound, LDS-staged 18x18 halo tiles
//   with the GroupNorm affine + SiLU applied once per staged element.
// Both are HBM-bound: algorithmic bytes = 4*(N*Cin*H*W + N*Cout*H*W).
// Stride 1, ks = 3 (pad 1) or ks = 1 only; anything else is routed to the MFMA kernel by launch_conv2d.
#include "adm_kernels.h"

namespace adm {

__device__ __forceinline__ float silu_s(float v) { return v / (1.0f + __expf(-v)); }

// ---- Cin <= 4: each thread one output pixel, loops over all couts -------------------------------------
template <int CIN, int KS>
__global__ void __launch_bounds__(256) conv_small_cin_kernel(const float* __restrict__ x, int N, int H, int W,
                                                             const float* __restrict__ wp,  // [Cin][tap][Cout]
                                                             const float* __restrict__ bias, int Cout,
                                                             const float* __restrict__ residual,
                                                             float* __restrict__ out) {
  constexpr int KS2 = KS * KS, PAD = KS / 2, K = CIN * KS2;
  const long HW = (long)H * W;
  const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;
  if (pix >= HW) return;
  const int y = (int)(pix / W), xx = (int)(pix % W);
  float v[K];
  ADM_UNROLL
  for (int c = 0; c < CIN; ++c)
    ADM_UNROLL
    for (int t = 0; t < KS2; ++t) {
      const int gy = y + t / KS - PAD, gx = xx + t % KS - PAD;
      v[c * KS2 + t] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? x[((long)n * CIN + c) * HW + (long)gy * W + gx] : 0.f;
    }
  for (int co = 0; co < Cout; ++co) {
    float acc = bias ? bias[co] : 0.f;
    ADM_UNROLL
    for (int k = 0; k < K; ++k) acc = fmaf(wp[(long)k * Cout + co], v[k], acc);
    const long o = ((long)n * Cout + co) * HW + pix;
    if (residual) acc += residual[o];
    out[o] = acc;
  }
}

// ---- Cin <= 4, four pixels per thread (round 2; W % 4 == 0): float4 stores, and optionally the GroupNorm partial sums of
// the output (adm_conv_args.stats_out): per (sample, cout, wave of 256 pixels) the pair (sum, sum of squares) — the lane's four
// values and the 64 lanes of the wave in fp32 (relative error ~1e-7 on a 256-term sum, below what the fp32 scale / shift of the
// consumer resolves), across tiles in fp64 by gn_finalize_kernel. conv_in's output is read by two GroupNorms (the first resnet
// and, through the skip connection, the last one): 0.6 ms of read-only statistics passes per forward at B = 32.
template <int CIN>
__global__ void __launch_bounds__(256) conv_small_cin_wide_kernel(const float* __restrict__ x, int N, int H, int W,
                                                                  const float* __restrict__ wp,  // [Cin][tap][Cout]
                                                                  const float* __restrict__ bias, int Cout,
                                                                  const float* __restrict__ residual,
                                                                  float* __restrict__ out, double* __restrict__ stats,
                                                                  int stats_tiles) {
  const long HW = (long)H * W;
  const long pix = 4 * ((long)blockIdx.x * blockDim.x + threadIdx.x);     // first of the thread's four pixels (same row)
  const int n = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool live = pix < HW;
  const int y = live ? (int)(pix / W) : 0, xx = live ? (int)(pix % W) : 0;
  const bool x_al = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  float v[CIN][3][6];                                                      // rows y - 1 .. y + 1, columns xx - 1 .. xx + 4
  ADM_UNROLL
  for (int c = 0; c < CIN; ++c)
    ADM_UNROLL
    for (int dy = 0; dy < 3; ++dy) {
      const int gy = y + dy - 1;
      const bool rok = live && gy >= 0 && gy < H;
      const float* row = x + ((long)n * CIN + c) * HW + (long)(rok ? gy : 0) * W + xx;
      float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rok) {                       // the network input is the caller's pointer: 16-byte loads only when it allows them (uniform)
        if (x_al) m = *reinterpret_cast<const float4*>(row);
        else m = make_float4(row[0], row[1], row[2], row[3]);
      }
      v[c][dy][0] = (rok && xx > 0) ? row[-1] : 0.f;
      v[c][dy][1] = m.x; v[c][dy][2] = m.y; v[c][dy][3] = m.z; v[c][dy][4] = m.w;
      v[c][dy][5] = (rok && xx + 4 < W) ? row[4] : 0.f;
    }
  // gridDim.z parts of the output channels (small images: a 32x32 latent at B = 16 is 16 workgroups of pixels — each used to walk
  // all 128 output channels, 75 us on 16 CUs); every output is computed exactly as before, whatever the split
  const int co_per = (Cout + (int)gridDim.z - 1) / (int)gridDim.z;
  const int co_begin = (int)blockIdx.z * co_per, co_end = co_begin + co_per < Cout ? co_begin + co_per : Cout;
  for (int co = co_begin; co < co_end; ++co) {
    const float b = bias ? bias[co] : 0.f;
    float acc[4] = {b, b, b, b};
    ADM_UNROLL
    for (int c = 0; c < CIN; ++c)
      ADM_UNROLL
      for (int t = 0; t < 9; ++t) {
        const float w = wp[(long)(c * 9 + t) * Cout + co];               // uniform -> scalar load
        ADM_UNROLL
        for (int px = 0; px < 4; ++px) acc[px] = fmaf(w, v[c][t / 3][px + t % 3], acc[px]);
      }
    const long o = ((long)n * Cout + co) * HW + pix;
    if (live) {
      if (residual) {
        const float4 q = *reinterpret_cast<const float4*>(residual + o);
        acc[0] += q.x; acc[1] += q.y; acc[2] += q.z; acc[3] += q.w;
      }
      *reinterpret_cast<float4*>(out + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    if (stats != nullptr) {                                               // uniform
      float s1 = live ? (acc[0] + acc[1]) + (acc[2] + acc[3]) : 0.f;
      float s2 = live ? (acc[0] * acc[0] + acc[1] * acc[1]) + (acc[2] * acc[2] + acc[3] * acc[3]) : 0.f;
      ADM_UNROLL
      for (int m = 32; m >= 1; m >>= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }
      if (lane == 0) {
        double* dst = stats + (((long)n * Cout + co) * stats_tiles + blockIdx.x * 4 + wave) * 2;
        dst[0] = (double)s1; dst[1] = (double)s2;
      }
    }
  }
}

// ---- Cout <= 4: 16x16 output tile per workgroup, channels staged 8 at a time through LDS ----------------
// Weights are wave-uniform: they are read straight from global memory (scalar loads into SGPRs), not from LDS,
// so the only LDS traffic in the inner loop is the 9 activated taps per channel.
constexpr int SC = 8;
template <int COUT>
__global__ void __launch_bounds__(256) conv_small_cout_kernel(const float* __restrict__ x, int Cin, int N, int H,
                                                              int W, const float* __restrict__ gn_scale,
                                                              const float* __restrict__ gn_shift, int act,
                                                              const float* __restrict__ wp,  // [Cin][tap][Cout]
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ residual,
                                                              float* __restrict__ out, int tiles_x,
                                                              float* __restrict__ part, long part_stride) {
  // part != nullptr (small images): gridDim.z workgroups share a tile, each walks 1 / gridDim.z of the channel chunks and stores
  // its partial sums to slab blockIdx.z; ksplit_finish_kernel adds the slabs in order with bias and residual (deterministic). One
  // 32x32 latent at B = 16 is 64 tiles: the 16-chunk channel loop ran 112 us on 64 CUs.
  __shared__ float tile[SC][18][18 + 1];
  const int tid = threadIdx.x;
  // Workgroups are dealt to the 8 XCDs round-robin and each XCD has its own L2: give every XCD a contiguous
  // run of (image, tile) pairs so the halo rows and the half-used 128-byte lines that neighbouring tiles share
  // are fetched from HBM once per XCD run instead of once per tile.
  int wg = blockIdx.y * gridDim.x + blockIdx.x;
  const int total = gridDim.x * gridDim.y;
  if ((total & 7) == 0) wg = (wg & 7) * (total >> 3) + (wg >> 3);
  const int n = wg / (int)gridDim.x, tl = wg - n * (int)gridDim.x;
  const int tx = tl % tiles_x, ty = tl / tiles_x;
  const int lx = tid & 15, ly = tid >> 4;
  const int ox = tx * 16 + lx, oy = ty * 16 + ly;
  const long HW = (long)H * W;
  // per-thread gather plan over the 18x18 halo (2 elements per thread: 324 = 256 + 68)
  int goff[2];
  ADM_UNROLL
  for (int k = 0; k < 2; ++k) {
    const int e = tid + 256 * k;
    goff[k] = -2;  // -2: not mine, -1: zero padding
    if (e < 324) {
      const int yy = e / 18, xx = e - yy * 18;
      const int gy = ty * 16 + yy - 1, gx = tx * 16 + xx - 1;
      goff[k] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? gy * W + gx : -1;
    }
  }
  float acc[COUT];
  ADM_UNROLL
  for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
  const int nchunk = (Cin + SC - 1) / SC;
  const int ch_lo = (int)((long)blockIdx.z * nchunk / gridDim.z) * SC, ch_hi = (int)((long)(blockIdx.z + 1) * nchunk / gridDim.z) * SC;
  for (int c0 = ch_lo; c0 < ch_hi && c0 < Cin; c0 += SC) {
    ADM_UNROLL
    for (int k = 0; k < 2; ++k) {
      const int e = tid + 256 * k;
      if (goff[k] != -2) {
        const int yy = e / 18, xx = e - yy * 18;
        ADM_UNROLL
        for (int c = 0; c < SC; ++c) {
          float v = 0.f;
          if (goff[k] >= 0 && c0 + c < Cin) {
            v = x[((long)n * Cin + c0 + c) * HW + goff[k]];
            if (gn_scale) v = v * gn_scale[(long)n * Cin + c0 + c] + gn_shift[(long)n * Cin + c0 + c];
            if (act) v = __fdividef(v, 1.0f + __expf(-v));
          }
          tile[c][yy][xx] = v;
        }
      }
    }
    __syncthreads();
    ADM_UNROLL
    for (int c = 0; c < SC; ++c) {
      if (c0 + c < Cin) {
        const float* wr = wp + (long)(c0 + c) * 9 * COUT;  // uniform address -> scalar loads
        ADM_UNROLL
        for (int t = 0; t < 9; ++t) {
          const float v = tile[c][ly + t / 3][lx + t % 3];
          ADM_UNROLL
          for (int co = 0; co < COUT; ++co) acc[co] = fmaf(wr[t * COUT + co], v, acc[co]);
        }
      }
    }
    __syncthreads();
  }
  if (ox < W && oy < H) {
    ADM_UNROLL
    for (int co = 0; co < COUT; ++co) {
      const long o = ((long)n * COUT + co) * HW + (long)oy * W + ox;
      if (part != nullptr) { part[(long)blockIdx.z * part_stride + o] = acc[co]; continue; }
      float v = acc[co] + (bias ? bias[co] : 0.f);
      if (residual) v += residual[o];
      out[o] = v;
    }
  }
}

// ---- Cout <= 4, wide tiles (round 2): 64 x 16 output pixels per workgroup ----------------------------------------------
// The 16x16 kernel above fetches 4.27 GB for a 1.07 GB input at B = 32 (profiles/r01_pmc_forward.json): its 18-pixel patch rows
// are 72-byte pieces of 128-byte lines, gathered one dword per lane (one cache line per ~40 useful bytes; the vector L1 charges
// ~4 cycles per line an instruction touches, profiles/r02_bf16_training.md). Here a patch row is the tile's 64 aligned pixels
// = two whole lines, loaded as float4 by 16 consecutive lanes (4 rows = 8 lines per instruction), plus one dword each side;
// the halo is 66 x 18 / (64 x 16) = 1.16x. A thread computes four horizontally adjacent outputs: 18 staged values per channel
// (one ds_read_b128 + two ds_read_b32 per row) instead of 36. Same arithmetic and summation order per output as above.
constexpr int SWP = 72;                          // LDS row pitch: [3] = left halo, [4..67] = the 64 pixels (16-byte aligned), [68] = right halo
template <int COUT>
__global__ void __launch_bounds__(256) conv_small_cout_wide_kernel(const float* __restrict__ x, int Cin, int N, int H,
                                                                   int W, const float* __restrict__ gn_scale,
                                                                   const float* __restrict__ gn_shift, int act,
                                                                   const float* __restrict__ wp,  // [Cin][tap][Cout]
                                                                   const float* __restrict__ bias,
                                                                   const float* __restrict__ residual,
                                                                   float* __restrict__ out, int tiles_x) {
  __shared__ __attribute__((aligned(16))) float tile[SC][18][SWP];
  const int tid = threadIdx.x;
  int wg = blockIdx.y * gridDim.x + blockIdx.x;
  const int total = gridDim.x * gridDim.y;
  if ((total & 7) == 0) wg = (wg & 7) * (total >> 3) + (wg >> 3);      // an XCD walks neighbouring tiles (shared halo rows)
  const int n = wg / (int)gridDim.x, tl = wg - n * (int)gridDim.x;
  const int tx = tl % tiles_x, ty = tl / tiles_x;
  const int lx4 = tid & 15, ly = tid >> 4;
  const long HW = (long)H * W;
  const int gx0 = tx * 64, gy0 = ty * 16 - 1;
  // staging plan: 9 float4 items (channel, patch row, 4-pixel group) per thread and chunk + the halo columns (288 items)
  int m_off[9], m_c[9], m_lds[9];
  ADM_UNROLL
  for (int k = 0; k < 9; ++k) {
    const int id = tid + 256 * k, x4 = id & 15, r = id >> 4, yy = r % 18, c = r / 18;
    const int gy = gy0 + yy;
    m_c[k] = c;
    m_lds[k] = (c * 18 + yy) * SWP + 4 + 4 * x4;
    m_off[k] = (gy >= 0 && gy < H) ? gy * W + gx0 + 4 * x4 : -1;
  }
  int e_off[2], e_c[2], e_lds[2];
  ADM_UNROLL
  for (int k = 0; k < 2; ++k) {
    const int id = tid + 256 * k;
    e_off[k] = -2;                                   // -2: not mine, -1: zero padding
    e_c[k] = 0; e_lds[k] = 0;
    if (id < SC * 18 * 2) {
      const int side = id & 1, r = id >> 1, yy = r % 18, c = r / 18;
      const int gy = gy0 + yy, gx = side ? gx0 + 64 : gx0 - 1;
      e_c[k] = c;
      e_lds[k] = (c * 18 + yy) * SWP + (side ? 68 : 3);
      e_off[k] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? gy * W + gx : -1;
    }
  }
  float acc[4][COUT];
  ADM_UNROLL
  for (int px = 0; px < 4; ++px)
    ADM_UNROLL
    for (int co = 0; co < COUT; ++co) acc[px][co] = 0.f;
  float* tl0 = &tile[0][0][0];
  // software pipeline: the raw values of chunk c0 + SC are requested before the FMAs of chunk c0 run (36 + 2 registers); the
  // first version loaded, activated, stored and computed chunk by chunk and sat at 2.0 TB/s with the load latency exposed
  float4 mr[9];
  float er[2];
  auto fetch = [&](int c0) __attribute__((always_inline)) {
    ADM_UNROLL
    for (int k = 0; k < 9; ++k) {
      const int c = c0 + m_c[k];
      mr[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m_off[k] >= 0 && c < Cin) mr[k] = *reinterpret_cast<const float4*>(x + ((long)n * Cin + c) * HW + m_off[k]);
    }
    ADM_UNROLL
    for (int k = 0; k < 2; ++k) {
      const int c = c0 + e_c[k];
      er[k] = 0.f;
      if (e_off[k] >= 0 && c < Cin) er[k] = x[((long)n * Cin + c) * HW + e_off[k]];
    }
  };
  fetch(0);
  for (int c0 = 0; c0 < Cin; c0 += SC) {
    ADM_UNROLL
    for (int k = 0; k < 9; ++k) {
      const int c = c0 + m_c[k];
      float4 v = mr[k];
      if (m_off[k] >= 0 && c < Cin) {
        if (gn_scale) {
          const float sc = gn_scale[(long)n * Cin + c], sh = gn_shift[(long)n * Cin + c];
          v.x = v.x * sc + sh; v.y = v.y * sc + sh; v.z = v.z * sc + sh; v.w = v.w * sc + sh;
        }
        if (act) {
          v.x = __fdividef(v.x, 1.0f + __expf(-v.x)); v.y = __fdividef(v.y, 1.0f + __expf(-v.y));
          v.z = __fdividef(v.z, 1.0f + __expf(-v.z)); v.w = __fdividef(v.w, 1.0f + __expf(-v.w));
        }
      }
      *reinterpret_cast<float4*>(tl0 + m_lds[k]) = v;
    }
    ADM_UNROLL
    for (int k = 0; k < 2; ++k) {
      if (e_off[k] != -2) {
        const int c = c0 + e_c[k];
        float v = er[k];
        if (e_off[k] >= 0 && c < Cin) {
          if (gn_scale) v = v * gn_scale[(long)n * Cin + c] + gn_shift[(long)n * Cin + c];
          if (act) v = __fdividef(v, 1.0f + __expf(-v));
        }
        tl0[e_lds[k]] = v;
      }
    }
    __syncthreads();
    if (c0 + SC < Cin) fetch(c0 + SC);
    ADM_UNROLL
    for (int c = 0; c < SC; ++c) {
      if (c0 + c < Cin) {
        const float* wr = wp + (long)(c0 + c) * 9 * COUT;  // uniform address -> scalar loads
        float v[3][6];
        ADM_UNROLL
        for (int dy = 0; dy < 3; ++dy) {
          const float* row = &tile[c][ly + dy][4 * lx4 + 3];
          const float4 m = *reinterpret_cast<const float4*>(row + 1);
          v[dy][0] = row[0]; v[dy][1] = m.x; v[dy][2] = m.y; v[dy][3] = m.z; v[dy][4] = m.w; v[dy][5] = row[5];
        }
        ADM_UNROLL
        for (int t = 0; t < 9; ++t)
          ADM_UNROLL
          for (int px = 0; px < 4; ++px)
            ADM_UNROLL
            for (int co = 0; co < COUT; ++co) acc[px][co] = fmaf(wr[t * COUT + co], v[t / 3][px + t % 3], acc[px][co]);
      }
    }
    __syncthreads();
  }
  const int oy = ty * 16 + ly, ox = gx0 + 4 * lx4;
  ADM_UNROLL
  for (int co = 0; co < COUT; ++co) {
    const long o = ((long)n * COUT + co) * HW + (long)oy * W + ox;
    const float b = bias ? bias[co] : 0.f;
    float4 r = make_float4(acc[0][co] + b, acc[1][co] + b, acc[2][co] + b, acc[3][co] + b);
    if (residual) {
      const float4 q = *reinterpret_cast<const float4*>(residual + o);
      r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
    }
    *reinterpret_cast<float4*>(out + o) = r;
  }
}

static bool use_wide_cin() {    // ADM_CONV_IN_WIDE=0 keeps the one-pixel-per-thread kernel (A/B timing)
  static const int v = [] { const char* e = getenv("ADM_CONV_IN_WIDE"); return e ? atoi(e) : 1; }();
  return v != 0;
}
// statistic tiles per (sample, cout) of the conv_in class kernel launch_conv_small would run for `a` (0: no epilogue)
int conv_small_stats_tiles(const adm_conv_args& a) {
  if (a.C1 > 4 || (a.x2 && a.C2) || a.ks != 3 || a.stride != 1 || a.up || a.W % 4 != 0 || !use_wide_cin()) return 0;
  if (a.gn_scale || a.act || a.chan_add) return 0;
  return (int)(((long)a.H * a.W / 4 + 255) / 256) * 4;
}
static bool use_wide_cout() {   // ADM_CONV_OUT_WIDE=0 keeps the 16x16 kernel (A/B timing)
  static const int v = [] { const char* e = getenv("ADM_CONV_OUT_WIDE"); return e ? atoi(e) : 1; }();
  return v != 0;
}

int launch_conv_small(const adm_conv_args& a, hipStream_t st) {
  ADM_REQUIRE(a.x2 == nullptr || a.C2 == 0, "conv_small: virtual concat not supported");
  ADM_REQUIRE(a.stride == 1 && !a.up, "conv_small: stride 1, no upsample only");
  ADM_REQUIRE(a.chan_add == nullptr, "conv_small: chan_add not supported");
  set_last_conv_variant(a.C1 <= 4 ? 1001 : 1002);
  if (a.C1 <= 4) {
    ADM_REQUIRE(a.gn_scale == nullptr && !a.act, "conv_small(cin): no fused norm/activation");
    ADM_REQUIRE(a.ks == 3 || a.ks == 1, "conv_small: ks");
    const long HW = (long)a.H * a.W;
    if (a.ks == 3 && a.W % 4 == 0 && use_wide_cin()) {     // four pixels per thread; the only variant with the statistics epilogue
      dim3 gw((unsigned)((HW / 4 + 255) / 256), a.N, 1);
      while ((long)gw.x * gw.y * gw.z < 256 && gw.z < 16 && a.Cout / (int)(2 * gw.z) >= 8) gw.z *= 2;
      ADM_REQUIRE(a.stats_out == nullptr || a.stats_tiles == (int)gw.x * 4, "conv_small(cin): stats_tiles mismatch");
#define ADM_CINW_CASE(CI)                                                                                          \
  if (a.C1 == CI) {                                                                                                \
    ADM_LAUNCH((conv_small_cin_wide_kernel<CI>), gw, dim3(256), 0, st, a.x1, a.N, a.H, a.W, a.wpacked, a.bias,      \
               a.Cout, a.residual, a.out, a.stats_out, a.stats_tiles);                                             \
    return ADM_CHECK_LAUNCH();                                                                                     \
  }
      ADM_CINW_CASE(1) ADM_CINW_CASE(2) ADM_CINW_CASE(3) ADM_CINW_CASE(4)
#undef ADM_CINW_CASE
    }
    ADM_REQUIRE(a.stats_out == nullptr, "conv_small(cin): this variant has no statistics epilogue");
    dim3 grid((unsigned)((HW + 255) / 256), a.N), block(256);
#define ADM_CIN_CASE(CI, KK)                                                                                       \
  if (a.C1 == CI && a.ks == KK) {                                                                                  \
    ADM_LAUNCH((conv_small_cin_kernel<CI, KK>), grid, block, 0, st, a.x1, a.N, a.H, a.W, a.wpacked, a.bias, a.Cout, \
               a.residual, a.out);                                                                                 \
    return ADM_CHECK_LAUNCH();                                                                                     \
  }
    ADM_CIN_CASE(1, 3) ADM_CIN_CASE(2, 3) ADM_CIN_CASE(3, 3) ADM_CIN_CASE(4, 3)
    ADM_CIN_CASE(1, 1) ADM_CIN_CASE(2, 1) ADM_CIN_CASE(3, 1) ADM_CIN_CASE(4, 1)
#undef ADM_CIN_CASE
    ADM_FAIL("conv_small(cin): unsupported (Cin, ks)");
  }
  ADM_REQUIRE(a.Cout <= 4 && a.ks == 3 && a.pad_lo == 1, "conv_small: unsupported shape (need Cin<=4 or Cout<=4, 3x3)");
  // the single-sample rule ("single_sample", by model): the channel split below on planes of any size — one 256x256 sample is 64 wide tiles,
  // each a serial walk over all input channels (93 us; 64x64: 4 tiles, 97 us)
  const bool single_split = conv_single_sample(a) && a.C1 >= 64;
  if (a.W % 64 == 0 && a.H % 16 == 0 && use_wide_cout() && !single_split) {   // whole 64 x 16 tiles: the wide kernel (16-byte aligned rows)
    const int wx = a.W / 64, wy = a.H / 16;
#define ADM_COUTW_CASE(CO)                                                                                              \
  if (a.Cout == CO) {                                                                                                   \
    ADM_LAUNCH((conv_small_cout_wide_kernel<CO>), dim3(wx * wy, a.N), dim3(256), 0, st, a.x1, a.C1, a.N, a.H, a.W,      \
               a.gn_scale, a.gn_shift, a.act, a.wpacked, a.bias, a.residual, a.out, wx);                                \
    return ADM_CHECK_LAUNCH();                                                                                          \
  }
    ADM_COUTW_CASE(1) ADM_COUTW_CASE(2) ADM_COUTW_CASE(3) ADM_COUTW_CASE(4)
#undef ADM_COUTW_CASE
  }
  const int tiles_x = ceil_div(a.W, 16), tiles_y = ceil_div(a.H, 16);
  // planes of at most 32x32 pixels (a function of the layer, not of the batch: the partition fixes the summation order): the channel
  // loop split over 8 workgroups per tile
  int S = 1;
  float* part = nullptr;
  const long total = (long)a.N * a.Cout * a.H * a.W;
  static const int use_ksp = [] { const char* e = getenv("ADM_CONV_KSPLIT"); return e ? atoi(e) : 1; }();
  if (use_ksp && ((long)a.H * a.W <= 1024 || single_split) && a.C1 >= 64) {
    part = conv_ksplit_scratch((size_t)8 * total, st);
    if (part == nullptr) return -1;       // (error recorded) the unsplit kernel sums in another order
    S = 8;
  }
#define ADM_COUT_CASE(CO)                                                                                          \
  if (a.Cout == CO) {                                                                                              \
    ADM_LAUNCH((conv_small_cout_kernel<CO>), dim3(tiles_x * tiles_y, a.N, S), dim3(256), 0, st, a.x1, a.C1, a.N, a.H, \
               a.W, a.gn_scale, a.gn_shift, a.act, a.wpacked, a.bias, a.residual, a.out, tiles_x, part, total);   \
    if (part != nullptr)                                                                                           \
      return launch_ksplit_finish(part, S, total, a.bias, nullptr, 0, a.residual, a.out, a.Cout, a.H * a.W, st);   \
    return ADM_CHECK_LAUNCH();                                                                                     \
  }
  ADM_COUT_CASE(1) ADM_COUT_CASE(2) ADM_COUT_CASE(3) ADM_COUT_CASE(4)
#undef ADM_COUT_CASE
  ADM_FAIL("conv_small(cout): unsupported Cout");
}

}  // namespace adm
