// k_conv_bf16_persist8.hip — the persistent chunk-stream kernel (k_conv_bf16_persist.hip) with 8 waves, two per SIMD
// (k_conv_bf16w8.hip): both opt-in ideas combined.  OPT-IN (ADM_BF16_PERSIST=2 / option "conv_bf16_persist" = 2), emulator
// parity only.  With 64 accumulator registers per wave the stream's extra state fits without AGPR spills.
// CAUTION: the register allocation of this kernel is fragile with ROCm 7.2 — small source changes (a compile-time patch
// index, a parity-indexed LDS constant set) flip it between 0 and ~125 spilled registers inside the tile loop, which makes
// every scratch reload a vmcnt(0) drain.  Re-check `-Rpass-analysis=kernel-resource-usage` after ANY edit before timing.
#include "adm_kernels.h"

namespace adm {

struct Bf16ConvParamsP8 {
  const float* x1; const float* x2; int C1, C2;
  int N, Hs, Ws, Hi, Wi;
  const float* gn_scale; const float* gn_shift; int gn_nstride;
  const u32x4* wb; const float* bias; int Cout;
  const float* chan_add; int chan_add_stride;
  const float* residual; float* out;
  int tiles_x, tiles_y, n_ct, nblk, tiles_per_wg;
  long x1_bs, x2_bs;
  int upshift;   // 1 with the nearest-x2 fold, else 0 — a RUNTIME value on purpose: with the compile-time form of the
                 // non-upsampled index the register allocator spilled ~100 registers into the tile loop (gfx950, ROCm 7.2)
};

constexpr int RPW = 18, RPP = RPW * RPW;

__device__ __forceinline__ float silu_p8(float v) { return v * ADM_RCP(1.0f + __expf(-v)); }

struct P8Stage { float v[2][8]; };
struct P8Filt { u32x4 a[9][2]; };
struct P8Tile {            // decoded tile: uniform fields + this lane's two patch offsets (0xFFFFFFFF = outside the image)
  int n, m0, ty, tx;
  unsigned so0, so1;
  bool valid;
};

template <bool UP, bool ACT, bool RES>
__global__ void __launch_bounds__(512, 1) conv_bf16p8_kernel(const Bf16ConvParamsP8 p) {
  ADM_DYN_SMEM(u32x4, lds);                 // [2 buffers][2 channel groups][324 pixels] + GroupNorm rows [2 sets][2][Ct]
  const int tid = threadIdx.x, lane = tid & 63, wave = ADM_UNIFORM(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  const int Ct = p.C1 + p.C2, KG = Ct >> 3, n_chunks = Ct >> 4;
  const int planeS = p.Hs * p.Ws;
  float* gnrows = reinterpret_cast<float*>(lds + 4 * RPP);          // set s: scale at gnrows + s*2*Ct, shift at + Ct
  float* ebias = gnrows + 4 * Ct;                                     // [128] epilogue constants of the current tile
  const int t_first = blockIdx.x * p.tiles_per_wg;
  int t_last = t_first + p.tiles_per_wg;
  if (t_last > p.nblk) t_last = p.nblk;
  if (t_first >= t_last) return;
  // staging items: id = tid + 512 r (r = 0, 1) over the 2 x 324 (channel group, patch pixel) items (tile-invariant part)
  const int id0 = tid, id1 = tid + 512;          // id1 < 648 only for the first 136 threads
  const int skg0 = id0 >= RPP ? 1 : 0, skg1 = 1;
  const int spx0 = id0 - skg0 * RPP, spx1 = id1 < 2 * RPP ? id1 - RPP : -1;

  auto decode = [&](int t) __attribute__((always_inline)) {
    P8Tile d;
    d.valid = t < t_last;
    int lid = d.valid ? t : t_last - 1;
    const int ct = lid % p.n_ct; lid /= p.n_ct;
    d.tx = lid % p.tiles_x; lid /= p.tiles_x;
    d.ty = lid % p.tiles_y; d.n = lid / p.tiles_y;
    d.m0 = ct * 128 + wm * 64;
    unsigned so[2];
    ADM_UNROLL
    for (int r = 0; r < 2; ++r) {
      const int sp_r = r == 0 ? spx0 : spx1;
      const int q = sp_r < 0 ? 0 : sp_r;
      const int ly = q / RPW, lx = q - ly * RPW;
      const int gy = d.ty * 16 + ly - 1, gx = d.tx * 16 + lx - 1;
      const bool ok = (sp_r >= 0) & (gy >= 0) & (gy < p.Hi) & (gx >= 0) & (gx < p.Wi);
      so[r] = ok ? (unsigned)((gy >> p.upshift) * p.Ws + (gx >> p.upshift)) : 0xFFFFFFFFu;   // runtime shift: see launcher
    }
    d.so0 = so[0]; d.so1 = so[1];
    return d;
  };
  auto fill_rows = [&](const P8Tile& d, int set) __attribute__((always_inline)) {
    float* gs = gnrows + set * 2 * Ct;
    for (int c = tid; c < Ct; c += 512) {
      gs[c] = p.gn_scale[(long)d.n * p.gn_nstride + c];
      gs[Ct + c] = p.gn_shift[(long)d.n * p.gn_nstride + c];
    }
  };

  P8Tile cur = decode(t_first), nxt = decode(t_first + 1);
  int seq = 0;                                   // tiles done by this workgroup: GroupNorm row set of `cur` = seq & 1

  // stream index g: chunk g of `cur` for g < n_chunks, chunk g - n_chunks of `nxt` beyond (clamped when there is no next tile).
  // Fields are selected one by one (scalar / v_cndmask selects): choosing between the two structs by reference made the
  // compiler put them in scratch memory, whose loads drain the vector-memory counter.
  auto issue = [&](P8Stage& s, int g) __attribute__((always_inline)) {
    const bool nx = g >= n_chunks, un = nx & nxt.valid;
    const int dn = un ? nxt.n : cur.n;
    const unsigned r0 = un ? nxt.so0 : cur.so0, r1 = un ? nxt.so1 : cur.so1;
    const int ch = nx ? (nxt.valid ? g - n_chunks : n_chunks - 1) : g;
    const int c0 = 16 * ch;
    const float* xc = c0 < p.C1 ? p.x1 + (long)dn * p.x1_bs + (long)c0 * planeS
                                : p.x2 + (long)dn * p.x2_bs + (long)(c0 - p.C1) * planeS;
    const unsigned a0 = r0 == 0xFFFFFFFFu ? 0u : r0, a1 = r1 == 0xFFFFFFFFu ? 0u : r1;
    ADM_UNROLL
    for (int r = 0; r < 2; ++r) {
      unsigned so = (r == 0 ? a0 : a1) + (unsigned)((r == 0 ? skg0 : skg1) * 8 * planeS);
      ADM_OPAQUE_V(so);
      ADM_UNROLL
      for (int e = 0; e < 8; ++e) s.v[r][e] = (xc + (long)e * planeS)[so];
    }
  };
  auto stash = [&](const P8Stage& s, u32x4* buf, int g) __attribute__((always_inline)) {
    const bool nx = g >= n_chunks;
    if (nx && !nxt.valid) return;
    const unsigned r0 = nx ? nxt.so0 : cur.so0, r1 = nx ? nxt.so1 : cur.so1;
    const int c0 = 16 * (nx ? g - n_chunks : g);
    const float* rows = gnrows + ((seq + (nx ? 1 : 0)) & 1) * 2 * Ct;
    ADM_UNROLL
    for (int r = 0; r < 2; ++r) {
      const int kg = r == 0 ? skg0 : skg1;
      const bool outside = (r == 0 ? r0 : r1) == 0xFFFFFFFFu;
      const float* gsp = rows + c0 + kg * 8;
      const float4 s0 = *reinterpret_cast<const float4*>(gsp), s1 = *reinterpret_cast<const float4*>(gsp + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(gsp + Ct), b1 = *reinterpret_cast<const float4*>(gsp + Ct + 4);
      const float gs[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      const float gb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float v[8];
      ADM_UNROLL
      for (int e = 0; e < 8; ++e) {
        float t = s.v[r][e] * gs[e] + gb[e];
        if (ACT) t = silu_p8(t);
        v[e] = outside ? 0.f : t;
      }
      u32x4 w;
      w[0] = ADM_PK_BF16(v[0], v[1]); w[1] = ADM_PK_BF16(v[2], v[3]);
      w[2] = ADM_PK_BF16(v[4], v[5]); w[3] = ADM_PK_BF16(v[6], v[7]);
      if ((r == 0 ? spx0 : spx1) >= 0) buf[kg * RPP + (r == 0 ? spx0 : spx1)] = w;
      ADM_SCHED_FENCE();
    }
  };
  const unsigned wlane = (unsigned)(h * p.Cout + l31);
  auto fetch_tap = [&](P8Filt& f, int g, int t) __attribute__((always_inline)) {
    const bool nx = g >= n_chunks, un = nx & nxt.valid;
    const int m0 = un ? nxt.m0 : cur.m0;
    const int ch = nx ? (nxt.valid ? g - n_chunks : n_chunks - 1) : g;
    const u32x4* wt = p.wb + m0 + ((long)(2 * ch) + (long)t * KG) * p.Cout;
    f.a[t][0] = wt[wlane]; f.a[t][1] = (wt + 32)[wlane];
  };

  f32x16 acc[2][2];
  const int bbase = h * RPP + (4 * wn + (l31 >> 4)) * RPW + (l31 & 15);
  P8Filt F;
  auto mfma_chunk = [&](const u32x4* bufc, int ch) __attribute__((always_inline)) {
    u32x4 Bc[2], Bn[2];
    ADM_UNROLL
    for (int pt = 0; pt < 2; ++pt) Bc[pt] = bufc[bbase + (2 * pt) * RPW];
    ADM_UNROLL
    for (int t = 0; t < 9; ++t) {
      if (t < 8) {
        ADM_UNROLL
        for (int pt = 0; pt < 2; ++pt) Bn[pt] = bufc[bbase + (2 * pt + (t + 1) / 3) * RPW + ((t + 1) % 3)];
      }
      ADM_SCHED_FENCE();
      ADM_UNROLL
      for (int pt = 0; pt < 2; ++pt) {
        acc[0][pt] = ADM_MFMA_BF16(F.a[t][0], Bc[pt], acc[0][pt]);
        acc[1][pt] = ADM_MFMA_BF16(F.a[t][1], Bc[pt], acc[1][pt]);
      }
      ADM_SCHED_FENCE();
      fetch_tap(F, ch + 1, t);
      ADM_UNROLL
      for (int pt = 0; pt < 2; ++pt) Bc[pt] = Bn[pt];
    }
  };

  P8Stage X, Y;
  u32x4* buf0 = lds;
  u32x4* buf1 = lds + 2 * RPP;
  // prime the stream (first tile only)
  fill_rows(cur, 0);
  issue(X, 0);
  ADM_UNROLL
  for (int t = 0; t < 9; ++t) fetch_tap(F, 0, t);
  issue(Y, 1);
  __syncthreads();
  stash(X, buf0, 0);
  issue(X, 2);
  __syncthreads();
  const long planeO = (long)p.Hi * p.Wi;
  for (int tile = t_first; tile < t_last; ++tile) {
    // GroupNorm rows of the next tile: requested now, written to LDS after the first MFMA chunk below (waiting for them
    // here would drain the patch / filter loads in flight: vector memory returns in order), published by the barriers of
    // the chunk loop long before the next tile's first conversion (n_chunks >= 4)
    float nrs[4], nrb[4];
    ADM_UNROLL
    for (int k = 0; k < 4; ++k) {
      const int c = tid + 512 * k;
      const long gi = (long)nxt.n * p.gn_nstride + (c < Ct ? c : 0);
      nrs[k] = p.gn_scale[gi]; nrb[k] = p.gn_shift[gi];
    }
    // ... and this tile's epilogue constants (bias + per-sample channel term of its 128 couts) take the same route, so that
    // the epilogue itself issues no vector load unless the convolution has a residual input
    const int eco = (cur.m0 - 64 * wm) + (tid & 127);
    const float eb0 = p.bias[eco], eb1 = p.chan_add[(long)cur.n * p.chan_add_stride + eco];   // summed at store time
    ADM_UNROLL
    for (int a = 0; a < 2; ++a)
      ADM_UNROLL
      for (int t = 0; t < 2; ++t)
        ADM_UNROLL
        for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.f;
    for (int ch = 0; ch < n_chunks; ch += 2) {
      mfma_chunk(buf0, ch);
      if (ch == 0) {
        __syncthreads();          // every wave has left the previous tile's epilogue (which reads ebias) before it is rewritten
        if (tid < 128) ebias[tid] = eb0 + eb1;
        if (nxt.valid) {
          float* gs = gnrows + ((seq + 1) & 1) * 2 * Ct;
          ADM_UNROLL
          for (int k = 0; k < 4; ++k) {
            const int c = tid + 512 * k;
            if (c < Ct) { gs[c] = nrs[k]; gs[Ct + c] = nrb[k]; }
          }
        }
      }
      stash(Y, buf1, ch + 1);
      issue(Y, ch + 3);
      __syncthreads();
      mfma_chunk(buf1, ch + 1);
      stash(X, buf0, ch + 2);
      if (ch + 2 < n_chunks) issue(X, ch + 4);      // the last one (next tile's chunk 2) is issued after the epilogue: a
      __syncthreads();                              // residual load there would otherwise wait for it (in-order return)
    }
    // epilogue of `cur`; the next tile's chunk 0 sits in buf0, its chunks 1 / 2 are in flight, its filters requested
    ADM_UNROLL
    for (int a = 0; a < 2; ++a) {
      const float* eb = ebias + 64 * wm + 32 * a + 4 * h;          // read per element below: 16 registers fewer live
      ADM_UNROLL
      for (int pt = 0; pt < 2; ++pt) {
        const int oy = cur.ty * 16 + 4 * wn + 2 * pt + (l31 >> 4), ox = cur.tx * 16 + (l31 & 15);
        const long pix = (long)oy * p.Wi + ox;
        ADM_UNROLL
        for (int half = 0; half < 2; ++half) {      // residual in two batches of 8: the stream's state leaves little room
          float rv[8];
          if (RES) {
            ADM_UNROLL
            for (int q = 0; q < 8; ++q) {
              const int r = 8 * half + q;
              const int co = cur.m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
              rv[q] = p.residual[((long)cur.n * p.Cout + co) * planeO + pix];
            }
          }
          ADM_UNROLL
          for (int q = 0; q < 8; ++q) {
            const int r = 8 * half + q;
            const int co = cur.m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
            float v = acc[a][pt][r] + eb[(r & 3) + 8 * (r >> 2)];
            if (RES) v += rv[q];
            p.out[((long)cur.n * p.Cout + co) * planeO + pix] = v;
          }
        }
      }
    }
    cur = nxt;
    nxt = decode(tile + 2);
    ++seq;
    if (cur.valid) issue(X, 2);
  }
}

// launched by launch_conv_bf16 (k_conv_bf16.hip) when enabled and Cin >= 64
int launch_conv_bf16_persist8(const adm_conv_args& a, hipStream_t st) {
  Bf16ConvParamsP8 p;
  const int C2 = a.x2 ? a.C2 : 0, Ct = a.C1 + C2;
  p.x1 = a.x1; p.x2 = a.x2; p.C1 = a.C1; p.C2 = C2;
  p.N = a.N; p.Hs = a.H; p.Ws = a.W;
  p.Hi = a.up ? 2 * a.H : a.H; p.Wi = a.up ? 2 * a.W : a.W;
  p.gn_scale = a.gn_scale; p.gn_shift = a.gn_shift; p.gn_nstride = Ct;
  if (p.gn_scale == nullptr) { p.gn_scale = conv_const_ones(Ct); p.gn_shift = conv_zero_bias(Ct); p.gn_nstride = 0; }
  p.wb = reinterpret_cast<const u32x4*>(a.bf16_packed);
  p.bias = a.bias ? a.bias : conv_zero_bias(a.Cout); p.Cout = a.Cout;
  p.chan_add = a.chan_add; p.chan_add_stride = a.chan_add_stride;
  if (p.chan_add == nullptr) { p.chan_add = conv_zero_bias(a.Cout); p.chan_add_stride = 0; }
  ADM_REQUIRE(p.gn_scale && p.gn_shift && p.bias && p.chan_add, "conv_bf16: constant buffers");
  p.residual = a.residual; p.out = a.out;
  p.tiles_x = p.Wi / 16; p.tiles_y = p.Hi / 16; p.n_ct = a.Cout / 128;
  p.nblk = p.tiles_x * p.tiles_y * a.N * p.n_ct;
  p.x1_bs = a.x1_bstride ? a.x1_bstride : (long)a.C1 * a.H * a.W;
  p.x2_bs = a.x2_bstride ? a.x2_bstride : (long)C2 * a.H * a.W;
  p.upshift = a.up ? 1 : 0;
#if !defined(ADM_EMU)
  static int n_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
#else
  const int n_cu = 3;                       // several tiles per workgroup on the emulator
#endif
  int grid = p.nblk < n_cu ? p.nblk : n_cu;
  p.tiles_per_wg = ceil_div(p.nblk, grid);
  grid = ceil_div(p.nblk, p.tiles_per_wg);
  const size_t smem = sizeof(u32x4) * 2 * 2 * RPP + sizeof(float) * (4 * Ct + 256);
  ADM_REQUIRE(smem <= 64 * 1024 && Ct <= 1024, "conv_bf16: too many input channels for the LDS GroupNorm rows");
  set_last_conv_variant(5000 + 319);
#define ADM_BF16P8_LAUNCH(UP_, ACT_)                                                                            \
  do {                                                                                                         \
    if (a.residual) ADM_LAUNCH((conv_bf16p8_kernel<UP_, ACT_, true>), dim3(grid), dim3(512), smem, st, p);      \
    else ADM_LAUNCH((conv_bf16p8_kernel<UP_, ACT_, false>), dim3(grid), dim3(512), smem, st, p);                \
  } while (0)
  if (a.up) {
    if (a.act) ADM_BF16P8_LAUNCH(true, true);
    else ADM_BF16P8_LAUNCH(true, false);
  } else {
    if (a.act) ADM_BF16P8_LAUNCH(false, true);
    else ADM_BF16P8_LAUNCH(false, false);
  }
#undef ADM_BF16P8_LAUNCH
  return ADM_CHECK_LAUNCH();
}

}  // namespace adm
