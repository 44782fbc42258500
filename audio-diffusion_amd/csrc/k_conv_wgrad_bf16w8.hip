// k_conv_wgrad_bf16w8.hip — 8-wave variant of the bf16 3x3 weight-gradient kernel (k_conv_bf16.hip).  OPT-IN
// (ADM_WGRAD_BF16_8W=1 / option "wgrad_bf16_8w"): written at the end of round 1 after the GPU budget was spent — parity-checked
// on the emulator only.  Motivation (profiles/r01_pmc_bf16.md): the 4-wave kernel keeps the matrix pipe 16 % busy because a
// single wave per SIMD has to issue ~500 conversion / LDS / address instructions per 36 MFMAs.
#include "adm_kernels.h"

namespace adm {

__device__ __forceinline__ float silu_w8(float v) { return v * ADM_RCP(1.0f + __expf(-v)); }

struct Bf16Wgrad8Params {
  const float* x1; const float* x2; int C1, C2;
  const float* dy; int Cout;
  int N, Hs, Ws, Hi, Wi;
  const float* gn_scale; const float* gn_shift; int gn_nstride;
  float* part;
  int tiles_x, tiles_y, n_ptiles, n_ct, n_chunks, split, tiles_per_block, nblk;
  long x1_bs, x2_bs;
  unsigned mTX, mTXY;       // floor(2^32 / d) + 1 for d = tiles_x, tiles_x * tiles_y (0: d == 1 or tile count >= 2^16)
};

__device__ __forceinline__ int bdiv8(int n, int d, unsigned magic) {   // n / d; exact via umulhi for n, d < 2^16
  return magic ? (int)(((unsigned long long)(unsigned)n * magic) >> 32) : n / d;
}

// raw fp32 prefetch of one 16x4-pixel tile: 4 dy items (8 pixels each), 7 patch pixel pairs, their in-bounds bits, image
struct Bf16Wg8Stage { float4 d[2][2]; float xa[4], xb[4]; unsigned ok; int n; };

template <bool UP, bool ACT>
__global__ void __launch_bounds__(512, 1) conv_wgrad_bf16w8_kernel(const Bf16Wgrad8Params p) {
  // 8 waves = two groups of 4: both groups share the staged tile (each of the 512 threads converts half as much as in the
  // 4-wave kernel), group g multiplies pixel rows 2g, 2g+1 of the 16x4 tile into its OWN 9 accumulator tiles and writes its
  // own partial-sum slab (2 sp + g): the two waves of a SIMD are independent instruction streams, so one converts while
  // the other's MFMAs run (the 4-wave kernel issues ~500 instructions per 36 MFMAs from a single wave per SIMD).
  constexpr int XROW = 12;                       // dwords per (row, cin) of the patch
  constexpr int DLD = 130;                       // fragment stride of a (row, half) line of 128 couts: 130 makes the 16 lanes of a
                                                 // b128 store group (2 couts x 8 (row, half)) hit 16 distinct 16-B slots (128: 8-way)
  constexpr int DFR = 8 * DLD;                   // dy fragments (u32x4) per buffer
  constexpr int XDW = 6 * 32 * XROW;             // patch dwords per buffer
  constexpr int BUF4 = DFR + XDW / 4;            // u32x4 per buffer
  ADM_DYN_SMEM(u32x4, lds4);
  unsigned* dummy = reinterpret_cast<unsigned*>(lds4 + 2 * BUF4);          // 512 dwords: disabled lanes store here
  float* gnS = reinterpret_cast<float*>(dummy + 512);                       // [N][32] scale, then [N][32] shift
  float* gnB = gnS + p.N * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = ADM_UNIFORM(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wq = wave & 3, grp = wave >> 2;      // cout quarter, pixel-row group
  int lid;
  {   // the n_chunks workgroups that read the same dy tiles are neighbours on one XCD
    const int b = blockIdx.x, q = p.nblk >> 3, r = p.nblk & 7, xcd = b & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int chunk = lid % p.n_chunks; lid /= p.n_chunks;
  const int ct = lid % p.n_ct, sp = lid / p.n_ct;
  const int m0 = ct * 128, c0 = chunk * 32;
  const int Ct = p.C1 + p.C2;
  const int planeS = p.Hs * p.Ws;
  const long planeO = (long)p.Hi * p.Wi;
  const float* xsrc = c0 < p.C1 ? p.x1 + (long)c0 * planeS : p.x2 + (long)(c0 - p.C1) * planeS;
  const long xbs = c0 < p.C1 ? p.x1_bs : p.x2_bs;
  const int t_begin = sp * p.tiles_per_block;
  int t_end = t_begin + p.tiles_per_block;
  if (t_end > p.n_ptiles) t_end = p.n_ptiles;

  for (int i = tid; i < p.N * 32; i += 512) {          // GroupNorm rows of this channel chunk, every image
    const long gi = (long)(i >> 5) * p.gn_nstride + c0 + (i & 31);
    gnS[i] = p.gn_scale[gi];
    gnB[i] = p.gn_shift[gi];
  }

  f32x16 acc[9];
  ADM_UNROLL
  for (int t = 0; t < 9; ++t)
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // tile-invariant staging roles.  dy: item = (cout, row, half), 2 per thread, 8 pixels (two float4) each.
  // patch: item = (row, cin, pixel pair), 1728 items in 4 rounds (the last one partial).
  unsigned dyo[2]; int ldsd[2];
  ADM_UNROLL
  for (int j = 0; j < 2; ++j) {
    const int id = tid + 512 * j;
    const int hh = id & 1, r = (id >> 1) & 3, co = id >> 3;
    dyo[j] = (unsigned)(co * (int)planeO + r * p.Wi + 8 * hh);
    ldsd[j] = (r * 2 + hh) * DLD + co;
  }
  int xcin[4], xrow[4], xq[4], ldsx[4];
  ADM_UNROLL
  for (int j = 0; j < 4; ++j) {
    const int id = tid + 512 * j;
    const int rc = id / 9;
    xq[j] = id - rc * 9; xrow[j] = rc >> 5; xcin[j] = rc & 31;
    ldsx[j] = xrow[j] < 6 ? (xrow[j] * 32 + xcin[j]) * XROW + xq[j] : -1;       // -1: past the end -> dummy word
  }

  auto load_tile = [&](Bf16Wg8Stage& s, int pt_raw) __attribute__((always_inline)) {
    const int pt = pt_raw < t_end ? pt_raw : t_end - 1;        // past the end: re-request the last tile (no branch)
    const int n = bdiv8(pt, p.tiles_x * p.tiles_y, p.mTXY);
    const int rem = pt - n * (p.tiles_x * p.tiles_y);
    const int ty = bdiv8(rem, p.tiles_x, p.mTX), tx = rem - ty * p.tiles_x;
    s.n = n;
    const float* dbase = p.dy + ((long)n * p.Cout + m0) * planeO + (long)(ty * 4) * p.Wi + tx * 16;   // uniform
    ADM_UNROLL
    for (int j = 0; j < 2; ++j) {
      s.d[j][0] = *reinterpret_cast<const float4*>(dbase + dyo[j]);
      s.d[j][1] = *reinterpret_cast<const float4*>(dbase + dyo[j] + 4);
    }
    const float* xt = xsrc + (long)n * xbs;                                                            // uniform
    const int gy0 = ty * 4 - 1, gx0 = tx * 16 - 1;
    unsigned ok = 0;
    ADM_UNROLL
    for (int j = 0; j < 4; ++j) {
      const int gy = gy0 + xrow[j], gx = gx0 + 2 * xq[j];
      const bool oky = (gy >= 0) & (gy < p.Hi) & (ldsx[j] >= 0);
      const bool ok0 = oky & (gx >= 0) & (gx < p.Wi), ok1 = oky & (gx + 1 < p.Wi);      // gx + 1 >= 0 always
      const int rowoff = xcin[j] * planeS + (UP ? (gy >> 1) : gy) * p.Ws;
      const unsigned o0 = ok0 ? (unsigned)(rowoff + (UP ? (gx >> 1) : gx)) : 0u;
      const unsigned o1 = ok1 ? (unsigned)(rowoff + (UP ? ((gx + 1) >> 1) : gx + 1)) : 0u;
      s.xa[j] = xt[o0]; s.xb[j] = xt[o1];
      ok |= ((ok0 ? 1u : 0u) | (ok1 ? 2u : 0u)) << (2 * j);
    }
    s.ok = ok;
  };
  auto stash_tile = [&](const Bf16Wg8Stage& s, u32x4* buf) __attribute__((always_inline)) {
    unsigned* bufX = reinterpret_cast<unsigned*>(buf + DFR);
    ADM_UNROLL
    for (int j = 0; j < 2; ++j) {
      const float4 v0 = s.d[j][0], v1 = s.d[j][1];
      u32x4 w;
      w[0] = ADM_PK_BF16(v0.x, v0.y); w[1] = ADM_PK_BF16(v0.z, v0.w);
      w[2] = ADM_PK_BF16(v1.x, v1.y); w[3] = ADM_PK_BF16(v1.z, v1.w);
      buf[ldsd[j]] = w;
    }
    ADM_UNROLL
    for (int j = 0; j < 4; ++j) {
      const float sc = gnS[s.n * 32 + xcin[j]], sh = gnB[s.n * 32 + xcin[j]];
      float a = s.xa[j] * sc + sh, b = s.xb[j] * sc + sh;
      if (ACT) { a = silu_w8(a); b = silu_w8(b); }
      a = (s.ok >> (2 * j)) & 1u ? a : 0.f;          // zero padding applies to the activated tensor
      b = (s.ok >> (2 * j)) & 2u ? b : 0.f;
      unsigned* dst = ldsx[j] >= 0 ? bufX + ldsx[j] : dummy + tid;
      *dst = ADM_PK_BF16(a, b);
    }
  };
  auto mfma_tile = [&](const u32x4* buf, bool valid) __attribute__((always_inline)) {
    const unsigned* bufX = reinterpret_cast<const unsigned*>(buf + DFR);
    ADM_UNROLL
    for (int rr = 0; rr < 2; ++rr) {
      const int r = 2 * grp + rr;
      u32x4 A = buf[(r * 2 + h) * DLD + 32 * wq + l31];
      if (!valid) { A[0] = 0u; A[1] = 0u; A[2] = 0u; A[3] = 0u; }       // tile past the end of an odd range: contributes zero
      ADM_UNROLL
      for (int dy3 = 0; dy3 < 3; ++dy3) {
        const unsigned* xr = bufX + ((r + dy3) * 32 + l31) * XROW + 4 * h;
        const u32x4 d = *reinterpret_cast<const u32x4*>(xr);
        const unsigned d4 = xr[4];
        u32x4 s1, s2;
        s1[0] = ADM_ALIGNBIT(d[1], d[0], 16); s1[1] = ADM_ALIGNBIT(d[2], d[1], 16);
        s1[2] = ADM_ALIGNBIT(d[3], d[2], 16); s1[3] = ADM_ALIGNBIT(d4, d[3], 16);
        s2[0] = d[1]; s2[1] = d[2]; s2[2] = d[3]; s2[3] = d4;
        acc[dy3 * 3 + 0] = ADM_MFMA_BF16(A, d, acc[dy3 * 3 + 0]);
        acc[dy3 * 3 + 1] = ADM_MFMA_BF16(A, s1, acc[dy3 * 3 + 1]);
        acc[dy3 * 3 + 2] = ADM_MFMA_BF16(A, s2, acc[dy3 * 3 + 2]);
      }
    }
  };

  // one prefetch register set (the 512-thread workgroup has 256 registers per wave; a second set spilled): tile t + 1 is in
  // registers while tile t multiplies, its conversion and the request for tile t + 2 follow the MFMAs; the partner wave of
  // the SIMD covers the remaining latency
  Bf16Wg8Stage P;
  u32x4* buf0 = lds4;
  u32x4* buf1 = lds4 + BUF4;
  load_tile(P, t_begin);
  __syncthreads();                         // GroupNorm rows are in LDS
  stash_tile(P, buf0);
  load_tile(P, t_begin + 1);
  __syncthreads();
  for (int pt = t_begin; pt < t_end; pt += 2) {
    mfma_tile(buf0, true);                 // tile pt; P holds pt + 1
    stash_tile(P, buf1);
    load_tile(P, pt + 2);
    __syncthreads();
    mfma_tile(buf1, pt + 1 < t_end);       // tile pt + 1; P holds pt + 2
    stash_tile(P, buf0);
    load_tile(P, pt + 3);
    __syncthreads();
  }
  float* out = p.part + (long)(2 * sp + grp) * p.Cout * Ct * 9;
  ADM_UNROLL
  for (int t = 0; t < 9; ++t) {
    const int cc = c0 + l31;
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int co = m0 + wq * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      out[((long)co * Ct + cc) * 9 + t] = acc[t][r];
    }
  }
}

static int g_bf16_mode = -1;   // -1: ADM_CONV_BF16 from the environment (default 0 = fp32 everywhere; 2 = 1x1 convs too)
bool conv_bf16_enabled();


static int g_w8 = -1;      // -1: ADM_WGRAD_BF16_8W from the environment (default 0)
void set_wgrad_bf16_8w(int v) { g_w8 = v; }
bool wgrad_bf16_8w_enabled() {
  if (g_w8 < 0) { const char* e = getenv("ADM_WGRAD_BF16_8W"); g_w8 = e ? atoi(e) : 0; }
  return g_w8 != 0;
}

// Same contract as launch_conv_wgrad_bf16 (partials in the workspace layout of conv_wgrad_workspace); returns the number
// of slabs written (2 per workgroup range), <= split; needs split >= 2.
int launch_conv_wgrad_bf16w8(const adm_conv_args& a, const float* dy, float* workspace, int split, hipStream_t st) {
  Bf16Wgrad8Params p;
  const int C2 = a.x2 ? a.C2 : 0, Ct = a.C1 + C2;
  p.x1 = a.x1; p.x2 = a.x2; p.C1 = a.C1; p.C2 = C2; p.dy = dy; p.Cout = a.Cout;
  p.N = a.N; p.Hs = a.H; p.Ws = a.W;
  p.Hi = a.up ? 2 * a.H : a.H; p.Wi = a.up ? 2 * a.W : a.W;
  ADM_REQUIRE((reinterpret_cast<uintptr_t>(dy) & 15) == 0, "conv_wgrad_bf16: dy must be 16-byte aligned");
  p.gn_scale = a.gn_scale; p.gn_shift = a.gn_shift; p.gn_nstride = Ct;
  if (p.gn_scale == nullptr) { p.gn_scale = conv_const_ones(Ct); p.gn_shift = conv_zero_bias(Ct); p.gn_nstride = 0; }
  ADM_REQUIRE(p.gn_scale && p.gn_shift, "conv_wgrad_bf16: constant buffers");
  p.part = workspace;
  p.tiles_x = p.Wi / 16; p.tiles_y = p.Hi / 4;
  p.n_ptiles = p.tiles_x * p.tiles_y * a.N;
  p.n_ct = a.Cout / 128; p.n_chunks = Ct / 32;
  int wsplit = split / 2;                                  // each workgroup range produces two slabs
  if (wsplit < 1) return -1;
  if (wsplit > p.n_ptiles) wsplit = p.n_ptiles;
  p.tiles_per_block = ceil_div(p.n_ptiles, wsplit);
  p.split = ceil_div(p.n_ptiles, p.tiles_per_block);
  p.nblk = p.n_ct * p.n_chunks * p.split;
  p.x1_bs = a.x1_bstride ? a.x1_bstride : (long)a.C1 * a.H * a.W;
  p.x2_bs = a.x2_bstride ? a.x2_bstride : (long)C2 * a.H * a.W;
  auto magic = [&](long d) { return (d <= 1 || p.n_ptiles >= 65536) ? 0u : (unsigned)((1ULL << 32) / (unsigned long long)d + 1ULL); };
  p.mTX = magic(p.tiles_x); p.mTXY = magic((long)p.tiles_x * p.tiles_y);
  const size_t smem = 2 * (sizeof(u32x4) * 8 * 130 + sizeof(unsigned) * 6 * 32 * 12) + sizeof(unsigned) * 512 +
                      sizeof(float) * 64 * (size_t)a.N;
  ADM_REQUIRE(smem <= 64 * 1024, "conv_wgrad_bf16: batch too large for the LDS GroupNorm rows");
  if (a.up) {
    if (a.act) ADM_LAUNCH((conv_wgrad_bf16w8_kernel<true, true>), dim3(p.nblk), dim3(512), smem, st, p);
    else ADM_LAUNCH((conv_wgrad_bf16w8_kernel<true, false>), dim3(p.nblk), dim3(512), smem, st, p);
  } else {
    if (a.act) ADM_LAUNCH((conv_wgrad_bf16w8_kernel<false, true>), dim3(p.nblk), dim3(512), smem, st, p);
    else ADM_LAUNCH((conv_wgrad_bf16w8_kernel<false, false>), dim3(p.nblk), dim3(512), smem, st, p);
  }
  if (ADM_CHECK_LAUNCH() != 0) return -1;
  return 2 * p.split;
}

}  // namespace adm
