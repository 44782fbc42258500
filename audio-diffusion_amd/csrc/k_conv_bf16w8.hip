// k_conv_bf16w8.hip — 8-wave variant of the bf16 3x3 forward / data-gradient kernel (k_conv_bf16.hip).  OPT-IN
// (ADM_BF16_8W=1 / option "conv_bf16_8w"): written at the end of round 1 after the GPU budget was spent — parity-checked on
// the emulator only.  Motivation (profiles/r01_pmc_bf16.md): with one wave per SIMD the conversion of the next chunk and
// the MFMAs of the current one are one in-order instruction stream (matrix pipe 24 % busy); two waves per SIMD give the
// hardware two streams to interleave.  Same operands and per-tile accumulation order as the 4-wave kernel.
#include "adm_kernels.h"

namespace adm {

struct Bf16ConvParams8 {
  const float* x1; const float* x2; int C1, C2;
  int N, Hs, Ws, Hi, Wi;
  const float* gn_scale; const float* gn_shift; int gn_nstride;
  const u32x4* wb; const float* bias; int Cout;
  const float* chan_add; int chan_add_stride;
  const float* residual; float* out;
  int tiles_x, tiles_y, n_ct, nblk;
  long x1_bs, x2_bs;
  int zins;   // UP kernels: 1 = zero-insertion x2 (data gradient of a stride-2 convolution) instead of nearest x2
};


constexpr int QPW = 18, QPP = QPW * QPW;   // input patch of a 16x16 output tile

__device__ __forceinline__ float silu_f8(float v) { return v * ADM_RCP(1.0f + __expf(-v)); }

struct Bf16Stage8 { float v[2][8]; };                // raw fp32 prefetch of one 16-channel chunk (two items of 8 channels)
template <int NA> struct Bf16Filt8 { u32x4 a[9][NA]; };  // the wave's A fragments of one chunk: 9 taps x NA cout sub-tiles

// WIDE = false: waves as 2 (64 couts) x 2 (8 pixel rows), 2 x 4 accumulator tiles, the two waves of a cout half fetch the
// same filter fragments.  WIDE = true: waves as 4 (32 couts) x 1, 1 x 8 accumulator tiles: every filter fragment is
// fetched once per workgroup (half the L2 requests, 36 registers less) at one LDS read per MFMA instead of one per two.
template <bool UP, bool ACT>
__global__ void __launch_bounds__(512, 1) conv_bf16w8_kernel(const Bf16ConvParams8 p) {
  constexpr int NA = 2, NP = 2;
  // 8 waves, two per SIMD: waves as 2 (64 couts) x 4 (4 pixel rows), 2 x 2 accumulator tiles each (64 registers), so that one
  // wave's conversion and loads run while its SIMD partner's MFMAs execute; every thread converts half as much.
  // One workgroup per CU (up to 512 registers per lane) so that everything that comes from memory is requested a full
  // chunk (filters, L2) or two chunks (input patch, HBM) before it is used: the first version requested the next tap's
  // filters 8 MFMAs ahead and queued them behind the patch loads of the in-order vector-memory counter — 17k cycles per
  // chunk against 2.3k of MFMA work (profiles/r01_train_bf16_v1_kernel_stats.md).
  ADM_DYN_SMEM(u32x4, lds);                 // [2 buffers][2 channel groups][324 pixels] + GroupNorm rows [2][Ct] floats
  const int tid = threadIdx.x, lane = tid & 63, wave = ADM_UNIFORM(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  int lid;
  {   // consecutive logical tiles (all cout tiles of a pixel tile, then the neighbouring pixel tile) share an XCD's L2
    const int b = blockIdx.x, q = p.nblk >> 3, r = p.nblk & 7, xcd = b & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int ct = lid % p.n_ct; lid /= p.n_ct;
  const int tx = lid % p.tiles_x; lid /= p.tiles_x;
  const int ty = lid % p.tiles_y, n = lid / p.tiles_y;
  const int m0 = ct * 128 + wm * (32 * NA);
  const int Ct = p.C1 + p.C2, KG = Ct >> 3;
  const int planeS = p.Hs * p.Ws;

  // staging plan (tile-invariant): item id = tid + 512 r (r = 0, 1) over the 2 x 324 (channel group, patch pixel) items;
  // the GroupNorm rows are read from LDS, so the group of an item need not be wave-uniform
  int soff[2], skg[2], spx[2];
  ADM_UNROLL
  for (int r = 0; r < 2; ++r) {
    const int id = tid + 512 * r;
    const int kg = id >= QPP ? 1 : 0, q = id - kg * QPP;
    const int ly = q / QPW, lx = q - ly * QPW;
    const int gy = ty * 16 + ly - 1, gx = tx * 16 + lx - 1;
    bool ok = (id < 2 * QPP) & (gy >= 0) & (gy < p.Hi) & (gx >= 0) & (gx < p.Wi);
    if (UP) ok = ok & !(p.zins && ((gy | gx) & 1));
    soff[r] = ok ? (UP ? (gy >> 1) * p.Ws + (gx >> 1) : gy * p.Ws + gx) : -1;
    skg[r] = kg; spx[r] = id < 2 * QPP ? q : -1;
  }
  float* gnS = reinterpret_cast<float*>(lds + 4 * QPP);     // [Ct] scale, then [Ct] shift of image n
  float* gnB = gnS + Ct;
  for (int c = tid; c < Ct; c += 512) {
    gnS[c] = p.gn_scale[(long)n * p.gn_nstride + c];
    gnB[c] = p.gn_shift[(long)n * p.gn_nstride + c];
  }
  const int n_chunks = Ct >> 4;

  const unsigned so0 = soff[0] < 0 ? 0u : (unsigned)soff[0], so1 = soff[1] < 0 ? 0u : (unsigned)soff[1];
  auto issue = [&](Bf16Stage8& s, int ch) __attribute__((always_inline)) {      // raw fp32 loads of chunk ch
    const int c0 = 16 * (ch < n_chunks ? ch : n_chunks - 1);     // past the end: harmless re-request, no branch
    const float* xc = c0 < p.C1 ? p.x1 + (long)n * p.x1_bs + (long)c0 * planeS
                                : p.x2 + (long)n * p.x2_bs + (long)(c0 - p.C1) * planeS;
    ADM_UNROLL
    for (int r = 0; r < 2; ++r) {
      const unsigned so = (r == 0 ? so0 : so1) + (unsigned)(skg[r] * 8 * planeS);
      ADM_UNROLL
      for (int e = 0; e < 8; ++e) s.v[r][e] = (xc + (long)e * planeS)[so];
    }
  };
  auto stash = [&](const Bf16Stage8& s, u32x4* buf, int ch) __attribute__((always_inline)) {   // affine + SiLU -> bf16 -> LDS
    if (ch >= n_chunks) return;
    const int c0 = 16 * ch;
    ADM_UNROLL
    for (int r = 0; r < 2; ++r) {
      const float* gsp = gnS + c0 + skg[r] * 8;
      const float4 s0 = *reinterpret_cast<const float4*>(gsp), s1 = *reinterpret_cast<const float4*>(gsp + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(gsp + Ct), b1 = *reinterpret_cast<const float4*>(gsp + Ct + 4);
      const float gs[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      const float gb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float v[8];
      ADM_UNROLL
      for (int e = 0; e < 8; ++e) {
        float t = s.v[r][e] * gs[e] + gb[e];
        if (ACT) t = silu_f8(t);
        v[e] = soff[r] < 0 ? 0.f : t;                              // zero padding applies to the activated tensor
      }
      u32x4 w;
      w[0] = ADM_PK_BF16(v[0], v[1]); w[1] = ADM_PK_BF16(v[2], v[3]);
      w[2] = ADM_PK_BF16(v[4], v[5]); w[3] = ADM_PK_BF16(v[6], v[7]);
      if (spx[r] >= 0) buf[skg[r] * QPP + spx[r]] = w;
      ADM_SCHED_FENCE();
    }
  };
  const unsigned wlane = (unsigned)(h * p.Cout + l31);             // per-lane part of the filter address (16-B units)
  // filters: ONE register set, refilled in place — as soon as the MFMAs of tap t are issued, the same registers receive
  // tap t of the next chunk, so every filter fragment is requested a full chunk (72 MFMAs) before its use at a cost of
  // 72 registers instead of 144
  auto fetch_tap = [&](Bf16Filt8<NA>& f, int ch, int t) __attribute__((always_inline)) {
    const u32x4* wt = p.wb + m0 + ((long)(2 * ch) + (long)t * KG) * p.Cout;     // uniform
    ADM_UNROLL
    for (int a = 0; a < NA; ++a) f.a[t][a] = (wt + 32 * a)[wlane];
  };
  f32x16 acc[NA][NP];
  ADM_UNROLL
  for (int a = 0; a < NA; ++a)
    ADM_UNROLL
    for (int t = 0; t < NP; ++t)
      ADM_UNROLL
      for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.f;

  // B fragment of pixel tile t (2 rows x 16 columns), tap (dy, dx): LDS slot (8 wn + 2 t + (l31 >> 4) + dy) * 18 + (l31 & 15) + dx
  const int bbase = h * QPP + (4 * wn + (l31 >> 4)) * QPW + (l31 & 15);
  Bf16Filt8<NA> F;
  auto mfma_chunk = [&](const u32x4* cur, int ch) __attribute__((always_inline)) {
    // B fragments one tap ahead, fenced: left alone, the scheduler hoists all 36 LDS reads of the chunk above the first
    // MFMA (144 registers) and the kernel spills
    const int chn = ch + 1 < n_chunks ? ch + 1 : ch;   // past the end: re-request the last chunk (no branch in the tap loop)
    u32x4 Bc[NP], Bn[NP];
    ADM_UNROLL
    for (int pt = 0; pt < NP; ++pt) Bc[pt] = cur[bbase + (2 * pt) * QPW];
    ADM_UNROLL
    for (int t = 0; t < 9; ++t) {
      if (t < 8) {
        ADM_UNROLL
        for (int pt = 0; pt < NP; ++pt) Bn[pt] = cur[bbase + (2 * pt + (t + 1) / 3) * QPW + ((t + 1) % 3)];
      }
      ADM_SCHED_FENCE();
      ADM_UNROLL
      for (int pt = 0; pt < NP; ++pt) {
        ADM_UNROLL
        for (int a = 0; a < NA; ++a) acc[a][pt] = ADM_MFMA_BF16(F.a[t][a], Bc[pt], acc[a][pt]);
      }
      ADM_SCHED_FENCE();
      fetch_tap(F, chn, t);
      ADM_UNROLL
      for (int pt = 0; pt < NP; ++pt) Bc[pt] = Bn[pt];
    }
  };

  // Pipeline: while chunk c multiplies out of LDS buffer c & 1, the filters of chunk c + 1 (rolling, above) and the raw
  // patch of chunk c + 2 are in flight, and the patch of chunk c + 1 (requested one iteration earlier) is converted into
  // the other buffer.  Two patch register sets alternate (X/Y): the loop body is written for an even/odd pair, and the
  // launcher only takes even chunk counts (Cin % 32 == 0) — an exit between the halves made the register allocator keep
  // two copies of the 128 accumulators and spill into the loop.
  Bf16Stage8 X, Y;
  u32x4* buf0 = lds;
  u32x4* buf1 = lds + 2 * QPP;
  // Order matters: vector memory returns in order, so a filter fragment (L2) requested after a patch load (HBM) cannot
  // arrive before it. The patch loads are therefore issued at the END of a chunk — after that chunk's rolling filter
  // requests, which the next chunk's MFMAs wait for — and only the conversion (stash), two chunks later, waits for them.
  // (Issued ahead of the MFMA phase they cost 9.5k cycles per chunk: every tap-0 wait inherited the HBM latency.)
  issue(X, 0);
  ADM_UNROLL
  for (int t = 0; t < 9; ++t) fetch_tap(F, 0, t);
  issue(Y, 1);
  __syncthreads();                        // GroupNorm rows are in LDS
  stash(X, buf0, 0);
  issue(X, 2);
  __syncthreads();
  for (int ch = 0; ch < n_chunks; ch += 2) {
    mfma_chunk(buf0, ch);                 // even chunk: LDS buffer 0; Y holds chunk ch + 1, X (in flight) ch + 2
    stash(Y, buf1, ch + 1);
    issue(Y, ch + 3);
    __syncthreads();
    mfma_chunk(buf1, ch + 1);             // odd chunk: LDS buffer 1; X holds chunk ch + 2, Y (in flight) ch + 3
    stash(X, buf0, ch + 2);
    issue(X, ch + 4);
    __syncthreads();
  }

  // epilogue: D row = output channel, column = pixel; fp32 bias + per-(n, channel) term + residual
  const long planeO = (long)p.Hi * p.Wi;
  ADM_UNROLL
  for (int a = 0; a < NA; ++a) {
    float bv[16];
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int co = m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
      bv[r] = p.bias[co] + p.chan_add[(long)n * p.chan_add_stride + co];
    }
    ADM_UNROLL
    for (int pt = 0; pt < NP; ++pt) {
      const int oy = ty * 16 + 4 * wn + 2 * pt + (l31 >> 4), ox = tx * 16 + (l31 & 15);
      const long pix = (long)oy * p.Wi + ox;
      float rv[16];
      if (p.residual) {
        ADM_UNROLL
        for (int r = 0; r < 16; ++r) {
          const int co = m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
          rv[r] = p.residual[((long)n * p.Cout + co) * planeO + pix];
        }
      }
      ADM_UNROLL
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = acc[a][pt][r] + bv[r];
        if (p.residual) v += rv[r];
        p.out[((long)n * p.Cout + co) * planeO + pix] = v;
      }
    }
  }
}


static int g_f8 = -1;      // -1: ADM_BF16_8W from the environment (default 0)
void set_conv_bf16_8w(int v) { g_f8 = v; }
bool conv_bf16_8w_enabled() {
  if (g_f8 < 0) { const char* e = getenv("ADM_BF16_8W"); g_f8 = e ? atoi(e) : 0; }
  return g_f8 != 0;
}

// launched by launch_conv_bf16 (k_conv_bf16.hip) when enabled; same eligibility
int launch_conv_bf16w8(const adm_conv_args& a, hipStream_t st) {
  Bf16ConvParams8 p;
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
  p.residual = a.residual; p.out = a.out; p.zins = a.up == 2;
  p.tiles_x = p.Wi / 16; p.tiles_y = p.Hi / 16; p.n_ct = a.Cout / 128;
  p.nblk = p.tiles_x * p.tiles_y * a.N * p.n_ct;
  p.x1_bs = a.x1_bstride ? a.x1_bstride : (long)a.C1 * a.H * a.W;
  p.x2_bs = a.x2_bstride ? a.x2_bstride : (long)C2 * a.H * a.W;
  const size_t smem = sizeof(u32x4) * 2 * 2 * QPP + sizeof(float) * 2 * Ct;
  ADM_REQUIRE(smem <= 64 * 1024, "conv_bf16: too many input channels for the LDS GroupNorm rows");
  set_last_conv_variant(5000 + 318);
  if (a.up) {
    if (a.act) ADM_LAUNCH((conv_bf16w8_kernel<true, true>), dim3(p.nblk), dim3(512), smem, st, p);
    else ADM_LAUNCH((conv_bf16w8_kernel<true, false>), dim3(p.nblk), dim3(512), smem, st, p);
  } else {
    if (a.act) ADM_LAUNCH((conv_bf16w8_kernel<false, true>), dim3(p.nblk), dim3(512), smem, st, p);
    else ADM_LAUNCH((conv_bf16w8_kernel<false, false>), dim3(p.nblk), dim3(512), smem, st, p);
  }
  return ADM_CHECK_LAUNCH();
}

}  // namespace adm
