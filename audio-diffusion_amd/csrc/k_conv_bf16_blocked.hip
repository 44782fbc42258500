// k_conv_bf16_blocked.hip — PROTOTYPE of the structural direction for the bf16 path: the activated conv input
// exists once per layer as bf16 in a channel-group-blocked layout  xb[n][C/8][H][W][8]  (16 B per pixel and group), written
// by one GroupNorm-apply pass and consumed by the convolution with NO conversion work: a patch item is one 16-byte load
// and one 16-byte LDS store.  Op-level only (adm_gn_apply_bf16_blocked / adm_conv2d_bf16_blocked; the executors do not use
// it), written after round 1's GPU budget was spent: emulator parity only — tools/bf16_blocked_probe.py times it against
// the fused-load kernel per layer shape.
//   Today (profiles/r01_pmc_bf16.md) the forward kernel issues 6.6 VALU instructions per MFMA, most of them GroupNorm +
//   SiLU + rounding of the patch, recomputed by every (cout tile, pixel tile) that touches a pixel; here that work is done
//   once per element by a streaming pass (read 4 B, write 2 B) and the patch bytes per conv halve.
#include "adm_kernels.h"

namespace adm {

__device__ __forceinline__ float silu_bl(float v) { return v * ADM_RCP(1.0f + __expf(-v)); }

// xb[n][kg][pixel] (u32x4 = 8 bf16 channels kg*8 .. kg*8+7) = bf16(act(x * scale[n][c] + shift[n][c])); virtual concat x1|x2
__global__ void __launch_bounds__(256) gn_apply_bf16_blocked_kernel(const float* __restrict__ x1, int C1,
                                                                     const float* __restrict__ x2, int C2,
                                                                     const float* __restrict__ gn_scale,
                                                                     const float* __restrict__ gn_shift, int gn_nstride,
                                                                     int act, u32x4* __restrict__ xb, long T) {
  const long px = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int kg = blockIdx.y, n = blockIdx.z, KG = (C1 + C2) >> 3;
  if (px >= T) return;
  const int c0 = kg * 8;
  const float* src = c0 < C1 ? x1 + ((long)n * C1 + c0) * T : x2 + ((long)n * C2 + (c0 - C1)) * T;
  const float* gs = gn_scale + (long)n * gn_nstride + c0;
  const float* gb = gn_shift + (long)n * gn_nstride + c0;
  float v[8];
  ADM_UNROLL
  for (int e = 0; e < 8; ++e) {
    float t = src[(long)e * T + px] * gs[e] + gb[e];
    v[e] = act ? silu_bl(t) : t;
  }
  u32x4 w;
  w[0] = ADM_PK_BF16(v[0], v[1]); w[1] = ADM_PK_BF16(v[2], v[3]);
  w[2] = ADM_PK_BF16(v[4], v[5]); w[3] = ADM_PK_BF16(v[6], v[7]);
  xb[((long)n * KG + kg) * T + px] = w;
}

int launch_gn_apply_bf16_blocked(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, const float* gn_scale,
                                 const float* gn_shift, int act, void* xb, hipStream_t st) {
  if (x2 == nullptr) C2 = 0;
  const int Ct = C1 + C2;
  ADM_REQUIRE(Ct % 8 == 0 && C1 % 8 == 0, "gn_apply_bf16_blocked: channel counts must be multiples of 8");
  int nstride = Ct;
  if (gn_scale == nullptr) { gn_scale = conv_const_ones(Ct); gn_shift = conv_zero_bias(Ct); nstride = 0; }
  ADM_REQUIRE(gn_scale && gn_shift, "gn_apply_bf16_blocked: constant buffers");
  const long T = (long)H * W;
  ADM_LAUNCH(gn_apply_bf16_blocked_kernel, dim3((unsigned)((T + 255) / 256), Ct / 8, N), dim3(256), 0, st, x1, C1, x2, C2,
             gn_scale, gn_shift, nstride, act, (u32x4*)xb, T);
  return ADM_CHECK_LAUNCH();
}

struct Bf16BlkParams {
  const u32x4* xb; int KGt;               // input: [N][KGt][Hs][Ws] x 16 B
  int N, Hs, Ws, Hi, Wi, upshift, zins;
  const u32x4* wb; const float* bias; int Cout;
  const float* chan_add; int chan_add_stride;
  const float* residual; float* out;
  int tiles_x, tiles_y, n_ct, nblk;
};

constexpr int KPW = 18, KPP = KPW * KPW;

struct BlkStage { u32x4 v[2]; };
struct BlkFilt { u32x4 a[9][2]; };

// 8 waves (two per SIMD) as 2 (64 couts) x 4 (4 pixel rows), 2 x 2 accumulator tiles — the layout of k_conv_bf16w8.hip
__global__ void __launch_bounds__(512, 1) conv_bf16_blocked_kernel(const Bf16BlkParams p) {
  ADM_DYN_SMEM(u32x4, lds);                 // [2 buffers][2 channel groups][324 pixels]
  const int tid = threadIdx.x, lane = tid & 63, wave = ADM_UNIFORM(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  int lid;
  {
    const int b = blockIdx.x, q = p.nblk >> 3, r = p.nblk & 7, xcd = b & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int ct = lid % p.n_ct; lid /= p.n_ct;
  const int tx = lid % p.tiles_x; lid /= p.tiles_x;
  const int ty = lid % p.tiles_y, n = lid / p.tiles_y;
  const int m0 = ct * 128 + wm * 64;
  const int n_chunks = p.KGt >> 1;
  const long planeS = (long)p.Hs * p.Ws;
  // staging items: id = tid + 512 r over the 2 x 324 (channel group, patch pixel) items; one 16-byte load each
  const int id1 = tid + 512;
  const int kg0 = tid >= KPP ? 1 : 0, px0 = tid - kg0 * KPP, px1 = id1 < 2 * KPP ? id1 - KPP : -1;
  auto src_off = [&](int q) __attribute__((always_inline)) {
    const int ly = q / KPW, lx = q - ly * KPW;
    const int gy = ty * 16 + ly - 1, gx = tx * 16 + lx - 1;
    bool ok = (q >= 0) & (gy >= 0) & (gy < p.Hi) & (gx >= 0) & (gx < p.Wi);
    ok = ok & !(p.zins && ((gy | gx) & 1));
    return ok ? (unsigned)((gy >> p.upshift) * p.Ws + (gx >> p.upshift)) : 0xFFFFFFFFu;
  };
  const unsigned so0 = src_off(px0) , so1 = src_off(px1 < 0 ? -1 : px1);
  const u32x4 zero = {0u, 0u, 0u, 0u};
  auto issue = [&](BlkStage& s, int ch) __attribute__((always_inline)) {
    const int c = ch < n_chunks ? ch : n_chunks - 1;
    const u32x4* base = p.xb + ((long)n * p.KGt + 2 * c) * planeS;          // uniform
    s.v[0] = (base + (long)kg0 * planeS)[so0 == 0xFFFFFFFFu ? 0u : so0];
    s.v[1] = (base + planeS)[so1 == 0xFFFFFFFFu ? 0u : so1];
  };
  auto stash = [&](const BlkStage& s, u32x4* buf, int ch) __attribute__((always_inline)) {
    if (ch >= n_chunks) return;
    buf[kg0 * KPP + px0] = so0 == 0xFFFFFFFFu ? zero : s.v[0];
    if (px1 >= 0) buf[KPP + px1] = so1 == 0xFFFFFFFFu ? zero : s.v[1];
  };
  const unsigned wlane = (unsigned)(h * p.Cout + l31);
  const int KG = p.KGt;
  auto fetch_tap = [&](BlkFilt& f, int ch, int t) __attribute__((always_inline)) {
    const u32x4* wt = p.wb + m0 + ((long)(2 * ch) + (long)t * KG) * p.Cout;
    f.a[t][0] = wt[wlane]; f.a[t][1] = (wt + 32)[wlane];
  };
  f32x16 acc[2][2];
  ADM_UNROLL
  for (int a = 0; a < 2; ++a)
    ADM_UNROLL
    for (int t = 0; t < 2; ++t)
      ADM_UNROLL
      for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.f;
  const int bbase = h * KPP + (4 * wn + (l31 >> 4)) * KPW + (l31 & 15);
  BlkFilt F;
  auto mfma_chunk = [&](const u32x4* cur, int ch) __attribute__((always_inline)) {
    const int chn = ch + 1 < n_chunks ? ch + 1 : ch;
    u32x4 Bc[2], Bn[2];
    ADM_UNROLL
    for (int pt = 0; pt < 2; ++pt) Bc[pt] = cur[bbase + (2 * pt) * KPW];
    ADM_UNROLL
    for (int t = 0; t < 9; ++t) {
      if (t < 8) {
        ADM_UNROLL
        for (int pt = 0; pt < 2; ++pt) Bn[pt] = cur[bbase + (2 * pt + (t + 1) / 3) * KPW + ((t + 1) % 3)];
      }
      ADM_SCHED_FENCE();
      ADM_UNROLL
      for (int pt = 0; pt < 2; ++pt) {
        acc[0][pt] = ADM_MFMA_BF16(F.a[t][0], Bc[pt], acc[0][pt]);
        acc[1][pt] = ADM_MFMA_BF16(F.a[t][1], Bc[pt], acc[1][pt]);
      }
      ADM_SCHED_FENCE();
      fetch_tap(F, chn, t);
      ADM_UNROLL
      for (int pt = 0; pt < 2; ++pt) Bc[pt] = Bn[pt];
    }
  };
  BlkStage X, Y;
  u32x4* buf0 = lds;
  u32x4* buf1 = lds + 2 * KPP;
  issue(X, 0);
  ADM_UNROLL
  for (int t = 0; t < 9; ++t) fetch_tap(F, 0, t);
  issue(Y, 1);
  stash(X, buf0, 0);
  issue(X, 2);
  __syncthreads();
  for (int ch = 0; ch < n_chunks; ch += 2) {
    mfma_chunk(buf0, ch);
    stash(Y, buf1, ch + 1);
    issue(Y, ch + 3);
    __syncthreads();
    mfma_chunk(buf1, ch + 1);
    stash(X, buf0, ch + 2);
    issue(X, ch + 4);
    __syncthreads();
  }
  const long planeO = (long)p.Hi * p.Wi;
  ADM_UNROLL
  for (int a = 0; a < 2; ++a) {
    float bv[16];
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int co = m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
      bv[r] = p.bias[co] + p.chan_add[(long)n * p.chan_add_stride + co];
    }
    ADM_UNROLL
    for (int pt = 0; pt < 2; ++pt) {
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

// 3x3 stride 1 "same" on a blocked bf16 input (Ct channels, H x W source, up = 0 | 1 nearest | 2 zero insertion)
int launch_conv2d_bf16_blocked(const void* xb, int Ct, int N, int H, int W, int up, const void* wb, const float* bias, int Cout,
                               const float* chan_add, int chan_add_stride, const float* residual, float* out, hipStream_t st) {
  Bf16BlkParams p;
  p.Hi = up ? 2 * H : H; p.Wi = up ? 2 * W : W;
  ADM_REQUIRE(Ct % 32 == 0 && Cout % 128 == 0 && p.Hi % 16 == 0 && p.Wi % 16 == 0 && up >= 0 && up <= 2,
              "conv2d_bf16_blocked: needs Cin % 32 == 0, Cout % 128 == 0, output a multiple of 16x16");
  p.xb = reinterpret_cast<const u32x4*>(xb); p.KGt = Ct / 8;
  p.N = N; p.Hs = H; p.Ws = W; p.upshift = up ? 1 : 0; p.zins = up == 2;
  p.wb = reinterpret_cast<const u32x4*>(wb);
  p.bias = bias ? bias : conv_zero_bias(Cout); p.Cout = Cout;
  p.chan_add = chan_add; p.chan_add_stride = chan_add_stride;
  if (p.chan_add == nullptr) { p.chan_add = conv_zero_bias(Cout); p.chan_add_stride = 0; }
  ADM_REQUIRE(p.bias && p.chan_add, "conv2d_bf16_blocked: constant buffers");
  p.residual = residual; p.out = out;
  p.tiles_x = p.Wi / 16; p.tiles_y = p.Hi / 16; p.n_ct = Cout / 128;
  p.nblk = p.tiles_x * p.tiles_y * N * p.n_ct;
  const size_t smem = sizeof(u32x4) * 2 * 2 * KPP;
  ADM_LAUNCH(conv_bf16_blocked_kernel, dim3(p.nblk), dim3(512), smem, st, p);
  return ADM_CHECK_LAUNCH();
}

}  // namespace adm
