// k_conv_bf16.hip — mixed-precision 3x3 stride-1 convolution for `--mixed_precision bf16` training
// (scripts/train_unet.py:391-401 hands the choice to accelerate, which wraps the step in torch.autocast: Conv2d runs on
// bf16 operands with fp32 accumulation).  Design: master weights, activations and gradients stay fp32 in HBM; only the
// MFMA operands are bf16 — the activated input (GroupNorm affine + SiLU applied on the load path) is rounded to bf16 as
// it is written into the LDS patch, the filters are repacked to bf16 once per optimizer step, v_mfma_f32_32x32x16_bf16
// accumulates in fp32 and the epilogue (bias, time-embedding bias, residual) is fp32.  Compared with autocast this keeps
// more precision (no bf16 rounding of stored activations / weight gradients); the arithmetic rate is the bf16 MFMA rate.
//
//   forward / data-gradient : conv_bf16_kernel   (data-gradient = the same kernel on dy with transposed, flipped filters)
//   weight gradient         : conv_wgrad_bf16_kernel (k = pixels; taps from funnel-shifted LDS rows)
//
// Workgroup tile of the forward kernel: 128 output channels x 16x16 output pixels, 4 waves as 2 (64 couts) x 2 (8 rows);
// each wave holds 2 x 4 accumulator tiles of 32x32 (128 registers) so that one A fragment serves 4 MFMAs and one B
// fragment 2 (the bf16 MFMA retires a 32x32x16 tile in 8 passes: operand delivery, not arithmetic, is the limit).
// K loop: 16 input channels per step = one MFMA k-step.  The 18x18 input patch of the step is staged as
// [2 channel groups of 8][324 pixels] x 16 B — a lane's B fragment (8 channels of one pixel) is one ds_read_b128 and the
// 32 lanes of a half-wave read 32 consecutive 16-B slots.  A fragments come straight from the packed filters
// [tap][Cin/8][Cout][8] (512 contiguous bytes per half-wave, L2-resident).  Two workgroups per CU: one stages while the
// other feeds the MFMA pipe.
#include "adm_kernels.h"

namespace adm {

struct Bf16ConvParams {
  const float* x1; const float* x2; int C1, C2;
  int N, Hs, Ws, Hi, Wi;
  const float* gn_scale; const float* gn_shift; int gn_nstride;
  const u32x4* wb; const float* bias; int Cout;
  const float* chan_add; int chan_add_stride;
  const float* residual; float* out;
  int tiles_x, tiles_y, n_ct, nblk;
  long x1_bs, x2_bs;
};

constexpr int BPW = 18, BPP = BPW * BPW;   // input patch of a 16x16 output tile

__device__ __forceinline__ float silu_b(float v) { return v * ADM_RCP(1.0f + __expf(-v)); }

template <bool UP, bool ACT>
__global__ void __launch_bounds__(256, 2) conv_bf16_kernel(const Bf16ConvParams p) {
  ADM_DYN_SMEM(u32x4, lds);                 // [2 buffers][2 channel groups][324 pixels]
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
  const int m0 = ct * 128 + wm * 64;
  const int Ct = p.C1 + p.C2, KG = Ct >> 3;
  const int planeS = p.Hs * p.Ws;

  // staging plan (tile-invariant): every wave converts three rounds of 64 pixels x 8 channels per chunk —
  // rounds 0/1: pixel 64*wave + lane of channel group 0/1; round 2: the 68 left-over pixels (wave&1 picks the 64-block,
  // wave>>1 the group).  The channel group of a round is wave-uniform, so the GroupNorm rows are scalar loads.
  int soff[2];
  ADM_UNROLL
  for (int r = 0; r < 2; ++r) {
    const int q = r == 0 ? 64 * wave + lane : 256 + 64 * (wave & 1) + lane;
    const int ly = q / BPW, lx = q - ly * BPW;
    const int gy = ty * 16 + ly - 1, gx = tx * 16 + lx - 1;
    const bool ok = (q < BPP) & (gy >= 0) & (gy < p.Hi) & (gx >= 0) & (gx < p.Wi);
    soff[r] = ok ? (UP ? (gy >> 1) * p.Ws + (gx >> 1) : gy * p.Ws + gx) : -1;
  }
  const int q2 = 256 + 64 * (wave & 1) + lane;
  const int kg2 = wave >> 1;
  const bool has2 = q2 < BPP;

  float xr[3][8];
  auto issue = [&](int c0) __attribute__((always_inline)) {      // raw fp32 loads of chunk [c0, c0 + 16)
    const float* xc = c0 < p.C1 ? p.x1 + (long)n * p.x1_bs + (long)c0 * planeS
                                : p.x2 + (long)n * p.x2_bs + (long)(c0 - p.C1) * planeS;
    ADM_UNROLL
    for (int r = 0; r < 3; ++r) {
      const int kg = r < 2 ? r : kg2;
      const int so = soff[r < 2 ? 0 : 1];
      const float* src = xc + (long)(kg * 8) * planeS + (so < 0 ? 0 : so);
      ADM_UNROLL
      for (int e = 0; e < 8; ++e) xr[r][e] = src[(long)e * planeS];
    }
  };
  auto stash = [&](u32x4* buf, int c0) __attribute__((always_inline)) {   // affine + SiLU, round to bf16, one 16-B LDS store
    ADM_UNROLL
    for (int r = 0; r < 3; ++r) {
      const int kg = r < 2 ? r : kg2;
      const int so = soff[r < 2 ? 0 : 1];
      const float* gs = p.gn_scale + (long)n * p.gn_nstride + c0 + kg * 8;
      const float* gb = p.gn_shift + (long)n * p.gn_nstride + c0 + kg * 8;
      float v[8];
      ADM_UNROLL
      for (int e = 0; e < 8; ++e) {
        float t = xr[r][e] * gs[e] + gb[e];
        if (ACT) t = silu_b(t);
        v[e] = so < 0 ? 0.f : t;                                   // zero padding applies to the activated tensor
      }
      u32x4 w;
      w[0] = ADM_PK_BF16(v[0], v[1]); w[1] = ADM_PK_BF16(v[2], v[3]);
      w[2] = ADM_PK_BF16(v[4], v[5]); w[3] = ADM_PK_BF16(v[6], v[7]);
      if (r < 2) buf[kg * BPP + 64 * wave + lane] = w;
      else if (has2) buf[kg * BPP + q2] = w;
    }
  };

  f32x16 acc[2][4];
  ADM_UNROLL
  for (int a = 0; a < 2; ++a)
    ADM_UNROLL
    for (int t = 0; t < 4; ++t)
      ADM_UNROLL
      for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.f;

  // B fragment of pixel tile t (2 rows x 16 columns), tap (dy, dx): LDS slot (8 wn + 2 t + (l31 >> 4) + dy) * 18 + (l31 & 15) + dx
  const int bbase = h * BPP + (8 * wn + (l31 >> 4)) * BPW + (l31 & 15);
  const u32x4* wrow = p.wb + (long)h * p.Cout + m0 + l31;          // + (tap * KG + 2 chunk) * Cout + 32 a

  const int n_chunks = Ct >> 4;
  issue(0);
  stash(lds, 0);
  if (n_chunks > 1) issue(16);
  __syncthreads();
  for (int ch = 0; ch < n_chunks; ++ch) {
    const u32x4* cur = lds + (ch & 1) * (2 * BPP);
    u32x4* nxt = lds + ((ch & 1) ^ 1) * (2 * BPP);
    const u32x4* wch = wrow + (long)(2 * ch) * p.Cout;
    u32x4 A[2], An[2];
    A[0] = wch[0]; A[1] = wch[32];
    ADM_UNROLL
    for (int t = 0; t < 9; ++t) {
      if (t < 8) {
        const u32x4* wt = wch + (long)(t + 1) * KG * p.Cout;
        An[0] = wt[0]; An[1] = wt[32];
      }
      ADM_UNROLL
      for (int pt = 0; pt < 4; ++pt) {
        const u32x4 B = cur[bbase + (2 * pt + t / 3) * BPW + (t % 3)];
        acc[0][pt] = ADM_MFMA_BF16(A[0], B, acc[0][pt]);
        acc[1][pt] = ADM_MFMA_BF16(A[1], B, acc[1][pt]);
      }
      A[0] = An[0]; A[1] = An[1];
    }
    if (ch + 1 < n_chunks) {
      stash(nxt, 16 * (ch + 1));
      if (ch + 2 < n_chunks) issue(16 * (ch + 2));
    }
    __syncthreads();
  }

  // epilogue: D row = output channel, column = pixel; fp32 bias + per-(n, channel) term + residual
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
    for (int pt = 0; pt < 4; ++pt) {
      const int oy = ty * 16 + 8 * wn + 2 * pt + (l31 >> 4), ox = tx * 16 + (l31 & 15);
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

// ---------------------------------------------------------------- weight gradient
// dW[co][ci][tap] = sum over pixels of dy[co][pixel] * a[ci][pixel + tap], a = upsample(act(GroupNorm(x))).
// GEMM with k = pixels: one MFMA k-step is a row of 16 output pixels (lane half h owns pixels 8h..8h+7); D = 32 couts x 32
// cins per tap, 9 taps = 144 accumulator registers per wave.  Workgroup = 128 couts (4 waves) x 32 cins x a range of
// 16x4-pixel tiles (split-K over pixel tiles, partial sums reduced by wgrad_reduce_kernel as in the fp32 path).
//   LDS dy tile : [row 4][half 2][cout 128] x 16 B (8 pixels)  — the A fragment is one ds_read_b128
//   LDS patch   : [row 6][cin 32] x 48 B (18 pixels used, padded to 24) — per patch row a lane reads pixels 8h..8h+9
//                 (b128 + b32) and forms the three horizontal taps in registers: dx = 0 as read, dx = 2 the same dwords
//                 shifted by one register, dx = 1 by four v_alignbit_b32 — 3.75 LDS dwords per MFMA.
struct Bf16WgradParams {
  const float* x1; const float* x2; int C1, C2;
  const float* dy; int Cout;
  int N, Hs, Ws, Hi, Wi;
  const float* gn_scale; const float* gn_shift; int gn_nstride;
  float* part;
  int tiles_x, tiles_y, n_ptiles, n_ct, n_chunks, split, tiles_per_block, nblk;
  long x1_bs, x2_bs;
};

template <bool UP, bool ACT>
__global__ void __launch_bounds__(256, 2) conv_wgrad_bf16_kernel(const Bf16WgradParams p) {
  constexpr int XROW = 12;                       // dwords per (row, cin) of the patch
  ADM_DYN_SMEM(u32x4, lds4);
  u32x4* ldsD = lds4;                            // 8 * 128 fragments
  unsigned* ldsX = reinterpret_cast<unsigned*>(lds4 + 8 * 128);   // 6 * 32 * 12 dwords
  const int tid = threadIdx.x, lane = tid & 63, wave = ADM_UNIFORM(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
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

  f32x16 acc[9];
  ADM_UNROLL
  for (int t = 0; t < 9; ++t)
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // tile-invariant staging roles
  // dy: item = (cout, row, half): 4 per thread, 8 pixels (two float4) each
  // patch: item = (row, cin, pixel pair): 1728 items, 7 rounds
  int xcin[7], xrow[7], xq[7];
  ADM_UNROLL
  for (int j = 0; j < 7; ++j) {
    const int id = tid + 256 * j;
    const int rc = id / 9;
    xq[j] = id - rc * 9; xrow[j] = rc >> 5; xcin[j] = rc & 31;     // rows >= 6: past the end (round 6 is partial)
  }

  const int t_begin = sp * p.tiles_per_block;
  int t_end = t_begin + p.tiles_per_block;
  if (t_end > p.n_ptiles) t_end = p.n_ptiles;
  for (int pt = t_begin; pt < t_end; ++pt) {
    const int tx = pt % p.tiles_x;
    const int ty = (pt / p.tiles_x) % p.tiles_y, n = pt / (p.tiles_x * p.tiles_y);
    // ---- dy tile -> bf16 A fragments
    ADM_UNROLL
    for (int j = 0; j < 4; ++j) {
      const int id = tid + 256 * j;
      const int hh = id & 1, r = (id >> 1) & 3, co = id >> 3;
      const float* src = p.dy + ((long)n * p.Cout + m0 + co) * planeO + (long)(ty * 4 + r) * p.Wi + tx * 16 + 8 * hh;
      const float4 v0 = *reinterpret_cast<const float4*>(src);
      const float4 v1 = *reinterpret_cast<const float4*>(src + 4);
      u32x4 w;
      w[0] = ADM_PK_BF16(v0.x, v0.y); w[1] = ADM_PK_BF16(v0.z, v0.w);
      w[2] = ADM_PK_BF16(v1.x, v1.y); w[3] = ADM_PK_BF16(v1.z, v1.w);
      ldsD[(r * 2 + hh) * 128 + co] = w;
    }
    // ---- activated input patch -> bf16 pixel pairs
    const float* xt = xsrc + (long)n * xbs;
    ADM_UNROLL
    for (int j = 0; j < 7; ++j) {
      if (j == 6 && xrow[j] >= 6) continue;
      const int gy = ty * 4 - 1 + xrow[j], gx = tx * 16 - 1 + 2 * xq[j];
      const bool oky = (gy >= 0) & (gy < p.Hi);
      const bool ok0 = oky & (gx >= 0) & (gx < p.Wi), ok1 = oky & (gx + 1 < p.Wi);      // gx + 1 >= 0 always
      const int sy = UP ? (gy >> 1) : gy;
      const float* row = xt + (long)xcin[j] * planeS + (long)(oky ? sy : 0) * p.Ws;
      const int sx0 = ok0 ? (UP ? (gx >> 1) : gx) : 0, sx1 = ok1 ? (UP ? ((gx + 1) >> 1) : gx + 1) : 0;
      float a = row[sx0], b = row[sx1];
      const long gi = (long)n * p.gn_nstride + c0 + xcin[j];
      const float sc = p.gn_scale[gi], sh = p.gn_shift[gi];
      a = a * sc + sh; b = b * sc + sh;
      if (ACT) { a = silu_b(a); b = silu_b(b); }
      a = ok0 ? a : 0.f; b = ok1 ? b : 0.f;
      ldsX[(xrow[j] * 32 + xcin[j]) * XROW + xq[j]] = ADM_PK_BF16(a, b);
    }
    __syncthreads();
    // ---- 4 k-steps x 9 taps
    ADM_UNROLL
    for (int r = 0; r < 4; ++r) {
      const u32x4 A = ldsD[(r * 2 + h) * 128 + 32 * wave + l31];
      ADM_UNROLL
      for (int dy3 = 0; dy3 < 3; ++dy3) {
        const unsigned* xr = ldsX + ((r + dy3) * 32 + l31) * XROW + 4 * h;
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
    __syncthreads();
  }
  float* out = p.part + (long)sp * p.Cout * Ct * 9;
  ADM_UNROLL
  for (int t = 0; t < 9; ++t) {
    const int cc = c0 + l31;
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int co = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      out[((long)co * Ct + cc) * 9 + t] = acc[t][r];
    }
  }
}

// ---------------------------------------------------------------- filters: fp32 (Cout,Cin,3,3) -> bf16 [tap][Cin/8][Cout][8]
__global__ void pack_bf16_weight_kernel(const float* __restrict__ w, unsigned* __restrict__ wb, int Cout, int Cin,
                                        int transposed) {
  // one thread per bf16 PAIR of the packed tensor. transposed: the data-gradient filters (roles of Cout/Cin swapped, taps
  // flipped), i.e. packed "Cout" = Cin and packed "Cin" = Cout.
  const int Co = transposed ? Cin : Cout, Ci = transposed ? Cout : Cin;
  const long total = (long)9 * Ci * Co / 2;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int e2 = (int)(i & 3);
  long r = i >> 2;
  const int co = (int)(r % Co); r /= Co;
  const int kg = (int)(r % (Ci >> 3));
  const int t = (int)(r / (Ci >> 3));
  const int ci = kg * 8 + 2 * e2;
  float a, b;
  if (!transposed) {
    a = w[((long)co * Cin + ci) * 9 + t];
    b = w[((long)co * Cin + ci + 1) * 9 + t];
  } else {
    a = w[((long)ci * Cin + co) * 9 + (8 - t)];
    b = w[((long)(ci + 1) * Cin + co) * 9 + (8 - t)];
  }
  wb[i] = ADM_PK_BF16(a, b);
}

int launch_pack_bf16_weight(const float* w, void* wb, int Cout, int Cin, int transposed, hipStream_t st) {
  ADM_REQUIRE(Cout % 8 == 0 && Cin % 8 == 0, "pack_bf16_weight: channel counts must be multiples of 8");
  const long total = (long)9 * Cin * Cout / 2;
  ADM_LAUNCH(pack_bf16_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, (unsigned*)wb, Cout, Cin,
             transposed);
  return ADM_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- dispatch
static int g_bf16_mode = -1;   // -1: ADM_CONV_BF16 from the environment (default 0 = fp32 everywhere)
void set_conv_bf16(int m) { g_bf16_mode = m; }
bool conv_bf16_enabled() {
  if (g_bf16_mode < 0) { const char* e = getenv("ADM_CONV_BF16"); g_bf16_mode = e ? atoi(e) : 0; }
  return g_bf16_mode != 0;
}

// 3x3 stride 1 "same", output a multiple of 16x16, channel chunks of 16 that do not straddle the concat seam, Cout % 128.
bool conv_bf16_eligible(const adm_conv_args& a) {
  if (a.ks != 3 || a.stride != 1 || a.pad_lo != 1 || a.w_bstride != 0 || a.bf16_packed == nullptr || a.up > 1) return false;
  const int C2 = a.x2 ? a.C2 : 0;
  const int Hi = a.up ? 2 * a.H : a.H, Wi = a.up ? 2 * a.W : a.W;
  return Wi % 16 == 0 && Hi % 16 == 0 && (a.C1 + C2) % 16 == 0 && a.C1 % 16 == 0 && a.Cout % 128 == 0 &&
         (a.gn_scale != nullptr || !a.act);
}

int launch_conv_bf16(const adm_conv_args& a, hipStream_t st) {
  Bf16ConvParams p;
  const int C2 = a.x2 ? a.C2 : 0, Ct = a.C1 + C2;
  p.x1 = a.x1; p.x2 = a.x2; p.C1 = a.C1; p.C2 = C2;
  p.N = a.N; p.Hs = a.H; p.Ws = a.W;
  p.Hi = a.up ? 2 * a.H : a.H; p.Wi = a.up ? 2 * a.W : a.W;
  p.gn_scale = a.gn_scale; p.gn_shift = a.gn_shift; p.gn_nstride = Ct;
  if (p.gn_scale == nullptr) {   // no GroupNorm on the load path: identity affine rows
    p.gn_scale = conv_const_ones(Ct); p.gn_shift = conv_zero_bias(Ct); p.gn_nstride = 0;
  }
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
  const size_t smem = sizeof(u32x4) * 2 * 2 * BPP;
  set_last_conv_variant(5000 + 316);
  if (a.up) {
    if (a.act) ADM_LAUNCH((conv_bf16_kernel<true, true>), dim3(p.nblk), dim3(256), smem, st, p);
    else ADM_LAUNCH((conv_bf16_kernel<true, false>), dim3(p.nblk), dim3(256), smem, st, p);
  } else {
    if (a.act) ADM_LAUNCH((conv_bf16_kernel<false, true>), dim3(p.nblk), dim3(256), smem, st, p);
    else ADM_LAUNCH((conv_bf16_kernel<false, false>), dim3(p.nblk), dim3(256), smem, st, p);
  }
  return ADM_CHECK_LAUNCH();
}

// 3x3 stride 1 "same", output a multiple of 16x4, 32-channel chunks inside one source tensor, full 128-cout tiles.
bool conv_wgrad_bf16_eligible(const adm_conv_args& a) {
  if (a.ks != 3 || a.stride != 1 || a.pad_lo != 1 || a.up > 1) return false;
  const int C2 = a.x2 ? a.C2 : 0;
  const int Hi = a.up ? 2 * a.H : a.H, Wi = a.up ? 2 * a.W : a.W;
  return Wi % 16 == 0 && Hi % 4 == 0 && (a.C1 + C2) % 32 == 0 && a.C1 % 32 == 0 && a.Cout % 128 == 0 &&
         (a.gn_scale != nullptr || !a.act);
}

// Partial sums into `workspace` ([split][Cout*Cin*9], the layout and split of conv_wgrad_workspace); the caller
// (launch_conv_wgrad) runs the reduction.
int launch_conv_wgrad_bf16(const adm_conv_args& a, const float* dy, float* dW, int accumulate, float* workspace, int split,
                           hipStream_t st) {
  (void)dW; (void)accumulate;
  Bf16WgradParams p;
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
  p.tiles_per_block = ceil_div(p.n_ptiles, split);
  p.split = ceil_div(p.n_ptiles, p.tiles_per_block);
  ADM_REQUIRE(p.split == split, "conv_wgrad_bf16: split must leave no empty workgroup");
  p.nblk = p.n_ct * p.n_chunks * p.split;
  p.x1_bs = a.x1_bstride ? a.x1_bstride : (long)a.C1 * a.H * a.W;
  p.x2_bs = a.x2_bstride ? a.x2_bstride : (long)C2 * a.H * a.W;
  const size_t smem = sizeof(u32x4) * 8 * 128 + sizeof(unsigned) * 6 * 32 * 12;
  if (a.up) {
    if (a.act) ADM_LAUNCH((conv_wgrad_bf16_kernel<true, true>), dim3(p.nblk), dim3(256), smem, st, p);
    else ADM_LAUNCH((conv_wgrad_bf16_kernel<true, false>), dim3(p.nblk), dim3(256), smem, st, p);
  } else {
    if (a.act) ADM_LAUNCH((conv_wgrad_bf16_kernel<false, true>), dim3(p.nblk), dim3(256), smem, st, p);
    else ADM_LAUNCH((conv_wgrad_bf16_kernel<false, false>), dim3(p.nblk), dim3(256), smem, st, p);
  }
  return ADM_CHECK_LAUNCH();
}

}  // namespace adm
