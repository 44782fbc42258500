// k_conv1x1_bf16.hip — bf16-operand 1x1 convolutions (shortcuts, attention projections) and their weight gradient for
// mixed-precision training.  OPT-IN (option "conv_bf16" = 2; 1 keeps them on the fp32 MFMA kernel): written at the end of
// round 1 without GPU time left — parity-checked on the emulator only, not yet timed.  Same numerics contract as
// k_conv_bf16.hip: operands rounded to bf16 (RNE) on the way into LDS / at packing time, fp32 accumulation and epilogue.
//
// A 1x1 convolution has 2 Cin Cout / (4 (Cin + Cout)) FLOP per byte — 43 for 256 -> 128: at bf16 MFMA rates it is a
// streaming kernel, so the design goal is full-width HBM access, not MFMA occupancy:
//   forward / data gradient: 128 couts x 256 consecutive pixels per workgroup; a chunk of 16 input channels is 2 x 8 channel
//     rows of 1 KiB each, read by the 256 lanes pixel-contiguous (4 B per lane, whole cache lines), converted
//     (GroupNorm affine + SiLU when the conv has them) and stored as [2 groups][256 pixels] x 16 B; filters
//     [Cin/8][Cout][8] straight from L2, one chunk ahead.  Output rows are written 128 B contiguous per half-wave.
//   weight gradient: dW[co][ci] = sum_px dy[co][px] a[ci][px], k = pixels; 128 couts x 128 cins per workgroup (4 waves as
//     2 x 2, 64 x 64 each), 64 pixels per stage read as 256-byte channel rows, split-K over pixel ranges with the fp32
//     path's workspace layout and reduction.
#include "adm_kernels.h"

namespace adm {

struct Bf16PwParams {
  const float* x1; const float* x2; int C1, C2;
  int N; long T;                       // pixels per plane
  const float* gn_scale; const float* gn_shift; int gn_nstride;
  const u32x4* wb; const float* bias; int Cout;
  const float* chan_add; int chan_add_stride;
  const float* residual; float* out;
  // split output (the data gradient of a convolution over a virtual concat): couts [0, split_c) go to out / residual (a tensor of
  // split_c channels), couts [split_c, Cout) to out2 / residual2 (a tensor of Cout - split_c channels); split_c == Cout: one output
  const float* residual2; float* out2; int split_c;
  int tiles, n_ct, nblk;
  long x1_bs, x2_bs;
};

__device__ __forceinline__ float silu_p(float v) { return v * ADM_RCP(1.0f + __expf(-v)); }

struct PwStage { float v[2][8]; };

template <bool ACT, bool F16 = false>
__global__ void __launch_bounds__(256, 1) conv1x1_bf16_kernel(const Bf16PwParams p) {
  ADM_DYN_SMEM(u32x4, lds);                 // [2 buffers][2 channel groups][256 pixels] + GroupNorm rows [2][Ct] floats
  const int tid = threadIdx.x, lane = tid & 63, wave = ADM_UNIFORM(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  int lid;
  {
    const int b = blockIdx.x, q = p.nblk >> 3, r = p.nblk & 7, xcd = b & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int ct = lid % p.n_ct; lid /= p.n_ct;
  const int tile = lid % p.tiles, n = lid / p.tiles;
  const int m0 = ct * 128 + wm * 64;
  const int Ct = p.C1 + p.C2, n_chunks = Ct >> 4;
  const long T = p.T, p0 = (long)tile * 256;
  float* gnS = reinterpret_cast<float*>(lds + 4 * 256);
  float* gnB = gnS + Ct;
  for (int c = tid; c < Ct; c += 256) {
    gnS[c] = p.gn_scale[(long)n * p.gn_nstride + c];
    gnB[c] = p.gn_shift[(long)n * p.gn_nstride + c];
  }
  auto issue = [&](PwStage& s, int ch) __attribute__((always_inline)) {
    const int c0 = 16 * (ch < n_chunks ? ch : n_chunks - 1);
    const float* xc = (c0 < p.C1 ? p.x1 + (long)n * p.x1_bs + (long)c0 * T
                                 : p.x2 + (long)n * p.x2_bs + (long)(c0 - p.C1) * T) + p0;
    ADM_UNROLL
    for (int g = 0; g < 2; ++g)
      ADM_UNROLL
      for (int e = 0; e < 8; ++e) s.v[g][e] = (xc + (long)(8 * g + e) * T)[tid];
  };
  auto stash = [&](const PwStage& s, u32x4* buf, int ch) __attribute__((always_inline)) {
    if (ch >= n_chunks) return;
    const int c0 = 16 * ch;
    ADM_UNROLL
    for (int g = 0; g < 2; ++g) {
      const float4 s0 = *reinterpret_cast<const float4*>(gnS + c0 + 8 * g), s1 = *reinterpret_cast<const float4*>(gnS + c0 + 8 * g + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(gnB + c0 + 8 * g), b1 = *reinterpret_cast<const float4*>(gnB + c0 + 8 * g + 4);
      const float gs[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      const float gb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float v[8];
      ADM_UNROLL
      for (int e = 0; e < 8; ++e) {
        float t = s.v[g][e] * gs[e] + gb[e];
        if (ACT) t = silu_p(t);
        v[e] = t;
      }
      u32x4 w;
      w[0] = ADM_PK16(F16, v[0], v[1]); w[1] = ADM_PK16(F16, v[2], v[3]);
      w[2] = ADM_PK16(F16, v[4], v[5]); w[3] = ADM_PK16(F16, v[6], v[7]);
      buf[g * 256 + tid] = w;
    }
  };
  const unsigned wlane = (unsigned)(h * p.Cout + l31);
  auto fetch = [&](u32x4 (&f)[2], int ch) __attribute__((always_inline)) {
    const int c = ch < n_chunks ? ch : n_chunks - 1;
    const u32x4* wt = p.wb + m0 + (long)(2 * c) * p.Cout;
    f[0] = wt[wlane]; f[1] = (wt + 32)[wlane];
  };
  f32x16 acc[2][4];
  ADM_UNROLL
  for (int a = 0; a < 2; ++a)
    ADM_UNROLL
    for (int t = 0; t < 4; ++t)
      ADM_UNROLL
      for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.f;
  const int bbase = h * 256 + 128 * wn + l31;
  auto mfma_chunk = [&](const u32x4 (&f)[2], const u32x4* cur) __attribute__((always_inline)) {
    ADM_UNROLL
    for (int pt = 0; pt < 4; ++pt) {
      const u32x4 B = cur[bbase + 32 * pt];
      acc[0][pt] = ADM_MFMA16(F16, f[0], B, acc[0][pt]);
      acc[1][pt] = ADM_MFMA16(F16, f[1], B, acc[1][pt]);
    }
  };
  // same pipeline as conv_bf16_kernel: LDS double buffer, two patch register sets (two chunks ahead), filters one chunk ahead
  PwStage X, Y;
  u32x4 F[2], G[2];
  u32x4* buf0 = lds;
  u32x4* buf1 = lds + 2 * 256;
  issue(X, 0);
  fetch(F, 0);
  issue(Y, 1);
  __syncthreads();
  stash(X, buf0, 0);
  issue(X, 2);
  __syncthreads();
  for (int ch = 0; ch < n_chunks; ch += 2) {
    fetch(G, ch + 1);
    mfma_chunk(F, buf0);
    stash(Y, buf1, ch + 1);
    issue(Y, ch + 3);
    __syncthreads();
    fetch(F, ch + 2);
    mfma_chunk(G, buf1);
    stash(X, buf0, ch + 2);
    issue(X, ch + 4);
    __syncthreads();
  }
  // epilogue: a wave's 32 couts x 32 pixels go through 4 KiB of its own in LDS (the operand buffers are dead behind the last
  // barrier) and come back as float4 per lane — bias, per-sample term, residual and the store are 16-byte operations, one store
  // instruction = 8 couts x 128 contiguous bytes (round 4: the dword version issued 128 stores per wave; k_conv_bf16b.hip).
  float* const stage = reinterpret_cast<float*>(lds) + 1024 * wave;
  const int srow = lane >> 3, scol = 4 * (lane & 7);
  ADM_UNROLL
  for (int a = 0; a < 2; ++a) {
    const int cb = m0 + 32 * a;                                    // the 32 couts of this sub-tile lie inside one output (launcher)
    const bool second = cb >= p.split_c;
    const int Cs = second ? p.Cout - p.split_c : p.split_c;
    float* const ob = (second ? p.out2 : p.out) + ((long)n * Cs + (cb - (second ? p.split_c : 0))) * T + p0 + 128 * wn + scol;
    const float* const rb0 = second ? p.residual2 : p.residual;
    const float* const rb = rb0 ? rb0 + ((long)n * Cs + (cb - (second ? p.split_c : 0))) * T + p0 + 128 * wn + scol : nullptr;
    float bv[4];
    ADM_UNROLL
    for (int k = 0; k < 4; ++k) {
      const int co = cb + srow + 8 * k;
      bv[k] = p.bias[co] + p.chan_add[(long)n * p.chan_add_stride + co];
    }
    ADM_UNROLL
    for (int pt = 0; pt < 4; ++pt) {
      ADM_UNROLL
      for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = acc[a][pt][r];
      ADM_WAVE_LDS_ORDER();
      float4 sv[4];
      ADM_UNROLL
      for (int k = 0; k < 4; ++k) sv[k] = *reinterpret_cast<const float4*>(stage + (srow + 8 * k) * 32 + scol);
      ADM_WAVE_LDS_ORDER();
      ADM_UNROLL
      for (int k = 0; k < 4; ++k) {
        const long o = (long)(srow + 8 * k) * T + 32 * pt;
        float4 v = sv[k];
        v.x += bv[k]; v.y += bv[k]; v.z += bv[k]; v.w += bv[k];
        if (rb) {
          const float4 rr = *reinterpret_cast<const float4*>(rb + o);
          v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        *reinterpret_cast<float4*>(ob + o) = v;
      }
    }
  }
}

// 1x1, stride 1, no upsample, pixel count a multiple of 256, an even number of 16-channel chunks inside one source tensor
bool conv1x1_bf16_eligible(const adm_conv_args& a) {
  if (a.ks != 1 || a.stride != 1 || a.up != 0 || a.w_bstride != 0 || a.bf16_packed == nullptr) return false;
  const int C2 = a.x2 ? a.C2 : 0;
  return ((long)a.H * a.W) % 256 == 0 && (a.C1 + C2) % 32 == 0 && a.C1 % 16 == 0 && a.Cout % 128 == 0 &&
         (a.gn_scale != nullptr || !a.act);
}

int launch_conv1x1_bf16(const adm_conv_args& a, hipStream_t st) { return launch_conv1x1_bf16_split(a, a.Cout, nullptr, nullptr, st); }

// couts [0, split_c) -> a.out (+ a.residual), couts [split_c, Cout) -> out2 (+ residual2): two tensors of split_c and Cout - split_c
// channels (the two halves of a virtual concat whose gradient this convolution produces); split_c == Cout: the plain call
int launch_conv1x1_bf16_split(const adm_conv_args& a, int split_c, float* out2, const float* residual2, hipStream_t st) {
  ADM_REQUIRE(split_c == a.Cout || (out2 != nullptr && split_c > 0 && split_c < a.Cout && split_c % 32 == 0), "conv1x1_bf16: split output");
  ADM_REQUIRE(((reinterpret_cast<uintptr_t>(a.out) | reinterpret_cast<uintptr_t>(a.residual) | reinterpret_cast<uintptr_t>(out2) |
                reinterpret_cast<uintptr_t>(residual2)) & 15) == 0, "conv1x1_bf16: outputs / residuals must be 16-byte aligned");
  Bf16PwParams p;
  const int C2 = a.x2 ? a.C2 : 0, Ct = a.C1 + C2;
  p.x1 = a.x1; p.x2 = a.x2; p.C1 = a.C1; p.C2 = C2; p.N = a.N; p.T = (long)a.H * a.W;
  p.gn_scale = a.gn_scale; p.gn_shift = a.gn_shift; p.gn_nstride = Ct;
  if (p.gn_scale == nullptr) { p.gn_scale = conv_const_ones(Ct); p.gn_shift = conv_zero_bias(Ct); p.gn_nstride = 0; }
  p.wb = reinterpret_cast<const u32x4*>(a.bf16_packed);
  p.bias = a.bias ? a.bias : conv_zero_bias(a.Cout); p.Cout = a.Cout;
  p.chan_add = a.chan_add; p.chan_add_stride = a.chan_add_stride;
  if (p.chan_add == nullptr) { p.chan_add = conv_zero_bias(a.Cout); p.chan_add_stride = 0; }
  ADM_REQUIRE(p.gn_scale && p.gn_shift && p.bias && p.chan_add, "conv1x1_bf16: constant buffers");
  p.residual = a.residual; p.out = a.out;
  p.residual2 = residual2; p.out2 = out2; p.split_c = split_c;
  p.tiles = (int)(p.T / 256); p.n_ct = a.Cout / 128;
  p.nblk = p.tiles * a.N * p.n_ct;
  p.x1_bs = a.x1_bstride ? a.x1_bstride : (long)a.C1 * p.T;
  p.x2_bs = a.x2_bstride ? a.x2_bstride : (long)C2 * p.T;
  const size_t smem = sizeof(u32x4) * 4 * 256 + sizeof(float) * 2 * Ct;
  ADM_REQUIRE(smem <= 64 * 1024, "conv1x1_bf16: too many input channels for the LDS GroupNorm rows");
  set_last_conv_variant(5000 + 116);
  if (conv_op16_f16()) {
    if (a.act) ADM_LAUNCH((conv1x1_bf16_kernel<true, true>), dim3(p.nblk), dim3(256), smem, st, p);
    else ADM_LAUNCH((conv1x1_bf16_kernel<false, true>), dim3(p.nblk), dim3(256), smem, st, p);
  } else {
    if (a.act) ADM_LAUNCH((conv1x1_bf16_kernel<true, false>), dim3(p.nblk), dim3(256), smem, st, p);
    else ADM_LAUNCH((conv1x1_bf16_kernel<false, false>), dim3(p.nblk), dim3(256), smem, st, p);
  }
  return ADM_CHECK_LAUNCH();
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of a 1x1 convolution: dW[co][ci] = sum_{n, px} dy[n][co][px] * a[n][ci][px].
struct Bf16PwWgradParams {
  const float* x1; const float* x2; int C1, C2;
  const float* dy; int Cout;
  int N; long T;
  const float* gn_scale; const float* gn_shift; int gn_nstride;
  float* part;                        // [split][Cout * Ct]
  int n_ct, n_chunks, n_stages, stages_per_block, split, nblk;
  long x1_bs, x2_bs;
};

// raw fp32 prefetch of one 64-pixel stage: 4 dy items + 4 input items per thread, 8 pixels (two float4) each
struct PwWgStage { float4 d[4][2]; float4 x[4][2]; int n; };

template <bool ACT, bool F16 = false>
__global__ void __launch_bounds__(256, 1) conv1x1_wgrad_bf16_kernel(const Bf16PwWgradParams p) {
  constexpr int LD = 130;                   // fragments per pixel-group row of 128 channels (padded: see k_conv_bf16.hip)
  constexpr int BUF4 = 2 * 8 * LD;          // dy rows then input rows: [8 pixel groups][130]
  ADM_DYN_SMEM(u32x4, lds4);
  float* gnS = reinterpret_cast<float*>(lds4 + 2 * BUF4);       // [N][128] scale, then shift, of this channel chunk
  float* gnB = gnS + p.N * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = ADM_UNIFORM(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  int lid;
  {
    const int b = blockIdx.x, q = p.nblk >> 3, r = p.nblk & 7, xcd = b & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int chunk = lid % p.n_chunks; lid /= p.n_chunks;
  const int ct = lid % p.n_ct, sp = lid / p.n_ct;
  const int m0 = ct * 128, c0 = chunk * 128;
  const int Ct = p.C1 + p.C2;
  const long T = p.T;
  const int stages_per_image = (int)(T / 64);
  const float* xsrc = c0 < p.C1 ? p.x1 + (long)c0 * T : p.x2 + (long)(c0 - p.C1) * T;
  const long xbs = c0 < p.C1 ? p.x1_bs : p.x2_bs;
  const int s_begin = sp * p.stages_per_block;
  int s_end = s_begin + p.stages_per_block;
  if (s_end > p.n_stages) s_end = p.n_stages;
  for (int i = tid; i < p.N * 128; i += 256) {
    const long gi = (long)(i >> 7) * p.gn_nstride + c0 + (i & 127);
    gnS[i] = p.gn_scale[gi];
    gnB[i] = p.gn_shift[gi];
  }
  f32x16 acc[2][2];
  ADM_UNROLL
  for (int a = 0; a < 2; ++a)
    ADM_UNROLL
    for (int b = 0; b < 2; ++b)
      ADM_UNROLL
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  // item j of a thread: pixel group g = tid & 7 (8 pixels = 32 B; 8 lanes cover a 256-byte channel row), channel (tid >> 3) + 32 j
  const int g = tid & 7, chb = tid >> 3;
  auto load_stage = [&](PwWgStage& s, int st_raw) __attribute__((always_inline)) {
    const int st = st_raw < s_end ? st_raw : s_end - 1;
    const int n = st / stages_per_image;
    const long px = (long)(st - n * stages_per_image) * 64 + 8 * g;
    s.n = n;
    const float* db = p.dy + ((long)n * p.Cout + m0) * T + px;
    const float* xb = xsrc + (long)n * xbs + px;
    ADM_UNROLL
    for (int j = 0; j < 4; ++j) {
      const long co = (long)(chb + 32 * j) * T;
      s.d[j][0] = *reinterpret_cast<const float4*>(db + co); s.d[j][1] = *reinterpret_cast<const float4*>(db + co + 4);
      s.x[j][0] = *reinterpret_cast<const float4*>(xb + co); s.x[j][1] = *reinterpret_cast<const float4*>(xb + co + 4);
    }
  };
  auto pack8 = [&](const float4& a, const float4& b) __attribute__((always_inline)) {
    u32x4 w;
    w[0] = ADM_PK16(F16, a.x, a.y); w[1] = ADM_PK16(F16, a.z, a.w);
    w[2] = ADM_PK16(F16, b.x, b.y); w[3] = ADM_PK16(F16, b.z, b.w);
    return w;
  };
  auto stash_stage = [&](const PwWgStage& s, u32x4* buf) __attribute__((always_inline)) {
    ADM_UNROLL
    for (int j = 0; j < 4; ++j) {
      const int ch = chb + 32 * j;
      buf[g * LD + ch] = pack8(s.d[j][0], s.d[j][1]);
      const float sc = gnS[s.n * 128 + ch], sh = gnB[s.n * 128 + ch];
      float4 a = s.x[j][0], b = s.x[j][1];
      a.x = a.x * sc + sh; a.y = a.y * sc + sh; a.z = a.z * sc + sh; a.w = a.w * sc + sh;
      b.x = b.x * sc + sh; b.y = b.y * sc + sh; b.z = b.z * sc + sh; b.w = b.w * sc + sh;
      if (ACT) {
        a.x = silu_p(a.x); a.y = silu_p(a.y); a.z = silu_p(a.z); a.w = silu_p(a.w);
        b.x = silu_p(b.x); b.y = silu_p(b.y); b.z = silu_p(b.z); b.w = silu_p(b.w);
      }
      buf[8 * LD + g * LD + ch] = pack8(a, b);
    }
  };
  auto mfma_stage = [&](const u32x4* buf, bool valid) __attribute__((always_inline)) {
    ADM_UNROLL
    for (int ks = 0; ks < 4; ++ks) {           // k-step = 16 pixels = pixel groups 2 ks (lane half 0) and 2 ks + 1 (half 1)
      u32x4 A[2], B[2];
      ADM_UNROLL
      for (int a = 0; a < 2; ++a) {
        A[a] = buf[(2 * ks + h) * LD + 64 * wm + 32 * a + l31];
        if (!valid) { A[a][0] = 0u; A[a][1] = 0u; A[a][2] = 0u; A[a][3] = 0u; }
        B[a] = buf[8 * LD + (2 * ks + h) * LD + 64 * wn + 32 * a + l31];
      }
      ADM_UNROLL
      for (int a = 0; a < 2; ++a)
        ADM_UNROLL
        for (int b = 0; b < 2; ++b) acc[a][b] = ADM_MFMA16(F16, A[a], B[b], acc[a][b]);
    }
  };
  PwWgStage P, Q;
  u32x4* buf0 = lds4;
  u32x4* buf1 = lds4 + BUF4;
  load_stage(P, s_begin);
  load_stage(Q, s_begin + 1);
  __syncthreads();
  stash_stage(P, buf0);
  load_stage(P, s_begin + 2);
  __syncthreads();
  for (int st = s_begin; st < s_end; st += 2) {
    mfma_stage(buf0, true);
    stash_stage(Q, buf1);
    load_stage(Q, st + 3);
    __syncthreads();
    mfma_stage(buf1, st + 1 < s_end);
    stash_stage(P, buf0);
    load_stage(P, st + 4);
    __syncthreads();
  }
  float* out = p.part + (long)sp * p.Cout * Ct;
  ADM_UNROLL
  for (int a = 0; a < 2; ++a)
    ADM_UNROLL
    for (int b = 0; b < 2; ++b) {
      const int cc = c0 + 64 * wn + 32 * b + l31;
      ADM_UNROLL
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + 64 * wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
        out[(long)co * Ct + cc] = acc[a][b][r];
      }
    }
}

// 1x1, stride 1, no upsample, pixel count a multiple of 64, 128-channel chunks inside one source tensor, full 128-cout tiles
bool conv1x1_wgrad_bf16_eligible(const adm_conv_args& a) {
  if (a.ks != 1 || a.stride != 1 || a.up != 0) return false;
  const int C2 = a.x2 ? a.C2 : 0;
  return ((long)a.H * a.W) % 64 == 0 && (a.C1 + C2) % 128 == 0 && a.C1 % 128 == 0 && a.Cout % 128 == 0 &&
         (a.gn_scale != nullptr || !a.act);
}

// partial sums into `workspace` ([split][Cout * Cin], the layout of conv_wgrad_workspace); the caller runs the reduction
int launch_conv1x1_wgrad_bf16(const adm_conv_args& a, const float* dy, float* workspace, int split, hipStream_t st) {
  Bf16PwWgradParams p;
  const int C2 = a.x2 ? a.C2 : 0, Ct = a.C1 + C2;
  p.x1 = a.x1; p.x2 = a.x2; p.C1 = a.C1; p.C2 = C2; p.dy = dy; p.Cout = a.Cout;
  p.N = a.N; p.T = (long)a.H * a.W;
  p.gn_scale = a.gn_scale; p.gn_shift = a.gn_shift; p.gn_nstride = Ct;
  if (p.gn_scale == nullptr) { p.gn_scale = conv_const_ones(Ct); p.gn_shift = conv_zero_bias(Ct); p.gn_nstride = 0; }
  ADM_REQUIRE(p.gn_scale && p.gn_shift, "conv1x1_wgrad_bf16: constant buffers");
  ADM_REQUIRE((reinterpret_cast<uintptr_t>(dy) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.x1) & 15) == 0 &&
              (a.x2 == nullptr || (reinterpret_cast<uintptr_t>(a.x2) & 15) == 0), "conv1x1_wgrad_bf16: 16-byte alignment");
  p.part = workspace;
  p.n_ct = a.Cout / 128; p.n_chunks = Ct / 128;
  p.n_stages = (int)(a.N * (p.T / 64));
  if (split > p.n_stages) split = p.n_stages;
  p.stages_per_block = ceil_div(p.n_stages, split);
  p.split = ceil_div(p.n_stages, p.stages_per_block);
  p.nblk = p.n_ct * p.n_chunks * p.split;
  p.x1_bs = a.x1_bstride ? a.x1_bstride : (long)a.C1 * p.T;
  p.x2_bs = a.x2_bstride ? a.x2_bstride : (long)C2 * p.T;
  const size_t smem = sizeof(u32x4) * 2 * 2 * 8 * 130 + sizeof(float) * 256 * (size_t)a.N;
  ADM_REQUIRE(smem <= 160 * 1024, "conv1x1_wgrad_bf16: batch too large for the LDS GroupNorm rows");
#if !defined(ADM_EMU)
  static bool once = [] {
    (void)hipFuncSetAttribute((const void*)conv1x1_wgrad_bf16_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv1x1_wgrad_bf16_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv1x1_wgrad_bf16_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv1x1_wgrad_bf16_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)once;
#endif
  if (conv_op16_f16()) {
    if (a.act) ADM_LAUNCH((conv1x1_wgrad_bf16_kernel<true, true>), dim3(p.nblk), dim3(256), smem, st, p);
    else ADM_LAUNCH((conv1x1_wgrad_bf16_kernel<false, true>), dim3(p.nblk), dim3(256), smem, st, p);
  } else {
    if (a.act) ADM_LAUNCH((conv1x1_wgrad_bf16_kernel<true, false>), dim3(p.nblk), dim3(256), smem, st, p);
    else ADM_LAUNCH((conv1x1_wgrad_bf16_kernel<false, false>), dim3(p.nblk), dim3(256), smem, st, p);
  }
  return p.split;      // > 0: number of partial slabs written (the reduction must sum exactly these)
}

}  // namespace adm
