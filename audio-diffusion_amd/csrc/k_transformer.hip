// k_transformer.hip — the pieces of diffusers' Transformer2DModel / BasicTransformerBlock that UNet2DConditionModel adds
// to the UNet (scripts/train_unet.py:139-159; called at pipeline_audio_diffusion.py:160-161 with `encoding`):
//   LayerNorm over channels per token, GEGLU gate, cross-attention on the (batch, seq, dim) audio encoding, and
//   self-attention for token counts whose K/V no longer fit LDS (64x64 latents of the 512-resolution latent model).
// Activations stay in the UNet's (N, C, T = H*W) layout, so every Linear of the block is a 1x1 convolution on the MFMA
// kernel (k_conv_mfma.hip) with its bias / residual epilogue; only the four operations here are new. All HBM-bound
// elementwise / small-GEMM work: lanes run along T (coalesced), no MFMA.
#include <cstdint>
#include <mutex>

#include "adm_kernels.h"

namespace adm {

// y[n][c][t] = (x[n][c][t] - mean_t) * rstd_t * gamma[c] + beta[c], statistics over c for every token (n, t).
// One lane per token, channel loop strided by T (coalesced across lanes); mean then centred variance (two passes, as
// ATen's LayerNorm kernel), then the write pass: 3 reads + 1 write of 4 B per element.
__global__ void __launch_bounds__(256) layernorm_nct_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y, int C,
                                                            long T, float eps) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const float* xp = x + (long)blockIdx.y * C * T + t;
  float* yp = y + (long)blockIdx.y * C * T + t;
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += xp[(long)c * T];
  const float mean = s / (float)C;
  float v = 0.f;
  for (int c = 0; c < C; ++c) { const float d = xp[(long)c * T] - mean; v = fmaf(d, d, v); }
  const float rstd = rsqrtf(v / (float)C + eps);
  for (int c = 0; c < C; ++c) yp[(long)c * T] = (xp[(long)c * T] - mean) * rstd * gamma[c] + beta[c];
}

// The same LayerNorm for token counts that are multiples of 64 (round 5): one workgroup per 64 tokens, the four waves split the channels
// (wave w takes c = w, w + 4, ...), lanes stay along T (256-byte coalesced rows), eight independent loads in flight per lane, the three
// reductions meet in LDS. One lane per token walked C dependent-looking strided loads on a quarter of the waves: 150-236 us per launch at
// 64x64 latents (7.2 of the 52 ms of a conditional forward, profiles/r05_conditional.md). Mean, then centred variance, then the write pass —
// the tile (64 tokens x C x 4 B <= 128 KiB) is read once from HBM and twice from the caches.
__global__ void __launch_bounds__(256) layernorm_nct_tile_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* __restrict__ y, int C,
                                                                 long T, float eps) {
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long t = (long)blockIdx.x * 64 + lane;
  const float* xp = x + (long)blockIdx.y * C * T + t;
  float* yp = y + (long)blockIdx.y * C * T + t;
  float s = 0.f;
  _Pragma("unroll 8")
  for (int c = w; c < C; c += 4) s += xp[(long)c * T];
  red[0][w][lane] = s;
  __syncthreads();
  const float mean = ((red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane])) / (float)C;
  float v = 0.f;
  _Pragma("unroll 8")
  for (int c = w; c < C; c += 4) { const float d = xp[(long)c * T] - mean; v = fmaf(d, d, v); }
  red[1][w][lane] = v;
  __syncthreads();
  const float rstd = rsqrtf(((red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane])) / (float)C + eps);
  _Pragma("unroll 8")
  for (int c = w; c < C; c += 4) yp[(long)c * T] = (xp[(long)c * T] - mean) * rstd * gamma[c] + beta[c];
}

int launch_layernorm_nct(const float* x, const float* gamma, const float* beta, float* y, int N, int C, long T, float eps,
                         hipStream_t st) {
  if (T % 64 == 0 && C >= 4) {
    ADM_LAUNCH(layernorm_nct_tile_kernel, dim3((unsigned)(T / 64), N), dim3(256), 0, st, x, gamma, beta, y, C, T, eps);
    return ADM_CHECK_LAUNCH();
  }
  ADM_LAUNCH(layernorm_nct_kernel, dim3((unsigned)((T + 255) / 256), N), dim3(256), 0, st, x, gamma, beta, y, C, T, eps);
  return ADM_CHECK_LAUNCH();
}

// GEGLU (diffusers attention.GEGLU): in (N, 2*C4, T) = [h | gate] on the channel axis -> out (N, C4, T) = h * gelu(gate),
// exact (erf) GELU as F.gelu's default.
__global__ void __launch_bounds__(256) geglu_kernel(const float* __restrict__ in, float* __restrict__ out, long per_sample) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per_sample) return;
  const float* ip = in + (long)blockIdx.y * 2 * per_sample;
  const float h = ip[i], g = ip[per_sample + i];
  out[(long)blockIdx.y * per_sample + i] = h * (0.5f * g * (1.0f + erff(g * 0.70710678118654752f)));
}

int launch_geglu(const float* in, float* out, int N, int C4, long T, hipStream_t st) {
  const long per = (long)C4 * T;
  ADM_LAUNCH(geglu_kernel, dim3((unsigned)((per + 255) / 256), N), dim3(256), 0, st, in, out, per);
  return ADM_CHECK_LAUNCH();
}

// Cross-attention of every token on the encoding: q (N, C, T) (already projected), ctx (N, S, Dc), Wk / Wv (C, Dc) the
// to_k / to_v Linear weights (no bias) -> out (N, C, T) = softmax_s(q_t . k_s * d^-0.5) v_s per head (d = C / heads).
// One workgroup per (256 tokens, head, sample): the head's K and V rows (S x d each) are computed from the encoding into
// LDS first (S * d * Dc MACs per matrix — S = 1 in the reference's use, where the result is just V broadcast), then each
// lane owns one query token.
template <int D>
__global__ void __launch_bounds__(256) cross_attention_kernel(const float* __restrict__ q, const float* __restrict__ ctx,
                                                              const float* __restrict__ Wk, const float* __restrict__ Wv,
                                                              float* __restrict__ out, int C, int T, int S, int Dc,
                                                              float scale) {
  ADM_DYN_SMEM(float, smem);
  float* Ks = smem;           // [S][D]
  float* Vs = smem + S * D;   // [S][D]
  const int head = blockIdx.y, n = blockIdx.z, tid = threadIdx.x;
  for (int e = tid; e < 2 * S * D; e += blockDim.x) {
    const int which = e / (S * D), r = e - which * S * D;
    const int s = r / D, d = r - s * D;
    const float* w = (which ? Wv : Wk) + (long)(head * D + d) * Dc;
    const float* cx = ctx + ((long)n * S + s) * Dc;
    float acc = 0.f;
    for (int k = 0; k < Dc; ++k) acc = fmaf(cx[k], w[k], acc);
    smem[e] = acc;
  }
  __syncthreads();
  const int t = blockIdx.x * blockDim.x + tid;
  if (t >= T) return;
  const float* qb = q + ((long)n * C + head * D) * T;
  float qv[D];
  ADM_UNROLL
  for (int d = 0; d < D; ++d) qv[d] = qb[(long)d * T + t];
  float m = -3.0e38f;
  for (int s = 0; s < S; ++s) {
    float a = 0.f;
    ADM_UNROLL
    for (int d = 0; d < D; ++d) a = fmaf(qv[d], Ks[s * D + d], a);
    m = fmaxf(m, a * scale);
  }
  float l = 0.f, o[D];
  ADM_UNROLL
  for (int d = 0; d < D; ++d) o[d] = 0.f;
  for (int s = 0; s < S; ++s) {
    float a = 0.f;
    ADM_UNROLL
    for (int d = 0; d < D; ++d) a = fmaf(qv[d], Ks[s * D + d], a);
    const float p = __expf(a * scale - m);
    l += p;
    ADM_UNROLL
    for (int d = 0; d < D; ++d) o[d] = fmaf(p, Vs[s * D + d], o[d]);
  }
  const float inv = 1.0f / l;
  float* ob = out + ((long)n * C + head * D) * T;
  ADM_UNROLL
  for (int d = 0; d < D; ++d) ob[(long)d * T + t] = o[d] * inv;
}

int launch_cross_attention(const float* q, const float* ctx, const float* Wk, const float* Wv, float* out, int N, int C,
                           int T, int S, int Dc, int head_dim, hipStream_t st) {
  ADM_REQUIRE(C % head_dim == 0 && S >= 1, "cross_attention: bad shape");
  const int heads = C / head_dim;
  const int bs = T >= 256 ? 256 : ((T + 63) / 64) * 64;
  dim3 grid(ceil_div(T, bs), heads, N), block(bs);
  const size_t smem = sizeof(float) * 2 * (size_t)S * head_dim;
  ADM_REQUIRE(smem <= 64 * 1024, "cross_attention: encoding sequence too long for the LDS K/V slab");
  const float scale = 1.0f / sqrtf((float)head_dim);
#define ADM_XATT_CASE(DD)                                                                                         \
  if (head_dim == DD) {                                                                                           \
    ADM_LAUNCH((cross_attention_kernel<DD>), grid, block, smem, st, q, ctx, Wk, Wv, out, C, T, S, Dc, scale);     \
    return ADM_CHECK_LAUNCH();                                                                                    \
  }
  ADM_XATT_CASE(4) ADM_XATT_CASE(8) ADM_XATT_CASE(16) ADM_XATT_CASE(32) ADM_XATT_CASE(64)
#undef ADM_XATT_CASE
  ADM_FAIL("cross_attention: unsupported head_dim (4/8/16/32/64)");
}

// Self-attention for long token counts: the same per-head math as attention_kernel (k_attention.hip) with the keys
// processed in LDS-sized blocks and an online softmax (running maximum m, denominator l and output o rescaled by
// exp(m_old - m_new) per block).  qkv (N, 3C, T) -> out (N, C, T).
template <int D>
__global__ void __launch_bounds__(256) attention_blocked_kernel(const float* __restrict__ qkv, float* __restrict__ out, int C,
                                                                int T, int KB, float scale) {
  ADM_DYN_SMEM(float, smem);
  float* Ks = smem;            // [KB][D]
  float* Vs = smem + KB * D;   // [KB][D]
  const int head = blockIdx.y, n = blockIdx.z, tid = threadIdx.x;
  const float* qb = qkv + ((long)n * 3 * C + head * D) * T;
  const float* kb = qb + (long)C * T;
  const float* vb = kb + (long)C * T;
  const int t = blockIdx.x * blockDim.x + tid;
  const bool live = t < T;
  float qv[D], o[D];
  ADM_UNROLL
  for (int d = 0; d < D; ++d) { qv[d] = live ? qb[(long)d * T + t] : 0.f; o[d] = 0.f; }
  float m = -3.0e38f, l = 0.f;
  for (int j0 = 0; j0 < T; j0 += KB) {
    const int nb = T - j0 < KB ? T - j0 : KB;
    __syncthreads();                                  // previous block fully consumed
    for (int e = tid; e < D * nb; e += blockDim.x) {
      const int d = e / nb, j = e - d * nb;
      Ks[j * D + d] = kb[(long)d * T + j0 + j];
      Vs[j * D + d] = vb[(long)d * T + j0 + j];
    }
    __syncthreads();
    float bm = m;
    for (int j = 0; j < nb; ++j) {
      float s = 0.f;
      ADM_UNROLL
      for (int d = 0; d < D; ++d) s = fmaf(qv[d], Ks[j * D + d], s);
      bm = fmaxf(bm, s * scale);
    }
    const float corr = __expf(m - bm);
    l *= corr;
    ADM_UNROLL
    for (int d = 0; d < D; ++d) o[d] *= corr;
    m = bm;
    for (int j = 0; j < nb; ++j) {
      float s = 0.f;
      ADM_UNROLL
      for (int d = 0; d < D; ++d) s = fmaf(qv[d], Ks[j * D + d], s);
      const float pj = __expf(s * scale - m);
      l += pj;
      ADM_UNROLL
      for (int d = 0; d < D; ++d) o[d] = fmaf(pj, Vs[j * D + d], o[d]);
    }
  }
  if (!live) return;
  const float inv = 1.0f / l;
  float* ob = out + ((long)n * C + head * D) * T;
  ADM_UNROLL
  for (int d = 0; d < D; ++d) ob[(long)d * T + t] = o[d] * inv;
}

int launch_attention_blocked(const float* qkv, float* out, int N, int C, int T, int head_dim, int key_block, hipStream_t st) {
  ADM_REQUIRE(C % head_dim == 0, "attention: C not divisible by head_dim");
  const int heads = C / head_dim;
  const int bs = T >= 256 ? 256 : ((T + 63) / 64) * 64;
  int KB = key_block > 0 ? key_block : (int)(64 * 1024 / (2 * sizeof(float) * head_dim));   // 64 KiB of K + V per block
  if (KB > T) KB = T;
  dim3 grid(ceil_div(T, bs), heads, N), block(bs);
  const size_t smem = sizeof(float) * 2 * (size_t)KB * head_dim;
  const float scale = 1.0f / sqrtf((float)head_dim);
#define ADM_ATTB_CASE(DD)                                                                          \
  if (head_dim == DD) {                                                                            \
    ADM_LAUNCH((attention_blocked_kernel<DD>), grid, block, smem, st, qkv, out, C, T, KB, scale);  \
    return ADM_CHECK_LAUNCH();                                                                     \
  }
  ADM_ATTB_CASE(4) ADM_ATTB_CASE(8) ADM_ATTB_CASE(16) ADM_ATTB_CASE(32) ADM_ATTB_CASE(64)
#undef ADM_ATTB_CASE
  ADM_FAIL("attention: unsupported head_dim (4/8/16/32/64)");
}

// ---------------------------------------------------------------------------------------------------------------------
// Self-attention of the Transformer2DModel blocks on the matrix pipe (round 5): flash form on v_mfma_f32_16x16x4_f32 for head
// dimensions 16 / 32 / 64 (the conditional UNet's 8 heads at 128 / 256 / 512 channels) and token counts that are multiples of 128.
// The VALU kernel above spent 5.2 ms per launch at 4096 tokens (B = 16: 26 of the 52 ms of a conditional forward,
// profiles/r05_conditional.md); here K and V of a key block are staged once per 128 queries and every operand of the two products
// is an MFMA fragment:
//   workgroup = 8 waves, wave w owns 16 queries; per 64 keys
//     S^T[key][query] = sum_d K[d][key] Q[d][query]      A = K (LDS, lanes along keys), B = Q (registers, loaded once)   D MFMAs
//     online softmax per query: the lane's 16 scores + two cross-lane maxima; p = exp(s * scale - m); O *= exp(m_old - m_new)
//     O^T[d][query] += sum_key V[d][key] p[key][query]    A = V (one ds_read_b128 = four k-steps), B = p                  D MFMAs
//   The accumulator layout of the first product (lane group g, register r <-> key 4 g + r, lane & 15 <-> query) IS the B-operand
//   layout of the second one when its k-step s takes key 4 k4 + s from lane group k4 — so the probabilities never move between lanes.
// LDS: K rows at a pitch = 16 (mod 32) words (the four d-rows of an A fragment fall into disjoint banks), V rows at a pitch = 4 (mod 64).
template <int D>
__global__ void __launch_bounds__(512) attention_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out, int C, int T,
                                                             float scale) {
  constexpr int KB = D <= 32 ? 256 : 128;          // keys per LDS block
  constexpr int PK = KB + 16, PV = KB + 4;
  ADM_DYN_SMEM(float, smem);
  float* Ks = smem;                                // [D][PK]
  float* Vs = smem + D * PK;                       // [D][PV]
  const int head = blockIdx.y, n = blockIdx.z, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, k4 = lane >> 4;
  const float* qb = qkv + ((long)n * 3 * C + head * D) * T;
  const float* kb = qb + (long)C * T;
  const float* vb = kb + (long)C * T;
  const int i0 = blockIdx.x * 128 + 16 * wave;     // this wave's 16 queries
  float qreg[D / 4];
  ADM_UNROLL
  for (int ks = 0; ks < D / 4; ++ks) qreg[ks] = qb[(long)(4 * ks + k4) * T + i0 + l15];
  f32x4 o[D / 16];
  ADM_UNROLL
  for (int dt = 0; dt < D / 16; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -3.0e38f, l = 0.f;                     // l: this lane's share of the denominator (its 16 of every 64 keys)
  for (int j0 = 0; j0 < T; j0 += KB) {
    __syncthreads();                               // previous block fully consumed
    ADM_UNROLL
    for (int e0 = 0; e0 < D * (KB / 4); e0 += 512) {
      const int e = e0 + tid;
      const int d = e / (KB / 4), c4 = e % (KB / 4);
      *reinterpret_cast<float4*>(Ks + d * PK + 4 * c4) = *reinterpret_cast<const float4*>(kb + (long)d * T + j0 + 4 * c4);
      *reinterpret_cast<float4*>(Vs + d * PV + 4 * c4) = *reinterpret_cast<const float4*>(vb + (long)d * T + j0 + 4 * c4);
    }
    __syncthreads();
    for (int jj = 0; jj < KB; jj += 64) {
      f32x4 st[4];
      ADM_UNROLL
      for (int kt = 0; kt < 4; ++kt) {
        st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        ADM_UNROLL
        for (int ks = 0; ks < D / 4; ++ks)
          st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[(4 * ks + k4) * PK + jj + 16 * kt + l15], qreg[ks], st[kt], 0, 0, 0);
      }
      float mx = st[0][0];
      ADM_UNROLL
      for (int kt = 0; kt < 4; ++kt)
        ADM_UNROLL
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m, mx * scale);
      const float corr = __expf(m - mn);
      m = mn;
      l *= corr;
      ADM_UNROLL
      for (int dt = 0; dt < D / 16; ++dt)
        ADM_UNROLL
        for (int r = 0; r < 4; ++r) o[dt][r] *= corr;
      ADM_UNROLL
      for (int kt = 0; kt < 4; ++kt)
        ADM_UNROLL
        for (int r = 0; r < 4; ++r) {
          const float pj = __expf(st[kt][r] * scale - m);
          st[kt][r] = pj;
          l += pj;
        }
      ADM_UNROLL
      for (int kt = 0; kt < 4; ++kt)
        ADM_UNROLL
        for (int dt = 0; dt < D / 16; ++dt) {
          const float4 v4 = *reinterpret_cast<const float4*>(Vs + (16 * dt + l15) * PV + jj + 16 * kt + 4 * k4);
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.x, st[kt][0], o[dt], 0, 0, 0);
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.y, st[kt][1], o[dt], 0, 0, 0);
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.z, st[kt][2], o[dt], 0, 0, 0);
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.w, st[kt][3], o[dt], 0, 0, 0);
        }
    }
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l;
  float* ob = out + ((long)n * C + head * D) * T + i0 + l15;
  ADM_UNROLL
  for (int dt = 0; dt < D / 16; ++dt)
    ADM_UNROLL
    for (int r = 0; r < 4; ++r) ob[(long)(16 * dt + 4 * k4 + r) * T] = o[dt][r] * inv;
}

// 68 - 70 KiB of dynamic LDS at head_dim 32 / 64: permission asked ONCE per device, under a lock, with the answer kept (the call is not legal
// inside a stream capture, and an ignored refusal would only show up as a generic launch error — ADVICE r5). Where the runtime refuses,
// attention_mfma_eligible says no and the caller takes launch_attention_blocked.
static bool attention_mfma_lds_ok(int head_dim) {
#if !defined(ADM_EMU)
  if (head_dim < 32) return true;                              // 34 KiB: no attribute needed
  static int state[16][2] = {};                                // per device slot x {32, 64}: 0 unknown, 1 granted, -1 refused
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  int& s = state[conv_dev_slot() & 15][head_dim == 64];
  if (s == 0) {
    const int KB = head_dim <= 32 ? 256 : 128;
    const int by = (int)(sizeof(float) * (size_t)head_dim * (KB + 16 + KB + 4));
    const hipError_t e = head_dim == 32
        ? hipFuncSetAttribute((const void*)attention_mfma_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, by)
        : hipFuncSetAttribute((const void*)attention_mfma_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, by);
    if (e != hipSuccess) (void)hipGetLastError();
    s = e == hipSuccess ? 1 : -1;
  }
  return s > 0;
#else
  (void)head_dim;
  return true;
#endif
}

bool attention_mfma_eligible(int C, int T, int head_dim) {
  if (head_dim != 16 && head_dim != 32 && head_dim != 64) return false;
  const int KB = head_dim <= 32 ? 256 : 128;
  return C % head_dim == 0 && T % 128 == 0 && T % KB == 0 && attention_mfma_lds_ok(head_dim);
}

int launch_attention_mfma(const float* qkv, float* out, int N, int C, int T, int head_dim, hipStream_t st) {
  ADM_REQUIRE(attention_mfma_eligible(C, T, head_dim), "attention_mfma: head_dim 16 / 32 / 64 and T % 128 == 0 (T % 256 for head_dim <= 32), and the device grants the kernel's dynamic LDS");
  ADM_REQUIRE((reinterpret_cast<uintptr_t>(qkv) & 15) == 0, "attention_mfma: qkv must be 16-byte aligned");
  const int heads = C / head_dim;
  const int KB = head_dim <= 32 ? 256 : 128;
  dim3 grid(T / 128, heads, N), block(512);
  const size_t smem = sizeof(float) * (size_t)head_dim * (KB + 16 + KB + 4);
  const float scale = 1.0f / sqrtf((float)head_dim);
#define ADM_ATTM_CASE(DD)                                                                                        \
  if (head_dim == DD) {                                                                                          \
    ADM_LAUNCH((attention_mfma_kernel<DD>), grid, block, smem, st, qkv, out, C, T, scale);                       \
    return ADM_CHECK_LAUNCH();                                                                                   \
  }
  ADM_ATTM_CASE(16) ADM_ATTM_CASE(32) ADM_ATTM_CASE(64)
#undef ADM_ATTM_CASE
  ADM_FAIL("attention_mfma: unsupported head_dim");
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward passes (training of the conditional UNet, scripts/train_unet.py:254-259 with --encodings)

// LayerNorm backward, pass 1 (one lane per token): with g_c = dy_c * gamma_c and xh_c = (x_c - mean) * rstd,
//   dx_c = rstd * (g_c - mean_c(g) - xh_c * mean_c(g * xh));  the token's (mean, rstd) are kept for pass 2.
__global__ void __launch_bounds__(256) layernorm_nct_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                   const float* __restrict__ gamma, float* __restrict__ dx,
                                                                   int accumulate, float* __restrict__ stats, int C, long T,
                                                                   float eps) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const long base = (long)blockIdx.y * C * T + t;
  const float* xp = x + base;
  const float* gp = dy + base;
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += xp[(long)c * T];
  const float mean = s / (float)C;
  float v = 0.f;
  for (int c = 0; c < C; ++c) { const float d = xp[(long)c * T] - mean; v = fmaf(d, d, v); }
  const float rstd = rsqrtf(v / (float)C + eps);
  float s1 = 0.f, s2 = 0.f;
  for (int c = 0; c < C; ++c) {
    const float g = gp[(long)c * T] * gamma[c], xh = (xp[(long)c * T] - mean) * rstd;
    s1 += g; s2 = fmaf(g, xh, s2);
  }
  s1 /= (float)C; s2 /= (float)C;
  float* dp = dx + base;
  for (int c = 0; c < C; ++c) {
    const float g = gp[(long)c * T] * gamma[c], xh = (xp[(long)c * T] - mean) * rstd;
    const float r = rstd * (g - s1 - xh * s2);
    dp[(long)c * T] = accumulate ? dp[(long)c * T] + r : r;
  }
  stats[2 * ((long)blockIdx.y * T + t)] = mean;
  stats[2 * ((long)blockIdx.y * T + t) + 1] = rstd;
}
// pass 2 (one workgroup per channel): dgamma_c += sum_{n,t} dy * xh, dbeta_c += sum_{n,t} dy
__global__ void __launch_bounds__(256) layernorm_nct_bwd_affine_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                       const float* __restrict__ stats, float* dgamma,
                                                                       float* dbeta, int N, int C, long T) {
  __shared__ float red[2][256];
  const int c = blockIdx.x, tid = threadIdx.x;
  float a = 0.f, b = 0.f;
  for (long i = tid; i < (long)N * T; i += 256) {
    const long n = i / T, t = i - n * T;
    const long e = (n * C + c) * T + t;
    const float g = dy[e];
    a = fmaf(g, (x[e] - stats[2 * i]) * stats[2 * i + 1], a);
    b += g;
  }
  red[0][tid] = a; red[1][tid] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; }
    __syncthreads();
  }
  if (tid == 0) { dgamma[c] += red[0][0]; dbeta[c] += red[1][0]; }
}

int launch_layernorm_nct_bwd(const float* x, const float* dy, const float* gamma, float* dx, int accumulate, float* stats,
                             float* dgamma, float* dbeta, int N, int C, long T, float eps, hipStream_t st) {
  ADM_LAUNCH(layernorm_nct_bwd_dx_kernel, dim3((unsigned)((T + 255) / 256), N), dim3(256), 0, st, x, dy, gamma, dx, accumulate,
             stats, C, T, eps);
  ADM_LAUNCH(layernorm_nct_bwd_affine_kernel, dim3(C), dim3(256), 0, st, x, dy, (const float*)stats, dgamma, dbeta, N, C, T);
  return ADM_CHECK_LAUNCH();
}

// GEGLU backward: out = h * gelu(g)  ->  dh = dy * gelu(g),  dg = dy * h * gelu'(g),
// gelu'(g) = Phi(g) + g * phi(g) (exact form: Phi = 0.5 (1 + erf(g / sqrt 2)), phi = exp(-g^2 / 2) / sqrt(2 pi)).
__global__ void __launch_bounds__(256) geglu_bwd_kernel(const float* __restrict__ in, const float* __restrict__ dy,
                                                        float* __restrict__ din, long per_sample) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per_sample) return;
  const float* ip = in + (long)blockIdx.y * 2 * per_sample;
  float* dp = din + (long)blockIdx.y * 2 * per_sample;
  const float h = ip[i], g = ip[per_sample + i], d = dy[(long)blockIdx.y * per_sample + i];
  const float Phi = 0.5f * (1.0f + erff(g * 0.70710678118654752f));
  const float phi = 0.39894228040143268f * __expf(-0.5f * g * g);
  dp[i] = d * g * Phi;
  dp[per_sample + i] = d * h * (Phi + g * phi);
}

int launch_geglu_bwd(const float* in, const float* dy, float* din, int N, int C4, long T, hipStream_t st) {
  const long per = (long)C4 * T;
  ADM_LAUNCH(geglu_bwd_kernel, dim3((unsigned)((per + 255) / 256), N), dim3(256), 0, st, in, dy, din, per);
  return ADM_CHECK_LAUNCH();
}

// Cross-attention backward: dq (N, C, T) and the gradients of to_k / to_v (C, Dc) (accumulated with atomics: every
// (sample, head) workgroup owns D rows of each, the batch sums over samples). The encoding itself is data (no gradient).
// One workgroup per (head, sample); K, V recomputed into LDS; lanes stride over the tokens; dK / dV of the head
// (S x D each) are reduced in LDS, then folded with the encoding into dWk / dWv.
template <int D>
__global__ void __launch_bounds__(256) cross_attention_bwd_kernel(const float* __restrict__ q, const float* __restrict__ ctx,
                                                                  const float* __restrict__ Wk, const float* __restrict__ Wv,
                                                                  const float* __restrict__ dy, float* __restrict__ dq,
                                                                  float* dWk, float* dWv, int C, int T, int S, int Dc,
                                                                  float scale) {
  ADM_DYN_SMEM(float, smem);
  float* Ks = smem;            // [S][D]
  float* Vs = Ks + S * D;
  float* dKs = Vs + S * D;
  float* dVs = dKs + S * D;
  const int head = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  for (int e = tid; e < 2 * S * D; e += blockDim.x) {
    const int which = e / (S * D), r = e - which * S * D;
    const int s = r / D, d = r - s * D;
    const float* w = (which ? Wv : Wk) + (long)(head * D + d) * Dc;
    const float* cx = ctx + ((long)n * S + s) * Dc;
    float acc = 0.f;
    for (int k = 0; k < Dc; ++k) acc = fmaf(cx[k], w[k], acc);
    smem[e] = acc;
    smem[2 * S * D + e] = 0.f;
  }
  __syncthreads();
  const float* qb = q + ((long)n * C + head * D) * T;
  const float* gb = dy + ((long)n * C + head * D) * T;
  float* dqb = dq + ((long)n * C + head * D) * T;
  for (int t = tid; t < T; t += blockDim.x) {
    float qv[D], go[D], dqv[D];
    ADM_UNROLL
    for (int d = 0; d < D; ++d) { qv[d] = qb[(long)d * T + t]; go[d] = gb[(long)d * T + t]; dqv[d] = 0.f; }
    float m = -3.0e38f;
    for (int s = 0; s < S; ++s) {
      float a = 0.f;
      ADM_UNROLL
      for (int d = 0; d < D; ++d) a = fmaf(qv[d], Ks[s * D + d], a);
      m = fmaxf(m, a * scale);
    }
    float l = 0.f, dsum = 0.f;
    for (int s = 0; s < S; ++s) {
      float a = 0.f, gv = 0.f;
      ADM_UNROLL
      for (int d = 0; d < D; ++d) { a = fmaf(qv[d], Ks[s * D + d], a); gv = fmaf(go[d], Vs[s * D + d], gv); }
      const float p = __expf(a * scale - m);
      l += p; dsum = fmaf(p, gv, dsum);
    }
    const float inv = 1.0f / l, Di = dsum * inv;
    for (int s = 0; s < S; ++s) {
      float a = 0.f, gv = 0.f;
      ADM_UNROLL
      for (int d = 0; d < D; ++d) { a = fmaf(qv[d], Ks[s * D + d], a); gv = fmaf(go[d], Vs[s * D + d], gv); }
      const float p = __expf(a * scale - m) * inv;
      const float ds = p * (gv - Di) * scale;
      ADM_UNROLL
      for (int d = 0; d < D; ++d) {
        dqv[d] = fmaf(ds, Ks[s * D + d], dqv[d]);
        atomicAdd(&dKs[s * D + d], ds * qv[d]);
        atomicAdd(&dVs[s * D + d], p * go[d]);
      }
    }
    ADM_UNROLL
    for (int d = 0; d < D; ++d) dqb[(long)d * T + t] = dqv[d];
  }
  __syncthreads();
  // dW[(head*D + d)][k] += sum_s dK[s][d] * ctx[n][s][k]
  for (int e = tid; e < 2 * D * Dc; e += blockDim.x) {
    const int which = e / (D * Dc), r = e - which * D * Dc;
    const int d = r / Dc, k = r - d * Dc;
    const float* g = which ? dVs : dKs;
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc = fmaf(g[s * D + d], ctx[((long)n * S + s) * Dc + k], acc);
    atomicAdd((which ? dWv : dWk) + (long)(head * D + d) * Dc + k, acc);
  }
}

int launch_cross_attention_bwd(const float* q, const float* ctx, const float* Wk, const float* Wv, const float* dy, float* dq,
                               float* dWk, float* dWv, int N, int C, int T, int S, int Dc, int head_dim, hipStream_t st) {
  ADM_REQUIRE(C % head_dim == 0 && S >= 1, "cross_attention_bwd: bad shape");
  const int heads = C / head_dim;
  const int bs = T >= 256 ? 256 : ((T + 63) / 64) * 64;
  const size_t smem = sizeof(float) * 4 * (size_t)S * head_dim;
  ADM_REQUIRE(smem <= 64 * 1024, "cross_attention_bwd: encoding sequence too long for the LDS K/V slab");
  const float scale = 1.0f / sqrtf((float)head_dim);
#define ADM_XATTB_CASE(DD)                                                                                              \
  if (head_dim == DD) {                                                                                                 \
    ADM_LAUNCH((cross_attention_bwd_kernel<DD>), dim3(heads, N), dim3(bs), smem, st, q, ctx, Wk, Wv, dy, dq, dWk, dWv, C, \
               T, S, Dc, scale);                                                                                        \
    return ADM_CHECK_LAUNCH();                                                                                          \
  }
  ADM_XATTB_CASE(4) ADM_XATTB_CASE(8) ADM_XATTB_CASE(16) ADM_XATTB_CASE(32) ADM_XATTB_CASE(64)
#undef ADM_XATTB_CASE
  ADM_FAIL("cross_attention_bwd: unsupported head_dim (4/8/16/32/64)");
}

// Self-attention backward for token counts whose head slab does not fit LDS (k_backward.hip: attn_bwd_kernel keeps
// Q, K, V and dout of a head resident). Flash-attention backward in two kernels, probabilities recomputed:
//   dq kernel  (lane = query i): key blocks through LDS; pass 1 online softmax gives m_i, 1/l_i and D_i = dout_i . out_i
//                                (kept in `stats` for the second kernel), pass 2 accumulates dq_i.
//   dkv kernel (lane = key j)  : query blocks (q, dout, stats) through LDS; dk_j, dv_j.
template <int D>
__global__ void __launch_bounds__(256) attn_bwd_blocked_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                  float* __restrict__ dqkv, float* __restrict__ stats, int C,
                                                                  int T, int KB, float scale) {
  ADM_DYN_SMEM(float, smem);
  float* Ks = smem;
  float* Vs = smem + KB * D;
  const int head = blockIdx.y, n = blockIdx.z, tid = threadIdx.x, heads = C / D;
  const float* qb = qkv + ((long)n * 3 * C + head * D) * T;
  const float* kb = qb + (long)C * T;
  const float* vb = kb + (long)C * T;
  const float* ob = dout + ((long)n * C + head * D) * T;
  const int i = blockIdx.x * blockDim.x + tid;
  const bool live = i < T;
  float q[D], go[D], dq[D];
  ADM_UNROLL
  for (int d = 0; d < D; ++d) { q[d] = live ? qb[(long)d * T + i] : 0.f; go[d] = live ? ob[(long)d * T + i] : 0.f; dq[d] = 0.f; }
  float m = -3.0e38f, l = 0.f, dsum = 0.f;
  for (int pass = 0; pass < 2; ++pass) {
    const float inv = pass ? 1.0f / l : 0.f, Di = pass ? dsum * inv : 0.f;
    for (int j0 = 0; j0 < T; j0 += KB) {
      const int nb = T - j0 < KB ? T - j0 : KB;
      __syncthreads();
      for (int e = tid; e < D * nb; e += blockDim.x) {
        const int d = e / nb, j = e - d * nb;
        Ks[j * D + d] = kb[(long)d * T + j0 + j];
        Vs[j * D + d] = vb[(long)d * T + j0 + j];
      }
      __syncthreads();
      if (pass == 0) {
        float bm = m;
        for (int j = 0; j < nb; ++j) {
          float s = 0.f;
          ADM_UNROLL
          for (int d = 0; d < D; ++d) s = fmaf(q[d], Ks[j * D + d], s);
          bm = fmaxf(bm, s * scale);
        }
        const float corr = __expf(m - bm);
        l *= corr; dsum *= corr; m = bm;
        for (int j = 0; j < nb; ++j) {
          float s = 0.f, gv = 0.f;
          ADM_UNROLL
          for (int d = 0; d < D; ++d) { s = fmaf(q[d], Ks[j * D + d], s); gv = fmaf(go[d], Vs[j * D + d], gv); }
          const float pj = __expf(s * scale - m);
          l += pj; dsum = fmaf(pj, gv, dsum);
        }
      } else {
        for (int j = 0; j < nb; ++j) {
          float s = 0.f, gv = 0.f;
          ADM_UNROLL
          for (int d = 0; d < D; ++d) { s = fmaf(q[d], Ks[j * D + d], s); gv = fmaf(go[d], Vs[j * D + d], gv); }
          const float ds = __expf(s * scale - m) * inv * (gv - Di) * scale;
          ADM_UNROLL
          for (int d = 0; d < D; ++d) dq[d] = fmaf(ds, Ks[j * D + d], dq[d]);
        }
      }
    }
  }
  if (!live) return;
  float* dqb = dqkv + ((long)n * 3 * C + head * D) * T;
  ADM_UNROLL
  for (int d = 0; d < D; ++d) dqb[(long)d * T + i] = dq[d];
  float* sp = stats + 3 * (((long)n * heads + head) * T + i);
  sp[0] = m; sp[1] = 1.0f / l; sp[2] = dsum / l;
}

template <int D>
__global__ void __launch_bounds__(256) attn_bwd_blocked_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                   float* __restrict__ dqkv, const float* __restrict__ stats,
                                                                   int C, int T, int QB, float scale) {
  ADM_DYN_SMEM(float, smem);
  float* Qs = smem;                 // [QB][D]
  float* Os = Qs + QB * D;          // dout [QB][D]
  float* Ss = Os + QB * D;          // [QB][3]: m, 1/l, D
  const int head = blockIdx.y, n = blockIdx.z, tid = threadIdx.x, heads = C / D;
  const float* qb = qkv + ((long)n * 3 * C + head * D) * T;
  const float* kb = qb + (long)C * T;
  const float* vb = kb + (long)C * T;
  const float* ob = dout + ((long)n * C + head * D) * T;
  const float* sb = stats + 3 * ((long)n * heads + head) * T;
  const int j = blockIdx.x * blockDim.x + tid;
  const bool live = j < T;
  float k[D], v[D], dk[D], dv[D];
  ADM_UNROLL
  for (int d = 0; d < D; ++d) { k[d] = live ? kb[(long)d * T + j] : 0.f; v[d] = live ? vb[(long)d * T + j] : 0.f; dk[d] = 0.f; dv[d] = 0.f; }
  for (int i0 = 0; i0 < T; i0 += QB) {
    const int nb = T - i0 < QB ? T - i0 : QB;
    __syncthreads();
    for (int e = tid; e < D * nb; e += blockDim.x) {
      const int d = e / nb, i = e - d * nb;
      Qs[i * D + d] = qb[(long)d * T + i0 + i];
      Os[i * D + d] = ob[(long)d * T + i0 + i];
    }
    for (int e = tid; e < 3 * nb; e += blockDim.x) Ss[e] = sb[3 * (long)i0 + e];
    __syncthreads();
    for (int i = 0; i < nb; ++i) {
      float s = 0.f, gv = 0.f;
      ADM_UNROLL
      for (int d = 0; d < D; ++d) { s = fmaf(Qs[i * D + d], k[d], s); gv = fmaf(Os[i * D + d], v[d], gv); }
      const float p = __expf(s * scale - Ss[3 * i]) * Ss[3 * i + 1];
      const float ds = p * (gv - Ss[3 * i + 2]) * scale;
      ADM_UNROLL
      for (int d = 0; d < D; ++d) { dk[d] = fmaf(ds, Qs[i * D + d], dk[d]); dv[d] = fmaf(p, Os[i * D + d], dv[d]); }
    }
  }
  if (!live) return;
  float* dkb = dqkv + ((long)n * 3 * C + C + head * D) * T;
  float* dvb = dkb + (long)C * T;
  ADM_UNROLL
  for (int d = 0; d < D; ++d) { dkb[(long)d * T + j] = dk[d]; dvb[(long)d * T + j] = dv[d]; }
}

// stats: 3 * N * (C / head_dim) * T floats of scratch
int launch_attention_bwd_blocked(const float* qkv, const float* dout, float* dqkv, float* stats, int N, int C, int T,
                                 int head_dim, int block, hipStream_t st) {
  ADM_REQUIRE(C % head_dim == 0 && stats != nullptr, "attention_bwd: bad argument");
  const int heads = C / head_dim;
  const int bs = T >= 256 ? 256 : ((T + 63) / 64) * 64;
  int KB = block > 0 ? block : (int)(64 * 1024 / (sizeof(float) * (2 * head_dim + 3)));
  if (KB > T) KB = T;
  dim3 grid(ceil_div(T, bs), heads, N);
  const size_t smem = sizeof(float) * (size_t)KB * (2 * head_dim + 3);
  const float scale = 1.0f / sqrtf((float)head_dim);
#define ADM_ATTBB_CASE(DD)                                                                                                  \
  if (head_dim == DD) {                                                                                                     \
    ADM_LAUNCH((attn_bwd_blocked_dq_kernel<DD>), grid, dim3(bs), smem, st, qkv, dout, dqkv, stats, C, T, KB, scale);          \
    ADM_LAUNCH((attn_bwd_blocked_dkv_kernel<DD>), grid, dim3(bs), smem, st, qkv, dout, dqkv, (const float*)stats, C, T, KB,   \
               scale);                                                                                                      \
    return ADM_CHECK_LAUNCH();                                                                                              \
  }
  ADM_ATTBB_CASE(4) ADM_ATTBB_CASE(8) ADM_ATTBB_CASE(16) ADM_ATTBB_CASE(32) ADM_ATTBB_CASE(64)
#undef ADM_ATTBB_CASE
  ADM_FAIL("attention_bwd: unsupported head_dim (4/8/16/32/64)");
}

}  // namespace adm
