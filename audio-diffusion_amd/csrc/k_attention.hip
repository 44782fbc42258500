// k_attention.hip — self-attention core of the deprecated-AttnBlock form:
//   per (n, head): out[:, t] = sum_j softmax_j(q_t . k_j * d^-0.5) v_j,  fp32 softmax, 64 heads x d = 8,
//   T = 256 (16x16) or 64 (8x8) tokens in the UNet.
// qkv comes from ONE fused 1x1 convolution (q|k|v stacked on the channel axis, GroupNorm folded into its load
// path) and stays NCHW = (N, 3C, T): each head's q/k/v is a [d][T] slab, so token-contiguous loads coalesce.
// One workgroup per (n, head, 256-query block): K and V of the head are staged once in LDS as [T][d]
// (one broadcast ds_read_b128 pair per key), each lane owns one query row. Two passes over the keys
// (max, then exp/sum/PV) reproduce torch's softmax exactly up to fp32 summation order.
// With d = 8 the QK^T / PV products are K=8 / N=8 GEMMs — below any MFMA tile's useful shape (a 16x16x4 f32
// MFMA version costs the same issue cycles) — so this stays on the vector ALUs.
// Algorithmic bytes: 4*(3*N*C*T + N*C*T); total work 4*N*heads*T*T*d FLOP (0.8 GFLOP/sample/forward).
#include <cstdint>
#include "adm_kernels.h"

namespace adm {

template <int D>
__global__ void __launch_bounds__(256) attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int C,
                                                        int T, float scale) {
  ADM_DYN_SMEM(float, smem);
  float* Ks = smem;          // [T][D]
  float* Vs = smem + T * D;  // [T][D]
  const int head = blockIdx.y, n = blockIdx.z;
  const int tid = threadIdx.x;
  const float* qb = qkv + ((long)n * 3 * C + head * D) * T;
  const float* kb = qb + (long)C * T;
  const float* vb = kb + (long)C * T;
  for (int e = tid; e < D * T; e += blockDim.x) {
    const int d = e / T, j = e - d * T;
    Ks[j * D + d] = kb[e];
    Vs[j * D + d] = vb[e];
  }
  __syncthreads();
  const int t = blockIdx.x * blockDim.x + tid;
  if (t >= T) return;
  float q[D];
  ADM_UNROLL
  for (int d = 0; d < D; ++d) q[d] = qb[(long)d * T + t];
  float m = -3.0e38f;
  for (int j = 0; j < T; ++j) {
    float s = 0.f;
    ADM_UNROLL
    for (int d = 0; d < D; ++d) s = fmaf(q[d], Ks[j * D + d], s);
    m = fmaxf(m, s * scale);
  }
  float l = 0.f, o[D];
  ADM_UNROLL
  for (int d = 0; d < D; ++d) o[d] = 0.f;
  for (int j = 0; j < T; ++j) {
    float s = 0.f;
    ADM_UNROLL
    for (int d = 0; d < D; ++d) s = fmaf(q[d], Ks[j * D + d], s);
    const float pj = __expf(s * scale - m);
    l += pj;
    ADM_UNROLL
    for (int d = 0; d < D; ++d) o[d] = fmaf(pj, Vs[j * D + d], o[d]);
  }
  const float inv = 1.0f / l;
  float* ob = out + ((long)n * C + head * D) * T;
  ADM_UNROLL
  for (int d = 0; d < D; ++d) ob[(long)d * T + t] = o[d] * inv;
}

// The single-sample form ("single_sample", by model): four lanes per query, each a quarter of the keys — one sample of the 256x256 model is 64
// workgroups of attention_kernel with one wave per SIMD, each lane a serial chain over all T keys twice (36-42 us per launch for 0.13 GFLOP).
// Here a wave holds 16 queries x 4 key quarters (lane = quarter * 16 + query: the 16 lanes of a quarter read one K / V row — a broadcast),
// a workgroup 64 queries: 4x the workgroups, chains a quarter as long. The row maximum is combined exactly (max), the sum and the PV
// products as (p0 + p1) + (p2 + p3) — another summation order than attention_kernel's, chosen by the model's rule and never by the batch.
template <int D>
__global__ void __launch_bounds__(256) attention_split4_kernel(const float* __restrict__ qkv, float* __restrict__ out, int C, int T, float scale) {
  ADM_DYN_SMEM(float, smem);
  float* Ks = smem;          // [T][D]
  float* Vs = smem + T * D;  // [T][D]
  const int head = blockIdx.y, n = blockIdx.z;
  const int tid = threadIdx.x;
  const float* qb = qkv + ((long)n * 3 * C + head * D) * T;
  const float* kb = qb + (long)C * T;
  const float* vb = kb + (long)C * T;
  for (int e = tid; e < D * T; e += 256) {
    const int d = e / T, j = e - d * T;
    Ks[j * D + d] = kb[e];
    Vs[j * D + d] = vb[e];
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int part = lane >> 4;
  const int t = blockIdx.x * 64 + wave * 16 + (lane & 15);
  const int tq = t < T ? t : T - 1;               // (a ragged last block: the spare lanes recompute the last query and do not store)
  const int j0 = part * (T / 4), j1 = part == 3 ? T : j0 + T / 4;
  float q[D];
  ADM_UNROLL
  for (int d = 0; d < D; ++d) q[d] = qb[(long)d * T + tq];
  float m = -3.0e38f;
  for (int j = j0; j < j1; ++j) {
    float s = 0.f;
    ADM_UNROLL
    for (int d = 0; d < D; ++d) s = fmaf(q[d], Ks[j * D + d], s);
    m = fmaxf(m, s * scale);
  }
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float l = 0.f, o[D];
  ADM_UNROLL
  for (int d = 0; d < D; ++d) o[d] = 0.f;
  for (int j = j0; j < j1; ++j) {
    float s = 0.f;
    ADM_UNROLL
    for (int d = 0; d < D; ++d) s = fmaf(q[d], Ks[j * D + d], s);
    const float pj = __expf(s * scale - m);
    l += pj;
    ADM_UNROLL
    for (int d = 0; d < D; ++d) o[d] = fmaf(pj, Vs[j * D + d], o[d]);
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  ADM_UNROLL
  for (int d = 0; d < D; ++d) {
    o[d] += __shfl_xor(o[d], 16, 64);
    o[d] += __shfl_xor(o[d], 32, 64);
  }
  if (part != 0 || t >= T) return;
  const float inv = 1.0f / l;
  float* ob = out + ((long)n * C + head * D) * T;
  ADM_UNROLL
  for (int d = 0; d < D; ++d) ob[(long)d * T + t] = o[d] * inv;
}

int launch_attention(const float* qkv, float* out, int N, int C, int T, int head_dim, hipStream_t st, int single_sample) {
  ADM_REQUIRE(C % head_dim == 0, "attention: C not divisible by head_dim");
  const int heads = C / head_dim;
  const int bs = T >= 256 ? 256 : ((T + 63) / 64) * 64;
  dim3 grid(ceil_div(T, bs), heads, N), block(bs);
  const size_t smem = sizeof(float) * 2 * (size_t)T * head_dim;
  // head dimensions 16 / 32 / 64 (the Transformer2DModel blocks of the conditional UNet): flash attention on the f32 MFMAs — chosen by the
  // layer's shape alone (a sample's bits must not depend on the batch it is in)
  if (attention_mfma_eligible(C, T, head_dim) && (reinterpret_cast<uintptr_t>(qkv) & 15) == 0) return launch_attention_mfma(qkv, out, N, C, T, head_dim, st);
  if (smem > 64 * 1024)   // K/V of a head no longer fit the default LDS window: key-blocked online-softmax kernel
    return launch_attention_blocked(qkv, out, N, C, T, head_dim, 0, st);
  const float scale = 1.0f / sqrtf((float)head_dim);
  if (single_sample_rule(single_sample) && T >= 64 && T % 4 == 0 && (head_dim == 8 || head_dim == 4 || head_dim == 16)) {
    const dim3 g4(ceil_div(T, 64), heads, N);
    if (head_dim == 8) ADM_LAUNCH((attention_split4_kernel<8>), g4, dim3(256), smem, st, qkv, out, C, T, scale);
    else if (head_dim == 4) ADM_LAUNCH((attention_split4_kernel<4>), g4, dim3(256), smem, st, qkv, out, C, T, scale);
    else ADM_LAUNCH((attention_split4_kernel<16>), g4, dim3(256), smem, st, qkv, out, C, T, scale);
    return ADM_CHECK_LAUNCH();
  }
#define ADM_ATT_CASE(DD)                                                                  \
  if (head_dim == DD) {                                                                   \
    ADM_LAUNCH((attention_kernel<DD>), grid, block, smem, st, qkv, out, C, T, scale);     \
    return ADM_CHECK_LAUNCH();                                                            \
  }
  ADM_ATT_CASE(8) ADM_ATT_CASE(4) ADM_ATT_CASE(16) ADM_ATT_CASE(32) ADM_ATT_CASE(64)
#undef ADM_ATT_CASE
  ADM_FAIL("attention: unsupported head_dim (4/8/16/32/64)");
}

}  // namespace adm
