// k_backward.hip — backward kernels of the UNet training step that are NOT convolutions (SURVEY.md §8(a) T4/T5;
// reference: `accelerator.backward(loss)` of scripts/train_unet.py:259, i.e. torch autograd of the U-sum forward).
// Data-gradients of the convolutions reuse the forward MFMA kernel (transposed/flipped weights, zero-insertion for the
// stride-2 convs); weight-gradients are in k_conv_wgrad.hip. Here:
//   sumpool2x2        : backward of the nearest-x2 upsample folded into Upsample2D's conv load path
//   accumulate        : dst (+)= src                      (residual / skip-connection gradient fan-in)
//   chan_sums         : per-(n,c) and per-c sums over pixels (time-embedding bias and conv bias gradients)
//   gn_bwd_stats/apply: GroupNorm(+SiLU) backward through the fused "normalise on load" path, virtual-concat aware
//   attn_bwd          : small-head self-attention core backward (recomputes the probabilities)
//   linear_bwd_*      : the tiny time-embedding MLP / time_emb_proj layers
//   conv_in/out small : gradients of the two degenerate convolutions (Cin = 1, Cout = 1)
// All are HBM- or latency-bound; fp32 with fp64 accumulation in the reductions.
#include "adm_kernels.h"

namespace adm {

__device__ __forceinline__ float sigmoid_f(float v) { return ADM_RCP(1.0f + __expf(-v)); }
__device__ __forceinline__ float silu_grad(float y) {  // d silu(y) / dy
  const float s = sigmoid_f(y);
  return s * (1.0f + y * (1.0f - s));
}
__device__ __forceinline__ double wave_sum_d(double v) {
  ADM_UNROLL
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
// Block-wide sum of two doubles (256 threads); every thread gets the result.
__device__ __forceinline__ void block_sum2(double& a, double& b, double (*red)[4]) {
  a = wave_sum_d(a); b = wave_sum_d(b);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { red[0][wave] = a; red[1][wave] = b; }
  __syncthreads();
  a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
}

// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sumpool2x2_kernel(const float* __restrict__ in, float* out, int H, int W,
                                                         long planes, int accumulate) {
  const int Ho = H / 2, Wo = W / 2;
  const long total = planes * Ho * Wo;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wo);
    const long r = i / Wo;
    const int y = (int)(r % Ho);
    const long pl = r / Ho;
    const float* p = in + (pl * H + 2 * y) * W + 2 * x;
    const float v = p[0] + p[1] + p[W] + p[W + 1];
    out[i] = accumulate ? out[i] + v : v;
  }
}

// dst[n][c][p] (+)= src[n][c][p] with independent batch strides (dst may be a channel slice of a wider tensor)
__global__ void __launch_bounds__(256) accumulate_kernel(float* dst, long dst_bs, const float* __restrict__ src,
                                                         long src_bs, long per_sample, int N, int accumulate) {
  // one sample per blockIdx.y: no 64-bit division per element; float4 when everything is 16-byte aligned
  const int n = blockIdx.y;
  const float* s = src + (long)n * src_bs;
  float* d = dst + (long)n * dst_bs;
  const long stride = (long)gridDim.x * blockDim.x;
  if (((per_sample | src_bs | dst_bs) & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
    const float4* s4 = reinterpret_cast<const float4*>(s);
    float4* d4 = reinterpret_cast<float4*>(d);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (per_sample >> 2); i += stride) {
      float4 v = s4[i];
      if (accumulate) { const float4 o = d4[i]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
      d4[i] = v;
    }
    return;
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += stride) {
    const float v = s[i];
    d[i] = accumulate ? d[i] + v : v;
  }
}

// out_nc[n*nc_stride + c] (=|+=) sum_p dy[n][c][p]  (NULL ok);  out_c[c] += sum_{n,p} dy (atomic; NULL ok)
__global__ void __launch_bounds__(256) chan_sums_kernel(const float* __restrict__ dy, int C, int HW,
                                                        float* out_nc, int nc_stride, int nc_accumulate,
                                                        float* out_c) {
  __shared__ double red[2][4];
  const int c = blockIdx.x, n = blockIdx.y;
  const float* p = dy + ((long)n * C + c) * HW;
  double s = 0.0, z = 0.0;
  if ((HW & 3) == 0) {
    const float4* p4 = reinterpret_cast<const float4*>(p);
    const int n4 = HW >> 2;
    int i = threadIdx.x;
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;                       // four loads in flight, four independent fp64 chains
    for (; i + 768 < n4; i += 1024) {
      const float4 a = p4[i], b = p4[i + 256], c4 = p4[i + 512], d = p4[i + 768];
      s += ((double)a.x + (double)a.y) + ((double)a.z + (double)a.w);
      s1 += ((double)b.x + (double)b.y) + ((double)b.z + (double)b.w);
      s2 += ((double)c4.x + (double)c4.y) + ((double)c4.z + (double)c4.w);
      s3 += ((double)d.x + (double)d.y) + ((double)d.z + (double)d.w);
    }
    for (; i < n4; i += 256) {
      const float4 v = p4[i];
      s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
    }
    s = (s + s1) + (s2 + s3);
  } else {
    for (int i = threadIdx.x; i < HW; i += 256) s += (double)p[i];
  }
  block_sum2(s, z, red);
  if (threadIdx.x == 0) {
    if (out_nc) {
      float* o = out_nc + (long)n * nc_stride + c;
      *o = nc_accumulate ? *o + (float)s : (float)s;
    }
    if (out_c) atomicAdd(out_c + c, (float)s);
  }
}

// ---------------------------------------------------------------------------------------------------------
// GroupNorm(+SiLU) backward, pass 1: per (n, group) s1 = mean(g*gamma), s2 = mean(g*gamma*xhat) over the group, and the
// affine gradients dgamma[c] += sum g*xhat, dbeta[c] += sum g (atomics over n). g = da * silu'(y) (or da).
// NT: streaming loads (the launcher sets it for passes over >= 64 MB: nothing of them survives in the L2 anyway)
template <bool NT>
__global__ void __launch_bounds__(256) gn_bwd_stats_kernel(const float* __restrict__ x1, int C1,
                                                           const float* __restrict__ x2, int C2,
                                                           const float* __restrict__ da, int HW, int groups,
                                                           const float* __restrict__ mean_rstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int act,
                                                           float* __restrict__ s12, float* dgamma, float* dbeta) {
  __shared__ double red[2][4];
  const int g = blockIdx.x, n = blockIdx.y;
  const int C = C1 + C2, cg = C / groups;
  const float mean = mean_rstd[((long)n * groups + g) * 2], rstd = mean_rstd[((long)n * groups + g) * 2 + 1];
  double S1 = 0.0, S2 = 0.0;
  for (int cl = 0; cl < cg; ++cl) {
    const int c = g * cg + cl;
    const float* xs = c < C1 ? x1 + ((long)n * C1 + c) * HW : x2 + ((long)n * C2 + (c - C1)) * HW;
    const float* ds = da + ((long)n * C + c) * HW;
    const float gm = gamma[c], bt = beta[c];
    double a = 0.0, b = 0.0;  // sum g, sum g*xhat
    if ((HW & 3) == 0) {      // float4 streaming, two independent loads per thread and iteration
      const float4* xs4 = reinterpret_cast<const float4*>(xs);
      const float4* ds4 = reinterpret_cast<const float4*>(ds);
      // eight loads in flight per thread (four iterations' x and da), then the arithmetic: one iteration at a time (two loads,
      // a transcendental chain, an fp64 accumulation) ran at 2.2 TB/s
      const int n4 = HW >> 2;
      for (int i0 = threadIdx.x; i0 < n4; i0 += 1024) {
        float4 xv[4], dv[4];
        ADM_UNROLL
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + 256 * u < n4 ? i0 + 256 * u : i0;          // past the end: re-read a valid element, contribution masked
          if (NT) { xv[u] = adm_ld_nt(xs4 + i); dv[u] = adm_ld_nt(ds4 + i); }
          else { xv[u] = xs4[i]; dv[u] = ds4[i]; }
        }
        ADM_UNROLL
        for (int u = 0; u < 4; ++u) {
          if (i0 + 256 * u >= n4) break;
          const float xe[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w}, de[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
          ADM_UNROLL
          for (int k = 0; k < 4; ++k) {
            const float xh = (xe[k] - mean) * rstd;
            float gy = de[k];
            if (act) gy *= silu_grad(xh * gm + bt);
            a += (double)gy;
            b += (double)gy * xh;
          }
        }
      }
    } else {
      for (int i = threadIdx.x; i < HW; i += 256) {
        const float xh = (xs[i] - mean) * rstd;
        float gy = ds[i];
        if (act) gy *= silu_grad(xh * gm + bt);
        a += (double)gy;
        b += (double)gy * xh;
      }
    }
    block_sum2(a, b, red);
    if (threadIdx.x == 0) {
      atomicAdd(dbeta + c, (float)a);
      atomicAdd(dgamma + c, (float)b);
    }
    S1 += a * gm;
    S2 += b * gm;
  }
  if (threadIdx.x == 0) {
    const double m = (double)cg * HW;
    s12[((long)n * groups + g) * 2] = (float)(S1 / m);
    s12[((long)n * groups + g) * 2 + 1] = (float)(S2 / m);
  }
}

// pass 2: dx = rstd * (g*gamma - s1 - xhat*s2), routed to the gradient buffers of x1 / x2 (each (=|+=)).
template <bool NT>
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const float* __restrict__ x1, int C1,
                                                           const float* __restrict__ x2, int C2,
                                                           const float* __restrict__ da, int HW, int groups,
                                                           const float* __restrict__ mean_rstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int act,
                                                           const float* __restrict__ s12, float* dx1, int acc1,
                                                           float* dx2, int acc2) {
  const int c = blockIdx.y, n = blockIdx.z;
  const int C = C1 + C2, cg = C / groups, g = c / cg;
  const float mean = mean_rstd[((long)n * groups + g) * 2], rstd = mean_rstd[((long)n * groups + g) * 2 + 1];
  const float s1 = s12[((long)n * groups + g) * 2], s2 = s12[((long)n * groups + g) * 2 + 1];
  const bool first = c < C1;
  const float* xs = first ? x1 + ((long)n * C1 + c) * HW : x2 + ((long)n * C2 + (c - C1)) * HW;
  float* dxs = first ? dx1 + ((long)n * C1 + c) * HW : dx2 + ((long)n * C2 + (c - C1)) * HW;
  const int acc = first ? acc1 : acc2;
  const float* ds = da + ((long)n * C + c) * HW;
  const float gm = gamma[c], bt = beta[c];
  if ((HW & 3) == 0) {
    const float4* xs4 = reinterpret_cast<const float4*>(xs);
    const float4* ds4 = reinterpret_cast<const float4*>(ds);
    float4* dx4 = reinterpret_cast<float4*>(dxs);
    // the (up to) four iterations of a thread issue all their loads first — 8..12 in flight instead of 2..3
    const int n4 = HW >> 2, step = gridDim.x * 256;
    for (int i0 = blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 4 * step) {
      float4 xv[4], dv[4], ov[4];
      ADM_UNROLL
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * step < n4 ? i0 + u * step : i0;
        if (NT) { xv[u] = adm_ld_nt(xs4 + i); dv[u] = adm_ld_nt(ds4 + i); }
        else { xv[u] = xs4[i]; dv[u] = ds4[i]; }
        ov[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (acc) ov[u] = NT ? adm_ld_nt(const_cast<const float4*>(dx4) + i) : dx4[i];   // uniform
      }
      ADM_UNROLL
      for (int u = 0; u < 4; ++u) {
        if (i0 + u * step >= n4) break;
        const float xe[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w}, de[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
        float r[4];
        ADM_UNROLL
        for (int k = 0; k < 4; ++k) {
          const float xh = (xe[k] - mean) * rstd;
          float gy = de[k];
          if (act) gy *= silu_grad(xh * gm + bt);
          r[k] = rstd * (gy * gm - s1 - xh * s2);
        }
        float4 o = ov[u];
        o.x += r[0]; o.y += r[1]; o.z += r[2]; o.w += r[3];
        if (NT) adm_st_nt(dx4 + i0 + u * step, o); else dx4[i0 + u * step] = o;
      }
    }
    return;
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const float xh = (xs[i] - mean) * rstd;
    float gy = ds[i];
    if (act) gy *= silu_grad(xh * gm + bt);
    const float v = rstd * (gy * gm - s1 - xh * s2);
    dxs[i] = acc ? dxs[i] + v : v;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Small-head attention core backward: qkv (N,3C,T), dout (N,C,T) -> dqkv (N,3C,T). One workgroup per (n, head).
// Phase A (thread = query i): softmax stats m_i, l_i, D_i = dout_i . out_i, and dq_i.
// Phase B (thread = key j)  : dk_j, dv_j. Probabilities are recomputed (flash-attention backward form).
template <int D>
__global__ void __launch_bounds__(256) attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                       float* __restrict__ dqkv, int C, int T, float scale) {
  ADM_DYN_SMEM(float, smem);
  float* Qs = smem;                 // [T][D]
  float* Ks = Qs + T * D;
  float* Vs = Ks + T * D;
  float* Os = Vs + T * D;           // dout [T][D]
  float* Ms = Os + T * D;           // m_i
  float* Ls = Ms + T;               // 1 / l_i
  float* Ds = Ls + T;               // D_i
  const int head = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const float* qb = qkv + ((long)n * 3 * C + head * D) * T;
  const float* kb = qb + (long)C * T;
  const float* vb = kb + (long)C * T;
  const float* ob = dout + ((long)n * C + head * D) * T;
  for (int e = tid; e < D * T; e += blockDim.x) {
    const int d = e / T, j = e - d * T;
    Qs[j * D + d] = qb[e]; Ks[j * D + d] = kb[e]; Vs[j * D + d] = vb[e]; Os[j * D + d] = ob[e];
  }
  __syncthreads();
  float* dqb = dqkv + ((long)n * 3 * C + head * D) * T;
  float* dkb = dqb + (long)C * T;
  float* dvb = dkb + (long)C * T;
  for (int i = tid; i < T; i += blockDim.x) {
    float q[D], go[D];
    ADM_UNROLL
    for (int d = 0; d < D; ++d) { q[d] = Qs[i * D + d]; go[d] = Os[i * D + d]; }
    float m = -3.0e38f;
    for (int j = 0; j < T; ++j) {
      float s = 0.f;
      ADM_UNROLL
      for (int d = 0; d < D; ++d) s = fmaf(q[d], Ks[j * D + d], s);
      m = fmaxf(m, s * scale);
    }
    float l = 0.f, dsum = 0.f;   // l = sum p~, dsum = sum p~ (go . v_j)
    for (int j = 0; j < T; ++j) {
      float s = 0.f, gv = 0.f;
      ADM_UNROLL
      for (int d = 0; d < D; ++d) { s = fmaf(q[d], Ks[j * D + d], s); gv = fmaf(go[d], Vs[j * D + d], gv); }
      const float pj = __expf(s * scale - m);
      l += pj;
      dsum = fmaf(pj, gv, dsum);
    }
    const float inv = 1.0f / l;
    const float Di = dsum * inv;   // = dout_i . out_i
    Ms[i] = m; Ls[i] = inv; Ds[i] = Di;
    float dq[D];
    ADM_UNROLL
    for (int d = 0; d < D; ++d) dq[d] = 0.f;
    for (int j = 0; j < T; ++j) {
      float s = 0.f, gv = 0.f;
      ADM_UNROLL
      for (int d = 0; d < D; ++d) { s = fmaf(q[d], Ks[j * D + d], s); gv = fmaf(go[d], Vs[j * D + d], gv); }
      const float p = __expf(s * scale - m) * inv;
      const float dsij = p * (gv - Di) * scale;
      ADM_UNROLL
      for (int d = 0; d < D; ++d) dq[d] = fmaf(dsij, Ks[j * D + d], dq[d]);
    }
    ADM_UNROLL
    for (int d = 0; d < D; ++d) dqb[(long)d * T + i] = dq[d];
  }
  __syncthreads();
  for (int j = tid; j < T; j += blockDim.x) {
    float k[D], v[D], dk[D], dv[D];
    ADM_UNROLL
    for (int d = 0; d < D; ++d) { k[d] = Ks[j * D + d]; v[d] = Vs[j * D + d]; dk[d] = 0.f; dv[d] = 0.f; }
    for (int i = 0; i < T; ++i) {
      float s = 0.f, gv = 0.f;
      ADM_UNROLL
      for (int d = 0; d < D; ++d) { s = fmaf(Qs[i * D + d], k[d], s); gv = fmaf(Os[i * D + d], v[d], gv); }
      const float p = __expf(s * scale - Ms[i]) * Ls[i];
      const float dsij = p * (gv - Ds[i]) * scale;
      ADM_UNROLL
      for (int d = 0; d < D; ++d) {
        dk[d] = fmaf(dsij, Qs[i * D + d], dk[d]);
        dv[d] = fmaf(p, Os[i * D + d], dv[d]);
      }
    }
    ADM_UNROLL
    for (int d = 0; d < D; ++d) { dkb[(long)d * T + j] = dk[d]; dvb[(long)d * T + j] = dv[d]; }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Tiny dense layers (time embedding MLP, time_emb_proj): Y[b][j] = bias[j] + sum_k W[j][k] X[b][k]
// dW[j][k] += sum_b dY[b][j] X'[b][k], db[j] += sum_b dY[b][j], with X' = silu(X) when x_silu (time_emb_proj input)
__global__ void __launch_bounds__(256) linear_bwd_weight_kernel(const float* __restrict__ dY, int ldy,
                                                                const float* __restrict__ X, int B, int J, int K,
                                                                int x_silu, float* dW, float* db) {
  const long total = (long)J * K;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e / K), k = (int)(e - (long)j * K);
    float acc = 0.f, accb = 0.f;
    for (int b = 0; b < B; ++b) {
      float xv = X[(long)b * K + k];
      if (x_silu) xv = xv * sigmoid_f(xv);
      const float g = dY[(long)b * ldy + j];
      acc = fmaf(g, xv, acc);
      accb += g;
    }
    dW[e] += acc;
    if (k == 0) db[j] += accb;
  }
}
// dX[b][k] = sum_j dY[b][j] W[j][k]  (then * silu'(X[b][k]) when x_silu: gradient w.r.t. the pre-activation input)
// One workgroup per (16 columns k, sample b): 16 lanes read 64 contiguous bytes of a W row, the 16 lane groups of the
// workgroup stride over the rows j and their partial sums meet in LDS.  (One thread per output with a serial loop over
// all J rows — 9.4k for the stacked time_emb_proj matrix — took 1.8 ms per call: 32 workgroups, pure latency.)
__global__ void __launch_bounds__(256) linear_bwd_input_kernel(const float* __restrict__ dY, int ldy,
                                                               const float* __restrict__ W, const float* __restrict__ X,
                                                               int B, int J, int K, int x_silu,
                                                               float* __restrict__ dX) {
  __shared__ float red[16][17];
  const int kk = threadIdx.x & 15, js = threadIdx.x >> 4;
  const int k = blockIdx.x * 16 + kk, b = blockIdx.y;
  float acc = 0.f;
  if (k < K) {
    const float* dyb = dY + (long)b * ldy;
    for (int j = js; j < J; j += 16) acc = fmaf(dyb[j], W[(long)j * K + k], acc);
  }
  red[js][kk] = acc;
  __syncthreads();
  if (threadIdx.x < 16 && k < K) {
    float s = 0.f;
    ADM_UNROLL
    for (int q = 0; q < 16; ++q) s += red[q][kk];
    if (x_silu) s *= silu_grad(X[(long)b * K + k]);
    dX[(long)b * K + k] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------
// conv_in class (Cin <= 4): dW[co][c][tap] += sum_{n,p} dy[n][co][p] x[n][c][p+tap]; one workgroup per (cout, n),
// register accumulators (compile-time Cin), one atomic per (tap, c) per workgroup.
// acc[t] += sum over the quad's four pixels of big[px] * small[y + t / 3 - 1][x0 + px + t % 3 - 1] (zero outside the plane): the
// small plane's three rows as one aligned float4 + the two edge words each — unconditional loads at clamped indices, masked
// afterwards (a load-or-not branch costs a vmcnt(0) round trip).  W % 4 == 0, x0 % 4 == 0.
__device__ __forceinline__ void quad_window_fma(const float4 big, const float* __restrict__ small, int y, int x0, int H, int W,
                                                float* acc9) {
  const float bq[4] = {big.x, big.y, big.z, big.w};
  ADM_UNROLL
  for (int ty = 0; ty < 3; ++ty) {
    const int r = y + ty - 1;
    const bool rok = r >= 0 && r < H;
    const float* row = small + (long)(rok ? r : y) * W;
    const float4 m = *reinterpret_cast<const float4*>(row + x0);
    const float lft = row[x0 > 0 ? x0 - 1 : x0], rgt = row[x0 + 4 < W ? x0 + 4 : x0];
    const float w6[6] = {rok && x0 > 0 ? lft : 0.f, rok ? m.x : 0.f, rok ? m.y : 0.f, rok ? m.z : 0.f, rok ? m.w : 0.f,
                         rok && x0 + 4 < W ? rgt : 0.f};
    ADM_UNROLL
    for (int tx = 0; tx < 3; ++tx) {
      float a = acc9[ty * 3 + tx];
      ADM_UNROLL
      for (int px = 0; px < 4; ++px) a = fmaf(bq[px], w6[px + tx], a);
      acc9[ty * 3 + tx] = a;
    }
  }
}

template <int CIN>
__global__ void __launch_bounds__(256) conv_small_cin_wgrad_kernel(const float* __restrict__ x, int H, int W,
                                                                   const float* __restrict__ dy, int Cout,
                                                                   float* dW /* (Cout,Cin,3,3) */) {
  __shared__ double red[2][4];
  const int co = blockIdx.x, n = blockIdx.y;
  const int HW = H * W;
  float acc[CIN * 9];
  ADM_UNROLL
  for (int k = 0; k < CIN * 9; ++k) acc[k] = 0.f;
  const float* dyp = dy + ((long)n * Cout + co) * HW;
  if ((W & 3) == 0 && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x)) & 15) == 0) {
    // four pixels of a row per thread: dy (the big tensor: Cout planes) streams as float4, 1 KiB per wave-load (round 4: the
    // one-pixel version below — a dword of dy, nine dwords of x, a division per pixel — read 537 MB in 370 us)
    const int W4 = W >> 2, nq = H * W4;
    for (int q = threadIdx.x; q < nq; q += 256) {
      const int y = q / W4, x0 = (q - y * W4) * 4;
      const float4 g = *reinterpret_cast<const float4*>(dyp + (long)y * W + x0);
      ADM_UNROLL
      for (int c = 0; c < CIN; ++c) quad_window_fma(g, x + ((long)n * CIN + c) * HW, y, x0, H, W, acc + c * 9);
    }
  } else
  for (int p = threadIdx.x; p < HW; p += 256) {
    const int y = p / W, xx = p - y * W;
    const float g = dyp[p];
    ADM_UNROLL
    for (int c = 0; c < CIN; ++c) {
      const float* xp = x + ((long)n * CIN + c) * HW;
      ADM_UNROLL
      for (int t = 0; t < 9; ++t) {
        const int gy = y + t / 3 - 1, gx = xx + t % 3 - 1;
        // unconditional load at a clamped index, masked afterwards: a per-element load-or-not branch costs a vmcnt(0) round trip each
        const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const float vl = xp[ok ? gy * W + gx : p];
        acc[c * 9 + t] = fmaf(g, ok ? vl : 0.f, acc[c * 9 + t]);
      }
    }
  }
  ADM_UNROLL
  for (int k = 0; k < CIN * 9; k += 2) {
    double a = acc[k], z = (k + 1 < CIN * 9) ? acc[k + 1] : 0.0;
    block_sum2(a, z, red);
    if (threadIdx.x == 0) {
      atomicAdd(dW + (long)co * CIN * 9 + k, (float)a);
      if (k + 1 < CIN * 9) atomicAdd(dW + (long)co * CIN * 9 + k + 1, (float)z);
    }
  }
}

// conv_out class (Cout <= 4) data gradient: da[n][c][p] = sum_co sum_tap dy[n][co][p - (tap - center)] w[co][c][tap]
// Write-bound (Cin planes out of Cout <= 4 planes in). One thread per four pixels of a row (float4 store), grid over (pixel
// block, input channel, sample): no 64-bit div/mod per element, the 18 dy values of a (co, row triple) loaded unconditionally at
// clamped indices and masked ONCE for 16 input channels (they do not depend on the channel), the 9 weights of (co, c)
// wave-uniform (scalar loads). The first version (flat index, four 64-bit divisions and nine load-or-not branches per element)
// took 0.67 ms for a 0.54 GB tensor.
constexpr int DG_CH = 16;     // input channels per thread: the dy values a thread holds do not depend on the channel
__global__ void __launch_bounds__(256) conv_small_cout_dgrad_kernel(const float* __restrict__ dy, int Cout, int N, int H,
                                                                    int W, const float* __restrict__ w /* (Cout,Cin,3,3) */,
                                                                    int Cin, float* __restrict__ da) {
  const int HW = H * W, c0 = blockIdx.y * DG_CH, n = blockIdx.z;
  const int W4 = (W + 3) >> 2;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;             // (row, group of four columns)
  if (q >= H * W4) return;
  const int y = q / W4, x0 = (q - y * W4) * 4;
  float acc[DG_CH][4];
  ADM_UNROLL
  for (int j = 0; j < DG_CH; ++j)
    ADM_UNROLL
    for (int px = 0; px < 4; ++px) acc[j][px] = 0.f;
  for (int co = 0; co < Cout; ++co) {
    const float* dp = dy + ((long)n * Cout + co) * HW;
    float v[3][6];                                                   // dy rows y - 1 .. y + 1, columns x0 - 1 .. x0 + 4
    ADM_UNROLL
    for (int r = 0; r < 3; ++r) {
      const int oy = y + r - 1;
      ADM_UNROLL
      for (int k = 0; k < 6; ++k) {
        const int ox = x0 + k - 1;
        const bool ok = oy >= 0 && oy < H && ox >= 0 && ox < W;
        const float t = dp[ok ? oy * W + ox : 0];
        v[r][k] = ok ? t : 0.f;
      }
    }
    // tap t of output pixel (oy, ox) touched input (oy + t/3 - 1, ox + t%3 - 1): input (y, x) gets dy(y - (t/3 - 1), x - (t%3 - 1))
    ADM_UNROLL
    for (int j = 0; j < DG_CH; ++j) {
      if (c0 + j < Cin) {                                            // uniform
        const float* wp = w + ((long)co * Cin + c0 + j) * 9;         // uniform -> scalar loads
        ADM_UNROLL
        for (int t = 0; t < 9; ++t) {
          const float wt = wp[t];
          ADM_UNROLL
          for (int px = 0; px < 4; ++px) acc[j][px] = fmaf(v[2 - t / 3][px + 2 - t % 3], wt, acc[j][px]);
        }
      }
    }
  }
  ADM_UNROLL
  for (int j = 0; j < DG_CH; ++j) {
    if (c0 + j >= Cin) break;
    float* dst = da + ((long)n * Cin + c0 + j) * HW + (long)y * W + x0;
    if (x0 + 3 < W && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0))
      *reinterpret_cast<float4*>(dst) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
    else
      for (int px = 0; px < 4 && x0 + px < W; ++px) dst[px] = acc[j][px];
  }
}

// conv_out class weight gradient with the fused GN+SiLU prologue recomputed: one workgroup per (input channel, n).
template <int COUT>
__global__ void __launch_bounds__(256) conv_small_cout_wgrad_kernel(const float* __restrict__ x, int Cin, int H, int W,
                                                                    const float* __restrict__ gn_scale,
                                                                    const float* __restrict__ gn_shift, int act,
                                                                    const float* __restrict__ dy,
                                                                    float* dW /* (Cout,Cin,3,3) */) {
  __shared__ double red[2][4];
  const int c = blockIdx.x, n = blockIdx.y;
  const int HW = H * W;
  float acc[COUT * 9];
  ADM_UNROLL
  for (int k = 0; k < COUT * 9; ++k) acc[k] = 0.f;
  const float* xp = x + ((long)n * Cin + c) * HW;
  const float sc = gn_scale ? gn_scale[(long)n * Cin + c] : 1.f, sh = gn_scale ? gn_shift[(long)n * Cin + c] : 0.f;
  if ((W & 3) == 0 && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x)) & 15) == 0) {
    // four pixels of a row per thread, the activated input (the big tensor: Cin planes) as float4.  a at pixel q is tap t of output
    // pixel q - offset(t): sum_q a[q] dy[q - off(t)] = the window sum of quad_window_fma with the tap index mirrored (t <-> 8 - t)
    float mir[COUT * 9];
    ADM_UNROLL
    for (int k = 0; k < COUT * 9; ++k) mir[k] = 0.f;
    const int W4 = W >> 2, nq = H * W4;
    for (int q = threadIdx.x; q < nq; q += 256) {
      const int y = q / W4, x0 = (q - y * W4) * 4;
      float4 a = *reinterpret_cast<const float4*>(xp + (long)y * W + x0);
      a.x = a.x * sc + sh; a.y = a.y * sc + sh; a.z = a.z * sc + sh; a.w = a.w * sc + sh;
      if (act) { a.x *= sigmoid_f(a.x); a.y *= sigmoid_f(a.y); a.z *= sigmoid_f(a.z); a.w *= sigmoid_f(a.w); }
      ADM_UNROLL
      for (int co = 0; co < COUT; ++co) quad_window_fma(a, dy + ((long)n * COUT + co) * HW, y, x0, H, W, mir + co * 9);
    }
    ADM_UNROLL
    for (int co = 0; co < COUT; ++co)
      ADM_UNROLL
      for (int t = 0; t < 9; ++t) acc[co * 9 + t] = mir[co * 9 + 8 - t];
  } else
  for (int p = threadIdx.x; p < HW; p += 256) {
    const int y = p / W, xx = p - y * W;
    float a = xp[p] * sc + sh;
    if (act) a = a * sigmoid_f(a);
    // a at (y,xx) is tap t of output pixel (y - (t/3-1), xx - (t%3-1))
    ADM_UNROLL
    for (int co = 0; co < COUT; ++co) {
      const float* dyp = dy + ((long)n * COUT + co) * HW;
      ADM_UNROLL
      for (int t = 0; t < 9; ++t) {
        const int oy = y - (t / 3 - 1), ox = xx - (t % 3 - 1);
        const bool ok = oy >= 0 && oy < H && ox >= 0 && ox < W;      // unconditional load, clamped index, masked (see above)
        const float gl = dyp[ok ? oy * W + ox : p];
        acc[co * 9 + t] = fmaf(a, ok ? gl : 0.f, acc[co * 9 + t]);
      }
    }
  }
  ADM_UNROLL
  for (int k = 0; k < COUT * 9; k += 2) {
    double a = acc[k], z = (k + 1 < COUT * 9) ? acc[k + 1] : 0.0;
    block_sum2(a, z, red);
    if (threadIdx.x == 0) {
      atomicAdd(dW + ((long)(k / 9) * Cin + c) * 9 + (k % 9), (float)a);
      if (k + 1 < COUT * 9) atomicAdd(dW + ((long)((k + 1) / 9) * Cin + c) * 9 + ((k + 1) % 9), (float)z);
    }
  }
}

static inline unsigned bgrid(long n) {
  long g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

int launch_sumpool2x2(const float* in, float* out, int H, int W, long planes, int accumulate, hipStream_t st) {
  ADM_REQUIRE(H % 2 == 0 && W % 2 == 0, "sumpool2x2: odd size");
  ADM_LAUNCH(sumpool2x2_kernel, dim3(bgrid(planes * (H / 2) * (W / 2))), dim3(256), 0, st, in, out, H, W, planes, accumulate);
  return ADM_CHECK_LAUNCH();
}
int launch_accumulate(float* dst, long dst_bs, const float* src, long src_bs, long per_sample, int N, int accumulate,
                      hipStream_t st) {
  ADM_LAUNCH(accumulate_kernel, dim3(bgrid((per_sample + 3) / 4), N), dim3(256), 0, st, dst, dst_bs, src, src_bs, per_sample, N,
             accumulate);
  return ADM_CHECK_LAUNCH();
}
int launch_chan_sums(const float* dy, int N, int C, int HW, float* out_nc, int nc_stride, int nc_accumulate, float* out_c,
                     hipStream_t st) {
  ADM_LAUNCH(chan_sums_kernel, dim3(C, N), dim3(256), 0, st, dy, C, HW, out_nc, nc_stride, nc_accumulate, out_c);
  return ADM_CHECK_LAUNCH();
}
// streaming loads / stores for the elementwise passes over a tensor of >= 64 MB (ADM_NT_STREAM=0: never; developer A/B)
bool gn_bwd_streaming(int N, int C, int HW) {
  static const int on = [] { const char* e = getenv("ADM_NT_STREAM"); return e ? atoi(e) : 1; }();
  return on && (long)N * C * HW * 4 >= (64L << 20);
}
int launch_gn_backward(const float* x1, int C1, const float* x2, int C2, const float* da, int N, int HW, int groups,
                       const float* mean_rstd, const float* gamma, const float* beta, int act, float* s12_scratch,
                       float* dgamma, float* dbeta, float* dx1, int acc1, float* dx2, int acc2, hipStream_t st) {
  if (x2 == nullptr) C2 = 0;
  const bool nt = gn_bwd_streaming(N, C1 + C2, HW);
  if (nt) ADM_LAUNCH(gn_bwd_stats_kernel<true>, dim3(groups, N), dim3(256), 0, st, x1, C1, x2, C2, da, HW, groups, mean_rstd, gamma,
                     beta, act, s12_scratch, dgamma, dbeta);
  else ADM_LAUNCH(gn_bwd_stats_kernel<false>, dim3(groups, N), dim3(256), 0, st, x1, C1, x2, C2, da, HW, groups, mean_rstd, gamma,
                  beta, act, s12_scratch, dgamma, dbeta);
  int gx = (HW + 4095) / 4096;   // 256 threads x float4 x 4 iterations per workgroup
  if (gx < 1) gx = 1;
  if (nt) ADM_LAUNCH(gn_bwd_apply_kernel<true>, dim3(gx, C1 + C2, N), dim3(256), 0, st, x1, C1, x2, C2, da, HW, groups, mean_rstd,
                     gamma, beta, act, (const float*)s12_scratch, dx1, acc1, dx2, acc2);
  else ADM_LAUNCH(gn_bwd_apply_kernel<false>, dim3(gx, C1 + C2, N), dim3(256), 0, st, x1, C1, x2, C2, da, HW, groups, mean_rstd,
                  gamma, beta, act, (const float*)s12_scratch, dx1, acc1, dx2, acc2);
  return ADM_CHECK_LAUNCH();
}
int launch_gn_backward_stats(const float* x1, int C1, const float* x2, int C2, const float* da, int N, int HW, int groups,
                             const float* mean_rstd, const float* gamma, const float* beta, int act, float* s12_scratch,
                             float* dgamma, float* dbeta, hipStream_t st) {
  if (x2 == nullptr) C2 = 0;
  if (gn_bwd_streaming(N, C1 + C2, HW))
    ADM_LAUNCH(gn_bwd_stats_kernel<true>, dim3(groups, N), dim3(256), 0, st, x1, C1, x2, C2, da, HW, groups, mean_rstd, gamma,
               beta, act, s12_scratch, dgamma, dbeta);
  else
    ADM_LAUNCH(gn_bwd_stats_kernel<false>, dim3(groups, N), dim3(256), 0, st, x1, C1, x2, C2, da, HW, groups, mean_rstd, gamma,
               beta, act, s12_scratch, dgamma, dbeta);
  return ADM_CHECK_LAUNCH();
}
int launch_attention_bwd(const float* qkv, const float* dout, float* dqkv, int N, int C, int T, int head_dim,
                         hipStream_t st) {
  ADM_REQUIRE(C % head_dim == 0, "attention_bwd: C not divisible by head_dim");
  const int heads = C / head_dim;
  const int bs = T >= 256 ? 256 : ((T + 63) / 64) * 64;
  const size_t smem = sizeof(float) * ((size_t)4 * T * head_dim + 3 * T);
  ADM_REQUIRE(smem <= 64 * 1024, "attention_bwd: head slab exceeds 64 KiB of LDS");
  const float scale = 1.0f / sqrtf((float)head_dim);
#define ADM_ATTB_CASE(DD)                                                                                   \
  if (head_dim == DD) {                                                                                     \
    ADM_LAUNCH((attn_bwd_kernel<DD>), dim3(heads, N), dim3(bs), smem, st, qkv, dout, dqkv, C, T, scale);     \
    return ADM_CHECK_LAUNCH();                                                                              \
  }
  ADM_ATTB_CASE(8) ADM_ATTB_CASE(4) ADM_ATTB_CASE(16) ADM_ATTB_CASE(32)
#undef ADM_ATTB_CASE
  ADM_FAIL("attention_bwd: unsupported head_dim (4/8/16/32)");
}
int launch_linear_bwd(const float* dY, int ldy, const float* X, const float* W, int B, int J, int K, int x_silu, float* dW,
                      float* db, float* dX, hipStream_t st) {
  if (dW) ADM_LAUNCH(linear_bwd_weight_kernel, dim3(bgrid((long)J * K)), dim3(256), 0, st, dY, ldy, X, B, J, K, x_silu, dW, db);
  if (dX) ADM_LAUNCH(linear_bwd_input_kernel, dim3((K + 15) / 16, B), dim3(256), 0, st, dY, ldy, W, X, B, J, K, x_silu, dX);
  return ADM_CHECK_LAUNCH();
}
int launch_conv_small_cin_wgrad(const float* x, int Cin, int N, int H, int W, const float* dy, int Cout, float* dW,
                                hipStream_t st) {
  ADM_REQUIRE(Cin >= 1 && Cin <= 4, "conv_small_cin_wgrad: Cin <= 4");
  dim3 grid(Cout, N), block(256);
  if (Cin == 1) { ADM_LAUNCH((conv_small_cin_wgrad_kernel<1>), grid, block, 0, st, x, H, W, dy, Cout, dW); }
  else if (Cin == 2) { ADM_LAUNCH((conv_small_cin_wgrad_kernel<2>), grid, block, 0, st, x, H, W, dy, Cout, dW); }
  else if (Cin == 3) { ADM_LAUNCH((conv_small_cin_wgrad_kernel<3>), grid, block, 0, st, x, H, W, dy, Cout, dW); }
  else { ADM_LAUNCH((conv_small_cin_wgrad_kernel<4>), grid, block, 0, st, x, H, W, dy, Cout, dW); }
  return ADM_CHECK_LAUNCH();
}
int launch_conv_small_cout_bwd(const float* x, int Cin, int N, int H, int W, const float* gn_scale, const float* gn_shift,
                               int act, const float* w, const float* dy, int Cout, float* da, float* dW, hipStream_t st) {
  ADM_REQUIRE(Cout <= 4, "conv_small_cout_bwd: Cout <= 4");
  if (da) ADM_LAUNCH(conv_small_cout_dgrad_kernel, dim3((unsigned)((H * ((W + 3) / 4) + 255) / 256), (unsigned)((Cin + 15) / 16), (unsigned)N), dim3(256), 0, st, dy,
                     Cout, N, H, W, w, Cin, da);
  if (dW) {
    dim3 grid(Cin, N), block(256);
    if (Cout == 1) { ADM_LAUNCH((conv_small_cout_wgrad_kernel<1>), grid, block, 0, st, x, Cin, H, W, gn_scale, gn_shift, act, dy, dW); }
    else if (Cout == 2) { ADM_LAUNCH((conv_small_cout_wgrad_kernel<2>), grid, block, 0, st, x, Cin, H, W, gn_scale, gn_shift, act, dy, dW); }
    else if (Cout == 3) { ADM_LAUNCH((conv_small_cout_wgrad_kernel<3>), grid, block, 0, st, x, Cin, H, W, gn_scale, gn_shift, act, dy, dW); }
    else { ADM_LAUNCH((conv_small_cout_wgrad_kernel<4>), grid, block, 0, st, x, Cin, H, W, gn_scale, gn_shift, act, dy, dW); }
  }
  return ADM_CHECK_LAUNCH();
}

}  // namespace adm
