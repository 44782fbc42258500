// k_groupnorm.hip — GroupNorm statistics -> per-(n,c) scale/shift (HBM-bound read-only pass).
// torch.nn.GroupNorm as used by ResnetBlock2D.norm1/norm2, conv_norm_out and the attention group_norm
// (SURVEY.md §2.2 "GroupNorm(32)+SiLU", §8(a) U3/U6/U8). The normalisation itself (x*scale+shift, SiLU) is
// applied in the consumer convolution's LDS-fill path, so the activation is read once here and once there,
// never written. Input may be a virtual channel concat (x1|x2) — a group can straddle the seam.
// One workgroup per (n, group); fp64 accumulation; wavefront shuffles (64 lanes) then LDS across the 4 waves.
// Algorithmic bytes: 4 B per input element.
#include "adm_kernels.h"
#include <atomic>

namespace adm {

__device__ __forceinline__ double wave_sum(double v) {
  ADM_UNROLL
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

__global__ void __launch_bounds__(256) gn_stats_kernel(const float* __restrict__ x1, int C1,
                                                       const float* __restrict__ x2, int C2, int HW, int groups,
                                                       float eps, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ scale,
                                                       float* __restrict__ shift, float* __restrict__ mean_rstd) {
  const int g = blockIdx.x, n = blockIdx.y;
  const int C = C1 + C2, cg = C / groups;
  const int tid = threadIdx.x;
  const long total = (long)cg * HW;
  double s = 0.0, ss = 0.0;
  const int HW4 = HW >> 2;
  if ((HW & 3) == 0 && (HW4 >= 256 || (256 % HW4) == 0)) {
    // channel-major walk without per-element divisions: `tpc` threads stream one channel plane with float4 loads (4 in
    // flight per thread), 256 / tpc channel planes at a time
    const int tpc = HW4 >= 256 ? 256 : HW4;
    const int sub = tid % tpc, cstep = 256 / tpc;
    for (int cl = tid / tpc; cl < cg; cl += cstep) {
      const int c = g * cg + cl;
      const float4* src = reinterpret_cast<const float4*>(c < C1 ? x1 + ((long)n * C1 + c) * HW
                                                                  : x2 + ((long)n * C2 + (c - C1)) * HW);
      int p = sub;
      for (; p + 3 * tpc < HW4; p += 4 * tpc) {
        const float4 v0 = src[p], v1 = src[p + tpc], v2 = src[p + 2 * tpc], v3 = src[p + 3 * tpc];
        s += ((double)v0.x + (double)v0.y + (double)v0.z + (double)v0.w) + ((double)v1.x + (double)v1.y + (double)v1.z + (double)v1.w) +
             ((double)v2.x + (double)v2.y + (double)v2.z + (double)v2.w) + ((double)v3.x + (double)v3.y + (double)v3.z + (double)v3.w);
        ss += ((double)v0.x * v0.x + (double)v0.y * v0.y + (double)v0.z * v0.z + (double)v0.w * v0.w) +
              ((double)v1.x * v1.x + (double)v1.y * v1.y + (double)v1.z * v1.z + (double)v1.w * v1.w) +
              ((double)v2.x * v2.x + (double)v2.y * v2.y + (double)v2.z * v2.z + (double)v2.w * v2.w) +
              ((double)v3.x * v3.x + (double)v3.y * v3.y + (double)v3.z * v3.z + (double)v3.w * v3.w);
      }
      for (; p < HW4; p += tpc) {
        const float4 v = src[p];
        s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
        ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
      }
    }
  } else if ((HW & 3) == 0) {
    for (long e = (long)tid * 4; e < total; e += 256 * 4) {
      const int cl = (int)(e / HW);
      const int p = (int)(e - (long)cl * HW);
      const int c = g * cg + cl;
      const float* src = c < C1 ? x1 + ((long)n * C1 + c) * HW : x2 + ((long)n * C2 + (c - C1)) * HW;
      const float4 v = *reinterpret_cast<const float4*>(src + p);
      s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
      ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
  } else {
    for (long e = tid; e < total; e += 256) {
      const int cl = (int)(e / HW);
      const int p = (int)(e - (long)cl * HW);
      const int c = g * cg + cl;
      const float* src = c < C1 ? x1 + ((long)n * C1 + c) * HW : x2 + ((long)n * C2 + (c - C1)) * HW;
      const double v = src[p];
      s += v;
      ss += v * v;
    }
  }
  s = wave_sum(s);
  ss = wave_sum(ss);
  __shared__ double red[2][4];
  __shared__ float stat[2];
  const int wave = tid >> 6, lane = tid & 63;
  if (lane == 0) { red[0][wave] = s; red[1][wave] = ss; }
  __syncthreads();
  if (tid == 0) {
    const double S = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const double SS = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const double mean = S / (double)total;
    double var = SS / (double)total - mean * mean;
    if (var < 0.0) var = 0.0;
    stat[0] = (float)mean;
    stat[1] = (float)(1.0 / sqrt(var + (double)eps));
    if (mean_rstd) {  // kept for the backward pass (training)
      mean_rstd[((long)n * groups + g) * 2] = stat[0];
      mean_rstd[((long)n * groups + g) * 2 + 1] = stat[1];
    }
  }
  __syncthreads();
  const float mean = stat[0], rstd = stat[1];
  for (int cl = tid; cl < cg; cl += 256) {
    const int c = g * cg + cl;
    const float sc = rstd * gamma[c];  // same form as ATen's CPU kernel: y = x*scale + (beta - scale*mean)
    scale[(long)n * C + c] = sc;
    shift[(long)n * C + c] = -sc * mean + beta[c];
  }
}

// GroupNorm statistics from per-tile partial sums that the PRODUCING convolutions wrote in their epilogues
// (adm_conv_args.stats_out: fp64 (sum, sum of squares) per (sample, channel, pixel tile)) — the read-only pass over the
// activation disappears: 16 bytes per 128-pixel tile instead of 512. One workgroup per (sample, group); the partials of
// the group's channels are summed in a fixed order (thread-strided, then the same shuffle / LDS tree as gn_stats_kernel), so
// the result is deterministic. Virtual concat: channels >= C1 come from the second producer's buffer.
__global__ void __launch_bounds__(256) gn_finalize_kernel(const double* __restrict__ st1, int C1, int tiles1,
                                                          const double* __restrict__ st2, int C2, int tiles2, int HW,
                                                          int groups, float eps, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ scale,
                                                          float* __restrict__ shift, float* __restrict__ mean_rstd) {
  const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int C = C1 + C2, cg = C / groups;
  double s = 0.0, ss = 0.0;
  for (int cl = 0; cl < cg; ++cl) {
    const int c = g * cg + cl;
    const double2* src;
    int tiles;
    if (c < C1) { src = reinterpret_cast<const double2*>(st1) + ((long)n * C1 + c) * tiles1; tiles = tiles1; }
    else { src = reinterpret_cast<const double2*>(st2) + ((long)n * C2 + (c - C1)) * tiles2; tiles = tiles2; }
    for (int t = tid; t < tiles; t += 256) { const double2 v = src[t]; s += v.x; ss += v.y; }
  }
  s = wave_sum(s);
  ss = wave_sum(ss);
  __shared__ double red[2][4];
  __shared__ float stat[2];
  const int wave = tid >> 6, lane = tid & 63;
  if (lane == 0) { red[0][wave] = s; red[1][wave] = ss; }
  __syncthreads();
  if (tid == 0) {
    const double total = (double)cg * HW;
    const double S = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const double SS = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const double mean = S / total;
    double var = SS / total - mean * mean;
    if (var < 0.0) var = 0.0;
    stat[0] = (float)mean;
    stat[1] = (float)(1.0 / sqrt(var + (double)eps));
    if (mean_rstd) {                                     // training: kept for the backward pass, as gn_stats_kernel does
      mean_rstd[((long)n * groups + g) * 2] = stat[0];
      mean_rstd[((long)n * groups + g) * 2 + 1] = stat[1];
    }
  }
  __syncthreads();
  const float mean = stat[0], rstd = stat[1];
  for (int cl = tid; cl < cg; cl += 256) {
    const int c = g * cg + cl;
    const float sc = rstd * gamma[c];
    scale[(long)n * C + c] = sc;
    shift[(long)n * C + c] = -sc * mean + beta[c];
  }
}

int launch_groupnorm_finalize(const double* st1, int C1, int tiles1, const double* st2, int C2, int tiles2, int N, int HW,
                              int groups, float eps, const float* gamma, const float* beta, float* scale, float* shift,
                              hipStream_t st, float* mean_rstd) {
  if (st2 == nullptr) { C2 = 0; tiles2 = 0; }
  ADM_REQUIRE(st1 && tiles1 > 0 && (C2 == 0 || tiles2 > 0), "groupnorm_finalize: missing partial sums");
  ADM_REQUIRE((C1 + C2) % groups == 0, "groupnorm: channels not divisible by groups");
  ADM_LAUNCH(gn_finalize_kernel, dim3(groups, N), dim3(256), 0, st, st1, C1, tiles1, st2, C2, tiles2, HW, groups, eps, gamma,
             beta, scale, shift, mean_rstd);
  return ADM_CHECK_LAUNCH();
}

// Split-K finish + GroupNorm statistics in one pass (the latency regime: planes of <= 8x8 pixels, where a forward is a chain of
// ~5 us launches): workgroup (group, sample) completes ITS slice of the split convolution's output — slabs added in order with
// bias / per-sample term / residual, the arithmetic of ksplit_finish_kernel, so the tensor is bit-identical to the two-launch
// path — and, holding every value of the group, leaves the scale / shift of the GroupNorm that reads the tensor next
// (gn_stats_kernel's formulas).  Replaces ksplit_finish + gn_stats: one launch instead of two, (group, sample) workgroups instead
// of total / 1024, slab loads eight at a time.
__global__ void __launch_bounds__(256) ksplit_finish_gn_kernel(const float* __restrict__ part, int S, long part_stride,
                                                               const float* __restrict__ bias, const float* __restrict__ chan_add,
                                                               int chan_add_stride, const float* residual, float* out, int Cout,
                                                               int HW, int groups, float eps, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ scale,
                                                               float* __restrict__ shift, float* __restrict__ mean_rstd) {
  const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int cg = Cout / groups;
  const int total = cg * HW;
  const long base = ((long)n * Cout + (long)g * cg) * HW;
  double s = 0.0, ss = 0.0;
  if ((HW & 3) == 0) {
    for (int e4 = tid * 4; e4 < total; e4 += 256 * 4) {
      const long e = base + e4;
      const int co = g * cg + e4 / HW;
      float b = bias[co];
      if (chan_add != nullptr) b += chan_add[(long)n * chan_add_stride + co];
      float4 v = *reinterpret_cast<const float4*>(part + e);
#pragma unroll 8
      for (int s2 = 1; s2 < S; ++s2) {
        const float4 q = *reinterpret_cast<const float4*>(part + (long)s2 * part_stride + e);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      }
      v.x += b; v.y += b; v.z += b; v.w += b;
      if (residual != nullptr) {
        const float4 q = *reinterpret_cast<const float4*>(residual + e);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      }
      *reinterpret_cast<float4*>(out + e) = v;
      s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
      ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
  } else {
    for (int el = tid; el < total; el += 256) {
      const long e = base + el;
      const int co = g * cg + el / HW;
      float v = part[e];
#pragma unroll 8
      for (int s2 = 1; s2 < S; ++s2) v += part[(long)s2 * part_stride + e];
      v += chan_add != nullptr ? bias[co] + chan_add[(long)n * chan_add_stride + co] : bias[co];
      if (residual != nullptr) v += residual[e];
      out[e] = v;
      s += (double)v;
      ss += (double)v * v;
    }
  }
  s = wave_sum(s);
  ss = wave_sum(ss);
  __shared__ double red[2][4];
  __shared__ float stat[2];
  const int wave = tid >> 6, lane = tid & 63;
  if (lane == 0) { red[0][wave] = s; red[1][wave] = ss; }
  __syncthreads();
  if (tid == 0) {
    const double Sm = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const double SS = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const double mean = Sm / (double)total;
    double var = SS / (double)total - mean * mean;
    if (var < 0.0) var = 0.0;
    stat[0] = (float)mean;
    stat[1] = (float)(1.0 / sqrt(var + (double)eps));
    if (mean_rstd) {
      mean_rstd[((long)n * groups + g) * 2] = stat[0];
      mean_rstd[((long)n * groups + g) * 2 + 1] = stat[1];
    }
  }
  __syncthreads();
  const float mean = stat[0], rstd = stat[1];
  for (int cl = tid; cl < cg; cl += 256) {
    const int c = g * cg + cl;
    const float sc = rstd * gamma[c];
    scale[(long)n * Cout + c] = sc;
    shift[(long)n * Cout + c] = -sc * mean + beta[c];
  }
}

// the request a caller (the executor) leaves for the NEXT launch_conv2d on this thread: "the tensor you write is read by this
// GroupNorm"; a split-K launch that can honour it (finish pass = one workgroup per (group, sample)) takes it and says so
static thread_local const GnFuse* g_gn_request = nullptr;
static thread_local bool g_gn_taken = false;
void conv_gn_fuse_request(const GnFuse* f) { g_gn_request = f; g_gn_taken = false; }
bool conv_gn_fuse_taken() { return g_gn_taken; }
static std::atomic<int> g_gn_fuse_finish{-1};     // option "gn_fuse_finish": -1 = ADM_GN_FUSE_FINISH from the environment (default 1)
void set_gn_fuse_finish(int v) { g_gn_fuse_finish.store(v); }
const GnFuse* conv_gn_fuse_pending(int Cout) {
  static const int env = [] { const char* e = getenv("ADM_GN_FUSE_FINISH"); return e ? atoi(e) : 1; }();
  const int opt = g_gn_fuse_finish.load();
  const bool on = (opt < 0 ? env : opt) != 0;
  const GnFuse* f = g_gn_request;
  return (on && f != nullptr && f->groups > 0 && Cout % f->groups == 0) ? f : nullptr;
}

int launch_ksplit_finish_gn(const float* part, int S, long part_stride, const float* bias, const float* chan_add, int chan_add_stride,
                            const float* residual, float* out, int N, int Cout, int HW, const GnFuse& f, hipStream_t st) {
  ADM_REQUIRE(bias != nullptr && Cout % f.groups == 0 && f.scale && f.shift && f.gamma && f.beta, "ksplit_finish_gn: arguments");
  ADM_LAUNCH(ksplit_finish_gn_kernel, dim3(f.groups, N), dim3(256), 0, st, part, S, part_stride, bias, chan_add, chan_add_stride,
             residual, out, Cout, HW, f.groups, f.eps, f.gamma, f.beta, f.scale, f.shift, f.mean_rstd);
  g_gn_taken = true;
  return ADM_CHECK_LAUNCH();
}

int launch_groupnorm_stats(const float* x1, int C1, const float* x2, int C2, int N, int HW, int groups, float eps,
                           const float* gamma, const float* beta, float* scale, float* shift, hipStream_t st,
                           float* mean_rstd) {
  if (x2 == nullptr) C2 = 0;
  ADM_REQUIRE((C1 + C2) % groups == 0, "groupnorm: channels not divisible by groups");
  ADM_LAUNCH(gn_stats_kernel, dim3(groups, N), dim3(256), 0, st, x1, C1, x2, C2, HW, groups, eps, gamma, beta, scale,
             shift, mean_rstd);
  return ADM_CHECK_LAUNCH();
}

}  // namespace adm
