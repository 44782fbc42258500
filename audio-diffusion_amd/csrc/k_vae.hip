// k_vae.hip — small kernels specific to AutoencoderKL:
//   * channel-axis softmax + transpose for the mid-block attention (1 head, d = 512, T = 1024): the two products
//     Q^T K and P V run on the MFMA 1x1-conv kernel with per-sample "weights" (K resp. V^T), so only the softmax
//     over keys (the channel axis of the (T_keys, T_queries) score tensor) and a (C,T)->(T,C) transpose remain;
//   * DiagonalGaussianDistribution.sample: mean + exp(0.5*clamp(logvar,-30,20)) * noise, times an output scale
//     (the 0.18215 the reference hard-codes, pipeline_audio_diffusion.py:147);
//   * scale: x * s (pipeline_audio_diffusion.py:189).
// All HBM/L2-bound elementwise or strided passes over <= 4 MB per sample.
#include "adm_kernels.h"

namespace adm {

// s: (N, J, T) scores with key index j on the channel axis; softmax over j for every (n, t), scaled by `scale`.
__global__ void __launch_bounds__(256) softmax_channels_kernel(float* __restrict__ s, int J, int T, float scale) {
  const int n = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  float* p = s + (long)n * J * T + t;
  float m = -3.0e38f;
  for (int j = 0; j < J; ++j) m = fmaxf(m, p[(long)j * T] * scale);
  float l = 0.f;
  for (int j = 0; j < J; ++j) {
    const float e = __expf(p[(long)j * T] * scale - m);
    p[(long)j * T] = e;
    l += e;
  }
  const float inv = 1.0f / l;
  for (int j = 0; j < J; ++j) p[(long)j * T] *= inv;
}

// in: (N, C, T) slice with batch stride in_bs -> out: (N, T, C)
__global__ void __launch_bounds__(256) transpose_ct_kernel(const float* __restrict__ in, long in_bs,
                                                           float* __restrict__ out, int C, int T) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8)
    tile[r][tx] = (c0 + r < C && t0 + tx < T) ? in[(long)n * in_bs + (long)(c0 + r) * T + t0 + tx] : 0.f;
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (t0 + r < T && c0 + tx < C) out[((long)n * T + t0 + r) * C + c0 + tx] = tile[tx][r];
}

// moments: (N, 2*Cz, HW) = [mean | logvar]; out (N, Cz, HW) = (mean + exp(0.5*clamp(logvar)) * noise) * out_scale
__global__ void __launch_bounds__(256) gaussian_sample_kernel(const float* __restrict__ moments,
                                                              const float* __restrict__ noise, float* __restrict__ out,
                                                              int Cz, long HW, float out_scale, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long per = (long)Cz * HW;
    const long n = i / per, r = i - n * per;
    const float mean = moments[n * 2 * per + r];
    float logvar = moments[n * 2 * per + per + r];
    logvar = fminf(fmaxf(logvar, -30.0f), 20.0f);
    const float std_ = expf(0.5f * logvar);
    out[i] = (mean + std_ * (noise ? noise[i] : 0.f)) * out_scale;
  }
}

__global__ void __launch_bounds__(256) scale_kernel(const float* __restrict__ x, float* __restrict__ out, float s, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = s * x[i];
}

static inline unsigned egrid(long n) {
  long g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

int launch_softmax_channels(float* s, int N, int J, int T, float scale, hipStream_t st) {
  ADM_LAUNCH(softmax_channels_kernel, dim3(ceil_div(T, 256), N), dim3(256), 0, st, s, J, T, scale);
  return ADM_CHECK_LAUNCH();
}
int launch_transpose_ct(const float* in, long in_bs, float* out, int N, int C, int T, hipStream_t st) {
  ADM_LAUNCH(transpose_ct_kernel, dim3(ceil_div(T, 32), ceil_div(C, 32), N), dim3(256), 0, st, in, in_bs, out, C, T);
  return ADM_CHECK_LAUNCH();
}
int launch_gaussian_sample(const float* moments, const float* noise, float* out, int N, int Cz, long HW, float out_scale,
                           hipStream_t st) {
  const long total = (long)N * Cz * HW;
  ADM_LAUNCH(gaussian_sample_kernel, dim3(egrid(total)), dim3(256), 0, st, moments, noise, out, Cz, HW, out_scale, total);
  return ADM_CHECK_LAUNCH();
}
int launch_scale(const float* x, float* out, float s, long n, hipStream_t st) {
  ADM_LAUNCH(scale_kernel, dim3(egrid(n)), dim3(256), 0, st, x, out, s, n);
  return ADM_CHECK_LAUNCH();
}

}  // namespace adm
