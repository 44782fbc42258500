// k_audio_encoder.hip — the reference's AudioEncoder (audiodiffusion/audio_encoder.py:62-84), the model that turns the mel
// slices of a track into the 100-d `encoding` the conditional UNet attends to (scripts/train_unet.py:93-94,158):
//   3 x [depthwise 3x3 (no bias) -> pointwise 1x1 (+bias) -> LeakyReLU(0.2) -> BatchNorm2d(eval) -> MaxPool 2x2]
//   -> flatten in NHWC order -> Linear(41472, 1024) -> LeakyReLU(0.2) -> BatchNorm1d(eval) -> Linear(1024, 100).
// Inference only (encode() runs under no_grad / eval, :86-88: Dropout is the identity, BatchNorm uses running statistics,
// folded by the host binding into one scale/shift per channel).  Small, HBM-bound work — direct kernels, lanes along W.
#include "adm_kernels.h"

namespace adm {

// depthwise 3x3, padding 1, no bias: (N, C, H, W) -> (N, C, H, W); weight (C, 1, 3, 3)
__global__ void __launch_bounds__(256) depthwise3x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           float* __restrict__ y, int C, int H, int W) {
  const int px = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y, n = blockIdx.z;
  if (px >= H * W) return;
  const int oy = px / W, ox = px - oy * W;
  const float* xp = x + ((long)n * C + c) * H * W;
  const float* wp = w + (long)c * 9;
  float acc = 0.f;
  ADM_UNROLL
  for (int t = 0; t < 9; ++t) {
    const int gy = oy + t / 3 - 1, gx = ox + t % 3 - 1;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) acc = fmaf(wp[t], xp[gy * W + gx], acc);
  }
  y[((long)n * C + c) * H * W + px] = acc;
}

// pointwise 1x1 (+bias) -> LeakyReLU(slope) -> per-channel affine (BatchNorm eval) -> MaxPool 2x2 (floor):
// (N, Ci, H, W) -> (N, Co, H/2, W/2); one lane per pooled output, weight (Co, Ci)
__global__ void __launch_bounds__(256) pointwise_act_bn_pool_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                    const float* __restrict__ bias,
                                                                    const float* __restrict__ bn_scale,
                                                                    const float* __restrict__ bn_shift, float slope,
                                                                    float* __restrict__ y, int Ci, int Co, int H, int W) {
  const int Hp = H >> 1, Wp = W >> 1;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int co = blockIdx.y, n = blockIdx.z;
  if (q >= Hp * Wp) return;
  const int py = q / Wp, pxx = q - py * Wp;
  const float* xp = x + (long)n * Ci * H * W + (long)(2 * py) * W + 2 * pxx;
  const float* wp = w + (long)co * Ci;
  float a00 = bias[co], a01 = a00, a10 = a00, a11 = a00;
  for (int ci = 0; ci < Ci; ++ci) {
    const float* p = xp + (long)ci * H * W;
    const float wv = wp[ci];
    a00 = fmaf(wv, p[0], a00); a01 = fmaf(wv, p[1], a01);
    a10 = fmaf(wv, p[W], a10); a11 = fmaf(wv, p[W + 1], a11);
  }
  const float sc = bn_scale[co], sh = bn_shift[co];
  auto f = [&](float v) { v = v > 0.f ? v : v * slope; return v * sc + sh; };
  y[((long)n * Co + co) * Hp * Wp + q] = fmaxf(fmaxf(f(a00), f(a01)), fmaxf(f(a10), f(a11)));
}

int launch_sepconv_block(const float* x, const float* dw, const float* pw, const float* pb, const float* bn_scale,
                         const float* bn_shift, float slope, float* tmp, float* y, int N, int Ci, int Co, int H, int W,
                         hipStream_t st) {
  ADM_REQUIRE(H >= 2 && W >= 2, "sepconv_block: input smaller than the 2x2 pooling window");
  ADM_LAUNCH(depthwise3x3_kernel, dim3(ceil_div(H * W, 256), Ci, N), dim3(256), 0, st, x, dw, tmp, Ci, H, W);
  ADM_LAUNCH(pointwise_act_bn_pool_kernel, dim3(ceil_div((H / 2) * (W / 2), 256), Co, N), dim3(256), 0, st,
             (const float*)tmp, pw, pb, bn_scale, bn_shift, slope, y, Ci, Co, H, W);
  return ADM_CHECK_LAUNCH();
}

// y[n][j] = post(b[j] + sum_k W[j][k] * x[n][k]),  post = LeakyReLU(slope) then per-output affine (both optional).
// x is read through a (channels, pixels) transpose when hwc_C > 0: flat index k = pixel * hwc_C + c addresses the NCHW
// tensor x[n][c][pixel] (the reference flattens x.permute(0, 2, 3, 1), audio_encoder.py:55).
// One workgroup per output j: every weight is read once and used for up to 8 samples (the 41472 x 1024 matrix is 170 MB —
// the encoder's only sizeable stream).
template <int NB>
__global__ void __launch_bounds__(256) dense_act_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                        const float* __restrict__ b, const float* __restrict__ post_scale,
                                                        const float* __restrict__ post_shift, float slope, int leaky,
                                                        float* __restrict__ y, int N, int K, int J, int hwc_C) {
  __shared__ float red[NB][256];
  const int j = blockIdx.x, n0 = blockIdx.y * NB, tid = threadIdx.x;
  const float* wr = W + (long)j * K;
  const int P = hwc_C > 0 ? K / hwc_C : 0;
  float acc[NB];
  ADM_UNROLL
  for (int i = 0; i < NB; ++i) acc[i] = 0.f;
  for (int k = tid; k < K; k += 256) {
    const float wv = wr[k];
    long xi = k;
    if (hwc_C > 0) { const int pix = k / hwc_C, c = k - pix * hwc_C; xi = (long)c * P + pix; }
    ADM_UNROLL
    for (int i = 0; i < NB; ++i)
      if (n0 + i < N) acc[i] = fmaf(wv, x[(long)(n0 + i) * K + xi], acc[i]);
  }
  ADM_UNROLL
  for (int i = 0; i < NB; ++i) red[i][tid] = acc[i];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      ADM_UNROLL
      for (int i = 0; i < NB; ++i) red[i][tid] += red[i][tid + s];
    }
    __syncthreads();
  }
  if (tid < NB && n0 + tid < N) {
    float v = red[tid][0] + b[j];
    if (leaky) v = v > 0.f ? v : v * slope;
    if (post_scale) v = v * post_scale[j] + post_shift[j];
    y[(long)(n0 + tid) * J + j] = v;
  }
}

int launch_dense_act(const float* x, const float* W, const float* b, const float* post_scale, const float* post_shift,
                     float slope, int leaky, float* y, int N, int K, int J, int hwc_C, hipStream_t st) {
  ADM_REQUIRE(hwc_C == 0 || K % hwc_C == 0, "dense_act: K not divisible by the channel count of the NHWC flatten");
  ADM_LAUNCH((dense_act_kernel<8>), dim3(J, ceil_div(N, 8)), dim3(256), 0, st, x, W, b, post_scale, post_shift, slope, leaky,
             y, N, K, J, hwc_C);
  return ADM_CHECK_LAUNCH();
}

}  // namespace adm
