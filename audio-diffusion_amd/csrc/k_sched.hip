// k_sched.hip — fused scheduler epilogue, add_noise and u8 dequantisation (HBM-bound elementwise).
// Replaces DDIMScheduler.step / DDPMScheduler.step + mask overwrite + final dequant
// (reference: audiodiffusion/pipeline_audio_diffusion.py:165-185,192-194; SURVEY.md §8(a) S2-S4,P4,P5).
// One float4 per lane per iteration, grid-stride; algorithmic bytes: 12 B/elem (x, eps in; out) +4 with noise.
#include "adm_kernels.h"

namespace adm {

__device__ __forceinline__ float sched_one(float x, float e, float nz, const adm_sched_coef& c) {
  float x0 = (x - c.sqrt_beta * e) / c.sqrt_alpha;
  if (c.clip >= 0.f) x0 = fminf(fmaxf(x0, -c.clip), c.clip);
  float prev = c.k_x0 * x0 + c.k_x * x;
  prev = prev + c.k_eps * e;
  prev = prev + c.k_noise * nz;
  return prev;
}

__device__ __forceinline__ unsigned char quant_u8(float v) {
  float q = fminf(fmaxf(v * 0.5f + 0.5f, 0.f), 1.f) * 255.f;
  return (unsigned char)rintf(q);  // round-half-even == numpy .round() (pipeline:194)
}

__global__ void __launch_bounds__(256) sched_step_kernel(
    const float* __restrict__ x, const float* __restrict__ eps, const float* noise, float* out,
    unsigned char* u8, const adm_sched_coef* __restrict__ table, const int* __restrict__ step_dev,
    int step, const float* __restrict__ mask, long mask_bstride, int mask_start, int mask_end, int W,
    long per_sample, long n4, long noise_step_stride, int u8_step) {
  const int s = step_dev ? *step_dev : step;
  const adm_sched_coef c = table[s];
  const bool use_noise = noise != nullptr && c.k_noise != 0.f;
  noise += (long)s * noise_step_stride;  // per-step slice of a (n_steps,B,C,H,W) noise tensor (0: single step)
  if (u8_step >= 0 && s != u8_step) u8 = nullptr;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 xv = reinterpret_cast<const float4*>(x)[i];
    const float4 ev = reinterpret_cast<const float4*>(eps)[i];
    float4 nv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (use_noise) nv = reinterpret_cast<const float4*>(noise)[i];
    float r[4] = {sched_one(xv.x, ev.x, nv.x, c), sched_one(xv.y, ev.y, nv.y, c), sched_one(xv.z, ev.z, nv.z, c),
                  sched_one(xv.w, ev.w, nv.w, c)};
    if (mask != nullptr) {
      const long e0 = i * 4;
      const long b = e0 / per_sample;
      const long p = e0 - b * per_sample;  // C == 1: p = row*W + col
      const int col0 = (int)(p % W);
      const float* mrow = mask + b * mask_bstride + (long)s * per_sample + p;
      ADM_UNROLL
      for (int k = 0; k < 4; ++k) {
        const int col = col0 + k;
        if (col < mask_start || col >= W - mask_end) r[k] = mrow[k];
      }
    }
    reinterpret_cast<float4*>(out)[i] = make_float4(r[0], r[1], r[2], r[3]);
    if (u8 != nullptr) {
      const unsigned q = (unsigned)quant_u8(r[0]) | ((unsigned)quant_u8(r[1]) << 8) |
                         ((unsigned)quant_u8(r[2]) << 16) | ((unsigned)quant_u8(r[3]) << 24);
      reinterpret_cast<unsigned*>(u8)[i] = q;
    }
  }
}

__global__ void step_advance_kernel(int* step_dev) { *step_dev += 1; }

// DDIM inversion update (pipeline_audio_diffusion.py:238-240).
__global__ void __launch_bounds__(256) encode_step_kernel(float* x, const float* __restrict__ eps,
                                                          const adm_sched_coef* __restrict__ table,
                                                          const int* __restrict__ step_dev, int step, long n4) {
  const adm_sched_coef c = table[step_dev ? *step_dev : step];
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 xv = reinterpret_cast<const float4*>(x)[i];
    const float4 ev = reinterpret_cast<const float4*>(eps)[i];
    float* xs = reinterpret_cast<float*>(&xv);
    const float* es = reinterpret_cast<const float*>(&ev);
    ADM_UNROLL
    for (int k = 0; k < 4; ++k) {
      float s = (xs[k] - c.sqrt_beta * es[k]) * c.sqrt_alpha;
      xs[k] = s * c.k_x0 + c.k_eps * es[k];
    }
    reinterpret_cast<float4*>(x)[i] = xv;
  }
}

__global__ void __launch_bounds__(256) add_noise_kernel(const float* __restrict__ x0, long x0_bstride,
                                                        const float* __restrict__ noise,
                                                        const float* __restrict__ sa, const float* __restrict__ sb,
                                                        int cb, int cn, float* __restrict__ out, int N, long P4) {
  const int b = blockIdx.z, n = blockIdx.y;
  const float a = sa[b * cb + n * cn], s = sb[b * cb + n * cn];
  const float4* xp = reinterpret_cast<const float4*>(x0 + (long)b * x0_bstride);
  const float4* np = reinterpret_cast<const float4*>(noise) + (long)b * P4;
  float4* op = reinterpret_cast<float4*>(out) + ((long)b * N + n) * P4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < P4; i += (long)gridDim.x * blockDim.x) {
    const float4 xv = xp[i], nv = np[i];
    op[i] = make_float4(a * xv.x + s * nv.x, a * xv.y + s * nv.y, a * xv.z + s * nv.z, a * xv.w + s * nv.w);
  }
}

__global__ void __launch_bounds__(256) dequant_kernel(const float* __restrict__ x, unsigned char* __restrict__ out,
                                                      long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    reinterpret_cast<unsigned*>(out)[i] = (unsigned)quant_u8(v.x) | ((unsigned)quant_u8(v.y) << 8) |
                                          ((unsigned)quant_u8(v.z) << 16) | ((unsigned)quant_u8(v.w) << 24);
  }
}

static inline int ew_grid(long n4) {
  long g = (n4 + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));  // cap + grid-stride (guide §6 G11)
}

int launch_sched_step_loop(const float* x, const float* eps, const float* noise, long noise_step_stride, float* out,
                           uint8_t* u8, int u8_step, const adm_sched_coef* table, const int* step_dev, int step,
                           const float* mask, int n_mask_steps, int mask_start, int mask_end, int B, int C, int H,
                           int W, hipStream_t st) {
  const long per_sample = (long)C * H * W, n = per_sample * B;
  ADM_REQUIRE(W % 4 == 0, "sched_step: W must be a multiple of 4");
  ADM_REQUIRE(mask == nullptr || C == 1, "sched_step: mask path requires C == 1 (as in the reference)");
  ADM_LAUNCH(sched_step_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, st, x, eps, noise, out, u8, table, step_dev, step,
             mask, (long)n_mask_steps * per_sample, mask_start, mask_end, W, per_sample, n / 4, noise_step_stride,
             u8_step);
  return ADM_CHECK_LAUNCH();
}

int launch_sched_step(const float* x, const float* eps, const float* noise, float* out, uint8_t* u8,
                      const adm_sched_coef* table, const int* step_dev, int step, const float* mask,
                      int n_mask_steps, int mask_start, int mask_end, int B, int C, int H, int W, hipStream_t st) {
  return launch_sched_step_loop(x, eps, noise, 0, out, u8, -1, table, step_dev, step, mask, n_mask_steps, mask_start,
                                mask_end, B, C, H, W, st);
}

int launch_step_advance(int* step_dev, hipStream_t st) {
  ADM_LAUNCH(step_advance_kernel, dim3(1), dim3(1), 0, st, step_dev);
  return ADM_CHECK_LAUNCH();
}

int launch_encode_step(float* x, const float* eps, const adm_sched_coef* table, const int* step_dev, int step, long n,
                       hipStream_t st) {
  ADM_REQUIRE(n % 4 == 0, "encode_step: size must be a multiple of 4");
  ADM_LAUNCH(encode_step_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, st, x, eps, table, step_dev, step, n / 4);
  return ADM_CHECK_LAUNCH();
}

int launch_add_noise(const float* x0, long x0_bstride, const float* noise, const float* sa, const float* sb, int cb,
                     int cn, float* out, int B, int N, long P, hipStream_t st) {
  ADM_REQUIRE(P % 4 == 0 && x0_bstride % 4 == 0, "add_noise: sizes must be multiples of 4");
  long g = (P / 4 + 255) / 256;
  if (g > 256) g = 256;
  ADM_LAUNCH(add_noise_kernel, dim3((unsigned)g, N, B), dim3(256), 0, st, x0, x0_bstride, noise, sa, sb, cb, cn, out, N,
             P / 4);
  return ADM_CHECK_LAUNCH();
}

// ---- spherical interpolation grid (pipeline_audio_diffusion.py:244-258, batched over alphas) -------------------------------------
// theta = acos(<x0, x1> / |x0| / |x1|);  out[a] = sin((1 - alpha_a) theta) x0 / sin(theta) + sin(alpha_a theta) x1 / sin(theta)
// The three reductions accumulate in fp64 (torch: fp32); the blend repeats torch's float32 operation order exactly: python
// double scalars are cast to float32, then (s0 * x0) / sin(theta) + (s1 * x1) / sin(theta) with IEEE multiplies / divisions.
__global__ void __launch_bounds__(256) slerp_reduce_kernel(const float* __restrict__ x0, const float* __restrict__ x1, long n,
                                                           double* __restrict__ acc3) {
  __shared__ double red[3][4];
  double d = 0.0, a = 0.0, b = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const double u = x0[i], v = x1[i];
    d += u * v; a += u * u; b += v * v;
  }
  for (int m = 32; m >= 1; m >>= 1) { d += __shfl_xor(d, m, 64); a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = d; red[1][wave] = a; red[2][wave] = b; }
  __syncthreads();
  if (threadIdx.x < 3) atomicAdd(acc3 + threadIdx.x, red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

__global__ void __launch_bounds__(256) slerp_blend_kernel(const float* __restrict__ x0, const float* __restrict__ x1, long n,
                                                          const double* __restrict__ acc3, const double* __restrict__ alphas,
                                                          int n_alpha, float* __restrict__ out) {
  // torch: dot / norm / norm on float32 0-d tensors, then math.acos of the float32 quotient
  const float dotf = (float)acc3[0], n0 = (float)sqrt(acc3[1]), n1 = (float)sqrt(acc3[2]);
  const double theta = acos((double)((dotf / n0) / n1));
  const float st = (float)sin(theta);
  const int a = blockIdx.y;
  const double al = alphas[a];   // a double, as the reference's Python float: s0, s1 are rounded to float32 once
  const float s0 = (float)sin((1.0 - al) * theta), s1 = (float)sin(al * theta);
  float* o = out + (long)a * n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    o[i] = __fadd_rn(__fdiv_rn(__fmul_rn(s0, x0[i]), st), __fdiv_rn(__fmul_rn(s1, x1[i]), st));
}

int launch_slerp_grid(const float* x0, const float* x1, long n, const double* alphas_dev, int n_alpha, float* out,
                      double* scratch3, hipStream_t st) {
  ADM_TRY(dmemset(scratch3, 0, 3 * sizeof(double), st));
  ADM_LAUNCH(slerp_reduce_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x0, x1, n, scratch3);
  ADM_LAUNCH(slerp_blend_kernel, dim3(ew_grid(n), n_alpha), dim3(256), 0, st, x0, x1, n, (const double*)scratch3, alphas_dev,
             n_alpha, out);
  return ADM_CHECK_LAUNCH();
}

int launch_dequant(const float* x, uint8_t* out, long n, hipStream_t st) {
  ADM_REQUIRE(n % 4 == 0, "dequant: size must be a multiple of 4");
  ADM_LAUNCH(dequant_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, st, x, out, n / 4);
  return ADM_CHECK_LAUNCH();
}

}  // namespace adm
