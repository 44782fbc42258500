// k_train.hip — optimizer-side kernels of the training step (SURVEY.md §8(a) rows T4, T6, T7, T9;
// reference: scripts/train_unet.py:258-267). All HBM-bound multi-tensor work is done over FLAT fp32 buffers
// (parameters, gradients, Adam moments and the EMA shadow are each one contiguous allocation of 113.67 M floats),
// so each is a single launch instead of ~450 per-tensor launches:
//   mse_loss        : loss = mean((pred-target)^2) (F.mse_loss, :258) and d loss / d pred = 2 (pred-target) / n
//   grad_sqnorm     : sum of squares of the flat gradient (clip_grad_norm_, :261-262), fp64 accumulation
//   adamw_ema_step  : torch.optim.AdamW single-tensor math (:263, hyper-params :166-172) fused with the global-norm
//                     clip factor (read from the device) and diffusers' EMAModel.step shadow update (:265-266):
//                     reads p,g,m,v,(ema) and writes p,m,v,(ema) once: 28-36 B per parameter.
#include "adm_kernels.h"

namespace adm {

__device__ __forceinline__ double block_sum_d(double v) {
  __shared__ double red[4];
  ADM_UNROLL
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256) mse_loss_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                       long n, double* __restrict__ acc, float* __restrict__ grad) {
  double s = 0.0;
  const float gscale = 2.0f / (float)n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = pred[i] - target[i];
    s += (double)d * d;
    if (grad) grad[i] = gscale * d;
  }
  s = block_sum_d(s);
  if (threadIdx.x == 0) atomicAdd(acc, s);
}

__global__ void __launch_bounds__(256) sqnorm_kernel(const float* __restrict__ g, long n, double* __restrict__ acc) {
  double s = 0.0;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long i = n4 << 2; i < n; ++i) s += (double)g[i] * g[i];
  s = block_sum_d(s);
  if (threadIdx.x == 0) atomicAdd(acc, s);
}

// out[0] = loss or norm helper: finalises scalars on the device so no host sync is needed inside the step.
__global__ void finalize_scalars_kernel(const double* __restrict__ acc, double inv_n, float max_norm,
                                        float* __restrict__ out, float inv_scale = 1.0f) {
  // mode by inv_n: >0 -> out[0] = acc*inv_n (mean); ==0 -> out[0] = total_norm, out[1] = clip coefficient.
  // inv_scale (fp16 loss scaling): the gradients carry the factor 1/inv_scale; the norm is reported unscaled and the
  // coefficient un-scales and clips in one multiply. A non-finite norm stays non-finite (the caller skips the step).
  if (inv_n > 0.0) {
    out[0] = (float)(acc[0] * inv_n);
  } else if (inv_scale != 1.0f) {
    const float total = (float)sqrt(acc[0]) * inv_scale;
    out[0] = total;
    const float coef = max_norm / (total + 1e-6f);
    out[1] = (coef < 1.0f ? coef : 1.0f) * inv_scale;
  } else {
    const float total = (float)sqrt(acc[0]);
    out[0] = total;
    const float coef = max_norm / (total + 1e-6f);
    out[1] = coef < 1.0f ? coef : 1.0f;   // torch.nn.utils.clip_grad_norm_: clamp(max_norm/(norm+1e-6), max=1)
  }
}

struct AdamArgs {
  float lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2_sqrt, ema_one_minus_decay;
};

__global__ void __launch_bounds__(256) adamw_ema_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        float* __restrict__ ema, long n, AdamArgs a,
                                                        const float* __restrict__ clip_coef) {
  const float gs = clip_coef ? *clip_coef : 1.0f;
  const float step_size = a.lr / a.bias_correction1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float grad = g[i] * gs;
    float param = p[i];
    param = param * (1.0f - a.lr * a.weight_decay);                       // param.mul_(1 - lr*wd)
    float ea = m[i];
    ea = ea + (1.0f - a.beta1) * (grad - ea);                              // exp_avg.lerp_(grad, 1-beta1)
    float es = v[i];
    es = es * a.beta2 + (1.0f - a.beta2) * grad * grad;                    // exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2)
    const float denom = sqrtf(es) / a.bias_correction2_sqrt + a.eps;
    param = param - step_size * (ea / denom);                              // param.addcdiv_(exp_avg, denom, -step_size)
    p[i] = param; m[i] = ea; v[i] = es;
    if (ema) {
      const float s = ema[i];
      ema[i] = s - a.ema_one_minus_decay * (s - param);                    // s_param.sub_((1-decay)*(s_param-param))
    }
  }
}

// Elementwise passes over a flat buffer that the optimizer side needs besides the fused AdamW kernel (gradient
// accumulation micro-steps, the DDP 1/world scaling, EMAModel.step on its own): one float4 grid-stride kernel.
enum { FLAT_ADD = 0, FLAT_SCALE_FROM = 1, FLAT_DIV = 2, FLAT_EMA = 3 };
template <int OP>
__device__ __forceinline__ float flat_apply(float y, float x, float a) {
  if (OP == FLAT_ADD) return y + x;                  // acc.add_(grads)
  if (OP == FLAT_SCALE_FROM) return x * a;           // torch.mul(acc, 1/k, out=grads)
  if (OP == FLAT_DIV) return y / a;                  // grads.div_(world)
  return y - a * (y - x);                            // s_param.sub_((1-decay)*(s_param-param)), a = 1-decay
}
template <int OP>
__global__ void __launch_bounds__(256) flat_op_kernel(float* __restrict__ y, const float* __restrict__ x, long n, float a) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 v = OP == FLAT_SCALE_FROM ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<float4*>(y)[i];
    const float4 u = OP == FLAT_DIV ? v : reinterpret_cast<const float4*>(x)[i];
    v.x = flat_apply<OP>(v.x, u.x, a); v.y = flat_apply<OP>(v.y, u.y, a);
    v.z = flat_apply<OP>(v.z, u.z, a); v.w = flat_apply<OP>(v.w, u.w, a);
    reinterpret_cast<float4*>(y)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long i = n4 << 2; i < n; ++i) y[i] = flat_apply<OP>(y[i], OP == FLAT_DIV ? 0.f : x[i], a);
}

static inline unsigned tgrid(long n) {
  long g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace adm

using namespace adm;

extern "C" {

// loss_out: device float[1]; grad_out: NULL or device (n); scratch: device double[1] (zeroed here).
int adm_mse_loss(const float* pred, const float* target, long n, float* loss_out, float* grad_out, double* scratch,
                 void* stream) {
  ADM_REQUIRE(pred && target && loss_out && scratch && n > 0, "mse_loss: bad argument");
  hipStream_t st = (hipStream_t)stream;
  ADM_TRY(dmemset(scratch, 0, sizeof(double), st));
  ADM_LAUNCH(mse_loss_kernel, dim3(tgrid(n)), dim3(256), 0, st, pred, target, n, scratch, grad_out);
  ADM_LAUNCH(finalize_scalars_kernel, dim3(1), dim3(1), 0, st, (const double*)scratch, 1.0 / (double)n, 0.f, loss_out);
  return ADM_CHECK_LAUNCH();
}

// norm_clip_out: device float[2] = {total L2 norm, clip coefficient min(1, max_norm/(norm+1e-6))}; scratch double[1].
int adm_grad_norm_clip(const float* grads, long n, float max_norm, float* norm_clip_out, double* scratch, void* stream) {
  ADM_REQUIRE(grads && norm_clip_out && scratch && n > 0, "grad_norm_clip: bad argument");
  hipStream_t st = (hipStream_t)stream;
  ADM_TRY(dmemset(scratch, 0, sizeof(double), st));
  ADM_LAUNCH(sqnorm_kernel, dim3(tgrid(n / 4 + 1)), dim3(256), 0, st, grads, n, scratch);
  ADM_LAUNCH(finalize_scalars_kernel, dim3(1), dim3(1), 0, st, (const double*)scratch, 0.0, max_norm, norm_clip_out);
  return ADM_CHECK_LAUNCH();
}

// clip_grad_norm_ on loss-SCALED gradients (GradScaler.unscale_ + clip in one): norm_clip_out = {||g||_2 * inv_scale,
// min(1, max_norm/(that + 1e-6)) * inv_scale}; non-finite gradients give a non-finite norm.
int adm_grad_norm_clip_scaled(const float* grads, long n, float max_norm, float inv_scale, float* norm_clip_out, double* scratch,
                              void* stream) {
  ADM_REQUIRE(grads && norm_clip_out && scratch && n > 0 && inv_scale > 0.f, "grad_norm_clip_scaled: bad argument");
  hipStream_t st = (hipStream_t)stream;
  ADM_TRY(dmemset(scratch, 0, sizeof(double), st));
  ADM_LAUNCH(sqnorm_kernel, dim3(tgrid(n / 4 + 1)), dim3(256), 0, st, grads, n, scratch);
  ADM_LAUNCH(finalize_scalars_kernel, dim3(1), dim3(1), 0, st, (const double*)scratch, 0.0, max_norm, norm_clip_out, inv_scale);
  return ADM_CHECK_LAUNCH();
}

// One AdamW step over flat buffers (+ optional EMA shadow update with `ema_decay`; ema == NULL skips it).
// `step` is the 1-based optimizer step (bias corrections 1-beta^step); clip_coef_dev: NULL or device float scale.
int adm_adamw_ema_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema, long n, float lr,
                       float beta1, float beta2, float eps, float weight_decay, int step, const float* clip_coef_dev,
                       float ema_decay, void* stream) {
  ADM_REQUIRE(params && grads && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adamw_ema_step: bad argument");
  AdamArgs a;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
  a.bias_correction1 = (float)(1.0 - pow((double)beta1, step));
  a.bias_correction2_sqrt = (float)sqrt(1.0 - pow((double)beta2, step));
  a.ema_one_minus_decay = 1.0f - ema_decay;
  ADM_LAUNCH(adamw_ema_kernel, dim3(tgrid(n)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, ema,
             n, a, clip_coef_dev);
  return ADM_CHECK_LAUNCH();
}

// y (op)= x over flat fp32 buffers (16-byte aligned): op 0: y += x; 1: y = x*a; 2: y /= a (x unused); 3: y -= a*(y-x).
int adm_flat_op(float* y, const float* x, long n, int op, float a, void* stream) {
  ADM_REQUIRE(y && n > 0 && op >= 0 && op <= 3 && (x || op == FLAT_DIV), "flat_op: bad argument");
  ADM_REQUIRE(((uintptr_t)y & 15) == 0 && (!x || ((uintptr_t)x & 15) == 0), "flat_op: buffers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const dim3 g(tgrid(n / 4 + 1)), b(256);
  switch (op) {
    case FLAT_ADD: ADM_LAUNCH(flat_op_kernel<FLAT_ADD>, g, b, 0, st, y, x, n, a); break;
    case FLAT_SCALE_FROM: ADM_LAUNCH(flat_op_kernel<FLAT_SCALE_FROM>, g, b, 0, st, y, x, n, a); break;
    case FLAT_DIV: ADM_LAUNCH(flat_op_kernel<FLAT_DIV>, g, b, 0, st, y, x, n, a); break;
    default: ADM_LAUNCH(flat_op_kernel<FLAT_EMA>, g, b, 0, st, y, x, n, a); break;
  }
  return ADM_CHECK_LAUNCH();
}

}  // extern "C"
