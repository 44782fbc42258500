// Developer microbenchmark (not part of the product): what bounds the GroupNorm-backward passes of the bf16 training step?
// Synthetic (N, C, H, W) = (16, 128, 256, 256) fp32 tensors; variants of the apply pass (read x, da [, dx]; write dx) and of the
// statistics pass (read x, da; two sums per channel).  hipcc --offload-arch=gfx950 -O3 -o gnbwd_bench gnbwd_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ float silu_grad(float y) { const float sg = 1.0f / (1.0f + __expf(-y)); return sg * (1.0f + y * (1.0f - sg)); }

struct P { const float* x; const float* da; float* dx; int HW, C, groups; const float* mr; const float* s12; const float* gamma; const float* beta; };

template <bool MATH, bool ACC, int NT, int GEOM>
__global__ void __launch_bounds__(256) apply_k(P p) {
  const int c = blockIdx.y, n = blockIdx.z, cg = p.C / p.groups, g = c / cg;
  const float mean = p.mr[(n * p.groups + g) * 2], rstd = p.mr[(n * p.groups + g) * 2 + 1];
  const float s1 = p.s12[(n * p.groups + g) * 2], s2 = p.s12[(n * p.groups + g) * 2 + 1];
  const float gm = p.gamma[c], bt = p.beta[c];
  const long base = ((long)n * p.C + c) * p.HW;
  const float4* xs4 = reinterpret_cast<const float4*>(p.x + base);
  const float4* ds4 = reinterpret_cast<const float4*>(p.da + base);
  float4* dx4 = reinterpret_cast<float4*>(p.dx + base);
  const int n4 = p.HW >> 2, step = GEOM == 0 ? gridDim.x * 256 : 256;
  const int i0 = GEOM == 0 ? blockIdx.x * 256 + threadIdx.x : blockIdx.x * 1024 + threadIdx.x;
  float4 xv[4], dv[4], ov[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = i0 + u * step;
    if (NT & 1) {
      typedef float f4v __attribute__((ext_vector_type(4)));
      const f4v a = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(xs4 + i)), b = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(ds4 + i));
      xv[u] = make_float4(a.x, a.y, a.z, a.w); dv[u] = make_float4(b.x, b.y, b.z, b.w);
    }
    else { xv[u] = xs4[i]; dv[u] = ds4[i]; }
    ov[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ACC) ov[u] = dx4[i];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float xe[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w}, de[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (MATH) {
        const float xh = (xe[k] - mean) * rstd;
        float gy = de[k] * silu_grad(xh * gm + bt);
        r[k] = rstd * (gy * gm - s1 - xh * s2);
      } else r[k] = xe[k] + de[k];
    }
    float4 o = ov[u];
    o.x += r[0]; o.y += r[1]; o.z += r[2]; o.w += r[3];
    if (NT & 2) {
      typedef float f4v __attribute__((ext_vector_type(4)));
      f4v ov_; ov_.x = o.x; ov_.y = o.y; ov_.z = o.z; ov_.w = o.w;
      __builtin_nontemporal_store(ov_, reinterpret_cast<f4v*>(dx4 + i0 + u * step));
    } else dx4[i0 + u * step] = o;
  }
}

// persistent: grid-stride over (n, c, 4096-float4 chunk) items, loads of the next item issued before the math of the current
template <bool MATH>
__global__ void __launch_bounds__(256) apply_persist_k(P p, int items, int chunks) {
  const int n4c = 1024;   // float4 per item
  auto load = [&](int it, float4* xv, float4* dv, long& off) {
    const int ch = it % chunks, nc = it / chunks;
    off = (long)nc * (p.HW >> 2) + (long)ch * n4c + threadIdx.x;
    const float4* xs4 = reinterpret_cast<const float4*>(p.x);
    const float4* ds4 = reinterpret_cast<const float4*>(p.da);
#pragma unroll
    for (int u = 0; u < 4; ++u) { xv[u] = xs4[off + 256 * u]; dv[u] = ds4[off + 256 * u]; }
  };
  float4 xa[4], da_[4], xb[4], db[4];
  long offa, offb;
  int it = blockIdx.x;
  if (it >= items) return;
  load(it, xa, da_, offa);
  for (; it < items; it += gridDim.x) {
    const int nx = it + gridDim.x;
    if (nx < items) load(nx, xb, db, offb);
    const int nc = it / chunks, c = nc % p.C, n = nc / p.C, cg = p.C / p.groups, g = c / cg;
    const float mean = p.mr[(n * p.groups + g) * 2], rstd = p.mr[(n * p.groups + g) * 2 + 1];
    const float s1 = p.s12[(n * p.groups + g) * 2], s2 = p.s12[(n * p.groups + g) * 2 + 1];
    const float gm = p.gamma[c], bt = p.beta[c];
    float4* dx4 = reinterpret_cast<float4*>(p.dx);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float xe[4] = {xa[u].x, xa[u].y, xa[u].z, xa[u].w}, de[4] = {da_[u].x, da_[u].y, da_[u].z, da_[u].w};
      float r[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (MATH) { const float xh = (xe[k] - mean) * rstd; float gy = de[k] * silu_grad(xh * gm + bt); r[k] = rstd * (gy * gm - s1 - xh * s2); }
        else r[k] = xe[k] + de[k];
      }
      dx4[offa + 256 * u] = make_float4(r[0], r[1], r[2], r[3]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { xa[u] = xb[u]; da_[u] = db[u]; }
    offa = offb;
  }
}

// statistics: GEOM 0 = one workgroup per (group, n) walking its channels; 1 = one per (channel, n); 2 = (channel, n, 4 parts)
template <int GEOM, bool F64, bool NT = false>
__global__ void __launch_bounds__(256) stats_k(P p, double* out) {
  const int cg = p.C / p.groups;
  const int n = blockIdx.y;
  const int c_lo = GEOM == 0 ? blockIdx.x * cg : blockIdx.x, c_hi = GEOM == 0 ? c_lo + cg : c_lo + 1;
  const int parts = GEOM == 2 ? 4 : 1, part = GEOM == 2 ? blockIdx.z : 0;
  double A = 0, B = 0;
  for (int c = c_lo; c < c_hi; ++c) {
    const int g = c / cg;
    const float mean = p.mr[(n * p.groups + g) * 2], rstd = p.mr[(n * p.groups + g) * 2 + 1];
    const float gm = p.gamma[c], bt = p.beta[c];
    const long base = ((long)n * p.C + c) * p.HW;
    const float4* xs4 = reinterpret_cast<const float4*>(p.x + base);
    const float4* ds4 = reinterpret_cast<const float4*>(p.da + base);
    const int n4 = (p.HW >> 2) / parts, o4 = part * n4;
    double a = 0, b = 0; float af = 0, bf = 0;
    for (int i0 = threadIdx.x; i0 < n4; i0 += 1024) {
      float4 xv[4], dv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (NT) {
          typedef float f4v __attribute__((ext_vector_type(4)));
          const f4v a = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(xs4 + o4 + i0 + 256 * u)), b = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(ds4 + o4 + i0 + 256 * u));
          xv[u] = make_float4(a.x, a.y, a.z, a.w); dv[u] = make_float4(b.x, b.y, b.z, b.w);
        } else { xv[u] = xs4[o4 + i0 + 256 * u]; dv[u] = ds4[o4 + i0 + 256 * u]; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float xe[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w}, de[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xh = (xe[k] - mean) * rstd;
          const float gy = de[k] * silu_grad(xh * gm + bt);
          if (F64) { a += (double)gy; b += (double)gy * xh; } else { af += gy; bf = fmaf(gy, xh, bf); }
        }
      }
      if (!F64) { a += af; b += bf; af = 0; bf = 0; }
    }
    A += a * gm; B += b * gm;
  }
  for (int m = 32; m >= 1; m >>= 1) { A += __shfl_xor(A, m); B += __shfl_xor(B, m); }
  if ((threadIdx.x & 63) == 0) { out[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (threadIdx.x >> 6) * 2] = A; out[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (threadIdx.x >> 6) * 2 + 1] = B; }
}

int main() {
  const int N = 16, C = 128, HW = 256 * 256, groups = 32;
  const size_t T = (size_t)N * C * HW;
  float *x, *da, *dx, *mr, *s12, *gamma, *beta; double* out;
  CK(hipMalloc(&x, T * 4)); CK(hipMalloc(&da, T * 4)); CK(hipMalloc(&dx, T * 4));
  CK(hipMalloc(&mr, N * groups * 8)); CK(hipMalloc(&s12, N * groups * 8)); CK(hipMalloc(&gamma, C * 4)); CK(hipMalloc(&beta, C * 4));
  CK(hipMalloc(&out, (size_t)N * C * 4 * 8 * 8));
  std::vector<float> h(T);
  for (size_t i = 0; i < T; ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 32768.f - 1.f;
  CK(hipMemcpy(x, h.data(), T * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(da, h.data(), T * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dx, 0, T * 4));
  std::vector<float> one(N * groups * 2, 0.5f), gb(C, 1.0f);
  CK(hipMemcpy(mr, one.data(), one.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(s12, one.data(), one.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(gamma, gb.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(beta, gb.data(), C * 4, hipMemcpyHostToDevice));
  P p{x, da, dx, HW, C, groups, mr, s12, gamma, beta};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, double bytes, auto launch) {
    for (int i = 0; i < 2; ++i) launch();
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
      (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-44s %8.1f us  %6.2f TB/s\n", name, best * 1e3, bytes / best / 1e9);
  };
  const dim3 ga(16, C, N);
  const double b3 = 3.0 * T * 4, b4 = 4.0 * T * 4, b2 = 2.0 * T * 4;
  timeit("apply current (strided, math)", b3, [&] { hipLaunchKernelGGL((apply_k<true, false, 0, 0>), ga, dim3(256), 0, 0, p); });
  timeit("apply current + accumulate", b4, [&] { hipLaunchKernelGGL((apply_k<true, true, 0, 0>), ga, dim3(256), 0, 0, p); });
  timeit("apply strided, no math", b3, [&] { hipLaunchKernelGGL((apply_k<false, false, 0, 0>), ga, dim3(256), 0, 0, p); });
  timeit("apply contiguous chunk, math", b3, [&] { hipLaunchKernelGGL((apply_k<true, false, 0, 1>), ga, dim3(256), 0, 0, p); });
  timeit("apply contiguous chunk, no math", b3, [&] { hipLaunchKernelGGL((apply_k<false, false, 0, 1>), ga, dim3(256), 0, 0, p); });
  timeit("apply contiguous, math, nontemporal", b3, [&] { hipLaunchKernelGGL((apply_k<true, false, 3, 1>), ga, dim3(256), 0, 0, p); });
  timeit("apply contiguous, math, nt LOADS only", b3, [&] { hipLaunchKernelGGL((apply_k<true, false, 1, 1>), ga, dim3(256), 0, 0, p); });
  timeit("apply contiguous, math, nt STORES only", b3, [&] { hipLaunchKernelGGL((apply_k<true, false, 2, 1>), ga, dim3(256), 0, 0, p); });
  timeit("apply strided, math, nontemporal", b3, [&] { hipLaunchKernelGGL((apply_k<true, false, 3, 0>), ga, dim3(256), 0, 0, p); });
  timeit("apply strided + accumulate, nontemporal", b4, [&] { hipLaunchKernelGGL((apply_k<true, true, 3, 0>), ga, dim3(256), 0, 0, p); });
  timeit("apply contiguous + accumulate", b4, [&] { hipLaunchKernelGGL((apply_k<true, true, 0, 1>), ga, dim3(256), 0, 0, p); });
  for (int wgs : {512, 1024, 2048, 4096}) {
    char nm[64]; snprintf(nm, 64, "apply persistent %d WGs, math", wgs);
    timeit(nm, b3, [&] { hipLaunchKernelGGL((apply_persist_k<true>), dim3(wgs), dim3(256), 0, 0, p, N * C * 16, 16); });
  }
  timeit("apply persistent 2048 WGs, no math", b3, [&] { hipLaunchKernelGGL((apply_persist_k<false>), dim3(2048), dim3(256), 0, 0, p, N * C * 16, 16); });
  timeit("stats current (group, n), fp64 sums", b2, [&] { hipLaunchKernelGGL((stats_k<0, true>), dim3(groups, N), dim3(256), 0, 0, p, out); });
  timeit("stats (channel, n), fp64 sums", b2, [&] { hipLaunchKernelGGL((stats_k<1, true>), dim3(C, N), dim3(256), 0, 0, p, out); });
  timeit("stats (channel, n, 4 parts), fp64 sums", b2, [&] { hipLaunchKernelGGL((stats_k<2, true>), dim3(C, N, 4), dim3(256), 0, 0, p, out); });
  timeit("stats (channel, n), fp32 inner sums", b2, [&] { hipLaunchKernelGGL((stats_k<1, false>), dim3(C, N), dim3(256), 0, 0, p, out); });
  timeit("stats (channel, n, 4 parts), fp32 inner", b2, [&] { hipLaunchKernelGGL((stats_k<2, false>), dim3(C, N, 4), dim3(256), 0, 0, p, out); });
  timeit("stats current (group, n), fp64, nontemporal", b2, [&] { hipLaunchKernelGGL((stats_k<0, true, true>), dim3(groups, N), dim3(256), 0, 0, p, out); });
  timeit("stats (channel, n), fp64, nontemporal", b2, [&] { hipLaunchKernelGGL((stats_k<1, true, true>), dim3(C, N), dim3(256), 0, 0, p, out); });
  return 0;
}
