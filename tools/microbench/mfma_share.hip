// Developer microbenchmark (not part of the product): does the fp32 matrix pipe of a SIMD run at the same pace when TWO waves feed it
// (conv_wino5_kernel: eight MFMA waves per workgroup) as when ONE does (conv_wino4_kernel: four)?  One workgroup per CU, every wave holds
// 32 independent 16x16x4 accumulators (128 registers) and issues 128 MFMAs per loop trip from register operands — nothing else.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_share mfma_share.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// MODE 3: 16x16x4 f32 MFMAs with 6 INDEPENDENT v_fma_f32 per 4 MFMAs; 4: bf16 32x32x16 MFMAs alone; 5: bf16 MFMAs with 6 independent v_fma_f32
// per MFMA (the same VALU : matrix-pipe-cycle ratio as mode 3) — does the VALU overlap a bf16 MFMA where it does not overlap an f32 one?
template <int THREADS, int MODE>
__global__ void __launch_bounds__(THREADS) k2(float* out, int iters, float av, float bv) {
  float v[6] = {av, bv, av + 1.f, bv + 2.f, av - 1.f, bv - 3.f};
  float s = 0.f;
  if (MODE == 3) {
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
          if ((i & 3) == 3) {
#pragma unroll
            for (int j = 0; j < 6; ++j) v[j] = v[j] * av + bv;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
  } else {
    f32x16 acc[8];
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)av; b[j] = (__bf16)bv; }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 16; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
          if (MODE == 5) {
#pragma unroll
            for (int j = 0; j < 6; ++j) v[j] = v[j] * av + bv;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
  }
  out[blockIdx.x * THREADS + threadIdx.x] = s + v[0] + v[1] + v[2] + v[3] + v[4] + v[5];
}

template <int THREADS, int MODE>
int run2(const char* name, int grid, int iters, float* out) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k2<THREADS, MODE>), dim3(grid), dim3(THREADS), 0, 0, out, iters, 1.0f, 0.0f);
  CK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k2<THREADS, MODE>), dim3(grid), dim3(THREADS), 0, 0, out, iters, 1.0f, 0.0f);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  // matrix-pipe cycles per trip and wave: 128 x 32 (f32 16x16x4) = 4096; 128 x 32 (bf16 32x32x16: 8 passes... 32 cycles) = 4096
  const double cyc = (double)iters * 128.0 * 32.0;
  printf("%-52s grid %4d x %3d threads: %8.3f ms  = %5.1f %% of the matrix pipe's own pace at 2.4 GHz\n", name, grid, THREADS, best,
         100.0 * cyc * (THREADS / 256) / 2.4e9 / (best * 1e-3));
  return 0;
}

template <int THREADS, int MODE>   // MODE 0: 16x16x4, 32 accumulators; 1: 32x32x2, 8 accumulators; 2: 16x16x4 with 6 VALU per 4 MFMAs
__global__ void __launch_bounds__(THREADS) k(float* out, int iters, float av, float bv) {
  if (MODE == 1) {
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 8; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
  } else {
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v0 = av, v1 = bv, v2 = av + 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
          if (MODE == 2 && (i & 3) == 3) {
            v0 = v0 * v1 + v2; v1 = v1 * v2 + v0; v2 = v2 * v0 + v1; v0 = v0 * v1 + v2; v1 = v1 * v2 + v0; v2 = v2 * v0 + v1;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    }
    float s = v0 + v1 + v2;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
  }
}

template <int THREADS, int MODE>
int run(const char* name, int grid, int iters, float* out) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<THREADS, MODE>), dim3(grid), dim3(THREADS), 0, 0, out, iters, 1.0f, 0.0f);
  CK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<THREADS, MODE>), dim3(grid), dim3(THREADS), 0, 0, out, iters, 1.0f, 0.0f);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  const double mf = (double)grid * (THREADS / 64) * iters * 128.0;      // MFMAs (16x16x4: 2048 FLOP; 32x32x2: 64 per trip x 4096 FLOP)
  const double flop = MODE == 1 ? (double)grid * (THREADS / 64) * iters * 64.0 * 4096.0 : mf * 2048.0;
  printf("%-44s grid %4d x %3d threads: %8.3f ms  %7.1f TF/s\n", name, grid, THREADS, best, flop / best / 1e9);
  return 0;
}

int main() {
  float* out; CK(hipMalloc(&out, sizeof(float) * 1024 * 512));
  const int it = 20000;
  run<256, 0>("16x16x4, 1 wave/SIMD (256-thread WG, 1/CU)", 256, it, out);
  run<512, 0>("16x16x4, 2 waves/SIMD (512-thread WG, 1/CU)", 256, it / 2, out);
  run<256, 0>("16x16x4, 2 waves/SIMD (2 x 256-thread WG/CU)", 512, it / 2, out);
  run<256, 1>("32x32x2, 1 wave/SIMD", 256, it, out);
  run<512, 1>("32x32x2, 2 waves/SIMD (512-thread WG)", 256, it / 2, out);
  run<256, 2>("16x16x4 + 6 VALU per 4 MFMAs, 1 wave/SIMD", 256, it, out);
  run<512, 2>("16x16x4 + 6 VALU per 4 MFMAs, 2 waves/SIMD", 256, it / 2, out);
  run<1024, 0>("16x16x4, 4 waves/SIMD (1024-thread WG)", 256, it / 4, out);
  run2<256, 3>("f32 16x16x4 + 6 independent VALU / 4 MFMAs, 1 wave/SIMD", 256, it, out);
  run2<512, 3>("f32 16x16x4 + 6 independent VALU / 4 MFMAs, 2 waves/SIMD", 256, it / 2, out);
  run2<256, 4>("bf16 32x32x16 alone, 1 wave/SIMD", 256, it, out);
  run2<256, 5>("bf16 32x32x16 + 6 independent VALU / MFMA, 1 wave/SIMD", 256, it, out);
  run2<512, 5>("bf16 32x32x16 + 6 independent VALU / MFMA, 2 waves/SIMD", 256, it / 2, out);
  return 0;
}
