// k_temb.hip — timestep embedding path (SURVEY.md §8(a) U1 and the per-resnet time_emb_proj of U3).
//   time_embedding: sinusoid(t) [cos|sin after flip_sin_to_cos] -> Linear(128->512) -> SiLU -> Linear(512->512)
//   temb_proj     : for ALL resnets at once, out[b][r] = bias[r] + W[r][:] . silu(emb[b][:]) with the 32
//                   time_emb_proj matrices stacked row-wise (9984 x 512 for the 256x256 model); the conv1
//                   epilogue then adds its slice as a per-(n,cout) bias.
// Tiny, launch-latency class work: two launches per UNet forward instead of ~70 eager ops.
// The timestep comes either from a device array (training / direct forward) or from the scheduler coefficient
// table indexed by the device-side step counter (hipGraph replay).
#include "adm_kernels.h"

namespace adm {

__device__ __forceinline__ float silu_t(float v) { return v / (1.0f + __expf(-v)); }

// grid (B, TE_SPLIT): every workgroup recomputes the sinusoid and linear_1 (+ SiLU) of its sample (65 k MACs) and produces a
// 1 / TE_SPLIT slice of linear_2's rows, one WAVE per row: lanes walk the row (coalesced) and a shuffle tree adds them up — a
// thread per row read 2 KB-strided words and made the B = 1 step wait ~50 us on one workgroup. emb_act = silu(emb) is stored
// beside emb: temb_proj_kernel used to recompute it for each of its 9984 rows (163 M SiLUs per forward at B = 32).
constexpr int TE_SPLIT = 8;
__global__ void __launch_bounds__(256) time_embedding_kernel(const float* __restrict__ t_dev, int t_stride,
                                                             const adm_sched_coef* __restrict__ table,
                                                             const int* __restrict__ step_dev,
                                                             const float* __restrict__ freqs, int half_dim, int flip,
                                                             const float* __restrict__ w1, const float* __restrict__ b1,
                                                             const float* __restrict__ w2, const float* __restrict__ b2,
                                                             int dim_in, int dim_emb, float* __restrict__ emb,
                                                             float* __restrict__ emb_act,
                                                             float* __restrict__ save_sinus,
                                                             float* __restrict__ save_z) {
  ADM_DYN_SMEM(float, smem);
  float* sinus = smem;          // dim_in
  float* hid = smem + dim_in;   // dim_emb
  const int b = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const float t = t_dev ? t_dev[b * t_stride] : table[*step_dev].timestep;
  for (int i = tid; i < half_dim; i += blockDim.x) {
    const float arg = t * freqs[i];  // fp32 product as in diffusers; sin/cos evaluated in fp64 then rounded
    const float s = (float)sin((double)arg), c = (float)cos((double)arg);
    if (flip) { sinus[i] = c; sinus[half_dim + i] = s; }
    else { sinus[i] = s; sinus[half_dim + i] = c; }
  }
  __syncthreads();
  if (save_sinus && part == 0)  // training: inputs of linear_1 kept for its weight gradient
    for (int i = tid; i < dim_in; i += blockDim.x) save_sinus[(long)b * dim_in + i] = sinus[i];
  // Both GEMVs: EIGHT rows per wave at a time — lane l works on row l / 8 and the k values congruent to l % 8 (mod 8) in float4
  // units, so a wave's load is 8 rows x 128 contiguous bytes, several independent row groups are in flight, and the reduction is
  // 3 shuffle steps.  (One row per wave and iteration — 2 loads, a 6-step shuffle tree, an LDS store, 128 times in a row per wave —
  // made this launch 186 us at ANY batch: a serial chain of L2 round trips)
  const int rl = lane >> 3, kq = lane & 7;
  auto gemv8 = [&](const float* __restrict__ w, const float* __restrict__ x, int K, int j0) __attribute__((always_inline)) -> float {
    const float* wr = w + (long)(j0 + rl) * K;
    float acc = 0.f;
    for (int k = 4 * kq; k < K; k += 32) {
      const float4 wv = *reinterpret_cast<const float4*>(wr + k);
      acc = fmaf(wv.x, x[k], acc); acc = fmaf(wv.y, x[k + 1], acc); acc = fmaf(wv.z, x[k + 2], acc); acc = fmaf(wv.w, x[k + 3], acc);
    }
    acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64);
    return acc;
  };
  const bool vec = (dim_in % 32 == 0) && (dim_emb % 32 == 0) && ((reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2)) & 15) == 0;
  if (vec && dim_in == 128 && dim_emb % 128 == 0) {
    // linear_1 of the shipped models (128 -> 512; every workgroup of a sample recomputes it: 65 k MACs): the lane's 16 sinusoid words
    // live in registers and FOUR row groups (32 rows per wave) are in flight at once — 16 independent float4 loads per lane instead
    // of a chain of 16 dependent iterations (LDS reads of x and the LDS store of hid kept the compiler from overlapping them).
    // Same products, same order per row as gemv8: bit-identical.
    float4 xr[4];
    ADM_UNROLL
    for (int q = 0; q < 4; ++q) xr[q] = *reinterpret_cast<const float4*>(sinus + 4 * kq + 32 * q);
    for (int j0 = 32 * wave; j0 < dim_emb; j0 += 128) {
      float4 wv[4][4];
      ADM_UNROLL
      for (int g = 0; g < 4; ++g)
        ADM_UNROLL
        for (int q = 0; q < 4; ++q) wv[g][q] = *reinterpret_cast<const float4*>(w1 + (long)(j0 + 8 * g + rl) * 128 + 4 * kq + 32 * q);
      ADM_UNROLL
      for (int g = 0; g < 4; ++g) {
        float acc = 0.f;
        ADM_UNROLL
        for (int q = 0; q < 4; ++q) {
          acc = fmaf(wv[g][q].x, xr[q].x, acc); acc = fmaf(wv[g][q].y, xr[q].y, acc);
          acc = fmaf(wv[g][q].z, xr[q].z, acc); acc = fmaf(wv[g][q].w, xr[q].w, acc);
        }
        acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64);
        const int j = j0 + 8 * g + rl;
        if (kq == 0) {
          acc += b1[j];
          hid[j] = silu_t(acc);
          if (save_z && part == 0) save_z[(long)b * dim_emb + j] = acc;  // pre-activation of linear_1 (training)
        }
      }
    }
  } else if (vec) {
    for (int j0 = 8 * wave; j0 < dim_emb; j0 += 32) {          // linear_1 (every workgroup of a sample recomputes it: 65 k MACs)
      const int j = j0 + rl;
      float acc = 0.f;
      if (j < dim_emb) acc = gemv8(w1, sinus, dim_in, j0);
      if (kq == 0 && j < dim_emb) {
        acc += b1[j];
        hid[j] = silu_t(acc);
        if (save_z && part == 0) save_z[(long)b * dim_emb + j] = acc;  // pre-activation of linear_1 (training)
      }
    }
  } else {
    for (int j = wave; j < dim_emb; j += 4) {          // generic shapes: one wave per row
      const float* wr = w1 + (long)j * dim_in;
      float acc = 0.f;
      for (int k = lane; k < dim_in; k += 64) acc = fmaf(wr[k], sinus[k], acc);
      ADM_UNROLL
      for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
      acc += b1[j];
      if (lane == 0) {
        hid[j] = silu_t(acc);
        if (save_z && part == 0) save_z[(long)b * dim_emb + j] = acc;
      }
    }
  }
  __syncthreads();
  const int rows = (dim_emb + TE_SPLIT - 1) / TE_SPLIT;
  const int j_end = (part + 1) * rows < dim_emb ? (part + 1) * rows : dim_emb;
  if (vec && dim_emb == 512 && rows % 8 == 0) {
    // linear_2 of the shipped models (512 -> 512): a row group's 16 float4 loads per lane issued together (the generic loop below
    // leaves it to the compiler, which keeps them behind the LDS reads of hid); same products in the same order: bit-identical
    for (int j0 = part * rows + 8 * wave; j0 < j_end; j0 += 32) {
      const int j = j0 + rl;
      float4 wv[16];
      ADM_UNROLL
      for (int q = 0; q < 16; ++q) wv[q] = *reinterpret_cast<const float4*>(w2 + (long)(j < j_end ? j : j0) * 512 + 4 * kq + 32 * q);
      float acc = 0.f;
      ADM_UNROLL
      for (int q = 0; q < 16; ++q) {
        const float4 xv = *reinterpret_cast<const float4*>(hid + 4 * kq + 32 * q);
        acc = fmaf(wv[q].x, xv.x, acc); acc = fmaf(wv[q].y, xv.y, acc); acc = fmaf(wv[q].z, xv.z, acc); acc = fmaf(wv[q].w, xv.w, acc);
      }
      acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64);
      if (kq == 0 && j < j_end) {
        acc += b2[j];
        emb[(long)b * dim_emb + j] = acc;
        if (emb_act) emb_act[(long)b * dim_emb + j] = silu_t(acc);
      }
    }
    return;
  }
  if (vec && rows % 8 == 0) {
    for (int j0 = part * rows + 8 * wave; j0 < j_end; j0 += 32) {   // linear_2: this workgroup's slice of the rows
      const int j = j0 + rl;
      float acc = 0.f;
      if (j < j_end) acc = gemv8(w2, hid, dim_emb, j0);
      if (kq == 0 && j < j_end) {
        acc += b2[j];
        emb[(long)b * dim_emb + j] = acc;
        if (emb_act) emb_act[(long)b * dim_emb + j] = silu_t(acc);
      }
    }
    return;
  }
  for (int j = part * rows + wave; j < j_end; j += 4) {
    const float* wr = w2 + (long)j * dim_emb;
    float acc = 0.f;
    for (int k = lane; k < dim_emb; k += 64) acc = fmaf(wr[k], hid[k], acc);
    ADM_UNROLL
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    acc += b2[j];
    if (lane == 0) {
      emb[(long)b * dim_emb + j] = acc;
      if (emb_act) emb_act[(long)b * dim_emb + j] = silu_t(acc);
    }
  }
}

// one wave per output row r; lanes split K (B < 8: at B = 1 the 2496 single-row workgroups finish in 19 us, the staged kernel
// below in 29)
template <bool ACTIVATED>
__global__ void __launch_bounds__(256) temb_proj_row_kernel(const float* __restrict__ emb, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ out, int B,
                                                            int K, int R) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* wr = w + (long)r * K;
  float acc[8];
  ADM_UNROLL
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float wv = wr[k];
    ADM_UNROLL
    for (int i = 0; i < 8; ++i)
      if (i < B) {
        const float e = emb[(long)i * K + k];
        acc[i] = fmaf(wv, ACTIVATED ? e : silu_t(e), acc[i]);
      }
  }
  ADM_UNROLL
  for (int i = 0; i < 8; ++i) {
    float v = acc[i];
    ADM_UNROLL
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if (lane == 0 && i < B) out[(long)i * R + r] = v + bias[r];
  }
}

// grid (R / 16, B / 8): a workgroup stages (the SiLU of) eight samples' embeddings in LDS once and produces 16 output rows for
// them, one wave per row at a time: lanes walk the row (coalesced, every weight read once per eight samples), the eight dot
// products take their embedding words from LDS. ACTIVATED: emb already holds silu(emb) (emb_act above). The per-row version
// fetched 8 embedding words per weight word straight from L2 — 1.4 M wave-level loads for 20 MB of weights (B = 16: 67 -> 38 us).
template <bool ACTIVATED>
__global__ void __launch_bounds__(256) temb_proj_kernel(const float* __restrict__ emb, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out, int B,
                                                        int K, int R) {
  ADM_DYN_SMEM(float, se);     // [8][K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.y * 8;
  for (int i = tid; i < 8 * K; i += 256) {
    const int b = i / K, k = i - b * K;
    float e = 0.f;
    if (b0 + b < B) { e = emb[(long)(b0 + b) * K + k]; if (!ACTIVATED) e = silu_t(e); }
    se[i] = e;
  }
  __syncthreads();
  if (K == 512) {
    // the shipped models: the wave's four rows' 32 weight words per lane requested together (the loop below takes the rows one
    // after the other: four L2 round trips in a row); same products in the same order, bit-identical
    float wv[4][8];
    ADM_UNROLL
    for (int j = 0; j < 4; ++j) {
      const int r = blockIdx.x * 16 + wave * 4 + j;
      ADM_UNROLL
      for (int q = 0; q < 8; ++q) wv[j][q] = w[(long)(r < R ? r : R - 1) * 512 + lane + 64 * q];
    }
    ADM_UNROLL
    for (int j = 0; j < 4; ++j) {
      const int r = blockIdx.x * 16 + wave * 4 + j;
      float acc[8];
      ADM_UNROLL
      for (int i = 0; i < 8; ++i) acc[i] = 0.f;
      ADM_UNROLL
      for (int q = 0; q < 8; ++q)
        ADM_UNROLL
        for (int i = 0; i < 8; ++i) acc[i] = fmaf(wv[j][q], se[i * 512 + lane + 64 * q], acc[i]);
      ADM_UNROLL
      for (int i = 0; i < 8; ++i) {
        float v = acc[i];
        ADM_UNROLL
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if (lane == 0 && b0 + i < B && r < R) out[(long)(b0 + i) * R + r] = v + bias[r];
      }
    }
    return;
  }
  for (int j = 0; j < 4; ++j) {
    const int r = blockIdx.x * 16 + wave * 4 + j;
    if (r >= R) break;
    const float* wr = w + (long)r * K;
    float acc[8];
    ADM_UNROLL
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int k = lane; k < K; k += 64) {
      const float wv = wr[k];
      ADM_UNROLL
      for (int i = 0; i < 8; ++i) acc[i] = fmaf(wv, se[i * K + k], acc[i]);
    }
    ADM_UNROLL
    for (int i = 0; i < 8; ++i) {
      float v = acc[i];
      ADM_UNROLL
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
      if (lane == 0 && b0 + i < B) out[(long)(b0 + i) * R + r] = v + bias[r];
    }
  }
}

int launch_time_embedding(const float* t_dev, int t_stride, const adm_sched_coef* table, const int* step_dev,
                          const float* freqs, int half_dim, int flip, const float* w1, const float* b1, const float* w2,
                          const float* b2, int dim_in, int dim_emb, float* emb, int B, hipStream_t st,
                          float* save_sinus, float* save_z, float* emb_act) {
  ADM_REQUIRE(t_dev != nullptr || (table != nullptr && step_dev != nullptr), "time_embedding: no timestep source");
  const size_t smem = sizeof(float) * (size_t)(dim_in + dim_emb);
  ADM_LAUNCH(time_embedding_kernel, dim3(B, TE_SPLIT), dim3(256), smem, st, t_dev, t_stride, table, step_dev, freqs, half_dim,
             flip, w1, b1, w2, b2, dim_in, dim_emb, emb, emb_act, save_sinus, save_z);
  return ADM_CHECK_LAUNCH();
}

int launch_temb_proj(const float* emb, const float* w, const float* bias, float* out, int B, int K, int R,
                     hipStream_t st, int emb_is_activated) {
  if (B < 8) {
    if (emb_is_activated) ADM_LAUNCH(temb_proj_row_kernel<true>, dim3(ceil_div(R, 4)), dim3(256), 0, st, emb, w, bias, out, B, K, R);
    else ADM_LAUNCH(temb_proj_row_kernel<false>, dim3(ceil_div(R, 4)), dim3(256), 0, st, emb, w, bias, out, B, K, R);
    return ADM_CHECK_LAUNCH();
  }
  ADM_REQUIRE(K <= 1536, "temb_proj: embedding width above 1536");
  const size_t smem = sizeof(float) * 8 * (size_t)K;
  const dim3 grid(ceil_div(R, 16), ceil_div(B, 8));
  if (emb_is_activated) ADM_LAUNCH(temb_proj_kernel<true>, grid, dim3(256), smem, st, emb, w, bias, out, B, K, R);
  else ADM_LAUNCH(temb_proj_kernel<false>, grid, dim3(256), smem, st, emb, w, bias, out, B, K, R);
  return ADM_CHECK_LAUNCH();
}

}  // namespace adm
