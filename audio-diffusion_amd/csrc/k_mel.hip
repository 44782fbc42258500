// k_mel.hip — the audio <-> 8-bit log-mel-spectrogram codec as HIP kernels.
//
// Replaces, for whole batches of slices/images at once (the reference does them one by one on a host core,
// pipeline_audio_diffusion.py:201):
//   Mel.audio_slice_to_image (audiodiffusion/mel.py:135-151): librosa melspectrogram (Hann-2048 STFT, |.|^2,
//     Slaney filterbank) -> power_to_db(ref=max, top_db) -> u8 quantise;
//   Mel.image_to_audio (mel.py:153-168): u8 -> dB -> power -> mel_to_stft (NNLS) -> Griffin-Lim (n_iter, momentum .99).
// Numerics follow librosa/numpy dtype-by-dtype: FFTs run in fp64 (the Hann window is fp64, so numpy's rfft runs
// in double) and are rounded to complex64 exactly where librosa stores into a complex64 array; the mel
// projection, log10 and quantisation run in the input precision (fp32 for fp32 audio, fp64 for fp64 audio).
// NNLS: librosa's L-BFGS-B starts from clip(pinv(A) S, 0) and — because the objective is scaled by 1/size —
// its projected gradient is already below pgtol=1e-5 there for any dB image, so it returns that point after
// zero iterations (oracle/mel.py records nit == 0). We compute exactly that point (fp64 GEMM) and verify the
// projected-gradient criterion on the device with the sparse filterbank; `status` reports its max.
// FFT: radix-2 in LDS (2048 complex fp64 = 32 KiB), one frame per workgroup, twiddles from an L2-resident table.
// All spectral arrays are laid out [b][frame][bin] so a workgroup's bins are contiguous (coalesced).
// Algorithmic bytes: forward 4 B/sample in + 1 B/pixel out; inverse 1 B/pixel in + 4 B/sample out.
#include <cstdlib>
#include <vector>

#include "adm_kernels.h"

struct adm_mel {
  adm_mel_config cfg;
  int n_bins = 0, n_mels = 0, log2n = 0, nnz = 0, nnz_t = 0, nnls_cols = 0;
  double *window = nullptr, *twiddle = nullptr, *fb_w64 = nullptr, *pinv = nullptr, *fbt_w64 = nullptr;
  float *fb_w32 = nullptr, *wss = nullptr;
  int *fb_start = nullptr, *fb_count = nullptr, *fb_off = nullptr, *fbt_off = nullptr, *fbt_idx = nullptr;
  std::vector<void*> owned;
  // scratch (grown on demand)
  size_t cap_fwd = 0, cap_inv = 0, cap_max = 0;
  void* melspec = nullptr;
  unsigned long long* spec_max = nullptr;   // per-spectrogram maximum (fast forward path)
  double *angles = nullptr, *mag = nullptr, *ytmp = nullptr, *Smel = nullptr, *Xpow = nullptr, *diff = nullptr;
  float *reb0 = nullptr, *reb1 = nullptr, *y = nullptr;
  float* pgmax = nullptr;      // [0] max |projected gradient| of the returned point, [1] of the start point
  float* pgblk = nullptr;      // per (image, NNLS column block): max |projected gradient| of the start point
  size_t cap_blk = 0;
  // NNLS solver (adm_mel_set_nnls_solver): step 1 / lipschitz = 1 / lambda_max(A A^T); 0 = start point only
  double lipschitz = 0.0;
  int nnls_max_iter = 0;
  int* iters_dev = nullptr;    // max iterations any column ran in the last inverse
  int last_iters = 0;
  float last_pg_start = 0.f;
};

namespace adm {

// ---------------------------------------------------------------------------------------------------------
// In-LDS radix-2 FFT over n = 2^log2n complex fp64 points held in x (bit-reversed order on entry).
// tw[q] = exp(-2*pi*i*q/n), q < n/2. inverse: conjugate twiddles (no scaling).
__device__ __forceinline__ void fft_lds(double2* x, int n, int log2n, const double2* __restrict__ tw, bool inverse) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int s = 1; s <= log2n; ++s) {
    const int half = 1 << (s - 1);
    const int tstride = n >> s;
    for (int j = tid; j < (n >> 1); j += nt) {
      const int k = j >> (s - 1), i = j & (half - 1);
      const int i0 = (k << s) + i, i1 = i0 + half;
      double2 w = tw[i * tstride];
      if (inverse) w.y = -w.y;
      const double2 a = x[i0], b = x[i1];
      const double tr = w.x * b.x - w.y * b.y, ti = w.x * b.y + w.y * b.x;
      x[i0] = make_double2(a.x + tr, a.y + ti);
      x[i1] = make_double2(a.x - tr, a.y - ti);
    }
    __syncthreads();
  }
}

__device__ __forceinline__ int bitrev(int v, int log2n) {
  unsigned r = 0, u = (unsigned)v;
  for (int i = 0; i < log2n; ++i) { r = (r << 1) | (u & 1u); u >>= 1; }
  return (int)r;
}

// ---------------------------------------------------------------------------------------------------------
// Forward: one workgroup per (frame, slice): centred Hann STFT frame -> |D|^2 -> sparse mel projection.
template <typename T>
__global__ void __launch_bounds__(256) mel_stft_power_kernel(const T* __restrict__ audio, long slice_stride,
                                                             int n_samples, int n_fft, int log2n, int hop,
                                                             const double* __restrict__ window,
                                                             const double2* __restrict__ tw,
                                                             const int* __restrict__ fb_start,
                                                             const int* __restrict__ fb_count,
                                                             const int* __restrict__ fb_off,
                                                             const float* __restrict__ fb_w32,
                                                             const double* __restrict__ fb_w64, int n_mels,
                                                             int n_frames, T* __restrict__ melspec) {
  ADM_DYN_SMEM(double2, x);                             // n_fft complex
  T* pw = reinterpret_cast<T*>(x + n_fft);               // n_fft/2+1 powers
  const int frame = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const T* y = audio + (long)b * slice_stride;
  for (int n = tid; n < n_fft; n += blockDim.x) {
    const long src = (long)frame * hop + n - n_fft / 2;  // center=True, pad_mode="constant"
    const double v = (src >= 0 && src < n_samples) ? (double)y[src] : 0.0;
    x[bitrev(n, log2n)] = make_double2(window[n] * v, 0.0);
  }
  __syncthreads();
  fft_lds(x, n_fft, log2n, tw, false);
  const int n_bins = n_fft / 2 + 1;
  for (int k = tid; k < n_bins; k += blockDim.x) {
    if (sizeof(T) == 4) {
      const float re = (float)x[k].x, im = (float)x[k].y;  // stored into a complex64 array by librosa.stft
      const float a = hypotf(re, im);                      // np.abs(complex64) -> float32
      pw[k] = (T)(a * a);
    } else {
      const double a = hypot(x[k].x, x[k].y);
      pw[k] = (T)(a * a);
    }
  }
  __syncthreads();
  for (int m = tid; m < n_mels; m += blockDim.x) {
    const int s = fb_start[m], c = fb_count[m], o = fb_off[m];
    double acc = 0.0;
    for (int i = 0; i < c; ++i)
      acc += (sizeof(T) == 4 ? (double)fb_w32[o + i] : fb_w64[o + i]) * (double)pw[s + i];
    melspec[((long)b * n_mels + m) * n_frames + frame] = (T)acc;
  }
}

// =====================================================================================================================
// n_fft = 2048 fast path (the configuration of every published audio-diffusion model): one WAVE per frame.
//   * real-input trick: the 2048 windowed samples are packed as 1024 complex points z[n] = x[2n] + i x[2n+1]; one 1024-point
//     complex FFT and an untangling pass give bins 0..1024 — half the arithmetic of the complex FFT the generic kernel runs;
//   * 1024 = 16 x 16 x 4: every lane keeps 16 points in registers and runs two radix-16 passes (each two radix-4 levels,
//     fully unrolled) and one radix-4 pass, with THREE trips through LDS instead of eleven;
//   * twiddles are built in registers from one table entry per lane and pass (w, w^2, w^4, w^8 by squaring, the rest by
//     one product each: <= 4 roundings deep) instead of 15 table gathers per pass;
//   * a frame belongs to one wave, so nothing in the transform needs a workgroup barrier (LDS executes a wave's operations
//     in order); LDS rows are pitched 65 complex (1040 B) so that all three exchange patterns are conflict-free;
//   * a workgroup (4 waves) handles 8 consecutive frames and stores their mel columns as 32-byte row segments.
// Numerics: fp64 throughout, rounded to float32 exactly where librosa stores into complex64 (as the generic kernel).
#if defined(ADM_EMU)
#define ADM_WAVE_SYNC() __syncthreads()      // fibers switch only at collectives: make the cross-lane LDS hand-over visible
#else
#define ADM_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif
constexpr int MF_PITCH = 65;                 // complex elements per LDS row (16 rows)

__device__ __forceinline__ double2 c_mul(double2 a, double2 b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 c_add(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 c_sub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 c_mul_mi(double2 a) { return make_double2(a.y, -a.x); }       // a * (-i)

// forward radix-4 butterfly: X[k] = sum_n a[n] (-i)^(n k)
__device__ __forceinline__ void radix4(double2& a0, double2& a1, double2& a2, double2& a3) {
  const double2 t0 = c_add(a0, a2), t1 = c_sub(a0, a2), t2 = c_add(a1, a3), t3 = c_mul_mi(c_sub(a1, a3));
  a0 = c_add(t0, t2); a1 = c_add(t1, t3); a2 = c_sub(t0, t2); a3 = c_sub(t1, t3);
}

// forward 16-point DFT in registers: v[n] -> v[k] (natural order in and out). n = 4 p + r, k = ka + 4 kb.
__device__ __forceinline__ void dft16(double2 (&v)[16]) {
  const double C1 = 0.92387953251128673848, S1 = 0.38268343236508978178, H = 0.70710678118654752440;
  // W16^m = exp(-2 pi i m / 16)
  const double2 W1 = make_double2(C1, -S1), W2 = make_double2(H, -H), W3 = make_double2(S1, -C1), W6 = make_double2(-H, -H),
                W9 = make_double2(-C1, S1);
  double2 a[4][4];                       // a[ka][r] after the first level
  ADM_UNROLL
  for (int r = 0; r < 4; ++r) {
    double2 x0 = v[r], x1 = v[4 + r], x2 = v[8 + r], x3 = v[12 + r];
    radix4(x0, x1, x2, x3);
    a[0][r] = x0; a[1][r] = x1; a[2][r] = x2; a[3][r] = x3;
  }
  a[1][1] = c_mul(a[1][1], W1); a[1][2] = c_mul(a[1][2], W2); a[1][3] = c_mul(a[1][3], W3);
  a[2][1] = c_mul(a[2][1], W2); a[2][2] = c_mul_mi(a[2][2]);   a[2][3] = c_mul(a[2][3], W6);
  a[3][1] = c_mul(a[3][1], W3); a[3][2] = c_mul(a[3][2], W6); a[3][3] = c_mul(a[3][3], W9);
  ADM_UNROLL
  for (int ka = 0; ka < 4; ++ka) {
    radix4(a[ka][0], a[ka][1], a[ka][2], a[ka][3]);
    v[ka] = a[ka][0]; v[ka + 4] = a[ka][1]; v[ka + 8] = a[ka][2]; v[ka + 12] = a[ka][3];
  }
}

// v[j] *= w^j, j = 1..15, with the powers of w built by squaring (depth <= 4 products)
__device__ __forceinline__ void twiddle16(double2 (&v)[16], double2 w) {
  const double2 w2 = c_mul(w, w), w4 = c_mul(w2, w2), w8 = c_mul(w4, w4);
  const double2 w3 = c_mul(w2, w), w5 = c_mul(w4, w), w6 = c_mul(w4, w2), w7 = c_mul(w4, w3);
  v[1] = c_mul(v[1], w); v[2] = c_mul(v[2], w2); v[3] = c_mul(v[3], w3); v[4] = c_mul(v[4], w4);
  v[5] = c_mul(v[5], w5); v[6] = c_mul(v[6], w6); v[7] = c_mul(v[7], w7); v[8] = c_mul(v[8], w8);
  v[9] = c_mul(v[9], c_mul(w8, w)); v[10] = c_mul(v[10], c_mul(w8, w2)); v[11] = c_mul(v[11], c_mul(w8, w3));
  v[12] = c_mul(v[12], c_mul(w8, w4)); v[13] = c_mul(v[13], c_mul(w8, w5)); v[14] = c_mul(v[14], c_mul(w8, w6));
  v[15] = c_mul(v[15], c_mul(w8, w7));
}

// 1024-point forward complex FFT of one wave's frame. In: lane l holds z[64 n1 + l] in v[n1]. Out: lane t holds
// Z[t + 64 c] in v[c]. buf: this wave's 16 x MF_PITCH complex LDS rows; tw[q] = exp(-2 pi i q / 2048), q < 1024.
__device__ __forceinline__ void fft1024_wave(double2 (&v)[16], double2* buf, const double2* __restrict__ tw, int lane) {
  dft16(v);                                              // over n1 -> A[k1]
  twiddle16(v, tw[2 * lane]);                            // A[k1] *= W1024^(l k1)
  ADM_UNROLL
  for (int k1 = 0; k1 < 16; ++k1) buf[k1 * MF_PITCH + lane] = v[k1];
  ADM_WAVE_SYNC();
  const int k1 = lane & 15, m2 = lane >> 4;
  ADM_UNROLL
  for (int m1 = 0; m1 < 16; ++m1) v[m1] = buf[k1 * MF_PITCH + 4 * m1 + m2];
  ADM_WAVE_SYNC();
  dft16(v);                                              // over m1 -> B[j1]
  twiddle16(v, tw[32 * m2]);                             // B[j1] *= W64^(m2 j1)
  ADM_UNROLL
  for (int j1 = 0; j1 < 16; ++j1) buf[k1 * MF_PITCH + 4 * j1 + m2] = v[j1];
  ADM_WAVE_SYNC();
  ADM_UNROLL
  for (int i = 0; i < 4; ++i) {                          // radix 4 over m2: (k1, j1 = (lane >> 4) + 4 i) -> j2 = 0..3
    const double2* src = buf + k1 * MF_PITCH + 4 * ((lane >> 4) + 4 * i);
    double2 c0 = src[0], c1 = src[1], c2 = src[2], c3 = src[3];
    radix4(c0, c1, c2, c3);
    v[i] = c0; v[i + 4] = c1; v[i + 8] = c2; v[i + 12] = c3;        // Z[lane + 64 (i + 4 j2)]
  }
  ADM_WAVE_SYNC();
}

// MINW = waves per SIMD: 1 = 4 waves per workgroup (8 frames), ~390 registers; 2 = 8 waves per workgroup (16 frames, one shared
// filterbank copy: 157 KiB of LDS), <= 256 registers — the loads of a frame are then issued in four batches instead of sixteen
// at once, which is what keeps the allocation under 256 without spills. ADM_MEL_OCC selects (measured on the MI355X).
// Round 6: PERSISTENT workgroups with the next frame's samples prefetched. The kernel was latency-bound on its own global loads (ablations,
// profiles/r06_mel.md: without the FFT it lost 16 % of its time, without the mel projection 14 % — the other 70 % were two waves per SIMD
// waiting for a frame's 2048 samples, four dependent batches per frame, and 4096 workgroups each staging the filterbank again): a workgroup
// now walks the groups of 2 NW frames with a stride of gridDim.x, stages the filterbank once, and every wave requests the samples of its
// NEXT frame (32 registers) before it transforms the current one. Same arithmetic in the same order: bit-identical spectrograms.
template <typename T, int MINW>
__global__ void __launch_bounds__(256 * MINW, 1) mel_stft2048_kernel(const T* __restrict__ audio, long slice_stride, int n_samples,
                                                          int hop, const double* __restrict__ window,
                                                          const double2* __restrict__ tw,
                                                          const int* __restrict__ fb_start, const int* __restrict__ fb_count,
                                                          const int* __restrict__ fb_off, const float* __restrict__ fb_w32,
                                                          const double* __restrict__ fb_w64, int n_mels, int n_frames,
                                                          T* __restrict__ melspec, unsigned long long* __restrict__ spec_max,
                                                          int groups_per_clip, int n_groups) {
  // spec_max (nullptr ok): per-spectrogram maximum for power_to_db(ref=np.max), as the bit pattern of the non-negative
  // double (order-preserving), so that the dB pass needs no reduction of its own
  ADM_DYN_SMEM(double2, sm);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  constexpr int NW = 4 * MINW, FR = 2 * NW;                        // waves and frames per group
  double2* buf = sm + wave * 16 * MF_PITCH;
  T* stage = reinterpret_cast<T*>(sm + NW * 16 * MF_PITCH);        // [n_mels][FR]
  // the filterbank, staged once per workgroup: taps as (first bin, count, offset) per filter + the weights in T
  int* f_start = reinterpret_cast<int*>(stage + n_mels * FR);
  int* f_count = f_start + n_mels;
  int* f_off = f_count + n_mels;
  T* f_w = reinterpret_cast<T*>(f_off + n_mels + (n_mels & 1));     // keeps 8-byte alignment for T = double
  const int nnz = fb_off[n_mels];
  for (int m = tid; m < n_mels; m += blockDim.x) { f_start[m] = fb_start[m]; f_count[m] = fb_count[m]; f_off[m] = fb_off[m]; }
  for (int i = tid; i < nnz; i += blockDim.x) f_w[i] = sizeof(T) == 4 ? (T)fb_w32[i] : (T)fb_w64[i];
  __syncthreads();
  // the 2048 samples of frame `slot` of group g, two per lane and n1 (zero outside the clip: center=True, pad_mode="constant";
  // frames beyond n_frames: computed on zeros, never stored)
  auto load_raw = [&](int g, int slot, T (&raw)[32]) {
    const int b = g / groups_per_clip, frame = (g - b * groups_per_clip) * FR + slot;
    const T* y = audio + (long)b * slice_stride;
    const bool live = frame < n_frames && g < n_groups;
    ADM_UNROLL
    for (int n1 = 0; n1 < 16; ++n1) {
      const int n = 64 * n1 + lane;
      const long src = (long)frame * hop + 2 * n - 1024;
      raw[2 * n1] = (live && src >= 0 && src < n_samples) ? y[src] : (T)0;
      raw[2 * n1 + 1] = (live && src + 1 >= 0 && src + 1 < n_samples) ? y[src + 1] : (T)0;
    }
  };
  // (the samples are consumed by the windowing below; the SAME registers then take the wave's next frame, requested before the transform)
  auto process = [&](int slot, T (&raw)[32], int next_g, int next_slot) {
    double2 v[16];
    ADM_UNROLL
    for (int n1 = 0; n1 < 16; ++n1) {
      const int n = 64 * n1 + lane;
      const double2 w = *reinterpret_cast<const double2*>(window + 2 * n);
      v[n1] = make_double2(w.x * (double)raw[2 * n1], w.y * (double)raw[2 * n1 + 1]);
      if (MINW == 2 && (n1 & 3) == 3) ADM_SCHED_FENCE();            // 256-register build: four batches of window loads, not sixteen
    }
    ADM_SCHED_FENCE();
    load_raw(next_g, next_slot, raw);
    ADM_SCHED_FENCE();
    fft1024_wave(v, buf, tw, lane);
    // untangle: X[k] = E + (-i) W2048^k O, E = (Z[k] + conj Z[1024-k]) / 2, O = (Z[k] - conj Z[1024-k]) / 2
    ADM_UNROLL
    for (int c = 0; c < 16; ++c) buf[lane + 64 * c] = v[c];
    ADM_WAVE_SYNC();
    // every partner Z[1024 - k] is read and consumed on the spot; only the 16 power values stay in registers until all
    // lanes have finished reading, then they overwrite the (dead) exchange rows as bins 0..1024
    T pv[16];
    double2 wk = tw[lane];                                          // W2048^(lane + 64 c), advanced by W32 per c
    const double2 w32 = tw[64];
    ADM_UNROLL
    for (int c = 0; c < 16; ++c) {
      const double2 zq = buf[(1024 - (lane + 64 * c)) & 1023];
      const double2 zk = v[c], zc = make_double2(zq.x, -zq.y);
      const double2 e = make_double2(0.5 * (zk.x + zc.x), 0.5 * (zk.y + zc.y));
      const double2 o = make_double2(0.5 * (zk.x - zc.x), 0.5 * (zk.y - zc.y));
      const double2 x = c_add(e, c_mul_mi(c_mul(wk, o)));
      if (sizeof(T) == 4) {
        // complex64 store, then np.abs -> float32 hypot. glibc's hypotf IS sqrt in double of the exactly representable
        // re^2 + im^2 rounded once to float: the same three operations here, without ocml's range-scaling hypotf
        const double re = (double)(float)x.x, im = (double)(float)x.y;
        const float a = (float)sqrt(re * re + im * im);
        pv[c] = (T)(a * a);
      } else {
        const double a = hypot(x.x, x.y);
        pv[c] = (T)(a * a);
      }
      wk = c_mul(wk, w32);
    }
    const double xn = v[0].x - v[0].y;                               // lane 0: X[1024] = Re Z[0] - Im Z[0] (real)
    ADM_WAVE_SYNC();
    T* pw = reinterpret_cast<T*>(buf);
    ADM_UNROLL
    for (int c = 0; c < 16; ++c) pw[lane + 64 * c] = pv[c];
    if (lane == 0) {
      if (sizeof(T) == 4) { const float a = fabsf((float)xn); pw[1024] = (T)(a * a); }
      else { const double a = fabs(xn); pw[1024] = (T)(a * a); }
    }
    ADM_WAVE_SYNC();
    for (int m = lane; m < n_mels; m += 64) {
      const int s = f_start[m], cnt = f_count[m], o = f_off[m];
      double acc = 0.0;                                             // taps in order, as the generic kernel sums them
      int i = 0;
      for (; i + 4 <= cnt; i += 4) {                                // four taps' operands in flight
        const double w0 = (double)f_w[o + i], w1 = (double)f_w[o + i + 1], w2 = (double)f_w[o + i + 2], w3 = (double)f_w[o + i + 3];
        const double p0 = (double)pw[s + i], p1 = (double)pw[s + i + 1], p2 = (double)pw[s + i + 2], p3 = (double)pw[s + i + 3];
        acc += w0 * p0; acc += w1 * p1; acc += w2 * p2; acc += w3 * p3;
      }
      for (; i < cnt; ++i) acc += (double)f_w[o + i] * (double)pw[s + i];
      stage[m * FR + slot] = (T)acc;
    }
    ADM_WAVE_SYNC();
  };
  T raw[32];
  int g = blockIdx.x;
  load_raw(g, wave, raw);
#pragma unroll 1
  for (; g < n_groups; g += gridDim.x) {
    const int b = g / groups_per_clip, f0 = (g - b * groups_per_clip) * FR;
    process(wave, raw, g, wave + NW);                                // ... requests this group's second frame
    process(wave + NW, raw, g + (int)gridDim.x, wave);               // ... requests the next group's first frame (past the end: zeros, unused)
    __syncthreads();
    // [n_mels][FR] -> melspec[b][m][f0 .. f0 + FR): one row segment per thread pass
    double mx = 0.0;
    for (int e = tid; e < n_mels * FR; e += blockDim.x) {
      const int m = e / FR, sl = e % FR;
      if (f0 + sl < n_frames) {
        melspec[((long)b * n_mels + m) * n_frames + f0 + sl] = stage[e];
        mx = fmax(mx, (double)stage[e]);
      }
    }
    if (spec_max != nullptr) {
      for (int m = 32; m >= 1; m >>= 1) mx = fmax(mx, __shfl_xor(mx, m, 64));
      if (lane == 0) atomicMax(spec_max + b, (unsigned long long)__double_as_longlong(mx));
    }
    __syncthreads();                                                 // the stage rows are free for the next group
  }
}

// power_to_db + u8 quantise with the per-spectrogram maximum already known (mel_stft2048_kernel's spec_max): plain
// elementwise grid, same arithmetic as mel_db_u8_kernel.
template <typename T>
__global__ void __launch_bounds__(256) mel_db_u8_grid_kernel(const T* __restrict__ melspec, int n, float top_db,
                                                              const unsigned long long* __restrict__ spec_max,
                                                              unsigned char* __restrict__ img) {
  const int b = blockIdx.y;
  const double mx = __longlong_as_double((long long)spec_max[b]);
  const T* s = melspec + (long)b * n;
  const T amin = (T)1e-10;
  const T ref_db = (T)10.0 * (T)log10((double)(((T)mx > amin) ? (T)mx : amin));
  const T peak = (T)10.0 * (T)log10((double)(((T)mx > amin) ? (T)mx : amin)) - ref_db;  // log_spec.max()
  const T floor_db = peak - (T)top_db;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const T v = s[i] > amin ? s[i] : amin;
    T db = (T)10.0 * (T)log10((double)v);
    db = db - ref_db;
    db = db > floor_db ? db : floor_db;
    T q = (db + (T)top_db) * (T)255 / (T)top_db;
    q = q < (T)0 ? (T)0 : (q > (T)255 ? (T)255 : q);
    img[(long)b * n + i] = (unsigned char)(q + (T)0.5);  // astype(uint8) truncation
  }
}

// power_to_db(ref=np.max, amin=1e-10, top_db) + u8 quantise (mel.py:148-150): one workgroup per spectrogram.
template <typename T>
__global__ void __launch_bounds__(256) mel_db_u8_kernel(const T* __restrict__ melspec, int n, float top_db,
                                                         unsigned char* __restrict__ img) {
  __shared__ double red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const T* s = melspec + (long)b * n;
  double mx = 0.0;
  for (int i = tid; i < n; i += blockDim.x) mx = fmax(mx, (double)s[i]);
  for (int m = 32; m >= 1; m >>= 1) mx = fmax(mx, __shfl_xor(mx, m, 64));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = red[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) mx = fmax(mx, red[i]);
  const T amin = (T)1e-10;
  const T ref_db = (T)10.0 * (T)log10((double)(((T)mx > amin) ? (T)mx : amin));
  const T peak = (T)10.0 * (T)log10((double)(((T)mx > amin) ? (T)mx : amin)) - ref_db;  // log_spec.max()
  const T floor_db = peak - (T)top_db;
  for (int i = tid; i < n; i += blockDim.x) {
    const T v = s[i] > amin ? s[i] : amin;
    T db = (T)10.0 * (T)log10((double)v);
    db = db - ref_db;
    db = db > floor_db ? db : floor_db;
    T q = (db + (T)top_db) * (T)255 / (T)top_db;
    q = q < (T)0 ? (T)0 : (q > (T)255 ? (T)255 : q);
    img[(long)b * n + i] = (unsigned char)(q + (T)0.5);  // astype(uint8) truncation
  }
}

// ---------------------------------------------------------------------------------------------------------
// Inverse step 1: u8 -> dB -> power (mel.py:162-164), fp64:  S = 10^(0.1*(u8*top_db/255 - top_db)).
__global__ void __launch_bounds__(256) mel_u8_to_power_kernel(const unsigned char* __restrict__ img, long n,
                                                              double top_db, double* __restrict__ S) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const double db = (double)img[i] * top_db / 255.0 - top_db;
    S[i] = pow(10.0, 0.1 * db);
  }
}

// Inverse step 2: X[b][t][f] = max(0, sum_m pinv[f][m] * S[b][m][t])  (fp64 GEMM, 64x64 tiles, K chunks of 16);
// also writes mag = sqrt(X) (mel_to_stft power=2).
__global__ void __launch_bounds__(256) mel_pinv_gemm_kernel(const double* __restrict__ pinv, const double* __restrict__ S,
                                                            int n_bins, int n_mels, int n_frames,
                                                            double* __restrict__ Xpow, double* __restrict__ mag) {
  __shared__ double As[16][64 + 1];  // [k][f]
  __shared__ double Bs[16][64 + 1];  // [k][t]
  const int b = blockIdx.z, f0 = blockIdx.x * 64, t0 = blockIdx.y * 64, tid = threadIdx.x;
  const int tf = (tid & 15) * 4, tt = (tid >> 4) * 4;
  double acc[4][4] = {{0}};
  const double* Sb = S + (long)b * n_mels * n_frames;
  for (int k0 = 0; k0 < n_mels; k0 += 16) {
    for (int e = tid; e < 16 * 64; e += 256) {
      const int kk = e & 15, ff = e >> 4;
      As[kk][ff] = (f0 + ff < n_bins && k0 + kk < n_mels) ? pinv[(long)(f0 + ff) * n_mels + k0 + kk] : 0.0;
      const int tt2 = e & 63, kk2 = e >> 6;
      Bs[kk2][tt2] = (t0 + tt2 < n_frames && k0 + kk2 < n_mels) ? Sb[(long)(k0 + kk2) * n_frames + t0 + tt2] : 0.0;
    }
    __syncthreads();
    for (int kk = 0; kk < 16; ++kk) {
      double a[4], c[4];
      for (int i = 0; i < 4; ++i) { a[i] = As[kk][tf + i]; c[i] = Bs[kk][tt + i]; }
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], c[j], acc[i][j]);
    }
    __syncthreads();
  }
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 4; ++i) {
      const int f = f0 + tf + i, t = t0 + tt + j;
      if (f < n_bins && t < n_frames) {
        const double v = acc[i][j] > 0.0 ? acc[i][j] : 0.0;
        const long o = ((long)b * n_frames + t) * n_bins + f;
        Xpow[o] = v;
        mag[o] = sqrt(v);
      }
    }
}

// NNLS check a: diff[b][m][t] = sum_k A[m][k] X[b][t][k] - S[b][m][t]
__global__ void __launch_bounds__(256) mel_nnls_diff_kernel(const double* __restrict__ X, const double* __restrict__ S,
                                                            const int* __restrict__ fb_start,
                                                            const int* __restrict__ fb_count,
                                                            const int* __restrict__ fb_off,
                                                            const double* __restrict__ fb_w64, int n_bins, int n_mels,
                                                            int n_frames, double* __restrict__ diff,
                                                            const float* __restrict__ known_blk = nullptr, int nnls_cols = 1,
                                                            float pgtol = 0.f) {
  // known_blk != nullptr (after the solver): columns of blocks the solver left alone keep the residual they have
  const int b = blockIdx.y;
  const long n = (long)n_mels * n_frames;
  const int n_blk = (n_frames + nnls_cols - 1) / nnls_cols;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const int m = (int)(e / n_frames), t = (int)(e - (long)m * n_frames);
    if (known_blk != nullptr && !(known_blk[(long)b * n_blk + t / nnls_cols] > pgtol)) continue;
    const int s = fb_start[m], c = fb_count[m], o = fb_off[m];
    const double* xr = X + ((long)b * n_frames + t) * n_bins;
    double acc = 0.0;
    for (int i = 0; i < c; ++i) acc += fb_w64[o + i] * xr[s + i];
    diff[(long)b * n + e] = acc - S[(long)b * n + e];
  }
}
// NNLS check b: projected gradient of 0.5*||AX-S||^2/size at X (bounds [0,inf)): max-abs over everything into pgmax and,
// when pgblk != nullptr, per (image, column block) — the unit librosa hands to L-BFGS-B. blockIdx.x = block * parts + part:
// a workgroup stays inside ONE column block, so both maxima cost one atomic per wave.
__global__ void __launch_bounds__(256) mel_nnls_pg_kernel(const double* __restrict__ X, const double* __restrict__ diff,
                                                          const int* __restrict__ fbt_off,
                                                          const int* __restrict__ fbt_idx,
                                                          const double* __restrict__ fbt_w64, int n_bins, int n_mels,
                                                          int n_frames, int nnls_cols, int parts, float* __restrict__ pgmax,
                                                          float* __restrict__ pgblk, const float* __restrict__ known_blk,
                                                          float pgtol) {
  // known_blk != nullptr (the pass AFTER the solver): a block whose start point already satisfied the rule was not touched,
  // its maximum is the one recorded before — no second walk over it
  const int b = blockIdx.y, blk = blockIdx.x / parts, part = blockIdx.x % parts;
  const int n_blk = (n_frames + nnls_cols - 1) / nnls_cols;
  if (known_blk != nullptr) {
    const float k = known_blk[(long)b * n_blk + blk];
    if (!(k > pgtol)) {
      if (threadIdx.x == 0 && part == 0) atomicMax(reinterpret_cast<unsigned*>(pgmax), __float_as_uint(k));
      return;
    }
  }
  const int blk0 = blk * nnls_cols;
  const int cols = (blk0 + nnls_cols <= n_frames) ? nnls_cols : n_frames - blk0;
  const double inv_size = 1.0 / ((double)n_mels * cols);
  const long n = (long)n_bins * cols;
  float local = 0.f;
  for (long e = (long)part * blockDim.x + threadIdx.x; e < n; e += (long)parts * blockDim.x) {
    const int t = blk0 + (int)(e / n_bins), f = (int)(e % n_bins);
    double g = 0.0;
    for (int i = fbt_off[f]; i < fbt_off[f + 1]; ++i)
      g += fbt_w64[i] * diff[((long)b * n_mels + fbt_idx[i]) * n_frames + t];
    g *= inv_size;
    const double x = X[((long)b * n_frames + t) * n_bins + f];
    const double pg = x > 0.0 ? g : (g < 0.0 ? g : 0.0);
    local = fmaxf(local, (float)fabs(pg));
  }
  for (int m = 32; m >= 1; m >>= 1) local = fmaxf(local, __shfl_xor(local, m, 64));
  if ((threadIdx.x & 63) == 0) {
    atomicMax(reinterpret_cast<unsigned*>(pgmax), __float_as_uint(local));
    if (pgblk != nullptr) atomicMax(reinterpret_cast<unsigned*>(pgblk + (long)b * n_blk + blk), __float_as_uint(local));
  }
}

// NNLS solver for the column blocks whose start point does NOT satisfy L-BFGS-B's stopping rule (max |projected
// gradient| > pgtol; librosa.util.nnls -> scipy.optimize.fmin_l_bfgs_b, audiodiffusion/mel.py:165). The problem
//   min_{X >= 0} 0.5 ||A X - S||^2 / size        (A: n_mels x n_bins sparse triangles)
// separates over the columns of X, so one workgroup solves one column (frame) entirely in LDS: accelerated projected
// gradient (FISTA, step 1 / lambda_max(A A^T)) with gradient-based adaptive restart, fp64; it stops when the column's
// own projected gradient is below tol (a tenth of pgtol: every column, hence every block, then satisfies scipy's rule with
// margin and the objective is below what L-BFGS-B's early stop reaches) or after max_iter. Blocks that already satisfy
// the rule are left exactly as they are — that is what scipy returns for them (nit = 0).
__global__ void __launch_bounds__(256) mel_nnls_solve_kernel(double* __restrict__ X, double* __restrict__ mag,
                                                             const double* __restrict__ S,
                                                             const int* __restrict__ fb_start,
                                                             const int* __restrict__ fb_count,
                                                             const int* __restrict__ fb_off,
                                                             const double* __restrict__ fb_w64,
                                                             const int* __restrict__ fbt_off,
                                                             const int* __restrict__ fbt_idx,
                                                             const double* __restrict__ fbt_w64, int n_bins, int n_mels,
                                                             int n_frames, int nnls_cols, const float* __restrict__ pgblk,
                                                             float pgtol, double inv_lip, double tol, int max_iter,
                                                             int* __restrict__ iters_out) {
  ADM_DYN_SMEM(double, sm);
  double* x = sm;                    // current iterate
  double* y = sm + n_bins;           // extrapolated point
  double* xn = sm + 2 * n_bins;      // candidate
  double* d = sm + 3 * n_bins;       // residual A y - s (n_mels)
  double* sv = d + n_mels;           // this column of S
  __shared__ double red[4];
  const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int n_blk = (n_frames + nnls_cols - 1) / nnls_cols;
  if (!(pgblk[(long)b * n_blk + t / nnls_cols] > pgtol)) return;      // block-uniform: scipy returns the start point
  const int blk0 = (t / nnls_cols) * nnls_cols;
  const int cols = (blk0 + nnls_cols <= n_frames) ? nnls_cols : n_frames - blk0;
  const double inv_size = 1.0 / ((double)n_mels * cols);
  double* xg = X + ((long)b * n_frames + t) * n_bins;
  for (int k = tid; k < n_bins; k += blockDim.x) { x[k] = xg[k]; y[k] = xg[k]; }
  for (int m = tid; m < n_mels; m += blockDim.x) sv[m] = S[((long)b * n_mels + m) * n_frames + t];
  __syncthreads();
  auto block_sum = [&](double v) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
  };
  auto block_max = [&](double v) {
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  };
  auto residual = [&](const double* p) {           // d = A p - s
    for (int m = tid; m < n_mels; m += blockDim.x) {
      const int s0 = fb_start[m], c = fb_count[m], o = fb_off[m];
      double acc = 0.0;
      for (int i = 0; i < c; ++i) acc += fb_w64[o + i] * p[s0 + i];
      d[m] = acc - sv[m];
    }
    __syncthreads();
  };
  auto grad_at = [&](int k) {                       // (A^T d)[k]
    double g = 0.0;
    for (int i = fbt_off[k]; i < fbt_off[k + 1]; ++i) g += fbt_w64[i] * d[fbt_idx[i]];
    return g;
  };
  double tk = 1.0;
  int it = 0;
  for (; it < max_iter; ++it) {
    if ((it & 15) == 0) {                           // stopping rule on the iterate itself, every 16 steps
      residual(x);
      double pg = 0.0;
      for (int k = tid; k < n_bins; k += blockDim.x) {
        const double g = grad_at(k) * inv_size;
        pg = fmax(pg, fabs(x[k] > 0.0 ? g : (g < 0.0 ? g : 0.0)));
      }
      pg = block_max(pg);
      if (pg <= tol) break;
    }
    residual(y);
    double r = 0.0;
    for (int k = tid; k < n_bins; k += blockDim.x) {
      const double v = y[k] - grad_at(k) * inv_lip;
      const double c = v > 0.0 ? v : 0.0;
      xn[k] = c;
      r += (y[k] - c) * (c - x[k]);
    }
    r = block_sum(r);
    if (r > 0.0 && tk > 1.0) {                      // momentum points uphill: restart from the current iterate
      tk = 1.0;
      for (int k = tid; k < n_bins; k += blockDim.x) y[k] = x[k];
    } else {
      const double tn = 0.5 * (1.0 + sqrt(1.0 + 4.0 * tk * tk));
      const double beta = (tk - 1.0) / tn;
      for (int k = tid; k < n_bins; k += blockDim.x) {
        const double c = xn[k];
        y[k] = c + beta * (c - x[k]);
        x[k] = c;
      }
      tk = tn;
    }
    __syncthreads();
  }
  double* mg = mag + ((long)b * n_frames + t) * n_bins;
  for (int k = tid; k < n_bins; k += blockDim.x) { xg[k] = x[k]; mg[k] = sqrt(x[k]); }
  if (tid == 0) atomicMax(reinterpret_cast<unsigned*>(iters_out), (unsigned)it);
}

// Griffin-Lim init: angles = (cos(2*pi*u) + i sin(2*pi*u)) * mag   (librosa.griffinlim init="random")
__global__ void __launch_bounds__(256) gl_init_kernel(const double* __restrict__ phase, const double* __restrict__ mag,
                                                      int n_bins, int n_frames, double2* __restrict__ angles) {
  const int b = blockIdx.y;
  const long n = (long)n_bins * n_frames;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const int t = (int)(e / n_bins), f = (int)(e - (long)t * n_bins);
    const double ph = 2.0 * 3.141592653589793 * phase[((long)b * n_bins + f) * n_frames + t];  // caller layout (B,bins,frames)
    const double m = mag[(long)b * n + e];
    angles[(long)b * n + e] = make_double2(cos(ph) * m, sin(ph) * m);
  }
}

// iSTFT part 1: per frame irfft (Hermitian, imag of DC/Nyquist ignored) * window -> ytmp[b][frame][n] fp64.
__global__ void __launch_bounds__(256) gl_istft_frames_kernel(const double2* __restrict__ angles, int n_fft, int log2n,
                                                              const double* __restrict__ window,
                                                              const double2* __restrict__ tw, int n_frames,
                                                              double* __restrict__ ytmp) {
  ADM_DYN_SMEM(double2, x);
  const int frame = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int n_bins = n_fft / 2 + 1;
  const double2* src = angles + ((long)b * n_frames + frame) * n_bins;
  for (int k = tid; k < n_fft; k += blockDim.x) {
    double2 v;
    if (k <= n_fft / 2) {
      v = src[k];
      if (k == 0 || k == n_fft / 2) v.y = 0.0;
    } else {
      v = src[n_fft - k];
      v.y = -v.y;
    }
    x[bitrev(k, log2n)] = v;
  }
  __syncthreads();
  fft_lds(x, n_fft, log2n, tw, true);
  const double fct = 1.0 / (double)n_fft;
  double* dst = ytmp + ((long)b * n_frames + frame) * n_fft;
  for (int n = tid; n < n_fft; n += blockDim.x) dst[n] = window[n] * (x[n].x * fct);
}

// iSTFT part 2: overlap-add in frame order with numpy's float32 in-place `+=` rounding, trim n_fft/2, divide by
// the window sum-square where it exceeds float32 tiny.
__global__ void __launch_bounds__(256) gl_overlap_add_kernel(const double* __restrict__ ytmp, int n_fft, int hop,
                                                             int n_frames, const float* __restrict__ wss, int out_len,
                                                             float* __restrict__ y) {
  const int b = blockIdx.y;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < out_len; j += gridDim.x * blockDim.x) {
    const int i = j + n_fft / 2;
    int f_lo = (i - n_fft + hop) / hop;  // ceil((i - n_fft + 1)/hop) for i - n_fft + 1 > 0
    if (i - n_fft + 1 <= 0) f_lo = 0;
    int f_hi = i / hop;
    if (f_hi > n_frames - 1) f_hi = n_frames - 1;
    float acc = 0.f;
    for (int f = f_lo; f <= f_hi; ++f)
      acc = (float)((double)acc + ytmp[((long)b * n_frames + f) * n_fft + (i - f * hop)]);
    const float w = wss[j];
    if (w > 1.17549435e-38f) acc = acc / w;
    y[(long)b * out_len + j] = acc;
  }
}

// STFT of y (float32) -> rebuilt (complex64) and the momentum / projection update of `angles`:
//   angles = rebuilt - (momentum/(1+momentum)) * tprev ; angles /= |angles| + tiny ; angles *= mag
__global__ void __launch_bounds__(256) gl_stft_update_kernel(const float* __restrict__ y, int out_len, int n_fft,
                                                             int log2n, int hop, const double* __restrict__ window,
                                                             const double2* __restrict__ tw, int n_frames,
                                                             float2* __restrict__ rebuilt,
                                                             const float2* __restrict__ tprev, float mom,
                                                             const double* __restrict__ mag,
                                                             double2* __restrict__ angles) {
  ADM_DYN_SMEM(double2, x);
  const int frame = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float* yb = y + (long)b * out_len;
  for (int n = tid; n < n_fft; n += blockDim.x) {
    const long src = (long)frame * hop + n - n_fft / 2;
    const double v = (src >= 0 && src < out_len) ? (double)yb[src] : 0.0;
    x[bitrev(n, log2n)] = make_double2(window[n] * v, 0.0);
  }
  __syncthreads();
  fft_lds(x, n_fft, log2n, tw, false);
  const int n_bins = n_fft / 2 + 1;
  const long base = ((long)b * n_frames + frame) * n_bins;
  for (int k = tid; k < n_bins; k += blockDim.x) {
    const float2 r = make_float2((float)x[k].x, (float)x[k].y);
    rebuilt[base + k] = r;
    double ar = (double)r.x, ai = (double)r.y;
    if (tprev != nullptr) {
      const float2 tp = tprev[base + k];
      ar -= (double)(mom * tp.x);  // python float * complex64 array stays complex64 (fp32 products)
      ai -= (double)(mom * tp.y);
    }
    const double den = hypot(ar, ai) + 2.2250738585072014e-308;
    const double scl = 1.0 / den;  // numpy complex/real division multiplies by the reciprocal
    const double m = mag[base + k];
    angles[base + k] = make_double2(ar * scl * m, ai * scl * m);
  }
}

// ---- n_fft = 2048 fast paths of the two Griffin-Lim FFT kernels (one wave per frame, fft1024_wave) --------------------------------
// irfft: X[0..1024] -> E[k] = (X[k] + conj X[1024-k]) / 2, Od[k] = conj(W2048^k) (X[k] - conj X[1024-k]) / 2, Z = E + i Od is the
// spectrum of z[n] = y[2n] + i y[2n+1]; z = conj(FFT(conj Z)) / 1024.
__global__ void __launch_bounds__(256) gl_istft_frames2048_kernel(const double2* __restrict__ angles,
                                                                  const double* __restrict__ window,
                                                                  const double2* __restrict__ tw, int n_frames,
                                                                  double* __restrict__ ytmp) {
  ADM_DYN_SMEM(double2, sm);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double2* buf = sm + wave * 16 * MF_PITCH;
  const int b = blockIdx.y, frame = blockIdx.x * 4 + wave;
  const bool live = frame < n_frames;
  const double2* X = angles + ((long)b * n_frames + (live ? frame : 0)) * 1025;
  double2 v[16];
  double2 wk = tw[lane];                                          // W2048^(lane + 64 n1)
  const double2 w32 = tw[64];
  ADM_UNROLL
  for (int n1 = 0; n1 < 16; ++n1) {
    const int k = 64 * n1 + lane;
    double2 xk = X[k], xp = X[1024 - k];
    if (k == 0) { xk.y = 0.0; xp.y = 0.0; }                       // numpy's irfft ignores the imaginary parts of DC and Nyquist
    const double2 e = make_double2(0.5 * (xk.x + xp.x), 0.5 * (xk.y - xp.y));
    const double2 d = make_double2(0.5 * (xk.x - xp.x), 0.5 * (xk.y + xp.y));
    const double2 od = c_mul(make_double2(wk.x, -wk.y), d);       // conj(W^k) * d
    const double2 z = make_double2(e.x - od.y, e.y + od.x);       // E + i Od
    v[n1] = make_double2(z.x, -z.y);                              // conj(Z): the forward engine then yields conj(z) * 1024
    wk = c_mul(wk, w32);
  }
  fft1024_wave(v, buf, tw, lane);
  const double fct = 1.0 / 1024.0;
  double* dst = ytmp + ((long)b * n_frames + (live ? frame : 0)) * 2048;
  ADM_UNROLL
  for (int c = 0; c < 16; ++c) {
    const int j = lane + 64 * c;
    const double2 w = *reinterpret_cast<const double2*>(window + 2 * j);
    if (live) *reinterpret_cast<double2*>(dst + 2 * j) = make_double2(w.x * (v[c].x * fct), w.y * (-v[c].y * fct));
  }
}

// STFT of y (float32) -> rebuilt (complex64) + the momentum / projection update of `angles` (see gl_stft_update_kernel).
__global__ void __launch_bounds__(256) gl_stft_update2048_kernel(const float* __restrict__ y, int out_len, int hop,
                                                                 const double* __restrict__ window,
                                                                 const double2* __restrict__ tw, int n_frames,
                                                                 float2* __restrict__ rebuilt, const float2* __restrict__ tprev,
                                                                 float mom, const double* __restrict__ mag,
                                                                 double2* __restrict__ angles) {
  ADM_DYN_SMEM(double2, sm);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double2* buf = sm + wave * 16 * MF_PITCH;
  const int b = blockIdx.y, frame = blockIdx.x * 4 + wave;
  const bool live = frame < n_frames;
  const float* yb = y + (long)b * out_len;
  double2 v[16];
  ADM_UNROLL
  for (int n1 = 0; n1 < 16; ++n1) {
    const int n = 64 * n1 + lane;
    const long src = (long)frame * hop + 2 * n - 1024;
    const double a0 = (live && src >= 0 && src < out_len) ? (double)yb[src] : 0.0;
    const double a1 = (live && src + 1 >= 0 && src + 1 < out_len) ? (double)yb[src + 1] : 0.0;
    const double2 w = *reinterpret_cast<const double2*>(window + 2 * n);
    v[n1] = make_double2(w.x * a0, w.y * a1);
  }
  fft1024_wave(v, buf, tw, lane);
  ADM_UNROLL
  for (int c = 0; c < 16; ++c) buf[lane + 64 * c] = v[c];
  ADM_WAVE_SYNC();
  const long base = ((long)b * n_frames + (live ? frame : 0)) * 1025;
  auto emit = [&](int k, double xr, double xi) {
    const float2 r = make_float2((float)xr, (float)xi);
    rebuilt[base + k] = r;
    double ar = (double)r.x, ai = (double)r.y;
    if (tprev != nullptr) {
      const float2 tp = tprev[base + k];
      ar -= (double)(mom * tp.x);
      ai -= (double)(mom * tp.y);
    }
    // angles /= |angles| + tiny: 1 / (hypot + tiny) as one reciprocal square root (|a| is far from the overflow range here;
    // a = 0 keeps angles = 0, as 0 / tiny does)
    const double s2 = ar * ar + ai * ai;
    const double scl = s2 > 0.0 ? rsqrt(s2) : 0.0;
    const double m = mag[base + k];
    angles[base + k] = make_double2(ar * scl * m, ai * scl * m);
  };
  double2 wk = tw[lane];
  const double2 w32 = tw[64];
  ADM_UNROLL
  for (int c = 0; c < 16; ++c) {
    const double2 zq = buf[(1024 - (lane + 64 * c)) & 1023];
    const double2 zk = v[c], zc = make_double2(zq.x, -zq.y);
    const double2 e = make_double2(0.5 * (zk.x + zc.x), 0.5 * (zk.y + zc.y));
    const double2 o = make_double2(0.5 * (zk.x - zc.x), 0.5 * (zk.y - zc.y));
    const double2 x = c_add(e, c_mul_mi(c_mul(wk, o)));
    if (live) emit(lane + 64 * c, x.x, x.y);
    wk = c_mul(wk, w32);
  }
  if (live && lane == 0) emit(1024, v[0].x - v[0].y, 0.0);
  ADM_WAVE_SYNC();
}

static int grow(adm_mel* h, void** p, size_t bytes) {
  if (*p) dfree(*p);
  *p = nullptr;
  return dmalloc(p, bytes);
}

template <typename T>
static int upload(adm_mel* h, T** dst, const T* src, size_t n) {
  ADM_TRY(dmalloc((void**)dst, sizeof(T) * (n ? n : 1)));
  h->owned.push_back(*dst);
  ADM_TRY(copy_h2d(*dst, src, sizeof(T) * n, nullptr));
  return 0;
}

static size_t fft_smem(const adm_mel* h) { return sizeof(double) * 2 * (size_t)h->cfg.n_fft + sizeof(double) * (h->n_bins + 3); }

}  // namespace adm

using namespace adm;

extern "C" {

int adm_mel_create(const adm_mel_config* cfg, const double* window, const double* twiddle, const int* fb_start,
                   const int* fb_count, const float* fb_w32, const double* fb_w64, int nnz, const int* fbt_off,
                   const int* fbt_idx, const double* fbt_w64, const double* pinv, const float* wss, int nnls_cols,
                   adm_mel_t** out) {
  ADM_REQUIRE(cfg && window && twiddle && fb_start && fb_count && fb_w32 && fb_w64 && pinv && wss && out,
              "mel_create: null argument");
  ADM_REQUIRE(cfg->n_fft >= 64 && cfg->n_fft <= 4096 && (cfg->n_fft & (cfg->n_fft - 1)) == 0,
              "mel_create: n_fft must be a power of two in [64, 4096]");
  adm_mel* h = new adm_mel();
  h->cfg = *cfg;
  h->n_bins = cfg->n_fft / 2 + 1;
  h->n_mels = cfg->y_res;
  h->nnz = nnz;
  h->nnls_cols = nnls_cols;
  while ((1 << h->log2n) < cfg->n_fft) ++h->log2n;
  std::vector<int> off(h->n_mels + 1, 0);
  for (int m = 0; m < h->n_mels; ++m) off[m + 1] = off[m] + fb_count[m];
  ADM_REQUIRE(off[h->n_mels] == nnz, "mel_create: filterbank tap count mismatch");
  int rc = 0;
  rc |= upload(h, &h->window, window, cfg->n_fft);
  rc |= upload(h, &h->twiddle, twiddle, cfg->n_fft);
  rc |= upload(h, &h->fb_start, fb_start, h->n_mels);
  rc |= upload(h, &h->fb_count, fb_count, h->n_mels);
  rc |= upload(h, &h->fb_off, off.data(), h->n_mels + 1);
  rc |= upload(h, &h->fb_w32, fb_w32, nnz);
  rc |= upload(h, &h->fb_w64, fb_w64, nnz);
  rc |= upload(h, &h->fbt_off, fbt_off, h->n_bins + 1);
  rc |= upload(h, &h->fbt_idx, fbt_idx, nnz);
  rc |= upload(h, &h->fbt_w64, fbt_w64, nnz);
  rc |= upload(h, &h->pinv, pinv, (size_t)h->n_bins * h->n_mels);
  rc |= upload(h, &h->wss, wss, (size_t)cfg->hop_length * (cfg->x_res - 1));
  rc |= dmalloc((void**)&h->pgmax, 2 * sizeof(float));
  rc |= dmalloc((void**)&h->iters_dev, sizeof(int));
  rc |= stream_sync(nullptr);
  if (rc) { delete h; return -1; }
  h->owned.push_back(h->pgmax);
  h->owned.push_back(h->iters_dev);
  *out = h;
  return 0;
}

void adm_mel_destroy(adm_mel_t* h) {
  if (!h) return;
  for (void* p : h->owned) dfree(p);
  for (void* p : {(void*)h->melspec, (void*)h->angles, (void*)h->mag, (void*)h->ytmp, (void*)h->Smel, (void*)h->Xpow,
                  (void*)h->diff, (void*)h->reb0, (void*)h->reb1, (void*)h->y, (void*)h->pgblk, (void*)h->spec_max})
    if (p) dfree(p);
  delete h;
}

// audio: device, B slices of n_samples (fp32 or fp64), consecutive slices `slice_stride` elements apart.
// melspec_out: device (B, n_mels, n_frames) in the audio's precision, n_frames = 1 + n_samples / hop:
// librosa.feature.melspectrogram (mel.py:140-147) before the dB conversion.
static bool mel_fast_path(const adm_mel* h) {
  static const int fast = [] { const char* e = getenv("ADM_MEL_FAST"); return e ? atoi(e) : 1; }();
  return h->cfg.n_fft == 2048 && fast;
}

static int mel_cu_count() {
#if !defined(ADM_EMU)
  static int n = [] { int dev = 0, v = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return v > 0 ? v : 256; }();
  return n;
#else
  return 3;                                   // the emulator: several groups per workgroup
#endif
}

static int mel_forward_power_impl(adm_mel_t* h, const void* audio, int is_f64, int B, long slice_stride, int n_samples,
                                  void* melspec_out, unsigned long long* spec_max, void* stream) {
  ADM_REQUIRE(h && audio && melspec_out && B > 0 && n_samples > 0, "mel_forward_power: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const adm_mel_config& c = h->cfg;
  const int n_frames = 1 + n_samples / c.hop_length;
  dim3 grid(n_frames, B);
  const size_t smem = fft_smem(h);
  if (mel_fast_path(h)) {                    // one wave per frame, real-input radix-16 FFT (mel_stft2048_kernel)
    static const int occ_env = [] { const char* e = getenv("ADM_MEL_OCC"); return e ? atoi(e) : 2; }();
    const size_t tsz = is_f64 ? 8 : 4;
    auto lds_bytes = [&](int nw_) {
      return sizeof(double2) * nw_ * 16 * MF_PITCH + tsz * (size_t)h->n_mels * 2 * nw_ + sizeof(int) * (3 * (size_t)h->n_mels + 2) +
             tsz * (size_t)h->nnz;
    };
    const int occ = (occ_env == 2 && lds_bytes(8) <= 160 * 1024) ? 2 : 1;     // fp64 audio at 256 mels: the 8-wave image exceeds the LDS
    const int nw = occ == 2 ? 8 : 4, fr = 2 * nw;
    const int gpc = ceil_div(n_frames, fr), n_groups = gpc * B;
    dim3 g2(n_groups < mel_cu_count() ? n_groups : mel_cu_count());      // persistent: one workgroup per CU (its LDS image fills the CU)
    const size_t sm2 = lds_bytes(nw);
    ADM_REQUIRE(sm2 <= 160 * 1024, "mel_forward: too many mel bands for the fast path staging buffer");
#if !defined(ADM_EMU)
    static bool once = [] {
      (void)hipFuncSetAttribute((const void*)mel_stft2048_kernel<float, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)mel_stft2048_kernel<double, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)mel_stft2048_kernel<float, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)mel_stft2048_kernel<double, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      return true;
    }();
    (void)once;
#endif
#define ADM_MEL_FAST_LAUNCH(T_, W_)                                                                                          \
  ADM_LAUNCH((mel_stft2048_kernel<T_, W_>), g2, dim3(256 * W_), sm2, st, (const T_*)audio, slice_stride, n_samples, c.hop_length,   \
             h->window, (const double2*)h->twiddle, h->fb_start, h->fb_count, h->fb_off, h->fb_w32, h->fb_w64, h->n_mels,     \
             n_frames, (T_*)melspec_out, spec_max, gpc, n_groups)
    if (is_f64) { if (occ == 2) ADM_MEL_FAST_LAUNCH(double, 2); else ADM_MEL_FAST_LAUNCH(double, 1); }
    else { if (occ == 2) ADM_MEL_FAST_LAUNCH(float, 2); else ADM_MEL_FAST_LAUNCH(float, 1); }
#undef ADM_MEL_FAST_LAUNCH
    return ADM_CHECK_LAUNCH();
  }
  if (is_f64) {
    ADM_LAUNCH((mel_stft_power_kernel<double>), grid, dim3(256), smem, st, (const double*)audio, slice_stride, n_samples,
               c.n_fft, h->log2n, c.hop_length, h->window, (const double2*)h->twiddle, h->fb_start, h->fb_count,
               h->fb_off, h->fb_w32, h->fb_w64, h->n_mels, n_frames, (double*)melspec_out);
  } else {
    ADM_LAUNCH((mel_stft_power_kernel<float>), grid, dim3(256), smem, st, (const float*)audio, slice_stride, n_samples,
               c.n_fft, h->log2n, c.hop_length, h->window, (const double2*)h->twiddle, h->fb_start, h->fb_count,
               h->fb_off, h->fb_w32, h->fb_w64, h->n_mels, n_frames, (float*)melspec_out);
  }
  return ADM_CHECK_LAUNCH();
}

int adm_mel_forward_power(adm_mel_t* h, const void* audio, int is_f64, int B, long slice_stride, int n_samples,
                          void* melspec_out, void* stream) {
  return mel_forward_power_impl(h, audio, is_f64, B, slice_stride, n_samples, melspec_out, nullptr, stream);
}

// image_out: device (B, n_mels, n_frames) uint8 = the dB conversion + quantisation (mel.py:148-150) of the above.
int adm_mel_forward(adm_mel_t* h, const void* audio, int is_f64, int B, long slice_stride, int n_samples,
                    uint8_t* image_out, void* stream) {
  ADM_REQUIRE(h && audio && image_out && B > 0, "mel_forward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const adm_mel_config& c = h->cfg;
  const int n_frames = 1 + n_samples / c.hop_length;
  const size_t need = (size_t)B * h->n_mels * n_frames * 8;
  if (h->cap_fwd < need) { ADM_TRY(stream_sync(st)); ADM_TRY(grow(h, &h->melspec, need)); h->cap_fwd = need; }
  if (mel_fast_path(h)) {                    // the STFT kernel also delivers each spectrogram's maximum
    if (h->cap_max < (size_t)B) { ADM_TRY(stream_sync(st)); ADM_TRY(grow(h, (void**)&h->spec_max, sizeof(unsigned long long) * B)); h->cap_max = B; }
    ADM_TRY(dmemset(h->spec_max, 0, sizeof(unsigned long long) * B, st));
    ADM_TRY(mel_forward_power_impl(h, audio, is_f64, B, slice_stride, n_samples, h->melspec, h->spec_max, stream));
    const int n = h->n_mels * n_frames;
    dim3 g(ceil_div(n, 256 * 4), B);
    if (is_f64) ADM_LAUNCH((mel_db_u8_grid_kernel<double>), g, dim3(256), 0, st, (const double*)h->melspec, n, (float)c.top_db,
                           (const unsigned long long*)h->spec_max, image_out);
    else ADM_LAUNCH((mel_db_u8_grid_kernel<float>), g, dim3(256), 0, st, (const float*)h->melspec, n, (float)c.top_db,
                    (const unsigned long long*)h->spec_max, image_out);
    return ADM_CHECK_LAUNCH();
  }
  ADM_TRY(adm_mel_forward_power(h, audio, is_f64, B, slice_stride, n_samples, h->melspec, stream));
  if (is_f64) {
    ADM_LAUNCH((mel_db_u8_kernel<double>), dim3(B), dim3(256), 0, st, (const double*)h->melspec, h->n_mels * n_frames,
               (float)c.top_db, image_out);
  } else {
    ADM_LAUNCH((mel_db_u8_kernel<float>), dim3(B), dim3(256), 0, st, (const float*)h->melspec, h->n_mels * n_frames,
               (float)c.top_db, image_out);
  }
  return ADM_CHECK_LAUNCH();
}

// images: device (B, n_mels, n_frames) uint8; init_phase: device (B, n_bins, n_frames) fp64 in [0,1);
// audio_out: device (B, hop*(n_frames-1)) fp32; stft_mag_out: optional device (B, n_bins, n_frames) fp64 copy of the
// NNLS magnitude (tests); pg_max_host: optional, receives max |projected gradient| of the NNLS start point
// (librosa's L-BFGS-B returns the start point iff this is <= 1e-5).
int adm_mel_inverse(adm_mel_t* h, const uint8_t* images, const double* init_phase, int B, int n_frames,
                    float* audio_out, double* stft_mag_out, float* pg_max_host, void* stream) {
  ADM_REQUIRE(h && images && init_phase && audio_out && B > 0, "mel_inverse: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const adm_mel_config& c = h->cfg;
  const int nb = h->n_bins, nm = h->n_mels, nfft = c.n_fft, hop = c.hop_length;
  const int out_len = hop * (n_frames - 1);
  ADM_REQUIRE(n_frames == c.x_res, "mel_inverse: image width must equal x_res (window-sum table is built for it)");
  const size_t spec = (size_t)B * n_frames * nb;
  const size_t need = spec * 16;
  if (h->cap_inv < need) {
    ADM_TRY(stream_sync(st));
    ADM_TRY(grow(h, (void**)&h->angles, spec * 16));
    ADM_TRY(grow(h, (void**)&h->mag, spec * 8));
    ADM_TRY(grow(h, (void**)&h->Xpow, spec * 8));
    ADM_TRY(grow(h, (void**)&h->reb0, spec * 8));
    ADM_TRY(grow(h, (void**)&h->reb1, spec * 8));
    ADM_TRY(grow(h, (void**)&h->ytmp, (size_t)B * n_frames * nfft * 8));
    ADM_TRY(grow(h, (void**)&h->Smel, (size_t)B * nm * n_frames * 8));
    ADM_TRY(grow(h, (void**)&h->diff, (size_t)B * nm * n_frames * 8));
    ADM_TRY(grow(h, (void**)&h->y, (size_t)B * out_len * 4));
    h->cap_inv = need;
  }
  const long npix = (long)B * nm * n_frames;
  ADM_LAUNCH(mel_u8_to_power_kernel, dim3((unsigned)((npix + 255) / 256 > 2048 ? 2048 : (npix + 255) / 256)), dim3(256), 0,
             st, images, npix, (double)c.top_db, h->Smel);
  ADM_LAUNCH(mel_pinv_gemm_kernel, dim3(ceil_div(nb, 64), ceil_div(n_frames, 64), B), dim3(256), 0, st, h->pinv, h->Smel,
             nb, nm, n_frames, h->Xpow, h->mag);
  // NNLS (mel.py:165 -> librosa.util.nnls): start point clip(pinv S, 0) above; its projected gradient per column block
  // decides, as in L-BFGS-B, whether the block is returned as it is (nit = 0: every dB image at the usual configurations)
  // or solved on the device (mel_nnls_solve_kernel: low sample rates / tiny filterbanks).
  const int n_blk = ceil_div(n_frames, h->nnls_cols);
  if (h->cap_blk < (size_t)B * n_blk) {
    ADM_TRY(stream_sync(st));
    ADM_TRY(grow(h, (void**)&h->pgblk, sizeof(float) * (size_t)B * n_blk));
    h->cap_blk = (size_t)B * n_blk;
  }
  ADM_TRY(dmemset(h->pgmax, 0, 2 * sizeof(float), st));
  ADM_TRY(dmemset(h->pgblk, 0, sizeof(float) * (size_t)B * n_blk, st));
  ADM_TRY(dmemset(h->iters_dev, 0, sizeof(int), st));
  ADM_LAUNCH(mel_nnls_diff_kernel, dim3(256, B), dim3(256), 0, st, h->Xpow, h->Smel, h->fb_start, h->fb_count, h->fb_off,
             h->fb_w64, nb, nm, n_frames, h->diff);
  const bool solve = h->lipschitz > 0.0 && h->nnls_max_iter > 0;
  const int pg_parts = ceil_div(256, n_blk) > 0 ? ceil_div(256, n_blk) : 1;     // ~256 workgroups per image, whole blocks each
  ADM_LAUNCH(mel_nnls_pg_kernel, dim3(n_blk * pg_parts, B), dim3(256), 0, st, h->Xpow, h->diff, h->fbt_off, h->fbt_idx, h->fbt_w64,
             nb, nm, n_frames, h->nnls_cols, pg_parts, h->pgmax + (solve ? 1 : 0), solve ? h->pgblk : (float*)nullptr,
             (const float*)nullptr, 0.f);
  if (solve) {
    const size_t nsm = sizeof(double) * (3 * (size_t)nb + 2 * (size_t)nm);
    ADM_REQUIRE(nsm <= 64 * 1024, "mel_inverse: filterbank too large for the in-LDS NNLS solver");
    ADM_LAUNCH(mel_nnls_solve_kernel, dim3(n_frames, B), dim3(256), nsm, st, h->Xpow, h->mag, (const double*)h->Smel,
               h->fb_start, h->fb_count, h->fb_off, h->fb_w64, h->fbt_off, h->fbt_idx, h->fbt_w64, nb, nm, n_frames,
               h->nnls_cols, (const float*)h->pgblk, 1e-5f, 1.0 / h->lipschitz, 1e-6, h->nnls_max_iter, h->iters_dev);
    // projected gradient of the point that is returned
    ADM_LAUNCH(mel_nnls_diff_kernel, dim3(256, B), dim3(256), 0, st, h->Xpow, h->Smel, h->fb_start, h->fb_count, h->fb_off,
               h->fb_w64, nb, nm, n_frames, h->diff, (const float*)h->pgblk, h->nnls_cols, 1e-5f);
    ADM_LAUNCH(mel_nnls_pg_kernel, dim3(n_blk * pg_parts, B), dim3(256), 0, st, h->Xpow, h->diff, h->fbt_off, h->fbt_idx,
               h->fbt_w64, nb, nm, n_frames, h->nnls_cols, pg_parts, h->pgmax, (float*)nullptr, (const float*)h->pgblk, 1e-5f);
  }
  ADM_LAUNCH(gl_init_kernel, dim3(256, B), dim3(256), 0, st, init_phase, h->mag, nb, n_frames, (double2*)h->angles);
  const size_t smem = fft_smem(h);
  const float mom = (float)(0.99 / (1.0 + 0.99));
  float2* reb = (float2*)h->reb0;
  float2* tprev = nullptr;
  float2* spare = (float2*)h->reb1;
  dim3 fgrid(n_frames, B), ogrid(ceil_div(out_len, 256) > 1024 ? 1024 : ceil_div(out_len, 256), B);
  const bool fast = mel_fast_path(h);
  const dim3 fgrid4(ceil_div(n_frames, 4), B);
  const size_t smem4 = sizeof(double2) * 4 * 16 * MF_PITCH;
#if !defined(ADM_EMU)
  static bool once_gl = [] {
    (void)hipFuncSetAttribute((const void*)gl_istft_frames2048_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)gl_stft_update2048_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    return true;
  }();
  (void)once_gl;
#endif
  auto istft_frames = [&]() {
    if (fast) ADM_LAUNCH(gl_istft_frames2048_kernel, fgrid4, dim3(256), smem4, st, (const double2*)h->angles, h->window,
                         (const double2*)h->twiddle, n_frames, h->ytmp);
    else ADM_LAUNCH(gl_istft_frames_kernel, fgrid, dim3(256), smem, st, (const double2*)h->angles, nfft, h->log2n, h->window,
                    (const double2*)h->twiddle, n_frames, h->ytmp);
  };
  for (int it = 0; it < c.n_iter; ++it) {
    istft_frames();
    ADM_LAUNCH(gl_overlap_add_kernel, ogrid, dim3(256), 0, st, h->ytmp, nfft, hop, n_frames, h->wss, out_len, h->y);
    if (fast) ADM_LAUNCH(gl_stft_update2048_kernel, fgrid4, dim3(256), smem4, st, h->y, out_len, hop, h->window,
                         (const double2*)h->twiddle, n_frames, reb, (const float2*)tprev, mom, h->mag, (double2*)h->angles);
    else ADM_LAUNCH(gl_stft_update_kernel, fgrid, dim3(256), smem, st, h->y, out_len, nfft, h->log2n, hop, h->window,
                    (const double2*)h->twiddle, n_frames, reb, (const float2*)tprev, mom, h->mag, (double2*)h->angles);
    // rebuilt, tprev = tprev, rebuilt
    float2* old = tprev;
    tprev = reb;
    reb = old ? old : spare;
  }
  istft_frames();
  ADM_LAUNCH(gl_overlap_add_kernel, ogrid, dim3(256), 0, st, h->ytmp, nfft, hop, n_frames, h->wss, out_len, audio_out);
  ADM_TRY(ADM_CHECK_LAUNCH());
  if (stft_mag_out) ADM_TRY(copy_d2d(stft_mag_out, h->mag, spec * 8, st));  // layout (B, n_frames, n_bins)
  if (pg_max_host) {
    float pg2[2] = {0.f, 0.f};
    ADM_TRY(copy_d2h(pg2, h->pgmax, 2 * sizeof(float), st));
    ADM_TRY(copy_d2h(&h->last_iters, h->iters_dev, sizeof(int), st));
    ADM_TRY(stream_sync(st));
    *pg_max_host = pg2[0];
    h->last_pg_start = (h->lipschitz > 0.0 && h->nnls_max_iter > 0) ? pg2[1] : pg2[0];
  }
  return 0;
}

// Enables the on-device NNLS solver for column blocks whose start point fails L-BFGS-B's projected-gradient rule:
// lipschitz = lambda_max(A A^T) of the fp64 filterbank (computed by the caller), max_iter = iteration cap per column.
int adm_mel_set_nnls_solver(adm_mel_t* h, double lipschitz, int max_iter) {
  ADM_REQUIRE(h && lipschitz >= 0.0 && max_iter >= 0, "mel_set_nnls_solver: bad argument");
  h->lipschitz = lipschitz;
  h->nnls_max_iter = max_iter;
  return 0;
}

// Diagnostics of the last adm_mel_inverse that was given pg_max_host: max |projected gradient| of the NNLS START point and
// the largest number of solver iterations any column ran (0: every block was returned as it was, scipy's nit = 0).
int adm_mel_last_nnls(adm_mel_t* h, float* pg_start, int* iterations) {
  ADM_REQUIRE(h && pg_start && iterations, "mel_last_nnls: null argument");
  *pg_start = h->last_pg_start;
  *iterations = h->last_iters;
  return 0;
}

}  // extern "C"
