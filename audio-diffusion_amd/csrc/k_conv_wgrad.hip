// k_conv_wgrad.hip — weight gradient of the fused convolution on the exact-f32 matrix core.
//   dW[co][c][tap] = sum_{n, oy, ox} dy[n][co][oy][ox] * a[n][c][oy*s + ky - pad][ox*s + kx - pad]
// where a = act(gn(concat(x1,x2))) (optionally nearest-x2 upsampled) is RECOMPUTED in the load path exactly as the
// forward kernel does (k_conv_mfma.hip), so the normalised/activated tensor never exists in HBM in training either.
// GEMM view: M = Cout, N = (tap, c), K = output pixels. Workgroup = 4 waves: 128 couts x {32 channels x 9 taps | 128
// channels (1x1)}; wave w owns cout rows [32w, 32w+32) and all column tiles (9 resp. 4 accumulator fragments).
// K runs over 64-pixel tiles (NI images x TH x TW, TW = min(Wo,16), TH = min(Ho,4)); per tile:
//   ldsD [64 px][129]     dy transposed on the way in, so the A operand (32 couts of one pixel) is conflict-free;
//   ldsP [channels][odd]  the haloed activated patch, odd channel stride -> the B operand (32 channels of one tap) too.
// Split-K: every workgroup reduces a contiguous range of pixel tiles and writes its 128 x (channels x taps) partial
// in final (Cout,Cin,ks,ks) order to a workspace slab; wgrad_reduce_kernel sums the slabs into the gradient buffer.
// Algorithmic FLOPs = forward FLOPs of the same layer; bytes = dy + x read once per (cout-tile, channel-chunk) pair.
#include <cstdlib>

#include "adm_kernels.h"

namespace adm {

struct WgradParams {
  const float* x1; const float* x2; int C1, C2;
  const float* dy; int Cout;
  int N, Hs, Ws, Hi, Wi, Ho, Wo, up, pad_lo;
  const float* gn_scale; const float* gn_shift; int act;
  float* part;              // [split][Cout*Cin*ks*ks]
  int lTW, lTH, tiles_x, tiles_y, n_ptiles, IH, IW, PS /* odd channel stride of the patch */;
  int n_ct, n_chunks, split, tiles_per_block;
  long x1_bs, x2_bs;
  // exact reciprocals (floor(2^32 / d) + 1) of the divisors of the per-element index math: n / d == umulhi(n, m) for
  // n, d < 2^16 — a runtime integer division costs ~40 VALU instructions, and the patch loop did 5 per element and tile
  unsigned mPE, mIHW, mIW, mTX, mTXY;
  unsigned long long* prof;   // developer aid (ADM_WGRAD_PROF=1, generic stride-2 kernel): per-phase cycle counters, else NULL
};

__device__ __forceinline__ int fdiv(int n, unsigned magic) {   // n / d for 0 <= n < 2^16; magic = floor(2^32 / d) + 1,
  return magic ? (int)(((unsigned long long)(unsigned)n * magic) >> 32) : n;   // or 0 for d == 1 (not representable)
}

__device__ __forceinline__ float silu_g(float v) { return v * ADM_RCP(1.0f + __expf(-v)); }

#if defined(ADM_EMU)
#define WG_CLK() 0ull
#else
#define WG_CLK() ((unsigned long long)__builtin_readcyclecounter())
#endif
#define WG_LAP(slot) do { if (PROF) { const unsigned long long tn_ = WG_CLK(); pr[slot] += tn_ - tq; tq = tn_; } } while (0)
template <int KS, int STRIDE, bool PROF = false>
__global__ void __launch_bounds__(256, 2) conv_wgrad_kernel(const WgradParams p) {
  constexpr int KS2 = KS * KS;
  unsigned long long pr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = PROF ? WG_CLK() : 0ull;
  const unsigned long long t_start = tq;
  constexpr int NT = KS == 3 ? 9 : 4;     // column tiles (of 32 channels) per wave
  constexpr int CB = KS == 3 ? 32 : 128;  // channels per workgroup
  constexpr int DLD = 129;
  ADM_DYN_SMEM(float, smem);
  float* ldsD = smem;                // 64 * 129
  float* ldsP = smem + 64 * DLD;     // CB * PS
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  int b = blockIdx.x;
  const int sp = b % p.split; b /= p.split;
  const int chunk = b % p.n_chunks, ct = b / p.n_chunks;
  const int m0 = ct * 128, c0 = chunk * CB;
  const int Ct = p.C1 + p.C2;
  const int TW = 1 << p.lTW, TH = 1 << p.lTH, NI = 64 >> (p.lTW + p.lTH);
  const int IHW = p.IH * p.IW, planeS = p.Hs * p.Ws;
  const long planeO = (long)p.Ho * p.Wo;

  f32x16 acc[NT];
  ADM_UNROLL
  for (int t = 0; t < NT; ++t)
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int t_begin = sp * p.tiles_per_block;
  int t_end = t_begin + p.tiles_per_block;
  if (t_end > p.n_ptiles) t_end = p.n_ptiles;
  for (int pt = t_begin; pt < t_end; ++pt) {
    const int tx = pt % p.tiles_x, ty = (pt / p.tiles_x) % p.tiles_y, ig = pt / (p.tiles_x * p.tiles_y);
    const int n0 = ig * NI;
    WG_LAP(0);
    // ---- dy tile, transposed into [px][co] --------------------------------------------------------------
    // Both staging loops issue their loads in UNCONDITIONAL batches of eight (clamped address, value masked afterwards): with a
    // per-element "load or not" branch hipcc waits vmcnt(0) after every load, and ADM_WGRAD_PROF=1 showed the stride-2 layers
    // spending 26k + 68k cycles per tile on 32 + 37 serialized load round trips next to 25k cycles of MFMAs (31 TF/s).
    for (int e0 = tid; e0 < 128 * 64; e0 += 256 * 8) {
      float v[8];
      bool okv[8];
      ADM_UNROLL
      for (int j = 0; j < 8; ++j) {
        const int e = e0 + 256 * j;
        const int co = e >> 6, pp = e & 63;
        const int px = pp & (TW - 1), py = (pp >> p.lTW) & (TH - 1), img = pp >> (p.lTW + p.lTH);
        const int oy = ty * TH + py, ox = tx * TW + px, n = n0 + img;
        okv[j] = m0 + co < p.Cout && n < p.N && oy < p.Ho && ox < p.Wo;
        const long off = okv[j] ? ((long)n * p.Cout + m0 + co) * planeO + (long)oy * p.Wo + ox : 0;
        v[j] = p.dy[off];
      }
      ADM_UNROLL
      for (int j = 0; j < 8; ++j) {
        const int e = e0 + 256 * j;
        ldsD[(e & 63) * DLD + (e >> 6)] = okv[j] ? v[j] : 0.f;
      }
    }
    WG_LAP(1);                // dy staging
    // ---- activated input patch: patch element outer (its placement costs three runtime divisions), channels inner ------------
    const bool has_gn = p.gn_scale != nullptr;
    for (int q = tid; q < NI * IHW; q += 256) {
      const int img = q / IHW, r2 = q - img * IHW;
      const int ly = r2 / p.IW, lx = r2 - ly * p.IW;
      const int gy = ty * TH * STRIDE + ly - p.pad_lo, gx = tx * TW * STRIDE + lx - p.pad_lo;
      const int n = n0 + img;
      const bool ok = n < p.N && gy >= 0 && gy < p.Hi && gx >= 0 && gx < p.Wi;
      const int sy = p.up ? (gy >> 1) : gy, sx = p.up ? (gx >> 1) : gx;
      const long o1 = ok ? (long)n * p.x1_bs + sy * p.Ws + sx : 0, o2 = ok ? (long)n * p.x2_bs + sy * p.Ws + sx : 0;
      const long gno = ok ? (long)n * Ct : 0;
      for (int cb = 0; cb < CB; cb += 8) {
        float v[8], sc[8], sh[8];
        ADM_UNROLL
        for (int j = 0; j < 8; ++j) {
          const int cc = c0 + cb + j;                       // wave-uniform: the source select is a scalar branch
          const int ccl = cc < Ct ? cc : 0;
          v[j] = ccl < p.C1 ? p.x1[o1 + (long)ccl * planeS] : p.x2[o2 + (long)(ccl - p.C1) * planeS];
          if (has_gn) { sc[j] = p.gn_scale[gno + ccl]; sh[j] = p.gn_shift[gno + ccl]; }
        }
        ADM_UNROLL
        for (int j = 0; j < 8; ++j) {
          float y = v[j];
          if (has_gn) y = y * sc[j] + sh[j];
          if (p.act) y = silu_g(y);
          ldsP[(cb + j) * p.PS + q] = (ok && c0 + cb + j < Ct) ? y : 0.f;
        }
      }
    }
    WG_LAP(2);                // patch staging
    __syncthreads();
    WG_LAP(3);                // barrier
    // ---- 32 k-steps (pixel pairs) x NT MFMAs -----------------------------------------------------------------
#pragma unroll 4
    for (int s = 0; s < 32; ++s) {       // four steps' operand reads in flight (rolled, every step paid its LDS latency: 25k cycles for 18.4k of MFMAs)
      const int pp = 2 * s + h;
      const int px = pp & (TW - 1), py = (pp >> p.lTW) & (TH - 1), img = pp >> (p.lTW + p.lTH);
      const int poff = img * IHW + py * STRIDE * p.IW + px * STRIDE;
      const float av = ldsD[pp * DLD + wave * 32 + l31];
      ADM_UNROLL
      for (int t = 0; t < NT; ++t) {
        float bv;
        if (KS == 3) bv = ldsP[l31 * p.PS + poff + (t / 3) * p.IW + (t % 3)];
        else bv = ldsP[(t * 32 + l31) * p.PS + poff];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
      }
    }
    WG_LAP(4);                // MFMA loop
    __syncthreads();
    WG_LAP(5);                // barrier
  }
  // ---- partial result in (Cout,Cin,ks,ks) order --------------------------------------------------------------
  float* out = p.part + (long)sp * p.Cout * Ct * KS2;
  ADM_UNROLL
  for (int t = 0; t < NT; ++t) {
    const int cc = KS == 3 ? c0 + l31 : c0 + t * 32 + l31;
    const int tap = KS == 3 ? t : 0;
    if (cc >= Ct) continue;
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int co = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (co < p.Cout) out[((long)tap * p.Cout + co) * Ct + cc] = acc[t][r];      // slab layout [tap][cout][cin]: see wgrad_reduce_kernel
    }
  }
  if (PROF) {
    WG_LAP(6);
    pr[7] = WG_CLK() - t_start;
    if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(p.prof + i, pr[i]);
  }
}


// ---------------------------------------------------------------------------------------------------------
// Software-pipelined variant (stride 1; patch of at most 128 elements per channel): the NEXT pixel tile's dy values and
// raw activations are loaded into registers before the 288 (resp. 128) MFMAs of the current tile and are
// normalised/activated/transposed into LDS after them, so global-load latency hides under the matrix work.
// One workgroup per CU (up to 512 VGPRs per lane: 144 accumulators + 48 prefetch registers + addresses, no spills).
template <int KS, bool FAST>
__global__ void __launch_bounds__(256, 1) conv_wgrad_pf_kernel(const WgradParams p) {
  constexpr int KS2 = KS * KS;
  constexpr int NT = KS == 3 ? 9 : 4;
  constexpr int CB = KS == 3 ? 32 : 128;
  constexpr int NPX = KS == 3 ? 16 : 32;   // patch elements per thread: CB * (<=128 | 64) / 256
  constexpr int DLD = 129;
  constexpr bool GN_PREFETCH = FAST || KS == 3;
  ADM_DYN_SMEM(float, smem);
  float* ldsD = smem;
  float* ldsP = smem + 64 * DLD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  int b = blockIdx.x;
  const int sp = b % p.split; b /= p.split;
  const int chunk = b % p.n_chunks, ct = b / p.n_chunks;
  const int m0 = ct * 128, c0 = chunk * CB;
  const int Ct = p.C1 + p.C2;
  const int TW = 1 << p.lTW, TH = 1 << p.lTH, NI = 64 >> (p.lTW + p.lTH);
  const int IHW = p.IH * p.IW, planeS = p.Hs * p.Ws, PE = NI * IHW;   // PE <= 128 (KS=3) / == 64 (KS=1)
  const long planeO = (long)p.Ho * p.Wo;
  const int n_el = CB * PE;

  f32x16 acc[NT];
  ADM_UNROLL
  for (int t = 0; t < NT; ++t)
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // per-thread roles, constant over tiles: dy -> pixel pp = tid & 63, couts (tid >> 6) + 4 i;
  // patch -> elements e = tid + 256 j  (channel ec[j], patch position eq[j])
  const int pp_d = tid & 63, co_d0 = tid >> 6;
  const int dpx = pp_d & (TW - 1), dpy = (pp_d >> p.lTW) & (TH - 1), dimg = pp_d >> (p.lTW + p.lTH);
  float dyr[32], xr[NPX];
  float gsr[NPX], ghr[NPX];   // GroupNorm scale/shift of the prefetched elements: fetched WITH them — loading them in stash()
                              // serialized 16 dependent L2 round trips per tile (each followed by vmcnt(0))
  unsigned xvalid = 0;   // bit j: patch element j of this thread is inside the image (else zero padding)
  int n0_st = 0;         // first image of the tile held in registers

  auto issue = [&](int pt) {
    const int ig = p.n_ptiles < 65536 ? fdiv(pt, p.mTXY) : pt / (p.tiles_x * p.tiles_y);   // wave-uniform
    const int rem = pt - ig * (p.tiles_x * p.tiles_y);
    const int ty = fdiv(rem, p.mTX), tx = rem - ty * p.tiles_x;
    const int n0 = ig * NI;
    n0_st = n0;
    xvalid = 0;
    {
      const int oy = ty * TH + dpy, ox = tx * TW + dpx, n = n0 + dimg;
      const bool ok = n < p.N && oy < p.Ho && ox < p.Wo;
      const float* src = p.dy + ((long)n * p.Cout + m0 + co_d0) * planeO + (long)oy * p.Wo + ox;
      ADM_UNROLL
      for (int i = 0; i < 32; ++i)
        dyr[i] = (ok && m0 + co_d0 + 4 * i < p.Cout) ? src[(long)(4 * i) * planeO] : 0.f;
    }
    ADM_UNROLL
    for (int j = 0; j < NPX; ++j) {
      const int e = tid + 256 * j;
      xr[j] = 0.f;
      if (e < n_el) {
        const int c = fdiv(e, p.mPE), q = e - c * PE;
        const int img = fdiv(q, p.mIHW), r2 = q - img * IHW;
        const int ly = fdiv(r2, p.mIW), lx = r2 - ly * p.IW;
        const int gy = ty * TH + ly - p.pad_lo, gx = tx * TW + lx - p.pad_lo;
        const int n = n0 + img, cc = c0 + c;
        if (cc < Ct && n < p.N && gy >= 0 && gy < p.Hi && gx >= 0 && gx < p.Wi) {
          const int sy = p.up ? (gy >> 1) : gy, sx = p.up ? (gx >> 1) : gx;
          xvalid |= 1u << j;
          xr[j] = cc < p.C1 ? p.x1[(long)n * p.x1_bs + (long)cc * planeS + sy * p.Ws + sx]
                            : p.x2[(long)n * p.x2_bs + (long)(cc - p.C1) * planeS + sy * p.Ws + sx];
          if (GN_PREFETCH && p.gn_scale != nullptr) {
            const long gi = (long)n * Ct + cc;
            gsr[j] = p.gn_scale[gi]; ghr[j] = p.gn_shift[gi];
          }
        }
      }
    }
  };
  // ---- fast prefetch path (no upsample fold, the channel chunk lies in ONE source tensor, full cout tile): everything that
  // does not depend on the tile is computed once — the generic `issue` below re-derives channel / image / row / column
  // and a 64-bit address with bounds branches for each of the 48 elements of every tile (~3500 instructions per tile and
  // thread, more issue time than the tile's 288 MFMAs); here a tile costs one uniform base + per-element compares, and
  // every load is unconditional (clamped address, zeroed through `xvalid`) so nothing waits at a branch join.
  constexpr bool fast = FAST;   // chosen by the launcher: p.up == 0, C1 % CB == 0, Ct % CB == 0, Cout % 128 == 0
  const float* xsrc = c0 < p.C1 ? p.x1 + (long)c0 * planeS : p.x2 + (long)(c0 - p.C1) * planeS;
  const long xbs = c0 < p.C1 ? p.x1_bs : p.x2_bs;
  int eoff[NPX];       // element offset relative to the tile origin: image * batch stride + channel plane + row + column
  int epk[NPX];        // packed roles: lx | ly << 8 | img << 16 | c << 24 | enabled << 31
  if constexpr (fast) {
    ADM_UNROLL
    for (int j = 0; j < NPX; ++j) {
      const int e = tid + 256 * j;
      const bool en = e < n_el;
      const int ec = en ? e : 0;
      const int c = fdiv(ec, p.mPE), q = ec - c * PE;
      const int img = fdiv(q, p.mIHW), r2 = q - img * IHW;
      const int ly = fdiv(r2, p.mIW), lx = r2 - ly * p.IW;
      eoff[j] = (int)(img * xbs) + c * planeS + (ly - p.pad_lo) * p.Ws + (lx - p.pad_lo);
      epk[j] = lx | (ly << 8) | (img << 16) | (c << 24) | (en ? (int)0x80000000 : 0);
    }
  }
  auto issue_fast = [&](int pt) __attribute__((always_inline)) {
    const int ig = p.n_ptiles < 65536 ? fdiv(pt, p.mTXY) : pt / (p.tiles_x * p.tiles_y);   // wave-uniform
    const int rem = pt - ig * (p.tiles_x * p.tiles_y);
    const int ty = fdiv(rem, p.mTX), tx = rem - ty * p.tiles_x;
    const int n0 = ig * NI;
    n0_st = n0;
    {
      const int oy = ty * TH + dpy, ox = tx * TW + dpx, n = n0 + dimg;
      const bool ok = (n < p.N) & (oy < p.Ho) & (ox < p.Wo);
      const float* src = p.dy + ((long)(ok ? n : 0) * p.Cout + m0 + co_d0) * planeO + (ok ? (long)oy * p.Wo + ox : 0);
      ADM_UNROLL
      for (int i = 0; i < 32; ++i) {
        const float v = src[(long)(4 * i) * planeO];
        dyr[i] = ok ? v : 0.f;
      }
    }
    const float* xt = xsrc + (long)n0 * xbs + (long)(ty * TH) * p.Ws + tx * TW;   // wave-uniform tile origin
    const float* gs = p.gn_scale ? p.gn_scale + (long)n0 * Ct + c0 : nullptr;
    const float* gh = p.gn_scale ? p.gn_shift + (long)n0 * Ct + c0 : nullptr;
    const int gy0 = ty * TH - p.pad_lo, gx0 = tx * TW - p.pad_lo;
    xvalid = 0;
    ADM_UNROLL
    for (int j = 0; j < NPX; ++j) {
      const int lx = epk[j] & 255, ly = (epk[j] >> 8) & 255, img = (epk[j] >> 16) & 255, c = (epk[j] >> 24) & 127;
      const int gy = gy0 + ly, gx = gx0 + lx;
      // bitwise, not short-circuit: && compiles to a branch per term and per element
      const bool ok = (epk[j] < 0) & (n0 + img < p.N) & ((unsigned)gy < (unsigned)p.Hi) & ((unsigned)gx < (unsigned)p.Wi);
      xr[j] = xt[ok ? eoff[j] : 0 - (ty * TH) * p.Ws - tx * TW];      // clamped to the first element of the chunk's plane
      if (gs != nullptr) {
        const int gi = ok ? img * Ct + c : 0;
        gsr[j] = gs[gi]; ghr[j] = gh[gi];
      }
      xvalid |= ok ? 1u << j : 0u;
    }
  };
  auto stash = [&]() {
    ADM_UNROLL
    for (int i = 0; i < 32; ++i) ldsD[pp_d * DLD + co_d0 + 4 * i] = dyr[i];
    ADM_UNROLL
    for (int j = 0; j < NPX; ++j) {
      const int e = tid + 256 * j;
      if (e < n_el) {
        const int c = fdiv(e, p.mPE), q = e - c * PE;
        float v = xr[j];
        if ((xvalid >> j) & 1u) {
          if (p.gn_scale != nullptr) {
            if (GN_PREFETCH) {
              v = v * gsr[j] + ghr[j];
            } else {   // generic 1x1 variant: 64 more prefetch registers would spill
              const long gi = (long)(n0_st + fdiv(q, p.mIHW)) * Ct + c0 + c;
              v = v * p.gn_scale[gi] + p.gn_shift[gi];
            }
          }
          if (p.act) v = silu_g(v);
        } else {
          v = 0.f;
        }
        ldsP[c * p.PS + q] = v;
      }
    }
  };

  const int t_begin = sp * p.tiles_per_block;
  int t_end = t_begin + p.tiles_per_block;
  if (t_end > p.n_ptiles) t_end = p.n_ptiles;
  if (t_begin < t_end) { if constexpr (fast) issue_fast(t_begin); else issue(t_begin); }
  for (int pt = t_begin; pt < t_end; ++pt) {
    stash();
    __syncthreads();
    if (pt + 1 < t_end) { if constexpr (fast) issue_fast(pt + 1); else issue(pt + 1); }
    // operand words of k-step s+1 are requested before the MFMAs of k-step s (one wave per SIMD: nothing else hides the
    // LDS latency; read -> wait -> MFMA per group of three left the matrix pipe ~45 % busy)
    auto fetch = [&](int s, float& av, float (&bv)[NT]) __attribute__((always_inline)) {
      const int pp = 2 * s + h;
      const int px = pp & (TW - 1), py = (pp >> p.lTW) & (TH - 1), img = pp >> (p.lTW + p.lTH);
      const int poff = img * IHW + py * p.IW + px;
      av = ldsD[pp * DLD + wave * 32 + l31];
      ADM_UNROLL
      for (int t = 0; t < NT; ++t) {
        if (KS == 3) bv[t] = ldsP[l31 * p.PS + poff + (t / 3) * p.IW + (t % 3)];
        else bv[t] = ldsP[(t * 32 + l31) * p.PS + poff];
      }
    };
    float a0, a1, b0[NT], b1[NT];
    fetch(0, a0, b0);
    for (int s = 0; s < 32; s += 2) {
      fetch(s + 1, a1, b1);
      ADM_SCHED_FENCE();
      ADM_UNROLL
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[t], acc[t], 0, 0, 0);
      if (s + 2 < 32) fetch(s + 2, a0, b0);
      ADM_SCHED_FENCE();
      ADM_UNROLL
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[t], acc[t], 0, 0, 0);
    }
    __syncthreads();
  }
  float* out = p.part + (long)sp * p.Cout * Ct * KS2;
  ADM_UNROLL
  for (int t = 0; t < NT; ++t) {
    const int cc = KS == 3 ? c0 + l31 : c0 + t * 32 + l31;
    const int tap = KS == 3 ? t : 0;
    if (cc >= Ct) continue;
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int co = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (co < p.Cout) out[((long)tap * p.Cout + co) * Ct + cc] = acc[t][r];      // slab layout [tap][cout][cin]: see wgrad_reduce_kernel
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Software-pipelined 3x3 stride-1 variant (the 40 % of a training step): with 144 accumulator registers there is ONE wave
// per SIMD, so nothing but the wave's own instruction order can overlap the staging work with the matrix pipe. The LDS
// tiles are double-buffered and the staging of the following tiles is cut into 16 pieces that ride behind the 16 MFMA
// groups of the current tile (an MFMA group = 18 MFMAs = 1152 pipe cycles, a piece ~60 instructions):
//   groups 0..7  : registers of tile t+1 -> LDS buffer (t+1) & 1   (dy transposed, GroupNorm + SiLU on the patch)
//   groups 8..15 : global loads of tile t+2 -> the registers just freed (half a tile ~ 4 us ahead of their use)
// One barrier per tile. Launcher guarantees the fast-path conditions: no upsample fold, C1 % 32 == 0, Ct % 32 == 0,
// Cout % 128 == 0, patch of a 64-pixel tile <= 128 elements per channel.
__global__ void __launch_bounds__(256, 1) conv_wgrad_sp_kernel(const WgradParams p) {
  // fixed tile geometry (launcher: Wo % 16 == 0 handled by TW = 16, TH = 4, one image per 64-pixel tile): every LDS address
  // of the MFMA loop is then a lane term plus a compile-time offset — with runtime geometry the fully unrolled loop kept
  // ~64 hoisted address registers alive and spilled
  constexpr int NT = 9, CB = 32, NPX = 16, DLD = 129;
  constexpr int TW = 16, TH = 4, NI = 1, IW = 18, IHW = 108, PE = 108, PS = 109;
  constexpr int PBUF = 64 * DLD + CB * PS;        // floats per LDS buffer: dy tile + patch
  ADM_DYN_SMEM(float, smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  int b = blockIdx.x;
  const int sp = b % p.split; b /= p.split;
  const int chunk = b % p.n_chunks, ct = b / p.n_chunks;
  const int m0 = ct * 128, c0 = chunk * CB;
  const int Ct = p.C1 + p.C2;
  const int planeS = p.Hs * p.Ws;
  const long planeO = (long)p.Ho * p.Wo;
  const int n_el = CB * PE;

  f32x16 acc[NT];
  ADM_UNROLL
  for (int t = 0; t < NT; ++t)
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // tile-invariant roles
  const int pp_d = tid & 63, co_d0 = tid >> 6;
  const int dpx = pp_d & (TW - 1), dpy = pp_d >> 4, dimg = 0;
  const float* xsrc = c0 < p.C1 ? p.x1 + (long)c0 * planeS : p.x2 + (long)(c0 - p.C1) * planeS;
  const long xbs = c0 < p.C1 ? p.x1_bs : p.x2_bs;
  int eoff[NPX], epk[NPX];
  ADM_UNROLL
  for (int j = 0; j < NPX; ++j) {
    const int e = tid + 256 * j;
    const bool en = e < n_el;
    const int ec = en ? e : 0;
    const int c = ec / PE, q = ec - c * PE;
    const int img = 0, r2 = q;
    const int ly = r2 / IW, lx = r2 - ly * IW;
    eoff[j] = (int)(img * xbs) + c * planeS + (ly - p.pad_lo) * p.Ws + (lx - p.pad_lo);
    epk[j] = lx | (ly << 8) | (img << 16) | (c << 24) | (en ? (int)0x80000000 : 0);
  }
  float dyr[32], xr[NPX];
  int n0s = 0;             // first image of the tile held in xr (GroupNorm rows are read at stash time: the latency hides
                           // behind the running MFMA group, and 32 prefetch registers fewer keep the kernel spill-free)
  unsigned xvalid = 0;
  // tile-uniform state of the tile being loaded
  const float *xt = xsrc, *dsrc = p.dy;
  int gy0 = 0, gx0 = 0, n0l = 0, back = 0;
  bool dok = false;
  auto load_begin = [&](int pt) __attribute__((always_inline)) {
    const int ig = p.n_ptiles < 65536 ? fdiv(pt, p.mTXY) : pt / (p.tiles_x * p.tiles_y);
    const int rem = pt - ig * (p.tiles_x * p.tiles_y);
    const int ty = fdiv(rem, p.mTX), tx = rem - ty * p.tiles_x;
    n0l = ig * NI;
    n0s = n0l;
    const int oy = ty * TH + dpy, ox = tx * TW + dpx, n = n0l + dimg;
    dok = (n < p.N) & (oy < p.Ho) & (ox < p.Wo);
    dsrc = p.dy + ((long)(dok ? n : 0) * p.Cout + m0 + co_d0) * planeO + (dok ? (long)oy * p.Wo + ox : 0);
    back = (ty * TH) * p.Ws + tx * TW;
    xt = xsrc + (long)n0l * xbs + back;
    gy0 = ty * TH - p.pad_lo; gx0 = tx * TW - p.pad_lo;
    xvalid = 0;
  };
  auto load_dy = [&](int i) __attribute__((always_inline)) {
    const float v = dsrc[(long)(4 * i) * planeO];
    dyr[i] = dok ? v : 0.f;
  };
  auto load_x = [&](int j) __attribute__((always_inline)) {
    const int lx = epk[j] & 255, ly = (epk[j] >> 8) & 255, img = (epk[j] >> 16) & 255, c = (epk[j] >> 24) & 127;
    const int gy = gy0 + ly, gx = gx0 + lx;
    const bool ok = (epk[j] < 0) & (n0l + img < p.N) & ((unsigned)gy < (unsigned)p.Hi) & ((unsigned)gx < (unsigned)p.Wi);
    xr[j] = xt[ok ? eoff[j] : -back];
    xvalid |= ok ? 1u << j : 0u;
  };
  auto stash_dy = [&](float* buf, int i) __attribute__((always_inline)) { buf[pp_d * DLD + co_d0 + 4 * i] = dyr[i]; };
  auto stash_x = [&](float* buf, int j) __attribute__((always_inline)) {
    const int lx = epk[j] & 255, ly = (epk[j] >> 8) & 255, img = (epk[j] >> 16) & 255, c = (epk[j] >> 24) & 127;
    const bool ok = (xvalid >> j) & 1u;
    float v = xr[j];
    if (p.gn_scale) {
      const long gi = ok ? (long)(n0s + img) * Ct + c0 + c : 0;
      v = v * p.gn_scale[gi] + p.gn_shift[gi];
    }
    const float sv = silu_g(v);
    v = p.act ? sv : v;
    float* dst = epk[j] < 0 ? buf + 64 * DLD + c * PS + img * IHW + ly * IW + lx
                            : smem + 2 * PBUF + tid;                       // disabled elements: private dummy word
    *dst = ok ? v : 0.f;
  };

  const int t_begin = sp * p.tiles_per_block;
  int t_end = t_begin + p.tiles_per_block;
  if (t_end > p.n_ptiles) t_end = p.n_ptiles;
  if (t_begin < t_end) {
    // prologue: tile t_begin -> buffer 0, tile t_begin + 1 -> registers
    load_begin(t_begin);
    ADM_UNROLL
    for (int i = 0; i < 32; ++i) load_dy(i);
    ADM_UNROLL
    for (int j = 0; j < NPX; ++j) load_x(j);
    ADM_UNROLL
    for (int i = 0; i < 32; ++i) stash_dy(smem, i);
    ADM_UNROLL
    for (int j = 0; j < NPX; ++j) stash_x(smem, j);
    if (t_begin + 1 < t_end) {
      load_begin(t_begin + 1);
      ADM_UNROLL
      for (int i = 0; i < 32; ++i) load_dy(i);
      ADM_UNROLL
      for (int j = 0; j < NPX; ++j) load_x(j);
    }
    __syncthreads();
  }
  for (int pt = t_begin; pt < t_end; ++pt) {
    const float* cur = smem + ((pt - t_begin) & 1) * PBUF;
    float* nxt = smem + (((pt - t_begin) & 1) ^ 1) * PBUF;
    const float* ldsD = cur;
    const float* ldsP = cur + 64 * DLD;
    const bool has1 = pt + 1 < t_end, has2 = pt + 2 < t_end;     // wave-uniform
    const int dlane = h * DLD + wave * 32 + l31, plane = l31 * PS + h;
    auto fetch = [&](int s, float& av, float (&bv)[NT]) __attribute__((always_inline)) {
      // pixel pp = 2 s + h: px = ((2 s) & 15) + h, py = s >> 3  ->  lane term + compile-time offset
      av = ldsD[dlane + s * (2 * DLD)];
      ADM_UNROLL
      for (int t = 0; t < NT; ++t) bv[t] = ldsP[plane + (s >> 3) * IW + ((2 * s) & 15) + (t / 3) * IW + (t % 3)];
    };
    float a0, a1, b0[NT], b1[NT];
    fetch(0, a0, b0);
    ADM_UNROLL
    for (int g = 0; g < 16; ++g) {
      const int s = 2 * g;
      fetch(s + 1, a1, b1);
      ADM_SCHED_FENCE();
      ADM_UNROLL
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[t], acc[t], 0, 0, 0);
      if (g < 15) fetch(s + 2, a0, b0);
      ADM_SCHED_FENCE();
      ADM_UNROLL
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[t], acc[t], 0, 0, 0);
      // ---- side work of this group, issued while the 18 MFMAs above run -------------------------------------------
      if (g < 8) {
        if (has1) {
          ADM_UNROLL
          for (int i = 4 * g; i < 4 * g + 4; ++i) stash_dy(nxt, i);
          stash_x(nxt, 2 * g);
          stash_x(nxt, 2 * g + 1);
        }
      } else {
        if (has2) {
          if (g == 8) load_begin(pt + 2);
          ADM_UNROLL
          for (int i = 4 * (g - 8); i < 4 * (g - 8) + 4; ++i) load_dy(i);
          load_x(2 * (g - 8));
          load_x(2 * (g - 8) + 1);
        }
      }
      ADM_SCHED_FENCE();
    }
    __syncthreads();
  }
  float* out = p.part + (long)sp * p.Cout * Ct * 9;
  ADM_UNROLL
  for (int t = 0; t < NT; ++t) {
    const int cc = c0 + l31;
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int co = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      out[((long)t * p.Cout + co) * Ct + cc] = acc[t][r];      // slab layout [tap][cout][cin]
    }
  }
}

// Eight-wave form of the kernel above: TWO waves per SIMD. Waves w and w + 4 land on the same SIMD, own the same 32 cout
// rows and split the nine taps 5 / 4 (80 / 64 accumulator registers instead of 144), so while one of them walks through its
// staging piece the other one's MFMAs keep the matrix pipe busy — the overlap a single in-order wave per SIMD cannot have.
// Staging per thread halves (512 threads): 16 dy words and 7 patch elements per tile.
template <int T0, int NT>
__device__ __forceinline__ void wgrad_sp8_body(const WgradParams& p, float* smem) {
  // fixed tile geometry (launcher: Wo % 16 == 0 handled by TW = 16, TH = 4, one image per 64-pixel tile): every LDS address
  // of the MFMA loop is then a lane term plus a compile-time offset — with runtime geometry the fully unrolled loop kept
  // ~64 hoisted address registers alive and spilled
  constexpr int CB = 32, NPX = 7, DLD = 129, NDY = 16;
  constexpr int TW = 16, TH = 4, NI = 1, IW = 18, IHW = 108, PE = 108, PS = 109;
  constexpr int PBUF = 64 * DLD + CB * PS;        // floats per LDS buffer: dy tile + patch
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3;
  const int l31 = lane & 31, h = lane >> 5;
  int b = blockIdx.x;
  const int sp = b % p.split; b /= p.split;
  const int chunk = b % p.n_chunks, ct = b / p.n_chunks;
  const int m0 = ct * 128, c0 = chunk * CB;
  const int Ct = p.C1 + p.C2;
  const int planeS = p.Hs * p.Ws;
  const long planeO = (long)p.Ho * p.Wo;
  const int n_el = CB * PE;

  f32x16 acc[NT];
  ADM_UNROLL
  for (int t = 0; t < NT; ++t)
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // tile-invariant roles
  const int pp_d = tid & 63, co_d0 = tid >> 6;
  const int dpx = pp_d & (TW - 1), dpy = pp_d >> 4, dimg = 0;
  const float* xsrc = c0 < p.C1 ? p.x1 + (long)c0 * planeS : p.x2 + (long)(c0 - p.C1) * planeS;
  const long xbs = c0 < p.C1 ? p.x1_bs : p.x2_bs;
  int eoff[NPX], epk[NPX];
  ADM_UNROLL
  for (int j = 0; j < NPX; ++j) {
    const int e = tid + 512 * j;
    const bool en = e < n_el;
    const int ec = en ? e : 0;
    const int c = ec / PE, q = ec - c * PE;
    const int img = 0, r2 = q;
    const int ly = r2 / IW, lx = r2 - ly * IW;
    eoff[j] = (int)(img * xbs) + c * planeS + (ly - p.pad_lo) * p.Ws + (lx - p.pad_lo);
    epk[j] = lx | (ly << 8) | (img << 16) | (c << 24) | (en ? (int)0x80000000 : 0);
  }
  float dyr[NDY], xr[NPX];
  int n0s = 0;             // first image of the tile held in xr (GroupNorm rows are read at stash time: the latency hides
                           // behind the running MFMA group, and 32 prefetch registers fewer keep the kernel spill-free)
  unsigned xvalid = 0;
  // tile-uniform state of the tile being loaded
  const float *xt = xsrc, *dsrc = p.dy;
  int gy0 = 0, gx0 = 0, n0l = 0, back = 0;
  bool dok = false;
  auto load_begin = [&](int pt) __attribute__((always_inline)) {
    const int ig = p.n_ptiles < 65536 ? fdiv(pt, p.mTXY) : pt / (p.tiles_x * p.tiles_y);
    const int rem = pt - ig * (p.tiles_x * p.tiles_y);
    const int ty = fdiv(rem, p.mTX), tx = rem - ty * p.tiles_x;
    n0l = ig * NI;
    n0s = n0l;
    const int oy = ty * TH + dpy, ox = tx * TW + dpx, n = n0l + dimg;
    dok = (n < p.N) & (oy < p.Ho) & (ox < p.Wo);
    dsrc = p.dy + ((long)(dok ? n : 0) * p.Cout + m0 + co_d0) * planeO + (dok ? (long)oy * p.Wo + ox : 0);   // co_d0 = tid >> 6 in 0..7
    back = (ty * TH) * p.Ws + tx * TW;
    xt = xsrc + (long)n0l * xbs + back;
    gy0 = ty * TH - p.pad_lo; gx0 = tx * TW - p.pad_lo;
    xvalid = 0;
  };
  auto load_dy = [&](int i) __attribute__((always_inline)) {
    const float v = dsrc[(long)(8 * i) * planeO];
    dyr[i] = dok ? v : 0.f;
  };
  auto load_x = [&](int j) __attribute__((always_inline)) {
    const int lx = epk[j] & 255, ly = (epk[j] >> 8) & 255, img = (epk[j] >> 16) & 255, c = (epk[j] >> 24) & 127;
    const int gy = gy0 + ly, gx = gx0 + lx;
    const bool ok = (epk[j] < 0) & (n0l + img < p.N) & ((unsigned)gy < (unsigned)p.Hi) & ((unsigned)gx < (unsigned)p.Wi);
    xr[j] = xt[ok ? eoff[j] : -back];
    xvalid |= ok ? 1u << j : 0u;
  };
  auto stash_dy = [&](float* buf, int i) __attribute__((always_inline)) { buf[pp_d * DLD + co_d0 + 8 * i] = dyr[i]; };
  auto stash_x = [&](float* buf, int j) __attribute__((always_inline)) {
    const int lx = epk[j] & 255, ly = (epk[j] >> 8) & 255, img = (epk[j] >> 16) & 255, c = (epk[j] >> 24) & 127;
    const bool ok = (xvalid >> j) & 1u;
    float v = xr[j];
    if (p.gn_scale) {
      const long gi = ok ? (long)(n0s + img) * Ct + c0 + c : 0;
      v = v * p.gn_scale[gi] + p.gn_shift[gi];
    }
    const float sv = silu_g(v);
    v = p.act ? sv : v;
    float* dst = epk[j] < 0 ? buf + 64 * DLD + c * PS + img * IHW + ly * IW + lx
                            : smem + 2 * PBUF + tid;                       // disabled elements: private dummy word (512)
    *dst = ok ? v : 0.f;
  };

  const int t_begin = sp * p.tiles_per_block;
  int t_end = t_begin + p.tiles_per_block;
  if (t_end > p.n_ptiles) t_end = p.n_ptiles;
  if (t_begin < t_end) {
    // prologue: tile t_begin -> buffer 0, tile t_begin + 1 -> registers
    load_begin(t_begin);
    ADM_UNROLL
    for (int i = 0; i < NDY; ++i) load_dy(i);
    ADM_UNROLL
    for (int j = 0; j < NPX; ++j) load_x(j);
    ADM_UNROLL
    for (int i = 0; i < NDY; ++i) stash_dy(smem, i);
    ADM_UNROLL
    for (int j = 0; j < NPX; ++j) stash_x(smem, j);
    if (t_begin + 1 < t_end) {
      load_begin(t_begin + 1);
      ADM_UNROLL
      for (int i = 0; i < NDY; ++i) load_dy(i);
      ADM_UNROLL
      for (int j = 0; j < NPX; ++j) load_x(j);
    }
    __syncthreads();
  }
  for (int pt = t_begin; pt < t_end; ++pt) {
    const float* cur = smem + ((pt - t_begin) & 1) * PBUF;
    float* nxt = smem + (((pt - t_begin) & 1) ^ 1) * PBUF;
    const float* ldsD = cur;
    const float* ldsP = cur + 64 * DLD;
    const bool has1 = pt + 1 < t_end, has2 = pt + 2 < t_end;     // wave-uniform
    const int dlane = h * DLD + wave * 32 + l31, plane = l31 * PS + h;
    auto fetch = [&](int s, float& av, float (&bv)[NT]) __attribute__((always_inline)) {
      // pixel pp = 2 s + h: px = ((2 s) & 15) + h, py = s >> 3  ->  lane term + compile-time offset
      av = ldsD[dlane + s * (2 * DLD)];
      ADM_UNROLL
      for (int t = 0; t < NT; ++t) bv[t] = ldsP[plane + (s >> 3) * IW + ((2 * s) & 15) + ((T0 + t) / 3) * IW + ((T0 + t) % 3)];
    };
    float a0, a1, b0[NT], b1[NT];
    fetch(0, a0, b0);
    ADM_UNROLL
    for (int g = 0; g < 16; ++g) {
      const int s = 2 * g;
      fetch(s + 1, a1, b1);
      ADM_SCHED_FENCE();
      ADM_UNROLL
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[t], acc[t], 0, 0, 0);
      if (g < 15) fetch(s + 2, a0, b0);
      ADM_SCHED_FENCE();
      ADM_UNROLL
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[t], acc[t], 0, 0, 0);
      // ---- side work of this group; the partner wave on this SIMD issues MFMAs meanwhile -------------------------------------------
      if (g < 8) {
        if (has1) {
          stash_dy(nxt, 2 * g);
          stash_dy(nxt, 2 * g + 1);
          if (g < NPX) stash_x(nxt, g);
        }
      } else {
        if (has2) {
          if (g == 8) load_begin(pt + 2);
          load_dy(2 * (g - 8));
          load_dy(2 * (g - 8) + 1);
          if (g - 8 < NPX) load_x(g - 8);
        }
      }
      ADM_SCHED_FENCE();
    }
    __syncthreads();
  }
  float* out = p.part + (long)sp * p.Cout * Ct * 9;
  ADM_UNROLL
  for (int t = 0; t < NT; ++t) {
    const int cc = c0 + l31;
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int co = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      out[((long)(T0 + t) * p.Cout + co) * Ct + cc] = acc[t][r];      // slab layout [tap][cout][cin]
    }
  }
}


__global__ void __launch_bounds__(512, 1) conv_wgrad_sp8_kernel(const WgradParams p) {
  ADM_DYN_SMEM(float, smem);
  if ((threadIdx.x >> 8) == 0) wgrad_sp8_body<0, 5>(p, smem);   // waves 0..3: taps 0..4
  else wgrad_sp8_body<5, 4>(p, smem);                           // waves 4..7: taps 5..8
}

// dW (=|+=) sum_k part[k]. The slabs are [tap][cout][cin] — a wave's 32 consecutive cins are one 128-byte run, so the kernels'
// partial-sum epilogue touches two cache lines per store instruction; in the final (cout, cin, tap) order the same store was 32
// words 36 bytes apart, ~18 lines, and on the small-spatial / wide layers (two 8x8 tiles per workgroup) the epilogue cost twice
// the MFMAs (ADM_WGRAD_PROF: 72k cycles). The transposition to (cout, cin, tap) happens here, once instead of `split` times.
// One workgroup per 256 consecutive slab elements: lane group e = tid & 63 owns one float4 column, wave g = tid >> 6 sums the
// slabs k = g, g + 4, ... (every wave-load is one contiguous KiB; four independent chains per thread keep four loads in
// flight), the four partial sums meet in LDS in a fixed order.
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ part, int split, long numel,
                                                           float* dW, int accumulate, int taps) {
  __shared__ float4 red[3][64];
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
  // float4 loads need numel % 4 == 0 (slab k starts at k * numel) and an aligned workspace; dW is a slice of the flat gradient
  // buffer at an arbitrary float offset: vector stores only when it happens to be aligned (and the layout is the same: taps == 1)
  const bool vec = (numel & 3) == 0 && (reinterpret_cast<uintptr_t>(part) & 15) == 0;
  const bool dvec = taps == 1 && (reinterpret_cast<uintptr_t>(dW) & 15) == 0;
  const long M = numel / taps;                                  // cout * cin
  for (long base = (long)blockIdx.x * 256; base < numel; base += (long)gridDim.x * 256) {
    const long i = base + 4 * e;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    if (vec && i + 3 < numel) {
      int k = g;
      for (; k + 12 < split; k += 16) {
        const float4 v0 = *reinterpret_cast<const float4*>(part + (long)k * numel + i);
        const float4 v1 = *reinterpret_cast<const float4*>(part + (long)(k + 4) * numel + i);
        const float4 v2 = *reinterpret_cast<const float4*>(part + (long)(k + 8) * numel + i);
        const float4 v3 = *reinterpret_cast<const float4*>(part + (long)(k + 12) * numel + i);
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
        a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
        a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
      }
      for (; k < split; k += 4) {
        const float4 v0 = *reinterpret_cast<const float4*>(part + (long)k * numel + i);
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      }
    } else if (i < numel) {           // ragged tail / unaligned workspace (never for the conv shapes; kept for generality)
      for (int k = g; k < split; k += 4)
        for (int j = 0; j < 4 && i + j < numel; ++j) (&a0.x)[j] += part[(long)k * numel + i + j];
    }
    float4 s = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                           (a0.w + a1.w) + (a2.w + a3.w));
    if (g > 0) red[g - 1][e] = s;
    __syncthreads();
    if (g == 0 && i < numel) {
      const float4 r0 = red[0][e], r1 = red[1][e], r2 = red[2][e];
      s.x = (s.x + r0.x) + (r1.x + r2.x); s.y = (s.y + r0.y) + (r1.y + r2.y);
      s.z = (s.z + r0.z) + (r1.z + r2.z); s.w = (s.w + r0.w) + (r1.w + r2.w);
      if (dvec && i + 3 < numel) {
        if (accumulate) { const float4 o = *reinterpret_cast<const float4*>(dW + i); s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
        *reinterpret_cast<float4*>(dW + i) = s;
      } else {
        for (int j = 0; j < 4 && i + j < numel; ++j) {
          const long sj = i + j, o = taps == 1 ? sj : (sj % M) * taps + sj / M;     // slab [tap][cout*cin] -> (cout, cin, tap)
          dW[o] = (accumulate ? dW[o] : 0.f) + (&s.x)[j];
        }
      }
    }
    __syncthreads();
  }
}

// The same for 3x3 weights with both sides coalesced: a workgroup owns 64 consecutive (cout, cin) pairs m — in every slab one
// contiguous 256-byte run per tap, in dW the 576 contiguous floats m * 9 + tap. Lane m of wave g sums its nine taps over the slabs
// k = g, g + 4, ... (nine independent chains), the four waves' sums meet in LDS in a fixed order and dW is written in order. (The generic
// kernel above writes dW[m * 9 + tap] straight from the slab order: 64 cache lines per store instruction.)
__global__ void __launch_bounds__(256) wgrad_reduce9_kernel(const float* __restrict__ part, int split, long M,
                                                            float* dW, int accumulate) {
  __shared__ float tile[4][64 * 9];
  const long numel = 9 * M;
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;         // pair m0 + e, slabs k = g, g + 4, ...
  for (long m0 = (long)blockIdx.x * 64; m0 < M; m0 += (long)gridDim.x * 64) {
    const long m = m0 + e;
    float a[9];
    ADM_UNROLL
    for (int t = 0; t < 9; ++t) a[t] = 0.f;
    if (m < M) {
      for (int k = g; k < split; k += 4) {
        const float* src = part + (long)k * numel + m;
        ADM_UNROLL
        for (int t = 0; t < 9; ++t) a[t] += src[(long)t * M];
      }
    }
    ADM_UNROLL
    for (int t = 0; t < 9; ++t) tile[g][e * 9 + t] = a[t];
    __syncthreads();
    const long o0 = m0 * 9;
    const int cnt = (int)(M - m0 < 64 ? M - m0 : 64) * 9;
    for (int i = threadIdx.x; i < cnt; i += 256) {
      const float v = (tile[0][i] + tile[1][i]) + (tile[2][i] + tile[3][i]);
      dW[o0 + i] = (accumulate ? dW[o0 + i] : 0.f) + v;
    }
    __syncthreads();
  }
}

static inline int ilog2w(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

int launch_wgrad_reduce(const float* workspace, int split, long numel, float* dW, int accumulate, int taps, hipStream_t st) {
  if (taps == 9) {
    const long M = numel / 9;
    long g = (M + 63) / 64;
    if (g > 16384) g = 16384;
    ADM_LAUNCH(wgrad_reduce9_kernel, dim3((unsigned)g), dim3(256), 0, st, workspace, split, M, dW, accumulate);
    return ADM_CHECK_LAUNCH();
  }
  long g = (numel + 255) / 256;
  if (g > 4096) g = 4096;
  ADM_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)g), dim3(256), 0, st, workspace, split, numel, dW, accumulate, taps);
  return ADM_CHECK_LAUNCH();
}

// workspace floats needed by launch_conv_wgrad for this shape
static int g_wgrad_max_split = 0;   // 0 = no cap; adm_set_option("wgrad_max_split", n) caps the split-K factor (tests use it
void set_wgrad_max_split(int v) { g_wgrad_max_split = v; }   // to put several pixel tiles on one workgroup)

long conv_wgrad_workspace(const adm_conv_args& a, int* split_out) {
  const int C2 = a.x2 ? a.C2 : 0, Ct = a.C1 + C2;
  int Ho, Wo;
  conv_out_dims(a.H, a.W, a.up, a.stride, a.ks, a.pad_lo, &Ho, &Wo);
  const int TW = Wo >= 16 ? 16 : Wo, TH = Ho >= 4 ? 4 : Ho;
  const int NI = 64 / (TW * TH);
  const int n_ptiles = ceil_div(Wo, TW) * ceil_div(Ho, TH) * ceil_div(a.N, NI);
  const int CB = a.ks == 3 ? 32 : 128;
  const int pairs = ceil_div(a.Cout, 128) * ceil_div(Ct, CB);
  int split = ceil_div(768, pairs);
  if (g_wgrad_max_split > 0 && split > g_wgrad_max_split) split = g_wgrad_max_split;
  if (split > n_ptiles) split = n_ptiles;
  if (split < 1) split = 1;
  if (split_out) *split_out = split;
  return (long)split * a.Cout * Ct * a.ks * a.ks;
}

int launch_conv_wgrad(const adm_conv_args& a, const float* dy, float* dW, int accumulate, float* workspace,
                      hipStream_t st) {
  ADM_REQUIRE(a.ks == 3 || a.ks == 1, "conv_wgrad: ks must be 1 or 3");
  ADM_REQUIRE(a.stride == 1 || (a.stride == 2 && a.ks == 3), "conv_wgrad: stride 2 only for 3x3");
  WgradParams p;
  const int C2 = a.x2 ? a.C2 : 0, Ct = a.C1 + C2;
  p.x1 = a.x1; p.x2 = a.x2; p.C1 = a.C1; p.C2 = C2; p.dy = dy; p.Cout = a.Cout;
  p.N = a.N; p.Hs = a.H; p.Ws = a.W;
  p.Hi = a.up ? 2 * a.H : a.H; p.Wi = a.up ? 2 * a.W : a.W;
  conv_out_dims(a.H, a.W, a.up, a.stride, a.ks, a.pad_lo, &p.Ho, &p.Wo);
  p.up = a.up; p.pad_lo = a.ks == 1 ? 0 : a.pad_lo;
  p.gn_scale = a.gn_scale; p.gn_shift = a.gn_shift; p.act = a.act;
  p.part = workspace; p.prof = nullptr;
  const int TW = p.Wo >= 16 ? 16 : p.Wo, TH = p.Ho >= 4 ? 4 : p.Ho;
  ADM_REQUIRE((TW & (TW - 1)) == 0 && (TH & (TH - 1)) == 0, "conv_wgrad: small output dims must be powers of two");
  p.lTW = ilog2w(TW); p.lTH = ilog2w(TH);
  const int NI = 64 / (TW * TH);
  p.tiles_x = ceil_div(p.Wo, TW); p.tiles_y = ceil_div(p.Ho, TH);
  p.n_ptiles = p.tiles_x * p.tiles_y * ceil_div(a.N, NI);
  p.IH = (TH - 1) * a.stride + a.ks; p.IW = (TW - 1) * a.stride + a.ks;
  p.PS = (NI * p.IH * p.IW) | 1;
  const int CB = a.ks == 3 ? 32 : 128;
  p.n_ct = ceil_div(a.Cout, 128); p.n_chunks = ceil_div(Ct, CB);
  conv_wgrad_workspace(a, &p.split);
  p.tiles_per_block = ceil_div(p.n_ptiles, p.split);
  p.split = ceil_div(p.n_ptiles, p.tiles_per_block);  // no empty workgroups
  p.x1_bs = a.x1_bstride ? a.x1_bstride : (long)a.C1 * a.H * a.W;
  p.x2_bs = a.x2_bstride ? a.x2_bstride : (long)C2 * a.H * a.W;
  auto magic = [](long d) { return d <= 1 ? 0u : (unsigned)((1ULL << 32) / (unsigned long long)d + 1ULL); };
  p.mPE = magic((long)NI * p.IH * p.IW); p.mIHW = magic((long)p.IH * p.IW); p.mIW = magic(p.IW);
  p.mTX = magic(p.tiles_x); p.mTXY = magic((long)p.tiles_x * p.tiles_y);
  const size_t smem = sizeof(float) * ((size_t)64 * 129 + (size_t)CB * p.PS);
  // 80 KiB covers every layer whose output is at least 4x4; the stride-2 convolutions that end at 2x2 / 1x1 (the deepest
  // Downsample2D of a 64x64 / 32x32 model) stage 16 / 64 images' 5x5 / 3x3 patches per tile: up to 107 KiB, generic kernel only
  ADM_REQUIRE(smem <= 128 * 1024, "conv_wgrad: patch too large for LDS");
  const long numel = (long)a.Cout * Ct * a.ks * a.ks;
  dim3 grid(p.n_ct * p.n_chunks * p.split), block(256);
#if !defined(ADM_EMU)
  static bool once = [] {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgrad_pf_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgrad_pf_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgrad_pf_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgrad_pf_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    return true;
  }();
  (void)once;
#endif
  static const int use_pf = [] { const char* e = getenv("ADM_WGRAD_PF"); return e ? atoi(e) : 1; }();
  const int PE = NI * p.IH * p.IW;
  if (conv_bf16_mode() >= 2 && conv1x1_wgrad_bf16_eligible(a)) {   // level 2: 1x1 weight gradient on bf16 operands
    const int slabs = launch_conv1x1_wgrad_bf16(a, dy, workspace, p.split, st);
    ADM_REQUIRE(slabs > 0 && slabs <= p.split, "conv1x1_wgrad_bf16: launch failed");
    return launch_wgrad_reduce(workspace, slabs, numel, dW, accumulate, a.ks * a.ks, st);
  }
  if (conv_bf16_enabled() && conv_wgrad_bf16_eligible(a)) {   // mixed precision: bf16 operands, fp32 partial sums
    ADM_TRY(launch_conv_wgrad_bf16(a, dy, dW, accumulate, workspace, p.split, st));
    return launch_wgrad_reduce(workspace, p.split, numel, dW, accumulate, a.ks * a.ks, st);
  }
  // tile-invariant prefetch path: no upsample fold, every channel chunk inside one source tensor, full cout tiles
  const bool fast = a.up == 0 && a.C1 % CB == 0 && Ct % CB == 0 && a.Cout % 128 == 0;
  static const int use_sp = [] { const char* e = getenv("ADM_WGRAD_SP"); return e ? atoi(e) : 1; }();
  if (use_pf && use_sp && fast && a.stride == 1 && a.ks == 3 && TW == 16 && TH == 4) {
    const size_t smem_sp = sizeof(float) * (2 * ((size_t)64 * 129 + (size_t)CB * 109) + 256);
#if !defined(ADM_EMU)
    static bool once_sp = [] {
      (void)hipFuncSetAttribute((const void*)conv_wgrad_sp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
      return true;
    }();
    (void)once_sp;
#endif
    static const int use_sp8 = [] { const char* e = getenv("ADM_WGRAD_SP8"); return e ? atoi(e) : 1; }();
    if (use_sp8) {
#if !defined(ADM_EMU)
      static bool once_sp8 = [] {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_sp8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
        return true;
      }();
      (void)once_sp8;
#endif
      ADM_LAUNCH(conv_wgrad_sp8_kernel, grid, dim3(512), smem_sp + sizeof(float) * 256, st, p);
    } else {
      ADM_LAUNCH(conv_wgrad_sp_kernel, grid, block, smem_sp, st, p);
    }
  } else if (use_pf && a.stride == 1 && a.ks == 3 && PE <= 128) {
    if (fast) ADM_LAUNCH((conv_wgrad_pf_kernel<3, true>), grid, block, smem, st, p);
    else ADM_LAUNCH((conv_wgrad_pf_kernel<3, false>), grid, block, smem, st, p);
  } else if (use_pf && a.ks == 1) {
    if (fast) ADM_LAUNCH((conv_wgrad_pf_kernel<1, true>), grid, block, smem, st, p);
    else ADM_LAUNCH((conv_wgrad_pf_kernel<1, false>), grid, block, smem, st, p);
  } else if (a.ks == 3 && a.stride == 1) {
    ADM_LAUNCH((conv_wgrad_kernel<3, 1>), grid, block, smem, st, p);
  } else if (a.ks == 3) {
#if !defined(ADM_EMU)
    static const bool want_prof = getenv("ADM_WGRAD_PROF") != nullptr;
    if (want_prof) {        // developer aid: per-phase cycle accounting of the generic stride-2 kernel, printed after the launch
      static unsigned long long* dprof = [] { void* q = nullptr; (void)hipMalloc(&q, 8 * sizeof(unsigned long long)); return (unsigned long long*)q; }();
      (void)hipMemsetAsync(dprof, 0, 8 * sizeof(unsigned long long), st);
      p.prof = dprof;
      (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<3, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      ADM_LAUNCH((conv_wgrad_kernel<3, 2, true>), grid, block, smem, st, p);
      unsigned long long h[8];
      (void)hipMemcpyAsync(h, dprof, sizeof(h), hipMemcpyDeviceToHost, st);
      (void)hipStreamSynchronize(st);
      const double w = 4.0 * grid.x, nt = (double)p.tiles_per_block;
      fprintf(stderr, "[wgrad s2 prof] %d->%d out %dx%d N=%d split=%d (%d workgroups, %.0f tiles each): per wave: total %.0f | per tile: "
              "loop head %.0f dy %.0f patch %.0f barrier %.0f mfma %.0f barrier %.0f | epilogue %.0f cycles (288 MFMAs = 18432)\n", Ct, a.Cout,
              p.Ho, p.Wo, a.N, p.split, (int)grid.x, nt, h[7] / w, h[0] / w / nt, h[1] / w / nt, h[2] / w / nt, h[3] / w / nt, h[4] / w / nt,
              h[5] / w / nt, h[6] / w);
    } else
#endif
    ADM_LAUNCH((conv_wgrad_kernel<3, 2>), grid, block, smem, st, p);
  } else {
    ADM_LAUNCH((conv_wgrad_kernel<1, 1>), grid, block, smem, st, p);
  }
  return launch_wgrad_reduce(workspace, p.split, numel, dW, accumulate, a.ks * a.ks, st);
}

}  // namespace adm
