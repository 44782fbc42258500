// k_conv_wino_f4.hip — conv_wino6_kernel: Winograd F(4x4,3x3) of the fused 3x3 stride-1 convolution (fp32 throughout), the headline kernel.
#include <type_traits>

#include "k_conv_wino.h"

namespace adm {

// =====================================================================================================================
// v6 (round 5) — Winograd F(4x4,3x3) on v5's skeleton. Round 5's accounting of v4 / v5 (profiles/r05_wino.md): 70 % of the kernel's cycles are
// MFMA cycles, vector instructions add to them one for one, and three different schedules of the same arithmetic land within 2 % of each
// other — what is left to cut is the MFMA count itself. F(4x4,3x3) multiplies 36 Winograd points per 16 outputs instead of 16 per 4:
// 1.78x fewer MFMAs than F(2x2,3x3), 4x fewer than the direct convolution; its fp32 error is 0.6-1.7e-5 of max|out| (rms 1-4e-6; `profiles/r05_accuracy.md`) on this network's layer
// shapes (F(2x2): 0.7-1.5e-6; the per-layer bar is 1e-4).
//   Y = A^T [ (G g G^T) . (B^T d B) ] A,  d = 6x6 input window, Y = 4x4 outputs, the standard matrices of Lavin & Gray.
// Workgroup tile = 128 couts x 16x16 pixels = 16 Winograd tiles; 8 waves, wave w owns couts 16 w .. 16 w + 15 x all 16 tiles x all 36 points
// (144 accumulators on v_mfma_f32_16x16x4_f32; the inverse transform is lane-local). Per 8-channel chunk and wave: 72 MFMAs (v5: 64 for HALF
// the pixels). Filters: their own image [chunk][cout block][k step 2][point group 9][lane 64][4 points], streamed L2 -> registers through a
// ring of six point groups; B operands: V slab [point 36][channel 8][tile 16] in LDS, one ds_read2_b32 per pair of points. Staging per PAIR of
// chunks, shared by all 512 threads: 1152 float4 row pieces + 576 halo elements (raw buffer loads one interval ahead -> GroupNorm affine +
// SiLU -> 18x18 patch per channel), 256 (channel, tile) windows transformed by two threads each (V rows 0-2 / 3-5: 72 VALU per thread).
// Rings of four V slabs / patch buffers, one workgroup barrier per pair of chunks — v5's protocol.
// Schedule (measured step by step, profiles/r05_wino.md §3): the two waves of a SIMD run an interval in antiphase ([MFMA block][staging] /
// [staging][MFMA block], one loop body); a staging block is B (the activations fetched an interval ago -> patch slab), the epilogue of a
// finished tile (its residual rows and bias fetched one stage B ahead), A (the next pair's loads), C (window transform): A in front of C
// because vmcnt retires in order — the MFMA block's first counted wait for a filter group also waits for every older load.
// Developer macros (timing / accounting builds, never the product): W6X_PROF (s_memtime accounting of waves W6X_PROFW / 64 and + 4),
// W6X_NO{A,B,C,EPI,RES,STATS,FILT,LDS,PRIO} (stage ablations), W6X_SWAP / W6X_ALLX / W6X_ALLY (roles), W6X_RING_IN_P, W6X_BFENCE / W6X_EFENCE.
constexpr int W6PP = 20;                           // patch row pitch (18 columns: left halo, 16 pixels, right halo)
constexpr int W6CS = 18 * W6PP;                    // 360 floats per channel
constexpr int W6PSLAB = WCK * W6CS + 512;          // + one dummy word per thread
constexpr int W6VSLAB = 36 * WCK * 16;             // 4608 floats
constexpr int W6LDS = 4 * (W6VSLAB + W6PSLAB);     // 32000 floats = 125 KiB
constexpr int W6AR = 6;                            // filter ring: point groups in flight (18 per chunk = 3 turns of the ring)

// 1D input transform B^T (6 x 6) on (d0 .. d5) -> (v0 .. v5): 12 operations
#define W6_BT(d0, d1, d2, d3, d4, d5, v0, v1, v2, v3, v4, v5)                  \
  do {                                                                          \
    const float a_ = fmaf(-4.f, d2, d4), b_ = fmaf(-4.f, d1, d3);               \
    const float c_ = d4 - d2, e_ = d3 - d1;                                     \
    v0 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));                                     \
    v1 = a_ + b_; v2 = a_ - b_;                                                 \
    v3 = fmaf(2.f, e_, c_); v4 = fmaf(-2.f, e_, c_);                            \
    v5 = fmaf(4.f, d1, fmaf(-5.f, d3, d5));                                     \
  } while (0)
// 1D inverse transform A^T (4 x 6) on (m0 .. m5) -> (y0 .. y3): 10 operations
#define W6_AT(m0, m1, m2, m3, m4, m5, y0, y1, y2, y3)                           \
  do {                                                                          \
    const float s1_ = m1 + m2, d1_ = m1 - m2, s2_ = m3 + m4, d2_ = m3 - m4;     \
    y0 = (m0 + s1_) + s2_;                                                      \
    y1 = fmaf(2.f, d2_, d1_);                                                   \
    y2 = fmaf(4.f, s2_, s1_);                                                   \
    y3 = fmaf(8.f, d2_, d1_) + m5;                                              \
  } while (0)

// KIND (staging slots of this wave; four slots per thread and pair of chunks): non-UP 0 = waves 0-1 (float4, float4, float4, halo),
// 1 = wave 2 (float4, float4, halo, halo), 2 = waves 3-7 (float4, float4, halo, -); UP (source-resolution 10x10 patches, scalars only)
// 0 = wave 0 (four scalars), 2 = the others (three).
template <bool UP, int KIND, int ACT>
__device__ __forceinline__ void wino6_wave(const WinoParams& p, float* ldsV, float* ldsP, const int tid, const int wave,
                                           const int b0, const int bs) {
  const bool yrole = wave >= 4;
  constexpr int NS = (KIND == 2) ? 3 : 4;                       // active slots
  constexpr int NF = UP ? 0 : (KIND == 0 ? 3 : 2);              // of which float4 pieces (the first NF)
  const int lane = tid & 63;
  const int l15 = lane & 15, k4 = lane >> 4;
  const int Ct = p.C1 + p.C2;
  const int planeS = p.Hs * p.Ws;
  const int nch = Ct / WCK;
  const int n_cblk = p.Cout >> 4;
  const int ntile = (p.nblk - b0 + bs - 1) / bs;
  const int total = ntile * nch;
  const int npairs = total >> 1;
  // ---- staging items of this thread (NS per pair of chunks) ---------------------------------------------------------------------
  // per slot ONE register: channel within the pair (4 bits) | LDS offset << 4 (the patch row / column are recomputed per tile in a_geometry:
  // sixteen per-lane constants beside 144 accumulators were sixteen spilled registers)
  auto slot_item = [&](const int tid, int s, int& chrel, int& prow, int& col, int& pofs) {
    int c2 = 0, ch = 0;
    prow = 0; col = 0; pofs = WCK * W6CS + tid;                  // (default: the thread's dummy word of slab 0)
    if (UP) {
      const int e = 512 * s + tid;
      if (e < 1600) {
        c2 = e / 800; const int rem = e % 800;
        ch = rem / 100; prow = (rem % 100) / 10; col = rem % 10;
        pofs = c2 * W6PSLAB + rem;
      }
    } else if (s < NF) {
      const int f = 512 * s + tid;                               // float4 piece 0..1151
      const int row = f >> 2, q = f & 3;
      c2 = row / 144; const int rr = row % 144;
      ch = rr / 18; prow = rr % 18; col = 4 * q;
      pofs = c2 * W6PSLAB + ch * W6CS + prow * W6PP + 1 + 4 * q;
    } else if (s < NS) {
      const int h = (s == 2) ? tid - 128 : 384 + tid;            // halo element 0..575
      const int row = h >> 1, side = h & 1;
      c2 = row / 144; const int rr = row % 144;
      ch = rr / 18; prow = rr % 18; col = side ? 16 : -1;
      pofs = c2 * W6PSLAB + ch * W6CS + prow * W6PP + (side ? 17 : 0);
    }
    chrel = 8 * c2 + ch;
  };
  int it_pk[4];
  ADM_UNROLL
  for (int s = 0; s < 4; ++s) {
    int chrel, prow, col, pofs;
    slot_item(tid, s, chrel, prow, col, pofs);
    it_pk[s] = chrel | (pofs << 4);
  }
  // stage C: half of a (chunk of the pair, channel, tile) window transform
  // (the half is wave-uniform — waves 2k and 2k + 1 share 64 items — so that the two code paths below are scalar branches)
  const int c_item = (tid & 63) | ((wave >> 1) << 6), c_half = wave & 1;
  const int c_c2 = c_item >> 7, c_ch = (c_item >> 4) & 7, c_tile = c_item & 15;
  const int c_tyy = c_tile >> 2, c_txx = c_tile & 3;
  const int c_wbase = UP ? c_ch * 100 + (2 * c_tyy) * 10 + 2 * c_txx : c_ch * W6CS + (4 * c_tyy) * W6PP + 4 * c_txx;
  const int c_vofs = c_ch * 16 + c_tile;
  // ---- stage A cursor (one PAIR of chunks per step) -----------------------------------------------------------------------------------
  int a_v = b0, a_ci = -2, a_left = total + 2;   // (stage A advances BEFORE it loads: the first call lands on chunks 0, 1)
  const float *a_x1 = nullptr, *a_x2 = nullptr;
  int a_vo[4];
  unsigned a_ok = 0;
  int a_sg = 0;                                // element offset of the pair's first channel in the GroupNorm rows (sample included)
#if !defined(ADM_EMU)
  __amdgpu_buffer_rsrc_t a_rx1, a_rx2;
  const __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gn_scale), (short)0, 0x7fffffff, 0x00027000);
  const __amdgpu_buffer_rsrc_t h_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gn_shift), (short)0, 0x7fffffff, 0x00027000);
#endif
  int a_n = 0;
  auto a_geometry = [&]() {
    const Wino3Tile t = wino5_tile(p, a_v);
    a_n = t.n;
    a_x1 = p.x1 + (long)t.n * p.x1_bs;
    a_x2 = p.x2 + (long)t.n * p.x2_bs - (long)p.C1 * planeS;
    a_ok = 0;
    int tid_o = tid;                           // opaque: the items' rows / columns are RE-computed here — hoisted out of the main loop as
    ADM_OPAQUE_V(tid_o);                       // invariants they are twelve more registers carried through every block
    ADM_UNROLL
    for (int s = 0; s < NS; ++s) {
      int chrel, prow, col, pofs;
      slot_item(tid_o, s, chrel, prow, col, pofs);
      const int sy = UP ? t.ty * 8 - 1 + prow : t.ty * 16 - 1 + prow;
      const int sx = UP ? t.tx * 8 - 1 + col : t.tx * 16 + col;
      const bool ok = sy >= 0 && sy < p.Hs && sx >= 0 && sx < p.Ws;
      a_vo[s] = (chrel * planeS + (ok ? sy * p.Ws + sx : 0)) * 4;
      a_ok |= ok ? 1u << s : 0u;
    }
#if !defined(ADM_EMU)
    a_rx1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_x1), (short)0, 0x7fffffff, 0x00027000);
    a_rx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_x2), (short)0, 0x7fffffff, 0x00027000);
#endif
  };
  a_geometry();
  struct Raw { f32x4 v[3]; float h0, h1; unsigned ok; int sg; };      // v[s]: float4 slots; h0 / h1: the scalar slots behind them
  // (UP: slots 0-2 use v[s][0], slot 3 uses h1)
  float b_sc[4], b_sh[4];                      // GroupNorm scale / shift of the four slots' channels: fetched with the activations they belong to
  auto stage_a = [&](Raw& r) {                 // advance to the next pair of chunks (saturating), then its global loads
    // (the advance comes first: a new tile's geometry needs ~30 temporaries, and here the previous pair's activations are already consumed)
    if (a_left > 2) {
      a_left -= 2;
      a_ci += 2;
      if (a_ci == nch) {
        ADM_SCHED_FENCE();
        a_ci = 0; a_v += bs;
        a_geometry();
      }
    }
    ADM_SCHED_FENCE();
    const int c0 = a_ci * WCK;
#if !defined(ADM_EMU)
    const __amdgpu_buffer_rsrc_t rx = c0 < p.C1 ? a_rx1 : a_rx2;
    const int so = c0 * planeS * 4;
    ADM_UNROLL
    for (int s = 0; s < NS; ++s) {
      if (s < NF) r.v[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, a_vo[s], so, 0));
      else {
        const float x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, a_vo[s], so, 0));
        if (UP && s < 3) r.v[s][0] = x;
        else if (s == 3) r.h1 = x;
        else r.h0 = x;
      }
    }
#else
    const float* base = (c0 < p.C1 ? a_x1 : a_x2) + (long)c0 * planeS;
    ADM_UNROLL
    for (int s = 0; s < NS; ++s) {
      if (s < NF) r.v[s] = *reinterpret_cast<const f32x4*>(base + a_vo[s] / 4);
      else {
        const float x = base[a_vo[s] / 4];
        if (UP && s < 3) r.v[s][0] = x;
        else if (s == 3) r.h1 = x;
        else r.h0 = x;
      }
    }
#endif
    r.ok = a_ok;
    r.sg = a_n * p.gn_nstride + c0;
    ADM_UNROLL
    for (int s = 0; s < NS; ++s) {             // (L2 / L1 hits; a whole MFMA block passes before stage B reads them)
#if !defined(ADM_EMU)
      b_sc[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g_rs, (it_pk[s] & 15) * 4, r.sg * 4, 0));
      b_sh[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(h_rs, (it_pk[s] & 15) * 4, r.sg * 4, 0));
#else
      b_sc[s] = p.gn_scale[r.sg + (it_pk[s] & 15)]; b_sh[s] = p.gn_shift[r.sg + (it_pk[s] & 15)];
#endif
    }
  };
  constexpr bool act_on = ACT != 0;
  auto act1 = [&](float x, float sc, float sh) { const float v0 = x * sc + sh; return act_on ? silu_w(v0) : v0; };
  auto stage_b = [&](const Raw& r, int g) {    // raw -> GroupNorm affine (+ SiLU) -> patch buffers of chunks g, g + 1 (zero padding = zeroed affine)
    float* P = ldsP + (g & 3) * W6PSLAB;       // (a pair never wraps the ring: g is even, so slab g + 1 follows slab g)
    // ONE scheduling region: beside a partner wave that keeps the matrix pipe full every dependent step of this block waits ~40 cycles for
    // its turn (cycle accounting, profiles/r05_wino.md), so the up to thirteen activation chains must run side by side, not one behind the
    // other (fenced slot by slot — as the registers demanded while the filter ring was alive here — the block took 2-3k cycles per pair).
    ADM_UNROLL
    for (int s = 0; s < NS; ++s) {
      const bool ok = (r.ok >> s) & 1u;
      const float c = ok ? b_sc[s] : 0.f, h = ok ? b_sh[s] : 0.f;
      float* dst = P + (it_pk[s] >> 4);
      if (s < NF) {
        dst[0] = act1(r.v[s][0], c, h); dst[1] = act1(r.v[s][1], c, h);
        dst[2] = act1(r.v[s][2], c, h); dst[3] = act1(r.v[s][3], c, h);
      } else {
        const float x = (UP && s < 3) ? r.v[s][0] : (s == 3 ? r.h1 : r.h0);
        dst[0] = act1(x, c, h);
      }
    }
  };
  auto stage_c_half = [&](int g, auto half_c) {  // one copy per half: each a single basic block, its 30 window reads free to run ahead of the math
    constexpr int HALF = decltype(half_c)::value;
    const float* P = ldsP + ((g + c_c2) & 3) * W6PSLAB + c_wbase;
    float* V = ldsV + ((g + c_c2) & 3) * W6VSLAB + c_vofs + (HALF ? 18 * 128 : 0);
    // rows of d this half needs: half 0 -> d rows 0..4 (V rows 0, 1, 2), half 1 -> d rows 1..5 (V rows 3, 4, 5)
    float t[3][6];
    ADM_UNROLL
    for (int l = 0; l < 6; ++l) {
      float r[5];                              // r[k] = d[HALF + k][l]
      ADM_UNROLL
      for (int k = 0; k < 5; ++k) {
        if (UP) r[k] = HALF ? P[((k + 2) >> 1) * 10 + ((l + 1) >> 1)] : P[((k + 1) >> 1) * 10 + ((l + 1) >> 1)];
        else r[k] = P[(HALF + k) * W6PP + l];
      }
      if (HALF) {                              // V rows 3, 4, 5 from d rows 1..5
        const float c_ = r[3] - r[1], e_ = r[2] - r[0];
        t[0][l] = fmaf(2.f, e_, c_); t[1][l] = fmaf(-2.f, e_, c_);
        t[2][l] = fmaf(4.f, r[0], fmaf(-5.f, r[2], r[4]));
      } else {                                 // V rows 0, 1, 2 from d rows 0..4
        const float a_ = fmaf(-4.f, r[2], r[4]), b_ = fmaf(-4.f, r[1], r[3]);
        t[0][l] = fmaf(4.f, r[0], fmaf(-5.f, r[2], r[4]));
        t[1][l] = a_ + b_; t[2][l] = a_ - b_;
      }
    }
    ADM_UNROLL
    for (int i = 0; i < 3; ++i) {
      float v0, v1, v2, v3, v4, v5;
      W6_BT(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], t[i][5], v0, v1, v2, v3, v4, v5);
      float* dst = V + (6 * i) * 128;
      dst[0] = v0; dst[128] = v1; dst[256] = v2; dst[384] = v3; dst[512] = v4; dst[640] = v5;
    }
  };
  auto stage_c = [&](int g) {                  // patches of chunks g, g + 1 -> this thread's half window -> V = B^T d B (three rows of it)
    if (c_half) stage_c_half(g, std::integral_constant<int, 1>{});
    else stage_c_half(g, std::integral_constant<int, 0>{});
  };
  // ---- filter stream: ring of W6AR point groups, in memory order [chunk][ks][pg] ---------------------------------------------------------
  // (raw buffer loads: resource = the whole image, lane term = lane * 16 bytes in ONE register, everything else — this wave's cout block, the
  // chunk, the group — a scalar offset: global loads 1 KiB apart needed a 64-bit VGPR pair per 4 KiB of immediate range)
  int d_v = b0, d_ci = 0, d_left = total;
  const int chunk_stride = n_cblk * W6ABLK;                      // floats; the image of a 512 -> 512 layer is 38 MB: 32-bit offsets
  int d_cur = ((wino5_tile(p, d_v).m0 >> 4) + wave) * W6ABLK;    // float offset of the chunk being consumed (this wave's cout block)
  int d_nxt = d_cur;                                             // ... of the chunk after it (saturating)
  auto advance_next = [&]() {
    if (d_left > 1) {
      --d_left;
      d_nxt += chunk_stride;
      if (++d_ci == nch) {
        ADM_SCHED_FENCE();
        d_ci = 0; d_v += bs;
        d_nxt = ((wino5_tile(p, d_v).m0 >> 4) + wave) * W6ABLK;
      }
    }
  };
  advance_next();                              // d_nxt = chunk 1
#if !defined(ADM_EMU)
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wu), (short)0, 0x7fffffff, 0x00027000);
  const int w_vo = lane * 16;
#define W6_LOAD_A(off_floats) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rs, w_vo, (off_floats) * 4, 0))
#else
#define W6_LOAD_A(off_floats) (*reinterpret_cast<const f32x4*>(p.wu + (off_floats) + lane * 4))
#endif
  f32x4 aR[W6AR];
  ADM_UNROLL
  for (int q = 0; q < W6AR; ++q) aR[q] = W6_LOAD_A(d_cur + q * 256);
  // ---- prologue ------------------------------------------------------------------------------------------------------------------
  Raw r0;
  ADM_UNROLL
  for (int s = 0; s < 3; ++s) r0.v[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  r0.h0 = 0.f; r0.h1 = 0.f;
  int pg = 0;
  stage_a(r0);                                 // chunks 0, 1
  stage_b(r0, 0);
  stage_a(r0);                                 // chunks 2, 3
  ADM_BARRIER_KEEP_VMEM(63);                   // patches 0, 1 complete
  stage_c(0);                                  // V(0), V(1)
  stage_b(r0, 2);
  stage_a(r0);                                 // chunks 4, 5
  pg = 2;
  ADM_BARRIER_KEEP_VMEM(63);                   // V(0), V(1) and patches 2, 3 complete
  const int vlane = k4 * 16 + l15;
  f32x4 acc[36];
  const long planeO = (long)p.Ho * p.Wo;
  int v = b0 - bs, ci = nch;
  Wino3Tile t = wino5_tile(p, b0);
  const int tyy = l15 >> 2, txx = l15 & 3;
#if !defined(ADM_EMU)
  __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc(p.out, (short)0, 0x7fffffff, 0x00027000), r_rs = o_rs;
  const int plane_b = (int)planeO * 4, row_b = p.Wo * 4;
#endif
  int o_vo = 0;                                // element (emulator) / byte offset of this lane's tile inside the wave's 16 cout planes
  // Epilogue operands, fetched one stage B ahead of the epilogue (the filter ring is dead there): the residual rows of TWO cout rows (2 x 16
  // registers; fetched row by row just in time, the sixteen HBM round trips of a tile ran one behind the other — 18 of a tile's 54 us) and the
  // four cout rows' bias and per-sample term (the time embedding projection).
  struct EpiOps { f32x4 res[2][4]; float bias[4], add[4]; };
  auto load_res = [&](int r, int a) {
#if !defined(ADM_EMU)
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rs, o_vo, r * plane_b + a * row_b, 0));
#else
    return *reinterpret_cast<const f32x4*>(p.residual + ((long)t.n * p.Cout + t.m0 + 16 * wave) * planeO + o_vo + r * planeO + a * p.Wo);
#endif
  };
  auto epilogue_fetch = [&](EpiOps& e) {
    ADM_UNROLL
    for (int r = 0; r < 4; ++r) {
      const int co = t.m0 + 16 * wave + 4 * k4 + r;
      e.bias[r] = p.bias[co];
      e.add[r] = p.chan_add[(long)t.n * p.chan_add_stride + co];
    }
    ADM_UNROLL
    for (int q = 0; q < 2; ++q)
      ADM_UNROLL
      for (int a = 0; a < 4; ++a) e.res[q][a] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.residual != nullptr) {
      ADM_UNROLL
      for (int q = 0; q < 2; ++q)
        ADM_UNROLL
        for (int a = 0; a < 4; ++a) e.res[q][a] = load_res(q, a);
    }
  };
  auto epilogue = [&](EpiOps& e) {             // lane-local inverse transform Y = A^T M A (6x6 -> 4x4), bias / per-sample term / residual, stores
    ADM_UNROLL
    for (int r = 0; r < 4; ++r) {
      const int co = t.m0 + 16 * wave + 4 * k4 + r;
      const float bsum = e.bias[r] + e.add[r];
      float f1 = 0.f, f2 = 0.f;
      ADM_UNROLL
      for (int a = 0; a < 4; ++a) {
        // row a of A^T M for the six columns, then that row times A — the column transforms are recomputed per output row (14 instead of 10
        // operations per column) so that six, not twenty-four, intermediate values are alive beside the 32 residual registers
        float tr[6];
        ADM_UNROLL
        for (int j = 0; j < 6; ++j) {
          const float m0 = acc[0 * 6 + j][r], m1 = acc[1 * 6 + j][r], m2 = acc[2 * 6 + j][r], m3 = acc[3 * 6 + j][r], m4 = acc[4 * 6 + j][r],
                      m5 = acc[5 * 6 + j][r];
          tr[j] = a == 0 ? (m0 + (m1 + m2)) + (m3 + m4) : a == 1 ? fmaf(2.f, m3 - m4, m1 - m2) : a == 2 ? fmaf(4.f, m3 + m4, m1 + m2)
                                                                                                  : fmaf(8.f, m3 - m4, m1 - m2) + m5;
        }
        f32x4 y;
        W6_AT(tr[0], tr[1], tr[2], tr[3], tr[4], tr[5], y[0], y[1], y[2], y[3]);
        ADM_UNROLL
        for (int b = 0; b < 4; ++b) y[b] += bsum;
        if (p.residual != nullptr) {
          ADM_UNROLL
          for (int b = 0; b < 4; ++b) y[b] += e.res[r & 1][a][b];
        }
#if !defined(ADM_EMU)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), o_rs, o_vo, r * plane_b + a * row_b, 0);
#else
        *reinterpret_cast<f32x4*>(p.out + ((long)t.n * p.Cout + t.m0 + 16 * wave) * planeO + o_vo + r * planeO + a * p.Wo) = y;
#endif
        f1 += (y[0] + y[1]) + (y[2] + y[3]);
        f2 += (y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3]);
      }
      if (p.stats != nullptr) {                // (sum, sum of squares) of this cout row over the 16x16 tile: 16 values per lane in fp32, lanes in fp64
        double s1 = (double)f1, s2 = (double)f2;
#if !defined(ADM_EMU)
        // rotations inside the 16-lane row as DPP moves (row_ror 8, 4, 2, 1): the same pairs as the xor butterfly — so the same bits — without
        // sixteen ds_bpermute round trips per cout row
        auto ror = [](double x, auto ctrl) {
          const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
          const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, decltype(ctrl)::value, 0xf, 0xf, false);
          const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), decltype(ctrl)::value, 0xf, 0xf, false);
          return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
        };
        s1 += ror(s1, std::integral_constant<int, 0x128>{}); s2 += ror(s2, std::integral_constant<int, 0x128>{});
        s1 += ror(s1, std::integral_constant<int, 0x124>{}); s2 += ror(s2, std::integral_constant<int, 0x124>{});
        s1 += ror(s1, std::integral_constant<int, 0x122>{}); s2 += ror(s2, std::integral_constant<int, 0x122>{});
        s1 += ror(s1, std::integral_constant<int, 0x121>{}); s2 += ror(s2, std::integral_constant<int, 0x121>{});
#else
        ADM_UNROLL
        for (int m = 8; m >= 1; m >>= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }
#endif
        if (l15 == 0) {
          const int tiles = p.tiles_x * p.tiles_y;
          double* dst = p.stats + (((long)t.n * p.Cout + co) * tiles + t.ty * p.tiles_x + t.tx) * 2;
          dst[0] = s1; dst[1] = s2;
        }
      }
      ADM_SCHED_FENCE();
      // the row after next is fetched here, not piece by piece above: the statistics' shuffles find these 16 registers free
      if (p.residual != nullptr && r + 2 < 4) {
        ADM_UNROLL
        for (int a = 0; a < 4; ++a) e.res[r & 1][a] = load_res(r + 2, a);
      }
    }
  };
  bool pend = false;                           // a finished tile waits for its inverse transform + stores
  auto tile_switch = [&]() {
    ADM_SCHED_FENCE();
    ci = 0; v += bs;
    t = wino5_tile(p, v);
    ADM_UNROLL
    for (int q = 0; q < 36; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int oy = t.ty * 16 + 4 * tyy, ox = t.tx * 16 + 4 * txx;
#if !defined(ADM_EMU)
    const long tbase = ((long)t.n * p.Cout + t.m0 + 16 * wave) * planeO;
    o_rs = __builtin_amdgcn_make_buffer_rsrc(p.out + tbase, (short)0, 0x7fffffff, 0x00027000);
    if (p.residual != nullptr) r_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual) + tbase, (short)0, 0x7fffffff, 0x00027000);
    o_vo = (4 * k4 * (int)planeO + oy * p.Wo + ox) * 4;
#else
    o_vo = 4 * k4 * (int)planeO + oy * p.Wo + ox;
#endif
  };
  // ---- staging block P: B(pg + 2, pg + 3), [the finished tile's epilogue], A(next pair), C(pg, pg + 1) ----------------------------------
  // (A in front of C: vmcnt retires in order, so the MFMA block's first wait for a filter group also waits for every older load and store —
  // the activations' HBM round trip and the epilogue's stores must be given stage C's time, not the MFMA block's.)
  // The epilogue sits in front of stage A: there the prefetched activations have been consumed and the filter ring
  // (24: not refilled behind a tile's last chunk) is dead, which is what its 32 residual registers need.
  auto staging = [&](bool more) {            // (more: false = nothing but the last tile's epilogue)
    // Priority: the SIMD's arbiter serves its older wave first, so the younger one (waves 4-7) staged only in the gaps of its partner's MFMA
    // stream — 10.5k cycles for a block that takes the older wave 7k (cycle accounting, profiles/r05_wino.md) — and every barrier waited for
    // it. A staging block is short dependent chains of VALU / LDS / memory instructions: it gets the issue slots first; the partner's MFMAs
    // need one slot in eight and fill the rest.
#if !defined(ADM_EMU)
    __builtin_amdgcn_s_setprio(2);
#endif

    if (pend) {                                // (stage B twice in the source: the epilogue's operands live in this branch only)
      EpiOps e;
      epilogue_fetch(e);                       // (their HBM / L2 round trips pass under stage B)
      if (more) {
        stage_b(r0, pg + 2);
        ADM_SCHED_FENCE();
      }

      epilogue(e);

      pend = false;
      ADM_UNROLL
      for (int q = 0; q < W6AR; ++q) aR[q] = W6_LOAD_A(d_cur + q * 256);
    } else {
      if (more) {
        stage_b(r0, pg + 2);
        ADM_SCHED_FENCE();
      }

    }
    // (unconditional — behind the last pair the saturated cursor re-reads it — so that the activations and their scale / shift are dead
    // across the epilogue in the compiler's eyes too)
    stage_a(r0);
    // (W6X_RING_IN_P, measured and not adopted: the filter ring's first six groups of the NEXT MFMA block fetched here instead of behind the
    // previous block's last groups — the MFMA blocks get 15 % shorter and stage B's waits stop covering these loads, but stage A grows by as
    // much: 58.5 vs 57.7 ms per forward, profiles/r05_wino.md)
    ADM_SCHED_FENCE();

    if (more) {
      stage_c(pg);
    }
    pg += 2;
    ADM_SCHED_FENCE();

#if !defined(ADM_EMU)
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  // The two halves of the workgroup run an interval in opposite order (inside an interval the staging block and the MFMA block touch disjoint
  // ring slots): waves 4-7 run P(it), M(it), barrier; waves 0-3 run M(it), P(it), barrier — written as ONE loop body [P; M] in which the
  // first half's P is the previous interval's and its barrier sits between the two blocks (s_barrier counts arrivals, not program counters):
  // while one wave of a SIMD stages, its partner owns the matrix pipe.
  for (int it = 0; it <= npairs; ++it) {
    if (yrole || it > 0) staging(yrole ? it < npairs : true);
    if (!yrole && it > 0) { ADM_BARRIER_KEEP_VMEM(63); }
    if (it == npairs) break;
    if (ci == nch) tile_switch();

    // ---- M: the 144 MFMAs of chunks g, g + 1 -------------------------------------------------------------------------------------------
    const int g = 2 * it;
    float rbw[3][4];                           // B operands: a window of three point groups (read three groups ahead of their MFMAs)
    auto read_b = [&](int slot, int gg, int gi) {
      const float* Vb = ldsV + (gg & 3) * W6VSLAB + vlane + (4 * (gi % 9)) * 128 + (4 * (gi / 9)) * 16;
      ADM_UNROLL
      for (int e = 0; e < 4; ++e) rbw[slot][e] = Vb[e * 128];
    };
    read_b(0, g, 0); read_b(1, g, 1); read_b(2, g, 2);
    // (a real two-trip loop, NOT unrolled — 18 groups = 3 turns of the filter ring and 6 of the B window, so both chunks run the same code:
    // the loop-carried values pin the 144 accumulators and the rings in place; unrolled, hipcc renamed them across the copies and spilled)
    _Pragma("clang loop unroll(disable)")
    for (int c2 = 0; c2 < 2; ++c2) {
      ADM_UNROLL
      for (int gi = 0; gi < 18; ++gi) {        // point group gi = 9 ks + pgi of this chunk
        const int pgi = gi % 9;
        ADM_UNROLL
        for (int e = 0; e < 4; ++e)
          acc[4 * pgi + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(aR[gi % W6AR][e], rbw[gi % 3][e], acc[4 * pgi + e], 0, 0, 0);
        // (behind the pair's second chunk these are words of a slab that is not certified yet — never used: the next block primes afresh)
        if (gi + 3 < 18) read_b(gi % 3, g + c2, gi + 3);
        else read_b(gi % 3, g + c2 + 1, gi + 3 - 18);
        // the ring slot takes the group W6AR places further down the stream (this chunk's, or the next chunk's first ones)
        // (behind a tile's LAST chunk the ring is not refilled: the epilogue that follows needs those 24 registers, and the next tile's
        // first six groups are loaded right behind it — one exposed L2 round trip per tile)
        if (gi + W6AR < 18) aR[gi % W6AR] = W6_LOAD_A(d_cur + (gi + W6AR) * 256);
        else if (!(c2 == 1 && ci + 2 == nch)) aR[gi % W6AR] = W6_LOAD_A(d_nxt + (gi + W6AR - 18) * 256);
        ADM_SCHED_FENCE();
      }
      d_cur = d_nxt;
      advance_next();
    }
    ci += 2;
    pend = ci == nch;

    if (yrole) { ADM_BARRIER_KEEP_VMEM(63); }
  }
#undef W6_LOAD_A
}

template <bool UP, int ACT>
__global__ void __launch_bounds__(512) conv_wino6_kernel(const WinoParams p) {
  ADM_DYN_SMEM(float, smem);
  float* ldsV = smem;
  float* ldsP = smem + 4 * W6VSLAB;
  const int tid = threadIdx.x;
  const int wave = ADM_UNIFORM(tid >> 6);
  if (UP) {
    if (wave == 0) wino6_wave<UP, 0, ACT>(p, ldsV, ldsP, tid, wave, (int)blockIdx.x, (int)gridDim.x);
    else wino6_wave<UP, 2, ACT>(p, ldsV, ldsP, tid, wave, (int)blockIdx.x, (int)gridDim.x);
  } else {
    if (wave <= 1) wino6_wave<UP, 0, ACT>(p, ldsV, ldsP, tid, wave, (int)blockIdx.x, (int)gridDim.x);
    else if (wave == 2) wino6_wave<UP, 1, ACT>(p, ldsV, ldsP, tid, wave, (int)blockIdx.x, (int)gridDim.x);
    else wino6_wave<UP, 2, ACT>(p, ldsV, ldsP, tid, wave, (int)blockIdx.x, (int)gridDim.x);
  }
}

int launch_wino6(const WinoParams& p, bool up, bool act, int grid, hipStream_t st) {
#if !defined(ADM_EMU)
  {
    static int state[16] = {};               // per device: 0 unknown, 1 granted, -1 refused
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    int& s = state[conv_dev_slot() & 15];
    if (s == 0) {
      const int by6 = (int)(sizeof(float) * W6LDS);
      bool ok6 = true;
      ok6 &= hipFuncSetAttribute((const void*)conv_wino6_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, by6) == hipSuccess;
      ok6 &= hipFuncSetAttribute((const void*)conv_wino6_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, by6) == hipSuccess;
      ok6 &= hipFuncSetAttribute((const void*)conv_wino6_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, by6) == hipSuccess;
      ok6 &= hipFuncSetAttribute((const void*)conv_wino6_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, by6) == hipSuccess;
      if (!ok6) (void)hipGetLastError();
      s = ok6 ? 1 : -1;
    }
    ADM_REQUIRE(s > 0, "conv_winograd: the runtime refused 125 KiB of dynamic LDS for conv_wino6_kernel");
  }
#endif
  const size_t need6 = sizeof(float) * W6LDS;
  if (up) {
    if (act) ADM_LAUNCH((conv_wino6_kernel<true, 1>), dim3(grid), dim3(512), need6, st, p);
    else ADM_LAUNCH((conv_wino6_kernel<true, 0>), dim3(grid), dim3(512), need6, st, p);
  } else {
    if (act) ADM_LAUNCH((conv_wino6_kernel<false, 1>), dim3(grid), dim3(512), need6, st, p);
    else ADM_LAUNCH((conv_wino6_kernel<false, 0>), dim3(grid), dim3(512), need6, st, p);
  }
  return ADM_CHECK_LAUNCH();
}

}  // namespace adm
