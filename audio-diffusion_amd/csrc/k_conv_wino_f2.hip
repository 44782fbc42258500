// k_conv_wino_f2.hip — Winograd F(2x2,3x3) kernels of the fused 3x3 stride-1 convolution (fp32 throughout): conv_wino4_kernel and
// conv_wino5_kernel. Same fusions as k_conv_mfma.hip's kernels (virtual concat, nearest-x2 upsample, zero padding, GroupNorm affine + SiLU on
// load; bias + time-embedding term + residual in the epilogue, GroupNorm partial sums of the output), the 9-tap correlation replaced by 16
// element-wise products in the Winograd domain: 2.25x fewer MFMA FLOPs per output.
//   Y = A^T [ (G g G^T) . (B^T d B) ] A        per 4x4 input tile d -> 2x2 output tile, summed over input channels
#include "k_conv_wino.h"

namespace adm {

// ---------------------------------------------------------------------------------------------------------------------
// Persistent, wave-specialised kernels (conv_wino4_kernel: 4 producer + 4 consumer waves). The design points that rounds 1-4 measured their
// way to (profiles/r01_pmc_wino.md, r02_wino_v4.md, r03_wino_pair.md, r04_wino.md):
//   * the raw haloed patch of a chunk (8 channels x 10 x 18) is fetched ONCE with float4 row loads several chunks ahead, activated once
//     and staged in a small LDS patch buffer from which the 4x4 windows are read (per-thread window gathers saturated the vector-memory path);
//   * consumers use v_mfma_f32_16x16x4_f32 with a wave owning ALL 16 Winograd points of its (cout, tile) sub-block, so A^T M A is lane-local
//     and outputs go straight from registers to HBM; workgroups are persistent (grid = #CUs, XCD-aware tile walk) and the producers run
//     into the next tile while the consumers finish the current one;
//   * operand words are read several Winograd points ahead of the MFMAs that use them; ONE workgroup barrier per TWO chunks: the V slabs and
//     the patch buffers are rings of four — a producer interval stages V(g), V(g + 1) [stage C twice], then the patches of g + 2, g + 3
//     [stage B twice] and the global loads of g + 6, g + 7, and only then meets the consumers, who by then have read chunks g - 2, g - 1.

// ---- producer role: 256 threads (waves 4..7) ------------------------------------------------------------------------
// Issue budget: every producer instruction costs the co-resident MFMA wave issue time, so every scalar / branch / address instruction
// counts. Hence: per-tile (not per-chunk) 32-bit offsets against buffer resources (the chunk's offset is an SGPR), no per-lane predication
// (disabled lanes write to dummy LDS words, whole-wave roles are scalar branches), the tile cursor's integer divisions behind a real
// (non-speculated) branch, packed forms wherever an operation has one.
struct Wino3Raw {                     // one chunk's raw activations of this thread, prefetched four chunks ahead
  float4 a, b;                        // item 0 / item 1 when it is a float4 row piece (UP: scalars in .x)
  float h;                            // item 1 when it is a halo element
  float sc0, sh0, sc1, sh1;           // GroupNorm scale / shift of the two items' channels
  unsigned ok;                        // bit k: item k lies inside the image (zero padding otherwise)
};

// ACT: compiled without / with SiLU.
template <bool UP, bool WIDE1, int ACT>
__device__ __forceinline__ void wino4_producer(const WinoParams& p, float* ldsV, float* ldsP, int tid, int b0, int bs) {
  constexpr int RING = 3;                     // buffer index mask: rings of four
  const int Ct = p.C1 + p.C2;
  const int planeS = p.Hs * p.Ws;
  const int nch = p.cps;              // chunks per (tile, part): every chunk of the layer unless K is split
  const int ntile = (p.nblk - b0 + bs - 1) / bs;
  const int total = ntile * nch;      // chunks this workgroup stages
  // Non-UP: every thread stages float4 row piece f = tid (item 0); producer wave 0 (WIDE1) also stages pieces 256..319,
  // waves 1..3 the 160 halo elements (item 1) — the role is a template parameter so that no load sits under a runtime
  // branch (a conditional load costs a register copy plus a premature vmcnt wait at the join). UP (source-resolution
  // patch 8 x 6 x 10): two scalars e = tid and 256 + tid. Items beyond the patch go to a private dummy word.
  constexpr bool wide1 = !UP && WIDE1;
  int it_ch[2], it_row[2], it_col[2], it_pofs[2];
  const int dummy = WCK * WPH * WPP + tid;
  if (UP) {
    ADM_UNROLL
    for (int k = 0; k < 2; ++k) {
      const int e = tid + 256 * k;
      const bool en = e < 480;
      const int ec = en ? e : 0;
      it_ch[k] = ec / 60; it_row[k] = (ec % 60) / 10; it_col[k] = ec % 10;
      it_pofs[k] = en ? ec : dummy;
    }
  } else {
    const int row0 = tid >> 2, q0 = tid & 3;
    it_ch[0] = row0 / WPH; it_row[0] = row0 % WPH; it_col[0] = 4 * q0;       // image x = tx*16 + col
    it_pofs[0] = row0 * WPP + 1 + 4 * q0;
    if (wide1) {
      const int f = 256 + tid;
      const int row = f >> 2, q = f & 3;
      it_ch[1] = row / WPH; it_row[1] = row % WPH; it_col[1] = 4 * q;
      it_pofs[1] = row * WPP + 1 + 4 * q;
    } else {
      const int hI = tid - 64;
      const bool en = hI < 160;
      const int hc = en ? hI : 0;
      const int hrow = hc >> 1, side = hc & 1;
      it_ch[1] = hrow / WPH; it_row[1] = hrow % WPH; it_col[1] = side ? 16 : -1;
      it_pofs[1] = en ? hrow * WPP + (side ? 17 : 0) : dummy;
    }
  }
  // stage C: window origin of this thread's (channel, tile) inside the patch
  const int pc = tid >> 5, ptile = tid & 31;
  const int tyy = ptile >> 3, txx = ptile & 7;
  const int wbase = UP ? pc * 60 + tyy * 10 + txx : pc * (WPH * WPP) + 2 * tyy * WPP + 2 * txx;
  const int vofs = pc * 32 + ptile;   // + xi * 256 (unswizzled: the consumers read whole 128-byte channel rows with ds_read_b64, conflict-free)

  // ---- stage A cursor: (tile, chunk) of the next global load ----------------------------------------------------------------
  int a_v = b0, a_ci = 0, a_left = total;
  int a_k0 = 0;                       // first chunk of the tile's part (split K; else 0)
  int a_off0 = 0, a_off1 = 0;         // element offset of the items inside the sample: channel plane + row + column
  unsigned a_ok = 0;
  const float *a_x1 = nullptr, *a_x2 = nullptr, *a_gs = nullptr, *a_gh = nullptr;   // per-tile wave-uniform bases
#if !defined(ADM_EMU)
  // the same bases as buffer resources (SGPR quads). A buffer load takes the per-lane byte offset as a 32-bit VGPR and the chunk's
  // offset as an SGPR, so the per-load 64-bit address arithmetic (sign extension + v_lshl_add_u64: ~10 VALU per chunk) leaves the
  // producers' instruction stream — which is what the co-resident MFMA wave pays for (profiles/r04_wino.md).
  __amdgpu_buffer_rsrc_t a_rx1, a_rx2, a_rgs, a_rgh;
  int a_vo0 = 0, a_vo1 = 0;
  const int ch_vo0 = it_ch[0] * 4, ch_vo1 = it_ch[1] * 4;
#endif
  auto a_geometry = [&]() {
    const Wino3Tile t = wino3_tile(p, a_v);
    a_k0 = t.kpart * nch;
    a_x1 = p.x1 + (long)t.n * p.x1_bs;
    a_x2 = p.x2 + (long)t.n * p.x2_bs - (long)p.C1 * planeS;     // indexed with the concatenated channel number
    a_gs = p.gn_scale + (long)t.n * p.gn_nstride;
    a_gh = p.gn_shift + (long)t.n * p.gn_nstride;
    a_ok = 0;
    int off[2];
    ADM_UNROLL
    for (int k = 0; k < 2; ++k) {
      const int sy = UP ? t.ty * 4 - 1 + it_row[k] : t.ty * 8 - 1 + it_row[k];
      const int sx = UP ? t.tx * 8 - 1 + it_col[k] : t.tx * 16 + it_col[k];
      const bool ok = sy >= 0 && sy < p.Hs && sx >= 0 && sx < p.Ws;   // interior pieces: only the row can fall outside
      off[k] = it_ch[k] * planeS + (ok ? sy * p.Ws + sx : 0);
      a_ok |= ok ? 1u << k : 0u;
    }
    a_off0 = off[0]; a_off1 = off[1];
#if !defined(ADM_EMU)
    a_rx1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_x1), (short)0, 0x7fffffff, 0x00027000);
    a_rx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_x2), (short)0, 0x7fffffff, 0x00027000);
    a_rgs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_gs), (short)0, 0x7fffffff, 0x00027000);
    a_rgh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_gh), (short)0, 0x7fffffff, 0x00027000);
    a_vo0 = a_off0 * 4; a_vo1 = a_off1 * 4;
#endif
  };
  a_geometry();
  auto stage_a = [&](Wino3Raw& r) {           // issue the global loads of chunk (a_v, a_ci); then advance the cursor
    const int c0 = (a_k0 + a_ci) * WCK;
#if !defined(ADM_EMU)
    const __amdgpu_buffer_rsrc_t rx = c0 < p.C1 ? a_rx1 : a_rx2;
    const int so = c0 * planeS * 4, sg = c0 * 4;            // wave-uniform byte offsets of the chunk (< 2^31: one sample's channels)
    if (UP) {
      r.a.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, a_vo0, so, 0));
      r.b.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, a_vo1, so, 0));
    } else {
      r.a = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, a_vo0, so, 0));
      if (wide1) r.b = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, a_vo1, so, 0));
      else r.h = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, a_vo1, so, 0));
    }
    r.sc0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rgs, ch_vo0, sg, 0));
    r.sh0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rgh, ch_vo0, sg, 0));
    r.sc1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rgs, ch_vo1, sg, 0));
    r.sh1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rgh, ch_vo1, sg, 0));
#else
    const float* base = (c0 < p.C1 ? a_x1 : a_x2) + (long)c0 * planeS;
    if (UP) {
      r.a.x = base[a_off0];
      r.b.x = base[a_off1];
    } else {
      r.a = *reinterpret_cast<const float4*>(base + a_off0);
      if (wide1) r.b = *reinterpret_cast<const float4*>(base + a_off1);   // each path loads into its own registers:
      else r.h = base[a_off1];                                            // no merge copies, no wait at the join
    }
    r.sc0 = a_gs[c0 + it_ch[0]]; r.sh0 = a_gh[c0 + it_ch[0]];
    r.sc1 = a_gs[c0 + it_ch[1]]; r.sh1 = a_gh[c0 + it_ch[1]];
#endif
    r.ok = a_ok;
    // advance; past the end the cursor stays on the last chunk (the loads stay unconditional, their data is never used)
    if (a_left > 1) {
      --a_left;
      if (++a_ci == nch) {
        ADM_SCHED_FENCE();             // keeps the divisions of wino3_tile behind this branch (no if-conversion)
        a_ci = 0; a_v += bs;
        a_geometry();
      }
    }
  };
  constexpr bool act_on = ACT != 0;
  // GroupNorm affine (+ SiLU). The caller has zeroed scale AND shift of an out-of-image item: the affine then gives 0, and SiLU(0) = 0 — no
  // per-element select (the padded positions read clamped, i.e. real and finite, activations)
  auto act1 = [&](float x, float sc, float sh) {
    const float v0 = x * sc + sh;
    return act_on ? silu_w(v0) : v0;
  };
  auto stage_b = [&](const Wino3Raw& r0_, int g) {        // raw -> activation -> patch buffer g & 3
    float* P = ldsP + (g & RING) * W3PSLAB;
    Wino3Raw r = r0_;
    r.sc0 = (r.ok & 1u) ? r.sc0 : 0.f; r.sh0 = (r.ok & 1u) ? r.sh0 : 0.f;      // zero padding as a zeroed affine: two selects per ITEM
    r.sc1 = (r.ok & 2u) ? r.sc1 : 0.f; r.sh1 = (r.ok & 2u) ? r.sh1 : 0.f;
    if (UP) {
      P[it_pofs[0]] = act1(r.a.x, r.sc0, r.sh0);
      P[it_pofs[1]] = act1(r.b.x, r.sc1, r.sh1);
    } else {
      float* P0 = P + it_pofs[0];
      float* P1 = P + it_pofs[1];
#if !defined(ADM_EMU)
      // two values per instruction wherever the operation has a packed form (affine, the exponent's scaling, 1 + e, the final product):
      // 8 VALU instructions per pair instead of 12; v_exp_f32 / v_rcp_f32 stay scalar. The operations and their order are act1's
      // (__expf(-v) = v_exp_f32(v * -log2(e)), ADM_RCP = v_rcp_f32): bit-identical.
      typedef float wf2 __attribute__((ext_vector_type(2)));
      auto act2 = [&](float x0, float x1, float sc, float sh, float* dst) __attribute__((always_inline)) {
        wf2 v = wf2{x0, x1} * sc + sh;
        if (act_on) {
          const wf2 t = v * -1.44269504088896340736f;
          wf2 e;
          e.x = __builtin_amdgcn_exp2f(t.x); e.y = __builtin_amdgcn_exp2f(t.y);
          const wf2 d = e + 1.0f;
          wf2 q;
          q.x = __builtin_amdgcn_rcpf(d.x); q.y = __builtin_amdgcn_rcpf(d.y);
          v = v * q;
        }
        dst[0] = v.x; dst[1] = v.y;
      };
      act2(r.a.x, r.a.y, r.sc0, r.sh0, P0); act2(r.a.z, r.a.w, r.sc0, r.sh0, P0 + 2);
      if (wide1) { act2(r.b.x, r.b.y, r.sc1, r.sh1, P1); act2(r.b.z, r.b.w, r.sc1, r.sh1, P1 + 2); }
      else P1[0] = act1(r.h, r.sc1, r.sh1);
#else
      P0[0] = act1(r.a.x, r.sc0, r.sh0); P0[1] = act1(r.a.y, r.sc0, r.sh0);
      P0[2] = act1(r.a.z, r.sc0, r.sh0); P0[3] = act1(r.a.w, r.sc0, r.sh0);
      if (wide1) {
        P1[0] = act1(r.b.x, r.sc1, r.sh1); P1[1] = act1(r.b.y, r.sc1, r.sh1);
        P1[2] = act1(r.b.z, r.sc1, r.sh1); P1[3] = act1(r.b.w, r.sc1, r.sh1);
      } else {
        P1[0] = act1(r.h, r.sc1, r.sh1);
      }
#endif
    }
  };
  auto stage_c = [&](int g) {                // patch g & 3 -> 4x4 window -> V = B^T d B -> V buffer g & 3
    const float* P = ldsP + (g & RING) * W3PSLAB + wbase;
    float d[16];
    ADM_UNROLL
    for (int i = 0; i < 4; ++i)
      ADM_UNROLL
      for (int j = 0; j < 4; ++j) d[i * 4 + j] = UP ? P[((i + 1) >> 1) * 10 + ((j + 1) >> 1)] : P[i * WPP + j];
    float* vdst = ldsV + (g & RING) * W3VSLAB + vofs;
#if !defined(ADM_EMU)
    // The 32 additions as 16 packed ones (v_pk_add_f32, full rate on gfx950): the rows first, two columns per instruction; then
    // per row (v0, v1) = (t0 - t2, t1 + t2) and (v2, v3) = (t2 - t1, t1 - t3) through the operand-select / negate modifiers.
    // Same additions on the same values (a - b issued as a + (-b)): bit-identical to the scalar form below.
    typedef float wf2 __attribute__((ext_vector_type(2)));
    wf2 D[4][2], T[4][2];
    ADM_UNROLL
    for (int i = 0; i < 4; ++i) { D[i][0] = wf2{d[i * 4 + 0], d[i * 4 + 1]}; D[i][1] = wf2{d[i * 4 + 2], d[i * 4 + 3]}; }
    ADM_UNROLL
    for (int h2 = 0; h2 < 2; ++h2) {
      T[0][h2] = D[0][h2] - D[2][h2];
      T[1][h2] = D[1][h2] + D[2][h2];
      T[2][h2] = D[2][h2] - D[1][h2];
      T[3][h2] = D[1][h2] - D[3][h2];
    }
    ADM_UNROLL
    for (int i = 0; i < 4; ++i) {
      wf2 lo, hi;
      asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(lo) : "v"(T[i][0]), "v"(T[i][1]));
      asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(hi) : "v"(T[i][0]), "v"(T[i][1]));
      vdst[(i * 4 + 0) * (WCK * 32)] = lo.x;     // t0 - t2
      vdst[(i * 4 + 1) * (WCK * 32)] = lo.y;     // t1 + t2
      vdst[(i * 4 + 2) * (WCK * 32)] = hi.x;     // t2 - t1
      vdst[(i * 4 + 3) * (WCK * 32)] = hi.y;     // t1 - t3
    }
#else
    float t[4][4];
    ADM_UNROLL
    for (int j = 0; j < 4; ++j) {
      t[0][j] = d[0 * 4 + j] - d[2 * 4 + j];
      t[1][j] = d[1 * 4 + j] + d[2 * 4 + j];
      t[2][j] = d[2 * 4 + j] - d[1 * 4 + j];
      t[3][j] = d[1 * 4 + j] - d[3 * 4 + j];
    }
    ADM_UNROLL
    for (int i = 0; i < 4; ++i) {
      vdst[(i * 4 + 0) * (WCK * 32)] = t[i][0] - t[i][2];
      vdst[(i * 4 + 1) * (WCK * 32)] = t[i][1] + t[i][2];
      vdst[(i * 4 + 2) * (WCK * 32)] = t[i][2] - t[i][1];
      vdst[(i * 4 + 3) * (WCK * 32)] = t[i][1] - t[i][3];
    }
#endif
  };
  // ---- pipeline: one barrier per pair of chunks; interval (g, g + 1) stages V(g), V(g + 1) [C], the patches of g + 2, g + 3 [B] and the
  // global loads of g + 6, g + 7 [A] ------------------------------------------------------------------------------------------------------
  Wino3Raw r0, r1, r2, r3;
  r0.b = make_float4(0.f, 0.f, 0.f, 0.f); r1.b = r0.b; r0.a = r0.b; r1.a = r0.b; r0.h = 0.f; r1.h = 0.f;
  r2.b = r0.b; r3.b = r0.b; r2.a = r0.b; r3.a = r0.b; r2.h = 0.f; r3.h = 0.f;
  stage_a(r0); stage_a(r1); stage_a(r2); stage_a(r3);            // chunks 0..3
  stage_b(r0, 0); stage_b(r1, 1);
  stage_a(r0); stage_a(r1);                                      // chunks 4, 5
  ADM_BARRIER_KEEP_VMEM(63);                                     // barrier "-2": patches 0 and 1 visible to every producer wave
  for (int g = 0; g < total; g += 4) {                           // total is a multiple of 4 (nch is)
    stage_c(g); stage_c(g + 1);
    stage_b(r2, g + 2); stage_b(r3, g + 3);
    stage_a(r2); stage_a(r3);                                    // chunks g + 6, g + 7
    ADM_BARRIER_KEEP_VMEM(63);
    stage_c(g + 2); stage_c(g + 3);
    stage_b(r0, g + 4); stage_b(r1, g + 5);
    stage_a(r0); stage_a(r1);                                    // chunks g + 8, g + 9
    ADM_BARRIER_KEEP_VMEM(63);
  }
  ADM_BARRIER_KEEP_VMEM(0);
}

// =====================================================================================================================
// v4 (mode 4) — v3 with the FILTER operand taken out of LDS. What v3's measurements asked for (profiles/r01_pmc_wino.md):
// its consumer stream alone needs 3200 cycles per chunk against 2048 of MFMA — 550 of them are the eight LDS-DMA pieces
// per wave that bring the 32 KiB U slab in, and two thirds of its 96 LDS operand reads per chunk are filter words.
//   * wave w owns 16 couts x ALL 32 Winograd tiles of the workgroup tile (v3: 32 couts x 16 tiles), so no two waves need
//     the same filter words and every A operand is loaded exactly once per workgroup: straight from L2 into registers,
//     8 global_load_dwordx4 per wave and chunk from a filter image packed for exactly this access
//     ([chunk][cout block][point group][k step][lane][4 points]: one contiguous KiB per load), refilled IN PLACE one whole
//     chunk ahead — the four points of a group are consumed, then the group's registers are reloaded for the next chunk;
//   * the B operand of both tile blocks comes from one ds_read_b64 (tiles 2 li, 2 li + 1): 32 LDS reads per wave and chunk
//     over plain, conflict-free 128-byte channel rows; a lane's two tiles are horizontal neighbours, so the lane-local
//     inverse transform ends in 16-byte stores;
//   * no LDS-DMA anywhere: the per-chunk barrier only hands V buffers over, and no vmcnt is ever drained at it;
//   * LDS: V 2 x 16 KiB + patch 2 x 5.6 KiB = 43 KiB.
constexpr int W4LDS = 2 * W3VSLAB + 2 * W3PSLAB;

__device__ __forceinline__ void wino4_consumer(const WinoParams& p, const float* ldsV, int tid, int wave, int b0, int bs) {
  constexpr int RING = 3;                   // rings of four V slabs
  const int lane = tid & 63;
  const int li = lane & 15, k4 = lane >> 4;
  const int nch = p.cps;                    // chunks per (tile, part)
  const int n_cblk = p.Cout >> 4;
  const int ntile = (p.nblk - b0 + bs - 1) / bs;
  const int total = ntile * nch;
  const int vlane = k4 * 32 + 2 * li;       // word pair (tile 2 li, 2 li + 1) of channel row k4 (+ 4 ks)
  // ---- filter stream cursor: (tile, chunk) of the NEXT chunk to load; saturates on the last one ---------------------------
  // (plain global loads: the eight loads of a chunk share one address register pair; raw buffer loads measured 1.5-4 % slower here, r04)
  int d_v = b0, d_ci = 0, d_left = total;
  const long chunk_stride = (long)n_cblk * W4ABLK;
  auto a_origin = [&](int v) {              // first chunk of virtual block v's (tile, part): this wave's 16-cout block, this lane's words
    const Wino3Tile t = wino3_tile(p, v);
    return p.wu + (long)t.kpart * nch * chunk_stride + ((long)(t.m0 >> 4) + wave) * W4ABLK + lane * 4;
  };
  const float* d_src = a_origin(d_v);
  f32x4 a[4][2];                            // [point group][k step]: component e = Winograd point 4 q + e
#define W4_LOAD_A(q)                                                                   \
  do {                                                                                 \
    a[q][0] = *reinterpret_cast<const f32x4*>(d_src + (q) * 512);                      \
    a[q][1] = *reinterpret_cast<const f32x4*>(d_src + (q) * 512 + 256);                \
  } while (0)
  auto advance_a = [&]() {
    if (d_left > 1) {
      --d_left;
      d_src += chunk_stride;
      if (++d_ci == nch) {
        ADM_SCHED_FENCE();
        d_ci = 0; d_v += bs;
        d_src = a_origin(d_v);
      }
    }
  };
  W4_LOAD_A(0); W4_LOAD_A(1); W4_LOAD_A(2); W4_LOAD_A(3);      // chunk 0
  advance_a();
  ADM_BARRIER_KEEP_VMEM(63);               // barrier "-2" (producers' patch hand-over)
  ADM_BARRIER_KEEP_VMEM(63);               // barrier "-1": V(0) complete

  f32x4 acc[16][2];
  // rolling B window, running across tile boundaries: RB Winograd points ahead. 4 points = 16 MFMAs = 512 cycles of cover for an LDS
  // read that the producers' traffic delays; the role accounting of round 4 had the consumer on the critical path with ~900 non-MFMA
  // cycles per chunk, so the window is 8 points (the chunk hand-over barrier sits at point 8, where the first read of the next chunk
  // is issued — the producers have 16 % of barrier slack)
  constexpr int RB = 8;
  float2 rb[RB][2];
  auto read_group = [&](int slot, int gg, int xi) {
    const float* V = ldsV + (gg & RING) * W3VSLAB + vlane;
    rb[slot][0] = *reinterpret_cast<const float2*>(V + (xi * WCK) * 32);
    rb[slot][1] = *reinterpret_cast<const float2*>(V + (xi * WCK + 4) * 32);
  };
  ADM_UNROLL
  for (int xi = 0; xi < RB; ++xi) read_group(xi, 0, xi);
  int g = 0;                               // running chunk index
  const long planeO = (long)p.Ho * p.Wo;
  for (int v = b0; v < p.nblk; v += bs) {
    const Wino3Tile t = wino3_tile(p, v);
    ADM_UNROLL
    for (int xi = 0; xi < 16; ++xi)
      ADM_UNROLL
      for (int c = 0; c < 2; ++c)
        ADM_UNROLL
        for (int r = 0; r < 4; ++r) acc[xi][c][r] = 0.f;
    const int oy = t.ty * 8 + 2 * (li >> 2), ox = t.tx * 16 + 4 * (li & 3);
    const long obase = (long)t.kpart * p.part_stride +      // (split K: slab kpart of the partial-sum buffer; the caller passes zero bias rows)
                       ((long)t.n * p.Cout + t.m0 + 16 * wave + 4 * k4) * planeO + (long)oy * p.Wo + ox;   // cout row r: + r * planeO
    // Bias, per-sample term and residual enter in the WINOGRAD domain: Y = A^T M A has Y00 / Y01 / Y10 / Y11 depend on the corner
    // entries M00 / M03 / M30 / M33 alone with weights +1 / -1 / -1 / +1, so adding (b + res) there is adding it to the output.
    // One cout row per chunk over the first four chunks: the loads are issued when the chunk starts and consumed when it ends
    // — a whole chunk of latency cover for 10 registers — and the epilogue is left with arithmetic and stores only.
    f32x4 fr0 = {0.f, 0.f, 0.f, 0.f}, fr1 = fr0;
    float fb0 = 0.f, fb1 = 0.f;
    for (int ci = 0; ci < nch; ++ci, ++g) {
      if (ci < 4) {                        // wave-uniform: this chunk carries cout row r = ci of the fold
        const int co = t.m0 + 16 * wave + 4 * k4 + ci;
        fb0 = p.bias[co];
        fb1 = p.chan_add[(long)t.n * p.chan_add_stride + co];
        if (p.residual != nullptr) {
          fr0 = *reinterpret_cast<const f32x4*>(p.residual + obase + ci * planeO);
          fr1 = *reinterpret_cast<const f32x4*>(p.residual + obase + ci * planeO + p.Wo);
        }
      }
      ADM_UNROLL
      for (int xi = 0; xi < 16; ++xi) {
        const int s = xi & (RB - 1), q = xi >> 2, e = xi & 3;
        ADM_UNROLL
        for (int ks = 0; ks < 2; ++ks) {
          acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][ks][e], rb[s][ks].x, acc[xi][0], 0, 0, 0);
          acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][ks][e], rb[s][ks].y, acc[xi][1], 0, 0, 0);
        }
        if (e == 3) {                      // group q consumed: its registers take the NEXT chunk's words
          if (q == 0) W4_LOAD_A(0);
          if (q == 1) W4_LOAD_A(1);
          if (q == 2) W4_LOAD_A(2);
          if (q == 3) { W4_LOAD_A(3); advance_a(); }
        }
        // the pair's barrier: every read of V(g - 1), V(g) has landed, V(g + 1), V(g + 2) are complete — only behind the second chunk of a
        // pair (its first chunk runs on into V(g + 1), which the previous pair's barrier certified)
        if (xi == 16 - RB && (g & 1)) ADM_BARRIER_KEEP_VMEM(63);
        if (xi < 16 - RB) read_group(s, g, xi + RB);
        else read_group(s, g + 1, xi - (16 - RB));   // next chunk — of this tile or the next one (past the end: stale words, unused)
        ADM_SCHED_FENCE();
      }
      if (ci < 4) {
        const float bsum = fb0 + fb1;
#define W4_FOLD(R)                                                                                         \
  do {                                                                                                     \
    acc[0][0][R] += bsum + fr0[0];  acc[0][1][R] += bsum + fr0[2];                                          \
    acc[3][0][R] -= bsum + fr0[1];  acc[3][1][R] -= bsum + fr0[3];                                          \
    acc[12][0][R] -= bsum + fr1[0]; acc[12][1][R] -= bsum + fr1[2];                                         \
    acc[15][0][R] += bsum + fr1[1]; acc[15][1][R] += bsum + fr1[3];                                         \
  } while (0)
        if (ci == 0) W4_FOLD(0);
        else if (ci == 1) W4_FOLD(1);
        else if (ci == 2) W4_FOLD(2);
        else W4_FOLD(3);
#undef W4_FOLD
      }
    }
    // ---- lane-local inverse transform Y = A^T M A: lane holds couts 4 k4 + r and tiles 2 li (c = 0), 2 li + 1 (c = 1) ----------
    ADM_UNROLL
    for (int r = 0; r < 4; ++r) {
      f32x4 y0, y1;
      ADM_UNROLL
      for (int c = 0; c < 2; ++c) {
        float t0[4], t1[4];
        ADM_UNROLL
        for (int j = 0; j < 4; ++j) {
          t0[j] = acc[0 * 4 + j][c][r] + acc[1 * 4 + j][c][r] + acc[2 * 4 + j][c][r];
          t1[j] = acc[1 * 4 + j][c][r] - acc[2 * 4 + j][c][r] - acc[3 * 4 + j][c][r];
        }
        y0[2 * c] = t0[0] + t0[1] + t0[2];
        y0[2 * c + 1] = t0[1] - t0[2] - t0[3];
        y1[2 * c] = t1[0] + t1[1] + t1[2];
        y1[2 * c + 1] = t1[1] - t1[2] - t1[3];
      }
      *reinterpret_cast<f32x4*>(p.out + obase + r * planeO) = y0;
      *reinterpret_cast<f32x4*>(p.out + obase + r * planeO + p.Wo) = y1;
      if (p.stats != nullptr) {            // wave-uniform: (sum, sum of squares) of this cout row over the 8 x 16 tile
        // the lane's 8 values in fp32 (8 + 8 operations), everything across lanes and tiles in fp64: the fp32 part adds a
        // relative error of ~1e-7 to a 8-term sum, far below what the consumer (an fp32 scale / shift) resolves
        float f1 = (y0[0] + y0[1]) + (y0[2] + y0[3]) + ((y1[0] + y1[1]) + (y1[2] + y1[3]));
        float f2 = (y0[0] * y0[0] + y0[1] * y0[1]) + (y0[2] * y0[2] + y0[3] * y0[3]) +
                   ((y1[0] * y1[0] + y1[1] * y1[1]) + (y1[2] * y1[2] + y1[3] * y1[3]));
        double s1 = (double)f1, s2 = (double)f2;
        ADM_UNROLL
        for (int m = 8; m >= 1; m >>= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }   // the 16 lanes of this k4
        if (li == 0) {
          const int tiles = p.tiles_x * p.tiles_y;
          double* dst = p.stats + (((long)t.n * p.Cout + t.m0 + 16 * wave + 4 * k4 + r) * tiles + t.ty * p.tiles_x + t.tx) * 2;
          dst[0] = s1; dst[1] = s2;
        }
      }
    }
  }
#undef W4_LOAD_A
}

template <bool UP, int ACT>
__global__ void __launch_bounds__(512) conv_wino4_kernel(const WinoParams p) {
  ADM_DYN_SMEM(float, smem);
  float* ldsV = smem;
  float* ldsP = smem + 4 * W3VSLAB;
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  if (wave >= 4) {
#if !defined(ADM_EMU)
    __builtin_amdgcn_s_setprio(1);   // producers: short dependent chains of VALU / LDS / memory instructions — they get the issue slots first
#endif
    if (!UP && wave == 4) wino4_producer<UP, true, ACT>(p, ldsV, ldsP, tid - 256, (int)blockIdx.x, (int)gridDim.x);
    else wino4_producer<UP, false, ACT>(p, ldsV, ldsP, tid - 256, (int)blockIdx.x, (int)gridDim.x);
  }
  else wino4_consumer(p, ldsV, tid, wave, (int)blockIdx.x, (int)gridDim.x);
}

// =====================================================================================================================
// v5 (round 5) — every input patch is transformed ONCE per 128 output channels: the workgroup tile is 128 couts x 8x16 pixels
// and there are no dedicated producer waves any more. What v4's measurements asked for (profiles/r04_wino.md): a v4
// workgroup transforms its patch for 64 couts, so every patch is fetched, activated and transformed Cout / 64 times, and 0.95 of the
// 1.16 ms a 128 -> 128 launch spends above the matrix pipe's own pace is what the co-resident producer wave issues. A 128-cout tile
// needs 128 x 32 x 16 accumulators = half of the CU's register file, i.e. ALL EIGHT waves must hold 128 of them:
//   * all 8 waves are MFMA waves (wave w owns couts 16 w .. 16 w + 15 of the tile x all 32 Winograd tiles x all 16 points: v4's
//     consumer body, filter image and lane-local inverse transform unchanged), and every wave also does 1/8 of the staging work
//     (v4's stages A / B / C re-mapped to 512 threads: per PAIR of chunks one (channel, tile) transform, two patch items, six loads);
//   * the two waves of a SIMD run in antiphase ("ping-pong", MI355X_MICROARCH.md "Two waves per SIMD"): waves 0-3 run
//     [128 MFMAs of a chunk pair][staging], waves 4-7 [staging][128 MFMAs], one workgroup barrier per pair — while one wave of a
//     SIMD stages, its partner owns the matrix pipe; while both are in their MFMA blocks the pipe is saturated by construction
//     (2 x 4096 cycles of MFMA per 8192-cycle interval against ~5500 cycles of serial instruction stream per wave);
//   * per MFMA the staging instructions are HALF of v4's at Cout = 128 (a quarter at 256: two cout tiles instead of four), the
//     filter traffic per MFMA is unchanged (each filter word once per workgroup tile, L2 -> registers), HBM / L2 input traffic per
//     launch halves.
// Ring protocol (rings of four V slabs / patch buffers, as v4 PAIR). Interval I = chunks 2I, 2I + 1 of the workgroup's chunk stream:
//   M(I) reads V(2I), V(2I+1);   P(I) = { C: patches 2I+2, 2I+3 -> V(2I+2), V(2I+3);  B: raw -> patches 2I+4, 2I+5;  A: global loads of
//   chunks 2I+6, 2I+7 into the registers B just emptied }.   Barrier I ends interval I for all eight waves; inside an interval the order of
//   M and P is free (they touch disjoint ring slots), which is what lets the two halves run them in opposite order.
// Arithmetic and summation order are v4's: outputs are bit-identical to conv_wino4_kernel (tests/test_conv_winograd.py).
constexpr int W5RB = 4;          // B window of the MFMA block, in Winograd points
// INTER: behind which MFMA group (0..31 = chunk * 16 + Winograd point) of an interval each staging piece is placed
constexpr int W5S_CR = 0, W5S_CM = 3, W5S_B0 = 7, W5S_B1 = 10, W5S_A = 14, W5S_SHIFT = 16;
struct Wino5Raw { f32x4 a; float sc, sh; unsigned ok; };      // one item of one chunk (HALO / UP: a[0] only)

// HALO: this wave's staging item is a halo element (waves 5-7 of the non-UP kernel), else a float4 row piece (UP: one scalar of the
// source-resolution patch). TUNE bit 0: static s_setprio 1 for the second half (waves 4-7); bit 1: B window of 4 points instead of 8.
// INTER: no halves — every wave runs [MFMA block with the staging pieces placed between its MFMA groups] barrier: both waves of a SIMD
// always have MFMAs to issue, and whatever one of them waits for (an LDS round trip, a vector-memory issue) the other's MFMAs cover.
template <bool UP, bool HALO, int ACT, bool INTER, bool H2 = false>
__device__ __forceinline__ void wino5_wave(const WinoParams& p, float* ldsV, float* ldsP, const int tid, const int wave,
                                           const int b0, const int bs) {
  const bool yrole = !INTER && wave >= 4;     // second half: staging first, MFMA block second
  const int lane = tid & 63;
  const int li = lane & 15, k4 = lane >> 4;
  const int Ct = p.C1 + p.C2;
  const int planeS = p.Hs * p.Ws;
  const int nch = Ct / WCK;
  const int n_cblk = p.Cout >> 4;
  const int ntile = (p.nblk - b0 + bs - 1) / bs;
  const int total = ntile * nch;              // chunks of this workgroup's stream (a multiple of 4)
  const int npairs = total >> 1;
  // ---- staging item of this thread (one per chunk) ----------------------------------------------------------------------------
  int it_ch, it_row, it_col, it_pofs;
  const int dummy = WCK * WPH * WPP + (tid & 255);
  if (UP) {                                   // source-resolution patch 8 x 6 x 10 = 480 scalars
    const bool en = tid < 480;
    const int ec = en ? tid : 0;
    it_ch = ec / 60; it_row = (ec % 60) / 10; it_col = ec % 10;
    it_pofs = en ? ec : dummy;
  } else if (!HALO) {                         // waves 0-4: float4 piece f = tid of the 320
    const int row0 = tid >> 2, q0 = tid & 3;
    it_ch = row0 / WPH; it_row = row0 % WPH; it_col = 4 * q0;
    it_pofs = row0 * WPP + 1 + 4 * q0;
  } else {                                    // waves 5-7: the 160 halo elements (threads 480-511 write a private dummy word)
    const int hI = tid - 320;
    const bool en = hI < 160;
    const int hc = en ? hI : 0;
    const int hrow = hc >> 1, side = hc & 1;
    it_ch = hrow / WPH; it_row = hrow % WPH; it_col = side ? 16 : -1;
    it_pofs = en ? hrow * WPP + (side ? 17 : 0) : dummy;
  }
  // stage C: this thread's (chunk of the pair, channel, Winograd tile)
  const int cpar = tid >> 8;
  const int pc = (tid >> 5) & 7, ptile = tid & 31;
  const int tyy = ptile >> 3, txx = ptile & 7;
  const int wbase = UP ? pc * 60 + tyy * 10 + txx : pc * (WPH * WPP) + 2 * tyy * WPP + 2 * txx;
  const int vofs = pc * 32 + ptile;
  // ---- stage A cursor ---------------------------------------------------------------------------------------------------------
  int a_v = b0, a_ci = 0, a_left = total;
  int a_off = 0;
  unsigned a_ok = 0;
  const float *a_x1 = nullptr, *a_x2 = nullptr, *a_gs = nullptr, *a_gh = nullptr;
#if !defined(ADM_EMU)
  __amdgpu_buffer_rsrc_t a_rx1, a_rx2, a_rgs, a_rgh;
  int a_vo = 0;
  const int ch_vo = it_ch * 4;
#endif
  auto a_geometry = [&]() {
    const Wino3Tile t = wino5_tile(p, a_v);
    a_x1 = p.x1 + (long)t.n * p.x1_bs;
    a_x2 = p.x2 + (long)t.n * p.x2_bs - (long)p.C1 * planeS;
    a_gs = p.gn_scale + (long)t.n * p.gn_nstride;
    a_gh = p.gn_shift + (long)t.n * p.gn_nstride;
    const int sy = UP ? t.ty * 4 - 1 + it_row : t.ty * 8 - 1 + it_row;
    const int sx = UP ? t.tx * 8 - 1 + it_col : t.tx * 16 + it_col;
    const bool ok = sy >= 0 && sy < p.Hs && sx >= 0 && sx < p.Ws;
    a_off = it_ch * planeS + (ok ? sy * p.Ws + sx : 0);
    a_ok = ok ? 1u : 0u;
#if !defined(ADM_EMU)
    a_rx1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_x1), (short)0, 0x7fffffff, 0x00027000);
    a_rx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_x2), (short)0, 0x7fffffff, 0x00027000);
    a_rgs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_gs), (short)0, 0x7fffffff, 0x00027000);
    a_rgh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_gh), (short)0, 0x7fffffff, 0x00027000);
    a_vo = a_off * 4;
#endif
  };
  a_geometry();
  // global loads of the next PAIR of chunks of the stream (a pair never straddles tiles: chunk counts are multiples of 4); then advance,
  // saturating on the last pair (the loads stay unconditional, their data is never used)
  auto stage_a2 = [&](Wino5Raw& ra, Wino5Raw& rb_) {
    const int c0 = a_ci * WCK;
#if !defined(ADM_EMU)
    const __amdgpu_buffer_rsrc_t rx = c0 < p.C1 ? a_rx1 : a_rx2;
    const int so = c0 * planeS * 4, sg = c0 * 4;
    const int so1 = so + WCK * planeS * 4, sg1 = sg + WCK * 4;
    if (UP || HALO) {
      ra.a[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, a_vo, so, 0));
      rb_.a[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, a_vo, so1, 0));
    } else {
      ra.a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, a_vo, so, 0));
      rb_.a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, a_vo, so1, 0));
    }
    ra.sc = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rgs, ch_vo, sg, 0));
    ra.sh = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rgh, ch_vo, sg, 0));
    rb_.sc = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rgs, ch_vo, sg1, 0));
    rb_.sh = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rgh, ch_vo, sg1, 0));
#else
    const float* base = (c0 < p.C1 ? a_x1 : a_x2) + (long)c0 * planeS;
    if (UP || HALO) { ra.a[0] = base[a_off]; rb_.a[0] = base[a_off + (long)WCK * planeS]; }
    else {
      ra.a = *reinterpret_cast<const f32x4*>(base + a_off);
      rb_.a = *reinterpret_cast<const f32x4*>(base + a_off + (long)WCK * planeS);
    }
    ra.sc = a_gs[c0 + it_ch]; ra.sh = a_gh[c0 + it_ch];
    rb_.sc = a_gs[c0 + WCK + it_ch]; rb_.sh = a_gh[c0 + WCK + it_ch];
#endif
    ra.ok = a_ok; rb_.ok = a_ok;
    if (a_left > 2) {
      a_left -= 2;
      a_ci += 2;
      if (a_ci == nch) {
        ADM_SCHED_FENCE();
        a_ci = 0; a_v += bs;
        a_geometry();
      }
    }
  };
  constexpr bool act_on = ACT != 0;
  auto stage_b = [&](const Wino5Raw& r_, int g) {     // raw -> GroupNorm affine (+ SiLU) -> patch buffer g & 3; zero padding = zeroed affine
    float* P0 = ldsP + (g & 3) * W3PSLAB + it_pofs;
    const float sc = r_.ok ? r_.sc : 0.f, sh = r_.ok ? r_.sh : 0.f;
    if (UP || HALO) {
      const float v0 = r_.a[0] * sc + sh;
      P0[0] = act_on ? silu_w(v0) : v0;
      return;
    }
#if !defined(ADM_EMU)
    typedef float wf2 __attribute__((ext_vector_type(2)));
    auto act2 = [&](float x0, float x1, float* dst) __attribute__((always_inline)) {   // v4's packed activation: bit-identical to silu_w
      wf2 v = wf2{x0, x1} * sc + sh;
      if (act_on) {
        const wf2 t = v * -1.44269504088896340736f;
        wf2 e;
        e.x = __builtin_amdgcn_exp2f(t.x); e.y = __builtin_amdgcn_exp2f(t.y);
        const wf2 d = e + 1.0f;
        wf2 q;
        q.x = __builtin_amdgcn_rcpf(d.x); q.y = __builtin_amdgcn_rcpf(d.y);
        v = v * q;
      }
      dst[0] = v.x; dst[1] = v.y;
    };
    act2(r_.a[0], r_.a[1], P0); act2(r_.a[2], r_.a[3], P0 + 2);
#else
    ADM_UNROLL
    for (int k = 0; k < 4; ++k) { const float v0 = r_.a[k] * sc + sh; P0[k] = act_on ? silu_w(v0) : v0; }
#endif
  };
  // stage C in two parts, so that the window's LDS round trip runs under stage B's arithmetic
  auto stage_c_read = [&](int g, float (&d)[16]) {     // patch g & 3 -> this thread's 4x4 window
    const float* P = ldsP + (g & 3) * W3PSLAB + wbase;
    ADM_UNROLL
    for (int i = 0; i < 4; ++i)
      ADM_UNROLL
      for (int j = 0; j < 4; ++j) d[i * 4 + j] = UP ? P[((i + 1) >> 1) * 10 + ((j + 1) >> 1)] : P[i * WPP + j];
  };
  auto stage_c_math = [&](int g, const float (&d)[16]) {   // V = B^T d B -> V slab g & 3
    float* vdst = ldsV + (g & 3) * W3VSLAB + vofs;
#if !defined(ADM_EMU)
    typedef float wf2 __attribute__((ext_vector_type(2)));
    wf2 D[4][2], T[4][2];
    ADM_UNROLL
    for (int i = 0; i < 4; ++i) { D[i][0] = wf2{d[i * 4 + 0], d[i * 4 + 1]}; D[i][1] = wf2{d[i * 4 + 2], d[i * 4 + 3]}; }
    ADM_UNROLL
    for (int h2 = 0; h2 < 2; ++h2) {
      T[0][h2] = D[0][h2] - D[2][h2];
      T[1][h2] = D[1][h2] + D[2][h2];
      T[2][h2] = D[2][h2] - D[1][h2];
      T[3][h2] = D[1][h2] - D[3][h2];
    }
    ADM_UNROLL
    for (int i = 0; i < 4; ++i) {
      wf2 lo, hi;
      asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(lo) : "v"(T[i][0]), "v"(T[i][1]));
      asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(hi) : "v"(T[i][0]), "v"(T[i][1]));
      vdst[(i * 4 + 0) * (WCK * 32)] = lo.x;     // t0 - t2
      vdst[(i * 4 + 1) * (WCK * 32)] = lo.y;     // t1 + t2
      vdst[(i * 4 + 2) * (WCK * 32)] = hi.x;     // t2 - t1
      vdst[(i * 4 + 3) * (WCK * 32)] = hi.y;     // t1 - t3
    }
#else
    float t[4][4];
    ADM_UNROLL
    for (int j = 0; j < 4; ++j) {
      t[0][j] = d[0 * 4 + j] - d[2 * 4 + j];
      t[1][j] = d[1 * 4 + j] + d[2 * 4 + j];
      t[2][j] = d[2 * 4 + j] - d[1 * 4 + j];
      t[3][j] = d[1 * 4 + j] - d[3 * 4 + j];
    }
    ADM_UNROLL
    for (int i = 0; i < 4; ++i) {
      vdst[(i * 4 + 0) * (WCK * 32)] = t[i][0] - t[i][2];
      vdst[(i * 4 + 1) * (WCK * 32)] = t[i][1] + t[i][2];
      vdst[(i * 4 + 2) * (WCK * 32)] = t[i][2] - t[i][1];
      vdst[(i * 4 + 3) * (WCK * 32)] = t[i][1] - t[i][3];
    }
#endif
  };
  // ---- filter stream cursor (v4's: [chunk][cout block][q][ks][lane][4 points], this wave's block = m0 / 16 + wave) -------------
  int d_v = b0, d_ci = 0, d_left = total;
  const long chunk_stride = (long)n_cblk * W4ABLK;
  const float* d_src = p.wu + ((long)(wino5_tile(p, d_v).m0 >> 4) + wave) * W4ABLK + lane * 4;
  f32x4 a[4][2];
#define W5_LOAD_A(q)                                                                 \
  do {                                                                               \
    a[q][0] = *reinterpret_cast<const f32x4*>(d_src + (q) * 512);                    \
    a[q][1] = *reinterpret_cast<const f32x4*>(d_src + (q) * 512 + 256);              \
  } while (0)
  auto advance_a = [&]() {
    if (d_left > 1) {
      --d_left;
      d_src += chunk_stride;
      if (++d_ci == nch) {
        ADM_SCHED_FENCE();
        d_ci = 0; d_v += bs;
        d_src = p.wu + ((long)(wino5_tile(p, d_v).m0 >> 4) + wave) * W4ABLK + lane * 4;
      }
    }
  };
  // ---- prologue ------------------------------------------------------------------------------------------------------------------
  Wino5Raw r0, r1;
  r0.a = f32x4{0.f, 0.f, 0.f, 0.f}; r1.a = r0.a;
  int pg = 0;                                 // first chunk of the pair the next staging block transforms (stage C)
  stage_a2(r0, r1);                           // chunks 0, 1
  stage_b(r0, 0); stage_b(r1, 1);
  stage_a2(r0, r1);                           // chunks 2, 3
  W5_LOAD_A(0); W5_LOAD_A(1); W5_LOAD_A(2); W5_LOAD_A(3);  // filters of chunk 0
  advance_a();
  ADM_BARRIER_KEEP_VMEM(63);                               // patches 0, 1 complete

  const int vlane = k4 * 32 + 2 * li;
  constexpr int RB = W5RB;
  float2 rb[RB][2];
  auto read_group = [&](int slot, int gg, int xi) {
    const float* V = ldsV + (gg & 3) * W3VSLAB + vlane;
    rb[slot][0] = *reinterpret_cast<const float2*>(V + (xi * WCK) * 32);
    rb[slot][1] = *reinterpret_cast<const float2*>(V + (xi * WCK + 4) * 32);
  };
  f32x4 acc[16][2];
  // ONE loop body serves the prologue as well: iterations -2 and -1 have no MFMA block. Barrier / staging schedule per iteration `it`:
  //   first half  (waves 0-3): [M(it)] [epilogue] P            barrier      — P from it = -1 on (P#0 = V(0), V(1), patches 2, 3, loads 4, 5)
  //   second half (waves 4-7): [M(it)] barrier    [epilogue] P              — P from it = -2 on, i.e. one staging block AHEAD of the first half
  // Both halves execute the same barriers (from it = -1 on); between two of them M and P of either half touch disjoint ring slots.
  const long planeO = (long)p.Ho * p.Wo;
  int v = b0 - bs, ci = 0;                    // tile / chunk cursor of the MFMA stream (ci == nch: step to the next tile)
  ci = nch;
  Wino3Tile t = wino5_tile(p, b0);
  long obase = 0;
#if !defined(ADM_EMU)
  // Output stores and residual loads as raw buffer operations: resource = this wave's 16 cout rows of the tile's sample (SGPRs, made once
  // per tile), lane term = one 32-bit byte offset, cout row / pixel row = SGPR offsets. The 64-bit per-row VGPR addresses (eight pairs
  // that hipcc kept alive through the whole tile) are gone.
  __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc(p.out, (short)0, 0x7fffffff, 0x00027000), r_rs = o_rs;
  int o_vo = 0;
  const int plane_b = (int)planeO * 4, row_b = p.Wo * 4;
#endif
  f32x4 fr0 = {0.f, 0.f, 0.f, 0.f}, fr1 = fr0;
  float fb0 = 0.f, fb1 = 0.f;
  auto epilogue = [&]() {
      ADM_UNROLL
      for (int r = 0; r < 4; ++r) {
        f32x4 y0, y1;
        ADM_UNROLL
        for (int c = 0; c < 2; ++c) {
          float t0[4], t1[4];
          ADM_UNROLL
          for (int j = 0; j < 4; ++j) {
            t0[j] = acc[0 * 4 + j][c][r] + acc[1 * 4 + j][c][r] + acc[2 * 4 + j][c][r];
            t1[j] = acc[1 * 4 + j][c][r] - acc[2 * 4 + j][c][r] - acc[3 * 4 + j][c][r];
          }
          y0[2 * c] = t0[0] + t0[1] + t0[2];
          y0[2 * c + 1] = t0[1] - t0[2] - t0[3];
          y1[2 * c] = t1[0] + t1[1] + t1[2];
          y1[2 * c + 1] = t1[1] - t1[2] - t1[3];
        }
#if !defined(ADM_EMU)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y0), o_rs, o_vo, r * plane_b, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y1), o_rs, o_vo, r * plane_b + row_b, 0);
#else
        *reinterpret_cast<f32x4*>(p.out + obase + r * planeO) = y0;
        *reinterpret_cast<f32x4*>(p.out + obase + r * planeO + p.Wo) = y1;
#endif
        if (p.stats != nullptr) {
          float f1 = (y0[0] + y0[1]) + (y0[2] + y0[3]) + ((y1[0] + y1[1]) + (y1[2] + y1[3]));
          float f2 = (y0[0] * y0[0] + y0[1] * y0[1]) + (y0[2] * y0[2] + y0[3] * y0[3]) +
                     ((y1[0] * y1[0] + y1[1] * y1[1]) + (y1[2] * y1[2] + y1[3] * y1[3]));
          double s1 = (double)f1, s2 = (double)f2;
          ADM_UNROLL
          for (int m = 8; m >= 1; m >>= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }
          if (li == 0) {
            const int tiles = p.tiles_x * p.tiles_y;
#if !defined(ADM_EMU)
            // (a buffer store: SGPR base + 32-bit lane term — the 64-bit lane part of the address was a loop invariant hipcc spilled)
            typedef double wd2 __attribute__((ext_vector_type(2)));
            const int so = (((t.n * p.Cout + t.m0 + 16 * wave) * tiles + t.ty * p.tiles_x + t.tx)) * 16;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, wd2{s1, s2}),
                                                   __builtin_amdgcn_make_buffer_rsrc(p.stats, (short)0, 0x7fffffff, 0x00027000),
                                                   (4 * k4 + r) * tiles * 16, so, 0);
#else
            double* dst = p.stats + (((long)t.n * p.Cout + t.m0 + 16 * wave + 4 * k4 + r) * tiles + t.ty * p.tiles_x + t.tx) * 2;
            dst[0] = s1; dst[1] = s2;
#endif
          }
        }
        ADM_SCHED_FENCE();
      }
  };
  bool pend = false;                          // a finished tile waits for its inverse transform + stores
  for (int it = -2; it <= npairs; ++it) {     // (iteration npairs: nothing but the last tile's epilogue)
    // Both halves store a finished tile at the top of the NEXT iteration — behind the barrier that ended its last interval: in front of
    // it, the first half's ~4000 cycles of inverse transform and stores kept the second half waiting once per tile.
    if (pend) { epilogue(); pend = false; }
    if (it >= 0 && it < npairs) {
      if (ci == nch) {                        // next tile
        ADM_SCHED_FENCE();
        ci = 0; v += bs;
        t = wino5_tile(p, v);
        ADM_UNROLL
        for (int xi = 0; xi < 16; ++xi)
          ADM_UNROLL
          for (int c = 0; c < 2; ++c)
            ADM_UNROLL
            for (int r = 0; r < 4; ++r) acc[xi][c][r] = 0.f;
        const int oy = t.ty * 8 + 2 * (li >> 2), ox = t.tx * 16 + 4 * (li & 3);
#if !defined(ADM_EMU)
        const long tbase = ((long)t.n * p.Cout + t.m0 + 16 * wave) * planeO;       // wave-uniform
        o_rs = __builtin_amdgcn_make_buffer_rsrc(p.out + tbase, (short)0, 0x7fffffff, 0x00027000);
        if (p.residual != nullptr) r_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual) + tbase, (short)0, 0x7fffffff, 0x00027000);
        o_vo = (4 * k4 * (int)planeO + oy * p.Wo + ox) * 4;
#else
        obase = ((long)t.n * p.Cout + t.m0 + 16 * wave + 4 * k4) * planeO + (long)oy * p.Wo + ox;
#endif
      }
      const int g = 2 * it;
      // ---- M: the 128 MFMAs of chunks g, g + 1 -----------------------------------------------------------------------------------
      ADM_UNROLL
      for (int xi = 0; xi < RB; ++xi) read_group(xi, g, xi);
      // (a real two-trip loop, NOT unrolled: the loop-carried values pin the accumulators and the filter registers in place — unrolled,
      // hipcc renamed them across the two copies: 156 accumulator and 48 filter registers instead of 128 + 32, and spilled)
      float cd[16];                            // (INTER) stage C's window, between its read and its transform
      auto chunk = [&](const int c2) __attribute__((always_inline)) {
        const int cc = ci + c2;
        if (cc < 4) {          // wave-uniform: this chunk carries cout row r = cc of the bias / residual fold (v4)
          const int co = t.m0 + 16 * wave + 4 * k4 + cc;
          fb0 = p.bias[co];
          fb1 = p.chan_add[(long)t.n * p.chan_add_stride + co];
          if (p.residual != nullptr) {
#if !defined(ADM_EMU)
            fr0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rs, o_vo, cc * plane_b, 0));
            fr1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rs, o_vo, cc * plane_b + row_b, 0));
#else
            fr0 = *reinterpret_cast<const f32x4*>(p.residual + obase + cc * planeO);
            fr1 = *reinterpret_cast<const f32x4*>(p.residual + obase + cc * planeO + p.Wo);
#endif
          }
        }
        ADM_UNROLL
        for (int xi = 0; xi < 16; ++xi) {
          const int s = xi & (RB - 1), q = xi >> 2, e = xi & 3;
          ADM_UNROLL
          for (int ks = 0; ks < 2; ++ks) {
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][ks][e], rb[s][ks].x, acc[xi][0], 0, 0, 0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][ks][e], rb[s][ks].y, acc[xi][1], 0, 0, 0);
          }
          if (e == 3) {                        // group q consumed: its registers take the NEXT chunk's words
            // (measured and dropped: the pair's second refill issued from the staging block behind stage B, so that stage B's wait for its
            // raw activations no longer waits for these loads too — the block's counted waits then ran into the raw HBM loads queued behind
            // the refill: 2.92 -> 3.04 ms on 128 -> 128 @256^2)
            if (q == 0) W5_LOAD_A(0);
            if (q == 1) W5_LOAD_A(1);
            if (q == 2) W5_LOAD_A(2);
            if (q == 3) { W5_LOAD_A(3); advance_a(); }
          }
          // the window runs on into the next chunk; behind the pair's second chunk those are words of a slab that is being written
          // (never used: the next block primes its window afresh behind the barrier) — unconditional, so the body has no branch
          if (xi < 16 - RB) read_group(s, g + c2, xi + RB);
          else read_group(s, g + c2 + 1, xi - (16 - RB));
          ADM_SCHED_FENCE();
          if (INTER) {                         // staging pieces of P(it) between the MFMA groups (s is a compile-time constant here)
            // The two waves of a SIMD (w and w + 4) run the same stream and restart together behind every barrier; with the SAME
            // placement their staging pieces — and the stalls that come with them — would coincide. The second half places its pieces
            // W5S_SHIFT groups later (same order, so the register hand-overs between the pieces hold).
            const int sl = c2 * 16 + xi;
#define W5_AT(S) (sl == (S) + (H2 ? W5S_SHIFT : 0))
            if (W5_AT(W5S_CR)) stage_c_read(pg + cpar, cd);
            if (W5_AT(W5S_CM)) stage_c_math(pg + cpar, cd);
            if (W5_AT(W5S_B0)) stage_b(r0, pg + 2);
            if (W5_AT(W5S_B1)) stage_b(r1, pg + 3);
            if (W5_AT(W5S_A)) { stage_a2(r0, r1); pg += 2; }
#undef W5_AT
            ADM_SCHED_FENCE();
          }
        }
        if (cc < 4) {
          // v4's fold (cout row r = cc gets bias + per-sample term + residual through the four corner points), written without a branch
          // per row: a four-way branch on cc made hipcc copy the eight accumulators through 32 spare registers (phi copies) and spill.
          // Row r adds its value, the other rows add 0 * value (x + 0 = x bit for bit, except -0 -> +0; a non-finite residual value
          // would reach the lane's other three rows as NaN — such a tensor is lost either way).
          const float bsum = fb0 + fb1;
          const float v00 = bsum + fr0[0], v01 = bsum + fr0[2], v30 = -(bsum + fr0[1]), v31 = -(bsum + fr0[3]);
          const float vc0 = -(bsum + fr1[0]), vc1 = -(bsum + fr1[2]), vf0 = bsum + fr1[1], vf1 = bsum + fr1[3];
          ADM_UNROLL
          for (int R = 0; R < 4; ++R) {
            // one fma per (accumulator, row) with a wave-uniform 1.0 / 0.0 factor: fma(1, v, acc) = acc + v exactly; fma(0, v, acc) = acc
            // (for finite v; acc - x = acc + (-x) exactly, so v4's subtractions are additions of the negated value)
            const float on = cc == R ? 1.f : 0.f;
            acc[0][0][R] = __builtin_fmaf(on, v00, acc[0][0][R]);   acc[0][1][R] = __builtin_fmaf(on, v01, acc[0][1][R]);
            acc[3][0][R] = __builtin_fmaf(on, v30, acc[3][0][R]);   acc[3][1][R] = __builtin_fmaf(on, v31, acc[3][1][R]);
            acc[12][0][R] = __builtin_fmaf(on, vc0, acc[12][0][R]); acc[12][1][R] = __builtin_fmaf(on, vc1, acc[12][1][R]);
            acc[15][0][R] = __builtin_fmaf(on, vf0, acc[15][0][R]); acc[15][1][R] = __builtin_fmaf(on, vf1, acc[15][1][R]);
          }
        }
      
      };
      if constexpr (INTER) { chunk(0); chunk(1); }          // straight-line: the slot numbers below are compile-time constants
      else {
        // (a real two-trip loop, NOT unrolled: see above)
        _Pragma("clang loop unroll(disable)")
        for (int c2 = 0; c2 < 2; ++c2) chunk(c2);
      }
      ci += 2;
      pend = ci == nch;
    }
    if (yrole && it >= -1 && it < npairs) ADM_BARRIER_KEEP_VMEM(63);
    if (INTER ? it == -1 : (yrole ? it + 1 < npairs : (it >= -1 && it < npairs))) {  // P: C(pg, pg + 1), B(pg + 2, pg + 3), A(the next pair of the stream)
      // Stage C's window read first (its LDS round trip runs under what follows). The second half stages right behind its MFMA block, whose
      // last filter refills are still in flight and sit in front of stage B's raw activations in the in-order counter: it finishes stage C
      // before stage B, the first half (a whole MFMA block between its loads and this point is not the issue there) the other way round.
      float cd[16];
      stage_c_read(pg + cpar, cd);
      if (yrole) {
        stage_c_math(pg + cpar, cd);
        ADM_SCHED_FENCE();
      }
      stage_b(r0, pg + 2);
      ADM_SCHED_FENCE();
      stage_b(r1, pg + 3);
      ADM_SCHED_FENCE();
      if (!yrole) {
        stage_c_math(pg + cpar, cd);
        ADM_SCHED_FENCE();
      }
      stage_a2(r0, r1);
      pg += 2;
    }
    if (!yrole && it >= -1 && it < npairs) ADM_BARRIER_KEEP_VMEM(63);
  }
#undef W5_LOAD_A
}

template <bool UP, int ACT, bool INTER>
__global__ void __launch_bounds__(512) conv_wino5_kernel(const WinoParams p) {
  ADM_DYN_SMEM(float, smem);
  float* ldsV = smem;
  float* ldsP = smem + 4 * W3VSLAB;
  const int tid = threadIdx.x;
  const int wave = ADM_UNIFORM(tid >> 6);     // an SGPR: role tests and the barrier placement become scalar branches
#if !defined(ADM_EMU)
  if (!INTER && wave >= 4 && (p.tune & 1)) __builtin_amdgcn_s_setprio(1);
#endif
  if (INTER && wave >= 4) {                   // (INTER: the second half is its own instantiation — its staging pieces sit at other places)
    if (!UP && wave >= 5) wino5_wave<UP, true, ACT, INTER, INTER>(p, ldsV, ldsP, tid, wave, (int)blockIdx.x, (int)gridDim.x);
    else wino5_wave<UP, false, ACT, INTER, INTER>(p, ldsV, ldsP, tid, wave, (int)blockIdx.x, (int)gridDim.x);
    return;
  }
  if (!UP && wave >= 5) wino5_wave<UP, true, ACT, INTER>(p, ldsV, ldsP, tid, wave, (int)blockIdx.x, (int)gridDim.x);
  else wino5_wave<UP, false, ACT, INTER>(p, ldsV, ldsP, tid, wave, (int)blockIdx.x, (int)gridDim.x);
}

// ---- launchers ------------------------------------------------------------------------------------------------------------------------
static bool wino_f2_lds_ok() {               // 91 KiB of dynamic LDS for the eight instantiations, asked once per device
#if !defined(ADM_EMU)
  static int state[16] = {};                 // 0 unknown, 1 granted, -1 refused
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  int& s = state[conv_dev_slot() & 15];
  if (s == 0) {
    const int by = (int)(sizeof(float) * W4LDS_PAIR);
    bool ok = true;
    ok &= hipFuncSetAttribute((const void*)conv_wino4_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
    ok &= hipFuncSetAttribute((const void*)conv_wino4_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
    ok &= hipFuncSetAttribute((const void*)conv_wino4_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
    ok &= hipFuncSetAttribute((const void*)conv_wino4_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
    ok &= hipFuncSetAttribute((const void*)conv_wino5_kernel<true, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
    ok &= hipFuncSetAttribute((const void*)conv_wino5_kernel<true, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
    ok &= hipFuncSetAttribute((const void*)conv_wino5_kernel<false, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
    ok &= hipFuncSetAttribute((const void*)conv_wino5_kernel<false, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
    ok &= hipFuncSetAttribute((const void*)conv_wino5_kernel<true, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
    ok &= hipFuncSetAttribute((const void*)conv_wino5_kernel<true, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
    ok &= hipFuncSetAttribute((const void*)conv_wino5_kernel<false, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
    ok &= hipFuncSetAttribute((const void*)conv_wino5_kernel<false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    s = ok ? 1 : -1;
  }
  return s > 0;
#else
  return true;
#endif
}

int launch_wino4(const WinoParams& p, bool up, bool act, int grid, hipStream_t st) {
  ADM_REQUIRE(wino_f2_lds_ok(), "conv_winograd: the runtime refused 91 KiB of dynamic LDS for conv_wino4_kernel");
  const size_t need = sizeof(float) * W4LDS_PAIR;
  if (up) {
    if (act) ADM_LAUNCH((conv_wino4_kernel<true, 1>), dim3(grid), dim3(512), need, st, p);
    else ADM_LAUNCH((conv_wino4_kernel<true, 0>), dim3(grid), dim3(512), need, st, p);
  } else {
    if (act) ADM_LAUNCH((conv_wino4_kernel<false, 1>), dim3(grid), dim3(512), need, st, p);
    else ADM_LAUNCH((conv_wino4_kernel<false, 0>), dim3(grid), dim3(512), need, st, p);
  }
  return ADM_CHECK_LAUNCH();
}

int launch_wino5(const WinoParams& p, bool up, bool act, int grid, bool two_halves, hipStream_t st) {
  ADM_REQUIRE(wino_f2_lds_ok(), "conv_winograd: the runtime refused 91 KiB of dynamic LDS for conv_wino5_kernel");
  const size_t need = sizeof(float) * W4LDS_PAIR;
  if (!two_halves) {                          // the interleaved schedule (the default)
    if (up) {
      if (act) ADM_LAUNCH((conv_wino5_kernel<true, 1, true>), dim3(grid), dim3(512), need, st, p);
      else ADM_LAUNCH((conv_wino5_kernel<true, 0, true>), dim3(grid), dim3(512), need, st, p);
    } else {
      if (act) ADM_LAUNCH((conv_wino5_kernel<false, 1, true>), dim3(grid), dim3(512), need, st, p);
      else ADM_LAUNCH((conv_wino5_kernel<false, 0, true>), dim3(grid), dim3(512), need, st, p);
    }
    return ADM_CHECK_LAUNCH();
  }
  if (up) {
    if (act) ADM_LAUNCH((conv_wino5_kernel<true, 1, false>), dim3(grid), dim3(512), need, st, p);
    else ADM_LAUNCH((conv_wino5_kernel<true, 0, false>), dim3(grid), dim3(512), need, st, p);
  } else {
    if (act) ADM_LAUNCH((conv_wino5_kernel<false, 1, false>), dim3(grid), dim3(512), need, st, p);
    else ADM_LAUNCH((conv_wino5_kernel<false, 0, false>), dim3(grid), dim3(512), need, st, p);
  }
  return ADM_CHECK_LAUNCH();
}

}  // namespace adm
