// k_conv_bf16.hip — mixed-precision 3x3 stride-1 convolution for `--mixed_precision bf16` training
// (scripts/train_unet.py:391-401 hands the choice to accelerate, which wraps the step in torch.autocast: Conv2d runs on
// bf16 operands with fp32 accumulation).  Design: master weights, activations and gradients stay fp32 in HBM; only the
// MFMA operands are bf16 — the activated input (GroupNorm affine + SiLU applied on the load path) is rounded to bf16 as
// it is written into the LDS patch, the filters are repacked to bf16 once per optimizer step, v_mfma_f32_32x32x16_bf16
// accumulates in fp32 and the epilogue (bias, time-embedding bias, residual) is fp32.  Compared with autocast this keeps
// more precision (no bf16 rounding of stored activations / weight gradients); the arithmetic rate is the bf16 MFMA rate.
//
//   forward / data-gradient : conv_bf16_kernel   (data-gradient = the same kernel on dy with transposed, flipped filters)
//   weight gradient         : conv_wgrad_bf16_kernel (k = pixels; taps from funnel-shifted LDS rows)
//
// Workgroup tile of the forward kernel: 128 output channels x 16x16 output pixels, 4 waves as 2 (64 couts) x 2 (8 rows);
// each wave holds 2 x 4 accumulator tiles of 32x32 (128 registers) so that one A fragment serves 4 MFMAs and one B
// fragment 2 (the bf16 MFMA retires a 32x32x16 tile in 8 passes: operand delivery, not arithmetic, is the limit).
// K loop: 16 input channels per step = one MFMA k-step.  The 18x18 input patch of the step is staged as
// [2 channel groups of 8][324 pixels] x 16 B — a lane's B fragment (8 channels of one pixel) is one ds_read_b128 and the
// 32 lanes of a half-wave read 32 consecutive 16-B slots.  A fragments come straight from the packed filters
// [tap][Cin/8][Cout][8] (512 contiguous bytes per half-wave, L2-resident).  Two workgroups per CU: one stages while the
// other feeds the MFMA pipe.
#include "adm_kernels.h"

namespace adm {

struct Bf16ConvParams {
  const float* x1; const float* x2; int C1, C2;
  int N, Hs, Ws, Hi, Wi;
  const float* gn_scale; const float* gn_shift; int gn_nstride;
  const u32x4* wb; const float* bias; int Cout;
  const float* chan_add; int chan_add_stride;
  const float* residual; float* out;
  int tiles_x, tiles_y, n_ct, nblk;
  long x1_bs, x2_bs;
  int zins;   // UP kernels: 1 = zero-insertion x2 (data gradient of a stride-2 convolution) instead of nearest x2
  unsigned long long* prof;   // developer aid (ADM_BF16_PROF=1): per-phase cycle counters, else NULL
};

constexpr int BPW = 18, BPP = BPW * BPW;   // input patch of a 16x16 output tile

#if defined(ADM_EMU)
#define B16_CLK() 0ull
#else
#define B16_CLK() ((unsigned long long)__builtin_readcyclecounter())
#endif
// per-phase cycle accounting of one wave (PROF kernels only): slot += time since the previous lap
#define B16_LAP(slot) do { if (PROF) { const unsigned long long tn_ = B16_CLK(); pr[slot] += tn_ - tq; tq = tn_; } } while (0)

__device__ __forceinline__ float silu_b(float v) { return v * ADM_RCP(1.0f + __expf(-v)); }

struct Bf16Stage { float v[3][8]; };                 // raw fp32 prefetch of one 16-channel chunk (three 8-channel rounds)
template <int NA> struct Bf16Filt { u32x4 a[9][NA]; };  // the wave's A fragments of one chunk: 9 taps x NA cout sub-tiles

// WIDE = false: waves as 2 (64 couts) x 2 (8 pixel rows), 2 x 4 accumulator tiles, the two waves of a cout half fetch the
// same filter fragments.  WIDE = true: waves as 4 (32 couts) x 1, 1 x 8 accumulator tiles: every filter fragment is
// fetched once per workgroup (half the L2 requests, 36 registers less) at one LDS read per MFMA instead of one per two.
template <bool UP, bool ACT, bool WIDE, bool F16 = false, bool PROF = false>
__global__ void __launch_bounds__(256, 1) conv_bf16_kernel(const Bf16ConvParams p) {
  constexpr int NA = WIDE ? 1 : 2, NP = WIDE ? 8 : 4;
  unsigned long long pr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = 0;
  const unsigned long long t_start = B16_CLK();
  if (PROF) tq = t_start;
  // One workgroup per CU (up to 512 registers per lane) so that everything that comes from memory is requested a full
  // chunk (filters, L2) or two chunks (input patch, HBM) before it is used: the first version requested the next tap's
  // filters 8 MFMAs ahead and queued them behind the patch loads of the in-order vector-memory counter — 17k cycles per
  // chunk against 2.3k of MFMA work (profiles/r01_train_bf16_v1_kernel_stats.md).
  ADM_DYN_SMEM(u32x4, lds);                 // [2 buffers][2 channel groups][324 pixels] + GroupNorm rows [2][Ct] floats
  const int tid = threadIdx.x, lane = tid & 63, wave = ADM_UNIFORM(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = WIDE ? wave : (wave & 1), wn = WIDE ? 0 : (wave >> 1);
  int lid;
  {   // consecutive logical tiles (all cout tiles of a pixel tile, then the neighbouring pixel tile) share an XCD's L2
    const int b = blockIdx.x, q = p.nblk >> 3, r = p.nblk & 7, xcd = b & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int ct = lid % p.n_ct; lid /= p.n_ct;
  const int tx = lid % p.tiles_x; lid /= p.tiles_x;
  const int ty = lid % p.tiles_y, n = lid / p.tiles_y;
  const int m0 = ct * 128 + wm * (32 * NA);
  const int Ct = p.C1 + p.C2, KG = Ct >> 3;
  const int planeS = p.Hs * p.Ws;

  // staging plan (tile-invariant): every wave converts three rounds of 64 pixels x 8 channels per chunk —
  // rounds 0/1: pixel 64*wave + lane of channel group 0/1; round 2: the 68 left-over pixels (wave&1 picks the 64-block,
  // wave>>1 the group).  The channel group of a round is wave-uniform, so the GroupNorm rows are scalar loads.
  int soff[2];
  ADM_UNROLL
  for (int r = 0; r < 2; ++r) {
    const int q = r == 0 ? 64 * wave + lane : 256 + 64 * (wave & 1) + lane;
    const int ly = q / BPW, lx = q - ly * BPW;
    const int gy = ty * 16 + ly - 1, gx = tx * 16 + lx - 1;
    bool ok = (q < BPP) & (gy >= 0) & (gy < p.Hi) & (gx >= 0) & (gx < p.Wi);
    if (UP) ok = ok & !(p.zins && ((gy | gx) & 1));          // zero insertion: odd rows / columns are the inserted zeros
    soff[r] = ok ? (UP ? (gy >> 1) * p.Ws + (gx >> 1) : gy * p.Ws + gx) : -1;
  }
  float* gnS = reinterpret_cast<float*>(lds + 4 * BPP);     // [Ct] scale, then [Ct] shift of image n
  float* gnB = gnS + Ct;
  for (int c = tid; c < Ct; c += 256) {
    gnS[c] = p.gn_scale[(long)n * p.gn_nstride + c];
    gnB[c] = p.gn_shift[(long)n * p.gn_nstride + c];
  }
  const int q2 = 256 + 64 * (wave & 1) + lane;
  const int kg2 = wave >> 1;
  const bool has2 = q2 < BPP;
  const int n_chunks = Ct >> 4;

  // every global address is a wave-uniform 64-bit base (SGPR pair, scalar arithmetic) plus a per-lane 32-bit offset that
  // never changes: per-load 64-bit VGPR addresses (24 patch + 18 filter loads, hoisted out of the loop) cost 130 registers
  const unsigned so0 = soff[0] < 0 ? 0u : (unsigned)soff[0], so1 = soff[1] < 0 ? 0u : (unsigned)soff[1];
  auto issue = [&](Bf16Stage& s, int ch) __attribute__((always_inline)) {      // raw fp32 loads of chunk ch
    const int c0 = 16 * (ch < n_chunks ? ch : n_chunks - 1);     // past the end: harmless re-request, no branch
    const float* xc = c0 < p.C1 ? p.x1 + (long)n * p.x1_bs + (long)c0 * planeS
                                : p.x2 + (long)n * p.x2_bs + (long)(c0 - p.C1) * planeS;
    ADM_UNROLL
    for (int r = 0; r < 3; ++r) {
      const int kg = r < 2 ? r : kg2;
      const unsigned so = r < 2 ? so0 : so1;
      ADM_UNROLL
      for (int e = 0; e < 8; ++e) {
        const float* pe = xc + (long)(kg * 8 + e) * planeS;       // uniform
        s.v[r][e] = pe[so];
      }
    }
  };
  auto stash = [&](const Bf16Stage& s, u32x4* buf, int ch) __attribute__((always_inline)) {   // affine + SiLU -> bf16 -> LDS
    if (ch >= n_chunks) return;
    const int c0 = 16 * ch;
    ADM_UNROLL
    for (int r = 0; r < 3; ++r) {
      const int kg = r < 2 ? r : kg2;
      const int so = soff[r < 2 ? 0 : 1];
      // GroupNorm rows of this image from LDS (copied once at kernel start): read straight from global memory they are
      // VECTOR loads (the compiler cannot prove the rows are not aliased by `out`, so no s_load), and waiting for them —
      // the youngest entries of the in-order vector-memory counter — drained the whole filter/patch prefetch (vmcnt(0))
      const float4 s0 = *reinterpret_cast<const float4*>(gnS + c0 + kg * 8), s1 = *reinterpret_cast<const float4*>(gnS + c0 + kg * 8 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(gnB + c0 + kg * 8), b1 = *reinterpret_cast<const float4*>(gnB + c0 + kg * 8 + 4);
      const float gs[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      const float gb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float v[8];
      ADM_UNROLL
      for (int e = 0; e < 8; ++e) {
        float t = s.v[r][e] * gs[e] + gb[e];
        if (ACT) t = silu_b(t);
        v[e] = so < 0 ? 0.f : t;                                   // zero padding applies to the activated tensor
      }
      u32x4 w;
      w[0] = ADM_PK16(F16, v[0], v[1]); w[1] = ADM_PK16(F16, v[2], v[3]);
      w[2] = ADM_PK16(F16, v[4], v[5]); w[3] = ADM_PK16(F16, v[6], v[7]);
      if (r < 2) buf[kg * BPP + 64 * wave + lane] = w;
      else if (has2) buf[kg * BPP + q2] = w;
      ADM_SCHED_FENCE();      // one round at a time: 24 interleaved SiLU chains cost ~70 temporaries
    }
  };
  const unsigned wlane = (unsigned)(h * p.Cout + l31);             // per-lane part of the filter address (16-B units)
  // filters: ONE register set, refilled in place — as soon as the MFMAs of tap t are issued, the same registers receive
  // tap t of the next chunk, so every filter fragment is requested a full chunk (72 MFMAs) before its use at a cost of
  // 72 registers instead of 144
  auto fetch_tap = [&](Bf16Filt<NA>& f, int ch, int t) __attribute__((always_inline)) {
    const u32x4* wt = p.wb + m0 + ((long)(2 * ch) + (long)t * KG) * p.Cout;     // uniform
    ADM_UNROLL
    for (int a = 0; a < NA; ++a) f.a[t][a] = (wt + 32 * a)[wlane];
  };
  f32x16 acc[NA][NP];
  ADM_UNROLL
  for (int a = 0; a < NA; ++a)
    ADM_UNROLL
    for (int t = 0; t < NP; ++t)
      ADM_UNROLL
      for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.f;

  // B fragment of pixel tile t (2 rows x 16 columns), tap (dy, dx): LDS slot (8 wn + 2 t + (l31 >> 4) + dy) * 18 + (l31 & 15) + dx
  const int bbase = h * BPP + (8 * wn + (l31 >> 4)) * BPW + (l31 & 15);
  Bf16Filt<NA> F;
  auto mfma_chunk = [&](const u32x4* cur, int ch) __attribute__((always_inline)) {
    // B fragments one tap ahead, fenced: left alone, the scheduler hoists all 36 LDS reads of the chunk above the first
    // MFMA (144 registers) and the kernel spills
    const int chn = ch + 1 < n_chunks ? ch + 1 : ch;   // past the end: re-request the last chunk (no branch in the tap loop)
    u32x4 Bc[NP], Bn[NP];
    ADM_UNROLL
    for (int pt = 0; pt < NP; ++pt) Bc[pt] = cur[bbase + (2 * pt) * BPW];
    ADM_UNROLL
    for (int t = 0; t < 9; ++t) {
      if (t < 8) {
        ADM_UNROLL
        for (int pt = 0; pt < NP; ++pt) Bn[pt] = cur[bbase + (2 * pt + (t + 1) / 3) * BPW + ((t + 1) % 3)];
      }
      ADM_SCHED_FENCE();
      ADM_UNROLL
      for (int pt = 0; pt < NP; ++pt) {
        ADM_UNROLL
        for (int a = 0; a < NA; ++a) acc[a][pt] = ADM_MFMA16(F16, F.a[t][a], Bc[pt], acc[a][pt]);
      }
      ADM_SCHED_FENCE();
      fetch_tap(F, chn, t);
      ADM_UNROLL
      for (int pt = 0; pt < NP; ++pt) Bc[pt] = Bn[pt];
    }
  };

  // Pipeline: while chunk c multiplies out of LDS buffer c & 1, the filters of chunk c + 1 (rolling, above) and the raw
  // patch of chunk c + 2 are in flight, and the patch of chunk c + 1 (requested one iteration earlier) is converted into
  // the other buffer.  Two patch register sets alternate (X/Y): the loop body is written for an even/odd pair, and the
  // launcher only takes even chunk counts (Cin % 32 == 0) — an exit between the halves made the register allocator keep
  // two copies of the 128 accumulators and spill into the loop.
  Bf16Stage X, Y;
  u32x4* buf0 = lds;
  u32x4* buf1 = lds + 2 * BPP;
  // Order matters: vector memory returns in order, so a filter fragment (L2) requested after a patch load (HBM) cannot
  // arrive before it. The patch loads are therefore issued at the END of a chunk — after that chunk's rolling filter
  // requests, which the next chunk's MFMAs wait for — and only the conversion (stash), two chunks later, waits for them.
  // (Issued ahead of the MFMA phase they cost 9.5k cycles per chunk: every tap-0 wait inherited the HBM latency.)
  issue(X, 0);
  ADM_UNROLL
  for (int t = 0; t < 9; ++t) fetch_tap(F, 0, t);
  issue(Y, 1);
  __syncthreads();                        // GroupNorm rows are in LDS
  stash(X, buf0, 0);
  issue(X, 2);
  __syncthreads();
  B16_LAP(1);                             // prologue
  for (int ch = 0; ch < n_chunks; ch += 2) {
    mfma_chunk(buf0, ch);                 // even chunk: LDS buffer 0; Y holds chunk ch + 1, X (in flight) ch + 2
    B16_LAP(2);
    stash(Y, buf1, ch + 1);
    B16_LAP(3);
    issue(Y, ch + 3);
    B16_LAP(4);
    __syncthreads();
    B16_LAP(5);
    mfma_chunk(buf1, ch + 1);             // odd chunk: LDS buffer 1; X holds chunk ch + 2, Y (in flight) ch + 3
    B16_LAP(2);
    stash(X, buf0, ch + 2);
    B16_LAP(3);
    issue(X, ch + 4);
    B16_LAP(4);
    __syncthreads();
    B16_LAP(5);
  }

  // epilogue: D row = output channel, column = pixel; fp32 bias + per-(n, channel) term + residual
  const long planeO = (long)p.Hi * p.Wi;
  ADM_UNROLL
  for (int a = 0; a < NA; ++a) {
    float bv[16];
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int co = m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
      bv[r] = p.bias[co] + p.chan_add[(long)n * p.chan_add_stride + co];
    }
    ADM_UNROLL
    for (int pt = 0; pt < NP; ++pt) {
      const int oy = ty * 16 + 8 * wn + 2 * pt + (l31 >> 4), ox = tx * 16 + (l31 & 15);
      const long pix = (long)oy * p.Wi + ox;
      float rv[16];
      if (p.residual) {
        ADM_UNROLL
        for (int r = 0; r < 16; ++r) {
          const int co = m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
          rv[r] = p.residual[((long)n * p.Cout + co) * planeO + pix];
        }
      }
      ADM_UNROLL
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = acc[a][pt][r] + bv[r];
        if (p.residual) v += rv[r];
        p.out[((long)n * p.Cout + co) * planeO + pix] = v;
      }
    }
  }
  if (PROF) {
    B16_LAP(6);                           // epilogue (issue only: the stores drain after the wave has left)
    pr[0] = B16_CLK() - t_start;
    if (lane == 0)
      for (int i = 0; i < 8; ++i) atomicAdd(p.prof + i, pr[i]);
  }
}

// ---------------------------------------------------------------- weight gradient
// dW[co][ci][tap] = sum over pixels of dy[co][pixel] * a[ci][pixel + tap], a = upsample(act(GroupNorm(x))).
// GEMM with k = pixels: one MFMA k-step is a row of 16 output pixels (lane half h owns pixels 8h..8h+7); D = 32 couts x 32
// cins per tap, 9 taps = 144 accumulator registers per wave.  Workgroup = 128 couts (4 waves) x 32 cins x a range of
// 16x4-pixel tiles (split-K over pixel tiles, partial sums reduced by wgrad_reduce_kernel as in the fp32 path).
//   LDS dy tile : [row 4][half 2][cout 128] x 16 B (8 pixels)  — the A fragment is one ds_read_b128
//   LDS patch   : [row 6][cin 32] x 48 B (18 pixels used, padded to 24) — per patch row a lane reads pixels 8h..8h+9
//                 (b128 + b32) and forms the three horizontal taps in registers: dx = 0 as read, dx = 2 the same dwords
//                 shifted by one register, dx = 1 by four v_alignbit_b32 — 3.75 LDS dwords per MFMA.
struct Bf16WgradParams {
  const float* x1; const float* x2; int C1, C2;
  const float* dy; int Cout;
  int N, Hs, Ws, Hi, Wi;
  const float* gn_scale; const float* gn_shift; int gn_nstride;
  float* part;
  int tiles_x, tiles_y, n_ptiles, n_ct, n_chunks, split, tiles_per_block, nblk;
  long x1_bs, x2_bs;
  unsigned mTX, mTXY;       // floor(2^32 / d) + 1 for d = tiles_x, tiles_x * tiles_y (0: d == 1 or tile count >= 2^16)
  unsigned long long* prof; // developer aid (ADM_BF16_PROF=1): per-phase cycle counters, else NULL
};

__device__ __forceinline__ int bdiv(int n, int d, unsigned magic) {   // n / d; exact via umulhi for n, d < 2^16
  return magic ? (int)(((unsigned long long)(unsigned)n * magic) >> 32) : n / d;
}

// raw fp32 prefetch of one 16x4-pixel tile: 4 dy items (8 pixels each), 7 patch pixel pairs, their in-bounds bits, image
struct Bf16WgStage { float4 d[8]; float xa[7], xb[7]; unsigned ok; int n; };

template <bool UP, bool ACT, bool F16 = false, bool PROF = false>
__global__ void __launch_bounds__(256, 1) conv_wgrad_bf16_kernel(const Bf16WgradParams p) {
  unsigned long long pr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = 0;
  const unsigned long long t_start = B16_CLK();
  if (PROF) tq = t_start;
  // One workgroup per CU, everything from memory requested two tiles ahead (two register sets P/Q), converted into the
  // LDS buffer the MFMAs are not reading, one barrier per tile.  The first version loaded, converted and multiplied tile
  // by tile: 13k cycles per tile against 1.2k of MFMA work (profiles/r01_train_bf16_v1_kernel_stats.md).
  constexpr int XROW = 12;                       // dwords per (row, cin) of the patch
  constexpr int DLD = 130;                       // fragment stride of a (row, half) line of 128 couts: 130 makes the 16 lanes of a
                                                 // b128 store group (2 couts x 8 (row, half)) hit 16 distinct 16-B slots (128: 8-way)
  constexpr int DFR = 8 * DLD;                   // dy fragments (u32x4) per buffer
  constexpr int XDW = 6 * 32 * XROW;             // patch dwords per buffer
  constexpr int BUF4 = DFR + XDW / 4;            // u32x4 per buffer
  ADM_DYN_SMEM(u32x4, lds4);
  unsigned* dummy = reinterpret_cast<unsigned*>(lds4 + 2 * BUF4);          // 256 dwords: disabled lanes store here
  float* gnS = reinterpret_cast<float*>(dummy + 256);                       // [N][32] scale, then [N][32] shift
  float* gnB = gnS + p.N * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = ADM_UNIFORM(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  int lid;
  {   // the n_chunks workgroups that read the same dy tiles are neighbours on one XCD
    const int b = blockIdx.x, q = p.nblk >> 3, r = p.nblk & 7, xcd = b & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int chunk = lid % p.n_chunks; lid /= p.n_chunks;
  const int ct = lid % p.n_ct, sp = lid / p.n_ct;
  const int m0 = ct * 128, c0 = chunk * 32;
  const int Ct = p.C1 + p.C2;
  const int planeS = p.Hs * p.Ws;
  const long planeO = (long)p.Hi * p.Wi;
  const float* xsrc = c0 < p.C1 ? p.x1 + (long)c0 * planeS : p.x2 + (long)(c0 - p.C1) * planeS;
  const long xbs = c0 < p.C1 ? p.x1_bs : p.x2_bs;
  const int t_begin = sp * p.tiles_per_block;
  int t_end = t_begin + p.tiles_per_block;
  if (t_end > p.n_ptiles) t_end = p.n_ptiles;

  for (int i = tid; i < p.N * 32; i += 256) {          // GroupNorm rows of this channel chunk, every image
    const long gi = (long)(i >> 5) * p.gn_nstride + c0 + (i & 31);
    gnS[i] = p.gn_scale[gi];
    gnB[i] = p.gn_shift[gi];
  }

  f32x16 acc[9];
  ADM_UNROLL
  for (int t = 0; t < 9; ++t)
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // tile-invariant staging roles.  dy: item = (cout, row, quarter of the 16-pixel row), 8 per thread, one float4 each: the four
  // lanes of a (cout, row) read its 64 bytes with ONE instruction, a wave-load touches 16 row segments. (Round 1: item = (cout,
  // row, half) as two float4 per lane — both instructions of an item touched the same 32 segments; the vector L1 charges per
  // segment an instruction touches, profiles/r02_bf16_training.md.)
  // patch: item = (row, cin, pixel pair), 1728 items in 7 rounds (the last one partial).
  unsigned dyo[8]; int ldsd[8];                   // ldsd: dword index of the item's 8-byte half inside the dy image
  ADM_UNROLL
  for (int j = 0; j < 8; ++j) {
    const int id = tid + 256 * j;
    const int qd = id & 3, r = (id >> 2) & 3, co = id >> 4;
    dyo[j] = (unsigned)(co * (int)planeO + r * p.Wi + 4 * qd);
    ldsd[j] = ((r * 2 + (qd >> 1)) * DLD + co) * 4 + 2 * (qd & 1);
  }
  int xcin[7], xrow[7], xq[7], ldsx[7];
  ADM_UNROLL
  for (int j = 0; j < 7; ++j) {
    const int id = tid + 256 * j;
    const int rc = id / 9;
    xq[j] = id - rc * 9; xrow[j] = rc >> 5; xcin[j] = rc & 31;
    ldsx[j] = xrow[j] < 6 ? (xrow[j] * 32 + xcin[j]) * XROW + xq[j] : -1;       // -1: past the end -> dummy word
  }

  auto load_tile = [&](Bf16WgStage& s, int pt_raw) __attribute__((always_inline)) {
    const int pt = pt_raw < t_end ? pt_raw : t_end - 1;        // past the end: re-request the last tile (no branch)
    const int n = bdiv(pt, p.tiles_x * p.tiles_y, p.mTXY);
    const int rem = pt - n * (p.tiles_x * p.tiles_y);
    const int ty = bdiv(rem, p.tiles_x, p.mTX), tx = rem - ty * p.tiles_x;
    s.n = n;
    const float* dbase = p.dy + ((long)n * p.Cout + m0) * planeO + (long)(ty * 4) * p.Wi + tx * 16;   // uniform
    ADM_UNROLL
    for (int j = 0; j < 8; ++j) s.d[j] = *reinterpret_cast<const float4*>(dbase + dyo[j]);
    const float* xt = xsrc + (long)n * xbs;                                                            // uniform
    const int gy0 = ty * 4 - 1, gx0 = tx * 16 - 1;
    unsigned ok = 0;
    ADM_UNROLL
    for (int j = 0; j < 7; ++j) {
      const int gy = gy0 + xrow[j], gx = gx0 + 2 * xq[j];
      const bool oky = (gy >= 0) & (gy < p.Hi) & (ldsx[j] >= 0);
      const bool ok0 = oky & (gx >= 0) & (gx < p.Wi), ok1 = oky & (gx + 1 < p.Wi);      // gx + 1 >= 0 always
      const int rowoff = xcin[j] * planeS + (UP ? (gy >> 1) : gy) * p.Ws;
      const unsigned o0 = ok0 ? (unsigned)(rowoff + (UP ? (gx >> 1) : gx)) : 0u;
      const unsigned o1 = ok1 ? (unsigned)(rowoff + (UP ? ((gx + 1) >> 1) : gx + 1)) : 0u;
      s.xa[j] = xt[o0]; s.xb[j] = xt[o1];
      ok |= ((ok0 ? 1u : 0u) | (ok1 ? 2u : 0u)) << (2 * j);
    }
    s.ok = ok;
  };
  auto stash_tile = [&](const Bf16WgStage& s, u32x4* buf) __attribute__((always_inline)) {
    unsigned* bufX = reinterpret_cast<unsigned*>(buf + DFR);
    unsigned* bufD = reinterpret_cast<unsigned*>(buf);
    ADM_UNROLL
    for (int j = 0; j < 8; ++j) {
      const float4 v = s.d[j];
      const unsigned long long w = (unsigned long long)ADM_PK16(F16, v.x, v.y) | ((unsigned long long)ADM_PK16(F16, v.z, v.w) << 32);
      *reinterpret_cast<unsigned long long*>(bufD + ldsd[j]) = w;              // ds_write_b64
    }
    ADM_UNROLL
    for (int j = 0; j < 7; ++j) {
      const float sc = gnS[s.n * 32 + xcin[j]], sh = gnB[s.n * 32 + xcin[j]];
      float a = s.xa[j] * sc + sh, b = s.xb[j] * sc + sh;
      if (ACT) { a = silu_b(a); b = silu_b(b); }
      a = (s.ok >> (2 * j)) & 1u ? a : 0.f;          // zero padding applies to the activated tensor
      b = (s.ok >> (2 * j)) & 2u ? b : 0.f;
      unsigned* dst = ldsx[j] >= 0 ? bufX + ldsx[j] : dummy + tid;
      *dst = ADM_PK16(F16, a, b);
    }
  };
  auto mfma_tile = [&](const u32x4* buf, bool valid) __attribute__((always_inline)) {
    const unsigned* bufX = reinterpret_cast<const unsigned*>(buf + DFR);
    ADM_UNROLL
    for (int r = 0; r < 4; ++r) {
      u32x4 A = buf[(r * 2 + h) * DLD + 32 * wave + l31];
      if (!valid) { A[0] = 0u; A[1] = 0u; A[2] = 0u; A[3] = 0u; }       // tile past the end of an odd range: contributes zero
      ADM_UNROLL
      for (int dy3 = 0; dy3 < 3; ++dy3) {
        const unsigned* xr = bufX + ((r + dy3) * 32 + l31) * XROW + 4 * h;
        const u32x4 d = *reinterpret_cast<const u32x4*>(xr);
        const unsigned d4 = xr[4];
        u32x4 s1, s2;
        s1[0] = ADM_ALIGNBIT(d[1], d[0], 16); s1[1] = ADM_ALIGNBIT(d[2], d[1], 16);
        s1[2] = ADM_ALIGNBIT(d[3], d[2], 16); s1[3] = ADM_ALIGNBIT(d4, d[3], 16);
        s2[0] = d[1]; s2[1] = d[2]; s2[2] = d[3]; s2[3] = d4;
        acc[dy3 * 3 + 0] = ADM_MFMA16(F16, A, d, acc[dy3 * 3 + 0]);
        acc[dy3 * 3 + 1] = ADM_MFMA16(F16, A, s1, acc[dy3 * 3 + 1]);
        acc[dy3 * 3 + 2] = ADM_MFMA16(F16, A, s2, acc[dy3 * 3 + 2]);
      }
    }
  };

  Bf16WgStage P, Q;
  u32x4* buf0 = lds4;
  u32x4* buf1 = lds4 + BUF4;
  load_tile(P, t_begin);
  load_tile(Q, t_begin + 1);
  __syncthreads();                         // GroupNorm rows are in LDS
  stash_tile(P, buf0);
  load_tile(P, t_begin + 2);
  __syncthreads();
  B16_LAP(1);                              // prologue
  for (int pt = t_begin; pt < t_end; pt += 2) {
    mfma_tile(buf0, true);                 // tile pt; Q holds pt + 1, P (in flight) pt + 2
    B16_LAP(2);
    stash_tile(Q, buf1);
    B16_LAP(3);
    load_tile(Q, pt + 3);
    B16_LAP(4);
    __syncthreads();
    B16_LAP(5);
    mfma_tile(buf1, pt + 1 < t_end);       // tile pt + 1; P holds pt + 2, Q (in flight) pt + 3
    B16_LAP(2);
    stash_tile(P, buf0);
    B16_LAP(3);
    load_tile(P, pt + 4);
    B16_LAP(4);
    __syncthreads();
    B16_LAP(5);
  }
  float* out = p.part + (long)sp * p.Cout * Ct * 9;
  ADM_UNROLL
  for (int t = 0; t < 9; ++t) {
    const int cc = c0 + l31;
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int co = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      out[((long)t * p.Cout + co) * Ct + cc] = acc[t][r];      // slab layout [tap][cout][cin] (k_conv_wgrad.hip: wgrad_reduce_kernel)
    }
  }
  if (PROF) {
    B16_LAP(6);
    pr[0] = B16_CLK() - t_start;
    pr[7] = (unsigned long long)(((t_end - t_begin) + 1) & ~1);      // tile slots this workgroup ran
    if (lane == 0)
      for (int i = 0; i < 8; ++i) atomicAdd(p.prof + i, pr[i]);
  }
}

bool conv_op16_f16();
static int g_bf16_mode = -1;   // -1: ADM_CONV_BF16 from the environment (default 0 = fp32 everywhere; 2 = 1x1 convs too)
bool conv_bf16_enabled();

// ---------------------------------------------------------------- filters: fp32 (Cout,Cin,3,3) -> bf16 [tap][Cin/8][Cout][8]
template <bool F16>
__device__ __forceinline__ void pack_bf16_body(const float* __restrict__ w, unsigned* __restrict__ wb, int Cout, int Cin,
                                               int transposed, int taps, long first, long step) {
  // one thread per bf16 PAIR of the packed tensor. transposed: the data-gradient filters (roles of Cout/Cin swapped, taps
  // flipped), i.e. packed "Cout" = Cin and packed "Cin" = Cout.
  const int Co = transposed ? Cin : Cout, Ci = transposed ? Cout : Cin;
  const long total = (long)taps * Ci * Co / 2;       // taps = 9 (3x3) or 1 (1x1: [Cin/8][Cout][8])
  for (long i = first; i < total; i += step) {
    const int e2 = (int)(i & 3);
    long r = i >> 2;
    const int co = (int)(r % Co); r /= Co;
    const int kg = (int)(r % (Ci >> 3));
    const int t = (int)(r / (Ci >> 3));
    const int ci = kg * 8 + 2 * e2;
    float a, b;
    if (!transposed) {
      a = w[((long)co * Cin + ci) * taps + t];
      b = w[((long)co * Cin + ci + 1) * taps + t];
    } else {
      a = w[((long)ci * Cin + co) * taps + (taps - 1 - t)];
      b = w[((long)(ci + 1) * Cin + co) * taps + (taps - 1 - t)];
    }
    wb[i] = ADM_PK16(F16, a, b);
  }
}
template <bool F16>
__global__ void pack_bf16_weight_kernel(const float* __restrict__ w, unsigned* __restrict__ wb, int Cout, int Cin,
                                        int transposed, int taps) {
  pack_bf16_body<F16>(w, wb, Cout, Cin, transposed, taps, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}
// The same image, one (64 couts x 32 cins) tile at a time through LDS: the gather above reads 8 bytes per thread at a 36-byte stride
// (and every source line once per tap) — 1.09 ms per optimizer step for the 113.7 M-parameter model's two images per layer, 1.25 TB/s.
// Here a workgroup reads 64 rows of 32 x taps CONTIGUOUS floats (float4), rounds them into LDS as [tap][cin][cout] halfwords (rows
// of 72: 144 bytes, so that 16-byte reads of consecutive cins fall into different banks) and writes runs of 64 (plain) / 32
// (transposed) consecutive 16-byte units. Same rounding as ADM_PK16 element by element: bit-identical images.
constexpr int PT_CO = 64, PT_CI = 32, PT_PAD = 72;
template <bool F16>
__device__ __forceinline__ void pack_bf16_tile(const float* __restrict__ w, u32x4* __restrict__ wb, int Cout, int Cin, int transposed,
                                               int taps, int tile, unsigned short* lds) {
  const int n_ci_t = Cin / PT_CI;
  const int co0 = (tile / n_ci_t) * PT_CO, ci0 = (tile % n_ci_t) * PT_CI;
  const int tid = threadIdx.x;
  const int row4 = PT_CI * taps / 4;                  // float4 per cout row of the tile: 72 (3x3) / 8 (1x1)
  for (int idx = tid; idx < PT_CO * row4; idx += 256) {
    const int co = idx / row4, f4 = idx - co * row4;
    const float4 v = *reinterpret_cast<const float4*>(w + ((long)(co0 + co) * Cin + ci0) * taps + 4 * f4);
    const float e[4] = {v.x, v.y, v.z, v.w};
    ADM_UNROLL
    for (int k = 0; k < 4; ++k) {
      const int f = 4 * f4 + k, ci = f / taps, t = f - ci * taps;
      lds[(t * PT_CI + ci) * PT_PAD + co] = (unsigned short)(ADM_PK16(F16, e[k], 0.f) & 0xffffu);
    }
  }
  __syncthreads();
  if (!transposed) {       // unit (tap, cin group kg, cout): 8 cins of one cout
    for (int u = tid; u < taps * 4 * 64; u += 256) {
      const int co = u & 63, kg = (u >> 6) & 3, t = u >> 8;
      unsigned h[8];
      ADM_UNROLL
      for (int e = 0; e < 8; ++e) h[e] = lds[(t * PT_CI + kg * 8 + e) * PT_PAD + co];
      u32x4 q;
      q[0] = h[0] | (h[1] << 16); q[1] = h[2] | (h[3] << 16); q[2] = h[4] | (h[5] << 16); q[3] = h[6] | (h[7] << 16);
      wb[((long)t * (Cin >> 3) + (ci0 >> 3) + kg) * Cout + co0 + co] = q;
    }
  } else {                 // data-gradient filters: unit (flipped tap, cout group kg, cin): 8 couts of one cin
    for (int u = tid; u < taps * 8 * 32; u += 256) {
      const int ci = u & 31, kg = (u >> 5) & 7, t = u >> 8;
      const u32x4 q = *reinterpret_cast<const u32x4*>(lds + (t * PT_CI + ci) * PT_PAD + kg * 8);
      wb[((long)(taps - 1 - t) * (Cout >> 3) + (co0 >> 3) + kg) * Cin + ci0 + ci] = q;
    }
  }
  __syncthreads();
}
// blockIdx.y = item of a device table (Net::refresh_weights after an optimizer step); flag = transposed
template <bool F16>
__global__ void __launch_bounds__(256) pack_bf16_batch_kernel(const PackItem* __restrict__ items) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[9 * PT_CI * PT_PAD];
  const PackItem it = items[blockIdx.y];
  const int taps = it.ks * it.ks;
  if (it.Cout % PT_CO == 0 && it.Cin % PT_CI == 0 && (reinterpret_cast<uintptr_t>(it.src) & 15) == 0 && (taps == 9 || taps == 1)) {
    const int n_tiles = (it.Cout / PT_CO) * (it.Cin / PT_CI);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x)
      pack_bf16_tile<F16>(it.src, reinterpret_cast<u32x4*>(it.dst), it.Cout, it.Cin, it.flag, taps, tile, lds);
    return;
  }
  pack_bf16_body<F16>(it.src, (unsigned*)it.dst, it.Cout, it.Cin, it.flag, taps, (long)blockIdx.x * blockDim.x + threadIdx.x,
                      (long)gridDim.x * blockDim.x);
}
int launch_pack_bf16_batch(const PackItem* items_dev, int n, hipStream_t st) {
  if (n <= 0) return 0;
  if (conv_op16_f16()) ADM_LAUNCH(pack_bf16_batch_kernel<true>, dim3(64, (unsigned)n), dim3(256), 0, st, items_dev);
  else ADM_LAUNCH(pack_bf16_batch_kernel<false>, dim3(64, (unsigned)n), dim3(256), 0, st, items_dev);
  return ADM_CHECK_LAUNCH();
}

template <bool F16>
__global__ void __launch_bounds__(256) pack_bf16_tiles_kernel(const float* __restrict__ w, u32x4* __restrict__ wb, int Cout, int Cin,
                                                              int transposed, int taps, int n_tiles) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[9 * PT_CI * PT_PAD];
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) pack_bf16_tile<F16>(w, wb, Cout, Cin, transposed, taps, tile, lds);
}
int launch_pack_bf16_weight(const float* w, void* wb, int Cout, int Cin, int transposed, hipStream_t st, int ks) {
  ADM_REQUIRE(Cout % 8 == 0 && Cin % 8 == 0, "pack_bf16_weight: channel counts must be multiples of 8");
  ADM_REQUIRE(ks == 3 || ks == 1, "pack_bf16_weight: ks must be 1 or 3");
  const int taps = ks * ks;
  const long total = (long)taps * Cin * Cout / 2;
  if (Cout % PT_CO == 0 && Cin % PT_CI == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {     // tiles through LDS (see pack_bf16_tile)
    const int n_tiles = (Cout / PT_CO) * (Cin / PT_CI);
    const int g = n_tiles < 2048 ? n_tiles : 2048;
    if (conv_op16_f16()) ADM_LAUNCH(pack_bf16_tiles_kernel<true>, dim3(g), dim3(256), 0, st, w, (u32x4*)wb, Cout, Cin, transposed, taps, n_tiles);
    else ADM_LAUNCH(pack_bf16_tiles_kernel<false>, dim3(g), dim3(256), 0, st, w, (u32x4*)wb, Cout, Cin, transposed, taps, n_tiles);
    return ADM_CHECK_LAUNCH();
  }
  if (conv_op16_f16())
    ADM_LAUNCH(pack_bf16_weight_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, (unsigned*)wb, Cout, Cin,
               transposed, taps);
  else
    ADM_LAUNCH(pack_bf16_weight_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, (unsigned*)wb, Cout, Cin,
               transposed, taps);
  return ADM_CHECK_LAUNCH();
}
int conv_bf16_mode() {
  conv_bf16_enabled();
  return g_bf16_mode;
}

// ---------------------------------------------------------------- dispatch
static int g_op16_f16 = 0;     // operand format of the 16-bit kernels: 0 = bf16, 1 = IEEE binary16 (`--mixed_precision fp16`)
void set_conv_op16_f16(int v) { g_op16_f16 = v != 0; }
bool conv_op16_f16() { return g_op16_f16 != 0; }
void set_conv_bf16(int m) { g_bf16_mode = m; }
bool conv_bf16_enabled() {
  if (g_bf16_mode < 0) { const char* e = getenv("ADM_CONV_BF16"); g_bf16_mode = e ? atoi(e) : 0; }
  return g_bf16_mode != 0;
}

// 3x3 stride 1 "same", output a multiple of 16x16, an even number of 16-channel chunks none of which straddles the
// concat seam, Cout % 128.
bool conv_bf16_eligible(const adm_conv_args& a) {
  if (a.ks != 3 || a.stride != 1 || a.pad_lo != 1 || a.w_bstride != 0 || a.bf16_packed == nullptr) return false;
  // up = 2 (zero insertion: the data gradient of a stride-2 convolution) at level 2
  if (a.up > (conv_bf16_mode() >= 2 ? 2 : 1)) return false;
  const int C2 = a.x2 ? a.C2 : 0;
  const int Hi = a.up ? 2 * a.H : a.H, Wi = a.up ? 2 * a.W : a.W;
  return Wi % 16 == 0 && Hi % 16 == 0 && (a.C1 + C2) % 32 == 0 && a.C1 % 16 == 0 && a.Cout % 128 == 0 &&
         (a.gn_scale != nullptr || !a.act);
}

int launch_conv_bf16(const adm_conv_args& a, hipStream_t st) {
  Bf16ConvParams p;
  const int C2 = a.x2 ? a.C2 : 0, Ct = a.C1 + C2;
  p.x1 = a.x1; p.x2 = a.x2; p.C1 = a.C1; p.C2 = C2;
  p.N = a.N; p.Hs = a.H; p.Ws = a.W;
  p.Hi = a.up ? 2 * a.H : a.H; p.Wi = a.up ? 2 * a.W : a.W;
  p.gn_scale = a.gn_scale; p.gn_shift = a.gn_shift; p.gn_nstride = Ct;
  if (p.gn_scale == nullptr) {   // no GroupNorm on the load path: identity affine rows
    p.gn_scale = conv_const_ones(Ct); p.gn_shift = conv_zero_bias(Ct); p.gn_nstride = 0;
  }
  p.wb = reinterpret_cast<const u32x4*>(a.bf16_packed);
  p.bias = a.bias ? a.bias : conv_zero_bias(a.Cout); p.Cout = a.Cout;
  p.chan_add = a.chan_add; p.chan_add_stride = a.chan_add_stride;
  if (p.chan_add == nullptr) { p.chan_add = conv_zero_bias(a.Cout); p.chan_add_stride = 0; }
  ADM_REQUIRE(p.gn_scale && p.gn_shift && p.bias && p.chan_add, "conv_bf16: constant buffers");
  p.residual = a.residual; p.out = a.out; p.zins = a.up == 2;
  p.tiles_x = p.Wi / 16; p.tiles_y = p.Hi / 16; p.n_ct = a.Cout / 128;
  p.nblk = p.tiles_x * p.tiles_y * a.N * p.n_ct;
  p.x1_bs = a.x1_bstride ? a.x1_bstride : (long)a.C1 * a.H * a.W;
  p.x2_bs = a.x2_bstride ? a.x2_bstride : (long)C2 * a.H * a.W;
  const size_t smem = sizeof(u32x4) * 2 * 2 * BPP + sizeof(float) * 2 * Ct;
  ADM_REQUIRE(smem <= 64 * 1024, "conv_bf16: too many input channels for the LDS GroupNorm rows");
  set_last_conv_variant(5000 + 316);
  p.prof = nullptr;
#if !defined(ADM_EMU)
  static const bool want_prof = getenv("ADM_BF16_PROF") != nullptr;
  if (want_prof && !a.up && a.act && !conv_op16_f16()) {   // developer aid: per-phase cycle accounting, printed after the launch (synchronous)
    static unsigned long long* dprof = [] { void* q = nullptr; (void)hipMalloc(&q, 8 * sizeof(unsigned long long)); return (unsigned long long*)q; }();
    (void)hipMemsetAsync(dprof, 0, 8 * sizeof(unsigned long long), st);
    p.prof = dprof;
    ADM_LAUNCH((conv_bf16_kernel<false, true, false, false, true>), dim3(p.nblk), dim3(256), smem, st, p);
    unsigned long long h[8];
    (void)hipMemcpyAsync(h, dprof, sizeof(h), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    const double w = 4.0 * p.nblk, nch = (double)(a.C1 + C2) / 16;
    fprintf(stderr, "[bf16 prof] %d->%d @%dx%d N=%d: per wave and workgroup: total %.0f | prologue %.0f | per chunk: mfma %.0f stash %.0f "
            "issue %.0f barrier %.0f | epilogue %.0f cycles (%d chunks; 72 MFMAs = 2304)\n", a.C1 + C2, a.Cout, p.Hi, p.Wi, a.N,
            h[0] / w, h[1] / w, h[2] / w / nch, h[3] / w / nch, h[4] / w / nch, h[5] / w / nch, h[6] / w, (int)nch);
    return ADM_CHECK_LAUNCH();
  }
#endif
  // WIDE = true (waves 4 x 1) measured 640 vs 634 us on 128->128 @256^2: not instantiated
#define ADM_BF16_LAUNCH(UP_, ACT_)                                                                          \
  do {                                                                                                      \
    if (conv_op16_f16()) ADM_LAUNCH((conv_bf16_kernel<UP_, ACT_, false, true>), dim3(p.nblk), dim3(256), smem, st, p);   \
    else ADM_LAUNCH((conv_bf16_kernel<UP_, ACT_, false, false>), dim3(p.nblk), dim3(256), smem, st, p);                  \
  } while (0)
  if (a.up) {
    if (a.act) ADM_BF16_LAUNCH(true, true);
    else ADM_BF16_LAUNCH(true, false);
  } else {
    if (a.act) ADM_BF16_LAUNCH(false, true);
    else ADM_BF16_LAUNCH(false, false);
  }
#undef ADM_BF16_LAUNCH
  return ADM_CHECK_LAUNCH();
}

// 3x3 stride 1 "same", output a multiple of 16x4, 32-channel chunks inside one source tensor, full 128-cout tiles.
bool conv_wgrad_bf16_eligible(const adm_conv_args& a) {
  if (a.ks != 3 || a.stride != 1 || a.pad_lo != 1 || a.up > 1) return false;
  const int C2 = a.x2 ? a.C2 : 0;
  const int Hi = a.up ? 2 * a.H : a.H, Wi = a.up ? 2 * a.W : a.W;
  return Wi % 16 == 0 && Hi % 4 == 0 && (a.C1 + C2) % 32 == 0 && a.C1 % 32 == 0 && a.Cout % 128 == 0 &&
         (a.gn_scale != nullptr || !a.act);
}

// Partial sums into `workspace` ([split][Cout*Cin*9], the layout and split of conv_wgrad_workspace); the caller
// (launch_conv_wgrad) runs the reduction.
int launch_conv_wgrad_bf16(const adm_conv_args& a, const float* dy, float* dW, int accumulate, float* workspace, int split,
                           hipStream_t st) {
  (void)dW; (void)accumulate;
  Bf16WgradParams p;
  const int C2 = a.x2 ? a.C2 : 0, Ct = a.C1 + C2;
  p.x1 = a.x1; p.x2 = a.x2; p.C1 = a.C1; p.C2 = C2; p.dy = dy; p.Cout = a.Cout;
  p.N = a.N; p.Hs = a.H; p.Ws = a.W;
  p.Hi = a.up ? 2 * a.H : a.H; p.Wi = a.up ? 2 * a.W : a.W;
  ADM_REQUIRE((reinterpret_cast<uintptr_t>(dy) & 15) == 0, "conv_wgrad_bf16: dy must be 16-byte aligned");
  p.gn_scale = a.gn_scale; p.gn_shift = a.gn_shift; p.gn_nstride = Ct;
  if (p.gn_scale == nullptr) { p.gn_scale = conv_const_ones(Ct); p.gn_shift = conv_zero_bias(Ct); p.gn_nstride = 0; }
  ADM_REQUIRE(p.gn_scale && p.gn_shift, "conv_wgrad_bf16: constant buffers");
  p.part = workspace;
  p.tiles_x = p.Wi / 16; p.tiles_y = p.Hi / 4;
  p.n_ptiles = p.tiles_x * p.tiles_y * a.N;
  p.n_ct = a.Cout / 128; p.n_chunks = Ct / 32;
  p.tiles_per_block = ceil_div(p.n_ptiles, split);
  p.split = ceil_div(p.n_ptiles, p.tiles_per_block);
  ADM_REQUIRE(p.split == split, "conv_wgrad_bf16: split must leave no empty workgroup");
  p.nblk = p.n_ct * p.n_chunks * p.split;
  p.x1_bs = a.x1_bstride ? a.x1_bstride : (long)a.C1 * a.H * a.W;
  p.x2_bs = a.x2_bstride ? a.x2_bstride : (long)C2 * a.H * a.W;
  auto magic = [&](long d) { return (d <= 1 || p.n_ptiles >= 65536) ? 0u : (unsigned)((1ULL << 32) / (unsigned long long)d + 1ULL); };
  p.mTX = magic(p.tiles_x); p.mTXY = magic((long)p.tiles_x * p.tiles_y);
  const size_t smem = 2 * (sizeof(u32x4) * 8 * 130 + sizeof(unsigned) * 6 * 32 * 12) + sizeof(unsigned) * 256 +
                      sizeof(float) * 64 * (size_t)a.N;
  ADM_REQUIRE(smem <= 64 * 1024, "conv_wgrad_bf16: batch too large for the LDS GroupNorm rows");
  p.prof = nullptr;
#if !defined(ADM_EMU)
  static const bool want_prof = getenv("ADM_BF16_PROF") != nullptr;
  if (want_prof && !a.up && a.act && !conv_op16_f16()) {   // developer aid: per-phase cycle accounting, printed after the launch (synchronous)
    static unsigned long long* dprof = [] { void* q = nullptr; (void)hipMalloc(&q, 8 * sizeof(unsigned long long)); return (unsigned long long*)q; }();
    (void)hipMemsetAsync(dprof, 0, 8 * sizeof(unsigned long long), st);
    p.prof = dprof;
    ADM_LAUNCH((conv_wgrad_bf16_kernel<false, true, false, true>), dim3(p.nblk), dim3(256), smem, st, p);
    unsigned long long h[8];
    (void)hipMemcpyAsync(h, dprof, sizeof(h), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    const double w = 4.0 * p.nblk, nt = (double)h[7] / w;
    fprintf(stderr, "[bf16 wgrad prof] %d->%d @%dx%d N=%d split=%d: per wave and workgroup: total %.0f | prologue %.0f | per tile: mfma %.0f "
            "stash %.0f load %.0f barrier %.0f | epilogue %.0f cycles (%.1f tiles; 36 MFMAs = 1152)\n", Ct, a.Cout, p.Hi, p.Wi, a.N, split,
            h[0] / w, h[1] / w, h[2] / w / nt, h[3] / w / nt, h[4] / w / nt, h[5] / w / nt, h[6] / w, nt);
    return ADM_CHECK_LAUNCH();
  }
#endif
#define ADM_WG16_LAUNCH(UP_, ACT_)                                                                             \
  do {                                                                                                         \
    if (conv_op16_f16()) ADM_LAUNCH((conv_wgrad_bf16_kernel<UP_, ACT_, true>), dim3(p.nblk), dim3(256), smem, st, p);  \
    else ADM_LAUNCH((conv_wgrad_bf16_kernel<UP_, ACT_, false>), dim3(p.nblk), dim3(256), smem, st, p);                 \
  } while (0)
  if (a.up) {
    if (a.act) ADM_WG16_LAUNCH(true, true);
    else ADM_WG16_LAUNCH(true, false);
  } else {
    if (a.act) ADM_WG16_LAUNCH(false, true);
    else ADM_WG16_LAUNCH(false, false);
  }
  return ADM_CHECK_LAUNCH();
}

}  // namespace adm
