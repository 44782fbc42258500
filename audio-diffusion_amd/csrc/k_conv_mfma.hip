// k_conv_mfma.hip — fused 2-D convolution as an implicit GEMM on the exact-f32 matrix core
// (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain, 157 TF peak = the f32 vector peak; guide §3).
//
// Replaces every nn.Conv2d the UNet forward reaches (SURVEY.md §2.2 / §8(a) U2-U5,U7): ResnetBlock2D
// conv1/conv2/conv_shortcut, Downsample2D (stride 2), Upsample2D (nearest x2 folded into the load),
// attention q/k/v/out projections (1x1), with these fusions so an activation is read once and written once:
//   load path : virtual channel concat (two base pointers), nearest x2 upsample by index, zero padding,
//               GroupNorm affine (per-(n,c) scale/shift from k_groupnorm) + SiLU;
//   epilogue  : + bias[co] + time-embedding bias[n][co] + residual[n][co][y][x].
//
// GEMM view: D[co][pixel] = sum_{c,tap} W[co][c][tap] * X[c][pixel+tap]; M = Cout, N = pixels, K = Cin*ks*ks.
// Workgroup = 4 waves, tile BM x 128 pixels (BM = 128/64/32), K in chunks of 8 input channels staged in LDS:
//   ldsX [8][NI][IH][IW]  the haloed (and already activated) input patch — NCHW rows, so a wave's 32 pixels are
//                         consecutive LDS words (conflict-free ds_read_b32 up to the row seam);
//   ldsW [8*ks*ks][BM]    weights pre-packed [Cin][tap][Cout] so a wave's 32 couts are consecutive words.
// MFMA operands: A = weights (lane l: co = l&31, k = l>>5), B = pixels (lane l: pixel = l&31, k = l>>5); the
// two k of one instruction are the two channels (2cp, 2cp+1) of the same tap. Output fragment: col = pixel,
// row = cout, so each store instruction writes 2 couts x 32 consecutive pixels (64-B..128-B segments, NCHW).
// Pixel tile = NI images x TH x TW with TW = min(Wo,16), TH = min(Ho,8): 128 pixels at every UNet level
// (256^2 ... 1x1). Algorithmic bytes per launch: 4*(N*Cin*Hs*Ws + N*Cout*Ho*Wo [+ residual]) + 4*Cout*Cin*ks^2.
#include <cstdlib>

#include <atomic>
#include <map>
#include <mutex>
#include <vector>

#include "adm_kernels.h"

namespace adm {

constexpr int CK = 8;      // input channels per K chunk
constexpr int MAXQ = 5;    // max ldsX elements per thread per channel plane (CS <= 1280)

struct ConvParams {
  const float* x1; const float* x2; int C1, C2;
  int N, Hs, Ws, Hi, Wi, Ho, Wo;
  int up, pad_lo;
  const float* gn_scale; const float* gn_shift; int act;
  const float* wp; const float* bias; int Cout;
  const float* chan_add; int chan_add_stride;
  const float* residual; float* out;
  int lTW, lTH, tiles_x, tiles_y, n_ct, IH, IW, CS, nblk;
  long x1_bs, x2_bs, wp_bs;  // batch strides (elements) of x1/x2 (channel-slice views) and of per-sample weights (0: shared)
  int tap_mask;                // split-K instantiations of the generic kernel: bit t set = tap t can meet a pixel inside the image (a 1-pixel-high plane only needs the middle row of taps)
  int single;                  // the call's single-sample rule (adm_conv_args.single_sample / the option): partition choices only
  int ksplit; long part_stride; // split-K instantiations: S workgroups per tile, each writes its partial sums to out + s * part_stride
};

__device__ __forceinline__ float silu_f(float v) { return v * ADM_RCP(1.0f + __expf(-v)); }


// Epilogue for one 32x32 accumulator tile: batch all loads (bias, time-embedding bias, residual) before use.
// The nullable-pointer conditions are template flags: a per-element "load or not" branch makes hipcc wait vmcnt(0)
// per element (guide §5 trap (c)) and serialises 64 dependent round trips per lane.
template <bool HAS_CHAN, bool HAS_RES>
__device__ __forceinline__ void store_tile(const f32x16& acc, const ConvParams& p, int co_base, int h, int n, long pix,
                                           long planeO) {
  float bv[16], cv[16], rv[16];
  ADM_UNROLL
  for (int r = 0; r < 16; ++r) {
    const int co = co_base + (r & 3) + 8 * (r >> 2) + 4 * h;
    bv[r] = p.bias[co];
    if (HAS_CHAN) cv[r] = p.chan_add[(long)n * p.chan_add_stride + co];
    if (HAS_RES) rv[r] = p.residual[((long)n * p.Cout + co) * planeO + pix];
  }
  ADM_UNROLL
  for (int r = 0; r < 16; ++r) {
    const int co = co_base + (r & 3) + 8 * (r >> 2) + 4 * h;
    float v = acc[r] + bv[r];
    if (HAS_CHAN) v += cv[r];
    if (HAS_RES) v += rv[r];
    p.out[((long)n * p.Cout + co) * planeO + pix] = v;
  }
}

template <int TM, int TN, bool HAS_CHAN, bool HAS_RES>
__device__ __forceinline__ void epilogue(const f32x16 (&acc)[TM][TN], const ConvParams& p, int m_wave, int wn, int l31,
                                         int h, int TW, int TH, int tx, int ty, int n0) {
  const long planeO = (long)p.Ho * p.Wo;
  ADM_UNROLL
  for (int tn = 0; tn < TN; ++tn) {
    const int pp = (wn * TN + tn) * 32 + l31;
    const int px = pp & (TW - 1), py = (pp >> p.lTW) & (TH - 1), img = pp >> (p.lTW + p.lTH);
    const int oy = ty * TH + py, ox = tx * TW + px, n = n0 + img;
    if (n >= p.N || oy >= p.Ho || ox >= p.Wo) continue;
    ADM_UNROLL
    for (int tm = 0; tm < TM; ++tm)
      store_tile<HAS_CHAN, HAS_RES>(acc[tm][tn], p, m_wave + tm * 32, h, n, (long)oy * p.Wo + ox, planeO);
  }
}

#define ADM_CONV_EPILOGUE(TM_, TN_)                                                                          \
  do {                                                                                                       \
    const int m_wave_ = m0 + wm * TM_ * 32;                                                                  \
    if (p.chan_add != nullptr) {                                                                             \
      if (p.residual != nullptr) epilogue<TM_, TN_, true, true>(acc, p, m_wave_, wn, l31, h, TW, TH, tx, ty, n0);   \
      else epilogue<TM_, TN_, true, false>(acc, p, m_wave_, wn, l31, h, TW, TH, tx, ty, n0);                 \
    } else {                                                                                                 \
      if (p.residual != nullptr) epilogue<TM_, TN_, false, true>(acc, p, m_wave_, wn, l31, h, TW, TH, tx, ty, n0);  \
      else epilogue<TM_, TN_, false, false>(acc, p, m_wave_, wn, l31, h, TW, TH, tx, ty, n0);                \
    }                                                                                                        \
  } while (0)

// KSP (split K; any shape whose tiles alone leave most CUs idle — the 4x4 / 2x2 / 1x1-pixel levels of a 64x64 or latent 32x32
// model, where this kernel used to walk a 512..1024-channel K loop on Cout / 32 = 16 workgroups: 270-540 us per layer for < 1
// GFLOP): S = p.ksplit workgroups share one output tile, each walks 1/S of the channel chunks and stores its partial sums to slab
// s of the scratch buffer; ksplit_finish_kernel adds the slabs in order with bias / per-sample term / residual (deterministic).
// Waves whose 32-pixel column blocks lie entirely beyond the batch skip their MFMAs (a 1x1-pixel level at B = 16 fills 16 of a
// tile's 128 columns).
// KPF (split K, stride 1): the register pipeline of the stride-2 instantiation for the split kernel's two-to-eight
// chunks — chunk c + 1's patch values and weight slab are requested before chunk c's MFMAs (4.4 us per chunk and wave) instead of
// after them: a 2-chunk workgroup spent two full load round trips per chunk around 8.8 us of MFMA (28 us per launch). Same
// MFMA order: bit-identical.
template <int KS, int STRIDE, int WM, int TM, bool KSP = false, bool KPF = false>
__global__ void __launch_bounds__(256, STRIDE == 2 ? 2 : 1) conv_mfma_kernel(const ConvParams p) {
  constexpr int WN = 4 / WM;
  constexpr int TN = 4 / WN;
  constexpr int BM = 32 * WM * TM;
  constexpr int KS2 = KS * KS;
  ADM_DYN_SMEM(float, smem);
  float* ldsX = smem;                 // CK * CS
  float* ldsW = smem + CK * p.CS;     // CK*KS2 * BM   (CS is rounded so this stays 16-B aligned)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware bijective remap: hardware places block b on XCD b%8; give each XCD a contiguous logical range so
  // workgroups sharing an input patch / weight slab hit the same L2 (guide §5 "XCD swizzle must be bijective").
  int lid;
  {
    const int b = blockIdx.x, q = p.nblk >> 3, r = p.nblk & 7, xcd = b & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int kpart = KSP ? lid % p.ksplit : 0;
  if (KSP) lid /= p.ksplit;
  const int ct = lid % p.n_ct, pt = lid / p.n_ct;
  const int tx = pt % p.tiles_x, ty = (pt / p.tiles_x) % p.tiles_y, ig = pt / (p.tiles_x * p.tiles_y);
  const int TW = 1 << p.lTW, TH = 1 << p.lTH, NI = 128 >> (p.lTW + p.lTH);
  const int m0 = ct * BM, n0 = ig * NI;
  const int Ct = p.C1 + p.C2;
  const int planeS = p.Hs * p.Ws;
  const int IHW = p.IH * p.IW;
  // channel range of this workgroup (KSP: chunk range kpart of p.ksplit, in chunks of CK channels)
  const int nchunks_all = Ct / CK;
  const int c_begin = KSP ? (int)((long)kpart * nchunks_all / p.ksplit) * CK : 0;
  const int c_end = KSP ? (int)((long)(kpart + 1) * nchunks_all / p.ksplit) * CK : Ct;
  const int tmask = KSP ? p.tap_mask : 0x1ff;

  // ---- per-thread gather plan for the input patch (same for every channel plane) -------------------
  int q_soff[MAXQ];   // offset inside a source channel plane, or -1 (zero padding / out of range)
  int q_img[MAXQ];    // image index inside the tile
  ADM_UNROLL
  for (int qi = 0; qi < MAXQ; ++qi) {
    const int q = tid + qi * 256;
    q_soff[qi] = -1;
    q_img[qi] = 0;
    if (q < NI * IHW) {
      const int img = q / IHW, r2 = q - img * IHW;
      const int ly = r2 / p.IW, lx = r2 - ly * p.IW;
      const int gy = ty * TH * STRIDE + ly - p.pad_lo, gx = tx * TW * STRIDE + lx - p.pad_lo;
      q_img[qi] = img;
      if (gy >= 0 && gy < p.Hi && gx >= 0 && gx < p.Wi && n0 + img < p.N &&
          !(p.up == 2 && ((gy | gx) & 1))) {   // up == 2: zero-insertion (transposed stride-2 conv, backward pass)
        const int sy = p.up ? (gy >> 1) : gy, sx = p.up ? (gx >> 1) : gx;
        q_soff[qi] = sy * p.Ws + sx;
      }
    }
  }
  // ---- per-lane B (pixel) offsets into ldsX, and output coordinates ---------------------------------
  int poff[TN];
  ADM_UNROLL
  for (int tn = 0; tn < TN; ++tn) {
    const int pp = (wn * TN + tn) * 32 + l31;
    const int px = pp & (TW - 1), py = (pp >> p.lTW) & (TH - 1), img = pp >> (p.lTW + p.lTH);
    poff[tn] = img * IHW + py * STRIDE * p.IW + px * STRIDE;
  }

  f32x16 acc[TM][TN];
  ADM_UNROLL
  for (int a = 0; a < TM; ++a)
    ADM_UNROLL
    for (int b = 0; b < TN; ++b)
      ADM_UNROLL
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int a_lane = wm * TM * 32 + l31;
  // KSP: a 32-pixel column block whose first pixel already lies beyond the batch holds no output at all (wave-uniform)
  bool blk_on[TN];
  ADM_UNROLL
  for (int tn = 0; tn < TN; ++tn) blk_on[tn] = !KSP || n0 + (((ADM_UNIFORM(wn) * TN + tn) * 32) >> (p.lTW + p.lTH)) < p.N;

  // STRIDE == 2 (Downsample2D: patches of 33 x 17 pixels, 3 elements per thread and channel plane — the single-element plan
  // of conv_mfma_pf_kernel does not fit): the same loop, software-pipelined through registers — the raw patch values and the
  // weight slab of chunk c + 1 are requested before the MFMAs of chunk c and written to LDS after them.
  constexpr bool PF = STRIDE == 2 || KPF;
  constexpr int ROW4 = BM / 4;
  constexpr int TOT4 = CK * KS2 * ROW4;
  constexpr int NW4 = (TOT4 + 255) / 256;
  constexpr int PFQ = 3;                      // patch elements per thread that travel through registers (16 x 8 outputs: 561 = 3 per
                                              // thread); the tiny-output layers' elements 4 and 5 are loaded at stash time as before
  float raw[PF ? PFQ : 1][CK];
  float4 wraw[PF ? NW4 : 1];
  auto fetch = [&](int c0) __attribute__((always_inline)) {
    const bool from1 = c0 < p.C1;
    const float* xb = from1 ? p.x1 : p.x2;
    const long xbs = from1 ? p.x1_bs : p.x2_bs;
    const int cb0 = from1 ? c0 : c0 - p.C1;
    ADM_UNROLL
    for (int qi = 0; qi < PFQ; ++qi) {
      const int q = tid + qi * 256;
      if (q < NI * IHW) {
        const int soff = q_soff[qi];
        const int n = n0 + q_img[qi];
        ADM_UNROLL
        for (int c = 0; c < CK; ++c) {
          raw[PF ? qi : 0][c] = 0.f;
          if (soff >= 0) raw[PF ? qi : 0][c] = xb[(long)n * xbs + (long)(cb0 + c) * planeS + soff];
        }
      }
    }
    const float* wsrc = p.wp + (long)n0 * p.wp_bs + (long)c0 * KS2 * p.Cout + m0;
    ADM_UNROLL
    for (int i = 0; i < NW4; ++i) {
      const int idx = tid + 256 * i;
      wraw[PF ? i : 0] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < TOT4) {
        const int row = idx / ROW4, c4 = idx - row * ROW4;
        const bool live_tap = !(KSP && KS2 > 1) || ((tmask >> (row % KS2)) & 1);      // a tap no pixel of this plane can use: never read
        if (m0 + c4 * 4 < p.Cout && live_tap) wraw[PF ? i : 0] = *reinterpret_cast<const float4*>(wsrc + (long)row * p.Cout + c4 * 4);
      }
    }
  };
  if (PF) fetch(c_begin);
  for (int c0 = c_begin; c0 < c_end; c0 += CK) {
    if constexpr (PF) {
      ADM_UNROLL
      for (int qi = 0; qi < MAXQ; ++qi) {
        const int q = tid + qi * 256;
        if (q < NI * IHW) {
          const int soff = q_soff[qi];
          const int n = n0 + q_img[qi];
          float v[CK];
          if (qi < PFQ) {
            ADM_UNROLL
            for (int c = 0; c < CK; ++c) v[c] = raw[qi][c];
          } else {
            const bool f1 = c0 < p.C1;
            const float* xs = (f1 ? p.x1 : p.x2) + (long)n * (f1 ? p.x1_bs : p.x2_bs) + (long)(f1 ? c0 : c0 - p.C1) * planeS;
            ADM_UNROLL
            for (int c = 0; c < CK; ++c) v[c] = soff >= 0 ? xs[(long)c * planeS + soff] : 0.f;
          }
          if (p.gn_scale != nullptr && soff >= 0) {
            ADM_UNROLL
            for (int c = 0; c < CK; ++c) {
              const float sc = p.gn_scale[(long)n * Ct + c0 + c], sh = p.gn_shift[(long)n * Ct + c0 + c];
              float y = v[c] * sc + sh;
              if (p.act) y = silu_f(y);
              v[c] = y;
            }
          } else if (p.act && soff >= 0) {
            ADM_UNROLL
            for (int c = 0; c < CK; ++c) v[c] = silu_f(v[c]);
          }
          ADM_UNROLL
          for (int c = 0; c < CK; ++c) ldsX[c * p.CS + q] = v[c];
        }
      }
      ADM_UNROLL
      for (int i = 0; i < NW4; ++i) {
        const int idx = tid + 256 * i;
        if (idx < TOT4) {
          const int row = idx / ROW4, c4 = idx - row * ROW4;
          *reinterpret_cast<float4*>(ldsW + row * BM + c4 * 4) = wraw[i];
        }
      }
      __syncthreads();
      if (c0 + CK < c_end) fetch(c0 + CK);
    } else {
      // ---- stage the activated input patch ----------------------------------------------------------
      const bool from1 = c0 < p.C1;
      const float* xb = from1 ? p.x1 : p.x2;
      const long xbs = from1 ? p.x1_bs : p.x2_bs;
      const int cb0 = from1 ? c0 : c0 - p.C1;
      ADM_UNROLL
      for (int qi = 0; qi < MAXQ; ++qi) {
        const int q = tid + qi * 256;
        if (q < NI * IHW) {
          const int soff = q_soff[qi];
          const int n = n0 + q_img[qi];
          float v[CK];
          ADM_UNROLL
          for (int c = 0; c < CK; ++c) {
            v[c] = 0.f;
            if (soff >= 0) v[c] = xb[(long)n * xbs + (long)(cb0 + c) * planeS + soff];
          }
          if (p.gn_scale != nullptr && soff >= 0) {
            ADM_UNROLL
            for (int c = 0; c < CK; ++c) {
              const float sc = p.gn_scale[(long)n * Ct + c0 + c], sh = p.gn_shift[(long)n * Ct + c0 + c];
              float y = v[c] * sc + sh;
              if (p.act) y = silu_f(y);
              v[c] = y;
            }
          } else if (p.act && soff >= 0) {
            ADM_UNROLL
            for (int c = 0; c < CK; ++c) v[c] = silu_f(v[c]);
          }
          ADM_UNROLL
          for (int c = 0; c < CK; ++c) ldsX[c * p.CS + q] = v[c];
        }
      }
      // ---- stage the weight slab: rows (c, tap), BM consecutive couts each --------------------------
      {
        const float* wsrc = p.wp + (long)n0 * p.wp_bs + (long)c0 * KS2 * p.Cout + m0;
        for (int idx = tid; idx < TOT4; idx += 256) {
          const int row = idx / ROW4, c4 = idx - row * ROW4;
          if (KSP && KS2 > 1 && !((tmask >> (row % KS2)) & 1)) continue;      // a tap no pixel of this plane can use: never read
          float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
          if (m0 + c4 * 4 < p.Cout) w = *reinterpret_cast<const float4*>(wsrc + (long)row * p.Cout + c4 * 4);
          *reinterpret_cast<float4*>(ldsW + row * BM + c4 * 4) = w;
        }
      }
      __syncthreads();
    }
    // ---- MFMA over this chunk: 9 taps x 4 channel pairs ------------------------------------------
    ADM_UNROLL
    for (int tap = 0; tap < KS2; ++tap) {
      if (KSP && KS2 > 1 && !((tmask >> tap) & 1)) continue;     // wave-uniform (kernel argument)
      const int toff = (tap / KS) * p.IW + (tap % KS);
      ADM_UNROLL
      for (int cp = 0; cp < CK / 2; ++cp) {
        const int ch = 2 * cp + h;
        float av[TM], bv[TN];
        ADM_UNROLL
        for (int a = 0; a < TM; ++a) av[a] = ldsW[(ch * KS2 + tap) * BM + a_lane + a * 32];
        ADM_UNROLL
        for (int b = 0; b < TN; ++b) bv[b] = ldsX[ch * p.CS + poff[b] + toff];
        ADM_UNROLL
        for (int a = 0; a < TM; ++a)
          ADM_UNROLL
          for (int b = 0; b < TN; ++b)
            if (blk_on[b]) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  if constexpr (KSP) {       // partial sums (no bias) to slab kpart; ksplit_finish_kernel completes them
    const int m_wave = m0 + wm * TM * 32;
    const long planeO = (long)p.Ho * p.Wo;
    float* part = p.out + (long)kpart * p.part_stride;
    ADM_UNROLL
    for (int tn = 0; tn < TN; ++tn) {
      const int pp = (wn * TN + tn) * 32 + l31;
      const int px = pp & (TW - 1), py = (pp >> p.lTW) & (TH - 1), img = pp >> (p.lTW + p.lTH);
      const int oy = ty * TH + py, ox = tx * TW + px, n = n0 + img;
      if (n >= p.N || oy >= p.Ho || ox >= p.Wo) continue;
      const long pix = (long)oy * p.Wo + ox;
      ADM_UNROLL
      for (int tm = 0; tm < TM; ++tm) {
        ADM_UNROLL
        for (int r = 0; r < 16; ++r) {
          const int co = m_wave + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          part[((long)n * p.Cout + co) * planeO + pix] = acc[tm][tn][r];
        }
      }
    }
  } else {
    // ---- epilogue: bias + temb bias + residual, NCHW store ---------------------------------------------
    ADM_CONV_EPILOGUE(TM, TN);
  }
}


// ---------------------------------------------------------------------------------------------------------
// Software-pipelined variant (stride 1, input patch <= 256 elements per channel plane, Cout % BM == 0):
//   * the weight slab of chunk c+1 streams HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPRs, asynchronous) into
//     the second half of a double buffer while the MFMAs of chunk c run;
//   * the raw activations (+ GroupNorm scale/shift) of chunk c+1 are prefetched into registers before the MFMAs of
//     chunk c and normalised/activated into LDS after them.
// Two workgroups per CU (79.5 KiB LDS each), so one workgroup's stash/barrier phase hides under the other's MFMAs.
// KSP (split K, 3x3 at tiny spatial sizes only): S = p.ksplit workgroups share one output tile, each walks 1/S of the
// input-channel chunks and stores its partial sums (no bias) to slab s of a scratch buffer; ksplit_finish_kernel adds the slabs in
// order together with bias / per-sample term / residual. Deterministic, and the tile can be 64 couts wide: at 8x8 pixels and B = 32
// the 32-cout tiles that fill the chip without it move 20.7 KB per 2304 MFMA cycles — two co-resident workgroups sit at the
// ~10 B/clk a CU is served with (109 us per 512->512 layer whatever the split).
// 1x1 (round 6): THREE workgroups per CU. A 1x1 tile is short (K = Cin in 8 chunks of 32 channels: 32.8 k cycles of MFMA in an 89 k-cycle
// workgroup lifetime, profiles/r03_pmc_conv1x1.md) and its two ends — the first chunk's HBM round trip, the epilogue's stores — are covered
// only by the co-resident workgroups; its LDS (48 KiB) allows a third, the registers do once the allocation is capped at 168 (four
// epilogue-only values go to scratch outside the loop). Bit-identical; 27 launches of a B = 32 forward 7.14 -> 6.87 ms (r06).
template <int KS, int WM, int TM, bool KSP = false>
__global__ void __launch_bounds__(256, (KS == 1 && !KSP) ? 3 : 2) conv_mfma_pf_kernel(const ConvParams p) {
  constexpr int WN = 4 / WM;
  constexpr int TN = 4 / WN;
  constexpr int BM = 32 * WM * TM;
  constexpr int KS2 = KS * KS;
  constexpr int CKP = KS == 1 ? 32 : CK;    // channels per K chunk: 1x1 has no taps, so take 32 channels (K = 32)
  constexpr int SPL = KS == 1 ? 2 : 1;      // threads per patch element (1x1: 128 pixels -> 2 threads each)
  constexpr int NCH = CKP / SPL;            // channels prefetched per thread
  constexpr int WSLAB = CKP * KS2 * BM;
  ADM_DYN_SMEM(float, smem);
  float* ldsX = smem;
  float* ldsW0 = smem + CKP * p.CS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  int lid;
  {
    const int b = blockIdx.x, q = p.nblk >> 3, r = p.nblk & 7, xcd = b & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int kpart = KSP ? lid % p.ksplit : 0;
  if (KSP) lid /= p.ksplit;
  const int ct = lid % p.n_ct, pt = lid / p.n_ct;
  const int tx = pt % p.tiles_x, ty = (pt / p.tiles_x) % p.tiles_y, ig = pt / (p.tiles_x * p.tiles_y);
  const int TW = 1 << p.lTW, TH = 1 << p.lTH, NI = 128 >> (p.lTW + p.lTH);
  const int m0 = ct * BM, n0 = ig * NI;
  const int Ct = p.C1 + p.C2;
  const int planeS = p.Hs * p.Ws;
  const int IHW = p.IH * p.IW;

  // gather plan: one patch element per thread; CS <= 256.  1x1 (the patch IS the 128-pixel tile, rows of TW >= 4 pixels):
  // four consecutive pixels x four channels per thread, so that every global load is a dwordx4 of a contiguous row and
  // every LDS store a float4 (lanes 0..31 cover one channel's 512 bytes, lanes 32..63 the next channel's)
  const int q = KS == 1 ? 4 * (tid & 31) : tid;
  const int csub = KS == 1 ? (tid >> 5) : 0;          // 1x1: channels csub + 8 k of the chunk, k = 0..3
  const bool qv = q < NI * IHW;
  int soff = -1, qn = n0;
  if (qv) {
    const int img = q / IHW, r2 = q - img * IHW;
    const int ly = r2 / p.IW, lx = r2 - ly * p.IW;
    const int gy = ty * TH + ly - p.pad_lo, gx = tx * TW + lx - p.pad_lo;
    qn = n0 + img;
    if (gy >= 0 && gy < p.Hi && gx >= 0 && gx < p.Wi && qn < p.N && !(p.up == 2 && ((gy | gx) & 1))) {
      const int sy = p.up ? (gy >> 1) : gy, sx = p.up ? (gx >> 1) : gx;
      soff = sy * p.Ws + sx;
    }
  }
  int poff[TN];
  ADM_UNROLL
  for (int tn = 0; tn < TN; ++tn) {
    const int pp = (wn * TN + tn) * 32 + l31;
    const int px = pp & (TW - 1), py = (pp >> p.lTW) & (TH - 1), img = pp >> (p.lTW + p.lTH);
    poff[tn] = img * IHW + py * p.IW + px;
  }
  f32x16 acc[TM][TN];
  ADM_UNROLL
  for (int a = 0; a < TM; ++a)
    ADM_UNROLL
    for (int b = 0; b < TN; ++b)
      ADM_UNROLL
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int a_lane = wm * TM * 32 + l31;
  const bool has_gn = p.gn_scale != nullptr;

  constexpr int NG = KS == 1 ? 4 : NCH;     // channels per thread (1x1: four, each with four pixels in xr)
  float xr[NCH], gs[NG], gh[NG];
  ADM_UNROLL
  for (int c = 0; c < NCH; ++c) xr[c] = 0.f;
  ADM_UNROLL
  for (int c = 0; c < NG; ++c) { gs[c] = 1.f; gh[c] = 0.f; }

  auto issue = [&](int c0, int buf) {
    const bool from1 = c0 < p.C1;
    const float* xb = from1 ? p.x1 : p.x2;
    const long xbs = from1 ? p.x1_bs : p.x2_bs;
    const int cb0 = (from1 ? c0 : c0 - p.C1) + csub;
    if (soff >= 0) {
      const float* src = xb + (long)qn * xbs + (long)cb0 * planeS + soff;
      if constexpr (KS == 1) {
        ADM_UNROLL
        for (int k = 0; k < 4; ++k) {
          const float4 v = *reinterpret_cast<const float4*>(src + (long)(8 * k) * planeS);
          xr[4 * k + 0] = v.x; xr[4 * k + 1] = v.y; xr[4 * k + 2] = v.z; xr[4 * k + 3] = v.w;
        }
      } else {
        ADM_UNROLL
        for (int c = 0; c < NCH; ++c) xr[c] = src[(long)c * planeS];
      }
      if (has_gn) {
        const float* sp = p.gn_scale + (long)qn * Ct + c0 + csub;
        const float* hp = p.gn_shift + (long)qn * Ct + c0 + csub;
        ADM_UNROLL
        for (int c = 0; c < NG; ++c) { gs[c] = sp[KS == 1 ? 8 * c : c]; gh[c] = hp[KS == 1 ? 8 * c : c]; }
      }
    }
    constexpr int ROW4 = BM / 4;
    constexpr int TOT4 = WSLAB / 4;               // multiple of 64: the tail guard below is wave-uniform
    constexpr int NIT = (TOT4 + 255) / 256;
    const float* wsrc = p.wp + (long)n0 * p.wp_bs + (long)c0 * KS2 * p.Cout + m0;
    float* wdst = ldsW0 + buf * WSLAB;
    ADM_UNROLL
    for (int i = 0; i < NIT; ++i) {
      const int idx = tid + 256 * i;
      if (idx < TOT4) {
        const int row = idx / ROW4, c4 = idx - row * ROW4;
        ADM_GLDS16(wsrc + (long)row * p.Cout + c4 * 4, wdst + (256 * i + wave * 64) * 4);
      }
    }
  };

  const int nchunks_all = Ct / CKP;
  const int ci0 = KSP ? (int)((long)kpart * nchunks_all / p.ksplit) : 0;
  const int nchunks = KSP ? (int)((long)(kpart + 1) * nchunks_all / p.ksplit) : nchunks_all;
  issue(ci0 * CKP, ci0 & 1);
  for (int ci = ci0; ci < nchunks; ++ci) {
#if !defined(ADM_EMU)
    __builtin_amdgcn_s_setprio(3);   // the short stash/issue phase should not queue behind the other workgroup's MFMAs
#endif
    if constexpr (KS == 1) {
      if (qv) {
        const bool live = soff >= 0;
        ADM_UNROLL
        for (int k = 0; k < 4; ++k) {
          float v[4];
          ADM_UNROLL
          for (int e = 0; e < 4; ++e) v[e] = xr[4 * k + e] * gs[k] + gh[k];      // gs = 1, gh = 0 without GroupNorm
          if (p.act) {                      // wave-uniform: shortcut convolutions and attention projections skip it
            ADM_UNROLL
            for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
          }
          *reinterpret_cast<float4*>(ldsX + (csub + 8 * k) * p.CS + q) =
              live ? make_float4(v[0], v[1], v[2], v[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    } else if (qv) {
      const bool live = soff >= 0;
      ADM_UNROLL
      for (int c = 0; c < NCH; ++c) {
        float v = xr[c] * gs[c] + gh[c];      // gs = 1, gh = 0 without GroupNorm
        const float sv = silu_f(v);
        v = p.act ? sv : v;
        ldsX[(csub + c) * p.CS + q] = live ? v : 0.f;   // zero padding is applied AFTER the activation
      }
    }
    __syncthreads();  // also drains the LDS-DMA of this chunk's weight slab (issued one chunk ago)
    if (ci + 1 < nchunks) issue((ci + 1) * CKP, (ci + 1) & 1);
#if !defined(ADM_EMU)
    __builtin_amdgcn_s_setprio(0);
#endif
    const float* ldsW = ldsW0 + (ci & 1) * WSLAB;
    ADM_UNROLL
    for (int tap = 0; tap < KS2; ++tap) {
      const int toff = (tap / KS) * p.IW + (tap % KS);
      ADM_UNROLL
      for (int cp = 0; cp < CKP / 2; ++cp) {
        const int ch = 2 * cp + h;
        float av[TM], bv[TN];
        ADM_UNROLL
        for (int a = 0; a < TM; ++a) av[a] = ldsW[(ch * KS2 + tap) * BM + a_lane + a * 32];
        ADM_UNROLL
        for (int b = 0; b < TN; ++b) bv[b] = ldsX[ch * p.CS + poff[b] + toff];
        ADM_UNROLL
        for (int a = 0; a < TM; ++a)
          ADM_UNROLL
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  if constexpr (KSP) {
    const int m_wave = m0 + wm * TM * 32;
    const long planeO = (long)p.Ho * p.Wo;
    float* part = p.out + (long)kpart * p.part_stride;
    ADM_UNROLL
    for (int tn = 0; tn < TN; ++tn) {
      const int pp = (wn * TN + tn) * 32 + l31;
      const int px = pp & (TW - 1), py = (pp >> p.lTW) & (TH - 1), img = pp >> (p.lTW + p.lTH);
      const int oy = ty * TH + py, ox = tx * TW + px, n = n0 + img;
      if (n >= p.N || oy >= p.Ho || ox >= p.Wo) continue;
      const long pix = (long)oy * p.Wo + ox;
      ADM_UNROLL
      for (int tm = 0; tm < TM; ++tm) {
        ADM_UNROLL
        for (int r = 0; r < 16; ++r) {
          const int co = m_wave + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          part[((long)n * p.Cout + co) * planeO + pix] = acc[tm][tn][r];
        }
      }
    }
  } else {
    ADM_CONV_EPILOGUE(TM, TN);
  }
}

void conv_out_dims(int H, int W, int up, int stride, int ks, int pad_lo, int* Ho, int* Wo) {
  const int Hi = up ? 2 * H : H, Wi = up ? 2 * W : W;
  if (stride == 1) {
    *Ho = Hi; *Wo = Wi;  // "same" (ks=3,pad 1) or 1x1
  } else {
    // symmetric pad 1: floor((Hi+2-3)/2)+1 ; asymmetric (0,1): floor((Hi+1-3)/2)+1
    *Ho = (Hi + (pad_lo ? 2 : 1) - ks) / stride + 1;
    *Wo = (Wi + (pad_lo ? 2 : 1) - ks) / stride + 1;
  }
}

static thread_local int g_last_variant = 0;
int last_conv_variant() { return g_last_variant; }
void set_last_conv_variant(int v) { g_last_variant = v; }

static inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

int launch_conv_small(const adm_conv_args& a, hipStream_t st);  // k_conv_small.hip
int conv_small_stats_tiles(const adm_conv_args& a);

#if !defined(ADM_EMU)
template <class K>
static void allow_big_lds(K kernel, size_t smem) {
  if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
}
#else
template <class K> static void allow_big_lds(K, size_t) {}
#endif

// bias is read unconditionally by the epilogue: a NULL bias maps to a shared all-zero device buffer.
const float* conv_zero_bias(int n);
static const float* zero_bias(int n) { return conv_zero_bias(n); }
// Both constant buffers are keyed by the current HIP device (a process normally owns one GPU, but nothing here assumes
// it) and grown outside any stream capture: the executors call them from finalize()/their uncaptured warm-up forward.
int conv_dev_slot() {
  int dev = 0;
#if !defined(ADM_EMU)
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
#endif
  return dev;
}
const float* conv_zero_bias(int n) {
  static float* z[16] = {};
  static int cap[16] = {};
  const int d = conv_dev_slot();
  if (n > cap[d]) {
    void* pz = nullptr;
    const int want = n < 8192 ? 8192 : n;
    if (dmalloc(&pz, sizeof(float) * want) != 0) return nullptr;
    dmemset(pz, 0, sizeof(float) * want, nullptr);
    stream_sync(nullptr);
    z[d] = (float*)pz;  // the previous (smaller) buffer is intentionally leaked: launches may still read it
    cap[d] = want;
  }
  return z[d];
}

// shared all-ones device buffer (identity GroupNorm scale for convolutions without a normalisation on their input)
const float* conv_const_ones(int n) {
  static float* z[16] = {};
  static int cap[16] = {};
  const int d = conv_dev_slot();
  if (n > cap[d]) {
    const int want = n < 8192 ? 8192 : n;
    std::vector<float> h((size_t)want, 1.0f);
    void* pz = nullptr;
    if (dmalloc(&pz, sizeof(float) * want) != 0) return nullptr;
    if (copy_h2d(pz, h.data(), sizeof(float) * want, nullptr) != 0) return nullptr;
    stream_sync(nullptr);
    z[d] = (float*)pz;
    cap[d] = want;
  }
  return z[d];
}

static bool use_pf() {
  static const int v = [] { const char* e = getenv("ADM_CONV_PF"); return e ? atoi(e) : 1; }();
  return v != 0;
}

// out[n][co][pix] = bias[co] + chan_add[n][co] + residual + sum_s part[s][...] (slabs in order); float4 over pixels when HW % 4 == 0
__global__ void __launch_bounds__(256) ksplit_finish_kernel(const float* __restrict__ part, int S, long part_stride,
                                                            const float* __restrict__ bias, const float* __restrict__ chan_add,
                                                            int chan_add_stride, const float* residual, float* out, int Cout, int HW,
                                                            long total4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const long e = i * 4, nc = e / HW;
    const int co = (int)(nc % Cout), n = (int)(nc / Cout);
    float b = bias[co];
    if (chan_add != nullptr) b += chan_add[(long)n * chan_add_stride + co];
    float4 v = *reinterpret_cast<const float4*>(part + e);
#pragma unroll 8
    for (int s2 = 1; s2 < S; ++s2) {
      const float4 q = *reinterpret_cast<const float4*>(part + (long)s2 * part_stride + e);
      v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    v.x += b; v.y += b; v.z += b; v.w += b;
    if (residual != nullptr) {
      const float4 q = *reinterpret_cast<const float4*>(residual + e);
      v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    *reinterpret_cast<float4*>(out + e) = v;
  }
}

// the same for planes of 1 or 2 pixels (HW % 4 != 0: four consecutive elements belong to different channels)
__global__ void __launch_bounds__(256) ksplit_finish1_kernel(const float* __restrict__ part, int S, long part_stride,
                                                             const float* __restrict__ bias, const float* __restrict__ chan_add,
                                                             int chan_add_stride, const float* residual, float* out, int Cout, int HW,
                                                             long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long nc = e / HW;
    const int co = (int)(nc % Cout), n = (int)(nc / Cout);
    float v = part[e];
#pragma unroll 8
    for (int s2 = 1; s2 < S; ++s2) v += part[(long)s2 * part_stride + e];
    v += chan_add != nullptr ? bias[co] + chan_add[(long)n * chan_add_stride + co] : bias[co];
    if (residual != nullptr) v += residual[e];
    out[e] = v;
  }
}

// the finish pass of a split convolution whose consumer is a GroupNorm and whose planes are multiples of 256 pixels (the split Winograd
// launches of the single-sample rule: 16x16 .. 64x64 planes): ksplit_finish_kernel's arithmetic, one wave per (sample, channel, strip of
// 256 pixels), and the strip's (sum, sum of squares) left in the layout of the convolutions' statistics epilogues (adm_conv_args.stats_out,
// HW / 256 tiles per channel) — gn_finalize_kernel reads them; no read pass over the tensor, no second launch beyond the finish itself.
__global__ void __launch_bounds__(256) ksplit_finish_stats_kernel(const float* __restrict__ part, int S, long part_stride,
                                                                  const float* __restrict__ bias, const float* __restrict__ chan_add,
                                                                  int chan_add_stride, const float* residual, float* out, int Cout, int HW,
                                                                  long nstrips, double* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  for (long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6); w < nstrips; w += (long)gridDim.x * 4) {
    const long e = w * 256 + lane * 4, nc = e / HW;
    const int co = (int)(nc % Cout), n = (int)(nc / Cout);
    float b = bias[co];
    if (chan_add != nullptr) b += chan_add[(long)n * chan_add_stride + co];
    float4 v = *reinterpret_cast<const float4*>(part + e);
#pragma unroll 8
    for (int s2 = 1; s2 < S; ++s2) {
      const float4 q = *reinterpret_cast<const float4*>(part + (long)s2 * part_stride + e);
      v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    v.x += b; v.y += b; v.z += b; v.w += b;
    if (residual != nullptr) {
      const float4 q = *reinterpret_cast<const float4*>(residual + e);
      v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    *reinterpret_cast<float4*>(out + e) = v;
    // the lane's four values in fp32, everything across lanes in fp64 (as the convolutions' own epilogues do)
    double s1 = (double)((v.x + v.y) + (v.z + v.w)), s2 = (double)((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
    ADM_UNROLL
    for (int m = 32; m >= 1; m >>= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }
    if (lane == 0) { stats[w * 2] = s1; stats[w * 2 + 1] = s2; }
  }
}

// "single_sample": the partition rules of a model that is sampled one spectrogram at a time (include/adm.h) — process-wide default, overridden
// per call (= per model) by adm_conv_args.single_sample (1 on, -1 off)
static std::atomic<int> g_single_sample{-1};
void set_single_sample(int v) { g_single_sample.store(v); }
bool conv_single_sample(const adm_conv_args& a) { return single_sample_rule(a.single_sample); }
bool single_sample_rule(int model_value) {
  if (model_value != 0) return model_value > 0;
  int v = g_single_sample.load();
  if (v < 0) { const char* e = getenv("ADM_SINGLE_SAMPLE"); v = e ? atoi(e) : 0; g_single_sample.store(v); }
  return v > 0;
}

// scratch for the split-K partial slabs, one buffer per (device, stream): launches on one stream are ordered, two streams must
// not share it. Grown only outside stream capture (the executors run one uncaptured forward before they capture), geometrically;
// a superseded buffer is RETIRED, not freed (round 6): a hipGraph captured on the stream — any model's sampling loop — holds its address
// and may be replayed after the growth; the retired buffers (less than the live one in total) go when the slot is released.
// The table is keyed dynamically (round 5): a slot is NEVER taken away from the stream it belongs to, so a process may use
// up to 1024 streams per device (>= 1 MiB each; every model owns one side stream) and the next fails loudly instead of stealing a live
// stream's scratch.
// nullptr = it cannot be provided (capture in progress and the buffer too small, table full, out of memory):
// the callers FAIL the launch — the unsplit kernel sums in another fp32 order, and a row's bits must not depend on such things.
// A stream's owner gives the slot back with conv_ksplit_release (adm_release_stream) before destroying the stream: a server with a stream
// per request would otherwise keep a buffer per stream it ever used and hit the slot limit.
namespace {
struct KsplitSlot { float* buf = nullptr; size_t cap = 0; std::vector<float*> retired; };   // retired: outgrown buffers, kept until the slot is released
std::map<std::pair<int, hipStream_t>, KsplitSlot>& ksplit_slots() { static std::map<std::pair<int, hipStream_t>, KsplitSlot> m; return m; }
int g_ksplit_per_dev[16] = {};
std::mutex g_ksplit_mu;
}  // namespace
void conv_ksplit_release(hipStream_t st) {
  std::lock_guard<std::mutex> lock(g_ksplit_mu);
  const int d = conv_dev_slot();
  auto& slots = ksplit_slots();
  auto it = slots.find(std::make_pair(d, st));
  if (it == slots.end()) return;
  if (it->second.buf != nullptr) {       // launches (or a graph replay) queued on the stream may still use the slabs: drain, then free
    (void)stream_sync(st);
    dfree(it->second.buf);
    for (float* q : it->second.retired) dfree(q);
  }
  slots.erase(it);
  --g_ksplit_per_dev[d & 15];
}
static float* ksplit_scratch(size_t floats, hipStream_t st) {
  typedef KsplitSlot Slot;
  auto& slots = ksplit_slots();
  int* per_dev = g_ksplit_per_dev;
  std::lock_guard<std::mutex> lock(g_ksplit_mu);
  const int d = conv_dev_slot();
  bool capturing = false;
#if !defined(ADM_EMU)
  {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    capturing = hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
  }
#endif
  auto it = slots.find(std::make_pair(d, st));
  if (it == slots.end()) {
    if (per_dev[d & 15] >= 1024) {
      set_error("split-K scratch: more than 1024 streams have run split-K convolutions on this device; a stream's scratch is never "
                "taken over (captured graphs hold its address) — reuse streams, or give a stream's slot back with adm_release_stream before destroying it");
      return nullptr;
    }
    ++per_dev[d & 15];
    it = slots.emplace(std::make_pair(d, st), Slot()).first;
  }
  Slot* sl = &it->second;
  if (floats <= sl->cap) return sl->buf;
  if (capturing) {
    set_error("split-K scratch must grow during stream capture: run one uncaptured pass at this batch size first");
    return nullptr;
  }
  // first buffer of a stream: what is asked for, at least 1 MiB (a model's side stream — Net::plan_side_overlap — only ever runs small-plane
  // launches; round 5's 32 MiB floor cost every live model that much). Growth is safe at any time outside a capture: see below.
  size_t want = floats < ((size_t)256 << 10) ? ((size_t)256 << 10) : floats;
  if (want < 2 * sl->cap) want = 2 * sl->cap;
  void* q = nullptr;
  if (dmalloc(&q, sizeof(float) * want) != 0) { set_error("split-K scratch: out of device memory"); return nullptr; }
  // An outgrown buffer is NOT freed: hipGraphs captured on this stream (any model's sampling loop) hold its address in their kernel
  // arguments and may be replayed after this call — they keep working on their old slabs (launches on one stream are ordered, so sharing
  // them is as safe as before). Buffers double, so the retired ones add up to less than the live one; adm_release_stream frees them all.
  if (sl->buf != nullptr) sl->retired.push_back(sl->buf);
  sl->buf = (float*)q;
  sl->cap = want;
  return sl->buf;
}

// the slab reduction and its scratch for other kernels' split-K paths (k_conv_small.hip: conv_out on small images)
float* conv_ksplit_scratch(size_t floats, hipStream_t st) { return ksplit_scratch(floats, st); }
int launch_ksplit_finish(const float* part, int S, long total, const float* bias, const float* chan_add, int chan_add_stride,
                         const float* residual, float* out, int Cout, int HW, hipStream_t st) {
  if (bias == nullptr) bias = zero_bias(Cout);
  ADM_REQUIRE(bias != nullptr, "ksplit_finish: zero-bias buffer");
  if (HW % 4 == 0) {
    long g = (total / 4 + 255) / 256;
    if (g > 4096) g = 4096;
    ADM_LAUNCH(ksplit_finish_kernel, dim3((unsigned)g), dim3(256), 0, st, part, S, total, bias, chan_add, chan_add_stride, residual,
               out, Cout, HW, total / 4);
  } else {
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    ADM_LAUNCH(ksplit_finish1_kernel, dim3((unsigned)g), dim3(256), 0, st, part, S, total, bias, chan_add, chan_add_stride, residual,
               out, Cout, HW, total);
  }
  return ADM_CHECK_LAUNCH();
}

int launch_ksplit_finish_stats(const float* part, int S, long total, const float* bias, const float* chan_add, int chan_add_stride,
                               const float* residual, float* out, int Cout, int HW, double* stats, hipStream_t st) {
  if (bias == nullptr) bias = zero_bias(Cout);
  ADM_REQUIRE(bias != nullptr, "ksplit_finish_stats: zero-bias buffer");
  ADM_REQUIRE(HW % 256 == 0 && stats != nullptr, "ksplit_finish_stats: planes of a multiple of 256 pixels only");
  const long nstrips = total / 256;
  long g = (nstrips + 3) / 4;
  if (g > 4096) g = 4096;
  ADM_LAUNCH(ksplit_finish_stats_kernel, dim3((unsigned)g), dim3(256), 0, st, part, S, total, bias, chan_add, chan_add_stride, residual,
             out, Cout, HW, nstrips, stats);
  return ADM_CHECK_LAUNCH();
}

template <int KS>
static int launch_ksplit(const ConvParams& p, int bm, int S, hipStream_t st) {
  constexpr int CKP = KS == 1 ? 32 : CK;
  const size_t smem = sizeof(float) * ((size_t)CKP * p.CS + 2 * (size_t)CKP * KS * KS * bm);
  const long total = (long)p.N * p.Cout * p.Ho * p.Wo;
  float* scratch = ksplit_scratch((size_t)S * total, st);
  if (scratch == nullptr) return -1;                      // (error recorded) never fall back silently: another summation order
  ConvParams q = p;
  q.ksplit = S; q.part_stride = total; q.out = scratch;
  q.n_ct = p.Cout / bm;
  q.nblk = (p.nblk / p.n_ct) * q.n_ct * S;
  if (bm == 64) {
    allow_big_lds(conv_mfma_pf_kernel<KS, 2, 1, true>, smem);
    ADM_LAUNCH((conv_mfma_pf_kernel<KS, 2, 1, true>), dim3(q.nblk), dim3(256), smem, st, q);
  } else {
    allow_big_lds(conv_mfma_pf_kernel<KS, 1, 1, true>, smem);
    ADM_LAUNCH((conv_mfma_pf_kernel<KS, 1, 1, true>), dim3(q.nblk), dim3(256), smem, st, q);
  }
  if (const GnFuse* f = conv_gn_fuse_pending(p.Cout))
    return launch_ksplit_finish_gn(scratch, S, total, p.bias, p.chan_add, p.chan_add_stride, p.residual, p.out, p.N, p.Cout, p.Ho * p.Wo, *f, st);
  long g = (total / 4 + 255) / 256;
  if (g > 4096) g = 4096;
  ADM_LAUNCH(ksplit_finish_kernel, dim3((unsigned)g), dim3(256), 0, st, (const float*)scratch, S, total, p.bias, p.chan_add,
             p.chan_add_stride, p.residual, p.out, p.Cout, p.Ho * p.Wo, total / 4);
  return ADM_CHECK_LAUNCH();
}

template <int KS>
static int dispatch_pf(const ConvParams& p, int bm, hipStream_t st) {
  constexpr int CKP = KS == 1 ? 32 : CK;
  const size_t smem = sizeof(float) * ((size_t)CKP * p.CS + 2 * (size_t)CKP * KS * KS * bm);
  dim3 grid(p.nblk), block(256);
  // 3x3 at tiny spatial sizes (the tiles alone cannot give every CU two workgroups): 64-cout tiles, K split over S workgroups
  static const int use_ksp = [] { const char* e = getenv("ADM_CONV_KSPLIT"); return e ? atoi(e) : 1; }();
  const int nch = (p.C1 + p.C2) / CKP;
  const long total = (long)p.N * p.Cout * p.Ho * p.Wo;
  // The decision and the number of parts depend on the LAYER only (output plane <= 8x8 pixels, channel count), never on the batch
  // size: the partition fixes the fp32 summation order, and a sample's result must not depend on how many samples share its
  // launch (round 2 chose S from the tile count: rows of a 32-batch and of a 256-batch then differed in the last bit, which a
  // random-weight sampler amplifies to different images — strong scaling would not have reproduced weak scaling's pictures).
  if (KS == 3 && use_ksp && p.Ho * p.Wo <= 64 && nch >= 16 && p.wp_bs == 0 && total % 4 == 0) {
    const int bm2 = p.Cout % 64 == 0 ? 64 : 32;
    // 512+ input channels (the 8x8 level of the 256x256 model), measured at B = 32 (the bench batch): 14 layers 1.54 ms with 4
    // parts, 1.90 ms with 8 (twice the slab traffic). Up to 384 channels (the 8x8 levels of the 64x64 and latent 32x32 models, whose
    // batches are 1..16 images: 16..128 workgroups with 4 parts): 8 parts, -85 us per config-4 step, -70 us per config-1 step.
    static const int s8_cout = [] { const char* e = getenv("ADM_KSP_PF_S8_COUT"); return e ? atoi(e) : 256; }();     // developer A/B
    int S = (nch <= 48 || p.Cout <= s8_cout) ? 8 : 4;     // (<= 256 couts: the up-path 8x8 layers of those models, 768 -> 256)
    // the single-sample rule ("single_sample", by model): 16 parts. One 256x256 sample: 14 such layers 61 -> 28 us each (32 workgroups of
    // 16 chunks -> 128 of 4; 256x256, B = 1: 5.21 -> 4.62 ms per step), where a batch of 32 pays 4x the slab traffic of the default
    if (p.single) S = 16;
    while (S > 1 && nch / S < 4) --S;
    if (S > 1) {
      const int rc = launch_ksplit<KS>(p, bm2, S, st);
      if (rc <= 0) { if (rc == 0) g_last_variant += 5; return rc; }      // 2316: the split-K instantiation
    }
  }
  if (bm == 128) {
    allow_big_lds(conv_mfma_pf_kernel<KS, 2, 2>, smem);
    ADM_LAUNCH((conv_mfma_pf_kernel<KS, 2, 2>), grid, block, smem, st, p);
  } else if (bm == 64) {
    allow_big_lds(conv_mfma_pf_kernel<KS, 2, 1>, smem);
    ADM_LAUNCH((conv_mfma_pf_kernel<KS, 2, 1>), grid, block, smem, st, p);
  } else {
    allow_big_lds(conv_mfma_pf_kernel<KS, 1, 1>, smem);
    ADM_LAUNCH((conv_mfma_pf_kernel<KS, 1, 1>), grid, block, smem, st, p);
  }
  return ADM_CHECK_LAUNCH();
}

// Split K for the generic kernel (see conv_mfma_kernel): taken for output planes of at most 8x8 pixels (the tiles of the unsplit
// launch then leave most CUs idle) when the K loop is long enough to be worth two launches. 128- (else 64- / 32-) cout tiles —
// every workgroup of a tile row stages the same patch, so wide tiles cut the staging work — and S = 64 / 32 / 8 parts for planes
// of <= 4 / <= 16 / <= 64 pixels, at least two 8-channel chunks each. Returns 1 when not taken (the caller runs the unsplit kernel).
template <int KS, int STRIDE>
static int launch_ksplit_generic(const ConvParams& p, hipStream_t st) {
  static const int use_ksp = [] { const char* e = getenv("ADM_CONV_KSPLIT"); return e ? atoi(e) : 1; }();
  const int nch = (p.C1 + p.C2) / CK;
  const int n_pt = p.nblk / p.n_ct;
  const int HWo = p.Ho * p.Wo;
  // Whether and how K is split depends on the LAYER only (output plane, channel counts), never on the batch size: the partition
  // fixes the fp32 summation order, and a sample's result must not depend on how many samples share its launch (rows sampled
  // alone, in another batch or on another number of GPUs are bit-identical; tests/test_full_size.py, tests/test_distributed.py).
  // The single-sample rule ("single_sample", by model) lifts the plane bound for whatever reaches this kernel on larger planes — the stride-2
  // Downsample2D convolutions: one sample gives a 256 -> 256 layer with a 16x16 output 16 workgroups of 32 chunks each (106 us for 0.3 GFLOP).
  const bool big = HWo > 64;
  if (!use_ksp || p.wp_bs != 0 || nch < 8 || (big && !(p.single && KS == 3))) return 1;
  // (64-cout tiles = twice the workgroups, two per CU: measured SLOWER in the latency regime, 3.85 vs 3.50 ms per config-4 step —
  // every workgroup of a tile row stages the same patch)
  const int bm = p.Cout % 128 == 0 ? 128 : (p.Cout % 64 == 0 ? 64 : 32);
  int S = HWo <= 4 ? 64 : (HWo <= 16 ? 32 : (p.single ? 32 : 8));      // (single-sample rule: the 8x8 planes as finely as the 4x4 ones)
  if (big) {                            // parts until ONE sample's workgroups reach 256 (the batch never enters), at most 16
    const int wg1 = p.tiles_x * p.tiles_y * (p.Cout / bm);
    if (wg1 >= 128) return 1;
    S = 2;
    while (S < 16 && S * wg1 < 256) S *= 2;
  }
  if (S > nch / 2) S = nch / 2;
  if (S < 2) return 1;
  const long total = (long)p.N * p.Cout * p.Ho * p.Wo;
  float* scratch = ksplit_scratch((size_t)S * total, st);
  if (scratch == nullptr) return -1;
  ConvParams q = p;
  q.ksplit = S; q.part_stride = total; q.out = scratch;
  q.n_ct = p.Cout / bm;
  q.nblk = n_pt * q.n_ct * S;
  // taps that can meet a pixel inside the (upsampled) input for SOME output pixel: row ty is live iff an output row oy exists with
  // 0 <= oy * STRIDE + ty - pad < Hi. At the 1x1-pixel level of the latent model that is the centre tap alone: 1 / 9 of the filter
  // bytes these weight-streaming layers move (18.9 MB per 512 -> 512 layer); stride 2 keeps every tap (its register-pipelined
  // staging is not masked).
  q.tap_mask = 0x1ff;
  if (KS == 3 && STRIDE == 1) {
    int rows = 0, cols = 0;
    for (int t = 0; t < 3; ++t) {
      bool r = false, c = false;
      for (int o = 0; o < p.Ho; ++o) r |= o + t - p.pad_lo >= 0 && o + t - p.pad_lo < p.Hi;
      for (int o = 0; o < p.Wo; ++o) c |= o + t - p.pad_lo >= 0 && o + t - p.pad_lo < p.Wi;
      rows |= r << t; cols |= c << t;
    }
    q.tap_mask = 0;
    for (int ty = 0; ty < 3; ++ty)
      for (int tx = 0; tx < 3; ++tx)
        if (((rows >> ty) & 1) && ((cols >> tx) & 1)) q.tap_mask |= 1 << (ty * 3 + tx);
  }
  const size_t smem = sizeof(float) * ((size_t)CK * p.CS + (size_t)CK * KS * KS * bm);
  static const int kpf = [] { const char* e = getenv("ADM_KSP_PIPE"); return e ? atoi(e) : 1; }();     // developer A/B
  if (bm == 128 && STRIDE == 1 && kpf) {       // (patch elements beyond three per thread are loaded at stash time, as in the stride-2 kernel)
    allow_big_lds(conv_mfma_kernel<KS, STRIDE, 2, 2, true, STRIDE == 1>, smem);
    ADM_LAUNCH((conv_mfma_kernel<KS, STRIDE, 2, 2, true, STRIDE == 1>), dim3(q.nblk), dim3(256), smem, st, q);
  } else if (bm == 128) {
    allow_big_lds(conv_mfma_kernel<KS, STRIDE, 2, 2, true>, smem);
    ADM_LAUNCH((conv_mfma_kernel<KS, STRIDE, 2, 2, true>), dim3(q.nblk), dim3(256), smem, st, q);
  } else if (bm == 64) {
    allow_big_lds(conv_mfma_kernel<KS, STRIDE, 2, 1, true>, smem);
    ADM_LAUNCH((conv_mfma_kernel<KS, STRIDE, 2, 1, true>), dim3(q.nblk), dim3(256), smem, st, q);
  } else {
    allow_big_lds(conv_mfma_kernel<KS, STRIDE, 1, 1, true>, smem);
    ADM_LAUNCH((conv_mfma_kernel<KS, STRIDE, 1, 1, true>), dim3(q.nblk), dim3(256), smem, st, q);
  }
  const int HW = p.Ho * p.Wo;
  g_last_variant = KS * 100 + STRIDE * 10 + bm / 32 + 5;      // x16 / x26 + ...: 3x3 stride 1 -> 316 / 317 / 319 (bm 32 / 64 / 128)
  if (const GnFuse* f = big ? nullptr : conv_gn_fuse_pending(p.Cout))      // (one workgroup per (group, sample) is a small-plane design)
    return launch_ksplit_finish_gn(scratch, S, total, p.bias, p.chan_add, p.chan_add_stride, p.residual, p.out, p.N, p.Cout, HW, *f, st);
  if (HW % 4 == 0) {
    long g = (total / 4 + 255) / 256;
    if (g > 4096) g = 4096;
    ADM_LAUNCH(ksplit_finish_kernel, dim3((unsigned)g), dim3(256), 0, st, (const float*)scratch, S, total, p.bias, p.chan_add,
               p.chan_add_stride, p.residual, p.out, p.Cout, HW, total / 4);
  } else {
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    ADM_LAUNCH(ksplit_finish1_kernel, dim3((unsigned)g), dim3(256), 0, st, (const float*)scratch, S, total, p.bias, p.chan_add,
               p.chan_add_stride, p.residual, p.out, p.Cout, HW, total);
  }
  return ADM_CHECK_LAUNCH();
}

template <int KS, int STRIDE>
static int dispatch_bm(const ConvParams& p, int bm, size_t smem, hipStream_t st) {
  {
    const int rc = launch_ksplit_generic<KS, STRIDE>(p, st);
    if (rc <= 0) return rc;
  }
  allow_big_lds(conv_mfma_kernel<KS, STRIDE, 2, 2>, smem);
  dim3 grid(p.nblk), block(256);
  if (bm == 128) {
    ADM_LAUNCH((conv_mfma_kernel<KS, STRIDE, 2, 2>), grid, block, smem, st, p);
  } else if (bm == 64) {
    ADM_LAUNCH((conv_mfma_kernel<KS, STRIDE, 2, 1>), grid, block, smem, st, p);
  } else {
    ADM_LAUNCH((conv_mfma_kernel<KS, STRIDE, 1, 1>), grid, block, smem, st, p);
  }
  return ADM_CHECK_LAUNCH();
}

// Mirrors launch_conv2d's dispatch: the number of GroupNorm statistic tiles per (sample, output channel) the kernel that would
// run for `a` emits into a.stats_out, or 0 when that kernel has no statistics epilogue (the consumer then runs gn_stats_kernel).
int conv_stats_tiles(const adm_conv_args& a) {
  const int C2 = a.x2 ? a.C2 : 0, Ct = a.C1 + C2;
  if (Ct % CK != 0 || a.C1 % CK != 0 || a.Cout % 4 != 0 || a.Cout < 32) return conv_small_stats_tiles(a);
  if (conv_bf16_enabled() && conv_bf16_eligible(a)) return 0;
  if (conv_bf16_mode() >= 2 && conv1x1_bf16_eligible(a)) return 0;
  if (winograd_enabled() && winograd_eligible(a)) return winograd_stats_tiles(a);
  return 0;
}

int launch_conv2d(const adm_conv_args& a, hipStream_t st) {
  const int C2 = a.x2 ? a.C2 : 0;
  const int Ct = a.C1 + C2;
  ADM_REQUIRE(a.ks == 3 || a.ks == 1, "conv2d: ks must be 1 or 3");
  ADM_REQUIRE(a.stride == 1 || a.stride == 2, "conv2d: stride must be 1 or 2");
  ADM_REQUIRE(a.stats_out == nullptr || a.stats_tiles == conv_stats_tiles(a),
              "conv2d: stats_tiles does not match adm_conv_stats_tiles for these arguments");
  if (Ct % CK != 0 || a.C1 % CK != 0 || a.Cout % 4 != 0 || a.Cout < 32)
    return launch_conv_small(a, st);  // conv_in / conv_out class (tiny Cin or Cout): direct kernel
  ADM_REQUIRE(!(a.ks == 1 && (a.stride != 1 || a.up)), "conv2d: 1x1 supports stride 1, no upsample");
  if (conv_bf16_enabled() && conv_bf16_eligible(a)) return launch_conv_bf16(a, st);
  if (conv_bf16_mode() >= 2 && conv1x1_bf16_eligible(a)) return launch_conv1x1_bf16(a, st);
  if (winograd_enabled() && winograd_eligible(a)) return launch_conv_winograd(a, st);
  ConvParams p;
  p.x1 = a.x1; p.x2 = a.x2; p.C1 = a.C1; p.C2 = C2;
  p.N = a.N; p.Hs = a.H; p.Ws = a.W;
  p.Hi = a.up ? 2 * a.H : a.H; p.Wi = a.up ? 2 * a.W : a.W;
  ADM_REQUIRE(a.up == 0 || a.up == 1 || a.up == 2, "conv2d: up must be 0 (none), 1 (nearest x2) or 2 (zero-insertion x2)");
  conv_out_dims(a.H, a.W, a.up, a.stride, a.ks, a.pad_lo, &p.Ho, &p.Wo);
  p.up = a.up; p.pad_lo = a.ks == 1 ? 0 : a.pad_lo;
  p.gn_scale = a.gn_scale; p.gn_shift = a.gn_shift; p.act = a.act;
  ADM_REQUIRE(a.Cout % 32 == 0, "conv2d: Cout must be a multiple of 32 for the MFMA kernel");
  p.wp = a.wpacked; p.bias = a.bias ? a.bias : zero_bias(a.Cout); p.Cout = a.Cout;
  ADM_REQUIRE(p.bias != nullptr, "conv2d: could not allocate the zero-bias buffer");
  p.chan_add = a.chan_add; p.chan_add_stride = a.chan_add_stride;
  p.residual = a.residual; p.out = a.out;
  p.x1_bs = a.x1_bstride ? a.x1_bstride : (long)a.C1 * a.H * a.W;
  p.x2_bs = a.x2_bstride ? a.x2_bstride : (long)C2 * a.H * a.W;
  p.wp_bs = a.w_bstride;
  p.ksplit = 1; p.part_stride = 0; p.tap_mask = 0x1ff;
  p.single = conv_single_sample(a) ? 1 : 0;
  // 3x3: 16 x 8 pixel tiles (small halo); 1x1 has no halo: rows as long as the image allows (<= 128 pixels), so that the
  // pipelined kernel loads and stores whole contiguous row segments
  int TW = p.Wo >= 16 ? 16 : p.Wo, TH = p.Ho >= 8 ? 8 : p.Ho;
  const bool wide1x1 = a.ks == 1 && use_pf() && (p.Wo & (p.Wo - 1)) == 0 && (p.Ho & (p.Ho - 1)) == 0 && p.Wo >= 4 &&
                       (long)p.Wo * p.Ho >= 128;
  if (wide1x1) { TW = p.Wo >= 128 ? 128 : p.Wo; TH = 128 / TW; }
  ADM_REQUIRE((TW & (TW - 1)) == 0 && (TH & (TH - 1)) == 0, "conv2d: output dims below 16x8 must be powers of two");
  p.lTW = ilog2(TW); p.lTH = ilog2(TH);
  const int NI = 128 / (TW * TH);
  p.tiles_x = ceil_div(p.Wo, TW); p.tiles_y = ceil_div(p.Ho, TH);
  const int img_groups = ceil_div(a.N, NI);
  p.IH = (TH - 1) * a.stride + a.ks; p.IW = (TW - 1) * a.stride + a.ks;
  p.CS = (NI * p.IH * p.IW + 3) & ~3;
  ADM_REQUIRE(p.CS <= MAXQ * 256, "conv2d: input patch too large for the gather plan");
  const int n_pt = p.tiles_x * p.tiles_y * img_groups;
  ADM_REQUIRE(a.w_bstride == 0 || NI == 1, "conv2d: per-sample weights need one image per tile (output >= 16x8)");
  // cout tile: largest of 128/64/32 that still gives >= 1 workgroup per CU (256 CUs), else the smallest.
  int bm = 32;
  if (a.Cout % 128 == 0 && (long)n_pt * (a.Cout / 128) >= 256) bm = 128;
  else if (a.Cout % 64 == 0 && (long)n_pt * (a.Cout / 64) >= 256) bm = 64;
  // (Cout % 32 == 0 is required above, so the chosen tile always divides Cout: no cout guards in the kernels)
  p.n_ct = ceil_div(a.Cout, bm);
  p.nblk = n_pt * p.n_ct;
  const size_t smem = sizeof(float) * ((size_t)CK * p.CS + (size_t)CK * a.ks * a.ks * bm);
  g_last_variant = a.ks * 100 + a.stride * 10 + bm / 32;
  if (use_pf() && a.stride == 1 && a.ks == 3 && p.CS <= 256) {
    g_last_variant += 2000;
    return dispatch_pf<3>(p, bm, st);
  }
  // (1x1 on planes of <= 4x4 pixels with >= 64 input channels: the split generic kernel — the pipelined one has no split for 1x1, and a
  // 768 -> 256 shortcut at 4x4 pixels, B = 16, was 16 workgroups walking 24 chunks: 33 us for 0.1 GFLOP; a function of the layer only)
  static const int small1x1 = [] { const char* e = getenv("ADM_KSP_1X1_SMALL"); return e ? atoi(e) : 1; }();     // developer A/B
  const bool split_1x1 = small1x1 && a.ks == 1 && p.Ho * p.Wo <= 16 && Ct / CK >= 8 && a.w_bstride == 0;
  if (!split_1x1 && use_pf() && a.ks == 1 && p.CS == 128 && Ct % 32 == 0 && a.C1 % 32 == 0 && TW % 4 == 0 && p.Wo % 4 == 0 && a.W % 4 == 0) {
    g_last_variant += 2000;
    return dispatch_pf<1>(p, bm, st);
  }
  if (a.ks == 3 && a.stride == 1) return dispatch_bm<3, 1>(p, bm, smem, st);
  if (a.ks == 3 && a.stride == 2) return dispatch_bm<3, 2>(p, bm, smem, st);
  return dispatch_bm<1, 1>(p, bm, smem, st);
}

// (Cout,Cin,ks,ks) -> [Cin][tap][Cout]
__device__ __forceinline__ void pack_weight_body(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int KS2,
                                                 long first, long step) {
  const long total = (long)Cout * Cin * KS2;
  for (long i = first; i < total; i += step) {
    const int co = (int)(i % Cout);
    const long r = i / Cout;
    const int tap = (int)(r % KS2), c = (int)(r / KS2);
    wp[i] = w[((long)co * Cin + c) * KS2 + tap];
  }
}
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int KS2) {
  pack_weight_body(w, wp, Cout, Cin, KS2, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

// Backward-data weights: the gradient w.r.t. a conv's input is a "same" conv of dy with the channel-transposed,
// spatially flipped kernel: (Cout,Cin,ks,ks) -> [Cout][ks*ks][Cin] with wpT[co][t][c] = w[co][c][ks*ks-1-t].
__device__ __forceinline__ void pack_weight_T_body(const float* __restrict__ w, float* __restrict__ wpT, int Cout, int Cin, int KS2,
                                                   long first, long step) {
  const long total = (long)Cout * Cin * KS2;
  for (long i = first; i < total; i += step) {
    const int c = (int)(i % Cin);
    const long r = i / Cin;
    const int t = (int)(r % KS2), co = (int)(r / KS2);
    wpT[i] = w[((long)co * Cin + c) * KS2 + (KS2 - 1 - t)];
  }
}
__global__ void pack_weight_T_kernel(const float* __restrict__ w, float* __restrict__ wpT, int Cout, int Cin, int KS2) {
  pack_weight_T_body(w, wpT, Cout, Cin, KS2, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

// blockIdx.y = item of a device table (Net::refresh_weights after an optimizer step)
__global__ void __launch_bounds__(256) pack_weight_batch_kernel(const PackItem* __restrict__ items) {
  const PackItem it = items[blockIdx.y];
  const long first = (long)blockIdx.x * blockDim.x + threadIdx.x, step = (long)gridDim.x * blockDim.x;
  if (it.flag) pack_weight_T_body(it.src, (float*)it.dst, it.Cout, it.Cin, it.ks * it.ks, first, step);
  else pack_weight_body(it.src, (float*)it.dst, it.Cout, it.Cin, it.ks * it.ks, first, step);
}
__global__ void __launch_bounds__(256) copy_batch_kernel(const PackItem* __restrict__ items) {
  const PackItem it = items[blockIdx.y];
  float* d = (float*)it.dst;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < it.flag; i += (long)gridDim.x * blockDim.x) d[i] = it.src[i];
}
int launch_pack_conv_weight_batch(const PackItem* items_dev, int n, hipStream_t st) {
  if (n <= 0) return 0;
  ADM_LAUNCH(pack_weight_batch_kernel, dim3(64, (unsigned)n), dim3(256), 0, st, items_dev);
  return ADM_CHECK_LAUNCH();
}
int launch_copy_batch(const PackItem* items_dev, int n, hipStream_t st) {
  if (n <= 0) return 0;
  ADM_LAUNCH(copy_batch_kernel, dim3(16, (unsigned)n), dim3(256), 0, st, items_dev);
  return ADM_CHECK_LAUNCH();
}

int launch_pack_conv_weight_T(const float* w, float* wpT, int Cout, int Cin, int ks, hipStream_t st) {
  const long total = (long)Cout * Cin * ks * ks;
  long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  ADM_LAUNCH(pack_weight_T_kernel, dim3((unsigned)g), dim3(256), 0, st, w, wpT, Cout, Cin, ks * ks);
  return ADM_CHECK_LAUNCH();
}

int launch_pack_conv_weight(const float* w, float* wp, int Cout, int Cin, int ks, hipStream_t st) {
  const long total = (long)Cout * Cin * ks * ks;
  long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  ADM_LAUNCH(pack_weight_kernel, dim3((unsigned)g), dim3(256), 0, st, w, wp, Cout, Cin, ks * ks);
  return ADM_CHECK_LAUNCH();
}

}  // namespace adm
