// k_conv_wino.hip — Winograd F(2x2,3x3) variant of the fused 3x3 stride-1 convolution (fp32 throughout).
//
// Same fusions and the same LDS patch as k_conv_mfma.hip's pipelined kernel (virtual concat, nearest-x2 upsample,
// zero padding, GroupNorm affine + SiLU on load; bias + temb bias + residual in the epilogue), but the 9-tap
// correlation is replaced by 16 element-wise products in the Winograd domain: 2.25x fewer MFMA FLOPs per output.
//   Y = A^T [ (G g G^T) . (B^T d B) ] A        per 4x4 input tile d -> 2x2 output tile, summed over input channels
// GEMM view: for each of the 16 Winograd positions xi: M[xi][co][tile] += U[xi][co][c] * V[xi][c][tile].
// Workgroup (4 waves): 32 couts x one 8x16-pixel output tile = 32 Winograd tiles (4 rows x 8 cols); wave w owns
// xi = 4w..4w+3 (4 accumulator fragments of 32 couts x 32 tiles on v_mfma_f32_32x32x2_f32). Per K chunk of 8 channels:
//   stash  : prefetched raw activations -> GN/SiLU -> ldsX (haloed 10x18 patch per channel)           [barrier]
//   transf : one (channel, tile) per thread: V = B^T d B (32 adds) -> ldsV[buf][xi][c][tile]           [barrier]
//   issue  : next chunk's activations -> registers; next chunk's U slab -> LDS by global_load_lds (double buffer)
//   MFMA   : 4 xi x 4 channel pairs = 16 MFMAs per wave (A = U[xi][c][co], B = V[xi][c][tile])
// U is pre-transformed once per layer: [Cin][16][Cout] (pack_winograd_weight). Epilogue: all 16 M fragments go through
// LDS, each thread applies A^T M A for 4 (cout, tile) pairs and stores 2x2 pixels.
// Numerics: fp32 Winograd F(2,3) differs from direct summation by O(1e-6) relative — far inside the 1e-3 parity bar.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>

#include "adm_kernels.h"

namespace adm {

struct WinoParams {
  const float* x1; const float* x2; int C1, C2;
  int N, Hs, Ws, Hi, Wi, Ho, Wo, up;
  const float* gn_scale; const float* gn_shift; int act;
  const float* wu; const float* bias; int Cout;
  const float* chan_add; int chan_add_stride;
  const float* residual; float* out;
  int tiles_x, tiles_y, n_ct, nblk;
  long x1_bs, x2_bs;
  int gn_nstride;             // per-sample stride of gn_scale / gn_shift (0: shared identity rows, conv without GroupNorm)
  unsigned long long* prof;   // optional cycle counters of the wave-specialised kernel (ADM_WINO_PROF=1), else NULL
  double* stats;              // optional (v4): GroupNorm partial sums of the output, [n][cout][tile][2] (adm_conv_args.stats_out)
  int tune;                   // conv_wino5_kernel: developer switches (ADM_WINO5_TUNE; bit 0 = s_setprio 1 for waves 4-7)
};

__device__ __forceinline__ float silu_w(float v) { return v * ADM_RCP(1.0f + __expf(-v)); }

constexpr int WCK = 8;            // input channels per chunk
constexpr int WPH = 10, WPW = 18; // haloed patch of an 8x16 output tile
constexpr int WCS = WPH * WPW;    // 180
constexpr int WBM = 32;           // couts per workgroup
constexpr int WUSLAB = WCK * 16 * WBM;   // 4096 floats = 16 KiB
constexpr int WVSLAB = 16 * WCK * 32;    // 4096 floats

#if defined(ADM_EXPERIMENTS)   // superseded kernel generations (modes 1 and 2): built only with -DADM_EXPERIMENTS (build.sh ... exp)
template <bool HAS_CHAN, bool HAS_RES>
__device__ __forceinline__ void wino_store(const WinoParams& p, const float* ldsM, int tid, int m0, int n, int ty0,
                                           int tx0) {
  const long planeO = (long)p.Ho * p.Wo;
  ADM_UNROLL
  for (int k = 0; k < 4; ++k) {
    const int pair = tid + 256 * k;           // 1024 (cout, tile) pairs
    const int co_l = pair >> 5, tile = pair & 31;
    const int tyy = tile >> 3, txx = tile & 7;
    float m[16];
    ADM_UNROLL
    for (int xi = 0; xi < 16; ++xi) m[xi] = ldsM[(xi * WBM + co_l) * 32 + tile];
    // Y = A^T M A with A^T = [[1,1,1,0],[0,1,-1,-1]]
    float t0[4], t1[4];
    ADM_UNROLL
    for (int j = 0; j < 4; ++j) {
      t0[j] = m[0 * 4 + j] + m[1 * 4 + j] + m[2 * 4 + j];
      t1[j] = m[1 * 4 + j] - m[2 * 4 + j] - m[3 * 4 + j];
    }
    float y[2][2];
    y[0][0] = t0[0] + t0[1] + t0[2]; y[0][1] = t0[1] - t0[2] - t0[3];
    y[1][0] = t1[0] + t1[1] + t1[2]; y[1][1] = t1[1] - t1[2] - t1[3];
    const int co = m0 + co_l;
    const float b = p.bias[co] + (HAS_CHAN ? p.chan_add[(long)n * p.chan_add_stride + co] : 0.f);
    const int oy = ty0 + 2 * tyy, ox = tx0 + 2 * txx;
    ADM_UNROLL
    for (int a = 0; a < 2; ++a) {
      const long o = ((long)n * p.Cout + co) * planeO + (long)(oy + a) * p.Wo + ox;
      float2 v = make_float2(y[a][0] + b, y[a][1] + b);
      if (HAS_RES) {
        const float2 r = *reinterpret_cast<const float2*>(p.residual + o);
        v.x += r.x; v.y += r.y;
      }
      *reinterpret_cast<float2*>(p.out + o) = v;
    }
  }
}

__global__ void __launch_bounds__(256, 2) conv_wino_kernel(const WinoParams p) {
  ADM_DYN_SMEM(float, smem);
  float* ldsX = smem;                       // WCK * 180  (padded to 1472)
  float* ldsV = smem + 1472;                // 2 * WVSLAB
  float* ldsU = ldsV + 2 * WVSLAB;          // 2 * WUSLAB
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  int lid;
  {
    const int b = blockIdx.x, q = p.nblk >> 3, r = p.nblk & 7, xcd = b & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int ct = lid % p.n_ct, pt = lid / p.n_ct;
  const int tx = pt % p.tiles_x, ty = (pt / p.tiles_x) % p.tiles_y, n = pt / (p.tiles_x * p.tiles_y);
  const int m0 = ct * WBM;
  const int Ct = p.C1 + p.C2;
  const int planeS = p.Hs * p.Ws;

  // gather plan: one patch element per thread (180 of 256 threads)
  const bool qv = tid < WCS;
  int soff = -1;
  if (qv) {
    const int ly = tid / WPW, lx = tid - ly * WPW;
    const int gy = ty * 8 + ly - 1, gx = tx * 16 + lx - 1;
    if (gy >= 0 && gy < p.Hi && gx >= 0 && gx < p.Wi) {
      const int sy = p.up ? (gy >> 1) : gy, sx = p.up ? (gx >> 1) : gx;
      soff = sy * p.Ws + sx;
    }
  }
  // transform role: channel tc, Winograd tile tt (row tt>>3, col tt&7) -> patch origin (2*row, 2*col)
  const int tc = tid >> 5, tt = tid & 31;
  const int torg = tc * WCS + (2 * (tt >> 3)) * WPW + 2 * (tt & 7);

  f32x16 acc[4];
  ADM_UNROLL
  for (int a = 0; a < 4; ++a)
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  const bool has_gn = p.gn_scale != nullptr;
  float xr[WCK], gs[WCK], gh[WCK];
  ADM_UNROLL
  for (int c = 0; c < WCK; ++c) { xr[c] = 0.f; gs[c] = 1.f; gh[c] = 0.f; }

  auto issue = [&](int c0, int buf) {
    const bool from1 = c0 < p.C1;
    const float* xb = from1 ? p.x1 : p.x2;
    const long xbs = from1 ? p.x1_bs : p.x2_bs;
    const int cb0 = from1 ? c0 : c0 - p.C1;
    if (soff >= 0) {
      const float* src = xb + (long)n * xbs + (long)cb0 * planeS + soff;
      ADM_UNROLL
      for (int c = 0; c < WCK; ++c) xr[c] = src[(long)c * planeS];
      if (has_gn) {
        const float* sp = p.gn_scale + (long)n * Ct + c0;
        const float* hp = p.gn_shift + (long)n * Ct + c0;
        ADM_UNROLL
        for (int c = 0; c < WCK; ++c) { gs[c] = sp[c]; gh[c] = hp[c]; }
      }
    }
    // U slab of this chunk: rows (c, xi) of WBM couts; [Cin][16][Cout] in global
    const float* usrc = p.wu + (long)c0 * 16 * p.Cout + m0;
    float* udst = ldsU + buf * WUSLAB;
    ADM_UNROLL
    for (int i = 0; i < 4; ++i) {  // 1024 float4 = 4 per thread
      const int idx = tid + 256 * i;
      const int row = idx >> 3, c4 = idx & 7;
      ADM_GLDS16(usrc + (long)row * p.Cout + c4 * 4, udst + (256 * i + wave * 64) * 4);
    }
  };

  const int nchunks = Ct / WCK;
  issue(0, 0);
  for (int ci = 0; ci < nchunks; ++ci) {
    if (qv) {
      const bool live = soff >= 0;
      ADM_UNROLL
      for (int c = 0; c < WCK; ++c) {
        float v = xr[c] * gs[c] + gh[c];
        const float sv = silu_w(v);
        v = p.act ? sv : v;
        ldsX[c * WCS + tid] = live ? v : 0.f;
      }
    }
    __syncthreads();  // patch visible; previous chunk's MFMAs done everywhere (V[buf^1], U[buf^1] free); U DMA of ci landed
    {
      // V = B^T d B for (channel tc, tile tt); B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
      float d[4][4];
      ADM_UNROLL
      for (int i = 0; i < 4; ++i)
        ADM_UNROLL
        for (int j = 0; j < 4; ++j) d[i][j] = ldsX[torg + i * WPW + j];
      float t[4][4];
      ADM_UNROLL
      for (int j = 0; j < 4; ++j) {
        t[0][j] = d[0][j] - d[2][j];
        t[1][j] = d[1][j] + d[2][j];
        t[2][j] = d[2][j] - d[1][j];
        t[3][j] = d[1][j] - d[3][j];
      }
      float* vdst = ldsV + (ci & 1) * WVSLAB + tc * 32 + tt;
      ADM_UNROLL
      for (int i = 0; i < 4; ++i) {
        vdst[(i * 4 + 0) * (WCK * 32)] = t[i][0] - t[i][2];
        vdst[(i * 4 + 1) * (WCK * 32)] = t[i][1] + t[i][2];
        vdst[(i * 4 + 2) * (WCK * 32)] = t[i][2] - t[i][1];
        vdst[(i * 4 + 3) * (WCK * 32)] = t[i][1] - t[i][3];
      }
    }
    __syncthreads();  // V of this chunk visible; ldsX free for the next stash
    if (ci + 1 < nchunks) issue((ci + 1) * WCK, (ci + 1) & 1);
    const float* U = ldsU + (ci & 1) * WUSLAB;
    const float* V = ldsV + (ci & 1) * WVSLAB;
    ADM_UNROLL
    for (int a = 0; a < 4; ++a) {
      const int xi = wave * 4 + a;
      ADM_UNROLL
      for (int cp = 0; cp < WCK / 2; ++cp) {
        const int ch = 2 * cp + h;
        const float av = U[(ch * 16 + xi) * WBM + l31];
        const float bv = V[(xi * WCK + ch) * 32 + l31];
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
      }
    }
  }
  __syncthreads();
  // ---- epilogue: M fragments -> LDS [xi][cout][tile], inverse transform, fused bias/temb/residual, 2x2 stores -----
  float* ldsM = smem;  // 16*32*32 floats = 64 KiB (the launch reserves max(main, epilogue))
  ADM_UNROLL
  for (int a = 0; a < 4; ++a) {
    const int xi = wave * 4 + a;
    ADM_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int co_l = (r & 3) + 8 * (r >> 2) + 4 * h;
      ldsM[(xi * WBM + co_l) * 32 + l31] = acc[a][r];
    }
  }
  __syncthreads();
  if (p.chan_add != nullptr) {
    if (p.residual != nullptr) wino_store<true, true>(p, ldsM, tid, m0, n, ty * 8, tx * 16);
    else wino_store<true, false>(p, ldsM, tid, m0, n, ty * 8, tx * 16);
  } else {
    if (p.residual != nullptr) wino_store<false, true>(p, ldsM, tid, m0, n, ty * 8, tx * 16);
    else wino_store<false, false>(p, ldsM, tid, m0, n, ty * 8, tx * 16);
  }
}


// ---------------------------------------------------------------------------------------------------------
// v2: wave-specialised Winograd kernel. 512 threads = 8 waves per workgroup (one workgroup per CU):
//   waves 0-3 (consumers): wave w owns xi = 4w..4w+3 x two 32-cout groups = 8 accumulator fragments (128 VGPRs) and
//       issues nothing but LDS operand reads and 32 MFMAs per 8-channel chunk;
//   waves 4-7 (producers): thread (channel c = t>>5, Winograd tile = t&31) loads the 4x4 input window of the NEXT chunk
//       straight from global memory (L1/L2 absorb the 4x window overlap), applies GroupNorm affine + SiLU + zero padding,
//       transforms it (V = B^T d B) in registers and writes the 16 V values to the other half of a double buffer; the
//       same waves stream the next chunk's U slab (8 ch x 16 xi x 64 couts) L2 -> LDS with global_load_lds.
// One __syncthreads per chunk; MFMA and VALU/LDS/VMEM pipes of each SIMD are fed by different waves, so the matrix
// pipe only waits when the producers are slower than 32 MFMAs (2048 cycles).
constexpr int W2BM = 64;
constexpr int W2USLAB = WCK * 16 * W2BM;   // 8192 floats = 32 KiB
constexpr int W2VSLAB = 16 * WCK * 32;     // 4096 floats = 16 KiB

template <bool HAS_CHAN, bool HAS_RES>
__device__ __forceinline__ void wino2_store(const WinoParams& p, const float* ldsM, int tid, int m0, int n, int ty0,
                                            int tx0) {
  const long planeO = (long)p.Ho * p.Wo;
  ADM_UNROLL
  for (int k = 0; k < 4; ++k) {
    const int pair = tid + 512 * k;            // 2048 (cout, tile) pairs
    const int co_l = pair >> 5, tile = pair & 31;
    const int tyy = tile >> 3, txx = tile & 7;
    float m[16];
    ADM_UNROLL
    for (int xi = 0; xi < 16; ++xi) m[xi] = ldsM[(xi * W2BM + co_l) * 32 + tile];
    float t0[4], t1[4];
    ADM_UNROLL
    for (int j = 0; j < 4; ++j) {
      t0[j] = m[0 * 4 + j] + m[1 * 4 + j] + m[2 * 4 + j];
      t1[j] = m[1 * 4 + j] - m[2 * 4 + j] - m[3 * 4 + j];
    }
    float y[2][2];
    y[0][0] = t0[0] + t0[1] + t0[2]; y[0][1] = t0[1] - t0[2] - t0[3];
    y[1][0] = t1[0] + t1[1] + t1[2]; y[1][1] = t1[1] - t1[2] - t1[3];
    const int co = m0 + co_l;
    const float b = p.bias[co] + (HAS_CHAN ? p.chan_add[(long)n * p.chan_add_stride + co] : 0.f);
    const int oy = ty0 + 2 * tyy, ox = tx0 + 2 * txx;
    ADM_UNROLL
    for (int a = 0; a < 2; ++a) {
      const long o = ((long)n * p.Cout + co) * planeO + (long)(oy + a) * p.Wo + ox;
      float2 v = make_float2(y[a][0] + b, y[a][1] + b);
      if (HAS_RES) {
        const float2 r = *reinterpret_cast<const float2*>(p.residual + o);
        v.x += r.x; v.y += r.y;
      }
      *reinterpret_cast<float2*>(p.out + o) = v;
    }
  }
}

struct Wino2Geom {
  int n, ty, tx, m0, Ct, planeS, nchunks;
};

// Producer role (waves 4..7): stage V (transformed activations) and U (filters) of chunk c+1 while chunk c is consumed.
// Barrier protocol (every wave of the workgroup executes the same NUMBER of barriers): 1 after the prologue, 1 per
// chunk, 1 after the consumers' fragment stash.
template <bool HAS_GN>
__device__ __forceinline__ void wino2_producer(const WinoParams& p, const Wino2Geom& g, float* ldsV, float* ldsU,
                                               int tid, int wave) {
  const int pt_id = tid & 255;
  const int pc = pt_id >> 5, ptile = pt_id & 31;
  // source offsets of the 4x4 window inside a channel plane (clamped to 0 where the window leaves the image: the load
  // is then unconditional — the counted barrier below relies on an exact VMEM instruction count — and the value is
  // zeroed after the activation through `wvalid`)
  int woff[16];
  unsigned wvalid = 0;
  {
    const int gy0 = g.ty * 8 + 2 * (ptile >> 3) - 1, gx0 = g.tx * 16 + 2 * (ptile & 7) - 1;
    ADM_UNROLL
    for (int i = 0; i < 4; ++i)
      ADM_UNROLL
      for (int j = 0; j < 4; ++j) {
        const int gy = gy0 + i, gx = gx0 + j;
        const bool ok = gy >= 0 && gy < p.Hi && gx >= 0 && gx < p.Wi;
        const int sy = p.up ? (gy >> 1) : gy, sx = p.up ? (gx >> 1) : gx;
        woff[i * 4 + j] = ok ? sy * p.Ws + sx : 0;
        wvalid |= ok ? 1u << (i * 4 + j) : 0u;
      }
  }
  constexpr bool has_gn = HAS_GN;         // compile-time: the producer's VMEM instruction count must be exact
  // two window register sets: while set A (chunk c+1) is transformed, set B (chunk c+2) is already in flight
  float xwA[16], xwB[16];
  float gscA = 1.f, gshA = 0.f, gscB = 1.f, gshB = 0.f;
  auto load_window = [&](int ci, float (&xw)[16], float& gsc, float& gsh) {   // raw 4x4 window + GN scale/shift
    const int cc = ci * WCK + pc;
    const float* src = cc < p.C1 ? p.x1 + (long)g.n * p.x1_bs + (long)cc * g.planeS
                                 : p.x2 + (long)g.n * p.x2_bs + (long)(cc - p.C1) * g.planeS;
    ADM_UNROLL
    for (int i = 0; i < 16; ++i) xw[i] = src[woff[i]];
    if (has_gn) { gsc = p.gn_scale[(long)g.n * g.Ct + cc]; gsh = p.gn_shift[(long)g.n * g.Ct + cc]; }
  };
  auto transform_store = [&](int ci, const float (&xw)[16], float gsc, float gsh) {   // window -> ldsV[ci & 1]
    float d[16];
    ADM_UNROLL
    for (int i = 0; i < 16; ++i) {
      float v = xw[i] * gsc + gsh;
      const float sv = silu_w(v);
      v = p.act ? sv : v;
      d[i] = ((wvalid >> i) & 1u) ? v : 0.f;     // zero padding is applied after the activation
    }
    float t[4][4];
    ADM_UNROLL
    for (int j = 0; j < 4; ++j) {
      t[0][j] = d[0 * 4 + j] - d[2 * 4 + j];
      t[1][j] = d[1 * 4 + j] + d[2 * 4 + j];
      t[2][j] = d[2 * 4 + j] - d[1 * 4 + j];
      t[3][j] = d[1 * 4 + j] - d[3 * 4 + j];
    }
    float* vdst = ldsV + (ci & 1) * W2VSLAB + pc * 32 + ptile;
    ADM_UNROLL
    for (int i = 0; i < 4; ++i) {
      vdst[(i * 4 + 0) * (WCK * 32)] = t[i][0] - t[i][2];
      vdst[(i * 4 + 1) * (WCK * 32)] = t[i][1] + t[i][2];
      vdst[(i * 4 + 2) * (WCK * 32)] = t[i][2] - t[i][1];
      vdst[(i * 4 + 3) * (WCK * 32)] = t[i][1] - t[i][3];
    }
  };
  // prologue: chunk 0 transformed, chunk 1 window in flight
  load_window(0, xwA, gscA, gshA);
  transform_store(0, xwA, gscA, gshA);
  load_window(1, xwB, gscB, gshB);
  ADM_BARRIER_KEEP_VMEM(63);
  // The chunk loop is unrolled by two (nchunks is even, checked by the launcher) with its last pair peeled, so the
  // instruction stream is branch-free: the window loads of chunk c+2 go into the register set that is NOT being
  // transformed and stay in flight across the barrier (vmcnt(63) = no vector-memory wait; the compiler's own counted
  // waits sit at the first use). The producers issue no LDS-DMA: next to one, hipcc waits vmcnt(0) for every plain load.
  int ci = 0;
  for (; ci + 2 < g.nchunks; ci += 2) {
    load_window(ci + 2, xwA, gscA, gshA);
    transform_store(ci + 1, xwB, gscB, gshB);
    ADM_BARRIER_KEEP_VMEM(63);
    load_window(ci + 3, xwB, gscB, gshB);
    transform_store(ci + 2, xwA, gscA, gshA);
    ADM_BARRIER_KEEP_VMEM(63);
  }
  transform_store(ci + 1, xwB, gscB, gshB);   // last pair: chunk nchunks-1 is the only operand still to be staged
  ADM_BARRIER_KEEP_VMEM(0);                // chunk ci consumed, chunk ci+1 staged
  ADM_BARRIER_KEEP_VMEM(0);                // chunk ci+1 consumed: operand buffers are dead
  ADM_BARRIER_KEEP_VMEM(0);                // consumers' fragments are in ldsM
}

// Consumer role (waves 0..3): wave w owns transform points 4w..4w+3, all 64 couts, all 32 tiles: 8 accumulator
// fragments (128 VGPRs), 32 MFMAs per chunk.
__device__ __forceinline__ void wino2_consumer(const WinoParams& p, const Wino2Geom& g, const float* ldsV, float* ldsU,
                                               float* ldsM, int tid, int wave) {
  const int lane = tid & 63;
  const int l31 = lane & 31, h = lane >> 5;
  auto issue_u = [&](int ci) {             // U slab of chunk ci -> ldsU[ci & 1]: 2048 float4 by 256 threads, LDS-DMA
    const float* usrc = p.wu + (long)ci * WCK * 16 * p.Cout + g.m0;
    float* udst = ldsU + (ci & 1) * W2USLAB;
    ADM_UNROLL
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx >> 4, c4 = idx & 15;
      ADM_GLDS16(usrc + (long)row * p.Cout + c4 * 4, udst + (256 * i + wave * 64) * 4);
    }
  };
  issue_u(0);
  f32x16 acc[4][2];
  ADM_UNROLL
  for (int a = 0; a < 4; ++a)
    ADM_UNROLL
    for (int f = 0; f < 2; ++f)
      ADM_UNROLL
      for (int r = 0; r < 16; ++r) acc[a][f][r] = 0.f;
  // Per chunk: 4 transform points x 4 channel pairs x 2 cout fragments = 32 MFMAs on 48 operand words. The words are
  // read into registers FIRST, then the barrier (which frees both operand buffers of this chunk for the producers and
  // for the next-but-one U slab), then the LDS-DMA of chunk c+2, then the MFMAs from registers: the matrix pipe works
  // while the producers transform and the DMA flies, and the LDS latency is paid once per chunk instead of per MFMA pair
  // (read -> wait -> MFMA pair interleaving held the pipe at ~45 %).
  ADM_BARRIER_KEEP_VMEM(0);                // prologue done: U(0), V(0) in place
  if (g.nchunks > 1) issue_u(1);
  for (int ci = 0; ci < g.nchunks; ++ci) {
    const float* U = ldsU + (ci & 1) * W2USLAB;
    const float* V = ldsV + (ci & 1) * W2VSLAB;
    float bv[4][WCK / 2], a0[4][WCK / 2], a1[4][WCK / 2];
    ADM_UNROLL
    for (int a = 0; a < 4; ++a) {
      const int xi = wave * 4 + a;
      ADM_UNROLL
      for (int cp = 0; cp < WCK / 2; ++cp) {
        const int ch = 2 * cp + h;
        bv[a][cp] = V[(xi * WCK + ch) * 32 + l31];
        a0[a][cp] = U[(ch * 16 + xi) * W2BM + l31];
        a1[a][cp] = U[(ch * 16 + xi) * W2BM + 32 + l31];
      }
    }
    ADM_BARRIER_KEEP_VMEM(0);              // operands of chunk ci are in registers; U(ci+1) (this wave's part) landed
    if (ci + 2 < g.nchunks) issue_u(ci + 2);
    ADM_SCHED_FENCE();
    ADM_UNROLL
    for (int a = 0; a < 4; ++a) {
      ADM_UNROLL
      for (int cp = 0; cp < WCK / 2; ++cp) {
        acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[a][cp], bv[a][cp], acc[a][0], 0, 0, 0);
        acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[a][cp], bv[a][cp], acc[a][1], 0, 0, 0);
      }
    }
  }
  // fragments -> ldsM [xi][cout][tile] (aliases the operand buffers, dead after the barrier above)
  ADM_UNROLL
  for (int a = 0; a < 4; ++a) {
    const int xi = wave * 4 + a;
    ADM_UNROLL
    for (int f = 0; f < 2; ++f)
      ADM_UNROLL
      for (int r = 0; r < 16; ++r) {
        const int co_l = f * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        ldsM[(xi * W2BM + co_l) * 32 + l31] = acc[a][f][r];
      }
  }
  ADM_BARRIER_KEEP_VMEM(0);
}

__global__ void __launch_bounds__(512, 2) conv_wino2_kernel(const WinoParams p) {
  ADM_DYN_SMEM(float, smem);
  float* ldsV = smem;                       // 2 * W2VSLAB
  float* ldsU = smem + 2 * W2VSLAB;         // 2 * W2USLAB
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  int lid;
  {
    const int b = blockIdx.x, q = p.nblk >> 3, r = p.nblk & 7, xcd = b & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int ct = lid % p.n_ct, pt = lid / p.n_ct;
  Wino2Geom g;
  g.tx = pt % p.tiles_x; g.ty = (pt / p.tiles_x) % p.tiles_y; g.n = pt / (p.tiles_x * p.tiles_y);
  g.m0 = ct * W2BM;
  g.Ct = p.C1 + p.C2;
  g.planeS = p.Hs * p.Ws;
  g.nchunks = g.Ct / WCK;
  // roles are wave-uniform; each role has its own register allocation (the accumulators live only in the consumers)
  if (wave >= 4) {
    if (p.gn_scale != nullptr) wino2_producer<true>(p, g, ldsV, ldsU, tid, wave);
    else wino2_producer<false>(p, g, ldsV, ldsU, tid, wave);
  } else wino2_consumer(p, g, ldsV, ldsU, smem, tid, wave);
  // ---- epilogue: inverse transform of ldsM (128 KiB) by all 512 threads ------------------------------------------
  const float* ldsM = smem;
  if (p.chan_add != nullptr) {
    if (p.residual != nullptr) wino2_store<true, true>(p, ldsM, tid, g.m0, g.n, g.ty * 8, g.tx * 16);
    else wino2_store<true, false>(p, ldsM, tid, g.m0, g.n, g.ty * 8, g.tx * 16);
  } else {
    if (p.residual != nullptr) wino2_store<false, true>(p, ldsM, tid, g.m0, g.n, g.ty * 8, g.tx * 16);
    else wino2_store<false, false>(p, ldsM, tid, g.m0, g.n, g.ty * 8, g.tx * 16);
  }
}


#endif  // ADM_EXPERIMENTS (v1, v2)
// ---------------------------------------------------------------------------------------------------------------------
// v3 — persistent, wave-specialised Winograd kernel (mode 3). What v2's measurements asked for:
//   * the producers' per-thread 4x4 window gathers (16 dword loads, every input pixel fetched 4x, GN+SiLU applied 4x)
//     saturated the CU's vector-memory path (-30 % when ablated): the raw haloed patch of a chunk (8 ch x 10 x 18) is now
//     fetched ONCE with float4 row loads two chunks ahead, activated once, and staged in a small LDS patch buffer from
//     which the 4x4 windows are read;
//   * the 128 KiB LDS round trip of the inverse transform and the per-tile prologue/epilogue bubble (1 workgroup per CU,
//     nothing to overlap with) are gone: consumers use v_mfma_f32_16x16x4_f32 with wave w owning ALL 16 Winograd points
//     of a 32-cout x 16-tile sub-block, so A^T M A is lane-local (the 16 points of a (cout, tile) pair sit in the same
//     lane/register slot of 16 accumulators) and outputs go straight from registers to HBM; workgroups are persistent
//     (grid = #CUs, tiles strided) and the producers run into the next tile while the consumers finish the current one;
//   * operand words are read 4 Winograd points ahead of the MFMAs that use them (rolling 24-register window), the
//     per-chunk barrier sits where the last read of the chunk has long landed, so the matrix pipe never waits on LDS.
// Barrier protocol (one workgroup barrier per 8-channel chunk; G = running chunk index over all tiles of the block):
//   barrier G certifies  (a) V(G+1) is complete [producers], (b) every consumer has read chunk G into registers,
//                        (c) each consumer's part of U(G+1) has landed (vmcnt(0) before its barrier).
//   After it the producers write V(G+2) and the consumers DMA U(G+2) into the buffers chunk G occupied.
// LDS: V 2x16 KiB + U 2x32 KiB + patch 2x5.6 KiB = 107.25 KiB. V and U images are swizzled by 16 words on odd channels so
// the four k-rows of a 16x16x4 operand read hit disjoint banks.
constexpr int W3BM = 64;
constexpr int W3USLAB = WCK * 16 * W3BM;    // 8192 floats
constexpr int W3VSLAB = 16 * WCK * 32;      // 4096 floats
// LDS pitch of a patch row in the wave-specialised kernels: 24 words instead of the 18 the patch is wide. Stage C reads the 4x4 windows of
// a channel's 32 tiles with ds_read2_b64 at word offsets 2 tyy P + 2 txx: with P = 18 the four tile rows start at banks 0 / 36 / 8 / 44
// and overlap pairwise (the 0.23 LDS conflict ratio of rounds 2-3); with P = 24 they start at 0 / 48 / 32 / 16 — conflict-free.
constexpr int WPP = 24;
constexpr int W3PSLAB = WCK * WPH * WPP + 264;   // 1920 floats (x2-upsample variant: 480) + one dummy word per producer lane
constexpr int W3LDS = 2 * W3VSLAB + 2 * W3USLAB + 2 * W3PSLAB;

struct Wino3Tile { int n, ty, tx, m0; };

__device__ __forceinline__ Wino3Tile wino3_tile(const WinoParams& p, int v) {
  // bijective XCD-aware remap of the virtual block id (v & 7 == XCD of the persistent block that owns it)
  const int q = p.nblk >> 3, r = p.nblk & 7, xcd = v & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
  const int ct = lid % p.n_ct, pt = lid / p.n_ct;
  Wino3Tile t;
  t.tx = pt % p.tiles_x; t.ty = (pt / p.tiles_x) % p.tiles_y; t.n = pt / (p.tiles_x * p.tiles_y);
  t.m0 = ct * W3BM;
  return t;
}

#if defined(ADM_EMU)
#define W3_CLK() 0ull
#define W3_BARRIER(N, prof, d, b) ADM_BARRIER_KEEP_VMEM(N)
#else
#define W3_CLK() ((unsigned long long)__builtin_readcyclecounter())
// barrier with optional accounting of the cycles spent in the counter drain (slot d) and in the barrier itself (slot b)
#define W3_BARRIER(N, prof, d, b)                                   \
  do {                                                              \
    if (PROF) {                                                     \
      const unsigned long long t0_ = W3_CLK();                      \
      asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory"); \
      const unsigned long long t1_ = W3_CLK();                      \
      ADM_BARRIER_KEEP_VMEM(N);                                     \
      const unsigned long long t2_ = W3_CLK();                      \
      (prof)[d] += t1_ - t0_; (prof)[b] += t2_ - t1_;               \
    } else {                                                        \
      ADM_BARRIER_KEEP_VMEM(N);                                     \
    }                                                               \
  } while (0)
#endif

// ---- producer role: 256 threads (waves 4..7) ------------------------------------------------------------------------
// Issue budget: a wave issues at most one instruction every ~4 cycles, so a producer wave has ~400 issue slots per
// 2048-cycle chunk and every scalar/branch/address instruction counts. Hence: per-tile (not per-chunk) 32-bit element
// offsets against a wave-uniform chunk base pointer, no per-lane predication (disabled lanes write to dummy LDS words,
// whole-wave roles are scalar branches), the tile cursor's integer divisions behind a real (non-speculated) branch.
struct Wino3Raw {                     // one chunk's raw activations of this thread, prefetched two chunks ahead
  float4 a, b;                        // item 0 / item 1 when it is a float4 row piece (UP: scalars in .x)
  float h;                            // item 1 when it is a halo element
  float sc0, sh0, sc1, sh1;           // GroupNorm scale / shift of the two items' channels
  unsigned ok;                        // bit k: item k lies inside the image (zero padding otherwise)
};

// V4 = true: the variant conv_wino4_kernel uses — V images unswizzled (its consumers read whole 128-byte channel rows with
// ds_read_b64, which is conflict-free as it is) and the raw activations prefetched FOUR chunks ahead instead of two (the
// registers are free: the kernel's allocation is set by the consumers' accumulators; with two chunks of ~3000 cycles in
// flight a producer is bound by the loaded HBM latency: measured ~2700 cycles per chunk with the MFMAs removed).
constexpr bool wino_abl_idle(int abl) { return abl == 1 || abl == 7 || abl == 9 || abl == 10 || abl == 11; }
// ABL (developer aid, timing only — results are wrong): 1 / 7 / 9 / 10 / 11 = this role keeps its barriers but stages nothing; 4 = stage C
// (window gather + transform + V write) skipped; 5 = stage B (activation + patch write) skipped.
// ACT: -1 = p.act decides at run time (v3); 0 / 1 = compiled without / with SiLU (v4: one select per element less).
// PAIR (conv_wino4_kernel only): ONE workgroup barrier per TWO chunks. The V slabs and the patch buffers become rings of four, a
// producer interval stages V(g), V(g + 1) [stage C twice], then the patches of g + 2, g + 3 [stage B twice] and the global loads of
// g + 6, g + 7, and only then meets the consumers — who by then have read chunks g - 2, g - 1 and go on to g, g + 1. Same arithmetic,
// same summation order (bit-identical to the one-chunk cadence); what changes is how often the two roles wait for each other
// (round 2's accounting: 9.5 % of the consumers' and 12 % of the producers' cycles are barrier waits at one barrier per chunk).
template <bool UP, bool WIDE1, bool PROF, bool V4 = false, int ABL = 0, int ACT = -1, bool PAIR = false>
__device__ __forceinline__ void wino3_producer(const WinoParams& p, float* ldsV, float* ldsP, int tid, int b0, int bs) {
  constexpr int RING = PAIR ? 3 : 1;          // buffer index mask: rings of four / two
  unsigned long long pr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_start = W3_CLK();
  const int Ct = p.C1 + p.C2;
  const int planeS = p.Hs * p.Ws;
  const int nch = Ct / WCK;
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
  const int vofs = V4 ? pc * 32 + ptile : pc * 32 + ((ptile + 16 * (pc & 1)) & 31);   // + xi * 256

  // ---- stage A cursor: (tile, chunk) of the next global load ----------------------------------------------------------------
  int a_v = b0, a_ci = 0, a_left = total;
  int a_off0 = 0, a_off1 = 0;         // element offset of the items inside the sample: channel plane + row + column
  unsigned a_ok = 0;
  const float *a_x1 = nullptr, *a_x2 = nullptr, *a_gs = nullptr, *a_gh = nullptr;   // per-tile wave-uniform bases
#if !defined(ADM_EMU)
  // v4: the same bases as buffer resources (SGPR quads). A buffer load takes the per-lane byte offset as a 32-bit VGPR and the chunk's
  // offset as an SGPR, so the per-load 64-bit address arithmetic (sign extension + v_lshl_add_u64: ~10 VALU per chunk) leaves the
  // producers' instruction stream — which is what the co-resident MFMA wave pays for (profiles/r04_wino.md).
  __amdgpu_buffer_rsrc_t a_rx1, a_rx2, a_rgs, a_rgh;
  int a_vo0 = 0, a_vo1 = 0;
  const int ch_vo0 = it_ch[0] * 4, ch_vo1 = it_ch[1] * 4;
#endif
  auto a_geometry = [&]() {
    const Wino3Tile t = wino3_tile(p, a_v);
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
    if constexpr (V4) {
      a_rx1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_x1), (short)0, 0x7fffffff, 0x00027000);
      a_rx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_x2), (short)0, 0x7fffffff, 0x00027000);
      a_rgs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_gs), (short)0, 0x7fffffff, 0x00027000);
      a_rgh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_gh), (short)0, 0x7fffffff, 0x00027000);
      a_vo0 = a_off0 * 4; a_vo1 = a_off1 * 4;
    }
#endif
  };
  a_geometry();
  auto stage_a = [&](Wino3Raw& r) {           // issue the global loads of chunk (a_v, a_ci); then advance the cursor
    const int c0 = a_ci * WCK;
#if !defined(ADM_EMU)
    if constexpr (V4) {
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
      r.ok = a_ok;
      if (a_left > 1) {
        --a_left;
        if (++a_ci == nch) {
          ADM_SCHED_FENCE();
          a_ci = 0; a_v += bs;
          a_geometry();
        }
      }
      return;
    }
#endif
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
  const bool act_on = ACT < 0 ? p.act != 0 : ACT != 0;
  auto act1 = [&](float x, float sc, float sh, unsigned ok) {   // GroupNorm affine (+ SiLU); zero padding applies after it
    if (V4) {
      // the caller has zeroed scale AND shift of an out-of-image item: the affine then gives 0, and SiLU(0) = 0 — no per-element
      // select (the padded positions read clamped, i.e. real and finite, activations)
      const float v0 = x * sc + sh;
      return act_on ? silu_w(v0) : v0;
    }
    const float v = x * sc + sh;
    const float a = act_on ? silu_w(v) : v;
    return ok ? a : 0.f;
  };
  auto stage_b = [&](const Wino3Raw& r0_, int g) {        // raw -> activation -> patch buffer g & 1
    float* P = ldsP + (g & RING) * W3PSLAB;
    Wino3Raw r = r0_;
    if (V4) {                                             // zero padding as a zeroed affine: two selects per ITEM
      r.sc0 = (r.ok & 1u) ? r.sc0 : 0.f; r.sh0 = (r.ok & 1u) ? r.sh0 : 0.f;
      r.sc1 = (r.ok & 2u) ? r.sc1 : 0.f; r.sh1 = (r.ok & 2u) ? r.sh1 : 0.f;
    }
    if (UP) {
      P[it_pofs[0]] = act1(r.a.x, r.sc0, r.sh0, r.ok & 1u);
      P[it_pofs[1]] = act1(r.b.x, r.sc1, r.sh1, r.ok & 2u);
    } else {
      float* P0 = P + it_pofs[0];
      float* P1 = P + it_pofs[1];
#if !defined(ADM_EMU)
      if constexpr (V4) {
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
        else P1[0] = act1(r.h, r.sc1, r.sh1, r.ok & 2u);
        return;
      }
#endif
      P0[0] = act1(r.a.x, r.sc0, r.sh0, r.ok & 1u); P0[1] = act1(r.a.y, r.sc0, r.sh0, r.ok & 1u);
      P0[2] = act1(r.a.z, r.sc0, r.sh0, r.ok & 1u); P0[3] = act1(r.a.w, r.sc0, r.sh0, r.ok & 1u);
      if (wide1) {
        P1[0] = act1(r.b.x, r.sc1, r.sh1, r.ok & 2u); P1[1] = act1(r.b.y, r.sc1, r.sh1, r.ok & 2u);
        P1[2] = act1(r.b.z, r.sc1, r.sh1, r.ok & 2u); P1[3] = act1(r.b.w, r.sc1, r.sh1, r.ok & 2u);
      } else {
        P1[0] = act1(r.h, r.sc1, r.sh1, r.ok & 2u);
      }
    }
  };
  auto stage_c = [&](int g) {                // patch g & 1 -> 4x4 window -> V = B^T d B -> V buffer g & 1
    const float* P = ldsP + (g & RING) * W3PSLAB + wbase;
    float d[16];
    ADM_UNROLL
    for (int i = 0; i < 4; ++i)
      ADM_UNROLL
      for (int j = 0; j < 4; ++j) d[i * 4 + j] = UP ? P[((i + 1) >> 1) * 10 + ((j + 1) >> 1)] : P[i * WPP + j];
#if !defined(ADM_EMU)
    if constexpr (V4) {
      // The 32 additions as 16 packed ones (v_pk_add_f32, full rate on gfx950): the rows first, two columns per instruction; then
      // per row (v0, v1) = (t0 - t2, t1 + t2) and (v2, v3) = (t2 - t1, t1 - t3) through the operand-select / negate modifiers.
      // Same additions on the same values (a - b issued as a + (-b)): bit-identical. Every producer instruction costs the
      // co-resident MFMA stream ~7 cycles of issue (profiles/r02_wino_v4.md), so 16 fewer per chunk is ~4 % of a chunk.
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
      float* vdst = ldsV + (g & RING) * W3VSLAB + vofs;
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
      return;
    }
#endif
    float t[4][4];
    ADM_UNROLL
    for (int j = 0; j < 4; ++j) {
      t[0][j] = d[0 * 4 + j] - d[2 * 4 + j];
      t[1][j] = d[1 * 4 + j] + d[2 * 4 + j];
      t[2][j] = d[2 * 4 + j] - d[1 * 4 + j];
      t[3][j] = d[1 * 4 + j] - d[3 * 4 + j];
    }
    float* vdst = ldsV + (g & RING) * W3VSLAB + vofs;
    ADM_UNROLL
    for (int i = 0; i < 4; ++i) {
      vdst[(i * 4 + 0) * (WCK * 32)] = t[i][0] - t[i][2];
      vdst[(i * 4 + 1) * (WCK * 32)] = t[i][1] + t[i][2];
      vdst[(i * 4 + 2) * (WCK * 32)] = t[i][2] - t[i][1];
      vdst[(i * 4 + 3) * (WCK * 32)] = t[i][1] - t[i][3];
    }
  };
  // ---- pipeline: interval g stages V(g) [C], the patch of g+1 [B] and the global loads of g+3 [A] ---------------------------
  Wino3Raw r0, r1;
  r0.b = make_float4(0.f, 0.f, 0.f, 0.f); r1.b = r0.b; r0.a = r0.b; r1.a = r0.b; r0.h = 0.f; r1.h = 0.f;
  unsigned long long tq = 0, tn;
#define W3_LAP(slot) do { if (PROF) { tn = W3_CLK(); pr[slot] += tn - tq; tq = tn; } } while (0)
  if constexpr (V4 && PAIR) {  // one barrier per pair of chunks (see above); the ABL instantiations never take this path
    Wino3Raw r2, r3;
    r2.b = r0.b; r3.b = r0.b; r2.a = r0.b; r3.a = r0.b; r2.h = 0.f; r3.h = 0.f;
    stage_a(r0); stage_a(r1); stage_a(r2); stage_a(r3);            // chunks 0..3
    stage_b(r0, 0); stage_b(r1, 1);
    stage_a(r0); stage_a(r1);                                      // chunks 4, 5
    ADM_BARRIER_KEEP_VMEM(63);                                     // barrier "-2": patches 0 and 1 visible to every producer wave
    if (PROF) tq = W3_CLK();
    for (int g = 0; g < total; g += 4) {                           // total is a multiple of 4 (nch is)
      stage_c(g); stage_c(g + 1);                  W3_LAP(3);
      stage_b(r2, g + 2); stage_b(r3, g + 3);      W3_LAP(4);
      stage_a(r2); stage_a(r3);                    W3_LAP(5);      // chunks g + 6, g + 7
      W3_BARRIER(63, pr, 1, 2);
      if (PROF) tq = W3_CLK();
      stage_c(g + 2); stage_c(g + 3);              W3_LAP(3);
      stage_b(r0, g + 4); stage_b(r1, g + 5);      W3_LAP(4);
      stage_a(r0); stage_a(r1);                    W3_LAP(5);      // chunks g + 8, g + 9
      W3_BARRIER(63, pr, 1, 2);
      if (PROF) tq = W3_CLK();
    }
    ADM_BARRIER_KEEP_VMEM(0);
    if (PROF && tid == 0) {
      pr[0] = W3_CLK() - t_start;
      for (int i = 0; i < 8; ++i) atomicAdd(p.prof + 8 + i, pr[i]);
    }
    return;
  }
  if constexpr (V4) {          // same schedule with the global loads of g + 5 in flight: four raw-chunk register sets
    Wino3Raw r2, r3;
    r2.b = r0.b; r3.b = r0.b; r2.a = r0.b; r3.a = r0.b; r2.h = 0.f; r3.h = 0.f;
    if (!wino_abl_idle(ABL)) stage_a(r0); if (!wino_abl_idle(ABL)) stage_a(r1); if (!wino_abl_idle(ABL)) stage_a(r2); if (!wino_abl_idle(ABL)) stage_a(r3);      // chunks 0..3
    if (!wino_abl_idle(ABL) && ABL != 5) stage_b(r0, 0);
    if (!wino_abl_idle(ABL)) stage_a(r0);                                             // chunk 4
    ADM_BARRIER_KEEP_VMEM(63);
    if (PROF) tq = W3_CLK();
    for (int g = 0; g < total; g += 4) {                     // total is a multiple of 4 (nch is)
      if (!wino_abl_idle(ABL) && ABL != 4) stage_c(g);                W3_LAP(3);
      if (!wino_abl_idle(ABL) && ABL != 5) stage_b(r1, g + 1);        W3_LAP(4);
      if (!wino_abl_idle(ABL)) stage_a(r1);               W3_LAP(5);   // chunk g + 5
      W3_BARRIER(63, pr, 1, 2);
      if (PROF) tq = W3_CLK();
      if (!wino_abl_idle(ABL) && ABL != 4) stage_c(g + 1);            W3_LAP(3);
      if (!wino_abl_idle(ABL) && ABL != 5) stage_b(r2, g + 2);        W3_LAP(4);
      if (!wino_abl_idle(ABL)) stage_a(r2);               W3_LAP(5);   // chunk g + 6
      W3_BARRIER(63, pr, 1, 2);
      if (PROF) tq = W3_CLK();
      if (!wino_abl_idle(ABL) && ABL != 4) stage_c(g + 2);            W3_LAP(3);
      if (!wino_abl_idle(ABL) && ABL != 5) stage_b(r3, g + 3);        W3_LAP(4);
      if (!wino_abl_idle(ABL)) stage_a(r3);               W3_LAP(5);   // chunk g + 7
      W3_BARRIER(63, pr, 1, 2);
      if (PROF) tq = W3_CLK();
      if (!wino_abl_idle(ABL) && ABL != 4) stage_c(g + 3);            W3_LAP(3);
      if (!wino_abl_idle(ABL) && ABL != 5) stage_b(r0, g + 4);        W3_LAP(4);
      if (!wino_abl_idle(ABL)) stage_a(r0);               W3_LAP(5);   // chunk g + 8
      W3_BARRIER(63, pr, 1, 2);
      if (PROF) tq = W3_CLK();
    }
    ADM_BARRIER_KEEP_VMEM(0);
    if (PROF && tid == 0) {
      pr[0] = W3_CLK() - t_start;
      for (int i = 0; i < 8; ++i) atomicAdd(p.prof + 8 + i, pr[i]);
    }
    return;
  }
  stage_a(r0);                 // chunk 0
  stage_a(r1);                 // chunk 1
  stage_b(r0, 0);
  stage_a(r0);                 // chunk 2
  ADM_BARRIER_KEEP_VMEM(63);   // barrier "-2": patch(0) visible to every producer wave
  tq = W3_CLK();
  for (int g = 0; g < total; g += 2) {      // total is even (nch is)
    stage_c(g);                W3_LAP(3);
    stage_b(r1, g + 1);        W3_LAP(4);
    stage_a(r1);               W3_LAP(5);   // chunk g + 3
    W3_BARRIER(63, pr, 1, 2);  // barrier g - 1
    if (PROF) tq = W3_CLK();
    stage_c(g + 1);            W3_LAP(3);
    stage_b(r0, g + 2);        W3_LAP(4);
    stage_a(r0);               W3_LAP(5);   // chunk g + 4
    W3_BARRIER(63, pr, 1, 2);  // barrier g
    if (PROF) tq = W3_CLK();
  }
#undef W3_LAP
  ADM_BARRIER_KEEP_VMEM(0);    // barrier total - 1 (the consumers' last chunk)
  if (PROF && tid == 0) {
    pr[0] = W3_CLK() - t_start;
    for (int i = 0; i < 8; ++i) atomicAdd(p.prof + 8 + i, pr[i]);
  }
}

#if defined(ADM_EXPERIMENTS)   // v3 (mode 3): the producer role above is shared with v4, this consumer and the kernel are not
// ---- consumer role: 256 threads (waves 0..3) ------------------------------------------------------------------------
template <bool PROF>
__device__ __forceinline__ void wino3_consumer(const WinoParams& p, const float* ldsV, float* ldsU, int tid, int wave,
                                               int b0, int bs) {
  unsigned long long pr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_start = W3_CLK();
  const int lane = tid & 63;
  const int li = lane & 15, k4 = lane >> 4;
  const int cw = wave & 1, tw = wave >> 1;
  const int nch = (p.C1 + p.C2) / WCK;
  const int ntile = (p.nblk - b0 + bs - 1) / bs;
  const int total = ntile * nch;
  // operand word addresses: chunk buffer + (compile-time xi / k-step terms) + these lane terms
  const int vlane = k4 * 32 + ((16 * tw + li + 16 * (k4 & 1)) & 31);
  const int ulane0 = k4 * 16 * W3BM + ((32 * cw + li + 16 * (k4 & 1)) & 63);
  const int ulane1 = k4 * 16 * W3BM + ((32 * cw + 16 + li + 16 * (k4 & 1)) & 63);
  // ---- U DMA cursor --------------------------------------------------------------------------------------------------
  int d_v = b0, d_ci = 0, d_g = 0;
  int d_m0 = wino3_tile(p, d_v).m0;
  int d_off[8];                            // source element offset of this lane's 8 float4 pieces inside a U slab
  ADM_UNROLL
  for (int i = 0; i < 8; ++i) {
    const int idx = tid + 256 * i;
    const int row = idx >> 4, c4 = idx & 15;            // row = ch * 16 + xi of the LDS image, c4 = float4 slot
    const int sc4 = (c4 - 4 * ((row >> 4) & 1)) & 15;    // odd channels are stored rotated by 16 couts
    d_off[i] = row * p.Cout + sc4 * 4;
  }
  auto issue_u = [&]() {                   // U slab of running chunk d_g -> ldsU[d_g & 1]; 2048 float4 by 256 threads
    if (d_g < total) {
      const float* usrc = p.wu + (long)d_ci * WCK * 16 * p.Cout + d_m0;   // wave-uniform
      float* udst = ldsU + (d_g & 1) * W3USLAB + wave * 256;
      ADM_UNROLL
      for (int i = 0; i < 8; ++i) ADM_GLDS16(usrc + d_off[i], udst + 1024 * i);
      ++d_g;
      if (++d_ci == nch) {
        d_ci = 0; d_v += bs;
        if (d_v < p.nblk) d_m0 = wino3_tile(p, d_v).m0;
      }
    }
  };
  issue_u();                               // U(0)
  issue_u();                               // U(1)
  ADM_BARRIER_KEEP_VMEM(63);               // barrier "-2" (producers' patch hand-over)
  ADM_BARRIER_KEEP_VMEM(0);                // barrier "-1": V(0) complete, U(0) landed

  f32x4 acc[16][2];
  float rb[4][2], ra[4][2][2];             // rolling operand window: 4 Winograd points ahead
  int g = 0;                               // running chunk index
  for (int v = b0; v < p.nblk; v += bs) {
    const Wino3Tile t = wino3_tile(p, v);
    ADM_UNROLL
    for (int xi = 0; xi < 16; ++xi)
      ADM_UNROLL
      for (int c = 0; c < 2; ++c)
        ADM_UNROLL
        for (int r = 0; r < 4; ++r) acc[xi][c][r] = 0.f;
    // epilogue constants of this lane's 8 couts, fetched now so their latency hides behind the whole tile
    float cb[2][4];
    ADM_UNROLL
    for (int c = 0; c < 2; ++c)
      ADM_UNROLL
      for (int r = 0; r < 4; ++r) {
        const int co = t.m0 + 32 * cw + 16 * c + 4 * k4 + r;
        cb[c][r] = p.bias[co] + p.chan_add[(long)t.n * p.chan_add_stride + co];
      }
    auto read_group = [&](int slot, int gg, int xi) {     // operand words of Winograd point xi of running chunk gg
      const float* V = ldsV + (gg & 1) * W3VSLAB + vlane;
      const float* U = ldsU + (gg & 1) * W3USLAB;
      ADM_UNROLL
      for (int ks = 0; ks < 2; ++ks) {
        rb[slot][ks] = V[(xi * WCK + 4 * ks) * 32];
        ra[slot][ks][0] = U[(4 * ks * 16 + xi) * W3BM + ulane0];
        ra[slot][ks][1] = U[(4 * ks * 16 + xi) * W3BM + ulane1];
      }
    };
    ADM_UNROLL
    for (int xi = 0; xi < 4; ++xi) read_group(xi, g, xi);
    for (int ci = 0; ci < nch; ++ci, ++g) {
      const bool more = ci + 1 < nch;      // the rolling window does not cross into the next tile
      ADM_UNROLL
      for (int xi = 0; xi < 16; ++xi) {
        const int s = xi & 3;
        ADM_UNROLL
        for (int ks = 0; ks < 2; ++ks) {
          acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[s][ks][0], rb[s][ks], acc[xi][0], 0, 0, 0);
          acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[s][ks][1], rb[s][ks], acc[xi][1], 0, 0, 0);
        }
        if (xi == 12) {                    // every read of chunk g was issued >= 1 step ago: free its buffers
          W3_BARRIER(0, pr, 1, 2);         // barrier g
          issue_u();                       // U(g + 2)
        }
        if (xi < 12) read_group(s, g, xi + 4);
        else if (more) read_group(s, g + 1, xi - 12);
        ADM_SCHED_FENCE();
      }
    }
    // ---- lane-local inverse transform Y = A^T M A and store: lane holds (cout = 4*k4 + r, tile = li) of each block ------
    const unsigned long long t_epi = W3_CLK();
    const int tile = 16 * tw + li;
    const int oy = t.ty * 8 + 2 * (tile >> 3), ox = t.tx * 16 + 2 * (tile & 7);
    const long planeO = (long)p.Ho * p.Wo;
    ADM_UNROLL
    for (int c = 0; c < 2; ++c) {
      float2 res[4][2];
      if (p.residual != nullptr) {
        ADM_UNROLL
        for (int r = 0; r < 4; ++r) {
          const int co = t.m0 + 32 * cw + 16 * c + 4 * k4 + r;
          const long o = ((long)t.n * p.Cout + co) * planeO + (long)oy * p.Wo + ox;
          res[r][0] = *reinterpret_cast<const float2*>(p.residual + o);
          res[r][1] = *reinterpret_cast<const float2*>(p.residual + o + p.Wo);
        }
      } else {
        ADM_UNROLL
        for (int r = 0; r < 4; ++r) { res[r][0] = make_float2(0.f, 0.f); res[r][1] = make_float2(0.f, 0.f); }
      }
      ADM_UNROLL
      for (int r = 0; r < 4; ++r) {
        float t0[4], t1[4];
        ADM_UNROLL
        for (int j = 0; j < 4; ++j) {
          t0[j] = acc[0 * 4 + j][c][r] + acc[1 * 4 + j][c][r] + acc[2 * 4 + j][c][r];
          t1[j] = acc[1 * 4 + j][c][r] - acc[2 * 4 + j][c][r] - acc[3 * 4 + j][c][r];
        }
        const float b = cb[c][r];
        const int co = t.m0 + 32 * cw + 16 * c + 4 * k4 + r;
        const long o = ((long)t.n * p.Cout + co) * planeO + (long)oy * p.Wo + ox;
        float2 y0 = make_float2(t0[0] + t0[1] + t0[2] + b + res[r][0].x, t0[1] - t0[2] - t0[3] + b + res[r][0].y);
        float2 y1 = make_float2(t1[0] + t1[1] + t1[2] + b + res[r][1].x, t1[1] - t1[2] - t1[3] + b + res[r][1].y);
        *reinterpret_cast<float2*>(p.out + o) = y0;
        *reinterpret_cast<float2*>(p.out + o + p.Wo) = y1;
      }
    }
    if (PROF) pr[3] += W3_CLK() - t_epi;
  }
  if (PROF && tid == 0) {
    pr[0] = W3_CLK() - t_start;
    for (int i = 0; i < 8; ++i) atomicAdd(p.prof + i, pr[i]);
  }
}

template <bool UP, bool PROF>
__global__ void __launch_bounds__(512, 2) conv_wino3_kernel(const WinoParams p) {
  ADM_DYN_SMEM(float, smem);
  float* ldsV = smem;
  float* ldsU = smem + 2 * W3VSLAB;
  float* ldsP = smem + 2 * W3VSLAB + 2 * W3USLAB;
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  // roles are wave-uniform and have separate register allocations (accumulators only in the consumers)
  if (wave >= 4) {
#if !defined(ADM_EMU)
    // The second-dispatched half of a 512-thread workgroup loses the per-SIMD VALU arbitration (priority, then age) to
    // its MFMA-issuing partner. The producers are the short, latency-critical role: static priority for the whole kernel.
    __builtin_amdgcn_s_setprio(1);
#endif
    if (!UP && wave == 4) wino3_producer<UP, true, PROF>(p, ldsV, ldsP, tid - 256, (int)blockIdx.x, (int)gridDim.x);
    else wino3_producer<UP, false, PROF>(p, ldsV, ldsP, tid - 256, (int)blockIdx.x, (int)gridDim.x);
  }
  else wino3_consumer<PROF>(p, ldsV, ldsU, tid, wave, (int)blockIdx.x, (int)gridDim.x);
}

#endif  // ADM_EXPERIMENTS (v3 consumer + kernel)
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
constexpr int W4LDS_PAIR = 4 * W3VSLAB + 4 * W3PSLAB;     // PAIR: rings of four (91 KiB)
constexpr int W4ABLK = 4 * 2 * 64 * 4;      // floats of one (chunk, 16-cout block) filter image: 8 KiB

// ABL (developer aid, timing only): 9 / 10 / 11 = producers idle and no filter loads / no LDS operand reads / no per-chunk barrier;
// 2 = the MFMAs are replaced by a register dependency (operands still fetched); 6 = barriers
// only (the producers' own pace); 7 = bare MFMA stream (no operand fetch; with idle producers: the matrix pipe's own pace).
template <bool PROF, int ABL = 0, bool PAIR = false>
__device__ __forceinline__ void wino4_consumer(const WinoParams& p, const float* ldsV, int tid, int wave, int b0, int bs) {
  constexpr int RING = PAIR ? 3 : 1;
  unsigned long long pr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_start = W3_CLK();
  const int lane = tid & 63;
  const int li = lane & 15, k4 = lane >> 4;
  const int nch = (p.C1 + p.C2) / WCK;
  const int n_cblk = p.Cout >> 4;
  const int ntile = (p.nblk - b0 + bs - 1) / bs;
  const int total = ntile * nch;
  const int vlane = k4 * 32 + 2 * li;       // word pair (tile 2 li, 2 li + 1) of channel row k4 (+ 4 ks)
  // ---- filter stream cursor: (tile, chunk) of the NEXT chunk to load; saturates on the last one ---------------------------
  int d_v = b0, d_ci = 0, d_left = total;
  const long chunk_stride = (long)n_cblk * W4ABLK;
  const float* d_src = p.wu + ((long)(wino3_tile(p, d_v).m0 >> 4) + wave) * W4ABLK + lane * 4;   // chunk 0 of the tile
  f32x4 a[4][2];                            // [point group][k step]: component e = Winograd point 4 q + e
  // ABL 13 (experiments build; CORRECT results, not an ablation): the filter stream as raw buffer loads — resource = this wave's
  // 16-cout block of the tile's filter image with an EXPLICITLY uniform base (readfirstlane: derived from the wave index, the compiler
  // does not prove it uniform and waterfalls every load — measured +20 % that way), lane offset = lane * 16 bytes, chunk offset = an
  // SGPR. The producers' loads gained 4.9 % from the same change; this one measured 1.5-4 % SLOWER (2.963 vs 2.919 ms, 0.687 vs 0.661):
  // the eight loads of a chunk already share one address register pair. Kept as a record, experiments build only.
  constexpr bool ABUF = ABL == 13;
#if !defined(ADM_EMU)
  auto tile_rsrc = [&](int v) {
    const float* b = p.wu + ((long)(wino3_tile(p, v).m0 >> 4) + ADM_UNIFORM(wave)) * W4ABLK;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(adm_uniform_ptr(b)), (short)0, 0x7fffffff, 0x00027000);
  };
  __amdgpu_buffer_rsrc_t d_rs = tile_rsrc(d_v);
  const int d_vo = lane * 16, chunk_stride_b = (int)(chunk_stride * 4);
  int d_so = 0;
#define W4_LOAD_A(q)                                                                                                          \
  do {                                                                                                                        \
    if (ABUF) {                                                                                                               \
      a[q][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(d_rs, d_vo, d_so + (q) * 2048, 0));           \
      a[q][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(d_rs, d_vo, d_so + (q) * 2048 + 1024, 0));    \
    } else {                                                                                                                  \
      a[q][0] = *reinterpret_cast<const f32x4*>(d_src + (q) * 512);                                                           \
      a[q][1] = *reinterpret_cast<const f32x4*>(d_src + (q) * 512 + 256);                                                     \
    }                                                                                                                         \
  } while (0)
#else
#define W4_LOAD_A(q)                                                                   \
  do {                                                                                 \
    a[q][0] = *reinterpret_cast<const f32x4*>(d_src + (q) * 512);                      \
    a[q][1] = *reinterpret_cast<const f32x4*>(d_src + (q) * 512 + 256);                \
  } while (0)
#endif
  auto advance_a = [&]() {
    if (d_left > 1) {
      --d_left;
      d_src += chunk_stride;
#if !defined(ADM_EMU)
      if (ABUF) d_so += chunk_stride_b;
#endif
      if (++d_ci == nch) {
        ADM_SCHED_FENCE();
        d_ci = 0; d_v += bs;
        d_src = p.wu + ((long)(wino3_tile(p, d_v).m0 >> 4) + wave) * W4ABLK + lane * 4;
#if !defined(ADM_EMU)
        if (ABUF) { d_so = 0; d_rs = tile_rsrc(d_v); }
#endif
      }
    }
  };
  W4_LOAD_A(0); W4_LOAD_A(1); W4_LOAD_A(2); W4_LOAD_A(3);      // chunk 0
  advance_a();
  ADM_BARRIER_KEEP_VMEM(63);               // barrier "-2" (producers' patch hand-over)
  ADM_BARRIER_KEEP_VMEM(63);               // barrier "-1": V(0) complete

  f32x4 acc[16][2];
  // rolling B window, running across tile boundaries: RB Winograd points ahead. 4 points = 16 MFMAs = 512 cycles of cover for an LDS
  // read that the producers' traffic delays; the role accounting (tools/wino_prof_probe.py) has the consumer on the critical path with
  // ~900 non-MFMA cycles per chunk, so the window is 8 points (16 more registers; the chunk hand-over barrier moves from point 12 to 8,
  // where the first read of the next chunk is issued — the producers have 16 % of barrier slack)
  constexpr int RB = (ABL == 0 || ABL == 12 || ABL == 13) ? 8 : 4;
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
    const long obase = ((long)t.n * p.Cout + t.m0 + 16 * wave + 4 * k4) * planeO + (long)oy * p.Wo + ox;   // cout row r: + r * planeO
    // Bias, per-sample term and residual enter in the WINOGRAD domain: Y = A^T M A has Y00 / Y01 / Y10 / Y11 depend on the corner
    // entries M00 / M03 / M30 / M33 alone with weights +1 / -1 / -1 / +1, so adding (b + res) there is adding it to the output.
    // One cout row per chunk over the first four chunks: the loads are issued when the chunk starts and consumed when it ends
    // — a whole chunk of latency cover for 10 registers — and the epilogue is left with arithmetic and stores only.
    f32x4 fr0 = {0.f, 0.f, 0.f, 0.f}, fr1 = fr0;
    float fb0 = 0.f, fb1 = 0.f;
    for (int ci = 0; ci < nch; ++ci, ++g) {
      if (ABL == 6 || ABL == 7 || ABL == 8) {       // 8 = bare MFMA stream beside WORKING producers
        ADM_UNROLL
        for (int xi = 0; xi < 16; ++xi) {
          if (ABL != 6) {
            ADM_UNROLL
            for (int ks = 0; ks < 2; ++ks) {
              acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb0, fb1, acc[xi][0], 0, 0, 0);
              acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb1, fb0, acc[xi][1], 0, 0, 0);
            }
          }
          if (xi == 12) W3_BARRIER(63, pr, 1, 2);
        }
        continue;
      }
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
          if (ABL == 2) {
            acc[xi][0][0] += a[q][ks][e] * rb[s][ks].x;
            acc[xi][1][0] += a[q][ks][e] * rb[s][ks].y;
          } else {
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][ks][e], rb[s][ks].x, acc[xi][0], 0, 0, 0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][ks][e], rb[s][ks].y, acc[xi][1], 0, 0, 0);
          }
        }
        if (e == 3 && ABL != 9 && ABL != 12) {   // group q consumed: its registers take the NEXT chunk's words (12: never, beside WORKING producers)
          if (q == 0) W4_LOAD_A(0);
          if (q == 1) W4_LOAD_A(1);
          if (q == 2) W4_LOAD_A(2);
          if (q == 3) { W4_LOAD_A(3); advance_a(); }
        }
        // barrier g: every read of V(g) has landed, V(g + 1) is complete. PAIR: only behind the second chunk of a pair (its
        // first chunk runs on into V(g + 1), which the previous pair's barrier certified)
        if (xi == 16 - RB && ABL != 11 && (!PAIR || (g & 1))) W3_BARRIER(63, pr, 1, 2);
        if (ABL != 10) {
          if (xi < 16 - RB) read_group(s, g, xi + RB);
          else read_group(s, g + 1, xi - (16 - RB));   // next chunk — of this tile or the next one (past the end: stale words, unused)
        }
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
    const unsigned long long t_epi = W3_CLK();
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
    if (PROF) pr[3] += W3_CLK() - t_epi;
  }
#undef W4_LOAD_A
  if (PROF && tid == 0) {
    pr[0] = W3_CLK() - t_start;
    for (int i = 0; i < 8; ++i) atomicAdd(p.prof + i, pr[i]);
  }
}

template <bool UP, bool PROF, int ABL = 0, int ACT = -1, bool PAIR = false>
__global__ void __launch_bounds__(512) conv_wino4_kernel(const WinoParams p) {
  ADM_DYN_SMEM(float, smem);
  float* ldsV = smem;
  float* ldsP = smem + (PAIR ? 4 : 2) * W3VSLAB;
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  if (wave >= 4) {
#if !defined(ADM_EMU)
    if (ABL != 3) __builtin_amdgcn_s_setprio(1);         // see conv_wino3_kernel (ABL 3: timing without it)
#endif
    if (!UP && wave == 4) wino3_producer<UP, true, PROF, true, ABL, ACT, PAIR>(p, ldsV, ldsP, tid - 256, (int)blockIdx.x, (int)gridDim.x);
    else wino3_producer<UP, false, PROF, true, ABL, ACT, PAIR>(p, ldsV, ldsP, tid - 256, (int)blockIdx.x, (int)gridDim.x);
  }
  else wino4_consumer<PROF, ABL, PAIR>(p, ldsV, tid, wave, (int)blockIdx.x, (int)gridDim.x);
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
constexpr int W5BM = 128;
constexpr int W5RB = 4;          // B window of the MFMA block, in Winograd points
// INTER: behind which MFMA group (0..31 = chunk * 16 + Winograd point) of an interval each staging piece is placed
constexpr int W5S_CR = 0, W5S_CM = 3, W5S_B0 = 7, W5S_B1 = 10, W5S_A = 14, W5S_SHIFT = 16;
struct Wino5Raw { f32x4 a; float sc, sh; unsigned ok; };      // one item of one chunk (HALO / UP: a[0] only)

__device__ __forceinline__ Wino3Tile wino5_tile(const WinoParams& p, int v) {
  const int q = p.nblk >> 3, r = p.nblk & 7, xcd = v & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
  const int ct = lid % p.n_ct, pt = lid / p.n_ct;
  Wino3Tile t;
  t.tx = pt % p.tiles_x; t.ty = (pt / p.tiles_x) % p.tiles_y; t.n = pt / (p.tiles_x * p.tiles_y);
  t.m0 = ct * W5BM;
  return t;
}

// HALO: this wave's staging item is a halo element (waves 5-7 of the non-UP kernel), else a float4 row piece (UP: one scalar of the
// source-resolution patch). TUNE bit 0: static s_setprio 1 for the second half (waves 4-7); bit 1: B window of 4 points instead of 8.
// ABL (experiments build, TIMING ONLY — results are wrong): bit 0 / 1 / 2 = stage C / B / A skipped, 3 = no MFMAs (operands still fetched),
// 4 = no filter loads, 5 = no LDS operand reads, 6 = no workgroup barriers, 7 = no bias / residual fold. PROF: per-half cycle accounting
// (s_memtime) into p.prof: [0] total, [1] MFMA blocks, [2] staging, [3] barrier waits, [4] epilogues; second half at +8.
// INTER: no halves — every wave runs [MFMA block with the staging pieces placed between its MFMA groups] barrier: both waves of a SIMD
// always have MFMAs to issue, and whatever one of them waits for (an LDS round trip, a vector-memory issue) the other's MFMAs cover.
template <bool UP, bool HALO, int ACT, int ABL = 0, bool PROF = false, bool INTER = false, bool H2 = false>
__device__ __forceinline__ void wino5_wave(const WinoParams& p, float* ldsV, float* ldsP, const int tid, const int wave,
                                           const int b0, const int bs) {
  const bool yrole = !INTER && wave >= 4;     // second half: staging first, MFMA block second
  unsigned long long pr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_start = (PROF || p.prof != nullptr) ? W3_CLK() : 0ull;   // (p.prof alone: total cycles only, two reads per wave)
  unsigned long long tq = t_start, tn;
#define W5_LAP(slot) do { if (PROF) { tn = W3_CLK(); pr[slot] += tn - tq; tq = tn; } } while (0)
#define W5_BARRIER() do { if (!(ABL & 64)) ADM_BARRIER_KEEP_VMEM(63); } while (0)
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
#define W5_LOAD_A_(q)                                                                \
  do {                                                                               \
    a[q][0] = *reinterpret_cast<const f32x4*>(d_src + (q) * 512);                    \
    a[q][1] = *reinterpret_cast<const f32x4*>(d_src + (q) * 512 + 256);              \
  } while (0)
#define W5_LOAD_A(q) do { if (!(ABL & 16)) W5_LOAD_A_(q); } while (0)
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
  W5_LOAD_A_(0); W5_LOAD_A_(1); W5_LOAD_A_(2); W5_LOAD_A_(3);  // filters of chunk 0
  advance_a();
  W5_BARRIER();                               // patches 0, 1 complete

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
      W5_LAP(4);
  };
  bool pend = false;                          // a finished tile waits for its inverse transform + stores
  for (int it = -2; it <= npairs; ++it) {     // (iteration npairs: nothing but the last tile's epilogue)
    // Both halves store a finished tile at the top of the NEXT iteration — behind the barrier that ended its last interval: in front of
    // it, the first half's ~4000 cycles of inverse transform and stores kept the second half waiting once per tile.
    if (pend) { if (PROF) tq = W3_CLK(); epilogue(); pend = false; }
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
      if (PROF) tq = W3_CLK();
      ADM_UNROLL
      for (int xi = 0; xi < RB; ++xi) read_group(xi, g, xi);
      // (a real two-trip loop, NOT unrolled: the loop-carried values pin the accumulators and the filter registers in place — unrolled,
      // hipcc renamed them across the two copies: 156 accumulator and 48 filter registers instead of 128 + 32, and spilled)
      float cd[16];                            // (INTER) stage C's window, between its read and its transform
      auto chunk = [&](const int c2) __attribute__((always_inline)) {
        const int cc = ci + c2;
        if (cc < 4 && !(ABL & 128)) {          // wave-uniform: this chunk carries cout row r = cc of the bias / residual fold (v4)
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
            if (ABL & 8) {
              acc[xi][0][0] += a[q][ks][e] * rb[s][ks].x;
              acc[xi][1][0] += a[q][ks][e] * rb[s][ks].y;
            } else {
              acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][ks][e], rb[s][ks].x, acc[xi][0], 0, 0, 0);
              acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][ks][e], rb[s][ks].y, acc[xi][1], 0, 0, 0);
            }
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
          if (!(ABL & 32)) {
            if (xi < 16 - RB) read_group(s, g + c2, xi + RB);
            else read_group(s, g + c2 + 1, xi - (16 - RB));
          }
          ADM_SCHED_FENCE();
          if (INTER) {                         // staging pieces of P(it) between the MFMA groups (s is a compile-time constant here)
            // The two waves of a SIMD (w and w + 4) run the same stream and restart together behind every barrier; with the SAME
            // placement their staging pieces — and the stalls that come with them — would coincide. The second half places its pieces
            // W5S_SHIFT groups later (same order, so the register hand-overs between the pieces hold).
            const int sl = c2 * 16 + xi;
#define W5_AT(S) (sl == (S) + (H2 ? W5S_SHIFT : 0))
            if (W5_AT(W5S_CR) && !(ABL & 1)) stage_c_read(pg + cpar, cd);
            if (W5_AT(W5S_CM) && !(ABL & 1)) stage_c_math(pg + cpar, cd);
            if (W5_AT(W5S_B0) && !(ABL & 2)) stage_b(r0, pg + 2);
            if (W5_AT(W5S_B1) && !(ABL & 2)) stage_b(r1, pg + 3);
            if (W5_AT(W5S_A)) { if (!(ABL & 4)) stage_a2(r0, r1); pg += 2; }
#undef W5_AT
            ADM_SCHED_FENCE();
          }
        }
        if (cc < 4 && !(ABL & 128)) {
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
      W5_LAP(1);
    }
    if (yrole && it >= -1 && it < npairs) { if (PROF) tq = W3_CLK(); W5_BARRIER(); W5_LAP(3); }
    if (INTER ? it == -1 : (yrole ? it + 1 < npairs : (it >= -1 && it < npairs))) {  // P: C(pg, pg + 1), B(pg + 2, pg + 3), A(the next pair of the stream)
      if (PROF) tq = W3_CLK();
      // Stage C's window read first (its LDS round trip runs under what follows). The second half stages right behind its MFMA block, whose
      // last filter refills are still in flight and sit in front of stage B's raw activations in the in-order counter: it finishes stage C
      // before stage B, the first half (a whole MFMA block between its loads and this point is not the issue there) the other way round.
      float cd[16];
      if (!(ABL & 1)) stage_c_read(pg + cpar, cd);
      if (yrole && !(ABL & 1)) {
        stage_c_math(pg + cpar, cd);
        ADM_SCHED_FENCE();
        W5_LAP(2);
      }
      if (!(ABL & 2)) {
        stage_b(r0, pg + 2);
        ADM_SCHED_FENCE();
        stage_b(r1, pg + 3);
        ADM_SCHED_FENCE();
      }
      W5_LAP(5);
      if (!yrole && !(ABL & 1)) {
        stage_c_math(pg + cpar, cd);
        ADM_SCHED_FENCE();
        W5_LAP(2);
      }
      if (!(ABL & 4)) stage_a2(r0, r1);
      pg += 2;
      W5_LAP(6);
    }
    if (!yrole && it >= -1 && it < npairs) { if (PROF) tq = W3_CLK(); W5_BARRIER(); W5_LAP(3); }
  }
  if ((PROF || p.prof != nullptr) && (tid & 255) == 0) {
    pr[0] = W3_CLK() - t_start;
    for (int i = 0; i < (PROF ? 8 : 1); ++i) atomicAdd(p.prof + (tid >= 256 ? 8 : 0) + i, pr[i]);
  }
#undef W5_LAP
#undef W5_BARRIER
#undef W5_LOAD_A_
#undef W5_LOAD_A
}

template <bool UP, int ACT, int ABL = 0, bool PROF = false, bool INTER = false>
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
    if (!UP && wave >= 5) wino5_wave<UP, true, ACT, ABL, PROF, INTER, INTER>(p, ldsV, ldsP, tid, wave, (int)blockIdx.x, (int)gridDim.x);
    else wino5_wave<UP, false, ACT, ABL, PROF, INTER, INTER>(p, ldsV, ldsP, tid, wave, (int)blockIdx.x, (int)gridDim.x);
    return;
  }
  if (!UP && wave >= 5) wino5_wave<UP, true, ACT, ABL, PROF, INTER>(p, ldsV, ldsP, tid, wave, (int)blockIdx.x, (int)gridDim.x);
  else wino5_wave<UP, false, ACT, ABL, PROF, INTER>(p, ldsV, ldsP, tid, wave, (int)blockIdx.x, (int)gridDim.x);
}

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
constexpr int W6ABLK = 2 * 9 * 64 * 4;             // floats of one (chunk, 16-cout block) filter image: 18 KiB
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
#if defined(W6X_ALLX)
  const bool yrole = false;
#elif defined(W6X_ALLY)
  const bool yrole = true;
#elif defined(W6X_SWAP)
  const bool yrole = wave < 4;
#else
  const bool yrole = wave >= 4;
#endif
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
#ifdef W6X_BFENCE
      ADM_SCHED_FENCE();
#endif
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
#ifndef W6X_NORES
    if (p.residual != nullptr) {
      ADM_UNROLL
      for (int q = 0; q < 2; ++q)
        ADM_UNROLL
        for (int a = 0; a < 4; ++a) e.res[q][a] = load_res(q, a);
    }
#endif
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
#ifndef W6X_NORES
        if (p.residual != nullptr) {
          ADM_UNROLL
          for (int b = 0; b < 4; ++b) y[b] += e.res[r & 1][a][b];
        }
#endif
#if !defined(ADM_EMU)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), o_rs, o_vo, r * plane_b + a * row_b, 0);
#else
        *reinterpret_cast<f32x4*>(p.out + ((long)t.n * p.Cout + t.m0 + 16 * wave) * planeO + o_vo + r * planeO + a * p.Wo) = y;
#endif
        f1 += (y[0] + y[1]) + (y[2] + y[3]);
        f2 += (y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3]);
#ifdef W6X_EFENCE
        ADM_SCHED_FENCE();                     // (one cout row = one scheduling region: its four output rows' chains run side by side, see stage B)
#endif
      }
#ifndef W6X_NOSTATS
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
#endif
      ADM_SCHED_FENCE();
#ifndef W6X_NORES
      // the row after next is fetched here, not piece by piece above: the statistics' shuffles find these 16 registers free
      if (p.residual != nullptr && r + 2 < 4) {
        ADM_UNROLL
        for (int a = 0; a < 4; ++a) e.res[r & 1][a] = load_res(r + 2, a);
      }
#endif
    }
  };
  bool pend = false;                           // a finished tile waits for its inverse transform + stores
#if defined(W6X_PROF) && !defined(ADM_EMU)     // developer build: cycle accounting of waves 0 / 4 ([1] M [2] B [3] epilogue [4] A [5] C [6] barrier)
  unsigned long long pr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = W3_CLK(), tn;
  const unsigned long long t_start = tq;
#define W6_LAP(slot) do { tn = W3_CLK(); pr[slot] += tn - tq; tq = tn; } while (0)
#else
#define W6_LAP(slot) ((void)0)
#endif
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
#if !defined(ADM_EMU) && !defined(W6X_NOPRIO)
    __builtin_amdgcn_s_setprio(2);
#endif
    W6_LAP(7);
    if (pend) {                                // (stage B twice in the source: the epilogue's operands live in this branch only)
      EpiOps e;
#ifndef W6X_NOEPI
      epilogue_fetch(e);                       // (their HBM / L2 round trips pass under stage B)
#endif
      if (more) {
#ifndef W6X_NOB
        stage_b(r0, pg + 2);
        ADM_SCHED_FENCE();
#endif
      }
      W6_LAP(2);
#ifndef W6X_NOEPI
      epilogue(e);
#endif
      W6_LAP(3);
      pend = false;
#ifndef W6X_RING_IN_P
      ADM_UNROLL
      for (int q = 0; q < W6AR; ++q) aR[q] = W6_LOAD_A(d_cur + q * 256);
#endif
    } else {
      if (more) {
#ifndef W6X_NOB
        stage_b(r0, pg + 2);
        ADM_SCHED_FENCE();
#endif
      }
      W6_LAP(2);
    }
    // (unconditional — behind the last pair the saturated cursor re-reads it — so that the activations and their scale / shift are dead
    // across the epilogue in the compiler's eyes too)
#ifndef W6X_NOA
    stage_a(r0);
#endif
    // (W6X_RING_IN_P, measured and not adopted: the filter ring's first six groups of the NEXT MFMA block fetched here instead of behind the
    // previous block's last groups — the MFMA blocks get 15 % shorter and stage B's waits stop covering these loads, but stage A grows by as
    // much: 58.5 vs 57.7 ms per forward, profiles/r05_wino.md)
#ifdef W6X_RING_IN_P
    ADM_UNROLL
    for (int q = 0; q < W6AR; ++q) aR[q] = W6_LOAD_A(d_cur + q * 256);
#endif
    ADM_SCHED_FENCE();
    W6_LAP(4);
    if (more) {
#ifndef W6X_NOC
      stage_c(pg);
#endif
    }
    pg += 2;
    ADM_SCHED_FENCE();
    W6_LAP(5);
#if !defined(ADM_EMU) && !defined(W6X_NOPRIO)
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  // The two halves of the workgroup run an interval in opposite order (inside an interval the staging block and the MFMA block touch disjoint
  // ring slots): waves 4-7 run P(it), M(it), barrier; waves 0-3 run M(it), P(it), barrier — written as ONE loop body [P; M] in which the
  // first half's P is the previous interval's and its barrier sits between the two blocks (s_barrier counts arrivals, not program counters):
  // while one wave of a SIMD stages, its partner owns the matrix pipe.
  for (int it = 0; it <= npairs; ++it) {
    if (yrole || it > 0) staging(yrole ? it < npairs : true);
    if (!yrole && it > 0) { ADM_BARRIER_KEEP_VMEM(63); W6_LAP(6); }
    if (it == npairs) break;
    if (ci == nch) tile_switch();
    W6_LAP(7);
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
#ifndef W6X_NOLDS
        if (gi + 3 < 18) read_b(gi % 3, g + c2, gi + 3);
        else read_b(gi % 3, g + c2 + 1, gi + 3 - 18);
#endif
        // the ring slot takes the group W6AR places further down the stream (this chunk's, or the next chunk's first ones)
        // (behind a tile's LAST chunk the ring is not refilled: the epilogue that follows needs those 24 registers, and the next tile's
        // first six groups are loaded right behind it — one exposed L2 round trip per tile)
#ifndef W6X_NOFILT
        if (gi + W6AR < 18) aR[gi % W6AR] = W6_LOAD_A(d_cur + (gi + W6AR) * 256);
#ifdef W6X_RING_IN_P
        else if (c2 == 0) aR[gi % W6AR] = W6_LOAD_A(d_nxt + (gi + W6AR - 18) * 256);
#else
        else if (!(c2 == 1 && ci + 2 == nch)) aR[gi % W6AR] = W6_LOAD_A(d_nxt + (gi + W6AR - 18) * 256);
#endif
#endif
        ADM_SCHED_FENCE();
      }
      d_cur = d_nxt;
      advance_next();
    }
    ci += 2;
    pend = ci == nch;
    W6_LAP(1);
    if (yrole) { ADM_BARRIER_KEEP_VMEM(63); W6_LAP(6); }
  }
#if defined(W6X_PROF) && !defined(ADM_EMU)
#ifndef W6X_PROFW
#define W6X_PROFW 0
#endif
  if (p.prof != nullptr && (tid & 255) == W6X_PROFW) {    // (W6X_PROFW = 64 k: waves k and 4 + k)
    pr[0] = W3_CLK() - t_start;
    for (int i = 0; i < 8; ++i) atomicAdd(p.prof + (yrole ? 8 : 0) + i, pr[i]);
  }
#endif
#undef W6_LAP
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

// Filter image of conv_wino6_kernel: U = G g G^T (6x6) as [Cin/8][Cout/16][k step 2][point group 9][lane = 16 k4 + l15][4 points] holding
// U[point = 4 pg + e][cout = 16 cblk + l15][cin = 8 chunk + 4 ks + k4]; transposed: the data-gradient filters (as pack_winograd4_body).
__device__ __forceinline__ void pack_winograd6_body(const float* __restrict__ w, float* __restrict__ wu, int Cout, int Cin,
                                                    int transposed, long first, long step) {
  const int PCo = transposed ? Cin : Cout, PCi = transposed ? Cout : Cin;
  const long total = (long)PCo * PCi;
  const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                         {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
  for (long i = first; i < total; i += step) {
    const int l15 = (int)(i & 15);
    long r = i >> 4;
    const int k4 = (int)(r & 3); r >>= 2;
    const int ks = (int)(r & 1); r >>= 1;
    const int cblk = (int)(r % (PCo >> 4));
    const int chunk = (int)(r / (PCo >> 4));
    const int po = cblk * 16 + l15, pi = chunk * 8 + 4 * ks + k4;
    const int co = transposed ? pi : po, c = transposed ? po : pi;
    const float* g = w + ((long)co * Cin + c) * 9;
    float gg[3][3];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) gg[a][b] = transposed ? g[(2 - a) * 3 + (2 - b)] : g[a * 3 + b];
    float t[6][3];
    for (int a = 0; a < 6; ++a)
      for (int b = 0; b < 3; ++b) t[a][b] = G[a][0] * gg[0][b] + G[a][1] * gg[1][b] + G[a][2] * gg[2][b];
    float* blk = wu + ((long)chunk * (PCo >> 4) + cblk) * W6ABLK + (long)ks * 9 * 256 + (k4 * 16 + l15) * 4;
    for (int a = 0; a < 6; ++a)
      for (int b = 0; b < 6; ++b) {
        const int pt = a * 6 + b;
        blk[(pt >> 2) * 256 + (pt & 3)] = t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2];
      }
  }
}

__global__ void pack_winograd6_weight_kernel(const float* __restrict__ w, float* __restrict__ wu, int Cout, int Cin, int transposed) {
  pack_winograd6_body(w, wu, Cout, Cin, transposed, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

// Filter image of conv_wino4_kernel: U = G g G^T as [Cin/8][Cout/16][point group q][k step ks][lane = 16 k4 + li][e]
// holding U[xi = 4 q + e][cout = 16 cblk + li][cin = 8 chunk + 4 ks + k4]. transposed: the data-gradient filters (roles of
// Cout / Cin swapped, taps flipped), as pack_winograd_weight_kernel.
__device__ __forceinline__ void pack_winograd4_body(const float* __restrict__ w, float* __restrict__ wu, int Cout, int Cin,
                                                    int transposed, long first, long step) {
  const int PCo = transposed ? Cin : Cout, PCi = transposed ? Cout : Cin;      // channel counts of the packed convolution
  const long total = (long)PCo * PCi;
  for (long i = first; i < total; i += step) {
    const int li = (int)(i & 15);
    long r = i >> 4;
    const int k4 = (int)(r & 3); r >>= 2;
    const int ks = (int)(r & 1); r >>= 1;
    const int cblk = (int)(r % (PCo >> 4));
    const int chunk = (int)(r / (PCo >> 4));
    const int po = cblk * 16 + li, pi = chunk * 8 + 4 * ks + k4;
    const int co = transposed ? pi : po, c = transposed ? po : pi;              // indices into w (Cout, Cin, 3, 3)
    const float* g = w + ((long)co * Cin + c) * 9;
    float t[4][3];
    for (int j = 0; j < 3; ++j) {
      const float g0 = transposed ? g[2 * 3 + (2 - j)] : g[0 * 3 + j];
      const float g1 = transposed ? g[1 * 3 + (2 - j)] : g[1 * 3 + j];
      const float g2 = transposed ? g[0 * 3 + (2 - j)] : g[2 * 3 + j];
      t[0][j] = g0;
      t[1][j] = 0.5f * (g0 + g1 + g2);
      t[2][j] = 0.5f * (g0 - g1 + g2);
      t[3][j] = g2;
    }
    float* blk = wu + ((long)chunk * (PCo >> 4) + cblk) * W4ABLK + ((long)ks * 64 + k4 * 16 + li) * 4;
    for (int a = 0; a < 4; ++a) {          // a = point group q (row of U), e = column
      f32x4 u;
      u[0] = t[a][0]; u[1] = 0.5f * (t[a][0] + t[a][1] + t[a][2]); u[2] = 0.5f * (t[a][0] - t[a][1] + t[a][2]); u[3] = t[a][2];
      *reinterpret_cast<f32x4*>(blk + a * 512) = u;
    }
  }
}

__global__ void pack_winograd4_weight_kernel(const float* __restrict__ w, float* __restrict__ wu, int Cout, int Cin,
                                             int transposed) {
  pack_winograd4_body(w, wu, Cout, Cin, transposed, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

// (Cout,Cin,3,3) -> U = G g G^T laid out [Cin][16][Cout]; G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]].
// transposed != 0: filters of the DATA-GRADIENT convolution (input channels = Cout, output channels = Cin, taps flipped):
// U' = G flip(g) G^T laid out [Cout][16][Cin].
__device__ __forceinline__ void pack_winograd3_body(const float* __restrict__ w, float* __restrict__ wu, int Cout, int Cin,
                                                    int transposed, long first, long step) {
  const long total = (long)Cout * Cin;
  for (long i = first; i < total; i += step) {
    int co, c;
    if (transposed) { c = (int)(i % Cin); co = (int)(i / Cin); }     // consecutive threads -> consecutive destination words
    else { co = (int)(i % Cout); c = (int)(i / Cout); }
    const float* g = w + ((long)co * Cin + c) * 9;
    float t[4][3];
    for (int j = 0; j < 3; ++j) {
      const float g0 = transposed ? g[2 * 3 + (2 - j)] : g[0 * 3 + j];
      const float g1 = transposed ? g[1 * 3 + (2 - j)] : g[1 * 3 + j];
      const float g2 = transposed ? g[0 * 3 + (2 - j)] : g[2 * 3 + j];
      t[0][j] = g0;
      t[1][j] = 0.5f * (g0 + g1 + g2);
      t[2][j] = 0.5f * (g0 - g1 + g2);
      t[3][j] = g2;
    }
    const long ostride = transposed ? Cin : Cout;
    for (int a = 0; a < 4; ++a) {
      const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]),
                  u3 = t[a][2];
      float* dst = transposed ? wu + ((long)co * 16 + a * 4) * Cin + c : wu + ((long)c * 16 + a * 4) * Cout + co;
      dst[0] = u0; dst[ostride] = u1; dst[2 * ostride] = u2; dst[3 * ostride] = u3;
    }
  }
}

__global__ void pack_winograd_weight_kernel(const float* __restrict__ w, float* __restrict__ wu, int Cout, int Cin,
                                            int transposed) {
  pack_winograd3_body(w, wu, Cout, Cin, transposed, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}
// blockIdx.y = item of a device table; flag bit 0 = transposed (data-gradient filters), bit 1 = the conv_wino4_kernel image
__global__ void __launch_bounds__(256) pack_winograd_batch_kernel(const PackItem* __restrict__ items) {
  const PackItem it = items[blockIdx.y];
  const long first = (long)blockIdx.x * blockDim.x + threadIdx.x, step = (long)gridDim.x * blockDim.x;
  if (it.flag & 2) pack_winograd4_body(it.src, (float*)it.dst, it.Cout, it.Cin, it.flag & 1, first, step);
  else pack_winograd3_body(it.src, (float*)it.dst, it.Cout, it.Cin, it.flag & 1, first, step);
  if (it.flag & 4) pack_winograd6_body(it.src, (float*)it.dst + (long)it.Cout * it.Cin * 16, it.Cout, it.Cin, it.flag & 1, first, step);
}

static int wino_mode();
static int wino_pair();
// Which filter image a convolution with these PACKED channel counts uses — decided by the mode and the channel counts
// alone, so that the packing (done once per layer) and every later launch agree: mode 4 and 64 | couts, 32 | cins -> the
// conv_wino4_kernel image (such a layer then runs on conv_wino4_kernel or, for arguments that kernel cannot take, on the
// direct kernel — never on v1-v3, which could not read it). The option must be set before the weights are packed.
// (mode 0 — no Winograd kernel runs — packs the mode-4 image as well: an image packed under "conv_wino" = 0 and convolved under the default
// would otherwise be read in the wrong layout and past its 16 floats per filter; only the experiments builds' modes 1-3 keep the v3 layout)
static bool wino4_layout(int couts, int cins) {
  const int m = wino_mode();
  return (m == 4 || m == 0) && couts % W3BM == 0 && cins % (4 * WCK) == 0;
}

// The F(4x4,3x3) image of conv_wino6_kernel FOLLOWS the F(2x2,3x3) image in the same buffer (16 + 36 transformed values per filter) whenever
// the packed channel counts allow the kernel (128 | couts on top of the v4 rule) — whatever the "wino6" option says at packing time, so that
// the option may change between packing and launch.
static bool wino6_layout(int couts, int cins) { return wino4_layout(couts, cins) && couts % W5BM == 0; }
// (the SIZE follows the channel counts alone — not the "conv_wino" mode at the time of the call: a buffer sized under one mode and re-packed
//  under another must hold whichever images that mode writes)
long winograd_packed_floats(int Cout, int Cin, int transposed) {
  const int couts = transposed ? Cin : Cout, cins = transposed ? Cout : Cin;
  const bool room6 = couts % W3BM == 0 && cins % (4 * WCK) == 0 && couts % W5BM == 0;
  return (long)Cout * Cin * (room6 ? 52 : 16);
}
static int pack_winograd(const float* w, float* wu, int Cout, int Cin, int transposed, hipStream_t st) {
  long g = ((long)Cout * Cin + 255) / 256;
  if (g > 4096) g = 4096;
  if (wino6_layout(transposed ? Cin : Cout, transposed ? Cout : Cin))
    ADM_LAUNCH(pack_winograd6_weight_kernel, dim3((unsigned)g), dim3(256), 0, st, w, wu + (long)Cout * Cin * 16, Cout, Cin, transposed);
  if (wino4_layout(transposed ? Cin : Cout, transposed ? Cout : Cin))
    ADM_LAUNCH(pack_winograd4_weight_kernel, dim3((unsigned)g), dim3(256), 0, st, w, wu, Cout, Cin, transposed);
  else
    ADM_LAUNCH(pack_winograd_weight_kernel, dim3((unsigned)g), dim3(256), 0, st, w, wu, Cout, Cin, transposed);
  return ADM_CHECK_LAUNCH();
}
int launch_pack_winograd_weight(const float* w, float* wu, int Cout, int Cin, hipStream_t st) {
  return pack_winograd(w, wu, Cout, Cin, 0, st);
}
// PackItem::flag of a Winograd image: bit 0 = transposed, bit 1 = the conv_wino4_kernel layout — decided exactly as pack_winograd
// decides it, so that the batched re-pack writes the image the kernels read
int winograd_pack_flag(int Cout, int Cin, int transposed) {
  return (transposed ? 1 : 0) | (wino4_layout(transposed ? Cin : Cout, transposed ? Cout : Cin) ? 2 : 0) |
         (wino6_layout(transposed ? Cin : Cout, transposed ? Cout : Cin) ? 4 : 0);
}
int launch_pack_winograd_batch(const PackItem* items_dev, int n, hipStream_t st) {
  if (n <= 0) return 0;
  ADM_LAUNCH(pack_winograd_batch_kernel, dim3(32, (unsigned)n), dim3(256), 0, st, items_dev);
  return ADM_CHECK_LAUNCH();
}
int launch_pack_winograd_weight_T(const float* w, float* wu, int Cout, int Cin, hipStream_t st) {
  return pack_winograd(w, wu, Cout, Cin, 1, st);
}

// 0 direct MFMA kernel only | 1 Winograd v1 | 2 wave-specialised v2 | 3 persistent wave-specialised v3 (1.33x the direct
// kernel on the whole UNet forward) | 4 (default) v3 with the filters loaded L2 -> registers (conv_wino4_kernel: bit-identical
// to v3, 87.7 vs 100.5 ms per B = 32 forward, profiles/r02_wino_v4.md; layers whose channel counts it cannot tile run as in
// mode 3); shapes a mode cannot take fall back to the direct kernel.
static int g_wino_pair = -1;   // -1: take ADM_WINO_PAIR from the environment (default 1) on first use
void set_winograd_pair(int v) { g_wino_pair = v; }
static int wino_pair() {
  if (g_wino_pair < 0) { const char* e = getenv("ADM_WINO_PAIR"); g_wino_pair = e ? atoi(e) : 1; }
  return g_wino_pair;
}
// conv_wino5_kernel (128-cout workgroup tiles, all eight waves MFMA + staging): 1 (default) = used wherever the layer has 128 | Cout
// and its 128-cout tiles fill the chip; 0 = conv_wino4_kernel everywhere; bit 1 (2) = also when the tiles do not fill the chip (tests);
// bit 3 (8) = the two-halves-in-antiphase schedule instead of the interleaved one. Bit-identical results in every case (same filter image).
static int g_wino5 = -1;       // -1: take ADM_WINO5 from the environment (default 1) on first use
void set_winograd_v5(int v) { g_wino5 = v; }
static int wino5_on() {
  if (g_wino5 < 0) { const char* e = getenv("ADM_WINO5"); g_wino5 = e ? atoi(e) : 1; }
  return g_wino5;
}
// conv_wino6_kernel (F(4x4,3x3)): 1 (default) = every layer the kernel tiles (128 | Cout, 32 | Cin, 16 | H, 16 | W) on a plane of at least
// 64x64 pixels whose tiles give ONE SAMPLE at least W6_MIN_WGS workgroups — a function of the LAYER only: F(4x4) is not bit-identical to
// F(2x2), so the choice must not depend on the batch (a sample's bits must not depend on the batch it is sampled in); 0 = F(2x2,3x3) kernels
// everywhere; 2 = every layer the kernel tiles (tests); n >= 16 = planes of at least n x n pixels, whatever their workgroup count.
static int g_wino6 = -1;
void set_winograd_v6(int v) { g_wino6 = v; }
static int wino6_on() {
  if (g_wino6 < 0) { const char* e = getenv("ADM_WINO6"); g_wino6 = e ? atoi(e) : 1; }
  return g_wino6;
}
// The rule, measured (profiles/r05_wino.md §5; captured loop, ms per step). One 16x16x128 tile is 2.25x the work of a 64-cout F(2x2)
// workgroup, so a plane whose tiles do not fill the chip pays for it at small batches — and the choice cannot follow the batch:
//                         B = 32 forward   256x256 B = 16   256x256 B = 4   256x256 B = 1   64x64 model B = 1
//   F(2x2) only                73.8             —                —              7.12             3.43
//   planes >= 256              63.5             —                —              6.79             3.42        ("wino6" = 256: the latency setting)
//   planes >= 128              59.7            30.8             11.3            7.29             3.41
//   default (below)            57.4            29.8             11.7            8.30             3.41
//   planes >= 64               57.7             —                —              8.33             4.00
// The default takes the 64x64 level of the 256x256 model (Cout = 256: 32 workgroups per sample) and leaves the 64x64 model's own top level
// (Cout = 128: 16) on F(2x2): the batched configurations gain 3.6 - 3.8 % over the 128 floor, single-sample sampling at 256x256 loses 14 %.
constexpr int W6_MIN_PLANE = 64, W6_MIN_WGS = 32;
static bool wino6_eligible(const adm_conv_args& a) {
  if (!wino6_on()) return false;
  const int C2 = a.x2 ? a.C2 : 0;
  const int Ho = a.up ? 2 * a.H : a.H, Wo = a.up ? 2 * a.W : a.W;
  if (!wino6_layout(a.Cout, a.C1 + C2) || a.C1 % 16 != 0 || Ho % 16 != 0 || Wo % 16 != 0) return false;
  if (wino6_on() == 2) return true;
  if (wino6_on() >= 16) return Ho >= wino6_on() && Wo >= wino6_on();
  return Ho >= W6_MIN_PLANE && Wo >= W6_MIN_PLANE && (Ho / 16) * (Wo / 16) * (a.Cout / W5BM) >= W6_MIN_WGS;
}
static int g_wino_mode = -1;   // -1: take ADM_CONV_WINO from the environment (default 4) on first use
bool winograd_mode_available(int m) {
#if defined(ADM_EXPERIMENTS)
  return m >= -1 && m <= 4;
#else
  return m == -1 || m == 0 || m == 4;      // modes 1-3 (earlier kernel generations) exist only in -DADM_EXPERIMENTS builds
#endif
}
void set_winograd_mode(int m) { g_wino_mode = winograd_mode_available(m) ? m : 4; }
static int wino_mode() {
  if (g_wino_mode < 0) {
    const char* e = getenv("ADM_CONV_WINO");
    g_wino_mode = e ? atoi(e) : 4;
    if (!winograd_mode_available(g_wino_mode) || g_wino_mode < 0) g_wino_mode = 4;
  }
  return g_wino_mode;
}
bool winograd_enabled() { return wino_mode() != 0; }

// Eligibility: 3x3 stride 1 "same", output at least 8x16 with Wo % 16 == 0 and Ho % 8 == 0, Cin % 8, Cout % 32.
static bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
// what the persistent kernels (v3, v4) need beyond the shape: float4 row loads of the activations, identity GroupNorm rows
// only without SiLU
static bool wino_persistent_args_ok(const adm_conv_args& a) {
  const int C2 = a.x2 ? a.C2 : 0;
  const long x1_bs = a.x1_bstride ? a.x1_bstride : (long)a.C1 * a.H * a.W, x2_bs = a.x2_bstride ? a.x2_bstride : (long)C2 * a.H * a.W;
  return (a.gn_scale != nullptr || !a.act) && aligned16(a.x1) && (a.x2 == nullptr || aligned16(a.x2)) && x1_bs % 4 == 0 &&
         x2_bs % 4 == 0;
}
bool winograd_eligible(const adm_conv_args& a) {
  if (a.ks != 3 || a.stride != 1 || a.pad_lo != 1 || a.w_bstride != 0 || a.wino_packed == nullptr || a.up > 1) return false;
  const int C2 = a.x2 ? a.C2 : 0;
  const int Hi = a.up ? 2 * a.H : a.H, Wi = a.up ? 2 * a.W : a.W;
  if (!(Wi % 16 == 0 && Hi % 8 == 0 && (a.C1 + C2) % 8 == 0 && a.C1 % 8 == 0 && a.Cout % 32 == 0)) return false;
  if (wino4_layout(a.Cout, a.C1 + C2))     // filters are in the v4 image: conv_wino4_kernel or nothing (-> direct kernel)
    return wino_persistent_args_ok(a) && aligned16(a.out) && (a.residual == nullptr || aligned16(a.residual));
#if defined(ADM_EXPERIMENTS)
  return true;                             // modes 1-3
#else
  return false;                            // shapes conv_wino4_kernel cannot tile take the direct MFMA kernel: the only fallback
#endif
}

const float* conv_zero_bias(int n);  // k_conv_mfma.hip

// GroupNorm statistic tiles of the output (one per 8 x 16-pixel tile) — only conv_wino4_kernel has the epilogue
int winograd_stats_tiles(const adm_conv_args& a) {
  const int C2 = a.x2 ? a.C2 : 0;
  if (!wino4_layout(a.Cout, a.C1 + C2)) return 0;
  const int Ho = a.up ? 2 * a.H : a.H, Wo = a.up ? 2 * a.W : a.W;
  if (wino6_eligible(a)) return (Wo / 16) * (Ho / 16);          // conv_wino6_kernel: one (sum, sum of squares) per 16x16-pixel tile
  return (Wo / 16) * (Ho / 8);
}

int launch_conv_winograd(const adm_conv_args& a, hipStream_t st) {
  WinoParams p;
  p.stats = nullptr;
  p.tune = 0;
  const int C2 = a.x2 ? a.C2 : 0;
  p.x1 = a.x1; p.x2 = a.x2; p.C1 = a.C1; p.C2 = C2;
  p.N = a.N; p.Hs = a.H; p.Ws = a.W;
  p.Hi = a.up ? 2 * a.H : a.H; p.Wi = a.up ? 2 * a.W : a.W;
  p.Ho = p.Hi; p.Wo = p.Wi; p.up = a.up;
  p.gn_scale = a.gn_scale; p.gn_shift = a.gn_shift; p.act = a.act;
  p.wu = a.wino_packed; p.bias = a.bias ? a.bias : conv_zero_bias(a.Cout); p.Cout = a.Cout;
  ADM_REQUIRE(p.bias != nullptr, "conv_winograd: zero-bias buffer");
  p.chan_add = a.chan_add; p.chan_add_stride = a.chan_add_stride;
  p.residual = a.residual; p.out = a.out;
  p.tiles_x = p.Wo / 16; p.tiles_y = p.Ho / 8; p.n_ct = a.Cout / WBM;
  p.nblk = p.tiles_x * p.tiles_y * a.N * p.n_ct;
  p.x1_bs = a.x1_bstride ? a.x1_bstride : (long)a.C1 * a.H * a.W;
  p.x2_bs = a.x2_bstride ? a.x2_bstride : (long)C2 * a.H * a.W;
  const bool v4 = wino4_layout(a.Cout, a.C1 + C2);             // (winograd_eligible has checked the kernel's other needs)
#if !defined(ADM_EXPERIMENTS)
  ADM_REQUIRE(v4, "conv_winograd: shape outside conv_wino4_kernel's tiling (winograd_eligible should have said no)");
#endif
  if (v4 || (wino_mode() >= 3 && a.Cout % W3BM == 0 && (a.C1 + C2) % (2 * WCK) == 0 && wino_persistent_args_ok(a))) {
    // persistent wave-specialised kernels
    p.n_ct = a.Cout / W3BM;
    p.nblk = p.tiles_x * p.tiles_y * a.N * p.n_ct;
    p.gn_nstride = a.C1 + C2;
    if (p.gn_scale == nullptr) {                               // no GroupNorm on the load path: identity affine rows
      p.gn_scale = conv_const_ones(a.C1 + C2); p.gn_shift = conv_zero_bias(a.C1 + C2); p.gn_nstride = 0;
      ADM_REQUIRE(p.gn_scale != nullptr && p.gn_shift != nullptr, "conv_winograd: constant buffers");
    }
    if (p.chan_add == nullptr) { p.chan_add = conv_zero_bias(a.Cout); p.chan_add_stride = 0; }
    ADM_REQUIRE(p.chan_add != nullptr, "conv_winograd: zero-bias buffer");
#if !defined(ADM_EMU)
    // Per device (a function-local `static once` ran for the device that happened to be current on first use only):
    // the CU count, and the permission for the PAIR kernels' 91 KiB of dynamic LDS — checked; where the runtime refuses it the
    // bit-identical one-barrier-per-chunk instantiation (43 KiB, no attribute needed) runs instead.
    struct DevInfo { int n_cu = 0; bool pair_ok = false; bool v5_ok = false; };
    static DevInfo info[16];
    static std::mutex info_mu;
    const int dslot = conv_dev_slot() & 15;
    {
      std::lock_guard<std::mutex> lk(info_mu);
      if (info[dslot].n_cu == 0) {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        info[dslot].n_cu = n > 0 ? n : 256;
        const int by = (int)(sizeof(float) * W4LDS_PAIR);
        bool ok = true;
        ok &= hipFuncSetAttribute((const void*)conv_wino4_kernel<true, false, 0, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
        ok &= hipFuncSetAttribute((const void*)conv_wino4_kernel<true, false, 0, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
        ok &= hipFuncSetAttribute((const void*)conv_wino4_kernel<false, false, 0, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
        ok &= hipFuncSetAttribute((const void*)conv_wino4_kernel<false, false, 0, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        info[dslot].pair_ok = ok;
        bool ok5 = true;
        ok5 &= hipFuncSetAttribute((const void*)conv_wino5_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
        ok5 &= hipFuncSetAttribute((const void*)conv_wino5_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
        ok5 &= hipFuncSetAttribute((const void*)conv_wino5_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
        ok5 &= hipFuncSetAttribute((const void*)conv_wino5_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
        ok5 &= hipFuncSetAttribute((const void*)conv_wino5_kernel<true, 1, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
        ok5 &= hipFuncSetAttribute((const void*)conv_wino5_kernel<true, 0, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
        ok5 &= hipFuncSetAttribute((const void*)conv_wino5_kernel<false, 1, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
        ok5 &= hipFuncSetAttribute((const void*)conv_wino5_kernel<false, 0, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, by) == hipSuccess;
        if (!ok5) (void)hipGetLastError();
        info[dslot].v5_ok = ok5;
#if defined(ADM_EXPERIMENTS)
        (void)hipFuncSetAttribute((const void*)conv_wino4_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * W4LDS));
        (void)hipFuncSetAttribute((const void*)conv_wino4_kernel<false, true, 0, -1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, by);
        (void)hipFuncSetAttribute((const void*)conv_wino3_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * W3LDS));
        (void)hipFuncSetAttribute((const void*)conv_wino3_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * W3LDS));
        (void)hipFuncSetAttribute((const void*)conv_wino3_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * W3LDS));
#endif
      }
    }
    const int n_cu = info[dslot].n_cu;
    const bool pair_ok = info[dslot].pair_ok;
    const bool v5_ok = info[dslot].v5_ok;
#else
    const int n_cu = 3;                                          // exercise persistence (several tiles per block) on the emulator
    const bool pair_ok = true, v5_ok = true;
#endif
    if (v4 && wino6_eligible(a)) {                               // F(4x4,3x3): chosen by the layer alone (see wino6_on)
#if !defined(ADM_EMU)
      static bool attr6[16] = {};
      std::lock_guard<std::mutex> lk6(info_mu);
      if (!attr6[dslot]) {
        const int by6 = (int)(sizeof(float) * W6LDS);
        bool ok6 = true;
        ok6 &= hipFuncSetAttribute((const void*)conv_wino6_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, by6) == hipSuccess;
        ok6 &= hipFuncSetAttribute((const void*)conv_wino6_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, by6) == hipSuccess;
        ok6 &= hipFuncSetAttribute((const void*)conv_wino6_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, by6) == hipSuccess;
        ok6 &= hipFuncSetAttribute((const void*)conv_wino6_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, by6) == hipSuccess;
        ADM_REQUIRE(ok6, "conv_winograd: the runtime refused 125 KiB of dynamic LDS for conv_wino6_kernel");
        attr6[dslot] = true;
      }
#endif
      p.tiles_x = p.Wo / 16; p.tiles_y = p.Ho / 16;
      p.n_ct = a.Cout / W5BM;
      p.nblk = p.tiles_x * p.tiles_y * a.N * p.n_ct;
      p.wu = a.wino_packed + (long)a.Cout * (a.C1 + C2) * 16;    // the F(4x4) image follows the F(2x2) image
      p.prof = nullptr;
      p.stats = a.stats_out;
      set_last_conv_variant(4000 + 316);
#if defined(W6X_PROF) && !defined(ADM_EMU)
      static unsigned long long* dprof6 = [] { void* q = nullptr; (void)hipMalloc(&q, 16 * sizeof(unsigned long long)); return (unsigned long long*)q; }();
      (void)hipMemsetAsync(dprof6, 0, 16 * sizeof(unsigned long long), st);
      p.prof = dprof6;
#endif
      const size_t need6 = sizeof(float) * W6LDS;
      const int grid6 = p.nblk < n_cu ? p.nblk : n_cu;
      if (a.up) {
        if (a.act) ADM_LAUNCH((conv_wino6_kernel<true, 1>), dim3(grid6), dim3(512), need6, st, p);
        else ADM_LAUNCH((conv_wino6_kernel<true, 0>), dim3(grid6), dim3(512), need6, st, p);
      } else {
        if (a.act) ADM_LAUNCH((conv_wino6_kernel<false, 1>), dim3(grid6), dim3(512), need6, st, p);
        else ADM_LAUNCH((conv_wino6_kernel<false, 0>), dim3(grid6), dim3(512), need6, st, p);
      }
#if defined(W6X_PROF) && !defined(ADM_EMU)
      {
        unsigned long long h[16];
        (void)hipMemcpyAsync(h, dprof6, sizeof(h), hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        const double nb = grid6;
        fprintf(stderr, "[wino6 prof] cycles of wave 0: total %.0f M %.0f B %.0f epilogue %.0f A %.0f C %.0f barrier %.0f other %.0f | wave 4: total %.0f M %.0f B %.0f epilogue %.0f A %.0f C %.0f barrier %.0f other %.0f\n",
                h[0] / nb, h[1] / nb, h[2] / nb, h[3] / nb, h[4] / nb, h[5] / nb, h[6] / nb, h[7] / nb,
                h[8] / nb, h[9] / nb, h[10] / nb, h[11] / nb, h[12] / nb, h[13] / nb, h[14] / nb, h[15] / nb);
      }
#endif
      return ADM_CHECK_LAUNCH();
    }
    // v5: 128-cout workgroup tiles (every patch transformed once per 128 couts), taken when those tiles still fill the chip — with
    // fewer, v4's 64-cout tiles are twice as many workgroups. The two kernels are bit-identical (same filter image, same summation
    // order), so this batch-dependent choice cannot move a sample's bits.
    if (v4 && v5_ok && wino5_on() && a.Cout % W5BM == 0) {
      const int nblk5 = p.tiles_x * p.tiles_y * a.N * (a.Cout / W5BM);
      if (nblk5 >= n_cu || (wino5_on() & 2)) {         // ("wino5" bit 1: wherever the shape allows — tests on small tensors)
        p.n_ct = a.Cout / W5BM;
        p.nblk = nblk5;
        p.prof = nullptr;
        p.stats = a.stats_out;
#if !defined(ADM_EMU)
        static const int tune = [] { const char* e = getenv("ADM_WINO5_TUNE"); return e ? atoi(e) : 1; }();
        p.tune = tune;
#endif
        set_last_conv_variant(4000 + 315);
        const size_t need5 = sizeof(float) * W4LDS_PAIR;
        const int grid5 = nblk5 < n_cu ? nblk5 : n_cu;
#if !defined(ADM_EMU) && defined(ADM_EXPERIMENTS)
        static const int abl5 = [] { const char* e = getenv("ADM_WINO5_ABL"); return e ? atoi(e) : 0; }();
        static const bool prof5 = getenv("ADM_WINO5_PROF") != nullptr;
        if (!a.up && a.act && (abl5 || prof5)) {   // developer aids (see wino5_wave): role / stage ablations (TIMING ONLY) and cycle accounting
#define W5_EXP(A, P)                                                                                                                \
  do {                                                                                                                              \
    (void)hipFuncSetAttribute((const void*)conv_wino5_kernel<false, 1, A, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need5); \
    ADM_LAUNCH((conv_wino5_kernel<false, 1, A, P>), dim3(grid5), dim3(512), need5, st, p);                                           \
  } while (0)
          if (prof5) {
            static unsigned long long* dprof = [] { void* q = nullptr; (void)hipMalloc(&q, 16 * sizeof(unsigned long long)); return (unsigned long long*)q; }();
            (void)hipMemsetAsync(dprof, 0, 16 * sizeof(unsigned long long), st);
            p.prof = dprof;
            W5_EXP(0, true);
            unsigned long long h[16];
            (void)hipMemcpyAsync(h, dprof, sizeof(h), hipMemcpyDeviceToHost, st);
            (void)hipStreamSynchronize(st);
            const double nb = 2.0 * grid5;      // two sampled waves (tid 0 / 256 of each half... one per half: waves 0 and 4) — see wino5_wave
            fprintf(stderr, "[wino5 prof] per-wave cycles, first half: total %.0f M %.0f C %.0f B %.0f A %.0f barrier %.0f epilogue %.0f | second half: total %.0f M %.0f C %.0f B %.0f A %.0f barrier %.0f epilogue %.0f\n",
                    h[0] / nb * 2, h[1] / nb * 2, h[2] / nb * 2, h[5] / nb * 2, h[6] / nb * 2, h[3] / nb * 2, h[4] / nb * 2, h[8] / nb * 2, h[9] / nb * 2, h[10] / nb * 2, h[13] / nb * 2, h[14] / nb * 2, h[11] / nb * 2, h[12] / nb * 2);
            return ADM_CHECK_LAUNCH();
          }
          static unsigned long long* dcyc = [] { void* q = nullptr; (void)hipMalloc(&q, 16 * sizeof(unsigned long long)); return (unsigned long long*)q; }();
          (void)hipMemsetAsync(dcyc, 0, 16 * sizeof(unsigned long long), st);
          p.prof = dcyc;
          switch (abl5) {
            case 1000: (void)hipFuncSetAttribute((const void*)conv_wino5_kernel<false, 1, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need5);
                       ADM_LAUNCH((conv_wino5_kernel<false, 1, 0, false, true>), dim3(grid5), dim3(512), need5, st, p); break;   // the interleaved schedule
            case 1007: (void)hipFuncSetAttribute((const void*)conv_wino5_kernel<false, 1, 7, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need5);
                       ADM_LAUNCH((conv_wino5_kernel<false, 1, 7, false, true>), dim3(grid5), dim3(512), need5, st, p); break;
            case 999: W5_EXP(0, false); break;      // the product kernel, with the cycle count
            case 7: W5_EXP(7, false); break;        // no staging: MFMA blocks + barriers
            case 56: W5_EXP(56, false); break;      // no MFMA block at all (no MFMAs, no operand fetch): staging + barriers
            case 32: W5_EXP(32, false); break;      // no LDS operand reads
            case 16: W5_EXP(16, false); break;      // no filter loads
            case 48: W5_EXP(48, false); break;      // bare MFMAs beside working staging
            case 55: W5_EXP(55, false); break;      // bare MFMAs, no staging: the pipe's own pace under this schedule
            case 64: W5_EXP(64, false); break;      // no barriers
            case 1: W5_EXP(1, false); break;
            case 2: W5_EXP(2, false); break;
            case 4: W5_EXP(4, false); break;
            case 128: W5_EXP(128, false); break;    // no bias / residual fold
            case 8: W5_EXP(8, false); break;        // MFMAs replaced by two FMAs (operands still fetched)
            default: W5_EXP(119, false); break;     // 119 = 55 | 64: bare MFMAs, nothing else
          }
#undef W5_EXP
          {
            unsigned long long h[16];
            (void)hipMemcpyAsync(h, dcyc, sizeof(h), hipMemcpyDeviceToHost, st);
            (void)hipStreamSynchronize(st);
            fprintf(stderr, "[wino5 cycles] ABL %d: %.0f cycles per wave (waves 0 / 4 of every workgroup)\n", abl5, (double)(h[0] + h[8]) / (2.0 * grid5));
          }
          return ADM_CHECK_LAUNCH();
        }
#endif
        if (!(wino5_on() & 8)) {                      // the interleaved schedule (INTER: the default); "wino5" bit 3 = the two halves in antiphase
          if (a.up) {
            if (a.act) ADM_LAUNCH((conv_wino5_kernel<true, 1, 0, false, true>), dim3(grid5), dim3(512), need5, st, p);
            else ADM_LAUNCH((conv_wino5_kernel<true, 0, 0, false, true>), dim3(grid5), dim3(512), need5, st, p);
          } else {
            if (a.act) ADM_LAUNCH((conv_wino5_kernel<false, 1, 0, false, true>), dim3(grid5), dim3(512), need5, st, p);
            else ADM_LAUNCH((conv_wino5_kernel<false, 0, 0, false, true>), dim3(grid5), dim3(512), need5, st, p);
          }
          return ADM_CHECK_LAUNCH();
        }
        if (a.up) {
          if (a.act) ADM_LAUNCH((conv_wino5_kernel<true, 1>), dim3(grid5), dim3(512), need5, st, p);
          else ADM_LAUNCH((conv_wino5_kernel<true, 0>), dim3(grid5), dim3(512), need5, st, p);
        } else {
          if (a.act) ADM_LAUNCH((conv_wino5_kernel<false, 1>), dim3(grid5), dim3(512), need5, st, p);
          else ADM_LAUNCH((conv_wino5_kernel<false, 0>), dim3(grid5), dim3(512), need5, st, p);
        }
        return ADM_CHECK_LAUNCH();
      }
    }
    const int grid = p.nblk < n_cu ? p.nblk : n_cu;
    set_last_conv_variant(4000 + (v4 ? 314 : 313));
    p.prof = nullptr;
    p.stats = v4 ? a.stats_out : nullptr;
#if !defined(ADM_EMU) && defined(ADM_EXPERIMENTS)
    static const bool want_prof = getenv("ADM_WINO_PROF") != nullptr;
    if (want_prof && !a.up) {   // developer aid: per-role cycle accounting, printed after every launch (synchronous)
      static unsigned long long* dprof = [] { void* q = nullptr; (void)hipMalloc(&q, 16 * sizeof(unsigned long long)); return (unsigned long long*)q; }();
      (void)hipMemsetAsync(dprof, 0, 16 * sizeof(unsigned long long), st);
      p.prof = dprof;
      if (v4 && wino_pair()) ADM_LAUNCH((conv_wino4_kernel<false, true, 0, -1, true>), dim3(grid), dim3(512), sizeof(float) * W4LDS_PAIR, st, p);
      else if (v4) ADM_LAUNCH((conv_wino4_kernel<false, true>), dim3(grid), dim3(512), sizeof(float) * W4LDS, st, p);
      else ADM_LAUNCH((conv_wino3_kernel<false, true>), dim3(grid), dim3(512), sizeof(float) * W3LDS, st, p);
      unsigned long long h[16];
      (void)hipMemcpyAsync(h, dprof, sizeof(h), hipMemcpyDeviceToHost, st);
      (void)hipStreamSynchronize(st);
      const double nb = grid;
      fprintf(stderr, "[wino3 prof] per-block cycles: consumer total %.0f drain %.0f barrier %.0f epilogue %.0f | producer total %.0f drain %.0f barrier %.0f C %.0f B %.0f A %.0f\n",
              h[0] / nb, h[1] / nb, h[2] / nb, h[3] / nb, h[8] / nb, h[9] / nb, h[10] / nb, h[11] / nb, h[12] / nb, h[13] / nb);
      return ADM_CHECK_LAUNCH();
    }
    static const int abl = [] { const char* e = getenv("ADM_WINO_ABL"); return e ? atoi(e) : 0; }();
    if (v4 && !a.up && abl) {   // developer aid: role ablations of conv_wino4_kernel (TIMING ONLY, wrong results)
      switch (abl) {
        case 1: ADM_LAUNCH((conv_wino4_kernel<false, false, 1>), dim3(grid), dim3(512), sizeof(float) * W4LDS, st, p); break;
        case 2: ADM_LAUNCH((conv_wino4_kernel<false, false, 2>), dim3(grid), dim3(512), sizeof(float) * W4LDS, st, p); break;
        case 3: ADM_LAUNCH((conv_wino4_kernel<false, false, 3>), dim3(grid), dim3(512), sizeof(float) * W4LDS, st, p); break;
        case 4: ADM_LAUNCH((conv_wino4_kernel<false, false, 4>), dim3(grid), dim3(512), sizeof(float) * W4LDS, st, p); break;
        case 5: ADM_LAUNCH((conv_wino4_kernel<false, false, 5>), dim3(grid), dim3(512), sizeof(float) * W4LDS, st, p); break;
        case 6: ADM_LAUNCH((conv_wino4_kernel<false, false, 6>), dim3(grid), dim3(512), sizeof(float) * W4LDS, st, p); break;
        case 7: ADM_LAUNCH((conv_wino4_kernel<false, false, 7>), dim3(grid), dim3(512), sizeof(float) * W4LDS, st, p); break;
        case 8: ADM_LAUNCH((conv_wino4_kernel<false, false, 8>), dim3(grid), dim3(512), sizeof(float) * W4LDS, st, p); break;
        case 9: ADM_LAUNCH((conv_wino4_kernel<false, false, 9>), dim3(grid), dim3(512), sizeof(float) * W4LDS, st, p); break;
        case 12: ADM_LAUNCH((conv_wino4_kernel<false, false, 12>), dim3(grid), dim3(512), sizeof(float) * W4LDS, st, p); break;
        case 13: ADM_LAUNCH((conv_wino4_kernel<false, false, 13>), dim3(grid), dim3(512), sizeof(float) * W4LDS, st, p); break;
        case 10: ADM_LAUNCH((conv_wino4_kernel<false, false, 10>), dim3(grid), dim3(512), sizeof(float) * W4LDS, st, p); break;
        default: ADM_LAUNCH((conv_wino4_kernel<false, false, 11>), dim3(grid), dim3(512), sizeof(float) * W4LDS, st, p); break;
      }
      return ADM_CHECK_LAUNCH();
    }
#endif
    if (v4) {
      // default since round 3: one workgroup barrier per TWO chunks (rings of four V slabs / patch buffers, 91 KiB of LDS). Measured on
      // one box, alternating, bit-identical outputs: 55 launches of a B = 32 forward 67.23 / 67.08 / 67.13 ms at one barrier per chunk,
      // 66.07 / 66.21 / 66.37 ms at one per pair; ADM_WINO_PAIR=0 restores the former.
      if (wino_pair() && pair_ok) {
        const size_t needp = sizeof(float) * W4LDS_PAIR;
        if (a.up) {
          if (a.act) ADM_LAUNCH((conv_wino4_kernel<true, false, 0, 1, true>), dim3(grid), dim3(512), needp, st, p);
          else ADM_LAUNCH((conv_wino4_kernel<true, false, 0, 0, true>), dim3(grid), dim3(512), needp, st, p);
        } else {
          if (a.act) ADM_LAUNCH((conv_wino4_kernel<false, false, 0, 1, true>), dim3(grid), dim3(512), needp, st, p);
          else ADM_LAUNCH((conv_wino4_kernel<false, false, 0, 0, true>), dim3(grid), dim3(512), needp, st, p);
        }
        return ADM_CHECK_LAUNCH();
      }
      const size_t need4 = sizeof(float) * W4LDS;
      if (a.up) {
        if (a.act) ADM_LAUNCH((conv_wino4_kernel<true, false, 0, 1>), dim3(grid), dim3(512), need4, st, p);
        else ADM_LAUNCH((conv_wino4_kernel<true, false, 0, 0>), dim3(grid), dim3(512), need4, st, p);
      } else {
        if (a.act) ADM_LAUNCH((conv_wino4_kernel<false, false, 0, 1>), dim3(grid), dim3(512), need4, st, p);
        else ADM_LAUNCH((conv_wino4_kernel<false, false, 0, 0>), dim3(grid), dim3(512), need4, st, p);
      }
      return ADM_CHECK_LAUNCH();
    }
#if defined(ADM_EXPERIMENTS)
    if (a.up) ADM_LAUNCH((conv_wino3_kernel<true, false>), dim3(grid), dim3(512), sizeof(float) * W3LDS, st, p);
    else ADM_LAUNCH((conv_wino3_kernel<false, false>), dim3(grid), dim3(512), sizeof(float) * W3LDS, st, p);
    return ADM_CHECK_LAUNCH();
#endif
  }
#if defined(ADM_EXPERIMENTS)
  if (wino_mode() == 2 && a.Cout % W2BM == 0 && (a.C1 + C2) % (2 * WCK) == 0) {   // wave-specialised kernel (even chunk count)
    p.n_ct = a.Cout / W2BM;
    p.nblk = p.tiles_x * p.tiles_y * a.N * p.n_ct;
    const size_t need2 = sizeof(float) * 16 * W2BM * 32;  // 128 KiB (epilogue) >= 2*16 + 2*32 KiB (main loop)
#if !defined(ADM_EMU)
    static bool once2 = [] {
      (void)hipFuncSetAttribute((const void*)conv_wino2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      return true;
    }();
    (void)once2;
#endif
    set_last_conv_variant(4000 + 312);
    ADM_LAUNCH(conv_wino2_kernel, dim3(p.nblk), dim3(512), need2, st, p);
    return ADM_CHECK_LAUNCH();
  }
  const size_t smem = sizeof(float) * 16 * WBM * 32;  // 64 KiB
  const size_t main_need = sizeof(float) * (1472 + 2 * WVSLAB + 2 * WUSLAB);
  const size_t need = smem > main_need ? smem : main_need;
#if !defined(ADM_EMU)
  static bool once = [] {
    (void)hipFuncSetAttribute((const void*)conv_wino_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    return true;
  }();
  (void)once;
#endif
  set_last_conv_variant(4000 + 311);
  ADM_LAUNCH(conv_wino_kernel, dim3(p.nblk), dim3(256), need, st, p);
  return ADM_CHECK_LAUNCH();
#else
  ADM_FAIL("conv_winograd: no kernel for this shape in a build without ADM_EXPERIMENTS");
#endif
}

}  // namespace adm
