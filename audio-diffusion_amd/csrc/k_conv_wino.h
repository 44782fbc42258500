// k_conv_wino.h — what the Winograd convolution's translation units share: the launch parameters, the persistent workgroups' tile walk,
// the tile constants, and the per-generation launchers.
//   k_conv_wino.hip     filter packing, options, eligibility and the dispatch (launch_conv_winograd)
//   k_conv_wino_f2.hip  F(2x2,3x3): conv_wino4_kernel (64-cout tiles, producer / consumer waves), conv_wino5_kernel (128-cout tiles)
//   k_conv_wino_f4.hip  F(4x4,3x3): conv_wino6_kernel (128 couts x 16x16 pixels) — the headline kernel
// The kernel generations before these (v1-v3), their ablation / cycle-accounting instantiations and the -DADM_EXPERIMENTS build that carried
// them were retired in round 6: their measurements are in profiles/ (r01-r05) and docs/history/, their sources in the git history.
#pragma once
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
  double* stats;              // optional: GroupNorm partial sums of the output, [n][cout][tile][2] (adm_conv_args.stats_out)
  int tune;                   // conv_wino5_kernel: developer switches (ADM_WINO5_TUNE; bit 0 = s_setprio 1 for waves 4-7)
  // conv_wino4_kernel, split K (the single-sample rule, "single_sample"): ksplit workgroups share one output tile, each walks cps of the
  // layer's 8-channel chunks and writes its partial sums (no bias / per-sample term / residual / statistics) to slab kpart of `out`
  // (slabs part_stride floats apart); nblk counts (tile, part) pairs. ksplit = 1: cps = every chunk, part_stride unused.
  int ksplit, cps; long part_stride;
};

__device__ __forceinline__ float silu_w(float v) { return v * ADM_RCP(1.0f + __expf(-v)); }

constexpr int WCK = 8;            // input channels per chunk
constexpr int WPH = 10, WPW = 18; // haloed patch of an 8x16 output tile
constexpr int WCS = WPH * WPW;    // 180
constexpr int W3BM = 64;                    // couts of a conv_wino4_kernel workgroup tile
constexpr int W3VSLAB = 16 * WCK * 32;      // 4096 floats: V slab of one chunk, F(2x2): [point 16][channel 8][tile 32]
// LDS pitch of a patch row in the wave-specialised kernels: 24 words instead of the 18 the patch is wide. Stage C reads the 4x4 windows of
// a channel's 32 tiles with ds_read2_b64 at word offsets 2 tyy P + 2 txx: with P = 18 the four tile rows start at banks 0 / 36 / 8 / 44
// and overlap pairwise (the 0.23 LDS conflict ratio of rounds 2-3); with P = 24 they start at 0 / 48 / 32 / 16 — conflict-free.
constexpr int WPP = 24;
constexpr int W3PSLAB = WCK * WPH * WPP + 264;   // 1920 floats (x2-upsample variant: 480) + one dummy word per producer lane

constexpr int W4LDS_PAIR = 4 * W3VSLAB + 4 * W3PSLAB;     // rings of four V slabs / patch buffers (91 KiB): conv_wino4_kernel, conv_wino5_kernel
constexpr int W5BM = 128;                   // couts of a conv_wino5_kernel / conv_wino6_kernel workgroup tile
constexpr int W4ABLK = 4 * 2 * 64 * 4;      // floats of one (chunk, 16-cout block) F(2x2) filter image: 8 KiB  (kernels and packers)
constexpr int W6ABLK = 2 * 9 * 64 * 4;      // ... of the F(4x4) image: 18 KiB

struct Wino3Tile { int n, ty, tx, m0, kpart; };

__device__ __forceinline__ Wino3Tile wino3_tile(const WinoParams& p, int v) {
  // bijective XCD-aware remap of the virtual block id (v & 7 == XCD of the persistent block that owns it)
  const int q = p.nblk >> 3, r = p.nblk & 7, xcd = v & 7;
  int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
  Wino3Tile t;
  t.kpart = 0;
  if (p.ksplit > 1) { t.kpart = lid % p.ksplit; lid /= p.ksplit; }      // the parts of a tile are neighbours (one XCD, one filter slab)
  const int ct = lid % p.n_ct, pt = lid / p.n_ct;
  t.tx = pt % p.tiles_x; t.ty = (pt / p.tiles_x) % p.tiles_y; t.n = pt / (p.tiles_x * p.tiles_y);
  t.m0 = ct * W3BM;
  return t;
}

__device__ __forceinline__ Wino3Tile wino5_tile(const WinoParams& p, int v) {
  const int q = p.nblk >> 3, r = p.nblk & 7, xcd = v & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
  const int ct = lid % p.n_ct, pt = lid / p.n_ct;
  Wino3Tile t;
  t.tx = pt % p.tiles_x; t.ty = (pt / p.tiles_x) % p.tiles_y; t.n = pt / (p.tiles_x * p.tiles_y);
  t.m0 = ct * W5BM; t.kpart = 0;
  return t;
}

// per-generation launchers (each asks ONCE per device, under a lock, for its kernels' dynamic LDS and fails loudly where the runtime refuses)
int launch_wino4(const WinoParams& p, bool up, bool act, int grid, hipStream_t st);                    // k_conv_wino_f2.hip
int launch_wino5(const WinoParams& p, bool up, bool act, int grid, bool two_halves, hipStream_t st);   // k_conv_wino_f2.hip
int launch_wino6(const WinoParams& p, bool up, bool act, int grid, hipStream_t st);                    // k_conv_wino_f4.hip
constexpr int W6_STATS_TILE = 16;           // conv_wino6_kernel: one (sum, sum of squares) per 16x16-pixel tile (F(2x2) kernels: 8x16)

}  // namespace adm
