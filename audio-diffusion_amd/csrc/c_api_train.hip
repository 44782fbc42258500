// c_api_train.hip — op-level C-ABI entry points of the backward kernels (parity tests; the training executor calls
// the launchers directly).
#include "adm_kernels.h"

using namespace adm;

extern "C" {

int adm_groupnorm_stats_ex(const float* x1, int C1, const float* x2, int C2, int N, int HW, int groups, float eps,
                           const float* gamma, const float* beta, float* scale, float* shift, float* mean_rstd,
                           void* stream) {
  ADM_REQUIRE(x1 && gamma && beta && scale && shift, "groupnorm_stats_ex: null argument");
  return launch_groupnorm_stats(x1, C1, x2, C2, N, HW, groups, eps, gamma, beta, scale, shift, (hipStream_t)stream,
                                mean_rstd);
}

int adm_groupnorm_backward(const float* x1, int C1, const float* x2, int C2, const float* da, int N, int HW, int groups,
                           const float* mean_rstd, const float* gamma, const float* beta, int act, float* s12_scratch,
                           float* dgamma, float* dbeta, float* dx1, int acc1, float* dx2, int acc2, void* stream) {
  ADM_REQUIRE(x1 && da && mean_rstd && gamma && beta && s12_scratch && dgamma && dbeta && dx1, "groupnorm_backward: null");
  return launch_gn_backward(x1, C1, x2, C2, da, N, HW, groups, mean_rstd, gamma, beta, act, s12_scratch, dgamma, dbeta,
                            dx1, acc1, dx2, acc2, (hipStream_t)stream);
}

long adm_conv_wgrad_workspace(const adm_conv_args* a) { return a ? conv_wgrad_workspace(*a, nullptr) : 0; }

int adm_conv2d_wgrad(const adm_conv_args* a, const float* dy, float* dW, int accumulate, float* workspace, void* stream) {
  ADM_REQUIRE(a && a->x1 && dy && dW && workspace, "conv2d_wgrad: null argument");
  return launch_conv_wgrad(*a, dy, dW, accumulate, workspace, (hipStream_t)stream);
}

size_t adm_blocked_image_bytes(int N, int C, int H, int W) { return blk_image_bytes(N, C, H, W); }
long adm_blocked_sums_scratch(int N, int C, int H, int W) { return blk_sums_scratch(N, C, H, W); }
int adm_blocked_apply(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, const float* scale,
                      const float* shift, int act, int zero_insert, void* img, float* sum_scratch, float* sum_nc, int nc_stride,
                      float* sum_c, void* stream) {
  ADM_REQUIRE(x1 && img, "blocked_apply: null argument");
  ADM_REQUIRE(sum_scratch != nullptr || (sum_nc == nullptr && sum_c == nullptr), "blocked_apply: sums need sum_scratch");
  ADM_REQUIRE(zero_insert >= 0 && zero_insert <= 2, "blocked_apply: zero_insert is 0, 1 (odd pixels) or 2 (even pixels)");
  ADM_TRY(launch_blk_apply(x1, C1, 0, x2, x2 ? C2 : 0, 0, N, H, W, scale, shift, act, img, sum_scratch, (hipStream_t)stream,
                           zero_insert));
  if (sum_scratch == nullptr) return 0;
  return launch_blk_sums_finalize(sum_scratch, N, C1 + (x2 ? C2 : 0), H, W, sum_nc, nc_stride, 0, sum_c, (hipStream_t)stream);
}
int adm_conv2d_bf16_blocked_eligible(int Cin, int Cout, int H, int W) { return conv_bf16b_eligible(Cin, Cout, H, W) ? 1 : 0; }
int adm_conv2d_bf16_blocked(const void* img, int Cin, int N, int H, int W, const void* wb, int Cout, const float* bias,
                            const float* chan_add, int chan_add_stride, const float* residual, float* out, int up,
                            double* stats_out, void* stream) {
  ADM_REQUIRE(img && wb && out, "conv2d_bf16_blocked: null argument");
  return launch_conv_bf16b(img, Cin, N, H, W, wb, Cout, bias, chan_add, chan_add_stride, residual, out, (hipStream_t)stream, up,
                           stats_out);
}
int adm_conv2d_wgrad_bf16_blocked_eligible(int Cin, int Cout, int H, int W) { return conv_wgradb_eligible(Cin, Cout, H, W) ? 1 : 0; }
long adm_conv_wgrad_blocked_workspace(int Cin, int Cout, int N, int H, int W) {
  return conv_wgradb_eligible(Cin, Cout, H, W, N) ? conv_wgradb_workspace(Cin, Cout, N, H, W, nullptr) : 0;
}
int adm_conv2d_wgrad_bf16_blocked(const void* x_img, int Cin, const void* dy_img, int Cout, int N, int H, int W, float* dW,
                                  int accumulate, float* workspace, int up, void* stream) {
  ADM_REQUIRE(x_img && dy_img && dW && workspace, "conv2d_wgrad_bf16_blocked: null argument");
  return launch_conv_wgradb(x_img, Cin, dy_img, Cout, N, H, W, dW, accumulate, workspace, (hipStream_t)stream, up);
}

int adm_sumpool2x2(const float* in, float* out, int H, int W, long planes, int accumulate, void* stream) {
  ADM_REQUIRE(in && out, "sumpool2x2: null argument");
  return launch_sumpool2x2(in, out, H, W, planes, accumulate, (hipStream_t)stream);
}

int adm_accumulate(float* dst, long dst_bs, const float* src, long src_bs, long per_sample, int N, int accumulate,
                   void* stream) {
  ADM_REQUIRE(dst && src, "accumulate: null argument");
  return launch_accumulate(dst, dst_bs, src, src_bs, per_sample, N, accumulate, (hipStream_t)stream);
}

int adm_chan_sums(const float* dy, int N, int C, int HW, float* out_nc, int nc_stride, int nc_accumulate, float* out_c,
                  void* stream) {
  ADM_REQUIRE(dy, "chan_sums: null argument");
  return launch_chan_sums(dy, N, C, HW, out_nc, nc_stride, nc_accumulate, out_c, (hipStream_t)stream);
}

int adm_attention_backward(const float* qkv, const float* dout, float* dqkv, int N, int C, int T, int head_dim,
                           void* stream) {
  ADM_REQUIRE(qkv && dout && dqkv, "attention_backward: null argument");
  return launch_attention_bwd(qkv, dout, dqkv, N, C, T, head_dim, (hipStream_t)stream);
}

int adm_linear_backward(const float* dY, int ldy, const float* X, const float* W, int B, int J, int K, int x_silu,
                        float* dW, float* db, float* dX, void* stream) {
  ADM_REQUIRE(dY && X, "linear_backward: null argument");
  return launch_linear_bwd(dY, ldy, X, W, B, J, K, x_silu, dW, db, dX, (hipStream_t)stream);
}

int adm_conv_small_cin_wgrad(const float* x, int Cin, int N, int H, int W, const float* dy, int Cout, float* dW,
                             void* stream) {
  ADM_REQUIRE(x && dy && dW, "conv_small_cin_wgrad: null argument");
  return launch_conv_small_cin_wgrad(x, Cin, N, H, W, dy, Cout, dW, (hipStream_t)stream);
}

int adm_conv_small_cout_backward(const float* x, int Cin, int N, int H, int W, const float* gn_scale,
                                 const float* gn_shift, int act, const float* w, const float* dy, int Cout, float* da,
                                 float* dW, void* stream) {
  ADM_REQUIRE(x && w && dy, "conv_small_cout_backward: null argument");
  return launch_conv_small_cout_bwd(x, Cin, N, H, W, gn_scale, gn_shift, act, w, dy, Cout, da, dW, (hipStream_t)stream);
}

}  // extern "C"
