// c_api.hip — error state + op-level C-ABI entry points (include/adm.h). The UNet executor and the
// sampling loop export their own entry points from unet_exec.hip.
#include <map>
#include <mutex>

#include "adm_kernels.h"

namespace adm {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const char* last_error() { return g_err.c_str(); }
}  // namespace adm

using namespace adm;

// ADM_SEGV_BACKTRACE=1: a SIGSEGV / SIGABRT inside the process prints the NATIVE stack (backtrace_symbols_fd) before the default action — Python's
// faulthandler stops at the ctypes call. Diagnostic only; nothing is installed without the variable.
#if !defined(ADM_EMU)
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
namespace {
void adm_segv_handler(int sig) {
  void* frames[64];
  const int n = backtrace(frames, 64);
  const char msg[] = "\n[adm] fatal signal, native stack:\n";
  (void)!write(2, msg, sizeof(msg) - 1);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
struct AdmSegvInstall {
  AdmSegvInstall() {
    const char* e = getenv("ADM_SEGV_BACKTRACE");
    if (e && atoi(e)) { signal(SIGSEGV, adm_segv_handler); signal(SIGABRT, adm_segv_handler); signal(SIGBUS, adm_segv_handler); }
  }
} g_adm_segv_install;
}  // namespace
#endif

extern "C" {

int adm_version(void) { return 104; }   // 104 (round 6): adm_conv_args.single_sample, option "single_sample"; 103 (round 6): adm_conv_args.wino6_rule, adm_unet_set_option, adm_release_stream
//   // 102 (round 5): Winograd buffers hold two images (adm_winograd_packed_floats)
//   // 101 (round 4): adm_slerp_grid takes double weights (round 3), blocked-image entry points
const char* adm_last_error(void) { return adm::last_error(); }
int adm_set_option(const char* name, int value) {
  ADM_REQUIRE(name, "set_option: null name");
  const std::string nm(name);
  static const char* known[] = {"conv_wino", "wino_pair", "wino5", "wino6", "single_sample", "wgrad_max_split", "conv_bf16", "conv_op16_f16", "blk_direct_dy", "gn_fuse_finish"};
  bool ok = false;
  for (const char* k : known) ok |= nm == k;
  if (!ok) ADM_FAIL(std::string("set_option: unknown option ") + name);
  // the dispatch epoch (training nets re-learn which weight images they read: a full re-pack) moves only when a value really
  // changes — a model that sets the options it already runs under (every enable_training does) leaves other nets alone
  if (nm == "conv_wino")   // validate BEFORE the value is recorded: a rejected value must neither move the epoch nor be remembered
    ADM_REQUIRE(adm::winograd_mode_available(value), "set_option: conv_wino takes -1 (environment), 0 (direct MFMA kernel only) or 4 (Winograd kernels: "
                "the default); modes 1-3 were earlier kernel generations, retired in round 6");
  if (nm == "wino6")
    ADM_REQUIRE(value == -1 || value == 0 || value == 1 || value == 2 || (value >= 16 && value <= 65536),
                "set_option: wino6 takes -1 (environment), 0 (off), 1 (default layer rule), 2 (every layer the kernel tiles) or a plane-size floor n >= 16");
  if (nm == "single_sample") ADM_REQUIRE(value >= -1 && value <= 1, "set_option: single_sample takes -1 (environment), 0 (off) or 1 (on)");
  static std::mutex mu;
  static std::map<std::string, int> last;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = last.find(nm);
    const bool same = it != last.end() && it->second == value;
    last[nm] = value;
    if (!same) adm::bump_dispatch_epoch();
  }
  if (nm == "conv_wino") { adm::set_winograd_mode(value); return 0; }
  if (nm == "wino5") { adm::set_winograd_v5(value); return 0; }
  if (nm == "wino6") { adm::set_winograd_v6(value); return 0; }
  if (nm == "single_sample") { adm::set_single_sample(value); return 0; }
  if (nm == "wino_pair") { adm::set_winograd_pair(value); return 0; }
  if (nm == "wgrad_max_split") { adm::set_wgrad_max_split(value); return 0; }
  if (nm == "conv_bf16") { adm::set_conv_bf16(value); return 0; }
  if (nm == "conv_op16_f16") { adm::set_conv_op16_f16(value); return 0; }
  if (nm == "gn_fuse_finish") { adm::set_gn_fuse_finish(value); return 0; }
  adm::set_blk_direct_dy(value);
  return 0;
}
int adm_last_conv_variant(void) { return adm::last_conv_variant(); }
int adm_release_stream(void* stream) { adm::conv_ksplit_release((hipStream_t)stream); return 0; }
int adm_has_experiments(void) { return 0; }   // (kept for ABI stability: the experiments builds were retired in round 6)
int adm_is_device_build(void) {
#if defined(ADM_EMU)
  return 0;
#else
  return 1;
#endif
}

int adm_sched_step(const float* x, const float* eps, const float* noise, float* out, uint8_t* u8_out,
                   const adm_sched_coef* coef_table, const int* step_dev, int step, const float* mask,
                   int n_mask_steps, int mask_start, int mask_end, int B, int C, int H, int W, void* stream) {
  ADM_REQUIRE(x && eps && out && coef_table, "sched_step: null argument");
  return launch_sched_step(x, eps, noise, out, u8_out, coef_table, step_dev, step, mask, n_mask_steps, mask_start,
                           mask_end, B, C, H, W, (hipStream_t)stream);
}

int adm_add_noise(const float* x0, long x0_bstride, const float* noise, const float* sa, const float* sb, int cb,
                  int cn, float* out, int B, int N, long P, void* stream) {
  ADM_REQUIRE(x0 && noise && sa && sb && out, "add_noise: null argument");
  return launch_add_noise(x0, x0_bstride, noise, sa, sb, cb, cn, out, B, N, P, (hipStream_t)stream);
}

int adm_dequant_u8(const float* x, uint8_t* out, long n, void* stream) {
  ADM_REQUIRE(x && out, "dequant: null argument");
  return launch_dequant(x, out, n, (hipStream_t)stream);
}

int adm_slerp_grid(const float* x0, const float* x1, long n, const double* alphas_dev, int n_alpha, float* out,
                   double* scratch3, void* stream) {
  ADM_REQUIRE(x0 && x1 && alphas_dev && out && scratch3 && n > 0 && n_alpha > 0, "slerp_grid: bad argument");
  return launch_slerp_grid(x0, x1, n, alphas_dev, n_alpha, out, scratch3, (hipStream_t)stream);
}

int adm_groupnorm_stats(const float* x1, int C1, const float* x2, int C2, int N, int HW, int groups, float eps,
                        const float* gamma, const float* beta, float* scale, float* shift, void* stream) {
  ADM_REQUIRE(x1 && gamma && beta && scale && shift, "groupnorm_stats: null argument");
  return launch_groupnorm_stats(x1, C1, x2, C2, N, HW, groups, eps, gamma, beta, scale, shift, (hipStream_t)stream);
}

int adm_conv_stats_tiles(const adm_conv_args* a) { return a ? conv_stats_tiles(*a) : 0; }

int adm_groupnorm_finalize(const double* stats1, int C1, int tiles1, const double* stats2, int C2, int tiles2, int N, int HW,
                           int groups, float eps, const float* gamma, const float* beta, float* scale, float* shift,
                           void* stream) {
  ADM_REQUIRE(stats1 && gamma && beta && scale && shift, "groupnorm_finalize: null argument");
  return launch_groupnorm_finalize(stats1, C1, tiles1, stats2, C2, tiles2, N, HW, groups, eps, gamma, beta, scale, shift,
                                   (hipStream_t)stream);
}

int adm_conv2d(const adm_conv_args* a, void* stream) {
  ADM_REQUIRE(a && a->x1 && a->wpacked && a->out, "conv2d: null argument");
  return launch_conv2d(*a, (hipStream_t)stream);
}

int adm_pack_conv_weight(const float* w, float* wpacked, int Cout, int Cin, int ks, void* stream) {
  ADM_REQUIRE(w && wpacked, "pack_conv_weight: null argument");
  return launch_pack_conv_weight(w, wpacked, Cout, Cin, ks, (hipStream_t)stream);
}

int adm_pack_conv_weight_T(const float* w, float* wpT, int Cout, int Cin, int ks, void* stream) {
  ADM_REQUIRE(w && wpT, "pack_conv_weight_T: null argument");
  return launch_pack_conv_weight_T(w, wpT, Cout, Cin, ks, (hipStream_t)stream);
}

long adm_winograd_packed_floats(int Cout, int Cin, int transposed) { return winograd_packed_floats(Cout, Cin, transposed); }
int adm_pack_winograd_weight(const float* w, float* wu, int Cout, int Cin, void* stream) {
  ADM_REQUIRE(w && wu, "pack_winograd_weight: null argument");
  return launch_pack_winograd_weight(w, wu, Cout, Cin, (hipStream_t)stream);
}

int adm_pack_winograd_weight_T(const float* w, float* wuT, int Cout, int Cin, void* stream) {
  ADM_REQUIRE(w && wuT, "pack_winograd_weight_T: null argument");
  return launch_pack_winograd_weight_T(w, wuT, Cout, Cin, (hipStream_t)stream);
}

int adm_pack_bf16_weight(const float* w, void* wb, int Cout, int Cin, int transposed, void* stream) {
  ADM_REQUIRE(w && wb, "pack_bf16_weight: null argument");
  return launch_pack_bf16_weight(w, wb, Cout, Cin, transposed, (hipStream_t)stream);
}

int adm_pack_bf16_weight_ks(const float* w, void* wb, int Cout, int Cin, int ks, int transposed, void* stream) {
  ADM_REQUIRE(w && wb, "pack_bf16_weight: null argument");
  return launch_pack_bf16_weight(w, wb, Cout, Cin, transposed, (hipStream_t)stream, ks);
}

void adm_conv_out_dims(int H, int W, int up, int stride, int ks, int pad_lo, int* Ho, int* Wo) {
  conv_out_dims(H, W, up, stride, ks, pad_lo, Ho, Wo);
}

int adm_attention(const float* qkv, float* out, int N, int C, int T, int head_dim, void* stream) {
  ADM_REQUIRE(qkv && out, "attention: null argument");
  return launch_attention(qkv, out, N, C, T, head_dim, (hipStream_t)stream);
}

int adm_layernorm_nct(const float* x, const float* gamma, const float* beta, float* y, int N, int C, long T, float eps,
                      void* stream) {
  ADM_REQUIRE(x && gamma && beta && y, "layernorm_nct: null argument");
  return launch_layernorm_nct(x, gamma, beta, y, N, C, T, eps, (hipStream_t)stream);
}
int adm_geglu(const float* in, float* out, int N, int C4, long T, void* stream) {
  ADM_REQUIRE(in && out, "geglu: null argument");
  return launch_geglu(in, out, N, C4, T, (hipStream_t)stream);
}
int adm_cross_attention(const float* q, const float* ctx, const float* Wk, const float* Wv, float* out, int N, int C, int T,
                        int S, int Dc, int head_dim, void* stream) {
  ADM_REQUIRE(q && ctx && Wk && Wv && out, "cross_attention: null argument");
  return launch_cross_attention(q, ctx, Wk, Wv, out, N, C, T, S, Dc, head_dim, (hipStream_t)stream);
}
int adm_attention_mfma_eligible(int C, int T, int head_dim) { return attention_mfma_eligible(C, T, head_dim) ? 1 : 0; }
int adm_attention_blocked(const float* qkv, float* out, int N, int C, int T, int head_dim, int key_block, void* stream) {
  ADM_REQUIRE(qkv && out, "attention_blocked: null argument");
  return launch_attention_blocked(qkv, out, N, C, T, head_dim, key_block, (hipStream_t)stream);
}

int adm_layernorm_nct_backward(const float* x, const float* dy, const float* gamma, float* dx, int accumulate, float* stats,
                               float* dgamma, float* dbeta, int N, int C, long T, float eps, void* stream) {
  ADM_REQUIRE(x && dy && gamma && dx && stats && dgamma && dbeta, "layernorm_nct_backward: null argument");
  return launch_layernorm_nct_bwd(x, dy, gamma, dx, accumulate, stats, dgamma, dbeta, N, C, T, eps, (hipStream_t)stream);
}
int adm_geglu_backward(const float* in, const float* dy, float* din, int N, int C4, long T, void* stream) {
  ADM_REQUIRE(in && dy && din, "geglu_backward: null argument");
  return launch_geglu_bwd(in, dy, din, N, C4, T, (hipStream_t)stream);
}
int adm_cross_attention_backward(const float* q, const float* ctx, const float* Wk, const float* Wv, const float* dy,
                                 float* dq, float* dWk, float* dWv, int N, int C, int T, int S, int Dc, int head_dim,
                                 void* stream) {
  ADM_REQUIRE(q && ctx && Wk && Wv && dy && dq && dWk && dWv, "cross_attention_backward: null argument");
  return launch_cross_attention_bwd(q, ctx, Wk, Wv, dy, dq, dWk, dWv, N, C, T, S, Dc, head_dim, (hipStream_t)stream);
}
int adm_attention_backward_blocked(const float* qkv, const float* dout, float* dqkv, float* stats, int N, int C, int T,
                                   int head_dim, int block, void* stream) {
  ADM_REQUIRE(qkv && dout && dqkv && stats, "attention_backward_blocked: null argument");
  return launch_attention_bwd_blocked(qkv, dout, dqkv, stats, N, C, T, head_dim, block, (hipStream_t)stream);
}
int adm_sepconv_block(const float* x, const float* dw, const float* pw, const float* pb, const float* bn_scale,
                      const float* bn_shift, float slope, float* tmp, float* y, int N, int Ci, int Co, int H, int W,
                      void* stream) {
  ADM_REQUIRE(x && dw && pw && pb && bn_scale && bn_shift && tmp && y, "sepconv_block: null argument");
  return launch_sepconv_block(x, dw, pw, pb, bn_scale, bn_shift, slope, tmp, y, N, Ci, Co, H, W, (hipStream_t)stream);
}
int adm_dense_act(const float* x, const float* W, const float* b, const float* post_scale, const float* post_shift,
                  float slope, int leaky, float* y, int N, int K, int J, int hwc_C, void* stream) {
  ADM_REQUIRE(x && W && b && y && ((post_scale == nullptr) == (post_shift == nullptr)), "dense_act: bad argument");
  return launch_dense_act(x, W, b, post_scale, post_shift, slope, leaky, y, N, K, J, hwc_C, (hipStream_t)stream);
}

}  // extern "C"
