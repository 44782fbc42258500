// net_exec.hip — implementation of the generic flat-op-list executor (see net_exec.h).
#include <cstdlib>

#include <atomic>
#include <tuple>
#include <utility>

#include "net_exec.h"

#include <cmath>
#include <cstring>

namespace adm {

// ---------------------------------------------------------------------------------------------- ParamStore
void ParamStore::declare(const std::string& key, std::vector<long> shape) {
  ParamSlot s;
  s.shape = shape;
  s.numel = 1;
  for (long d : shape) s.numel *= (size_t)d;
  params[key] = s;
}
void ParamStore::declare_conv(const std::string& p, int co, int ci, int ks) {
  declare(p + ".weight", {co, ci, ks, ks});
  declare(p + ".bias", {co});
}
void ParamStore::declare_lin(const std::string& p, int co, int ci) {
  declare(p + ".weight", {co, ci});
  declare(p + ".bias", {co});
}
void ParamStore::declare_gn(const std::string& p, int c) {
  declare(p + ".weight", {c});
  declare(p + ".bias", {c});
}
void ParamStore::declare_resnet(const std::string& p, int ci, int co, int temb) {
  declare_gn(p + ".norm1", ci);
  declare_conv(p + ".conv1", co, ci, 3);
  if (temb > 0) declare_lin(p + ".time_emb_proj", co, temb);
  declare_gn(p + ".norm2", co);
  declare_conv(p + ".conv2", co, co, 3);
  if (ci != co) declare_conv(p + ".conv_shortcut", co, ci, 1);
}
void ParamStore::declare_transformer(const std::string& p, int c, int cross_dim) {
  declare_gn(p + ".norm", c);
  declare_conv(p + ".proj_in", c, c, 1);
  const std::string tb = p + ".transformer_blocks.0";
  declare_gn(tb + ".norm1", c);
  for (const char* n : {".attn1.to_q", ".attn1.to_k", ".attn1.to_v", ".attn2.to_q"}) declare(tb + n + ".weight", {c, c});
  declare_lin(tb + ".attn1.to_out.0", c, c);
  declare_gn(tb + ".norm2", c);
  declare(tb + ".attn2.to_k.weight", {c, cross_dim});
  declare(tb + ".attn2.to_v.weight", {c, cross_dim});
  declare_lin(tb + ".attn2.to_out.0", c, c);
  declare_gn(tb + ".norm3", c);
  declare_lin(tb + ".ff.net.0.proj", 8 * c, c);
  declare_lin(tb + ".ff.net.2", c, 4 * c);
  declare_conv(p + ".proj_out", c, c, 1);
}
void ParamStore::declare_attn(const std::string& p, int c) {
  declare_gn(p + ".group_norm", c);
  declare_lin(p + ".to_q", c, c);
  declare_lin(p + ".to_k", c, c);
  declare_lin(p + ".to_v", c, c);
  declare_lin(p + ".to_out.0", c, c);
}
int ParamStore::set(const char* key, const float* host_data, size_t numel) {
  std::string k(key);
  static const char* oldn[4] = {".query.", ".key.", ".value.", ".proj_attn."};
  static const char* newn[4] = {".to_q.", ".to_k.", ".to_v.", ".to_out.0."};
  if (k.find(".attentions.") != std::string::npos)
    for (int i = 0; i < 4; ++i) {
      size_t pos = k.find(oldn[i]);
      if (pos != std::string::npos) k.replace(pos, strlen(oldn[i]), newn[i]);
    }
  auto it = params.find(k);
  ADM_REQUIRE(it != params.end(), "set_param: unexpected key " + k);
  ADM_REQUIRE(it->second.numel == numel, "set_param: size mismatch for " + k);
  if (!it->second.dev) ADM_TRY(dmalloc((void**)&it->second.dev, sizeof(float) * numel));
  ADM_TRY(copy_h2d(it->second.dev, host_data, sizeof(float) * numel, nullptr));
  ADM_TRY(stream_sync(nullptr));
  it->second.set = true;
  return 0;
}
int ParamStore::bind(const char* key, float* dev_ptr) {
  auto it = params.find(key);
  ADM_REQUIRE(it != params.end(), std::string("bind_param: unexpected key ") + key);
  if (it->second.dev && !it->second.external) dfree(it->second.dev);
  it->second.dev = dev_ptr;
  it->second.external = true;
  it->second.set = true;
  return 0;
}
int ParamStore::missing(std::string* names) const {
  int n = 0;
  for (auto& kv : params)
    if (!kv.second.set) { ++n; if (names && n <= 16) *names += kv.first + " "; }
  return n;
}
void ParamStore::free_all() {
  for (auto& kv : params)
    if (kv.second.dev && !kv.second.external) { dfree(kv.second.dev); kv.second.dev = nullptr; }
}

// ---------------------------------------------------------------------------------------------- OpTimer
void OpTimer::begin() {
#if !defined(ADM_EMU)
  if (!recs) return;
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  (void)hipEventRecord(a, st);
  evs.push_back({a, b});
#endif
}
void OpTimer::end(int kind, int variant, double flops, double bytes) {
  if (!recs) return;
#if !defined(ADM_EMU)
  (void)hipEventRecord(evs.back().second, st);
#endif
  adm_op_profile r; r.kind = kind; r.variant = variant; r.ms = 0.f; r.flops = flops; r.bytes = bytes;
  recs->push_back(r);
}
void OpTimer::finish() {
#if !defined(ADM_EMU)
  if (!recs) return;
  (void)hipStreamSynchronize(st);
  for (size_t i = 0; i < evs.size(); ++i) {
    (void)hipEventElapsedTime(&(*recs)[i].ms, evs[i].first, evs[i].second);
    (void)hipEventDestroy(evs[i].first); (void)hipEventDestroy(evs[i].second);
  }
  evs.clear();
#endif
}

// ---------------------------------------------------------------------------------------------- Net: build
int Net::dalloc(void** p, size_t bytes) {
  ADM_TRY(dmalloc(p, bytes));
  owned.push_back(*p);
  return 0;
}
static int pack_one(Net* net, ConvW& w, hipStream_t st) {
  const float* src = w.stacked;
  if (!w.qkv_prefix.empty()) {
    const int C = w.Cin;
    const char* names[3] = {".to_q", ".to_k", ".to_v"};
    for (int i = 0; i < 3; ++i) {
      ADM_TRY(copy_d2d(w.stacked + (size_t)i * C * C, net->ps->P(w.qkv_prefix + names[i] + ".weight"), sizeof(float) * (size_t)C * C, st));
      if (w.has_bias) ADM_TRY(copy_d2d(w.bias + (size_t)i * C, net->ps->P(w.qkv_prefix + names[i] + ".bias"), sizeof(float) * (size_t)C, st));
    }
  } else {
    src = net->ps->P(w.key + ".weight");
    w.bias = w.has_bias ? net->ps->P(w.key + ".bias") : nullptr;
  }
  // every packing is produced when the layer is first built (allocation + contents); a REFRESH after an optimizer step skips
  // the packings the dispatch has not read during the last complete step (e.g. the fp32 and Winograd images of a layer whose
  // three passes all run on bf16 operands): `need` is the mask of what to write now
  const unsigned need = (net->training && net->use_known) ? w.used : ~0u;
  if (need & PK_WP) ADM_TRY(launch_pack_conv_weight(src, w.wp, w.Cout, w.Cin, w.ks, st));
  if (w.ks == 3 && w.qkv_prefix.empty() && winograd_enabled() && w.Cout % 32 == 0 && w.Cin % 8 == 0) {
    if (!w.wu) ADM_TRY(net->dalloc((void**)&w.wu, sizeof(float) * (size_t)winograd_packed_floats(w.Cout, w.Cin, 0)));
    if (need & PK_WU) ADM_TRY(launch_pack_winograd_weight(src, w.wu, w.Cout, w.Cin, st));
  }
  if (net->training) {
    if (!w.wpT) ADM_TRY(net->dalloc((void**)&w.wpT, sizeof(float) * (size_t)w.Cout * w.Cin * w.ks * w.ks));
    if (need & PK_WPT) ADM_TRY(launch_pack_conv_weight_T(src, w.wpT, w.Cout, w.Cin, w.ks, st));
    if (w.ks == 3 && w.qkv_prefix.empty() && winograd_enabled() && w.Cin % 32 == 0 && w.Cout % 8 == 0) {
      if (!w.wuT) ADM_TRY(net->dalloc((void**)&w.wuT, sizeof(float) * (size_t)winograd_packed_floats(w.Cout, w.Cin, 1)));
      if (need & PK_WUT) ADM_TRY(launch_pack_winograd_weight_T(src, w.wuT, w.Cout, w.Cin, st));
    }
    // mixed precision (`--mixed_precision bf16`): the filters as bf16 MFMA operands, re-rounded from the fp32 masters after
    // every optimizer step; only training nets carry them, so sampling stays fp32 whatever the option says
    const bool bf3 = w.ks == 3 && w.qkv_prefix.empty() && conv_bf16_enabled();
    const bool bf1 = w.ks == 1 && conv_bf16_mode() >= 2;          // level 2: shortcuts and attention projections too
    if ((bf3 || bf1) && w.Cin % 16 == 0 && w.Cout % 16 == 0) {
      const size_t bytes = 2 * (size_t)w.Cout * w.Cin * w.ks * w.ks;
      if (!w.wb) ADM_TRY(net->dalloc(&w.wb, bytes));
      if (!w.wbT) ADM_TRY(net->dalloc(&w.wbT, bytes));
      if (need & PK_WB) ADM_TRY(launch_pack_bf16_weight(src, w.wb, w.Cout, w.Cin, 0, st, w.ks));
      if (need & PK_WBT) ADM_TRY(launch_pack_bf16_weight(src, w.wbT, w.Cout, w.Cin, 1, st, w.ks));
    }
  }
  return 0;
}

int Net::make_conv(const std::string& p, int co, int ci, int ks, const ConvW** out, bool bias) {
  ConvW w;
  w.Cin = ci; w.Cout = co; w.ks = ks; w.key = p; w.has_bias = bias;
  ADM_TRY(dalloc((void**)&w.wp, sizeof(float) * (size_t)co * ci * ks * ks));
  convs.push_back(w);
  ADM_TRY(pack_one(this, convs.back(), nullptr));
  *out = &convs.back();
  return 0;
}

int Net::set_bucket_hook(int n_buckets, const long* bounds, void (*fn)(void*, int), void* user) {
  bk_fn = nullptr; bk_lo.clear(); bk_total.clear(); bk_pending.clear();
  if (n_buckets <= 0 || fn == nullptr) return 0;
  ADM_REQUIRE(params_base != nullptr && bounds != nullptr, "set_bucket_hook: parameters are not bound to a flat buffer");
  bk_lo.assign(bounds, bounds + n_buckets + 1);
  for (int b = 0; b < n_buckets; ++b) ADM_REQUIRE(bk_lo[b] < bk_lo[b + 1], "set_bucket_hook: bounds must ascend");
  bk_total.assign(n_buckets, 0);
  for (auto& kv : ps->params) {
    const long lo = kv.second.dev - params_base, hi = lo + (long)kv.second.numel;
    for (int b = 0; b < n_buckets; ++b)
      if (lo < bk_lo[b + 1] && hi > bk_lo[b]) ++bk_total[b];
  }
  bk_fn = fn; bk_user = user;
  bucket_reset();
  return 0;
}

void Net::mark_ready(const float* master_param, size_t numel) {
  if (bk_fn == nullptr) return;
  const long lo = master_param - params_base, hi = lo + (long)numel;
  for (size_t b = 0; b + 1 < bk_lo.size(); ++b)
    if (lo < bk_lo[b + 1] && hi > bk_lo[b] && --bk_pending[b] == 0) bk_fn(bk_user, (int)b);
}

static int g_blk_direct_dy = -1;    // -1: ADM_BLK_DIRECT_DY from the environment (default 1); read at plan time
void set_blk_direct_dy(int v) { g_blk_direct_dy = v; }
static bool blk_direct_dy() {
  if (g_blk_direct_dy < 0) { const char* e = getenv("ADM_BLK_DIRECT_DY"); g_blk_direct_dy = e ? atoi(e) : 1; }
  return g_blk_direct_dy != 0;
}
static std::atomic<unsigned> g_dispatch_epoch{1};
void bump_dispatch_epoch() { g_dispatch_epoch.fetch_add(1); }
unsigned dispatch_epoch() { return g_dispatch_epoch.load(); }

int Net::note_packing(const ConvW& w, unsigned pk) {
  if (training && use_known && stale_packings && (pk & ~w.used) != 0)
    ADM_FAIL("conv dispatch of layer " + w.key + " changed after the weight packings were learned (packing mask " +
             std::to_string(pk) + " not in " + std::to_string(w.used) + "): the kernel read weights from before the last "
             "optimizer steps; call adm_unet_refresh_weights after changing options / input alignment");
  if (pk & ~w.used) pk_valid = false;       // a new packing joins the mask: the batched re-pack tables are rebuilt
  w.used |= pk;
  return 0;
}

int Net::begin_training_batch(int B, hipStream_t st) {
  for (int b : learned_B) if (b == B) return 0;
  if (stale_packings) {
    const bool known = use_known;
    use_known = false;
    for (ConvW& w : convs) ADM_TRY(pack_one(this, w, st));
    use_known = known;
    stale_packings = false;
  }
  return 0;
}

// The packings a learned mask asks for, as device tables for the batched kernels: the same decisions as pack_one (a packing
// exists iff pack_one allocated it; it is written iff the mask has its bit), made once per mask change instead of per step.
int Net::build_pack_tables(hipStream_t st) {
  for (auto& v : pk_host) v.clear();
  for (ConvW& w : convs) {
    const float* src = w.stacked;
    if (!w.qkv_prefix.empty()) {
      const int C = w.Cin;
      const char* names[3] = {".to_q", ".to_k", ".to_v"};
      for (int i = 0; i < 3; ++i) {
        pk_host[0].push_back(PackItem{ps->P(w.qkv_prefix + names[i] + ".weight"), w.stacked + (size_t)i * C * C, 0, 0, 0, C * C});
        if (w.has_bias) pk_host[0].push_back(PackItem{ps->P(w.qkv_prefix + names[i] + ".bias"), w.bias + (size_t)i * C, 0, 0, 0, C});
      }
    } else {
      src = ps->P(w.key + ".weight");
    }
    const unsigned need = w.used;
    if (need & PK_WP) pk_host[1].push_back(PackItem{src, w.wp, w.Cout, w.Cin, w.ks, 0});
    if (w.wpT && (need & PK_WPT)) pk_host[1].push_back(PackItem{src, w.wpT, w.Cout, w.Cin, w.ks, 1});
    if (w.wu && (need & PK_WU)) pk_host[2].push_back(PackItem{src, w.wu, w.Cout, w.Cin, 3, winograd_pack_flag(w.Cout, w.Cin, 0)});
    if (w.wuT && (need & PK_WUT)) pk_host[2].push_back(PackItem{src, w.wuT, w.Cout, w.Cin, 3, winograd_pack_flag(w.Cout, w.Cin, 1)});
    if (w.wb && (need & PK_WB)) pk_host[3].push_back(PackItem{src, w.wb, w.Cout, w.Cin, w.ks, 0});
    if (w.wbT && (need & PK_WBT)) pk_host[3].push_back(PackItem{src, w.wbT, w.Cout, w.Cin, w.ks, 1});
  }
  size_t n = 0;
  for (auto& v : pk_host) n += v.size();
  if (n > pk_dev_cap) {
    ADM_TRY(dalloc((void**)&pk_dev, sizeof(PackItem) * (n + 64)));     // (a superseded smaller table stays owned until destroy)
    pk_dev_cap = n + 64;
  }
  size_t off = 0;
  for (auto& v : pk_host) {
    if (!v.empty()) ADM_TRY(copy_h2d(pk_dev + off, v.data(), sizeof(PackItem) * v.size(), st));
    off += v.size();
  }
  ADM_TRY(stream_sync(st));     // the host vectors may be rebuilt before the copies would otherwise have run
  pk_valid = true;
  return 0;
}

int Net::refresh_weights(hipStream_t st) {
  if (use_known && known_epoch != dispatch_epoch()) {   // adm_set_option since the masks were learned: re-learn them
    use_known = false;
    learned_B.clear();
    pk_valid = false;
    for (ConvW& w : convs) w.used = 0;
  }
  static const int batched = [] { const char* e = getenv("ADM_PACK_BATCH"); return e ? atoi(e) : 1; }();
  if (training && use_known && batched) {
    // after an optimizer step, masks learned: ~6 launches over device tables instead of one tiny launch per (layer, packing)
    if (!pk_valid) ADM_TRY(build_pack_tables(st));
    const PackItem* t = pk_dev;
    ADM_TRY(launch_copy_batch(t, (int)pk_host[0].size(), st)); t += pk_host[0].size();
    ADM_TRY(launch_pack_conv_weight_batch(t, (int)pk_host[1].size(), st)); t += pk_host[1].size();
    ADM_TRY(launch_pack_winograd_batch(t, (int)pk_host[2].size(), st)); t += pk_host[2].size();
    ADM_TRY(launch_pack_bf16_batch(t, (int)pk_host[3].size(), st));
    stale_packings = true;
    return 0;
  }
  for (ConvW& w : convs) ADM_TRY(pack_one(this, w, st));
  stale_packings = training && use_known;               // the packings no training pass reads were left as they were
  return 0;
}
// An inference entry point on a TRAINING net (evaluation samples through the live model): bring every packing up to date first,
// and keep what the inference dispatch reads out of the training masks.
int Net::begin_inference(hipStream_t st, std::vector<unsigned>* saved) {
  saved->clear();
  if (!training) return 0;
  if (stale_packings) {
    const bool known = use_known;
    use_known = false;
    for (ConvW& w : convs) ADM_TRY(pack_one(this, w, st));
    use_known = known;
    stale_packings = false;
  }
  for (const ConvW& w : convs) saved->push_back(w.used);
  return 0;
}
void Net::end_inference(const std::vector<unsigned>& saved) {
  if (!training) return;
  size_t i = 0;
  for (ConvW& w : convs) {
    if (w.used != saved[i]) pk_valid = false;
    w.used = saved[i++];
  }
}
const GNW* Net::make_gn(const std::string& p, int c) {
  GNW g;
  g.gamma = ps->P(p + ".weight"); g.beta = ps->P(p + ".bias"); g.C = c;
  gns.push_back(g);
  return &gns.back();
}
int Net::new_tensor(int C, int H, int W, bool ext) {
  Tensor t; t.C = C; t.H = H; t.W = W; t.external = ext;
  tensors.push_back(t);
  return (int)tensors.size() - 1;
}
int Net::gn_op(int in1, int in2, const GNW* g) {
  GnBuf b; b.C = g->C; b.g = g;
  gnbufs.push_back(b);
  Op o; o.kind = Op::GN; o.in1 = in1; o.in2 = in2; o.g = g; o.gn = (int)gnbufs.size() - 1;
  ops.push_back(o);
  return o.gn;
}
int Net::conv_op(int in1, int in2, const ConvW* w, int gn, int act, int up, int stride, int pad_lo, int res,
                 int temb_off, int out_ext) {
  const Tensor& ti = tensors[in1];
  int Ho, Wo;
  conv_out_dims(ti.H, ti.W, up, stride, w->ks, pad_lo, &Ho, &Wo);
  Op o; o.kind = Op::CONV; o.in1 = in1; o.in2 = in2; o.w = w; o.gn = gn; o.act = act; o.up = up; o.stride = stride;
  o.ks = w->ks; o.pad_lo = w->ks == 3 ? pad_lo : 0; o.res = res; o.temb_off = temb_off;
  o.out = out_ext >= 0 ? out_ext : new_tensor(w->Cout, Ho, Wo);
  ops.push_back(o);
  return o.out;
}
int Net::resnet(const std::string& p, int x1, int x2, int ci, int co, bool temb, int* rc) {
  const ConvW *c1, *c2, *sc = nullptr;
  if ((*rc = make_conv(p + ".conv1", co, ci, 3, &c1))) return -1;
  if ((*rc = make_conv(p + ".conv2", co, co, 3, &c2))) return -1;
  if (ci != co && (*rc = make_conv(p + ".conv_shortcut", co, ci, 1, &sc))) return -1;
  int temb_off = -1;
  if (temb) {
    temb_off = 0;
    for (auto& r : temb_rows) temb_off += r.second;
    temb_rows.push_back({p + ".time_emb_proj", co});
  }
  const int g1 = gn_op(x1, x2, make_gn(p + ".norm1", ci));
  const int hmid = conv_op(x1, x2, c1, g1, 1, 0, 1, 1, -1, temb_off);
  const int g2 = gn_op(hmid, -1, make_gn(p + ".norm2", co));
  int res = x1;
  if (sc) res = conv_op(x1, x2, sc, -1, 0, 0, 1, 0, -1, -1);
  return conv_op(hmid, -1, c2, g2, 1, 0, 1, 1, res, -1);
}
int Net::stacked_qkv(const std::string& prefix, int C, bool bias, const ConvW** out) {
  // q|k|v stacked into one 1x1 conv: weights (3C, C), bias 3C (all-zero for Linear(bias=False) projections)
  ConvW qkv; qkv.Cin = C; qkv.Cout = 3 * C; qkv.ks = 1; qkv.qkv_prefix = prefix; qkv.has_bias = bias;
  ADM_TRY(dalloc((void**)&qkv.stacked, sizeof(float) * (size_t)3 * C * C));
  ADM_TRY(dalloc((void**)&qkv.wp, sizeof(float) * (size_t)3 * C * C));
  ADM_TRY(dalloc((void**)&qkv.bias, sizeof(float) * (size_t)3 * C));
  if (!bias) { ADM_TRY(dmemset(qkv.bias, 0, sizeof(float) * (size_t)3 * C, nullptr)); ADM_TRY(stream_sync(nullptr)); }
  convs.push_back(qkv);
  ADM_TRY(pack_one(this, convs.back(), nullptr));
  *out = &convs.back();
  return 0;
}
int Net::attention(const std::string& p, int x, int C, int head_dim, int* rc) {
  // GroupNorm (no SiLU) folded into the load path of the stacked q|k|v 1x1 conv
  const ConvW* wqkv_;
  if ((*rc = stacked_qkv(p, C, true, &wqkv_))) return -1;
  const ConvW* wqkv = wqkv_;
  const ConvW* wo;
  if ((*rc = make_conv(p + ".to_out.0", C, C, 1, &wo))) return -1;
  const int g = gn_op(x, -1, make_gn(p + ".group_norm", C));
  const int t_qkv = conv_op(x, -1, wqkv, g, 0, 0, 1, 0, -1, -1);
  const Tensor tx = tensors[x];
  const int T = tx.H * tx.W;
  if (head_dim <= 64) {  // UNet form: many small heads -> fused per-head kernel
    Op o; o.kind = Op::ATTN; o.in1 = t_qkv; o.head_dim = head_dim;
    o.out = new_tensor(C, tx.H, tx.W);
    ops.push_back(o);
    return conv_op(o.out, -1, wo, -1, 0, 0, 1, 0, x, -1);
  }
  // VAE form: one head, d = C. Both products run on the MFMA 1x1 kernel with per-sample weights:
  //   S'[j][t] = sum_c K[c][j] Q[c][t]   (weights = k slice, whose [c][j] memory IS the packed [Cin][Cout] layout)
  //   softmax over j (channel axis) with scale d^-0.5
  //   O[c][t]  = sum_j V[c][j] P[j][t]   (weights = V^T, produced by a transpose op)
  if (head_dim != C) { set_error("attention: only 1 head (head_dim == channels) or head_dim <= 64 are implemented"); *rc = -1; return -1; }
  Op s; s.kind = Op::CONV; s.in1 = t_qkv; s.in1_coff = 0; s.in1_C = C; s.wt = t_qkv; s.wt_coff = C; s.dyn_cout = T;
  s.ks = 1; s.pad_lo = 0; s.out = new_tensor(T, tx.H, tx.W);
  ops.push_back(s);
  Op sm; sm.kind = Op::SOFTMAXC; sm.in1 = s.out; sm.out = s.out; sm.scale = 1.0f / sqrtf((float)head_dim);
  ops.push_back(sm);
  Op tr; tr.kind = Op::TRANSP; tr.in1 = t_qkv; tr.in1_coff = 2 * C; tr.in1_C = C; tr.out = new_tensor(T, C, 1);
  ops.push_back(tr);
  Op pv; pv.kind = Op::CONV; pv.in1 = s.out; pv.wt = tr.out; pv.wt_coff = 0; pv.dyn_cout = C; pv.ks = 1; pv.pad_lo = 0;
  pv.out = new_tensor(C, tx.H, tx.W);
  ops.push_back(pv);
  return conv_op(pv.out, -1, wo, -1, 0, 0, 1, 0, x, -1);
}
int Net::transformer(const std::string& p, int x, int C, int heads, int cross_dim, int* rc) {
  ADM_REQUIRE(heads > 0 && C % heads == 0, "transformer: channels not divisible by the head count");
  (void)cross_dim;
  const int hd = C / heads;
  const std::string tb = p + ".transformer_blocks.0";
  const Tensor tx = tensors[x];
  const ConvW *w_in, *w_qkv, *w_o1, *w_q2, *w_o2, *w_ff1, *w_ff2, *w_out;
  if ((*rc = make_conv(p + ".proj_in", C, C, 1, &w_in))) return -1;
  if ((*rc = stacked_qkv(tb + ".attn1", C, false, &w_qkv))) return -1;
  if ((*rc = make_conv(tb + ".attn1.to_out.0", C, C, 1, &w_o1))) return -1;
  if ((*rc = make_conv(tb + ".attn2.to_q", C, C, 1, &w_q2, false))) return -1;
  if ((*rc = make_conv(tb + ".attn2.to_out.0", C, C, 1, &w_o2))) return -1;
  if ((*rc = make_conv(tb + ".ff.net.0.proj", 8 * C, C, 1, &w_ff1))) return -1;
  if ((*rc = make_conv(tb + ".ff.net.2", C, 4 * C, 1, &w_ff2))) return -1;
  if ((*rc = make_conv(p + ".proj_out", C, C, 1, &w_out))) return -1;
  auto ln = [&](int in, const std::string& name) {
    Op o; o.kind = Op::LN; o.in1 = in; o.g = make_gn(name, C); o.eps = 1e-5f;
    o.out = new_tensor(C, tx.H, tx.W);
    ops.push_back(o);
    return o.out;
  };
  // hidden = proj_in(GroupNorm(x)), GroupNorm eps = 1e-6 (transformer_2d.py), folded into the conv's load path
  const int g = gn_op(x, -1, make_gn(p + ".norm", C));
  ops.back().eps = 1e-6f;
  const int h0 = conv_op(x, -1, w_in, g, 0, 0, 1, 0, -1, -1);
  // self-attention: h1 = to_out(attn(qkv(LN1(h0)))) + h0
  const int qkv = conv_op(ln(h0, tb + ".norm1"), -1, w_qkv, -1, 0, 0, 1, 0, -1, -1);
  Op a; a.kind = Op::ATTN; a.in1 = qkv; a.head_dim = hd; a.out = new_tensor(C, tx.H, tx.W);
  ops.push_back(a);
  const int h1 = conv_op(a.out, -1, w_o1, -1, 0, 0, 1, 0, h0, -1);
  // cross-attention on the encoding: h2 = to_out(xattn(to_q(LN2(h1)), ctx)) + h1
  const int q2 = conv_op(ln(h1, tb + ".norm2"), -1, w_q2, -1, 0, 0, 1, 0, -1, -1);
  Op xa; xa.kind = Op::XATTN; xa.in1 = q2; xa.head_dim = hd; xa.out = new_tensor(C, tx.H, tx.W);
  xa.wk = ps->P(tb + ".attn2.to_k.weight"); xa.wv = ps->P(tb + ".attn2.to_v.weight");
  ops.push_back(xa);
  const int h2 = conv_op(xa.out, -1, w_o2, -1, 0, 0, 1, 0, h1, -1);
  // feed-forward: h3 = W2(GEGLU(W1(LN3(h2)))) + h2
  const int ff = conv_op(ln(h2, tb + ".norm3"), -1, w_ff1, -1, 0, 0, 1, 0, -1, -1);
  Op ge; ge.kind = Op::GEGLU; ge.in1 = ff; ge.out = new_tensor(4 * C, tx.H, tx.W);
  ops.push_back(ge);
  const int h3 = conv_op(ge.out, -1, w_ff2, -1, 0, 0, 1, 0, h2, -1);
  return conv_op(h3, -1, w_out, -1, 0, 0, 1, 0, x, -1);
}
void Net::finish_liveness() {
  for (size_t i = 0; i < ops.size(); ++i) {
    const Op& o = ops[i];
    for (int t : {o.in1, o.in2, o.res, o.wt})
      if (t >= 0) tensors[t].last_use = (int)i;
    if (o.kind == Op::SOFTMAXC) tensors[o.out].last_use = (int)i;  // in place
  }
}

// ---------------------------------------------------------------------------------------------- Net: run
int Net::arena_alloc(void** p, size_t bytes) {
  ADM_TRY(dmalloc(p, bytes));
  arena.push_back(*p);
  arena_bytes += bytes;
  return 0;
}
void Net::free_plan() {
  for (void* p : arena) dfree(p);
  arena.clear();
  arena_bytes = 0;
  planned_B = 0;
}
void Net::destroy() {
  free_plan();
#if !defined(ADM_EMU)
  if (side) { conv_ksplit_release(side); (void)hipStreamDestroy(side); side = nullptr; }
  if (ev_fork) { (void)hipEventDestroy((hipEvent_t)ev_fork); ev_fork = nullptr; }
  if (ev_join) { (void)hipEventDestroy((hipEvent_t)ev_join); ev_join = nullptr; }
#endif
  for (void* p : owned) dfree(p);
  owned.clear();
}
// Assign activation buffers for batch B: exact-size free lists driven by liveness.
bool Net::plan_current(int B) const { return planned_B == B && plan_epoch == dispatch_epoch(); }
int Net::plan(int B) {
  // (an option set since the plan was made may have moved a layer to another kernel: the per-tensor statistic-tile counts and
  //  partial-sum buffers below follow the kernels, so the plan is rebuilt — ADVICE r5)
  if (plan_current(B)) return 0;
  planned_B = 0;
  use_known = false;                       // another batch size may dispatch other kernels: re-learn which packings are read
  learned_B.clear();
  pk_valid = false;
  for (ConvW& w : convs) w.used = 0;
  // the shared all-zero bias buffer is created lazily with a device allocation: do it here, outside any stream capture
  ADM_REQUIRE(conv_zero_bias(8192) != nullptr && conv_const_ones(8192) != nullptr, "plan: constant buffers");
  free_plan();
  std::multimap<size_t, float*> freelist;
  std::vector<std::vector<int>> dying(ops.size());
  for (size_t t = 0; t < tensors.size(); ++t)
    if (!tensors[t].external && tensors[t].last_use >= 0) dying[tensors[t].last_use].push_back((int)t);
  for (size_t i = 0; i < ops.size(); ++i) {
    const Op& o = ops[i];
    if (o.out >= 0 && !tensors[o.out].external && o.kind != Op::SOFTMAXC) {
      Tensor& t = tensors[o.out];
      const size_t bytes = sizeof(float) * (size_t)B * t.C * t.H * t.W;
      auto it = training ? freelist.end() : freelist.find(bytes);
      if (it != freelist.end()) { t.ptr = it->second; freelist.erase(it); }
      else ADM_TRY(arena_alloc((void**)&t.ptr, bytes));
    }
    if (!training)
      for (int t : dying[i]) {
        const Tensor& tt = tensors[t];
        freelist.insert({sizeof(float) * (size_t)B * tt.C * tt.H * tt.W, tt.ptr});
      }
  }
  for (GnBuf& g : gnbufs) {
    ADM_TRY(arena_alloc((void**)&g.scale, sizeof(float) * (size_t)B * g.C));
    ADM_TRY(arena_alloc((void**)&g.shift, sizeof(float) * (size_t)B * g.C));
    if (training) ADM_TRY(arena_alloc((void**)&g.mean_rstd, sizeof(float) * (size_t)B * groups * 2));
  }
  // GroupNorm statistics folded into the producing convolution (inference; after the scale / shift buffers exist, which the
  // kernels' eligibility rules look at): every tensor some GroupNorm reads gets a
  // partial-sum buffer when the kernel that will produce it has the epilogue (adm_conv_stats_tiles > 0 for its arguments).
  static const int fold = [] { const char* e = getenv("ADM_GN_FOLD"); return e ? atoi(e) : 1; }();
  for (Tensor& t : tensors) { t.stats = nullptr; t.stat_tiles = 0; t.want_stats = false; }
  if (!training && fold) {
    for (const Op& o : ops)
      if (o.kind == Op::GN) {
        tensors[o.in1].want_stats = true;
        if (o.in2 >= 0) tensors[o.in2].want_stats = true;
      }
    for (const Op& o : ops) {
      if (o.kind != Op::CONV || o.wt >= 0 || !tensors[o.out].want_stats || tensors[o.out].external) continue;
      adm_conv_args a;
      fill_conv_args(o, B, nullptr, 0, &a);
      // an external input's pointer (alignment) is unknown until run: only the conv_in class kernel does not depend on it
      if ((tensors[o.in1].external || (o.in2 >= 0 && tensors[o.in2].external)) && a.C1 > 4) continue;
      const int tiles = conv_stats_tiles(a);
      if (tiles <= 0) continue;
      Tensor& t = tensors[o.out];
      ADM_TRY(arena_alloc((void**)&t.stats, sizeof(double) * 2 * (size_t)B * t.C * tiles));
      t.stat_tiles = tiles;
    }
  }
  // GroupNorm statistics folded into the split-K finish pass of the producing convolution (planes of <= 8x8 pixels; both modes):
  // the FIRST single-input GroupNorm that reads a convolution's output is announced to that convolution's launch (run()); when
  // the launch took the request, the statistics op is skipped. Which launches split K is the kernels' business, decided per call.
  gn_fuse_of.assign(ops.size(), -1);
  {
    std::vector<int> prod(tensors.size(), -1);
    for (size_t i = 0; i < ops.size(); ++i) {
      const Op& o = ops[i];
      if (o.kind == Op::GN) {
        const int t = o.in1, pi = t >= 0 ? prod[t] : -1;
        if (o.in2 >= 0 || pi < 0 || tensors[t].external || t == t_in || tensors[t].C % groups != 0) continue;
        const Op& po = ops[pi];
        if (po.kind != Op::CONV || po.wt >= 0 || po.w == nullptr || po.w->Cout != tensors[t].C || gn_fuse_of[pi] >= 0) continue;
        gn_fuse_of[pi] = (int)i;
      } else if (o.out >= 0) {
        prod[o.out] = (int)i;
      }
    }
  }
  if (training) {
    // gradient buffer for every tensor except the network input; scratch sized for the largest layer
    size_t max_da = 0, max_ws = 0, max_w = 0;
    for (size_t t = 0; t < tensors.size(); ++t) {
      Tensor& tt = tensors[t];
      if ((int)t == t_in) continue;
      ADM_TRY(arena_alloc((void**)&tt.grad, sizeof(float) * (size_t)B * tt.C * tt.H * tt.W));
    }
    for (const Op& o : ops) {
      if (o.kind != Op::CONV || o.wt >= 0) continue;
      const Tensor& t1 = tensors[o.in1];
      const int C2 = o.in2 >= 0 ? tensors[o.in2].C : 0, Ct = (o.in1_C ? o.in1_C : t1.C) + C2;
      const int Hi = o.up ? 2 * t1.H : t1.H, Wi = o.up ? 2 * t1.W : t1.W;
      const size_t da = (size_t)B * Ct * Hi * Wi;
      if (da > max_da) max_da = da;
      adm_conv_args a; memset(&a, 0, sizeof(a));
      a.x1 = t1.ptr; a.C1 = t1.C; a.C2 = C2; a.x2 = C2 ? t1.ptr : nullptr; a.N = B; a.H = t1.H; a.W = t1.W;
      a.up = o.up; a.stride = o.stride; a.ks = o.ks; a.pad_lo = o.pad_lo; a.Cout = o.w->Cout;
      if (Ct > 4 && o.w->Cout > 4) {
        const size_t ws = (size_t)conv_wgrad_workspace(a, nullptr);
        if (ws > max_ws) max_ws = ws;
      }
      const size_t wn = (size_t)o.w->Cout * Ct * o.ks * o.ks;
      if (wn > max_w) max_w = wn;
    }
    blk.clear();
    if (conv_bf16_mode() >= 3) {
      blk.resize(ops.size());
      std::map<std::tuple<int, int, int>, void*> dy_imgs;
      size_t max_part = 4;
      for (size_t i = 0; i < ops.size(); ++i) {
        const Op& o = ops[i];
        if (o.kind != Op::CONV || o.wt >= 0 || o.ks != 3 || o.up > 1 || o.in1_C != 0) continue;
        // stride 1 "same", or Downsample2D.conv (stride 2: every other pixel of the stride-1 convolution; padding 1 -> the even
        // pixels, pad (0, 1, 0, 1) -> the odd ones)
        const bool s2 = o.stride == 2 && (o.pad_lo == 0 || o.pad_lo == 1) && !o.up && o.gn < 0 && !o.act && o.in2 < 0 && o.res < 0 && o.temb_off < 0;
        if (!s2 && (o.stride != 1 || o.pad_lo != 1)) continue;
        if (o.w == nullptr || o.w->wb == nullptr || o.w->wbT == nullptr || o.in1 == t_in) continue;
        if (o.up && (o.gn >= 0 || o.act || o.in2 >= 0)) continue;       // Upsample2D.conv: plain nearest x2, nothing on the load path
        const Tensor& t1 = tensors[o.in1];
        const int C1 = t1.C, C2 = o.in2 >= 0 ? tensors[o.in2].C : 0, Ct = C1 + C2, Cout = o.w->Cout, H = t1.H, W = t1.W;
        if (s2 && ((H | W) & 1)) continue;
        // the image has the SOURCE dims; the kernels tile Ho x Wo: the output (stride 1, up) or, for stride 2, the input plane again
        const int Ho = o.up ? 2 * H : H, Wo = o.up ? 2 * W : W;
        if (!blk_apply_eligible(C1, C2, H, W) || !blk_apply_eligible(Cout, 0, s2 ? H / 2 : Ho, s2 ? W / 2 : Wo)) continue;
        if (o.act && o.gn < 0) continue;
        // Upsample2D.conv onto an 8x8 plane: the blocked kernels have no nearest-x2 variant for 8-pixel rows (such a layer
        // was planned and then failed in train_step) — it stays on launch_conv2d / launch_conv_wgrad
        if (o.up && Wo == 8) continue;
        BlkOp& b = blk[i];
        b.s2 = s2;
        b.fwd = conv_bf16b_eligible(Ct, Cout, Ho, Wo, s2 ? 0 : B);      // (B: rows of 16 / 8 pixels tile 2 / 4 images side by side)
        b.wg = conv_wgradb_eligible(Ct, Cout, Ho, Wo, s2 ? 0 : B);
        b.dg = conv_bf16b_eligible(Cout, Ct, Ho, Wo, s2 ? 0 : B);
        if (s2 && !(b.fwd && b.wg && b.dg)) { b = BlkOp(); continue; }      // all three passes or none (one zero-inserted dy image)
        if (b.fwd || b.wg) {
          const size_t bytes = blk_image_bytes(B, Ct, H, W);
          ADM_TRY(arena_alloc(&b.xa, bytes));
          ADM_TRY(dmemset(b.xa, 0, bytes, nullptr));          // the halo stays zero for the life of the plan
        }
        if (b.wg || b.dg) {
          // (a zero-inserted image keeps its own zeros — the pad-0 variant dirties the odd pixels, the pad-1 variant the even ones, so
          // the two never share an image: key -Cout / -Cout - 2^20)
          void*& img = dy_imgs[std::make_tuple(s2 ? -Cout - (o.pad_lo == 0 ? (1 << 20) : 0) : Cout, Ho, Wo)];
          if (img == nullptr) {
            const size_t bytes = blk_image_bytes(B, Cout, Ho, Wo);
            ADM_TRY(arena_alloc(&img, bytes));
            ADM_TRY(dmemset(img, 0, bytes, nullptr));
          }
          b.dyb = img;
          const size_t pf = (size_t)blk_sums_scratch(B, Cout, s2 ? H / 2 : Ho, s2 ? W / 2 : Wo);
          if (pf > max_part) max_part = pf;
          if (b.wg) {
            const size_t ws = (size_t)conv_wgradb_workspace(Ct, Cout, B, Ho, Wo, nullptr);
            if (ws > max_ws) max_ws = ws;
          }
        }
      }
      // conv1 of a resnet: its output feeds one GroupNorm (+ SiLU) + convolution and nothing else, so its dy is that GroupNorm's dx —
      // the consumer's backward writes the 16-bit image (and the channel sums) straight away (launch_blk_gn_bwd_image)
      producer_of.assign(tensors.size(), -1);
      reader_count.assign(tensors.size(), 0);
      for (size_t i = 0; i < ops.size(); ++i) {
        if (ops[i].out >= 0) producer_of[ops[i].out] = (int)i;
        if (ops[i].kind == Op::GN) continue;                    // (the statistics op of a convolution is that convolution's read)
        for (int t : {ops[i].in1, ops[i].in2, ops[i].res, ops[i].wt}) if (t >= 0) ++reader_count[t];
      }
      if (blk_direct_dy()) {
        std::vector<int> producer(tensors.size(), -1), readers(tensors.size(), 0);
        for (size_t i = 0; i < ops.size(); ++i) {
          const Op& o = ops[i];
          if (o.out >= 0) producer[o.out] = (int)i;
          if (o.kind == Op::GN) continue;                       // the statistics op of the consuming convolution is not a reader of its own
          for (int t : {o.in1, o.in2, o.res, o.wt}) if (t >= 0) ++readers[t];
        }
        for (size_t i = 0; i < ops.size(); ++i) {
          const Op& o = ops[i];
          if (o.kind != Op::CONV || o.gn < 0 || o.in2 >= 0 || o.up || o.in1_C != 0 || o.wt >= 0) continue;
          const int t = o.in1, pi = producer[t];
          if (pi < 0 || readers[t] != 1 || tensors[t].external || t == t_in || t == t_out) continue;
          const Op& po = ops[pi];
          const BlkOp& pb = blk[pi];
          if (po.kind != Op::CONV || !pb.wg || !pb.dg || pb.s2 || po.up || po.res >= 0 || po.w == nullptr || !po.w->qkv_prefix.empty()) continue;
          if (tensors[t].C % groups != 0 || tensors[t].C <= 4) continue;
          // the consumer's GroupNorm backward writes the producer's dy image when op i runs in the reverse walk; the producer reads it
          // when op pi runs. No blocked convolution in between may use the same image (one buffer per (Cout, H, W)): true for the resnet
          // order conv1, GN, shortcut, conv2 — checked, not assumed
          bool clash = false;
          for (int j = pi + 1; j < (int)i; ++j) clash |= blk[j].dyb != nullptr && blk[j].dyb == pb.dyb;
          if (clash) continue;
          blk[i].img_for = pi;
        }
      }
      // GroupNorm statistics from the producing convolution's epilogue (as the inference plan does): tensors a blocked forward
      // kernel writes and some GroupNorm reads get per-tile partial sums; the read pass (gn_stats_kernel) disappears for them
      static const int fold_t = [] { const char* e = getenv("ADM_GN_FOLD_TRAIN"); return e ? atoi(e) : 1; }();
      if (fold_t) {
        for (const Op& o : ops)
          if (o.kind == Op::GN) {
            tensors[o.in1].want_stats = true;
            if (o.in2 >= 0) tensors[o.in2].want_stats = true;
          }
        for (size_t i = 0; i < ops.size(); ++i) {
          const Op& o = ops[i];
          if (!blk[i].fwd || blk[i].s2 || !tensors[o.out].want_stats || tensors[o.out].external) continue;
          Tensor& t = tensors[o.out];
          const int tiles = conv_bf16b_stats_tiles(t.H, t.W);
          if (tiles <= 0) continue;                              // rows of 16 / 8 pixels: no statistics epilogue
          ADM_TRY(arena_alloc((void**)&t.stats, sizeof(double) * 2 * (size_t)B * t.C * tiles));
          t.stat_tiles = tiles;
        }
      }
      ADM_TRY(arena_alloc((void**)&blk_part, sizeof(float) * max_part));
      ADM_TRY(stream_sync(nullptr));
    }
    ADM_TRY(arena_alloc((void**)&tmp_da, sizeof(float) * max_da)); tmp_da_floats = max_da;
    ADM_TRY(arena_alloc((void**)&wgrad_ws, sizeof(float) * (max_ws ? max_ws : 4))); wgrad_ws_floats = max_ws;
    ADM_TRY(arena_alloc((void**)&tmp_w, sizeof(float) * (max_w + 4096))); tmp_w_floats = max_w + 4096;
    ADM_TRY(arena_alloc((void**)&s12, sizeof(float) * (size_t)B * groups * 2));
  }
  ADM_TRY(plan_side_overlap(B));
  planned_B = B;
  plan_epoch = dispatch_epoch();
  return 0;
}

// which weight packing the convolution that just ran read: `fwd` = forward pass (wp / wu / wb) or data gradient (wpT / wuT / wbT)
static unsigned packing_of_variant(int var, bool fwd) {
  if (var / 1000 == 5) return fwd ? PK_WB : PK_WBT;         // bf16-operand kernels (3x3: 5316, 1x1: 5116)
  if (var / 1000 == 4) return fwd ? PK_WU : PK_WUT;         // Winograd kernels
  return fwd ? PK_WP : PK_WPT;                              // direct MFMA / small-channel kernels
}

// Which convolutions run on the side stream (see net_exec.h). Op j qualifies when it is a plain 1x1 convolution (no GroupNorm / activation on
// its load path, no statistics epilogue, weights of its own), its inputs are produced at least two ops before it, and NO tensor dies between the
// hoist point and j: the arena hands a dead tensor's buffer to later outputs in op order, and only with no death in the window can neither j's
// output buffer nor anything j reads be touched by the ops it now runs beside. One side launch in flight at a time (the two events are reused).
int Net::plan_side_overlap(int B) {
  const int n = (int)ops.size();
  hoist_from.assign(n, -1);
  hoist_at.assign(n, {});
  static const int on = [] { const char* e = getenv("ADM_SIDE_OVERLAP"); return e ? atoi(e) : 1; }();
  if (training || !on) return 0;
  std::vector<int> prod(tensors.size(), -1), deaths(n, 0);
  for (int i = 0; i < n; ++i) if (ops[i].out >= 0) prod[ops[i].out] = i;
  for (const Tensor& t : tensors) if (!t.external && t.last_use >= 0 && t.last_use < n) ++deaths[t.last_use];
  int busy_until = -1;                                          // the previous side launch's own position: windows must not overlap
  bool any = false;
  for (int j = 0; j < n; ++j) {
    const Op& o = ops[j];
    if (o.kind != Op::CONV || o.ks != 1 || o.stride != 1 || o.up || o.gn >= 0 || o.act || o.wt >= 0 || o.w == nullptr || o.out < 0) continue;
    const Tensor& to = tensors[o.out];
    if (to.external || to.stats != nullptr || ((size_t)j < gn_fuse_of.size() && gn_fuse_of[j] >= 0)) continue;
    if ((double)B * to.C * to.H * to.W > 4.0 * 1024 * 1024) continue;      // a launch of this size fills the chip by itself
    int ready = -1;                                             // last producer of an input
    bool ext_in = false;
    for (int t : {o.in1, o.in2, o.res}) if (t >= 0) { ready = prod[t] > ready ? prod[t] : ready; ext_in |= tensors[t].external && t != t_in; }
    if (ext_in) continue;
    int i = j;                                                  // walk back while nothing dies and the inputs are ready
    while (i - 1 > ready && i - 1 > busy_until && deaths[i - 1] == 0) --i;
    if (j - i < 2) continue;
    hoist_from[j] = i;
    hoist_at[i].push_back(j);
    busy_until = j;
    any = true;
  }
  (void)any;
#if !defined(ADM_EMU)
  if (any && side == nullptr) {                                 // (created here, outside any stream capture)
    ADM_HIP_OK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t e1 = nullptr, e2 = nullptr;
    ADM_HIP_OK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    ADM_HIP_OK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    ev_fork = e1; ev_join = e2;
  }
#endif
  return 0;
}

int Net::launch_side_conv(const Op& o, int B, const float* temb_all, int temb_stride, hipStream_t st) {
  adm_conv_args a;
  fill_conv_args(o, B, temb_all, temb_stride, &a);
  ADM_TRY(launch_conv2d(a, st));
  if (o.w) ADM_TRY(note_packing(*o.w, packing_of_variant(last_conv_variant(), true)));
  return 0;
}

void Net::fill_conv_args(const Op& o, int B, const float* temb_all, int temb_stride, adm_conv_args* ap) const {
  adm_conv_args& a = *ap;
  memset(&a, 0, sizeof(a));
  const Tensor& t1 = tensors[o.in1];
  const long plane = (long)t1.H * t1.W;
  a.x1 = t1.ptr + (long)o.in1_coff * plane;
  a.C1 = o.in1_C ? o.in1_C : t1.C;
  a.x1_bstride = (long)t1.C * plane;
  if (o.in2 >= 0) { a.x2 = tensors[o.in2].ptr; a.C2 = tensors[o.in2].C; }
  a.N = B; a.H = t1.H; a.W = t1.W;
  a.up = o.up; a.stride = o.stride; a.ks = o.ks; a.pad_lo = o.pad_lo;
  if (o.gn >= 0) { a.gn_scale = gnbufs[o.gn].scale; a.gn_shift = gnbufs[o.gn].shift; }
  a.act = o.act;
  if (o.wt >= 0) {  // per-sample weights living in an activation tensor
    const Tensor& tw = tensors[o.wt];
    const long wplane = (long)tw.H * tw.W;
    a.wpacked = tw.ptr + (long)o.wt_coff * wplane;
    a.w_bstride = (long)tw.C * wplane;
    a.bias = nullptr;
    a.Cout = o.dyn_cout;
  } else {
    a.wpacked = o.w->wp; a.bias = o.w->bias; a.Cout = o.w->Cout;
    if (o.stride == 1) { a.wino_packed = o.w->wu; a.bf16_packed = o.w->wb; }   // eligibility (shape, mode): the launcher
  }
  if (o.temb_off >= 0 && temb_all) { a.chan_add = temb_all + o.temb_off; a.chan_add_stride = temb_stride; }
  if (o.res >= 0) a.residual = tensors[o.res].ptr;
  a.out = tensors[o.out].ptr;
  a.wino6_rule = wino6_rule;
  a.single_sample = single_sample;
}

int Net::run(const float* x, float* out, int B, const float* temb_all, int temb_stride, hipStream_t st, OpTimer* tm) {
  OpTimer none;
  if (!tm) tm = &none;
  tm->st = st;
  tensors[t_in].ptr = const_cast<float*>(x);
  tensors[t_out].ptr = out;
  gn_skip.assign(ops.size(), 0);
  // side-stream overlap: only on the product path proper (no per-op timers) and only for a plan made for this batch size
  const bool overlap = tm == &none && !training && hoist_from.size() == ops.size() && planned_B == B;
  for (const Op& o : ops) {
    const Tensor& t1 = tensors[o.in1];
    const size_t oidx = (size_t)(&o - ops.data());
    if (overlap) {
      for (int j : hoist_at[oidx]) {
#if !defined(ADM_EMU)
        ADM_HIP_OK(hipEventRecord((hipEvent_t)ev_fork, st));            // everything enqueued so far (the inputs' producers) ...
        ADM_HIP_OK(hipStreamWaitEvent(side, (hipEvent_t)ev_fork, 0));    // ... precedes the side launch
        ADM_TRY(launch_side_conv(ops[j], B, temb_all, temb_stride, side));
        ADM_HIP_OK(hipEventRecord((hipEvent_t)ev_join, side));
#else
        ADM_TRY(launch_side_conv(ops[j], B, temb_all, temb_stride, st));  // emulator: the hoisted ORDER, in line (exercises the buffer-reuse rule)
#endif
      }
      if (hoist_from[oidx] >= 0) {                                   // its own position: join
#if !defined(ADM_EMU)
        ADM_HIP_OK(hipStreamWaitEvent(st, (hipEvent_t)ev_join, 0));
#endif
        continue;
      }
    }
    tm->begin();
    if (o.kind == Op::GN && gn_skip[(size_t)(&o - ops.data())]) {
      tm->end(0, 2, 0.0, 0.0);           // its scale / shift came with the producing convolution's split-K finish pass
    } else if (o.kind == Op::GN) {
      const GnBuf& g = gnbufs[o.gn];
      const float* x2 = o.in2 >= 0 ? tensors[o.in2].ptr : nullptr;
      const int C2 = o.in2 >= 0 ? tensors[o.in2].C : 0;
      const bool folded = t1.stats != nullptr && (o.in2 < 0 || tensors[o.in2].stats != nullptr) &&
                          (!training || conv_bf16_mode() >= 3);   // training plan: the blocked forward kernels write them
      if (folded) {    // the producing convolutions left per-tile partial sums: no pass over the activation
        const Tensor* t2 = o.in2 >= 0 ? &tensors[o.in2] : nullptr;
        ADM_TRY(launch_groupnorm_finalize(t1.stats, t1.C, t1.stat_tiles, t2 ? t2->stats : nullptr, C2, t2 ? t2->stat_tiles : 0, B,
                                          t1.H * t1.W, groups, o.eps > 0.f ? o.eps : eps, o.g->gamma, o.g->beta, g.scale, g.shift,
                                          st, g.mean_rstd));
        tm->end(0, 1, 3.0 * B * (t1.C + C2) * t1.H * t1.W,
                16.0 * B * ((double)t1.C * t1.stat_tiles + (t2 ? (double)C2 * t2->stat_tiles : 0.0)));
      } else {
        ADM_TRY(launch_groupnorm_stats(t1.ptr, t1.C, x2, C2, B, t1.H * t1.W, groups, o.eps > 0.f ? o.eps : eps, o.g->gamma,
                                       o.g->beta, g.scale, g.shift, st, g.mean_rstd));
        tm->end(0, 0, 3.0 * B * (t1.C + C2) * t1.H * t1.W, 4.0 * B * (t1.C + C2) * t1.H * t1.W);
      }
    } else if (o.kind == Op::CONV) {
      adm_conv_args a;
      fill_conv_args(o, B, temb_all, temb_stride, &a);
      if (tensors[o.out].stats != nullptr && !training) { a.stats_out = tensors[o.out].stats; a.stats_tiles = tensors[o.out].stat_tiles; }
      const size_t oi = (size_t)(&o - ops.data());
      const BlkOp* bo = (training && conv_bf16_mode() >= 3 && oi < blk.size() && blk[oi].xa != nullptr) ? &blk[oi] : nullptr;
      if (bo)    // level 3: the activated input once as a blocked 16-bit image (kept for the weight gradient)
        ADM_TRY(launch_blk_apply(a.x1, a.C1, a.x1_bstride, a.x2, a.C2, a.x2_bstride, B, a.H, a.W, a.gn_scale, a.gn_shift, a.act, bo->xa,
                                 nullptr, st));
      {
        // the first single-input GroupNorm that reads this output, announced to the launch: a split-K finish pass that can leave its
        // scale / shift takes the request (k_groupnorm.hip), and the statistics op below is then skipped
        GnFuse gf;
        const int gk = oi < gn_fuse_of.size() ? gn_fuse_of[oi] : -1;
        const bool ask = gk >= 0 && tensors[o.out].stats == nullptr;
        if (ask) {
          const Op& go = ops[gk];
          const GnBuf& gb = gnbufs[go.gn];
          gf.gamma = go.g->gamma; gf.beta = go.g->beta; gf.eps = go.eps > 0.f ? go.eps : eps; gf.groups = groups;
          gf.scale = gb.scale; gf.shift = gb.shift; gf.mean_rstd = gb.mean_rstd;
          conv_gn_fuse_request(&gf);
        }
        int rc;
        if (bo && bo->fwd && conv_bf16b_eligible(a.C1 + a.C2, a.Cout, o.up ? 2 * a.H : a.H, o.up ? 2 * a.W : a.W, bo->s2 ? 0 : B))
          rc = launch_conv_bf16b(bo->xa, a.C1 + a.C2, B, o.up ? 2 * a.H : a.H, o.up ? 2 * a.W : a.W, o.w->wb, a.Cout, a.bias, a.chan_add,
                                 a.chan_add_stride, a.residual, a.out, st, bo->s2 ? (o.pad_lo ? 3 : 2) : o.up, bo->s2 ? nullptr : tensors[o.out].stats);
        else
          rc = launch_conv2d(a, st);
        if (ask) {
          if (rc == 0 && conv_gn_fuse_taken()) gn_skip[gk] = 1;
          conv_gn_fuse_request(nullptr);
        }
        ADM_TRY(rc);
      }
      if (o.w) ADM_TRY(note_packing(*o.w, packing_of_variant(last_conv_variant(), true)));
      const Tensor& to = tensors[o.out];
      const double Cin = a.C1 + a.C2, outel = (double)B * to.C * to.H * to.W;
      const int var = last_conv_variant();
      tm->end((var >= 1000 && var < 2000) ? 3 : 1, var, 2.0 * outel * Cin * o.ks * o.ks,
              4.0 * ((double)B * Cin * t1.H * t1.W + outel * (o.res >= 0 ? 2 : 1) + (double)to.C * Cin * o.ks * o.ks));
    } else if (o.kind == Op::ATTN) {
      const int C = t1.C / 3, T = t1.H * t1.W;
      ADM_TRY(launch_attention(t1.ptr, tensors[o.out].ptr, B, C, T, o.head_dim, st, single_sample));
      tm->end(2, o.head_dim, 4.0 * B * C * (double)T * T, 16.0 * B * C * T);
    } else if (o.kind == Op::LN) {
      const long T = (long)t1.H * t1.W;
      ADM_TRY(launch_layernorm_nct(t1.ptr, o.g->gamma, o.g->beta, tensors[o.out].ptr, B, t1.C, T, o.eps, st));
      tm->end(7, 0, 8.0 * B * t1.C * T, 16.0 * B * t1.C * T);
    } else if (o.kind == Op::GEGLU) {
      const long T = (long)t1.H * t1.W;
      ADM_TRY(launch_geglu(t1.ptr, tensors[o.out].ptr, B, t1.C / 2, T, st));
      tm->end(8, 0, 10.0 * B * (t1.C / 2) * T, 6.0 * B * t1.C * T);
    } else if (o.kind == Op::XATTN) {
      ADM_REQUIRE(ctx != nullptr && ctx_S > 0, "conditional UNet: no encoding set (adm_unet_set_encoding) before the forward");
      const int T = t1.H * t1.W;
      ADM_TRY(launch_cross_attention(t1.ptr, ctx, o.wk, o.wv, tensors[o.out].ptr, B, t1.C, T, ctx_S, ctx_D, o.head_dim, st));
      tm->end(9, o.head_dim, 4.0 * B * t1.C * (double)T * ctx_S, 8.0 * B * t1.C * T);
    } else if (o.kind == Op::SOFTMAXC) {
      const int T = t1.H * t1.W;
      ADM_TRY(launch_softmax_channels(t1.ptr, B, t1.C, T, o.scale, st));
      tm->end(5, 0, 5.0 * B * t1.C * T, 8.0 * B * t1.C * T);
    } else {  // TRANSP: channel slice (C, T) -> (T, C)
      const int T = t1.H * t1.W;
      ADM_TRY(launch_transpose_ct(t1.ptr + (long)o.in1_coff * T, (long)t1.C * T, tensors[o.out].ptr, B, o.in1_C, T, st));
      tm->end(6, 0, 0.0, 8.0 * B * o.in1_C * T);
    }
  }
  tm->finish();
  return 0;
}

}  // namespace adm

// ---------------------------------------------------------------------------------------------- Net: backward
// Reverse pass over the op list (training). Gradient fan-in uses per-tensor "already initialised" flags: the first
// contribution writes, later ones accumulate, so no gradient buffer needs a memset. Parameter gradients go to
// grads_base at the offset of their master parameter (the caller zeroes that flat buffer once per step).
namespace adm {

int Net::run_backward(int B, float* dtemb_all, int temb_stride, hipStream_t st) {
  ADM_REQUIRE(training && params_base && grads_base, "run_backward: training mode is not enabled");
  for (Tensor& t : tensors) t.ginit = false;
  tensors[t_out].ginit = true;
  for (BlkOp& b : blk) b.img_done = false;
  bias_done.assign(ops.size(), 0);
  auto contribute = [&](int t, const float* src, long src_bs, int C) -> int {
    Tensor& tt = tensors[t];
    if (t == t_in) return 0;
    const long plane = (long)tt.H * tt.W;
    ADM_REQUIRE(C == tt.C, "run_backward: partial-channel gradient fan-in is not supported");
    ADM_TRY(launch_accumulate(tt.grad, (long)tt.C * plane, src, src_bs, (long)C * plane, B, tt.ginit ? 1 : 0, st));
    tt.ginit = true;
    return 0;
  };
  for (int i = (int)ops.size() - 1; i >= 0; --i) {
    const Op& o = ops[i];
    if (o.kind == Op::GN) continue;
    Tensor& t1 = tensors[o.in1];
    if (o.kind == Op::ATTN) {
      Tensor& to = tensors[o.out];
      ADM_REQUIRE(to.ginit && !t1.ginit, "run_backward: attention gradient state");
      const int C = t1.C / 3, T = t1.H * t1.W;
      if (o.head_dim <= 32 && sizeof(float) * ((size_t)4 * T * o.head_dim + 3 * T) <= 64 * 1024) {
        ADM_TRY(launch_attention_bwd(t1.ptr, to.grad, t1.grad, B, C, T, o.head_dim, st));
      } else {   // head slab too large for LDS (transformer blocks at 32x32 / 64x64 latents): key / query blocks
        ADM_REQUIRE(tmp_da_floats >= (size_t)3 * B * (C / o.head_dim) * T, "run_backward: scratch too small");
        ADM_TRY(launch_attention_bwd_blocked(t1.ptr, to.grad, t1.grad, tmp_da, B, C, T, o.head_dim, 0, st));
      }
      t1.ginit = true;
      continue;
    }
    if (o.kind == Op::LN) {
      Tensor& to = tensors[o.out];
      const long T = (long)t1.H * t1.W;
      ADM_REQUIRE(to.ginit, "run_backward: LayerNorm output gradient missing");
      ADM_REQUIRE(tmp_da_floats >= (size_t)2 * B * T, "run_backward: scratch too small");
      ADM_TRY(launch_layernorm_nct_bwd(t1.ptr, to.grad, o.g->gamma, t1.grad, t1.ginit ? 1 : 0, tmp_da, grad_of(o.g->gamma),
                                       grad_of(o.g->beta), B, t1.C, T, o.eps, st));
      mark_ready(o.g->gamma, (size_t)t1.C);
      mark_ready(o.g->beta, (size_t)t1.C);
      t1.ginit = true;
      continue;
    }
    if (o.kind == Op::GEGLU) {
      Tensor& to = tensors[o.out];
      ADM_REQUIRE(to.ginit && !t1.ginit, "run_backward: GEGLU gradient state");
      ADM_TRY(launch_geglu_bwd(t1.ptr, to.grad, t1.grad, B, t1.C / 2, (long)t1.H * t1.W, st));
      t1.ginit = true;
      continue;
    }
    if (o.kind == Op::XATTN) {
      Tensor& to = tensors[o.out];
      ADM_REQUIRE(to.ginit && !t1.ginit, "run_backward: cross-attention gradient state");
      ADM_REQUIRE(ctx != nullptr && ctx_S > 0, "run_backward: no encoding set");
      ADM_TRY(launch_cross_attention_bwd(t1.ptr, ctx, o.wk, o.wv, to.grad, t1.grad, grad_of(o.wk), grad_of(o.wv), B, t1.C,
                                         t1.H * t1.W, ctx_S, ctx_D, o.head_dim, st));
      mark_ready(o.wk, (size_t)t1.C * ctx_D);
      mark_ready(o.wv, (size_t)t1.C * ctx_D);
      t1.ginit = true;
      continue;
    }
    ADM_REQUIRE(o.kind == Op::CONV && o.wt < 0, "run_backward: op kind not supported in training (UNet ops only)");
    Tensor& to = tensors[o.out];
    ADM_REQUIRE(to.ginit, "run_backward: output gradient missing");
    const float* dy = to.grad;
    const long plane_o = (long)to.H * to.W;
    const int C1 = t1.C, C2 = o.in2 >= 0 ? tensors[o.in2].C : 0, Ct = C1 + C2, Cout = to.C;
    const int Hi = o.up ? 2 * t1.H : t1.H, Wi = o.up ? 2 * t1.W : t1.W;
    const float* x2 = o.in2 >= 0 ? tensors[o.in2].ptr : nullptr;
    const bool qkv = !o.w->qkv_prefix.empty();
    // ---- residual fan-in --------------------------------------------------------------------------------
    bool res_swapped = false;
    if (o.res >= 0) {
      Tensor& tr = tensors[o.res];
      if (o.res != t_in && !tr.ginit && !tr.external && !to.external && o.out != t_out && tr.C == Cout && tr.H == to.H && tr.W == to.W) {
        // first contribution to the residual's gradient = dy itself: hand the BUFFER over instead of copying it (the two tensors
        // have one shape). `dy` keeps pointing at the data for the rest of this op; later contributions accumulate into it, by
        // which time this convolution's own backward kernels — queued before them on the stream — have read it; the output
        // tensor's gradient is dead after this op (its consumers ran earlier in the reverse walk)
        std::swap(tr.grad, to.grad);
        tr.ginit = true;
        res_swapped = true;
      } else {
        ADM_TRY(contribute(o.res, dy, (long)Cout * plane_o, Cout));
      }
    }
    // ---- bias (+ time-embedding bias) gradients ------------------------------------------------------------
    float* dW = qkv ? tmp_w : grad_of(ps->P(o.w->key + ".weight"));
    float* dbias = qkv ? tmp_w + (size_t)Cout * Ct : (o.w->has_bias ? grad_of(ps->P(o.w->key + ".bias")) : nullptr);
    if (qkv) ADM_TRY(dmemset(dbias, 0, sizeof(float) * Cout, st));
    const BlkOp* bo = (conv_bf16_mode() >= 3 && (size_t)i < blk.size() && (blk[i].wg || blk[i].dg)) ? &blk[i] : nullptr;
    float* dtemb_o = (o.temb_off >= 0 && dtemb_all) ? dtemb_all + o.temb_off : nullptr;
    // the shortcut convolution of this resnet (the producer of the residual) sees the SAME dy — the buffer has just been handed to
    // its output's gradient and nothing else adds to it — so its bias gradient is this convolution's channel sums: second target
    float* dbias_sc = nullptr;
    // (only when this convolution is the residual's ONLY reader — a conv_shortcut output; the output of an attention block's to_out
    // also feeds the resnet's norm1 / conv1, and its gradient is more than this dy)
    if (res_swapped && (size_t)o.res < producer_of.size() && producer_of[o.res] >= 0 && reader_count[o.res] == 1) {
      const int ri = producer_of[o.res];
      const Op& ro = ops[ri];
      const bool r_blocked = (size_t)ri < blk.size() && (blk[ri].wg || blk[ri].dg);
      if (ro.kind == Op::CONV && ro.wt < 0 && ro.w && ro.w->has_bias && ro.w->qkv_prefix.empty() && ro.temb_off < 0 && !r_blocked &&
          ro.w->Cout == Cout && ri < i) {
        dbias_sc = grad_of(ps->P(ro.w->key + ".bias"));
        if (!(conv_bf16_mode() >= 3 && (size_t)i < blk.size() && (blk[i].wg || blk[i].dg) && !blk[i].img_done)) dbias_sc = nullptr;
        if (dbias_sc) bias_done[ri] = 1;
      }
    }
    if (bias_done[i]) {
      // (this convolution's bias gradient has been added with its resnet's conv2 channel sums)
    } else if (bo && bo->img_done) {
      // the consumer's GroupNorm backward has written this convolution's dy image and its channel sums already
    } else if (bo) {    // level 3: dy once as a blocked 16-bit image for the weight-gradient and the data-gradient kernel; the pass
                 // also leaves the channel sums that adm_chan_sums would read dy a second time for
      ADM_TRY(launch_blk_apply(dy, Cout, 0, nullptr, 0, 0, B, to.H, to.W, nullptr, nullptr, 0, bo->dyb, blk_part, st, bo->s2 ? (o.pad_lo ? 2 : 1) : 0));
      ADM_TRY(launch_blk_sums_finalize(blk_part, B, Cout, to.H, to.W, dtemb_o, temb_stride, 0, dbias, st, dbias_sc));
    } else {
      ADM_TRY(launch_chan_sums(dy, B, Cout, (int)plane_o, dtemb_o, temb_stride, 0, dbias, st));
    }
    // ---- weight gradient -----------------------------------------------------------------------------------
    const float* gsc = o.gn >= 0 ? gnbufs[o.gn].scale : nullptr;
    const float* gsh = o.gn >= 0 ? gnbufs[o.gn].shift : nullptr;
    const bool small_cin = Ct <= 4, small_cout = Cout <= 4;
    if (small_cin) {
      ADM_REQUIRE(o.ks == 3 && o.stride == 1 && !o.up && o.gn < 0 && !o.act, "run_backward: conv_in class shape");
      ADM_TRY(launch_conv_small_cin_wgrad(t1.ptr, Ct, B, t1.H, t1.W, dy, Cout, dW, st));
    } else if (small_cout) {
      ADM_REQUIRE(o.ks == 3 && o.stride == 1 && !o.up && C2 == 0, "run_backward: conv_out class shape");
      ADM_REQUIRE(tmp_da_floats >= (size_t)B * Ct * Hi * Wi, "run_backward: scratch too small");
      ADM_TRY(launch_conv_small_cout_bwd(t1.ptr, Ct, B, t1.H, t1.W, gsc, gsh, o.act, ps->P(o.w->key + ".weight"), dy, Cout,
                                         tmp_da, dW, st));
    } else if (bo && bo->wg && conv_wgradb_eligible(Ct, Cout, bo->s2 ? t1.H : to.H, bo->s2 ? t1.W : to.W, bo->s2 ? 0 : B)) {
      ADM_TRY(launch_conv_wgradb(bo->xa, Ct, bo->dyb, Cout, B, bo->s2 ? t1.H : to.H, bo->s2 ? t1.W : to.W, dW, 0, wgrad_ws, st, o.up));
    } else {
      adm_conv_args a;
      memset(&a, 0, sizeof(a));
      a.x1 = t1.ptr; a.C1 = C1; a.x2 = x2; a.C2 = C2; a.N = B; a.H = t1.H; a.W = t1.W;
      a.up = o.up; a.stride = o.stride; a.ks = o.ks; a.pad_lo = o.pad_lo;
      a.gn_scale = gsc; a.gn_shift = gsh; a.act = o.act; a.Cout = Cout;
      ADM_TRY(launch_conv_wgrad(a, dy, dW, 0, wgrad_ws, st));
    }
    if (qkv) {  // scatter the stacked q|k|v gradients to the three master parameters
      const int C = Ct;
      const char* names[3] = {".to_q", ".to_k", ".to_v"};
      for (int k = 0; k < 3; ++k) {
        ADM_TRY(copy_d2d(grad_of(ps->P(o.w->qkv_prefix + names[k] + ".weight")), tmp_w + (size_t)k * C * C,
                         sizeof(float) * (size_t)C * C, st));
        mark_ready(ps->P(o.w->qkv_prefix + names[k] + ".weight"), (size_t)C * C);
        if (!o.w->has_bias) continue;
        ADM_TRY(copy_d2d(grad_of(ps->P(o.w->qkv_prefix + names[k] + ".bias")), dbias + (size_t)k * C, sizeof(float) * C, st));
        mark_ready(ps->P(o.w->qkv_prefix + names[k] + ".bias"), (size_t)C);
      }
    } else {
      mark_ready(ps->P(o.w->key + ".weight"), (size_t)Cout * Ct * o.ks * o.ks);
      if (o.w->has_bias) mark_ready(ps->P(o.w->key + ".bias"), (size_t)Cout);
    }
    // ---- data gradient ---------------------------------------------------------------------------------------
    if (o.in1 == t_in) continue;  // the network input needs no gradient
    const bool direct = o.gn < 0 && !o.up && o.in2 < 0 && !small_cout;
    if (!small_cout) {
      ADM_REQUIRE(o.ks == 1 || o.pad_lo == 1, "run_backward: only symmetric padding is supported");
      adm_conv_args a;
      memset(&a, 0, sizeof(a));
      a.x1 = dy; a.C1 = Cout; a.N = B; a.H = to.H; a.W = to.W;
      a.up = o.stride == 2 ? 2 : 0; a.stride = 1; a.ks = o.ks; a.pad_lo = o.ks == 3 ? 1 : 0;
      a.wpacked = o.w->wpT; a.bias = nullptr; a.Cout = Ct;
      if (o.stride == 1) a.wino_packed = o.w->wuT;   // 3x3 stride-1: the data gradient is a Winograd-eligible convolution too
      a.bf16_packed = o.w->wbT;                      // bf16: stride-1, and (opt-in level 2) stride-2 via zero insertion
      if (direct) {
        a.out = t1.grad;
        if (t1.ginit) a.residual = t1.grad;   // accumulate in the epilogue (same thread reads then writes)
      } else {
        ADM_REQUIRE(tmp_da_floats >= (size_t)B * Ct * Hi * Wi, "run_backward: scratch too small");
        a.out = tmp_da;
      }
      if (o.ks == 1 && o.in2 >= 0 && o.gn < 0 && !o.act && o.in1_C == 0 && !o.up && conv_bf16_mode() >= 2 && C1 % 32 == 0 &&
          a.bf16_packed != nullptr) {
        // 1x1 convolution over a virtual concat (the shortcuts of the up blocks): its data gradient goes straight into the two source
        // tensors' gradient buffers (each written or accumulated in the epilogue) instead of a scratch tensor + two fan-in passes
        adm_conv_args a2 = a;
        Tensor& t2 = tensors[o.in2];
        a2.out = t1.grad; a2.residual = t1.ginit ? t1.grad : nullptr;
        if (conv1x1_bf16_eligible(a2) && o.in1 != t_in && o.in2 != t_in) {
          ADM_TRY(launch_conv1x1_bf16_split(a2, C1, t2.grad, t2.ginit ? t2.grad : nullptr, st));
          ADM_TRY(note_packing(*o.w, PK_WBT));
          t1.ginit = true; t2.ginit = true;
          continue;
        }
      }
      if (bo && bo->dg && conv_bf16b_eligible(Cout, Ct, bo->s2 ? t1.H : to.H, bo->s2 ? t1.W : to.W, bo->s2 ? 0 : B))
                             // (stride 2: the zero-inserted dy image has the INPUT's dims; the kernel is the plain stride-1 one)
        ADM_TRY(launch_conv_bf16b(bo->dyb, Cout, B, bo->s2 ? t1.H : to.H, bo->s2 ? t1.W : to.W, o.w->wbT, Ct, nullptr, nullptr, 0, a.residual,
                                  a.out, st));
      else
        ADM_TRY(launch_conv2d(a, st));
      ADM_TRY(note_packing(*o.w, packing_of_variant(last_conv_variant(), false)));
      if (direct) { t1.ginit = true; continue; }
    }
    const long plane_i = (long)t1.H * t1.W;
    if (o.gn >= 0) {
      ADM_REQUIRE(!o.up, "run_backward: GroupNorm + upsample in one conv is not supported");
      const GnBuf& gb = gnbufs[o.gn];
      Tensor* t2 = o.in2 >= 0 ? &tensors[o.in2] : nullptr;
      int pi = (conv_bf16_mode() >= 3 && (size_t)i < blk.size()) ? blk[i].img_for : -1;
      if (pi >= 0) {     // a batch smaller than the planned one may not fill the narrow-row tiles: the producer then reads the fp32 dy
        const Op& po = ops[pi];
        const int pCt = tensors[po.in1].C + (po.in2 >= 0 ? tensors[po.in2].C : 0);
        if (!conv_wgradb_eligible(pCt, C1, t1.H, t1.W, B) || !conv_bf16b_eligible(C1, pCt, t1.H, t1.W, B)) pi = -1;
      }
      if (pi >= 0 && !t1.ginit) {
        const Op& po = ops[pi];
        ADM_TRY(launch_gn_backward_stats(t1.ptr, C1, nullptr, 0, tmp_da, B, (int)plane_i, groups, gb.mean_rstd, gb.g->gamma, gb.g->beta, o.act,
                                         s12, grad_of(gb.g->gamma), grad_of(gb.g->beta), st));
        ADM_TRY(launch_blk_gn_bwd_image(t1.ptr, C1, tmp_da, B, t1.H, t1.W, groups, gb.mean_rstd, gb.g->gamma, gb.g->beta, o.act, s12,
                                        blk[pi].dyb, blk_part, st));
        float* pdb = po.w->has_bias ? grad_of(ps->P(po.w->key + ".bias")) : nullptr;
        ADM_TRY(launch_blk_sums_finalize(blk_part, B, C1, t1.H, t1.W, (po.temb_off >= 0 && dtemb_all) ? dtemb_all + po.temb_off : nullptr,
                                         temb_stride, 0, pdb, st));
        blk[pi].img_done = true;
      } else
      ADM_TRY(launch_gn_backward(t1.ptr, C1, x2, C2, tmp_da, B, (int)plane_i, groups, gb.mean_rstd, gb.g->gamma, gb.g->beta,
                                 o.act, s12, grad_of(gb.g->gamma), grad_of(gb.g->beta), t1.grad, t1.ginit ? 1 : 0,
                                 t2 ? t2->grad : nullptr, (t2 && t2->ginit) ? 1 : 0, st));
      mark_ready(gb.g->gamma, (size_t)gb.g->C);
      mark_ready(gb.g->beta, (size_t)gb.g->C);
      t1.ginit = true;
      if (t2) t2->ginit = true;
    } else if (o.up) {
      ADM_REQUIRE(o.in2 < 0 && !o.act, "run_backward: upsample conv with concat/activation is not supported");
      ADM_TRY(launch_sumpool2x2(tmp_da, t1.grad, Hi, Wi, (long)B * Ct, t1.ginit ? 1 : 0, st));
      t1.ginit = true;
    } else {
      ADM_REQUIRE(!o.act, "run_backward: activation without GroupNorm is not supported");
      ADM_TRY(contribute(o.in1, tmp_da, (long)Ct * plane_i, C1));
      if (o.in2 >= 0) ADM_TRY(contribute(o.in2, tmp_da + (long)C1 * plane_i, (long)Ct * plane_i, C2));
    }
  }
  known_epoch = dispatch_epoch();
  { bool seen = false; for (int b : learned_B) seen |= b == B; if (!seen) learned_B.push_back(B); }
  use_known = true;      // a complete forward + backward has run on this plan: ConvW::used now lists every packing that is read
  return 0;
}

}  // namespace adm
