// unet_exec.hip — native UNet2DModel executor + whole-loop sampler behind the C-ABI (include/adm.h).
//
// Replaces `self.unet(images, t)["sample"]` and the Python denoising loop of
// audiodiffusion/pipeline_audio_diffusion.py:159-185. Design (MI355X-first):
//  * the layer graph is built ONCE from the diffusers config into a flat op list (GroupNorm-stats, fused conv,
//    attention core) — ~190 kernel launches per forward instead of ~300 eager ops, no concat / upsample /
//    residual / activation kernels at all (they are folded into the conv load path and epilogue);
//  * weights are uploaded by diffusers state-dict key, then repacked once ([Cin][tap][Cout]; q|k|v stacked;
//    all time_emb_proj matrices stacked) so every hot load is a coalesced float4 row;
//  * activations live in one arena planned by liveness (skip connections stay resident until consumed);
//  * one denoising step {temb, UNet, scheduler epilogue, step++} is captured into a hipGraph and replayed
//    n_steps times; step-dependent scalars (timestep, scheduler coefficients, noise/mask slices) are read on the
//    device from a coefficient table indexed by a device-side step counter, so one graph serves every step.
#include <cmath>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "adm_kernels.h"

namespace adm {

struct ParamSlot {
  std::vector<long> shape;
  size_t numel = 0;
  float* dev = nullptr;
  bool set = false;
};
struct ConvW {
  float* wp = nullptr;
  float* bias = nullptr;
  int Cin = 0, Cout = 0, ks = 3;
};
struct GNW {
  float* gamma = nullptr;
  float* beta = nullptr;
  int C = 0;
};
struct Tensor {
  int C = 0, H = 0, W = 0;
  float* ptr = nullptr;
  int last_use = -1;
  bool external = false;
};
struct GnBuf {
  int C = 0;
  float* scale = nullptr;
  float* shift = nullptr;
};
struct Op {
  enum Kind { GN, CONV, ATTN } kind = CONV;
  int in1 = -1, in2 = -1, out = -1, res = -1, gn = -1;
  int up = 0, stride = 1, ks = 3, pad_lo = 1, act = 0;
  const ConvW* w = nullptr;
  const GNW* g = nullptr;
  int temb_off = -1;
  int head_dim = 0;
};

}  // namespace adm

using namespace adm;

struct adm_unet {
  adm_unet_config cfg;
  std::map<std::string, ParamSlot> params;
  std::deque<ConvW> convs;
  std::deque<GNW> gns;
  std::vector<Tensor> tensors;
  std::vector<GnBuf> gnbufs;
  std::vector<Op> ops;
  std::vector<void*> owned;  // device allocations freed on destroy
  bool finalized = false;
  int temb_dim = 0, temb_rows = 0;
  float *freqs = nullptr, *temb_w = nullptr, *temb_b = nullptr;
  int t_in = -1, t_out = -1;
  // per-batch plan
  int planned_B = 0;
  std::vector<void*> arena;
  size_t arena_bytes = 0;
  float *emb = nullptr, *temb_all = nullptr, *t_dev = nullptr, *eps_buf = nullptr;
  adm_sched_coef* coef_dev = nullptr;
  int coef_cap = 0;
  int* step_dev = nullptr;
#if !defined(ADM_EMU)
  hipStream_t own_stream = nullptr;
  hipGraphExec_t gexec = nullptr;
  std::vector<uint64_t> gkey;
#endif
};

namespace adm {

static void declare(adm_unet* h, const std::string& key, std::vector<long> shape) {
  ParamSlot s;
  s.shape = shape;
  s.numel = 1;
  for (long d : shape) s.numel *= (size_t)d;
  h->params[key] = s;
}
static void declare_conv(adm_unet* h, const std::string& p, int co, int ci, int ks) {
  declare(h, p + ".weight", {co, ci, ks, ks});
  declare(h, p + ".bias", {co});
}
static void declare_lin(adm_unet* h, const std::string& p, int co, int ci) {
  declare(h, p + ".weight", {co, ci});
  declare(h, p + ".bias", {co});
}
static void declare_gn(adm_unet* h, const std::string& p, int c) {
  declare(h, p + ".weight", {c});
  declare(h, p + ".bias", {c});
}
static void declare_resnet(adm_unet* h, const std::string& p, int ci, int co, int temb) {
  declare_gn(h, p + ".norm1", ci);
  declare_conv(h, p + ".conv1", co, ci, 3);
  declare_lin(h, p + ".time_emb_proj", co, temb);
  declare_gn(h, p + ".norm2", co);
  declare_conv(h, p + ".conv2", co, co, 3);
  if (ci != co) declare_conv(h, p + ".conv_shortcut", co, ci, 1);
}
static void declare_attn(adm_unet* h, const std::string& p, int c) {
  declare_gn(h, p + ".group_norm", c);
  declare_lin(h, p + ".to_q", c, c);
  declare_lin(h, p + ".to_k", c, c);
  declare_lin(h, p + ".to_v", c, c);
  declare_lin(h, p + ".to_out.0", c, c);
}

struct ResnetSpec { std::string prefix; int ci, co; };

// Enumerates parameters exactly as diffusers' UNet2DModel.__init__ does for this config.
static void declare_all(adm_unet* h) {
  const adm_unet_config& c = h->cfg;
  const int nb = c.n_blocks, L = c.layers_per_block;
  const int* boc = c.block_out_channels;
  const int temb = boc[0] * 4;
  h->temb_dim = temb;
  declare_conv(h, "conv_in", boc[0], c.in_channels, 3);
  declare_lin(h, "time_embedding.linear_1", temb, boc[0]);
  declare_lin(h, "time_embedding.linear_2", temb, temb);
  int out = boc[0];
  for (int i = 0; i < nb; ++i) {
    const int cin = out;
    out = boc[i];
    const std::string bp = "down_blocks." + std::to_string(i);
    for (int j = 0; j < L; ++j) {
      declare_resnet(h, bp + ".resnets." + std::to_string(j), j == 0 ? cin : out, out, temb);
      if (c.down_attn[i]) declare_attn(h, bp + ".attentions." + std::to_string(j), out);
    }
    if (i != nb - 1) declare_conv(h, bp + ".downsamplers.0.conv", out, out, 3);
  }
  const int mid = boc[nb - 1];
  declare_resnet(h, "mid_block.resnets.0", mid, mid, temb);
  declare_attn(h, "mid_block.attentions.0", mid);
  declare_resnet(h, "mid_block.resnets.1", mid, mid, temb);
  out = boc[nb - 1];
  for (int i = 0; i < nb; ++i) {
    const int prev = out;
    out = boc[nb - 1 - i];
    const int idx_in = nb - 1 - (i + 1 < nb - 1 ? i + 1 : nb - 1);
    const int cin = boc[idx_in];
    const std::string bp = "up_blocks." + std::to_string(i);
    for (int j = 0; j < L + 1; ++j) {
      const int skip = (j == L) ? cin : out;
      const int rin = (j == 0) ? prev : out;
      declare_resnet(h, bp + ".resnets." + std::to_string(j), rin + skip, out, temb);
      if (c.up_attn[i]) declare_attn(h, bp + ".attentions." + std::to_string(j), out);
    }
    if (i != nb - 1) declare_conv(h, bp + ".upsamplers.0.conv", out, out, 3);
  }
  declare_gn(h, "conv_norm_out", boc[0]);
  declare_conv(h, "conv_out", c.out_channels, boc[0], 3);
}

static int dalloc(adm_unet* h, void** p, size_t bytes) {
  ADM_TRY(dmalloc(p, bytes));
  h->owned.push_back(*p);
  return 0;
}

static float* P(adm_unet* h, const std::string& k) { return h->params.at(k).dev; }

static int make_conv(adm_unet* h, const std::string& p, int co, int ci, int ks, const ConvW** out) {
  ConvW w;
  w.Cin = ci; w.Cout = co; w.ks = ks;
  ADM_TRY(dalloc(h, (void**)&w.wp, sizeof(float) * (size_t)co * ci * ks * ks));
  ADM_TRY(launch_pack_conv_weight(P(h, p + ".weight"), w.wp, co, ci, ks, nullptr));
  w.bias = P(h, p + ".bias");
  h->convs.push_back(w);
  *out = &h->convs.back();
  return 0;
}
static const GNW* make_gn(adm_unet* h, const std::string& p, int c) {
  GNW g;
  g.gamma = P(h, p + ".weight"); g.beta = P(h, p + ".bias"); g.C = c;
  h->gns.push_back(g);
  return &h->gns.back();
}

struct Builder {
  adm_unet* h;
  std::vector<std::pair<std::string, int>> temb_rows;  // (time_emb_proj prefix, Cout) in op order
  int new_tensor(int C, int H, int W, bool ext = false) {
    Tensor t; t.C = C; t.H = H; t.W = W; t.external = ext;
    h->tensors.push_back(t);
    return (int)h->tensors.size() - 1;
  }
  int new_gn(int C) {
    GnBuf g; g.C = C;
    h->gnbufs.push_back(g);
    return (int)h->gnbufs.size() - 1;
  }
  int gn_op(int in1, int in2, const GNW* g) {
    Op o; o.kind = Op::GN; o.in1 = in1; o.in2 = in2; o.g = g; o.gn = new_gn(g->C);
    h->ops.push_back(o);
    return o.gn;
  }
  int conv_op(int in1, int in2, const ConvW* w, int gn, int act, int up, int stride, int res, int temb_off,
              int out_ext = -1) {
    const Tensor& ti = h->tensors[in1];
    int Ho, Wo;
    conv_out_dims(ti.H, ti.W, up, stride, w->ks, 1, &Ho, &Wo);
    Op o; o.kind = Op::CONV; o.in1 = in1; o.in2 = in2; o.w = w; o.gn = gn; o.act = act; o.up = up; o.stride = stride;
    o.ks = w->ks; o.pad_lo = w->ks == 3 ? 1 : 0; o.res = res; o.temb_off = temb_off;
    o.out = out_ext >= 0 ? out_ext : new_tensor(w->Cout, Ho, Wo);
    h->ops.push_back(o);
    return o.out;
  }
  int resnet(const std::string& p, int x1, int x2, int ci, int co, int* rc) {
    const ConvW *c1, *c2, *sc = nullptr;
    if ((*rc = make_conv(h, p + ".conv1", co, ci, 3, &c1))) return -1;
    if ((*rc = make_conv(h, p + ".conv2", co, co, 3, &c2))) return -1;
    if (ci != co && (*rc = make_conv(h, p + ".conv_shortcut", co, ci, 1, &sc))) return -1;
    int temb_off = 0;
    for (auto& r : temb_rows) temb_off += r.second;
    temb_rows.push_back({p + ".time_emb_proj", co});
    const int g1 = gn_op(x1, x2, make_gn(h, p + ".norm1", ci));
    const int hmid = conv_op(x1, x2, c1, g1, 1, 0, 1, -1, temb_off);
    const int g2 = gn_op(hmid, -1, make_gn(h, p + ".norm2", co));
    int res = x1;
    if (sc) res = conv_op(x1, x2, sc, -1, 0, 0, 1, -1, -1);
    return conv_op(hmid, -1, c2, g2, 1, 0, 1, res, -1);
  }
  int attention(const std::string& p, int x, int C, int* rc) {
    // q|k|v stacked into one 1x1 conv: weights (3C, C), bias 3C
    ConvW qkv; qkv.Cin = C; qkv.Cout = 3 * C; qkv.ks = 1;
    float* stacked = nullptr;
    if ((*rc = dalloc(h, (void**)&stacked, sizeof(float) * (size_t)3 * C * C))) return -1;
    if ((*rc = dalloc(h, (void**)&qkv.wp, sizeof(float) * (size_t)3 * C * C))) return -1;
    if ((*rc = dalloc(h, (void**)&qkv.bias, sizeof(float) * (size_t)3 * C))) return -1;
    const char* names[3] = {".to_q", ".to_k", ".to_v"};
    for (int i = 0; i < 3; ++i) {
      copy_d2d(stacked + (size_t)i * C * C, P(h, p + names[i] + ".weight"), sizeof(float) * (size_t)C * C, nullptr);
      copy_d2d(qkv.bias + (size_t)i * C, P(h, p + names[i] + ".bias"), sizeof(float) * (size_t)C, nullptr);
    }
    if ((*rc = launch_pack_conv_weight(stacked, qkv.wp, 3 * C, C, 1, nullptr))) return -1;
    h->convs.push_back(qkv);
    const ConvW* wqkv = &h->convs.back();
    const ConvW* wo;
    if ((*rc = make_conv(h, p + ".to_out.0", C, C, 1, &wo))) return -1;
    const int g = gn_op(x, -1, make_gn(h, p + ".group_norm", C));
    const int t_qkv = conv_op(x, -1, wqkv, g, 0, 0, 1, -1, -1);
    const Tensor tx = h->tensors[x];
    Op o; o.kind = Op::ATTN; o.in1 = t_qkv; o.head_dim = h->cfg.attention_head_dim > 0 ? h->cfg.attention_head_dim : C;
    o.out = new_tensor(C, tx.H, tx.W);
    h->ops.push_back(o);
    return conv_op(o.out, -1, wo, -1, 0, 0, 1, x, -1);
  }
};

static int finalize(adm_unet* h) {
  if (h->finalized) return 0;
  std::string missing;
  int nmiss = 0;
  for (auto& kv : h->params)
    if (!kv.second.set) { if (nmiss++ < 8) missing += kv.first + " "; }
  ADM_REQUIRE(nmiss == 0, "unet: " + std::to_string(nmiss) + " parameters not set: " + missing);
  const adm_unet_config& c = h->cfg;
  const int nb = c.n_blocks, L = c.layers_per_block;
  const int* boc = c.block_out_channels;
  Builder b{h};
  int rc = 0;
  h->t_in = b.new_tensor(c.in_channels, c.sample_h, c.sample_w, true);
  const ConvW* w;
  ADM_TRY(make_conv(h, "conv_in", boc[0], c.in_channels, 3, &w));
  int x = b.conv_op(h->t_in, -1, w, -1, 0, 0, 1, -1, -1);
  std::vector<int> skips{x};
  int out = boc[0];
  for (int i = 0; i < nb; ++i) {
    const int cin = out;
    out = boc[i];
    const std::string bp = "down_blocks." + std::to_string(i);
    for (int j = 0; j < L; ++j) {
      x = b.resnet(bp + ".resnets." + std::to_string(j), x, -1, j == 0 ? cin : out, out, &rc);
      ADM_TRY(rc);
      if (c.down_attn[i]) { x = b.attention(bp + ".attentions." + std::to_string(j), x, out, &rc); ADM_TRY(rc); }
      skips.push_back(x);
    }
    if (i != nb - 1) {
      ADM_TRY(make_conv(h, bp + ".downsamplers.0.conv", out, out, 3, &w));
      x = b.conv_op(x, -1, w, -1, 0, 0, 2, -1, -1);
      skips.push_back(x);
    }
  }
  const int mid = boc[nb - 1];
  x = b.resnet("mid_block.resnets.0", x, -1, mid, mid, &rc); ADM_TRY(rc);
  x = b.attention("mid_block.attentions.0", x, mid, &rc); ADM_TRY(rc);
  x = b.resnet("mid_block.resnets.1", x, -1, mid, mid, &rc); ADM_TRY(rc);
  out = boc[nb - 1];
  for (int i = 0; i < nb; ++i) {
    const int prev = out;
    out = boc[nb - 1 - i];
    const int cin = boc[nb - 1 - (i + 1 < nb - 1 ? i + 1 : nb - 1)];
    const std::string bp = "up_blocks." + std::to_string(i);
    for (int j = 0; j < L + 1; ++j) {
      const int skip = (j == L) ? cin : out;
      const int rin = (j == 0) ? prev : out;
      const int s = skips.back();
      skips.pop_back();
      ADM_REQUIRE(h->tensors[s].C == skip, "unet: skip channel mismatch at " + bp);
      x = b.resnet(bp + ".resnets." + std::to_string(j), x, s, rin + skip, out, &rc);
      ADM_TRY(rc);
      if (c.up_attn[i]) { x = b.attention(bp + ".attentions." + std::to_string(j), x, out, &rc); ADM_TRY(rc); }
    }
    if (i != nb - 1) {
      ADM_TRY(make_conv(h, bp + ".upsamplers.0.conv", out, out, 3, &w));
      x = b.conv_op(x, -1, w, -1, 0, 1, 1, -1, -1);
    }
  }
  const int g = b.gn_op(x, -1, make_gn(h, "conv_norm_out", boc[0]));
  ADM_TRY(make_conv(h, "conv_out", c.out_channels, boc[0], 3, &w));
  h->t_out = b.new_tensor(c.out_channels, c.sample_h, c.sample_w, true);
  b.conv_op(x, -1, w, g, 1, 0, 1, -1, -1, h->t_out);
  // stacked time_emb_proj
  int R = 0;
  for (auto& r : b.temb_rows) R += r.second;
  h->temb_rows = R;
  ADM_TRY(dalloc(h, (void**)&h->temb_w, sizeof(float) * (size_t)R * h->temb_dim));
  ADM_TRY(dalloc(h, (void**)&h->temb_b, sizeof(float) * (size_t)R));
  int off = 0;
  for (auto& r : b.temb_rows) {
    copy_d2d(h->temb_w + (size_t)off * h->temb_dim, P(h, r.first + ".weight"), sizeof(float) * (size_t)r.second * h->temb_dim, nullptr);
    copy_d2d(h->temb_b + off, P(h, r.first + ".bias"), sizeof(float) * (size_t)r.second, nullptr);
    off += r.second;
  }
  // sinusoid frequencies: exp(-ln(10000) * i / (half - freq_shift)) in fp32, as diffusers computes them
  const int half = boc[0] / 2;
  std::vector<float> fr(half);
  for (int i = 0; i < half; ++i) {
    float e = -logf(10000.0f) * (float)i;
    e = e / ((float)half - c.freq_shift);
    fr[i] = expf(e);
  }
  ADM_TRY(dalloc(h, (void**)&h->freqs, sizeof(float) * half));
  ADM_TRY(copy_h2d(h->freqs, fr.data(), sizeof(float) * half, nullptr));
  ADM_TRY(stream_sync(nullptr));
  // liveness
  for (size_t i = 0; i < h->ops.size(); ++i) {
    const Op& o = h->ops[i];
    for (int t : {o.in1, o.in2, o.res})
      if (t >= 0) h->tensors[t].last_use = (int)i;
  }
  h->finalized = true;
  return 0;
}

static void free_plan(adm_unet* h) {
  for (void* p : h->arena) dfree(p);
  h->arena.clear();
  h->arena_bytes = 0;
  h->planned_B = 0;
#if !defined(ADM_EMU)
  if (h->gexec) { (void)hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
  h->gkey.clear();
#endif
}

static int arena_alloc(adm_unet* h, void** p, size_t bytes) {
  ADM_TRY(dmalloc(p, bytes));
  h->arena.push_back(*p);
  h->arena_bytes += bytes;
  return 0;
}

// Assign activation buffers for batch B: exact-size free lists driven by liveness.
static int plan(adm_unet* h, int B) {
  if (h->planned_B == B) return 0;
  free_plan(h);
  std::multimap<size_t, float*> freelist;
  std::vector<std::vector<int>> dying(h->ops.size());
  for (size_t t = 0; t < h->tensors.size(); ++t)
    if (!h->tensors[t].external && h->tensors[t].last_use >= 0) dying[h->tensors[t].last_use].push_back((int)t);
  for (size_t i = 0; i < h->ops.size(); ++i) {
    const Op& o = h->ops[i];
    if (o.out >= 0 && !h->tensors[o.out].external) {
      Tensor& t = h->tensors[o.out];
      const size_t bytes = sizeof(float) * (size_t)B * t.C * t.H * t.W;
      auto it = freelist.find(bytes);
      if (it != freelist.end()) { t.ptr = it->second; freelist.erase(it); }
      else ADM_TRY(arena_alloc(h, (void**)&t.ptr, bytes));
    }
    for (int t : dying[i]) {
      const Tensor& tt = h->tensors[t];
      freelist.insert({sizeof(float) * (size_t)B * tt.C * tt.H * tt.W, tt.ptr});
    }
  }
  for (GnBuf& g : h->gnbufs) {
    ADM_TRY(arena_alloc(h, (void**)&g.scale, sizeof(float) * (size_t)B * g.C));
    ADM_TRY(arena_alloc(h, (void**)&g.shift, sizeof(float) * (size_t)B * g.C));
  }
  ADM_TRY(arena_alloc(h, (void**)&h->emb, sizeof(float) * (size_t)B * h->temb_dim));
  ADM_TRY(arena_alloc(h, (void**)&h->temb_all, sizeof(float) * (size_t)B * h->temb_rows));
  ADM_TRY(arena_alloc(h, (void**)&h->t_dev, sizeof(float) * (size_t)B));
  ADM_TRY(arena_alloc(h, (void**)&h->eps_buf,
                      sizeof(float) * (size_t)B * h->cfg.out_channels * h->cfg.sample_h * h->cfg.sample_w));
  ADM_TRY(arena_alloc(h, (void**)&h->step_dev, sizeof(int)));
  h->planned_B = B;
  return 0;
}

struct OpTimer {  // optional per-op HIP-event timing (adm_unet_profile); disabled (null) on the product path
  std::vector<adm_op_profile>* recs = nullptr;
#if !defined(ADM_EMU)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;
#endif
  hipStream_t st = nullptr;
  void begin() {
#if !defined(ADM_EMU)
    if (!recs) return;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a, st);
    evs.push_back({a, b});
#endif
  }
  void end(int kind, int variant, double flops, double bytes) {
    if (!recs) return;
#if !defined(ADM_EMU)
    (void)hipEventRecord(evs.back().second, st);
#endif
    adm_op_profile r; r.kind = kind; r.variant = variant; r.ms = 0.f; r.flops = flops; r.bytes = bytes;
    recs->push_back(r);
  }
  void finish() {
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
};

// Enqueue one UNet forward. Timestep source: t_dev (B floats) when table == nullptr, else table[*step_dev].
static int run_forward(adm_unet* h, const float* x, float* out, int B, const adm_sched_coef* table, hipStream_t st,
                       OpTimer* tm = nullptr) {
  OpTimer none;
  if (!tm) tm = &none;
  tm->st = st;
  const adm_unet_config& c = h->cfg;
  h->tensors[h->t_in].ptr = const_cast<float*>(x);
  h->tensors[h->t_out].ptr = out;
  const int dim_in = c.block_out_channels[0];
  ADM_TRY(launch_time_embedding(table ? nullptr : h->t_dev, 1, table, h->step_dev, h->freqs, dim_in / 2,
                                c.flip_sin_to_cos, P(h, "time_embedding.linear_1.weight"),
                                P(h, "time_embedding.linear_1.bias"), P(h, "time_embedding.linear_2.weight"),
                                P(h, "time_embedding.linear_2.bias"), dim_in, h->temb_dim, h->emb, B, st));
  tm->begin();
  ADM_TRY(launch_temb_proj(h->emb, h->temb_w, h->temb_b, h->temb_all, B, h->temb_dim, h->temb_rows, st));
  tm->end(4, 0, 2.0 * B * h->temb_dim * h->temb_rows, 4.0 * h->temb_dim * h->temb_rows);
  for (const Op& o : h->ops) {
    const Tensor& t1 = h->tensors[o.in1];
    tm->begin();
    if (o.kind == Op::GN) {
      const GnBuf& g = h->gnbufs[o.gn];
      const float* x2 = o.in2 >= 0 ? h->tensors[o.in2].ptr : nullptr;
      const int C2 = o.in2 >= 0 ? h->tensors[o.in2].C : 0;
      ADM_TRY(launch_groupnorm_stats(t1.ptr, t1.C, x2, C2, B, t1.H * t1.W, c.norm_num_groups, c.norm_eps, o.g->gamma,
                                     o.g->beta, g.scale, g.shift, st));
      tm->end(0, 0, 3.0 * B * (t1.C + C2) * t1.H * t1.W, 4.0 * B * (t1.C + C2) * t1.H * t1.W);
    } else if (o.kind == Op::CONV) {
      adm_conv_args a;
      memset(&a, 0, sizeof(a));
      a.x1 = t1.ptr; a.C1 = t1.C;
      if (o.in2 >= 0) { a.x2 = h->tensors[o.in2].ptr; a.C2 = h->tensors[o.in2].C; }
      a.N = B; a.H = t1.H; a.W = t1.W;
      a.up = o.up; a.stride = o.stride; a.ks = o.ks; a.pad_lo = o.pad_lo;
      if (o.gn >= 0) { a.gn_scale = h->gnbufs[o.gn].scale; a.gn_shift = h->gnbufs[o.gn].shift; }
      a.act = o.act;
      a.wpacked = o.w->wp; a.bias = o.w->bias; a.Cout = o.w->Cout;
      if (o.temb_off >= 0) { a.chan_add = h->temb_all + o.temb_off; a.chan_add_stride = h->temb_rows; }
      if (o.res >= 0) a.residual = h->tensors[o.res].ptr;
      a.out = h->tensors[o.out].ptr;
      ADM_TRY(launch_conv2d(a, st));
      const Tensor& to = h->tensors[o.out];
      const double Cin = a.C1 + a.C2, outel = (double)B * to.C * to.H * to.W;
      tm->end((last_conv_variant() >= 1000 && last_conv_variant() < 2000) ? 3 : 1, last_conv_variant(), 2.0 * outel * Cin * o.ks * o.ks,
              4.0 * ((double)B * Cin * t1.H * t1.W + outel * (o.res >= 0 ? 2 : 1) + (double)to.C * Cin * o.ks * o.ks));
    } else {
      const int C = t1.C / 3, T = t1.H * t1.W;
      ADM_TRY(launch_attention(t1.ptr, h->tensors[o.out].ptr, B, C, T, o.head_dim, st));
      tm->end(2, o.head_dim, 4.0 * B * C * (double)T * T, 16.0 * B * C * T);
    }
  }
  tm->finish();
  return 0;
}

static int ensure_coef(adm_unet* h, const adm_sched_coef* coef_host, int n, hipStream_t st) {
  if (h->coef_cap < n) {
    ADM_TRY(dalloc(h, (void**)&h->coef_dev, sizeof(adm_sched_coef) * (size_t)n));
    h->coef_cap = n;
  }
  ADM_TRY(copy_h2d(h->coef_dev, coef_host, sizeof(adm_sched_coef) * (size_t)n, st));
  ADM_TRY(dmemset(h->step_dev, 0, sizeof(int), st));
  return 0;
}

struct LoopArgs {
  float* x; int B; int n_steps; const float* step_noise; const float* mask; int mask_start, mask_end;
  uint8_t* u8; int encode;
};

// One denoising step; every step-dependent scalar is read on the device through *step_dev.
static int enqueue_step(adm_unet* h, const LoopArgs& a, int step, hipStream_t st) {
  const adm_unet_config& c = h->cfg;
  const adm_sched_coef* table = h->coef_dev;
  ADM_TRY(run_forward(h, a.x, h->eps_buf, a.B, table, st));
  const long n = (long)a.B * c.in_channels * c.sample_h * c.sample_w;
  if (a.encode) {
    ADM_TRY(launch_encode_step(a.x, h->eps_buf, table, h->step_dev, step, n, st));
  } else {
    ADM_TRY(launch_sched_step_loop(a.x, h->eps_buf, a.step_noise, n, a.x, a.u8, a.n_steps - 1, table, h->step_dev, step,
                                   a.mask, a.n_steps, a.mask_start, a.mask_end, a.B, c.in_channels, c.sample_h,
                                   c.sample_w, st));
  }
  ADM_TRY(launch_step_advance(h->step_dev, st));
  return 0;
}

static int run_loop(adm_unet* h, const LoopArgs& a, const adm_sched_coef* coef_host, int use_graph, hipStream_t st) {
  ADM_TRY(finalize(h));
  ADM_REQUIRE(h->cfg.in_channels == h->cfg.out_channels, "sample_loop: in/out channels differ");
  ADM_TRY(plan(h, a.B));
#if defined(ADM_EMU)
  (void)use_graph;
  ADM_TRY(ensure_coef(h, coef_host, a.n_steps, st));
  for (int s = 0; s < a.n_steps; ++s) ADM_TRY(enqueue_step(h, a, s, st));
  return 0;
#else
  hipStream_t run = st;
  hipEvent_t ev = nullptr;
  if (use_graph && st == nullptr) {  // the legacy default stream cannot be captured: hop onto an owned stream
    if (!h->own_stream) ADM_HIP_OK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    run = h->own_stream;
    ADM_HIP_OK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    ADM_HIP_OK(hipEventRecord(ev, st));
    ADM_HIP_OK(hipStreamWaitEvent(run, ev, 0));
  }
  ADM_TRY(ensure_coef(h, coef_host, a.n_steps, run));
  if (!use_graph) {
    for (int s = 0; s < a.n_steps; ++s) ADM_TRY(enqueue_step(h, a, s, run));
  } else {
    std::vector<uint64_t> key = {(uint64_t)a.x, (uint64_t)a.B, (uint64_t)a.n_steps, (uint64_t)a.step_noise,
                                 (uint64_t)a.mask, (uint64_t)a.mask_start, (uint64_t)a.mask_end, (uint64_t)a.u8,
                                 (uint64_t)a.encode, (uint64_t)h->coef_dev, (uint64_t)run};
    if (!h->gexec || key != h->gkey) {
      if (h->gexec) { (void)hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
      hipGraph_t graph = nullptr;
      ADM_HIP_OK(hipStreamBeginCapture(run, hipStreamCaptureModeThreadLocal));
      int rc = enqueue_step(h, a, 0, run);
      hipError_t e = hipStreamEndCapture(run, &graph);
      if (rc != 0) { if (graph) (void)hipGraphDestroy(graph); return rc; }
      ADM_HIP_OK(e);
      ADM_HIP_OK(hipGraphInstantiate(&h->gexec, graph, nullptr, nullptr, 0));
      ADM_HIP_OK(hipGraphDestroy(graph));
      h->gkey = key;
    }
    for (int s = 0; s < a.n_steps; ++s) ADM_HIP_OK(hipGraphLaunch(h->gexec, run));
  }
  if (ev) {
    ADM_HIP_OK(hipEventRecord(ev, run));
    ADM_HIP_OK(hipStreamWaitEvent(st, ev, 0));
    ADM_HIP_OK(hipEventDestroy(ev));
  }
  return 0;
#endif
}

}  // namespace adm

extern "C" {

int adm_unet_create(const adm_unet_config* cfg, adm_unet_t** out) {
  ADM_REQUIRE(cfg && out, "unet_create: null argument");
  ADM_REQUIRE(cfg->n_blocks >= 1 && cfg->n_blocks <= 8, "unet_create: n_blocks out of range");
  ADM_REQUIRE(cfg->norm_num_groups > 0 && cfg->layers_per_block >= 1, "unet_create: bad config");
  for (int i = 0; i < cfg->n_blocks; ++i)
    ADM_REQUIRE(cfg->block_out_channels[i] % cfg->norm_num_groups == 0 && cfg->block_out_channels[i] % 32 == 0,
                "unet_create: block_out_channels must be multiples of 32 and of norm_num_groups");
  adm_unet* h = new adm_unet();
  h->cfg = *cfg;
  declare_all(h);
  *out = h;
  return 0;
}

void adm_unet_destroy(adm_unet_t* h) {
  if (!h) return;
  free_plan(h);
  for (auto& kv : h->params)
    if (kv.second.dev) dfree(kv.second.dev);
  for (void* p : h->owned) dfree(p);
#if !defined(ADM_EMU)
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
#endif
  delete h;
}

int adm_unet_set_param(adm_unet_t* h, const char* key, const float* host_data, size_t numel) {
  ADM_REQUIRE(h && key && host_data, "unet_set_param: null argument");
  ADM_REQUIRE(!h->finalized, "unet_set_param: model already finalized (first forward ran)");
  std::string k(key);
  static const char* oldn[4] = {".query.", ".key.", ".value.", ".proj_attn."};
  static const char* newn[4] = {".to_q.", ".to_k.", ".to_v.", ".to_out.0."};
  if (k.find(".attentions.") != std::string::npos)
    for (int i = 0; i < 4; ++i) {
      size_t pos = k.find(oldn[i]);
      if (pos != std::string::npos) k.replace(pos, strlen(oldn[i]), newn[i]);
    }
  auto it = h->params.find(k);
  ADM_REQUIRE(it != h->params.end(), "unet_set_param: unexpected key " + k);
  ADM_REQUIRE(it->second.numel == numel, "unet_set_param: size mismatch for " + k);
  if (!it->second.dev) ADM_TRY(dmalloc((void**)&it->second.dev, sizeof(float) * numel));
  ADM_TRY(copy_h2d(it->second.dev, host_data, sizeof(float) * numel, nullptr));
  ADM_TRY(stream_sync(nullptr));
  it->second.set = true;
  return 0;
}

int adm_unet_missing_params(adm_unet_t* h) {
  int n = 0;
  std::string names;
  for (auto& kv : h->params)
    if (!kv.second.set) { ++n; if (n <= 16) names += kv.first + " "; }
  if (n) set_error("missing: " + names);
  return n;
}

int adm_unet_forward(adm_unet_t* h, const float* x, const float* timesteps_host, int n_timesteps, float* out, int B,
                     void* stream) {
  ADM_REQUIRE(h && x && out && timesteps_host, "unet_forward: null argument");
  ADM_REQUIRE(n_timesteps == 1 || n_timesteps == B, "unet_forward: need 1 or B timesteps");
  hipStream_t st = (hipStream_t)stream;
  ADM_TRY(finalize(h));
  ADM_TRY(plan(h, B));
  std::vector<float> t(B);
  for (int i = 0; i < B; ++i) t[i] = timesteps_host[n_timesteps == 1 ? 0 : i];
  ADM_TRY(copy_h2d(h->t_dev, t.data(), sizeof(float) * B, st));
  ADM_TRY(stream_sync(st));  // t is a stack/vector buffer: make the copy complete before returning
  return run_forward(h, x, out, B, nullptr, st);
}

size_t adm_unet_workspace_bytes(adm_unet_t* h) { return h ? h->arena_bytes : 0; }

int adm_unet_profile(adm_unet_t* h, const float* x, float timestep, float* out, int B, adm_op_profile* recs, int cap,
                     int* n_out, void* stream) {
  ADM_REQUIRE(h && x && out && recs && n_out, "unet_profile: null argument");
  hipStream_t st = (hipStream_t)stream;
  ADM_TRY(finalize(h));
  ADM_TRY(plan(h, B));
  std::vector<float> t(B, timestep);
  ADM_TRY(copy_h2d(h->t_dev, t.data(), sizeof(float) * B, st));
  ADM_TRY(stream_sync(st));
  std::vector<adm_op_profile> v;
  OpTimer tm;
  tm.recs = &v;
  ADM_TRY(run_forward(h, x, out, B, nullptr, st, &tm));
  *n_out = (int)v.size();
  for (int i = 0; i < (int)v.size() && i < cap; ++i) recs[i] = v[i];
  return 0;
}

int adm_sample_loop(adm_unet_t* h, float* x, int B, const adm_sched_coef* coef_host, int n_steps,
                    const float* step_noise, const float* mask, int mask_start, int mask_end, uint8_t* u8_out,
                    int use_graph, void* stream) {
  ADM_REQUIRE(h && x && coef_host && n_steps > 0, "sample_loop: bad argument");
  LoopArgs a{x, B, n_steps, step_noise, mask, mask_start, mask_end, u8_out, 0};
  return run_loop(h, a, coef_host, use_graph, (hipStream_t)stream);
}

int adm_encode_loop(adm_unet_t* h, float* x, int B, const adm_sched_coef* coef_host, int n_steps, int use_graph,
                    void* stream) {
  ADM_REQUIRE(h && x && coef_host && n_steps > 0, "encode_loop: bad argument");
  LoopArgs a{x, B, n_steps, nullptr, nullptr, 0, 0, nullptr, 1};
  return run_loop(h, a, coef_host, use_graph, (hipStream_t)stream);
}

}  // extern "C"
