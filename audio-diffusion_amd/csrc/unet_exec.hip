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
#include <string>
#include <vector>

#include "net_exec.h"

extern "C" int adm_mse_loss(const float* pred, const float* target, long n, float* loss_out, float* grad_out,
                            double* scratch, void* stream);

using namespace adm;

struct adm_unet {
  adm_unet_config cfg;
  ParamStore ps;
  Net net;
  std::vector<void*> owned;  // device allocations freed on destroy
  bool finalized = false;
  int temb_dim = 0, temb_rows = 0;
  float *freqs = nullptr, *temb_w = nullptr, *temb_b = nullptr;
  // per-batch plan
  int planned_B = 0;
  std::vector<void*> extra;  // per-batch buffers besides the net arena
  PackItem* temb_items = nullptr; int n_temb_items = 0;   // restack_temb's device table
  float *emb = nullptr, *emb_act = nullptr, *temb_all = nullptr, *t_dev = nullptr, *eps_buf = nullptr;
  adm_sched_coef* coef_dev = nullptr;
  int coef_cap = 0;
  int* step_dev = nullptr;
  // training
  bool training = false;
  int warm_B = 0;         // batch size at which an uncaptured forward has already run (see run_loop)
  int bf16_level = 0;     // mixed-precision level of THIS model (option conv_bf16 as it stood at adm_unet_enable_training)
  int op16_f16 = 0;       // ... and the operand format of its 16-bit kernels (option conv_op16_f16: fp16 instead of bf16)
  float loss_scale = 1.f; // fp16 loss scaling (adm_unet_set_loss_scale)
  long params_numel = 0;
  float *dtemb_all = nullptr, *demb = nullptr, *dz = nullptr, *save_sinus = nullptr, *save_z = nullptr;
  double* scratch_d = nullptr;
#if !defined(ADM_EMU)
  hipStream_t own_stream = nullptr;
  hipGraphExec_t gexec = nullptr;
  hipStream_t gstream = nullptr;     // the stream gexec was captured on and is replayed on
  std::vector<uint64_t> gkey;
#endif
};

namespace adm {

// Enumerates parameters exactly as diffusers' UNet2DModel.__init__ does for this config.
static void declare_all(adm_unet* h) {
  const adm_unet_config& c = h->cfg;
  ParamStore& ps = h->ps;
  const int nb = c.n_blocks, L = c.layers_per_block;
  const int* boc = c.block_out_channels;
  const int temb = boc[0] * 4;
  h->temb_dim = temb;
  ps.declare_conv("conv_in", boc[0], c.in_channels, 3);
  ps.declare_lin("time_embedding.linear_1", temb, boc[0]);
  ps.declare_lin("time_embedding.linear_2", temb, temb);
  int out = boc[0];
  for (int i = 0; i < nb; ++i) {
    const int cin = out;
    out = boc[i];
    const std::string bp = "down_blocks." + std::to_string(i);
    for (int j = 0; j < L; ++j) {
      ps.declare_resnet(bp + ".resnets." + std::to_string(j), j == 0 ? cin : out, out, temb);
      if (c.down_attn[i] == 2) ps.declare_transformer(bp + ".attentions." + std::to_string(j), out, c.cross_attention_dim);
      else if (c.down_attn[i]) ps.declare_attn(bp + ".attentions." + std::to_string(j), out);
    }
    if (i != nb - 1) ps.declare_conv(bp + ".downsamplers.0.conv", out, out, 3);
  }
  const int mid = boc[nb - 1];
  ps.declare_resnet("mid_block.resnets.0", mid, mid, temb);
  if (c.cross_attention_dim > 0) ps.declare_transformer("mid_block.attentions.0", mid, c.cross_attention_dim);
  else ps.declare_attn("mid_block.attentions.0", mid);
  ps.declare_resnet("mid_block.resnets.1", mid, mid, temb);
  out = boc[nb - 1];
  for (int i = 0; i < nb; ++i) {
    const int prev = out;
    out = boc[nb - 1 - i];
    const int cin = boc[nb - 1 - (i + 1 < nb - 1 ? i + 1 : nb - 1)];
    const std::string bp = "up_blocks." + std::to_string(i);
    for (int j = 0; j < L + 1; ++j) {
      const int skip = (j == L) ? cin : out;
      const int rin = (j == 0) ? prev : out;
      ps.declare_resnet(bp + ".resnets." + std::to_string(j), rin + skip, out, temb);
      if (c.up_attn[i] == 2) ps.declare_transformer(bp + ".attentions." + std::to_string(j), out, c.cross_attention_dim);
      else if (c.up_attn[i]) ps.declare_attn(bp + ".attentions." + std::to_string(j), out);
    }
    if (i != nb - 1) ps.declare_conv(bp + ".upsamplers.0.conv", out, out, 3);
  }
  ps.declare_gn("conv_norm_out", boc[0]);
  ps.declare_conv("conv_out", c.out_channels, boc[0], 3);
}

static int dalloc(adm_unet* h, void** p, size_t bytes) {
  ADM_TRY(dmalloc(p, bytes));
  h->owned.push_back(*p);
  return 0;
}

static float* P(adm_unet* h, const std::string& k) { return h->ps.P(k); }

// stacked copy of all time_emb_proj matrices (re-done after every optimizer step in training)
// the 32 time_emb_proj matrices (+ biases) stacked row-wise: ONE batched copy launch over a device table built once (the
// parameter pointers are stable: master parameters never move after finalize) instead of 64 copies per optimizer step
static int restack_temb(adm_unet* h, hipStream_t st) {
  if (h->temb_items == nullptr) {
    std::vector<PackItem> items;
    int off = 0;
    for (auto& r : h->net.temb_rows) {
      items.push_back(PackItem{P(h, r.first + ".weight"), h->temb_w + (size_t)off * h->temb_dim, 0, 0, 0, r.second * h->temb_dim});
      items.push_back(PackItem{P(h, r.first + ".bias"), h->temb_b + off, 0, 0, 0, r.second});
      off += r.second;
    }
    h->n_temb_items = (int)items.size();
    if (items.empty()) return 0;
    ADM_TRY(h->net.dalloc((void**)&h->temb_items, sizeof(PackItem) * items.size()));
    ADM_TRY(copy_h2d(h->temb_items, items.data(), sizeof(PackItem) * items.size(), st));
    ADM_TRY(stream_sync(st));
  }
  return launch_copy_batch(h->temb_items, h->n_temb_items, st);
}

// The conv dispatchers read the process-wide option conv_bf16; a model's own level is put in force only while one of ITS
// training entry points runs (and while its weights are packed), and every inference entry point runs at level 0 — so the
// sampling path stays fp32 even through a training handle, and one model's setting never leaks into another's.
struct Bf16Scope {
  int prev, prev_fmt;
  explicit Bf16Scope(int level, int f16 = 0) : prev(conv_bf16_mode()), prev_fmt(conv_op16_f16() ? 1 : 0) {
    set_conv_bf16(level);
    set_conv_op16_f16(f16);
  }
  ~Bf16Scope() { set_conv_bf16(prev); set_conv_op16_f16(prev_fmt); }
};

// Inference through a training handle: packings refreshed in full first (begin_inference), training's usage masks restored after.
struct InferenceScope {
  Net* n; std::vector<unsigned> saved; int rc;
  InferenceScope(Net* net, hipStream_t st) : n(net) { rc = n->begin_inference(st, &saved); }
  ~InferenceScope() { if (rc == 0) n->end_inference(saved); }
};

static int finalize(adm_unet* h) {
  Bf16Scope pack_scope(h->training ? h->bf16_level : 0, h->training ? h->op16_f16 : 0);
  if (h->finalized) return 0;
  std::string missing;
  const int nmiss = h->ps.missing(&missing);
  ADM_REQUIRE(nmiss == 0, "unet: " + std::to_string(nmiss) + " parameters not set: " + missing);
  const adm_unet_config& c = h->cfg;
  const int nb = c.n_blocks, L = c.layers_per_block;
  const int* boc = c.block_out_channels;
  Net& b = h->net;
  b.ps = &h->ps;
  b.groups = c.norm_num_groups;
  b.eps = c.norm_eps;
  b.training = h->training;
  int rc = 0;
  b.t_in = b.new_tensor(c.in_channels, c.sample_h, c.sample_w, true);
  const ConvW* w;
  ADM_TRY(b.make_conv("conv_in", boc[0], c.in_channels, 3, &w));
  int x = b.conv_op(b.t_in, -1, w, -1, 0, 0, 1, 1, -1, -1);
  std::vector<int> skips{x};
  int out = boc[0];
  const int hd = c.attention_head_dim > 0 ? c.attention_head_dim : 0;
  for (int i = 0; i < nb; ++i) {
    const int cin = out;
    out = boc[i];
    const std::string bp = "down_blocks." + std::to_string(i);
    for (int j = 0; j < L; ++j) {
      x = b.resnet(bp + ".resnets." + std::to_string(j), x, -1, j == 0 ? cin : out, out, true, &rc);
      ADM_TRY(rc);
      if (c.down_attn[i] == 2) { x = b.transformer(bp + ".attentions." + std::to_string(j), x, out, hd, c.cross_attention_dim, &rc); ADM_TRY(rc); }
      else if (c.down_attn[i]) { x = b.attention(bp + ".attentions." + std::to_string(j), x, out, hd ? hd : out, &rc); ADM_TRY(rc); }
      skips.push_back(x);
    }
    if (i != nb - 1) {
      ADM_TRY(b.make_conv(bp + ".downsamplers.0.conv", out, out, 3, &w));
      x = b.conv_op(x, -1, w, -1, 0, 0, 2, 1, -1, -1);
      skips.push_back(x);
    }
  }
  const int mid = boc[nb - 1];
  x = b.resnet("mid_block.resnets.0", x, -1, mid, mid, true, &rc); ADM_TRY(rc);
  if (c.cross_attention_dim > 0) x = b.transformer("mid_block.attentions.0", x, mid, hd, c.cross_attention_dim, &rc);
  else x = b.attention("mid_block.attentions.0", x, mid, hd ? hd : mid, &rc);
  ADM_TRY(rc);
  x = b.resnet("mid_block.resnets.1", x, -1, mid, mid, true, &rc); ADM_TRY(rc);
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
      ADM_REQUIRE(b.tensors[s].C == skip, "unet: skip channel mismatch at " + bp);
      x = b.resnet(bp + ".resnets." + std::to_string(j), x, s, rin + skip, out, true, &rc);
      ADM_TRY(rc);
      if (c.up_attn[i] == 2) { x = b.transformer(bp + ".attentions." + std::to_string(j), x, out, hd, c.cross_attention_dim, &rc); ADM_TRY(rc); }
      else if (c.up_attn[i]) { x = b.attention(bp + ".attentions." + std::to_string(j), x, out, hd ? hd : out, &rc); ADM_TRY(rc); }
    }
    if (i != nb - 1) {
      ADM_TRY(b.make_conv(bp + ".upsamplers.0.conv", out, out, 3, &w));
      x = b.conv_op(x, -1, w, -1, 0, 1, 1, 1, -1, -1);
    }
  }
  const int g = b.gn_op(x, -1, b.make_gn("conv_norm_out", boc[0]));
  ADM_TRY(b.make_conv("conv_out", c.out_channels, boc[0], 3, &w));
  b.t_out = b.new_tensor(c.out_channels, c.sample_h, c.sample_w, true);
  b.conv_op(x, -1, w, g, 1, 0, 1, 1, -1, -1, b.t_out);
  // stacked time_emb_proj
  int R = 0;
  for (auto& r : b.temb_rows) R += r.second;
  h->temb_rows = R;
  ADM_TRY(dalloc(h, (void**)&h->temb_w, sizeof(float) * (size_t)R * h->temb_dim));
  ADM_TRY(dalloc(h, (void**)&h->temb_b, sizeof(float) * (size_t)R));
  ADM_TRY(restack_temb(h, nullptr));
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
  b.finish_liveness();
  h->finalized = true;
  return 0;
}

static void free_plan(adm_unet* h) {
  h->net.free_plan();
  for (void* p : h->extra) dfree(p);
  h->extra.clear();
  h->planned_B = 0;
  h->warm_B = 0;       // the next capture is preceded by an uncaptured forward again (another kernel's one-time set-up)
#if !defined(ADM_EMU)
  if (h->gexec) {                     // (a replay may still be in flight: drain the stream it ran on before the executable graph goes)
    (void)stream_sync(h->gstream);
    (void)hipGraphExecDestroy(h->gexec);
    h->gexec = nullptr;
  }
  h->gkey.clear();
#endif
}

static int extra_alloc(adm_unet* h, void** p, size_t bytes) {
  ADM_TRY(dmalloc(p, bytes));
  h->extra.push_back(*p);
  return 0;
}

static int plan(adm_unet* h, int B) {
  // (re-planned as well when adm_set_option has moved since: a layer may have changed kernel, and the partial-sum buffers of the
  //  GroupNorm statistics follow the kernels — free_plan also drops the captured graph, which holds the old kernels)
  if (h->planned_B == B && h->net.plan_current(B)) return 0;
  free_plan(h);
  ADM_TRY(h->net.plan(B));
  ADM_TRY(extra_alloc(h, (void**)&h->emb, sizeof(float) * (size_t)B * h->temb_dim));
  ADM_TRY(extra_alloc(h, (void**)&h->emb_act, sizeof(float) * (size_t)B * h->temb_dim));
  ADM_TRY(extra_alloc(h, (void**)&h->temb_all, sizeof(float) * (size_t)B * h->temb_rows));
  ADM_TRY(extra_alloc(h, (void**)&h->t_dev, sizeof(float) * (size_t)B));
  ADM_TRY(extra_alloc(h, (void**)&h->eps_buf,
                      sizeof(float) * (size_t)B * h->cfg.out_channels * h->cfg.sample_h * h->cfg.sample_w));
  ADM_TRY(extra_alloc(h, (void**)&h->step_dev, sizeof(int)));
  if (h->training) {
    ADM_TRY(extra_alloc(h, (void**)&h->dtemb_all, sizeof(float) * (size_t)B * h->temb_rows));
    ADM_TRY(extra_alloc(h, (void**)&h->demb, sizeof(float) * (size_t)B * h->temb_dim));
    ADM_TRY(extra_alloc(h, (void**)&h->dz, sizeof(float) * (size_t)B * h->temb_dim));
    ADM_TRY(extra_alloc(h, (void**)&h->save_sinus, sizeof(float) * (size_t)B * h->cfg.block_out_channels[0]));
    ADM_TRY(extra_alloc(h, (void**)&h->save_z, sizeof(float) * (size_t)B * h->temb_dim));
    ADM_TRY(extra_alloc(h, (void**)&h->scratch_d, sizeof(double)));
  }
  h->planned_B = B;
  return 0;
}

// Enqueue one UNet forward. Timestep source: t_dev (B floats) when table == nullptr, else table[*step_dev].
static int run_forward(adm_unet* h, const float* x, float* out, int B, const adm_sched_coef* table, hipStream_t st,
                       OpTimer* tm = nullptr) {
  const adm_unet_config& c = h->cfg;
  const int dim_in = c.block_out_channels[0];
  ADM_TRY(launch_time_embedding(table ? nullptr : h->t_dev, 1, table, h->step_dev, h->freqs, dim_in / 2,
                                c.flip_sin_to_cos, P(h, "time_embedding.linear_1.weight"),
                                P(h, "time_embedding.linear_1.bias"), P(h, "time_embedding.linear_2.weight"),
                                P(h, "time_embedding.linear_2.bias"), dim_in, h->temb_dim, h->emb, B, st,
                                h->training ? h->save_sinus : nullptr, h->training ? h->save_z : nullptr, h->emb_act));
  if (tm) { tm->st = st; tm->begin(); }
  ADM_TRY(launch_temb_proj(h->emb_act, h->temb_w, h->temb_b, h->temb_all, B, h->temb_dim, h->temb_rows, st, 1));
  if (tm) tm->end(4, 0, 2.0 * B * h->temb_dim * h->temb_rows, 4.0 * h->temb_dim * h->temb_rows);
  return h->net.run(x, out, B, h->temb_all, h->temb_rows, st, tm);
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
                                 (uint64_t)a.encode, (uint64_t)h->coef_dev, (uint64_t)run, (uint64_t)h->net.ctx,
                                 (uint64_t)h->net.ctx_S};
    if (!h->gexec || key != h->gkey) {
      if (h->gexec) {
        // the previous loop's replays may still be running (a caller that samples again without a host synchronisation in between — 50
        // single-step calls in tests/test_pipeline.py): destroying an executable graph under its own replays made that test die with SIGSEGV
        // inside this call — 4 of 11 runs of test_full_size.py + test_pipeline.py without the drain, 0 of 12 with it.
        // Re-capture is the rare path; a drain costs nothing there.
        static const int drain = [] { const char* e = getenv("ADM_GRAPH_DRAIN"); return e ? atoi(e) : 1; }();   // (0: tools/graph_churn_probe.py's A/B)
        if (drain) (void)stream_sync(h->gstream);
        (void)hipGraphExecDestroy(h->gexec);
        h->gexec = nullptr;
      }
      if (h->warm_B != a.B) {
        // One UNCAPTURED forward (into eps_buf; the sample is not touched) before the first capture at this batch size:
        // the launchers' one-time work — constant buffers (hipMalloc + null-stream copy), hipFuncSetAttribute for the
        // >64 KiB LDS kernels, device-attribute queries — is not legal inside a stream capture.
        ADM_TRY(run_forward(h, a.x, h->eps_buf, a.B, h->coef_dev, run));
        h->warm_B = a.B;
      }
      hipGraph_t graph = nullptr;
      ADM_HIP_OK(hipStreamBeginCapture(run, hipStreamCaptureModeThreadLocal));
      int rc = enqueue_step(h, a, 0, run);
      hipError_t e = hipStreamEndCapture(run, &graph);
      if (rc != 0) { if (graph) (void)hipGraphDestroy(graph); return rc; }
      ADM_HIP_OK(e);
      ADM_HIP_OK(hipGraphInstantiate(&h->gexec, graph, nullptr, nullptr, 0));
      ADM_HIP_OK(hipGraphDestroy(graph));
      h->gkey = key;
      h->gstream = run;
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
  ADM_REQUIRE(cfg->cross_attention_dim >= 0, "unet_create: cross_attention_dim < 0");
  for (int i = 0; i < cfg->n_blocks; ++i)
    ADM_REQUIRE((cfg->down_attn[i] != 2 && cfg->up_attn[i] != 2) || cfg->cross_attention_dim > 0,
                "unet_create: CrossAttn blocks need cross_attention_dim > 0");
  if (cfg->cross_attention_dim > 0) {
    ADM_REQUIRE(cfg->attention_head_dim > 0, "unet_create: conditional model needs attention_head_dim (= number of heads)");
    for (int i = 0; i < cfg->n_blocks; ++i)
      ADM_REQUIRE(cfg->block_out_channels[i] % cfg->attention_head_dim == 0, "unet_create: channels not divisible by the head count");
  }
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
  h->net.destroy();
  h->ps.free_all();
  for (void* p : h->owned) dfree(p);
#if !defined(ADM_EMU)
  if (h->own_stream) { conv_ksplit_release(h->own_stream); (void)hipStreamDestroy(h->own_stream); }
#endif
  delete h;
}

int adm_unet_set_option(adm_unet_t* h, const char* name, int value) {
  ADM_REQUIRE(h && name, "unet_set_option: null argument");
  const std::string nm(name);
  ADM_REQUIRE(nm == "wino6" || nm == "single_sample", "unet_set_option: the per-model options are: wino6, single_sample");
  if (nm == "single_sample") {
    ADM_REQUIRE(value >= -1 && value <= 1, "unet_set_option: single_sample takes 0 (follow the process-wide option), 1 (on) or -1 (off)");
    if (h->net.single_sample != value) {
      h->net.single_sample = value;
      free_plan(h);
    }
    return 0;
  }
  ADM_REQUIRE(value == 0 || value == 1 || value == 2 || (value >= 16 && value <= 65536),
              "unet_set_option: wino6 takes 0 (follow the process-wide option), 1 (default layer rule), 2 (every layer the kernel tiles) or a plane-size floor n >= 16");
  if (h->net.wino6_rule != value) {
    h->net.wino6_rule = value;
    free_plan(h);                     // the statistic-tile counts and a captured loop follow the kernels: plan again on the next call
  }
  return 0;
}

int adm_unet_set_param(adm_unet_t* h, const char* key, const float* host_data, size_t numel) {
  ADM_REQUIRE(h && key && host_data, "unet_set_param: null argument");
  ADM_REQUIRE(!h->finalized, "unet_set_param: model already finalized (first forward ran)");
  return h->ps.set(key, host_data, numel);
}

int adm_unet_missing_params(adm_unet_t* h) {
  std::string names;
  const int n = h->ps.missing(&names);
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
  Bf16Scope fp32(0);
  InferenceScope inf(&h->net, st);
  ADM_TRY(inf.rc);
  return run_forward(h, x, out, B, nullptr, st);
}

int adm_unet_set_encoding(adm_unet_t* h, const float* encoding_dev, int seq_len) {
  ADM_REQUIRE(h, "unet_set_encoding: null handle");
  ADM_REQUIRE(h->cfg.cross_attention_dim > 0, "unet_set_encoding: this model has no cross-attention (UNet2DModel)");
  ADM_REQUIRE((encoding_dev != nullptr) == (seq_len > 0), "unet_set_encoding: pointer and seq_len must both be given (or both cleared)");
  h->net.ctx = encoding_dev; h->net.ctx_S = seq_len; h->net.ctx_D = h->cfg.cross_attention_dim;
  return 0;
}

int adm_unet_bind_param(adm_unet_t* h, const char* key, float* dev_ptr) {
  ADM_REQUIRE(h && key && dev_ptr, "unet_bind_param: null argument");
  ADM_REQUIRE(!h->finalized, "unet_bind_param: model already finalized");
  return h->ps.bind(key, dev_ptr);
}

int adm_unet_enable_training(adm_unet_t* h, const float* params_base, long numel) {
  ADM_REQUIRE(h && params_base && numel > 0, "unet_enable_training: bad argument");
  ADM_REQUIRE(!h->finalized, "unet_enable_training: must be called before the first forward");
  for (auto& kv : h->ps.params)
    ADM_REQUIRE(kv.second.set && kv.second.dev >= params_base && kv.second.dev + kv.second.numel <= params_base + numel,
                "unet_enable_training: parameter " + kv.first + " is not bound inside the flat buffer");
  h->training = true;
  h->bf16_level = conv_bf16_mode();
  h->op16_f16 = conv_op16_f16() ? 1 : 0;
  h->params_numel = numel;
  h->net.params_base = params_base;
  return 0;
}

int adm_unet_set_loss_scale(adm_unet_t* h, float scale) {
  ADM_REQUIRE(h && scale > 0.f, "unet_set_loss_scale: bad argument");
  h->loss_scale = scale;
  return 0;
}

int adm_unet_refresh_weights(adm_unet_t* h, void* stream) {
  ADM_REQUIRE(h && h->finalized, "unet_refresh_weights: model not finalized");
  Bf16Scope own(h->bf16_level, h->op16_f16);
  ADM_TRY(h->net.refresh_weights((hipStream_t)stream));
  return restack_temb(h, (hipStream_t)stream);
}

// One training forward + backward (scripts/train_unet.py:257-259 without the optimizer): loss_dev[0] = mse(unet(x, t),
// target); grads_base (flat, same offsets as the parameter buffer) receives d loss / d parameters.
int adm_unet_forward_backward(adm_unet_t* h, const float* x, const float* timesteps_host, int n_timesteps,
                              const float* target, float* loss_dev, float* grads_base, int B, void* stream) {
  ADM_REQUIRE(h && x && timesteps_host && target && loss_dev && grads_base, "unet_forward_backward: null argument");
  ADM_REQUIRE(h->training, "unet_forward_backward: call adm_unet_enable_training first");
  ADM_REQUIRE(n_timesteps == 1 || n_timesteps == B, "unet_forward_backward: need 1 or B timesteps");
  hipStream_t st = (hipStream_t)stream;
  Bf16Scope own(h->bf16_level, h->op16_f16);
  ADM_TRY(finalize(h));
  // `drop_last=False` (train_unet.py:181): the partial last batch of an epoch runs INSIDE the full batch's plan (every buffer
  // is sized per sample and the split-K workspace shrinks with B) instead of re-planning twice per epoch
  if (!(h->planned_B > 0 && B <= h->planned_B)) ADM_TRY(plan(h, B));
  ADM_TRY(h->net.begin_training_batch(B, st));
  std::vector<float> t(B);
  for (int i = 0; i < B; ++i) t[i] = timesteps_host[n_timesteps == 1 ? 0 : i];
  ADM_TRY(copy_h2d(h->t_dev, t.data(), sizeof(float) * B, st));
  ADM_TRY(stream_sync(st));
  ADM_TRY(run_forward(h, x, h->eps_buf, B, nullptr, st));
  Net& net = h->net;
  const adm_unet_config& c = h->cfg;
  const long n_out = (long)B * c.out_channels * c.sample_h * c.sample_w;
  ADM_TRY(adm_mse_loss(h->eps_buf, target, n_out, loss_dev, net.tensors[net.t_out].grad, h->scratch_d, st));
  if (h->loss_scale != 1.f)   // fp16 loss scaling: every gradient of the reverse pass carries the factor
    ADM_TRY(adm_flat_op(net.tensors[net.t_out].grad, net.tensors[net.t_out].grad, n_out, 1, h->loss_scale, st));
  net.grads_base = grads_base;
  net.bucket_reset();
  ADM_TRY(dmemset(grads_base, 0, sizeof(float) * (size_t)h->params_numel, st));
  ADM_TRY(net.run_backward(B, h->dtemb_all, h->temb_rows, st));
  // ---- time-embedding path: time_emb_proj (per resnet), then the 2-layer MLP ---------------------------------
  const int K = h->temb_dim, R = h->temb_rows, dim_in = c.block_out_channels[0];
  int off = 0;
  for (auto& r : net.temb_rows) {
    ADM_TRY(launch_linear_bwd(h->dtemb_all + off, R, h->emb, nullptr, B, r.second, K, 1,
                              net.grad_of(P(h, r.first + ".weight")), net.grad_of(P(h, r.first + ".bias")), nullptr, st));
    net.mark_ready(P(h, r.first + ".weight"), (size_t)r.second * K);
    net.mark_ready(P(h, r.first + ".bias"), (size_t)r.second);
    off += r.second;
  }
  ADM_TRY(launch_linear_bwd(h->dtemb_all, R, h->emb, h->temb_w, B, R, K, 1, nullptr, nullptr, h->demb, st));
  ADM_TRY(launch_linear_bwd(h->demb, K, h->save_z, P(h, "time_embedding.linear_2.weight"), B, K, K, 1,
                            net.grad_of(P(h, "time_embedding.linear_2.weight")),
                            net.grad_of(P(h, "time_embedding.linear_2.bias")), h->dz, st));
  ADM_TRY(launch_linear_bwd(h->dz, K, h->save_sinus, nullptr, B, K, dim_in, 0,
                            net.grad_of(P(h, "time_embedding.linear_1.weight")),
                            net.grad_of(P(h, "time_embedding.linear_1.bias")), nullptr, st));
  net.mark_ready(P(h, "time_embedding.linear_2.weight"), (size_t)K * K);
  net.mark_ready(P(h, "time_embedding.linear_2.bias"), (size_t)K);
  net.mark_ready(P(h, "time_embedding.linear_1.weight"), (size_t)K * dim_in);
  net.mark_ready(P(h, "time_embedding.linear_1.bias"), (size_t)K);
  return 0;
}

// Data-parallel overlap (scripts/train_unet.py:259: DDP all-reduces gradient buckets while autograd is still running):
// fn(user, b) fires on the calling thread, during adm_unet_forward_backward, right after the last kernel writing into
// bucket b = [bounds[b], bounds[b+1]) of the flat gradient buffer has been enqueued. n_buckets = 0 removes the hook.
int adm_unet_set_grad_bucket_hook(adm_unet_t* h, int n_buckets, const long* bounds, adm_bucket_fn fn, void* user) {
  ADM_REQUIRE(h, "unet_set_grad_bucket_hook: null handle");
  ADM_REQUIRE(n_buckets == 0 || h->training, "unet_set_grad_bucket_hook: call adm_unet_enable_training first");
  if (n_buckets > 0) ADM_TRY(finalize(h));   // the op list (and its parameter table) must exist to count a bucket's writers
  return h->net.set_bucket_hook(n_buckets, bounds, fn, user);
}

size_t adm_unet_workspace_bytes(adm_unet_t* h) { return h ? h->net.arena_bytes : 0; }

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
  Bf16Scope fp32(0);
  InferenceScope inf(&h->net, st);
  ADM_TRY(inf.rc);
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
  Bf16Scope fp32(0);
  ADM_TRY(finalize(h));
  InferenceScope inf(&h->net, (hipStream_t)stream);
  ADM_TRY(inf.rc);
  return run_loop(h, a, coef_host, use_graph, (hipStream_t)stream);
}

int adm_encode_loop(adm_unet_t* h, float* x, int B, const adm_sched_coef* coef_host, int n_steps, int use_graph,
                    void* stream) {
  ADM_REQUIRE(h && x && coef_host && n_steps > 0, "encode_loop: bad argument");
  LoopArgs a{x, B, n_steps, nullptr, nullptr, 0, 0, nullptr, 1};
  Bf16Scope fp32(0);
  ADM_TRY(finalize(h));
  InferenceScope inf(&h->net, (hipStream_t)stream);
  ADM_TRY(inf.rc);
  return run_loop(h, a, coef_host, use_graph, (hipStream_t)stream);
}

}  // extern "C"
