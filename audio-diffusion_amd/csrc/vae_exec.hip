// vae_exec.hip — native AutoencoderKL executor behind the C-ABI (include/adm.h).
// Replaces `vqvae.encode(x).latent_dist.sample(generator)` (audiodiffusion/pipeline_audio_diffusion.py:144,
// scripts/train_unet.py:104,233) and `vqvae.decode(z)["sample"]` (pipeline_audio_diffusion.py:190).
// Architecture = diffusers==0.24.0 Encoder/Decoder as configured by audiodiffusion/utils.py:132-153 from
// config/ldm_autoencoder_kl.yaml:18-28. Built from the same op set as the UNet (GroupNorm statistics + fused MFMA
// convolutions); the stride-2 encoder convs use diffusers' asymmetric (0,1,0,1) zero pad (pad_lo = 0); the single-head
// d = C mid-block attention runs its two products on the MFMA 1x1 kernel with per-sample weights (net_exec.hip).
#include <string>
#include <vector>

#include "net_exec.h"

using namespace adm;

struct adm_vae {
  adm_vae_config cfg;
  ParamStore ps;
  Net enc, dec;
  bool finalized = false;
  int planned_B_enc = 0, planned_B_dec = 0;
  float *moments = nullptr, *zq = nullptr;  // per-batch scratch: encoder moments, post-quant latent
  int lat_h = 0, lat_w = 0;
};

namespace adm {

static void vae_declare(adm_vae* h) {
  const adm_vae_config& c = h->cfg;
  ParamStore& ps = h->ps;
  const int nb = c.n_blocks, L = c.layers_per_block;
  const int* boc = c.block_out_channels;
  ps.declare_conv("encoder.conv_in", boc[0], c.in_channels, 3);
  int out = boc[0];
  for (int i = 0; i < nb; ++i) {
    const int cin = out;
    out = boc[i];
    const std::string bp = "encoder.down_blocks." + std::to_string(i);
    for (int j = 0; j < L; ++j) ps.declare_resnet(bp + ".resnets." + std::to_string(j), j == 0 ? cin : out, out, 0);
    if (i != nb - 1) ps.declare_conv(bp + ".downsamplers.0.conv", out, out, 3);
  }
  const int mid = boc[nb - 1];
  for (const char* side : {"encoder", "decoder"}) {
    const std::string s(side);
    ps.declare_resnet(s + ".mid_block.resnets.0", mid, mid, 0);
    ps.declare_attn(s + ".mid_block.attentions.0", mid);
    ps.declare_resnet(s + ".mid_block.resnets.1", mid, mid, 0);
  }
  ps.declare_gn("encoder.conv_norm_out", mid);
  ps.declare_conv("encoder.conv_out", 2 * c.latent_channels, mid, 3);
  ps.declare_conv("quant_conv", 2 * c.latent_channels, 2 * c.latent_channels, 1);
  ps.declare_conv("post_quant_conv", c.latent_channels, c.latent_channels, 1);
  ps.declare_conv("decoder.conv_in", mid, c.latent_channels, 3);
  out = boc[nb - 1];
  for (int i = 0; i < nb; ++i) {
    const int prev = out;
    out = boc[nb - 1 - i];
    const std::string bp = "decoder.up_blocks." + std::to_string(i);
    for (int j = 0; j < L + 1; ++j) ps.declare_resnet(bp + ".resnets." + std::to_string(j), j == 0 ? prev : out, out, 0);
    if (i != nb - 1) ps.declare_conv(bp + ".upsamplers.0.conv", out, out, 3);
  }
  ps.declare_gn("decoder.conv_norm_out", boc[0]);
  ps.declare_conv("decoder.conv_out", c.out_channels, boc[0], 3);
}

static int vae_finalize(adm_vae* h) {
  if (h->finalized) return 0;
  std::string missing;
  const int nmiss = h->ps.missing(&missing);
  ADM_REQUIRE(nmiss == 0, "vae: " + std::to_string(nmiss) + " parameters not set: " + missing);
  const adm_vae_config& c = h->cfg;
  const int nb = c.n_blocks, L = c.layers_per_block;
  const int* boc = c.block_out_channels;
  int rc = 0;
  const ConvW* w;
  // ---- encoder: x (Cin,H,W) -> moments (2*Cz, H/2^(nb-1), W/2^(nb-1)) incl. quant_conv ------------------
  {
    Net& b = h->enc;
    b.ps = &h->ps; b.groups = c.norm_num_groups; b.eps = 1e-6f;
    b.t_in = b.new_tensor(c.in_channels, c.sample_h, c.sample_w, true);
    ADM_TRY(b.make_conv("encoder.conv_in", boc[0], c.in_channels, 3, &w));
    int x = b.conv_op(b.t_in, -1, w, -1, 0, 0, 1, 1, -1, -1);
    int out = boc[0];
    for (int i = 0; i < nb; ++i) {
      const int cin = out;
      out = boc[i];
      const std::string bp = "encoder.down_blocks." + std::to_string(i);
      for (int j = 0; j < L; ++j) {
        x = b.resnet(bp + ".resnets." + std::to_string(j), x, -1, j == 0 ? cin : out, out, false, &rc);
        ADM_TRY(rc);
      }
      if (i != nb - 1) {
        ADM_TRY(b.make_conv(bp + ".downsamplers.0.conv", out, out, 3, &w));
        x = b.conv_op(x, -1, w, -1, 0, 0, 2, /*pad_lo=*/0, -1, -1);  // F.pad(x, (0,1,0,1)) + conv stride 2, padding 0
      }
    }
    const int mid = boc[nb - 1];
    x = b.resnet("encoder.mid_block.resnets.0", x, -1, mid, mid, false, &rc); ADM_TRY(rc);
    x = b.attention("encoder.mid_block.attentions.0", x, mid, mid, &rc); ADM_TRY(rc);
    x = b.resnet("encoder.mid_block.resnets.1", x, -1, mid, mid, false, &rc); ADM_TRY(rc);
    const int g = b.gn_op(x, -1, b.make_gn("encoder.conv_norm_out", mid));
    ADM_TRY(b.make_conv("encoder.conv_out", 2 * c.latent_channels, mid, 3, &w));
    x = b.conv_op(x, -1, w, g, 1, 0, 1, 1, -1, -1);
    h->lat_h = b.tensors[x].H; h->lat_w = b.tensors[x].W;
    ADM_TRY(b.make_conv("quant_conv", 2 * c.latent_channels, 2 * c.latent_channels, 1, &w));
    b.t_out = b.new_tensor(2 * c.latent_channels, h->lat_h, h->lat_w, true);
    b.conv_op(x, -1, w, -1, 0, 0, 1, 0, -1, -1, b.t_out);
    b.finish_liveness();
  }
  // ---- decoder: z (Cz,h,w) -> sample (Cout,H,W) incl. post_quant_conv --------------------------------------
  {
    Net& b = h->dec;
    b.ps = &h->ps; b.groups = c.norm_num_groups; b.eps = 1e-6f;
    b.t_in = b.new_tensor(c.latent_channels, h->lat_h, h->lat_w, true);
    ADM_TRY(b.make_conv("post_quant_conv", c.latent_channels, c.latent_channels, 1, &w));
    int x = b.conv_op(b.t_in, -1, w, -1, 0, 0, 1, 0, -1, -1);
    const int mid = boc[nb - 1];
    ADM_TRY(b.make_conv("decoder.conv_in", mid, c.latent_channels, 3, &w));
    x = b.conv_op(x, -1, w, -1, 0, 0, 1, 1, -1, -1);
    x = b.resnet("decoder.mid_block.resnets.0", x, -1, mid, mid, false, &rc); ADM_TRY(rc);
    x = b.attention("decoder.mid_block.attentions.0", x, mid, mid, &rc); ADM_TRY(rc);
    x = b.resnet("decoder.mid_block.resnets.1", x, -1, mid, mid, false, &rc); ADM_TRY(rc);
    int out = boc[nb - 1];
    for (int i = 0; i < nb; ++i) {
      const int prev = out;
      out = boc[nb - 1 - i];
      const std::string bp = "decoder.up_blocks." + std::to_string(i);
      for (int j = 0; j < L + 1; ++j) {
        x = b.resnet(bp + ".resnets." + std::to_string(j), x, -1, j == 0 ? prev : out, out, false, &rc);
        ADM_TRY(rc);
      }
      if (i != nb - 1) {
        ADM_TRY(b.make_conv(bp + ".upsamplers.0.conv", out, out, 3, &w));
        x = b.conv_op(x, -1, w, -1, 0, 1, 1, 1, -1, -1);
      }
    }
    const int g = b.gn_op(x, -1, b.make_gn("decoder.conv_norm_out", boc[0]));
    ADM_TRY(b.make_conv("decoder.conv_out", c.out_channels, boc[0], 3, &w));
    b.t_out = b.new_tensor(c.out_channels, b.tensors[x].H, b.tensors[x].W, true);
    b.conv_op(x, -1, w, g, 1, 0, 1, 1, -1, -1, b.t_out);
    b.finish_liveness();
  }
  ADM_TRY(stream_sync(nullptr));
  h->finalized = true;
  return 0;
}

}  // namespace adm

extern "C" {

int adm_vae_create(const adm_vae_config* cfg, adm_vae_t** out) {
  ADM_REQUIRE(cfg && out, "vae_create: null argument");
  ADM_REQUIRE(cfg->n_blocks >= 1 && cfg->n_blocks <= 8 && cfg->layers_per_block >= 1, "vae_create: bad config");
  for (int i = 0; i < cfg->n_blocks; ++i)
    ADM_REQUIRE(cfg->block_out_channels[i] % 32 == 0 && cfg->block_out_channels[i] % cfg->norm_num_groups == 0,
                "vae_create: block_out_channels must be multiples of 32 and of norm_num_groups");
  ADM_REQUIRE(cfg->latent_channels >= 1 && 2 * cfg->latent_channels <= 4, "vae_create: latent_channels must be 1 or 2");
  adm_vae* h = new adm_vae();
  h->cfg = *cfg;
  vae_declare(h);
  *out = h;
  return 0;
}

void adm_vae_destroy(adm_vae_t* h) {
  if (!h) return;
  h->enc.destroy();
  h->dec.destroy();
  h->ps.free_all();
  if (h->moments) dfree(h->moments);
  if (h->zq) dfree(h->zq);
  delete h;
}

int adm_vae_set_param(adm_vae_t* h, const char* key, const float* host_data, size_t numel) {
  ADM_REQUIRE(h && key && host_data, "vae_set_param: null argument");
  ADM_REQUIRE(!h->finalized, "vae_set_param: model already finalized");
  return h->ps.set(key, host_data, numel);
}

int adm_vae_latent_dims(adm_vae_t* h, int* lat_h, int* lat_w) {
  ADM_REQUIRE(h && lat_h && lat_w, "vae_latent_dims: null argument");
  ADM_TRY(vae_finalize(h));
  *lat_h = h->lat_h; *lat_w = h->lat_w;
  return 0;
}

int adm_vae_encode(adm_vae_t* h, const float* x, const float* noise, float out_scale, float* z_out, float* moments_out,
                   int B, void* stream) {
  ADM_REQUIRE(h && x && z_out, "vae_encode: null argument");
  hipStream_t st = (hipStream_t)stream;
  ADM_TRY(vae_finalize(h));
  const int Cz = h->cfg.latent_channels;
  const long hw = (long)h->lat_h * h->lat_w;
  if (h->planned_B_enc != B || !h->enc.plan_current(B)) {
    ADM_TRY(stream_sync(st));
    ADM_TRY(h->enc.plan(B));
    if (h->moments) dfree(h->moments);
    ADM_TRY(dmalloc((void**)&h->moments, sizeof(float) * (size_t)B * 2 * Cz * hw));
    h->planned_B_enc = B;
  }
  float* mom = moments_out ? moments_out : h->moments;
  ADM_TRY(h->enc.run(x, mom, B, nullptr, 0, st, nullptr));
  return launch_gaussian_sample(mom, noise, z_out, B, Cz, hw, out_scale, st);
}

int adm_vae_decode(adm_vae_t* h, const float* z, float in_scale, float* out, int B, void* stream) {
  ADM_REQUIRE(h && z && out, "vae_decode: null argument");
  hipStream_t st = (hipStream_t)stream;
  ADM_TRY(vae_finalize(h));
  const long n = (long)B * h->cfg.latent_channels * h->lat_h * h->lat_w;
  if (h->planned_B_dec != B || !h->dec.plan_current(B)) {
    ADM_TRY(stream_sync(st));
    ADM_TRY(h->dec.plan(B));
    if (h->zq) dfree(h->zq);
    ADM_TRY(dmalloc((void**)&h->zq, sizeof(float) * (size_t)n));
    h->planned_B_dec = B;
  }
  const float* zin = z;
  if (in_scale != 1.0f) {
    ADM_TRY(launch_scale(z, h->zq, in_scale, n, st));
    zin = h->zq;
  }
  return h->dec.run(zin, out, B, nullptr, 0, st, nullptr);
}

}  // extern "C"
