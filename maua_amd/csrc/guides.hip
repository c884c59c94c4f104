// Grad modules of the guided sampler as library objects, so that a LIST of them can be evaluated inside the captured guided loop
// (maua_unet_set_guides, unet.hip) - the reference sums up to four modules per step (maua/diffusion/image.py:92-97,
// guided.py:258-266: `for grad_mod in self.grad_modules: ... img_grad += sub_grad`).
//
// A guide holds what one module's forward needs besides the image: VGGGrads (maua/grad.py:90-93) = the perceptor handle, its style
// taps, the target Gram matrices and the strength; LPIPSGrads (:189-193) = the network handle, taps, the target's unit-normalised
// features, the lin layers, the scale; ColorMatchGrads (:67-70) = bins, weighting, the target histogram, the scale.  The target
// tensors stay the caller's (device pointers; updated in place they are read by the next evaluation, also from a captured graph).
// guide_eval launches on the context's stream and allocates nothing once guide_prepare has seen the batch shape.
#include <atomic>
#include <vector>

#include "common.h"
#include "internal.h"

using namespace maua;

struct maua_guide {
  maua_ctx* ctx = nullptr;
  unsigned long long uid = 0;
  int kind = 0;                       // MAUA_GUIDE_STYLE / _LPIPS / _COLORMATCH
  maua_vgg* vgg = nullptr;
  std::vector<int> taps;
  std::vector<const float*> targets;
  std::vector<long> strides;
  std::vector<const float*> lins;
  float scale = 1.f;
  int nbins = 255, sat = 1, per_sample = 0;
  // colour-match workspaces (grow-only)
  unsigned long long* fix = nullptr;
  float* gr = nullptr;
  int cm_cap = 0;                     // samples the workspaces hold
  unsigned long long cm_epoch = 0;
};

namespace maua {

maua_ctx* guide_ctx(maua_guide* g) { return g ? g->ctx : nullptr; }
unsigned long long guide_uid(maua_guide* g) { return g ? g->uid : 0; }
unsigned long long guide_epoch(maua_guide* g) { return !g ? 0 : g->kind == MAUA_GUIDE_COLORMATCH ? g->cm_epoch : vgg_epoch(g->vgg); }

int guide_prepare(maua_guide* g, int B, int H, int W) {
  if (g->kind != MAUA_GUIDE_COLORMATCH) return MAUA_OK;   // (the perceptors size their workspaces in their first eager evaluation)
  if (B <= g->cm_cap) return MAUA_OK;
  MAUA_HIP_CHECK(hipStreamSynchronize(g->ctx->stream));
  if (g->fix) hipFree(g->fix);
  if (g->gr) hipFree(g->gr);
  g->fix = nullptr; g->gr = nullptr; g->cm_cap = 0; g->cm_epoch++;
  MAUA_HIP_CHECK(hipMalloc((void**)&g->fix, colormatch_fix_bytes(B, g->nbins)));
  MAUA_HIP_CHECK(hipMalloc((void**)&g->gr, (size_t)B * g->nbins * 4));
  g->cm_cap = B;
  return MAUA_OK;
}

int guide_eval(maua_guide* g, const float* img, int B, int H, int W, float* out) {
  switch (g->kind) {
    case MAUA_GUIDE_STYLE:
      return maua_vgg_style_grad(g->vgg, img, B, H, W, g->taps.data(), (int)g->taps.size(), g->targets.data(), g->strides.data(), g->scale, out,
                                 nullptr);
    case MAUA_GUIDE_LPIPS:
      return maua_vgg_lpips_grad(g->vgg, img, B, H, W, g->taps.data(), (int)g->taps.size(), g->targets.data(), g->strides.data(),
                                 g->lins.data(), g->scale, out, nullptr);
    case MAUA_GUIDE_COLORMATCH:
      MAUA_REQUIRE(B <= g->cm_cap, "maua_guide: colour-match workspaces were prepared for a smaller batch");
      return colormatch_grad_into(g->ctx->stream, img, B, H, W, g->nbins, g->sat, g->targets[0], g->per_sample, g->scale, g->fix, g->gr, out,
                                  nullptr);
  }
  return fail("maua_guide: unknown kind");
}

}  // namespace maua

extern "C" {

int maua_guide_create(maua_ctx* ctx, int kind, maua_vgg* vgg, const int* taps, int n_taps, const float* const* targets,
                      const long* target_bstride, const float* const* lins, float scale, int nbins, int sat_weighting, maua_guide** out) {
  MAUA_REQUIRE(ctx && out && targets, "maua_guide_create: NULL argument");
  MAUA_REQUIRE(kind == MAUA_GUIDE_STYLE || kind == MAUA_GUIDE_LPIPS || kind == MAUA_GUIDE_COLORMATCH, "maua_guide_create: unknown kind");
  if (kind == MAUA_GUIDE_COLORMATCH) {
    MAUA_REQUIRE(targets[0] && nbins >= 2 && nbins <= 1024, "maua_guide_create: colour match needs a target histogram and 2 <= bins <= 1024");
  } else {
    MAUA_REQUIRE(vgg && taps && n_taps > 0, "maua_guide_create: a perceptor guide needs its network and taps");
    MAUA_REQUIRE(vgg_ctx(vgg) == ctx, "maua_guide_create: the perceptor lives on another context");
    MAUA_REQUIRE(kind != MAUA_GUIDE_LPIPS || lins, "maua_guide_create: lpips needs the lin layers");
  }
  maua_guide* g = new maua_guide();
  static std::atomic<unsigned long long> next_uid{1};
  g->uid = next_uid.fetch_add(1);
  g->ctx = ctx; g->kind = kind; g->vgg = vgg; g->scale = scale; g->nbins = nbins; g->sat = sat_weighting;
  const int n = kind == MAUA_GUIDE_COLORMATCH ? 1 : n_taps;
  for (int k = 0; k < n; k++) {
    if (kind != MAUA_GUIDE_COLORMATCH) g->taps.push_back(taps[k]);
    g->targets.push_back(targets[k]);
    g->strides.push_back(target_bstride ? target_bstride[k] : 0L);
    if (kind == MAUA_GUIDE_LPIPS) g->lins.push_back(lins[k]);
  }
  g->per_sample = kind == MAUA_GUIDE_COLORMATCH && g->strides[0] != 0;
  *out = g;
  return MAUA_OK;
}

void maua_guide_destroy(maua_guide* g) {
  if (!g) return;
  hipStreamSynchronize(g->ctx->stream);
  if (g->fix) hipFree(g->fix);
  if (g->gr) hipFree(g->gr);
  delete g;
}

// the module as an operator: grad = d loss / d img of img device f32 [B][3][H][W]
int maua_guide_grad(maua_guide* g, const float* img, int B, int H, int W, float* grad) {
  MAUA_REQUIRE(g && img && grad, "maua_guide_grad: NULL argument");
  MAUA_REQUIRE(B >= 0 && H > 0 && W > 0, "maua_guide_grad: bad shape");
  if (B == 0) return MAUA_OK;
  if (int rc = guide_prepare(g, B, H, W)) return rc;
  return guide_eval(g, img, B, H, W, grad);
}

}  // extern "C"
