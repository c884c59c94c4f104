// StyleGAN2 synthesis network object: parameters resident in HBM, one batched forward = a fixed sequence of
// launches on the ctx stream (styles -> [conv0(up2) -> conv1 -> toRGB+skip] x blocks -> optional u8 pack).
//
// Replaces (reference): inference/stylegan2.py:385-436 SynthesisNetwork, :275-382 SynthesisBlock,
// :195-251 SynthesisLayer, :254-272 ToRGBLayer; wrappers/stylegan2.py:85-102 (per-batch noise install).
// Activations are NHWC in the network dtype (bf16 or f32); the RGB skip image stays f32 planar.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "internal.h"

using namespace maua;

namespace {

struct ConvLayer {
  int block, which;  // which: 0 = conv0 (up 2), 1 = conv1
  int Ci, Co, res, up, w_index;
  int ih = 0, iw = 0;   // input grid of this launch
  int oh = 0, ow = 0;   // conv output grid (= where the noise is added)
  int fh = 0, fw = 0;   // grid handed to the next layer (differs from oh x ow only on the resized layer)
  float* affine_w = nullptr;  // [Ci][w_dim]
  float* affine_b = nullptr;  // [Ci]
  float* bias = nullptr;      // [Co]
  float* noise_const = nullptr;  // [res][res]
  float noise_strength = 0.f;    // loaded value (used under nv_compat bit1)
  void* wt = nullptr;            // prepared weights (up-layers: 4 phase kernels, 9 taps each)
  void* wt_t = nullptr;          // up-layers: transposed-conv class weights (minimal MACs; FIR done afterwards)
  void* wt_h = nullptr;          // up-layers on the row-walk kernel: half-folded weights (modconv_upwalk.hip)
  float* wsq = nullptr;          // [Co][Ci]
  float* s = nullptr;            // [Bcap][Ci]
  float* d = nullptr;            // [Bcap][Co]
  void* feat = nullptr;          // keep_features buffer
};
struct RgbLayer {
  int block, C, res, w_index;
  int h = 0, w = 0;     // grid toRGB runs on (the block's feature grid)
  float* affine_w = nullptr;
  float* affine_b = nullptr;
  float* wrgb = nullptr;  // [3][C]
  float* bias = nullptr;  // [3]
  float* s = nullptr;     // [Bcap][C]
  float* wmod = nullptr;  // [Bcap][3][C]
};

template <typename T>
int dev_alloc(T** p, size_t n) {
  MAUA_HIP_CHECK(hipMalloc((void**)p, n * sizeof(T)));
  return MAUA_OK;
}

}  // namespace

struct maua_synth {
  maua_ctx* ctx;
  int res, w_dim, channel_base, channel_max, dtype, nv_compat;
  int nblocks, num_ws;
  size_t esize;
  std::vector<ConvLayer> convs;
  std::vector<RgbLayer> rgbs;
  void* const_x = nullptr;  // NHWC [4][4][C0]
  int keep_features = 0;
  int lowres = 1;      // <= 8x8 layers as one batch-wide split-K GEMM (option "lowres")
  int use_hires = 1;   // weights-in-registers kernels for the 512^2 / 1024^2 layers (bf16 / f16)
  int upwalk = 2;      // ... and their 64 -> 32 channel up-layer on the half-folded row walk (modconv_upwalk.hip);
                       // 2: the last block as one fused walk when nothing else reads its features
  int walk_segs = 0;   // fused walk: force this many row segments (0: cost model); walk_narrow: a <= 32-column last strip as two
  int walk_narrow = 1; // half-height sub-items walked at once (options "walk_segs" / "walk_narrow"; results do not depend on either)
  int fuse_torgb = 1;  // toRGB + skip fused into those conv1 epilogues
  int tconv_up = 1;    // up-layers: minimal transposed conv + separate FIR/epilogue pass (0 = 4 phase kernels)
  int tconv_fir = 256; // up-layers with inputs of at least this size: transposed conv + FIR + epilogue in ONE kernel, t stays in
                       // LDS (modconv_tconv_fir.hip); 0 = never.  Measured at B = 128 (pair -> fused): 256^2 inputs 3.61 -> 3.48 ms,
                       // 128^2 2.33 -> 2.62, 64^2 1.89 -> 2.65: the 1.42x MACs pay only where the t round trip was HBM-bound.
  const float* nz_scales = nullptr;   // [num_layers][nz_scale_stride] per-sample noise factors (maua_synth_set_noise_scale) or NULL
  long nz_scale_stride = 0;
  int tconv_min = 32;  // ... from this input size up (below: the phase kernels / the batch-wide low-resolution GEMM)
  int dma_conv = 1;    // conv1 layers behind such an up-layer: LDS-direct-load kernel on pre-modulated input (bf16 / f16)
  int dual_store = 1;  // ... whose toRGB is a separate pass (512 channels): plain + style-scaled output in one epilogue (no premod pass)
  int tconv_dma = 2;   // the up-layers' transposed conv on LDS-direct loads (main block; pre-modulated input)
  float* ones = nullptr;   // [Bcap][max channels] unit styles (kernels that take already-modulated input)
  void* xm = nullptr;      // [Bcap] pre-modulated copy of an up-layer's input when its producer could not scale it
  void* tbuf = nullptr;  // [Bcap] transposed-conv tensor of the largest up-layer
  void* lowres_xm = nullptr;   // [Bcap] modconv_lowres workspaces (premodulated input, split-K partial sums)
  float* lowres_ws = nullptr;
  // one feature-space resize (wrappers/stylegan2.py:104-151): rs_layer = -1 none, 0 = before layer 0, L = after layer L-1
  int rs_layer = -1, rs_mode = 0, rs_th = 0, rs_tw = 0, rs_pl = 0, rs_pr = 0, rs_pt = 0, rs_pb = 0, rs_how = 3;
  float rs_value = 0.f;
  float* rs_noise = nullptr;   // [C][th][tw] fill noise or NULL
  void* const_rs = nullptr;    // resized const input (rs_layer == 0)
  float* rgb_tmp[2] = {nullptr, nullptr};
  int out_h = 0, out_w = 0;    // final image
  // geometric transform hooks (wrappers/stylegan2.py:153-194): up to 3 warps, applied in slot order after their layer
  int warp_layer[3] = {-1, -1, -1};
  const float* warp_minv[3] = {nullptr, nullptr, nullptr};  // device [B][6], owned by the caller
  // profile mode: HIP events recorded on the ctx stream around every launch of a forward
  int profile = 0;
  std::vector<hipEvent_t> ev;
  std::vector<std::string> ev_names;
  size_t ev_used = 0;
  std::vector<size_t> ev_fwd_start;
  // workspace
  int bcap = 0;
  void* act[2] = {nullptr, nullptr};
  float* img[2] = {nullptr, nullptr};
  StyleLayer* style_table_dev = nullptr;
  float fir[16];
};

static void prof_mark(maua_synth* n, const char* name) {
  if (!n->profile || n->ev_used >= (1u << 16)) return;
  if (n->ev_used == n->ev.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    n->ev.push_back(e);
  }
  hipEventRecord(n->ev[n->ev_used++], n->ctx->stream);
  n->ev_names.push_back(name);
}

static int channels_for(int res, int base, int maxc) { return std::min(base / res, maxc); }

// per-layer grids: native power-of-two sizes, scaled from the resized layer on
static void compute_dims(maua_synth* n) {
  int h = 4, w = 4;
  if (n->rs_layer == 0) { h = n->rs_th; w = n->rs_tw; }
  size_t li = 0;
  for (int blk = 0; blk < n->nblocks; blk++) {
    const int nconv = blk == 0 ? 1 : 2;
    for (int k = 0; k < nconv; k++, li++) {
      ConvLayer& c = n->convs[li];
      c.ih = h; c.iw = w;
      c.oh = h * c.up; c.ow = w * c.up;
      h = c.oh; w = c.ow;
      if (n->rs_layer == (int)li + 1) { h = n->rs_th; w = n->rs_tw; }
      c.fh = h; c.fw = w;
    }
    n->rgbs[blk].h = h; n->rgbs[blk].w = w;
  }
  n->out_h = h; n->out_w = w;
}

static int free_workspace(maua_synth* n) {
  for (auto& c : n->convs) {
    if (c.s) hipFree(c.s);
    if (c.d) hipFree(c.d);
    if (c.feat) hipFree(c.feat);
    c.s = c.d = nullptr;
    c.feat = nullptr;
  }
  for (auto& r : n->rgbs) {
    if (r.s) hipFree(r.s);
    if (r.wmod) hipFree(r.wmod);
    r.s = r.wmod = nullptr;
  }
  for (int i = 0; i < 2; i++) {
    if (n->act[i]) hipFree(n->act[i]);
    if (n->img[i]) hipFree(n->img[i]);
    n->act[i] = nullptr;
    n->img[i] = nullptr;
  }
  if (n->style_table_dev) hipFree(n->style_table_dev);
  n->style_table_dev = nullptr;
  if (n->tbuf) hipFree(n->tbuf);
  n->tbuf = nullptr;
  if (n->ones) hipFree(n->ones);
  if (n->xm) hipFree(n->xm);
  n->ones = nullptr;
  n->xm = nullptr;
  if (n->lowres_xm) hipFree(n->lowres_xm);
  if (n->lowres_ws) hipFree(n->lowres_ws);
  n->lowres_xm = nullptr;
  n->lowres_ws = nullptr;
  if (n->const_rs) hipFree(n->const_rs);
  n->const_rs = nullptr;
  for (int i = 0; i < 2; i++) {
    if (n->rgb_tmp[i]) hipFree(n->rgb_tmp[i]);
    n->rgb_tmp[i] = nullptr;
  }
  n->bcap = 0;
  return MAUA_OK;
}

static int ensure_workspace(maua_synth* n, int B) {
  if (B <= n->bcap) return MAUA_OK;
  MAUA_HIP_CHECK(hipStreamSynchronize(n->ctx->stream));
  free_workspace(n);
  size_t max_act = 0;
  for (auto& c : n->convs) {
    MAUA_HIP_CHECK(hipMalloc((void**)&c.s, (size_t)B * c.Ci * sizeof(float)));
    MAUA_HIP_CHECK(hipMalloc((void**)&c.d, (size_t)B * c.Co * sizeof(float)));
    size_t e = (size_t)c.fh * c.fw * c.Co;
    max_act = std::max(max_act, std::max(e, (size_t)c.oh * c.ow * c.Co));
    if (n->keep_features) MAUA_HIP_CHECK(hipMalloc(&c.feat, (size_t)B * e * n->esize));
  }
  for (auto& r : n->rgbs) {
    MAUA_HIP_CHECK(hipMalloc((void**)&r.s, (size_t)B * r.C * sizeof(float)));
    MAUA_HIP_CHECK(hipMalloc((void**)&r.wmod, (size_t)B * 3 * r.C * sizeof(float)));
  }
  // (a resized / warped layer goes through scratch buffers, also when every layer keeps its own)
  for (int i = 0; i < 2; i++) MAUA_HIP_CHECK(hipMalloc(&n->act[i], (size_t)B * max_act * n->esize));
  if (n->rs_layer == 0)
    MAUA_HIP_CHECK(hipMalloc(&n->const_rs, (size_t)n->rs_th * n->rs_tw * n->convs[0].Ci * n->esize));
  if (n->rs_layer >= 1) {
    const ConvLayer& hc = n->convs[n->rs_layer - 1];
    const size_t px = std::max((size_t)hc.oh * hc.ow, (size_t)n->rs_th * n->rs_tw);
    for (int i = 0; i < 2; i++) MAUA_HIP_CHECK(hipMalloc((void**)&n->rgb_tmp[i], (size_t)B * 3 * px * sizeof(float)));
  }
  size_t max_t = 0;
  for (auto& c : n->convs)
    if (c.up == 2) max_t = std::max(max_t, (size_t)(c.oh + 1) * (c.ow + 1) * c.Co);
  if (max_t) MAUA_HIP_CHECK(hipMalloc(&n->tbuf, (size_t)B * max_t * n->esize));
  size_t lx = 0, lw = 0;
  for (auto& c : n->convs)
    if (lowres_supported(n->dtype, c.Ci, c.Co, c.up, c.ih, c.iw)) {
      size_t x1, w1;
      lowres_workspace(n->dtype, B, c.ih, c.iw, c.Ci, c.Co, c.up, &x1, &w1);
      lx = std::max(lx, x1); lw = std::max(lw, w1);
    }
  {
    int maxc = 0;
    size_t xm_elems = 0;
    for (auto& c : n->convs) {
      maxc = std::max(maxc, std::max(c.Ci, c.Co));
      // (only the up-layers whose producer has no fused toRGB need the copy: inputs up to 64^2 at 1024^2 networks;
      //  sized for any up-layer so that hooks / options can fall back to it)
      if (c.up == 2 && tconv_dma_supported(n->dtype, c.Ci, c.Co, c.ih, c.iw)) xm_elems = std::max(xm_elems, (size_t)c.ih * c.iw * c.Ci);
    }
    std::vector<float> h1((size_t)B * maxc, 1.f);
    MAUA_HIP_CHECK(hipMalloc((void**)&n->ones, h1.size() * sizeof(float)));
    MAUA_HIP_CHECK(hipMemcpy(n->ones, h1.data(), h1.size() * sizeof(float), hipMemcpyHostToDevice));
    if (xm_elems) MAUA_HIP_CHECK(hipMalloc(&n->xm, (size_t)B * xm_elems * n->esize));
  }
  if (lx) MAUA_HIP_CHECK(hipMalloc(&n->lowres_xm, lx));
  if (lw) MAUA_HIP_CHECK(hipMalloc((void**)&n->lowres_ws, lw));
  for (int i = 0; i < 2; i++)
    MAUA_HIP_CHECK(hipMalloc((void**)&n->img[i], (size_t)B * 3 * std::max(n->out_h * n->out_w, n->res * n->res) * sizeof(float)));
  // style table
  std::vector<StyleLayer> tab;
  for (auto& c : n->convs) {
    StyleLayer L{};
    L.affine_w = c.affine_w; L.affine_b = c.affine_b; L.wsq = c.wsq; L.wrgb = nullptr;
    L.s = c.s; L.d = c.d; L.wmod = nullptr; L.w_index = c.w_index;
    L.Cin = c.Ci; L.Co = c.Co; L.Cs = c.Ci; L.Cd = c.Co; L.scale = 1.f;
    tab.push_back(L);
  }
  for (auto& r : n->rgbs) {
    StyleLayer L{};
    L.affine_w = r.affine_w; L.affine_b = r.affine_b; L.wsq = nullptr; L.wrgb = r.wrgb;
    L.s = r.s; L.d = nullptr; L.wmod = r.wmod; L.w_index = r.w_index;
    L.Cin = r.C; L.Co = 3; L.Cs = r.C; L.Cd = 0; L.scale = 1.f / std::sqrt((float)r.C);
    tab.push_back(L);
  }
  MAUA_HIP_CHECK(hipMalloc((void**)&n->style_table_dev, tab.size() * sizeof(StyleLayer)));
  MAUA_HIP_CHECK(hipMemcpy(n->style_table_dev, tab.data(), tab.size() * sizeof(StyleLayer), hipMemcpyHostToDevice));
  n->bcap = B;
  return MAUA_OK;
}

extern "C" {

int maua_synth_create(maua_ctx* ctx, int img_resolution, int w_dim, int channel_base, int channel_max, int dtype,
                      int nv_compat, maua_synth** out) {
  MAUA_REQUIRE(ctx && out, "maua_synth_create: NULL argument");
  MAUA_REQUIRE(img_resolution >= 4 && (img_resolution & (img_resolution - 1)) == 0,
               "maua_synth_create: img_resolution must be a power of two >= 4");
  MAUA_REQUIRE(dtype == MAUA_F32 || dtype == MAUA_BF16 || dtype == MAUA_F16,
               "maua_synth_create: dtype must be MAUA_F32, MAUA_BF16 or MAUA_F16");
  MAUA_REQUIRE(w_dim > 0 && w_dim <= 4096, "maua_synth_create: bad w_dim");
  const int kc = dtype == MAUA_F32 ? 16 : 32;
  maua_synth* n = new maua_synth();
  n->ctx = ctx; n->res = img_resolution; n->w_dim = w_dim; n->channel_base = channel_base;
  n->channel_max = channel_max; n->dtype = dtype; n->nv_compat = nv_compat;
  n->esize = dtype == MAUA_F32 ? 4 : 2;
  int nb = 0;
  for (int r = 4; r <= img_resolution; r *= 2) nb++;
  n->nblocks = nb;
  n->num_ws = 2 * nb;
  int widx = 0;
  for (int i = 0; i < nb; i++) {
    int r = 4 << i;
    int co = channels_for(r, channel_base, channel_max);
    if (co % 32 != 0 || co % kc != 0) {
      delete n;
      return fail("maua_synth_create: channel counts must be multiples of 32");
    }
    if (i > 0) {
      ConvLayer c{};
      c.block = i; c.which = 0; c.Ci = channels_for(r / 2, channel_base, channel_max); c.Co = co; c.res = r; c.up = 2;
      c.w_index = widx++;
      n->convs.push_back(c);
    }
    ConvLayer c{};
    c.block = i; c.which = 1; c.Ci = co; c.Co = co; c.res = r; c.up = 1; c.w_index = widx++;
    n->convs.push_back(c);
    RgbLayer g{};
    g.block = i; g.C = co; g.res = r; g.w_index = widx;  // toRGB shares the next block's first w (stylegan2.py:431-433)
    n->rgbs.push_back(g);
  }
  compute_dims(n);
  hipError_t e = hipSuccess;
  auto A = [&](void** p, size_t bytes) {
    if (e == hipSuccess) e = hipMalloc(p, bytes);
    if (e == hipSuccess) e = hipMemset(*p, 0, bytes);
  };
  for (auto& c : n->convs) {
    A((void**)&c.affine_w, (size_t)c.Ci * w_dim * 4);
    A((void**)&c.affine_b, (size_t)c.Ci * 4);
    A((void**)&c.bias, (size_t)c.Co * 4);
    A((void**)&c.noise_const, (size_t)c.res * c.res * 4);
    A(&c.wt, prepped_weight_elems(3, c.up, c.Co, c.Ci) * n->esize);
    if (c.up == 2) A(&c.wt_t, prepped_weight_elems(3, c.up, c.Co, c.Ci) * n->esize);
    // (the half-folded weights do not depend on the layer's size - it can change with a resize hook - only the routing does)
    if (upwalk_supported(n->dtype, c.Ci, c.Co, c.up, 64, 64)) A(&c.wt_h, upwalk_weight_elems(c.Co, c.Ci) * 2);
    A((void**)&c.wsq, (size_t)c.Co * c.Ci * 4);
  }
  for (auto& g : n->rgbs) {
    A((void**)&g.affine_w, (size_t)g.C * w_dim * 4);
    A((void**)&g.affine_b, (size_t)g.C * 4);
    A((void**)&g.wrgb, (size_t)3 * g.C * 4);
    A((void**)&g.bias, 3 * 4);
  }
  A(&n->const_x, (size_t)16 * n->convs[0].Ci * n->esize);
  if (e != hipSuccess) {
    maua_synth_destroy(n);
    return fail(std::string("maua_synth_create: hipMalloc: ") + hipGetErrorString(e));
  }
  const float g4[4] = {0.25f, 0.75f, 0.75f, 0.25f};  // upsample2d: f*gain(4) = outer(g4,g4)
  for (int u = 0; u < 4; u++)
    for (int v = 0; v < 4; v++) n->fir[u * 4 + v] = g4[u] * g4[v];
  *out = n;
  return MAUA_OK;
}

void maua_synth_destroy(maua_synth* n) {
  if (!n) return;
  hipStreamSynchronize(n->ctx->stream);
  free_workspace(n);
  for (auto e : n->ev) hipEventDestroy(e);
  for (auto& c : n->convs) {
    hipFree(c.affine_w); hipFree(c.affine_b); hipFree(c.bias); hipFree(c.noise_const); hipFree(c.wt); hipFree(c.wsq);
    if (c.wt_t) hipFree(c.wt_t);
    if (c.wt_h) hipFree(c.wt_h);
  }
  for (auto& g : n->rgbs) {
    hipFree(g.affine_w); hipFree(g.affine_b); hipFree(g.wrgb); hipFree(g.bias);
  }
  hipFree(n->const_x);
  if (n->rs_noise) hipFree(n->rs_noise);
  delete n;
}

int maua_synth_num_ws(const maua_synth* n) { return n ? n->num_ws : 0; }
int maua_synth_num_layers(const maua_synth* n) { return n ? (int)n->convs.size() : 0; }

int maua_synth_set_resize(maua_synth* n, int layer, int mode, int target_h, int target_w, int pad_left, int pad_right,
                          int pad_top, int pad_bottom, int pad_how, float pad_value, const float* fill_noise_host) {
  MAUA_REQUIRE(n, "maua_synth_set_resize: net is NULL");
  MAUA_REQUIRE(layer >= -1 && layer <= (int)n->convs.size(), "maua_synth_set_resize: no such layer");
  MAUA_HIP_CHECK(hipStreamSynchronize(n->ctx->stream));
  const std::vector<ConvLayer> before = n->convs;
  if (layer >= 0) {
    MAUA_REQUIRE(mode == 0 || mode == 1, "maua_synth_set_resize: mode must be 0 (stretch) or 1 (pad)");
    MAUA_REQUIRE(target_h >= 1 && target_w >= 1, "maua_synth_set_resize: empty target size");
    // native grid of the hooked tensor
    const int nat = layer == 0 ? 4 : n->convs[layer - 1].res;
    if (mode == 1) {
      // negative entries crop, as F.pad does.  The reference reaches them only through the pre-hook of layer 0
      // (wrappers/stylegan2.py:294); behind a later layer its toRGB inverse slices with negative bounds and the forward
      // fails (:313-323, the "TODO negative padding" at :278), so there is no behaviour to reproduce there.
      MAUA_REQUIRE(layer == 0 || (pad_left >= 0 && pad_right >= 0 && pad_top >= 0 && pad_bottom >= 0),
                   "maua_synth_set_resize: negative padding (cropping) is only defined at layer 0");
      MAUA_REQUIRE(-pad_left < nat && -pad_right < nat && -pad_top < nat && -pad_bottom < nat,
                   "maua_synth_set_resize: a crop must leave part of the layer");
      MAUA_REQUIRE(pad_how >= 0 && pad_how <= 3, "maua_synth_set_resize: unknown padding mode");
      MAUA_REQUIRE(target_h == nat + pad_top + pad_bottom && target_w == nat + pad_left + pad_right,
                   "maua_synth_set_resize: target size must equal the layer size plus the padding");
    }
    // the up-layers double the grid: every layer after the hook must stay even where toRGB upsamples the skip image
    n->rs_layer = layer; n->rs_mode = mode; n->rs_th = target_h; n->rs_tw = target_w;
    n->rs_pl = pad_left; n->rs_pr = pad_right; n->rs_pt = pad_top; n->rs_pb = pad_bottom;
    n->rs_how = pad_how; n->rs_value = pad_value;
  } else {
    n->rs_layer = -1;
  }
  if (n->rs_noise) hipFree(n->rs_noise);
  n->rs_noise = nullptr;
  if (layer >= 0 && fill_noise_host) {
    const int C = layer == 0 ? n->convs[0].Ci : n->convs[layer - 1].Co;
    const size_t cnt = (size_t)C * target_h * target_w;
    MAUA_HIP_CHECK(hipMalloc((void**)&n->rs_noise, cnt * sizeof(float)));
    MAUA_HIP_CHECK(hipMemcpy(n->rs_noise, fill_noise_host, cnt * sizeof(float), hipMemcpyHostToDevice));
  }
  compute_dims(n);
  free_workspace(n);
  // layers whose grid changed get a zeroed noise_const of the new size (the caller uploads fresh noise, :141-150)
  for (size_t i = 0; i < n->convs.size(); i++) {
    ConvLayer& c = n->convs[i];
    if (c.oh != before[i].oh || c.ow != before[i].ow) {
      hipFree(c.noise_const);
      c.noise_const = nullptr;
      MAUA_HIP_CHECK(hipMalloc((void**)&c.noise_const, (size_t)c.oh * c.ow * sizeof(float)));
      MAUA_HIP_CHECK(hipMemset(c.noise_const, 0, (size_t)c.oh * c.ow * sizeof(float)));
    }
  }
  return MAUA_OK;
}

int maua_synth_set_warp(maua_synth* n, int slot, int layer, const float* inv_matrices_dev) {
  MAUA_REQUIRE(n, "maua_synth_set_warp: net is NULL");
  MAUA_REQUIRE(slot >= 0 && slot < 3, "maua_synth_set_warp: slot must be 0..2");
  MAUA_REQUIRE(layer >= 1 && layer <= (int)n->convs.size(), "maua_synth_set_warp: layer must name a synthesis layer (1-based index into layer_names)");
  n->warp_layer[slot] = inv_matrices_dev ? layer : -1;
  n->warp_minv[slot] = inv_matrices_dev;
  return MAUA_OK;
}

int maua_synth_layer_size(const maua_synth* n, int layer, int* h, int* w) {
  MAUA_REQUIRE(n && h && w, "maua_synth_layer_size: NULL argument");
  MAUA_REQUIRE(layer >= -1 && layer < (int)n->convs.size(), "maua_synth_layer_size: no such layer");
  if (layer < 0) { *h = n->out_h; *w = n->out_w; }
  else { *h = n->convs[layer].oh; *w = n->convs[layer].ow; }
  return MAUA_OK;
}

int maua_synth_set_option(maua_synth* n, const char* key, int value) {
  MAUA_REQUIRE(n && key, "maua_synth_set_option: NULL argument");
  if (!strcmp(key, "profile")) {
    n->profile = value;
    n->ev_used = 0;
    n->ev_names.clear();
    n->ev_fwd_start.clear();
    return MAUA_OK;
  }
  if (!strcmp(key, "lowres")) {
    n->lowres = value;
    return MAUA_OK;
  }
  if (!strcmp(key, "use_hires")) {
    n->use_hires = value;
    return MAUA_OK;
  }
  if (!strcmp(key, "upwalk")) {
    n->upwalk = value;
    return MAUA_OK;
  }
  if (!strcmp(key, "dual_store")) {
    n->dual_store = value;
    return MAUA_OK;
  }
  if (!strcmp(key, "walk_segs")) {
    n->walk_segs = value;
    return MAUA_OK;
  }
  if (!strcmp(key, "walk_narrow")) {
    n->walk_narrow = value;
    return MAUA_OK;
  }
  if (!strcmp(key, "tconv_min")) {
    n->tconv_min = value;
    return MAUA_OK;
  }
  if (!strcmp(key, "tconv_fir")) {
    n->tconv_fir = value;
    return MAUA_OK;
  }
  if (!strcmp(key, "tconv_up")) {
    n->tconv_up = value;
    return MAUA_OK;
  }
  if (!strcmp(key, "tconv_dma")) {
    n->tconv_dma = value;
    return MAUA_OK;
  }
  if (!strcmp(key, "dma_conv")) {
    n->dma_conv = value;
    return MAUA_OK;
  }
  if (!strcmp(key, "fuse_torgb")) {
    n->fuse_torgb = value;
    return MAUA_OK;
  }
  if (!strcmp(key, "keep_features")) {
    if (n->keep_features != value) {
      MAUA_HIP_CHECK(hipStreamSynchronize(n->ctx->stream));
      free_workspace(n);
      n->keep_features = value;
    }
    return MAUA_OK;
  }
  return fail(std::string("maua_synth_set_option: unknown option ") + key);
}

// where maua_synth_load's source lives: host memory (maua_synth_load) or this device (maua_synth_load_device, large tensors)
static thread_local hipMemcpyKind g_load_kind = hipMemcpyHostToDevice;

static int upload(float* dst, const float* host, size_t count, size_t expect, const char* name) {
  if (count != expect)
    return fail(std::string("maua_synth_load: ") + name + ": expected " + std::to_string(expect) + " values, got " +
                std::to_string(count));
  MAUA_HIP_CHECK(hipMemcpy(dst, host, count * sizeof(float), g_load_kind));
  return MAUA_OK;
}

int maua_synth_load(maua_synth* n, const char* name, const float* host, size_t count) {
  MAUA_REQUIRE(n && name && host, "maua_synth_load: NULL argument");
  std::string s(name);
  int blk = -1, pos = 0;
  if (sscanf(name, "bs.%d.%n", &blk, &pos) < 1 || blk < 0 || blk >= n->nblocks || pos == 0)
    return fail("maua_synth_load: unknown parameter name: " + s);
  std::string rest = s.substr(pos);
  hipStream_t st = n->ctx->stream;
  auto check_filter = [&]() -> int {
    if (count != 16) return fail("maua_synth_load: " + s + ": only the 4x4 [1,3,3,1] resample filter is supported");
    const float t[4] = {1, 3, 3, 1};
    for (int u = 0; u < 4; u++)
      for (int v = 0; v < 4; v++)
        if (std::fabs(host[u * 4 + v] - t[u] * t[v] / 64.f) > 1e-6f)
          return fail("maua_synth_load: " + s + ": only the 4x4 [1,3,3,1] resample filter is supported");
    return MAUA_OK;
  };
  if (rest == "resample_filter") return check_filter();
  if (rest == "const") {
    if (blk != 0) return fail("maua_synth_load: only block 0 has a const input");
    int C = n->convs[0].Ci;
    if (count != (size_t)C * 16) return fail("maua_synth_load: bs.0.const: wrong size");
    float* tmp;
    MAUA_HIP_CHECK(hipMalloc((void**)&tmp, count * 4));
    MAUA_HIP_CHECK(hipMemcpy(tmp, host, count * 4, g_load_kind));
    int rc = n->dtype == MAUA_BF16   ? launch_nchw_to_nhwc<float, bf16_t>(st, tmp, n->const_x, 1, C, 16, C)
             : n->dtype == MAUA_F16 ? launch_nchw_to_nhwc<float, f16_t>(st, tmp, n->const_x, 1, C, 16, C)
                                    : launch_nchw_to_nhwc<float, float>(st, tmp, n->const_x, 1, C, 16, C);
    hipStreamSynchronize(st);
    hipFree(tmp);
    return rc;
  }
  size_t dot = rest.find('.');
  if (dot == std::string::npos) return fail("maua_synth_load: unknown parameter name: " + s);
  std::string mod = rest.substr(0, dot), par = rest.substr(dot + 1);
  if (mod == "torgb") {
    RgbLayer& g = n->rgbs[blk];
    if (par == "weight") return upload(g.wrgb, host, count, (size_t)3 * g.C, name);
    if (par == "bias") return upload(g.bias, host, count, 3, name);
    if (par == "affine.weight") return upload(g.affine_w, host, count, (size_t)g.C * n->w_dim, name);
    if (par == "affine.bias") return upload(g.affine_b, host, count, g.C, name);
    return fail("maua_synth_load: unknown parameter name: " + s);
  }
  ConvLayer* c = nullptr;
  for (auto& cc : n->convs)
    if (cc.block == blk && ((mod == "conv0" && cc.which == 0) || (mod == "conv1" && cc.which == 1))) c = &cc;
  if (!c) return fail("maua_synth_load: unknown parameter name: " + s);
  if (par == "resample_filter") return check_filter();
  if (par == "bias") return upload(c->bias, host, count, c->Co, name);
  if (par == "affine.weight") return upload(c->affine_w, host, count, (size_t)c->Ci * n->w_dim, name);
  if (par == "affine.bias") return upload(c->affine_b, host, count, c->Ci, name);
  if (par == "noise_const") return upload(c->noise_const, host, count, (size_t)c->oh * c->ow, name);
  if (par == "noise_strength") {
    if (count != 1) return fail("maua_synth_load: noise_strength is a scalar");
    c->noise_strength = host[0];
    return MAUA_OK;
  }
  if (par == "weight") {
    size_t expect = (size_t)c->Co * c->Ci * 9;
    if (count != expect) return fail("maua_synth_load: " + s + ": wrong size");
    float* tmp;
    MAUA_HIP_CHECK(hipMalloc((void**)&tmp, count * 4));
    MAUA_HIP_CHECK(hipMemcpy(tmp, host, count * 4, g_load_kind));
    // F16 networks: the reference's FP16 pre-normalisation of the weight (ops.py:161-163; every convolution of the synthesis
    // network demodulates) - the styles' half is applied per batch (launch_styles)
    if (n->dtype == MAUA_F16)
      if (int rc0 = launch_f16_prenorm_weights(st, tmp, tmp, c->Co, c->Ci, 9)) { hipFree(tmp); return rc0; }
    int rc = launch_prep_weights(st, n->dtype, tmp, c->wt, c->wsq, c->Co, c->Ci, 3, c->up,
                                 (c->up == 2) ? (n->nv_compat & 1) : 0, c->Co, c->Ci);
    if (!rc && c->up == 2)
      rc = launch_prep_tconv_weights(st, n->dtype, tmp, c->wt_t, c->Co, c->Ci, n->nv_compat & 1);
    if (!rc && c->wt_h) rc = launch_prep_upwalk_weights(st, tmp, c->wt_h, c->Co, c->Ci, n->nv_compat & 1, n->dtype);
    hipStreamSynchronize(st);
    hipFree(tmp);
    return rc;
  }
  return fail("maua_synth_load: unknown parameter name: " + s);
}

// The same from DEVICE memory (parameters drawn on the device: maua_philox_normal): large tensors are copied device to device,
// small ones (filters, scalars, <= 4096 values: the loader inspects them on the host) take a detour through a host buffer.
int maua_synth_load_device(maua_synth* n, const char* name, const float* dev, size_t count) {
  MAUA_REQUIRE(n && name && dev, "maua_synth_load_device: NULL argument");
  MAUA_HIP_CHECK(hipStreamSynchronize(n->ctx->stream));   // (the producer of `dev` ran on the context's stream)
  if (count <= 4096) {
    std::vector<float> h(count);
    MAUA_HIP_CHECK(hipMemcpy(h.data(), dev, count * 4, hipMemcpyDeviceToHost));
    return maua_synth_load(n, name, h.data(), count);
  }
  g_load_kind = hipMemcpyDeviceToDevice;
  const int rc = maua_synth_load(n, name, dev, count);
  g_load_kind = hipMemcpyHostToDevice;
  return rc;
}

// does this up-layer run the LDS-direct transposed-conv kernel?  (same routing conditions as in the forward below)
static bool up_uses_tconv_dma(const maua_synth* n, const ConvLayer& c) {
  if (!n->tconv_dma || n->tconv_up != 1 || c.up != 2) return false;
  const bool hires_up = n->use_hires && hires_supported(n->dtype, c.Ci, c.Co, c.up, c.ih, c.iw);
  const int hin = std::min(c.ih, c.iw), hmax = std::max(c.ih, c.iw);
  return !hires_up && hin >= n->tconv_min && hmax <= 512 && tconv_dma_supported(n->dtype, c.Ci, c.Co, c.ih, c.iw);
}

int maua_synth_render_rgb8(maua_synth* n, const float* ws, const float* const* noise, const long* noise_bstride, int B,
                           float* img_out, uint8_t* rgb8_out) {
  MAUA_REQUIRE(n && ws, "maua_synth_forward: NULL argument");
  MAUA_REQUIRE(B >= 0, "maua_synth_forward: negative batch");
  // the per-sample noise factors (maua_synth_set_noise_scale) belong to THIS call: forgotten on every way out, so that a caller
  // cannot leak one batch's factors into a later forward
  struct ForgetScales { maua_synth* n; ~ForgetScales() { n->nz_scales = nullptr; n->nz_scale_stride = 0; } } forget{n};
  MAUA_REQUIRE(img_out || rgb8_out, "maua_synth_forward: no output buffer");
  if (B == 0) return MAUA_OK;
  if (int rc = ensure_workspace(n, B)) return rc;
  hipStream_t st = n->ctx->stream;
  const int ntab = (int)(n->convs.size() + n->rgbs.size());
  int max_c = 0;
  for (auto& c : n->convs) max_c = std::max(max_c, std::max(c.Ci, c.Co));
  // events accumulate across forwards until maua_synth_get_profile() reads and resets them
  if (n->profile) n->ev_fwd_start.push_back(n->ev_used);
  prof_mark(n, "begin");
  if (int rc = launch_styles(st, n->style_table_dev, ntab, ws, n->num_ws, n->w_dim, B, max_c, n->dtype == MAUA_F16)) return rc;

  prof_mark(n, "styles");
  const void* x = n->const_x;
  long x_bstride = 0;
  int cur = 0;
  // the one feature-space resize (get_hook's resize(x, feat=True)): bicubic / pad + fill noise, NHWC
  auto resize_feat = [&](const void* src, long src_bstride, int nb, int H, int W, int C, void* dst) -> int {
    ResizeArgs r{};
    r.x = src; r.x_bstride = src_bstride; r.y = dst; r.B = nb; r.H = H; r.W = W; r.C = C; r.oh = n->rs_th; r.ow = n->rs_tw;
    r.mode = n->rs_mode; r.pl = n->rs_pl; r.pt = n->rs_pt; r.how = n->rs_how; r.value = n->rs_value; r.noise = n->rs_noise;
    return launch_resize2d(st, n->dtype, true, r);
  };
  if (n->rs_layer == 0) {  // pre-hook on the first layer: its input (the learned const) is resized
    if (int rc = resize_feat(n->const_x, 0, 1, 4, 4, n->convs[0].Ci, n->const_rs)) return rc;
    x = n->const_rs;
  }
  const float* prev_img = nullptr;
  int img_cur = 0;
  size_t li = 0;
  bool rgb8_done = false;
  bool walk_skip = false;      // the previous up-layer ran the whole block (modconv_upwalk.hip): its conv1 is done
  bool x_premod = false;       // the current x already carries the styles of the conv1 that reads it (modconv_dma.hip)
  bool premod_for_up = false;  // ... of the up-layer that reads it (modconv_tconv_dma.hip)
  bool premod_in_xm = false;   // ... or the premod buffer already holds x times that up-layer's styles (dual store of the conv1 before it)
  for (int blk = 0; blk < n->nblocks; blk++) {
    const int nconv = blk == 0 ? 1 : 2;
    RgbLayer& g = n->rgbs[blk];
    const bool last = blk == n->nblocks - 1;
    float* rgb_out = (last && img_out) ? img_out : n->img[img_cur];
    bool rgb_fused = false;
    for (int k = 0; k < nconv; k++, li++) {
      ConvLayer& c = n->convs[li];
      const float* nz = (noise && noise[li]) ? noise[li] : c.noise_const;
      const long nz_stride = (noise && noise[li]) ? (noise_bstride ? noise_bstride[li] : (long)c.oh * c.ow) : 0;
      const float nz_strength = (n->nv_compat & 2) ? c.noise_strength : 1.f;
      // (un-normalised Loop maps: the factor 1 / (rms + eps) of each sample rides on the noise strength; only with caller-supplied maps)
      const float* nz_scale = (n->nz_scales && noise && noise[li]) ? n->nz_scales + (long)li * n->nz_scale_stride : nullptr;
      const bool hooked = n->rs_layer == (int)li + 1;  // this layer's output is resized before anything reads it
      void* y = hooked ? n->act[cur] : n->keep_features ? c.feat : n->act[cur];
      const int hin = std::min(c.ih, c.iw), hin_max = std::max(c.ih, c.iw);
      const int tconv_max = n->tconv_up == 1 ? 512 : n->tconv_up;  // option value > 1 = largest input size routed
      // (default routing: where the register-stationary kernel exists (bf16 64 -> 32 channels, the 1024^2 layer) its
      //  FIR-folded phase form beats tconv + upfir, whose t round trip is HBM-bound there: 1.33 vs 1.56 ms at B = 32)
      const bool hires_up = c.up == 2 && n->tconv_up == 1 && n->use_hires &&
                            hires_supported(n->dtype, c.Ci, c.Co, c.up, c.ih, c.iw);
      const bool via_tconv = c.up == 2 && n->tconv_up && !hires_up && hin >= (n->tconv_up == 1 ? n->tconv_min : 1) &&
                             hin_max <= tconv_max;
      const bool rs_block = n->rs_layer >= 1 && n->convs[n->rs_layer - 1].block == blk;  // toRGB needs the hook path
      // a translate / zoom / rotate hook on this layer replaces its output before toRGB reads it: no fused toRGB then
      bool warped = false;
      for (int wsl = 0; wsl < 3; wsl++) warped = warped || (n->warp_layer[wsl] == (int)li + 1 && n->warp_minv[wsl]);
      const bool fuse_rgb_ok = c.which == 1 && n->fuse_torgb && !rs_block && !warped;
      // does the conv1 that follows this up-layer take pre-modulated input?  (then the epilogue below multiplies the
      // output by that layer's styles; nothing else reads an up-layer's output)
      const bool premod_in = x_premod;
      x_premod = false;
      const bool premod_up_in = premod_for_up;   // x carries this up-layer's styles (set by the conv1 that produced it)
      premod_for_up = false;
      const bool xm_ready = premod_in_xm;        // ... or that conv1 wrote the scaled copy into the premod buffer
      premod_in_xm = false;
      bool premod_out = false;
      // (producers that can scale their output: the FIR pass of a tconv up-layer, the generic kernel's epilogue)
      const bool generic_up = c.up == 2 && !via_tconv && !hires_up &&
                              !(n->lowres && lowres_supported(n->dtype, c.Ci, c.Co, c.up, c.ih, c.iw));
      if ((via_tconv || generic_up) && n->dma_conv && !hooked && !warped && !n->keep_features && c.which == 0 &&
          li + 1 < n->convs.size()) {
        const ConvLayer& nx = n->convs[li + 1];
        premod_out = nx.block == blk && dma_conv_supported(n->dtype, nx.Ci, nx.Co, nx.up, nx.ih, nx.iw);
      }
      if (premod_in) {
        ConvArgs a{};
        a.x = x; a.x_bstride = x_bstride; a.w = c.wt; a.s = nullptr; a.d = c.d;
        a.noise = nz; a.noise_bstride = nz_stride; a.noise_strength = nz_strength; a.noise_scale = nz_scale;
        a.bias = c.bias; a.y = y;
        a.B = B; a.H = c.ih; a.W = c.iw; a.Ci = c.Ci; a.Co = c.Co; a.up = 1;
        a.act = MAUA_ACT_LRELU; a.alpha = 0.2f; a.gain = std::sqrt(2.0f); a.clamp = 256.f;
        if (fuse_rgb_ok && dma_rgb_fusable(c.Co) && (c.ih % 2) == 0 && (c.iw % 2) == 0) {
          a.rgb_wmod = g.wmod; a.rgb_bias = g.bias; a.rgb_prev = prev_img; a.rgb_out = rgb_out; a.rgb_clamp = 256.f;
          memcpy(a.fir, n->fir, sizeof(a.fir));
          rgb_fused = true;
        }
        // the up-layer that follows reads only these features (the block's toRGB is fused right here): store them
        // already multiplied by its styles when it runs the LDS-direct transposed-conv kernel
        if (rgb_fused && !hooked && !warped && !n->keep_features && li + 1 < n->convs.size()) {
          const ConvLayer& nx = n->convs[li + 1];
          if (nx.up == 2 && up_uses_tconv_dma(n, nx)) {
            a.out_scale = nx.s;
            premod_for_up = true;
          }
        } else if (!rgb_fused && n->dual_store && n->xm && !hooked && !warped && !n->keep_features && li + 1 < n->convs.size()) {
          // (round 5) the 512-channel conv1 layers keep a separate toRGB pass, which reads the PLAIN features: they are stored
          // twice - plain to y, multiplied by the next up-layer's styles into the premod buffer - instead of a pass over y later
          const ConvLayer& nx = n->convs[li + 1];
          bool warped_nx = n->rs_layer == (int)li + 2;
          for (int wsl = 0; wsl < 3; wsl++) warped_nx = warped_nx || (n->warp_layer[wsl] == (int)li + 2 && n->warp_minv[wsl]);
          if (nx.up == 2 && up_uses_tconv_dma(n, nx) && nx.ih == c.oh && nx.iw == c.ow && !warped_nx && n->rs_layer != (int)li + 1) {
            a.out_scale = nx.s;
            a.y_scaled = n->xm;
            premod_in_xm = true;
          }
        }
        if (int rc = launch_modconv_dma(st, a, n->dtype)) return rc;
      } else if (!via_tconv && n->use_hires && hires_supported(n->dtype, c.Ci, c.Co, c.up, c.ih, c.iw)) {
        HiresArgs a{};
        a.x = x; a.w = c.wt; a.s = c.s; a.d = c.d; a.noise = nz; a.noise_bstride = nz_stride;
        a.noise_strength = nz_strength; a.noise_scale = nz_scale; a.bias = c.bias; a.y = y;
        a.B = B; a.H = c.ih; a.W = c.iw; a.Ci = c.Ci; a.Co = c.Co; a.up = c.up;
        a.act = MAUA_ACT_LRELU; a.alpha = 0.2f; a.gain = std::sqrt(2.0f); a.clamp = 256.f;
        if (fuse_rgb_ok) {  // conv1: the block's toRGB + skip rides on the epilogue tile
          a.rgb_wmod = g.wmod; a.rgb_bias = g.bias; a.rgb_prev = prev_img; a.rgb_out = rgb_out; a.rgb_clamp = 256.f;
          memcpy(a.fir, n->fir, sizeof(a.fir));
          rgb_fused = true;
          if (last && rgb8_out) {  // the u8 frame is packed in the same epilogue
            a.rgb8_out = rgb8_out;
            a.rgb_skip_f32 = img_out == nullptr;
            rgb8_done = true;
          }
          // the last block's features have no reader besides the toRGB fused here: skip their HBM store
          if (last && !n->keep_features && !hooked) a.y = nullptr;
        }
        if (walk_skip) {
          // (conv1 of a block that ran as one fused walk: nothing left to launch)
          rgb_fused = true;
          rgb8_done = rgb8_done || (last && rgb8_out);
          walk_skip = false;
        } else if (c.up == 2 && n->upwalk && c.wt_h && !fuse_rgb_ok &&
                   upwalk_supported(n->dtype, c.Ci, c.Co, c.up, c.ih, c.iw)) {  // half the matrix work of the phase form
          a.w = c.wt_h;
          // the last block as ONE walk (conv0 up -> conv1 -> toRGB + skip -> image / u8): when nothing else reads its
          // features (no hooks, no feature capture) they never reach HBM
          bool fused_walk = false;
          if (n->upwalk >= 2 && last && li + 1 < n->convs.size() && !hooked && !warped && !n->keep_features &&
              n->fuse_torgb && !rs_block) {
            ConvLayer& c1 = n->convs[li + 1];
            bool warped1 = n->rs_layer == (int)li + 2;
            for (int wsl = 0; wsl < 3; wsl++) warped1 = warped1 || (n->warp_layer[wsl] == (int)li + 2 && n->warp_minv[wsl]);
            if (c1.block == blk && c1.up == 1 && c1.Ci == c.Co && c1.Co == c.Co && c1.ih == c.oh && c1.iw == c.ow &&
                !warped1 && upwalk_fused_supported(n->dtype, c.Ci, c.Co, c.ih, c.iw)) {
              const float* nz1 = (noise && noise[li + 1]) ? noise[li + 1] : c1.noise_const;
              HiresArgs f{};
              f.x = nullptr; f.w = c1.wt; f.s = c1.s; f.d = c1.d; f.noise = nz1;
              f.noise_bstride = (noise && noise[li + 1]) ? (noise_bstride ? noise_bstride[li + 1] : (long)c1.oh * c1.ow) : 0;
              f.noise_strength = (n->nv_compat & 2) ? c1.noise_strength : 1.f;
              f.noise_scale = (n->nz_scales && noise && noise[li + 1]) ? n->nz_scales + (long)(li + 1) * n->nz_scale_stride : nullptr;
              f.bias = c1.bias; f.y = nullptr;
              f.B = B; f.H = c1.ih; f.W = c1.iw; f.Ci = c1.Ci; f.Co = c1.Co; f.up = 1;
              f.act = MAUA_ACT_LRELU; f.alpha = 0.2f; f.gain = std::sqrt(2.0f); f.clamp = 256.f;
              f.rgb_wmod = g.wmod; f.rgb_bias = g.bias; f.rgb_prev = prev_img; f.rgb_out = rgb_out; f.rgb_clamp = 256.f;
              memcpy(f.fir, n->fir, sizeof(f.fir));
              if (rgb8_out) {
                f.rgb8_out = rgb8_out;
                f.rgb_skip_f32 = img_out == nullptr;
              }
              a.y = nullptr;
              if (int rc = launch_upwalk_fused(st, a, f, n->walk_segs, n->walk_narrow, n->dtype)) return rc;
              fused_walk = true;
              walk_skip = true;
            }
          }
          if (!fused_walk)
            if (int rc = launch_upwalk(st, a, n->dtype)) return rc;
        } else if (int rc = launch_modconv_hires(st, a, n->dtype)) {
          return rc;
        }
      } else if (via_tconv) {
        // (measured: pays off from 32^2 inputs up; below, the extra launch costs more than the MACs it saves,
        //  a tconv_up value > 1 sets the largest routed input size)
        // minimal up-layer: t = conv_transpose2d(x*s, W, stride 2) on the matrix cores, then FIR + epilogue
        ConvArgs a{};
        a.x = x; a.x_bstride = x_bstride; a.w = c.wt_t; a.s = c.s; a.d = nullptr; a.noise = nullptr; a.bias = nullptr;
        a.y = n->tbuf; a.B = B; a.H = c.ih; a.W = c.iw; a.Ci = c.Ci; a.Co = c.Co; a.up = 2;
        bool up_fused = false;
        if (up_uses_tconv_dma(n, c)) {
          // main block on the LDS-direct kernel (input already multiplied by the styles: by the producing conv1, or by
          // a pass over the - small - input here), last row / column of positions on the register-staged kernel
          if (xm_ready) {          // the producing conv1 left the scaled copy in the premod buffer
            a.x = n->xm;
            a.x_bstride = (long)c.ih * c.iw * c.Ci;
          } else if (!premod_up_in) {
            if (int rc = launch_premod_nhwc(st, x, x_bstride, c.s, n->xm, B, (long)c.ih * c.iw, c.Ci, n->dtype)) return rc;
            a.x = n->xm;
            a.x_bstride = (long)c.ih * c.iw * c.Ci;
          }
          a.s = n->ones;
          if (n->tconv_fir > 0 && hin >= n->tconv_fir && tconv_fir_supported(n->dtype, c.Ci, c.Co, c.ih, c.iw)) {
            // the whole layer in one kernel: t never leaves LDS (bit-identical output)
            UpfirArgs u{};
            u.y = y; u.d = c.d; u.noise = nz; u.noise_bstride = nz_stride; u.noise_strength = nz_strength; u.noise_scale = nz_scale;
            u.bias = c.bias; u.B = B; u.H = c.ih; u.W = c.iw; u.Co = c.Co;
            if (premod_out) {
              u.out_scale = n->convs[li + 1].s;
              x_premod = true;
            }
            u.act = MAUA_ACT_LRELU; u.alpha = 0.2f; u.gain = std::sqrt(2.0f); u.clamp = 256.f;
            if (int rc = launch_tconv_fir(st, a, u, n->dtype)) return rc;
            prof_mark(n, "conv0_tconv");   // (two profile slots like the two-launch path: the second measures ~0)
            up_fused = true;
          }
          // (the thin edges first, the main block behind them; running the edges on a side stream beside the main block
          //  measured no different: 8.06 vs 8.08 ms per forward)
          if (up_fused) {
          } else if (n->tconv_dma >= 2) {          // dedicated edge kernel (3 of 9 weight blocks, no tile waste)
            if (int rc = launch_tconv_edges(st, a, n->dtype)) return rc;
          } else {
            ConvArgs e = a;
            e.variant = TCONV_EDGES_ONLY;
            if (int rc = launch_tconv2(st, n->dtype, e)) return rc;
          }
          a.variant = 0;
          if (!up_fused)
            if (int rc = launch_tconv_dma(st, a, n->dtype)) return rc;
        } else if (int rc = launch_tconv2(st, n->dtype, a)) {
          return rc;
        }
        if (!up_fused) {
        prof_mark(n, "conv0_tconv");  // (profile mode: this up-layer occupies two slots)
        UpfirArgs u{};
        u.t = n->tbuf; u.y = y; u.d = c.d; u.noise = nz; u.noise_bstride = nz_stride; u.noise_strength = nz_strength; u.noise_scale = nz_scale;
        u.bias = c.bias; u.B = B; u.H = c.ih; u.W = c.iw; u.Co = c.Co;
        if (premod_out) {
          u.out_scale = n->convs[li + 1].s;
          x_premod = true;
        }
        u.act = MAUA_ACT_LRELU; u.alpha = 0.2f; u.gain = std::sqrt(2.0f); u.clamp = 256.f;
        if (int rc = launch_upfir_epilogue(st, n->dtype, u)) return rc;
        }
      } else {
        ConvArgs a{};
        a.x = x; a.x_bstride = x_bstride; a.w = c.wt; a.s = c.s; a.d = c.d;
        a.noise = nz; a.noise_bstride = nz_stride; a.noise_strength = nz_strength; a.noise_scale = nz_scale;
        a.bias = c.bias; a.y = y;
        a.B = B; a.H = c.ih; a.W = c.iw; a.Ci = c.Ci; a.Co = c.Co; a.up = c.up;
        a.act = MAUA_ACT_LRELU; a.alpha = 0.2f; a.gain = std::sqrt(2.0f); a.clamp = 256.f;
        if (n->lowres && lowres_supported(n->dtype, c.Ci, c.Co, c.up, c.ih, c.iw)) {
          // <= 8x8 input pixels: one GEMM over all samples, split-K (modconv_lowres.hip)
          if (int rc = launch_modconv_lowres(st, n->dtype, a, n->lowres_xm, n->lowres_ws)) return rc;
        } else {
        if (premod_out) {
          a.out_scale = n->convs[li + 1].s;
          x_premod = true;
        }
        if (fuse_rgb_ok && modconv_rgb_fusable(n->dtype, c.Ci, c.Co, c.up, c.ih, c.iw)) {  // the block's toRGB + skip in the epilogue
          a.rgb_wmod = g.wmod; a.rgb_bias = g.bias; a.rgb_prev = prev_img; a.rgb_out = rgb_out; a.rgb_clamp = 256.f;
          memcpy(a.fir, n->fir, sizeof(a.fir));
          rgb_fused = true;
        }
        if (int rc = launch_modconv3x3(st, n->dtype, a)) return rc;
        }
      }
      prof_mark(n, c.which == 0 ? "conv0" : "conv1");
      x = y;
      x_bstride = (long)c.oh * c.ow * c.Co;
      cur ^= 1;
      if (hooked) {  // forward hook: resize(output, feat=True)
        void* dst = n->keep_features ? c.feat : n->act[cur];
        if (int rc = resize_feat(x, x_bstride, B, c.oh, c.ow, c.Co, dst)) return rc;
        x = dst;
        x_bstride = (long)c.fh * c.fw * c.Co;
        cur ^= 1;
      }
      for (int wsl = 0; wsl < 3; wsl++) {  // translate / zoom / rotate hooks on this layer (registered after the
        if (n->warp_layer[wsl] != (int)li + 1 || !n->warp_minv[wsl]) continue;  // resize hook, so they run after it)
        void* dst = n->act[cur];
        if (dst == x) dst = n->act[cur ^ 1];
        if (int rc = launch_warp_affine_nhwc(st, n->dtype, x, dst, n->warp_minv[wsl], B, c.fh, c.fw, c.Co)) return rc;
        if (n->keep_features) {  // the hook's output replaces the layer's
          MAUA_HIP_CHECK(hipMemcpyAsync(c.feat, dst, (size_t)B * c.fh * c.fw * c.Co * n->esize, hipMemcpyDeviceToDevice, st));
        } else {
          x = dst;
          cur = (dst == n->act[0]) ? 1 : 0;
        }
      }
    }
    const bool rs_here = n->rs_layer >= 1 && n->convs[n->rs_layer - 1].block == blk;
    if (rs_here) {
      // the reference's rgb_hook / img_hook around the resized block (get_hook :325-338): toRGB runs on the resized
      // features, its output goes back to the layer's native grid (bicubic back / crop), joins the skip image there,
      // and the block's image is resized forward again (no fill noise on images)
      const ConvLayer& hc = n->convs[n->rs_layer - 1];
      const int nh = hc.oh, nw = hc.ow;  // native grid of the block
      RgbArgs r{};
      r.x = x; r.wmod = g.wmod; r.bias = g.bias; r.prev = nullptr;
      r.out = n->rgb_tmp[0]; r.B = B; r.H = g.h; r.W = g.w; r.C = g.C; r.clamp = 256.f;
      memcpy(r.fir, n->fir, sizeof(r.fir));
      if (int rc = launch_torgb(st, n->dtype, r)) return rc;
      ResizeArgs inv{};
      inv.x = n->rgb_tmp[0]; inv.x_bstride = 3L * g.h * g.w; inv.y = n->rgb_tmp[1]; inv.B = B; inv.H = g.h; inv.W = g.w;
      inv.C = 3; inv.oh = nh; inv.ow = nw; inv.mode = n->rs_mode; inv.pl = -n->rs_pl; inv.pt = -n->rs_pt;
      inv.how = MAUA_PAD_CONSTANT; inv.value = 0.f; inv.noise = nullptr;
      if (int rc = launch_resize2d(st, MAUA_F32, false, inv)) return rc;
      const float* native = n->rgb_tmp[1];
      if (prev_img) {
        if (int rc = launch_skip_add(st, n->rgb_tmp[1], prev_img, n->rgb_tmp[0], B, nh, nw, n->fir)) return rc;
        native = n->rgb_tmp[0];
      }
      ResizeArgs fwd{};
      fwd.x = native; fwd.x_bstride = 3L * nh * nw; fwd.y = rgb_out; fwd.B = B; fwd.H = nh; fwd.W = nw; fwd.C = 3;
      fwd.oh = g.h; fwd.ow = g.w; fwd.mode = n->rs_mode; fwd.pl = n->rs_pl; fwd.pt = n->rs_pt; fwd.how = n->rs_how;
      fwd.value = n->rs_value; fwd.noise = nullptr;
      if (int rc = launch_resize2d(st, MAUA_F32, false, fwd)) return rc;
    } else if (!rgb_fused) {
      RgbArgs r{};
      r.x = x; r.wmod = g.wmod; r.bias = g.bias; r.prev = prev_img;
      r.out = rgb_out; r.B = B; r.H = g.h; r.W = g.w; r.C = g.C; r.clamp = 256.f;
      memcpy(r.fir, n->fir, sizeof(r.fir));
      if (int rc = launch_torgb(st, n->dtype, r)) return rc;
    }
    prof_mark(n, "torgb");  // zero-length when fused into conv1
    prev_img = rgb_out;
    img_cur ^= 1;
  }
  if (rgb8_out) {
    if (!rgb8_done)
      if (int rc = launch_pack_rgb8(st, prev_img, rgb8_out, B, n->out_h, n->out_w)) return rc;
    prof_mark(n, "pack_rgb8");  // zero-length when the last block's epilogue packed the frame
  }
  return MAUA_OK;
}

int maua_synth_set_noise_scale(maua_synth* n, const float* scales, long layer_stride) {
  MAUA_REQUIRE(n, "maua_synth_set_noise_scale: NULL argument");
  MAUA_REQUIRE(!scales || layer_stride > 0, "maua_synth_set_noise_scale: layer_stride must be positive");
  n->nz_scales = scales;
  n->nz_scale_stride = layer_stride;
  return MAUA_OK;
}

int maua_synth_get_profile(maua_synth* n, float* ms_out, int capacity, int* count) {
  MAUA_REQUIRE(n && count, "maua_synth_get_profile: NULL argument");
  // one duration per launch; the "begin" marker of every recorded forward is skipped
  int k = 0;
  if (n->ev_used > 0) MAUA_HIP_CHECK(hipEventSynchronize(n->ev[n->ev_used - 1]));
  for (size_t f = 0; f < n->ev_fwd_start.size(); f++) {
    size_t lo = n->ev_fwd_start[f], hi = (f + 1 < n->ev_fwd_start.size()) ? n->ev_fwd_start[f + 1] : n->ev_used;
    for (size_t i = lo; i + 1 < hi; i++, k++)
      if (ms_out && k < capacity) MAUA_HIP_CHECK(hipEventElapsedTime(&ms_out[k], n->ev[i], n->ev[i + 1]));
  }
  *count = k;
  if (ms_out) {  // reading resets the recording
    n->ev_used = 0;
    n->ev_names.clear();
    n->ev_fwd_start.clear();
  }
  return MAUA_OK;
}

int maua_synth_forward(maua_synth* n, const float* ws, const float* const* noise, const long* noise_bstride, int B,
                       float* img_out) {
  MAUA_REQUIRE(img_out, "maua_synth_forward: img_out is NULL");
  return maua_synth_render_rgb8(n, ws, noise, noise_bstride, B, img_out, nullptr);
}

int maua_synth_get_feature(maua_synth* n, int layer, int B, float* out_nchw) {
  MAUA_REQUIRE(n && out_nchw, "maua_synth_get_feature: NULL argument");
  MAUA_REQUIRE(n->keep_features, "maua_synth_get_feature: enable with maua_synth_set_option(net, \"keep_features\", 1)");
  MAUA_REQUIRE(layer >= 0 && layer < (int)n->convs.size(), "maua_synth_get_feature: no such layer");
  MAUA_REQUIRE(B <= n->bcap, "maua_synth_get_feature: batch larger than the last forward");
  ConvLayer& c = n->convs[layer];
  hipStream_t st = n->ctx->stream;
  if (n->dtype == MAUA_BF16) return launch_nhwc_to_nchw<bf16_t, float>(st, c.feat, out_nchw, B, c.Co, c.fh * c.fw, c.Co);
  if (n->dtype == MAUA_F16) return launch_nhwc_to_nchw<f16_t, float>(st, c.feat, out_nchw, B, c.Co, c.fh * c.fw, c.Co);
  return launch_nhwc_to_nchw<float, float>(st, c.feat, out_nchw, B, c.Co, c.fh * c.fw, c.Co);
}

}  // extern "C"
