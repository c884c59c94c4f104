// RealESRGAN x4 generator (RRDBNet) forward: the per-frame 4x up-scaler of BASELINE configs[4] ("StyleGAN2 render ->
// RealESRGAN 4x"), first slice.
//
// Replaces (reference): maua/super/image/models/realesrgan.py:22-49 (load_model builds basicsr's RRDBNet(3, 3, 64, 23 |
// 6, 32, scale 4) and runs it through RealESRGANer.enhance: [0,1] input -> network -> clamp -> x255 round -> u8) and the
// per-frame loop of maua/super/video/frame_by_frame.py:22-33.  basicsr / realesrgan are un-vendored (setup.py:32,85,
// unpinned): the architecture is the published ESRGAN / Real-ESRGAN one; oracle/super.py restates it.  Parity unpinned.
//
//   RDB :  x1 = lrelu(c1(x)); x2 = lrelu(c2([x,x1])); x3 = lrelu(c3([x,x1,x2])); x4 = lrelu(c4([x..x3]));
//          out = c5([x..x4]) * 0.2 + x                       (growth 32, lrelu slope 0.2)
//   RRDB:  out = rdb3(rdb2(rdb1(x))) * 0.2 + x
//   net :  f = conv_first(img); f = f + conv_body(RRDB^n(f)); f = lrelu(conv_up1(up2(f))); f = lrelu(conv_up2(up2(f)));
//          out = conv_last(lrelu(conv_hr(f)))                (up2 = nearest neighbour x2)
//
// MI355X design: activations stay NHWC in the network dtype for the whole forward; a dense block is ONE buffer of
// num_feat + 4 * grow channels per pixel - every convolution reads a channel prefix of it and writes its own channel
// slice (no torch.cat copies: the conv kernels take pixel strides), the block's scaled residual is fused into conv5's
// epilogue.  The 3x3 convolutions run on the MFMA implicit-GEMM kernels with unit styles (a plain convolution is a
// modulated one with s = 1 and no demodulation): bf16 at H % 8 == 0, W % 32 == 0 on the narrow N tiles of the LDS-direct
// kernel (modconv_dma.hip), everything else on the generic kernel (modconv.hip).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "internal.h"

using namespace maua;

namespace {

struct PlainConv {
  int Ci, Co, Cip, Cop;   // real / padded-to-32 channel counts
  void* wt = nullptr;     // prepared [9][Cop][Cip]
  float* bias = nullptr;  // [Cop] (zero padded)
};

// dst[p][c] = a * x[p][c] + b * y[p][c] over n_pix pixels x C channels (16-byte pieces), each operand with its own pixel
// stride; y may be NULL (b ignored).  The RRDB-level residual and the block-input copy.
template <typename T>
__global__ __launch_bounds__(256) void lincomb_nhwc_kernel(T* __restrict__ dst, int dps, float a, const T* __restrict__ x,
                                                           int xps, float bsc, const T* __restrict__ y, int yps,
                                                           long n_pix, int C) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const int ppp = C / EPC;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_pix * ppp) return;
  const long p = idx / ppp;
  const int c = (int)(idx - p * ppp) * EPC;
  float v[EPC];
#pragma unroll
  for (int e = 0; e < EPC; e++) v[e] = a * Elem<T>::load(x + p * xps + c + e);
  if (y) {
#pragma unroll
    for (int e = 0; e < EPC; e++) v[e] += bsc * Elem<T>::load(y + p * yps + c + e);
  }
#pragma unroll
  for (int e = 0; e < EPC; e++) Elem<T>::store(dst + p * dps + c + e, v[e]);
}

// nearest-neighbour x2 (F.interpolate(scale_factor=2, mode="nearest")) on NHWC, 16-byte pieces
template <typename T>
__global__ __launch_bounds__(256) void upsample2_nearest_nhwc_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H,
                                                                     int W, int C) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const int ppp = C / EPC;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * 4 * H * W * ppp;
  if (idx >= total) return;
  const int pc = (int)(idx % ppp);
  long p = idx / ppp;
  const int ox = (int)(p % (2 * W)); p /= 2 * W;
  const int oy = (int)(p % (2 * H));
  const int b = (int)(p / (2 * H));
  const uint4 v = *reinterpret_cast<const uint4*>(x + (((long)b * H + (oy >> 1)) * W + (ox >> 1)) * C + pc * EPC);
  *reinterpret_cast<uint4*>(y + (((long)b * 2 * H + oy) * 2 * W + ox) * C + pc * EPC) = v;
}

// final image: NHWC network dtype (first 3 of Cp channels) -> planar f32 [B][3][H][W] and / or u8 HWC, clamped to [0,1]
// (RealESRGANer.enhance: output.clamp_(0, 1), then (x * 255).round() for 8-bit images)
template <typename T>
__global__ __launch_bounds__(256) void rrdb_output_kernel(const T* __restrict__ x, int Cp, long HW, int B, int do_clamp,
                                                          float* __restrict__ out_f32, uint8_t* __restrict__ out_u8, int W, int oh,
                                                          int ow) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * HW) return;
  const long b = idx / HW, p = idx - b * HW;
  const int yy = (int)(p / W), xx = (int)(p - (long)yy * W);
#pragma unroll
  for (int c = 0; c < 3; c++) {
    float v = Elem<T>::load(x + idx * Cp + c);
    if (out_f32) out_f32[(b * 3 + c) * HW + p] = do_clamp ? fminf(fmaxf(v, 0.f), 1.f) : v;
    if (out_u8 && yy < oh && xx < ow)   // (the u8 frame may be a cropped window: oh x ow, dense)
      out_u8[((b * oh + yy) * ow + xx) * 3 + c] = (uint8_t)__float2int_rn(fminf(fmaxf(v, 0.f), 1.f) * 255.0f);
  }
}

// u8 HWC frames [B][h][w][3] -> the first convolution's input: NHWC [B][H][W][32] in the network dtype (3 real channels), value / 255,
// REFLECT-padded on the right / bottom to H x W (RealESRGANer.enhance: img / 255, F.pad(.., (0, pre_pad, 0, pre_pad), "reflect")) -
// what enhance_frames used to build with three torch passes (convert, pad, layout)
template <typename T>
__global__ __launch_bounds__(256) void frames_u8_to_nhwc32_kernel(const uint8_t* __restrict__ f, T* __restrict__ y, int B, int h, int w,
                                                                  int H, int W) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * H * W) return;
  const int xx = (int)(idx % W), yy = (int)((idx / W) % H);
  const long b = idx / ((long)W * H);
  const int sy = yy < h ? yy : 2 * (h - 1) - yy, sx = xx < w ? xx : 2 * (w - 1) - xx;
  const uint8_t* src = f + ((b * h + sy) * w + sx) * 3;
  T* dst = y + idx * 32;
#pragma unroll
  for (int c = 0; c < 32; c++) Elem<T>::store(dst + c, c < 3 ? (float)src[c] / 255.0f : 0.f);
}

}  // namespace

struct maua_rrdbnet {
  maua_ctx* ctx;
  int num_feat, num_block, grow, dtype;
  size_t esize;
  PlainConv conv_first, conv_body, conv_up1, conv_up2, conv_hr, conv_last;
  std::vector<PlainConv> rdb;  // [block][3 rdb][5 conv]
  int use_dma = 1;             // bf16 trunk convolutions on the LDS-direct kernel (MAUA_RRDB_DMA=0: the generic kernel)
  float* ones = nullptr;       // unit styles [bcap][max Ci]
  int ones_b = 0;
  // workspace (grow-only)
  size_t cap_px = 0;           // B * H * W the buffers were sized for
  void *in32 = nullptr, *feat0 = nullptr, *dense[3] = {nullptr, nullptr, nullptr}, *f1 = nullptr, *up1 = nullptr,
       *f2 = nullptr, *up2 = nullptr, *f3 = nullptr, *f4 = nullptr, *f5 = nullptr;
};

static int alloc_conv(PlainConv& c, int Ci, int Co, size_t esize) {
  c.Ci = Ci; c.Co = Co; c.Cip = (Ci + 31) / 32 * 32; c.Cop = (Co + 31) / 32 * 32;
  // bf16: K in whole 64-channel chunks (zero weights beyond Ci) so that the trunk runs on the LDS-direct kernel; the
  // channels read beyond Ci belong to the same dense-block buffer (zero-initialised, only ever finite)
  // (32 output channels: the narrow tile also takes an odd number of 32-channel chunks, so 96 / 160-channel layers are not padded)
  if (esize == 2 && Ci >= 64 && !(c.Cop == 32 && Ci >= 96)) c.Cip = (Ci + 63) / 64 * 64;
  MAUA_HIP_CHECK(hipMalloc(&c.wt, (size_t)9 * c.Cop * c.Cip * esize));
  MAUA_HIP_CHECK(hipMemset(c.wt, 0, (size_t)9 * c.Cop * c.Cip * esize));
  MAUA_HIP_CHECK(hipMalloc((void**)&c.bias, (size_t)c.Cop * 4));
  MAUA_HIP_CHECK(hipMemset(c.bias, 0, (size_t)c.Cop * 4));
  return MAUA_OK;
}

static void free_ws(maua_rrdbnet* n) {
  void** ps[] = {&n->in32, &n->feat0, &n->dense[0], &n->dense[1], &n->dense[2], &n->f1, &n->up1, &n->f2, &n->up2, &n->f3,
                 &n->f4, &n->f5};
  for (void** p : ps) {
    if (*p) hipFree(*p);
    *p = nullptr;
  }
  n->cap_px = 0;
}

extern "C" {

int maua_rrdb_create(maua_ctx* ctx, int num_feat, int num_block, int num_grow_ch, int dtype, maua_rrdbnet** out) {
  MAUA_REQUIRE(ctx && out, "maua_rrdb_create: NULL argument");
  MAUA_REQUIRE(dtype == MAUA_F32 || dtype == MAUA_BF16, "maua_rrdb_create: dtype must be MAUA_F32 or MAUA_BF16");
  MAUA_REQUIRE(num_feat % 32 == 0 && num_grow_ch % 32 == 0 && num_feat > 0 && num_grow_ch > 0 && num_block > 0,
               "maua_rrdb_create: num_feat / num_grow_ch must be positive multiples of 32");
  // a dense-block convolution reads whole 64-channel K chunks of the block's buffer (zero weights beyond its own Ci):
  // every padded prefix must stay inside the num_feat + 4 * grow channels of a pixel
  for (int k = 0; k < 5; k++)
    MAUA_REQUIRE((num_feat + k * num_grow_ch + 63) / 64 * 64 <= num_feat + 4 * num_grow_ch || dtype != MAUA_BF16,
                 "maua_rrdb_create: num_feat + k * num_grow_ch padded to 64 exceeds the dense-block width (use 64 / 32)");
  maua_rrdbnet* n = new maua_rrdbnet();
  n->ctx = ctx; n->num_feat = num_feat; n->num_block = num_block; n->grow = num_grow_ch; n->dtype = dtype;
  n->esize = dtype == MAUA_BF16 ? 2 : 4;
  if (const char* e = getenv("MAUA_RRDB_DMA")) n->use_dma = atoi(e);
  int rc = alloc_conv(n->conv_first, 3, num_feat, n->esize);
  n->rdb.resize((size_t)num_block * 15);
  for (int i = 0; i < num_block * 3 && !rc; i++)
    for (int k = 0; k < 5 && !rc; k++)
      rc = alloc_conv(n->rdb[(size_t)i * 5 + k], num_feat + k * num_grow_ch, k == 4 ? num_feat : num_grow_ch, n->esize);
  if (!rc) rc = alloc_conv(n->conv_body, num_feat, num_feat, n->esize);
  if (!rc) rc = alloc_conv(n->conv_up1, num_feat, num_feat, n->esize);
  if (!rc) rc = alloc_conv(n->conv_up2, num_feat, num_feat, n->esize);
  if (!rc) rc = alloc_conv(n->conv_hr, num_feat, num_feat, n->esize);
  if (!rc) rc = alloc_conv(n->conv_last, num_feat, 3, n->esize);
  if (rc) {
    maua_rrdb_destroy(n);
    return rc;
  }
  *out = n;
  return MAUA_OK;
}

void maua_rrdb_destroy(maua_rrdbnet* n) {
  if (!n) return;
  hipStreamSynchronize(n->ctx->stream);
  free_ws(n);
  auto fc = [](PlainConv& c) {
    if (c.wt) hipFree(c.wt);
    if (c.bias) hipFree(c.bias);
  };
  fc(n->conv_first); fc(n->conv_body); fc(n->conv_up1); fc(n->conv_up2); fc(n->conv_hr); fc(n->conv_last);
  for (auto& c : n->rdb) fc(c);
  if (n->ones) hipFree(n->ones);
  delete n;
}

// parameter names of basicsr's RRDBNet state dict: conv_first | conv_body | conv_up1 | conv_up2 | conv_hr | conv_last
// | body.<i>.rdb<1..3>.conv<1..5>, each .weight ([Co][Ci][3][3]) or .bias ([Co])
int maua_rrdb_load(maua_rrdbnet* n, const char* name, const float* host, size_t count) {
  MAUA_REQUIRE(n && name && host, "maua_rrdb_load: NULL argument");
  std::string s(name);
  const size_t dot = s.rfind('.');
  MAUA_REQUIRE(dot != std::string::npos, "maua_rrdb_load: unknown parameter name");
  const std::string mod = s.substr(0, dot), par = s.substr(dot + 1);
  PlainConv* c = nullptr;
  int blk = -1, r = -1, k = -1;
  if (mod == "conv_first") c = &n->conv_first;
  else if (mod == "conv_body") c = &n->conv_body;
  else if (mod == "conv_up1") c = &n->conv_up1;
  else if (mod == "conv_up2") c = &n->conv_up2;
  else if (mod == "conv_hr") c = &n->conv_hr;
  else if (mod == "conv_last") c = &n->conv_last;
  else if (sscanf(mod.c_str(), "body.%d.rdb%d.conv%d", &blk, &r, &k) == 3 && blk >= 0 && blk < n->num_block && r >= 1 &&
           r <= 3 && k >= 1 && k <= 5)
    c = &n->rdb[((size_t)blk * 3 + (r - 1)) * 5 + (k - 1)];
  if (!c) return fail("maua_rrdb_load: unknown parameter name: " + s);
  hipStream_t st = n->ctx->stream;
  if (par == "bias") {
    if (count != (size_t)c->Co) return fail("maua_rrdb_load: " + s + ": wrong size");
    MAUA_HIP_CHECK(hipMemcpy(c->bias, host, count * 4, hipMemcpyHostToDevice));
    return MAUA_OK;
  }
  if (par != "weight") return fail("maua_rrdb_load: unknown parameter name: " + s);
  if (count != (size_t)c->Co * c->Ci * 9) return fail("maua_rrdb_load: " + s + ": wrong size");
  float* tmp;
  MAUA_HIP_CHECK(hipMalloc((void**)&tmp, count * 4));
  MAUA_HIP_CHECK(hipMemcpy(tmp, host, count * 4, hipMemcpyHostToDevice));
  MAUA_HIP_CHECK(hipMemsetAsync(c->wt, 0, (size_t)9 * c->Cop * c->Cip * n->esize, st));
  int rc = launch_prep_weights(st, n->dtype, tmp, c->wt, nullptr, c->Co, c->Ci, 3, 1, 0, c->Cop, c->Cip);
  hipStreamSynchronize(st);
  hipFree(tmp);
  return rc;
}

}  // extern "C"

namespace {

template <typename T>
int forward_t(maua_rrdbnet* n, const float* img, int B, int H, int W, int do_clamp, float* out_f32, uint8_t* out_u8,
              const uint8_t* frames = nullptr, int fh = 0, int fw = 0) {
  hipStream_t st = n->ctx->stream;
  const int F = n->num_feat, G = n->grow, D = F + 4 * G;
  const size_t es = n->esize;
  const size_t px = (size_t)B * H * W;
  if (px > n->cap_px) {
    MAUA_HIP_CHECK(hipStreamSynchronize(st));
    free_ws(n);
    MAUA_HIP_CHECK(hipMalloc(&n->in32, px * 32 * es));
    MAUA_HIP_CHECK(hipMalloc(&n->feat0, px * F * es));
    for (int i = 0; i < 3; i++) {
      MAUA_HIP_CHECK(hipMalloc(&n->dense[i], px * D * es));
      MAUA_HIP_CHECK(hipMemsetAsync(n->dense[i], 0, px * D * es, st));
    }
    MAUA_HIP_CHECK(hipMalloc(&n->f1, px * F * es));
    MAUA_HIP_CHECK(hipMalloc(&n->f2, px * 4 * F * es));
    MAUA_HIP_CHECK(hipMalloc(&n->f3, px * 16 * F * es));
    MAUA_HIP_CHECK(hipMalloc(&n->f4, px * 16 * F * es));
    n->cap_px = px;
  }
  if (B > n->ones_b) {  // unit styles: [B][D] floats of 1
    MAUA_HIP_CHECK(hipStreamSynchronize(st));
    if (n->ones) hipFree(n->ones);
    std::vector<float> h((size_t)B * D, 1.f);
    MAUA_HIP_CHECK(hipMalloc((void**)&n->ones, h.size() * 4));
    MAUA_HIP_CHECK(hipMemcpy(n->ones, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    n->ones_b = B;
  }
  // a plain 3x3 convolution on channel slices: x = first c.Cip channels of a buffer with xps channels per pixel,
  // y = c.Cop channels at channel offset ycoff of a buffer with yps per pixel (+ optional residual, yps-strided source)
  auto conv = [&](const PlainConv& c, const void* x, int xps, void* y, int yps, int ycoff, int h, int w, bool lrelu,
                  float gain, const void* res, int rps, const void* res2 = nullptr, int r2ps = 0, float res_gain = 1.f) -> int {
    ConvArgs a{};
    a.x = x; a.x_bstride = (long)h * w * xps; a.x_pstride = xps; a.w = c.wt; a.s = n->ones; a.d = nullptr;
    a.noise = nullptr; a.bias = c.bias; a.y = y; a.y_pstride = yps; a.y_coff = ycoff; a.y_bstride = (long)h * w * yps;
    a.B = B; a.H = h; a.W = w; a.Ci = c.Cip; a.Co = c.Cop; a.up = 1;
    a.act = lrelu ? MAUA_ACT_LRELU : MAUA_ACT_LINEAR; a.alpha = 0.2f; a.gain = gain; a.clamp = -1.f;
    a.res = res; a.res_pstride = rps; a.res_bstride = (long)h * w * rps;
    a.res2 = res2; a.res2_pstride = r2ps; a.res2_bstride = (long)h * w * r2ps; a.res_gain = res_gain;
    a.Ci_read = std::max(c.Ci, 64);   // (the zero-weight padding of K is not fetched; at least one chunk pair)
    // 32 / 64 output channels with K in 64-channel chunks: both operands by LDS-direct loads (modconv_dma.hip narrow tiles)
    if (n->use_dma && dma_conv_narrow_supported(n->dtype, c.Cip, c.Cop, h, w)) return launch_modconv_dma(st, a);
    return launch_modconv3x3(st, n->dtype, a);
  };
  auto lincomb = [&](void* dst, int dps, float aa, const void* x, int xps, float bb, const void* y, int yps, long npix,
                     int C) -> int {
    const long total = npix * (C / (16 / (int)sizeof(T)));
    hipLaunchKernelGGL(lincomb_nhwc_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (T*)dst, dps, aa,
                       (const T*)x, xps, bb, (const T*)y, yps, npix, C);
    MAUA_HIP_CHECK(hipGetLastError());
    return MAUA_OK;
  };
  auto up2 = [&](const void* x, void* y, int h, int w) -> int {
    const long total = (long)B * 4 * h * w * (F / (16 / (int)sizeof(T)));
    hipLaunchKernelGGL(upsample2_nearest_nhwc_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const T*)x, (T*)y, B, h, w, F);
    MAUA_HIP_CHECK(hipGetLastError());
    return MAUA_OK;
  };

  int rc = MAUA_OK;
  const int oh = frames ? 4 * fh : 4 * H, ow = frames ? 4 * fw : 4 * W;   // the u8 frame's size (frames: the un-padded input's x4)
  if (frames) {
    const long total = (long)B * H * W;
    hipLaunchKernelGGL(frames_u8_to_nhwc32_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, frames, (T*)n->in32, B, fh, fw,
                       H, W);
    MAUA_HIP_CHECK(hipGetLastError());
  } else {
    rc = launch_nchw_to_nhwc<float, T>(st, img, n->in32, B, 3, H * W, 32);
  }
  if (rc) return rc;
  if ((rc = conv(n->conv_first, n->in32, 32, n->feat0, F, 0, H, W, false, 1.f, nullptr, 0))) return rc;
  // the trunk: block input in channels [0, F) of dense[ia].  Three dense-block buffers rotate: a block's input stays untouched in
  // [0, F) of its buffer (the dense blocks only write the growth slices there) until the block's last conv5 adds it back, so
  // neither a saved copy of the input nor a separate "out * 0.2 + x" pass is needed - both ride on that conv5's epilogue.
  int ia = 0, ib = 1, ic = 2;
  if ((rc = lincomb(n->dense[ia], D, 1.f, n->feat0, F, 0.f, nullptr, 0, (long)px, F))) return rc;
  for (int blk = 0; blk < n->num_block; blk++) {
    const int src[3] = {ia, ib, ic}, dst[3] = {ib, ic, ib};
    for (int r = 0; r < 3; r++) {
      const PlainConv* cs = &n->rdb[((size_t)blk * 3 + r) * 5];
      char* db = (char*)n->dense[src[r]];
      for (int k = 0; k < 4; k++)  // x_{k+1} = lrelu(conv_{k+1}(prefix)) -> its own channel slice of the same buffer
        if ((rc = conv(cs[k], db, D, db, D, F + k * G, H, W, true, 1.f, nullptr, 0))) return rc;
      // out = conv5(all) * 0.2 + x  -> the next dense block's input slice; the third one also: RRDB out = out * 0.2 + block input
      if (r < 2) {
        if ((rc = conv(cs[4], db, D, n->dense[dst[r]], D, 0, H, W, false, 0.2f, db, D))) return rc;
      } else if ((rc = conv(cs[4], db, D, n->dense[dst[r]], D, 0, H, W, false, 0.2f, db, D, n->dense[ia], D, 0.2f))) {
        return rc;
      }
    }
    const int t = ia;   // the next block's input sits in [0, F) of dense[ib]
    ia = ib; ib = ic; ic = t;
  }
  // feat = feat0 + conv_body(trunk)
  if ((rc = conv(n->conv_body, n->dense[ia], D, n->f1, F, 0, H, W, false, 1.f, n->feat0, F))) return rc;
  // conv_up(interpolate(x, 2, "nearest")): on the LDS-direct kernel the up-sampled tensor is never materialised (its halo loads
  // address the half-size source); otherwise one repetition pass in front of the generic kernel
  auto conv_up = [&](const PlainConv& c, const void* src, void** tmp, void* dst, int h, int w) -> int {   // h, w: source size
    if (n->use_dma && dma_conv_narrow_supported(n->dtype, c.Cip, c.Cop, 2 * h, 2 * w)) {
      ConvArgs a{};
      a.x = src; a.x_bstride = (long)h * w * F; a.x_pstride = F; a.x_up2 = 1; a.w = c.wt; a.s = n->ones; a.bias = c.bias;
      a.y = dst; a.y_pstride = F; a.y_bstride = (long)4 * h * w * F;
      a.B = B; a.H = 2 * h; a.W = 2 * w; a.Ci = c.Cip; a.Co = c.Cop; a.up = 1;
      a.act = MAUA_ACT_LRELU; a.alpha = 0.2f; a.gain = 1.f; a.clamp = -1.f;
      return launch_modconv_dma(st, a);
    }
    // (only this path needs the up-sampled tensor; sized like every other buffer, for the capacity in pixels)
    if (!*tmp) MAUA_HIP_CHECK(hipMalloc(tmp, n->cap_px * (size_t)(4 * h * w / (H * W)) * F * es));
    if (int r2 = up2(src, *tmp, h, w)) return r2;
    return conv(c, *tmp, F, dst, F, 0, 2 * h, 2 * w, true, 1.f, nullptr, 0);
  };
  if ((rc = conv_up(n->conv_up1, n->f1, &n->up1, n->f2, H, W))) return rc;
  if ((rc = conv_up(n->conv_up2, n->f2, &n->up2, n->f3, 2 * H, 2 * W))) return rc;
  if ((rc = conv(n->conv_hr, n->f3, F, n->f4, F, 0, 4 * H, 4 * W, true, 1.f, nullptr, 0))) return rc;
  if (n->use_dma && n->conv_last.Cop == 32 && dma_conv_narrow_supported(n->dtype, n->conv_last.Cip, 32, 4 * H, 4 * W)) {
    // conv_last writes the image itself (clamp, f32 planes / u8 frame in its epilogue): no 32-channel tensor, no output pass
    const PlainConv& c = n->conv_last;
    ConvArgs a{};
    a.x = n->f4; a.x_bstride = (long)16 * H * W * F; a.x_pstride = F; a.w = c.wt; a.s = n->ones; a.bias = c.bias;
    a.B = B; a.H = 4 * H; a.W = 4 * W; a.Ci = c.Cip; a.Co = 32; a.up = 1;
    a.act = MAUA_ACT_LINEAR; a.alpha = 0.2f; a.gain = 1.f; a.clamp = -1.f;
    a.img_f32 = out_f32; a.img_u8 = out_u8; a.img_clamp = do_clamp; a.img_h = oh; a.img_w = ow;
    return launch_modconv_dma(st, a);
  }
  if (!n->f5) MAUA_HIP_CHECK(hipMalloc(&n->f5, n->cap_px * 16 * 32 * es));
  if ((rc = conv(n->conv_last, n->f4, F, n->f5, 32, 0, 4 * H, 4 * W, false, 1.f, nullptr, 0))) return rc;
  const long opx = (long)B * 16 * H * W;
  hipLaunchKernelGGL(rrdb_output_kernel<T>, dim3((unsigned)((opx + 255) / 256)), dim3(256), 0, st, (const T*)n->f5, 32,
                     (long)16 * H * W, B, do_clamp, out_f32, out_u8, 4 * W, oh, ow);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace

// RealESRGANer.enhance for a batch of device-resident u8 frames in ONE call (BASELINE configs[4]: render -> x4 per frame): frames
// [B][h][w][3] u8 -> out [B][4h][4w][3] u8 = round(255 clamp(net(reflect_pad(frames / 255, pre_pad)), 0, 1)) cropped to 4h x 4w - the
// conversion and the padding ride on the first convolution's input staging, the clamp / round / crop on the last one's store
extern "C" int maua_rrdb_enhance_u8(maua_rrdbnet* n, const uint8_t* frames, int B, int h, int w, int pre_pad, uint8_t* out_rgb8) {
  MAUA_REQUIRE(n && frames && out_rgb8, "maua_rrdb_enhance_u8: NULL argument");
  MAUA_REQUIRE(B >= 0 && h > 0 && w > 0 && pre_pad >= 0 && pre_pad < h && pre_pad < w, "maua_rrdb_enhance_u8: bad shape (reflect padding needs pre_pad < h, w)");
  if (B == 0) return MAUA_OK;
  const int H = h + pre_pad, W = w + pre_pad;
  return n->dtype == MAUA_BF16 ? forward_t<bf16_t>(n, nullptr, B, H, W, 1, nullptr, out_rgb8, frames, h, w)
                               : forward_t<float>(n, nullptr, B, H, W, 1, nullptr, out_rgb8, frames, h, w);
}

extern "C" int maua_rrdb_forward_ex(maua_rrdbnet* n, const float* img_nchw, int B, int H, int W, int clamp01, float* out_nchw,
                                    uint8_t* out_rgb8) {
  MAUA_REQUIRE(n && img_nchw, "maua_rrdb_forward: NULL argument");
  MAUA_REQUIRE(out_nchw || out_rgb8, "maua_rrdb_forward: no output buffer");
  MAUA_REQUIRE(B >= 0 && H > 0 && W > 0, "maua_rrdb_forward: bad shape");
  if (B == 0) return MAUA_OK;
  return n->dtype == MAUA_BF16 ? forward_t<bf16_t>(n, img_nchw, B, H, W, clamp01, out_nchw, out_rgb8)
                               : forward_t<float>(n, img_nchw, B, H, W, clamp01, out_nchw, out_rgb8);
}
extern "C" int maua_rrdb_forward(maua_rrdbnet* n, const float* img_nchw, int B, int H, int W, float* out_nchw,
                                 uint8_t* out_rgb8) {
  return maua_rrdb_forward_ex(n, img_nchw, B, H, W, 1, out_nchw, out_rgb8);
}

// ================================================================================================ SRVGGNetCompact
// Replaces (reference): maua/super/image/models/realesrgan.py:34-35 - the "xsx4-animevideo" model,
// realesrgan.archs.srvgg_arch.SRVGGNetCompact(3, 3, num_feat 64, num_conv 16, upscale 4, act_type "prelu") (un-vendored:
// published architecture, oracle/super.py restates it; parity unpinned):
//     out = conv_last(act(conv_k(... act(conv_0(x))))) -> PixelShuffle(upscale) -> + nearest_upsample(x, upscale)
// Every convolution runs on the MFMA conv kernels with the PReLU slopes in the epilogue (ConvArgs.prelu); the pixel shuffle,
// the nearest-neighbour base image, the clamp and the u8 pack are ONE output pass over the last convolution's NHWC tile.
namespace {

// y[c][s y + dy][s x + dx] = feat[y][x][c s^2 + dy s + dx] + img[c][y][x]   (PixelShuffle + F.interpolate(nearest))
template <typename T>
__global__ __launch_bounds__(256) void srvgg_output_kernel(const T* __restrict__ feat, int Cp, const float* __restrict__ img,
                                                           int B, int H, int W, int S, int do_clamp,
                                                           float* __restrict__ out_f32, uint8_t* __restrict__ out_u8) {
  const long Ho = (long)H * S, Wo = (long)W * S;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * Ho * Wo) return;
  const int ox = (int)(idx % Wo);
  const long t = idx / Wo;
  const int oy = (int)(t % Ho), b = (int)(t / Ho);
  const int x = ox / S, dx = ox - x * S, y = oy / S, dy = oy - y * S;
  const T* fp = feat + (((long)b * H + y) * W + x) * Cp;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    float v = Elem<T>::load(fp + c * S * S + dy * S + dx) + img[(((long)b * 3 + c) * H + y) * W + x];
    if (do_clamp) v = fminf(fmaxf(v, 0.f), 1.f);
    if (out_f32) out_f32[(((long)b * 3 + c) * Ho + oy) * Wo + ox] = v;
    if (out_u8) out_u8[idx * 3 + c] = (uint8_t)__float2int_rn(fminf(fmaxf(v, 0.f), 1.f) * 255.0f);
  }
}

}  // namespace

struct maua_srvgg {
  maua_ctx* ctx;
  int num_feat, num_conv, upscale, dtype, act;  // act: 0 prelu, 1 relu, 2 leakyrelu(0.1)
  size_t esize;
  std::vector<PlainConv> convs;   // num_conv + 2
  std::vector<float*> slopes;     // num_conv + 1 PReLU vectors [num_feat]
  float* ones = nullptr;
  int ones_b = 0;
  size_t cap_px = 0;
  void *in32 = nullptr, *fa = nullptr, *fb = nullptr, *flast = nullptr;
};

extern "C" {

int maua_srvgg_create(maua_ctx* ctx, int num_feat, int num_conv, int upscale, int act_type, int dtype, maua_srvgg** out) {
  MAUA_REQUIRE(ctx && out, "maua_srvgg_create: NULL argument");
  MAUA_REQUIRE(dtype == MAUA_F32 || dtype == MAUA_BF16, "maua_srvgg_create: dtype must be MAUA_F32 or MAUA_BF16");
  MAUA_REQUIRE(num_feat > 0 && num_feat % 32 == 0 && num_conv >= 0 && upscale >= 1 && 3 * upscale * upscale <= 64,
               "maua_srvgg_create: num_feat % 32 == 0, upscale <= 4");
  MAUA_REQUIRE(act_type >= 0 && act_type <= 2, "maua_srvgg_create: act_type 0 prelu / 1 relu / 2 leakyrelu");
  maua_srvgg* n = new maua_srvgg();
  n->ctx = ctx; n->num_feat = num_feat; n->num_conv = num_conv; n->upscale = upscale; n->dtype = dtype; n->act = act_type;
  n->esize = dtype == MAUA_BF16 ? 2 : 4;
  n->convs.resize((size_t)num_conv + 2);
  int rc = alloc_conv(n->convs[0], 3, num_feat, n->esize);
  for (int i = 1; i <= num_conv && !rc; i++) rc = alloc_conv(n->convs[i], num_feat, num_feat, n->esize);
  if (!rc) rc = alloc_conv(n->convs[num_conv + 1], num_feat, 3 * upscale * upscale, n->esize);
  n->slopes.assign((size_t)num_conv + 1, nullptr);
  for (auto& p : n->slopes) {
    if (rc) break;
    if (hipMalloc((void**)&p, (size_t)num_feat * 4) != hipSuccess) { rc = fail("maua_srvgg_create: out of memory"); break; }
    std::vector<float> init((size_t)num_feat, 0.25f);   // nn.PReLU's default slope
    hipMemcpy(p, init.data(), init.size() * 4, hipMemcpyHostToDevice);
  }
  if (rc) {
    maua_srvgg_destroy(n);
    return rc;
  }
  *out = n;
  return MAUA_OK;
}

void maua_srvgg_destroy(maua_srvgg* n) {
  if (!n) return;
  hipStreamSynchronize(n->ctx->stream);
  for (auto& c : n->convs) {
    if (c.wt) hipFree(c.wt);
    if (c.bias) hipFree(c.bias);
  }
  for (float* p : n->slopes)
    if (p) hipFree(p);
  for (void* p : {n->in32, n->fa, n->fb, n->flast, (void*)n->ones})
    if (p) hipFree(p);
  delete n;
}

// name: a key of SRVGGNetCompact's state dict: body.<2k>.weight / .bias (convolutions), body.<2k+1>.weight (PReLU slopes)
int maua_srvgg_load(maua_srvgg* n, const char* name, const float* host, size_t count) {
  MAUA_REQUIRE(n && name && host, "maua_srvgg_load: NULL argument");
  int idx = -1;
  char par[16] = {0};
  if (sscanf(name, "body.%d.%15s", &idx, par) != 2 || idx < 0 || idx > 2 * (n->num_conv + 1))
    return fail(std::string("maua_srvgg_load: unknown parameter name: ") + name);
  hipStream_t st = n->ctx->stream;
  if (idx & 1) {  // activation module
    if (strcmp(par, "weight") || n->act != 0) return fail(std::string("maua_srvgg_load: unknown parameter name: ") + name);
    if (count != (size_t)n->num_feat) return fail(std::string("maua_srvgg_load: ") + name + ": wrong size");
    MAUA_HIP_CHECK(hipMemcpy(n->slopes[idx / 2], host, count * 4, hipMemcpyHostToDevice));
    return MAUA_OK;
  }
  PlainConv& c = n->convs[idx / 2];
  if (!strcmp(par, "bias")) {
    if (count != (size_t)c.Co) return fail(std::string("maua_srvgg_load: ") + name + ": wrong size");
    MAUA_HIP_CHECK(hipMemcpy(c.bias, host, count * 4, hipMemcpyHostToDevice));
    return MAUA_OK;
  }
  if (strcmp(par, "weight")) return fail(std::string("maua_srvgg_load: unknown parameter name: ") + name);
  if (count != (size_t)c.Co * c.Ci * 9) return fail(std::string("maua_srvgg_load: ") + name + ": wrong size");
  float* tmp;
  MAUA_HIP_CHECK(hipMalloc((void**)&tmp, count * 4));
  MAUA_HIP_CHECK(hipMemcpy(tmp, host, count * 4, hipMemcpyHostToDevice));
  MAUA_HIP_CHECK(hipMemsetAsync(c.wt, 0, (size_t)9 * c.Cop * c.Cip * n->esize, st));
  int rc = launch_prep_weights(st, n->dtype, tmp, c.wt, nullptr, c.Co, c.Ci, 3, 1, 0, c.Cop, c.Cip);
  hipStreamSynchronize(st);
  hipFree(tmp);
  return rc;
}

}  // extern "C"

namespace {

template <typename T>
int srvgg_forward_t(maua_srvgg* n, const float* img, int B, int H, int W, int do_clamp, float* out_f32, uint8_t* out_u8) {
  hipStream_t st = n->ctx->stream;
  const int F = n->num_feat;
  const size_t es = n->esize, px = (size_t)B * H * W;
  const int CL = n->convs.back().Cop;
  if (px > n->cap_px) {
    MAUA_HIP_CHECK(hipStreamSynchronize(st));
    for (void** p : {&n->in32, &n->fa, &n->fb, &n->flast}) { if (*p) hipFree(*p); *p = nullptr; }
    MAUA_HIP_CHECK(hipMalloc(&n->in32, px * 32 * es));
    MAUA_HIP_CHECK(hipMalloc(&n->fa, px * F * es));
    MAUA_HIP_CHECK(hipMalloc(&n->fb, px * F * es));
    MAUA_HIP_CHECK(hipMalloc(&n->flast, px * CL * es));
    n->cap_px = px;
  }
  if (B > n->ones_b) {
    MAUA_HIP_CHECK(hipStreamSynchronize(st));
    if (n->ones) hipFree(n->ones);
    std::vector<float> h((size_t)B * std::max(F, 32), 1.f);
    MAUA_HIP_CHECK(hipMalloc((void**)&n->ones, h.size() * 4));
    MAUA_HIP_CHECK(hipMemcpy(n->ones, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    n->ones_b = B;
  }
  auto conv = [&](const PlainConv& c, const void* x, void* y, const float* slopes, bool activated) -> int {
    ConvArgs a{};
    a.x = x; a.x_bstride = (long)H * W * c.Cip; a.w = c.wt; a.s = n->ones; a.bias = c.bias; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Ci = c.Cip; a.Co = c.Cop; a.up = 1; a.gain = 1.f; a.clamp = -1.f; a.alpha = 0.1f;
    a.act = !activated ? MAUA_ACT_LINEAR : (n->act == 1 ? MAUA_ACT_RELU : MAUA_ACT_LRELU);
    a.prelu = activated && n->act == 0 ? slopes : nullptr;
    if (dma_conv_narrow_supported(n->dtype, c.Cip, c.Cop, H, W)) return launch_modconv_dma(st, a);
    return launch_modconv3x3(st, n->dtype, a);
  };
  int rc = launch_nchw_to_nhwc<float, T>(st, img, n->in32, B, 3, H * W, 32);
  if (rc) return rc;
  if ((rc = conv(n->convs[0], n->in32, n->fa, n->slopes[0], true))) return rc;
  void *cur = n->fa, *nxt = n->fb;
  for (int i = 1; i <= n->num_conv; i++) {
    if ((rc = conv(n->convs[i], cur, nxt, n->slopes[i], true))) return rc;
    std::swap(cur, nxt);
  }
  if ((rc = conv(n->convs[n->num_conv + 1], cur, n->flast, nullptr, false))) return rc;
  const long total = (long)B * H * W * n->upscale * n->upscale;
  hipLaunchKernelGGL(srvgg_output_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const T*)n->flast, CL, img,
                     B, H, W, n->upscale, do_clamp, out_f32, out_u8);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace

// img: device f32 [B][3][H][W]; out_nchw: device f32 [B][3][sH][sW] (the network's raw output; clamped to [0,1] when clamp01) or
// NULL; out_rgb8: device u8 [B][sH][sW][3] = round(clamp(y, 0, 1) * 255) or NULL
extern "C" int maua_srvgg_forward(maua_srvgg* n, const float* img_nchw, int B, int H, int W, int clamp01, float* out_nchw,
                                  uint8_t* out_rgb8) {
  MAUA_REQUIRE(n && img_nchw, "maua_srvgg_forward: NULL argument");
  MAUA_REQUIRE(out_nchw || out_rgb8, "maua_srvgg_forward: no output buffer");
  MAUA_REQUIRE(B >= 0 && H > 0 && W > 0, "maua_srvgg_forward: bad shape");
  if (B == 0) return MAUA_OK;
  return n->dtype == MAUA_BF16 ? srvgg_forward_t<bf16_t>(n, img_nchw, B, H, W, clamp01, out_nchw, out_rgb8)
                               : srvgg_forward_t<float>(n, img_nchw, B, H, W, clamp01, out_nchw, out_rgb8);
}
