// The secondary diffusion model of the reference's DEFAULT guidance speed ("fast") - forward AND the vector-Jacobian product with
// respect to its input, evaluated by hand on the transposed network (the library has no autograd).
//
// Replaces (reference): maua/diffusion/processors/guided.py:68-143 (SecondaryDiffusionImageNet2: FourierFeatures timestep planes,
// 24 3x3 convolutions + ReLU, five AvgPool2d(2) / bilinear-x2 levels, skip concatenations; forward -> v, pred = x alpha - v sigma,
// eps = x sigma + v alpha) and the torch.autograd.grad(img, x, img_grad) of :236-272 that back-propagates through it.
//
//   forward :  h = [x | emb(t)] -> c0 -> c1 = s0 | down -> c2 -> c3 = s1 | down -> c4 -> c5 = s2 | down -> c6 -> c7 = s3 | down -> c8 -> c9 = s4
//              | down -> c10 .. c13 -> up -> [. | s4] -> c14 -> c15 -> up -> [. | s3] -> c16 -> c17 -> up -> [. | s2] -> c18 -> c19 -> up -> [. | s1]
//              -> c20 -> c21 -> up -> [. | s0] -> c22 -> c23 = v                                        (channels 64 128 128 256 256 512)
//   vjp     :  the same graph backwards: a 3x3 convolution's input gradient is the 3x3 convolution with the transposed, flipped
//              kernel (prepared at load time: the caller hands both layouts), ReLU's is the mask of the STORED forward activation,
//              AvgPool2d's is a quarter of the gradient at each of the four pixels, the bilinear up-sampling's is its adjoint
//              (gathered per input pixel from the <= 4 x 4 output pixels it touches, same clamped-source rule), a concatenation's
//              is the split; where a skip leaves (s_k feeds both the pool and the concatenation) the two gradients add.
//
// MI355X design: everything NHWC in the network dtype, concatenations are channel slices of one buffer (the convolutions write
// slices, the pool reads them: no copies), all 48 convolutions run on the MFMA implicit-GEMM kernel of modconv.hip (a plain
// convolution = unit styles, no demodulation; ReLU = leaky ReLU with slope 0 on its fast epilogue), the element-wise steps are
// 16-byte streaming kernels with the mask / add / pool-gradient fused.  Forward keeps every activation (13.9 M parameters, about 1 GB
// of activations per 16 samples at 256^2 in bf16); vjp() consumes the state of the LAST forward of the same shape.
#include <atomic>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "internal.h"

using namespace maua;

namespace {

constexpr int NCONV = 24;
constexpr int CS[6] = {64, 128, 128, 256, 256, 512};

struct SConv {
  int Ci, Co, Cip, Cop;     // real / padded-to-32 channels
  void *wt = nullptr, *wt_t = nullptr;   // prepared [9][Cop][Cip] and the transposed network's [9][Cip][Cop]
  float* bias = nullptr;    // [Cop], zero padded
  float* zero_bias = nullptr;
};

__device__ __forceinline__ void bil_src(int o, int n, int& i0, int& i1, float& f) {
  // F.interpolate(scale_factor=2, mode="bilinear", align_corners=False): src = (o + 0.5) / 2 - 0.5, clamped at 0
  float s = fmaxf((o + 0.5f) * 0.5f - 0.5f, 0.f);
  i0 = (int)s;
  i1 = min(i0 + 1, n - 1);
  f = s - (float)i0;
}

// in0[p][0..2] = x, [3..18] = cos / sin(2 pi t w_k) (FourierFeatures :57-66), [19..31] = 0
template <typename T>
__global__ __launch_bounds__(256) void sec_input_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                        const float* __restrict__ wemb, T* __restrict__ out, int B, long HW) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * HW) return;
  const long b = idx / HW, p = idx - b * HW;
  const float tb = t[b];
  T* o = out + idx * 32;
#pragma unroll
  for (int c = 0; c < 3; c++) Elem<T>::store(o + c, x[(b * 3 + c) * HW + p]);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const float f = 2.f * 3.14159265358979323846f * tb * wemb[k];
    Elem<T>::store(o + 3 + k, cosf(f));
    Elem<T>::store(o + 11 + k, sinf(f));
  }
#pragma unroll
  for (int c = 19; c < 32; c++) Elem<T>::store(o + c, 0.f);
}

// AvgPool2d(2): src = C channels at src (pixel stride sps), H x W -> dense [H/2][W/2][C]
template <typename T>
__global__ __launch_bounds__(256) void avgpool2_kernel(const T* __restrict__ src, int sps, T* __restrict__ dst, int B, int H, int W,
                                                       int C) {
  constexpr int E = 16 / (int)sizeof(T);
  const int ppp = C / E, h2 = H / 2, w2 = W / 2;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * h2 * w2 * ppp) return;
  const int pc = (int)(idx % ppp);
  long p = idx / ppp;
  const int x = (int)(p % w2); p /= w2;
  const int y = (int)(p % h2);
  const int b = (int)(p / h2);
  float v[E];
#pragma unroll
  for (int e = 0; e < E; e++) v[e] = 0.f;
#pragma unroll
  for (int dy = 0; dy < 2; dy++)
#pragma unroll
    for (int dx = 0; dx < 2; dx++) {
      const T* s = src + (((long)b * H + 2 * y + dy) * W + 2 * x + dx) * sps + pc * E;
#pragma unroll
      for (int e = 0; e < E; e++) v[e] += Elem<T>::load(s + e);
    }
  T* d = dst + (((long)b * h2 + y) * w2 + x) * C + pc * E;
#pragma unroll
  for (int e = 0; e < E; e++) Elem<T>::store(d + e, v[e] * 0.25f);
}

// bilinear x2: dense [h][w][C] -> C channels at dst (pixel stride dps) of the [2h][2w] grid
template <typename T>
__global__ __launch_bounds__(256) void bilinear_up2_kernel(const T* __restrict__ src, T* __restrict__ dst, int dps, int B, int h, int w,
                                                           int C) {
  constexpr int E = 16 / (int)sizeof(T);
  const int ppp = C / E;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * 4 * h * w * ppp) return;
  const int pc = (int)(idx % ppp);
  long p = idx / ppp;
  const int ox = (int)(p % (2 * w)); p /= 2 * w;
  const int oy = (int)(p % (2 * h));
  const int b = (int)(p / (2 * h));
  int y0, y1, x0, x1;
  float fy, fx;
  bil_src(oy, h, y0, y1, fy);
  bil_src(ox, w, x0, x1, fx);
  const T* s = src + (long)b * h * w * C + pc * E;
  float v[E];
#pragma unroll
  for (int e = 0; e < E; e++) {
    const float a = Elem<T>::load(s + ((long)y0 * w + x0) * C + e), bq = Elem<T>::load(s + ((long)y0 * w + x1) * C + e);
    const float c = Elem<T>::load(s + ((long)y1 * w + x0) * C + e), d = Elem<T>::load(s + ((long)y1 * w + x1) * C + e);
    v[e] = (1.f - fy) * ((1.f - fx) * a + fx * bq) + fy * ((1.f - fx) * c + fx * d);
  }
  T* o = dst + (((long)b * 2 * h + oy) * 2 * w + ox) * dps + pc * E;
#pragma unroll
  for (int e = 0; e < E; e++) Elem<T>::store(o + e, v[e]);
}

// the adjoint of bilinear_up2: g_in[y][x] = sum over the output pixels that read (y, x) of their weight x g_out; then the ReLU
// mask of the stored activation `act` (dense [h][w][C]; NULL: none).  gout: C channels at pixel stride gps of the [2h][2w] grid.
template <typename T>
__global__ __launch_bounds__(256) void bilinear_up2_adjoint_kernel(const T* __restrict__ gout, int gps, const T* __restrict__ act,
                                                                   T* __restrict__ gin, int B, int h, int w, int C) {
  constexpr int E = 16 / (int)sizeof(T);
  const int ppp = C / E;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * h * w * ppp) return;
  const int pc = (int)(idx % ppp);
  long p = idx / ppp;
  const int x = (int)(p % w); p /= w;
  const int y = (int)(p % h);
  const int b = (int)(p / h);
  float v[E];
#pragma unroll
  for (int e = 0; e < E; e++) v[e] = 0.f;
  for (int oy = max(2 * y - 1, 0); oy <= min(2 * y + 2, 2 * h - 1); oy++) {
    int y0, y1;
    float fy;
    bil_src(oy, h, y0, y1, fy);
    const float wy = (y0 == y ? 1.f - fy : 0.f) + (y1 == y ? fy : 0.f);
    if (wy == 0.f) continue;
    for (int ox = max(2 * x - 1, 0); ox <= min(2 * x + 2, 2 * w - 1); ox++) {
      int x0, x1;
      float fx;
      bil_src(ox, w, x0, x1, fx);
      const float wx = (x0 == x ? 1.f - fx : 0.f) + (x1 == x ? fx : 0.f);
      if (wx == 0.f) continue;
      const T* g = gout + (((long)b * 2 * h + oy) * 2 * w + ox) * gps + pc * E;
#pragma unroll
      for (int e = 0; e < E; e++) v[e] += wy * wx * Elem<T>::load(g + e);
    }
  }
  const long o = (((long)b * h + y) * w + x) * C + pc * E;
#pragma unroll
  for (int e = 0; e < E; e++) {
    const bool on = !act || Elem<T>::load(act + o + e) > 0.f;
    Elem<T>::store(gin + o + e, on ? v[e] : 0.f);
  }
}

// g[p][c] *= (act[p][c] > 0): dense g, act with pixel stride aps
template <typename T>
__global__ __launch_bounds__(256) void relu_mask_kernel(T* __restrict__ g, const T* __restrict__ act, int aps, long n_pix, int C) {
  constexpr int E = 16 / (int)sizeof(T);
  const int ppp = C / E;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_pix * ppp) return;
  const long p = idx / ppp;
  const int c = (int)(idx - p * ppp) * E;
#pragma unroll
  for (int e = 0; e < E; e++)
    if (!(Elem<T>::load(act + p * aps + c + e) > 0.f)) Elem<T>::store(g + p * C + c + e, 0.f);
}

// where a skip leaves: g_skip[p][c] = (act > 0) ? gcat[p][c] + 0.25 * gpool[p / 2][c] : 0
// (gcat / act: C channels at pixel stride cps of the H x W grid; gpool dense [H/2][W/2][C]; out dense [H][W][C])
template <typename T>
__global__ __launch_bounds__(256) void skip_grad_kernel(const T* __restrict__ gcat, const T* __restrict__ act, int cps,
                                                        const T* __restrict__ gpool, T* __restrict__ out, int B, int H, int W, int C) {
  constexpr int E = 16 / (int)sizeof(T);
  const int ppp = C / E;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * H * W * ppp) return;
  const int pc = (int)(idx % ppp);
  long p = idx / ppp;
  const int x = (int)(p % W); p /= W;
  const int y = (int)(p % H);
  const int b = (int)(p / H);
  const long pi = ((long)b * H + y) * W + x;
  const long pp = (((long)b * (H / 2) + (y >> 1)) * (W / 2) + (x >> 1)) * C + pc * E;
#pragma unroll
  for (int e = 0; e < E; e++) {
    const float a = Elem<T>::load(act + pi * cps + pc * E + e);
    const float v = Elem<T>::load(gcat + pi * cps + pc * E + e) + 0.25f * Elem<T>::load(gpool + pp + e);
    Elem<T>::store(out + pi * C + pc * E + e, a > 0.f ? v : 0.f);
  }
}

// v (first 3 of 32 channels, NHWC) -> v / pred / eps as planar f32 (any NULL skipped): pred = x alpha - v sigma, eps = x sigma + v alpha
template <typename T>
__global__ __launch_bounds__(256) void sec_output_kernel(const T* __restrict__ vn, const float* __restrict__ x, const float* __restrict__ t,
                                                         float* __restrict__ v_out, float* __restrict__ pred, float* __restrict__ eps,
                                                         int B, long HW) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * HW) return;
  const long b = idx / HW, p = idx - b * HW;
  const float ang = t[b] * 1.57079632679489661923f;
  const float al = cosf(ang), si = sinf(ang);
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float v = Elem<T>::load(vn + idx * 32 + c);
    const long o = (b * 3 + c) * HW + p;
    const float xv = x[o];
    if (v_out) v_out[o] = v;
    if (pred) pred[o] = xv * al - v * si;
    if (eps) eps[o] = xv * si + v * al;
  }
}

// planar f32 [B][3][HW] -> NHWC T [.][32] (channels 3.. zero)
template <typename T>
__global__ __launch_bounds__(256) void sec_grad_in_kernel(const float* __restrict__ g, T* __restrict__ out, int B, long HW) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * HW) return;
  const long b = idx / HW, p = idx - b * HW;
#pragma unroll
  for (int c = 0; c < 32; c++) Elem<T>::store(out + idx * 32 + c, c < 3 ? g[(b * 3 + c) * HW + p] : 0.f);
}

// NHWC T [.][32] -> planar f32 [B][3][HW]
template <typename T>
__global__ __launch_bounds__(256) void sec_grad_out_kernel(const T* __restrict__ g, float* __restrict__ out, int B, long HW) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * HW) return;
  const long b = idx / HW, p = idx - b * HW;
#pragma unroll
  for (int c = 0; c < 3; c++) out[(b * 3 + c) * HW + p] = Elem<T>::load(g + idx * 32 + c);
}

}  // namespace

struct maua_secondary {
  maua_ctx* ctx;
  // identity of this model and of its device buffers: a captured graph that holds pointers into them (unet.hip's guided loop) is valid
  // for one (uid, epoch) only.  uid is never reused (an allocator may hand a later model the freed one's address); epoch moves whenever
  // a workspace or table is freed or reallocated.
  unsigned long long uid = 0, epoch = 0;
  int dtype;
  int conv_dtype;
  size_t esize;
  SConv conv[NCONV];
  float* wemb = nullptr;   // [8] FourierFeatures weight
  float* ones = nullptr;   // unit styles [bcap][512]
  int ones_b = 0;
  // activations of the last forward (grow-only): see forward_t for the layout
  size_t cap_px = 0;       // B * H * W
  int B = 0, H = 0, W = 0; // shape of the last forward (0: none)
  void* in0 = nullptr;
  void* a[5] = {};         // first conv of each level: a[0] = c0 out, a[k] = c(2k) out
  void* cat[5] = {};       // [up(main) | skip] per level
  void* pool[5] = {};      // pool[k] = down(skip k-1) (k = 1..5; index k-1)
  void* bott[4] = {};      // c10 .. c13 outputs
  void* dec[9] = {};       // c14 .. c22 outputs
  void* vbuf = nullptr;    // c23 output (32 channels, 3 real)
  // gradient workspaces (vjp)
  void *ga = nullptr, *gb = nullptr, *gcat = nullptr, *gpool = nullptr;
};

namespace {

int level_ch_main(int k) { return k == 0 ? CS[0] : k == 1 ? CS[1] : k == 2 ? CS[2] : k == 3 ? CS[3] : CS[4]; }   // channels of up(main) at level k
int level_ch_skip(int k) { return CS[k]; }                                                                      // ... of the skip s_k

void free_ws(maua_secondary* n) {
  auto f = [](void*& p) { if (p) hipFree(p); p = nullptr; };
  f(n->in0); f(n->vbuf); f(n->ga); f(n->gb); f(n->gcat); f(n->gpool);
  for (auto& p : n->a) f(p);
  for (auto& p : n->cat) f(p);
  for (auto& p : n->pool) f(p);
  for (auto& p : n->bott) f(p);
  for (auto& p : n->dec) f(p);
  n->cap_px = 0;
  n->B = n->H = n->W = 0;
  n->epoch++;
}

// the 24 convolutions in execution order: channels in / out (guided.py:77-134)
void conv_plan(int (&ci)[NCONV], int (&co)[NCONV]) {
  const int c0 = CS[0], c1 = CS[1], c2 = CS[2], c3 = CS[3], c4 = CS[4], c5 = CS[5];
  const int I[NCONV] = {19, c0, c0, c1, c1, c2, c2, c3, c3, c4, c4, c5, c5, c5, 2 * c4, c4, 2 * c3, c3, 2 * c2, c2, 2 * c1, c1, 2 * c0, c0};
  const int O[NCONV] = {c0, c0, c1, c1, c2, c2, c3, c3, c4, c4, c5, c5, c5, c4, c4, c3, c3, c2, c2, c1, c1, c0, c0, 3};
  for (int i = 0; i < NCONV; i++) { ci[i] = I[i]; co[i] = O[i]; }
}

template <typename T>
int ensure_ws(maua_secondary* n, int B, int H, int W) {
  hipStream_t st = n->ctx->stream;
  const size_t px = (size_t)B * H * W, es = n->esize;
  if (px > n->cap_px) {
    MAUA_HIP_CHECK(hipStreamSynchronize(st));
    free_ws(n);
    auto al = [&](void*& p, size_t elems) -> int {
      MAUA_HIP_CHECK(hipMalloc(&p, elems * es));
      return MAUA_OK;
    };
    int rc = al(n->in0, px * 32);
    if (!rc) rc = al(n->vbuf, px * 32);
    for (int k = 0; k < 5 && !rc; k++) {
      const size_t lp = px >> (2 * k);
      rc = al(n->a[k], lp * CS[k]);
      if (!rc) rc = al(n->cat[k], lp * (level_ch_main(k) + level_ch_skip(k)));
      if (!rc) rc = al(n->pool[k], (lp >> 2) * CS[k]);
    }
    const size_t l5 = px >> 10;
    const int bc[4] = {CS[5], CS[5], CS[5], CS[4]};
    for (int i = 0; i < 4 && !rc; i++) rc = al(n->bott[i], l5 * bc[i]);
    // decoder outputs: c14, c15 at level 4; c16, c17 at 3; c18, c19 at 2; c20, c21 at 1; c22 at 0
    const int dc[9] = {CS[4], CS[3], CS[3], CS[2], CS[2], CS[1], CS[1], CS[0], CS[0]};
    const int dl[9] = {4, 4, 3, 3, 2, 2, 1, 1, 0};
    for (int i = 0; i < 9 && !rc; i++) rc = al(n->dec[i], (px >> (2 * dl[i])) * dc[i]);
    // gradient workspaces: the widest tensors of the backward pass are the level-0 ones (128 channels)
    if (!rc) rc = al(n->ga, px * 128);
    if (!rc) rc = al(n->gb, px * 128);
    if (!rc) rc = al(n->gcat, px * 224);   // the concatenations' gradients of all five levels, back to back: 128 + 64 + 16 + 8 + 2 per level-0 pixel
    if (!rc) rc = al(n->gpool, px * 128 / 4);
    if (rc) return rc;
    n->cap_px = px;
  }
  if (B > n->ones_b) {
    MAUA_HIP_CHECK(hipStreamSynchronize(st));
    if (n->ones) hipFree(n->ones);
    n->epoch++;
    std::vector<float> h((size_t)B * 512, 1.f);
    MAUA_HIP_CHECK(hipMalloc((void**)&n->ones, h.size() * 4));
    MAUA_HIP_CHECK(hipMemcpy(n->ones, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    n->ones_b = B;
  }
  return MAUA_OK;
}

// y slice = act(conv3x3(x) + bias): x = all Cip channels of a buffer with xps channels per pixel, y = Cop channels at offset ycoff of
// a buffer with yps per pixel.  transposed: the input-gradient convolution of the same layer (Cop -> Cip, no bias, no activation)
int run_conv(maua_secondary* n, const SConv& c, bool transposed, const void* x, int xps, void* y, int yps, int ycoff, int B, int h, int w,
             bool relu) {
  ConvArgs a{};
  const int Ci = transposed ? c.Cop : c.Cip, Co = transposed ? c.Cip : c.Cop;
  a.x = x; a.x_bstride = (long)h * w * xps; a.x_pstride = xps; a.w = transposed ? c.wt_t : c.wt; a.s = n->ones; a.d = nullptr;
  a.noise = nullptr; a.bias = transposed ? c.zero_bias : c.bias; a.y = y; a.y_pstride = yps; a.y_coff = ycoff; a.y_bstride = (long)h * w * yps;
  a.B = B; a.H = h; a.W = w; a.Ci = Ci; a.Co = Co; a.up = 1;
  a.act = relu ? MAUA_ACT_LRELU : MAUA_ACT_LINEAR; a.alpha = relu ? 0.f : 1.f; a.gain = 1.f; a.clamp = -1.f;
  return launch_modconv3x3(n->ctx->stream, n->conv_dtype, a);
}

#define SEC_LAUNCH(KERNEL, TOTAL, ...)                                                                         \
  do {                                                                                                         \
    const long total_ = (TOTAL);                                                                               \
    if (total_ > 0) hipLaunchKernelGGL(KERNEL, dim3((unsigned)((total_ + 255) / 256)), dim3(256), 0, st, __VA_ARGS__); \
    MAUA_HIP_CHECK(hipGetLastError());                                                                         \
  } while (0)

template <typename T>
int forward_t(maua_secondary* n, const float* x, const float* t, int B, int H, int W, float* v_out, float* pred, float* eps) {
  hipStream_t st = n->ctx->stream;
  if (int rc = ensure_ws<T>(n, B, H, W)) return rc;
  constexpr int E = 16 / (int)sizeof(T);
  const long HW = (long)H * W;
  int rc;
  SEC_LAUNCH(sec_input_kernel<T>, (long)B * HW, x, t, n->wemb, (T*)n->in0, B, HW);
  // encoder: level k's skip s_k lands in the upper channel slice of cat[k]
  const void* cur = n->in0;
  int cur_ps = 32;
  for (int k = 0; k < 5; k++) {
    const int h = H >> k, w = W >> k, cm = level_ch_main(k), csk = level_ch_skip(k), cps = cm + csk;
    if ((rc = run_conv(n, n->conv[2 * k], false, cur, cur_ps, n->a[k], CS[k], 0, B, h, w, true))) return rc;
    if ((rc = run_conv(n, n->conv[2 * k + 1], false, n->a[k], CS[k], n->cat[k], cps, cm, B, h, w, true))) return rc;
    SEC_LAUNCH(avgpool2_kernel<T>, (long)B * (h / 2) * (w / 2) * (csk / E), (const T*)n->cat[k] + cm, cps, (T*)n->pool[k], B, h, w, csk);
    cur = n->pool[k];
    cur_ps = csk;
  }
  // bottleneck at level 5: c10 .. c13
  {
    const int h = H >> 5, w = W >> 5;
    const int bc[4] = {CS[5], CS[5], CS[5], CS[4]};
    for (int i = 0; i < 4; i++) {
      if ((rc = run_conv(n, n->conv[10 + i], false, cur, cur_ps, n->bott[i], bc[i], 0, B, h, w, true))) return rc;
      cur = n->bott[i];
      cur_ps = bc[i];
    }
  }
  // decoder: up into the lower slice of cat[k], two convolutions, k = 4 .. 1; then level 0
  for (int k = 4; k >= 0; k--) {
    const int h = H >> k, w = W >> k, cm = level_ch_main(k), cps = cm + level_ch_skip(k);
    SEC_LAUNCH(bilinear_up2_kernel<T>, (long)B * h * w * (cm / E), (const T*)cur, (T*)n->cat[k], cps, B, h / 2, w / 2, cm);
    if (k > 0) {
      const int i0 = 14 + 2 * (4 - k);
      const int c_mid = n->conv[i0].Cop, c_out = n->conv[i0 + 1].Cop;
      if ((rc = run_conv(n, n->conv[i0], false, n->cat[k], cps, n->dec[2 * (4 - k)], c_mid, 0, B, h, w, true))) return rc;
      if ((rc = run_conv(n, n->conv[i0 + 1], false, n->dec[2 * (4 - k)], c_mid, n->dec[2 * (4 - k) + 1], c_out, 0, B, h, w, true))) return rc;
      cur = n->dec[2 * (4 - k) + 1];
    } else {
      if ((rc = run_conv(n, n->conv[22], false, n->cat[0], cps, n->dec[8], CS[0], 0, B, h, w, true))) return rc;
      if ((rc = run_conv(n, n->conv[23], false, n->dec[8], CS[0], n->vbuf, 32, 0, B, h, w, false))) return rc;
    }
  }
  SEC_LAUNCH(sec_output_kernel<T>, (long)B * HW, (const T*)n->vbuf, x, t, v_out, pred, eps, B, HW);
  n->B = B; n->H = H; n->W = W;
  return MAUA_OK;
}

template <typename T>
int vjp_t(maua_secondary* n, const float* g_v, float* g_x) {
  hipStream_t st = n->ctx->stream;
  constexpr int E = 16 / (int)sizeof(T);
  const int B = n->B, H = n->H, W = n->W;
  const long HW = (long)H * W;
  int rc;
  T *ga = (T*)n->ga, *gb = (T*)n->gb, *gcat = (T*)n->gcat, *gpool = (T*)n->gpool;
  // v = c23(d22): no ReLU on v
  SEC_LAUNCH(sec_grad_in_kernel<T>, (long)B * HW, g_v, ga, B, HW);
  if ((rc = run_conv(n, n->conv[23], true, ga, 32, gb, CS[0], 0, B, H, W, false))) return rc;       // G(d22) before its mask
  SEC_LAUNCH(relu_mask_kernel<T>, (long)B * HW * (CS[0] / E), gb, (const T*)n->dec[8], CS[0], (long)B * HW, CS[0]);
  // decoder backwards, level 0 .. 4: G(cat[k]) = convT(first decoder conv of the level); its lower slice goes down through the
  // up-sampling's adjoint, its upper slice waits in gcat_k for the encoder's way back.  The cat gradients of all levels are kept
  // (they are needed again on the way up): level k's lives in its own buffer - reuse the forward's bott / pool scratch? no: allocate
  // from the gradient arena below.
  // (sizes: G(cat[k]) has (cm + cs) channels at level k: 128 px-channels at level 0, 64 at level 1, 16, 8, 2: all fit gcat back to back)
  T* gcat_k[5];
  {
    size_t off = 0;
    for (int k = 0; k < 5; k++) {
      gcat_k[k] = gcat + off;
      off += ((size_t)B * HW >> (2 * k)) * (level_ch_main(k) + level_ch_skip(k));
    }
  }
  T* gcur = gb;      // G(output of the second decoder conv of the level above), masked; at level 0: G(d22)
  T* gother = ga;
  for (int k = 0; k <= 4; k++) {
    const int h = H >> k, w = W >> k, cm = level_ch_main(k), cps = cm + level_ch_skip(k);
    const long lp = (long)B * h * w;
    const int i_first = k == 0 ? 22 : 14 + 2 * (4 - k);
    if (k > 0) {
      // gcur = G(second conv's output) -> through the second conv -> mask of the first conv's output
      const int c_mid = n->conv[i_first].Cop;
      if ((rc = run_conv(n, n->conv[i_first + 1], true, gcur, n->conv[i_first + 1].Cop, gother, c_mid, 0, B, h, w, false))) return rc;
      SEC_LAUNCH(relu_mask_kernel<T>, lp * (c_mid / E), gother, (const T*)n->dec[2 * (4 - k)], c_mid, lp, c_mid);
      std::swap(gcur, gother);
    }
    if ((rc = run_conv(n, n->conv[i_first], true, gcur, n->conv[i_first].Cop, gcat_k[k], cps, 0, B, h, w, false))) return rc;
    // lower slice -> adjoint of the up-sampling -> G(source), masked by the source's stored activation
    const void* src_act = k == 4 ? n->bott[3] : n->dec[2 * (4 - (k + 1)) + 1];
    SEC_LAUNCH(bilinear_up2_adjoint_kernel<T>, (long)B * (h / 2) * (w / 2) * (cm / E), (const T*)gcat_k[k], cps, (const T*)src_act, gother, B,
               h / 2, w / 2, cm);
    std::swap(gcur, gother);
  }
  // bottleneck backwards: gcur = G(c13 out), masked
  {
    const int h = H >> 5, w = W >> 5;
    const long lp = (long)B * h * w;
    for (int i = 3; i >= 1; i--) {
      const int c_prev = n->conv[10 + i].Cip;
      if ((rc = run_conv(n, n->conv[10 + i], true, gcur, n->conv[10 + i].Cop, gother, c_prev, 0, B, h, w, false))) return rc;
      SEC_LAUNCH(relu_mask_kernel<T>, lp * (c_prev / E), gother, (const T*)n->bott[i - 1], c_prev, lp, c_prev);
      std::swap(gcur, gother);
    }
    if ((rc = run_conv(n, n->conv[10], true, gcur, n->conv[10].Cop, gpool, CS[4], 0, B, h, w, false))) return rc;   // G(pool[4])
  }
  // encoder backwards, level 4 .. 0: G(s_k) = G(cat[k])[skip slice] + pool gradient, masked; through c(2k+1), mask, through c(2k)
  for (int k = 4; k >= 0; k--) {
    const int h = H >> k, w = W >> k, cm = level_ch_main(k), csk = level_ch_skip(k), cps = cm + csk;
    const long lp = (long)B * h * w;
    SEC_LAUNCH(skip_grad_kernel<T>, lp * (csk / E), (const T*)gcat_k[k] + cm, (const T*)n->cat[k] + cm, cps, (const T*)gpool, gcur, B, h, w, csk);
    if ((rc = run_conv(n, n->conv[2 * k + 1], true, gcur, csk, gother, CS[k], 0, B, h, w, false))) return rc;
    SEC_LAUNCH(relu_mask_kernel<T>, lp * (CS[k] / E), gother, (const T*)n->a[k], CS[k], lp, CS[k]);
    // through the level's first conv: into the pool gradient of the level above (k > 0) or the network input (k = 0)
    const int cin = n->conv[2 * k].Cip;
    if ((rc = run_conv(n, n->conv[2 * k], true, gother, CS[k], k > 0 ? (void*)gpool : (void*)gcur, cin, 0, B, h, w, false))) return rc;
  }
  SEC_LAUNCH(sec_grad_out_kernel<T>, (long)B * HW, (const T*)gcur, g_x, B, HW);
  return MAUA_OK;
}

}  // namespace

namespace maua {
maua_ctx* secondary_ctx(maua_secondary* n) { return n ? n->ctx : nullptr; }
void secondary_stamp(maua_secondary* n, unsigned long long* uid, unsigned long long* epoch) {
  *uid = n ? n->uid : 0;
  *epoch = n ? n->epoch : 0;
}
}

extern "C" {

int maua_secondary_create(maua_ctx* ctx, int dtype, maua_secondary** out) {
  MAUA_REQUIRE(ctx && out, "maua_secondary_create: NULL argument");
  MAUA_REQUIRE(dtype == MAUA_F32 || dtype == MAUA_BF16 || dtype == MAUA_F32_SPLIT,
               "maua_secondary_create: dtype must be MAUA_F32, MAUA_F32_SPLIT or MAUA_BF16");
  maua_secondary* n = new maua_secondary();
  static std::atomic<unsigned long long> next_uid{1};
  n->uid = next_uid.fetch_add(1);
  // MAUA_F32_SPLIT: float32 tensors everywhere (dtype below), only the convolutions' products differ (conv_dtype)
  n->ctx = ctx; n->conv_dtype = dtype; n->dtype = dtype == MAUA_F32_SPLIT ? MAUA_F32 : dtype; n->esize = n->dtype == MAUA_BF16 ? 2 : 4;
  int ci[NCONV], co[NCONV];
  conv_plan(ci, co);
  for (int i = 0; i < NCONV; i++) {
    SConv& c = n->conv[i];
    c.Ci = ci[i]; c.Co = co[i]; c.Cip = (ci[i] + 31) / 32 * 32; c.Cop = (co[i] + 31) / 32 * 32;
    const size_t we = (size_t)9 * c.Cop * c.Cip * n->esize;
    if (hipMalloc(&c.wt, we) != hipSuccess || hipMalloc(&c.wt_t, we) != hipSuccess ||
        hipMalloc((void**)&c.bias, (size_t)c.Cop * 4) != hipSuccess ||
        hipMalloc((void**)&c.zero_bias, (size_t)std::max(c.Cop, c.Cip) * 4) != hipSuccess) {
      maua_secondary_destroy(n);
      return fail("maua_secondary_create: out of device memory");
    }
    hipMemset(c.wt, 0, we); hipMemset(c.wt_t, 0, we);
    hipMemset(c.bias, 0, (size_t)c.Cop * 4); hipMemset(c.zero_bias, 0, (size_t)std::max(c.Cop, c.Cip) * 4);
  }
  if (hipMalloc((void**)&n->wemb, 32) != hipSuccess) {
    maua_secondary_destroy(n);
    return fail("maua_secondary_create: out of device memory");
  }
  hipMemset(n->wemb, 0, 32);
  *out = n;
  return MAUA_OK;
}

void maua_secondary_destroy(maua_secondary* n) {
  if (!n) return;
  hipStreamSynchronize(n->ctx->stream);
  free_ws(n);
  for (auto& c : n->conv) {
    if (c.wt) hipFree(c.wt);
    if (c.wt_t) hipFree(c.wt_t);
    if (c.bias) hipFree(c.bias);
    if (c.zero_bias) hipFree(c.zero_bias);
  }
  if (n->wemb) hipFree(n->wemb);
  if (n->ones) hipFree(n->ones);
  delete n;
}

int maua_secondary_conv_shape(int index, int* ci, int* co) {
  MAUA_REQUIRE(index >= 0 && index < NCONV && ci && co, "maua_secondary_conv_shape: bad argument");
  int I[NCONV], O[NCONV];
  conv_plan(I, O);
  *ci = I[index];
  *co = O[index];
  return MAUA_OK;
}

// what: 0 = convolution weight [Co][Ci][3][3] (the transposed network's layout is derived here), 1 = bias [Co],
// 2 = timestep_embed.weight [8] (index ignored)
int maua_secondary_load(maua_secondary* n, int index, int what, const float* host, size_t count) {
  MAUA_REQUIRE(n && host, "maua_secondary_load: NULL argument");
  hipStream_t st = n->ctx->stream;
  if (what == 2) {
    MAUA_REQUIRE(count == 8, "maua_secondary_load: timestep_embed.weight has 8 values");
    MAUA_HIP_CHECK(hipMemcpy(n->wemb, host, 32, hipMemcpyHostToDevice));
    return MAUA_OK;
  }
  MAUA_REQUIRE(index >= 0 && index < NCONV, "maua_secondary_load: no such convolution");
  SConv& c = n->conv[index];
  if (what == 1) {
    MAUA_REQUIRE(count == (size_t)c.Co, "maua_secondary_load: bias: wrong size");
    MAUA_HIP_CHECK(hipMemcpy(c.bias, host, count * 4, hipMemcpyHostToDevice));
    return MAUA_OK;
  }
  MAUA_REQUIRE(what == 0, "maua_secondary_load: what must be 0, 1 or 2");
  MAUA_REQUIRE(count == (size_t)c.Co * c.Ci * 9, "maua_secondary_load: weight: wrong size");
  // the input-gradient convolution: Wt[ci][co][ky][kx] = W[co][ci][2 - ky][2 - kx]
  std::vector<float> wt(count);
  for (int o = 0; o < c.Co; o++)
    for (int i = 0; i < c.Ci; i++)
      for (int k = 0; k < 9; k++) wt[((size_t)i * c.Co + o) * 9 + k] = host[((size_t)o * c.Ci + i) * 9 + (8 - k)];
  float* tmp;
  MAUA_HIP_CHECK(hipMalloc((void**)&tmp, count * 4));
  const size_t we = (size_t)9 * c.Cop * c.Cip * n->esize;
  MAUA_HIP_CHECK(hipMemcpy(tmp, host, count * 4, hipMemcpyHostToDevice));
  MAUA_HIP_CHECK(hipMemsetAsync(c.wt, 0, we, st));
  int rc = launch_prep_weights(st, n->dtype, tmp, c.wt, nullptr, c.Co, c.Ci, 3, 1, 0, c.Cop, c.Cip);
  hipStreamSynchronize(st);
  if (!rc) {
    MAUA_HIP_CHECK(hipMemcpy(tmp, wt.data(), count * 4, hipMemcpyHostToDevice));
    MAUA_HIP_CHECK(hipMemsetAsync(c.wt_t, 0, we, st));
    rc = launch_prep_weights(st, n->dtype, tmp, c.wt_t, nullptr, c.Ci, c.Co, 3, 1, 0, c.Cip, c.Cop);
    if (!rc && n->conv_dtype == MAUA_F32_SPLIT) {   // both copies in the split form the matrix cores read
      rc = launch_f32_split_inplace(st, c.wt, (long)9 * c.Cop * c.Cip);
      if (!rc) rc = launch_f32_split_inplace(st, c.wt_t, (long)9 * c.Cop * c.Cip);
    }
    hipStreamSynchronize(st);
  }
  hipFree(tmp);
  return rc;
}

int maua_secondary_forward(maua_secondary* n, const float* x, const float* t, int B, int H, int W, float* v_out, float* pred_out,
                           float* eps_out) {
  MAUA_REQUIRE(n && x && t, "maua_secondary_forward: NULL argument");
  MAUA_REQUIRE(B >= 0 && H > 0 && W > 0 && H % 32 == 0 && W % 32 == 0, "maua_secondary_forward: H and W must be positive multiples of 32");
  if (B == 0) return MAUA_OK;
  return n->dtype == MAUA_BF16 ? forward_t<bf16_t>(n, x, t, B, H, W, v_out, pred_out, eps_out)
                               : forward_t<float>(n, x, t, B, H, W, v_out, pred_out, eps_out);
}

int maua_secondary_vjp(maua_secondary* n, const float* g_v, int B, int H, int W, float* g_x) {
  MAUA_REQUIRE(n && g_v && g_x, "maua_secondary_vjp: NULL argument");
  MAUA_REQUIRE(n->B > 0 && B == n->B && H == n->H && W == n->W, "maua_secondary_vjp: call maua_secondary_forward with the same shape first");
  return n->dtype == MAUA_BF16 ? vjp_t<bf16_t>(n, g_v, g_x) : vjp_t<float>(n, g_v, g_x);
}

}  // extern "C"
