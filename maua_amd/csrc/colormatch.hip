// ColorMatchGrads on the device: the saturation-weighted hue histogram of an image batch, its MSE to a style image's histogram and
// the gradient of that loss with respect to the image - evaluated in closed form (the library has no autograd).
//
// Replaces (reference): maua/grad.py:27-47 differentiable_histogram (255 masked passes over the image, one per bin),
// maua/grad.py:50-70 ColorMatchGrads.histogram / forward (kornia.color.rgb_to_hsv of clamp((img + 1) / 2, 1e-8, 1 - 1e-8), clamp(0, 1)
// - kornia's hue is in RADIANS, so every hue above one radian sits on the last edge -, weighting sqrt(sat * val), histogram,
// mse_loss, torch.autograd.grad).
//
// One pass instead of 255: a value x in [e_k, e_k+1) (e_j = float32(j) * float32(1 / (nbins - 1)), the reference's edges) adds
// (e_k+1 - x) w to bin k and (x - e_k) w to bin k + 1 (where those bins exist).  The sums are taken in 64-bit FIXED POINT (2^-40):
// integer atomics commute, so the histogram - and the gradient - are bit-identical from run to run, which float atomics are not.
//   hist  : per pixel hsv + the two contributions -> LDS histogram per workgroup -> global u64 [B][nbins]
//   final : per sample H = R / sum R, the loss's share, gR = dL/dR (through the normalisation)
//   grad  : per pixel again hsv, then dL/dx = w (gR[k + 1] - gR[k]), dL/dw = (e_k+1 - x) gR[k] + (x - e_k) gR[k + 1], back through
//           sqrt(s v), the clamps, rgb_to_hsv (max / min send their gradient to the FIRST maximal / minimal channel, as torch's
//           max(dim) backward does) and (img + 1) / 2.  A pixel whose weight is exactly zero gets no gradient through the square
//           root (the reference's autograd yields inf / NaN there).
#include <cmath>

#include "common.h"
#include "internal.h"

using namespace maua;

namespace {

constexpr float TWO_PI = 6.283185307179586f;
constexpr double FIX = 1099511627776.0;   // 2^40
constexpr int MAX_BINS = 1024;

struct Hsv {
  float c[3];       // clamped rgb in [1e-8, 1]
  bool live[3];     // the clamp passes the gradient (1e-8 <= u <= 1)
  int im, in;       // first maximal / minimal channel
  float M, delta, deltap, num, hue_raw, x, s, v, w;
};

// one pixel of ColorMatchGrads.histogram up to the histogram's inputs (x = clamped hue, w = weight)
__device__ __forceinline__ Hsv pixel_hsv(float r, float g, float b, bool sat_weighting) {
  Hsv h;
  const float in3[3] = {r, g, b};
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const float u = __fdiv_rn(__fadd_rn(in3[i], 1.f), 2.f);
    h.live[i] = u >= 1e-8f && u <= 1.f;
    h.c[i] = fminf(fmaxf(u, 1e-8f), 1.f);
  }
  h.im = 0; h.in = 0;
  if (h.c[1] > h.c[h.im]) h.im = 1;
  if (h.c[2] > h.c[h.im]) h.im = 2;
  if (h.c[1] < h.c[h.in]) h.in = 1;
  if (h.c[2] < h.c[h.in]) h.in = 2;
  h.M = h.c[h.im];
  const float m = h.c[h.in];
  h.delta = __fsub_rn(h.M, m);
  h.v = h.M;
  h.s = __fdiv_rn(h.delta, __fadd_rn(h.M, 1e-8f));
  h.deltap = h.delta == 0.f ? 1.f : h.delta;
  const float rc = __fsub_rn(h.M, h.c[0]), gc = __fsub_rn(h.M, h.c[1]), bc = __fsub_rn(h.M, h.c[2]);
  float hh;
  if (h.im == 0) { h.num = __fsub_rn(bc, gc); hh = __fdiv_rn(h.num, h.deltap); }
  else if (h.im == 1) { h.num = __fsub_rn(rc, bc); hh = __fdiv_rn(__fadd_rn(h.num, __fmul_rn(2.f, h.deltap)), h.deltap); }
  else { h.num = __fsub_rn(gc, rc); hh = __fdiv_rn(__fadd_rn(h.num, __fmul_rn(4.f, h.deltap)), h.deltap); }
  const float q = __fdiv_rn(hh, 6.f);
  float fr = fmodf(q, 1.f);                       // torch's float % 1.0: fmod, then + 1 when the signs differ (a tiny negative q gives exactly 1)
  if (fr < 0.f) fr = __fadd_rn(fr, 1.f);
  h.hue_raw = __fmul_rn(TWO_PI, fr);
  h.x = fminf(fmaxf(h.hue_raw, 0.f), 1.f);
  const float sc = fminf(fmaxf(h.s, 0.f), 1.f), vc = fminf(fmaxf(h.v, 0.f), 1.f);
  h.w = sat_weighting ? __fsqrt_rn(__fmul_rn(sc, vc)) : 1.f;
  return h;
}

__device__ __forceinline__ float edge(int j, float delta) { return __fmul_rn((float)j, delta); }

// k with e_k <= x < e_k+1 (x in [0, 1]: 0 <= k <= nbins - 1 because e_nbins-1 = 1 rounds to at most 1 and e_nbins > 1)
__device__ __forceinline__ int bin_of(float x, int nbins, float delta) {
  int k = (int)floorf(x * (float)(nbins - 1));
  k = min(max(k, 0), nbins - 1);
  while (k > 0 && x < edge(k, delta)) k--;
  while (k < nbins - 1 && x >= edge(k + 1, delta)) k++;
  return k;
}

__global__ __launch_bounds__(256) void cm_hist_kernel(const float* __restrict__ img, long HW, int nbins, float delta, int sat_weighting,
                                                      unsigned long long* __restrict__ hist) {
  __shared__ unsigned long long sh[MAX_BINS];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < nbins; i += 256) sh[i] = 0ull;
  __syncthreads();
  const float* base = img + (long)b * 3 * HW;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
    const Hsv h = pixel_hsv(base[p], base[HW + p], base[2 * HW + p], sat_weighting != 0);
    const int k = bin_of(h.x, nbins, delta);
    const float lo = edge(k, delta), hi = edge(k + 1, delta);
    if (h.x >= lo && h.x < hi) {
      const float a = __fmul_rn(__fsub_rn(hi, h.x), h.w);
      atomicAdd(&sh[k], (unsigned long long)((double)a * FIX + 0.5));
      if (k + 1 <= nbins - 1) {
        const float c = __fmul_rn(__fsub_rn(h.x, lo), h.w);
        atomicAdd(&sh[k + 1], (unsigned long long)((double)c * FIX + 0.5));
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nbins; i += 256)
    if (sh[i]) atomicAdd(&hist[(long)b * nbins + i], sh[i]);
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int wv = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wv] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += red[i];
  return t;
}

// per sample: normalised histogram (hist_out, optional) and, when target != NULL, gR = dL/dR and the sample's share of the loss
__global__ __launch_bounds__(256) void cm_final_kernel(const unsigned long long* __restrict__ fix, int nbins, const float* __restrict__ target,
                                                       long target_stride, float coef /* scale / (B nbins) */, float* __restrict__ hist_out,
                                                       float* __restrict__ gR, float* __restrict__ loss) {
  __shared__ float red[4];
  __shared__ float Hs[MAX_BINS];
  const int b = blockIdx.x;
  float part = 0.f;
  for (int i = threadIdx.x; i < nbins; i += 256) {
    Hs[i] = (float)((double)fix[(long)b * nbins + i] / FIX);
    part += Hs[i];
  }
  const float S = block_sum(part, red);
  float dot = 0.f, lp = 0.f;
  for (int i = threadIdx.x; i < nbins; i += 256) {
    const float Hn = Hs[i] / S;
    if (hist_out) hist_out[(long)b * nbins + i] = Hn;
    if (target) {
      const float d = Hn - target[(long)b * target_stride + i];
      lp += d * d;
      dot += 2.f * coef * d * Hn;
    }
    Hs[i] = Hn;
  }
  if (!target) return;
  const float gdot = block_sum(dot, red);
  const float lsum = block_sum(lp, red);
  for (int i = threadIdx.x; i < nbins; i += 256) {
    const float gk = 2.f * coef * (Hs[i] - target[(long)b * target_stride + i]);
    gR[(long)b * nbins + i] = (gk - gdot) / S;
  }
  if (loss && threadIdx.x == 0) loss[b] = coef * lsum;
}

__global__ __launch_bounds__(256) void cm_grad_kernel(const float* __restrict__ img, long HW, int nbins, float delta, int sat_weighting,
                                                      const float* __restrict__ gR, float* __restrict__ grad) {
  const int b = blockIdx.y;
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const float* base = img + (long)b * 3 * HW;
  const Hsv h = pixel_hsv(base[p], base[HW + p], base[2 * HW + p], sat_weighting != 0);
  const float* g = gR + (long)b * nbins;
  const int k = bin_of(h.x, nbins, delta);
  const float lo = edge(k, delta), hi = edge(k + 1, delta);
  float dx = 0.f, dw = 0.f;
  if (h.x >= lo && h.x < hi) {
    const float g0 = g[k], g1 = k + 1 <= nbins - 1 ? g[k + 1] : 0.f;
    dx = h.w * (g1 - g0);
    dw = (hi - h.x) * g0 + (h.x - lo) * g1;
  }
  float dc[3] = {0.f, 0.f, 0.f};
  // weight = sqrt(clamp(s) clamp(v))
  float ds = 0.f, dv = 0.f;
  if (sat_weighting && h.w > 0.f) {
    const float sc = fminf(fmaxf(h.s, 0.f), 1.f), vc = fminf(fmaxf(h.v, 0.f), 1.f);
    if (h.s >= 0.f && h.s <= 1.f) ds = dw * vc / (2.f * h.w);
    if (h.v >= 0.f && h.v <= 1.f) dv = dw * sc / (2.f * h.w);
  }
  // v = max
  dc[h.im] += dv;
  // s = delta / (max + eps), delta = max - min
  const float Me = h.M + 1e-8f;
  dc[h.im] += ds * (1.f / Me - h.delta / (Me * Me));
  dc[h.in] += ds * (-1.f / Me);
  // hue: clamp(2 pi ((num / delta' + off) / 6 mod 1), 0, 1)
  if (h.hue_raw >= 0.f && h.hue_raw <= 1.f && dx != 0.f) {
    const float dhh = dx * (TWO_PI / 6.f);
    const float dnum = dhh / h.deltap;
    const int ia = h.im == 0 ? 1 : h.im == 1 ? 2 : 0;   // num = c[ia] - c[ib]: (g - b), (b - r), (r - g)
    const int ib = h.im == 0 ? 2 : h.im == 1 ? 0 : 1;
    dc[ia] += dnum;
    dc[ib] -= dnum;
    if (h.delta != 0.f) {
      const float dd = -dhh * h.num / (h.deltap * h.deltap);
      dc[h.im] += dd;
      dc[h.in] -= dd;
    }
  }
  float* o = grad + (long)b * 3 * HW + p;
#pragma unroll
  for (int i = 0; i < 3; i++) o[(long)i * HW] = h.live[i] ? 0.5f * dc[i] : 0.f;
}

// ---- the conditioning's sum over grad modules (guided.py:258-266): acc (+)= sub, or nothing when sub holds a NaN
__global__ __launch_bounds__(256) void nan_flag_kernel(const float* __restrict__ x, long n, int* __restrict__ flag) {
  bool bad = false;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) bad |= (x[i] != x[i]);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
__global__ __launch_bounds__(256) void screened_add_kernel(const float* __restrict__ sub, const int* __restrict__ flag, float* __restrict__ acc,
                                                          long n, int first) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = *flag ? 0.f : sub[i];
  acc[i] = first ? v : acc[i] + v;
}

// (zeroing by kernel, not by a memset node: these calls are also captured into the guided loop's hipGraph, where replays of small
//  memset nodes were seen misbehaving on ROCm 7.0.2 - unet.hip)
__global__ __launch_bounds__(256) void zero_u32_kernel(uint32_t* __restrict__ p, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0u;
}

int hist_into(hipStream_t st, const float* img, int B, long HW, int nbins, int sat_weighting, unsigned long long* fix) {
  const long words = (long)B * nbins * 2;
  hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, (uint32_t*)fix, words);
  const float delta = (float)(1.0 / (nbins - 1));
  const int blocks = (int)std::min<long>((HW + 255) / 256, 1024);
  hipLaunchKernelGGL(cm_hist_kernel, dim3(blocks, B), dim3(256), 0, st, img, HW, nbins, delta, sat_weighting, fix);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int run_hist(maua_ctx* ctx, const float* img, int B, long HW, int nbins, int sat_weighting, unsigned long long** fix_out, float** gr_out) {
  const size_t fix_bytes = colormatch_fix_bytes(B, nbins);
  if (int rc = scratch_reserve(ctx, fix_bytes + (size_t)B * nbins * 4)) return rc;
  unsigned long long* fix = (unsigned long long*)ctx->scratch;
  *fix_out = fix;
  *gr_out = (float*)((char*)ctx->scratch + fix_bytes);
  return hist_into(ctx->stream, img, B, HW, nbins, sat_weighting, fix);
}

}  // namespace

namespace maua {
size_t colormatch_fix_bytes(int B, int nbins) { return ((size_t)B * nbins * 8 + 255) / 256 * 256; }

// ColorMatchGrads.forward on caller-owned workspaces (fix: colormatch_fix_bytes, gr: B * nbins floats) - nothing is allocated
int colormatch_grad_into(hipStream_t st, const float* img, int B, int H, int W, int nbins, int sat_weighting, const float* target,
                         int target_per_sample, float scale, unsigned long long* fix, float* gr, float* grad, float* loss) {
  const long HW = (long)H * W;
  if (int rc = hist_into(st, img, B, HW, nbins, sat_weighting, fix)) return rc;
  const float coef = scale / ((float)B * (float)nbins);      // mse_loss: the mean over [B, nbins]
  hipLaunchKernelGGL(cm_final_kernel, dim3(B), dim3(256), 0, st, (const unsigned long long*)fix, nbins, target,
                     target_per_sample ? (long)nbins : 0L, coef, (float*)nullptr, gr, loss);
  MAUA_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(cm_grad_kernel, dim3((unsigned)((HW + 255) / 256), B), dim3(256), 0, st, img, HW, nbins, (float)(1.0 / (nbins - 1)),
                     sat_weighting, (const float*)gr, grad);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// acc = (first ? 0 : acc) + (sub holds a NaN ? 0 : sub); flag: one device int of the caller's
int screened_accumulate(hipStream_t st, const float* sub, float* acc, long n, int first, int* flag) {
  if (n == 0) return MAUA_OK;
  hipLaunchKernelGGL(zero_u32_kernel, dim3(1), dim3(256), 0, st, (uint32_t*)flag, 1L);
  const int blocks = (int)std::min<long>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(nan_flag_kernel, dim3(blocks), dim3(256), 0, st, sub, n, flag);
  hipLaunchKernelGGL(screened_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sub, (const int*)flag, acc, n, first);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}
}  // namespace maua

extern "C" {

int maua_colormatch_hist(maua_ctx* ctx, const float* img, int B, int H, int W, int nbins, int sat_weighting, float* hist) {
  MAUA_REQUIRE(ctx && img && hist, "maua_colormatch_hist: NULL argument");
  MAUA_REQUIRE(B >= 0 && H > 0 && W > 0 && nbins >= 2 && nbins <= MAX_BINS, "maua_colormatch_hist: bad shape (2 <= bins <= 1024)");
  if (B == 0) return MAUA_OK;
  unsigned long long* fix;
  float* gr;
  if (int rc = run_hist(ctx, img, B, (long)H * W, nbins, sat_weighting, &fix, &gr)) return rc;
  hipLaunchKernelGGL(cm_final_kernel, dim3(B), dim3(256), 0, ctx->stream, fix, nbins, (const float*)nullptr, 0L, 0.f, hist,
                     (float*)nullptr, (float*)nullptr);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_colormatch_grad(maua_ctx* ctx, const float* img, int B, int H, int W, int nbins, int sat_weighting, const float* target,
                         int target_per_sample, float scale, float* grad, float* loss) {
  MAUA_REQUIRE(ctx && img && target && grad, "maua_colormatch_grad: NULL argument");
  MAUA_REQUIRE(B >= 0 && H > 0 && W > 0 && nbins >= 2 && nbins <= MAX_BINS, "maua_colormatch_grad: bad shape (2 <= bins <= 1024)");
  if (B == 0) return MAUA_OK;
  const size_t fix_bytes = colormatch_fix_bytes(B, nbins);
  if (int rc = scratch_reserve(ctx, fix_bytes + (size_t)B * nbins * 4)) return rc;
  return colormatch_grad_into(ctx->stream, img, B, H, W, nbins, sat_weighting, target, target_per_sample, scale,
                              (unsigned long long*)ctx->scratch, (float*)((char*)ctx->scratch + fix_bytes), grad, loss);
}

// GradientGuidedConditioning's loop over its grad modules (guided.py:258-266): acc = (first ? 0 : acc) + (sub holds a NaN ? 0 : sub),
// n floats, no host round trip (the reference's `if torch.isnan(sub).any()` synchronises every step)
int maua_grad_accumulate(maua_ctx* ctx, const float* sub, float* acc, long n, int first) {
  MAUA_REQUIRE(ctx && sub && acc && n >= 0, "maua_grad_accumulate: bad argument");
  if (n == 0) return MAUA_OK;
  if (int rc = scratch_reserve(ctx, 256)) return rc;
  return screened_accumulate(ctx->stream, sub, acc, n, first, (int*)ctx->scratch);
}

}  // extern "C"
