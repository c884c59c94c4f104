// Operator-level modulated convolution entry point (reference-layout tensors in, reference-layout out).
// The hot path never takes this route (maua_synth_forward keeps activations NHWC and weights prepared);
// this entry exists so that users of the reference's ops.modulated_conv2d — and the parity tests — have the
// same operator, executed by the same MFMA kernel.
#include <cmath>

#include "common.h"
#include "internal.h"

namespace maua {

// d[b][co] = rsqrt(sum_ci s^2 * wsq + 1e-8) (ops.py:168-171); one wave per (b, co)
__global__ __launch_bounds__(256) void demod_kernel(const float* __restrict__ s, const float* __restrict__ wsq,
                                                    float* __restrict__ d, int Ci, int Co, int Cs, int Cd) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int co = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (co >= Cd) return;
  float dv = 0.f;
  if (co < Co) {
    float acc = 0.f;
    for (int k = lane; k < Ci; k += 64) {
      float sv = s[(long)b * Cs + k];
      acc += sv * sv * wsq[(long)co * Ci + k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    dv = rsqrtf(acc + 1e-8f);
  }
  if (lane == 0) d[(long)b * Cd + co] = dv;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace maua

using namespace maua;

extern "C" int maua_modconv2d(maua_ctx* ctx, const void* x, const float* weight, const float* styles,
                              const float* noise, long noise_batch_stride, float noise_strength, const float* bias,
                              void* y, int N, int Ci, int Co, int H, int W, int k, int up, int demodulate,
                              int flip_weight, int act, float alpha, float gain, float clamp, int dtype) {
  MAUA_REQUIRE(ctx, "maua_modconv2d: ctx is NULL");
  if (N == 0) return MAUA_OK;
  MAUA_REQUIRE(x && weight && styles && y, "maua_modconv2d: NULL argument");
  MAUA_REQUIRE(dtype == MAUA_F32 || dtype == MAUA_BF16 || dtype == MAUA_F16, "maua_modconv2d: unsupported dtype");
  MAUA_REQUIRE(k == 1 || k == 3, "maua_modconv2d: kernel size must be 1 or 3");
  MAUA_REQUIRE(up == 1 || (up == 2 && k == 3), "maua_modconv2d: up must be 1, or 2 with a 3x3 kernel");
  MAUA_REQUIRE(N >= 0 && Ci > 0 && Co > 0 && H > 0 && W > 0, "maua_modconv2d: bad shape");
  if (N == 0) return MAUA_OK;
  const size_t es = dtype == MAUA_F32 ? 4 : 2;
  const int Cip = (Ci + 31) / 32 * 32, Cop = (Co + 31) / 32 * 32;
  const int Ho = H * up, Wo = W * up;
  const size_t wt_elems = prepped_weight_elems(3, up, Cop, Cip);
  // carve the scratch arena
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    size_t o = off;
    off += align256(bytes);
    return o;
  };
  size_t o_x = carve((size_t)N * H * W * Cip * es), o_y = carve((size_t)N * Ho * Wo * Cop * es);
  size_t o_w = carve(wt_elems * es), o_wsq = carve((size_t)Co * Ci * 4), o_s = carve((size_t)N * Cip * 4);
  size_t o_d = carve((size_t)N * Cop * 4), o_b = carve((size_t)Cop * 4);
  // ops.py:161-165: "if x.dtype == torch.float16 and demodulate" the weight and the styles are pre-normalised
  const bool prenorm = dtype == MAUA_F16 && demodulate;
  size_t o_wn = prenorm ? carve((size_t)Co * Ci * k * k * 4) : 0;
  const bool dma = ctx->dma_conv && k == 3 && dma_conv_supported(dtype, Cip, Cop, up, H, W);
  size_t o_xm = dma ? carve((size_t)N * H * W * Cip * es) : 0;
  if (int rc = scratch_reserve(ctx, off)) return rc;
  char* base = (char*)ctx->scratch;
  hipStream_t st = ctx->stream;
  void* xn = base + o_x;
  void* yn = base + o_y;
  void* wt = base + o_w;
  float* wsq = (float*)(base + o_wsq);
  float* sp = (float*)(base + o_s);
  float* dp = (float*)(base + o_d);
  float* bp = (float*)(base + o_b);

  int rc = dtype == MAUA_BF16   ? launch_nchw_to_nhwc<bf16_t, bf16_t>(st, x, xn, N, Ci, H * W, Cip)
           : dtype == MAUA_F16 ? launch_nchw_to_nhwc<f16_t, f16_t>(st, x, xn, N, Ci, H * W, Cip)
                               : launch_nchw_to_nhwc<float, float>(st, x, xn, N, Ci, H * W, Cip);
  if (rc) return rc;
  if (prenorm) {
    float* wn = (float*)(base + o_wn);
    if ((rc = launch_f16_prenorm_weights(st, weight, wn, Co, Ci, k * k))) return rc;
    weight = wn;
  }
  MAUA_HIP_CHECK(hipMemsetAsync(wt, 0, wt_elems * es, st));
  // a 1x1 kernel is executed as the centre tap of a zero 3x3 kernel
  void* wt_dst = (k == 1) ? (void*)((char*)wt + (size_t)4 * Cop * Cip * es) : wt;
  if ((rc = launch_prep_weights(st, dtype, weight, wt_dst, wsq, Co, Ci, k, up, (up == 2) ? flip_weight : 0, Cop, Cip)))
    return rc;
  MAUA_HIP_CHECK(hipMemsetAsync(sp, 0, (size_t)N * Cip * 4, st));
  MAUA_HIP_CHECK(hipMemcpy2DAsync(sp, (size_t)Cip * 4, styles, (size_t)Ci * 4, (size_t)Ci * 4, N,
                                  hipMemcpyDeviceToDevice, st));
  if (prenorm)
    if ((rc = launch_f16_prenorm_styles(st, sp, N, Cip, Ci))) return rc;
  if (demodulate) {
    hipLaunchKernelGGL(demod_kernel, dim3(cdiv(Cop, 4), N), dim3(256), 0, st, sp, wsq, dp, Ci, Co, Cip, Cop);
    MAUA_HIP_CHECK(hipGetLastError());
  }
  MAUA_HIP_CHECK(hipMemsetAsync(bp, 0, (size_t)Cop * 4, st));
  if (bias) MAUA_HIP_CHECK(hipMemcpyAsync(bp, bias, (size_t)Co * 4, hipMemcpyDeviceToDevice, st));

  ConvArgs a{};
  a.x = xn; a.x_bstride = (long)H * W * Cip; a.w = wt; a.s = sp; a.d = demodulate ? dp : nullptr;
  a.noise = noise; a.noise_bstride = noise_batch_stride; a.noise_strength = noise_strength;
  a.bias = bp; a.y = yn; a.B = N; a.H = H; a.W = W; a.Ci = Cip; a.Co = Cop; a.up = up;
  a.act = act; a.alpha = alpha; a.gain = gain; a.clamp = clamp;
  if (dma) {  // the hot path's kernel for this shape: styles applied to the input first (there the producer does it)
    if ((rc = launch_premod_nhwc(st, xn, a.x_bstride, sp, base + o_xm, N, (long)H * W, Cip, dtype))) return rc;
    a.x = base + o_xm;
    if ((rc = launch_modconv_dma(st, a, dtype))) return rc;
  } else if ((rc = launch_modconv3x3(st, dtype, a))) {
    return rc;
  }
  return dtype == MAUA_BF16   ? launch_nhwc_to_nchw<bf16_t, bf16_t>(st, yn, y, N, Co, Ho * Wo, Cop)
         : dtype == MAUA_F16 ? launch_nhwc_to_nchw<f16_t, f16_t>(st, yn, y, N, Co, Ho * Wo, Cop)
                             : launch_nhwc_to_nchw<float, float>(st, yn, y, N, Co, Ho * Wo, Cop);
}
