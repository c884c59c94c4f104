// Feature-space / image resizing for arbitrary output sizes.
//
// Replaces (reference, maua/GAN/wrappers/stylegan2.py:216-340 get_hook): the "stretch" strategy's
// torch.nn.functional.interpolate(x, size, mode="bicubic", align_corners=False) (:231, :253) and the "pad-*" strategies'
// torch.nn.functional.pad(x, padding, mode, value) (:294) with their inverses (bicubic back / crop, :253, :313-323),
// plus the per-channel fill noise added to resized features (:233-248, :296-311).
// Bicubic follows ATen's upsample_bicubic2d: source index (o + 0.5) * in/out - 0.5, Keys kernel A = -0.75, the four
// taps clamped to the image, x pass then y pass, float32 arithmetic.  HBM-bound gathers; NHWC for features
// (16-byte channel pieces are overkill here: these run once per forward on one layer), planar f32 for RGB images.
#include "common.h"
#include "internal.h"

namespace maua {

__device__ __forceinline__ void cubic_coeffs(float t, float (&c)[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
  c[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
  c[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
  c[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
  c[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

// source coordinate of padded position p for F.pad's modes; returns -1 when the value is the constant
__device__ __forceinline__ int pad_src(int p, int n, int how) {
  if (p >= 0 && p < n) return p;
  switch (how) {
    case MAUA_PAD_REFLECT: {
      if (n == 1) return 0;
      const int period = 2 * (n - 1);
      int q = p % period;
      if (q < 0) q += period;
      return q < n ? q : period - q;
    }
    case MAUA_PAD_REPLICATE: return p < 0 ? 0 : n - 1;
    case MAUA_PAD_CIRCULAR: {
      int q = p % n;
      return q < 0 ? q + n : q;
    }
    default: return -1;
  }
}

template <typename T, bool NHWC>
__global__ __launch_bounds__(256) void resize2d_kernel(ResizeArgs a) {
  // one thread per output element; NHWC: channel fastest, planar: x fastest
  const long total = (long)a.B * a.C * a.oh * a.ow;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int b, c, oy, ox;
  if (NHWC) {
    c = (int)(idx % a.C);
    long r = idx / a.C;
    ox = (int)(r % a.ow); r /= a.ow;
    oy = (int)(r % a.oh);
    b = (int)(r / a.oh);
  } else {
    ox = (int)(idx % a.ow);
    long r = idx / a.ow;
    oy = (int)(r % a.oh); r /= a.oh;
    c = (int)(r % a.C);
    b = (int)(r / a.C);
  }
  const T* xb = reinterpret_cast<const T*>(a.x) + (long)b * a.x_bstride;
  auto at = [&](int y, int x) -> float {
    const long o = NHWC ? ((long)y * a.W + x) * a.C + c : ((long)c * a.H + y) * a.W + x;
    return Elem<T>::load(xb + o);
  };
  float v;
  if (a.mode == 0) {  // bicubic
    // ATen area_pixel_compute_source_index: align_corners -> o * (in - 1) / (out - 1), else half-pixel centres
    const float sy = a.align ? (a.oh > 1 ? oy * ((float)(a.H - 1) / (float)(a.oh - 1)) : 0.f)
                             : (oy + 0.5f) * ((float)a.H / (float)a.oh) - 0.5f;
    const float sx = a.align ? (a.ow > 1 ? ox * ((float)(a.W - 1) / (float)(a.ow - 1)) : 0.f)
                             : (ox + 0.5f) * ((float)a.W / (float)a.ow) - 0.5f;
    const float fy = floorf(sy), fx = floorf(sx);
    float cy[4], cx[4];
    cubic_coeffs(sy - fy, cy);
    cubic_coeffs(sx - fx, cx);
    const int iy = (int)fy, ix = (int)fx;
    float rows[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int y = min(max(iy - 1 + i, 0), a.H - 1);
      float r = 0.f;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int x = min(max(ix - 1 + j, 0), a.W - 1);
        r += at(y, x) * cx[j];
      }
      rows[i] = r;
    }
    v = rows[0] * cy[0] + rows[1] * cy[1] + rows[2] * cy[2] + rows[3] * cy[3];
  } else {  // pad (positive offsets) / crop (negative offsets): output (oy, ox) <- input (oy - pt, ox - pl)
    const int y = pad_src(oy - a.pt, a.H, a.how), x = pad_src(ox - a.pl, a.W, a.how);
    v = (y < 0 || x < 0) ? a.value : at(y, x);
  }
  if (a.noise) v += a.noise[((long)c * a.oh + oy) * a.ow + ox];
  T* yb = reinterpret_cast<T*>(a.y) + (long)b * a.C * a.oh * a.ow;
  const long o = NHWC ? ((long)oy * a.ow + ox) * a.C + c : ((long)c * a.oh + oy) * a.ow + ox;
  Elem<T>::store(yb + o, v);
}

int launch_resize2d(hipStream_t stream, int dtype, bool nhwc, const ResizeArgs& a) {
  const long total = (long)a.B * a.C * a.oh * a.ow;
  if (total == 0) return MAUA_OK;
  MAUA_REQUIRE(a.H > 0 && a.W > 0, "resize2d: empty input");
  MAUA_REQUIRE(a.mode == 0 || (a.how >= 0 && a.how <= 3), "resize2d: unknown padding mode");
  if (a.mode == 1 && a.how == MAUA_PAD_REFLECT)
    MAUA_REQUIRE(a.pl < a.W && a.ow - a.W - a.pl < a.W && a.pt < a.H && a.oh - a.H - a.pt < a.H,
                 "resize2d: reflect padding must be smaller than the input (as torch)");
  const dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == MAUA_F32) {
    if (nhwc) hipLaunchKernelGGL((resize2d_kernel<float, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((resize2d_kernel<float, false>), grid, dim3(256), 0, stream, a);
  } else if (dtype == MAUA_BF16) {
    if (nhwc) hipLaunchKernelGGL((resize2d_kernel<bf16_t, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((resize2d_kernel<bf16_t, false>), grid, dim3(256), 0, stream, a);
  } else if (dtype == MAUA_F16) {
    if (nhwc) hipLaunchKernelGGL((resize2d_kernel<f16_t, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((resize2d_kernel<f16_t, false>), grid, dim3(256), 0, stream, a);
  } else {
    return fail("resize2d: unsupported dtype");
  }
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// depthwise 1-D correlation along H (axis 0) or W (axis 1) of planar f32 [planes][H][W] with reflect padding: the
// lanczos pre-filter of maua/ops/image.py:226-236 (F.pad reflect + F.conv2d with a [k,1] / [1,k] kernel)
__global__ __launch_bounds__(256) void conv1d_planar_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            const float* __restrict__ taps, int radius, int axis, int H,
                                                            int W, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int xx = (int)(idx % W);
  const long r = idx / W;
  const int yy = (int)(r % H);
  const float* pl = x + (r / H) * H * W;
  float acc = 0.f;
  for (int k = -radius; k <= radius; k++) {
    const int sy = axis == 0 ? pad_src(yy + k, H, MAUA_PAD_REFLECT) : yy;
    const int sx = axis == 1 ? pad_src(xx + k, W, MAUA_PAD_REFLECT) : xx;
    acc += pl[(long)sy * W + sx] * taps[k + radius];
  }
  y[idx] = acc;
}

// adjoint of conv1d_planar_kernel: gx[i] = sum over (j, k) with reflect(j + k) == i of taps[k + radius] gy[j].  The padded positions that
// reflect onto i are m = i, m = -i (i >= 1) and m = 2 (n - 1) - i (i <= n - 2) (radius < n: one reflection at most)
__global__ __launch_bounds__(256) void conv1d_planar_vjp_kernel(const float* __restrict__ gy, float* __restrict__ gx, const float* __restrict__ taps,
                                                                int radius, int axis, int H, int W, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int xx = (int)(idx % W);
  const long r = idx / W;
  const int yy = (int)(r % H);
  const float* pl = gy + (r / H) * H * W;
  const int n = axis == 0 ? H : W, i = axis == 0 ? yy : xx;
  const long stride = axis == 0 ? W : 1;
  const float* line = pl + (axis == 0 ? (long)xx : (long)yy * W);
  const int ms[3] = {i, -i, 2 * (n - 1) - i};
  const bool ok[3] = {true, i >= 1, i <= n - 2 && n > 1};
  float acc = 0.f;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    if (!ok[c]) continue;
    for (int k = -radius; k <= radius; k++) {
      const int j = ms[c] - k;
      if (j >= 0 && j < n) acc += taps[k + radius] * line[(long)j * stride];
    }
  }
  gx[idx] = acc;
}

// one axis of the bicubic adjoint: src [planes][rows][n_out] -> dst [planes][rows][n_in] (axis 1) or src [planes][n_out][cols] ->
// dst [planes][n_in][cols] (axis 0; `rows` is then the column count): dst[i] = sum_o w(o -> i) src[o] over the outputs whose clamped taps hit i
__global__ __launch_bounds__(256) void bicubic_axis_vjp_kernel(const float* __restrict__ src, float* __restrict__ dst, int n_out_rows, int n_out_or_cols,
                                                               int n_in, int axis, int align, long total) {
  // axis 1: src [planes][R = n_out_rows][n_out = n_out_or_cols], dst [planes][R][n_in]
  // axis 0: src [planes][n_out = n_out_rows][Cc = n_out_or_cols], dst [planes][n_in][Cc]
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int i, n_out;
  const float* line;
  long stride;
  if (axis == 1) {
    n_out = n_out_or_cols;
    i = (int)(idx % n_in);
    const long row = idx / n_in;                      // plane * R + r
    line = src + row * n_out;
    stride = 1;
  } else {
    n_out = n_out_rows;
    const int cols = n_out_or_cols;
    const int x = (int)(idx % cols);
    const long q = idx / cols;
    i = (int)(q % n_in);
    line = src + (q / n_in) * (long)n_out * cols + x;
    stride = cols;
  }
  const float sc = align ? (n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f) : (float)n_in / (float)n_out;
  int lo = 0, hi = n_out - 1;
  if (sc > 0.f) {
    lo = max(0, (int)floorf(((float)i - 3.f) / sc) - 1);
    hi = min(n_out - 1, (int)ceilf(((float)i + 3.f) / sc) + 1);
  }
  float acc = 0.f;
  for (int o = lo; o <= hi; o++) {
    const float sp = align ? (n_out > 1 ? o * sc : 0.f) : (o + 0.5f) * sc - 0.5f;
    const float f = floorf(sp);
    float c[4];
    cubic_coeffs(sp - f, c);
    const int i0 = (int)f - 1;
    float w = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (min(max(i0 + k, 0), n_in - 1) == i) w += c[k];
    if (w != 0.f) acc += w * line[(long)o * stride];
  }
  dst[idx] = acc;
}

// Per-sample affine warp of NHWC features with bilinear sampling and reflection padding — what kornia's translate /
// rotate / scale (wrappers/stylegan2.py:153-194; kornia = warp_affine -> F.affine_grid + F.grid_sample(bilinear,
// padding_mode="reflection", align_corners=True)) do to a layer's output.  minv [B][6] maps OUTPUT pixel (x, y) to the
// SOURCE pixel: sx = m0 x + m1 y + m2, sy = m3 x + m4 y + m5.
__device__ __forceinline__ float reflect_coord(float x, int size) {
  // grid_sample reflection, align_corners=True: reflect about 0 and size-1, then clip
  if (size <= 1) return 0.f;
  const float span = (float)(size - 1);
  x = fabsf(x);
  const float flips = floorf(x / span);
  const float extra = x - flips * span;
  x = (((int)flips) & 1) ? span - extra : extra;
  return fminf(fmaxf(x, 0.f), span);
}

template <typename T>
__global__ __launch_bounds__(256) void warp_affine_nhwc_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                               const float* __restrict__ minv, int H, int W, int C,
                                                               long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  long r = idx / C;
  const int ox = (int)(r % W); r /= W;
  const int oy = (int)(r % H);
  const int b = (int)(r / H);
  const float* m = minv + (long)b * 6;
  const float sx = reflect_coord(m[0] * ox + m[1] * oy + m[2], W);
  const float sy = reflect_coord(m[3] * ox + m[4] * oy + m[5], H);
  const float fx = floorf(sx), fy = floorf(sy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float tx = sx - fx, ty = sy - fy;
  const T* xb = x + (long)b * H * W * C;
  auto at = [&](int yy, int xx) -> float {
    return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? Elem<T>::load(xb + ((long)yy * W + xx) * C + c) : 0.f;
  };
  // grid_sample's order: nw, ne, sw, se
  const float v = at(y0, x0) * ((1.f - tx) * (1.f - ty)) + at(y0, x0 + 1) * (tx * (1.f - ty)) +
                  at(y0 + 1, x0) * ((1.f - tx) * ty) + at(y0 + 1, x0 + 1) * (tx * ty);
  Elem<T>::store(y + idx, v);
}

int launch_warp_affine_nhwc(hipStream_t stream, int dtype, const void* x, void* y, const float* minv, int B, int H, int W,
                            int C) {
  const long total = (long)B * H * W * C;
  if (total == 0) return MAUA_OK;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == MAUA_F32)
    hipLaunchKernelGGL(warp_affine_nhwc_kernel<float>, grid, dim3(256), 0, stream, (const float*)x, (float*)y, minv, H, W, C, total);
  else if (dtype == MAUA_BF16)
    hipLaunchKernelGGL(warp_affine_nhwc_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, minv, H, W, C, total);
  else if (dtype == MAUA_F16)
    hipLaunchKernelGGL(warp_affine_nhwc_kernel<f16_t>, grid, dim3(256), 0, stream, (const f16_t*)x, (f16_t*)y, minv, H, W, C, total);
  else
    return fail("warp_affine: unsupported dtype");
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// img = upsample2d(prev) + y (stylegan2.py:372-378 with ops.py:117-133), planar f32 [B][3][H][W], prev [B][3][H/2][W/2]
__global__ __launch_bounds__(256) void skip_add_kernel(const float* __restrict__ y, const float* __restrict__ prev,
                                                       float* __restrict__ out, int H, int W, float f0, float f1, float f4,
                                                       float f5, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int x = (int)(idx % W);
  long r = idx / W;
  const int yy = (int)(r % H);
  const long plane = r / H;  // b * 3 + c
  const int Hp = H >> 1, Wp = W >> 1;
  const float* pv = prev + plane * Hp * Wp;
  const int iy0 = (yy - 1) >> 1, ix0 = (x - 1) >> 1;
  float u = 0.f;
#pragma unroll
  for (int dy = 0; dy < 2; dy++) {
    const int iy = iy0 + dy, uu = 2 * iy - yy + 2;
    const bool oky = iy >= 0 && iy < Hp;
#pragma unroll
    for (int dx = 0; dx < 2; dx++) {
      const int ix = ix0 + dx, vv = 2 * ix - x + 2;
      const bool ok = oky && ix >= 0 && ix < Wp;
      const bool uh = uu == 1 || uu == 2, vh = vv == 1 || vv == 2;
      const float f = !ok ? 0.f : uh ? (vh ? f5 : f4) : (vh ? f1 : f0);
      u += pv[ok ? (long)iy * Wp + ix : 0] * f;
    }
  }
  out[idx] = u + y[idx];
}

int launch_skip_add(hipStream_t stream, const float* y, const float* prev, float* out, int B, int H, int W,
                    const float* fir16) {
  const long total = (long)B * 3 * H * W;
  if (total == 0) return MAUA_OK;
  MAUA_REQUIRE(H % 2 == 0 && W % 2 == 0, "skip_add: the image must be twice the previous one");
  hipLaunchKernelGGL(skip_add_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, y, prev, out, H, W,
                     fir16[0], fir16[1], fir16[4], fir16[5], total);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace maua

using namespace maua;

extern "C" {

int maua_conv1d_reflect(maua_ctx* ctx, const float* x, float* y, const float* taps, int radius, int axis, long planes,
                        int H, int W) {
  MAUA_REQUIRE(ctx, "maua_conv1d_reflect: ctx is NULL");
  const long total = planes * H * W;
  if (total == 0) return MAUA_OK;
  MAUA_REQUIRE(x && y && taps && x != y, "maua_conv1d_reflect: NULL or aliased argument");
  MAUA_REQUIRE(axis == 0 || axis == 1, "maua_conv1d_reflect: axis must be 0 (rows) or 1 (columns)");
  MAUA_REQUIRE(radius >= 0 && radius < (axis == 0 ? H : W), "maua_conv1d_reflect: reflect padding must be smaller than the image");
  hipLaunchKernelGGL(conv1d_planar_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, x, y, taps,
                     radius, axis, H, W, total);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// ---- adjoints (the gradient of a loss through `resample`, maua/ops/image.py:214-240: what autograd gives LPIPSGrads, grad.py:191-192)
int maua_conv1d_reflect_vjp(maua_ctx* ctx, const float* gy, float* gx, const float* taps, int radius, int axis, long planes, int H, int W) {
  MAUA_REQUIRE(ctx, "maua_conv1d_reflect_vjp: ctx is NULL");
  const long total = planes * H * W;
  if (total == 0) return MAUA_OK;
  MAUA_REQUIRE(gy && gx && taps && gy != gx, "maua_conv1d_reflect_vjp: NULL or aliased argument");
  MAUA_REQUIRE(axis == 0 || axis == 1, "maua_conv1d_reflect_vjp: axis must be 0 (rows) or 1 (columns)");
  MAUA_REQUIRE(radius >= 0 && radius < (axis == 0 ? H : W), "maua_conv1d_reflect_vjp: reflect padding must be smaller than the image");
  hipLaunchKernelGGL(conv1d_planar_vjp_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, gy, gx, taps, radius, axis,
                     H, W, total);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// gy [N][C][out_h][out_w] -> gx [N][C][H][W]: the adjoint of maua_resize2d's bicubic modes (0: align_corners False, 2: True), as two
// 1-D gather passes (columns, then rows) over the outputs whose four clamped taps touch an input sample - no atomics
int maua_resize2d_bicubic_vjp(maua_ctx* ctx, const float* gy, float* gx, int N, int C, int H, int W, int out_h, int out_w, int align_corners) {
  MAUA_REQUIRE(ctx, "maua_resize2d_bicubic_vjp: ctx is NULL");
  const long planes = (long)N * C;
  if (planes * H * W == 0) return MAUA_OK;
  MAUA_REQUIRE(gy && gx && out_h > 0 && out_w > 0, "maua_resize2d_bicubic_vjp: bad argument");
  if (int rc = scratch_reserve(ctx, (size_t)planes * out_h * W * 4)) return rc;
  float* tmp = (float*)ctx->scratch;
  const long t1 = planes * out_h * W, t2 = planes * H * W;
  hipLaunchKernelGGL(bicubic_axis_vjp_kernel, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, ctx->stream, gy, tmp, out_h, out_w, W, 1,
                     align_corners, t1);
  hipLaunchKernelGGL(bicubic_axis_vjp_kernel, dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, ctx->stream, (const float*)tmp, gx, out_h, W, H, 0,
                     align_corners, t2);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_resize2d(maua_ctx* ctx, const void* x, void* y, int N, int C, int H, int W, int out_h, int out_w, int mode,
                  int pad_left, int pad_top, int pad_how, float pad_value, int dtype) {
  MAUA_REQUIRE(ctx, "maua_resize2d: ctx is NULL");
  if ((long)N * C * out_h * out_w == 0) return MAUA_OK;
  MAUA_REQUIRE(x && y, "maua_resize2d: NULL argument");
  MAUA_REQUIRE(mode >= 0 && mode <= 2, "maua_resize2d: mode must be 0 (bicubic), 1 (pad / crop) or 2 (bicubic, align_corners)");
  ResizeArgs a{};
  a.align = mode == 2;
  if (mode == 2) mode = 0;
  a.x = x; a.x_bstride = (long)C * H * W; a.y = y; a.B = N; a.H = H; a.W = W; a.C = C; a.oh = out_h; a.ow = out_w;
  a.mode = mode; a.pl = pad_left; a.pt = pad_top; a.how = pad_how; a.value = pad_value; a.noise = nullptr;
  return launch_resize2d(ctx->stream, dtype, false, a);
}

}  // extern "C"
