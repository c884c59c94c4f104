// Operator-layer kernels (NCHW, reference tensor layout): bias_act, upfirdn2d, pack_rgb8, layout converters.
// HBM-bound byte/elementwise work: coalesced 16-byte accesses, grid-stride, no GEMM reshaping.
#include "common.h"
#include "internal.h"

namespace maua {

// ------------------------------------------------------------------------------------------------ bias_act
// reference ops.py:65-84.  4 elements per thread (16 B for f32, 8 B for bf16) when H*W % 4 == 0 so that a
// vector never straddles a channel; scalar tail kernel otherwise.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void bias_act_kernel(const T* __restrict__ x, const float* __restrict__ b,
                                                       T* __restrict__ y, long total_vec, int C, int HW, int act,
                                                       float alpha, float gain, float clamp) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < total_vec; i += stride) {
    long e = i * VEC;
    int c = (int)((e / HW) % C);
    float bv = b ? b[c] : 0.f;
    float v[VEC];
    if constexpr (VEC == 4 && sizeof(T) == 4) {
      float4 t = *reinterpret_cast<const float4*>(x + e);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else if constexpr (VEC == 4 && sizeof(T) == 2) {
      uint2 t = *reinterpret_cast<const uint2*>(x + e);
      v[0] = Fmt16<T>::lo(t.x); v[1] = Fmt16<T>::hi(t.x);
      v[2] = Fmt16<T>::lo(t.y); v[3] = Fmt16<T>::hi(t.y);
    } else {
      v[0] = Elem<T>::load(x + e);
    }
#pragma unroll
    for (int k = 0; k < VEC; k++) {
      float t = activate(v[k] + bv, act, alpha);
      if (gain != 1.f) t *= gain;
      if (clamp >= 0.f) t = fminf(fmaxf(t, -clamp), clamp);
      v[k] = t;
    }
    if constexpr (VEC == 4 && sizeof(T) == 4) {
      *reinterpret_cast<float4*>(y + e) = make_float4(v[0], v[1], v[2], v[3]);
    } else if constexpr (VEC == 4 && sizeof(T) == 2) {
      *reinterpret_cast<uint2*>(y + e) = make_uint2(Fmt16<T>::pack2(v[0], v[1]), Fmt16<T>::pack2(v[2], v[3]));
    } else {
      Elem<T>::store(y + e, v[0]);
    }
  }
}

template <typename T>
static int launch_bias_act(maua_ctx* ctx, const void* x, const float* b, void* y, int N, int C, int H, int W, int act,
                           float alpha, float gain, float clamp) {
  long total = (long)N * C * H * W;
  if (total == 0) return MAUA_OK;
  int HW = H * W;
  bool vec = (HW % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0);
  long tv = vec ? total / 4 : total;
  int grid = (int)std::min<long>((tv + 255) / 256, 256 * 8);
  if (vec)
    hipLaunchKernelGGL((bias_act_kernel<T, 4>), dim3(grid), dim3(256), 0, ctx->stream, (const T*)x, b, (T*)y, tv, C, HW,
                       act, alpha, gain, clamp);
  else
    hipLaunchKernelGGL((bias_act_kernel<T, 1>), dim3(grid), dim3(256), 0, ctx->stream, (const T*)x, b, (T*)y, tv, C, HW,
                       act, alpha, gain, clamp);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// ------------------------------------------------------------------------------------------------ residual add
// out = a + b (summed in f32): the "resnet" blocks' x = y + x and the skip images' img + y (stylegan2.py:360, :373) for callers
// that run the network a layer at a time.  HBM-bound: 16 bytes per lane per operand.
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, long n,
                                                  int vec) {
  constexpr int V = 16 / sizeof(T);
  const long stride = (long)gridDim.x * blockDim.x;
  if (vec) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n / V; i += stride) {
      if constexpr (sizeof(T) == 4) {
        float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
        reinterpret_cast<float4*>(o)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
      } else {
        uint4 x = reinterpret_cast<const uint4*>(a)[i], y = reinterpret_cast<const uint4*>(b)[i];
        const unsigned xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
        unsigned r[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
          r[k] = Fmt16<T>::pack2(Fmt16<T>::lo(xs[k]) + Fmt16<T>::lo(ys[k]), Fmt16<T>::hi(xs[k]) + Fmt16<T>::hi(ys[k]));
        reinterpret_cast<uint4*>(o)[i] = make_uint4(r[0], r[1], r[2], r[3]);
      }
    }
    for (long i = (n / V) * V + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
      Elem<T>::store(o + i, Elem<T>::load(a + i) + Elem<T>::load(b + i));
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
      Elem<T>::store(o + i, Elem<T>::load(a + i) + Elem<T>::load(b + i));
  }
}

template <typename T>
static int launch_add(maua_ctx* ctx, const void* a, const void* b, void* o, long n) {
  const int vec = ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0) && ((uintptr_t)o % 16 == 0);
  const long work = vec ? (n * (long)sizeof(T) + 15) / 16 : n;
  const int grid = (int)std::min<long>((work + 255) / 256, 256 * 16);
  hipLaunchKernelGGL((add_kernel<T>), dim3(grid), dim3(256), 0, ctx->stream, (const T*)a, (const T*)b, (T*)o, n, vec);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// ------------------------------------------------------------------------------------------------ upfirdn2d
// reference ops.py:87-114.  One workgroup = one 16x64 output tile of one (n,c) plane.  The input footprint of
// the tile (after zero-insertion only every up-th sample is non-zero, so the footprint is stored compacted at
// input resolution) is staged in LDS once; each thread then accumulates its 4 outputs over the taps whose
// parity lands on a real sample.  Correlation, no flip (ops.py:107-111).
constexpr int UF_TH = 16, UF_TW = 64, UF_MAXF = 32;

template <typename T>
__global__ __launch_bounds__(256) void upfirdn2d_kernel(const T* __restrict__ x, const float* __restrict__ f, int fh,
                                                        int fw, T* __restrict__ y, int H, int W, int Ho, int Wo, int up,
                                                        int down, int px0, int py0, float gain, int lh, int lw) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* fs = reinterpret_cast<float*>(smem_raw);  // fh*fw taps (gain folded in)
  float* tile = fs + UF_MAXF * UF_MAXF;            // lh x lw compact input footprint
  const int plane = blockIdx.z;
  const int oy0 = blockIdx.y * UF_TH, ox0 = blockIdx.x * UF_TW;
  for (int i = threadIdx.x; i < fh * fw; i += blockDim.x) fs[i] = f[i] * gain;
  // footprint in zero-inserted coordinates starts at Y0 = oy0*down - py0; first real sample row >= Y0:
  const int Y0 = oy0 * down - py0, X0 = ox0 * down - px0;
  // floor-div that is safe for negatives
  auto cdiv_up = [](int a, int b) { return (a >= 0) ? (a + b - 1) / b : -((-a) / b); };
  const int iy0 = cdiv_up(Y0, up), ix0 = cdiv_up(X0, up);
  const T* xp = x + (long)plane * H * W;
  for (int i = threadIdx.x; i < lh * lw; i += blockDim.x) {
    int r = i / lw, c = i - r * lw;
    int iy = iy0 + r, ix = ix0 + c;
    float v = 0.f;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = Elem<T>::load(xp + (long)iy * W + ix);
    tile[i] = v;
  }
  __syncthreads();
  T* yp = y + (long)plane * Ho * Wo;
  const int tx = threadIdx.x & 63, ty4 = threadIdx.x >> 6;  // 64 columns, 4 row groups of 4 rows
#pragma unroll
  for (int rr = 0; rr < 4; rr++) {
    int oy = oy0 + ty4 * 4 + rr, ox = ox0 + tx;
    if (oy >= Ho || ox >= Wo) continue;
    int Yb = oy * down - py0, Xb = ox * down - px0;  // zero-inserted coordinate of tap (0,0)
    float acc = 0.f;
    for (int u = 0; u < fh; u++) {
      int Y = Yb + u;
      if (Y < 0 || (Y % up) != 0) continue;
      int r = Y / up - iy0;
      if (r < 0 || r >= lh) continue;
      for (int v = 0; v < fw; v++) {
        int X = Xb + v;
        if (X < 0 || (X % up) != 0) continue;
        int c = X / up - ix0;
        if (c < 0 || c >= lw) continue;
        acc += tile[r * lw + c] * fs[u * fw + v];
      }
    }
    Elem<T>::store(yp + (long)oy * Wo + ox, acc);
  }
}

template <typename T>
static int launch_upfirdn2d(maua_ctx* ctx, const void* x, const float* f, int fh, int fw, void* y, int N, int C, int H,
                            int W, int up, int down, int px0, int px1, int py0, int py1, float gain) {
  int Ho = (H * up + py0 + py1 - fh) / down + 1, Wo = (W * up + px0 + px1 - fw) / down + 1;
  MAUA_REQUIRE(Ho > 0 && Wo > 0, "maua_upfirdn2d: empty output");
  // compact footprint rows/cols needed by one tile (+2 slack for the ceil at both ends)
  int lh = ((UF_TH - 1) * down + fh - 1) / up + 2, lw = ((UF_TW - 1) * down + fw - 1) / up + 2;
  size_t smem = (UF_MAXF * UF_MAXF + (size_t)lh * lw) * sizeof(float);
  MAUA_REQUIRE(smem <= 64 * 1024, "maua_upfirdn2d: tile footprint too large (down factor too big)");
  dim3 grid(cdiv(Wo, UF_TW), cdiv(Ho, UF_TH), N * C);
  hipLaunchKernelGGL((upfirdn2d_kernel<T>), grid, dim3(256), smem, ctx->stream, (const T*)x, f, fh, fw, (T*)y, H, W, Ho,
                     Wo, up, down, px0, py0, gain, lh, lw);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// ------------------------------------------------------------------------------------------------ pack_rgb8
// render/ffmpeg.py:72 + ops/io.py:47-70.  4 pixels per thread: 3x float4 planar loads, 3x u32 interleaved store.

__global__ __launch_bounds__(256) void pack_rgb8_kernel(const float* __restrict__ img, uint8_t* __restrict__ out,
                                                        long npix4, long HW) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < npix4; i += stride) {
    long p = i * 4;
    long b = p / HW, q = p - b * HW;
    const float* base = img + b * 3 * HW + q;
    float4 r = *reinterpret_cast<const float4*>(base);
    float4 g = *reinterpret_cast<const float4*>(base + HW);
    float4 bl = *reinterpret_cast<const float4*>(base + 2 * HW);
    uint32_t r0 = to_u8(r.x), g0 = to_u8(g.x), b0 = to_u8(bl.x);
    uint32_t r1 = to_u8(r.y), g1 = to_u8(g.y), b1 = to_u8(bl.y);
    uint32_t r2 = to_u8(r.z), g2 = to_u8(g.z), b2 = to_u8(bl.z);
    uint32_t r3 = to_u8(r.w), g3 = to_u8(g.w), b3 = to_u8(bl.w);
    uint32_t w0 = r0 | (g0 << 8) | (b0 << 16) | (r1 << 24);
    uint32_t w1 = g1 | (b1 << 8) | (r2 << 16) | (g2 << 24);
    uint32_t w2 = b2 | (r3 << 8) | (g3 << 16) | (b3 << 24);
    uint32_t* o = reinterpret_cast<uint32_t*>(out + p * 3);
    o[0] = w0; o[1] = w1; o[2] = w2;
  }
}

__global__ __launch_bounds__(256) void pack_rgb8_scalar_kernel(const float* __restrict__ img, uint8_t* __restrict__ out,
                                                               long npix, long HW) {
  long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  long b = p / HW, q = p - b * HW;
  const float* base = img + b * 3 * HW + q;
  out[p * 3 + 0] = (uint8_t)to_u8(base[0]);
  out[p * 3 + 1] = (uint8_t)to_u8(base[HW]);
  out[p * 3 + 2] = (uint8_t)to_u8(base[2 * HW]);
}

int launch_pack_rgb8(hipStream_t stream, const float* img, uint8_t* out, int B, int H, int W) {
  long HW = (long)H * W, npix = (long)B * HW;
  if (npix == 0) return MAUA_OK;
  if (HW % 4 == 0 && (uintptr_t)img % 16 == 0 && (uintptr_t)out % 4 == 0) {
    long n4 = npix / 4;
    int grid = (int)std::min<long>((n4 + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(pack_rgb8_kernel, dim3(grid), dim3(256), 0, stream, img, out, n4, HW);
  } else {
    hipLaunchKernelGGL(pack_rgb8_scalar_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, stream, img, out,
                       npix, HW);
  }
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// ops/io.py:47-70 tensor2bytes for any value range and channel count, in the reference's operation order: clamp(mn, mx) - mn,
// / (mx - mn), * 255, round half to even -> u8, NCHW -> HWC
__global__ __launch_bounds__(256) void tensor2bytes_kernel(const float* __restrict__ img, uint8_t* __restrict__ out, long total,
                                                           int Cc, long HW, float mn, float mx, float width) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // output element (b, pixel, channel)
  if (i >= total) return;
  const int c = (int)(i % Cc);
  const long bp = i / Cc, b = bp / HW, q = bp - b * HW;
  float v = img[(b * Cc + c) * HW + q];
  v = fminf(fmaxf(v, mn), mx);
  v = __fdiv_rn(__fsub_rn(v, mn), width);
  out[i] = (uint8_t)__float2int_rn(__fmul_rn(v, 255.0f));
}

int launch_tensor2bytes(hipStream_t stream, const float* img, uint8_t* out, int B, int Cc, int H, int W, double mn, double mx) {
  const long total = (long)B * Cc * H * W;
  if (total == 0) return MAUA_OK;
  hipLaunchKernelGGL(tensor2bytes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, img, out, total, Cc,
                     (long)H * W, (float)mn, (float)mx, (float)(mx - mn));   // the width is formed in double, like the Python scalar
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// ------------------------------------------------------------------------------------------------ layout converters
// NCHW f32/bf16 <-> NHWC T with channel padding (operator-level API plumbing; LDS-transposed 32x32 tiles so both
// sides are coalesced).
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const TI* __restrict__ x, TO* __restrict__ y, int C, int HW,
                                                           int Cp) {
  __shared__ float t[32][33];
  const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    int c = c0 + r, p = p0 + tx;
    t[r][tx] = (c < C && p < HW) ? Elem<TI>::load(x + ((long)n * C + c) * HW + p) : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    int p = p0 + r, c = c0 + tx;
    if (p < HW && c < Cp) Elem<TO>::store(y + ((long)n * HW + p) * Cp + c, t[tx][r]);
  }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const TI* __restrict__ x, TO* __restrict__ y, int C, int HW,
                                                           int Cp) {
  __shared__ float t[32][33];
  const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    int p = p0 + r, c = c0 + tx;
    t[r][tx] = (p < HW && c < Cp) ? Elem<TI>::load(x + ((long)n * HW + p) * Cp + c) : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    int c = c0 + r, p = p0 + tx;
    if (c < C && p < HW) Elem<TO>::store(y + ((long)n * C + c) * HW + p, t[tx][r]);
  }
}

template <typename TI, typename TO>
int launch_nchw_to_nhwc(hipStream_t s, const void* x, void* y, int N, int C, int HW, int Cp) {
  dim3 grid(cdiv(HW, 32), cdiv(Cp, 32), N);
  hipLaunchKernelGGL((nchw_to_nhwc_kernel<TI, TO>), grid, dim3(256), 0, s, (const TI*)x, (TO*)y, C, HW, Cp);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}
template <typename TI, typename TO>
int launch_nhwc_to_nchw(hipStream_t s, const void* x, void* y, int N, int C, int HW, int Cp) {
  dim3 grid(cdiv(HW, 32), cdiv(Cp, 32), N);
  hipLaunchKernelGGL((nhwc_to_nchw_kernel<TI, TO>), grid, dim3(256), 0, s, (const TI*)x, (TO*)y, C, HW, Cp);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}
template int launch_nchw_to_nhwc<float, float>(hipStream_t, const void*, void*, int, int, int, int);
template int launch_nchw_to_nhwc<float, bf16_t>(hipStream_t, const void*, void*, int, int, int, int);
template int launch_nchw_to_nhwc<bf16_t, bf16_t>(hipStream_t, const void*, void*, int, int, int, int);
template int launch_nhwc_to_nchw<float, float>(hipStream_t, const void*, void*, int, int, int, int);
template int launch_nhwc_to_nchw<bf16_t, float>(hipStream_t, const void*, void*, int, int, int, int);
template int launch_nhwc_to_nchw<bf16_t, bf16_t>(hipStream_t, const void*, void*, int, int, int, int);
template int launch_nchw_to_nhwc<float, f16_t>(hipStream_t, const void*, void*, int, int, int, int);
template int launch_nchw_to_nhwc<f16_t, f16_t>(hipStream_t, const void*, void*, int, int, int, int);
template int launch_nhwc_to_nchw<f16_t, float>(hipStream_t, const void*, void*, int, int, int, int);
template int launch_nhwc_to_nchw<f16_t, f16_t>(hipStream_t, const void*, void*, int, int, int, int);

}  // namespace maua

// ================================================================================================ C ABI
extern "C" {

int maua_bias_act(maua_ctx* ctx, const void* x, const float* b, void* y, int N, int C, int H, int W, int dtype, int act,
                  float alpha, float gain, float clamp) {
  MAUA_REQUIRE(ctx, "maua_bias_act: ctx is NULL");
  MAUA_REQUIRE(act >= MAUA_ACT_LINEAR && act <= MAUA_ACT_SWISH, "maua_bias_act: unknown activation");
  MAUA_REQUIRE(N >= 0 && C >= 0 && H >= 0 && W >= 0, "maua_bias_act: negative size");
  if ((long)N * C * H * W == 0) return MAUA_OK;
  MAUA_REQUIRE(x && y, "maua_bias_act: NULL argument");
  if (dtype == MAUA_F32) return maua::launch_bias_act<float>(ctx, x, b, y, N, C, H, W, act, alpha, gain, clamp);
  if (dtype == MAUA_BF16) return maua::launch_bias_act<maua::bf16_t>(ctx, x, b, y, N, C, H, W, act, alpha, gain, clamp);
  if (dtype == MAUA_F16) return maua::launch_bias_act<maua::f16_t>(ctx, x, b, y, N, C, H, W, act, alpha, gain, clamp);
  return maua::fail("maua_bias_act: unsupported dtype");
}

int maua_add(maua_ctx* ctx, const void* a, const void* b, void* out, long n, int dtype) {
  MAUA_REQUIRE(ctx, "maua_add: ctx is NULL");
  MAUA_REQUIRE(n >= 0, "maua_add: negative size");
  if (n == 0) return MAUA_OK;
  MAUA_REQUIRE(a && b && out, "maua_add: NULL argument");
  if (dtype == MAUA_F32) return maua::launch_add<float>(ctx, a, b, out, n);
  if (dtype == MAUA_BF16) return maua::launch_add<maua::bf16_t>(ctx, a, b, out, n);
  if (dtype == MAUA_F16) return maua::launch_add<maua::f16_t>(ctx, a, b, out, n);
  return maua::fail("maua_add: unsupported dtype");
}

int maua_upfirdn2d(maua_ctx* ctx, const void* x, const float* f, int fh, int fw, void* y, int N, int C, int H, int W,
                   int dtype, int up, int down, int px0, int px1, int py0, int py1, float gain) {
  MAUA_REQUIRE(ctx, "maua_upfirdn2d: ctx is NULL");
  if (N * C == 0) return MAUA_OK;
  MAUA_REQUIRE(x && y && f, "maua_upfirdn2d: NULL argument");
  MAUA_REQUIRE(up >= 1 && down >= 1, "maua_upfirdn2d: up/down must be >= 1");
  MAUA_REQUIRE(fh >= 1 && fw >= 1 && fh <= maua::UF_MAXF && fw <= maua::UF_MAXF, "maua_upfirdn2d: filter size 1..32");
  if (N * C == 0) return MAUA_OK;
  if (dtype == MAUA_F32)
    return maua::launch_upfirdn2d<float>(ctx, x, f, fh, fw, y, N, C, H, W, up, down, px0, px1, py0, py1, gain);
  if (dtype == MAUA_BF16)
    return maua::launch_upfirdn2d<maua::bf16_t>(ctx, x, f, fh, fw, y, N, C, H, W, up, down, px0, px1, py0, py1, gain);
  if (dtype == MAUA_F16)
    return maua::launch_upfirdn2d<maua::f16_t>(ctx, x, f, fh, fw, y, N, C, H, W, up, down, px0, px1, py0, py1, gain);
  return maua::fail("maua_upfirdn2d: unsupported dtype");
}

int maua_pack_rgb8(maua_ctx* ctx, const float* img, uint8_t* out_hwc, int B, int H, int W) {
  MAUA_REQUIRE(ctx, "maua_pack_rgb8: ctx is NULL");
  if ((long)B * H * W == 0) return MAUA_OK;
  MAUA_REQUIRE(img && out_hwc, "maua_pack_rgb8: NULL argument");
  return maua::launch_pack_rgb8(ctx->stream, img, out_hwc, B, H, W);
}

int maua_tensor2bytes(maua_ctx* ctx, const float* img, uint8_t* out_hwc, int B, int C, int H, int W, double value_min,
                      double value_max) {
  MAUA_REQUIRE(ctx, "maua_tensor2bytes: ctx is NULL");
  if ((long)B * C * H * W == 0) return MAUA_OK;
  MAUA_REQUIRE(img && out_hwc && C > 0, "maua_tensor2bytes: NULL argument");
  MAUA_REQUIRE(value_max > value_min, "maua_tensor2bytes: empty value range");
  return maua::launch_tensor2bytes(ctx->stream, img, out_hwc, B, C, H, W, value_min, value_max);
}

}  // extern "C"
