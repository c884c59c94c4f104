// Index-addressed noise modules of the selfsupervised patch (reference
// maua/audiovisual/audioreactive/selfsupervised/noise.py): Loop :42-53, Blend :11-24, Multiply :27-39.
// HBM-bound elementwise work with one per-frame reduction (RMS); two launches per call, deterministic
// (fixed-order partial sums, no atomics).
#include "common.h"
#include "internal.h"

namespace maua {

constexpr int NZ_BLOCKS = 32;  // partial-sum blocks per frame

// sin / cos for the moderate arguments of the Loop module (|x| < ~1e4; here |x| < 64): two-constant FMA Cody-Waite
// reduction by pi/2 and the cephes single-precision minimax polynomials on [-pi/4, pi/4] — ~1 ulp like the library
// versions, at less than half their instruction count (no large-argument path, no branches).  The two Loop kernels were
// bound by exactly these instructions (1.55 TB/s written vs ~4.5 achievable).
__device__ __forceinline__ void reduce_pio2(float x, float& r, int& q) {
  const float k = rintf(x * 0.63661977236758134f);
  q = (int)k;
  r = fmaf(-k, 1.5707963705062866f, x);      // float(pi/2)
  r = fmaf(-k, -4.3711388286737929e-08f, r); // pi/2 - float(pi/2)
}
__device__ __forceinline__ float poly_sin(float r) {
  const float z = r * r;
  return fmaf(r * z, fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), r);
}
__device__ __forceinline__ float poly_cos(float r) {
  const float z = r * r;
  return fmaf(z * z, fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
              fmaf(-0.5f, z, 1.0f));
}
__device__ __forceinline__ float fast_sin(float x) {
  float r; int q;
  reduce_pio2(x, r, q);
  const float v = (q & 1) ? poly_cos(r) : poly_sin(r);
  return (q & 2) ? -v : v;
}
__device__ __forceinline__ float fast_cos(float x) {
  float r; int q;
  reduce_pio2(x, r, q);
  const float v = (q & 1) ? poly_sin(r) : poly_cos(r);
  return ((q + 1) & 2) ? -v : v;
}

__device__ __forceinline__ float loop_value(float idx, float n0, float n1, float n2, float inv_sigma50) {
  float freqs = fast_cos(idx + n0) * inv_sigma50;   // noise.py:50 (division by sigma/50 as a multiplication)
  return fast_sin(freqs + n1) * n2;                 // noise.py:51
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += sh[i];
  return t;  // valid in thread 0
}

// The two passes of one (layer, frame), shared by the single-module and the batched kernels so that both produce the
// same bits: 16-byte loads/stores when the plane size allows (hw % 4 == 0: every size the generator uses), the
// per-thread accumulation order is fixed by (block, nblk).
static int noise_nparts(int hw) { return std::min(NZ_BLOCKS, cdiv(hw, (hw & 3) ? 256 : 1024)); }
static int noise_wblk(int hw) { return std::min(256, cdiv(hw, (hw & 3) ? 256 : 1024)); }  // blocks of the write pass

// pass 1: this thread's part of sum(value^2) over slice `blk` of `nblk`
__device__ __forceinline__ float loop_sumsq_slice(const float* __restrict__ planes, int hw, float id, float inv_s, int blk,
                                                  int nblk) {
  float acc = 0.f;
  if ((hw & 3) == 0) {
    const float4* p0 = reinterpret_cast<const float4*>(planes);
    const float4* p1 = reinterpret_cast<const float4*>(planes + hw);
    const float4* p2 = reinterpret_cast<const float4*>(planes + 2 * (long)hw);
    for (int v = blk * 256 + threadIdx.x; v < (hw >> 2); v += nblk * 256) {
      const float4 a = p0[v], b = p1[v], c = p2[v];
      const float x = loop_value(id, a.x, b.x, c.x, inv_s), y = loop_value(id, a.y, b.y, c.y, inv_s);
      const float z = loop_value(id, a.z, b.z, c.z, inv_s), w = loop_value(id, a.w, b.w, c.w, inv_s);
      acc += x * x; acc += y * y; acc += z * z; acc += w * w;
    }
  } else {
    for (int p = blk * 256 + threadIdx.x; p < hw; p += nblk * 256) {
      const float v = loop_value(id, planes[p], planes[hw + p], planes[2 * (long)hw + p], inv_s);
      acc += v * v;
    }
  }
  return acc;
}

// pass 2: out = value / (sqrt(mean) + eps)   (noise.py:52; the division as a multiplication by the reciprocal)
__device__ __forceinline__ void loop_write_slice(const float* __restrict__ planes, int hw, float id, float inv_s,
                                                 float inv_denom, int blk, int nblk, float* __restrict__ out) {
  if ((hw & 3) == 0) {
    const float4* p0 = reinterpret_cast<const float4*>(planes);
    const float4* p1 = reinterpret_cast<const float4*>(planes + hw);
    const float4* p2 = reinterpret_cast<const float4*>(planes + 2 * (long)hw);
    float4* o = reinterpret_cast<float4*>(out);
    for (int v = blk * 256 + threadIdx.x; v < (hw >> 2); v += nblk * 256) {
      const float4 a = p0[v], b = p1[v], c = p2[v];
      o[v] = make_float4(loop_value(id, a.x, b.x, c.x, inv_s) * inv_denom, loop_value(id, a.y, b.y, c.y, inv_s) * inv_denom,
                         loop_value(id, a.z, b.z, c.z, inv_s) * inv_denom, loop_value(id, a.w, b.w, c.w, inv_s) * inv_denom);
    }
  } else {
    for (int p = blk * 256 + threadIdx.x; p < hw; p += nblk * 256)
      out[p] = loop_value(id, planes[p], planes[hw + p], planes[2 * (long)hw + p], inv_s) * inv_denom;
  }
}

__device__ __forceinline__ float loop_inv_denom(const float* __restrict__ partial, int nparts, int hw) {
  float tot = 0.f;
  for (int i = 0; i < nparts; i++) tot += partial[i];  // fixed order, identical in every thread
  return 1.f / (sqrtf(tot / (float)hw) + 1.1920928955078125e-07f);  // torch.finfo(float32).eps
}

__global__ __launch_bounds__(256) void noise_loop_sumsq_kernel(const float* __restrict__ planes,
                                                               const float* __restrict__ idx, int i0, int hw,
                                                               float inv_s, float* __restrict__ partial) {
  __shared__ float sh[4];
  const int b = blockIdx.y;
  const float t = block_sum(loop_sumsq_slice(planes, hw, idx[i0 + b], inv_s, blockIdx.x, gridDim.x), sh);
  if (threadIdx.x == 0) partial[b * gridDim.x + blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void noise_loop_write_kernel(const float* __restrict__ planes,
                                                               const float* __restrict__ idx, int i0, int hw,
                                                               float inv_s, const float* __restrict__ partial,
                                                               int nparts, float* __restrict__ out) {
  const int b = blockIdx.y;
  const float inv_denom = loop_inv_denom(partial + b * nparts, nparts, hw);
  loop_write_slice(planes, hw, idx[i0 + b], inv_s, inv_denom, blockIdx.x, gridDim.x, out + (long)b * hw);
}

// All layers of a batch in two launches: the per-layer descriptors travel in the kernel arguments.
constexpr int NZ_MAX_LAYERS = 24;
struct NoiseBatch {
  const float* planes[NZ_MAX_LAYERS];
  const float* idx[NZ_MAX_LAYERS];
  float* out[NZ_MAX_LAYERS];
  int hw[NZ_MAX_LAYERS];
  float inv_s[NZ_MAX_LAYERS];
  int nparts[NZ_MAX_LAYERS];
  int wblk[NZ_MAX_LAYERS];   // blocks of the write pass (the grid is sized for the largest layer)
  int i0, B;
};

__global__ __launch_bounds__(256) void noise_loop_batch_sumsq_kernel(NoiseBatch nb, float* __restrict__ partial) {
  __shared__ float sh[4];
  const int l = blockIdx.z, b = blockIdx.y;
  if ((int)blockIdx.x >= nb.nparts[l]) return;
  const float t = block_sum(
      loop_sumsq_slice(nb.planes[l], nb.hw[l], nb.idx[l][nb.i0 + b], nb.inv_s[l], blockIdx.x, nb.nparts[l]), sh);
  if (threadIdx.x == 0) partial[((long)l * nb.B + b) * NZ_BLOCKS + blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void noise_loop_batch_write_kernel(NoiseBatch nb, const float* __restrict__ partial) {
  const int l = blockIdx.z, b = blockIdx.y;
  if ((int)blockIdx.x >= nb.wblk[l]) return;
  const int hw = nb.hw[l];
  const float inv_denom = loop_inv_denom(partial + ((long)l * nb.B + b) * NZ_BLOCKS, nb.nparts[l], hw);
  loop_write_slice(nb.planes[l], hw, nb.idx[l][nb.i0 + b], nb.inv_s[l], inv_denom, blockIdx.x, nb.wblk[l],
                   nb.out[l] + (long)b * hw);
}

// RAW mode (round 4): ONE pass over every map - the un-normalised values are written and the squares summed per block in the same
// sweep (the two-pass form evaluates sin(cos(.)) twice per value: 0.31 + 0.50 ms per 128 frames, both passes bound by those
// instructions); a tiny second kernel turns the partial sums into the per-(layer, sample) factor 1 / (rms + eps) that the
// consuming convolution epilogues multiply into their noise strength (ConvArgs / HiresArgs / UpfirArgs.noise_scale).
// Same per-thread sweep as loop_write_slice; partial sums in a fixed order: deterministic.
constexpr int NZ_RAW_BLOCKS = 256;

__global__ __launch_bounds__(256) void noise_loop_batch_raw_kernel(NoiseBatch nb, float* __restrict__ partial) {
  __shared__ float sh[4];
  const int l = blockIdx.z, b = blockIdx.y;
  const int nblk = nb.wblk[l];
  if ((int)blockIdx.x >= nblk) return;
  const int hw = nb.hw[l];
  const float* __restrict__ planes = nb.planes[l];
  const float id = nb.idx[l][nb.i0 + b], inv_s = nb.inv_s[l];
  float* __restrict__ out = nb.out[l] + (long)b * hw;
  float acc = 0.f;
  if ((hw & 3) == 0) {
    const float4* p0 = reinterpret_cast<const float4*>(planes);
    const float4* p1 = reinterpret_cast<const float4*>(planes + hw);
    const float4* p2 = reinterpret_cast<const float4*>(planes + 2 * (long)hw);
    float4* o = reinterpret_cast<float4*>(out);
    for (int v = blockIdx.x * 256 + threadIdx.x; v < (hw >> 2); v += nblk * 256) {
      const float4 a = p0[v], bb = p1[v], c = p2[v];
      const float x = loop_value(id, a.x, bb.x, c.x, inv_s), y = loop_value(id, a.y, bb.y, c.y, inv_s);
      const float z = loop_value(id, a.z, bb.z, c.z, inv_s), w = loop_value(id, a.w, bb.w, c.w, inv_s);
      o[v] = make_float4(x, y, z, w);
      acc += x * x; acc += y * y; acc += z * z; acc += w * w;
    }
  } else {
    for (int p = blockIdx.x * 256 + threadIdx.x; p < hw; p += nblk * 256) {
      const float v = loop_value(id, planes[p], planes[hw + p], planes[2 * (long)hw + p], inv_s);
      out[p] = v;
      acc += v * v;
    }
  }
  const float t = block_sum(acc, sh);
  if (threadIdx.x == 0) partial[((long)l * nb.B + b) * NZ_RAW_BLOCKS + blockIdx.x] = t;
}

// scales[l * B + b] = 1 / (sqrt(mean v^2) + eps) from the blocks' partial sums (fixed order)
__global__ __launch_bounds__(256) void noise_loop_batch_scale_kernel(NoiseBatch nb, int n, const float* __restrict__ partial,
                                                                     float* __restrict__ scales) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * nb.B) return;
  const int l = i / nb.B;
  scales[i] = loop_inv_denom(partial + (long)i * NZ_RAW_BLOCKS, nb.wblk[l], nb.hw[l]);
}

__global__ __launch_bounds__(256) void noise_mix_kernel(const float* __restrict__ noise, const float* __restrict__ noise2,
                                                        const float* __restrict__ mod, int M, int hw,
                                                        float* __restrict__ out) {
  extern __shared__ float ms[];
  const int b = blockIdx.y;
  for (int m = threadIdx.x; m < M; m += blockDim.x) ms[m] = mod[(long)b * M + m];
  __syncthreads();
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += gridDim.x * blockDim.x) {
    float l = 0.f, r = 0.f;
    for (int m = 0; m < M; m++) {
      l += noise[(long)m * hw + p] * ms[m];
      if (noise2) r += noise2[(long)m * hw + p] * (1.f - ms[m]);
    }
    out[(long)b * hw + p] = noise2 ? l + r : l;
  }
}

// Average (mode 0), Modulate (mode 1), ScaleBias (mode 2)   (noise.py:56-86)
__global__ __launch_bounds__(256) void noise_combine_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ mod, int mode, float scale,
                                                            float bias, long hw, float* __restrict__ out) {
  const int b = blockIdx.y;
  const float m = (mode == 1) ? mod[b] : 0.f;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += (long)gridDim.x * blockDim.x) {
    long o = (long)b * hw + p;
    float v;
    if (mode == 0) v = (x[o] + y[o]) / 2.f;
    else if (mode == 1) v = x[o] * m + y[o] * (1.f - m);
    else v = scale * x[o] + bias;
    out[o] = v;
  }
}

}  // namespace maua

using namespace maua;

extern "C" {

int maua_noise_loop(maua_ctx* ctx, const float* planes, const float* idx, int i0, int B, int h, int w, float sigma,
                    float* out) {
  MAUA_REQUIRE(ctx, "maua_noise_loop: ctx is NULL");
  if (B == 0 || h * w == 0) return MAUA_OK;
  MAUA_REQUIRE(planes && idx && out, "maua_noise_loop: NULL argument");
  MAUA_REQUIRE(sigma != 0.f, "maua_noise_loop: sigma must be non-zero");
  const int hw = h * w;
  const int nblk = noise_nparts(hw);
  if (int rc = scratch_reserve(ctx, (size_t)B * nblk * sizeof(float))) return rc;
  float* partial = (float*)ctx->scratch;
  const float inv_s = (float)(50.0 / (double)sigma);
  hipLaunchKernelGGL(noise_loop_sumsq_kernel, dim3(nblk, B), dim3(256), 0, ctx->stream, planes, idx, i0, hw, inv_s,
                     partial);
  MAUA_HIP_CHECK(hipGetLastError());
  const int wblk = noise_wblk(hw);
  hipLaunchKernelGGL(noise_loop_write_kernel, dim3(wblk, B), dim3(256), 0, ctx->stream, planes, idx, i0, hw, inv_s,
                     partial, nblk, out);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_noise_loop_batch(maua_ctx* ctx, int n, const float* const* planes, const float* const* idx, const int* h,
                          const int* w, const float* sigma, int i0, int B, float* const* out) {
  MAUA_REQUIRE(ctx, "maua_noise_loop_batch: ctx is NULL");
  if (B == 0 || n == 0) return MAUA_OK;
  MAUA_REQUIRE(planes && idx && h && w && sigma && out, "maua_noise_loop_batch: NULL argument");
  MAUA_REQUIRE(n <= NZ_MAX_LAYERS, "maua_noise_loop_batch: at most 24 layers per call");
  NoiseBatch nb{};
  int max_wblk = 1;
  for (int l = 0; l < n; l++) {
    MAUA_REQUIRE(planes[l] && idx[l] && out[l] && sigma[l] != 0.f, "maua_noise_loop_batch: NULL layer argument");
    nb.planes[l] = planes[l]; nb.idx[l] = idx[l]; nb.out[l] = out[l];
    nb.hw[l] = h[l] * w[l];
    nb.inv_s[l] = (float)(50.0 / (double)sigma[l]);
    nb.nparts[l] = noise_nparts(nb.hw[l]);
    nb.wblk[l] = noise_wblk(nb.hw[l]);
    max_wblk = std::max(max_wblk, nb.wblk[l]);
  }
  nb.i0 = i0; nb.B = B;
  if (int rc = scratch_reserve(ctx, (size_t)n * B * NZ_BLOCKS * sizeof(float))) return rc;
  float* partial = (float*)ctx->scratch;
  hipLaunchKernelGGL(noise_loop_batch_sumsq_kernel, dim3(NZ_BLOCKS, B, n), dim3(256), 0, ctx->stream, nb, partial);
  hipLaunchKernelGGL(noise_loop_batch_write_kernel, dim3(max_wblk, B, n), dim3(256), 0,
                     ctx->stream, nb, partial);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_noise_loop_batch_raw(maua_ctx* ctx, int n, const float* const* planes, const float* const* idx, const int* h,
                              const int* w, const float* sigma, int i0, int B, float* const* out, float* scales) {
  MAUA_REQUIRE(ctx, "maua_noise_loop_batch_raw: ctx is NULL");
  if (B == 0 || n == 0) return MAUA_OK;
  MAUA_REQUIRE(planes && idx && h && w && sigma && out && scales, "maua_noise_loop_batch_raw: NULL argument");
  MAUA_REQUIRE(n <= NZ_MAX_LAYERS, "maua_noise_loop_batch_raw: at most 24 layers per call");
  NoiseBatch nb{};
  int max_wblk = 1;
  for (int l = 0; l < n; l++) {
    MAUA_REQUIRE(planes[l] && idx[l] && out[l] && sigma[l] != 0.f, "maua_noise_loop_batch_raw: NULL layer argument");
    nb.planes[l] = planes[l]; nb.idx[l] = idx[l]; nb.out[l] = out[l];
    nb.hw[l] = h[l] * w[l];
    nb.inv_s[l] = (float)(50.0 / (double)sigma[l]);
    nb.wblk[l] = noise_wblk(nb.hw[l]);
    max_wblk = std::max(max_wblk, nb.wblk[l]);
  }
  static_assert(NZ_RAW_BLOCKS >= 256, "noise_wblk() returns at most 256 blocks");
  nb.i0 = i0; nb.B = B;
  if (int rc = scratch_reserve(ctx, (size_t)n * B * NZ_RAW_BLOCKS * sizeof(float))) return rc;
  float* partial = (float*)ctx->scratch;
  hipLaunchKernelGGL(noise_loop_batch_raw_kernel, dim3(max_wblk, B, n), dim3(256), 0, ctx->stream, nb, partial);
  hipLaunchKernelGGL(noise_loop_batch_scale_kernel, dim3((unsigned)((n * B + 255) / 256)), dim3(256), 0, ctx->stream, nb, n,
                     partial, scales);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_noise_mix(maua_ctx* ctx, const float* noise, const float* noise2, const float* mod, int M, int B, int h,
                   int w, float* out) {
  MAUA_REQUIRE(ctx, "maua_noise_mix: ctx is NULL");
  if (B == 0 || h * w == 0) return MAUA_OK;
  MAUA_REQUIRE(noise && mod && out && M > 0, "maua_noise_mix: NULL argument");
  const int hw = h * w;
  hipLaunchKernelGGL(noise_mix_kernel, dim3(std::min(256, cdiv(hw, 256)), B), dim3(256), (size_t)M * sizeof(float),
                     ctx->stream, noise, noise2, mod, M, hw, out);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_noise_combine(maua_ctx* ctx, const float* x, const float* y, const float* mod, int mode, float scale,
                       float bias, int B, int h, int w, float* out) {
  MAUA_REQUIRE(ctx, "maua_noise_combine: ctx is NULL");
  if (B == 0 || h * w == 0) return MAUA_OK;
  MAUA_REQUIRE(x && out, "maua_noise_combine: NULL argument");
  MAUA_REQUIRE(mode >= 0 && mode <= 2, "maua_noise_combine: mode must be 0 (average), 1 (modulate) or 2 (scale+bias)");
  MAUA_REQUIRE(mode == 2 || y, "maua_noise_combine: y is NULL");
  MAUA_REQUIRE(mode != 1 || mod, "maua_noise_combine: mod is NULL");
  const long hw = (long)h * w;
  hipLaunchKernelGGL(noise_combine_kernel, dim3((unsigned)std::min<long>(256, (hw + 255) / 256), B), dim3(256), 0,
                     ctx->stream, x, y, mod, mode, scale, bias, hw, out);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // extern "C"
