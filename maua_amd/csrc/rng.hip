// Build-owned counter RNG ("a build-owned counter RNG ... identical on host / device / all ranks"): Philox4x32-10
// (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11 - the published algorithm; the oracle twin
// oracle/rng.py is pinned to the paper's known-answer vectors).  No reference counterpart: the reference draws its random-init
// weights and noise planes from torch's host generator (inference/stylegan2.py:216-227, selfsupervised/noise.py:42-53); the
// benchmark's synthetic network and planes come from here instead, so that a clip's set-up costs kernels, not 32 M host draws and
// their upload, and every rank / device / the CPU baseline hold the same numbers without exchanging them.
//   element i of (seed, stream):  counter = {lo(i / 4), hi(i / 4), lo(stream), hi(stream)}, key = {lo(seed), hi(seed)}, word i % 4
//   normal:  words (x0, x1) and (x2, x3) of a counter -> Box-Muller pairs; u = ((x >> 9) + 0.5) 2^-23 (exact in f32, never 0 or 1)
#include "common.h"
#include "internal.h"

namespace maua {

namespace {

struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

__device__ __forceinline__ float unit23(uint32_t x) { return ((float)(x >> 9) + 0.5f) * 1.1920928955078125e-07f; }

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
  const float r = sqrtf(-2.0f * logf(unit23(a)));
  const float th = 6.283185307179586f * unit23(b);
  z0 = r * cosf(th);
  z1 = r * sinf(th);
}

// one thread per counter (4 outputs); offset: first element of the stream this call produces (any alignment)
__global__ __launch_bounds__(256) void philox_kernel(uint32_t s0, uint32_t s1, uint32_t t0, uint32_t t1, unsigned long long offset,
                                                     long n, uint32_t* __restrict__ out_u32, float* __restrict__ out_f, float mean,
                                                     float stdev) {
  const unsigned long long first = offset >> 2;
  const unsigned long long c = first + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long base = (long)((c << 2) - offset);   // index in `out` of this counter's word 0 (may be negative for the first counter)
  if (base >= n) return;
  const U4 v = philox4x32_10((uint32_t)c, (uint32_t)(c >> 32), t0, t1, s0, s1);
  if (out_u32) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (base + k >= 0 && base + k < n) out_u32[base + k] = w[k];
  } else {
    float z[4];
    box_muller(v.x, v.y, z[0], z[1]);
    box_muller(v.z, v.w, z[2], z[3]);
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (base + k >= 0 && base + k < n) out_f[base + k] = fmaf(z[k], stdev, mean);
  }
}

// SURVEY 8(d)'s synthetic clip, four samples per thread (counter c of both streams):
//   y[i] = 0.3 sin(2 pi 220 t) + 0.2 (u - 0.5) [(2 t mod 1) < 0.05] + 0.01 n,   t = i / sr (float64 phase)
//   u = unit23 of stream 0's word i, n = stream 1's normal i of `seed`
__global__ __launch_bounds__(256) void clip_audio_kernel(uint32_t s0, uint32_t s1, long n, double sr, float* __restrict__ out) {
  const unsigned long long c = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long base = (long)(c << 2);
  if (base >= n) return;
  const U4 uw = philox4x32_10((uint32_t)c, (uint32_t)(c >> 32), 0u, 0u, s0, s1);
  const U4 nw = philox4x32_10((uint32_t)c, (uint32_t)(c >> 32), 1u, 0u, s0, s1);
  float z[4];
  box_muller(nw.x, nw.y, z[0], z[1]);
  box_muller(nw.z, nw.w, z[2], z[3]);
  const uint32_t w[4] = {uw.x, uw.y, uw.z, uw.w};
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (base + k >= n) break;
    const double t = (double)(base + k) / sr;
    const double ph = 2.0 * t;
    const bool click = ph - floor(ph) < 0.05;
    const float tone = (float)(0.3 * sin(6.283185307179586 * 220.0 * t));
    out[base + k] = tone + (click ? 0.2f * (unit23(w[k]) - 0.5f) : 0.f) + 0.01f * z[k];
  }
}

int launch(maua_ctx* ctx, unsigned long long seed, unsigned long long stream, unsigned long long offset, long n, uint32_t* u, float* f,
           float mean, float stdev) {
  if (n == 0) return MAUA_OK;
  const unsigned long long counters = ((offset + (unsigned long long)n + 3) >> 2) - (offset >> 2);
  MAUA_REQUIRE(counters < (1ull << 39), "philox: too many elements in one call");
  hipLaunchKernelGGL(philox_kernel, dim3((unsigned)((counters + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t)seed,
                     (uint32_t)(seed >> 32), (uint32_t)stream, (uint32_t)(stream >> 32), offset, n, u, f, mean, stdev);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace

}  // namespace maua

using namespace maua;

extern "C" {

int maua_philox_u32(maua_ctx* ctx, unsigned long long seed, unsigned long long stream, unsigned long long offset, uint32_t* out, long n) {
  MAUA_REQUIRE(ctx, "maua_philox_u32: ctx is NULL");
  MAUA_REQUIRE(n >= 0 && (n == 0 || out), "maua_philox_u32: bad argument");
  return launch(ctx, seed, stream, offset, n, out, nullptr, 0.f, 1.f);
}

int maua_philox_normal(maua_ctx* ctx, unsigned long long seed, unsigned long long stream, unsigned long long offset, float* out, long n,
                       float mean, float stdev) {
  MAUA_REQUIRE(ctx, "maua_philox_normal: ctx is NULL");
  MAUA_REQUIRE(n >= 0 && (n == 0 || out), "maua_philox_normal: bad argument");
  return launch(ctx, seed, stream, offset, n, nullptr, out, mean, stdev);
}

// The benchmark's synthetic waveform ("220 Hz tone + 2 Hz click train + noise from a build-owned counter RNG") drawn on
// the device: out [n] f32 at sample rate sr; streams 0 (click amplitudes) and 1 (noise floor) of `seed`.  oracle/rng.py clip_audio is
// its host twin.
int maua_philox_clip_audio(maua_ctx* ctx, unsigned long long seed, long n, double sample_rate, float* out) {
  MAUA_REQUIRE(ctx, "maua_philox_clip_audio: ctx is NULL");
  MAUA_REQUIRE(n >= 0 && (n == 0 || out) && sample_rate > 0, "maua_philox_clip_audio: bad argument");
  if (n == 0) return MAUA_OK;
  const long counters = (n + 3) >> 2;
  hipLaunchKernelGGL(clip_audio_kernel, dim3((unsigned)((counters + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t)seed,
                     (uint32_t)(seed >> 32), n, sample_rate, out);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // extern "C"
