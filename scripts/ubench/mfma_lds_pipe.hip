// micro-benchmark: an LDS-fed MFMA stream shaped like the fused walk's producer phase - 12 iterations of
// (R ds_read_b128 requested PD - 1 iterations ahead, 3 MFMAs on independent accumulators) - cycles per 36 MFMAs,
// for R = 0..4 reads per iteration, PD = 2..4 register slots, 1 or 2 waves per SIMD, with / without scheduling barriers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int R, int PD, bool PIN>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  __shared__ u32x4 lds[48 * 64 + 64];
  for (int i = threadIdx.x; i < 48 * 64 + 64; i += blockDim.x) lds[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u32x4* p = lds + lane;
  f32x16 a0 = {0}, a1 = {0}, a2 = {0};
  u32x4 s0[PD], s1[PD], s2[PD], s3[PD];
  const u32x4 kc = {1u, 2u, 3u, 4u};
#pragma unroll
  for (int j = 0; j < PD; j++) { s0[j] = kc; s1[j] = kc; s2[j] = kc; s3[j] = kc; }
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
#define LOAD(IT)                                                   \
  {                                                                \
    if (R > 0) s0[(IT) % PD] = p[(4 * (IT)) * 64];                 \
    if (R > 1) s1[(IT) % PD] = p[(4 * (IT) + 1) * 64];             \
    if (R > 2) s2[(IT) % PD] = p[(4 * (IT) + 2) * 64];             \
    if (R > 3) s3[(IT) % PD] = p[(4 * (IT) + 3) * 64];             \
  }
  for (int it0 = 0; it0 < iters; it0++) {
#pragma unroll
    for (int it = 0; it < PD - 1; it++) LOAD(it)
#pragma unroll
    for (int it = 0; it < 12; it++) {
      if (it + PD - 1 < 12) LOAD(it + PD - 1)
      if (PIN) __builtin_amdgcn_sched_barrier(0);
      const bf16x8 b = __builtin_bit_cast(bf16x8, s0[it % PD]);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, s1[it % PD]), b, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, s2[it % PD]), b, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, s3[it % PD]), b, a2, 0, 0, 0);
      if (PIN) __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; i++) s += a0[i] + a1[i] + a2[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int R, int PD, bool PIN>
void run(float* out, long long* cyc) {
  const int iters = 200;
  for (int threads : {256, 512}) {
    hipLaunchKernelGGL((k<R, PD, PIN>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[8]; (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("reads/iter %d  slots %d  pinned %d  %d waves/SIMD: %6.0f cycles per 36 MFMAs\n", R, PD, (int)PIN, threads / 256, (double)h[0] / iters);
  }
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  run<0, 3, true>(out, cyc);
  run<1, 3, true>(out, cyc);
  run<2, 3, true>(out, cyc);
  run<4, 3, true>(out, cyc);
  run<4, 2, true>(out, cyc);
  run<4, 4, true>(out, cyc);
  run<4, 3, false>(out, cyc);
  run<2, 3, false>(out, cyc);
  return 0;
}
