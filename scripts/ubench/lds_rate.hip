// micro-benchmark: ds_read_b128 throughput of one wave (16 independent reads in flight, then one wait), with 1 / 2 waves
// per SIMD, and the same stream feeding MFMAs with the reads requested `ahead` groups before use.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int mode, int iters) {
  __shared__ u32x4 lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) lds[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u32x4* p = lds + lane;
  u32x4 r[16];
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  if (mode == 0) {   // 16 reads, wait, repeat (48 reads per "iteration")
    for (int it = 0; it < iters * 3; it++) {
#pragma unroll
      for (int j = 0; j < 16; j++) r[j] = p[((it + j) & 63) * 64];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 16; j++) asm volatile("" ::"v"(r[j]));
    }
  } else {           // mode = reads per MFMA group of 4 MFMAs (independent accumulators); software-pipelined by one group
    u32x4 q[4], n[4];
#pragma unroll
    for (int j = 0; j < 4; j++) q[j] = p[j * 64];
    for (int it = 0; it < iters * 9; it++) {   // 36 MFMAs per "iteration"
#pragma unroll
      for (int j = 0; j < 4; j++) if (j < mode) n[j] = p[((it * 4 + j) & 63) * 64];
      __builtin_amdgcn_sched_barrier(0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, q[0]), __builtin_bit_cast(bf16x8, q[1]), a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, q[1]), __builtin_bit_cast(bf16x8, q[2]), a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, q[2]), __builtin_bit_cast(bf16x8, q[3]), a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, q[3]), __builtin_bit_cast(bf16x8, q[0]), a3, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; j++) if (j < mode) q[j] = n[j];
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; i++) s += a0[i] + a1[i] + a2[i] + a3[i] + r[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  const int iters = 200;
  for (int threads : {256, 512})
    for (int mode : {0, 1, 2, 3, 4}) {
      hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, cyc, mode, iters);
      hipDeviceSynchronize();
      long long h[8]; (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
      if (mode == 0) printf("%d waves/SIMD, 48 ds_read_b128 (16 in flight): %.0f cycles = %.1f per read\n", threads / 256, (double)h[0] / iters, (double)h[0] / iters / 48);
      else printf("%d waves/SIMD, 36 MFMA with %d ds_read_b128 per 4 MFMAs (%d reads): %.0f cycles\n", threads / 256, mode, mode * 9, (double)h[0] / iters);
    }
  return 0;
}
