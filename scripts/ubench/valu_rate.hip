// micro-benchmark: VALU issue rate per wave on gfx950 with 1 or 2 waves per SIMD, alone or beside an MFMA wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int mode, int iters) {
  const int wave = threadIdx.x >> 6;
  float v[32];
  for (int i = 0; i < 32; i++) v[i] = threadIdx.x * 0.001f + i;
  f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0};
  bf16x8 a = {0}, b = {0};
  const bool mf = (mode == 2 && wave >= 4) || mode == 3;   // mode 2: waves 4-7 run MFMAs beside VALU waves 0-3; 3: all MFMA
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  if (mf) {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int j = 0; j < 12; j++) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
      }
    }
  } else {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int j = 0; j < 10; j++)
#pragma unroll
        for (int i = 0; i < 32; i++) v[i] = fmaf(v[i], 1.0001f, 0.5f);   // 320 independent-ish VALU per iteration (chains of depth 10 x 32 wide)
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 32; i++) s += v[i];
  for (int i = 0; i < 16; i++) s += acc0[i] + acc1[i] + acc2[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  const int iters = 200;
  for (int threads : {256, 512})
    for (int mode : {1, 2, 3}) {
      if (threads == 256 && mode == 2) continue;
      hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, cyc, mode, iters);
      hipDeviceSynchronize();
      long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
      printf("threads %d mode %d: wave0 %.1f cycles/iter (%s), wave%d %.1f cycles/iter\n", threads, mode, (double)h[0] / iters,
             mode == 3 ? "36 MFMA" : "320 VALU", threads / 64 - 1, (double)h[threads / 64 - 1] / iters);
    }
  return 0;
}
