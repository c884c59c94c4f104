// micro-benchmark: what one wave's instruction stream costs its SIMD partner (gfx950, 2 waves per SIMD).
// wave A (waves 0-3) runs workload X, wave B (waves 4-7) workload Y; cycles per iteration of each.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
enum { IDLE = 0, VALU = 1, MFMA = 2, LDSR = 3, MIX = 4, VMED = 5, VCVT = 6, VFMA1 = 7 };
__device__ __forceinline__ void work(int kind, int iters, float* v, f32x16& a0, f32x16& a1, f32x16& a2, const u32x4* lds, u32x4& sink) {
  bf16x8 a = {0}, b = {0};
  for (int it = 0; it < iters; it++) {
    if (kind == VALU) {
#pragma unroll
      for (int j = 0; j < 10; j++)
#pragma unroll
        for (int i = 0; i < 32; i++) v[i] = fmaf(v[i], 1.0001f, 0.5f);
    } else if (kind == VMED) {   // 320 unpaired VALU instructions (v_med3_f32: no packed form exists)
#pragma unroll
      for (int j = 0; j < 10; j++)
#pragma unroll
        for (int i = 0; i < 32; i++) v[i] = __builtin_amdgcn_fmed3f(v[i], v[(i + 1) & 31], 0.25f);
    } else if (kind == VCVT) {   // 320 x v_cvt_pk_bf16_f32 (two values -> one register)
#pragma unroll
      for (int j = 0; j < 10; j++)
#pragma unroll
        for (int i = 0; i < 32; i++) {
          typedef __bf16 hb2 __attribute__((ext_vector_type(2)));
          typedef float f2 __attribute__((ext_vector_type(2)));
          f2 t = {v[i], v[(i + 1) & 31]};
          hb2 c = __builtin_convertvector(t, hb2);
          v[i] = __uint_as_float(__builtin_bit_cast(unsigned, c) | 0x3f000000u);
        }
    } else if (kind == VFMA1) {  // 320 unpaired v_fma_f32 (a different multiplier per register defeats the pairing)
#pragma unroll
      for (int j = 0; j < 10; j++)
#pragma unroll
        for (int i = 0; i < 32; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 7) & 31]), "v"(v[(i + 13) & 31]));
    } else if (kind == MFMA) {
#pragma unroll
      for (int j = 0; j < 12; j++) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, a2, 0, 0, 0);
      }
    } else if (kind == LDSR) {   // 48 ds_read_b128, conflict-free
#pragma unroll
      for (int j = 0; j < 48; j++) { u32x4 t = lds[j * 64]; sink[0] ^= t[0]; sink[1] ^= t[1]; sink[2] ^= t[2]; sink[3] ^= t[3]; }
    } else if (kind == MIX) {    // 36 MFMA with 48 ds_read_b128 feeding them (the producer's multiply phase)
#pragma unroll
      for (int j = 0; j < 12; j++) {
        u32x4 t0 = lds[(4 * j) * 64], t1 = lds[(4 * j + 1) * 64], t2 = lds[(4 * j + 2) * 64], t3 = lds[(4 * j + 3) * 64];
        bf16x8 bb = __builtin_bit_cast(bf16x8, t0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, t1), bb, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, t2), bb, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, t3), bb, a2, 0, 0, 0);
      }
    }
  }
}
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int kindA, int kindB, int iters) {
  __shared__ u32x4 lds[48 * 64 + 64];
  for (int i = threadIdx.x; i < 48 * 64 + 64; i += 512) lds[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float v[32];
  for (int i = 0; i < 32; i++) v[i] = threadIdx.x * 0.001f + i;
  f32x16 a0 = {0}, a1 = {0}, a2 = {0};
  u32x4 sink = {0, 0, 0, 0};
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  work(wave < 4 ? kindA : kindB, iters, v, a0, a1, a2, lds + lane, sink);
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = sink[0] + sink[1] + sink[2] + sink[3];
  for (int i = 0; i < 32; i++) s += v[i];
  for (int i = 0; i < 16; i++) s += a0[i] + a1[i] + a2[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  const int iters = 200;
  const char* nm[] = {"idle", "320 VALU", "36 MFMA", "48 ds_read_b128", "36 MFMA + 48 ds_read", "320 v_med3", "320 v_cvt_pk_bf16 (+or)", "320 v_fma_f32 unpaired"};
  int pairs[][2] = {{VALU, IDLE}, {MFMA, IDLE}, {LDSR, IDLE}, {MIX, IDLE}, {VALU, VALU}, {VALU, MFMA}, {MFMA, VALU}, {VALU, LDSR}, {LDSR, VALU},
                    {VALU, MIX}, {MIX, VALU}, {MFMA, LDSR}, {MIX, MFMA}, {MIX, MIX}, {LDSR, LDSR},
                    {VMED, IDLE}, {VMED, MFMA}, {MFMA, VMED}, {VMED, MIX}, {VMED, VMED}, {VCVT, IDLE}, {VCVT, MFMA}, {VFMA1, IDLE}, {VFMA1, MFMA}, {MFMA, VFMA1}, {VFMA1, MIX}};
  for (auto& p : pairs) {
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, cyc, p[0], p[1], iters);
    hipDeviceSynchronize();
    long long h[8]; (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("A(older) = %-26s B = %-26s : A %.0f  B %.0f cycles/iter\n", nm[p[0]], nm[p[1]], (double)h[0] / iters, (double)h[4] / iters);
  }
  return 0;
}
