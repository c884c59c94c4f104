// micro-benchmark for the fused walk's structure question: the work of one step of one SIMD - 76 MFMAs
// (32x32x16 bf16), 46 operand fragments read from LDS, ~420 VALU instructions - issued either
//   D : as today, by TWO 256-register waves per SIMD (producer: 48 reads incl. 36 weight fragments + 36 MFMAs, then 200 VALU;
//       consumer: 34 reads + 40 MFMAs, then 220 VALU; one workgroup barrier per step), or
//   S : by ONE 512-register wave per SIMD with every weight fragment register-resident (54 fragments = 216 registers) and
//       the three streams interleaved in program order (1 MFMA, ~5.5 VALU, 0.6 ds_read per group, nothing may cross a group).
// Prints cycles per step (s_memtime) and microseconds per step (events) for each, plus the single-stream floors.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0)
constexpr int LDS_ELEMS = 48 * 64 + 64;

// ---- S: one wave per SIMD.  MODE bit 0 = MFMAs, bit 1 = VALU, bit 2 = LDS reads
template <int MODE, bool PIN, bool PLAIN = false>
__global__ __launch_bounds__(256) void ks(float* out, long long* cyc, int iters) {
  extern __shared__ u32x4 lds[];
  for (int i = threadIdx.x; i < LDS_ELEMS; i += blockDim.x) lds[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u32x4* p = lds + lane;
  u32x4 w[54];
#pragma unroll
  for (int i = 0; i < 54; i++) { w[i] = p[(i % 48) * 64]; w[i][0] += i; }
  f32x16 acc[6];
#pragma unroll
  for (int i = 0; i < 6; i++) acc[i] = f32x16{0};
  float v[64];
#pragma unroll
  for (int i = 0; i < 64; i++) v[i] = threadIdx.x * 0.001f + i;
  u32x4 s[8];
#pragma unroll
  for (int i = 0; i < 8; i++) s[i] = u32x4{1u, 2u, 3u, 4u};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    int vk = 0;
#pragma unroll
    for (int j = 0; j < 76; j++) {
      const int rd = (j * 46) / 76, rd_next = ((j + 1) * 46) / 76;
      if ((MODE & 4) && rd_next != rd) s[rd_next % 8] = p[(rd_next % 48) * 64];
      if (MODE & 1) acc[j % 6] = MFMA(s[(rd + 5) % 8], w[j % 54], acc[j % 6]);   // fragment requested >= 3 groups ago
      if (MODE & 2) {
        const int nv = (j & 1) ? 6 : 5;      // 418 per step
#pragma unroll
        for (int q = 0; q < nv; q++) {
          if (PLAIN) v[vk % 64] = __builtin_amdgcn_fmed3f(v[vk % 64], v[(vk + 1) % 64], 0.25f);
          else v[vk % 64] = fmaf(v[vk % 64], 1.0001f, 0.5f);
          vk++;
        }
      }
      if (PIN) __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
  for (int i = 0; i < 64; i++) r += v[i];
  for (int i = 0; i < 6; i++)
    for (int e = 0; e < 16; e++) r += acc[i][e];
  for (int i = 0; i < 8; i++) r += s[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// ---- D: two waves per SIMD, phases in sequence inside each wave, one barrier per step.  WREG: the producer's 36 weight
// fragments in registers instead of LDS (what 256 registers do not allow in the real kernel)
template <bool WREG, bool BARRIER>
__global__ __launch_bounds__(512) void kd(float* out, long long* cyc, int iters) {
  extern __shared__ u32x4 lds[];
  for (int i = threadIdx.x; i < LDS_ELEMS; i += blockDim.x) lds[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u32x4* p = lds + lane;
  const bool producer = wave < 4;
  u32x4 w[18];
#pragma unroll
  for (int i = 0; i < 18; i++) { w[i] = p[i * 64]; w[i][0] += i; }
  f32x16 a0 = {0}, a1 = {0}, a2 = {0};
  float v[32];
#pragma unroll
  for (int i = 0; i < 32; i++) v[i] = threadIdx.x * 0.001f + i;
  u32x4 sink = {0, 0, 0, 0};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    if (producer) {
#pragma unroll
      for (int j = 0; j < 12; j++) {   // 12 x fragments, each feeding three chains whose weight fragments come from LDS
        u32x4 x = p[(4 * j) * 64];
        u32x4 b0 = WREG ? w[j] : p[(4 * j + 1) * 64], b1 = WREG ? w[(j + 3) % 18] : p[(4 * j + 2) * 64], b2 = WREG ? w[(j + 6) % 18] : p[(4 * j + 3) * 64];
        a0 = MFMA(x, b0, a0); a1 = MFMA(x, b1, a1); a2 = MFMA(x, b2, a2);
      }
#pragma unroll
      for (int q = 0; q < 200; q++) v[q % 32] = fmaf(v[q % 32], 1.0001f, 0.5f);
    } else {
#pragma unroll
      for (int j = 0; j < 34; j++) {   // ring fragments; weights in registers
        u32x4 x = p[(j % 48) * 64];
        if (j % 3 == 0) a0 = MFMA(x, w[j % 18], a0); else if (j % 3 == 1) a1 = MFMA(x, w[j % 18], a1); else a2 = MFMA(x, w[j % 18], a2);
        if (j < 6) a2 = MFMA(x, w[(j + 9) % 18], a2);
      }
#pragma unroll
      for (int q = 0; q < 220; q++) v[q % 32] = fmaf(v[q % 32], 1.0001f, 0.5f);
    }
    if (BARRIER) __syncthreads();
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float r = sink[0];
  for (int i = 0; i < 32; i++) r += v[i];
  for (int e = 0; e < 16; e++) r += a0[e] + a1[e] + a2[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}


// ---- the same question with the 4-pass 16x16x32 instruction (half the flops per instruction: 152 per step for the same work)
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int MODE>
__global__ __launch_bounds__(256) void ks16(float* out, long long* cyc, int iters) {
  extern __shared__ u32x4 lds[];
  for (int i = threadIdx.x; i < LDS_ELEMS; i += blockDim.x) lds[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u32x4* p = lds + lane;
  u32x4 w[24];
#pragma unroll
  for (int i = 0; i < 24; i++) { w[i] = p[(i % 48) * 64]; w[i][0] += i; }
  f32x4 acc[12];
#pragma unroll
  for (int i = 0; i < 12; i++) acc[i] = f32x4{0};
  float v[64];
#pragma unroll
  for (int i = 0; i < 64; i++) v[i] = threadIdx.x * 0.001f + i;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    int vk = 0;
#pragma unroll
    for (int j = 0; j < 152; j++) {
      if (MODE & 1)
        acc[j % 12] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[j % 24]), __builtin_bit_cast(bf16x8, w[(j + 7) % 24]), acc[j % 12], 0, 0, 0);
      if (MODE & 2) {
        const int nv = (j & 3) == 3 ? 2 : 3;      // 418 per step
#pragma unroll
        for (int q = 0; q < nv; q++) { v[vk % 64] = fmaf(v[vk % 64], 1.0001f, 0.5f); vk++; }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
  for (int i = 0; i < 64; i++) r += v[i];
  for (int i = 0; i < 12; i++)
    for (int e = 0; e < 4; e++) r += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// scalar (unpaired) VALU beside the 8-pass instruction: v_max / v_med3 cannot be packed
template <int MODE>
__global__ __launch_bounds__(256) void ksmax(float* out, long long* cyc, int iters) {
  extern __shared__ u32x4 lds[];
  for (int i = threadIdx.x; i < LDS_ELEMS; i += blockDim.x) lds[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u32x4* p = lds + lane;
  u32x4 w[24];
#pragma unroll
  for (int i = 0; i < 24; i++) { w[i] = p[(i % 48) * 64]; w[i][0] += i; }
  f32x16 acc[6];
#pragma unroll
  for (int i = 0; i < 6; i++) acc[i] = f32x16{0};
  float v[64];
#pragma unroll
  for (int i = 0; i < 64; i++) v[i] = threadIdx.x * 0.001f + i;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    int vk = 0;
#pragma unroll
    for (int j = 0; j < 76; j++) {
      if (MODE & 1) acc[j % 6] = MFMA(w[j % 24], w[(j + 7) % 24], acc[j % 6]);
      if (MODE & 2) {
        const int nv = (j & 1) ? 6 : 5;
#pragma unroll
        for (int q = 0; q < nv; q++) { v[vk % 64] = __builtin_amdgcn_fmed3f(v[vk % 64], v[(vk + 1) % 64], 0.25f); vk++; }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
  for (int i = 0; i < 64; i++) r += v[i];
  for (int i = 0; i < 6; i++)
    for (int e = 0; e < 16; e++) r += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <typename K>
static void run(const char* name, K kern, int threads, float* out, long long* cyc) {
  const int iters = 400;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t shm = 100 * 1024;   // one workgroup per CU
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), shm, 0, out, cyc, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), shm, 0, out, cyc, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  long long h[8]; (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("%-64s %7.0f ticks/step (wave 0)  %7.0f (wave %d)   %7.3f us/step\n", name, (double)h[0] / iters,
         (double)h[threads / 64 - 1] / iters, threads / 64 - 1, ms * 1e3 / iters);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  run("S  one wave/SIMD: MFMA only (76)", ks<1, true>, 256, out, cyc);
  run("S  one wave/SIMD: VALU only (418)", ks<2, true>, 256, out, cyc);
  run("S  one wave/SIMD: LDS reads only (46)", ks<4, true>, 256, out, cyc);
  run("S  one wave/SIMD: MFMA + LDS", ks<5, true>, 256, out, cyc);
  run("S  one wave/SIMD: MFMA + VALU", ks<3, true>, 256, out, cyc);
  run("S  one wave/SIMD: MFMA + VALU + LDS, groups pinned", ks<7, true>, 256, out, cyc);
  run("S  one wave/SIMD: MFMA + VALU + LDS, compiler's order", ks<7, false>, 256, out, cyc);
  run("D  two waves/SIMD, phases in sequence, barrier per step", kd<false, true>, 512, out, cyc);
  run("D  ... without the barrier", kd<false, false>, 512, out, cyc);
  run("D  ... producer weights in registers, barrier", kd<true, true>, 512, out, cyc);
  run("S  one wave/SIMD, UNPAIRED VALU (v_med3): MFMA + VALU", ks<3, true, true>, 256, out, cyc);
  run("S  one wave/SIMD, UNPAIRED VALU (v_med3): MFMA + VALU + LDS, groups pinned", ks<7, true, true>, 256, out, cyc);
  run("16x16x32: MFMA only (152)", ks16<1>, 256, out, cyc);
  run("16x16x32: VALU only (418, paired by the compiler)", ks16<2>, 256, out, cyc);
  run("16x16x32: MFMA + VALU interleaved 1 : 2.75", ks16<3>, 256, out, cyc);
  run("32x32x16 + UNPAIRED VALU (v_med3): MFMA only", ksmax<1>, 256, out, cyc);
  run("32x32x16 + UNPAIRED VALU (v_med3): VALU only (418)", ksmax<2>, 256, out, cyc);
  run("32x32x16 + UNPAIRED VALU (v_med3): both", ksmax<3>, 256, out, cyc);
  return 0;
}
