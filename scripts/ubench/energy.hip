// micro-benchmark for the per-instruction ENERGY price list: one instruction class at a time on random
// operands, on all CUs or on half of them, for a few seconds each while scripts/energy_prices.py samples rocm-smi.  The
// program prints the rate it reached (instructions per second over the chip) and the clock that rate implies; the driver
// script turns (power, rate) pairs into joules per instruction: E = (P_all - P_half) / (rate_all - rate_half), and a fixed
// part P - rate x E.
//   energy <mode> <workgroups> <seconds>     mode: mfma | lds | valu | pkvalu | mix | dma | hbm | idle
// mfma  v_mfma_f32_32x32x16_bf16, 2 waves per SIMD, 4 independent accumulators each, random bf16 operands in registers
// lds   ds_read_b128, conflict-free, 2 waves per SIMD
// valu  v_fma_f32, 8 independent chains, 4 waves per SIMD;  pkvalu: v_pk_fma_f32
// mix   the convolution kernels' ratio: 6 ds_read_b128 + 8 MFMAs per step (0.75 reads per MFMA), 2 waves per SIMD
// dma   global_load_lds_dwordx4 from an L2-resident 1 MB region (the weight stream of modconv_dma.hip), 2 waves per SIMD
// hbm   16-byte-per-lane copy of a 2 GiB buffer
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ unsigned hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// a random bf16 pair in [-1, 1): sign, exponent 2^-1 .. 2^-8, random mantissa
__device__ __forceinline__ unsigned rnd_bf16x2(unsigned s) {
  const unsigned a = hash(s), b = hash(s + 0x9e3779b9u);
  const unsigned ea = 119u + (a & 7u), eb = 119u + (b & 7u);
  const unsigned ha = ((a >> 31) << 15) | (ea << 7) | ((a >> 8) & 0x7f), hb = ((b >> 31) << 15) | (eb << 7) | ((b >> 8) & 0x7f);
  return ha | (hb << 16);
}

constexpr int INNER = 256;   // instructions of the class per loop iteration and wave (mix: MFMAs)

__global__ __launch_bounds__(512) void k_mfma(float* out, int iters) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  u32x4 a[4], b[4];
  for (int i = 0; i < 4; i++)
    for (int k = 0; k < 4; k++) { a[i][k] = rnd_bf16x2(t * 64 + i * 8 + k); b[i][k] = rnd_bf16x2(t * 64 + 32 + i * 8 + k); }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < INNER / 4; j++) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[j & 3]), __builtin_bit_cast(bf16x8, b[0]), c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[j & 3]), __builtin_bit_cast(bf16x8, b[1]), c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[j & 3]), __builtin_bit_cast(bf16x8, b[2]), c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[j & 3]), __builtin_bit_cast(bf16x8, b[3]), c3, 0, 0, 0);
    }
    // keep the accumulators bounded (operands ~ 2^-4: sums stay small) and the loop from being collapsed
    asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
  }
  float s = 0;
  for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[t] = s;
}

__global__ __launch_bounds__(512) void k_lds(float* out, int iters) {
  __shared__ u32x4 lds[16 * 64 * 8];   // 128 KB
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = threadIdx.x; i < 16 * 64 * 8; i += blockDim.x) lds[i] = u32x4{hash(i), hash(i + 77777u), hash(i * 3u), hash(i * 7u)};
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4* p = lds + wave * 16 * 64 + lane;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < INNER; j += 16) {
      u32x4 v[16];
#pragma unroll
      for (int q = 0; q < 16; q++) v[q] = p[q * 64];
#pragma unroll
      for (int q = 0; q < 16; q++) asm volatile("" :: "v"(v[q]));
      acc ^= v[0];
      asm volatile("" ::: "memory");   // (the next 16 reads are re-issued, not hoisted)
    }
  }
  out[t] = (float)(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
}

template <bool PK>
__global__ __launch_bounds__(1024) void k_valu(float* out, int iters) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  float x[16];
  for (int i = 0; i < 16; i++) x[i] = __uint_as_float(0x3f000000u | (hash(t * 16 + i) & 0x7fffffu)) - 0.75f;
  // x -> c - x -> x -> ...: every operation flips the sign and most mantissa bits (a contracting map would settle on its fixed
  // point and stop toggling the data path)
  const float m = -1.0f, c = x[3];
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < INNER / 8; j++) {
      if (PK) {
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
          f32x2 v = {x[q], x[q + 1]};
          v = v * f32x2{m, m} + f32x2{c, c};
          asm volatile("" : "+v"(v));
          x[q] = v[0]; x[q + 1] = v[1];
        }
      } else {
#pragma unroll
        for (int q = 0; q < 8; q++) { x[q] = fmaf(x[q], m, c); asm volatile("" : "+v"(x[q])); }
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; i++) s += x[i];
  out[t] = s;
}

__global__ __launch_bounds__(512) void k_mix(float* out, int iters) {
  __shared__ u32x4 lds[16 * 64 * 8];
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = threadIdx.x; i < 16 * 64 * 8; i += blockDim.x) lds[i] = u32x4{rnd_bf16x2(i), rnd_bf16x2(i + 7777u), rnd_bf16x2(i * 3u), rnd_bf16x2(i * 7u)};
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4* p = lds + wave * 16 * 64 + lane;
  f32x16 c[8];
  for (int i = 0; i < 8; i++) c[i] = f32x16{0};
  u32x4 af[4], bf[2], af1[4], bf1[2];
#define LD(A_, B_, O_) { for (int q = 0; q < 4; q++) A_[q] = p[((O_) + q) * 64]; for (int q = 0; q < 2; q++) B_[q] = p[((O_) + 4 + q) * 64]; }
#define MM(A_, B_) { _Pragma("unroll") for (int i = 0; i < 4; i++) _Pragma("unroll") for (int j = 0; j < 2; j++) \
    c[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, B_[j]), __builtin_bit_cast(bf16x8, A_[i]), c[i * 2 + j], 0, 0, 0); }
  LD(af, bf, 0)
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < INNER / 16; j++) {
      LD(af1, bf1, 6)
      MM(af, bf)
      asm volatile("" ::: "memory");
      LD(af, bf, (j & 1) * 2)
      MM(af1, bf1)
      asm volatile("" ::: "memory");
    }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) for (int e = 0; e < 16; e++) s += c[i][e];
  out[t] = s;
}

__global__ __launch_bounds__(512) void k_dma(float* out, const char* src, int iters) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // 128 KB
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem) + wave * 16384;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const unsigned voff = (unsigned)((((it * 16 + j) * 8 + wave) & 1023) * 1024 + lane * 16);
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(lds0 + j * 1024) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  out[t] = (float)smem[threadIdx.x * 16];
}

__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = a[i];
}

__global__ void k_fill(unsigned* p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = hash((unsigned)i);
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: energy <mode> <workgroups> <seconds>\n"); return 2; }
  const char* mode = argv[1];
  const int wgs = atoi(argv[2]);
  const double secs = atof(argv[3]);
  float* out;
  (void)hipMalloc((void**)&out, 4096 * 1024 * 4);
  char* buf = nullptr;
  const long copy_bytes = 1L << 31;
  if (!strcmp(mode, "hbm")) {
    (void)hipMalloc((void**)&buf, 2 * copy_bytes);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned*)buf, copy_bytes / 4);
  } else if (!strcmp(mode, "dma")) {
    (void)hipMalloc((void**)&buf, 1 << 20);
    hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, (unsigned*)buf, (1 << 20) / 4);
    (void)hipFuncSetAttribute((const void*)k_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  }
  (void)hipDeviceSynchronize();
  const int iters = 2000;
  double per_launch = 0;   // instructions of the class per launch (wave-instructions), or bytes for hbm
  auto launch = [&]() {
    if (!strcmp(mode, "mfma")) { hipLaunchKernelGGL(k_mfma, dim3(wgs), dim3(512), 0, 0, out, iters); per_launch = (double)wgs * 8 * iters * INNER; }
    else if (!strcmp(mode, "lds")) { hipLaunchKernelGGL(k_lds, dim3(wgs), dim3(512), 0, 0, out, iters); per_launch = (double)wgs * 8 * iters * INNER; }
    else if (!strcmp(mode, "valu")) { hipLaunchKernelGGL(k_valu<false>, dim3(wgs), dim3(1024), 0, 0, out, iters); per_launch = (double)wgs * 16 * iters * INNER; }
    else if (!strcmp(mode, "pkvalu")) { hipLaunchKernelGGL(k_valu<true>, dim3(wgs), dim3(1024), 0, 0, out, iters); per_launch = (double)wgs * 16 * iters * INNER; }
    else if (!strcmp(mode, "mix")) { hipLaunchKernelGGL(k_mix, dim3(wgs), dim3(512), 0, 0, out, iters); per_launch = (double)wgs * 8 * iters * INNER; }
    else if (!strcmp(mode, "dma")) { hipLaunchKernelGGL(k_dma, dim3(wgs), dim3(512), 128 * 1024, 0, out, buf, iters / 4); per_launch = (double)wgs * 8 * (iters / 4) * 16; }
    else if (!strcmp(mode, "hbm")) { hipLaunchKernelGGL(k_copy, dim3(wgs * 8), dim3(256), 0, 0, (const uint4*)buf, (uint4*)(buf + copy_bytes), copy_bytes / 16); per_launch = 2.0 * copy_bytes; }
    else if (!strcmp(mode, "idle")) { per_launch = 0; }
    else { fprintf(stderr, "unknown mode %s\n", mode); exit(2); }
  };
  launch();
  (void)hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  double total = 0;
  long launches = 0;
  while (true) {
    for (int i = 0; i < 4; i++) { launch(); total += per_launch; launches++; }
    (void)hipDeviceSynchronize();
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (el >= secs) {
      printf("{\"mode\": \"%s\", \"workgroups\": %d, \"seconds\": %.3f, \"rate\": %.6e, \"launches\": %ld}\n", mode, wgs, el, total / el, launches);
      break;
    }
    if (!strcmp(mode, "idle")) { struct timespec ts = {0, 100000000}; nanosleep(&ts, nullptr); }
  }
  return 0;
}
