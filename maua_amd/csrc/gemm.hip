// C = A x W^T (+ bias, + residual) on the matrix cores: the 1x1 convolutions of the guided-diffusion UNet (unet.hip).
//
// Replaces (reference): the `conv_nd(1, C, 3C, 1)` / `conv_nd(1, C, C, 1)` of guided_diffusion's AttentionBlock (qkv,
// proj_out + the block's residual) and the 1x1 `skip_connection` of its ResBlock, as built by
// maua/diffusion/processors/guided.py:164-209 (create_models; the network source is the un-vendored submodule
// maua/submodules/guided_diffusion).  In NHWC a 1x1 convolution over [B, H, W, C] IS the row-major GEMM
// [B*H*W, C] x [N, C]^T, so there is no layout change around it.
//
// A may come from TWO tensors (columns [0, K0) from a0, [K0, K0 + K1) from a1): the UNet's decoder concatenates the
// running features with the encoder's skip tensor along channels (`th.cat([h, hs.pop()], dim=1)`) - here the
// concatenation is never materialised, the consumers read both sources.
//
// Tile: 64 rows x 128 columns per 256-thread workgroup, K in 64-byte chunks, register prefetch of the next chunk while
// the current one is multiplied (the structure of modconv_lowres.hip without the tap gather); v_mfma_f32_32x32x16_bf16
// or 4 x v_mfma_f32_32x32x2_f32 (exact-f32 parity mode) from identical 16-byte LDS fragments; rows of 64 + 16 bytes keep
// the ds_read_b128 fragment reads bank-conflict-free.  These GEMMs are ~3 % of the UNet's FLOPs (the 3x3 convolutions
// are the rest), so the tile is sized for simplicity and edge handling (any M, N % 32 == 0), not for the MFMA roof.
#include <cstdlib>

#include "common.h"
#include "internal.h"

namespace maua {

namespace {

template <typename T> struct GMma;
template <> struct GMma<bf16_t> {
  __device__ static __forceinline__ void step(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0,
                                                  0, 0);
  }
};
template <> struct GMma<float> {
  __device__ static __forceinline__ void step(f32x16& acc, const u32x4& a, const u32x4& b) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], acc, 0, 0, 0);
  }
};

constexpr int GKCB = 64, GRS = GKCB + 16, GBM = 64, GBN = 128;

template <typename T>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs g) {
  constexpr int KC = GKCB / (int)sizeof(T), EPC = 16 / (int)sizeof(T);
  __shared__ __attribute__((aligned(16))) char a_s[GBM * GRS];
  __shared__ __attribute__((aligned(16))) char b_s[GBN * GRS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const long m0 = (long)blockIdx.x * GBM;
  const int n0 = blockIdx.y * GBN;
  const int q = tid & 3, row0 = tid >> 2;  // staging role: 16-byte piece q of A row row0 and W rows row0, row0 + 64
  const int K = g.K0 + g.K1, stages = K / KC;
  const T* a0 = reinterpret_cast<const T*>(g.a0) + (long)blockIdx.z * g.a_bstride;
  const T* a1 = reinterpret_cast<const T*>(g.a1);
  const T* w = reinterpret_cast<const T*>(g.w) + (long)blockIdx.z * g.w_bstride;
  g.c = reinterpret_cast<char*>(g.c) + (long)blockIdx.z * g.c_bstride * (g.c_f32 ? 4 : (long)sizeof(T));
  const long am = m0 + row0;
  const bool a_ok = am < g.M;
  const bool w_ok0 = n0 + row0 < g.N, w_ok1 = n0 + row0 + 64 < g.N;

  u32x4 areg, breg[2];
#define GEMM_LOAD(S)                                                                                         \
  {                                                                                                          \
    const int kc_ = (S) * KC + q * EPC;                                                                      \
    areg = u32x4{0u, 0u, 0u, 0u};                                                                            \
    if (a_ok)                                                                                                \
      areg = kc_ < g.K0 ? *reinterpret_cast<const u32x4*>(a0 + am * g.lda0 + kc_)                            \
                        : *reinterpret_cast<const u32x4*>(a1 + am * g.lda1 + (kc_ - g.K0));                  \
    breg[0] = breg[1] = u32x4{0u, 0u, 0u, 0u};                                                               \
    if (w_ok0) breg[0] = *reinterpret_cast<const u32x4*>(w + (long)(n0 + row0) * K + kc_);                   \
    if (w_ok1) breg[1] = *reinterpret_cast<const u32x4*>(w + (long)(n0 + row0 + 64) * K + kc_);              \
  }

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[i][e] = 0.f;

  GEMM_LOAD(0)
  for (int s = 0; s < stages; s++) {
    __syncthreads();  // the previous stage's fragment reads are done
    *reinterpret_cast<u32x4*>(a_s + row0 * GRS + q * 16) = areg;
    *reinterpret_cast<u32x4*>(b_s + row0 * GRS + q * 16) = breg[0];
    *reinterpret_cast<u32x4*>(b_s + (row0 + 64) * GRS + q * 16) = breg[1];
    __syncthreads();
    if (s + 1 < stages) GEMM_LOAD(s + 1)  // flies during the MFMAs
#pragma unroll
    for (int ks = 0; ks < GKCB / 32; ks++) {
      const u32x4 bf = *reinterpret_cast<const u32x4*>(b_s + (wave * 32 + r) * GRS + ks * 32 + h * 16);
      const u32x4 f0 = *reinterpret_cast<const u32x4*>(a_s + r * GRS + ks * 32 + h * 16);
      const u32x4 f1 = *reinterpret_cast<const u32x4*>(a_s + (32 + r) * GRS + ks * 32 + h * 16);
      GMma<T>::step(acc[0], bf, f0);  // rows = output columns n, columns = rows m: a lane owns one m and 4-column runs
      GMma<T>::step(acc[1], bf, f1);
    }
  }
#undef GEMM_LOAD

  const int nb = n0 + wave * 32;
  if (nb >= g.N) return;  // (wave-uniform; N % 32 == 0)
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const long m = m0 + i * 32 + r;
    if (m >= g.M) continue;
#pragma unroll
    for (int qd = 0; qd < 4; qd++) {
      const int n = nb + 8 * qd + 4 * h;
      float v[4] = {acc[i][qd * 4], acc[i][qd * 4 + 1], acc[i][qd * 4 + 2], acc[i][qd * 4 + 3]};
      if (g.bias) {
        const float4 bv = *reinterpret_cast<const float4*>(g.bias + n);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
      }
      if (g.res) {
        const T* rp = reinterpret_cast<const T*>(g.res) + m * g.ldr + n;
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] += Elem<T>::load(rp + k);
      }
      if (g.c_f32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.c) + m * g.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
      } else if constexpr (sizeof(T) == 2) {
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(g.c) + m * g.ldc + n) =
            make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
      } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.c) + m * g.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
}


// ---- the wide form: 128 x 128 tile, 4 waves of 64 x 64 (2 x 2 MFMA blocks), 128-byte K chunks (4 k-steps = 16 MFMAs per
// wave between barriers instead of 4), and the tile leaves through LDS as full 16-byte row pieces (bias / residual added in
// the copy-out) instead of 8-byte scattered stores.  Needs both K parts in whole 128-byte chunks; the kernel above serves
// the rest (narrow test networks, 96-channel concatenations).
constexpr int WKCB = 128, WRS = WKCB + 16, WBM = 128, WBN = 128;

template <typename T>
__global__ __launch_bounds__(256) void gemm_nt128_kernel(GemmArgs g) {
  constexpr int KC = WKCB / (int)sizeof(T), EPC = 16 / (int)sizeof(T);
  constexpr int ES = WBN * (int)sizeof(T) + 16, PPP = WBN / EPC;   // epilogue tile row stride, 16-byte pieces per row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* a_s = smem;
  char* b_s = smem + WBM * WRS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  long mt = blockIdx.x;
  int nt = blockIdx.y;
  if (g.remap_nt) {   // (see GemmArgs.remap_nt)
    const long L = blockIdx.x, j = L >> 3;
    nt = (int)(j % g.remap_nt);
    mt = (j / g.remap_nt) * 8 + (L & 7);
    if (mt * WBM >= g.M) return;
  }
  const long m0 = mt * WBM;
  const int n0 = nt * WBN;
  const int q = tid & 7, row0 = tid >> 3;  // staging role: 16-byte piece q of rows row0 + 32 i (i < 4), A and W alike
  const int K = g.K0 + g.K1, stages = K / KC;
  const T* a0 = reinterpret_cast<const T*>(g.a0) + (long)blockIdx.z * g.a_bstride;
  const T* a1 = reinterpret_cast<const T*>(g.a1);
  const T* w = reinterpret_cast<const T*>(g.w) + (long)blockIdx.z * g.w_bstride;
  g.c = reinterpret_cast<char*>(g.c) + (long)blockIdx.z * g.c_bstride * (long)sizeof(T);
  u32x4 areg[4], breg[4];
#define GEMMW_LOAD(S)                                                                                        \
  {                                                                                                          \
    const int kc_ = (S) * KC + q * EPC;                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; i++) {                                                         \
      const long am_ = m0 + row0 + 32 * i;                                                                   \
      const int wn_ = n0 + row0 + 32 * i;                                                                    \
      areg[i] = u32x4{0u, 0u, 0u, 0u};                                                                       \
      if (am_ < g.M)                                                                                         \
        areg[i] = kc_ < g.K0 ? *reinterpret_cast<const u32x4*>(a0 + am_ * g.lda0 + kc_)                      \
                             : *reinterpret_cast<const u32x4*>(a1 + am_ * g.lda1 + (kc_ - g.K0));            \
      breg[i] = u32x4{0u, 0u, 0u, 0u};                                                                       \
      if (wn_ < g.N) breg[i] = *reinterpret_cast<const u32x4*>(w + (long)wn_ * K + kc_);                     \
    }                                                                                                        \
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  GEMMW_LOAD(0)
  for (int s = 0; s < stages; s++) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
      *reinterpret_cast<u32x4*>(a_s + (row0 + 32 * i) * WRS + q * 16) = areg[i];
      *reinterpret_cast<u32x4*>(b_s + (row0 + 32 * i) * WRS + q * 16) = breg[i];
    }
    __syncthreads();
    if (s + 1 < stages) GEMMW_LOAD(s + 1)
#pragma unroll
    for (int ks = 0; ks < WKCB / 32; ks++) {
      u32x4 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        af[i] = *reinterpret_cast<const u32x4*>(a_s + (wm * 64 + i * 32 + r) * WRS + ks * 32 + h * 16);
        bf[i] = *reinterpret_cast<const u32x4*>(b_s + (wn * 64 + i * 32 + r) * WRS + ks * 32 + h * 16);
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) GMma<T>::step(acc[i][j], bf[j], af[i]);   // rows = columns n, columns = rows m
    }
  }
#undef GEMMW_LOAD
  // ---- accumulators -> LDS tile [m][n] (T) -> 16-byte pieces with bias / residual
  __syncthreads();
  char* epi = smem;
  // the bias pieces of the wave's 8 column groups in one round trip (each used to be loaded, and waited for, inside its branch)
  float4 bq[2][4];
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int qd = 0; qd < 4; qd++) bq[j][qd] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g.bias) {
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int qd = 0; qd < 4; qd++) {
        const int n = wn * 64 + j * 32 + 8 * qd + 4 * h;
        if (n0 + n < g.N) bq[j][qd] = *reinterpret_cast<const float4*>(g.bias + n0 + n);
      }
  }
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int m = wm * 64 + i * 32 + r;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int qd = 0; qd < 4; qd++) {
        const int n = wn * 64 + j * 32 + 8 * qd + 4 * h;
        const float4 bv = bq[j][qd];
        const float v[4] = {acc[i][j][qd * 4] + bv.x, acc[i][j][qd * 4 + 1] + bv.y, acc[i][j][qd * 4 + 2] + bv.z,
                            acc[i][j][qd * 4 + 3] + bv.w};
        if constexpr (sizeof(T) == 2)
          *reinterpret_cast<uint2*>(epi + m * ES + n * 2) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
        else
          *reinterpret_cast<float4*>(epi + m * ES + n * 4) = make_float4(v[0], v[1], v[2], v[3]);
      }
  }
  __syncthreads();
  static_assert((WBM * PPP) % 256 == 0, "whole copy-out steps");
  constexpr int NIT = WBM * PPP / 256;
  u32x4 rvs[NIT];   // residual pieces of all copy-out steps: one round trip
  if (g.res) {
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int p = tid + it * 256, m = p / PPP, pc = p - m * PPP;
      const long gm = m0 + m;
      const int gn = n0 + pc * EPC;
      rvs[it] = u32x4{0u, 0u, 0u, 0u};
      if (gm < g.M && gn < g.N) rvs[it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(g.res) + gm * g.ldr + gn);
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    const int p = tid + it * 256;
    const int m = p / PPP, pc = p - m * PPP;
    const long gm = m0 + m;
    const int gn = n0 + pc * EPC;
    if (gm >= g.M || gn >= g.N) continue;
    u32x4 v = *reinterpret_cast<const u32x4*>(epi + m * ES + pc * 16);
    if (g.res) {
      const u32x4 rv = rvs[it];
      if constexpr (sizeof(T) == 2) {
        // (the bias-added value was rounded to bf16 in the tile; the residual is added to that, like the conv kernels do)
#pragma unroll
        for (int k = 0; k < 4; k++)
          v[k] = pack2bf(bf2f((bf16_t)(v[k] & 0xffff)) + bf2f((bf16_t)(rv[k] & 0xffff)), bf2f((bf16_t)(v[k] >> 16)) + bf2f((bf16_t)(rv[k] >> 16)));
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = __float_as_uint(__uint_as_float(v[k]) + __uint_as_float(rv[k]));
      }
    }
    *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(g.c) + gm * g.ldc + gn) = v;
  }
}

}  // namespace

int launch_gemm_nt(hipStream_t stream, int dtype, const GemmArgs& g) {
  MAUA_REQUIRE(dtype == MAUA_BF16 || dtype == MAUA_F32, "gemm_nt: unsupported dtype");
  const int kc = dtype == MAUA_BF16 ? 32 : 16;
  MAUA_REQUIRE(g.a0 && g.w && g.c && g.K0 > 0 && g.K0 % kc == 0 && g.K1 % kc == 0 && (g.K1 == 0 || g.a1),
               "gemm_nt: K parts must be multiples of 64 bytes");
  MAUA_REQUIRE(g.N % 32 == 0 && g.N > 0 && g.lda0 % 4 == 0 && g.ldc % 4 == 0, "gemm_nt: N must be a multiple of 32");
  if (g.M == 0) return MAUA_OK;
  const unsigned nb = g.batch > 1 ? (unsigned)g.batch : 1u;
  MAUA_REQUIRE(nb == 1 || (!g.a1 && g.K1 == 0 && !g.res && !g.bias && nb <= 65535), "gemm_nt: a batched launch takes one A source, no bias, no residual");
  if (nb == 1 && g.prefer_dma && gemm_dma_supported(dtype, g)) return launch_gemm_dma(stream, g);
  MAUA_REQUIRE(g.epi == 0, "gemm_nt: the QuickGELU epilogue forms exist on the LDS-direct kernel only (callers check gemm_dma_supported)");
  const int kcw = dtype == MAUA_BF16 ? 64 : 32;   // channels per 128-byte chunk
  const int epc = dtype == MAUA_BF16 ? 8 : 4;
  if (!g.c_f32 && g.K0 % kcw == 0 && g.K1 % kcw == 0 && g.M >= 128 && g.ldc % epc == 0 && (!g.res || g.ldr % epc == 0) &&
      g.lda0 % epc == 0 && (g.K1 == 0 || g.lda1 % epc == 0)) {
    const size_t smem = std::max<size_t>((size_t)2 * WBM * WRS, (size_t)WBM * (WBN * (dtype == MAUA_BF16 ? 2 : 4) + 16));
    dim3 gridw((unsigned)((g.M + WBM - 1) / WBM), (unsigned)((g.N + WBN - 1) / WBN), nb);
    GemmArgs gr = g;
    static const bool xcd_off = getenv("MAUA_GEMM_XCD_OFF") != nullptr;
    if (nb == 1 && gridw.y >= 2 && gridw.y <= 8 && gridw.x >= 64 && !xcd_off) {   // few N tiles over many rows: keep an M tile's N tiles on one XCD
      gr.remap_nt = (int)gridw.y;
      gridw = dim3((gridw.x + 7) / 8 * 8 * gridw.y, 1, 1);
    }
    if (dtype == MAUA_BF16) {
      MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt128_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      hipLaunchKernelGGL(gemm_nt128_kernel<bf16_t>, gridw, dim3(256), smem, stream, gr);
    } else {
      MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt128_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      hipLaunchKernelGGL(gemm_nt128_kernel<float>, gridw, dim3(256), smem, stream, gr);
    }
    MAUA_HIP_CHECK(hipGetLastError());
    return MAUA_OK;
  }
  dim3 grid((unsigned)((g.M + GBM - 1) / GBM), (unsigned)((g.N + GBN - 1) / GBN), nb);
  if (dtype == MAUA_BF16)
    hipLaunchKernelGGL(gemm_nt_kernel<bf16_t>, grid, dim3(256), 0, stream, g);
  else
    hipLaunchKernelGGL(gemm_nt_kernel<float>, grid, dim3(256), 0, stream, g);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace maua
