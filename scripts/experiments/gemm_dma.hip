// C = A x W^T (+ bias, + residual) for the LARGE 1x1 layers of the guided-diffusion UNet, both operands streamed HBM -> LDS by
// LDS-direct loads (global_load_lds_dwordx4) two K chunks ahead of the matrix cores.
//
// Replaces (reference): the same layers as gemm.hip - guided_diffusion's AttentionBlock qkv / proj_out and the 1x1 ResBlock
// skip_connection of the network maua/diffusion/processors/guided.py:164-209 builds - where they are big enough to fill the chip
// with 256 x 128 tiles (launch_gemm_nt routes; everything else stays on gemm.hip's register-staged kernels).
//
// Why a second kernel: gemm.hip stages a chunk through registers one stage ahead - 16 MFMAs per wave are shorter than an HBM round trip
// under load, and two register sets (two stages ahead) cost 260 VGPRs or 72 spilled ones (measured, round 5).  LDS-direct loads need no
// registers: chunk c + 3 is requested as soon as the barrier behind chunk c's last fragment reads has passed, two chunks of MFMAs
// before it is needed (the structure of modconv_tconv_dma.hip without the halo).
//
// Tile 256 (rows of A) x 128 (columns) per 512-thread workgroup: 8 waves of 64 x 64 (2 x 2 MFMA blocks), K in 128-byte chunks
// (64 bf16 channels; 4 k-steps = 16 MFMAs per wave per chunk), a ring of three 48 KB LDS buffers (two chunks in flight).  LDS rows are 128 bytes with the 16-byte
// piece index XOR-ed by (row >> 1) & 7 - applied to the SOURCE address of the load and to the fragment read - so that the 16 lanes of a
// ds_read_b128 group hit 16 different bank groups.  A comes from up to two tensors (the decoder's virtual concatenation), switched
// per chunk.  bf16 only; f32 (parity mode) and small / ragged shapes run on gemm.hip.
#include "common.h"
#include "internal.h"

namespace maua {

namespace {

constexpr int DBM = 256, DBN = 128, DKB = 128, DNW = 8, DNT = DNW * 64;
constexpr int ABUF = DBM * DKB, BBUF = DBN * DKB, STAGE = ABUF + BBUF;   // 32 KB + 16 KB
constexpr int AJ = ABUF / 1024 / DNW, BJ = BBUF / 1024 / DNW;             // 4 + 2 LDS-direct instructions per wave per chunk
constexpr int DES = DBN * 2 + 16, DPPP = DBN / 8;                          // epilogue tile row stride, 16-byte pieces per row

__device__ __forceinline__ unsigned lds_off(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ void dma16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ void mma(f32x16& acc, const u32x4& w, const u32x4& x) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

__global__ __launch_bounds__(DNT) void gemm_dma_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_off(smem));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  // Workgroup order (1-D grid): the dispatcher places block L on XCD L % 8, each XCD has its own L2.  The N / 128 column tiles of one
  // 256-row tile of A run back to back ON ONE XCD, so A is fetched from HBM once and found in that L2 by the others (a tall A against
  // 256 columns - the decoder's skip convolutions - is otherwise streamed from HBM once per column tile: 3.7 TB/s for 0.38 PFLOP/s).
  const int NTL = g.N / DBN;
  const int L = blockIdx.x, xcd = L & 7, idx = L >> 3;
  const long mt = (long)(idx / NTL) * 8 + xcd;
  if (mt * DBM >= g.M) return;
  const long m0 = mt * DBM;
  const int n0 = (idx % NTL) * DBN;
  const int K = g.K0 + g.K1;
  const int n_chunks = K / 64, c_split = g.K0 / 64;   // chunks [0, c_split) come from a0, the rest from a1
  const char* a0 = reinterpret_cast<const char*>(g.a0);
  const char* a1 = reinterpret_cast<const char*>(g.a1);
  const char* wp = reinterpret_cast<const char*>(g.w);

  // sources of this lane's loads: instruction ii = wave + 8 j fills LDS rows [8 ii, 8 ii + 8) x 8 pieces; LDS piece p of row R holds
  // global piece p ^ swz(R).  Rows past M re-read the last row (their results are never stored).
  unsigned aoff0[AJ], aoff1[AJ], boff[BJ];
#pragma unroll
  for (int j = 0; j < AJ; j++) {
    const int R = 8 * (wave + DNW * j) + (lane >> 3);
    const long gm = m0 + R < g.M ? m0 + R : g.M - 1;
    const int q = (lane & 7) ^ swz(R);
    aoff0[j] = (unsigned)((gm * g.lda0 + q * 8) * 2);
    aoff1[j] = (unsigned)((gm * g.lda1 + q * 8) * 2);
  }
#pragma unroll
  for (int j = 0; j < BJ; j++) {
    const int R = 8 * (wave + DNW * j) + (lane >> 3);
    const int q = (lane & 7) ^ swz(R);
    boff[j] = (unsigned)((((long)(n0 + R)) * K + q * 8) * 2);
  }
#define GD_ISSUE(C_, BUF_)                                                                                     \
  {                                                                                                            \
    const bool first_ = (C_) < c_split;                                                                        \
    const char* as_ = first_ ? a0 + (long)(C_) * DKB : a1 + (long)((C_) - c_split) * DKB;                      \
    _Pragma("unroll") for (int j = 0; j < AJ; j++)                                                            \
        dma16_s(as_, first_ ? aoff0[j] : aoff1[j], lds0 + (BUF_) * STAGE + (wave + DNW * j) * 1024);           \
    const char* ws_ = wp + (long)(C_) * DKB;                                                                   \
    _Pragma("unroll") for (int j = 0; j < BJ; j++)                                                            \
        dma16_s(ws_, boff[j], lds0 + (BUF_) * STAGE + ABUF + (wave + DNW * j) * 1024);                         \
  }
  // fragment addresses: rows of this wave's 64 x 64 block; lane (r, h) reads piece (2 ks + h) ^ swz(row)
  int arow[2], brow[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    arow[i] = wm * 64 + i * 32 + r;
    brow[i] = wn * 64 + i * 32 + r;
  }
#define GD_A(I_, KS_, BUF_) \
  (*reinterpret_cast<const u32x4*>(smem + (BUF_) * STAGE + arow[I_] * DKB + (((2 * (KS_) + h) ^ swz(arow[I_])) << 4)))
#define GD_B(J_, KS_, BUF_) \
  (*reinterpret_cast<const u32x4*>(smem + (BUF_) * STAGE + ABUF + brow[J_] * DKB + (((2 * (KS_) + h) ^ swz(brow[J_])) << 4)))

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  // three buffers: chunk c + 3 is requested behind chunk c's last fragment reads, chunk c + 1 is waited for with chunk c + 2 still in
  // flight (every wave issues exactly AJ + BJ loads per chunk, loads retire in order: a counted vmcnt)
  GD_ISSUE(0, 0)
  if (n_chunks > 1) GD_ISSUE(1, 1)
  if (n_chunks > 2) GD_ISSUE(2, 2)
  if (n_chunks > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (n_chunks > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  static_assert(AJ + BJ == 6, "the counted waits assume six loads per wave and chunk");
  int buf = 0;
  for (int c = 0; c < n_chunks; c++) {
#pragma unroll
    for (int ks = 0; ks < 3; ks++) {
      const u32x4 A0 = GD_A(0, ks, buf), A1 = GD_A(1, ks, buf), B0 = GD_B(0, ks, buf), B1 = GD_B(1, ks, buf);
      mma(acc[0][0], B0, A0); mma(acc[0][1], B1, A0);
      mma(acc[1][0], B0, A1); mma(acc[1][1], B1, A1);
    }
    {
      // the chunk's last fragments are read BEFORE the barrier that frees its buffer for chunk c + 2
      const u32x4 A0 = GD_A(0, 3, buf), A1 = GD_A(1, 3, buf), B0 = GD_B(0, 3, buf), B1 = GD_B(1, 3, buf);
      if (c + 2 < n_chunks) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // chunk c + 1 has landed (c + 2 may still be in flight)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                   // ... for everybody; everybody's reads of chunk c have returned
      if (c + 3 < n_chunks) GD_ISSUE(c + 3, buf)
      mma(acc[0][0], B0, A0); mma(acc[0][1], B1, A0);
      mma(acc[1][0], B0, A1); mma(acc[1][1], B1, A1);
    }
    buf = buf == 2 ? 0 : buf + 1;
  }
#undef GD_ISSUE
#undef GD_A
#undef GD_B

  // ---- accumulators (+ bias) -> LDS tile [m][n] bf16 -> 16-byte row pieces (+ residual, added to the rounded value like gemm.hip)
  __syncthreads();
  char* epi = smem;
  float4 bq[2][4];
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int qd = 0; qd < 4; qd++) {
      bq[j][qd] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (g.bias) bq[j][qd] = *reinterpret_cast<const float4*>(g.bias + n0 + wn * 64 + j * 32 + 8 * qd + 4 * h);
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
        *reinterpret_cast<uint2*>(epi + m * DES + n * 2) =
            make_uint2(pack2bf(acc[i][j][qd * 4] + bv.x, acc[i][j][qd * 4 + 1] + bv.y),
                       pack2bf(acc[i][j][qd * 4 + 2] + bv.z, acc[i][j][qd * 4 + 3] + bv.w));
      }
  }
  __syncthreads();
  constexpr int NIT = DBM * DPPP / DNT;   // 8 copy-out steps
  u32x4 rvs[NIT];
  if (g.res) {
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int p = tid + it * DNT, m = p / DPPP, pc = p - m * DPPP;
      const long gm = m0 + m;
      rvs[it] = u32x4{0u, 0u, 0u, 0u};
      if (gm < g.M) rvs[it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(g.res) + gm * g.ldr + n0 + pc * 8);
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    const int p = tid + it * DNT, m = p / DPPP, pc = p - m * DPPP;
    const long gm = m0 + m;
    if (gm >= g.M) continue;
    u32x4 v = *reinterpret_cast<const u32x4*>(epi + m * DES + pc * 16);
    if (g.res) {
      const u32x4 rv = rvs[it];
#pragma unroll
      for (int k = 0; k < 4; k++)
        v[k] = pack2bf(bf2f((bf16_t)(v[k] & 0xffff)) + bf2f((bf16_t)(rv[k] & 0xffff)), bf2f((bf16_t)(v[k] >> 16)) + bf2f((bf16_t)(rv[k] >> 16)));
    }
    *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(g.c) + gm * g.ldc + n0 + pc * 8) = v;
  }
}

}  // namespace

// shapes the LDS-direct kernel takes: bf16, both K parts in whole 64-channel chunks, N in whole 128-column tiles, 32-bit byte offsets
// into A and W, and enough tiles for the chip
bool gemm_dma_supported(int dtype, const GemmArgs& g) {
  if (dtype != MAUA_BF16 || g.c_f32 || g.K0 % 64 || g.K1 % 64 || g.N % DBN || g.M < DBM) return false;
  if (g.lda0 % 8 || (g.K1 && g.lda1 % 8) || g.ldc % 8 || (g.res && g.ldr % 8)) return false;
  const long K = g.K0 + g.K1;
  if (g.M * g.lda0 * 2 >= (1L << 32) || (g.K1 && g.M * g.lda1 * 2 >= (1L << 32)) || (long)g.N * K * 2 >= (1L << 32)) return false;
  return ((g.M + DBM - 1) / DBM) * (g.N / DBN) >= 256;
}

int launch_gemm_dma(hipStream_t stream, const GemmArgs& g) {
  MAUA_REQUIRE(gemm_dma_supported(MAUA_BF16, g), "gemm_dma: unsupported shape");
  const size_t smem = std::max<size_t>((size_t)3 * STAGE, (size_t)DBM * DES);
  MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const long mtiles8 = ((g.M + DBM - 1) / DBM + 7) / 8 * 8;
  dim3 grid((unsigned)(mtiles8 * (g.N / DBN)));
  hipLaunchKernelGGL(gemm_dma_kernel, grid, dim3(DNT), smem, stream, g);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace maua
