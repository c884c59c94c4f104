// C = A x W^T (+ bias, + residual, + QuickGELU forms) for LARGE plain GEMMs, both operands streamed HBM -> LDS by LDS-direct loads
// (global_load_lds_dwordx4) two K chunks ahead of the matrix cores.
//
// Replaces (reference): the nn.Linear / nn.MultiheadAttention projections of the CLIP image tower that maua/grad.py:96-165 (CLIPGrads)
// evaluates and differentiates per cutout batch (clip/model.py ResidualAttentionBlock: in_proj, out_proj, mlp.c_fc, mlp.c_proj - and
// the same four products against the transposed weights on the way back): 200 000 rows x 768 ... 3072 columns, K = 768 ... 3072,
// 95 % of a text-guided step's FLOPs.  Callers opt in with GemmArgs.prefer_dma (clip.hip); everything else - the diffusion UNet's 1x1
// layers (K of 8-16 chunks: measured no faster here, scripts/experiments/README.md round 5 (v)), f32 parity mode, ragged shapes -
// stays on gemm.hip's register-staged kernels.
//
// Epilogue forms (GemmArgs.epi), both applied to the value as stored (rounded to bf16), so that they equal the separate element-wise
// kernels of clip.hip bit for bit:  1 = c2 <- QuickGELU(c) stored beside c (mlp.c_fc: the pre-activation is kept for the gradient,
// the activation feeds c_proj);  2 = c <- c * QuickGELU'(aux) (the gradient through the activation on the way back).
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
#include <cstdlib>

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
// the same under a wave-uniform predicate (mask = all ones or zero): with EXEC cleared the request is not issued - a branch-free
// "only if that chunk exists" inside a straight-line MFMA sequence
__device__ __forceinline__ void dma16_s_if(const void* sbase, unsigned voff, unsigned lds_dst, int go) {
  unsigned long long saved;
  asm volatile("s_mov_b32 m0, %3\n\ts_mov_b64 %0, exec\n\ts_cmp_lg_u32 %4, 0\n\ts_cselect_b64 exec, %0, 0\n\t"
               "global_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, %0"
               : "=&s"(saved)
               : "v"(voff), "s"(sbase), "s"(lds_dst), "s"(go)
               : "memory", "scc");
}
__device__ __forceinline__ void mma(f32x16& acc, const u32x4& w, const u32x4& x) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }
// x * sigmoid(1.702 x) (clip/model.py QuickGELU) and its derivative s + 1.702 x s (1 - s)
__device__ __forceinline__ float qgelu(float x) { return x / (1.f + __expf(-1.702f * x)); }
__device__ __forceinline__ float qgelu_grad(float x) {
  const float s = 1.f / (1.f + __expf(-1.702f * x));
  return s * (1.f + 1.702f * x * (1.f - s));
}

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
    if (g.epi == 2) {
      const u32x4 hv = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(g.aux) + gm * g.ldaux + n0 + pc * 8);
#pragma unroll
      for (int k = 0; k < 4; k++)
        v[k] = pack2bf(bf2f((bf16_t)(v[k] & 0xffff)) * qgelu_grad(bf2f((bf16_t)(hv[k] & 0xffff))),
                       bf2f((bf16_t)(v[k] >> 16)) * qgelu_grad(bf2f((bf16_t)(hv[k] >> 16))));
    }
    *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(g.c) + gm * g.ldc + n0 + pc * 8) = v;
    if (g.epi == 1) {
      u32x4 a;
#pragma unroll
      for (int k = 0; k < 4; k++) a[k] = pack2bf(qgelu(bf2f((bf16_t)(v[k] & 0xffff))), qgelu(bf2f((bf16_t)(v[k] >> 16))));
      *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(g.c2) + gm * g.ldc2 + n0 + pc * 8) = a;
    }
  }
}


// ---- 256 x 256 tile (round 6: the CLIP tower's GEMMs, M ~ 200 000, N = 768 ... 3072, K = 768 ... 3072).  8 waves as 2 (M) x 4 (N), a wave
// owns 128 x 64 = 4 x 2 MFMA blocks: 6 fragment reads per 8 MFMAs (0.75 per MFMA; the 256 x 128 tile above: 1.0), 32 MFMAs per wave between
// barriers.  K in 128-byte chunks, TWO 64 KB stages (A 256 rows + W 256 rows): chunk c + 1 is requested right behind the barrier that
// ended chunk c - 1's reads of its stage and is waited for at the end of chunk c - one barrier per chunk, 128 KB of LDS, one workgroup
// per CU (the guide's "glds, 2 LDS buffers, BK = 64" row).  Same XOR swizzle, same XCD-aware tile order, same epilogue forms.
constexpr int QBM = 256, QBN = 256, QNW = 8, QNT = QNW * 64;
constexpr int QABUF = QBM * DKB, QBBUF = QBN * DKB, QSTAGE = QABUF + QBBUF;   // 32 KB + 32 KB
constexpr int QAJ = QABUF / 1024 / QNW, QBJ = QBBUF / 1024 / QNW;               // 4 + 4 LDS-direct instructions per wave per chunk
constexpr int QES = QBN * 2 + 16, QPPP = QBN / 8;

__global__ __launch_bounds__(QNT) void gemm256_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_off(smem));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;
  const int NTL = g.N / QBN;
  const int L = blockIdx.x, xcd = L & 7, idx = L >> 3;
  const long mt = (long)(idx / NTL) * 8 + xcd;
  if (mt * QBM >= g.M) return;
  const long m0 = mt * QBM;
  const int n0 = (idx % NTL) * QBN;
  const int K = g.K0;
  const int n_chunks = K / 64;
  const char* a0 = reinterpret_cast<const char*>(g.a0);
  const char* wp = reinterpret_cast<const char*>(g.w);

  unsigned aoff[QAJ], boff[QBJ];
#pragma unroll
  for (int j = 0; j < QAJ; j++) {
    const int R = 8 * (wave + QNW * j) + (lane >> 3);
    const long gm = m0 + R < g.M ? m0 + R : g.M - 1;
    aoff[j] = (unsigned)((gm * g.lda0 + (((lane & 7) ^ swz(R)) << 3)) * 2);
  }
#pragma unroll
  for (int j = 0; j < QBJ; j++) {
    const int R = 8 * (wave + QNW * j) + (lane >> 3);
    boff[j] = (unsigned)((((long)(n0 + R)) * K + (((lane & 7) ^ swz(R)) << 3)) * 2);
  }
#define GQ_ISSUE(C_, BUF_)                                                                                  \
  {                                                                                                         \
    const char* as_ = a0 + (long)(C_) * DKB;                                                                \
    const char* ws_ = wp + (long)(C_) * DKB;                                                                \
    _Pragma("unroll") for (int j = 0; j < QAJ; j++)                                                        \
        dma16_s(as_, aoff[j], lds0 + (BUF_) * QSTAGE + (wave + QNW * j) * 1024);                            \
    _Pragma("unroll") for (int j = 0; j < QBJ; j++)                                                        \
        dma16_s(ws_, boff[j], lds0 + (BUF_) * QSTAGE + QABUF + (wave + QNW * j) * 1024);                    \
  }
  // fragment addresses: block rows are multiples of 32, so the swizzle term (row >> 1) & 7 is the lane's (r >> 1) & 7 for every block
  const int sw = (r >> 1) & 7;
  const char* abase = smem + (wm * 128 + r) * DKB;
  const char* bbase = smem + QABUF + (wn * 64 + r) * DKB;
#define GQ_A(I_, KS_, BUF_) (*reinterpret_cast<const u32x4*>(abase + (BUF_) * QSTAGE + (I_) * 32 * DKB + (((2 * (KS_) + h) ^ sw) << 4)))
#define GQ_B(J_, KS_, BUF_) (*reinterpret_cast<const u32x4*>(bbase + (BUF_) * QSTAGE + (J_) * 32 * DKB + (((2 * (KS_) + h) ^ sw) << 4)))

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  // Software pipeline, one barrier per chunk, placed BEHIND the chunk's last fragment reads and AHEAD of its last k-step's MFMAs:
  //   k-steps 0 .. 2 of chunk c : fragments of the next k-step requested, then 8 MFMAs
  //   vmcnt(0) - this wave's pieces of chunk c + 1 have landed - lgkmcnt(0), barrier: chunk c + 1 is complete for everybody and nobody
  //                               reads chunk c's stage any more
  //   k-step 3                  : fragments (c + 1, 0) requested from the other stage; its 8 MFMAs with the 8 LDS-direct requests of chunk
  //                               c + 2 (into the stage just freed) issued between them - a request costs its wave 60 - 180 cycles of issue
  //                               time, which now passes under MFMAs instead of at an idle matrix pipe - due 3 k-steps later
  u32x4 Af[2][4], Bf[2][2];
#define GQ_FRAGS(F_, KS_, BUF_)                                                 \
  {                                                                             \
    Bf[F_][0] = GQ_B(0, KS_, BUF_); Bf[F_][1] = GQ_B(1, KS_, BUF_);             \
    _Pragma("unroll") for (int i = 0; i < 4; i++) Af[F_][i] = GQ_A(i, KS_, BUF_); \
  }
#define GQ_MMA(F_)                                                              \
  _Pragma("unroll") for (int i = 0; i < 4; i++) {                              \
    mma(acc[i][0], Bf[F_][0], Af[F_][i]);                                       \
    mma(acc[i][1], Bf[F_][1], Af[F_][i]);                                       \
  }
  // the last k-step: MFMA pairs with one A piece and one W piece of chunk C2_ requested behind each (C2_ < 0: nothing to request)
#define GQ_MMA_ISSUE(F_, C2_, BUF_)                                                                                  \
  {                                                                                                                  \
    const int go_ = __builtin_amdgcn_readfirstlane((C2_) < n_chunks ? 1 : 0);                                        \
    const char* as_ = a0 + (long)(C2_) * DKB;                                                                        \
    const char* ws_ = wp + (long)(C2_) * DKB;                                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; i++) {                                                                 \
      mma(acc[i][0], Bf[F_][0], Af[F_][i]);                                                                          \
      dma16_s_if(as_, aoff[i], lds0 + (BUF_) * QSTAGE + (wave + QNW * i) * 1024, go_);                               \
      mma(acc[i][1], Bf[F_][1], Af[F_][i]);                                                                          \
      dma16_s_if(ws_, boff[i], lds0 + (BUF_) * QSTAGE + QABUF + (wave + QNW * i) * 1024, go_);                       \
    }                                                                                                                \
  }
  static_assert(QAJ == 4 && QBJ == 4, "GQ_MMA_ISSUE pairs four A and four W pieces with the four MFMA pairs of a k-step");

  // the k-step's interleave is pinned - one fragment read of the NEXT k-step behind each of the first six MFMAs (the compiler otherwise
  // sinks the reads to just ahead of their first use and the wave waits out the LDS latency eight times per chunk): + 2 %
#define GQ_SCHED()                                                        \
  {                                                                       \
    _Pragma("unroll") for (int q_ = 0; q_ < 6; q_++) {                   \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                  \
    }                                                                     \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                    \
  }
  GQ_ISSUE(0, 0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (n_chunks > 1) GQ_ISSUE(1, 1)
  GQ_FRAGS(0, 0, 0)
  int buf = 0;
  for (int c = 0; c < n_chunks; c++) {
    GQ_FRAGS(1, 1, buf) GQ_MMA(0) GQ_SCHED()
    GQ_FRAGS(0, 2, buf) GQ_MMA(1) GQ_SCHED()
    GQ_FRAGS(1, 3, buf) GQ_MMA(0) GQ_SCHED()
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (c + 1 < n_chunks) GQ_FRAGS(0, 0, buf ^ 1)
    GQ_MMA_ISSUE(1, c + 2, buf)
    buf ^= 1;
  }
  __syncthreads();                       // (everybody's last fragment reads are behind: the stages become the epilogue tile)
#undef GQ_FRAGS
#undef GQ_SCHED
#undef GQ_MMA
#undef GQ_MMA_ISSUE
#undef GQ_CHUNK
#undef GQ_ISSUE
#undef GQ_A
#undef GQ_B

  // ---- accumulators (+ bias) -> LDS tile [m][n] bf16 -> 16-byte row pieces (+ residual / QuickGELU forms, as in the kernel above)
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
  for (int i = 0; i < 4; i++) {
    const int m = wm * 128 + i * 32 + r;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int qd = 0; qd < 4; qd++) {
        const int n = wn * 64 + j * 32 + 8 * qd + 4 * h;
        const float4 bv = bq[j][qd];
        *reinterpret_cast<uint2*>(epi + m * QES + n * 2) =
            make_uint2(pack2bf(acc[i][j][qd * 4] + bv.x, acc[i][j][qd * 4 + 1] + bv.y),
                       pack2bf(acc[i][j][qd * 4 + 2] + bv.z, acc[i][j][qd * 4 + 3] + bv.w));
      }
  }
  __syncthreads();
  constexpr int NIT = QBM * QPPP / QNT;   // 16 copy-out steps
#pragma unroll 4
  for (int it = 0; it < NIT; it++) {
    const int p = tid + it * QNT, m = p / QPPP, pc = p - m * QPPP;
    const long gm = m0 + m;
    if (gm >= g.M) continue;
    u32x4 v = *reinterpret_cast<const u32x4*>(epi + m * QES + pc * 16);
    if (g.res) {
      const u32x4 rv = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(g.res) + gm * g.ldr + n0 + pc * 8);
#pragma unroll
      for (int k = 0; k < 4; k++)
        v[k] = pack2bf(bf2f((bf16_t)(v[k] & 0xffff)) + bf2f((bf16_t)(rv[k] & 0xffff)), bf2f((bf16_t)(v[k] >> 16)) + bf2f((bf16_t)(rv[k] >> 16)));
    }
    if (g.epi == 2) {
      const u32x4 hv = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(g.aux) + gm * g.ldaux + n0 + pc * 8);
#pragma unroll
      for (int k = 0; k < 4; k++)
        v[k] = pack2bf(bf2f((bf16_t)(v[k] & 0xffff)) * qgelu_grad(bf2f((bf16_t)(hv[k] & 0xffff))),
                       bf2f((bf16_t)(v[k] >> 16)) * qgelu_grad(bf2f((bf16_t)(hv[k] >> 16))));
    }
    *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(g.c) + gm * g.ldc + n0 + pc * 8) = v;
    if (g.epi == 1) {
      u32x4 a;
#pragma unroll
      for (int k = 0; k < 4; k++) a[k] = pack2bf(qgelu(bf2f((bf16_t)(v[k] & 0xffff))), qgelu(bf2f((bf16_t)(v[k] >> 16))));
      *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(g.c2) + gm * g.ldc2 + n0 + pc * 8) = a;
    }
  }
}

}  // namespace

// shapes the LDS-direct kernel takes: bf16, both K parts in whole 64-channel chunks, N in whole 128-column tiles, 32-bit byte offsets
// into A and W, and enough tiles for the chip
bool gemm_dma_supported(int dtype, const GemmArgs& g) {
  if (dtype != MAUA_BF16 || g.c_f32 || g.K0 % 64 || g.K1 % 64 || g.N % DBN || g.M < DBM) return false;
  if (g.lda0 % 8 || (g.K1 && g.lda1 % 8) || g.ldc % 8 || (g.res && g.ldr % 8)) return false;
  if ((g.epi == 1 && (!g.c2 || g.ldc2 % 8)) || (g.epi == 2 && (!g.aux || g.ldaux % 8)) || g.epi < 0 || g.epi > 2) return false;
  const long K = g.K0 + g.K1;
  if (g.M * g.lda0 * 2 >= (1L << 32) || (g.K1 && g.M * g.lda1 * 2 >= (1L << 32)) || (long)g.N * K * 2 >= (1L << 32)) return false;
  return ((g.M + DBM - 1) / DBM) * (g.N / DBN) >= 256;
}

// the 256 x 256 form: one K source, N in whole 256-column tiles, at least a chip's worth of tiles
static bool gemm256_takes(const GemmArgs& g) {
  return g.K1 == 0 && g.N % QBN == 0 && ((g.M + QBM - 1) / QBM) * (g.N / QBN) >= 256;
}

int launch_gemm_dma(hipStream_t stream, const GemmArgs& g) {
  MAUA_REQUIRE(gemm_dma_supported(MAUA_BF16, g), "gemm_dma: unsupported shape");
  if (gemm256_takes(g) && !getenv("MAUA_GEMM_DMA_128")) {
    const size_t smem = std::max<size_t>((size_t)2 * QSTAGE, (size_t)QBM * QES);
    const long mtiles8 = ((g.M + QBM - 1) / QBM + 7) / 8 * 8;
    const dim3 grid((unsigned)(mtiles8 * (g.N / QBN)));
    MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)gemm256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(gemm256_kernel, grid, dim3(QNT), smem, stream, g);
    MAUA_HIP_CHECK(hipGetLastError());
    return MAUA_OK;
  }
  const size_t smem = std::max<size_t>((size_t)3 * STAGE, (size_t)DBM * DES);
  MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const long mtiles8 = ((g.M + DBM - 1) / DBM + 7) / 8 * 8;
  dim3 grid((unsigned)(mtiles8 * (g.N / DBN)));
  hipLaunchKernelGGL(gemm_dma_kernel, grid, dim3(DNT), smem, stream, g);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace maua
