// Fused modulated 3x3 convolution for gfx950 (CDNA4) as an MFMA implicit GEMM.
//
// Replaces (reference): ops.py:146-186 modulated_conv2d, :189-233 conv2d_resample (up=1 and up=2 branches),
// :87-114 upfirdn2d inside the up=2 branch, :65-84 bias_act that follows every layer
// (stylegan2.py:238-250).
//
// Formulation (shared weights, no per-sample weight tensor):
//   y[b,co,p] = act( d[b,co] * sum_{tap,ci} W[co,ci,tap] * (s[b,ci] * x[b,ci,p+tap]) + ns*noise[b,p] + bias[co] ) * gain
//   d[b,co]   = rsqrt( sum_ci s[b,ci]^2 * sum_tap W[co,ci,tap]^2 + 1e-8 )          (launch_styles)
// which equals the reference's per-sample-weight grouped convolution up to rounding order.
//
// up = 2: the reference's stride-2 transposed conv (pad 0) followed by the 4x4 FIR (pad 1, gain 4) is exactly
// four 3x3 correlations on the input grid, one per output parity (a,b), with phase kernels
//   Kp[a][b][ky][kx] = K[2ky+1-a][2kx+1-b],  K = full_conv2d(flip(W), 4f)   (6x6)
// prepared once by prep_weights_kernel; the conv kernel then only differs in where it stores.
//
// GEMM mapping per workgroup: M = a TH x TW patch of output-grid pixels of one sample (halo tile staged once per
// K-chunk in LDS and re-read for the 9 taps), N = BN output channels, K = 9 taps x Ci.
// MFMA: v_mfma_f32_32x32x16_bf16 (bf16 operands, f32 accumulate) or 4x v_mfma_f32_32x32x2_f32 (exact f32 parity
// mode) per 32-byte K-step; both read identical 16-byte-per-lane LDS fragments.
// LDS rows are 64 B of K + 16 B pad (80 B stride): ds_read_b128 fragment reads of 32 consecutive rows are
// bank-conflict-free (5*i mod 16 is a bijection).
#include <cstdlib>

#include "common.h"
#include "internal.h"

namespace maua {

// KCB = bytes of K (input channels) per LDS row chunk (template parameter: 64 or 128); the LDS row stride is KCB + 16

// An operand fragment is prepared once (Mma<T>::prep_w for the weight side, prep_x for the pixel side) and used in WM x WN steps:
// the identity for every type but the split-f32 one.
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  using FW = u32x4;
  using FX = u32x4;
  __device__ static __forceinline__ FW prep_w(const u32x4& v) { return v; }
  __device__ static __forceinline__ FX prep_x(const u32x4& v) { return v; }
  __device__ static __forceinline__ void step(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0,
                                                  0, 0);
  }
  // scale 8 bf16 by 8 f32 styles, round to nearest even
  __device__ static __forceinline__ u32x4 scale(const u32x4& v, const float* sv) {
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; k++)
      o[k] = pack2bf(bf2f((bf16_t)(v[k] & 0xffff)) * sv[2 * k], bf2f((bf16_t)(v[k] >> 16)) * sv[2 * k + 1]);
    return o;
  }
};
template <> struct Mma<f16_t> {
  using FW = u32x4;
  using FX = u32x4;
  __device__ static __forceinline__ FW prep_w(const u32x4& v) { return v; }
  __device__ static __forceinline__ FX prep_x(const u32x4& v) { return v; }
  __device__ static __forceinline__ void step(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
  }
  // scale 8 halves by 8 f32 styles, round to nearest even (the styles arrive pre-normalised, |s| <= 1: the product cannot overflow)
  __device__ static __forceinline__ u32x4 scale(const u32x4& v, const float* sv) {
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = pack2h(Fmt16<f16_t>::lo(v[k]) * sv[2 * k], Fmt16<f16_t>::hi(v[k]) * sv[2 * k + 1]);
    return o;
  }
};
template <> struct Mma<float> {
  using FW = u32x4;
  using FX = u32x4;
  __device__ static __forceinline__ FW prep_w(const u32x4& v) { return v; }
  __device__ static __forceinline__ FX prep_x(const u32x4& v) { return v; }
  // lane half h holds k = 8j+4h+e (e = 0..3): MFMA e consumes element e of both operands, so A and B see the
  // same K permutation and the sum is over the same set of products.
  __device__ static __forceinline__ void step(f32x16& acc, const u32x4& a, const u32x4& b) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], acc, 0, 0, 0);
  }
  __device__ static __forceinline__ u32x4 scale(const u32x4& v, const float* sv) {
    f32x4 f = __builtin_bit_cast(f32x4, v);
    f[0] *= sv[0]; f[1] *= sv[1]; f[2] *= sv[2]; f[3] *= sv[3];
    return __builtin_bit_cast(u32x4, f);
  }
};

// MAUA_F32_SPLIT: float32 tensors, products on the bf16 matrix cores.  x = hi + lo, hi = bf16(x) (round to nearest even),
// lo = bf16(x - hi) (x - hi is exact in f32).  Every aligned group of 8 floats (two 16-byte pieces: what the two lane halves of one
// k-step hold) is kept in LDS - and, for the weights, in HBM: launch_f32_split_inplace after launch_prep_weights - as
// [hi0 .. hi7 | lo0 .. lo7] bf16: piece 2 j = the eight hi parts, piece 2 j + 1 = the eight lo parts.  The pixel side is split ONCE
// per element when the halo tile is staged (scale() + one exchange between the two lanes that hold a group); a k-step of the
// convolution is then three full v_mfma_f32_32x32x16_bf16 over 16 real k:
//     w_hi x_hi  +  w_hi x_lo  +  w_lo x_hi          (w_lo x_lo, <= 2^-16 of the product, is dropped)
// 96 matrix-core cycles per 16 k where the exact path's eight v_mfma_f32_32x32x2_f32 take 512 (a first form with [hi x 4 | lo x 4] per
// piece spent one of four operand slots on zeros: 128 cycles).
__device__ __forceinline__ u32x4 f32_split4(const f32x4& f) {
  const uint32_t h0 = pack2bf(f[0], f[1]), h1 = pack2bf(f[2], f[3]);
  return u32x4{h0, h1, pack2bf(f[0] - __uint_as_float(h0 << 16), f[1] - __uint_as_float(h0 & 0xffff0000u)),
               pack2bf(f[2] - __uint_as_float(h1 << 16), f[3] - __uint_as_float(h1 & 0xffff0000u))};
}
// {hi x 4 | lo x 4} of this lane's piece q and of lane ^ 1's piece q ^ 1 -> this lane's piece of the group format above
__device__ __forceinline__ u32x4 f32_split_pair(const u32x4& v, int q) {
  const bool odd = q & 1;
  const uint32_t r0 = __shfl_xor(odd ? v[0] : v[2], 1), r1 = __shfl_xor(odd ? v[1] : v[3], 1);   // even lanes receive hi, odd lanes lo
  return odd ? u32x4{r0, r1, v[2], v[3]} : u32x4{v[0], v[1], r0, r1};
}
template <> struct Mma<f32s_t> {
  using FW = u32x4;
  using FX = u32x4;
  __device__ static __forceinline__ FW prep_w(const u32x4& v) { return v; }
  __device__ static __forceinline__ FX prep_x(const u32x4& v) { return v; }
  __device__ static __forceinline__ void step(f32x16& acc, const FW& w, const FX& x) {   // (one of the three products of a k-step)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
  }
  // the halo tile's 4 floats x 4 styles -> {hi x 4 | lo x 4} (the staging code then exchanges halves with the neighbouring lane)
  __device__ static __forceinline__ u32x4 scale(const u32x4& v, const float* sv) {
    f32x4 f = __builtin_bit_cast(f32x4, v);
    f[0] *= sv[0]; f[1] *= sv[1]; f[2] *= sv[2]; f[3] *= sv[3];
    return f32_split4(f);
  }
};
template <typename T> struct IsSplit { static constexpr bool value = false; };
template <> struct IsSplit<f32s_t> { static constexpr bool value = true; };

struct ConvGeom {
  int tw_log2, th;      // tile = th x (1 << tw_log2) pixels
  int tiles_x;          // tiles per row
  int hw2;              // tw + 2
  unsigned inv_hw2;     // ceil(2^20 / hw2) for the halo pixel -> (row, col) split
  int halo_px;          // (th+2)*(tw+2)
  int phases;           // up*up
};

// Pipeline: the K loop is a sequence of stages (input-channel chunk c, tap group g).  While stage s is multiplied
// out of LDS, the global loads of stage s+1 (weights, and the halo when a new chunk starts) are already in flight
// into registers; they are written to LDS between the two barriers that separate the stages.
// Accumulators are kept TRANSPOSED (MFMA called as W x X^T): a lane owns one pixel and 4-channel runs, so the
// epilogue needs one noise value per lane, float4 demod/bias loads, packs 4 channels per LDS store, and the tile
// leaves the workgroup as full 16-byte coalesced NHWC rows.
template <typename T, int WAVES_M, int WAVES_N, int WM, int WN, int TG, int KCB>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void modconv3x3_kernel(ConvArgs a, ConvGeom g) {
  constexpr int RS = KCB + 16;
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
  constexpr int KC = KCB / (int)sizeof(T);   // channels per K chunk
  constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-byte piece
  constexpr int WREGS = (TG * BN * (KCB / 16) + NT - 1) / NT;         // weight pieces per thread per stage
  constexpr int HREGS = ((BM == 128 ? 204 : 396) * (KCB / 16) + NT - 1) / NT;  // halo pieces per thread per chunk (max over tile shapes)
  // The 16-wave / 128-byte-chunk variant (the 64^2..256^2 conv1 layers) brings its weights in by LDS-direct loads: two
  // unpadded buffers (XOR-swizzled 16-byte pieces instead of the +16 row padding, which a linear fill cannot produce)
  // still fit next to the halo (57 + 2 x 48 KB).  No prefetch registers, no ds_writes for the weights.
  constexpr bool DMAW = KCB == 128 && NT == 1024 && TG == 3;
  constexpr int WBUF = TG * BN * KCB;                        // bytes of one weight stage (DMAW)
  constexpr int WDMA = DMAW ? WBUF / 1024 / (NT / 64) : 1;   // LDS-direct load instructions per wave per stage
  static_assert(!DMAW || WBUF % (1024 * (NT / 64)) == 0, "weight stage must split evenly over the waves");
  constexpr int ES = BN * (int)sizeof(T) + 16;               // epilogue tile row stride (bytes)
  constexpr int PPP = BN * (int)sizeof(T) / 16;              // 16-byte pieces per output pixel
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* wt = smem + g.halo_px * RS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int r = lane & 31, h = lane >> 5;
  const int tw = 1 << g.tw_log2;

  const int tile = blockIdx.x;
  const int tyi = tile / g.tiles_x, txi = tile - tyi * g.tiles_x;
  const int ty0 = tyi * g.th, tx0 = txi * tw;
  const int b = blockIdx.y;
  const int n0 = blockIdx.z * BN;        // virtual output channel n_v = phase * Co + co
  const int CoV = a.Co * g.phases;

  const T* xb = reinterpret_cast<const T*>(a.x) + (long)b * a.x_bstride;
  const T* wp = reinterpret_cast<const T*>(a.w);
  const float* sb = a.s + (long)b * a.Ci;

  // per-lane fragment base offsets (bytes)
  int offa[WM], offb[WN];
#pragma unroll
  for (int i = 0; i < WM; i++) {
    int m = (wm * WM + i) * 32 + r;
    int ty = m >> g.tw_log2, tx = m & (tw - 1);
    offa[i] = ((ty + 1) * g.hw2 + (tx + 1)) * RS + h * (IsSplit<T>::value ? 32 : 16);   // (split f32: a half owns a 32-byte group)
  }
#pragma unroll
  for (int j = 0; j < WN; j++)
    offb[j] = DMAW ? ((wn * WN + j) * 32 + r) * KCB : ((wn * WN + j) * 32 + r) * RS + h * (IsSplit<T>::value ? 32 : 16);
  const int swz = (r >> 1) & 7;  // DMAW: piece p of weight row n sits at piece p ^ ((n >> 1) & 7) (conflict-free b128 reads)

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; i++)
#pragma unroll
    for (int j = 0; j < WN; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  constexpr int PR = KCB / 16;  // 16-byte pieces per LDS row
  const int q = tid % PR;       // which piece of a row this thread stages (NT % PR == 0)
  const int rq = tid / PR;      // first row / halo pixel this thread stages

  // halo pixel -> global offset (elements), -1 outside the image; fixed for the whole K loop
  long hoff[HREGS];
#pragma unroll
  for (int i = 0; i < HREGS; i++) {
    int p = rq + i * (NT / PR);
    hoff[i] = -1;
    if (p < g.halo_px) {
      int py = (int)(((unsigned)p * g.inv_hw2) >> 20);
      int px = p - py * g.hw2;
      int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
      if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) hoff[i] = ((long)gy * a.W + gx) * (a.x_pstride ? a.x_pstride : a.Ci) + q * EPC;
    }
  }

  // Prefetch registers.  No lambdas / runtime indices here: everything below is fully unrolled so that these arrays
  // stay in VGPRs (a by-reference capture sent them to scratch and serialised the pipeline).
  u32x4 wreg[WREGS], hreg[HREGS];
  float sv[EPC];
  const T* wrow[WREGS];  // per-thread weight row base for tap 0 / chunk 0 of this N tile (row -> (t, n) is fixed)
  int wlds[WREGS];
#pragma unroll
  for (int i = 0; i < WREGS; i++) {
    int row = rq + i * (NT / PR);
    if (row >= TG * BN) row = TG * BN - 1;  // only when TG*BN*4 % NT != 0: duplicate a valid row, never stored
    int t = row / BN, n = row - t * BN;
    wrow[i] = wp + ((long)t * CoV + n0 + n) * a.Ci + q * EPC;
    wlds[i] = row * RS + q * 16;
  }
  const long tap_stride = (long)CoV * a.Ci;
  // DMAW: instruction ii (0 .. WBUF/1024) fills the 1 KB slot of rows 8 ii .. 8 ii + 7; lane l supplies row 8 ii + (l >> 3),
  // LDS piece l & 7 = logical piece (l & 7) ^ ((row >> 1) & 7); wave w issues ii = w + 16 j
  unsigned woff[WDMA];
  if constexpr (DMAW) {
#pragma unroll
    for (int j = 0; j < WDMA; j++) {
      const int row = 8 * (wave + (NT / 64) * j) + (lane >> 3);
      const int t = row / BN, n = row - t * BN;
      const int piece = (lane & 7) ^ ((row >> 1) & 7);
      woff[j] = (unsigned)((((long)t * CoV + n0 + n) * a.Ci + piece * EPC) * (long)sizeof(T));
    }
  }
#define MAUA_DMA_W(C0, TG0, BUF)                                                                          \
  {                                                                                                       \
    const T* wsrc_ = wp + ((long)(TG0) * tap_stride + (C0));                                              \
    _Pragma("unroll") for (int j = 0; j < WDMA; j++)                                                     \
      lds_dma_b128(wsrc_, woff[j], wt + (BUF) * WBUF + (wave + (NT / 64) * j) * 1024);                    \
  }

#define MAUA_LOAD_W(C0, TG0)                                                                              \
  _Pragma("unroll") for (int i = 0; i < WREGS; i++)                                                      \
      wreg[i] = *reinterpret_cast<const u32x4*>(wrow[i] + ((long)(TG0) * tap_stride + (C0)));
#define MAUA_LOAD_H(C0)                                                                                   \
  {                                                                                                       \
    _Pragma("unroll") for (int e = 0; e < EPC; e++) sv[e] = sb[(C0) + q * EPC + e];                       \
    _Pragma("unroll") for (int i = 0; i < HREGS; i++) {                                                  \
      hreg[i] = u32x4{0u, 0u, 0u, 0u};                                                                   \
      if (hoff[i] >= 0) hreg[i] = *reinterpret_cast<const u32x4*>(xb + hoff[i] + (C0)); \
    }                                                                                                     \
  }
#define MAUA_STORE_W()                                                                                    \
  _Pragma("unroll") for (int i = 0; i < WREGS; i++) {                                                    \
    if ((TG * BN * PR) % NT == 0 || rq + i * (NT / PR) < TG * BN)                                           \
      *reinterpret_cast<u32x4*>(wt + wlds[i]) = wreg[i];                                                  \
  }
#define MAUA_STORE_H()                                                                                    \
  _Pragma("unroll") for (int i = 0; i < HREGS; i++) {                                                    \
    int p = rq + i * (NT / PR);                                                                            \
    if (p < g.halo_px) {                                                                                   \
      u32x4 sv_ = Mma<T>::scale(hreg[i], sv);                                                              \
      if constexpr (IsSplit<T>::value) sv_ = f32_split_pair(sv_, q);                                       \
      *reinterpret_cast<u32x4*>(halo + p * RS + q * 16) = sv_;                                             \
    }                                                                                                      \
  }

  constexpr int NG = 9 / TG;
  const int n_chunks = a.Ci / KC;
  // DMAW ordering: a stage's weight request is issued BEFORE the register loads of the same stage boundary; vmcnt
  // retires in order, so consuming those registers (halo stages) - or an explicit vmcnt(0) where the request is the
  // wave's newest memory operation (the other stages) - means the wave's pieces have landed; the barrier that follows
  // makes every wave's pieces visible.
  if constexpr (DMAW) { MAUA_DMA_W(0, 0, 0) } else { MAUA_LOAD_W(0, 0) }
  MAUA_LOAD_H(0)
  int wbuf = 0;
  for (int c = 0; c < n_chunks; c++) {
    const int c0 = c * KC;
#pragma unroll
    for (int gi = 0; gi < NG; gi++) {
      __syncthreads();  // every wave is done reading the previous stage's LDS
      if (gi == 0) MAUA_STORE_H()
      if constexpr (DMAW) {
        if (gi != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        MAUA_STORE_W()
      }
      __syncthreads();
      // next stage's global loads fly while this stage is multiplied
      if (gi + 1 < NG) {
        if constexpr (DMAW) { MAUA_DMA_W(c0, (gi + 1) * TG, wbuf ^ 1) } else { MAUA_LOAD_W(c0, (gi + 1) * TG) }
      } else if (c + 1 < n_chunks) {
        if constexpr (DMAW) { MAUA_DMA_W(c0 + KC, 0, wbuf ^ 1) } else { MAUA_LOAD_W(c0 + KC, 0) }
        MAUA_LOAD_H(c0 + KC)
      }
      const char* wtb = DMAW ? wt + wbuf * WBUF : wt;
      wbuf ^= 1;
#pragma unroll
      for (int t = 0; t < TG; t++) {
        const int tap = gi * TG + t;                       // compile-time after unrolling
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const int tapoff = (dy * g.hw2 + dx) * RS;
        if constexpr (IsSplit<T>::value) {
          // split f32: a k-step covers 16 floats = two 32-byte groups (one per lane half), each as a hi piece and a lo piece
#pragma unroll
          for (int s2 = 0; s2 < KCB / 64; s2++) {
            u32x4 xh[WM], xl[WM], wh[WN], wl[WN];
#pragma unroll
            for (int i = 0; i < WM; i++) {
              xh[i] = *reinterpret_cast<const u32x4*>(halo + offa[i] + tapoff + s2 * 64);
              xl[i] = *reinterpret_cast<const u32x4*>(halo + offa[i] + tapoff + s2 * 64 + 16);
            }
#pragma unroll
            for (int j = 0; j < WN; j++) {
              if constexpr (DMAW) {
                wh[j] = *reinterpret_cast<const u32x4*>(wtb + t * BN * KCB + offb[j] + (((2 * (2 * s2 + h)) ^ swz) << 4));
                wl[j] = *reinterpret_cast<const u32x4*>(wtb + t * BN * KCB + offb[j] + (((2 * (2 * s2 + h) + 1) ^ swz) << 4));
              } else {
                wh[j] = *reinterpret_cast<const u32x4*>(wtb + t * BN * RS + offb[j] + s2 * 64);
                wl[j] = *reinterpret_cast<const u32x4*>(wtb + t * BN * RS + offb[j] + s2 * 64 + 16);
              }
            }
#pragma unroll
            for (int i = 0; i < WM; i++)
#pragma unroll
              for (int j = 0; j < WN; j++) {
                Mma<T>::step(acc[i][j], wh[j], xh[i]);
                Mma<T>::step(acc[i][j], wh[j], xl[i]);
                Mma<T>::step(acc[i][j], wl[j], xh[i]);
              }
          }
        } else
#pragma unroll
        for (int ks = 0; ks < KCB / 32; ks++) {
          typename Mma<T>::FX af[WM];
          typename Mma<T>::FW bf[WN];
#pragma unroll
          for (int i = 0; i < WM; i++) af[i] = Mma<T>::prep_x(*reinterpret_cast<const u32x4*>(halo + offa[i] + tapoff + ks * 32));
#pragma unroll
          for (int j = 0; j < WN; j++)
            bf[j] = Mma<T>::prep_w(DMAW ? *reinterpret_cast<const u32x4*>(wtb + t * BN * KCB + offb[j] + (((h + 2 * ks) ^ swz) << 4))
                                        : *reinterpret_cast<const u32x4*>(wtb + t * BN * RS + offb[j] + ks * 32));
#pragma unroll
          for (int i = 0; i < WM; i++)
#pragma unroll
            for (int j = 0; j < WN; j++) Mma<T>::step(acc[i][j], bf[j], af[i]);  // rows = channels, cols = pixels
        }
      }
    }
  }
#undef MAUA_LOAD_W
#undef MAUA_DMA_W
#undef MAUA_LOAD_H
#undef MAUA_STORE_W
#undef MAUA_STORE_H

  // ---- epilogue: demod, noise, bias, activation, gain, clamp -> LDS tile [pixel][virtual channel] -> coalesced
  // NHWC rows.  A virtual channel n_v decodes to (output parity, real channel): phase = n_v / Co, co = n_v % Co.
  const int Ho = a.H * a.up, Wo = a.W * a.up;
  const float* nb = a.noise ? a.noise + (long)b * a.noise_bstride : nullptr;
  __syncthreads();  // main-loop LDS is dead from here on
  char* epi = smem;
  // lrelu / linear (every synthesis layer): positively homogeneous, so the gain is folded into the coefficients,
  // act(acc*d + nz + b)*g == act(acc*(d*g) + (nz + b)*g); other activations keep the reference order of operations
  const float alpha = a.act == MAUA_ACT_LINEAR ? 1.f : a.alpha;
  const bool fast = (a.act == MAUA_ACT_LRELU || a.act == MAUA_ACT_LINEAR) && alpha >= 0.f && alpha <= 1.f && a.gain > 0.f && !a.prelu;
  const float cl = a.clamp >= 0.f ? a.clamp : 3.0e38f;
  // Per N tile the global operands are requested as groups (the WM noise values; the demodulation / bias / next-layer-style vectors
  // of the tile's 16 channels), each group one round trip; loaded inside the activation's branches every load was waited for alone.
#pragma unroll
  for (int j = 0; j < WN; j++) {
    // a 32-channel group never straddles an output parity (Co % 32 == 0): phase and channel base are wave-uniform
    const int nvb = n0 + (wn * WN + j) * 32;
    const int ph = g.phases == 1 ? 0 : (nvb >= a.Co) + (nvb >= 2 * a.Co) + (nvb >= 3 * a.Co);
    const int cob = nvb - ph * a.Co;
    const int pa = ph >> (a.up - 1), pb = ph & (a.up - 1);
    float nzr[WM];
#pragma unroll
    for (int i = 0; i < WM; i++) nzr[i] = 0.f;
    if (nb) {
#pragma unroll
      for (int i = 0; i < WM; i++) {
        const int m = (wm * WM + i) * 32 + r;
        const int gy = ty0 + (m >> g.tw_log2), gx = tx0 + (m & (tw - 1));
        if (gy < a.H && gx < a.W) nzr[i] = nb[(long)(gy * a.up + pa) * Wo + gx * a.up + pb];
      }
#pragma unroll
      for (int i = 0; i < WM; i++) nzr[i] *= a.noise_strength * (a.noise_scale ? a.noise_scale[b] : 1.f);
    }
#pragma unroll
    for (int q0 = 0; q0 < 4; q0 += 2) {   // two register groups (8 channels) per round trip
      float4 dq[2], bq[2], sq[2];
#pragma unroll
      for (int q = 0; q < 2; q++) {
        dq[q] = sq[q] = make_float4(1.f, 1.f, 1.f, 1.f);
        bq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const int co0 = cob + 8 * q0 + 4 * h;
      if (a.d) {
#pragma unroll
        for (int q = 0; q < 2; q++) dq[q] = *reinterpret_cast<const float4*>(a.d + (long)b * a.Co + co0 + 8 * q);
      }
      if (a.bias) {
#pragma unroll
        for (int q = 0; q < 2; q++) bq[q] = *reinterpret_cast<const float4*>(a.bias + co0 + 8 * q);
      }
      if (a.out_scale) {  // the layer that reads these features wants them pre-multiplied by its styles
#pragma unroll
        for (int q = 0; q < 2; q++) sq[q] = *reinterpret_cast<const float4*>(a.out_scale + (long)b * a.Co + co0 + 8 * q);
      }
#pragma unroll
      for (int i = 0; i < WM; i++) {
        const int m = (wm * WM + i) * 32 + r;
        const int ty = m >> g.tw_log2, tx = m & (tw - 1);
        const int gy = ty0 + ty, gx = tx0 + tx;
        const bool inside = gy < a.H && gx < a.W;
        const float nz = nzr[i];
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const int qd = q0 + q;
          const int nl = (wn * WN + j) * 32 + 8 * qd + 4 * h;  // first of 4 consecutive virtual channels (tile-local)
          const int co = co0 + 8 * q;
          const float dd[4] = {dq[q].x, dq[q].y, dq[q].z, dq[q].w}, bb[4] = {bq[q].x, bq[q].y, bq[q].z, bq[q].w};
          float v[4];
          if (fast) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
              float t = fmaf(acc[i][j][qd * 4 + k], dd[k] * a.gain, (nz + bb[k]) * a.gain);
              t = fmaxf(t, t * alpha);
              v[k] = __builtin_amdgcn_fmed3f(t, -cl, cl);
            }
          } else if (a.prelu) {  // PReLU: per-channel slope on the negative side
            const float4 pv = *reinterpret_cast<const float4*>(a.prelu + co);
            const float pp[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
              float t = acc[i][j][qd * 4 + k] * dd[k] + nz + bb[k];
              t = (t >= 0.f ? t : t * pp[k]) * a.gain;
              if (a.clamp >= 0.f) t = fminf(fmaxf(t, -a.clamp), a.clamp);
              v[k] = t;
            }
          } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
              float t = activate(acc[i][j][qd * 4 + k] * dd[k] + nz + bb[k], a.act, a.alpha) * a.gain;
              if (a.clamp >= 0.f) t = fminf(fmaxf(t, -a.clamp), a.clamp);
              v[k] = t;
            }
          }
          if (a.out_scale) { v[0] *= sq[q].x; v[1] *= sq[q].y; v[2] *= sq[q].z; v[3] *= sq[q].w; }
          if (a.res && inside) {  // residual connection (RRDB blocks): added after activation, gain and clamp
            const T* rp = reinterpret_cast<const T*>(a.res) + (long)b * a.res_bstride +
                          ((long)gy * a.W + gx) * a.res_pstride + co;
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] += Elem<T>::load(rp + k);
            if (a.res2) {  // second residual on the value as stored (rounded to T): y = res_gain * y + res2
              const T* rp2 = reinterpret_cast<const T*>(a.res2) + (long)b * a.res2_bstride +
                             ((long)gy * a.W + gx) * a.res2_pstride + co;
#pragma unroll
              for (int k = 0; k < 4; k++) {
                float t = v[k];
                if constexpr (sizeof(T) == 2) t = Fmt16<T>::round(t);
                v[k] = a.res_gain * t + Elem<T>::load(rp2 + k);
              }
            }
          }
          char* dst = epi + m * ES + nl * (int)sizeof(T);
          if constexpr (sizeof(T) == 2)
            *reinterpret_cast<uint2*>(dst) = make_uint2(Fmt16<T>::pack2(v[0], v[1]), Fmt16<T>::pack2(v[2], v[3]));
          else
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
  }
  __syncthreads();
  // ---- fused toRGB + upsampled skip (conv1 layers whose Co == BN, bf16): [32 px x Co] x [Co x 3(+3)] on the matrix
  // cores, the activated bf16 outputs are the operand straight from the epilogue tile (the block's features are not
  // re-read by a toRGB pass).  Rows 0..2 of the weight operand = bf16(hi) part of the pre-modulated RGB weights, rows
  // 8..10 = the bf16 remainder (w = hi + lo to ~2^-17); rgb[c] = acc[row c] + acc[row 8 + c], both of which land in
  // the h == 0 lane of the pixel.  One wave per 32-pixel sub-tile.
  if constexpr (sizeof(T) == 2) {
    if (a.rgb_out && wave < BM / 32) {
      const int c_rgb = r < 3 ? r : (r >= 8 && r < 11 ? r - 8 : -1);
      const int mrow = wave * 32 + r;
      const int y = ty0 + (mrow >> g.tw_log2), x = tx0 + (mrow & (tw - 1));
      const bool px_ok = h == 0 && y < a.H && x < a.W;
      // This phase is the workgroup's tail (nothing else hides it): every global load it needs is requested up
      // front - the 2x2 taps of the previous image and all weight pieces - so that one memory latency is exposed
      // instead of one per k-step plus one for the skip.
      // upsample2d (zero-insert x2, pad (2,1,2,1), 4x4 FIR) in its 2x2 form: only the taps whose parity hits a real
      // sample are non-zero -> rows {iy0, iy0+1}, cols {ix0, ix0+1}, filter index u = 2*iy - y + 2 (same products,
      // same u-major order as the 16-tap correlation)
      float pvv[3][4], pf[4];
#pragma unroll
      for (int t4 = 0; t4 < 4; t4++) { pf[t4] = 0.f; pvv[0][t4] = pvv[1][t4] = pvv[2][t4] = 0.f; }
      if (a.rgb_prev && px_ok) {
        const int Hp = a.H >> 1, Wp = a.W >> 1;
        const float* pv = a.rgb_prev + (long)b * 3 * Hp * Wp;
        const int iy0 = (y - 1) >> 1, ix0 = (x - 1) >> 1;
#pragma unroll
        for (int dy = 0; dy < 2; dy++) {
          const int iy = iy0 + dy, u = 2 * iy - y + 2;
          const bool oky = iy >= 0 && iy < Hp;
#pragma unroll
          for (int dx = 0; dx < 2; dx++) {
            const int ix = ix0 + dx, v = 2 * ix - x + 2;
            const bool ok = oky && ix >= 0 && ix < Wp;
            const bool uh = u == 1 || u == 2, vh = v == 1 || v == 2;  // fir[u][v] takes 3 distinct values
            pf[dy * 2 + dx] = !ok ? 0.f : uh ? (vh ? a.fir[5] : a.fir[4]) : (vh ? a.fir[1] : a.fir[0]);
            const unsigned o = ok ? (unsigned)(iy * Wp + ix) : 0u;
            pvv[0][dy * 2 + dx] = pv[o];
            pvv[1][dy * 2 + dx] = pv[(unsigned)(Hp * Wp) + o];
            pvv[2][dy * 2 + dx] = pv[2u * (unsigned)(Hp * Wp) + o];
          }
        }
      }
      float4 wsrc[BN / 16][2];
      const float* wbase = a.rgb_wmod + ((long)b * 3 + (c_rgb >= 0 ? c_rgb : 0)) * a.Co + 8 * h;
#pragma unroll
      for (int ks = 0; ks < BN / 16; ks++) {
        wsrc[ks][0] = *reinterpret_cast<const float4*>(wbase + ks * 16);
        wsrc[ks][1] = *reinterpret_cast<const float4*>(wbase + ks * 16 + 4);
      }
      const float rb0 = a.rgb_bias[0], rb1 = a.rgb_bias[1], rb2 = a.rgb_bias[2];
      const float row_mask = c_rgb >= 0 ? 1.f : 0.f, lo_mask = r >= 8 ? 1.f : 0.f;
      f32x16 racc;
#pragma unroll
      for (int e = 0; e < 16; e++) racc[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < BN / 16; ks++) {
        // branch-free on purpose (per-lane masks as factors): with conditionals the compiler wraps every element in
        // its own exec-mask branch and sinks the loads into them
        const float4 w0 = wsrc[ks][0], w1 = wsrc[ks][1];
        float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        u32x4 wf;
#pragma unroll
        for (int k = 0; k < 8; k++) wv[k] = (wv[k] - lo_mask * Fmt16<T>::round(wv[k])) * row_mask;  // hi rows: w, lo rows: w - T(w)
#pragma unroll
        for (int k = 0; k < 4; k++) wf[k] = Fmt16<T>::pack2(wv[2 * k], wv[2 * k + 1]);
        const u32x4 av = *reinterpret_cast<const u32x4*>(epi + mrow * ES + (ks * 16 + 8 * h) * 2);
        Mma<T>::step(racc, wf, av);
      }
      if (px_ok) {
        float o3[3] = {racc[0] + racc[4] + rb0, racc[1] + racc[5] + rb1, racc[2] + racc[6] + rb2};
#pragma unroll
        for (int c = 0; c < 3; c++)
          if (a.rgb_clamp >= 0.f) o3[c] = fminf(fmaxf(o3[c], -a.rgb_clamp), a.rgb_clamp);
        const unsigned HWl = (unsigned)(a.H * a.W);
        if (a.rgb_prev) {
#pragma unroll
          for (int c = 0; c < 3; c++) {
            float u = 0.f;
#pragma unroll
            for (int t4 = 0; t4 < 4; t4++) u += pvv[c][t4] * pf[t4];
            o3[c] = u + o3[c];
          }
        }
        float* ob = a.rgb_out + (long)b * 3 * HWl + (unsigned)(y * a.W + x);
        ob[0] = o3[0]; ob[HWl] = o3[1]; ob[2 * HWl] = o3[2];
      }
    }
  }
  const long ypix = a.y_pstride ? a.y_pstride : a.Co;
  char* yb = reinterpret_cast<char*>(a.y) +
             ((long)b * (a.y_bstride ? a.y_bstride : (long)Ho * Wo * a.Co) + a.y_coff) * (long)sizeof(T);
  // (with the fused toRGB the waves that did not take part in it write the whole tile meanwhile)
  const bool rgb_split = sizeof(T) == 2 && a.rgb_out && NT > (BM / 32) * 64;
  const int ro_tid = rgb_split ? tid - (BM / 32) * 64 : tid, ro_nt = rgb_split ? NT - (BM / 32) * 64 : NT;
  if (ro_tid >= 0)
  for (int p = ro_tid; p < BM * PPP; p += ro_nt) {
    const int m = p / PPP, pc = p - m * PPP;
    const int ty = m >> g.tw_log2, tx = m & (tw - 1);
    const int gy = ty0 + ty, gx = tx0 + tx;
    if (gy < a.H && gx < a.W) {
      const int nv = n0 + pc * EPC;
      const int ph = g.phases == 1 ? 0 : (nv >= a.Co) + (nv >= 2 * a.Co) + (nv >= 3 * a.Co);
      const int co = nv - ph * a.Co;
      const int pa = ph >> (a.up - 1), pb = ph & (a.up - 1);
      const long pix = (long)(gy * a.up + pa) * Wo + gx * a.up + pb;
      *reinterpret_cast<uint4*>(yb + (pix * ypix + co) * (long)sizeof(T)) =
          *reinterpret_cast<const uint4*>(epi + m * ES + pc * 16);
    }
  }
}

template <typename T, int WAVES_M, int WAVES_N, int WM, int WN, int TG, int KCB>
static int launch_variant(hipStream_t stream, const ConvArgs& a) {
  constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32, NT = WAVES_M * WAVES_N * 64, RS = KCB + 16;
  MAUA_REQUIRE(a.Ci % (KCB / (int)sizeof(T)) == 0, "modconv3x3: Ci must be a multiple of the K chunk (pad channels)");
  ConvGeom g;
  int tw = a.W > 16 ? 32 : a.W > 8 ? 16 : a.W > 4 ? 8 : 4;
  g.tw_log2 = tw == 32 ? 5 : tw == 16 ? 4 : tw == 8 ? 3 : 2;
  g.th = BM / tw;
  g.tiles_x = cdiv(a.W, tw);
  int tiles_y = cdiv(a.H, g.th);
  g.hw2 = tw + 2;
  g.inv_hw2 = ((1u << 20) + g.hw2 - 1) / g.hw2;
  g.halo_px = (g.th + 2) * g.hw2;
  g.phases = a.up * a.up;
  MAUA_REQUIRE(g.halo_px <= (((BM == 128 ? 204 : 396) * (KCB / 16) + NT - 1) / NT) * (NT / (KCB / 16)),
               "modconv3x3: halo does not fit the prefetch registers");
  MAUA_REQUIRE(!a.rgb_out || (sizeof(T) == 2 && a.up == 1 && a.Co == BN && a.rgb_wmod && a.rgb_bias),
               "modconv3x3: fused toRGB needs bf16, up == 1 and all output channels in one N tile");
  constexpr bool DMAW = KCB == 128 && NT == 1024 && TG == 3;  // as in the kernel
  size_t smem_main = (size_t)g.halo_px * RS + (DMAW ? (size_t)2 * TG * BN * KCB : (size_t)TG * BN * RS);
  size_t smem_epi = (size_t)BM * (BN * sizeof(T) + 16);
  size_t smem = std::max(smem_main, smem_epi);
  MAUA_REQUIRE(smem <= 160 * 1024, "modconv3x3: LDS budget exceeded");
  MAUA_REQUIRE(a.B <= 65535 && (a.Co * g.phases / BN) <= 65535, "modconv3x3: grid too large");
  auto kern = modconv3x3_kernel<T, WAVES_M, WAVES_N, WM, WN, TG, KCB>;
  if (smem > 64 * 1024)
    MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  // x = spatial tiles, y = sample, z = tile of virtual output channels (phase * Co + co): consecutive workgroups
  // share one weight slice (L2-resident per XCD) while they stream different activation tiles
  dim3 grid(g.tiles_x * tiles_y, a.B, a.Co * g.phases / BN);
  hipLaunchKernelGGL(kern, grid, dim3(NT), smem, stream, a, g);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

template <typename T>
static int launch_modconv_t(hipStream_t stream, const ConvArgs& a) {
  MAUA_REQUIRE(a.Ci % 32 == 0, "modconv3x3: Ci must be a multiple of 32 (pad channels)");
  MAUA_REQUIRE(a.Co % 32 == 0, "modconv3x3: Co must be a multiple of 32 (pad channels)");
  MAUA_REQUIRE(a.up == 1 || a.up == 2, "modconv3x3: up must be 1 or 2");
  if (a.B == 0) return MAUA_OK;
  const int cov = a.Co * a.up * a.up;  // virtual output channels: up-layers carry their 4 parities in N
  // 8 / 16 waves per workgroup: measured 1.15x over 4 waves at the same tile (more waves hide the two barriers
  // per stage); the 256-pixel tile pays off once a sample has >= 16 of them
  // K chunk: 128 bytes (fewer, longer stages) measured 1.1-1.2x over 64 except on the 32^2 layers
  const bool k128 = a.Ci % (128 / (int)sizeof(T)) == 0;
  // tiny up=1 layers (4^2..16^2) are bound by the latency of streaming the weights through few workgroups: 32-channel
  // N tiles give 4x the workgroups and 9-tap stages a third of the dependent stages (measured 0.058 -> 0.047 ms at 4^2,
  // 0.065 -> 0.049 at 16^2; up=2 layers would re-stage the halo for 64 virtual-channel tiles and lose)
  if (a.up == 1 && a.H * a.W <= 256 && cov % 128 == 0) return launch_variant<T, 4, 1, 2, 1, 9, 64>(stream, a);
  if (cov % 128 == 0 && a.H * a.W >= 4096 && k128) return launch_variant<T, 4, 4, 2, 1, 3, 128>(stream, a);
  if (cov % 128 == 0 && a.H * a.W >= 4096) return launch_variant<T, 4, 4, 2, 1, 3, 64>(stream, a);
  // 16 x 16 inputs (the 16^2 -> 32^2 up-layer of the synthesis network): the whole image as one 256-pixel tile of 16 waves with
  // 128-byte K chunks - 0.70 -> 0.59 ms at B = 128 against the 128-pixel / 64-byte tile (round 5; 128 px x 128 B: 0.86, 256 x 64 B: 0.69)
  if (cov % 128 == 0 && a.H * a.W == 256 && k128) return launch_variant<T, 4, 4, 2, 1, 3, 128>(stream, a);
  if (cov % 128 == 0 && a.H * a.W < 256 && k128) return launch_variant<T, 2, 4, 2, 1, 3, 128>(stream, a);
  if (cov % 128 == 0) return launch_variant<T, 2, 4, 2, 1, 3, 64>(stream, a);
  // 64 virtual channels on large maps (the secondary diffusion model's and the VGG perceptors' 64-channel layers at 256^2 / 128^2): the
  // same 256 x 64 tile on 8 waves with 3-tap stages - secondary forward + VJP 8.12 -> 7.16 ms (split f32), 4.41 -> 3.77 (bf16) at batch 16
  static const bool w4 = getenv("MAUA_CONV64_4W") != nullptr;
  if (!w4 && cov % 64 == 0 && a.H * a.W >= 4096) return launch_variant<T, 4, 2, 2, 1, 3, 64>(stream, a);
  if (cov % 64 == 0) return launch_variant<T, 4, 1, 2, 2, 9, 64>(stream, a);
  return launch_variant<T, 4, 1, 2, 1, 9, 64>(stream, a);
}

// conv1 layers whose toRGB can ride on the epilogue tile: the variants with a 128-channel N tile (launch_modconv_t)
bool modconv_rgb_fusable(int dtype, int Ci, int Co, int up, int H, int W) {
  return (dtype == MAUA_BF16 || dtype == MAUA_F16) && up == 1 && Co == 128 && Ci % 32 == 0 && H * W >= 4096 && (H % 2) == 0 && (W % 2) == 0;
}

int launch_modconv3x3(hipStream_t stream, int dtype, const ConvArgs& a) {
  if (dtype == MAUA_BF16) return launch_modconv_t<bf16_t>(stream, a);
  if (dtype == MAUA_F16) return launch_modconv_t<f16_t>(stream, a);
  if (dtype == MAUA_F32) return launch_modconv_t<float>(stream, a);
  if (dtype == MAUA_F32_SPLIT) return launch_modconv_t<f32s_t>(stream, a);
  return fail("modconv3x3: unsupported dtype");
}

// a prepared float32 weight buffer (launch_prep_weights, MAUA_F32) -> the split form above, in place: n floats, n % 8 == 0 (K, the
// innermost dimension, is a multiple of 32 channels)
__global__ __launch_bounds__(256) void f32_split_inplace_kernel(float* __restrict__ w, long n8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const u32x4 a = f32_split4(*reinterpret_cast<const f32x4*>(w + 8 * i)), b = f32_split4(*reinterpret_cast<const f32x4*>(w + 8 * i + 4));
  *reinterpret_cast<u32x4*>(w + 8 * i) = u32x4{a[0], a[1], b[0], b[1]};        // hi of floats 0 .. 7
  *reinterpret_cast<u32x4*>(w + 8 * i + 4) = u32x4{a[2], a[3], b[2], b[3]};    // lo of floats 0 .. 7
}
int launch_f32_split_inplace(hipStream_t stream, void* w, long n) {
  MAUA_REQUIRE(w && n >= 0 && n % 8 == 0, "f32_split_inplace: bad argument");
  if (n == 0) return MAUA_OK;
  hipLaunchKernelGGL(f32_split_inplace_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, stream, (float*)w, n / 8);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// ------------------------------------------------------------------------------------------------ weight prep
// f32 [Co][Ci][k][k] -> T [k*k][phases][Cop][Cip] (zero padded; the phase is part of a 'virtual' output-channel
// index n_v = phase*Cop + co, so one workgroup can produce several output parities from one input halo) and
// Wsq[co][ci] = sum_taps W^2.
// up == 2 (k == 3): phase kernels from K = full_conv2d(flip(W), 4f), f = outer([1,3,3,1])/64.
template <typename T>
__global__ __launch_bounds__(256) void prep_weights_kernel(const float* __restrict__ w, T* __restrict__ wt,
                                                           float* __restrict__ wsq, int Co, int Ci, int k, int up,
                                                           int flip, int Cop, int Cip) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)Cop * Cip) return;
  int co = (int)(idx / Cip), ci = (int)(idx - (long)co * Cip);
  const int kk = k * k;
  float wv[9];
  bool real = co < Co && ci < Ci;
  float sq = 0.f;
  for (int t = 0; t < kk; t++) {
    wv[t] = real ? w[((long)co * Ci + ci) * kk + t] : 0.f;
    sq += wv[t] * wv[t];
  }
  if (wsq && real) wsq[(long)co * Ci + ci] = sq;
  const long plane = (long)Cop * Cip;
  if (up == 1) {
    for (int t = 0; t < kk; t++) Elem<T>::store(wt + (long)t * plane + idx, wv[t]);
    return;
  }
  // up == 2, k == 3
  const float g4[4] = {0.25f, 0.75f, 0.75f, 0.25f};  // 2 * [1,3,3,1]/8 per axis  (4f = outer(g4, g4))
  float K[6][6];
  for (int u = 0; u < 6; u++)
    for (int v = 0; v < 6; v++) {
      float s = 0.f;
      for (int i = 0; i < 3; i++) {
        int fu = u - i;
        if (fu < 0 || fu > 3) continue;
        for (int j = 0; j < 3; j++) {
          int fv = v - j;
          if (fv < 0 || fv > 3) continue;
          // A = flip(W) in-tree (no flip before the transposed conv), A = W under nv_compat
          float aij = flip ? wv[i * 3 + j] : wv[(2 - i) * 3 + (2 - j)];
          s += aij * g4[fu] * g4[fv];
        }
      }
      K[u][v] = s;
    }
  for (int pa = 0; pa < 2; pa++)
    for (int pb = 0; pb < 2; pb++)
      for (int ky = 0; ky < 3; ky++)
        for (int kx = 0; kx < 3; kx++) {
          int ph = pa * 2 + pb, t = ky * 3 + kx;
          Elem<T>::store(wt + ((long)t * 4 + ph) * plane + idx, K[2 * ky + 1 - pa][2 * kx + 1 - pb]);
        }
}

size_t prepped_weight_elems(int k, int up, int Cop, int Cip) { return (size_t)up * up * k * k * Cop * Cip; }

int launch_prep_weights(hipStream_t stream, int dtype, const float* w, void* wt, float* wsq, int Co, int Ci, int k,
                        int up, int flip, int Cop, int Cip) {
  MAUA_REQUIRE(k == 1 || k == 3, "prep_weights: kernel size must be 1 or 3");
  MAUA_REQUIRE(up == 1 || (up == 2 && k == 3), "prep_weights: up=2 needs a 3x3 kernel");
  long n = (long)Cop * Cip;
  dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == MAUA_BF16)
    hipLaunchKernelGGL(prep_weights_kernel<bf16_t>, grid, dim3(256), 0, stream, w, (bf16_t*)wt, wsq, Co, Ci, k, up, flip,
                       Cop, Cip);
  else if (dtype == MAUA_F16)
    hipLaunchKernelGGL(prep_weights_kernel<f16_t>, grid, dim3(256), 0, stream, w, (f16_t*)wt, wsq, Co, Ci, k, up, flip,
                       Cop, Cip);
  else if (dtype == MAUA_F32)
    hipLaunchKernelGGL(prep_weights_kernel<float>, grid, dim3(256), 0, stream, w, (float*)wt, wsq, Co, Ci, k, up, flip,
                       Cop, Cip);
  else
    return fail("prep_weights: unsupported dtype");
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// ------------------------------------------------------------------------------------------------ styles / demod
// Two launches per batch for ALL layers (17 conv + 9 toRGB at 1024^2): (1) s = affine(w) (stylegan2.py:48-58,:230,
// :269), (2) demod coefficients (ops.py:168-171) / pre-modulated toRGB weights.  Both are small GEMMs
// out[row][sample] = sum_k M[row][k] * vec[sample][k] (rows = channels, <= 32 samples per pass), done with the exact
// f32 MFMA (v_mfma_f32_32x32x2_f32): a workgroup owns 32 matrix rows of one layer for all samples, its 4 waves split
// K and reduce through LDS.  A sample is a column of the MFMA, so its arithmetic does not depend on where it sits in
// the batch (frames stay bit-identical under any batching / sharding).
//   lane (r = l & 31, h = l >> 5) feeds M[row0 + r][k8 + 4h + v] and vec[n = r][k8 + 4h + v] to MFMA number v of
//   the 8-column block k8: one float4 load per operand per lane per block.
template <typename VecLoad, typename Store>
__device__ __forceinline__ void rows_times_samples(const float* __restrict__ M, int K, int row0, int rows_valid, int B,
                                                   VecLoad vec_load, Store store, float* red /* LDS [4][16][64] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int kq = (K / 4 + 7) / 8 * 8;  // K columns per wave, a multiple of the 8-column block
  const int k_lo = wave * kq, k_hi = min(K, k_lo + kq);
  const bool row_ok = r < rows_valid;
  const float* mrow = M + (long)(row0 + (row_ok ? r : 0)) * K;
  for (int b0 = 0; b0 < B; b0 += 32) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    const bool smp_ok = b0 + r < B;
#pragma unroll 4
    for (int k8 = k_lo; k8 < k_hi; k8 += 8) {
      const int k = k8 + 4 * h;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row_ok && k < K) a = *reinterpret_cast<const float4*>(mrow + k);
      if (smp_ok && k < K) v = vec_load(b0 + r, k);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, v.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, v.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, v.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, v.w, acc, 0, 0, 0);
    }
    // K-split reduction, fixed order wave 0 + 1 + 2 + 3
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; e++) red[(wave * 16 + e) * 64 + lane] = acc[e];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const float t = ((red[e * 64 + lane] + red[(16 + e) * 64 + lane]) + red[(32 + e) * 64 + lane]) + red[(48 + e) * 64 + lane];
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;  // D layout: lane = column (sample), 16 rows per lane
        if (row < rows_valid && smp_ok) store(b0 + r, row0 + row, t);
      }
    }
  }
}

__global__ __launch_bounds__(256) void styles_affine_kernel(const StyleLayer* __restrict__ layers,
                                                            const float* __restrict__ ws, int num_ws, int w_dim, int B) {
  __shared__ float red[4 * 16 * 64];
  const StyleLayer L = layers[blockIdx.x];
  const int c0 = blockIdx.y * 32;
  if (c0 >= L.Cs) return;
  const float wgain = rsqrtf((float)w_dim);
  const float* wbase = ws + (long)L.w_index * w_dim;
  const long wstride = (long)num_ws * w_dim;
  rows_times_samples(
      L.affine_w, w_dim, c0, min(32, L.Cin - c0), B,
      [&](int b, int k) { return *reinterpret_cast<const float4*>(wbase + b * wstride + k); },
      [&](int b, int ci, float t) { L.s[(long)b * L.Cs + ci] = (t * wgain + L.affine_b[ci]) * L.scale; }, red);
  // channel padding of the styles buffer
  for (int i = threadIdx.x; i < B * 32; i += blockDim.x) {
    const int b = i >> 5, ci = c0 + (i & 31);
    if (ci >= L.Cin && ci < L.Cs) L.s[(long)b * L.Cs + ci] = 0.f;
  }
}

__global__ __launch_bounds__(256) void styles_demod_kernel(const StyleLayer* __restrict__ layers, int B) {
  __shared__ float red[4 * 16 * 64];
  const StyleLayer L = layers[blockIdx.x];
  if (L.d) {
    const int c0 = blockIdx.y * 32;
    if (c0 >= L.Cd) return;
    rows_times_samples(
        L.wsq, L.Cin, c0, min(32, L.Co - c0), B,
        [&](int b, int k) {
          const float4 v = *reinterpret_cast<const float4*>(L.s + (long)b * L.Cs + k);
          return make_float4(v.x * v.x, v.y * v.y, v.z * v.z, v.w * v.w);
        },
        [&](int b, int co, float t) { L.d[(long)b * L.Cd + co] = rsqrtf(t + 1e-8f); }, red);
    for (int i = threadIdx.x; i < B * 32; i += blockDim.x) {
      const int b = i >> 5, co = c0 + (i & 31);
      if (co >= L.Co && co < L.Cd) L.d[(long)b * L.Cd + co] = 0.f;
    }
  } else if (L.wmod) {
    const int n = B * 3 * L.Cin;
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x) {
      const int b = i / (3 * L.Cin), r = i - b * 3 * L.Cin;
      L.wmod[i] = L.wrgb[r] * L.s[(long)b * L.Cs + (r % L.Cin)];
    }
  }
}

// ---- ops.py:161-165, the FP16 pre-normalisation of a demodulated convolution ("Pre-normalize inputs to avoid FP16 overflow"):
//   weight = weight / (amax |weight| over (ci, ky, kx) per output channel * sqrt(Ci k k));  styles = styles / amax |styles| per sample
// Demodulation makes the layer's output invariant to both factors (up to its 1e-8); what they buy is range: with |s| <= 1 the
// modulated activation x * s, which this library rounds to the network dtype, cannot leave the half range when x did not.
// One workgroup per output channel
__global__ __launch_bounds__(256) void f16_prenorm_weights_kernel(const float* __restrict__ w, float* __restrict__ out, int row,
                                                                  float rsqrt_fan) {
  __shared__ float red[4];
  const float* src = w + (long)blockIdx.x * row;
  float m = 0.f;
  for (int i = threadIdx.x; i < row; i += 256) m = fmaxf(m, fabsf(src[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float k = rsqrt_fan / m;   // (an all-zero channel divides by zero, as the reference does)
  for (int i = threadIdx.x; i < row; i += 256) out[(long)blockIdx.x * row + i] = src[i] * k;
}
int launch_f16_prenorm_weights(hipStream_t stream, const float* w, float* out, int Co, int Ci, int kk) {
  if (Co == 0) return MAUA_OK;
  hipLaunchKernelGGL(f16_prenorm_weights_kernel, dim3(Co), dim3(256), 0, stream, w, out, Ci * kk, 1.f / sqrtf((float)(Ci * kk)));
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}
// one wave per (layer, sample): s[b][0 .. Cin) /= max |s[b][.]|, for the layers that demodulate (L.d != NULL)
__global__ __launch_bounds__(64) void f16_prenorm_styles_kernel(const StyleLayer* __restrict__ layers, int B) {
  const StyleLayer L = layers[blockIdx.x];
  if (!L.d) return;
  float* sp = L.s + (long)blockIdx.y * L.Cs;
  float m = 0.f;
  for (int i = threadIdx.x; i < L.Cin; i += 64) m = fmaxf(m, fabsf(sp[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  for (int i = threadIdx.x; i < L.Cin; i += 64) sp[i] = sp[i] / m;
}
// the same for one plain styles tensor [B][Cs] (operator-level entry point)
__global__ __launch_bounds__(64) void f16_prenorm_styles_plain_kernel(float* __restrict__ s, int Cs, int Cin) {
  float* sp = s + (long)blockIdx.x * Cs;
  float m = 0.f;
  for (int i = threadIdx.x; i < Cin; i += 64) m = fmaxf(m, fabsf(sp[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  for (int i = threadIdx.x; i < Cin; i += 64) sp[i] = sp[i] / m;
}
int launch_f16_prenorm_styles(hipStream_t stream, float* s, int B, int Cs, int Cin) {
  if (B == 0) return MAUA_OK;
  hipLaunchKernelGGL(f16_prenorm_styles_plain_kernel, dim3(B), dim3(64), 0, stream, s, Cs, Cin);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int launch_styles(hipStream_t stream, const StyleLayer* layers_dev, int n_layers, const float* ws, int num_ws,
                  int w_dim, int B, int max_channels, int f16_prenorm) {
  if (B == 0 || n_layers == 0) return MAUA_OK;
  MAUA_REQUIRE(w_dim % 4 == 0 && max_channels % 4 == 0, "styles: w_dim and channel counts must be multiples of 4");
  const int ny = cdiv(std::max(max_channels, 32), 32);
  hipLaunchKernelGGL(styles_affine_kernel, dim3(n_layers, ny), dim3(256), 0, stream, layers_dev, ws, num_ws, w_dim, B);
  if (f16_prenorm) hipLaunchKernelGGL(f16_prenorm_styles_kernel, dim3(n_layers, B), dim3(64), 0, stream, layers_dev, B);
  hipLaunchKernelGGL(styles_demod_kernel, dim3(n_layers, ny), dim3(256), 0, stream, layers_dev, B);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// ------------------------------------------------------------------------------------------------ toRGB + skip
// stylegan2.py:268-272 (1x1 modulated conv without demodulation, bias, clamp) fused with SynthesisBlock's
// img = upsample2d(img) + y (stylegan2.py:372-378; ops.py:117-133).  HBM-bound: x is read once in 16-byte
// pieces, LP lanes cooperate on one pixel and reduce with wave shuffles.
// bias + clamp of one pixel's RGB, the FIR-upsampled skip of the previous image, planar f32 store
__device__ __forceinline__ void torgb_finish(const RgbArgs& a, int b, long p, float r0, float r1, float r2) {
  const long HW = (long)a.H * a.W;
  const int Hp = a.H >> 1, Wp = a.W >> 1;
  int y = (int)(p / a.W), x = (int)(p - (long)y * a.W);
  float o3[3] = {r0 + a.bias[0], r1 + a.bias[1], r2 + a.bias[2]};
#pragma unroll
  for (int c = 0; c < 3; c++)
    if (a.clamp >= 0.f) o3[c] = fminf(fmaxf(o3[c], -a.clamp), a.clamp);
  if (a.prev) {
    // upsample2d: zero-insert x2, pad (2,1,2,1), correlate with 4f
    const float* pv = a.prev + (long)b * 3 * Hp * Wp;
    // upsample2d (zero-insert x2, pad (2,1,2,1), 4x4 FIR) in its branch-free 2x2 form: only the taps whose
    // parity hits a real sample are non-zero -> rows {iy0, iy0+1}, cols {ix0, ix0+1}, filter index
    // u = 2*iy - y + 2 (same products, same u-major order as the 16-tap correlation)
    const int iy0 = (y - 1) >> 1, ix0 = (x - 1) >> 1;
    float u3[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 2; dy++) {
      const int iy = iy0 + dy, u = 2 * iy - y + 2;
      const bool oky = iy >= 0 && iy < Hp;
#pragma unroll
      for (int dx = 0; dx < 2; dx++) {
        const int ix = ix0 + dx, v = 2 * ix - x + 2;
        const bool ok = oky && ix >= 0 && ix < Wp;
        const bool uh = u == 1 || u == 2, vh = v == 1 || v == 2;  // fir[u][v] takes 3 distinct values
        const float f = !ok ? 0.f : uh ? (vh ? a.fir[5] : a.fir[4]) : (vh ? a.fir[1] : a.fir[0]);
        const long o = ok ? (long)iy * Wp + ix : 0;
        u3[0] += pv[o] * f;
        u3[1] += pv[(long)Hp * Wp + o] * f;
        u3[2] += pv[2L * Hp * Wp + o] * f;
      }
    }
    o3[0] = u3[0] + o3[0]; o3[1] = u3[1] + o3[1]; o3[2] = u3[2] + o3[2];
  }
  float* ob = a.out + (long)b * 3 * HW + p;
  ob[0] = o3[0]; ob[HW] = o3[1]; ob[2 * HW] = o3[2];
}

template <typename T>
__global__ __launch_bounds__(256) void torgb_kernel(RgbArgs a, int lp_log2) {
  constexpr int EPC = 16 / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* wm = reinterpret_cast<float*>(smem);  // [3][C]
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < 3 * a.C; i += blockDim.x) wm[i] = a.wmod[(long)b * 3 * a.C + i];
  __syncthreads();
  const int LP = 1 << lp_log2;
  const int sub = threadIdx.x & (LP - 1);
  const int ppb = blockDim.x >> lp_log2;  // pixels per block-iteration
  const long HW = (long)a.H * a.W;
  const T* xb = reinterpret_cast<const T*>(a.x) + (long)b * HW * a.C;
  const int pieces = a.C / EPC;
  const int Hp = a.H >> 1, Wp = a.W >> 1;
  for (long p = (long)blockIdx.x * ppb + (threadIdx.x >> lp_log2); p < HW; p += (long)gridDim.x * ppb) {
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    for (int pc = sub; pc < pieces; pc += LP) {
      uint4 v = *reinterpret_cast<const uint4*>(xb + p * a.C + pc * EPC);
      float xv[EPC];
      if constexpr (sizeof(T) == 2) {
        xv[0] = Fmt16<T>::lo(v.x); xv[1] = Fmt16<T>::hi(v.x);
        xv[2] = Fmt16<T>::lo(v.y); xv[3] = Fmt16<T>::hi(v.y);
        xv[4] = Fmt16<T>::lo(v.z); xv[5] = Fmt16<T>::hi(v.z);
        xv[6] = Fmt16<T>::lo(v.w); xv[7] = Fmt16<T>::hi(v.w);
      } else {
        xv[0] = __uint_as_float(v.x); xv[1] = __uint_as_float(v.y);
        xv[2] = __uint_as_float(v.z); xv[3] = __uint_as_float(v.w);
      }
      // the lane's EPC weights of each row as 16-byte LDS reads (scalar reads at a 32-byte lane stride conflicted: lds_conf 0.65)
      float wv[3][EPC];
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
        for (int q4 = 0; q4 < EPC / 4; q4++) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(wm + c * a.C + pc * EPC + q4 * 4);
          wv[c][q4 * 4] = t[0]; wv[c][q4 * 4 + 1] = t[1]; wv[c][q4 * 4 + 2] = t[2]; wv[c][q4 * 4 + 3] = t[3];
        }
#pragma unroll
      for (int e = 0; e < EPC; e++) {
        r0 += xv[e] * wv[0][e];
        r1 += xv[e] * wv[1][e];
        r2 += xv[e] * wv[2][e];
      }
    }
    for (int o = LP >> 1; o > 0; o >>= 1) {
      r0 += __shfl_xor(r0, o);
      r1 += __shfl_xor(r1, o);
      r2 += __shfl_xor(r2, o);
    }
    if (sub == 0) torgb_finish(a, b, p, r0, r1, r2);
  }
}

// bf16 inputs: the 1x1 convolution itself on the matrix cores.  out[3][px] = Wmod[3][C] x X^T[C][px] per 32 pixels:
// A = pre-modulated weights split into bf16 hi (rows 0..2) + lo remainder (rows 8..10; w = hi + lo to ~2^-17, the
// same trick as the fused toRGB of modconv_hires.hip), resident in registers; B = the pixels' channel pieces, loaded
// straight from HBM in the MFMA operand layout (lane = pixel, 16 bytes = 8 channels) — no LDS, no cross-lane
// reduction, ~1 VALU op per 16 bytes instead of 24 FMAs + shuffles.
template <int NF>
__global__ __launch_bounds__(256) void torgb_mfma_kernel(RgbArgs a) {
  constexpr int C = NF * 16;
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  u32x4 wf[NF];
  {
    const int c_rgb = r < 3 ? r : (r >= 8 && r < 11 ? r - 8 : -1);
#pragma unroll
    for (int ks = 0; ks < NF; ks++) {
      u32x4 o = u32x4{0u, 0u, 0u, 0u};
      if (c_rgb >= 0) {
        const float* src = a.wmod + ((long)b * 3 + c_rgb) * C + ks * 16 + 8 * h;
        const float4 s0 = *reinterpret_cast<const float4*>(src), s1 = *reinterpret_cast<const float4*>(src + 4);
        const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
          float w0 = sv[2 * k], w1 = sv[2 * k + 1];
          const float h0 = bf2f(f2bf(w0)), h1 = bf2f(f2bf(w1));
          if (r >= 8) { w0 -= h0; w1 -= h1; }
          o[k] = pack2bf(w0, w1);
        }
      }
      wf[ks] = o;
    }
  }
  const long HW = (long)a.H * a.W;
  const bf16_t* xb = reinterpret_cast<const bf16_t*>(a.x) + (long)b * HW * C;
  const long n_groups = (HW + 31) / 32;
  for (long gi = (long)blockIdx.x * 4 + (threadIdx.x >> 6); gi < n_groups; gi += (long)gridDim.x * 4) {
    const long p = gi * 32 + r;
    const bf16_t* px = xb + (p < HW ? p : HW - 1) * C + 8 * h;
    u32x4 xv[NF];
#pragma unroll
    for (int ks = 0; ks < NF; ks++) xv[ks] = *reinterpret_cast<const u32x4*>(px + ks * 16);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NF; ks++)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[ks]), __builtin_bit_cast(bf16x8, xv[ks]),
                                                    acc, 0, 0, 0);
    // D rows 0..2 (hi) and 8..10 (lo) of pixel r live in the h == 0 lane: acc[0..2] and acc[4..6]
    if (h == 0 && p < HW) torgb_finish(a, b, p, acc[0] + acc[4], acc[1] + acc[5], acc[2] + acc[6]);
  }
}

int launch_torgb(hipStream_t stream, int dtype, const RgbArgs& a) {
  if (a.B == 0) return MAUA_OK;
  const int epc = dtype == MAUA_F32 ? 4 : 8;
  MAUA_REQUIRE(a.C % epc == 0, "torgb: C must be a multiple of the 16-byte piece");
  // (C = 512 layers are <= 64^2: the fragment set-up of 32 k-steps costs more than it saves there)
  if (dtype == MAUA_BF16 && (a.C == 32 || a.C == 64 || a.C == 128 || a.C == 256) && ((uintptr_t)a.wmod % 16) == 0) {
    const long groups = ((long)a.H * a.W + 31) / 32;
    const dim3 grid((unsigned)std::min<long>((groups + 3) / 4, 1024), a.B);
    switch (a.C / 16) {
      case 2: hipLaunchKernelGGL(torgb_mfma_kernel<2>, grid, dim3(256), 0, stream, a); break;
      case 4: hipLaunchKernelGGL(torgb_mfma_kernel<4>, grid, dim3(256), 0, stream, a); break;
      case 8: hipLaunchKernelGGL(torgb_mfma_kernel<8>, grid, dim3(256), 0, stream, a); break;
      default: hipLaunchKernelGGL(torgb_mfma_kernel<16>, grid, dim3(256), 0, stream, a); break;
    }
    MAUA_HIP_CHECK(hipGetLastError());
    return MAUA_OK;
  }
  int pieces = a.C / epc;
  int lp_log2 = 0;
  while ((1 << (lp_log2 + 1)) <= pieces && lp_log2 < 4) lp_log2++;
  long HW = (long)a.H * a.W;
  int ppb = 256 >> lp_log2;
  int gx = (int)std::min<long>((HW + ppb - 1) / ppb, 2048);
  size_t smem = (size_t)3 * a.C * sizeof(float);
  if (dtype == MAUA_BF16)
    hipLaunchKernelGGL(torgb_kernel<bf16_t>, dim3(gx, a.B), dim3(256), smem, stream, a, lp_log2);
  else if (dtype == MAUA_F16)
    hipLaunchKernelGGL(torgb_kernel<f16_t>, dim3(gx, a.B), dim3(256), smem, stream, a, lp_log2);
  else if (dtype == MAUA_F32)
    hipLaunchKernelGGL(torgb_kernel<float>, dim3(gx, a.B), dim3(256), smem, stream, a, lp_log2);
  else
    return fail("torgb: unsupported dtype");
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace maua
