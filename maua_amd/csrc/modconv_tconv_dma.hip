// Minimal-MAC up-layer, first half (t = conv_transpose2d(x * s, W, stride 2): reference ops.py:211-224), on the
// LDS-direct-load pipeline of modconv_dma.hip - the main H x W block of the (H+1) x (W+1) position grid; the thin
// last row / last column keep the register-staged kernel of modconv_tconv.hip.
//
// Same arithmetic and the same K order as tconv2_kernel (chunks of 32 channels, 9 (shift, class) weight blocks per
// 16-channel k-step, classes accumulate their taps in the same order), so both produce bit-identical t from the same
// bf16 operands.  What changes is the data path:
//   * the input is ALREADY multiplied by the styles (the producing conv1's epilogue, or a premod pass), so the halo
//     (9 x 33 pixels x 64 B per chunk) and the weights (9 blocks x 32 rows x 64 B per chunk) both arrive by
//     global_load_lds_dwordx4 into double buffers - no staging registers, no ds_write, no style multiplication;
//   * ONE barrier per chunk (tconv2_kernel: two), placed before the chunk's last k-step so that the refill of the
//     buffers and the first fragment reads of the next chunk overlap the last MFMAs;
//   * a wave owns TWO position rows (64 positions x 4 classes x 32 channels = 128 accumulator registers): the 9 weight
//     fragments of a k-step feed 18 MFMAs instead of 9, the 6 distinct halo fragments (3 rows x 2 column shifts) are
//     shared by the two rows - 0.83 ds_read_b128 per MFMA instead of 1.44;
//   * 4 waves / 75 KB of LDS per workgroup: two workgroups per CU run out of phase.
#include "common.h"
#include "internal.h"

namespace maua {

namespace {

constexpr int PTH = 8, PTW = 32;                   // positions per workgroup: 8 rows x 32 columns
constexpr int HW1 = PTW + 1, HPX = (PTH + 1) * HW1;  // 9 x 33 halo pixels
constexpr int KB = 64;                             // bytes of K per LDS row (32 bf16 channels = one chunk)
constexpr int HBUF = HPX * KB;                     // 19 008
constexpr int WROWS = 9 * 32;                      // weight rows per chunk: 9 (shift, class) blocks of 32 channels
constexpr int WBUF = WROWS * KB;                   // 18 432
constexpr int OFF_H = 2 * WBUF;
constexpr int NW = 4, NT = NW * 64;
constexpr int WJ = (WBUF / 1024 + NW - 1) / NW;    // 5 weight instructions per wave per chunk (18 in all)
constexpr int HJ = (HPX * 4 + NT - 1) / NT;        // 5 halo instructions per wave per chunk (1188 pieces)

__device__ __constant__ const int kSlot[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};  // as modconv_tconv.hip
__device__ __constant__ const int kCls[9] = {0, 0, 1, 0, 2, 0, 1, 2, 3};

__device__ __forceinline__ unsigned lds_off(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ void dma16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
template <typename F>
__device__ __forceinline__ void mma(f32x16& acc, const u32x4& w, const u32x4& x) {
  Mma16<F>::step(acc, w, x);
}
__device__ __forceinline__ int swz(int n) { return (n >> 2) & 3; }

}  // namespace

template <typename F>
__global__ __launch_bounds__(NT, 2) void tconv_dma_kernel(ConvArgs a) {
  constexpr int ES = 128 * 2 + 16, PPP = 16;  // epilogue tile row stride, 16-byte pieces per position (128 virtual ch)
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_off(smem));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  // Workgroup order (1-D grid): the dispatcher places block L on XCD L % 8, each XCD has its own L2.  The channel
  // blocks of one (tile, sample) read the same input halo, so they run back to back ON ONE XCD (cb fastest inside an
  // XCD, in groups of <= 8 blocks = <= 2.4 MB of weights, which stay L2-resident next to the halos): x is fetched from
  // HBM once per group instead of once per channel block.  Placement is a speed matter only.
  const int tiles_x = a.W >> 5, tiles = tiles_x * (a.H >> 3), CB = a.Co >> 5;
  const int cbg = CB < 8 ? CB : 8, n_ts = tiles * a.B, per_group = ((n_ts + 7) >> 3) * 8 * cbg;
  const int L = blockIdx.x, grp = L / per_group, Lg = L - grp * per_group;
  const int xcd = Lg & 7, idx = Lg >> 3;
  const int cb = grp * cbg + idx % cbg, ts = (idx / cbg) * 8 + xcd;
  if (ts >= n_ts) return;
  const int b = ts / tiles, tile = ts - b * tiles;
  const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
  const int ty0 = tyi * PTH, tx0 = txi * PTW;
  const char* xb = reinterpret_cast<const char*>(a.x) + (long)b * a.x_bstride * 2;
  const char* wp = reinterpret_cast<const char*>(a.w);

  // halo sources (pixels above / left of the image are never loaded: zeroed once)
  unsigned hoff[HJ];
#pragma unroll
  for (int j = 0; j < HJ; j++) {
    const int P = (wave + NW * j) * 64 + lane;
    const int hp = P >> 2, q = (P & 3) ^ swz(hp);
    const int py = (hp * 1986) >> 16;  // hp / 33 for hp < 297
    const int px = hp - py * HW1;
    const int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
    hoff[j] = 0xffffffffu;
    if (P < HPX * 4) {
      if (gy >= 0 && gx >= 0) {
        hoff[j] = (unsigned)(((gy * a.W + gx) * a.Ci + q * 8) * 2);
      } else {
        *reinterpret_cast<u32x4*>(smem + OFF_H + P * 16) = u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(smem + OFF_H + HBUF + P * 16) = u32x4{0u, 0u, 0u, 0u};
      }
    }
  }
  // weight sources: LDS row R = 32 k + n of the chunk's [288][64 B] image <- global row of block k, channel n
  unsigned woff[WJ];
#pragma unroll
  for (int j = 0; j < WJ; j++) {
    const int ii = wave + NW * j;
    const int R = min(16 * ii + (lane >> 2), WROWS - 1);
    const int k = R >> 5, n = R & 31;
    const int q = (lane & 3) ^ swz(R);
    woff[j] = (unsigned)((((((kSlot[k] * CB + cb) * 4 + kCls[k]) * 32 + n) * a.Ci) + q * 8) * 2);
  }
#define TD_ISSUE(C_, BUF_)                                                                               \
  {                                                                                                      \
    const char* ws_ = wp + (long)(C_) * KB;                                                              \
    _Pragma("unroll") for (int j = 0; j < WJ; j++)                                                      \
      if (wave + NW * j < WBUF / 1024) dma16_s(ws_, woff[j], lds0 + (BUF_) * WBUF + (wave + NW * j) * 1024); \
    const char* xs_ = xb + (long)(C_) * KB;                                                              \
    _Pragma("unroll") for (int j = 0; j < HJ; j++)                                                      \
      if (hoff[j] != 0xffffffffu) dma16_s(xs_, hoff[j], lds0 + OFF_H + (BUF_) * HBUF + (wave + NW * j) * 1024); \
  }

  // fragment addresses.  A: halo pixel of (local row 2 wave + R, column r) under shift (p, q) = row + 1 - p, r + 1 - q
  const int hp00 = (2 * wave) * HW1 + r;   // halo row 2 wave, halo column r  (= shift (1,1) of the wave's first row)
  const unsigned b0 = (unsigned)(r * KB + ((swz(r) ^ h) << 4));
#define TD_A(HR_, Q1_, KS_, BUF_)                                                                        \
  ({                                                                                                     \
    const int hp_ = hpv + (HR_) * HW1 + (Q1_);                                                           \
    *reinterpret_cast<const u32x4*>(smem + ((OFF_H + (BUF_) * HBUF + hp_ * KB + ((swz(hp_) ^ h) << 4)) ^ ((KS_) << 5))); \
  })
#define TD_B(K_, KS_, BUF_) (*reinterpret_cast<const u32x4*>(smem + (BUF_) * WBUF + (K_) * 32 * KB + (b0 ^ ((KS_) << 5))))

  f32x16 acc[2][4];
#pragma unroll
  for (int R = 0; R < 2; R++)
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[R][c][e] = 0.f;

  const int n_chunks = a.Ci >> 5;
  TD_ISSUE(0, 0)
  if (n_chunks > 1) TD_ISSUE(1, 1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int hpv = hp00;
  // one k-step: 6 halo fragments (rows 2w .. 2w+2 of the halo x column shifts q = 1, 0), 9 weight fragments, 18 MFMAs;
  // per class the taps accumulate in the order of tconv2_kernel (shift slots 0, 1, 2, 3)
#define TD_STEP(KS_, BUF_)                                                                               \
  {                                                                                                      \
    u32x4 A00 = TD_A(0, 0, KS_, BUF_), A01 = TD_A(0, 1, KS_, BUF_);   /* halo row 2w:   q = 1, q = 0 */   \
    u32x4 A10 = TD_A(1, 0, KS_, BUF_), A11 = TD_A(1, 1, KS_, BUF_);   /* halo row 2w+1 */                 \
    u32x4 A20 = TD_A(2, 0, KS_, BUF_), A21 = TD_A(2, 1, KS_, BUF_);   /* halo row 2w+2 */                 \
    {                                                                                                    \
      const u32x4 B0 = TD_B(0, KS_, BUF_), B1 = TD_B(1, KS_, BUF_), B2 = TD_B(2, KS_, BUF_);              \
      mma<F>(acc[0][0], B0, A00); mma<F>(acc[1][0], B0, A10);            /* shift (1,1) */                     \
      mma<F>(acc[0][1], B2, A01); mma<F>(acc[1][1], B2, A11);            /* shift (1,0), class 1 */            \
      mma<F>(acc[0][0], B1, A01); mma<F>(acc[1][0], B1, A11);            /* shift (1,0), class 0 */            \
    }                                                                                                    \
    {                                                                                                    \
      const u32x4 B3 = TD_B(3, KS_, BUF_), B4 = TD_B(4, KS_, BUF_);                                       \
      mma<F>(acc[0][2], B4, A10); mma<F>(acc[1][2], B4, A20);            /* shift (0,1), class 2 */            \
      mma<F>(acc[0][0], B3, A10); mma<F>(acc[1][0], B3, A20);            /* shift (0,1), class 0 */            \
    }                                                                                                    \
    {                                                                                                    \
      const u32x4 B5 = TD_B(5, KS_, BUF_), B6 = TD_B(6, KS_, BUF_), B7 = TD_B(7, KS_, BUF_), B8 = TD_B(8, KS_, BUF_); \
      mma<F>(acc[0][3], B8, A11); mma<F>(acc[1][3], B8, A21);            /* shift (0,0) */                     \
      mma<F>(acc[0][1], B6, A11); mma<F>(acc[1][1], B6, A21);                                                  \
      mma<F>(acc[0][2], B7, A11); mma<F>(acc[1][2], B7, A21);                                                  \
      mma<F>(acc[0][0], B5, A11); mma<F>(acc[1][0], B5, A21);                                                  \
    }                                                                                                    \
  }
  for (int c = 0; c < n_chunks; c++) {
    const int buf = c & 1;
    asm volatile("" : "+v"(hpv));  // (keeps the fragment addresses out of loop-invariant registers)
    TD_STEP(0, buf)
    // the chunk's last fragment reads are issued inside the next step; the barrier that frees the buffers comes after
    // they have returned, so it splits that step by hand: loads, wait, barrier, refill, MFMAs
    {
      u32x4 A00 = TD_A(0, 0, 1, buf), A01 = TD_A(0, 1, 1, buf), A10 = TD_A(1, 0, 1, buf), A11 = TD_A(1, 1, 1, buf);
      u32x4 A20 = TD_A(2, 0, 1, buf), A21 = TD_A(2, 1, 1, buf);
      const u32x4 B0 = TD_B(0, 1, buf), B1 = TD_B(1, 1, buf), B2 = TD_B(2, 1, buf), B3 = TD_B(3, 1, buf), B4 = TD_B(4, 1, buf);
      const u32x4 B5 = TD_B(5, 1, buf), B6 = TD_B(6, 1, buf), B7 = TD_B(7, 1, buf), B8 = TD_B(8, 1, buf);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // chunk c+1 (issued one chunk ago) has landed
      __syncthreads();                                  // ... for everybody; chunk c's buffers are free
      if (c + 2 < n_chunks) TD_ISSUE(c + 2, buf)
      mma<F>(acc[0][0], B0, A00); mma<F>(acc[1][0], B0, A10);
      mma<F>(acc[0][1], B2, A01); mma<F>(acc[1][1], B2, A11);
      mma<F>(acc[0][0], B1, A01); mma<F>(acc[1][0], B1, A11);
      mma<F>(acc[0][2], B4, A10); mma<F>(acc[1][2], B4, A20);
      mma<F>(acc[0][0], B3, A10); mma<F>(acc[1][0], B3, A20);
      mma<F>(acc[0][3], B8, A11); mma<F>(acc[1][3], B8, A21);
      mma<F>(acc[0][1], B6, A11); mma<F>(acc[1][1], B6, A21);
      mma<F>(acc[0][2], B7, A11); mma<F>(acc[1][2], B7, A21);
      mma<F>(acc[0][0], B5, A11); mma<F>(acc[1][0], B5, A21);
    }
  }
#undef TD_ISSUE
#undef TD_A
#undef TD_B
#undef TD_STEP

  // ---- raw t tile -> LDS [position][class * 32 + ch] -> 16-byte NHWC pieces of t [2H+1][2W+1][Co]
  const int Wt = 2 * a.W + 1, Ht = 2 * a.H + 1;
  __syncthreads();
  char* epi = smem;
#pragma unroll
  for (int R = 0; R < 2; R++) {
    const int m = (2 * wave + R) * 32 + r;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int qd = 0; qd < 4; qd++)
        *reinterpret_cast<uint2*>(epi + m * ES + (c * 32 + 8 * qd + 4 * h) * 2) =
            make_uint2(Fmt16<F>::pack2(acc[R][c][qd * 4 + 0], acc[R][c][qd * 4 + 1]), Fmt16<F>::pack2(acc[R][c][qd * 4 + 2], acc[R][c][qd * 4 + 3]));
  }
  __syncthreads();
  char* yb = reinterpret_cast<char*>(a.y) + (long)b * Ht * Wt * a.Co * 2;
  for (int p = tid; p < PTH * PTW * PPP; p += NT) {
    const int mm = p / PPP, pc = p - mm * PPP;
    const int gy = ty0 + (mm >> 5), gx = tx0 + (mm & 31);
    const int nv = pc * 8, cls = nv >> 5, ch = nv & 31;
    const int oy = 2 * gy + (cls >> 1), ox = 2 * gx + (cls & 1);
    *reinterpret_cast<uint4*>(yb + (((long)oy * Wt + ox) * a.Co + cb * 32 + ch) * 2) =
        *reinterpret_cast<const uint4*>(epi + mm * ES + pc * 16);
  }
}

// ---- the thin last row (i = H) and last column (j = W) of the (H + 1) x (W + 1) position grid ---------------------------
// Beyond the image only the taps that reach back inside contribute: the row edge sees x[H - 1][.] through the shift slots
// (1,1) / (1,0) and produces the classes (0,0), (0,1) (t row 2H); the column edge sees x[.][W - 1] through (1,1) / (0,1)
// and produces (0,0), (1,0) (t column 2W).  That is 3 weight blocks and 2 input fragments per k-step instead of 9 and 4,
// on 32 consecutive edge positions per wave - no tile of mostly-absent neighbours, no LDS: both operands come straight
// from global memory (L2) as MFMA fragments.  One wave per (32 positions, 32 channels, sample); K runs over all input
// channels in 16-channel steps.  (tconv2_kernel's thin regions did this work in 8 x 32 / 64 x 4 tiles at 12 - 25 % use
// with all nine blocks: 0.06 ms per layer at B = 32 for 0.4 % of the layer's MACs.)
template <typename F>
__global__ __launch_bounds__(64) void tconv_edges_kernel(ConvArgs a, int row_tiles) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int b = blockIdx.y, cb = blockIdx.z, CB = a.Co >> 5;
  const bool row_edge = (int)blockIdx.x < row_tiles;
  const int e0 = (row_edge ? (int)blockIdx.x : (int)blockIdx.x - row_tiles) * 32 + r;   // j (row edge) or i (column edge)
  const uint16_t* xb = reinterpret_cast<const uint16_t*>(a.x) + (long)b * a.x_bstride;
  const uint16_t* wp = reinterpret_cast<const uint16_t*>(a.w);
  // input fragments: X0 = shift (1,1), X1 = shift (1,0) [row edge] or (0,1) [column edge]
  long x0 = -1, x1 = -1;
  if (row_edge) {
    if (e0 >= 1 && e0 <= a.W) x0 = ((long)(a.H - 1) * a.W + (e0 - 1)) * a.Ci;
    if (e0 < a.W) x1 = ((long)(a.H - 1) * a.W + e0) * a.Ci;
  } else {
    if (e0 >= 1 && e0 <= a.H - 1) x0 = ((long)(e0 - 1) * a.W + (a.W - 1)) * a.Ci;
    if (e0 < a.H) x1 = ((long)e0 * a.W + (a.W - 1)) * a.Ci;
  }
  // weight blocks [slot][CB][class][32][Ci]: W0 = (slot 0, class 0); row edge: W1 = (1, 0), W2 = (1, 1); column: (2, 0), (2, 2)
  const int s1 = row_edge ? 1 : 2, c2 = row_edge ? 1 : 2;
  const uint16_t* w0 = wp + ((((long)0 * CB + cb) * 4 + 0) * 32 + r) * a.Ci + 8 * h;
  const uint16_t* w1 = wp + ((((long)s1 * CB + cb) * 4 + 0) * 32 + r) * a.Ci + 8 * h;
  const uint16_t* w2 = wp + ((((long)s1 * CB + cb) * 4 + c2) * 32 + r) * a.Ci + 8 * h;
  f32x16 acc0, acc1;
#pragma unroll
  for (int e = 0; e < 16; e++) { acc0[e] = 0.f; acc1[e] = 0.f; }
  const u32x4 zero = u32x4{0u, 0u, 0u, 0u};
#pragma unroll 4
  for (int k = 0; k < a.Ci; k += 16) {
    const u32x4 A0 = *reinterpret_cast<const u32x4*>(w0 + k), A1 = *reinterpret_cast<const u32x4*>(w1 + k),
                A2 = *reinterpret_cast<const u32x4*>(w2 + k);
    const u32x4 X0 = x0 >= 0 ? *reinterpret_cast<const u32x4*>(xb + x0 + k + 8 * h) : zero;
    const u32x4 X1 = x1 >= 0 ? *reinterpret_cast<const u32x4*>(xb + x1 + k + 8 * h) : zero;
    mma<F>(acc0, A0, X0);
    mma<F>(acc1, A2, X1);
    mma<F>(acc0, A1, X1);
  }
  const int Wt = 2 * a.W + 1, Ht = 2 * a.H + 1;
  uint16_t* yb = reinterpret_cast<uint16_t*>(a.y) + (long)b * Ht * Wt * a.Co + cb * 32 + 4 * h;
  long o0 = -1, o1 = -1;   // class (0,0) and the edge's second class
  if (row_edge) {
    if (e0 <= a.W) o0 = ((long)(2 * a.H) * Wt + 2 * e0) * a.Co;
    if (e0 < a.W) o1 = ((long)(2 * a.H) * Wt + 2 * e0 + 1) * a.Co;
  } else if (e0 < a.H) {
    o0 = ((long)(2 * e0) * Wt + 2 * a.W) * a.Co;
    o1 = ((long)(2 * e0 + 1) * Wt + 2 * a.W) * a.Co;
  }
#pragma unroll
  for (int qd = 0; qd < 4; qd++) {
    if (o0 >= 0)
      *reinterpret_cast<uint2*>(yb + o0 + 8 * qd) = make_uint2(Fmt16<F>::pack2(acc0[qd * 4], acc0[qd * 4 + 1]), Fmt16<F>::pack2(acc0[qd * 4 + 2], acc0[qd * 4 + 3]));
    if (o1 >= 0)
      *reinterpret_cast<uint2*>(yb + o1 + 8 * qd) = make_uint2(Fmt16<F>::pack2(acc1[qd * 4], acc1[qd * 4 + 1]), Fmt16<F>::pack2(acc1[qd * 4 + 2], acc1[qd * 4 + 3]));
  }
}

// last row / column of positions; a.x must already carry the styles (as for launch_tconv_dma)
int launch_tconv_edges(hipStream_t stream, const ConvArgs& a, int dtype) {
  MAUA_REQUIRE(a.Ci % 16 == 0 && a.Co % 32 == 0, "tconv_edges: Ci % 16, Co % 32");
  if (a.B == 0) return MAUA_OK;
  const int row_tiles = (a.W + 1 + 31) / 32, col_tiles = (a.H + 31) / 32;
  MAUA_REQUIRE(a.B <= 65535 && a.Co / 32 <= 65535, "tconv_edges: grid too large");
  if (dtype == MAUA_F16)
    hipLaunchKernelGGL(tconv_edges_kernel<f16_t>, dim3(row_tiles + col_tiles, a.B, a.Co / 32), dim3(64), 0, stream, a, row_tiles);
  else
    hipLaunchKernelGGL(tconv_edges_kernel<bf16_t>, dim3(row_tiles + col_tiles, a.B, a.Co / 32), dim3(64), 0, stream, a, row_tiles);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

bool tconv_dma_supported(int dtype, int Ci, int Co, int H, int W) {
  return (dtype == MAUA_BF16 || dtype == MAUA_F16) && Ci % 32 == 0 && Co % 32 == 0 && H % PTH == 0 && W % PTW == 0 &&
         (long)H * W * Ci * 2 < (1L << 32) && 16L * Co * Ci * 2 < (1L << 32);
}

// main H x W block of the position grid; a.x must already carry the styles.  The caller adds the last row / column with
// launch_tconv2 (variant = TCONV_EDGES_ONLY, unit styles).
int launch_tconv_dma(hipStream_t stream, const ConvArgs& a, int dtype) {
  MAUA_REQUIRE(tconv_dma_supported(dtype, a.Ci, a.Co, a.H, a.W), "tconv_dma: unsupported shape");
  if (a.B == 0) return MAUA_OK;
  const int tiles = (a.H / PTH) * (a.W / PTW), CB = a.Co / 32, cbg = CB < 8 ? CB : 8;
  MAUA_REQUIRE(CB % cbg == 0, "tconv_dma: channel blocks must split into groups of 8");
  const long n_ts = (long)tiles * a.B, grid = ((n_ts + 7) / 8) * 8 * cbg * (CB / cbg);
  MAUA_REQUIRE(grid < (1L << 31), "tconv_dma: grid too large");
  const size_t smem = std::max<size_t>((size_t)2 * WBUF + 2 * HBUF, (size_t)PTH * PTW * (128 * 2 + 16));
  if (dtype == MAUA_F16) {
    MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)tconv_dma_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(tconv_dma_kernel<f16_t>, dim3((unsigned)grid), dim3(NT), smem, stream, a);
  } else {
    MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)tconv_dma_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(tconv_dma_kernel<bf16_t>, dim3((unsigned)grid), dim3(NT), smem, stream, a);
  }
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace maua
