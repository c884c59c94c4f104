// Minimal-MAC up-layer as ONE persistent kernel with producer and consumer waves: t = conv_transpose2d(x * s, W, stride 2) on
// the matrix cores (waves 0-3), the 4 x 4 FIR + layer epilogue from an LDS tile of t (waves 4-7) - of the PREVIOUS tile, while
// the producers already multiply the next one.  The (2H+1) x (2W+1) tensor never reaches HBM.
//
// Replaces (reference): ops.py:211-225 (conv2d_resample, up = 2: conv_transpose2d, then upfirdn2d pad 1 gain 4) + :184-185
// noise + :65-84 bias_act.  Same tiles and the same K loop as modconv_tconv_fir.hip (8 x 32 positions x 32 channels per item,
// 6 x 30 useful: the frame of one position is recomputed instead of exchanged).
//
// Why a third form of this layer (round 4): timing-only builds of tconv_fir_kernel showed that neither of its phases is bound
// by a pipe - both are chains of exposed latencies.  A workgroup alone on a CU needs 14 200 cycles for a tile's K loop (4 608 cycles
// of MFMAs: the first two chunks' loads are waited for in full, then every chunk waits for the one issued a chunk earlier) and
// 9 100 cycles for its FIR phase (noise / per-channel loads, two barriers, the tile write, then nine dependent row steps), and two
// co-resident workgroups hide only part of each other's gaps: 16 500 cycles per tile and CU.  Here:
//   * one 512-thread workgroup per CU walks a contiguous run of items (sample, tile, channel block) - the LDS-direct loads of
//     chunk g + 2 are issued while chunk g is multiplied ACROSS item boundaries, so the load latency is paid once per launch,
//     not once per tile;
//   * the K-loop buffers (75 KB) and the t tile (72 KB) do not alias, the consumers filter tile i - 1 while the producers
//     multiply tile i; their global operands (noise, d, bias, next-layer styles) are requested one hand-over ahead;
//   * the horizontal [1,3,3,1] pass is a banded-Toeplitz product on the matrix cores (t is bf16, the taps are exact in bf16,
//     f32 accumulation: exact up to the order of four additions), the vertical pass + epilogue run lane-locally on its
//     accumulators (a lane = one pixel x 16 channels), lane pairs exchange halves and store 32 contiguous bytes each.
// Synchronisation is the workgroup barrier only: every wave executes the same number of s_barrier per item (n_chunks chunk
// barriers + "tile free" + "tile ready"); the consumers spread the chunk barriers over their nine row steps.
#include <stdlib.h>

#include "common.h"
#include "internal.h"

#ifndef TP_SKIP
#define TP_SKIP 0   // timing-only builds (scripts/mk_variant.sh): 1 no multiplies in the K loop, 2 no FIR / epilogue, 4 no output stores
#endif

namespace maua {

namespace {

constexpr int PTH = 8, PTW = 32, UPR = PTH - 2;    // position rows / columns per item, useful rows
constexpr int HW1 = PTW + 1;                       // halo columns
constexpr int KB = 64;                             // bytes of K per LDS row (32 bf16 channels = one chunk)
constexpr int WROWS = 9 * 32;
constexpr int WBUF = WROWS * KB;                   // 18 432
constexpr int HPX = (PTH + 1) * HW1, HBUF = HPX * KB;   // 297 halo pixels, 19 008 bytes
constexpr int OFF_H = 2 * WBUF;
constexpr int OFF_T = ((2 * WBUF + 2 * HBUF + 1023) / 1024) * 1024;
constexpr int RS = 128 + 16;                       // t tile [t row][ch][64 t columns] bf16: bytes per (row, ch); the 16-byte pad makes the
                                                   // producers' 16-byte writes (8 consecutive lanes = 8 channels) and the consumers' fragment
                                                   // reads (16-lane groups of channels) conflict-free
constexpr int TBYTES = 2 * PTH * 32 * RS;          // 73 728
constexpr int SMEM = OFF_T + TBYTES;               // 149 504: one workgroup per CU
constexpr int NW = 4;                              // producer waves (= consumer waves)
constexpr int WJ = (WBUF / 1024 + NW - 1) / NW, HJ = (HPX * 4 + NW * 64 - 1) / (NW * 64);

__device__ __constant__ const int kSlotP[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};  // as modconv_tconv.hip
__device__ __constant__ const int kClsP[9] = {0, 0, 1, 0, 2, 0, 1, 2, 3};

__device__ __forceinline__ unsigned lds_off(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ void dma16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
// D[m = position][n = channel]: a lane holds ONE channel of 4 x 4 consecutive positions (what the tile write wants)
__device__ __forceinline__ void mma_t(f32x16& acc, const u32x4& w, const u32x4& x) {
  if (TP_SKIP & 1) return;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, w), acc, 0, 0, 0);
}
// ... the first product of an accumulator in an item starts from zero (no 128 v_mov per item)
__device__ __forceinline__ void mma_0(f32x16& acc, const u32x4& w, const u32x4& x) {
  f32x16 z;
#pragma unroll
  for (int e = 0; e < 16; e++) z[e] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, w), z, 0, 0, 0);
}
__device__ __forceinline__ int swz(int n) { return (n >> 2) & 3; }
// every wave of the workgroup, the same number of times per item.  lgkmcnt(0): this wave's LDS reads / writes are done (the
// producers' LDS-direct loads are waited for with vmcnt where it matters); the memory clobber keeps the compiler's LDS accesses on
// their side of the barrier.
__device__ __forceinline__ void bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// item = (sample b, tile (tyi, txi), channel block cb), cb fastest, then the tiles of a sample column by column (vertical
// neighbours share 3 of 9 halo rows: back to back), then the samples.  Decoded once per role, then advanced by carries.
struct PcItem {
  int b, cb, tyi, txi;
  __device__ __forceinline__ int ty0() const { return tyi * UPR - 1; }
  __device__ __forceinline__ int tx0() const { return txi * 30 - 1; }
};
__device__ __forceinline__ PcItem pc_decode(int I, int CB, int tiles, int tiles_y) {
  PcItem it;
  const int ts = I / CB;
  it.cb = I - ts * CB;
  it.b = ts / tiles;
  const int tile = ts - it.b * tiles;
  it.txi = tile / tiles_y;
  it.tyi = tile - it.txi * tiles_y;
  return it;
}
__device__ __forceinline__ void pc_next(PcItem& it, int CB, int tiles_x, int tiles_y) {
  if (++it.cb < CB) return;
  it.cb = 0;
  if (++it.tyi < tiles_y) return;
  it.tyi = 0;
  if (++it.txi < tiles_x) return;
  it.txi = 0;
  it.b++;
}

}  // namespace

__global__ __launch_bounds__(512, 1) void tconv_pc_kernel(ConvArgs a, UpfirArgs u, int items_per_wg, int n_total) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_off(smem));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int tiles_x = (a.W + 29) / 30, tiles_y = (a.H + UPR - 1) / UPR, tiles = tiles_x * tiles_y, CB = a.Co >> 5;
  const int I0 = blockIdx.x * items_per_wg;
  const int n_items = min(items_per_wg, n_total - I0);
  if (n_items <= 0) return;
  const int nc = a.Ci >> 5;
  char* tt = smem + OFF_T;

  if (wave8 < NW) {
    // =========================================================== producers: the K loop of item after item
    const int wave = wave8;
    // static part of this lane's halo / weight sources
    int hpy[HJ], hpx[HJ], hq[HJ];
    bool hslot[HJ];
#pragma unroll
    for (int j = 0; j < HJ; j++) {
      const int P = (wave + NW * j) * 64 + lane;
      const int hp = P >> 2;
      hq[j] = ((P & 3) ^ swz(hp)) * 16;
      hpy[j] = (hp * 1986) >> 16;  // hp / 33 for hp < 561
      hpx[j] = hp - hpy[j] * HW1;
      hslot[j] = P < HPX * 4;
    }
    const char* wp = reinterpret_cast<const char*>(a.w);
    unsigned woff[WJ];
#pragma unroll
    for (int j = 0; j < WJ; j++) {
      const int ii = wave + NW * j;
      const int R = min(16 * ii + (lane >> 2), WROWS - 1);
      const int k = R >> 5, n = R & 31;
      const int q = (lane & 3) ^ swz(R);
      woff[j] = (unsigned)((((((kSlotP[k] * CB) * 4 + kClsP[k]) * 32 + n) * a.Ci) + q * 8) * 2);
    }
    // the weights' cursor runs half a chunk behind the halo's (its pieces go out in the NEXT k-step's gaps: L2-warm, 250-400 cycles)
    int w_c = 0, w_cb = I0 % CB, w_left = n_items * nc;
    const char* wsrc = nullptr;
#define TP_ISSUE_W(J_, BUF_)                                                                              \
  if (wave + NW * (J_) < WBUF / 1024 && !(TP_SKIP & 16)) dma16_s(wsrc, woff[J_], lds0 + (BUF_) * WBUF + (wave + NW * (J_)) * 1024);
#define TP_W_BEGIN() { wsrc = wp + (long)w_cb * (256 * a.Ci) + (long)w_c * KB; }
#define TP_W_END()                                                                                        \
  {                                                                                                       \
    w_left--;                                                                                             \
    if (++w_c == nc) {                                                                                    \
      w_c = 0;                                                                                            \
      if (++w_cb == CB) w_cb = 0;                                                                         \
    }                                                                                                     \
  }
    // the issue cursor: the item / chunk whose loads go out next (two chunks ahead of the one being multiplied)
    PcItem ic = pc_decode(I0, CB, tiles, tiles_y);
    int iss_c = 0, iss_items = n_items;
    unsigned hoff[HJ];
    const char* xs = nullptr;   // sample base of the cursor's item
#define TP_CURSOR()                                                                                       \
  {                                                                                                       \
    xs = reinterpret_cast<const char*>(a.x) + (long)ic.b * a.x_bstride * 2;                               \
    const int ty0_ = ic.ty0(), tx0_ = ic.tx0();                                                           \
    _Pragma("unroll") for (int j = 0; j < HJ; j++) {                                                     \
      const int gy = ty0_ - 1 + hpy[j], gx = tx0_ - 1 + hpx[j];                                           \
      const bool in = hslot[j] && gy >= 0 && gx >= 0 && gy < a.H && gx < a.W;                             \
      hoff[j] = in ? ((unsigned)(gy * a.W + gx) * (unsigned)a.Ci * 2u + (unsigned)hq[j]) : 0xffffffffu;  \
    }                                                                                                     \
  }
    // halo piece J_ of the cursor's chunk into buffer BUF_; pixels outside the image are zeroed by hand in both buffers when an
    // item starts (its first two chunks) and not touched again
#define TP_ISSUE_H(J_, BUF_)                                                                              \
  {                                                                                                       \
    if (hoff[J_] != 0xffffffffu) {                                                                        \
      if (!(TP_SKIP & 8)) dma16_s(xs + (long)iss_c * KB, hoff[J_], lds0 + OFF_H + (BUF_) * HBUF + (wave + NW * (J_)) * 1024); \
    } else if (iss_c < 2 && hslot[J_]) {                                                                  \
      *reinterpret_cast<u32x4*>(smem + OFF_H + (BUF_) * HBUF + ((wave + NW * (J_)) * 64 + lane) * 16) = u32x4{0u, 0u, 0u, 0u}; \
    }                                                                                                     \
  }
#define TP_ADVANCE()                                                                                      \
  {                                                                                                       \
    if (++iss_c == nc) {                                                                                  \
      iss_c = 0;                                                                                          \
      pc_next(ic, CB, tiles_x, tiles_y);                                                                  \
      if (--iss_items > 0 && !(TP_SKIP & 64)) TP_CURSOR()                                                 \
    }                                                                                                     \
  }
    // fragment addresses.  A: halo pixel of (local row 2 wave + R, column r) under shift (p, q) = row + 1 - p, r + 1 - q
    const int hp00 = (2 * wave) * HW1 + r;
    const unsigned b0 = (unsigned)(r * KB + ((swz(r) ^ h) << 4));
#define TD_A(HR_, Q1_, KS_, BUF_)                                                                        \
  ({                                                                                                     \
    const int hp_ = hpv + (HR_) * HW1 + (Q1_);                                                           \
    *reinterpret_cast<const u32x4*>(smem + ((OFF_H + (BUF_) * HBUF + hp_ * KB + ((swz(hp_) ^ h) << 4)) ^ ((KS_) << 5))); \
  })
#define TD_B(K_, KS_, BUF_) (*reinterpret_cast<const u32x4*>(smem + (BUF_) * WBUF + (K_) * 32 * KB + (b0 ^ ((KS_) << 5))))
    // one k-step: 6 halo fragments (rows 2w .. 2w+2 of the halo x column shifts q = 1, 0), 9 weight fragments, 18 MFMAs; per
    // class the taps accumulate in the order of tconv2_kernel (shift slots 0, 1, 2, 3).  M0_: the product that touches an
    // accumulator first (mma_0 in the first k-step of an item)
#define TD_STEP(KS_, BUF_, M0_, WB_)                                                                     \
  {                                                                                                      \
    u32x4 A00 = TD_A(0, 0, KS_, BUF_), A01 = TD_A(0, 1, KS_, BUF_);   /* halo row 2w:   q = 1, q = 0 */   \
    u32x4 A10 = TD_A(1, 0, KS_, BUF_), A11 = TD_A(1, 1, KS_, BUF_);   /* halo row 2w+1 */                 \
    u32x4 A20 = TD_A(2, 0, KS_, BUF_), A21 = TD_A(2, 1, KS_, BUF_);   /* halo row 2w+2 */                 \
    const bool wi_ = w_left > 0 && g > 0;     /* the weight pieces of chunk g + 1 into the buffer the last barrier freed */ \
    if (wi_) TP_W_BEGIN()                                                                                \
    {                                                                                                    \
      const u32x4 B0 = TD_B(0, KS_, BUF_), B1 = TD_B(1, KS_, BUF_), B2 = TD_B(2, KS_, BUF_);              \
      M0_(acc[0][0], B0, A00); M0_(acc[1][0], B0, A10);            /* shift (1,1) */                     \
      M0_(acc[0][1], B2, A01); M0_(acc[1][1], B2, A11);            /* shift (1,0), class 1 */            \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
      if (wi_) TP_ISSUE_W(0, WB_)                                                                        \
      mma_t(acc[0][0], B1, A01); mma_t(acc[1][0], B1, A11);        /* shift (1,0), class 0 */            \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
      if (wi_) TP_ISSUE_W(1, WB_)                                                                        \
    }                                                                                                    \
    {                                                                                                    \
      const u32x4 B3 = TD_B(3, KS_, BUF_), B4 = TD_B(4, KS_, BUF_);                                       \
      M0_(acc[0][2], B4, A10); M0_(acc[1][2], B4, A20);            /* shift (0,1), class 2 */            \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
      if (wi_) TP_ISSUE_W(2, WB_)                                                                        \
      mma_t(acc[0][0], B3, A10); mma_t(acc[1][0], B3, A20);        /* shift (0,1), class 0 */            \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
      if (wi_) TP_ISSUE_W(3, WB_)                                                                        \
    }                                                                                                    \
    {                                                                                                    \
      const u32x4 B5 = TD_B(5, KS_, BUF_), B6 = TD_B(6, KS_, BUF_), B7 = TD_B(7, KS_, BUF_), B8 = TD_B(8, KS_, BUF_); \
      M0_(acc[0][3], B8, A11); M0_(acc[1][3], B8, A21);            /* shift (0,0) */                     \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
      if (wi_) { TP_ISSUE_W(4, WB_) TP_W_END() }                                                         \
      mma_t(acc[0][1], B6, A11); mma_t(acc[1][1], B6, A21);                                              \
      mma_t(acc[0][2], B7, A11); mma_t(acc[1][2], B7, A21);                                              \
      mma_t(acc[0][0], B5, A11); mma_t(acc[1][0], B5, A21);                                              \
    }                                                                                                    \
  }
    f32x16 acc[2][4];
#pragma unroll
    for (int R = 0; R < 2; R++)
#pragma unroll
      for (int c = 0; c < 4; c++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[R][c][e] = 0.f;

    TP_CURSOR()
#pragma unroll
    for (int j = 0; j < HJ; j++) TP_ISSUE_H(j, 0)
    TP_ADVANCE()
#pragma unroll
    for (int j = 0; j < HJ; j++) TP_ISSUE_H(j, 1)
    TP_ADVANCE()
    TP_W_BEGIN()
#pragma unroll
    for (int j = 0; j < WJ; j++) TP_ISSUE_W(j, 0)
    TP_W_END()
    TP_W_BEGIN()
#pragma unroll
    for (int j = 0; j < WJ; j++) TP_ISSUE_W(j, 1)
    TP_W_END()
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();
    int hpv = hp00;
    int g = 0;   // chunks multiplied so far (buffer parity)
    int iss_left = n_items * nc - 2;
    for (int i = 0; i <= n_items; i++) {
      if (i == n_items) {   // the consumers filter the last tile: keep their barriers company
        for (int c = 0; c < nc + 2; c++) bar();
        break;
      }
      for (int c = 0; c < nc; c++, g++) {
        if (TP_SKIP & 256) { bar(); continue; }   // (timing-only: the consumers alone)
        const int buf = g & 1;
        asm volatile("" : "+v"(hpv));  // (keeps the fragment addresses out of loop-invariant registers)
        if (c == 0) {
          TD_STEP(0, buf, mma_0, buf ^ 1)
        } else {
          TD_STEP(0, buf, mma_t, buf ^ 1)
        }
        // the chunk's last fragment reads are issued inside the next step; the barrier that frees the buffers comes after they
        // have returned, so that step is split by hand: loads, wait, barrier, then the MFMAs with the refill's five LDS-direct
        // pieces between them (an in-order wave cannot issue them behind a queue of MFMAs: every piece sits in a gap of two)
        {
          u32x4 A00 = TD_A(0, 0, 1, buf), A01 = TD_A(0, 1, 1, buf), A10 = TD_A(1, 0, 1, buf), A11 = TD_A(1, 1, 1, buf);
          u32x4 A20 = TD_A(2, 0, 1, buf), A21 = TD_A(2, 1, 1, buf);
          const u32x4 B0 = TD_B(0, 1, buf), B1 = TD_B(1, 1, buf), B2 = TD_B(2, 1, buf), B3 = TD_B(3, 1, buf), B4 = TD_B(4, 1, buf);
          const u32x4 B5 = TD_B(5, 1, buf), B6 = TD_B(6, 1, buf), B7 = TD_B(7, 1, buf), B8 = TD_B(8, 1, buf);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // chunk g + 1 (issued one chunk ago) has landed
          bar();                                            // ... for everybody; chunk g's buffers are free
          const bool iss = iss_left > 0;
          iss_left--;
          mma_t(acc[0][0], B0, A00); mma_t(acc[1][0], B0, A10);
          __builtin_amdgcn_sched_barrier(0);
          if (iss) TP_ISSUE_H(0, buf)
          mma_t(acc[0][1], B2, A01); mma_t(acc[1][1], B2, A11);
          __builtin_amdgcn_sched_barrier(0);
          if (iss) TP_ISSUE_H(1, buf)
          mma_t(acc[0][0], B1, A01); mma_t(acc[1][0], B1, A11);
          __builtin_amdgcn_sched_barrier(0);
          if (iss) TP_ISSUE_H(2, buf)
          mma_t(acc[0][2], B4, A10); mma_t(acc[1][2], B4, A20);
          __builtin_amdgcn_sched_barrier(0);
          if (iss) TP_ISSUE_H(3, buf)
          mma_t(acc[0][0], B3, A10); mma_t(acc[1][0], B3, A20);
          __builtin_amdgcn_sched_barrier(0);
          if (iss) TP_ISSUE_H(4, buf)
          mma_t(acc[0][3], B8, A11); mma_t(acc[1][3], B8, A21);
          __builtin_amdgcn_sched_barrier(0);
          if (iss) TP_ADVANCE()
          mma_t(acc[0][1], B6, A11); mma_t(acc[1][1], B6, A21);
          mma_t(acc[0][2], B7, A11); mma_t(acc[1][2], B7, A21);
          mma_t(acc[0][0], B5, A11); mma_t(acc[1][0], B5, A21);
        }
      }
      bar();   // "tile free": the consumers have read the previous item's tile
      // t tile -> LDS [t row][ch][t column] as bf16 (the rounding the two-launch path's HBM tensor has): classes (rp, 0) and (rp, 1)
      // interleave into 8 consecutive t columns
      if (!(TP_SKIP & 32) || u.gain == -123.f)
#pragma unroll
      for (int R = 0; R < 2; R++)
#pragma unroll
        for (int rp = 0; rp < 2; rp++) {
          const int tr = 4 * wave + 2 * R + rp;
#pragma unroll
          for (int qd = 0; qd < 4; qd++)
            *reinterpret_cast<u32x4*>(tt + (tr * 32 + r) * RS + (2 * qd + h) * 16) =
                u32x4{pack2bf(acc[R][2 * rp][4 * qd + 0], acc[R][2 * rp + 1][4 * qd + 0]),
                      pack2bf(acc[R][2 * rp][4 * qd + 1], acc[R][2 * rp + 1][4 * qd + 1]),
                      pack2bf(acc[R][2 * rp][4 * qd + 2], acc[R][2 * rp + 1][4 * qd + 2]),
                      pack2bf(acc[R][2 * rp][4 * qd + 3], acc[R][2 * rp + 1][4 * qd + 3])};
        }
      bar();   // "tile ready"
    }
#undef TP_ISSUE_W
#undef TP_W_BEGIN
#undef TP_W_END
#undef TP_CURSOR
#undef TP_ISSUE_H
#undef TP_ADVANCE
#undef TD_A
#undef TD_B
#undef TD_STEP
  } else {
    // =========================================================== consumers: FIR + epilogue of the previous item's tile
    // Wave (xt, rh): output columns 32 xt .. 32 xt + 31 of the tile (t columns 16 (xt + s) .. + 15, s = 0, 1, 2: three k-steps per
    // t row), output rows 2 + 6 rh .. 7 + 6 rh: nine t rows through a rolling window of four accumulators.
    const int wave = wave8 - NW;
    const int xt = wave & 1, rh = wave >> 1;
    const int xout = 32 * xt + r;
    const int yl0 = 2 + 6 * rh;
    const int Wo = 2 * a.W, Ho = 2 * a.H;
#define TP_CBAR() bar();
    // the three Toeplitz fragments of the lane (column xout, k = 8 h + j <-> t column 16 (xt + s) + 8 h + j)
    u32x4 Ff[3];
#pragma unroll
    for (int s3 = 0; s3 < 3; s3++) {
      const int base = 16 * (xt + s3) + 8 * h - xout + 1;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const unsigned d0 = (unsigned)(base + 2 * q), d1 = d0 + 1u;
        const unsigned lo = d0 > 3u ? 0u : ((d0 - 1u) < 2u ? 0x4040u : 0x3f80u);
        const unsigned hi = d1 > 3u ? 0u : ((d1 - 1u) < 2u ? 0x4040u : 0x3f80u);
        Ff[s3][q] = lo | (hi << 16);
      }
    }
    const float nzs = u.noise_strength * u.gain;
    const float cl = u.clamp >= 0.f ? u.clamp : 3.0e38f;
    const float dscale = 0.0625f * u.gain;
    const unsigned toff = (unsigned)(r * RS + (2 * xt + h) * 16);
    // per-item global operands: the set in use (dv, bv, sv, nzv) and the NEXT item's, requested at the top of the row loop so that
    // their latency is covered by a whole tile of work (double register set instead of a wait at the hand-over)
    f32x4 dv[4], bv[4], sv[4], dvn[4], bvn[4], svn[4];
    float nzv[6], nzn[6];
    const bool has_sv = u.out_scale != nullptr;
    PcItem cur = pc_decode(I0, CB, tiles, tiles_y), nxt = cur;
#define TP_LOAD_NEXT()                                                                                    \
  {                                                                                                       \
    _Pragma("unroll") for (int qd = 0; qd < 4; qd++) {                                                   \
      const int cho = nxt.cb * 32 + 8 * qd + 4 * h;                                                       \
      svn[qd] = has_sv ? *reinterpret_cast<const f32x4*>(u.out_scale + (long)nxt.b * a.Co + cho) : f32x4{1.f, 1.f, 1.f, 1.f}; \
      dvn[qd] = u.d ? *reinterpret_cast<const f32x4*>(u.d + (long)nxt.b * a.Co + cho) : f32x4{1.f, 1.f, 1.f, 1.f}; \
      bvn[qd] = u.bias ? *reinterpret_cast<const f32x4*>(u.bias + cho) : f32x4{0.f, 0.f, 0.f, 0.f};      \
    }                                                                                                     \
    const int Xn = 2 * nxt.tx0() + xout;                                                                  \
    const bool on_n = xout >= 2 && xout < 62 && Xn < Wo;                                                  \
    _Pragma("unroll") for (int k = 0; k < 6; k++) nzn[k] = 0.f;                                          \
    if (u.noise && on_n) {                                                                                \
      const float* nb = u.noise + (long)nxt.b * u.noise_bstride;                                          \
      _Pragma("unroll") for (int k = 0; k < 6; k++) {                                                    \
        const int Yn = 2 * nxt.ty0() + yl0 + k;                                                           \
        if (Yn < Ho) nzn[k] = nb[(long)Yn * Wo + Xn];                                                     \
      }                                                                                                   \
    }                                                                                                     \
  }
    // one t row's horizontal pass: three k-steps into a fresh accumulator
#define TP_HROW(K_)                                                                                       \
  {                                                                                                       \
    const char* trow = tt + (yl0 - 1 + (K_)) * (32 * RS) + toff;                                          \
    const u32x4 A0 = *reinterpret_cast<const u32x4*>(trow);                                               \
    const u32x4 A1 = *reinterpret_cast<const u32x4*>(trow + 32);                                          \
    const u32x4 A2 = *reinterpret_cast<const u32x4*>(trow + 64);                                          \
    f32x16 z;                                                                                             \
    _Pragma("unroll") for (int e = 0; e < 16; e++) z[e] = 0.f;                                           \
    z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A0), __builtin_bit_cast(bf16x8, Ff[0]), z, 0, 0, 0); \
    z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A1), __builtin_bit_cast(bf16x8, Ff[1]), z, 0, 0, 0); \
    z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A2), __builtin_bit_cast(bf16x8, Ff[2]), z, 0, 0, 0); \
    Hh[(K_) % 5] = z;                                                                                     \
  }
    bar();   // (the first two chunks have landed)
    for (int it = 0; it <= n_items; it++) {
      if (it == 0 || (TP_SKIP & 2)) {
        if (it < n_items) TP_LOAD_NEXT()
        for (int c = 0; c < nc; c++) TP_CBAR()
      } else {
        if (it < n_items) TP_LOAD_NEXT()
        const int X = 2 * cur.tx0() + xout;
        const bool px_on = xout >= 2 && xout < 62 && X < Wo;
        char* yb = reinterpret_cast<char*>(u.y) + (long)cur.b * Ho * Wo * a.Co * 2;
        // rolling window of FIVE row accumulators: row k + 1's products are issued before output row k - 3 is finished from rows
        // k - 3 .. k, so the matrix pipe and the LDS work behind the epilogue's vector instructions
        f32x16 Hh[5];
        TP_HROW(0)
        TP_HROW(1)
        TP_HROW(2)
        TP_HROW(3)
#pragma unroll
        for (int k = 3; k < 9; k++) {
          if (k + 1 < 9) TP_HROW(k + 1)
          __builtin_amdgcn_sched_barrier(0);
          {
            const int ko = k - 3;
            const int Y = 2 * cur.ty0() + yl0 + ko;
            const f32x16& h0 = Hh[ko % 5];
            const f32x16& h1 = Hh[(ko + 1) % 5];
            const f32x16& h2 = Hh[(ko + 2) % 5];
            const f32x16& h3 = Hh[(ko + 3) % 5];
            const float nz = nzv[ko] * nzs;
            unsigned Q[4][2];
#pragma unroll
            for (int qd = 0; qd < 4; qd++)
#pragma unroll
              for (int ip = 0; ip < 2; ip++) {
                const int e = 4 * qd + 2 * ip;
                const f32x2_t a0 = {h0[e], h0[e + 1]}, a1 = {h1[e], h1[e + 1]}, a2 = {h2[e], h2[e + 1]}, a3 = {h3[e], h3[e + 1]};
                const f32x2_t dvp = {dv[qd][2 * ip], dv[qd][2 * ip + 1]}, bvp = {bv[qd][2 * ip], bv[qd][2 * ip + 1]};
                const f32x2_t accv = (a0 + a3) + 3.f * (a1 + a2);
                f32x2_t t = accv * dvp + (bvp + nz);
                const f32x2_t ta = t * u.alpha;
                t = f32x2_t{fmaxf(t[0], ta[0]), fmaxf(t[1], ta[1])};
                f32x2_t o = f32x2_t{__builtin_amdgcn_fmed3f(t[0], -cl, cl), __builtin_amdgcn_fmed3f(t[1], -cl, cl)};
                if (has_sv) o *= f32x2_t{sv[qd][2 * ip], sv[qd][2 * ip + 1]};
                Q[qd][ip] = pack2bf(o[0], o[1]);
              }
            // lanes l and l + 32 hold the two halves of each 8-channel piece of one pixel: exchange, then each stores 32 contiguous bytes
            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
            const u32x2_t s00 = __builtin_amdgcn_permlane32_swap(Q[0][0], Q[2][0], false, false);
            const u32x2_t s01 = __builtin_amdgcn_permlane32_swap(Q[0][1], Q[2][1], false, false);
            const u32x2_t s10 = __builtin_amdgcn_permlane32_swap(Q[1][0], Q[3][0], false, false);
            const u32x2_t s11 = __builtin_amdgcn_permlane32_swap(Q[1][1], Q[3][1], false, false);
            if (px_on && Y < Ho && (!(TP_SKIP & 4) || u.gain == -123.f)) {
              char* yp = yb + (((long)Y * Wo + X) * a.Co + cur.cb * 32 + 16 * h) * 2;
              *reinterpret_cast<u32x4*>(yp) = u32x4{s00[0], s01[0], s00[1], s01[1]};
              *reinterpret_cast<u32x4*>(yp + 16) = u32x4{s10[0], s11[0], s10[1], s11[1]};
            }
          }
          // this output row's share of the item's chunk barriers
          for (int q = ((k - 3) * nc) / 6; q < ((k - 2) * nc) / 6; q++) TP_CBAR()
        }
      }
      bar();   // "tile free"
      if (it < n_items) {   // the operands requested a tile ago become the current ones
        cur = nxt;
        pc_next(nxt, CB, tiles_x, tiles_y);
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
          sv[qd] = svn[qd];
          dv[qd] = dvn[qd] * dscale;
          bv[qd] = bvn[qd] * u.gain;
        }
#pragma unroll
        for (int k = 0; k < 6; k++) nzv[k] = nzn[k];
      }
      bar();   // "tile ready"
    }
#undef TP_LOAD_NEXT
#undef TP_HROW
#undef TP_CBAR
  }
}

bool tconv_pc_supported(int dtype, int Ci, int Co, int H, int W) {
  return dtype == MAUA_BF16 && Ci % 32 == 0 && Ci >= 64 && Co % 32 == 0 && H >= 16 && W >= 32 && (long)H * W * Ci * 2 < (1L << 32) &&
         16L * Co * Ci * 2 < (1L << 32);
}

// the whole up-layer: a = the transposed convolution's arguments (x already multiplied by the styles, w from
// launch_prep_tconv_weights; y unused), u = the FIR / epilogue arguments (t unused; lrelu with 0 <= alpha <= 1, gain > 0)
int launch_tconv_pc(hipStream_t stream, const ConvArgs& a, const UpfirArgs& u) {
  MAUA_REQUIRE(tconv_pc_supported(MAUA_BF16, a.Ci, a.Co, a.H, a.W), "tconv_pc: unsupported shape");
  MAUA_REQUIRE(u.act == MAUA_ACT_LRELU && u.alpha >= 0.f && u.alpha <= 1.f && u.gain > 0.f, "tconv_pc: lrelu epilogue only");
  if (a.B == 0) return MAUA_OK;
  const long tiles = (long)((a.H + UPR - 1) / UPR) * ((a.W + 29) / 30), CB = a.Co / 32;
  const long total = tiles * a.B * CB;
  MAUA_REQUIRE(total < (1L << 31), "tconv_pc: too many items");
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    MAUA_HIP_CHECK(hipGetDevice(&dev));
    MAUA_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)tconv_pc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
  }
  const long wgs = std::min<long>(cus, total);
  const int per = (int)((total + wgs - 1) / wgs);
  const int grid = (int)((total + per - 1) / per);
  hipLaunchKernelGGL(tconv_pc_kernel, dim3((unsigned)grid), dim3(512), SMEM, stream, a, u, per, (int)total);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace maua
