// Minimal-MAC up-layer in ONE kernel: t = conv_transpose2d(x * s, W, stride 2) on the matrix cores, the 4 x 4 FIR and the
// layer epilogue straight from the workgroup's LDS tile of t - the (2H+1) x (2W+1) tensor never reaches HBM.
//
// Replaces (reference): ops.py:211-225 (conv2d_resample, up = 2: conv_transpose2d, then upfirdn2d pad 1 gain 4) + :184-185
// noise + :65-84 bias_act, i.e. what modconv_tconv_dma.hip + upfir.hip do in two launches with t written and re-read
// (round 2: the pair ran at 2.6x its algorithmic bytes; the 256^2 -> 512^2 layer alone was 12 % of the step).
//
// A workgroup computes t for 8 x 32 positions x 32 channels exactly like tconv_dma_kernel (same K loop, same LDS-direct
// pipeline, same accumulation order: t is bit-identical), leaves the tile in LDS as bf16 - the rounding the HBM tensor had -
// and evaluates the FIR + epilogue with upfir_epilogue_kernel's arithmetic (same operation order: the layer output is
// bit-identical to the two-launch path).  An output pixel needs t rows Y-1 .. Y+2 and columns X-1 .. X+2, so a tile of
// 16 x 64 t values yields 12 x 60 outputs: tiles advance by 6 x 30 positions and overlap by a one-position frame
// (1.42x the transposed convolution's MACs - the price of not exchanging t between workgroups; the matrix cores absorb it
// where the round trip was HBM-bound).  A row-walk form of this kernel (8 position rows per step, the three t rows a step needs
// from the one above carried in registers as horizontal sums: 32 / 30 of the MACs instead of 1.42x) was built and verified
// bit-identical in round 3, and measured SLOWER (4.20 vs 3.48 ms on the 256^2 -> 512^2 layer at B = 128): 128 accumulators +
// 60 fragment registers + 48 carry registers do not fit 256 VGPRs at two workgroups per CU (228 bytes / lane of scratch),
// and the carry does not fit LDS beside two resident workgroups either (75 KB + 12 KB each).  Zero halo pixels on every side make positions outside the image (and the thin
// last row / column of positions, which the two-launch path needs an extra kernel for) come out right by themselves.
#include "common.h"
#include "internal.h"

namespace maua {

namespace {

constexpr int PTW = 32;                            // position columns per workgroup (rows: template parameter, 8 or 16)
constexpr int HW1 = PTW + 1;                       // halo columns
constexpr int KB = 64;                             // bytes of K per LDS row (32 bf16 channels = one chunk)
constexpr int WROWS = 9 * 32;
constexpr int WBUF = WROWS * KB;
constexpr int OFF_H = 2 * WBUF;
constexpr int ES = 128 * 2;                        // t tile: [position][class * 32 + ch] bf16, row stride (unpadded: XOR-swizzled)

__device__ __constant__ const int kSlotF[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};  // as modconv_tconv.hip
__device__ __constant__ const int kClsF[9] = {0, 0, 1, 0, 2, 0, 1, 2, 3};

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

// The t tile, [position p = (tr >> 1) * 32 + (tc >> 1)][class (tr & 1) * 2 + (tc & 1)][32 channels] bf16, rows of 256 bytes
// swizzled on two levels (round 5; the padded 272-byte rows of rounds 3-4 had 35 % of this kernel's LDS cycles in conflicts):
//   the 64-byte class block sits at slot  cls ^ (p & 3)        - the four 2-column strips of a ds_read_b128 lane group differ in p & 3
//   its 16-byte piece pc at               pc ^ ((p >> 2) & 3)  - 16 consecutive positions of a ds_write_b64 lane group take the 16
//                                                                 (slot, piece) combinations: both sides conflict-free
// column part of the byte offset of piece pc of t[.][tc]; the row part is (tr >> 1) * 32 * ES and XOR ((tr & 1) << 7)
__device__ __forceinline__ int t_col(int tc, int pc) {
  const int p = tc >> 1;
  return p * ES + ((((tc & 1) ^ (p & 3)) << 6) | ((pc ^ ((p >> 2) & 3)) << 4));
}

// the two horizontal [1,3,3,1] sums of one t row for the output columns xl, xl + 1 (upfir_hrow's arithmetic); tcol[rx] =
// t_col(xl - 1 + rx, pc)
template <typename F>
__device__ __forceinline__ void fir_hrow(const char* tile, int tr, const int (&tcol)[5], f32x2_t (&h)[2][4]) {
  f32x2_t c[5][4];
  const int rowb = (tr >> 1) * (PTW * ES), rsw = (tr & 1) << 7;
#pragma unroll
  for (int rx = 0; rx < 5; rx++) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(tile + rowb + (tcol[rx] ^ rsw));
#pragma unroll
    for (int k = 0; k < 4; k++) c[rx][k] = f32x2_t{Fmt16<F>::lo(v[k]), Fmt16<F>::hi(v[k])};
  }
#pragma unroll
  for (int e = 0; e < 4; e++) {
    h[0][e] = (c[0][e] + c[3][e]) + 3.f * (c[1][e] + c[2][e]);
    h[1][e] = (c[1][e] + c[4][e]) + 3.f * (c[2][e] + c[3][e]);
  }
}

}  // namespace

// PTH_ = 8: 4 waves, 75 KB of LDS, two workgroups per CU, 6 x 30 useful positions of 8 x 32 (1.42x MACs).  The kernel is
// written for any even PTH_; PTH_ = 16 (8 waves, 136 KB, ONE workgroup per CU, 14 x 30 of 16 x 32 = 1.22x MACs) measured
// 3.78 vs 3.50 ms on the 256^2 -> 512^2 layer at B = 128 - two workgroups out of phase are worth more than the saved rows.
template <int PTH_, typename F>
__global__ __launch_bounds__(PTH_ * 32, PTH_ == 8 ? 2 : 1) void tconv_fir_kernel(ConvArgs a, UpfirArgs u) {
  constexpr int PTH = PTH_, UPR = PTH - 2;           // position rows computed / useful
  constexpr int HPX = (PTH + 1) * HW1, HBUF = HPX * KB;
  constexpr int NW = PTH / 2, NT = NW * 64;
  constexpr int WJ = (WBUF / 1024 + NW - 1) / NW, HJ = (HPX * 4 + NT - 1) / NT;
  constexpr int FG = NT / 128, FROWS = 2 * UPR / FG;  // FIR thread groups of 120, output rows per group (6 or 7)
  static_assert(FG * FROWS == 2 * UPR, "output rows must split evenly over the FIR groups");
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_off(smem));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  // Workgroup order (1-D grid): the dispatcher places block L on XCD L % 8, each XCD has its own L2.  The channel
  // blocks of one (tile, sample) read the same input halo, so they run back to back ON ONE XCD (cb fastest inside an
  // XCD, in groups of <= 8 blocks = <= 2.4 MB of weights, which stay L2-resident next to the halos): x is fetched from
  // HBM once per group instead of once per channel block.  Placement is a speed matter only.
  const int tiles_x = (a.W + 29) / 30, tiles = tiles_x * ((a.H + UPR - 1) / UPR), CB = a.Co >> 5;
  const int cbg = CB < 8 ? CB : 8, n_ts = tiles * a.B, per_group = ((n_ts + 7) >> 3) * 8 * cbg;
  const int L = blockIdx.x, grp = L / per_group, Lg = L - grp * per_group;
  const int xcd = Lg & 7, idx = Lg >> 3;
  const int cb = grp * cbg + idx % cbg, ts = (idx / cbg) * 8 + xcd;
  if (ts >= n_ts) return;
  const int b = ts / tiles, tile = ts - b * tiles;
  const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
  const int ty0 = tyi * UPR - 1, tx0 = txi * 30 - 1;   // first position of the tile (a one-position frame around UPR x 30)
  const char* xb = reinterpret_cast<const char*>(a.x) + (long)b * a.x_bstride * 2;
  const char* wp = reinterpret_cast<const char*>(a.w);

  // halo sources (pixels above / left of the image are never loaded: zeroed once)
  unsigned hoff[HJ];
#pragma unroll
  for (int j = 0; j < HJ; j++) {
    const int P = (wave + NW * j) * 64 + lane;
    const int hp = P >> 2, q = (P & 3) ^ swz(hp);
    const int py = (hp * 1986) >> 16;  // hp / 33 for hp < 561
    const int px = hp - py * HW1;
    const int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
    hoff[j] = 0xffffffffu;
    if (P < HPX * 4) {
      if (gy >= 0 && gx >= 0 && gy < a.H && gx < a.W) {
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
    woff[j] = (unsigned)((((((kSlotF[k] * CB + cb) * 4 + kClsF[k]) * 32 + n) * a.Ci) + q * 8) * 2);
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

  // ---- the epilogue's global operands are requested BEFORE the tile is written: per-channel vectors and the thread's noise values
  // (one float2 per output row) are in flight across the two barriers and the first three FIR rows instead of being waited for row
  // by row (every wait would also have drained the row's stores: 6 exposed round trips per thread, 1.88 of the kernel's 3.5 ms).
  // Local t row / column 0 = global t row 2 ty0 / column 2 tx0; local output (yl, xl) = global (2 ty0 + yl, 2 tx0 + xl) reads local
  // t rows yl-1 .. yl+2: yl in [2, 14), xl in [2, 62).  A thread owns a 2-column strip x one 16-byte channel piece and walks 6
  // output rows: 30 strips x 4 pieces x 2 row halves.
  const int fhalf = tid / 120, fw = tid - fhalf * 120;
  const int strip = fw >> 2, pc = fw & 3;
  const int xl = 2 + 2 * strip, yl0 = 2 + FROWS * fhalf;
  const int X = 2 * tx0 + xl, Wo = 2 * a.W, Ho = 2 * a.H;
  const int cho = cb * 32 + pc * 8;
  const bool fir_on = tid < 120 * FG && X < Wo;
  f32x2_t dv[4], bv[4], sv[4];
  float2 nzv[FROWS];
#pragma unroll
  for (int k = 0; k < FROWS; k++) nzv[k] = make_float2(0.f, 0.f);
  if (fir_on) {
#pragma unroll
    for (int e4 = 0; e4 < 8; e4 += 4) {
      const float4 s4 = u.out_scale ? *reinterpret_cast<const float4*>(u.out_scale + (long)b * a.Co + cho + e4)
                                    : make_float4(1.f, 1.f, 1.f, 1.f);
      sv[e4 / 2] = f32x2_t{s4.x, s4.y}; sv[e4 / 2 + 1] = f32x2_t{s4.z, s4.w};
      const float4 d4 = u.d ? *reinterpret_cast<const float4*>(u.d + (long)b * a.Co + cho + e4) : make_float4(1.f, 1.f, 1.f, 1.f);
      const float4 b4 = u.bias ? *reinterpret_cast<const float4*>(u.bias + cho + e4) : make_float4(0.f, 0.f, 0.f, 0.f);
      dv[e4 / 2] = f32x2_t{d4.x, d4.y}; dv[e4 / 2 + 1] = f32x2_t{d4.z, d4.w};
      bv[e4 / 2] = f32x2_t{b4.x, b4.y}; bv[e4 / 2 + 1] = f32x2_t{b4.z, b4.w};
    }
    if (u.noise) {
      const float* nb = u.noise + (long)b * u.noise_bstride;
#pragma unroll
      for (int k = 0; k < FROWS; k++) {
        const int Y = 2 * ty0 + yl0 + k;
        if (Y < Ho) nzv[k] = *reinterpret_cast<const float2*>(nb + (long)Y * Wo + X);
      }
    }
  }
  // ---- t tile -> LDS [position][class * 32 + ch] as bf16 (the rounding of the two-launch path's HBM tensor)
  __syncthreads();
  char* tt = smem;
#pragma unroll
  for (int R = 0; R < 2; R++) {
    const int m = (2 * wave + R) * 32 + r;   // (m & 3 = r & 3, (m >> 2) & 3 = (r >> 2) & 3: the swizzle terms are per-lane constants)
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int qd = 0; qd < 4; qd++)
        *reinterpret_cast<uint2*>(tt + m * ES + (((c ^ (r & 3)) << 6) | ((qd ^ ((r >> 2) & 3)) << 4) | (h << 3))) =
            make_uint2(Fmt16<F>::pack2(acc[R][c][qd * 4 + 0], acc[R][c][qd * 4 + 1]), Fmt16<F>::pack2(acc[R][c][qd * 4 + 2], acc[R][c][qd * 4 + 3]));
  }
  __syncthreads();
  // ---- FIR + epilogue (upfir_epilogue_kernel's arithmetic)
  {
    if (fir_on) {
#pragma unroll
      for (int e = 0; e < 4; e++) { dv[e] *= 0.0625f * u.gain; bv[e] *= u.gain; }
      const float nzs = u.noise_strength * u.gain * (u.noise_scale ? u.noise_scale[b] : 1.f);
      const float cl = u.clamp >= 0.f ? u.clamp : 3.0e38f;
      char* yb = reinterpret_cast<char*>(u.y) + (long)b * Ho * Wo * a.Co * 2;
      int tcol[5];
#pragma unroll
      for (int rx = 0; rx < 5; rx++) tcol[rx] = t_col(xl - 1 + rx, pc);
      f32x2_t hr[4][2][4];
      fir_hrow<F>(tt, yl0 - 1, tcol, hr[0]);
      fir_hrow<F>(tt, yl0, tcol, hr[1]);
      fir_hrow<F>(tt, yl0 + 1, tcol, hr[2]);
#pragma unroll
      for (int k = 0; k < FROWS; k++) {
        fir_hrow<F>(tt, yl0 + k + 2, tcol, hr[(k + 3) & 3]);
        const int Y = 2 * ty0 + yl0 + k;
        if (Y < Ho) {
          const float nz[2] = {nzv[k].x * nzs, nzv[k].y * nzs};
#pragma unroll
          for (int j = 0; j < 2; j++) {
            f32x2_t o[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
              const f32x2_t accv = (hr[k & 3][j][e] + hr[(k + 3) & 3][j][e]) + 3.f * (hr[(k + 1) & 3][j][e] + hr[(k + 2) & 3][j][e]);
              f32x2_t t = accv * dv[e] + (bv[e] + nz[j]);
              const f32x2_t ta = t * u.alpha;
              t = f32x2_t{fmaxf(t[0], ta[0]), fmaxf(t[1], ta[1])};
              o[e] = f32x2_t{__builtin_amdgcn_fmed3f(t[0], -cl, cl), __builtin_amdgcn_fmed3f(t[1], -cl, cl)} * sv[e];
            }
            *reinterpret_cast<u32x4*>(yb + (((long)Y * Wo + X + j) * a.Co + cho) * 2) =
                u32x4{Fmt16<F>::pack2(o[0][0], o[0][1]), Fmt16<F>::pack2(o[1][0], o[1][1]), Fmt16<F>::pack2(o[2][0], o[2][1]), Fmt16<F>::pack2(o[3][0], o[3][1])};
          }
        }
      }
    }
  }
}

bool tconv_fir_supported(int dtype, int Ci, int Co, int H, int W) {
  return (dtype == MAUA_BF16 || dtype == MAUA_F16) && Ci % 32 == 0 && Co % 32 == 0 && H >= 16 && W >= 32 && (long)H * W * Ci * 2 < (1L << 32) &&
         16L * Co * Ci * 2 < (1L << 32);
}

// the whole up-layer: a = the transposed convolution's arguments (x already multiplied by the styles, w from
// launch_prep_tconv_weights; y unused), u = the FIR / epilogue arguments (t unused; lrelu with 0 <= alpha <= 1, gain > 0)
int launch_tconv_fir(hipStream_t stream, const ConvArgs& a, const UpfirArgs& u, int dtype) {
  MAUA_REQUIRE(tconv_fir_supported(dtype, a.Ci, a.Co, a.H, a.W), "tconv_fir: unsupported shape");
  MAUA_REQUIRE(u.act == MAUA_ACT_LRELU && u.alpha >= 0.f && u.alpha <= 1.f && u.gain > 0.f, "tconv_fir: lrelu epilogue only");
  MAUA_REQUIRE(!u.noise || (((uintptr_t)u.noise % 8) == 0 && u.noise_bstride % 2 == 0), "tconv_fir: noise must be 8-byte aligned");
  if (a.B == 0) return MAUA_OK;
  const int pth = 8, upr = pth - 2;
  const int tiles = ((a.H + upr - 1) / upr) * ((a.W + 29) / 30), CB = a.Co / 32, cbg = CB < 8 ? CB : 8;
  MAUA_REQUIRE(CB % cbg == 0, "tconv_fir: channel blocks must split into groups of 8");
  const long n_ts = (long)tiles * a.B, grid = ((n_ts + 7) / 8) * 8 * cbg * (CB / cbg);
  MAUA_REQUIRE(grid < (1L << 31), "tconv_fir: grid too large");
  const size_t smem = std::max<size_t>((size_t)2 * WBUF + 2 * (pth + 1) * HW1 * KB, (size_t)pth * PTW * ES);
  if (dtype == MAUA_F16) {
    MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)tconv_fir_kernel<8, f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL((tconv_fir_kernel<8, f16_t>), dim3((unsigned)grid), dim3(256), smem, stream, a, u);
  } else {
    MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)tconv_fir_kernel<8, bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL((tconv_fir_kernel<8, bf16_t>), dim3((unsigned)grid), dim3(256), smem, stream, a, u);
  }
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace maua
