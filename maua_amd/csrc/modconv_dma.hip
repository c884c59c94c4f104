// Modulated 3x3 convolution (up = 1, bf16) for the MFMA-bound middle of the network, built around LDS-direct loads.
//
// Replaces (reference): ops.py:146-186 modulated_conv2d + :189-233 conv2d_resample (up = 1 branch) + :65-84 bias_act
// (as called from stylegan2.py:238-250), for the layers whose input has ALREADY been multiplied by the layer's styles
// (x' = x * s[b, ci]: the reference's own w = weight * styles, moved to the other operand; the producing layer's
// epilogue applies it for free, see upfir.hip / synth.hip).  With the modulation out of the staging path neither
// operand needs a register on its way from HBM/L2 to LDS, and the K loop becomes a plain software pipeline:
//
//   stage s = (64-channel chunk c, tap t):  A = a shifted window of the halo tile of chunk c (staged once per chunk,
//                                           re-read for the 9 taps), B = the tap's [BN x 64] weight slice
//   * both operands arrive by global_load_lds_dwordx4 (no VGPRs in flight, no ds_write): weights two stages ahead
//     into a 2-slot ring, the next chunk's halo spread over the first stages of the current chunk into the other of
//     two halo buffers;
//   * ONE barrier per stage, placed before the stage's last k-step: by then every wave has issued (and waited for)
//     its last fragment reads of the stage, so the slot can be refilled right behind the barrier, and the first
//     fragments of stage s+1 are requested before the last MFMAs of stage s are issued - the matrix pipe does not
//     drain at stage boundaries;
//   * 8 waves, each owning a (WM*32 pixels) x (WN*32 channels) accumulator block (128 x 64 for the 256-channel N
//     tile): 0.75 ds_read_b128 per MFMA instead of the 1.5 of the 16-wave / 64 x 32 blocks of modconv.hip.
// LDS rows are 128 bytes, unpadded (an LDS-direct load fills 1 KB linearly); the 16-byte pieces of row p sit at piece
// index q ^ ((p >> 1) & 7), applied on the SOURCE address of the load and on the fragment read, which makes every
// ds_read_b128 of 32 consecutive rows bank-conflict-free.  Halo pixels outside the image are not loaded: their
// rows are zeroed once before the loop.  Layout of the 149 KB (BN = 256): W[0] | W[1] | H[0] | H[1]; the epilogue tile reuses it.
#include "common.h"
#include "internal.h"

// K-loop schedule experiments (bit mask; results do not depend on it - scripts/experiments/README.md, round 5):
//   1  static priority for the second-dispatched half of the workgroup (waves >= NW / 2: the younger wave of every SIMD)
//   2  the first half issues its share of the stage's LDS-direct loads half a stage later (between the k-steps of the NEXT stage
//      instead of right behind the barrier), so that the two waves of a SIMD do not both sit in their load-issue block at once
#ifndef MAUA_DMA_SCHED
#define MAUA_DMA_SCHED 0
#endif

namespace maua {

namespace {

constexpr int TH = 8, TW = 32, HW2 = TW + 2, HALO_PX = (TH + 2) * HW2;  // 8 x 32 output pixels, 10 x 34 halo
// K bytes per LDS row (KB, template parameter): 128 (64 bf16 channels; one workgroup per CU) or 64 (32 channels: half
// the halo, two workgroups per CU - the short-K layers, whose per-tile prologue / epilogue otherwise idles the CU)

__device__ __forceinline__ unsigned lds_off(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
// LDS-direct loads; lds_dst must be wave-uniform (SGPR).  M0 is written in the same statement that uses it.
__device__ __forceinline__ void dma16_v(const void* gptr, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}

template <typename F>
__device__ __forceinline__ void mma(f32x16& acc, const u32x4& a, const u32x4& b) {
  Mma16<F>::step(acc, a, b);
}

}  // namespace

// x' = bf16(x * s): only for callers whose producer could not apply the styles (operator-level entry point, hooks)
template <typename F>
__global__ __launch_bounds__(256) void premod_nhwc_kernel(const uint16_t* __restrict__ x, long x_bstride,
                                                          const float* __restrict__ s, uint16_t* __restrict__ y, int B,
                                                          long HW, int Ci) {
  const int ppp = Ci / 8;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * HW * ppp) return;
  const int pc = (int)(idx % ppp);
  const long bp = idx / ppp;
  const long p = bp % HW;
  const int b = (int)(bp / HW);
  const float4 s0 = *reinterpret_cast<const float4*>(s + (long)b * Ci + pc * 8);
  const float4 s1 = *reinterpret_cast<const float4*>(s + (long)b * Ci + pc * 8 + 4);
  const u32x4 v = *reinterpret_cast<const u32x4*>(x + (long)b * x_bstride + p * Ci + pc * 8);
  u32x4 o;
  o[0] = Fmt16<F>::pack2(Fmt16<F>::lo(v[0]) * s0.x, Fmt16<F>::hi(v[0]) * s0.y);
  o[1] = Fmt16<F>::pack2(Fmt16<F>::lo(v[1]) * s0.z, Fmt16<F>::hi(v[1]) * s0.w);
  o[2] = Fmt16<F>::pack2(Fmt16<F>::lo(v[2]) * s1.x, Fmt16<F>::hi(v[2]) * s1.y);
  o[3] = Fmt16<F>::pack2(Fmt16<F>::lo(v[3]) * s1.z, Fmt16<F>::hi(v[3]) * s1.w);
  *reinterpret_cast<u32x4*>(y + bp * Ci + pc * 8) = o;
}

int launch_premod_nhwc(hipStream_t stream, const void* x, long x_bstride, const float* s, void* y, int B, long HW, int Ci, int dtype) {
  MAUA_REQUIRE(Ci % 8 == 0, "premod: Ci must be a multiple of 8");
  if (B == 0) return MAUA_OK;
  const long n = (long)B * HW * (Ci / 8);
  if (dtype == MAUA_F16)
    hipLaunchKernelGGL(premod_nhwc_kernel<f16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const uint16_t*)x, x_bstride, s,
                       (uint16_t*)y, B, HW, Ci);
  else
    hipLaunchKernelGGL(premod_nhwc_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const uint16_t*)x, x_bstride, s,
                       (uint16_t*)y, B, HW, Ci);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// PSUM: also emit ConvArgs.psum (its own instantiation: the 16 accumulators of the copy-out loop cost the 128-register
// variants a few spilled registers, which the StyleGAN2 path's launches do not pay)
template <int WAVES_M, int WAVES_N, int WM, int WN, int TPS, int KB, bool PSUM = false, bool ODDK = false, typename F = bf16_t>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, KB == 64 ? 4 : 2) void modconv_dma_kernel(ConvArgs a) {
  constexpr int NW = WAVES_M * WAVES_N, NT = NW * 64;
  static_assert(WAVES_M * WM == TH, "the M tile is 8 image rows of 32 pixels");
  constexpr int BM = TH * TW, BN = WAVES_N * WN * 32;
  constexpr int KCB = KB, KC = KB / 2;                  // bytes / bf16 channels of K per LDS row (= per chunk)
  constexpr int PPR = KB / 16, PSH = KB == 128 ? 3 : 2; // 16-byte pieces per row
  constexpr int HB = HALO_PX * KCB;                     // one halo buffer
  constexpr int TB = BN * KCB;                          // bytes of one tap's weight slice (BN rows)
  constexpr int WB = TPS * TB;                          // bytes of one weight stage (TPS taps)
  constexpr int WPIECES = TB / 1024;                    // 1 KB weight load instructions per tap (all waves together)
  constexpr int WJ = (WPIECES + NW - 1) / NW;           // ... per wave (narrow N tiles: only the first WPIECES waves load)
  constexpr int HJ = (HALO_PX * PPR + NT - 1) / NT;     // halo load instructions per wave per chunk
  constexpr int KSPT = KB / 32;                         // 32-byte k-steps per tap
  constexpr int Q = KSPT * TPS;                         // k-steps per stage
  static_assert(TB % 1024 == 0 && (WPIECES % NW == 0 || WPIECES < NW) && (HJ == 6 || HJ == 3) && (TPS == 1 || TPS == 2) &&
                    (Q == 4 || Q == 8), "stage split");
  constexpr int OFF_H = 2 * WB;
  // piece p of row n sits at piece index p ^ swz(n): 8 rows x 8 pieces or 16 rows x 4 pieces tile one 1 KB bank period
#define MAUA_SWZ(N_) (KB == 128 ? (((N_) >> 1) & 7) : (((N_) >> 2) & 3))
  constexpr int ES = BN * 2 + 16, PPP = BN / 8;         // epilogue tile row stride, 16-byte pieces per pixel
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_off(smem));

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int r = lane & 31, h = lane >> 5;
  const int tiles_x = (a.W + TW - 1) >> 5;   // (narrow plain convolutions may overhang the image: their stores are masked)
  const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
  const int ty0 = tyi * TH, tx0 = txi * TW;
  const int b = blockIdx.y, n0 = blockIdx.z * BN;
  const char* xb = reinterpret_cast<const char*>(a.x) + (long)b * a.x_bstride * 2;
  const char* wp = reinterpret_cast<const char*>(a.w);
  const int xps = a.x_pstride ? a.x_pstride : a.Ci;   // elements between pixels (channel-sliced inputs: a prefix of a wider buffer)

  // ---- sources of this lane's LDS-direct loads (fixed for the whole K loop apart from the chunk offset)
  // halo: instruction ii = wave + NW j covers pieces [64 ii, 64 ii + 64) of the [340 px][8 pieces] buffer
  // (out-of-image pixels are never loaded: their LDS rows are zeroed once, below, and keep that value)
  unsigned hoff[HJ];
#pragma unroll
  for (int j = 0; j < HJ; j++) {
    const int P = (wave + NW * j) * 64 + lane;
    const int hp = P >> PSH, q = (P & (PPR - 1)) ^ MAUA_SWZ(hp);
    const int py = (hp * 1928) >> 16;  // hp / 34 for hp < 340
    const int px = hp - py * HW2;
    const int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
    const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    hoff[j] = 0xffffffffu;
    if (P < HALO_PX * PPR) {
      if (in) {
        // (x_up2: the input is the nearest-neighbour x2 up-sampling of a half-size tensor - read the source pixel directly)
        hoff[j] = a.x_up2 ? (unsigned)((((gy >> 1) * (a.W >> 1) + (gx >> 1)) * xps + q * 8) * 2)
                          : (unsigned)(((gy * a.W + gx) * xps + q * 8) * 2);
      } else {
        *reinterpret_cast<u32x4*>(smem + OFF_H + P * 16) = u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(smem + OFF_H + HB + P * 16) = u32x4{0u, 0u, 0u, 0u};
      }
    }
  }
  // weights: instruction ii fills rows 8 ii .. 8 ii + 7 of a tap's [BN][8 pieces] slice
  unsigned woff[WJ];
#pragma unroll
  for (int j = 0; j < WJ; j++) {
    const int row = min((1024 / KB) * (wave + NW * j) + (lane >> PSH), BN - 1);
    const int q = (lane & (PPR - 1)) ^ MAUA_SWZ(row);
    woff[j] = (unsigned)((row * a.Ci + q * 8) * 2);
  }
  const long tap_stride = (long)a.Co * a.Ci * 2;  // bytes between taps ([tap][Co][Ci])
  const char* wtile = wp + (long)n0 * a.Ci * 2;
  const int n_chunks = a.Ci / KC;
  // chunks that are read from memory: a layer whose K was padded with zero weights (Ci_read < Ci) skips the halo loads of the
  // padding - those MFMAs multiply whatever finite values an earlier chunk left in the halo buffer by zero
  const int n_hchunks = a.Ci_read ? (a.Ci_read + KC - 1) / KC : n_chunks;

  // one tap (chunk C_, tap T_) into slice J_ of weight stage buffer BUF_
#define MAUA_ISSUE_WTAP(C_, T_, BUF_, J_)                                                                \
  {                                                                                                      \
    const char* ws_ = wtile + (long)(T_) * tap_stride + (long)(C_) * (KC * 2);                           \
    _Pragma("unroll") for (int jj = 0; jj < WJ; jj++)                                                   \
        if (WPIECES >= NW || wave + NW * jj < WPIECES)                                                   \
          dma16_s(ws_, woff[jj], lds0 + (BUF_) * WB + (J_) * TB + (wave + NW * jj) * 1024);              \
  }
  // ODDK (two taps per stage, an ODD number of chunks: plain convolutions on the 96 / 160-channel prefixes of a dense-block
  // buffer, super.hip): the last period has one chunk.  Its stage 4 pairs the chunk's last tap with a tap that does not exist:
  // that weight slice is zeroed in LDS instead of loaded (the halo buffer it multiplies still holds an earlier chunk - finite
  // values x 0), and the period ends there instead of running four more stages of padding.
#define MAUA_ZERO_WTAP(BUF_, J_)                                                                         \
  {                                                                                                      \
    _Pragma("unroll") for (int jj = 0; jj < WJ; jj++)                                                   \
        if (WPIECES >= NW || wave + NW * jj < WPIECES)                                                   \
          *reinterpret_cast<u32x4*>(smem + (BUF_) * WB + (J_) * TB + (wave + NW * jj) * 1024 + lane * 16) = u32x4{0u, 0u, 0u, 0u}; \
  }
  // the stage at position K_ of the period that starts at chunk CC_ (positions >= 9 belong to the next period)
#define MAUA_ISSUE_WSTAGE(CC_, K_, BUF_)                                                                 \
  {                                                                                                      \
    _Pragma("unroll") for (int j_ = 0; j_ < TPS; j_++) {                                                \
      const int lp_ = ((K_) % 9) * TPS + j_;                                                             \
      const int c_ = (CC_) + ((K_) / 9) * TPS + lp_ / 9;                                                 \
      if (c_ < n_chunks) MAUA_ISSUE_WTAP(c_, lp_ % 9, BUF_, j_)                                          \
      else if (ODDK && c_ == n_chunks && lp_ == 9) MAUA_ZERO_WTAP(BUF_, j_)                              \
    }                                                                                                    \
  }
#define MAUA_ISSUE_H(J_, C_)                                                                             \
  {                                                                                                      \
    if ((C_) < n_hchunks && hoff[J_] != 0xffffffffu)                                                     \
      dma16_s(xb + (long)(C_) * (KC * 2), hoff[J_], lds0 + OFF_H + ((C_) & 1) * HB + (wave + NW * (J_)) * 1024); \
  }

  // ---- fragment addresses: A (pixels) from the halo, B (channels) from the weight stage
  const int hp00 = (wm * WM + 1) * HW2 + r + 1;                          // halo pixel of block row 0, centre tap
  const unsigned b0 = (unsigned)(((wn * WN) * 32 + r) * KCB + ((MAUA_SWZ(r) ^ h) << 4));
  // k-step Q_ of the stage at position K_ of the period starting at chunk CC_ (weight stage buffer WBUF_)
#define MAUA_LOAD_FRAGS(AF_, BF_, CC_, K_, Q_, WBUF_)                                                    \
  {                                                                                                      \
    const int lp_ = (K_) * TPS + (Q_) / KSPT, t_ = lp_ % 9, ks_ = (Q_) % KSPT;                           \
    const int hb_ = ((CC_) + lp_ / 9) & 1;                                                               \
    _Pragma("unroll") for (int i = 0; i < WM; i++) {                                                    \
      const int hp_ = hpv + (i + t_ / 3 - 1) * HW2 + (t_ % 3 - 1);                                       \
      const unsigned o_ = OFF_H + hb_ * HB + hp_ * KCB + ((MAUA_SWZ(hp_) ^ h) << 4);                     \
      AF_[i] = *reinterpret_cast<const u32x4*>(smem + (o_ ^ (ks_ << 5)));                                \
    }                                                                                                    \
    _Pragma("unroll") for (int j = 0; j < WN; j++)                                                      \
        BF_[j] = *reinterpret_cast<const u32x4*>(smem + (WBUF_) * WB + ((Q_) / KSPT) * TB + j * 32 * KCB + (b0 ^ (ks_ << 5))); \
  }
#define MAUA_MMA(AF_, BF_)                                                                               \
  _Pragma("unroll") for (int i = 0; i < WM; i++) _Pragma("unroll") for (int j = 0; j < WN; j++)         \
      mma<F>(acc[i][j], BF_[j], AF_[i]);  /* rows = channels, columns = pixels */

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; i++)
#pragma unroll
    for (int j = 0; j < WN; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  // ---- prologue: halo of chunk 0 (and, with two taps per stage, the first third of chunk 1's), weight stages 0 and 1
#pragma unroll
  for (int j = 0; j < HJ; j++) MAUA_ISSUE_H(j, 0)
  if constexpr (TPS == 2) {
#pragma unroll
    for (int j = 0; j < HJ / 3; j++) MAUA_ISSUE_H(j, 1)
  }
  MAUA_ISSUE_WSTAGE(0, 0, 0)
  MAUA_ISSUE_WSTAGE(0, 1, 1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  u32x4 af[WM], bf[WN], af1[WM], bf1[WN];
  int hpv = hp00;
  MAUA_LOAD_FRAGS(af, bf, 0, 0, 0, 0)

  if constexpr ((MAUA_DMA_SCHED & 1) != 0) {
    if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
  }
  // A period = 9 stages = 9 TPS taps = TPS chunks; inside it every tap offset is a compile-time constant.
  int pp = 0;  // parity of the period index (odd number of stages per period: the weight ring flips with it)
  for (int cc = 0; cc < n_chunks; cc += TPS, pp ^= 1) {
#pragma unroll
    for (int k = 0; k < 9; k++) {
      if (ODDK && k == 5 && cc + 1 == n_chunks) break;   // (the single chunk of the last period ends inside stage 4)
      const int wbuf = pp ^ (k & 1);
      // (opaque copy per stage: keeps the tap x row fragment addresses from being hoisted out of the loop into
      //  registers the accumulators need)
      asm volatile("" : "+v"(hpv));
      // k-steps 0 .. Q-2: fragments one step ahead of the MFMAs that consume them
#define MAUA_STEP2(Q0_)                                                                                  \
      MAUA_LOAD_FRAGS(af1, bf1, cc, k, Q0_ + 1, wbuf)                                                    \
      MAUA_MMA(af, bf)                                                                                   \
      if constexpr (Q0_ + 2 < Q) {                                                                       \
        MAUA_LOAD_FRAGS(af, bf, cc, k, Q0_ + 2, wbuf)                                                    \
        MAUA_MMA(af1, bf1)                                                                               \
      }
      MAUA_STEP2(0)
      if constexpr (Q == 8) { MAUA_STEP2(2) }
      if constexpr ((MAUA_DMA_SCHED & 2) != 0) {
        // (schedule 2) the first half's weight loads of the PREVIOUS stage's refill: that stage's slot (the other one) has been
        // free since its barrier; what lands here is first read behind this stage's barrier, after this wave's vmcnt(0)
        if (wave < NW / 2) {
          if (k >= 1) MAUA_ISSUE_WSTAGE(cc, k + 1, wbuf ^ 1)
          else if (cc > 0) MAUA_ISSUE_WSTAGE(cc - TPS, 10, wbuf ^ 1)
        }
      }
      if constexpr (Q == 8) { MAUA_STEP2(4) }
      MAUA_STEP2(Q - 2)
#undef MAUA_STEP2
      // here: MFMAs of k-steps 0 .. Q-2 issued, fragments of k-step Q-1 in af1 / bf1.
      // Every load this wave issued behind the previous barrier (the next stage's weights, pieces of a coming halo)
      // has landed; the barrier publishes them and tells everybody that this stage's buffers have been read for the
      // last time.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (k + 1 < 9) {
        MAUA_LOAD_FRAGS(af, bf, cc, (k + 1) % 9, 0, wbuf ^ 1)
      } else if (cc + TPS < n_chunks) {
        MAUA_LOAD_FRAGS(af, bf, cc + TPS, 0, 0, wbuf ^ 1)
      }
      if ((MAUA_DMA_SCHED & 2) == 0 || wave >= NW / 2)
        MAUA_ISSUE_WSTAGE(cc, k + 2, wbuf)  // stage s+2 into the slot stage s has just finished with
      if constexpr (TPS == 1) {
        if (k < HJ) MAUA_ISSUE_H(k, cc + 1)
      } else {
        // chunk cc+1 -> H[1]: free since the previous period's last stage (whose post-barrier block issues the first
        // third), first read by the second tap of stage 4; chunk cc+2 -> H[0]: free once stage 4 has read tap 8, first
        // read by the next period's stage 0.  HJ / 3 load instructions per wave at each of the six points.
#pragma unroll
        for (int j = 0; j < HJ / 3; j++) {
          if (k == 0) MAUA_ISSUE_H(HJ / 3 + j, cc + 1)
          if (k == 1) MAUA_ISSUE_H(2 * (HJ / 3) + j, cc + 1)
          if (k == 4) MAUA_ISSUE_H(j, cc + 2)
          if (k == 5) MAUA_ISSUE_H(HJ / 3 + j, cc + 2)
          if (k == 6) MAUA_ISSUE_H(2 * (HJ / 3) + j, cc + 2)
          if (k == 8) MAUA_ISSUE_H(j, cc + 3)
        }
      }
      MAUA_MMA(af1, bf1)
    }
  }
#undef MAUA_ISSUE_WTAP
#undef MAUA_ZERO_WTAP
#undef MAUA_ISSUE_WSTAGE
#undef MAUA_ISSUE_H
#undef MAUA_LOAD_FRAGS
#undef MAUA_MMA
#undef MAUA_SWZ

  if constexpr ((MAUA_DMA_SCHED & 1) != 0) __builtin_amdgcn_s_setprio(0);
  // ---- epilogue: demod, noise, bias, activation, gain, clamp -> LDS tile [pixel][channel] -> coalesced NHWC rows
  const float* nb = a.noise ? a.noise + (long)b * a.noise_bstride : nullptr;
  __syncthreads();  // main-loop LDS is dead from here on
  char* epi = smem;
  const float alpha = a.act == MAUA_ACT_LINEAR ? 1.f : a.alpha;
  const bool fast = (a.act == MAUA_ACT_LRELU || a.act == MAUA_ACT_LINEAR) && alpha >= 0.f && alpha <= 1.f && a.gain > 0.f && !a.prelu;
  const float cl = a.clamp >= 0.f ? a.clamp : 3.0e38f;
  // Global operands first, grouped: the WM noise values, then per N tile the demodulation / bias (/ PReLU slope) vectors of its 16
  // channels - each group is ONE round trip.  (Loaded where they were used, inside the activation's branches, every load was waited
  // for on its own: ~ 3 serial round trips for each of the WM x WN x 4 register groups of the tile.)
  float nzr[WM];
#pragma unroll
  for (int i = 0; i < WM; i++) nzr[i] = 0.f;
  if (nb) {
#pragma unroll
    for (int i = 0; i < WM; i++) nzr[i] = nb[(long)(ty0 + wm * WM + i) * a.W + tx0 + r];
#pragma unroll
    for (int i = 0; i < WM; i++) nzr[i] *= a.noise_strength * (a.noise_scale ? a.noise_scale[b] : 1.f);
  }
  constexpr int QG = KB == 64 ? 2 : 4;   // register groups per round trip (the 128-register variants take them in halves)
#pragma unroll
  for (int j = 0; j < WN; j++) {
#pragma unroll
    for (int q0 = 0; q0 < 4; q0 += QG) {
      float4 dq[QG], bq[QG];
#pragma unroll
      for (int q = 0; q < QG; q++) {
        dq[q] = make_float4(1.f, 1.f, 1.f, 1.f);
        bq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const int nl0 = (wn * WN + j) * 32 + 4 * h + 8 * q0;   // + 8 q: first of 4 consecutive channels (tile-local)
      if (a.d) {
#pragma unroll
        for (int q = 0; q < QG; q++) dq[q] = *reinterpret_cast<const float4*>(a.d + (long)b * a.Co + n0 + nl0 + 8 * q);
      }
      if (a.bias) {
#pragma unroll
        for (int q = 0; q < QG; q++) bq[q] = *reinterpret_cast<const float4*>(a.bias + n0 + nl0 + 8 * q);
      }
#pragma unroll
      for (int i = 0; i < WM; i++) {
        const int m = (wm * WM + i) * 32 + r;
        const float nz = nzr[i];
#pragma unroll
        for (int q = 0; q < QG; q++) {
          const int nl = nl0 + 8 * q, qd = q0 + q;
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
            const float4 pv = *reinterpret_cast<const float4*>(a.prelu + n0 + nl);
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
          *reinterpret_cast<uint2*>(epi + m * ES + nl * 2) = make_uint2(Fmt16<F>::pack2(v[0], v[1]), Fmt16<F>::pack2(v[2], v[3]));
        }
      }
    }
  }
  __syncthreads();
  // ---- fused toRGB + upsampled skip (conv1 layers whose channels all sit in this N tile): [32 px x Co] x [Co x 3(+3)]
  // on the matrix cores straight from the epilogue tile, as in modconv.hip (weights split hi + lo to ~2^-17)
  if (NW == 8 && a.rgb_out) {   // (wave w owns image row w of the tile)
    const int c_rgb = r < 3 ? r : (r >= 8 && r < 11 ? r - 8 : -1);
    const int mrow = wave * 32 + r;  // wave w owns image row w of the tile
    const int y = ty0 + wave, x = tx0 + r;
    const bool px_ok = h == 0;
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
          const bool uh = u == 1 || u == 2, vh = v == 1 || v == 2;
          pf[dy * 2 + dx] = !ok ? 0.f : uh ? (vh ? a.fir[5] : a.fir[4]) : (vh ? a.fir[1] : a.fir[0]);
          const unsigned o = ok ? (unsigned)(iy * Wp + ix) : 0u;
          pvv[0][dy * 2 + dx] = pv[o];
          pvv[1][dy * 2 + dx] = pv[(unsigned)(Hp * Wp) + o];
          pvv[2][dy * 2 + dx] = pv[2u * (unsigned)(Hp * Wp) + o];
        }
      }
    }
    const float* wbase = a.rgb_wmod + ((long)b * 3 + (c_rgb >= 0 ? c_rgb : 0)) * a.Co + 8 * h;
    const float rb0 = a.rgb_bias[0], rb1 = a.rgb_bias[1], rb2 = a.rgb_bias[2];
    const float row_mask = c_rgb >= 0 ? 1.f : 0.f, lo_mask = r >= 8 ? 1.f : 0.f;
    f32x16 racc;
#pragma unroll
    for (int e = 0; e < 16; e++) racc[e] = 0.f;
#pragma unroll 8
    for (int ks = 0; ks < BN / 16; ks++) {
      const float4 w0 = *reinterpret_cast<const float4*>(wbase + ks * 16);
      const float4 w1 = *reinterpret_cast<const float4*>(wbase + ks * 16 + 4);
      float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
      u32x4 wf;
#pragma unroll
      for (int k = 0; k < 8; k++) wv[k] = (wv[k] - lo_mask * Fmt16<F>::round(wv[k])) * row_mask;  // hi rows: w, lo rows: w - bf16(w)
#pragma unroll
      for (int k = 0; k < 4; k++) wf[k] = Fmt16<F>::pack2(wv[2 * k], wv[2 * k + 1]);
      const u32x4 av = *reinterpret_cast<const u32x4*>(epi + mrow * ES + (ks * 16 + 8 * h) * 2);
      Mma16<F>::step(racc, wf, av);
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
  // ---- image output (the last convolution of an up-scaler: 3 real channels of a 32-channel tile): the tile's pixels go out as
  // planar f32 [B][3][H][W] and / or u8 HWC, clamped to [0, 1], instead of as an NHWC feature tensor that a second pass would
  // only read back.  Values are the tile's (rounded to the network dtype), i.e. what that pass would have seen.
  if (BN == 32 && (a.img_f32 || a.img_u8)) {
    static_assert(BN != 32 || NT == BM, "one pixel per thread");
    const int m = tid, yy = ty0 + (m >> 5), xx = tx0 + (m & 31);
    if (yy < a.H && xx < a.W) {
      const long HWl = (long)a.H * a.W, pix = (long)yy * a.W + xx;
      const uint2 t2 = *reinterpret_cast<const uint2*>(epi + m * ES);
      const float v3[3] = {Fmt16<F>::lo(t2.x), Fmt16<F>::hi(t2.x), Fmt16<F>::lo(t2.y)};
#pragma unroll
      for (int cch = 0; cch < 3; cch++) {
        const float cv = fminf(fmaxf(v3[cch], 0.f), 1.f);
        if (a.img_f32) a.img_f32[((long)b * 3 + cch) * HWl + pix] = a.img_clamp ? cv : v3[cch];
        if (a.img_u8) {
          const int oh = a.img_h ? a.img_h : a.H, ow = a.img_w ? a.img_w : a.W;
          if (yy < oh && xx < ow) a.img_u8[(((long)b * oh + yy) * ow + xx) * 3 + cch] = (uint8_t)__float2int_rn(cv * 255.0f);
        }
      }
    }
    return;
  }
  const int yps = a.y_pstride ? a.y_pstride : a.Co;   // channel-sliced outputs: Co channels at offset y_coff of a wider pixel
  char* yb = reinterpret_cast<char*>(a.y) + ((long)b * (a.y_bstride ? a.y_bstride : (long)a.H * a.W * yps) + a.y_coff) * 2;
  const char* rb = a.res ? reinterpret_cast<const char*>(a.res) + (long)b * a.res_bstride * 2 : nullptr;
  // The stored features may carry the NEXT layer's styles (that layer's kernel then needs no modulation on its load
  // path); the fused toRGB above read the unscaled tile.  A thread always copies the same piece column (NT % PPP == 0).
  static_assert(NT % PPP == 0, "piece column per thread");
  float osc[8];
  if (a.out_scale) {
    const float* sp = a.out_scale + (long)b * a.Co + n0 + (tid % PPP) * 8;
    const float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
    osc[0] = s0.x; osc[1] = s0.y; osc[2] = s0.z; osc[3] = s0.w; osc[4] = s1.x; osc[5] = s1.y; osc[6] = s1.z; osc[7] = s1.w;
  }
  float ps_s[8], ps_q[8];   // (PSUM) this thread's piece column: sums / sums of squares of what it stores
#pragma unroll
  for (int k = 0; k < 8; k++) { ps_s[k] = 0.f; ps_q[k] = 0.f; }
  // the residual pieces of all of this thread's copy-out steps are requested up front (one round trip instead of one per step)
  static_assert((BM * PPP) % NT == 0, "whole copy-out steps");
  constexpr int NIT = BM * PPP / NT;
  u32x4 rvs[NIT], rvs2[NIT];
  const char* rb2 = (rb && a.res2) ? reinterpret_cast<const char*>(a.res2) + (long)b * a.res2_bstride * 2 : nullptr;
  if (rb) {
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int p = tid + it * NT, m = p / PPP, pc = p - m * PPP;
      const int yy = min(ty0 + (m >> 5), a.H - 1), xx = min(tx0 + (m & 31), a.W - 1);   // (overhanging pixels read a valid one)
      const long pix = (long)yy * a.W + xx;
      rvs[it] = *reinterpret_cast<const u32x4*>(rb + (pix * a.res_pstride + n0 + pc * 8) * 2);
      if (rb2) rvs2[it] = *reinterpret_cast<const u32x4*>(rb2 + (pix * a.res2_pstride + n0 + pc * 8) * 2);
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    const int p = tid + it * NT;
    const int m = p / PPP, pc = p - m * PPP;
    const long pix = (long)(ty0 + (m >> 5)) * a.W + tx0 + (m & 31);
    u32x4 v = *reinterpret_cast<const u32x4*>(epi + m * ES + pc * 16);
    if (a.out_scale) {
      u32x4 vs;
#pragma unroll
      for (int k = 0; k < 4; k++)
        vs[k] = Fmt16<F>::pack2(Fmt16<F>::lo(v[k]) * osc[2 * k], Fmt16<F>::hi(v[k]) * osc[2 * k + 1]);
      if (a.y_scaled) {   // both forms leave: the plain features to y (below), the scaled ones here (what premod_nhwc_kernel would write)
        if (ty0 + (m >> 5) < a.H && tx0 + (m & 31) < a.W)
          *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.y_scaled) + (((long)b * a.H * a.W + pix) * a.Co + n0 + pc * 8) * 2) = vs;
      } else {
        v = vs;
      }
    }
    if (rb) {   // residual added to the activated output (RRDB: out = conv5(..) * 0.2 + x); both operands bf16
      const u32x4 rv = rvs[it];
#pragma unroll
      for (int k = 0; k < 4; k++)
        v[k] = Fmt16<F>::pack2(Fmt16<F>::lo(v[k]) + Fmt16<F>::lo(rv[k]), Fmt16<F>::hi(v[k]) + Fmt16<F>::hi(rv[k]));
      if (rb2) {   // second residual on the rounded sum (RRDB: (conv5 * 0.2 + x) * 0.2 + block input): what a separate pass would compute
        const u32x4 r2 = rvs2[it];
#pragma unroll
        for (int k = 0; k < 4; k++)
          v[k] = Fmt16<F>::pack2(a.res_gain * Fmt16<F>::lo(v[k]) + Fmt16<F>::lo(r2[k]),
                         a.res_gain * Fmt16<F>::hi(v[k]) + Fmt16<F>::hi(r2[k]));
      }
    }
    if (ty0 + (m >> 5) < a.H && tx0 + (m & 31) < a.W) *reinterpret_cast<u32x4*>(yb + (pix * yps + n0 + pc * 8) * 2) = v;
    if constexpr (PSUM) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const float f0 = Fmt16<F>::lo(v[k]), f1 = Fmt16<F>::hi(v[k]);
        ps_s[2 * k] += f0; ps_s[2 * k + 1] += f1;
        ps_q[2 * k] = fmaf(f0, f0, ps_q[2 * k]); ps_q[2 * k + 1] = fmaf(f1, f1, ps_q[2 * k + 1]);
      }
    }
  }
  if constexpr (PSUM) {
    // lanes with the same lane % PPP hold the same piece column: fixed-order butterfly over them, then the waves' rows are
    // added in wave order through LDS (the epilogue tile is dead by now) - ONE row of Co / 8 pieces per workgroup
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
      for (int o = PPP; o < 64; o <<= 1) {
        ps_s[k] += __shfl_xor(ps_s[k], o);
        ps_q[k] += __shfl_xor(ps_q[k], o);
      }
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);   // [NW][PPP][16]
    if (lane < PPP) {
      float* dstl = red + (wave * PPP + lane) * 16;
#pragma unroll
      for (int k = 0; k < 8; k++) { dstl[k] = ps_s[k]; dstl[8 + k] = ps_q[k]; }
    }
    __syncthreads();
    if (tid < PPP * 16) {
      const int pc = tid >> 4, k = tid & 15;
      float acc = 0.f;
#pragma unroll
      for (int w = 0; w < NW; w++) acc += red[(w * PPP + pc) * 16 + k];
      a.psum[(((long)b * gridDim.x + blockIdx.x) * (a.Co >> 3) + (n0 >> 3) + pc) * 16 + k] = acc;
    }
  }
}

// rows per sample of ConvArgs.psum for a launch of launch_modconv_dma with these arguments
int dma_psum_rows(const ConvArgs& a) { return (a.H / TH) * (a.W / TW); }   // one row per 8 x 32-pixel tile

bool dma_conv_supported(int dtype, int Ci, int Co, int up, int H, int W) {
  return (dtype == MAUA_BF16 || dtype == MAUA_F16) && up == 1 && Ci % 64 == 0 && Co % 128 == 0 && H % TH == 0 && W % TW == 0;
}
// ... and the narrow plain convolutions of the RRDB up-scaler (super.hip): 32 or 64 output channels, K a multiple of 64
// (any H, W >= one tile: tiles that overhang the image read zeros - the convolution's own padding - and mask their stores;
//  RealESRGANer's pre_pad makes a 1024^2 frame 1034 x 1034)
bool dma_conv_narrow_supported(int dtype, int Ci, int Co, int H, int W) {
  // (K in 32-channel chunks, two per period: 32 output channels also take an odd chunk count >= 3, ending the last period early)
  return dtype == MAUA_BF16 && (Ci % 64 == 0 || (Co == 32 && Ci % 32 == 0 && Ci >= 96)) && (Co == 32 || Co == 64) && H >= TH &&
         W >= TW;
}

template <int WAVES_M, int WAVES_N, int WM, int WN, int TPS, int KB, bool PSUM = false, bool ODDK = false, typename F = bf16_t>
static int launch_dma_variant(hipStream_t stream, const ConvArgs& a) {
  constexpr int BN = WAVES_N * WN * 32, NT = WAVES_M * WAVES_N * 64;
  const size_t smem = std::max<size_t>((size_t)2 * TPS * BN * KB + 2 * HALO_PX * KB, (size_t)TH * TW * (BN * 2 + 16));
  MAUA_REQUIRE(smem <= 160 * 1024, "modconv_dma: LDS budget exceeded");
  MAUA_REQUIRE(((a.Ci / (KB / 2)) % TPS == 0) != ODDK, "modconv_dma: chunk count must be a multiple of the taps per stage");
  MAUA_REQUIRE(!a.rgb_out || (a.Co == BN && a.rgb_wmod && a.rgb_bias), "modconv_dma: fused toRGB needs all channels in one N tile");
  auto kern = modconv_dma_kernel<WAVES_M, WAVES_N, WM, WN, TPS, KB, PSUM, ODDK, F>;
  MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  dim3 grid(((a.H + TH - 1) / TH) * ((a.W + TW - 1) / TW), a.B, a.Co / BN);
  hipLaunchKernelGGL(kern, grid, dim3(NT), smem, stream, a);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// a.x must already carry the styles (x * s[b, ci]); a.s is not read.  dtype: MAUA_BF16, or MAUA_F16 on the two wide tiles (round 6: the
// reference's own render dtype, render/ffmpeg.py:45 - same kernel, v_mfma_f32_32x32x16_f16, half conversions)
int launch_modconv_dma(hipStream_t stream, const ConvArgs& a, int dtype) {
  if (dtype == MAUA_F16) {
    MAUA_REQUIRE(dma_conv_supported(MAUA_F16, a.Ci, a.Co, a.up, a.H, a.W) && !a.psum && !a.x_up2 && a.B <= 65535 &&
                     (long)a.H * a.W * (a.x_pstride ? a.x_pstride : a.Ci) * 2 < (1L << 32) && (!a.y_scaled || (a.out_scale && !a.res)),
                 "modconv_dma (f16): unsupported shape / arguments");
    if (a.B == 0) return MAUA_OK;
    if (a.Co % 256 == 0 && a.variant != 128) return launch_dma_variant<2, 4, 4, 2, 1, 128, false, false, f16_t>(stream, a);
    return launch_dma_variant<4, 2, 2, 2, 2, 64, false, false, f16_t>(stream, a);
  }
  const bool narrow = dma_conv_narrow_supported(MAUA_BF16, a.Ci, a.Co, a.H, a.W) && a.up == 1;
  MAUA_REQUIRE(narrow || dma_conv_supported(MAUA_BF16, a.Ci, a.Co, a.up, a.H, a.W), "modconv_dma: unsupported shape");
  MAUA_REQUIRE(a.B <= 65535, "modconv_dma: grid too large");
  if (a.B == 0) return MAUA_OK;
  MAUA_REQUIRE(!a.x_up2 || (a.H % 2 == 0 && a.W % 2 == 0 && a.up == 1), "modconv_dma: x_up2 needs even output sizes");
  MAUA_REQUIRE(!a.y_scaled || (a.out_scale && !a.res), "modconv_dma: y_scaled goes with out_scale (and no residual)");
  MAUA_REQUIRE((long)a.H * a.W * (a.x_pstride ? a.x_pstride : a.Ci) * 2 < (1L << 32),
               "modconv_dma: a sample must stay below 4 GiB (32-bit offsets)");
  // narrow N tiles (4 waves, 64-byte K rows, two taps per stage): 64 channels = 2 x 2 blocks per wave, 32 = 2 x 1
  if (narrow) {
    MAUA_REQUIRE(!a.rgb_out && !a.out_scale && !a.psum, "modconv_dma: the narrow tiles carry no toRGB / style scaling / piece sums");
    MAUA_REQUIRE(!(a.img_f32 || a.img_u8) || (a.Co == 32 && !a.res), "modconv_dma: image output is the 32-channel tile's, without a residual");
    MAUA_REQUIRE((a.H % TH == 0 && a.W % TW == 0) || !a.noise, "modconv_dma: overhanging tiles take no noise operand");
    if (a.Co == 64) return launch_dma_variant<4, 1, 2, 2, 2, 64>(stream, a);
    return (a.Ci / 32) % 2 ? launch_dma_variant<4, 1, 2, 1, 2, 64, false, true>(stream, a) : launch_dma_variant<4, 1, 2, 1, 2, 64>(stream, a);
  }
  // (channel-sliced operands and the residual are honoured by every tile shape: the kernel's address arithmetic is shared)
  // 256-channel N tile: 128-byte K rows, one tap per stage, 149 KB of LDS, one workgroup per CU.
  // 128-channel N tile: 64-byte K rows, two taps per stage, 75 KB -> two workgroups per CU (measured on the 256^2 layer,
  // K = 1152: 0.90 -> 0.65 ms against the same tile with 128-byte rows and one workgroup per CU; for the 256-channel
  // layers the two-workgroup shape measured the same as the big tile, which also keeps their toRGB fused)
  // (a.variant == 128: the caller asks for the 128-channel tile although 256 would divide - twice the workgroups for
  //  launches that would otherwise leave CUs idle, e.g. the diffusion UNet's 64^2 level at small batch)
  if (a.psum) {
    if (a.Co % 256 == 0 && a.variant != 128) return launch_dma_variant<2, 4, 4, 2, 1, 128, true>(stream, a);
    return launch_dma_variant<4, 2, 2, 2, 2, 64, true>(stream, a);
  }
  if (a.Co % 256 == 0 && a.variant != 128) return launch_dma_variant<2, 4, 4, 2, 1, 128>(stream, a);
  return launch_dma_variant<4, 2, 2, 2, 2, 64>(stream, a);
}

bool dma_rgb_fusable(int Co) { return Co == 128 || Co == 256; }

}  // namespace maua
