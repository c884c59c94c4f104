// 3x3 modulated convolution (up = 1) of the 512^2 block as a ROW WALK: b512.conv1, 64 -> 64 channels (+ the block's toRGB
// and FIR-upsampled skip).  Same math as modconv_hires.hip <64,64,1> (reference ops.py:146-186 modulated_conv2d,
// :189-233 conv2d_resample up = 1, :65-84 bias_act; inference/stylegan2.py:268-272 ToRGBLayer, :372-378 skip), other data
// movement: the tiled kernel re-reads a 6 x 34 halo per 4 x 32 tile and fetches one LDS fragment per MFMA; here a wave walks
// DOWN a 32-pixel column with its 36 weight fragments in registers, every input row is staged once (LDS-direct DMA, no
// vertical halo) and each of its 12 fragments feeds THREE output rows (taps ky = 0, 1, 2 -> the rows below / at / above):
// 12 + 12 ds_read_b128 per 36 MFMAs (the ky = 0 weights sit in LDS), one barrier per row.
// MEASURED EQUAL to the tiled kernel (0.80 vs 0.79 ms at B = 32; the layer's floor is its 2.2 GB at ~4.7 TB/s = 0.47 ms):
// with two 256-register waves per SIMD every phase of a step is instruction-issue bound and the phases add up (ablation
// by MAUA_CW_SKIP, bits 1 = no row DMA, 2 = no MFMA loop, 4 = no epilogue, 8 = no stores: skeleton 0.15 ms, + rows in
// 0.15, + stores 0.26, + MFMA loop 0.09 .. 0.22, + epilogue 0.02 .. 0.15).  Kept as synth option "cwalk".
//   workgroup = 8 waves = 4 pixel blocks of 32 x 2 halves of the output channels -> a strip 128 pixels wide; persistent,
//   items (sample, row segment, strip) in contiguous runs; weights (x styles x demodulation x gain) reloaded when the sample
//   changes.  Step k of a segment [r0, r1): input row r0 - 1 + k is multiplied; output row r0 - 2 + k gets its epilogue
//   (noise, lrelu, clamp -> bf16 tile in LDS; its toRGB partial sums over the wave's 32 channels straight from the
//   registers, K permuted to the accumulator layout as in modconv_upwalk.hip); output row r0 - 3 + k leaves: the tile as
//   whole 128-byte pixels, the RGB row after adding the other channel half's partial sums, bias, clamp and the skip.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "internal.h"

namespace maua {

namespace {

constexpr int CW_CI = 64, CW_CO = 64, CW_KS = CW_CI / 16, CW_KSR = 2;
constexpr int CW_SW = 128;                    // strip width (output pixels)
constexpr int CW_XPX = CW_SW + 2;             // staged input pixels per row
constexpr int CW_XG = (CW_XPX + 7) / 8;       // 1 KB DMA pieces (8 pixels of 128 bytes)
constexpr int CW_XROWB = CW_XG * 1024;
constexpr int CW_TILEB = CW_SW * CW_CO * 2;   // one output row of the strip, [pixel][channel] bf16, pieces XOR-swizzled
constexpr int CW_PVP = CW_SW / 2 + 2, CW_PVW = 256;   // staged row of the previous image: [3][66] floats, padded

__device__ __forceinline__ void cw_dma_b32(const void* sbase, unsigned voff_bytes, void* lds_wave_base) {
  const unsigned base =
      __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff_bytes), "s"(sbase), "s"(base)
               : "memory");
}

}  // namespace

__global__ __launch_bounds__(512, 1) void conv_walk_kernel(HiresArgs a, int seg_rows, int nseg, int strips, int n_items,
                                                           int items_per_wg, int skip) {
  constexpr int CI = CW_CI, CO = CW_CO, KS = CW_KS, KSR = CW_KSR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xrow = smem;                                              // [3][XROWB]: rows are staged two steps ahead
  char* tile = smem + 3 * CW_XROWB;                               // [2][TILEB]
  float* rgbp = reinterpret_cast<float*>(tile + 2 * CW_TILEB);    // [2][2 halves][3][SW]: toRGB partial sums per channel half
  float* pvs = rgbp + 2 * 2 * 3 * CW_SW;                             // [3 slots][PVW]
  float* bias_s = pvs + 3 * CW_PVW;                               // [2 halves][2 h][16]: bias * gain in accumulator order
  u32x4* wl = reinterpret_cast<u32x4*>(bias_s + CO);              // [2 halves][3 kx][KS][64 lanes]: the ky = 0 A fragments
  u32x4* rfl = wl + 2 * 3 * KS * 64;                              // [2 halves][KSR][64 lanes]: toRGB A fragments
  float* nzs = reinterpret_cast<float*>(rfl + 2 * KSR * 64);      // [3][SW]: staged noise rows

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int pbk = wave & 3, ch = wave >> 2;     // (waves w and w + 4 share a SIMD: same pixels, the two channel halves)
  const int Hp = a.H >> 1, Wp = a.W >> 1;
  const unsigned HWl = (unsigned)a.H * (unsigned)a.W;
  const int item0 = blockIdx.x * items_per_wg, item1 = min(item0 + items_per_wg, n_items);
  const bool rgb_on = a.rgb_out != nullptr;
  const float cl = a.clamp >= 0.f ? a.clamp : 3.0e38f;
  const float nz_scale = a.noise_strength * a.gain;
  const f32x2_t al = a.alpha;

  // A fragments: W[tap][co][ci] * s[ci] * d[co] * gain of this wave's 32 output channels; the taps ky = 1, 2 in registers,
  // ky = 0 in LDS (36 fragments + 3 accumulators + the prefetch slots do not fit 256 registers)
  u32x4 w1[6 * KS];

  // items are ordered (sample, segment, strip): an outer loop over the samples of this workgroup's run loads the weights
  // unconditionally (a conditional reload inside one item loop makes the compiler keep two copies of the fragments)
  int item = item0;
  while (item < item1) {
    const int b = item / (nseg * strips);
    const int item_end = min(item1, (b + 1) * nseg * strips);
    __syncthreads();                  // the previous sample's LDS reads are done
    {
      if (tid < CO) {
        const int cc = tid >> 5, hh = (tid >> 4) & 1, e = tid & 15, chn = 32 * cc + 8 * (e >> 2) + 4 * hh + (e & 3);
        bias_s[tid] = (a.bias ? a.bias[chn] : 0.f) * a.gain;
      }
      // (opaque lane indices: the 36 weight addresses are loop-invariant and would otherwise be hoisted out of the item loop
      //  into 72 registers that stay live across the walk)
      int r_ = r, h_ = h;
      asm volatile("" : "+v"(r_), "+v"(h_));
      const float dco = (a.d ? a.d[(long)b * CO + 32 * ch + r_] : 1.f) * a.gain;
      const float* sb = a.s + (long)b * CI;
      float sv[KS][8];
#pragma unroll
      for (int cs = 0; cs < KS; cs++)
#pragma unroll
        for (int e = 0; e < 8; e++) sv[cs][e] = sb[cs * 16 + 8 * h_ + e] * dco;
      const bf16_t* wbase = reinterpret_cast<const bf16_t*>(a.w);
#pragma unroll
      for (int tap = 0; tap < 9; tap++)
#pragma unroll
        for (int cs = 0; cs < KS; cs++) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(wbase + ((long)tap * CO + 32 * ch + r_) * CI + cs * 16 + 8 * h_);
          u32x4 o;
#pragma unroll
          for (int k = 0; k < 4; k++)
            o[k] = pack2bf(bf2f((bf16_t)(v[k] & 0xffff)) * sv[cs][2 * k], bf2f((bf16_t)(v[k] >> 16)) * sv[cs][2 * k + 1]);
          if (tap >= 3) w1[(tap - 3) * KS + cs] = o;
          else if (pbk == 0) wl[((ch * 3 + tap) * KS + cs) * 64 + lane] = o;
        }
      if (rgb_on) {
        const int c_rgb = (r_ < 16 && (r_ & 3) < 3) ? (r_ & 3) : -1;
#pragma unroll
        for (int ks = 0; ks < KSR; ks++) {
          u32x4 o = u32x4{0u, 0u, 0u, 0u};
          if (c_rgb >= 0) {
            const float* src = a.rgb_wmod + ((long)b * 3 + c_rgb) * CO + 32 * ch + ks * 16 + 4 * h_;
#pragma unroll
            for (int k = 0; k < 4; k++) {
              float w0 = src[8 * (k >> 1) + 2 * (k & 1)], w1v = src[8 * (k >> 1) + 2 * (k & 1) + 1];
              const float h0 = bf2f(f2bf(w0)), h1 = bf2f(f2bf(w1v));
              if (r_ >= 8) { w0 -= h0; w1v -= h1; }
              o[k] = pack2bf(w0, w1v);
            }
          }
          // (hi rows 0..2, bf16 remainder rows 8..10, K in accumulator order; kept in LDS: two reads per row)
          if (pbk == 0) rfl[(ch * KSR + ks) * 64 + lane] = o;
        }
      }
    }
   for (; item < item_end; item++) {
    const int rem = item - b * (nseg * strips);
    const int seg = rem / strips, strip = rem - seg * strips;
    const int X0 = strip * CW_SW;
    const int r0 = seg * seg_rows, r1 = min(r0 + seg_rows, a.H);
    const int nsteps = r1 - r0 + 3;   // two rows to fill the 3-row window + one step for the last row to leave
    __syncthreads();                  // the previous item's LDS reads are done (and the weights in LDS are written)
    for (int i = tid; i < 3 * CW_XROWB / 16; i += 512) reinterpret_cast<u32x4*>(xrow)[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();

    // ---- per-item lane constants
    const int pxl = pbk * 32 + r;                 // pixel inside the strip; staged-row position of tap dx: pxl + dx
    const int px = X0 + pxl;
    const bool px_ok = px < a.W;
    const int pxc = min(px, a.W - 1);
    // byte offsets of the B fragments inside a staged row: position q = pxl + dx, piece (2 cs + h) ^ ((q >> 1) & 7), i.e.
    // fbase[dx] ^ (cs << 5) (the k-step only flips bits 5, 6 of the offset: three registers instead of twelve)
    int fbase[3];
#pragma unroll
    for (int dx = 0; dx < 3; dx++) {
      const int q = pxl + dx;
      fbase[dx] = q * 128 + ((h ^ ((q >> 1) & 7)) << 4);
    }
    // DMA descriptors of the input row: piece j = wave + 8 jj covers positions 8 j .. 8 j + 7; lane = (position, slot) and
    // fetches the 16-byte source piece that belongs in its slot after the XOR swizzle
    unsigned xoff[3];
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
      const int grp = wave + 8 * jj, q = grp * 8 + (lane >> 3), gx = X0 - 1 + q;
      const bool ok = grp < CW_XG && q < CW_XPX && gx >= 0 && gx < a.W;
      xoff[jj] = ok ? (unsigned)((gx * CI + (((lane & 7) ^ ((q >> 1) & 7)) * 8)) * 2) : 0xffffffffu;
    }
    const char* xbase = reinterpret_cast<const char*>(a.x) + (long)b * a.H * a.W * CI * 2;
    const char* nbase = a.noise ? reinterpret_cast<const char*>(a.noise + (long)b * a.noise_bstride) : nullptr;
    const char* pvbase = (rgb_on && a.rgb_prev) ? reinterpret_cast<const char*>(a.rgb_prev + (long)b * 3 * Hp * Wp) : nullptr;
    const int Jp0 = (X0 - 1) >> 1;                // first column of the previous image the strip's skip taps read
    unsigned pvoff = 0xffffffffu;
    {
      const int f = wave * 64 + lane, c3 = f / CW_PVP, pxi = f - c3 * CW_PVP, gx = Jp0 + pxi;
      if (f < 3 * CW_PVP && pvbase) pvoff = (unsigned)((c3 * Hp) * Wp + min(max(gx, 0), Wp - 1)) * 4u;
    }
    // skip = upsample2d of the previous image in its branch-free 2 x 2 form (modconv_hires.hip): output pixel (y, x) reads
    // rows (y - 1) >> 1, + 1 and columns (x - 1) >> 1, + 1; the column parity is the lane's, the row parity alternates:
    // two coefficient sets (odd / even rows); columns outside the image get a zero coefficient, rows are staged as zeros
    const int ix0 = (pxc - 1) >> 1;
    const int pvi = ix0 - Jp0;
    float fc[4];   // odd rows; on even rows the two staged rows swap their roles (the 2 x 2 taps are symmetric in dy)
    {
      const bool xo = pxc & 1;
#pragma unroll
      for (int dy = 0; dy < 2; dy++)
#pragma unroll
        for (int dx = 0; dx < 2; dx++) {
          const bool vh = (dx == 0) == xo;
          const bool okk = ix0 + dx >= 0 && ix0 + dx < Wp;
          fc[dy * 2 + dx] = !okk ? 0.f : (dy == 0) ? (vh ? a.fir[5] : a.fir[4]) : (vh ? a.fir[1] : a.fir[0]);
        }
    }
    const float rgb_b0 = rgb_on ? a.rgb_bias[0] : 0.f, rgb_b1 = rgb_on ? a.rgb_bias[1] : 0.f, rgb_b2 = rgb_on ? a.rgb_bias[2] : 0.f;
    const f32x16* bias_p = reinterpret_cast<const f32x16*>(bias_s + (ch * 2 + h) * 16);
    const u32x4* wlp = wl + ch * 3 * KS * 64 + lane;
    const u32x4* rfp = rfl + ch * KSR * 64 + lane;

    // (pieces of this wave that have any pixel inside the image: uniform, so that the number of DMA instructions a step
    //  issues is known - the step waits for everything OLDER than them, s_waitcnt vmcnt(that number))
    int nx = 0;
    bool xact[3];
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
      xact[jj] = __ballot(xoff[jj] != 0xffffffffu) != 0ull;
      nx += xact[jj] ? 1 : 0;
    }
#define MAUA_CW_STAGE_X(RHO, BUF, ND_)                                                                   \
  {                                                                                                      \
    const int gy_ = (RHO);                                                                               \
    char* dst_ = xrow + (BUF) * CW_XROWB;                                                                \
    if (gy_ >= 0 && gy_ < a.H) {                                                                         \
      const char* rb_ = xbase + (long)gy_ * a.W * CI * 2;                                                \
      _Pragma("unroll") for (int jj = 0; jj < 3; jj++)                                                  \
        if (xact[jj]) {                                                                                  \
          if (xoff[jj] != 0xffffffffu) lds_dma_b128(rb_, xoff[jj], dst_ + (wave + 8 * jj) * 1024);       \
        }                                                                                                \
      ND_ += nx;                                                                                         \
    } else {                                                                                             \
      for (int i = tid; i < CW_XROWB / 16; i += 512) reinterpret_cast<u32x4*>(dst_)[i] = u32x4{0u, 0u, 0u, 0u}; \
    }                                                                                                    \
  }
#define MAUA_CW_STAGE_NZ(ROW, SLOT, ND_)                                                                 \
  if (nbase && wave >= 4 && wave < 6) {                                                                  \
    cw_dma_b32(nbase + (long)min(max((ROW), 0), a.H - 1) * a.W * 4, nzoff, nzs + (SLOT) * CW_SW + (wave - 4) * 64); \
    ND_ += 1;                                                                                            \
  }
#define MAUA_CW_STAGE_PV(M, ND_)                                                                         \
  if (rgb_on && wave < 4) {                                                                              \
    const int m_ = (M);                                                                                  \
    float* dst_ = pvs + ((m_ + 3) % 3) * CW_PVW;                                                         \
    if (m_ >= 0 && m_ < Hp && pvbase) {                                                                  \
      if (pvoff != 0xffffffffu) cw_dma_b32(pvbase + (long)m_ * Wp * 4, pvoff, dst_ + wave * 64);         \
      ND_ += 1;                                                                                          \
    } else {                                                                                             \
      dst_[wave * 64 + lane] = 0.f;                                                                      \
    }                                                                                                    \
  }
// wait until at most N_ (uniform, <= 5) vector-memory operations are outstanding
#define MAUA_CW_WAIT_VM(N_)                                                                              \
  switch (N_) {                                                                                          \
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;                                     \
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;                                     \
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;                                     \
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;                                     \
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;                                     \
    default: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;                                    \
  }
    const unsigned nzoff = (unsigned)min(X0 + (wave & 1) * 64 + lane, a.W - 1) * 4u;
    {
      int nd0 = 0;   // prologue: the rows of steps 0 and 1, the two rows of the previous image the first row's skip reads
      MAUA_CW_STAGE_X(r0 - 1, 0, nd0)
      MAUA_CW_STAGE_X(r0, 1, nd0)
      MAUA_CW_STAGE_NZ(r0 - 2, 0, nd0)
      MAUA_CW_STAGE_NZ(r0 - 1, 1, nd0)
      MAUA_CW_STAGE_PV((r0 - 1) >> 1, nd0)
      MAUA_CW_STAGE_PV(((r0 - 1) >> 1) + 1, nd0)
    }
    f32x16 accP = *bias_p, accC = accP, accN = accP;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // One step; (P, C, N) = accumulators of output rows rho - 1, rho, rho + 1 where rho = the input row multiplied now.
    // The three roles rotate from step to step, so the loop body is three steps with the names permuted (no copies).
#define MAUA_CW_FOFF(IT_) ((((IT_) / KS) == 0 ? fb0_ : ((IT_) / KS) == 1 ? fb1_ : fb2_) ^ (((IT_) % KS) << 5))
#define MAUA_CW_STEP(K_, J_, P_, C_, N_)                                                                                \
  {                                                                                                                   \
    const int k_ = (K_);                                                                                              \
    const int rho_ = r0 - 1 + k_, epi_row_ = rho_ - 1, fin_row_ = rho_ - 2;                                           \
    /* (opaque copies: everything derived from them is recomputed in the step instead of living in registers the      \
        weights need) */                                                                                              \
    int tid_ = tid, pxl_ = pxl;                                                                                       \
    asm volatile("" : "+v"(tid_), "+v"(pxl_));                                                                        \
    __syncthreads();                                                                                                  \
    /* ---- staging, two steps ahead: input row rho + 2, the noise of the epilogue two steps on, the row of the previous  \
       image that the skip of the row leaving two steps on adds to the staged ones */                                 \
    int nd_ = 0;                                                                                                      \
    if (k_ + 2 < nsteps - 1 && !(skip & 1)) MAUA_CW_STAGE_X(rho_ + 2, ((J_) + 2) % 3, nd_)                                           \
    MAUA_CW_STAGE_NZ(epi_row_ + 2, ((J_) + 2) % 3, nd_)                                                               \
    MAUA_CW_STAGE_PV((epi_row_ >> 1) + 1, nd_)                                                                        \
    if (k_ < nsteps - 1) {                                                                                            \
     if (!(skip & 2)) {                                                                                                \
      /* ---- multiply input row rho: 12 fragments x 3 taps, operands requested two iterations ahead */              \
      u32x4 pa_[3], pw_[3];                                                                                           \
      const int fb0_ = fbase[0] + (J_) * CW_XROWB, fb1_ = fbase[1] + (J_) * CW_XROWB;                                 \
      const int fb2_ = fbase[2] + (J_) * CW_XROWB;                                                                    \
      pa_[0] = *reinterpret_cast<const u32x4*>(xrow + MAUA_CW_FOFF(0));                                               \
      pw_[0] = wlp[0];                                                                                                \
      pa_[1] = *reinterpret_cast<const u32x4*>(xrow + MAUA_CW_FOFF(1));                                               \
      pw_[1] = wlp[64];                                                                                               \
      _Pragma("unroll") for (int it_ = 0; it_ < 3 * KS; it_++) {                                                     \
        if (it_ + 2 < 3 * KS) {                                                                                       \
          pa_[(it_ + 2) % 3] = *reinterpret_cast<const u32x4*>(xrow + MAUA_CW_FOFF(it_ + 2));                         \
          pw_[(it_ + 2) % 3] = wlp[(it_ + 2) * 64];                                                                   \
        }                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        const bf16x8 bv_ = __builtin_bit_cast(bf16x8, pa_[it_ % 3]);                                                  \
        P_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1[3 * KS + it_]), bv_, P_, 0, 0, 0); \
        C_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1[it_]), bv_, C_, 0, 0, 0);          \
        N_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pw_[it_ % 3]), bv_, N_, 0, 0, 0);     \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
      }                                                                                                               \
     }                                                                                                                \
     if (!(skip & 4)) {                                                                                               \
      /* ---- epilogue of output row rho - 1 (complete now): bias rides in the accumulator */                         \
      char* tw_ = tile + (k_ & 1) * CW_TILEB + pxl_ * 128 + 8 * h;                                                             \
      const f32x2_t nn_ = (nbase ? nzs[(J_) * CW_SW + pxl_] : 0.f) * nz_scale;                                                                           \
      u32x4 fa_[KSR];                                                                                                 \
      _Pragma("unroll") for (int e_ = 0; e_ < 16; e_ += 2) {                                                         \
        const f32x2_t y_ = f32x2_t{P_[e_], P_[e_ + 1]} + nn_;                                                         \
        const f32x2_t s_ = y_ * al;                                                                                   \
        fa_[e_ >> 3][(e_ >> 1) & 3] = pack2bf(__builtin_amdgcn_fmed3f(fmaxf(y_[0], s_[0]), -cl, cl),                  \
                                              __builtin_amdgcn_fmed3f(fmaxf(y_[1], s_[1]), -cl, cl));                 \
      }                                                                                                               \
      _Pragma("unroll") for (int g_ = 0; g_ < 4; g_++)                                                               \
        *reinterpret_cast<uint2*>(tw_ + (((4 * ch + g_) ^ (pxl_ & 7)) << 4)) =                                              \
            make_uint2(fa_[g_ >> 1][2 * (g_ & 1)], fa_[g_ >> 1][2 * (g_ & 1) + 1]);                                   \
      if (rgb_on) {                                                                                                   \
        f32x16 ra_;                                                                                                   \
        _Pragma("unroll") for (int e_ = 0; e_ < 16; e_++) ra_[e_] = 0.f;                                             \
        _Pragma("unroll") for (int ks_ = 0; ks_ < KSR; ks_++)                                                        \
          ra_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rfp[ks_ * 64]),                    \
                                                        __builtin_bit_cast(bf16x8, fa_[ks_]), ra_, 0, 0, 0);         \
        const float p0_ = ra_[0] + ra_[4], p1_ = ra_[1] + ra_[5], p2_ = ra_[2] + ra_[6];                              \
        if (h == 0) {                                                                                                 \
          float* rp_ = rgbp + ((k_ & 1) * 2 + ch) * 3 * CW_SW + pxl_;                                                 \
          rp_[0] = p0_; rp_[CW_SW] = p1_; rp_[2 * CW_SW] = p2_;                                                       \
        }                                                                                                             \
      }                                                                                                               \
     }                                                                                                                \
      P_ = *bias_p;   /* this accumulator is row rho + 2's from the next step on */                                   \
    }                                                                                                                 \
    MAUA_CW_WAIT_VM(nd_)   /* everything older than this step's DMAs has landed: the next step's rows, the last stores */ \
    /* ---- output row fin_row leaves (tile and partial sums of the previous step): features as whole pixels, RGB after  \
       the cross-half sum.  Issued behind the wait so that these stores have the whole next step to drain */         \
    if (fin_row_ >= r0 && fin_row_ < r1 && !(skip & 8)) {                                                                           \
      const char* tb_ = tile + ((k_ + 1) & 1) * CW_TILEB;                                                             \
      if (a.y) {                                                                                                      \
        char* yrow_ = reinterpret_cast<char*>(a.y) + (((long)b * a.H + fin_row_) * a.W + X0) * (CO * 2);             \
        _Pragma("unroll") for (int t_ = 0; t_ < 2; t_++) {                                                           \
          const int id_ = tid_ + 512 * t_, pxt_ = id_ >> 3, j_ = id_ & 7;                                              \
          if (X0 + pxt_ < a.W)                                                                                        \
            *reinterpret_cast<u32x4*>(yrow_ + pxt_ * 128 + j_ * 16) =                                                 \
                *reinterpret_cast<const u32x4*>(tb_ + pxt_ * 128 + ((j_ ^ (pxt_ & 7)) << 4));                         \
        }                                                                                                             \
      }                                                                                                               \
      if (rgb_on && ch == 0 && h == 0) {                                                                              \
        const float* rp_ = rgbp + ((k_ + 1) & 1) * 6 * CW_SW + pxl_;                                                  \
        float o0_ = rp_[0] + rp_[3 * CW_SW] + rgb_b0, o1_ = rp_[CW_SW] + rp_[4 * CW_SW] + rgb_b1;                      \
        float o2_ = rp_[2 * CW_SW] + rp_[5 * CW_SW] + rgb_b2;                                                         \
        if (a.rgb_clamp >= 0.f) {                                                                                     \
          o0_ = fminf(fmaxf(o0_, -a.rgb_clamp), a.rgb_clamp);                                                         \
          o1_ = fminf(fmaxf(o1_, -a.rgb_clamp), a.rgb_clamp);                                                         \
          o2_ = fminf(fmaxf(o2_, -a.rgb_clamp), a.rgb_clamp);                                                         \
        }                                                                                                             \
        if (pvbase) {                                                                                                 \
          const int m0_ = (fin_row_ - 1) >> 1;                                                                        \
          const bool odd_ = fin_row_ & 1;                                                                             \
          const float* sA_ = pvs + ((m0_ + (odd_ ? 3 : 4)) % 3) * CW_PVW + pvi;                                       \
          const float* sB_ = pvs + ((m0_ + (odd_ ? 4 : 3)) % 3) * CW_PVW + pvi;                                       \
          o0_ = (sA_[0] * fc[0] + sA_[1] * fc[1] + sB_[0] * fc[2] + sB_[1] * fc[3]) + o0_;                            \
          o1_ = (sA_[CW_PVP] * fc[0] + sA_[CW_PVP + 1] * fc[1] + sB_[CW_PVP] * fc[2] + sB_[CW_PVP + 1] * fc[3]) + o1_; \
          o2_ = (sA_[2 * CW_PVP] * fc[0] + sA_[2 * CW_PVP + 1] * fc[1] + sB_[2 * CW_PVP] * fc[2] + sB_[2 * CW_PVP + 1] * fc[3]) + o2_; \
        }                                                                                                             \
        if (px_ok) {                                                                                                  \
          float* ob_ = a.rgb_out + (long)b * 3 * HWl + (unsigned)(fin_row_ * a.W + X0 + pxl_);                              \
          ob_[0] = o0_; ob_[HWl] = o1_; ob_[2 * HWl] = o2_;                                                           \
        }                                                                                                             \
      }                                                                                                               \
    }                                                                                                                 \
  }

#pragma unroll 1
    for (int k = 0; k < nsteps; k += 3) {
      MAUA_CW_STEP(k, 0, accP, accC, accN)
      if (k + 1 >= nsteps) break;
      MAUA_CW_STEP(k + 1, 1, accC, accN, accP)
      if (k + 2 >= nsteps) break;
      MAUA_CW_STEP(k + 2, 2, accN, accP, accC)
    }
#undef MAUA_CW_STEP
#undef MAUA_CW_FOFF
#undef MAUA_CW_STAGE_X
#undef MAUA_CW_STAGE_NZ
#undef MAUA_CW_STAGE_PV
#undef MAUA_CW_WAIT_VM
   }
  }
}

bool cwalk_supported(int dtype, int Ci, int Co, int up, int H, int W) {
  return dtype == MAUA_BF16 && Ci == CW_CI && Co == CW_CO && up == 1 && H >= 2 && W >= 2 && (H % 2) == 0 && (W % 2) == 0;
}

int launch_conv_walk(hipStream_t stream, const HiresArgs& a0) {
  if (a0.B == 0) return MAUA_OK;
  MAUA_REQUIRE(cwalk_supported(MAUA_BF16, a0.Ci, a0.Co, a0.up, a0.H, a0.W), "conv_walk: unsupported shape");
  MAUA_REQUIRE((long)a0.H * a0.W * CW_CI * 2 < (1L << 31), "conv_walk: a sample must stay below 2 GiB (32-bit in-sample offsets)");
  MAUA_REQUIRE(a0.act == MAUA_ACT_LRELU || a0.act == MAUA_ACT_LINEAR, "conv_walk: lrelu / linear only");
  MAUA_REQUIRE(!a0.rgb8_out, "conv_walk: the u8 pack belongs to the last block");
  MAUA_REQUIRE(a0.y || a0.rgb_out, "conv_walk: no output");
  MAUA_REQUIRE(!a0.rgb_out || (a0.rgb_wmod && a0.rgb_bias), "conv_walk: toRGB needs its weights");
  HiresArgs a = a0;
  if (a.act == MAUA_ACT_LINEAR) a.alpha = 1.f;
  MAUA_REQUIRE(a.alpha >= 0.f && a.alpha <= 1.f && a.gain > 0.f, "conv_walk: needs 0 <= alpha <= 1 and gain > 0");
  const size_t smem = 3 * CW_XROWB + 2 * CW_TILEB + 2 * 2 * 3 * CW_SW * 4 + 3 * CW_PVW * 4 + CW_CO * 4 + 2 * 3 * CW_KS * 64 * 16 + 2 * CW_KSR * 64 * 16 + 3 * CW_SW * 4;
  auto kern = conv_walk_kernel;
  MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  int dev = 0, cus = 256;
  MAUA_HIP_CHECK(hipGetDevice(&dev));
  MAUA_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int strips = (a.W + CW_SW - 1) / CW_SW;
  // items = (sample, row segment, strip), about nine per CU (a segment costs three extra steps); one persistent workgroup
  // per CU takes a contiguous run of them
  int nseg = std::max(1, (9 * cus + strips * a.B - 1) / (strips * a.B));
  nseg = std::min(nseg, std::max(1, a.H / 16));
  const int seg_rows = (a.H + nseg - 1) / nseg;
  nseg = (a.H + seg_rows - 1) / seg_rows;
  const int n_items = strips * nseg * a.B;
  const int wgs = std::min(n_items, cus);
  const int ipw = (n_items + wgs - 1) / wgs;
  static const int skip = getenv("MAUA_CW_SKIP") ? atoi(getenv("MAUA_CW_SKIP")) : 0;
  hipLaunchKernelGGL(kern, dim3((n_items + ipw - 1) / ipw), dim3(512), smem, stream, a, seg_rows, nseg, strips, n_items, ipw, skip);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace maua
