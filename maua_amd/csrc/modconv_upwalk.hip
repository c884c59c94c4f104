// Up-sampling modulated convolution for the high-resolution layers (512^2 -> 1024^2, 64 -> 32 channels), half-folded
// form with a row walk.  Same math as the phase form of modconv.hip / modconv_hires.hip<64,32,2> (reference
// ops.py:211-225: conv_transpose2d stride 2, then upfirdn2d with 4 f, f = outer([1,3,3,1]) / 64; fused with
// ops.py:146-186 modulation / demodulation, stylegan2.py:236-257 noise + bias + lrelu + clamp), at HALF the matrix work.
//
// Only the HORIZONTAL FIR is folded into the weights.  With A the (flipped) 3x3 kernel and g = [1,3,3,1] / 4:
//     Kh[i][v] = sum_j A[i][j] g[v - j]                      (3 x 6)
//     U_i,pb[rho][j] = sum_kx Kh[i][2 kx + 1 - pb] * x[rho][j + kx - 1]     (3 taps, per kernel row i and column parity pb)
// are rows of the transposed convolution that are already final horizontally.  Vertically, the sequence
//     ..., O[r-1], E[r], O[r], E[r+1], ...   with   E[r] = U_0[r-1] + U_2[r],   O[r] = U_1[r]
// is filtered by the 4-tap g:   y[2r] = g0 O[r-1] + g1 E[r] + g2 O[r] + g3 E[r+1],   y[2r+1] = g0 E[r] + g1 O[r] + g2 E[r+1] +
// g3 O[r+1].  A wave walks DOWN a 32-position column: at input row rho it reads each x fragment of that row once and
// feeds three MFMA chains with it (E[rho] += U_2, O[rho] = U_1, E[rho+1] = U_0: 36 MFMAs per 12 LDS reads), then
// advances the vertical FIR on the f32 accumulators (three running partial sums, no copies) and emits two output
// rows.  No vertical halo is ever recomputed; the phase form spends 36 tap-MFMAs per position and output parity
// PAIR, this one 18.  The weights (36 fragments: one column parity) stay in registers for the whole walk.
//   workgroup = 64 positions x a segment of rows of one sample; wave = (column parity pb, 32-position half)
//   LDS: the current + next input row (2 x 66 px), two epilogue tiles of 2 rows x 128 px (de-interleaved by parity and
//   XOR-swizzled so the 8-byte epilogue writes spread over the banks; read out as full 16-byte NHWC pieces), the O
//   chain's weight fragments, bias
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "internal.h"

namespace maua {

namespace {
constexpr int UW_TW = 64;  // positions per strip
}

// the 32x32x16 matrix instruction of the kernel's 16-bit format on operands held as any 16-byte register type
template <typename F, typename A, typename B>
__device__ __forceinline__ f32x16 mfma16(const A& a, const B& b, const f32x16& c) {
  f32x16 r = c;
  Mma16<F>::step(r, __builtin_bit_cast(u32x4, a), __builtin_bit_cast(u32x4, b));
  return r;
}

template <int CI, int CO, typename F>
__global__ __launch_bounds__(256, 2) void upwalk_kernel(HiresArgs a, int seg_rows) {
  constexpr int KS = CI / 16;
  constexpr int PIECES = CI * 2 / 16;
  constexpr int RSH = CI * 2 + 16;
  constexpr int TW = UW_TW, HPX = TW + 2;
  constexpr int OPX = 2 * TW;
  constexpr int ES = CO * 2;      // epilogue tile pixel stride: no padding, the 16-byte pieces are XOR-swizzled instead
  constexpr int HREGS = (HPX * PIECES + 255) / 256;
  constexpr int PPP = CO * 2 / 16;  // 16-byte pieces per output pixel
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xrow = smem;                              // [2][HPX * RSH]
  char* epi = smem + 2 * HPX * RSH;               // [2][2 rows][OPX * ES]
  u32x4* wl = reinterpret_cast<u32x4*>(epi + 4 * OPX * ES);   // [pb 2][kx 3][KS][64 lanes]: A fragments of the O chain
  float* bias_s = reinterpret_cast<float*>(wl + 2 * 3 * KS * 64);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int pb = wave & 1, jbase = (wave >> 1) * 32;
  const int b = blockIdx.z;
  const int j0 = blockIdx.x * TW;
  const int r0 = blockIdx.y * seg_rows, r1 = min(r0 + seg_rows, a.H);
  const int nsteps = r1 - r0 + 2;
  const uint16_t* xb = reinterpret_cast<const uint16_t*>(a.x) + (long)b * a.H * a.W * CI;
  const int Wo = a.W * 2;
  char* yb = reinterpret_cast<char*>(a.y) + (long)b * (a.H * 2) * Wo * CO * 2;
  const float* nb = a.noise ? a.noise + (long)b * a.noise_bstride : nullptr;

  // ---- A fragments: Kh[i][2 kx + 1 - pb][co][ci] * s[ci] * d[co] * gain -> bf16, resident for the whole walk: the two E
  // chains (i = 0, 2) in registers, the O chain (i = 1) in LDS (36 fragments + 6 accumulators do not fit 256 registers)
  u32x4 wf[6 * KS];
  {
    const float dco = (a.d ? a.d[(long)b * CO + r] : 1.f) * a.gain;
    const float* sb = a.s + (long)b * CI;
    float sv[KS][8];
#pragma unroll
    for (int cs = 0; cs < KS; cs++)
#pragma unroll
      for (int e = 0; e < 8; e++) sv[cs][e] = sb[cs * 16 + 8 * h + e] * dco;
    const uint16_t* wbase = reinterpret_cast<const uint16_t*>(a.w);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int kx = 0; kx < 3; kx++)
#pragma unroll
        for (int cs = 0; cs < KS; cs++) {
          const uint16_t* src = wbase + ((((long)i * 2 + pb) * 3 + kx) * CO + r) * CI + cs * 16 + 8 * h;
          const u32x4 v = *reinterpret_cast<const u32x4*>(src);
          u32x4 o;
#pragma unroll
          for (int k = 0; k < 4; k++)
            o[k] = Fmt16<F>::pack2(Fmt16<F>::lo(v[k]) * sv[cs][2 * k], Fmt16<F>::hi(v[k]) * sv[cs][2 * k + 1]);
          if (i == 1) { if (wave < 2) wl[((pb * 3 + kx) * KS + cs) * 64 + lane] = o; }
          else wf[((i >> 1) * 3 + kx) * KS + cs] = o;
        }
  }
  if (tid < CO) bias_s[tid] = (a.bias ? a.bias[tid] : 0.f) * a.gain;
  const float nz_scale = a.noise_strength * a.gain * (a.noise_scale ? a.noise_scale[b] : 1.f);
  const float cl = a.clamp >= 0.f ? a.clamp : 3.0e38f;

  // ---- input rows: piece e = tid + 256 i of the 66-pixel row -> (pixel, 16-byte piece); rows / columns outside the
  // image are zero (the transposed convolution's zero padding)
  u32x4 hreg[HREGS];
#define MAUA_UW_LOAD_ROW(RHO)                                                                          \
  {                                                                                                     \
    const int gy_ = (RHO);                                                                              \
    _Pragma("unroll") for (int i = 0; i < HREGS; i++) {                                                \
      const int e = tid + i * 256, px = e / PIECES, q = e - px * PIECES;                                \
      const int gx = j0 - 1 + px;                                                                       \
      hreg[i] = u32x4{0u, 0u, 0u, 0u};                                                                  \
      if (e < HPX * PIECES && gy_ >= 0 && gy_ < a.H && gx >= 0 && gx < a.W)                             \
        hreg[i] = *reinterpret_cast<const u32x4*>(xb + (unsigned)((gy_ * a.W + gx) * CI + q * 8));      \
    }                                                                                                   \
  }
#define MAUA_UW_STORE_ROW(BUF)                                                                          \
  {                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < HREGS; i++) {                                                \
      const int e = tid + i * 256, px = e / PIECES, q = e - px * PIECES;                                \
      if (e < HPX * PIECES) *reinterpret_cast<u32x4*>(xrow + (BUF) * (HPX * RSH) + px * RSH + q * 16) = hreg[i]; \
    }                                                                                                   \
  }
  MAUA_UW_LOAD_ROW(r0 - 1)
  MAUA_UW_STORE_ROW(0)
  MAUA_UW_LOAD_ROW(r0)

  f32x16 ecur, p0, p1, q0;
#pragma unroll
  for (int e = 0; e < 16; e++) { ecur[e] = 0.f; p0[e] = 0.f; p1[e] = 0.f; q0[e] = 0.f; }
  const float g0 = 0.25f, g1 = 0.75f;  // g = [1,3,3,1] / 4 (g2 = g1, g3 = g0)
  float nz_cur = 0.f;

#pragma unroll 1
  for (int k = 0; k < nsteps; k++) {
    const int rho = r0 - 1 + k;
    // this step completes output rows 2 (rho - 1), 2 (rho - 1) + 1; the first two steps of a segment only fill the
    // pipeline (their tiles are written to LDS like any other and never stored)
    __syncthreads();  // row k & 1 is in LDS; the other row buffer and the epilogue tile written two steps ago are free
    MAUA_UW_STORE_ROW((k + 1) & 1)
    MAUA_UW_LOAD_ROW(rho + 2)      // flies during this step's MFMAs (rows past the segment: loaded, never used)
    // lane (h, r): the noise of output pixel (2 rho + h, 2 (j0 + jbase + r) + pb) - the NEXT step's rows, in a register
    float nz_next = 0.f;
    if (nb) nz_next = nb[(unsigned)((2 * min(max(rho, 0), a.H - 1) + h) * Wo + 2 * (j0 + jbase + r) + pb)];
    // ---- read-out of the previous step's two output rows: full 16-byte NHWC pieces
    if (k >= 3) {
      const char* et = epi + ((k - 1) & 1) * (2 * OPX * ES);
      const int orow = 2 * (rho - 2);
#pragma unroll
      for (int i = 0; i < 2 * OPX * PPP / 256; i++) {
        int p = tid + i * 256;
        asm volatile("" : "+v"(p));  // re-derive the piece coordinates per step instead of keeping them in registers
        const int row = p / (OPX * PPP), rem = p - row * (OPX * PPP);
        const int px = rem / PPP, pc = rem - px * PPP;
        const int slot = (px & 1) * TW + (px >> 1);
        *reinterpret_cast<uint4*>(yb + ((unsigned)((orow + row) * Wo + 2 * j0 + px) * CO) * 2 + pc * 16) =
            *reinterpret_cast<const uint4*>(et + (row * OPX + slot) * ES + ((pc ^ ((slot >> 2) & 3)) * 16));
      }
    }
    // ---- multiply: every fragment of input row rho feeds the three chains
    f32x16 o, enext;
#pragma unroll
    for (int e = 0; e < 16; e++) { o[e] = 0.f; enext[e] = 0.f; }
    const char* abase = xrow + (k & 1) * (HPX * RSH) + (jbase + r) * RSH + h * 16;
#pragma unroll
    for (int kx = 0; kx < 3; kx++)
#pragma unroll
      for (int cs = 0; cs < KS; cs++) {
        const bf16x8 av = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(abase + kx * RSH + cs * 32));
        const bf16x8 wo = __builtin_bit_cast(bf16x8, wl[((pb * 3 + kx) * KS + cs) * 64 + lane]);
        ecur = mfma16<F>(__builtin_bit_cast(bf16x8, wf[(1 * 3 + kx) * KS + cs]), av, ecur);
        o = mfma16<F>(wo, av, o);
        enext = mfma16<F>(__builtin_bit_cast(bf16x8, wf[(0 * 3 + kx) * KS + cs]), av, enext);
      }
    // ---- vertical FIR on the accumulators + epilogue: lane = position jbase + r, 16 channels in 4 quads
    // (each lane finishes both rows of its pixel: the other row's noise sits in the lane 32 away)
    const float nz_other = __shfl_xor(nz_cur, 32);
    const float nz0 = (h ? nz_other : nz_cur) * nz_scale, nz1 = (h ? nz_cur : nz_other) * nz_scale;
    nz_cur = nz_next;
    const int eslot = pb * TW + jbase + r, esw = (eslot >> 2) & 3;
    char* et = epi + (k & 1) * (2 * OPX * ES) + eslot * ES + h * 8;
#pragma unroll
    for (int qd = 0; qd < 4; qd++) {
      const float4 b4 = *reinterpret_cast<const float4*>(bias_s + 8 * qd + 4 * h);
      const float bq[4] = {b4.x, b4.y, b4.z, b4.w};
      float v0[4], v1[4];
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        const int e = qd * 4 + kk;
        const float ev = ecur[e], ov = o[e];
        float y0 = fmaf(g0, ev, p0[e]);
        float y1 = fmaf(g0, ov, fmaf(g1, ev, p1[e]));
        p0[e] = fmaf(g1, ov, fmaf(g1, ev, q0[e]));
        p1[e] = fmaf(g1, ov, g0 * ev);
        q0[e] = g0 * ov;
        y0 += nz0 + bq[kk];
        y1 += nz1 + bq[kk];
        y0 = fmaxf(y0, y0 * a.alpha);
        y1 = fmaxf(y1, y1 * a.alpha);
        v0[kk] = __builtin_amdgcn_fmed3f(y0, -cl, cl);
        v1[kk] = __builtin_amdgcn_fmed3f(y1, -cl, cl);
      }
      *reinterpret_cast<uint2*>(et + ((qd ^ esw) * 16)) = make_uint2(Fmt16<F>::pack2(v0[0], v0[1]), Fmt16<F>::pack2(v0[2], v0[3]));
      *reinterpret_cast<uint2*>(et + OPX * ES + ((qd ^ esw) * 16)) = make_uint2(Fmt16<F>::pack2(v1[0], v1[1]), Fmt16<F>::pack2(v1[2], v1[3]));
    }
    ecur = enext;
  }
  __syncthreads();
  {  // the last step's rows
    const int k = nsteps;
    const char* et = epi + ((k - 1) & 1) * (2 * OPX * ES);
    const int orow = 2 * (r1 - 1);
#pragma unroll
    for (int i = 0; i < 2 * OPX * PPP / 256; i++) {
      const int p = tid + i * 256;
      const int row = p / (OPX * PPP), rem = p - row * (OPX * PPP);
      const int px = rem / PPP, pc = rem - px * PPP;
      const int slot = (px & 1) * TW + (px >> 1);
      *reinterpret_cast<uint4*>(yb + ((unsigned)((orow + row) * Wo + 2 * j0 + px) * CO) * 2 + pc * 16) =
          *reinterpret_cast<const uint4*>(et + (row * OPX + slot) * ES + ((pc ^ ((slot >> 2) & 3)) * 16));
    }
  }
#undef MAUA_UW_LOAD_ROW
#undef MAUA_UW_STORE_ROW
}

// ------------------------------------------------------------------------------------------------------------------
// The whole last block in one walk: conv0 (up, half-folded as above) -> conv1 (3x3) -> toRGB + up-sampled skip -> f32
// image / u8 frame (stylegan2.py:351-378 for one block).  The block's 1024^2 x 32 feature maps never reach HBM: four
// PRODUCER waves run the walk above and write the activated conv0 rows (bf16) into a ring of six rows in LDS, four
// CONSUMER waves follow one step behind, convolve the ring rows with conv1's weights (in registers, styles /
// demodulation / gain folded in), finish noise + bias + lrelu + clamp in registers, feed the result - without leaving
// the registers: toRGB's K index is permuted to the accumulator layout - to the toRGB MFMA, add the FIR-upsampled
// previous image and store the pixels.  Per step (one input row): producers 4 x 36 MFMAs, consumers 4 x 40.
//   strip: 126 output px (conv1 needs one conv0 pixel either side: the producers cover 128 px; parity-1 waves start one
//   position early), i.e. 63 input positions; 9 strips cover 1024 px.  Rows: a segment of r1 - r0 input rows needs
//   r1 - r0 + 5 steps (two to fill the vertical FIR, one row of conv0 above / below for conv1, one step of lag).
//   One workgroup of 8 waves per CU (256 registers per wave); wave w and w + 4 share a SIMD.  Workgroups are
//   persistent: each takes a contiguous run of (sample, segment, strip) items and reloads its weights only when the
//   sample changes.
//   Every wave is a serial instruction stream and the step is as long as the slower role's stream, so per-step
//   instructions are what counts: all staging is LDS-direct DMA issued by the producers with scalar row bases and
//   per-lane offsets fixed per item (input row: 9 x 1 KB pieces, 16-byte pieces XOR-swizzled on the source side; one row
//   of the previous image per step: 198 dwords), the skip's FIR coefficients are per-lane constants (rows / columns
//   outside the image are staged as zeros), LDS operands are requested two iterations ahead of their MFMAs.
struct WalkFusedArgs {
  HiresArgs up, c1;
};

__device__ __forceinline__ void lds_dma_b32(const void* sbase, unsigned voff_bytes, void* lds_wave_base) {
  const unsigned base =
      __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff_bytes), "s"(sbase), "s"(base)
               : "memory");
}

template <int CI, int CM, bool DBG, typename F>
__global__ __launch_bounds__(512, 1) void upwalk_fused_kernel(WalkFusedArgs A, int seg_rows, int nseg, int strips,
                                                             int n_items, int items_per_wg, int narrow_last, long long* dbg) {
  constexpr int KS = CI / 16, KS1 = CM / 16;
  constexpr int PITCH = 126, XPX = 67, RPX = 132;
  constexpr int XG = (XPX + 7) / 8;         // 1 KB DMA pieces (8 positions) per staged input row
  constexpr int XROWB = XG * 1024;          // bytes per staged input row (128 bytes per position, pieces XOR-swizzled)
  constexpr int RROW = RPX * CM * 2;        // bytes per ring row (64 bytes per pixel, 16-byte pieces XOR-swizzled)
  constexpr int PVW = 256, PVP = 66;        // staged row of the previous image: [3 channels][66 px] floats, padded
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xrow = smem;                                            // [2][XROWB]
  char* ring = smem + 2 * XROWB;                                // [3 slots][2 rows][RROW]
  u32x4* wl = reinterpret_cast<u32x4*>(ring + 6 * RROW);        // [pb 2][chain i 3][kx 3][KS][64 lanes]: conv0's A fragments
  float* pvs = reinterpret_cast<float*>(wl + 2 * 3 * 3 * KS * 64);   // [3 slots][PVW]
  float* bias0_s = pvs + 3 * PVW;
  float* bias1_s = bias0_s + CM;
  // (both in accumulator order: lane half h, element e = channel 8 (e >> 2) + 4 h + (e & 3); conv0's halved: its four
  //  FIR taps sum to 2)

  const HiresArgs& a = A.up;
  const HiresArgs& c = A.c1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const bool producer = wave < 4;
  const int widx = wave & 3;
  const int Ho = a.H * 2, Wo = a.W * 2, Hp = a.H, Wp = a.W;
  const unsigned HWl = (unsigned)Ho * (unsigned)Wo;
  const int item0 = blockIdx.x * items_per_wg, item1 = min(item0 + items_per_wg, n_items);
  int b_loaded = -1;
  long long dsum[DBG ? 5 : 1] = {0}, tlast = 0, steps_total = 0;   // (DBG: s_memtime per phase, one workgroup reports)
#define MAUA_TICK(stmt) if constexpr (DBG) { stmt; }
#define MAUA_NOW() (DBG ? (long long)__builtin_amdgcn_s_memtime() : 0LL)

  const int pb = widx & 1, t = (widx >> 1) * 32 + r;
  // (each role has its own item loop - same barrier count per item - so that only its own state is live in it)
  if (producer) {
    for (int item = item0; item < item1; item++) {
      const int b = item / (nseg * strips);
      const int rem = item - b * (nseg * strips);
      const int seg = rem / strips, strip = rem - seg * strips;
      const int X0i = strip * PITCH, J0 = strip * (PITCH / 2);
      const int r0i = seg * seg_rows, r1i = min(r0i + seg_rows, a.H);
      // A NARROW last strip (<= 32 output columns: 1024 = 8 x 126 + 16) is walked as two sub-items at once - the upper and the
      // lower half of the item's rows - in half the steps: sub-item A lives where the strip's first 32 positions / 64 ring pixels
      // live (producer waves 0, 1 and consumer wave 0), sub-item B in the coordinates of positions 32.. / ring pixels 64..
      // (producer waves 2, 3 and consumer wave 2: the same columns of the image, `drow` rows further down).  Every wave sees an
      // ordinary strip whose origin is X0 (and whose rows are r0 .. r1); only the cooperative staging knows about the halves.
      const bool narrow = narrow_last && strip == strips - 1;
      const int hrows = narrow ? (r1i - r0i + 1) >> 1 : r1i - r0i, drow = narrow ? hrows : 0;
      const int sub = narrow ? (widx >> 1) : 0;
      const int X0 = X0i - 64 * sub;
      const int r0 = r0i + sub * hrows, r1 = min(r0 + hrows, r1i);
      const int nsteps = hrows + 5;
      __syncthreads();   // the previous item's LDS reads are done
      // ---- zero both staged-row buffers (positions outside the image stay zero: the DMA skips them)
      for (int i = tid; i < 2 * XROWB / 16; i += 256) reinterpret_cast<u32x4*>(xrow)[i] = u32x4{0u, 0u, 0u, 0u};
      if (b != b_loaded) {
        b_loaded = b;
        if (tid < 2 * CM) {
          const int hh = (tid >> 4) & 1, e = tid & 15, ch = 8 * (e >> 2) + 4 * hh + (e & 3);
          if (tid < CM) bias0_s[tid] = 0.5f * (a.bias ? a.bias[ch] : 0.f) * a.gain;
          else bias1_s[tid - CM] = (c.bias ? c.bias[ch] : 0.f) * c.gain;
        }
        // A fragments: Kh[i][2 kx + 1 - pb][co][ci] * s[ci] * d[co] * gain -> bf16, in LDS (per column parity; read next
        // to the x fragments: 36 fragments + 6 accumulators + the prefetch ring do not fit 256 registers)
        const float dco = (a.d ? a.d[(long)b * CM + r] : 1.f) * a.gain;
        const float* sb = a.s + (long)b * CI;
        float sv[KS][8];
#pragma unroll
        for (int cs = 0; cs < KS; cs++)
#pragma unroll
          for (int e = 0; e < 8; e++) sv[cs][e] = sb[cs * 16 + 8 * h + e] * dco;
        const uint16_t* wbase = reinterpret_cast<const uint16_t*>(a.w);
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int kx = 0; kx < 3; kx++)
#pragma unroll
            for (int cs = 0; cs < KS; cs++) {
              const uint16_t* src = wbase + ((((long)i * 2 + pb) * 3 + kx) * CM + r) * CI + cs * 16 + 8 * h;
              const u32x4 v = *reinterpret_cast<const u32x4*>(src);
              u32x4 o;
#pragma unroll
              for (int k = 0; k < 4; k++)
                o[k] = Fmt16<F>::pack2(Fmt16<F>::lo(v[k]) * sv[cs][2 * k], Fmt16<F>::hi(v[k]) * sv[cs][2 * k + 1]);
              if (widx < 2) wl[(((pb * 3 + i) * 3 + kx) * KS + cs) * 64 + lane] = o;
            }
      }
      __syncthreads();
      // ============================================================ conv0: half-folded walk (see upwalk_kernel) + staging
      const float nz_scale = a.noise_strength * a.gain * (a.noise_scale ? a.noise_scale[b] : 1.f);
      const float cl = a.clamp >= 0.f ? a.clamp : 3.0e38f;
      const char* nbase = a.noise ? reinterpret_cast<const char*>(a.noise + (long)b * a.noise_bstride) : nullptr;
      const char* xbase = reinterpret_cast<const char*>(a.x) + (long)b * a.H * a.W * CI * 2;
      const char* pvbase = c.rgb_prev ? reinterpret_cast<const char*>(c.rgb_prev + (long)b * 3 * Hp * Wp) : nullptr;
      const int p = 2 * t + (1 - pb);              // ring pixel of this lane; image pixel X0 - 1 + p
      const int px = X0 - 1 + p;
      const bool px_ok = px >= 0 && px < Wo;
      const unsigned nzoff = (unsigned)(h * Wo + min(max(px, 0), Wo - 1)) * 4u;
      // ring rows keep even and odd pixels in separate halves (slot = (p & 1) * 66 + (p >> 1)) with the 16-byte pieces
      // XOR-swizzled by (p >> 2) & 3: the consumers' ds_read_b128 are conflict-free, these 8-byte writes 2-way
      const int ring_off = ((p & 1) * (RPX / 2) + (p >> 1)) * (CM * 2) + h * 8, esw = (p >> 2) & 3;
      // DMA descriptors: piece j of this wave = positions 8 (widx + 4 j) .. + 7; lane = (position, 16-byte slot); the
      // lane fetches the source piece that belongs in its slot after the XOR swizzle
      unsigned xoff[3];
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int grp = widx + 4 * j, q = grp * 8 + (lane >> 3), gx = J0 - 2 + q - ((narrow && grp >= 4) ? 32 : 0);
        const bool ok = grp < XG && q < XPX && gx >= 0 && gx < a.W;
        xoff[j] = ok ? (unsigned)((gx * CI + (((lane & 7) ^ ((q >> 1) & 7)) * 8)) * 2) : 0xffffffffu;
      }
      const int Jp0 = (X0i - 1) >> 1;              // first column of the previous image the strip's skip taps read
      unsigned pvoff;
      int pvd;                                     // rows between this lane's staged pixel and sub-item A's row
      {
        const int f = widx * 64 + lane, ch = f / PVP, pxi = f - ch * PVP;
        const bool lower = narrow && pxi >= 32;    // (staged pixels 32.. belong to sub-item B)
        const int gx = Jp0 + pxi - (lower ? 32 : 0);
        pvd = lower ? drow : 0;
        pvoff = (f < 3 * PVP) ? (unsigned)((ch * Hp + pvd) * Wp + min(max(gx, 0), Wp - 1)) * 4u : 0xffffffffu;
      }
      // byte offsets of the x fragments inside a staged row: position q = (1 - pb) + t + kx, piece (2 cs + h) ^ swizzle
      int xfo[3 * KS];
#pragma unroll
      for (int kx = 0; kx < 3; kx++)
#pragma unroll
        for (int cs = 0; cs < KS; cs++) {
          const int q = (1 - pb) + t + kx;
          xfo[kx * KS + cs] = q * 128 + (((2 * cs + h) ^ ((q >> 1) & 7)) << 4);
        }
      // (RHO: sub-item A's row; positions 32.. - pieces 4.. - hold the row `drow` further down, which is the same row unless narrow)
#define MAUA_UWF_STAGE_X(RHO, BUF)                                                                      \
  {                                                                                                     \
    char* dst_ = xrow + (BUF) * XROWB;                                                                  \
    _Pragma("unroll") for (int hs_ = 0; hs_ < 2; hs_++) {                                              \
      const int gy_ = (RHO) + hs_ * drow;                                                               \
      if (gy_ >= 0 && gy_ < a.H) {                                                                      \
        const char* rb_ = xbase + (long)gy_ * a.W * CI * 2;                                             \
        _Pragma("unroll") for (int j = hs_; j < 1 + 2 * hs_; j++)                                      \
          if (xoff[j] != 0xffffffffu) lds_dma_b128(rb_, xoff[j], dst_ + (widx + 4 * j) * 1024);         \
      } else {                                                                                          \
        for (int i = tid + 256 * hs_; i < (hs_ ? XROWB / 16 : 256); i += 256)                           \
          reinterpret_cast<u32x4*>(dst_)[i] = u32x4{0u, 0u, 0u, 0u};                                    \
      }                                                                                                 \
    }                                                                                                   \
  }
      MAUA_UWF_STAGE_X(r0i - 2, 0)
      const f32x16* bhalf_p = reinterpret_cast<const f32x16*>(bias0_s + 16 * h);   // bias * gain / 2 of the lane's 16 channels
      f32x16 ecur = *bhalf_p, p0, p1, q0;
#pragma unroll
      for (int e = 0; e < 16; e++) { p0[e] = 0.f; p1[e] = 0.f; q0[e] = 0.f; }
      const float g0 = 0.25f, g1 = 0.75f;
      float nz_cur = 0.f;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
      for (int k = 0; k < nsteps; k++) {
        const int rho = r0 - 2 + k;
        const long long tz = MAUA_NOW();
        __syncthreads();
        const long long ta = MAUA_NOW();
        MAUA_TICK(if (k) dsum[2] += tz - tlast)
        MAUA_TICK(dsum[3] += ta - tz)
        if (k == nsteps - 1) continue;   // (the consumers' last step; the loop ends right after)
        // ---- staging: input row rho + 1 and row rho - 1 of the previous image (slot k % 3) for the next step
        MAUA_UWF_STAGE_X(r0i - 2 + k + 1, (k + 1) & 1)
        if (pvbase) {
          const int m = r0i - 2 + k - 1;
          float* dst = pvs + (k % 3) * PVW;
          if (m + pvd >= 0 && m + pvd < Hp) {
            if (pvoff != 0xffffffffu) lds_dma_b32(pvbase + (long)m * Wp * 4, pvoff, dst + widx * 64);
          } else {
            dst[widx * 64 + lane] = 0.f;
          }
        }
        float nz_next = 0.f;
        if (nbase) nz_next = *reinterpret_cast<const float*>(nbase + (long)(2 * min(max(rho, 0), a.H - 1)) * Wo * 4 + nzoff);
        f32x16 o = *bhalf_p, enext = o;
        const char* abase = xrow + (k & 1) * XROWB;
        const u32x4* wl0 = wl + (pb * 3 + 0) * 3 * KS * 64 + lane;
        const u32x4* wl1 = wl + (pb * 3 + 1) * 3 * KS * 64 + lane;
        const u32x4* wl2 = wl + (pb * 3 + 2) * 3 * KS * 64 + lane;
        // 12 (kx, k-step) iterations of 3 MFMAs; the LDS operands of iteration it + 2 are requested before the MFMAs
        // of iteration it (three register slots), pinned with scheduling barriers: left alone the compiler reads each
        // fragment right before its use and the wave sits out every LDS latency
        constexpr int PD = 3;   // register slots: operands are requested PD - 1 iterations ahead (4 measured no faster)
        u32x4 pa[PD], pw0[PD], pw1[PD], pw2[PD];
#define MAUA_UWF_PLOAD(IT)                                                  \
  {                                                                         \
    pa[(IT) % PD] = *reinterpret_cast<const u32x4*>(abase + xfo[IT]);       \
    pw0[(IT) % PD] = wl0[(IT) * 64];                                        \
    pw1[(IT) % PD] = wl1[(IT) * 64];                                        \
    pw2[(IT) % PD] = wl2[(IT) * 64];                                        \
  }
#pragma unroll
        for (int it = 0; it < PD - 1; it++) MAUA_UWF_PLOAD(it)
#pragma unroll
        for (int it = 0; it < 3 * KS; it++) {
          if (it + PD - 1 < 3 * KS) MAUA_UWF_PLOAD(it + PD - 1)
          __builtin_amdgcn_sched_barrier(0);
          const bf16x8 av = __builtin_bit_cast(bf16x8, pa[it % PD]);
          ecur = mfma16<F>(__builtin_bit_cast(bf16x8, pw2[it % PD]), av, ecur);
          o = mfma16<F>(__builtin_bit_cast(bf16x8, pw1[it % PD]), av, o);
          enext = mfma16<F>(__builtin_bit_cast(bf16x8, pw0[it % PD]), av, enext);
          __builtin_amdgcn_sched_barrier(0);
        }
#undef MAUA_UWF_PLOAD
        const long long tb = MAUA_NOW();
        MAUA_TICK(dsum[0] += tb - ta)
        const float nz_other = __shfl_xor(nz_cur, 32);
        const f32x2_t nz0 = (h ? nz_other : nz_cur) * nz_scale, nz1 = (h ? nz_cur : nz_other) * nz_scale;
        nz_cur = nz_next;
        // vertical FIR + epilogue on register pairs (packed f32: 8 FIR operations per pair); the bias rides in the
        // accumulators (each chain starts at bias / 2 and the four taps sum to 2)
        char* et = ring + (k % 3) * (2 * RROW) + ring_off;
        const f32x2_t al = a.alpha;
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
          uint32_t w0[2], w1[2];
#pragma unroll
          for (int kk = 0; kk < 4; kk += 2) {
            const int e = qd * 4 + kk;
            const f32x2_t ev = {ecur[e], ecur[e + 1]}, ov = {o[e], o[e + 1]};
            const f32x2_t pp0 = {p0[e], p0[e + 1]}, pp1 = {p1[e], p1[e + 1]}, qq0 = {q0[e], q0[e + 1]};
            f32x2_t y0 = g0 * ev + pp0;
            f32x2_t y1 = g0 * ov + (g1 * ev + pp1);
            const f32x2_t n0 = g1 * ov + (g1 * ev + qq0);
            const f32x2_t n1 = g1 * ov + g0 * ev;
            const f32x2_t nq = g0 * ov;
            p0[e] = n0[0]; p0[e + 1] = n0[1];
            p1[e] = n1[0]; p1[e + 1] = n1[1];
            q0[e] = nq[0]; q0[e + 1] = nq[1];
            y0 += nz0;
            y1 += nz1;
            const f32x2_t s0 = y0 * al, s1 = y1 * al;
            w0[kk >> 1] = Fmt16<F>::pack2(__builtin_amdgcn_fmed3f(fmaxf(y0[0], s0[0]), -cl, cl), __builtin_amdgcn_fmed3f(fmaxf(y0[1], s0[1]), -cl, cl));
            w1[kk >> 1] = Fmt16<F>::pack2(__builtin_amdgcn_fmed3f(fmaxf(y1[0], s1[0]), -cl, cl), __builtin_amdgcn_fmed3f(fmaxf(y1[1], s1[1]), -cl, cl));
          }
          *reinterpret_cast<uint2*>(et + ((qd ^ esw) * 16)) = make_uint2(w0[0], w0[1]);
          *reinterpret_cast<uint2*>(et + RROW + ((qd ^ esw) * 16)) = make_uint2(w1[0], w1[1]);
        }
        // rows 2 (rho - 1), 2 (rho - 1) + 1 of conv0's output: outside the image, and in the pixels left / right of it,
        // conv1 sees zeros (rare lanes / steps: fixed up after the fact)
        if (!(px_ok && rho - 1 >= 0 && rho - 1 < a.H)) {
#pragma unroll
          for (int qd = 0; qd < 4; qd++) {
            *reinterpret_cast<uint2*>(et + qd * 16) = make_uint2(0u, 0u);
            *reinterpret_cast<uint2*>(et + RROW + qd * 16) = make_uint2(0u, 0u);
          }
        }
        ecur = enext;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the rows staged for the next step have landed
        tlast = MAUA_NOW();
        MAUA_TICK(dsum[1] += tlast - tb)
      }
#undef MAUA_UWF_STAGE_X
      MAUA_TICK(steps_total += nsteps)
    }
  } else {
    u32x4 w1[9 * KS1];       // conv1's A fragments
    u32x4 rf[KS1];           // toRGB's A fragments
    for (int item = item0; item < item1; item++) {
      const int b = item / (nseg * strips);
      const int rem = item - b * (nseg * strips);
      const int seg = rem / strips, strip = rem - seg * strips;
      const int X0i = strip * PITCH;
      const int r0i = seg * seg_rows, r1i = min(r0i + seg_rows, a.H);
      // (narrow last strip: consumer wave 0 finishes sub-item A, wave 2 sub-item B, waves 1 and 3 only keep the barriers - see the
      //  producers' item set-up)
      const bool narrow = narrow_last && strip == strips - 1;
      const int hrows = narrow ? (r1i - r0i + 1) >> 1 : r1i - r0i;
      const int sub = narrow ? (widx >> 1) : 0;
      const bool active = !narrow || (widx & 1) == 0;
      const int X0 = X0i - 64 * sub;
      const int r0 = r0i + sub * hrows, r1 = min(r0 + hrows, r1i);
      const int nsteps = hrows + 5;
      __syncthreads();
      if (b != b_loaded) {
        b_loaded = b;
        // conv1's A fragments: W1[tap][co][ci] * s1[ci] * d1[co] * gain
        const float dco = (c.d ? c.d[(long)b * CM + r] : 1.f) * c.gain;
        const float* sb = c.s + (long)b * CM;
        float sv[KS1][8];
#pragma unroll
        for (int cs = 0; cs < KS1; cs++)
#pragma unroll
          for (int e = 0; e < 8; e++) sv[cs][e] = sb[cs * 16 + 8 * h + e] * dco;
        const uint16_t* wbase = reinterpret_cast<const uint16_t*>(c.w);
#pragma unroll
        for (int tap = 0; tap < 9; tap++)
#pragma unroll
          for (int cs = 0; cs < KS1; cs++) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(wbase + ((long)tap * CM + r) * CM + cs * 16 + 8 * h);
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; k++)
              o[k] = Fmt16<F>::pack2(Fmt16<F>::lo(v[k]) * sv[cs][2 * k], Fmt16<F>::hi(v[k]) * sv[cs][2 * k + 1]);
            w1[tap * KS1 + cs] = o;
          }
        // toRGB A fragments: rows 0..2 = bf16(hi) of the pre-modulated RGB weights, rows 8..10 the bf16 remainder, rows
        // 4..6 / 12..14 repeat them (the h == 1 lanes then hold the same sums); K in ACCUMULATOR order: element e of
        // k-step ks in lane half h is channel 16 ks + 8 (e >> 2) + 4 h + (e & 3), so conv1's activated outputs are the B
        // operand as they sit in the registers
        const int c_rgb = (r < 16 && (r & 3) < 3) ? (r & 3) : -1;
#pragma unroll
        for (int ks = 0; ks < KS1; ks++) {
          u32x4 o = u32x4{0u, 0u, 0u, 0u};
          if (c_rgb >= 0) {
            const float* src = c.rgb_wmod + ((long)b * 3 + c_rgb) * CM + ks * 16 + 4 * h;
#pragma unroll
            for (int k = 0; k < 4; k++) {
              float w0 = src[8 * (k >> 1) + 2 * (k & 1)], w1v = src[8 * (k >> 1) + 2 * (k & 1) + 1];
              const float h0 = Fmt16<F>::round(w0), h1 = Fmt16<F>::round(w1v);
              if (r >= 8) { w0 -= h0; w1v -= h1; }
              o[k] = Fmt16<F>::pack2(w0, w1v);
            }
          }
          rf[ks] = o;
        }
      }
      __syncthreads();
      // ============================================================ conv1 + toRGB + skip on the ring rows
      const float rgb_b0 = c.rgb_bias[0], rgb_b1 = c.rgb_bias[1], rgb_b2 = c.rgb_bias[2];
      const float nz_scale = c.noise_strength * c.gain * (c.noise_scale ? c.noise_scale[b] : 1.f);
      const float cl = c.clamp >= 0.f ? c.clamp : 3.0e38f;
      const char* nbase = c.noise ? reinterpret_cast<const char*>(c.noise + (long)b * c.noise_bstride) : nullptr;
      const int pcol = widx * 32 + r;                     // column inside the strip; ring pixel of tap dx: pcol + dx
      const int px = X0 + pcol;
      const bool px_ok = pcol < PITCH && px < Wo;
      const int pxc = min(px, Wo - 1);
      const unsigned nzoff = (unsigned)pxc * 4u;
      int foff[3][KS1];                                  // byte offsets of the B fragments inside a ring row
#pragma unroll
      for (int dx = 0; dx < 3; dx++)
#pragma unroll
        for (int ks = 0; ks < KS1; ks++) {
          const int pp = pcol + dx;
          foff[dx][ks] = ((pp & 1) * (RPX / 2) + (pp >> 1)) * (CM * 2) + (((2 * ks + h) ^ ((pp >> 2) & 3)) * 16);
        }
      // skip image = upsample2d of the previous image in its branch-free 2x2 form (modconv_hires.hip): output pixel
      // (y, x) reads rows (y - 1) >> 1, + 1 and columns (x - 1) >> 1, + 1 with the FIR taps its parities select.  The
      // lane's pixel has fixed parities - row arow - 1 + h: odd for h == 0 - so the four coefficients are constants;
      // rows outside the image are staged as zeros, columns outside get a zero coefficient
      const bool pv_on = c.rgb_prev != nullptr;
      const int ix0 = (pxc - 1) >> 1;
      const int pvi = ix0 - ((X0 - 1) >> 1);              // index into the staged row (0 .. 64)
      float fc[4];
      {
        const bool yo = h == 0, xo = pxc & 1;
#pragma unroll
        for (int dy = 0; dy < 2; dy++)
#pragma unroll
          for (int dx = 0; dx < 2; dx++) {
            const bool uh = (dy == 0) == yo, vh = (dx == 0) == xo;
            const bool okk = ix0 + dx >= 0 && ix0 + dx < Wp;
            fc[dy * 2 + dx] = !okk ? 0.f : uh ? (vh ? c.fir[5] : c.fir[4]) : (vh ? c.fir[1] : c.fir[0]);
          }
      }
      const f32x16* bias1_p = reinterpret_cast<const f32x16*>(bias1_s + 16 * h);   // the accumulators start from bias * gain
      long long ta = 0;
      // A consumer step has the producers' phase order - multiply, then VALU - because that is how two waves share a SIMD
      // best here (scripts/ubench/issue_mix.hip: a VALU stream beside a partner that multiplies with LDS operands runs at
      // ~14 cycles per instruction, beside a partner's VALU stream at ~5; two LDS-fed MFMA streams barely slow each other:
      // each is bound by its own wave's LDS read rate, ~50 cycles per ds_read_b128):
      //   M  conv1 of rows arow - 1 (accA), arow (accB), arow = 2 (rho - 2), from ring rows arow - 2 .. arow + 1
      //   A  noise + lrelu + clamp in registers -> bf16 B fragments -> the four toRGB MFMAs
      //   B  + bias, clamp, + skip taps from LDS, store the pixel (lanes h == 0: row arow - 1, h == 1: row arow)
#pragma unroll 1
      for (int k = 0; k < nsteps; k++) {
        const int rho = r0 - 2 + k;
        const int arow = 2 * (rho - 2);
        const long long tz = MAUA_NOW();
        MAUA_TICK(if (k) dsum[4] += tz - tlast)
        __syncthreads();
        ta = MAUA_NOW();
        MAUA_TICK(dsum[0] += ta - tz)
        if (!active) continue;
        f32x2_t nz = {0.f, 0.f};   // noise of the lane's column in rows arow - 1, arow (needed after the multiply)
        if (nbase) {               // (rows clamped into the image: the clamped ones are never stored)
          nz[0] = *reinterpret_cast<const float*>(nbase + (long)min(max(arow - 1, 0), Ho - 1) * Wo * 4 + nzoff);
          nz[1] = *reinterpret_cast<const float*>(nbase + (long)min(max(arow, 0), Ho - 1) * Wo * 4 + nzoff);
        }
        const long long tb = MAUA_NOW();
        MAUA_TICK(dsum[1] += tb - ta)
        // ---- M: ring rows arow - 2 .. arow + 1 = slots (k - 2) % 3 and (k - 1) % 3.  Row arow - 2 feeds only accA and
        // row arow + 1 only accB: those two are interleaved so no two consecutive MFMAs share an accumulator; the
        // fragments are requested a stage ahead
        f32x16 accA = *bias1_p, accB = accA;
        const char* s2 = ring + ((k + 1) % 3) * (2 * RROW);
        const char* s1 = ring + ((k + 2) % 3) * (2 * RROW);
        u32x4 f0[3 * KS1], f3[3 * KS1], f1[3 * KS1];
#define MAUA_UWF_CLOAD(DST, ROWP)                                                                       \
  {                                                                                                     \
    _Pragma("unroll") for (int dx = 0; dx < 3; dx++) _Pragma("unroll") for (int ks = 0; ks < KS1; ks++) \
      DST[dx * KS1 + ks] = *reinterpret_cast<const u32x4*>((ROWP) + foff[dx][ks]);                      \
  }
        MAUA_UWF_CLOAD(f0, s2)
        MAUA_UWF_CLOAD(f3, s1 + RROW)
        MAUA_UWF_CLOAD(f1, s2 + RROW)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 3 * KS1; i++) {   // taps dy = -1 of accA / dy = +1 of accB
          accA = mfma16<F>(__builtin_bit_cast(bf16x8, w1[i]), __builtin_bit_cast(bf16x8, f0[i]), accA);
          accB = mfma16<F>(__builtin_bit_cast(bf16x8, w1[2 * 3 * KS1 + i]), __builtin_bit_cast(bf16x8, f3[i]), accB);
        }
        __builtin_amdgcn_sched_barrier(0);
        MAUA_UWF_CLOAD(f0, s1)               // ring row arow
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 3 * KS1; i++) {   // ring row arow - 1: dy = 0 of accA, dy = -1 of accB
          accA = mfma16<F>(__builtin_bit_cast(bf16x8, w1[3 * KS1 + i]), __builtin_bit_cast(bf16x8, f1[i]), accA);
          accB = mfma16<F>(__builtin_bit_cast(bf16x8, w1[i]), __builtin_bit_cast(bf16x8, f1[i]), accB);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 3 * KS1; i++) {   // ring row arow: dy = +1 of accA, dy = 0 of accB
          accA = mfma16<F>(__builtin_bit_cast(bf16x8, w1[2 * 3 * KS1 + i]), __builtin_bit_cast(bf16x8, f0[i]), accA);
          accB = mfma16<F>(__builtin_bit_cast(bf16x8, w1[3 * KS1 + i]), __builtin_bit_cast(bf16x8, f0[i]), accB);
        }
#undef MAUA_UWF_CLOAD
        const long long tc = MAUA_NOW();
        MAUA_TICK(dsum[3] += tc - tb)
        // ---- A: epilogue in registers (the bias rides in the accumulators) -> toRGB MFMAs
        u32x4 fa[KS1], fb[KS1];
        {
          const f32x2_t nA = nz[0] * nz_scale, nB = nz[1] * nz_scale, al = c.alpha;
#pragma unroll
          for (int e = 0; e < 16; e += 2) {
            const f32x2_t ya = f32x2_t{accA[e], accA[e + 1]} + nA, yb = f32x2_t{accB[e], accB[e + 1]} + nB;
            const f32x2_t sa = ya * al, sb = yb * al;
            fa[e >> 3][(e >> 1) & 3] = Fmt16<F>::pack2(__builtin_amdgcn_fmed3f(fmaxf(ya[0], sa[0]), -cl, cl), __builtin_amdgcn_fmed3f(fmaxf(ya[1], sa[1]), -cl, cl));
            fb[e >> 3][(e >> 1) & 3] = Fmt16<F>::pack2(__builtin_amdgcn_fmed3f(fmaxf(yb[0], sb[0]), -cl, cl), __builtin_amdgcn_fmed3f(fmaxf(yb[1], sb[1]), -cl, cl));
          }
        }
        f32x16 ra, rb;
#pragma unroll
        for (int e = 0; e < 16; e++) { ra[e] = 0.f; rb[e] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS1; ks++) {
          ra = mfma16<F>(__builtin_bit_cast(bf16x8, rf[ks]), __builtin_bit_cast(bf16x8, fa[ks]), ra);
          rb = mfma16<F>(__builtin_bit_cast(bf16x8, rf[ks]), __builtin_bit_cast(bf16x8, fb[ks]), rb);
        }
        // ---- B: skip = rows m - 1 (staged at step k - 2) and m (step k - 1) of the previous image, m = rho - 2
        const int oy = arow - 1 + h;
        float u3[3] = {0.f, 0.f, 0.f};
        if (pv_on) {
          const float* sA = pvs + ((k + 1) % 3) * PVW + pvi;
          const float* sB = pvs + ((k + 2) % 3) * PVW + pvi;
#pragma unroll
          for (int ch = 0; ch < 3; ch++)
            u3[ch] = sA[ch * PVP] * fc[0] + sA[ch * PVP + 1] * fc[1] + sB[ch * PVP] * fc[2] + sB[ch * PVP + 1] * fc[3];
        }
        float o3[3];
        o3[0] = (h ? rb[0] + rb[4] : ra[0] + ra[4]) + rgb_b0;
        o3[1] = (h ? rb[1] + rb[5] : ra[1] + ra[5]) + rgb_b1;
        o3[2] = (h ? rb[2] + rb[6] : ra[2] + ra[6]) + rgb_b2;
        if (c.rgb_clamp >= 0.f) {
#pragma unroll
          for (int ch = 0; ch < 3; ch++) o3[ch] = fminf(fmaxf(o3[ch], -c.rgb_clamp), c.rgb_clamp);
        }
        o3[0] = u3[0] + o3[0]; o3[1] = u3[1] + o3[1]; o3[2] = u3[2] + o3[2];
        if (px_ok && oy >= 2 * r0 && oy < 2 * r1) {
          if (!c.rgb_skip_f32) {
            float* ob = c.rgb_out + (long)b * 3 * HWl + (unsigned)(oy * Wo + px);
            ob[0] = o3[0]; ob[HWl] = o3[1]; ob[2 * HWl] = o3[2];
          }
          if (c.rgb8_out) {
            uint8_t* o8 = c.rgb8_out + ((long)b * HWl + (unsigned)(oy * Wo + px)) * 3;
            o8[0] = (uint8_t)to_u8(o3[0]); o8[1] = (uint8_t)to_u8(o3[1]); o8[2] = (uint8_t)to_u8(o3[2]);
          }
        }
        tlast = MAUA_NOW();
        MAUA_TICK(dsum[2] += tlast - tc)
      }
      MAUA_TICK(steps_total += nsteps)
    }
  }
  if constexpr (DBG) {
    if (dbg && blockIdx.x == 37 && (tid == 0 || tid == 256)) {
      long long* d = dbg + (tid ? 8 : 0);
      for (int i = 0; i < 5; i++) d[i] = dsum[i];
      d[7] = steps_total;
    }
  }
#undef MAUA_TICK
#undef MAUA_NOW
}

bool upwalk_fused_supported(int dtype, int Ci, int Cm, int H, int W) {
  return (dtype == MAUA_BF16 || dtype == MAUA_F16) && Ci == 64 && Cm == 32 && H >= 2 && W >= 2 && (long)H * W * 4 * 3 < (1L << 31);
}

int launch_upwalk_fused(hipStream_t stream, const HiresArgs& up, const HiresArgs& c1, int force_segs, int narrow_ok, int dtype) {
  if (up.B == 0) return MAUA_OK;
  MAUA_REQUIRE(upwalk_fused_supported(dtype, up.Ci, up.Co, up.H, up.W) && c1.Ci == up.Co && c1.Co == up.Co &&
                   c1.H == 2 * up.H && c1.W == 2 * up.W && up.up == 2 && c1.up == 1,
               "upwalk_fused: unsupported shapes");
  MAUA_REQUIRE((long)up.H * up.W * up.Ci * 2 < (1L << 31), "upwalk_fused: a sample must stay below 2 GiB");
  MAUA_REQUIRE(c1.rgb_out && c1.rgb_wmod && c1.rgb_bias, "upwalk_fused: needs the block's toRGB (the features are not stored)");
  MAUA_REQUIRE(!c1.rgb_skip_f32 || c1.rgb8_out, "upwalk_fused: no output");
  WalkFusedArgs A;
  A.up = up;
  A.c1 = c1;
  for (HiresArgs* q : {&A.up, &A.c1}) {
    MAUA_REQUIRE(q->act == MAUA_ACT_LRELU || q->act == MAUA_ACT_LINEAR, "upwalk_fused: lrelu / linear only");
    if (q->act == MAUA_ACT_LINEAR) q->alpha = 1.f;
    MAUA_REQUIRE(q->alpha >= 0.f && q->alpha <= 1.f && q->gain > 0.f, "upwalk_fused: needs 0 <= alpha <= 1 and gain > 0");
  }
  constexpr int CI = 64, CM = 32;
  const size_t smem = 2 * 9 * 1024 + 6 * 132 * (CM * 2) + 2 * 3 * 3 * (CI / 16) * 64 * 16 + 3 * 256 * 4 + 2 * CM * 4;
  static const bool want_dbg = getenv("MAUA_UW_DBG") != nullptr;
  auto kern = dtype == MAUA_F16 ? upwalk_fused_kernel<CI, CM, false, f16_t>
                                : want_dbg ? upwalk_fused_kernel<CI, CM, true, bf16_t> : upwalk_fused_kernel<CI, CM, false, bf16_t>;
  MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  int dev = 0, cus = 256;
  MAUA_HIP_CHECK(hipGetDevice(&dev));
  MAUA_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int Wo = up.W * 2;
  const int strips = (Wo + 125) / 126;
  // a last strip of <= 32 columns is walked as two half-height sub-items at once (see the kernel): about half the steps
  const int narrow_last = (narrow_ok && strips >= 2 && Wo - (strips - 1) * 126 <= 32) ? 1 : 0;
  // items = (sample, row segment, strip), about nine per CU (a segment costs five extra steps); one persistent workgroup
  // per CU takes a contiguous run of them
  // the number of row segments that minimises the LONGEST workgroup's step count (an item = rows of its segment + 5 steps, a
  // narrow one half the rows + 5)
  int nseg = 1, seg_rows = up.H, n_items = strips * up.B, ipw = (n_items + cus - 1) / cus;
  {
    long best = -1;
    for (int cand = 1; cand <= std::max(1, up.H / 32); cand++) {
      if (force_segs > 0 && cand != std::min(force_segs, std::max(1, up.H / 32))) continue;   // (tests: odd splits)
      const int rows = (up.H + cand - 1) / cand, segs = (up.H + rows - 1) / rows;
      const int items = strips * segs * up.B, per = (items + std::min(items, cus) - 1) / std::min(items, cus);
      const int last_rows = up.H - (segs - 1) * rows;
      long cost = 0;   // the worst contiguous run of `per` items (items ordered sample, segment, strip)
      for (int i0 = 0; i0 < items; i0 += per) {
        long c = 0;
        for (int it = i0; it < std::min(i0 + per, items); it++) {
          const int rem = it % (segs * strips), sg = rem / strips, st = rem % strips;
          const int rr = sg == segs - 1 ? last_rows : rows;
          c += ((narrow_last && st == strips - 1) ? (rr + 1) / 2 : rr) + 5;
        }
        cost = std::max(cost, c);
      }
      if (best < 0 || cost < best) { best = cost; nseg = segs; seg_rows = rows; n_items = items; ipw = per; }
    }
  }
  static long long* dbg = nullptr;
  if (want_dbg && !dbg) { (void)hipMalloc((void**)&dbg, 16 * 8); (void)hipMemset(dbg, 0, 128); }
  hipLaunchKernelGGL(kern, dim3((n_items + ipw - 1) / ipw), dim3(512), smem, stream, A, seg_rows, nseg, strips, n_items,
                     ipw, narrow_last, dbg);
  MAUA_HIP_CHECK(hipGetLastError());
  if (want_dbg) {
    long long hb[16];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(hb, dbg, 128, hipMemcpyDeviceToHost);
    const double n = hb[7] ? (double)hb[7] : 1.0;
    fprintf(stderr, "uwdbg steps %.0f | producer mfma %.0f fir %.0f loopend %.0f barrier %.0f | consumer barrier %.0f noise %.0f epi %.0f mfma %.0f top %.0f\n",
            n, hb[0] / n, hb[1] / n, hb[2] / n, hb[3] / n, hb[8] / n, hb[9] / n, hb[10] / n, hb[11] / n, hb[12] / n);
  }
  return MAUA_OK;
}

bool upwalk_supported(int dtype, int Ci, int Co, int up, int H, int W) {
  return (dtype == MAUA_BF16 || dtype == MAUA_F16) && Ci == 64 && Co == 32 && up == 2 && W % UW_TW == 0 && H >= 2;
}

size_t upwalk_weight_elems(int Co, int Ci) { return (size_t)18 * Co * Ci; }

int launch_upwalk(hipStream_t stream, const HiresArgs& a, int dtype) {
  if (a.B == 0) return MAUA_OK;
  MAUA_REQUIRE(upwalk_supported(dtype, a.Ci, a.Co, a.up, a.H, a.W), "upwalk: unsupported shape");
  MAUA_REQUIRE((long)a.H * 2 * a.W * 2 * std::max(a.Ci, a.Co) * 2 < (1L << 31),
               "upwalk: a sample must stay below 2 GiB (32-bit in-sample offsets)");
  MAUA_REQUIRE(a.act == MAUA_ACT_LRELU || a.act == MAUA_ACT_LINEAR, "upwalk: lrelu / linear only");
  MAUA_REQUIRE(a.y && !a.rgb_out, "upwalk: features out, no toRGB fusion");
  HiresArgs b = a;
  if (a.act == MAUA_ACT_LINEAR) b.alpha = 1.f;
  MAUA_REQUIRE(b.alpha >= 0.f && b.alpha <= 1.f && b.gain > 0.f, "upwalk: needs 0 <= alpha <= 1 and gain > 0");
  constexpr int CI = 64, CO = 32;
  const size_t smem = 2 * (UW_TW + 2) * (CI * 2 + 16) + 4 * (2 * UW_TW) * (CO * 2) + 2 * 3 * (CI / 16) * 64 * 16 + CO * 4;
  auto kern = dtype == MAUA_F16 ? upwalk_kernel<CI, CO, f16_t> : upwalk_kernel<CI, CO, bf16_t>;
  MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  // two workgroups per CU over the whole batch: split every 64-position strip into row segments (each segment pays two
  // extra steps for the rows above / below it)
  const int strips = a.W / UW_TW;
  int nseg = std::max(1, (512 + strips * a.B - 1) / (strips * a.B));
  nseg = std::min(nseg, std::max(1, a.H / 16));
  const int seg_rows = (a.H + nseg - 1) / nseg;
  nseg = (a.H + seg_rows - 1) / seg_rows;
  MAUA_REQUIRE(a.B <= 65535 && nseg <= 65535, "upwalk: grid too large");
  hipLaunchKernelGGL(kern, dim3(strips, nseg, a.B), dim3(256), smem, stream, b, seg_rows);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// ---- weights: f32 [Co][Ci][3][3] -> bf16 [i 3][pb 2][kx 3][Co][Ci] = Kh[i][2 kx + 1 - pb]
template <typename F>
__global__ __launch_bounds__(256) void prep_upwalk_weights_kernel(const float* __restrict__ w, uint16_t* __restrict__ wt,
                                                                  int Co, int Ci, int flip) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)Co * Ci) return;
  float wv[9];
  for (int t = 0; t < 9; t++) wv[t] = w[idx * 9 + t];
  const float g4[4] = {0.25f, 0.75f, 0.75f, 0.25f};
  const long plane = (long)Co * Ci;
  for (int i = 0; i < 3; i++) {
    float kh[6];
    for (int v = 0; v < 6; v++) {
      float s = 0.f;
      for (int j = 0; j < 3; j++) {
        const int fv = v - j;
        if (fv < 0 || fv > 3) continue;
        // A = flip(W) in-tree (no flip before the transposed conv), A = W under nv_compat (as prep_weights_kernel)
        const float aij = flip ? wv[i * 3 + j] : wv[(2 - i) * 3 + (2 - j)];
        s += aij * g4[fv];
      }
      kh[v] = s;
    }
    for (int pb = 0; pb < 2; pb++)
      for (int kx = 0; kx < 3; kx++) wt[(((long)i * 2 + pb) * 3 + kx) * plane + idx] = (uint16_t)(Fmt16<F>::pack2(kh[2 * kx + 1 - pb], 0.f) & 0xffffu);
  }
}

int launch_prep_upwalk_weights(hipStream_t stream, const float* w, void* wt, int Co, int Ci, int flip, int dtype) {
  const long n = (long)Co * Ci;
  if (dtype == MAUA_F16)
    hipLaunchKernelGGL(prep_upwalk_weights_kernel<f16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, (uint16_t*)wt, Co, Ci, flip);
  else
    hipLaunchKernelGGL(prep_upwalk_weights_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, (uint16_t*)wt, Co, Ci, flip);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace maua
