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

template <int CI, int CO>
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
  const bf16_t* xb = reinterpret_cast<const bf16_t*>(a.x) + (long)b * a.H * a.W * CI;
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
    const bf16_t* wbase = reinterpret_cast<const bf16_t*>(a.w);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int kx = 0; kx < 3; kx++)
#pragma unroll
        for (int cs = 0; cs < KS; cs++) {
          const bf16_t* src = wbase + ((((long)i * 2 + pb) * 3 + kx) * CO + r) * CI + cs * 16 + 8 * h;
          const u32x4 v = *reinterpret_cast<const u32x4*>(src);
          u32x4 o;
#pragma unroll
          for (int k = 0; k < 4; k++)
            o[k] = pack2bf(bf2f((bf16_t)(v[k] & 0xffff)) * sv[cs][2 * k], bf2f((bf16_t)(v[k] >> 16)) * sv[cs][2 * k + 1]);
          if (i == 1) { if (wave < 2) wl[((pb * 3 + kx) * KS + cs) * 64 + lane] = o; }
          else wf[((i >> 1) * 3 + kx) * KS + cs] = o;
        }
  }
  if (tid < CO) bias_s[tid] = (a.bias ? a.bias[tid] : 0.f) * a.gain;
  const float nz_scale = a.noise_strength * a.gain;
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
        ecur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[(1 * 3 + kx) * KS + cs]), av, ecur, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wo, av, o, 0, 0, 0);
        enext = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[(0 * 3 + kx) * KS + cs]), av, enext, 0, 0, 0);
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
      *reinterpret_cast<uint2*>(et + ((qd ^ esw) * 16)) = make_uint2(pack2bf(v0[0], v0[1]), pack2bf(v0[2], v0[3]));
      *reinterpret_cast<uint2*>(et + OPX * ES + ((qd ^ esw) * 16)) = make_uint2(pack2bf(v1[0], v1[1]), pack2bf(v1[2], v1[3]));
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
//   One workgroup of 8 waves per CU (the producers need 256 registers); wave w and w + 4 share a SIMD.
struct WalkFusedArgs {
  HiresArgs up, c1;
};

template <int CI, int CM>
__global__ __launch_bounds__(512, 1) void upwalk_fused_kernel(WalkFusedArgs A, int seg_rows, int abl, long long* dbg) {
  constexpr int KS = CI / 16, KS1 = CM / 16;
  constexpr int PIECES = CI * 2 / 16;
  constexpr int RSH = CI * 2 + 16;
  constexpr int PITCH = 126, XPX = 67, RPX = 132;
  constexpr int XROWB = XPX * RSH;          // bytes per staged input row
  constexpr int RROW = RPX * CM * 2;        // bytes per ring row (64 bytes per pixel, 16-byte pieces XOR-swizzled)
  constexpr int HREGS = (XPX * PIECES + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xrow = smem;                                            // [2][XROWB]
  char* ring = smem + 2 * XROWB;                                // [3 slots][2 rows][RROW]
  u32x4* wl = reinterpret_cast<u32x4*>(ring + 6 * RROW);        // [pb 2][chain i = 0, 1][kx 3][KS][64 lanes]
  float* bias0_s = reinterpret_cast<float*>(wl + 2 * 2 * 3 * KS * 64);
  float* bias1_s = bias0_s + CM;

  const HiresArgs& a = A.up;
  const HiresArgs& c = A.c1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const bool producer = wave < 4;
  const int widx = wave & 3;
  const int b = blockIdx.z;
  const int X0 = blockIdx.x * PITCH, J0 = blockIdx.x * (PITCH / 2);
  const int r0 = blockIdx.y * seg_rows, r1 = min(r0 + seg_rows, a.H);
  const int nsteps = r1 - r0 + 5;
  const int Ho = a.H * 2, Wo = a.W * 2;
  if (tid < CM) bias0_s[tid] = (a.bias ? a.bias[tid] : 0.f) * a.gain;
  else if (tid < 2 * CM) bias1_s[tid - CM] = (c.bias ? c.bias[tid - CM] : 0.f) * c.gain;

  if (producer) {
    // ============================================================ conv0: half-folded walk (see upwalk_kernel)
    const int pb = widx & 1, t = (widx >> 1) * 32 + r;
    // (E[rho] += U_2 keeps its 12 fragments in registers; the O chain (i = 1) and the E[rho + 1] = U_0 chain read theirs
    //  from LDS next to the x fragments: 36 + 6 accumulators + a prefetch ring do not fit 256 registers)
    u32x4 wf[3 * KS];
    {
      const float dco = (a.d ? a.d[(long)b * CM + r] : 1.f) * a.gain;
      const float* sb = a.s + (long)b * CI;
      float sv[KS][8];
#pragma unroll
      for (int cs = 0; cs < KS; cs++)
#pragma unroll
        for (int e = 0; e < 8; e++) sv[cs][e] = sb[cs * 16 + 8 * h + e] * dco;
      const bf16_t* wbase = reinterpret_cast<const bf16_t*>(a.w);
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int kx = 0; kx < 3; kx++)
#pragma unroll
          for (int cs = 0; cs < KS; cs++) {
            const bf16_t* src = wbase + ((((long)i * 2 + pb) * 3 + kx) * CM + r) * CI + cs * 16 + 8 * h;
            const u32x4 v = *reinterpret_cast<const u32x4*>(src);
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; k++)
              o[k] = pack2bf(bf2f((bf16_t)(v[k] & 0xffff)) * sv[cs][2 * k], bf2f((bf16_t)(v[k] >> 16)) * sv[cs][2 * k + 1]);
            if (i < 2) { if (widx < 2) wl[(((pb * 2 + i) * 3 + kx) * KS + cs) * 64 + lane] = o; }
            else wf[kx * KS + cs] = o;
          }
    }
    const float nz_scale = a.noise_strength * a.gain;
    const float cl = a.clamp >= 0.f ? a.clamp : 3.0e38f;
    const float* nb = a.noise ? a.noise + (long)b * a.noise_bstride : nullptr;
    const int p = 2 * t + (1 - pb);              // ring pixel of this lane; image pixel X0 - 1 + p
    const int px = X0 - 1 + p;
    const bool px_ok = px >= 0 && px < Wo;
    const int pxc = min(max(px, 0), Wo - 1);
    const int ring_off = p * (CM * 2) + h * 8, esw = (p >> 2) & 3;
    f32x16 ecur, p0, p1, q0;
#pragma unroll
    for (int e = 0; e < 16; e++) { ecur[e] = 0.f; p0[e] = 0.f; p1[e] = 0.f; q0[e] = 0.f; }
    const float g0 = 0.25f, g1 = 0.75f;
    float nz_cur = 0.f;
    long long dsum[4] = {0, 0, 0, 0}, tlast = 0;
#pragma unroll 1
    for (int k = 0; k < nsteps; k++) {
      const int rho = r0 - 2 + k;
      const long long tz = __builtin_amdgcn_s_memtime();
      if (!(abl & 32)) __syncthreads();
      const long long ta = __builtin_amdgcn_s_memtime();
      if (k) { dsum[2] += tz - tlast; }
      dsum[3] += ta - tz;
      if (k == nsteps - 1) continue;   // (the consumers' last step; the loop ends right after)
      float nz_next = 0.f;
      if (nb && !(abl & 128)) nz_next = nb[(unsigned)((2 * min(max(rho, 0), a.H - 1) + h) * Wo + pxc)];
      f32x16 o, enext;
#pragma unroll
      for (int e = 0; e < 16; e++) { o[e] = 0.f; enext[e] = 0.f; }
      const char* abase = xrow + (k & 1) * XROWB + ((1 - pb) + t) * RSH + h * 16;
      const u32x4* wl0 = wl + (pb * 2 + 0) * 3 * KS * 64 + lane;
      const u32x4* wl1 = wl + (pb * 2 + 1) * 3 * KS * 64 + lane;
      // 12 (kx, k-step) iterations of 3 MFMAs; the LDS operands of iteration it + 2 are requested before the MFMAs of
      // iteration it (three register slots), pinned with scheduling barriers: left alone the compiler reads each
      // fragment right before its use and the wave sits out every LDS latency
      u32x4 pa[3], pw0[3], pw1[3];
#define MAUA_UWF_PLOAD(IT)                                                                          \
  {                                                                                                 \
    pa[(IT) % 3] = *reinterpret_cast<const u32x4*>(abase + ((IT) / KS) * RSH + ((IT) % KS) * 32);   \
    pw0[(IT) % 3] = wl0[(IT) * 64];                                                                 \
    pw1[(IT) % 3] = wl1[(IT) * 64];                                                                 \
  }
      if (!(abl & 1)) {
      MAUA_UWF_PLOAD(0)
      MAUA_UWF_PLOAD(1)
#pragma unroll
      for (int it = 0; it < 3 * KS; it++) {
        if (it + 2 < 3 * KS) MAUA_UWF_PLOAD(it + 2)
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 av = __builtin_bit_cast(bf16x8, pa[it % 3]);
        ecur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[it]), av, ecur, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pw1[it % 3]), av, o, 0, 0, 0);
        enext = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pw0[it % 3]), av, enext, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      }
#undef MAUA_UWF_PLOAD
      const long long tb = __builtin_amdgcn_s_memtime();
      dsum[0] += tb - ta;
      const float nz_other = __shfl_xor(nz_cur, 32);
      const float nz0 = (h ? nz_other : nz_cur) * nz_scale, nz1 = (h ? nz_cur : nz_other) * nz_scale;
      nz_cur = nz_next;
      // rows 2 (rho - 1), 2 (rho - 1) + 1 of conv0's output; outside the image (and in the pixels left / right of
      // it) conv1 sees zeros
      const bool ok = px_ok && rho - 1 >= 0 && rho - 1 < a.H;
      char* et = ring + (k % 3) * (2 * RROW) + ring_off;
      if (!(abl & 2))
#pragma unroll
      for (int qd = 0; qd < 4; qd++) {
        const float4 b4 = *reinterpret_cast<const float4*>(bias0_s + 8 * qd + 4 * h);
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
          v0[kk] = ok ? __builtin_amdgcn_fmed3f(y0, -cl, cl) : 0.f;
          v1[kk] = ok ? __builtin_amdgcn_fmed3f(y1, -cl, cl) : 0.f;
        }
        *reinterpret_cast<uint2*>(et + ((qd ^ esw) * 16)) = make_uint2(pack2bf(v0[0], v0[1]), pack2bf(v0[2], v0[3]));
        *reinterpret_cast<uint2*>(et + RROW + ((qd ^ esw) * 16)) = make_uint2(pack2bf(v1[0], v1[1]), pack2bf(v1[2], v1[3]));
      }
      ecur = enext;
      tlast = __builtin_amdgcn_s_memtime();
      dsum[1] += tlast - tb;
    }
    if (dbg && blockIdx.x == 1 && blockIdx.y == 1 && blockIdx.z == 1 && tid == 0)
      for (int i = 0; i < 4; i++) dbg[i] = dsum[i];
  } else {
    // ============================================================ conv1 + toRGB + skip on the ring rows
    const int tid2 = tid - 256;
    const bf16_t* xb = reinterpret_cast<const bf16_t*>(a.x) + (long)b * a.H * a.W * CI;
    u32x4 hreg[HREGS];
#define MAUA_UWF_LOAD_ROW(RHO)                                                                          \
  {                                                                                                     \
    const int gy_ = (RHO);                                                                              \
    _Pragma("unroll") for (int i = 0; i < HREGS; i++) {                                                \
      const int e = tid2 + i * 256, q = e / PIECES, pc = e - q * PIECES;                                \
      const int gx = J0 - 2 + q;                                                                        \
      hreg[i] = u32x4{0u, 0u, 0u, 0u};                                                                  \
      if (e < XPX * PIECES && gy_ >= 0 && gy_ < a.H && gx >= 0 && gx < a.W)                             \
        hreg[i] = *reinterpret_cast<const u32x4*>(xb + (unsigned)((gy_ * a.W + gx) * CI + pc * 8));     \
    }                                                                                                   \
  }
#define MAUA_UWF_STORE_ROW(BUF)                                                                         \
  {                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < HREGS; i++) {                                                \
      const int e = tid2 + i * 256, q = e / PIECES, pc = e - q * PIECES;                                \
      if (e < XPX * PIECES) *reinterpret_cast<u32x4*>(xrow + (BUF) * XROWB + q * RSH + pc * 16) = hreg[i]; \
    }                                                                                                   \
  }
    MAUA_UWF_LOAD_ROW(r0 - 2)
    MAUA_UWF_STORE_ROW(0)
    MAUA_UWF_LOAD_ROW(r0 - 1)
    // ---- conv1's A fragments: W1[tap][co][ci] * s1[ci] * d1[co] * gain, in registers
    u32x4 w1[9 * KS1];
    {
      const float dco = (c.d ? c.d[(long)b * CM + r] : 1.f) * c.gain;
      const float* sb = c.s + (long)b * CM;
      float sv[KS1][8];
#pragma unroll
      for (int cs = 0; cs < KS1; cs++)
#pragma unroll
        for (int e = 0; e < 8; e++) sv[cs][e] = sb[cs * 16 + 8 * h + e] * dco;
      const bf16_t* wbase = reinterpret_cast<const bf16_t*>(c.w);
#pragma unroll
      for (int tap = 0; tap < 9; tap++)
#pragma unroll
        for (int cs = 0; cs < KS1; cs++) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(wbase + ((long)tap * CM + r) * CM + cs * 16 + 8 * h);
          u32x4 o;
#pragma unroll
          for (int k = 0; k < 4; k++)
            o[k] = pack2bf(bf2f((bf16_t)(v[k] & 0xffff)) * sv[cs][2 * k], bf2f((bf16_t)(v[k] >> 16)) * sv[cs][2 * k + 1]);
          w1[tap * KS1 + cs] = o;
        }
    }
    // ---- toRGB A fragments: rows 0..2 = bf16(hi) of the pre-modulated RGB weights, rows 8..10 the bf16 remainder, rows
    // 4..6 / 12..14 repeat them (the h == 1 lanes then hold the same sums); K in ACCUMULATOR order: element e of k-step ks
    // in lane half h is channel 16 ks + 8 (e >> 2) + 4 h + (e & 3), so conv1's activated outputs are the B operand as
    // they sit in the registers
    u32x4 rf[KS1];
    {
      const int c_rgb = (r < 16 && (r & 3) < 3) ? (r & 3) : -1;
#pragma unroll
      for (int ks = 0; ks < KS1; ks++) {
        u32x4 o = u32x4{0u, 0u, 0u, 0u};
        if (c_rgb >= 0) {
          const float* src = c.rgb_wmod + ((long)b * 3 + c_rgb) * CM + ks * 16 + 4 * h;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            float w0 = src[8 * (k >> 1) + 2 * (k & 1)], w1v = src[8 * (k >> 1) + 2 * (k & 1) + 1];
            const float h0 = bf2f(f2bf(w0)), h1 = bf2f(f2bf(w1v));
            if (r >= 8) { w0 -= h0; w1v -= h1; }
            o[k] = pack2bf(w0, w1v);
          }
        }
        rf[ks] = o;
      }
    }
    const float rgb_b0 = c.rgb_bias[0], rgb_b1 = c.rgb_bias[1], rgb_b2 = c.rgb_bias[2];
    const float nz_scale = c.noise_strength * c.gain;
    const float cl = c.clamp >= 0.f ? c.clamp : 3.0e38f;
    const float* nb = c.noise ? c.noise + (long)b * c.noise_bstride : nullptr;
    const int pcol = widx * 32 + r;                     // column inside the strip; ring pixel of tap dx: pcol + dx
    const int px = X0 + pcol;
    const bool px_ok = pcol < PITCH && px < Wo;
    const int pxc = min(px, Wo - 1);
    int foff[3][KS1];                                  // byte offsets of the B fragments inside a ring row
#pragma unroll
    for (int dx = 0; dx < 3; dx++)
#pragma unroll
      for (int ks = 0; ks < KS1; ks++) {
        const int pp = pcol + dx;
        foff[dx][ks] = pp * (CM * 2) + (((2 * ks + h) ^ ((pp >> 2) & 3)) * 16);
      }
    // skip image: the 2x2 window of upsample2d's branch-free form (modconv_hires.hip); column part is per lane
    const int Hp = a.H, Wp = a.W;
    const float* pvb = c.rgb_prev ? c.rgb_prev + (long)b * 3 * Hp * Wp : nullptr;
    const int ix0 = (pxc - 1) >> 1;
    const bool xo = pxc & 1;
    const unsigned HWl = (unsigned)Ho * (unsigned)Wo;
    float nz_next = 0.f;
    f32x16 accA, accB;
#pragma unroll
    for (int e = 0; e < 16; e++) { accA[e] = 0.f; accB[e] = 0.f; }
    float pv[12], fc[4];
#pragma unroll
    for (int j = 0; j < 12; j++) pv[j] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) fc[j] = 0.f;
    // The consumers run half a step out of phase with the producers on the same SIMD: a step starts with the EPILOGUE of
    // the rows multiplied in the previous step (VALU, while the producers multiply) and ends with this step's MFMAs
    // (while the producers run their FIR / epilogue).
    float pvn[12], fcn[4], nz_use = 0.f;
    long long csum[5] = {0, 0, 0, 0, 0}, tlast = 0, ta = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) pvn[j] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) fcn[j] = 0.f;
#pragma unroll 1
    for (int k = 0; k <= nsteps; k++) {
      const int rho = r0 - 2 + k;
      // every global load of a step is issued right here, a whole step before its data are needed: the staged input row
      // goes to LDS at the next step's start, noise + skip window (for the rows multiplied in THIS step) are used by the
      // next step's epilogue
#pragma unroll
      for (int j = 0; j < 12; j++) pv[j] = pvn[j];
#pragma unroll
      for (int j = 0; j < 4; j++) fc[j] = fcn[j];
      nz_use = nz_next;
      const long long tz = __builtin_amdgcn_s_memtime();
      if (k) csum[4] += tz - tlast;
      if (k < nsteps) {
        if (!(abl & 32)) __syncthreads();
        ta = __builtin_amdgcn_s_memtime();
        csum[0] += ta - tz;
        if (!(abl & 16)) {
        MAUA_UWF_STORE_ROW((k + 1) & 1)
        MAUA_UWF_LOAD_ROW(rho + 2)
        }
        const int arow = 2 * (rho - 2);
        const int oyc = min(max(arow - 1 + h, 0), Ho - 1);
        if (nb && !(abl & 64)) nz_next = nb[(unsigned)(oyc * Wo + pxc)];
        if (pvb && !(abl & 64)) {
          const int iy0 = (oyc - 1) >> 1;
          const bool yo = oyc & 1;
#pragma unroll
          for (int dy = 0; dy < 2; dy++) {
            const bool uh = (dy == 0) == yo;
            const bool oky = iy0 + dy >= 0 && iy0 + dy < Hp;
            const int gy = min(max(iy0 + dy, 0), Hp - 1);
#pragma unroll
            for (int dx = 0; dx < 2; dx++) {
              const bool vh = (dx == 0) == xo;
              const bool okk = oky && ix0 + dx >= 0 && ix0 + dx < Wp;
              const int gx = min(max(ix0 + dx, 0), Wp - 1);
              fcn[dy * 2 + dx] = !okk ? 0.f : uh ? (vh ? c.fir[5] : c.fir[4]) : (vh ? c.fir[1] : c.fir[0]);
#pragma unroll
              for (int ch = 0; ch < 3; ch++) pvn[(dy * 2 + dx) * 3 + ch] = pvb[(unsigned)((ch * Hp + gy) * Wp + gx)];
            }
          }
        }
      }
      const long long tb = __builtin_amdgcn_s_memtime();
      csum[1] += tb - ta;
      if (k > 0 && !(abl & 8)) {
        // ---- epilogue of step k - 1 (output rows arow - 1: lanes h == 0, arow: h == 1) in registers -> toRGB MFMA
        const int arow = 2 * (rho - 3);
        const int oy = arow - 1 + h;
        const float nz_other = __shfl_xor(nz_use, 32);   // (requested in the previous step)
        const float nzA = (h ? nz_other : nz_use) * nz_scale, nzB = (h ? nz_use : nz_other) * nz_scale;
        u32x4 fa[KS1], fb[KS1];
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
          const float4 b4 = *reinterpret_cast<const float4*>(bias1_s + 8 * qd + 4 * h);
          const float bq[4] = {b4.x, b4.y, b4.z, b4.w};
          float va[4], vb[4];
#pragma unroll
          for (int kk = 0; kk < 4; kk++) {
            float ya = accA[qd * 4 + kk] + (nzA + bq[kk]);
            float yb = accB[qd * 4 + kk] + (nzB + bq[kk]);
            ya = fmaxf(ya, ya * c.alpha);
            yb = fmaxf(yb, yb * c.alpha);
            va[kk] = __builtin_amdgcn_fmed3f(ya, -cl, cl);
            vb[kk] = __builtin_amdgcn_fmed3f(yb, -cl, cl);
          }
          fa[qd >> 1][(qd & 1) * 2] = pack2bf(va[0], va[1]);
          fa[qd >> 1][(qd & 1) * 2 + 1] = pack2bf(va[2], va[3]);
          fb[qd >> 1][(qd & 1) * 2] = pack2bf(vb[0], vb[1]);
          fb[qd >> 1][(qd & 1) * 2 + 1] = pack2bf(vb[2], vb[3]);
        }
        f32x16 ra, rb;
#pragma unroll
        for (int e = 0; e < 16; e++) { ra[e] = 0.f; rb[e] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS1; ks++) {
          ra = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rf[ks]), __builtin_bit_cast(bf16x8, fa[ks]), ra, 0, 0, 0);
          rb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rf[ks]), __builtin_bit_cast(bf16x8, fb[ks]), rb, 0, 0, 0);
        }
        float o3[3];
        o3[0] = (h ? rb[0] + rb[4] : ra[0] + ra[4]) + rgb_b0;
        o3[1] = (h ? rb[1] + rb[5] : ra[1] + ra[5]) + rgb_b1;
        o3[2] = (h ? rb[2] + rb[6] : ra[2] + ra[6]) + rgb_b2;
        if (c.rgb_clamp >= 0.f) {
#pragma unroll
          for (int ch = 0; ch < 3; ch++) o3[ch] = fminf(fmaxf(o3[ch], -c.rgb_clamp), c.rgb_clamp);
        }
        if (pvb) {
          float u3[3] = {0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 4; j++)
#pragma unroll
            for (int ch = 0; ch < 3; ch++) u3[ch] += pv[j * 3 + ch] * fc[j];
          o3[0] = u3[0] + o3[0]; o3[1] = u3[1] + o3[1]; o3[2] = u3[2] + o3[2];
        }
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
      }
      const long long tc = __builtin_amdgcn_s_memtime();
      csum[2] += tc - tb;
      if (k == nsteps) break;
      // ---- conv1: ring rows arow - 2 .. arow + 1 = slots (k - 2) % 3 and (k - 1) % 3
#pragma unroll
      for (int e = 0; e < 16; e++) { accA[e] = 0.f; accB[e] = 0.f; }
      const char* s2 = ring + ((k + 1) % 3) * (2 * RROW);
      const char* s1 = ring + ((k + 2) % 3) * (2 * RROW);
      // (one ring row = 6 fragments ahead of the MFMAs that use them, pinned like the producers' pipeline)
      u32x4 fr[2][3 * KS1];
#define MAUA_UWF_CLOAD(RR)                                                                             \
  {                                                                                                    \
    const char* rowp = ((RR) < 2 ? s2 : s1) + ((RR) & 1) * RROW;                                       \
    _Pragma("unroll") for (int dx = 0; dx < 3; dx++) _Pragma("unroll") for (int ks = 0; ks < KS1; ks++) \
      fr[(RR) & 1][dx * KS1 + ks] = *reinterpret_cast<const u32x4*>(rowp + foff[dx][ks]);              \
  }
      if (!(abl & 4)) {
      MAUA_UWF_CLOAD(0)
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {        // rr = ring row arow - 2 + rr
        if (rr < 3) MAUA_UWF_CLOAD(rr + 1)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dx = 0; dx < 3; dx++)
#pragma unroll
          for (int ks = 0; ks < KS1; ks++) {
            const bf16x8 bv = __builtin_bit_cast(bf16x8, fr[rr & 1][dx * KS1 + ks]);
            if (rr <= 2)  // tap dy = rr - 1 of output row arow - 1
              accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1[(rr * 3 + dx) * KS1 + ks]), bv, accA, 0, 0, 0);
            if (rr >= 1)  // tap dy = rr - 2 of output row arow
              accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1[((rr - 1) * 3 + dx) * KS1 + ks]), bv, accB, 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      }
#undef MAUA_UWF_CLOAD
      tlast = __builtin_amdgcn_s_memtime();
      csum[3] += tlast - tc;
    }
    if (dbg && blockIdx.x == 1 && blockIdx.y == 1 && blockIdx.z == 1 && tid == 256)
      for (int i = 0; i < 5; i++) dbg[8 + i] = csum[i];
#undef MAUA_UWF_LOAD_ROW
#undef MAUA_UWF_STORE_ROW
  }
}

bool upwalk_fused_supported(int dtype, int Ci, int Cm, int H, int W) {
  return dtype == MAUA_BF16 && Ci == 64 && Cm == 32 && H >= 2 && W >= 2 && (long)H * W * 4 * 3 < (1L << 31);
}

int launch_upwalk_fused(hipStream_t stream, const HiresArgs& up, const HiresArgs& c1) {
  if (up.B == 0) return MAUA_OK;
  MAUA_REQUIRE(upwalk_fused_supported(MAUA_BF16, up.Ci, up.Co, up.H, up.W) && c1.Ci == up.Co && c1.Co == up.Co &&
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
  const size_t smem = 2 * 67 * (CI * 2 + 16) + 6 * 132 * (CM * 2) + 2 * 2 * 3 * (CI / 16) * 64 * 16 + 2 * CM * 4;
  auto kern = upwalk_fused_kernel<CI, CM>;
  MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const int Wo = up.W * 2;
  const int strips = (Wo + 125) / 126;
  // about nine workgroups per CU over the batch (one resident at a time); a segment costs five extra steps
  int nseg = std::max(1, (2304 + strips * up.B - 1) / (strips * up.B));
  nseg = std::min(nseg, std::max(1, up.H / 32));
  const int seg_rows = (up.H + nseg - 1) / nseg;
  nseg = (up.H + seg_rows - 1) / seg_rows;
  MAUA_REQUIRE(up.B <= 65535 && nseg <= 65535, "upwalk_fused: grid too large");
  static const int abl = getenv("MAUA_UW_ABL") ? atoi(getenv("MAUA_UW_ABL")) : 0;
  static long long* dbg = nullptr;
  static const bool want_dbg = getenv("MAUA_UW_DBG") != nullptr;
  if (want_dbg && !dbg) { hipMalloc((void**)&dbg, 16 * 8); hipMemset(dbg, 0, 128); }
  hipLaunchKernelGGL(kern, dim3(strips, nseg, up.B), dim3(512), smem, stream, A, seg_rows, abl, dbg);
  if (want_dbg) {
    long long hbuf[16];
    hipStreamSynchronize(stream);
    hipMemcpy(hbuf, dbg, 128, hipMemcpyDeviceToHost);
    const double n = (up.H + nseg - 1) / nseg + 5;
    fprintf(stderr, "uwdbg steps %.0f | producer mfma %.0f fir %.0f loopend %.0f barrier %.0f | consumer barrier %.0f stage %.0f epi %.0f mfma %.0f top %.0f\n",
            n, hbuf[0] / n, hbuf[1] / n, hbuf[2] / n, hbuf[3] / n, hbuf[8] / n, hbuf[9] / n, hbuf[10] / n, hbuf[11] / n, hbuf[12] / n);
  }
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

bool upwalk_supported(int dtype, int Ci, int Co, int up, int H, int W) {
  return dtype == MAUA_BF16 && Ci == 64 && Co == 32 && up == 2 && W % UW_TW == 0 && H >= 2;
}

size_t upwalk_weight_elems(int Co, int Ci) { return (size_t)18 * Co * Ci; }

int launch_upwalk(hipStream_t stream, const HiresArgs& a) {
  if (a.B == 0) return MAUA_OK;
  MAUA_REQUIRE(upwalk_supported(MAUA_BF16, a.Ci, a.Co, a.up, a.H, a.W), "upwalk: unsupported shape");
  MAUA_REQUIRE((long)a.H * 2 * a.W * 2 * std::max(a.Ci, a.Co) * 2 < (1L << 31),
               "upwalk: a sample must stay below 2 GiB (32-bit in-sample offsets)");
  MAUA_REQUIRE(a.act == MAUA_ACT_LRELU || a.act == MAUA_ACT_LINEAR, "upwalk: lrelu / linear only");
  MAUA_REQUIRE(a.y && !a.rgb_out, "upwalk: features out, no toRGB fusion");
  HiresArgs b = a;
  if (a.act == MAUA_ACT_LINEAR) b.alpha = 1.f;
  MAUA_REQUIRE(b.alpha >= 0.f && b.alpha <= 1.f && b.gain > 0.f, "upwalk: needs 0 <= alpha <= 1 and gain > 0");
  constexpr int CI = 64, CO = 32;
  const size_t smem = 2 * (UW_TW + 2) * (CI * 2 + 16) + 4 * (2 * UW_TW) * (CO * 2) + 2 * 3 * (CI / 16) * 64 * 16 + CO * 4;
  auto kern = upwalk_kernel<CI, CO>;
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
__global__ __launch_bounds__(256) void prep_upwalk_weights_kernel(const float* __restrict__ w, bf16_t* __restrict__ wt,
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
      for (int kx = 0; kx < 3; kx++) wt[(((long)i * 2 + pb) * 3 + kx) * plane + idx] = f2bf(kh[2 * kx + 1 - pb]);
  }
}

int launch_prep_upwalk_weights(hipStream_t stream, const float* w, void* wt, int Co, int Ci, int flip) {
  const long n = (long)Co * Ci;
  hipLaunchKernelGGL(prep_upwalk_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w,
                     (bf16_t*)wt, Co, Ci, flip);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace maua
