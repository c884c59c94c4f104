// High-resolution (few-channel, HBM-bound) modulated 3x3 convolution for gfx950: weights stationary in VGPRs.
//
// Same math as modconv.hip (reference ops.py:146-186,189-233,87-114,65-84 fused), specialised for the layers where
// the activations, not the MACs, are the cost: 512^2 / 1024^2 with 32..64 channels.  At these shapes the whole
// weight set of a wave's output slice is only 18..36 MFMA B-fragments, so it is loaded ONCE per workgroup into
// registers — pre-multiplied by the sample's styles, which is the reference's own "w = weight * styles" — and the
// workgroup then walks over spatial tiles of its sample: the only per-tile traffic is the input halo (prefetched
// into registers while the previous tile is multiplied), the noise row and the output.  LDS holds just the halo
// and the [pixel][channel] epilogue tile; B operands never touch LDS.
//   <32,32,1>  b1024.conv1   tile 8x32 px, wave = 2 image rows x 32 channels
//   <64,64,1>  b512.conv1    tile 4x32 px, wave = 2 image rows x one 32-channel half
//   <64,32,2>  b1024.conv0   tile 4x32 input px, wave = one output parity (phase kernels of modconv.hip) x 32 ch
// The conv1 variants optionally fuse the block's toRGB (stylegan2.py:268-272) + FIR-upsampled skip + add
// (stylegan2.py:372-378) on the epilogue tile while it is still in LDS: the 1x1 conv then never re-reads x from HBM.
#include "common.h"
#include "internal.h"

namespace maua {

template <int CI, int CO, int UP, typename F>
__global__ __launch_bounds__(256, CI == 64 ? 2 : 3) void modconv_hires_kernel(HiresArgs a) {
  constexpr int P = UP * UP;                      // output parities
  constexpr int NV = CO * P;                      // virtual output channels
  constexpr int KS = CI / 16;                     // MFMA k-steps per tap
  constexpr int NKS = 9 * KS;                     // B fragments per wave
  constexpr int PIECES = CI * 2 / 16;             // 16-byte pieces per input pixel
  constexpr int RSH = CI * 2 + 16;                // halo row stride (bytes), +16 keeps ds_read_b128 conflict-free
  constexpr int TH = CI == 32 ? 8 : 4, TW = 32;   // tile (input grid)
  constexpr int BM = TH * TW;
  constexpr int HW2 = TW + 2, HALO_PX = (TH + 2) * HW2;
  constexpr int HREGS = (HALO_PX * PIECES + 255) / 256;
  constexpr int ES = NV * 2 + 16;                 // epilogue tile row stride
  constexpr int PPP = NV * 2 / 16;                // 16-byte pieces per pixel of the epilogue tile
  constexpr int MSW = (UP == 2) ? 4 : 2;          // image rows (M sub-tiles of 32 px) per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* epi = smem + HALO_PX * RSH;
  // CI == 64: 36 weight fragments leave no room for the epilogue constants at 2 waves per SIMD -> they live in LDS
  constexpr bool LEAN = true;
  float* bias_s = reinterpret_cast<float*>(epi + BM * ES);            // [CO] bias * gain
  u32x4* rf_s = reinterpret_cast<u32x4*>(epi + BM * ES + CO * 4);      // [CO/16][64 lanes] toRGB B fragments
  // the tile's window of the previous block's image (the toRGB skip): [3][PH][PW] f32, zero outside the image
  constexpr int PH = TH / 2 + 2, PW = TW / 2 + 2, PREGS = (3 * PH * PW + 255) / 256;
  float* prev_s = reinterpret_cast<float*>(epi + BM * ES + CO * 4 + (CO / 16) * 64 * 16);
  float* noise_s = prev_s + PREGS * 256;  // [4 waves][MSW rows][32]: the noise of each wave's output pixels

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  const uint16_t* xb = reinterpret_cast<const uint16_t*>(a.x) + (long)b * a.H * a.W * CI;
  const float* sb = a.s + (long)b * CI;

  // ---- role of this wave
  int phase = 0, nsub = 0, ms0 = 0;
  if constexpr (UP == 2) { phase = wave; }
  else if constexpr (CO == 64) { nsub = wave & 1; ms0 = (wave >> 1) * 2; }
  else { ms0 = wave * 2; }

  // ---- B fragments: W[tap][phase][co][ci] * s[b][ci] -> bf16, resident for the whole kernel
  u32x4 wf[NKS];
  {
    // CI == 64: the lane's output channel (MFMA A row r) also gets its demodulation coefficient and the layer gain
    // here, act(d*acc + nz + b)*g == act(acc' + (nz + b)*g) with w' = w*s*d*g (lrelu is positively homogeneous);
    // CI == 32 keeps d*g in registers (measured faster there)
    const float dco = LEAN ? (a.d ? a.d[(long)b * CO + nsub * 32 + r] : 1.f) * a.gain : 1.f;
    float sv[KS][8];
#pragma unroll
    for (int cs = 0; cs < KS; cs++)
#pragma unroll
      for (int e = 0; e < 8; e++) sv[cs][e] = sb[cs * 16 + 8 * h + e] * dco;
    const uint16_t* wbase = reinterpret_cast<const uint16_t*>(a.w);
#pragma unroll
    for (int tap = 0; tap < 9; tap++)
#pragma unroll
      for (int cs = 0; cs < KS; cs++) {
        const uint16_t* src = wbase + (((long)tap * P + phase) * CO + nsub * 32 + r) * CI + cs * 16 + 8 * h;
        u32x4 v = *reinterpret_cast<const u32x4*>(src);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; k++)
          o[k] = Fmt16<F>::pack2(Fmt16<F>::lo(v[k]) * sv[cs][2 * k], Fmt16<F>::hi(v[k]) * sv[cs][2 * k + 1]);
        wf[tap * KS + cs] = o;
      }
  }
  // per-lane epilogue constants: bias * gain for 4 quads of 4 consecutive channels
  float bv[LEAN ? 1 : 16], dv[LEAN ? 1 : 16];
  if constexpr (LEAN) {
    if (tid < CO) bias_s[tid] = (a.bias ? a.bias[tid] : 0.f) * a.gain;
  } else {
#pragma unroll
  for (int qd = 0; qd < 4; qd++) {
    const int co = nsub * 32 + 8 * qd + 4 * h;
    const float4 b4 = a.bias ? *reinterpret_cast<const float4*>(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    bv[qd * 4] = b4.x * a.gain; bv[qd * 4 + 1] = b4.y * a.gain; bv[qd * 4 + 2] = b4.z * a.gain; bv[qd * 4 + 3] = b4.w * a.gain;
    const float4 d4 = a.d ? *reinterpret_cast<const float4*>(a.d + (long)b * CO + co) : make_float4(1.f, 1.f, 1.f, 1.f);
    dv[qd * 4] = d4.x * a.gain; dv[qd * 4 + 1] = d4.y * a.gain; dv[qd * 4 + 2] = d4.z * a.gain; dv[qd * 4 + 3] = d4.w * a.gain;
  }
  }
  const float nz_scale = a.noise_strength * a.gain * (a.noise_scale ? a.noise_scale[b] : 1.f);
  const float cl = a.clamp >= 0.f ? a.clamp : 3.0e38f;
  // fused toRGB as one more MFMA: B rows 0..2 = bf16(hi) part of the pre-modulated RGB weights, rows 8..10 = the
  // bf16 remainder (w = hi + lo to ~2^-17), everything else zero; rgb[c] = acc[row c] + acc[row 8+c], both of
  // which land in the h == 0 lane of the pixel.
  u32x4 rf[LEAN ? 1 : CO / 16];
  if (a.rgb_out) {
    // (rows 4..6 / 12..14 repeat them: the h == 1 lanes then hold the same sums and can finish a second image row)
    const int c_rgb = (r < 16 && (r & 3) < 3) ? (r & 3) : -1;
#pragma unroll
    for (int ks = 0; ks < CO / 16; ks++) {
      u32x4 o = u32x4{0u, 0u, 0u, 0u};
      if (c_rgb >= 0) {
        const float* src = a.rgb_wmod + ((long)b * 3 + c_rgb) * CO + ks * 16 + 8 * h;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          float w0 = src[2 * k], w1 = src[2 * k + 1];
          const float h0 = Fmt16<F>::round(w0), h1 = Fmt16<F>::round(w1);
          if (r >= 8) { w0 -= h0; w1 -= h1; }
          o[k] = Fmt16<F>::pack2(w0, w1);
        }
      }
      if constexpr (LEAN) { if (wave == 0) rf_s[ks * 64 + lane] = o; }
      else rf[ks] = o;
    }
  }

  float rgb_b[3] = {0.f, 0.f, 0.f};  // read once (scalar): a load inside the tile loop would drain vmcnt every tile
  if (a.rgb_out) { rgb_b[0] = a.rgb_bias[0]; rgb_b[1] = a.rgb_bias[1]; rgb_b[2] = a.rgb_bias[2]; }
  const int tiles_x = a.W / TW, n_tiles = tiles_x * (a.H / TH);
  const int Ho = a.H * UP, Wo = a.W * UP;
  const float* nb = a.noise ? a.noise + (long)b * a.noise_bstride : nullptr;
  char* yb = reinterpret_cast<char*>(a.y) + (long)b * Ho * Wo * CO * 2;

  // halo staging: piece index -> (pixel, 16-byte piece)
  const int q = tid % PIECES, rq = tid / PIECES;
  u32x4 hreg[HREGS];
  const bool with_prev = UP == 1 && a.rgb_out && a.rgb_prev;
  const float* pvb = a.rgb_prev + (long)b * 3 * (a.H >> 1) * (a.W >> 1);
#define MAUA_HIRES_LOAD_HALO(TILE)                                                               \
  {                                                                                               \
    const int tyi_ = (TILE) / tiles_x, txi_ = (TILE) - tyi_ * tiles_x;                            \
    _Pragma("unroll") for (int i = 0; i < HREGS; i++) {                                          \
      const int p = rq + i * (256 / PIECES);                                                      \
      hreg[i] = u32x4{0u, 0u, 0u, 0u};                                                            \
      if (p < HALO_PX) {                                                                          \
        const int py = p / HW2, px = p - py * HW2;                                                \
        const int gy = tyi_ * TH - 1 + py, gx = txi_ * TW - 1 + px;                               \
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)                                           \
          hreg[i] = *reinterpret_cast<const u32x4*>(xb + (unsigned)((gy * a.W + gx) * CI + q * 8));     \
      }                                                                                           \
    }                                                                                             \
  }

  int tile = blockIdx.x;
  if (tile < n_tiles) MAUA_HIRES_LOAD_HALO(tile)
  for (; tile < n_tiles; tile += gridDim.x) {
    const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
    const int ty0 = tyi * TH, tx0 = txi * TW;
    __syncthreads();  // previous tile: MFMA reads of the halo and read-out of the epilogue tile are done
#pragma unroll
    for (int i = 0; i < HREGS; i++) {
      const int p = rq + i * (256 / PIECES);
      if (p < HALO_PX) *reinterpret_cast<u32x4*>(halo + p * RSH + q * 16) = hreg[i];
    }
    __syncthreads();
    if constexpr (UP == 2) {
      // (same LDS-direct path as below; lane (h, r) fetches the noise of output pixel (2 gy + pa, 2 gx + pb) for
      //  tile rows 2 j + h: two loads cover the wave's four rows)
      if (nb) {
        const int pa = phase >> 1, pb = phase & 1;
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const float* src = nb + (unsigned)(((ty0 + 2 * j + h) * 2 + pa) * Wo + (tx0 + r) * 2 + pb);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
              (__attribute__((address_space(3))) void*)(noise_s + wave * 128 + j * 64), 4, 0, 0);
        }
      }
    }
    if constexpr (UP == 1) {
      // the tile's noise and its window of the previous image travel global -> LDS without passing through
      // registers (LDS-direct loads: lane l of a wave fills dword l of the wave's 256-byte slot); they are waited
      // for (vmcnt) in the epilogue, the whole multiply phase later
      if (nb) {  // lane -> pixel (ms0 + (lane >> 5), lane & 31): exactly the pixels this wave's epilogue covers
        const float* src = nb + (unsigned)((ty0 + ms0 + h) * a.W + tx0 + r);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
            (__attribute__((address_space(3))) void*)(noise_s + wave * (MSW * 32)), 4, 0, 0);
      }
      if (with_prev) {
        const int Hp = a.H >> 1, Wp = a.W >> 1;
#pragma unroll
        for (int i = 0; i < PREGS; i++) {
          int e = tid + i * 256;
          asm volatile("" : "+v"(e));  // re-derive the window coordinates per tile instead of keeping them in registers
          const int c = e / (PH * PW), py = (e - c * PH * PW) / PW;
          const int px = e - c * PH * PW - py * PW;
          // (coordinates clamped into the image: the FIR below zeroes the taps that fall outside)
          const int gy = min(max(tyi * (TH / 2) - 1 + py, 0), Hp - 1), gx = min(max(txi * (TW / 2) - 1 + px, 0), Wp - 1);
          if (e < 3 * PH * PW)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(pvb + (unsigned)((c * Hp + gy) * Wp + gx)),
                (__attribute__((address_space(3))) void*)(prev_s + i * 256 + wave * 64), 4, 0, 0);
        }
      }
    }
    if (tile + (int)gridDim.x < n_tiles) MAUA_HIRES_LOAD_HALO(tile + (int)gridDim.x)  // flies during the MFMAs

    // ---- multiply: PAIR image rows (M sub-tiles) at a time share every B fragment (CI == 64: one row at a time,
    // the second accumulator would not fit next to 36 weight fragments at 2 waves per SIMD)
    constexpr int PAIR = 1;
#pragma unroll
    for (int mp = 0; mp < MSW; mp += PAIR) {
      f32x16 acc[PAIR];
#pragma unroll
      for (int i = 0; i < PAIR; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
      const int ms = ms0 + mp;
      const char* abase = halo + ((ms + 1) * HW2 + (r + 1)) * RSH + h * 16;
#pragma unroll
      for (int tap = 0; tap < 9; tap++) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
        for (int cs = 0; cs < KS; cs++) {
          const char* ap = abase + (dy * HW2 + dx) * RSH + cs * 32;
          const u32x4 wv = wf[tap * KS + cs];
#pragma unroll
          for (int i = 0; i < PAIR; i++) {
            const u32x4 av = *reinterpret_cast<const u32x4*>(ap + i * HW2 * RSH);
            Mma16<F>::step(acc[i], wv, av);
          }
        }
      }
      // ---- epilogue into the LDS tile: lane = pixel (ms + i, r), 16 channels in 4 quads
#pragma unroll
      for (int half = 0; half < PAIR; half++) {
        const int m = (ms + half) * 32 + r;
        const int gy = ty0 + ms + half, gx = tx0 + r;
        const int pa = phase / UP, pb = phase - pa * UP;
        float nz = 0.f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the LDS-direct loads of this tile have landed
        if (nb) nz = noise_s[wave * (MSW * 32) + (ms + half - ms0) * 32 + r] * nz_scale;
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
          float v[4], bq[4];
          if constexpr (LEAN) {
            const float4 b4 = *reinterpret_cast<const float4*>(bias_s + nsub * 32 + 8 * qd + 4 * h);
            bq[0] = b4.x; bq[1] = b4.y; bq[2] = b4.z; bq[3] = b4.w;
          } else {
#pragma unroll
            for (int k = 0; k < 4; k++) bq[k] = bv[qd * 4 + k];
          }
#pragma unroll
          for (int k = 0; k < 4; k++) {
            float t;
            if constexpr (LEAN) t = acc[half][qd * 4 + k] + (nz + bq[k]);  // d * gain is inside the weights
            else t = fmaf(acc[half][qd * 4 + k], dv[qd * 4 + k], nz + bq[k]);
            t = fmaxf(t, t * a.alpha);                 // lrelu for 0 <= alpha <= 1 (alpha = 1 gives linear)
            v[k] = __builtin_amdgcn_fmed3f(t, -cl, cl);  // clamp
          }
          const int nv = phase * CO + nsub * 32 + 8 * qd + 4 * h;
          *reinterpret_cast<uint2*>(epi + m * ES + nv * 2) = make_uint2(Fmt16<F>::pack2(v[0], v[1]), Fmt16<F>::pack2(v[2], v[3]));
        }
      }
    }
    __syncthreads();
    // ---- read-out: full 16-byte NHWC pieces (y == NULL: the last block's conv1 output only feeds the fused toRGB)
    if (a.y) for (int p = tid; p < BM * PPP; p += 256) {
      const int m = p / PPP, pc = p - m * PPP;
      const int gy = ty0 + (m >> 5), gx = tx0 + (m & 31);
      const int nv = pc * 8;
      const int ph = nv / CO, co = nv - ph * CO;
      const int pa = ph / UP, pb = ph - pa * UP;
      const unsigned pix = (unsigned)((gy * UP + pa) * Wo + gx * UP + pb);
      *reinterpret_cast<uint4*>(yb + (pix * CO + co) * 2) = *reinterpret_cast<const uint4*>(epi + m * ES + pc * 16);
    }
    // ---- fused toRGB + upsampled skip (conv1 layers only): [32 px x CO] x [CO x 3(+3)] on the matrix cores, the
    // activated bf16 outputs are the A operand straight from the epilogue tile.  Every lane finishes ONE pixel:
    // with two image rows per wave the h == 0 lanes take the first and the h == 1 lanes the second (the repeated
    // weight rows put the sums into both halves); with one row per wave the h == 1 lanes idle.
    if constexpr (UP == 1) {
      if (a.rgb_out) {
        constexpr int ROWS_W = TH / 4;  // image rows per wave
        float o3[3];
#pragma unroll 1
        for (int rw = 0; rw < ROWS_W; rw++) {
          const int row_m = wave * ROWS_W + rw;
          f32x16 racc;
#pragma unroll
          for (int e = 0; e < 16; e++) racc[e] = 0.f;
#pragma unroll
          for (int ks = 0; ks < CO / 16; ks++) {
            const u32x4 av = *reinterpret_cast<const u32x4*>(epi + (row_m * 32 + r) * ES + (ks * 16 + 8 * h) * 2);
            Mma16<F>::step(racc, rf_s[ks * 64 + lane], av);
          }
          if (rw == 0 || h == rw) { o3[0] = racc[0] + racc[4]; o3[1] = racc[1] + racc[5]; o3[2] = racc[2] + racc[6]; }
        }
        if (ROWS_W == 2 || h == 0) {
          const int row = wave * ROWS_W + (ROWS_W == 2 ? h : 0);
          const int y = ty0 + row, x = tx0 + r;
#pragma unroll
          for (int c = 0; c < 3; c++) {
            o3[c] += rgb_b[c];
            if (a.rgb_clamp >= 0.f) o3[c] = fminf(fmaxf(o3[c], -a.rgb_clamp), a.rgb_clamp);
          }
          const unsigned HWl = (unsigned)(a.H * a.W);
          if (a.rgb_prev) {
            // upsample2d (zero-insert x2, pad (2,1,2,1), 4x4 FIR) in its branch-free 2x2 form: only the taps whose
            // parity hits a real sample are non-zero -> window rows {iy0, iy0+1}, cols {ix0, ix0+1} with
            // iy0 = (y-1)>>1, filter index u = 2*iy - y + 2 (same products, same u-major order as the 16-tap
            // correlation; taps outside the image get a zero coefficient)
            const int py0 = ((row - 1) >> 1) + 1, px0 = ((r - 1) >> 1) + 1;
            const int iy0 = (ty0 >> 1) - 1 + py0, ix0 = (tx0 >> 1) - 1 + px0;
            int par = (row & 1) * 2 + (r & 1);
            asm volatile("" : "+v"(par));  // (keeps the four per-lane coefficients out of the loop-invariant registers)
            const bool yo = par & 2, xo = par & 1;  // odd output coordinate: first tap is fir index 1, second 3
            float u3[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 2; dy++) {
              const bool uh = (dy == 0) == yo;  // fir[u][v] takes 3 distinct values: u, v in {1,2} or {0,3}
              const bool oky = iy0 + dy >= 0 && iy0 + dy < (a.H >> 1);
#pragma unroll
              for (int dx = 0; dx < 2; dx++) {
                const bool vh = (dx == 0) == xo;
                const bool ok = oky && ix0 + dx >= 0 && ix0 + dx < (a.W >> 1);
                const float f = !ok ? 0.f : uh ? (vh ? a.fir[5] : a.fir[4]) : (vh ? a.fir[1] : a.fir[0]);
                const float* ps = prev_s + (py0 + dy) * PW + px0 + dx;
                u3[0] += ps[0] * f;
                u3[1] += ps[PH * PW] * f;
                u3[2] += ps[2 * PH * PW] * f;
              }
            }
            o3[0] = u3[0] + o3[0]; o3[1] = u3[1] + o3[1]; o3[2] = u3[2] + o3[2];
          }
          if (!a.rgb_skip_f32) {
            float* ob = a.rgb_out + (long)b * 3 * HWl + (unsigned)(y * a.W + x);
            ob[0] = o3[0]; ob[HWl] = o3[1]; ob[2 * HWl] = o3[2];
          }
          if (a.rgb8_out) {  // final block: the u8 HWC frame leaves from here (no separate pack pass)
            uint8_t* o8 = a.rgb8_out + ((long)b * HWl + (unsigned)(y * a.W + x)) * 3;
            o8[0] = (uint8_t)to_u8(o3[0]); o8[1] = (uint8_t)to_u8(o3[1]); o8[2] = (uint8_t)to_u8(o3[2]);
          }
        }
      }
    }
  }
#undef MAUA_HIRES_LOAD_HALO
}

template <int CI, int CO, int UP, typename F>
static int launch_hires_variant(hipStream_t stream, const HiresArgs& a) {
  constexpr int TH = CI == 32 ? 8 : 4, TW = 32, BM = TH * TW, NV = CO * UP * UP;
  constexpr int HALO_PX = (TH + 2) * (TW + 2);
  size_t smem = (size_t)HALO_PX * (CI * 2 + 16) + (size_t)BM * (NV * 2 + 16) + CO * 4 + (CO / 16) * 64 * 16 +
                ((3 * (TH / 2 + 2) * (TW / 2 + 2) + 255) / 256 + UP) * 1024;  // + previous-image window + noise
  auto kern = modconv_hires_kernel<CI, CO, UP, F>;
  if (smem > 64 * 1024)
    MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const int n_tiles = (a.W / TW) * (a.H / TH);
  // persistent workgroups: ~2 per CU over all samples, each walks the tiles of ONE sample (its styles are baked
  // into the register-resident weights)
  constexpr int WG_PER_CU = CI == 64 ? 2 : 3;
  // (rounded DOWN: one workgroup more than the resident slots puts a second, almost empty round of workgroups behind the
  //  first - a batch of 112 ran 1.5x slower per frame than one of 128 with the rounding up)
  int per_sample = std::max(1, std::min(n_tiles, (256 * WG_PER_CU) / a.B));
  hipLaunchKernelGGL(kern, dim3(per_sample, a.B), dim3(256), smem, stream, a);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

bool hires_supported(int dtype, int Ci, int Co, int up, int H, int W) {
  if (dtype != MAUA_BF16 && dtype != MAUA_F16) return false;
  const bool shape = (Ci == 32 && Co == 32 && up == 1) || (Ci == 64 && Co == 64 && up == 1) ||
                     (Ci == 64 && Co == 32 && up == 2);
  if (!shape) return false;
  const int th = Ci == 32 ? 8 : 4;
  return W % 32 == 0 && H % th == 0;
}

int launch_modconv_hires(hipStream_t stream, const HiresArgs& a, int dtype) {
  if (a.B == 0) return MAUA_OK;
  MAUA_REQUIRE(hires_supported(dtype, a.Ci, a.Co, a.up, a.H, a.W), "modconv_hires: unsupported shape");
  MAUA_REQUIRE((long)a.H * a.up * a.W * a.up * std::max(a.Ci, a.Co) * 2 < (1L << 31),
               "modconv_hires: a sample must stay below 2 GiB (32-bit in-sample offsets)");
  MAUA_REQUIRE(a.act == MAUA_ACT_LRELU || a.act == MAUA_ACT_LINEAR, "modconv_hires: lrelu / linear only");
  MAUA_REQUIRE(a.y || a.rgb_out, "modconv_hires: no output (y is optional only with the fused toRGB)");
  HiresArgs b = a;
  if (a.act == MAUA_ACT_LINEAR) b.alpha = 1.f;
  MAUA_REQUIRE(b.alpha >= 0.f && b.alpha <= 1.f && b.gain > 0.f, "modconv_hires: needs 0 <= alpha <= 1 and gain > 0");
  if (dtype == MAUA_F16) {   // (round 6: the reference's own render dtype on the same kernels - v_mfma_f32_32x32x16_f16, half conversions)
    if (a.Ci == 32) return launch_hires_variant<32, 32, 1, f16_t>(stream, b);
    if (a.up == 1) return launch_hires_variant<64, 64, 1, f16_t>(stream, b);
    MAUA_REQUIRE(!a.rgb_out, "modconv_hires: toRGB fusion is for conv1 layers");
    return launch_hires_variant<64, 32, 2, f16_t>(stream, b);
  }
  if (a.Ci == 32) return launch_hires_variant<32, 32, 1, bf16_t>(stream, b);
  if (a.up == 1) return launch_hires_variant<64, 64, 1, bf16_t>(stream, b);
  MAUA_REQUIRE(!a.rgb_out, "modconv_hires: toRGB fusion is for conv1 layers");
  return launch_hires_variant<64, 32, 2, bf16_t>(stream, b);
}

}  // namespace maua
