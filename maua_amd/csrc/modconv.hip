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
#include "common.h"
#include "internal.h"

namespace maua {

constexpr int KCB = 64;       // bytes of K (input channels) per LDS row chunk
constexpr int RS = KCB + 16;  // LDS row stride in bytes

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  __device__ static __forceinline__ void step(f32x16& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0,
                                                  0, 0);
  }
  // scale 8 bf16 by 8 f32 styles, round to nearest even
  __device__ static __forceinline__ uint4 scale(const uint4& v, const float* sv) {
    uint4 o;
    o.x = pack2bf(bf2f((bf16_t)(v.x & 0xffff)) * sv[0], bf2f((bf16_t)(v.x >> 16)) * sv[1]);
    o.y = pack2bf(bf2f((bf16_t)(v.y & 0xffff)) * sv[2], bf2f((bf16_t)(v.y >> 16)) * sv[3]);
    o.z = pack2bf(bf2f((bf16_t)(v.z & 0xffff)) * sv[4], bf2f((bf16_t)(v.z >> 16)) * sv[5]);
    o.w = pack2bf(bf2f((bf16_t)(v.w & 0xffff)) * sv[6], bf2f((bf16_t)(v.w >> 16)) * sv[7]);
    return o;
  }
};
template <> struct Mma<float> {
  // lane half h holds k = 8j+4h+e (e = 0..3): MFMA e consumes element e of both operands, so A and B see the
  // same K permutation and the sum is over the same set of products.
  __device__ static __forceinline__ void step(f32x16& acc, const uint4& a, const uint4& b) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], acc, 0, 0, 0);
  }
  __device__ static __forceinline__ uint4 scale(const uint4& v, const float* sv) {
    f32x4 f = __builtin_bit_cast(f32x4, v);
    f[0] *= sv[0]; f[1] *= sv[1]; f[2] *= sv[2]; f[3] *= sv[3];
    return __builtin_bit_cast(uint4, f);
  }
};

struct ConvGeom {
  int tw_log2, th;      // tile = th x (1 << tw_log2) pixels
  int tiles_x;          // tiles per row
  int hw2;              // tw + 2
  unsigned inv_hw2;     // ceil(2^20 / hw2) for the halo pixel -> (row, col) split
  int halo_px;          // (th+2)*(tw+2)
  int phases;           // up*up
};

template <typename T, int WAVES_M, int WAVES_N, int WM, int WN, int TG>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void modconv3x3_kernel(ConvArgs a, ConvGeom g) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int BN = WAVES_N * WN * 32;
  constexpr int KC = KCB / (int)sizeof(T);   // channels per K chunk
  constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-byte piece
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
  const int n0 = blockIdx.y * BN;
  const int b = blockIdx.z / g.phases, phase = blockIdx.z - b * g.phases;

  const T* xb = reinterpret_cast<const T*>(a.x) + (long)b * a.x_bstride;
  const T* wp = reinterpret_cast<const T*>(a.w) + (long)phase * 9 * a.Co * a.Ci;
  const float* sb = a.s + (long)b * a.Ci;

  // per-lane fragment base offsets (bytes)
  int offa[WM], offb[WN];
#pragma unroll
  for (int i = 0; i < WM; i++) {
    int m = (wm * WM + i) * 32 + r;
    int ty = m >> g.tw_log2, tx = m & (tw - 1);
    offa[i] = ((ty + 1) * g.hw2 + (tx + 1)) * RS + h * 16;
  }
#pragma unroll
  for (int j = 0; j < WN; j++) offb[j] = ((wn * WN + j) * 32 + r) * RS + h * 16;

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; i++)
#pragma unroll
    for (int j = 0; j < WN; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  const int q = tid & 3;  // which 16-byte piece of the 64-byte row this thread stages (NT % 4 == 0)

  for (int c0 = 0; c0 < a.Ci; c0 += KC) {
    // styles of this thread's piece
    float sv[EPC];
#pragma unroll
    for (int e = 0; e < EPC; e++) sv[e] = sb[c0 + q * EPC + e];
    __syncthreads();  // all waves done with the previous chunk's halo / weights
    // ---- stage the (th+2) x (tw+2) input halo, scaled by the styles, zero outside the image
    for (int p = tid >> 2; p < g.halo_px; p += NT / 4) {
      int py = (int)(((unsigned)p * g.inv_hw2) >> 20);
      int px = p - py * g.hw2;
      int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
        v = *reinterpret_cast<const uint4*>(xb + ((long)gy * a.W + gx) * a.Ci + c0 + q * EPC);
        v = Mma<T>::scale(v, sv);
      }
      *reinterpret_cast<uint4*>(halo + p * RS + q * 16) = v;
    }
    for (int tg = 0; tg < 9; tg += TG) {
      if (tg > 0) __syncthreads();  // previous tap group consumed
      // ---- stage TG taps of weights: rows (t, n) of KC channels
      for (int row = tid >> 2; row < TG * BN; row += NT / 4) {
        int t = row / BN, n = row - t * BN;
        uint4 v = *reinterpret_cast<const uint4*>(wp + ((long)(tg + t) * a.Co + n0 + n) * a.Ci + c0 + q * EPC);
        *reinterpret_cast<uint4*>(wt + row * RS + q * 16) = v;
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < TG; t++) {
        const int tap = tg + t;
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const int tapoff = (dy * g.hw2 + dx) * RS;
#pragma unroll
        for (int ks = 0; ks < KCB / 32; ks++) {
          uint4 af[WM], bf[WN];
#pragma unroll
          for (int i = 0; i < WM; i++) af[i] = *reinterpret_cast<const uint4*>(halo + offa[i] + tapoff + ks * 32);
#pragma unroll
          for (int j = 0; j < WN; j++) bf[j] = *reinterpret_cast<const uint4*>(wt + t * BN * RS + offb[j] + ks * 32);
#pragma unroll
          for (int i = 0; i < WM; i++)
#pragma unroll
            for (int j = 0; j < WN; j++) Mma<T>::step(acc[i][j], af[i], bf[j]);
        }
      }
    }
  }

  // ---- epilogue: demod, noise, bias, activation, gain, clamp, store NHWC
  const int Ho = a.H * a.up, Wo = a.W * a.up;
  const int pa = phase / a.up, pb = phase - pa * a.up;  // output parity (row, col); 0,0 when up == 1
  T* yb = reinterpret_cast<T*>(a.y) + (long)b * Ho * Wo * a.Co;
  const float* nb = a.noise ? a.noise + (long)b * a.noise_bstride : nullptr;
#pragma unroll
  for (int j = 0; j < WN; j++) {
    const int n = n0 + (wn * WN + j) * 32 + r;
    const float dv = a.d ? a.d[(long)b * a.Co + n] : 1.f;
    const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < WM; i++) {
#pragma unroll
      for (int e = 0; e < 16; e++) {
        int m = (wm * WM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        int ty = m >> g.tw_log2, tx = m & (tw - 1);
        int gy = ty0 + ty, gx = tx0 + tx;
        if (ty < g.th && gy < a.H && gx < a.W) {
          int oy = gy * a.up + pa, ox = gx * a.up + pb;
          long pix = (long)oy * Wo + ox;
          float v = acc[i][j][e] * dv;
          if (nb) v += nb[pix] * a.noise_strength;
          v = activate(v + bv, a.act, a.alpha);
          v *= a.gain;
          if (a.clamp >= 0.f) v = fminf(fmaxf(v, -a.clamp), a.clamp);
          Elem<T>::store(yb + pix * a.Co + n, v);
        }
      }
    }
  }
}

template <typename T, int WAVES_M, int WAVES_N, int WM, int WN, int TG>
static int launch_variant(hipStream_t stream, const ConvArgs& a) {
  constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32, NT = WAVES_M * WAVES_N * 64;
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
  size_t smem = (size_t)g.halo_px * RS + (size_t)TG * BN * RS;
  MAUA_REQUIRE(smem <= 160 * 1024, "modconv3x3: LDS budget exceeded");
  auto kern = modconv3x3_kernel<T, WAVES_M, WAVES_N, WM, WN, TG>;
  if (smem > 64 * 1024)
    MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  dim3 grid(g.tiles_x * tiles_y, a.Co / BN, a.B * g.phases);
  hipLaunchKernelGGL(kern, grid, dim3(NT), smem, stream, a, g);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

template <typename T>
static int launch_modconv_t(hipStream_t stream, const ConvArgs& a) {
  constexpr int KC = KCB / (int)sizeof(T);
  MAUA_REQUIRE(a.Ci % KC == 0, "modconv3x3: Ci must be a multiple of the K chunk (pad channels)");
  MAUA_REQUIRE(a.Co % 32 == 0, "modconv3x3: Co must be a multiple of 32 (pad channels)");
  MAUA_REQUIRE(a.up == 1 || a.up == 2, "modconv3x3: up must be 1 or 2");
  if (a.B == 0) return MAUA_OK;
  if (a.Co % 128 == 0) return launch_variant<T, 2, 2, 2, 2, 3>(stream, a);
  if (a.Co % 64 == 0) return launch_variant<T, 4, 1, 2, 2, 9>(stream, a);
  return launch_variant<T, 4, 1, 2, 1, 9>(stream, a);
}

int launch_modconv3x3(hipStream_t stream, int dtype, const ConvArgs& a) {
  if (dtype == MAUA_BF16) return launch_modconv_t<bf16_t>(stream, a);
  if (dtype == MAUA_F32) return launch_modconv_t<float>(stream, a);
  return fail("modconv3x3: unsupported dtype");
}

// ------------------------------------------------------------------------------------------------ weight prep
// f32 [Co][Ci][k][k] -> T [phases][k*k][Cop][Cip] (zero padded) and Wsq[co][ci] = sum_taps W^2.
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
          Elem<T>::store(wt + ((long)ph * 9 + t) * plane + idx, K[2 * ky + 1 - pa][2 * kx + 1 - pb]);
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
  else if (dtype == MAUA_F32)
    hipLaunchKernelGGL(prep_weights_kernel<float>, grid, dim3(256), 0, stream, w, (float*)wt, wsq, Co, Ci, k, up, flip,
                       Cop, Cip);
  else
    return fail("prep_weights: unsupported dtype");
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// ------------------------------------------------------------------------------------------------ styles / demod
// One workgroup per (layer, sample).  Phase 1: s = affine(w) (stylegan2.py:48-58,:230,:269); phase 2: demod
// coefficients (ops.py:168-171) or pre-modulated toRGB weights.
__global__ __launch_bounds__(256) void styles_kernel(const StyleLayer* __restrict__ layers, const float* __restrict__ ws,
                                                     int num_ws, int w_dim) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* wv = reinterpret_cast<float*>(smem);  // w_dim
  float* ss = wv + w_dim;                      // Cin
  const StyleLayer L = layers[blockIdx.x];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const float* wsrc = ws + ((long)b * num_ws + L.w_index) * w_dim;
  for (int i = threadIdx.x; i < w_dim; i += blockDim.x) wv[i] = wsrc[i];
  __syncthreads();
  const float wgain = rsqrtf((float)w_dim);
  for (int ci = wave; ci < L.Cs; ci += nw) {
    float s = 0.f;
    if (ci < L.Cin) {
      const float* row = L.affine_w + (long)ci * w_dim;
      float acc = 0.f;
      for (int k = lane; k < w_dim; k += 64) acc += row[k] * wv[k];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
      s = (acc * wgain + L.affine_b[ci]) * L.scale;
    }
    if (lane == 0) {
      if (ci < L.Cin) ss[ci] = s;
      L.s[(long)b * L.Cs + ci] = s;
    }
  }
  __syncthreads();
  if (L.d) {
    for (int co = wave; co < L.Cd; co += nw) {
      float dv = 0.f;
      if (co < L.Co) {
        const float* row = L.wsq + (long)co * L.Cin;
        float acc = 0.f;
        for (int k = lane; k < L.Cin; k += 64) acc += ss[k] * ss[k] * row[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        dv = rsqrtf(acc + 1e-8f);
      }
      if (lane == 0) L.d[(long)b * L.Cd + co] = dv;
    }
  }
  if (L.wmod) {
    for (int i = threadIdx.x; i < 3 * L.Cin; i += blockDim.x) {
      int ci = i % L.Cin;
      L.wmod[(long)b * 3 * L.Cin + i] = L.wrgb[i] * ss[ci];
    }
  }
}

int launch_styles(hipStream_t stream, const StyleLayer* layers_dev, int n_layers, const float* ws, int num_ws,
                  int w_dim, int B) {
  if (B == 0 || n_layers == 0) return MAUA_OK;
  size_t smem = (size_t)(w_dim + 1024) * sizeof(float);
  hipLaunchKernelGGL(styles_kernel, dim3(n_layers, B), dim3(256), smem, stream, layers_dev, ws, num_ws, w_dim);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// ------------------------------------------------------------------------------------------------ toRGB + skip
// stylegan2.py:268-272 (1x1 modulated conv without demodulation, bias, clamp) fused with SynthesisBlock's
// img = upsample2d(img) + y (stylegan2.py:372-378; ops.py:117-133).  HBM-bound: x is read once in 16-byte
// pieces, LP lanes cooperate on one pixel and reduce with wave shuffles.
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
        xv[0] = bf2f((bf16_t)(v.x & 0xffff)); xv[1] = bf2f((bf16_t)(v.x >> 16));
        xv[2] = bf2f((bf16_t)(v.y & 0xffff)); xv[3] = bf2f((bf16_t)(v.y >> 16));
        xv[4] = bf2f((bf16_t)(v.z & 0xffff)); xv[5] = bf2f((bf16_t)(v.z >> 16));
        xv[6] = bf2f((bf16_t)(v.w & 0xffff)); xv[7] = bf2f((bf16_t)(v.w >> 16));
      } else {
        xv[0] = __uint_as_float(v.x); xv[1] = __uint_as_float(v.y);
        xv[2] = __uint_as_float(v.z); xv[3] = __uint_as_float(v.w);
      }
      const float* w0 = wm + pc * EPC;
#pragma unroll
      for (int e = 0; e < EPC; e++) {
        r0 += xv[e] * w0[e];
        r1 += xv[e] * w0[a.C + e];
        r2 += xv[e] * w0[2 * a.C + e];
      }
    }
    for (int o = LP >> 1; o > 0; o >>= 1) {
      r0 += __shfl_xor(r0, o);
      r1 += __shfl_xor(r1, o);
      r2 += __shfl_xor(r2, o);
    }
    if (sub == 0) {
      int y = (int)(p / a.W), x = (int)(p - (long)y * a.W);
      float o3[3] = {r0 + a.bias[0], r1 + a.bias[1], r2 + a.bias[2]};
#pragma unroll
      for (int c = 0; c < 3; c++)
        if (a.clamp >= 0.f) o3[c] = fminf(fmaxf(o3[c], -a.clamp), a.clamp);
      if (a.prev) {
        // upsample2d: zero-insert x2, pad (2,1,2,1), correlate with 4f
        const float* pv = a.prev + (long)b * 3 * Hp * Wp;
        float u3[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; u++) {
          int Y = y + u - 2;
          if (Y < 0 || (Y & 1)) continue;
          int iy = Y >> 1;
          if (iy >= Hp) continue;
#pragma unroll
          for (int v = 0; v < 4; v++) {
            int X = x + v - 2;
            if (X < 0 || (X & 1)) continue;
            int ix = X >> 1;
            if (ix >= Wp) continue;
            float f = a.fir[u * 4 + v];
            long o = (long)iy * Wp + ix;
            u3[0] += pv[o] * f;
            u3[1] += pv[(long)Hp * Wp + o] * f;
            u3[2] += pv[2L * Hp * Wp + o] * f;
          }
        }
        // reference order: img = upsample2d(img) + y
        o3[0] = u3[0] + o3[0]; o3[1] = u3[1] + o3[1]; o3[2] = u3[2] + o3[2];
      }
      float* ob = a.out + (long)b * 3 * HW + p;
      ob[0] = o3[0]; ob[HW] = o3[1]; ob[2 * HW] = o3[2];
    }
  }
}

int launch_torgb(hipStream_t stream, int dtype, const RgbArgs& a) {
  if (a.B == 0) return MAUA_OK;
  const int epc = dtype == MAUA_BF16 ? 8 : 4;
  MAUA_REQUIRE(a.C % epc == 0, "torgb: C must be a multiple of the 16-byte piece");
  int pieces = a.C / epc;
  int lp_log2 = 0;
  while ((1 << (lp_log2 + 1)) <= pieces && lp_log2 < 4) lp_log2++;
  long HW = (long)a.H * a.W;
  int ppb = 256 >> lp_log2;
  int gx = (int)std::min<long>((HW + ppb - 1) / ppb, 2048);
  size_t smem = (size_t)3 * a.C * sizeof(float);
  if (dtype == MAUA_BF16)
    hipLaunchKernelGGL(torgb_kernel<bf16_t>, dim3(gx, a.B), dim3(256), smem, stream, a, lp_log2);
  else if (dtype == MAUA_F32)
    hipLaunchKernelGGL(torgb_kernel<float>, dim3(gx, a.B), dim3(256), smem, stream, a, lp_log2);
  else
    return fail("torgb: unsupported dtype");
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace maua
