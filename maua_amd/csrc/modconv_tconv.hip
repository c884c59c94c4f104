// Up-layers with the MINIMUM number of MACs: the reference's stride-2 transposed convolution (ops.py:211-224) on
// the matrix cores, leaving its 4x4 FIR + epilogue (ops.py:225, :184-185, :65-84) to upfir.hip.
//
//   t[2m+a', 2n+b', co] = sum_{p,q in {0,1}} sum_ci (s[ci] x[m-p, n-q, ci]) * Wc[a',b'][p][q][co][ci]
// where class (a',b') = output parity of t and Wc picks W[2p or 1][2q or 1] (odd parity: p/q = 0 only).  The four
// classes need 4 / 2 / 2 / 1 of the four input shifts = 9 taps per input pixel, against 36 for the four 3x3 phase
// kernels of modconv.hip.  One wave owns 32 positions x (4 classes x 32 channels): every wave runs the same static
// sequence of 9 MFMAs per 16-channel K-step (balanced, no runtime masks), A fragments are shared by the classes.
// Workgroup = 8 waves = 256 positions; weights of a K chunk = 9 [32 x KC] blocks, staged with the same
// register-prefetch pipeline as modconv.hip; the styles are applied while the halo is staged.
// The M domain is (H+1) x (W+1); its last row/column are launched as two thin strips so that the main launch keeps
// power-of-two tiles on H x W.
#include "common.h"
#include "internal.h"

namespace maua {

constexpr int TKCB = 64, TRS = TKCB + 16, TPR = TKCB / 16;  // K chunk bytes, LDS row stride, pieces per row

template <typename T> struct TMma;
template <> struct TMma<bf16_t> {
  __device__ static __forceinline__ void step(f32x16& acc, const u32x4& w, const u32x4& x) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0,
                                                  0, 0);
  }
  __device__ static __forceinline__ u32x4 scale(const u32x4& v, const float* sv) {
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; k++)
      o[k] = pack2bf(bf2f((bf16_t)(v[k] & 0xffff)) * sv[2 * k], bf2f((bf16_t)(v[k] >> 16)) * sv[2 * k + 1]);
    return o;
  }
};
template <> struct TMma<f16_t> {
  __device__ static __forceinline__ void step(f32x16& acc, const u32x4& w, const u32x4& x) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), acc, 0, 0, 0);
  }
  __device__ static __forceinline__ u32x4 scale(const u32x4& v, const float* sv) {
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = pack2h(Fmt16<f16_t>::lo(v[k]) * sv[2 * k], Fmt16<f16_t>::hi(v[k]) * sv[2 * k + 1]);
    return o;
  }
};
template <> struct TMma<float> {
  __device__ static __forceinline__ void step(f32x16& acc, const u32x4& w, const u32x4& x) {
    f32x4 wf = __builtin_bit_cast(f32x4, w), xf = __builtin_bit_cast(f32x4, x);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[0], xf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[1], xf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[2], xf[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[3], xf[3], acc, 0, 0, 0);
  }
  __device__ static __forceinline__ u32x4 scale(const u32x4& v, const float* sv) {
    f32x4 f = __builtin_bit_cast(f32x4, v);
    f[0] *= sv[0]; f[1] *= sv[1]; f[2] *= sv[2]; f[3] *= sv[3];
    return __builtin_bit_cast(u32x4, f);
  }
};

struct TconvGeom {
  int tw_log2, th, tiles_x, hw1, halo_px;
  unsigned inv_hw1;
  int oy0, ox0, hm, wm;  // sub-domain of the (H+1) x (W+1) position grid
  int tile0;             // first blockIdx.x of this region
};
struct TconvRegions {
  TconvGeom r[3];        // main H x W block, last column (n = W), last row (m = H)
  int halo_max;          // LDS carve: max halo_px over the regions
};

// weight block k of a stage: (shift slot, class).  slot s: (p,q) = (1,1), (1,0), (0,1), (0,0); class = 2a' + b'.
__device__ __constant__ const int kTconvSlot[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};
__device__ __constant__ const int kTconvCls[9] = {0, 0, 1, 0, 2, 0, 1, 2, 3};

template <typename T>
__global__ __launch_bounds__(512, 4) void tconv2_kernel(ConvArgs a, TconvRegions regs) {
  // one launch covers the three regions; a workgroup's geometry is uniform
  const TconvGeom g = (int)blockIdx.x >= regs.r[2].tile0 ? regs.r[2] : (int)blockIdx.x >= regs.r[1].tile0 ? regs.r[1] : regs.r[0];
  constexpr int NT = 512, BM = 256;
  constexpr int KC = TKCB / (int)sizeof(T);
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int WROW = TKCB;                          // weight rows are unpadded in LDS (LDS-direct loads are linear)
  constexpr int WBUF = 9 * 32 * WROW;                 // one K chunk of weights: 9 [32 x KC] blocks = 18 x 1 KB
  constexpr int WDMA = (WBUF / 1024 + 7) / 8;         // LDS-direct load instructions per wave per chunk
  constexpr int HREGS = (325 * TPR + NT - 1) / NT;   // halo <= 65*5 = 325 px (tile shapes: 9*33, 17*17, 33*9, 65*5)
  constexpr int ES = 128 * (int)sizeof(T) + 16;
  constexpr int PPP = 128 * (int)sizeof(T) / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* wt = smem + regs.halo_max * TRS;  // two buffers: chunk c + 1 arrives while chunk c is multiplied

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int tw = 1 << g.tw_log2;
  const int tile = blockIdx.x - g.tile0;
  const int tyi = tile / g.tiles_x, txi = tile - tyi * g.tiles_x;
  const int ty0 = g.oy0 + tyi * g.th, tx0 = g.ox0 + txi * tw;  // first position of the tile
  const int b = blockIdx.y, cb = blockIdx.z, CB = a.Co / 32;

  const T* xb = reinterpret_cast<const T*>(a.x) + (long)b * a.x_bstride;
  const T* wp = reinterpret_cast<const T*>(a.w);
  const float* sb = a.s + (long)b * a.Ci;

  // this wave's 32 positions
  const int m = wave * 32 + r;
  const int pty = m >> g.tw_log2, ptx = m & (tw - 1);
  const int offa = ((pty + 1) * g.hw1 + (ptx + 1)) * TRS + h * 16;
  // B fragments: the 16-byte piece p of weight row n sits at piece p ^ ((n >> 2) & 3) - with 64-byte rows this
  // XOR makes a quarter-wave's ds_read_b128 hit 16 distinct bank groups (the padding trick needs a non-linear fill)
  const int swz = (r >> 2) & 3;
  const int offb0 = r * WROW + ((h ^ swz) << 4), offb1 = r * WROW + (((h ^ swz) ^ 2) << 4);

  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; c++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[c][e] = 0.f;

  const int q = tid % TPR, rq = tid / TPR;
  int hoff[HREGS];  // element offsets inside the sample (a sample stays below 2^31 elements: launcher check)
#pragma unroll
  for (int i = 0; i < HREGS; i++) {
    const int p = rq + i * (NT / TPR);
    hoff[i] = -1;
    if (p < g.halo_px) {
      const int py = (int)(((unsigned)p * g.inv_hw1) >> 20);
      const int px = p - py * g.hw1;
      const int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
      if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) hoff[i] = (gy * a.W + gx) * a.Ci + q * EPC;
    }
  }
  u32x4 hreg[HREGS];
  float sv[EPC];
  // weights travel global -> LDS directly: instruction ii (0..17) fills the 1 KB slot of rows 16 ii .. 16 ii + 15, lane
  // l supplies row 16 ii + (l >> 2), LDS piece l & 3 = logical piece (l & 3) ^ ((row >> 2) & 3); wave w issues ii = w + 8 j
  int woff[WDMA];
#pragma unroll
  for (int j = 0; j < WDMA; j++) {
    const int ii = wave + 8 * j;
    const int row = std::min(16 * ii + (lane >> 2), 9 * 32 - 1);
    const int k = row >> 5, n = row & 31;
    const int piece = (lane & 3) ^ ((row >> 2) & 3);
    woff[j] = (((kTconvSlot[k] * CB + cb) * 4 + kTconvCls[k]) * 32 + n) * a.Ci + piece * EPC;
  }
#define TC_DMA_W(C0, BUF)                                                                                \
  {                                                                                                      \
    _Pragma("unroll") for (int j = 0; j < WDMA; j++)                                                    \
      if (wave + 8 * j < WBUF / 1024)                                                                    \
        lds_dma_b128(wp + (C0), (unsigned)woff[j] * (unsigned)sizeof(T), wt + (BUF) * WBUF + (wave + 8 * j) * 1024); \
  }

#define TC_LOAD(C0)                                                                                     \
  {                                                                                                     \
    _Pragma("unroll") for (int e = 0; e < EPC; e++) sv[e] = sb[(C0) + q * EPC + e];                     \
    _Pragma("unroll") for (int i = 0; i < HREGS; i++) {                                                \
      hreg[i] = u32x4{0u, 0u, 0u, 0u};                                                                  \
      if (hoff[i] >= 0) hreg[i] = *reinterpret_cast<const u32x4*>(xb + (hoff[i] + (C0)));                 \
    }                                                                                                   \
  }

  // Order matters: the weight request of a chunk is issued BEFORE that chunk's register loads (styles, halo).  vmcnt
  // retires in order, so once a wave has consumed those registers (the LDS writes at the top of the chunk) its weight
  // pieces have landed; the barrier after the LDS writes then makes all waves' pieces visible to all.
  const int n_chunks = a.Ci / KC;
  TC_DMA_W(0, 0)
  TC_LOAD(0)
  for (int c = 0; c < n_chunks; c++) {
    const char* wtb = wt + (c & 1) * WBUF;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < HREGS; i++) {
      const int p = rq + i * (NT / TPR);
      if (p < g.halo_px) *reinterpret_cast<u32x4*>(halo + p * TRS + q * 16) = TMma<T>::scale(hreg[i], sv);
    }
    __syncthreads();
    if (c + 1 < n_chunks) {
      TC_DMA_W((c + 1) * KC, (c + 1) & 1)
      TC_LOAD((c + 1) * KC)
    }
#pragma unroll
    for (int ks = 0; ks < TKCB / 32; ks++) {
      // A fragments for the four shifts (dy,dx) = (-1,-1), (-1,0), (0,-1), (0,0), each followed by the weight blocks
      // that use it (few fragments live at a time: the kernel runs at 128 VGPRs for 2 workgroups per CU)
#define TC_A(SH) (*reinterpret_cast<const u32x4*>(halo + offa + (SH) * TRS + ks * 32))
#define TC_B(K) (*reinterpret_cast<const u32x4*>(wtb + (K) * 32 * WROW + (ks == 0 ? offb0 : offb1)))
      {
        const u32x4 a0 = TC_A(-g.hw1 - 1);
        TMma<T>::step(acc[0], TC_B(0), a0);
      }
      {
        const u32x4 a1 = TC_A(-g.hw1);
        TMma<T>::step(acc[1], TC_B(2), a1);
        TMma<T>::step(acc[0], TC_B(1), a1);
      }
      {
        const u32x4 a2 = TC_A(-1);
        TMma<T>::step(acc[2], TC_B(4), a2);
        TMma<T>::step(acc[0], TC_B(3), a2);
      }
      {
        const u32x4 a3 = TC_A(0);
        TMma<T>::step(acc[3], TC_B(8), a3);
        TMma<T>::step(acc[1], TC_B(6), a3);
        TMma<T>::step(acc[2], TC_B(7), a3);
        TMma<T>::step(acc[0], TC_B(5), a3);
      }
#undef TC_A
#undef TC_B
    }
  }
#undef TC_LOAD
#undef TC_DMA_W

  // ---- raw t tile -> LDS [position][class*32 + ch] -> 16-byte NHWC pieces of t [2H+1][2W+1][Co]
  const int Ht = 2 * a.H + 1, Wt = 2 * a.W + 1;
  __syncthreads();
  char* epi = smem;
#pragma unroll
  for (int c = 0; c < 4; c++)
#pragma unroll
    for (int qd = 0; qd < 4; qd++) {
      char* dst = epi + m * ES + (c * 32 + 8 * qd + 4 * h) * (int)sizeof(T);
      if constexpr (sizeof(T) == 2)
        *reinterpret_cast<uint2*>(dst) = make_uint2(Fmt16<T>::pack2(acc[c][qd * 4 + 0], acc[c][qd * 4 + 1]),
                                                    Fmt16<T>::pack2(acc[c][qd * 4 + 2], acc[c][qd * 4 + 3]));
      else
        *reinterpret_cast<float4*>(dst) =
            make_float4(acc[c][qd * 4 + 0], acc[c][qd * 4 + 1], acc[c][qd * 4 + 2], acc[c][qd * 4 + 3]);
    }
  __syncthreads();
  char* yb = reinterpret_cast<char*>(a.y) + (long)b * Ht * Wt * a.Co * (long)sizeof(T);
  for (int p = tid; p < BM * PPP; p += NT) {
    const int mm = p / PPP, pc = p - mm * PPP;
    const int gy = ty0 + (mm >> g.tw_log2), gx = tx0 + (mm & (tw - 1));
    if (gy < g.oy0 + g.hm && gx < g.ox0 + g.wm) {
      const int nv = pc * EPC, cls = nv >> 5, ch = nv & 31;
      const int oy = 2 * gy + (cls >> 1), ox = 2 * gx + (cls & 1);
      if (oy < Ht && ox < Wt)
        *reinterpret_cast<uint4*>(yb + (((long)oy * Wt + ox) * a.Co + cb * 32 + ch) * (long)sizeof(T)) =
            *reinterpret_cast<const uint4*>(epi + mm * ES + pc * 16);
    }
  }
}

static int make_region(TconvGeom& g, int oy0, int ox0, int hm, int wm, int tile0) {
  int tw = wm > 16 ? 32 : wm > 8 ? 16 : wm > 4 ? 8 : 4;
  g.tw_log2 = tw == 32 ? 5 : tw == 16 ? 4 : tw == 8 ? 3 : 2;
  g.th = 256 / tw;
  g.tiles_x = cdiv(wm, tw);
  g.hw1 = tw + 1;
  g.inv_hw1 = ((1u << 20) + g.hw1 - 1) / g.hw1;
  g.halo_px = (g.th + 1) * g.hw1;
  g.oy0 = oy0; g.ox0 = ox0; g.hm = hm; g.wm = wm;
  g.tile0 = tile0;
  return g.tiles_x * cdiv(hm, g.th);
}

template <typename T>
static int launch_tconv_t(hipStream_t stream, const ConvArgs& a) {
  constexpr int KC = TKCB / (int)sizeof(T);
  MAUA_REQUIRE(a.Ci % KC == 0 && a.Co % 32 == 0, "tconv2: channel counts must be multiples of 32");
  MAUA_REQUIRE((long)a.H * a.W * a.Ci < (1L << 31) && 16L * a.Co * a.Ci < (1L << 31), "tconv2: 32-bit offsets");
  if (a.B == 0) return MAUA_OK;
  // main H x W block, then the last column (n = W) and the last row (m = H) of the (H+1) x (W+1) position grid
  TconvRegions regs;
  int nt = make_region(regs.r[0], 0, 0, a.H, a.W, 0);
  if (a.variant == TCONV_EDGES_ONLY) nt = 0;  // the main block is somebody else's (modconv_tconv_dma.hip): thin regions only
  nt += make_region(regs.r[1], 0, a.W, a.H + 1, 1, nt);
  nt += make_region(regs.r[2], a.H, 0, 1, a.W, nt);
  regs.halo_max = std::max(regs.r[0].halo_px, std::max(regs.r[1].halo_px, regs.r[2].halo_px));
  MAUA_REQUIRE(regs.halo_max <= 325, "tconv2: halo does not fit the prefetch registers");
  size_t smem = std::max((size_t)regs.halo_max * TRS + (size_t)2 * 9 * 32 * TKCB, (size_t)256 * (128 * sizeof(T) + 16));
  auto kern = tconv2_kernel<T>;
  if (smem > 64 * 1024)
    MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(kern, dim3(nt, a.B, a.Co / 32), dim3(512), smem, stream, a, regs);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int launch_tconv2(hipStream_t stream, int dtype, const ConvArgs& a) {
  if (dtype == MAUA_BF16) return launch_tconv_t<bf16_t>(stream, a);
  if (dtype == MAUA_F16) return launch_tconv_t<f16_t>(stream, a);
  if (dtype == MAUA_F32) return launch_tconv_t<float>(stream, a);
  return fail("tconv2: unsupported dtype");
}

// ---- weights: f32 [Co][Ci][3][3] -> T [slot 4][Co/32][class 4][32][Ci]  (zero where a class does not use a slot)
template <typename T>
__global__ __launch_bounds__(256) void prep_tconv_weights_kernel(const float* __restrict__ w, T* __restrict__ wt, int Co,
                                                                 int Ci, int flip) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)Co * Ci) return;
  const int co = (int)(idx / Ci), ci = (int)(idx - (long)co * Ci);
  float wv[9];
  for (int t = 0; t < 9; t++) wv[t] = w[((long)co * Ci + ci) * 9 + t];
  const int CB = Co / 32, cb = co >> 5, n = co & 31;
  for (int s = 0; s < 4; s++) {
    const int p = s < 2 ? 1 : 0, q = (s == 0 || s == 2) ? 1 : 0;
    for (int cls = 0; cls < 4; cls++) {
      const int ap = cls >> 1, bp = cls & 1;
      const int ky = ap == 0 ? 2 * p : (p == 0 ? 1 : -1);
      const int kx = bp == 0 ? 2 * q : (q == 0 ? 1 : -1);
      float v = 0.f;
      if (ky >= 0 && kx >= 0) v = flip ? wv[(2 - ky) * 3 + (2 - kx)] : wv[ky * 3 + kx];
      Elem<T>::store(wt + ((((long)s * CB + cb) * 4 + cls) * 32 + n) * Ci + ci, v);
    }
  }
}

int launch_prep_tconv_weights(hipStream_t stream, int dtype, const float* w, void* wt, int Co, int Ci, int flip) {
  MAUA_REQUIRE(Co % 32 == 0, "prep_tconv_weights: Co must be a multiple of 32");
  const long n = (long)Co * Ci;
  dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == MAUA_BF16)
    hipLaunchKernelGGL(prep_tconv_weights_kernel<bf16_t>, grid, dim3(256), 0, stream, w, (bf16_t*)wt, Co, Ci, flip);
  else if (dtype == MAUA_F16)
    hipLaunchKernelGGL(prep_tconv_weights_kernel<f16_t>, grid, dim3(256), 0, stream, w, (f16_t*)wt, Co, Ci, flip);
  else if (dtype == MAUA_F32)
    hipLaunchKernelGGL(prep_tconv_weights_kernel<float>, grid, dim3(256), 0, stream, w, (float*)wt, Co, Ci, flip);
  else
    return fail("prep_tconv_weights: unsupported dtype");
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace maua
