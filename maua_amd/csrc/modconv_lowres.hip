// Modulated 3x3 convolution for the lowest-resolution layers (<= 8 x 8 input pixels per sample: the 4^2 / 8^2 blocks and
// the up-layers that leave them).  Same arithmetic as modconv.hip (reference ops.py:146-233 + the bias_act of
// stylegan2.py:238-250), different decomposition.
//
// Why a separate kernel: modconv.hip tiles ONE sample's pixels (256 or 128 per workgroup).  A 4^2 sample fills 6 % of
// that M tile, an 8^2 sample 25 %: ablation builds showed these layers spend ~25 us on padded MFMAs and ~35 us in a
// chain of 16-24 dependent load -> LDS -> barrier stages, for ~0.01 ms of useful matrix work.  Here the GEMM is taken
// over ALL samples at once,
//     M = B*H*W pixels (batch-major),  N = Co * up^2 (virtual channels, parity in N like modconv.hip),  K = 9 * Ci,
// in 64 x 128 tiles, and K is split into `ksplit` slices so that a few hundred short workgroups (2-4 stages each)
// cover the chip.  Three launches:
//   1. premod:  xm = bf16(x * s)            (the styles differ per sample, so they are applied before the samples share
//                                            an M tile; same rounding as the staging of modconv.hip)
//   2. conv:    partial[slice][m][n] (f32)  A rows are gathered per tap straight from xm (no halo reuse: the whole
//                                            activation tensor of such a layer is 16-130 KB per sample and L2-resident)
//   3. reduce + epilogue: slices added in a fixed order (deterministic), then demod / noise / bias / lrelu / clamp and
//                         the NHWC store with the parity decode of the up-layers.
#include "common.h"
#include "internal.h"

namespace maua {

namespace {

template <typename T> struct LMma;
template <> struct LMma<bf16_t> {
  __device__ static __forceinline__ void step(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0,
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
template <> struct LMma<float> {
  // lane half h holds k = 8j+4h+e (e = 0..3): MFMA e consumes element e of both operands (same K permutation for A, B)
  __device__ static __forceinline__ void step(f32x16& acc, const u32x4& a, const u32x4& b) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], acc, 0, 0, 0);
  }
  __device__ static __forceinline__ u32x4 scale(const u32x4& v, const float* sv) {
    f32x4 f = __builtin_bit_cast(f32x4, v);
    f[0] *= sv[0]; f[1] *= sv[1]; f[2] *= sv[2]; f[3] *= sv[3];
    return __builtin_bit_cast(u32x4, f);
  }
};

constexpr int LKCB = 128, LRS = LKCB + 16;  // bytes of K per stage and LDS row stride (conflict-free ds_read_b128)
constexpr int LBM = 64, LBN = 128;

struct LowresGeom {
  int M, CoV, stages_total, ksplit;
};

// ---- 1. xm[b][p][ci] = x[b][p][ci] * s[b][ci]
template <typename T>
__global__ __launch_bounds__(256) void lowres_premod_kernel(const T* __restrict__ x, long x_bstride,
                                                            const float* __restrict__ s, T* __restrict__ xm, int B, int HW,
                                                            int Ci) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const int ppp = Ci / EPC;  // 16-byte pieces per pixel
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * HW * ppp) return;
  const int pc = (int)(idx % ppp);
  const long bp = idx / ppp;
  const int p = (int)(bp % HW), b = (int)(bp / HW);
  float sv[EPC];
#pragma unroll
  for (int e = 0; e < EPC; e++) sv[e] = s[(long)b * Ci + pc * EPC + e];
  const u32x4 v = *reinterpret_cast<const u32x4*>(x + (long)b * x_bstride + (long)p * Ci + pc * EPC);
  *reinterpret_cast<u32x4*>(xm + bp * Ci + pc * EPC) = LMma<T>::scale(v, sv);
}

// ---- 2. one K slice of a 64 x 128 tile
template <typename T>
__global__ __launch_bounds__(256) void lowres_conv_kernel(const T* __restrict__ xm, const T* __restrict__ wp,
                                                          float* __restrict__ ws, int H, int W, int Ci, LowresGeom g) {
  constexpr int KC = LKCB / (int)sizeof(T), EPC = 16 / (int)sizeof(T);
  __shared__ __attribute__((aligned(16))) char a_s[LBM * LRS];
  __shared__ __attribute__((aligned(16))) char b_s[LBN * LRS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * LBM, n0 = blockIdx.y * LBN, ksi = blockIdx.z;
  const int spk = g.stages_total / g.ksplit;
  const int s_begin = ksi * spk, s_end = s_begin + spk;
  const int HW = H * W;
  const int q = tid & 7, row0 = tid >> 3;  // staging role: 16-byte piece q of rows row0 (+32 ...)

  int apix[2], ay[2], ax[2];  // this thread's two A rows: sample-pixel base (elements / Ci), coordinates; -1: padding row
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int m = m0 + row0 + i * 32;
    if (m < g.M) {
      const int b = m / HW, p = m - b * HW;
      ay[i] = p / W; ax[i] = p - ay[i] * W;
      apix[i] = b * HW;
    } else {
      apix[i] = -1; ay[i] = 0; ax[i] = 0;
    }
  }
  u32x4 areg[2], breg[4];
#define LOWRES_LOAD(S)                                                                                   \
  {                                                                                                      \
    const int c_ = (S) / 9, t_ = (S) - c_ * 9;                                                           \
    const int dy_ = t_ / 3 - 1, dx_ = t_ - (t_ / 3) * 3 - 1;                                             \
    _Pragma("unroll") for (int i = 0; i < 2; i++) {                                                     \
      const int sy = ay[i] + dy_, sx = ax[i] + dx_;                                                      \
      areg[i] = u32x4{0u, 0u, 0u, 0u};                                                                   \
      if (apix[i] >= 0 && sy >= 0 && sy < H && sx >= 0 && sx < W)                                        \
        areg[i] = *reinterpret_cast<const u32x4*>(xm + ((long)(apix[i] + sy * W + sx) * Ci + c_ * KC + q * EPC)); \
    }                                                                                                    \
    _Pragma("unroll") for (int j = 0; j < 4; j++)                                                       \
      breg[j] = *reinterpret_cast<const u32x4*>(wp + (((long)t_ * g.CoV + n0 + row0 + j * 32) * Ci + c_ * KC + q * EPC)); \
  }

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[i][e] = 0.f;

  LOWRES_LOAD(s_begin)
  for (int s = s_begin; s < s_end; s++) {
    __syncthreads();  // the previous stage's fragment reads are done
#pragma unroll
    for (int i = 0; i < 2; i++) *reinterpret_cast<u32x4*>(a_s + (row0 + i * 32) * LRS + q * 16) = areg[i];
#pragma unroll
    for (int j = 0; j < 4; j++) *reinterpret_cast<u32x4*>(b_s + (row0 + j * 32) * LRS + q * 16) = breg[j];
    __syncthreads();
    if (s + 1 < s_end) LOWRES_LOAD(s + 1)  // flies during the MFMAs
#pragma unroll
    for (int ks = 0; ks < LKCB / 32; ks++) {
      const u32x4 bf = *reinterpret_cast<const u32x4*>(b_s + (wave * 32 + r) * LRS + ks * 32 + h * 16);
      const u32x4 a0 = *reinterpret_cast<const u32x4*>(a_s + r * LRS + ks * 32 + h * 16);
      const u32x4 a1 = *reinterpret_cast<const u32x4*>(a_s + (32 + r) * LRS + ks * 32 + h * 16);
      LMma<T>::step(acc[0], bf, a0);  // rows = channels, cols = pixels: a lane owns one pixel and 4-channel runs
      LMma<T>::step(acc[1], bf, a1);
    }
  }
#undef LOWRES_LOAD
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int m = m0 + i * 32 + r;
    if (m < g.M) {
      float* dst = ws + ((long)ksi * g.M + m) * g.CoV + n0 + wave * 32 + 4 * h;
#pragma unroll
      for (int qd = 0; qd < 4; qd++)
        *reinterpret_cast<float4*>(dst + 8 * qd) =
            make_float4(acc[i][qd * 4], acc[i][qd * 4 + 1], acc[i][qd * 4 + 2], acc[i][qd * 4 + 3]);
    }
  }
}

// ---- 3. y = act((sum_slices partial) * d + noise + bias) * gain, clamp; one thread per 4 virtual channels of one pixel
template <typename T>
__global__ __launch_bounds__(256) void lowres_epilogue_kernel(ConvArgs a, const float* __restrict__ ws, int ksplit) {
  const int CoV = a.Co * a.up * a.up, q4 = CoV / 4;
  const int HW = a.H * a.W;
  const long total = (long)a.B * HW * q4;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int nv = (int)(idx % q4) * 4;
  const long m = idx / q4;  // sample * H*W + pixel
  const int pix = (int)(m % HW), b = (int)(m / HW);
  const int gy = pix / a.W, gx = pix - gy * a.W;
  const long slice = (long)a.B * HW * CoV;
  const float* src = ws + m * CoV + nv;
  float4 s4 = *reinterpret_cast<const float4*>(src);
  for (int k = 1; k < ksplit; k++) {  // fixed order
    const float4 t = *reinterpret_cast<const float4*>(src + k * slice);
    s4.x += t.x; s4.y += t.y; s4.z += t.z; s4.w += t.w;
  }
  const int ph = nv / a.Co, co = nv - ph * a.Co;  // a 4-channel run never straddles a parity (Co % 32 == 0)
  const int pa = ph >> (a.up - 1), pb = ph & (a.up - 1);
  const int Ho = a.H * a.up, Wo = a.W * a.up;
  const long opix = (long)(gy * a.up + pa) * Wo + gx * a.up + pb;
  float nz = 0.f;
  if (a.noise) nz = a.noise[(long)b * a.noise_bstride + opix] * a.noise_strength * (a.noise_scale ? a.noise_scale[b] : 1.f);
  float4 dv = make_float4(1.f, 1.f, 1.f, 1.f), bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.d) dv = *reinterpret_cast<const float4*>(a.d + (long)b * a.Co + co);
  if (a.bias) bv = *reinterpret_cast<const float4*>(a.bias + co);
  const float acc[4] = {s4.x, s4.y, s4.z, s4.w}, dd[4] = {dv.x, dv.y, dv.z, dv.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
  const float alpha = a.act == MAUA_ACT_LINEAR ? 1.f : a.alpha;
  const bool fast = (a.act == MAUA_ACT_LRELU || a.act == MAUA_ACT_LINEAR) && alpha >= 0.f && alpha <= 1.f && a.gain > 0.f;
  const float cl = a.clamp >= 0.f ? a.clamp : 3.0e38f;
  float v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (fast) {  // the folded-gain form of modconv.hip's epilogue (lrelu is positively homogeneous)
      float t = fmaf(acc[k], dd[k] * a.gain, (nz + bb[k]) * a.gain);
      t = fmaxf(t, t * alpha);
      v[k] = __builtin_amdgcn_fmed3f(t, -cl, cl);
    } else {
      float t = activate(acc[k] * dd[k] + nz + bb[k], a.act, a.alpha) * a.gain;
      if (a.clamp >= 0.f) t = fminf(fmaxf(t, -a.clamp), a.clamp);
      v[k] = t;
    }
  }
  if (a.res) {  // residual added after activation / gain / clamp (plain convolutions of the diffusion UNet, unet.hip)
    const T* rp = reinterpret_cast<const T*>(a.res) + (long)b * a.res_bstride + opix * a.res_pstride + co;
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] += Elem<T>::load(rp + k);
  }
  const int yps = a.y_pstride ? a.y_pstride : a.Co;
  T* dst = reinterpret_cast<T*>(a.y) + (long)b * (a.y_bstride ? a.y_bstride : (long)Ho * Wo * yps) + opix * yps + a.y_coff + co;
  if constexpr (sizeof(T) == 2)
    *reinterpret_cast<uint2*>(dst) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
  else
    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
}

LowresGeom lowres_geom(int esize, int B, int H, int W, int Ci, int Co, int up, int want_override = 0) {
  LowresGeom g;
  g.M = B * H * W;
  g.CoV = Co * up * up;
  g.stages_total = 9 * (Ci / (LKCB / esize));
  // K slices: chosen from the layer shape only, never from the batch size - the order in which the slices are added
  // must not depend on how many frames share a batch (frames are bit-identical wherever they are rendered).  Aim: about
  // 18 slice-tiles per sample (~600 workgroups at 32 frames, 4-18 stages each); more slices cost more in workspace
  // traffic than they save in chain length.
  const int per_sample = std::max(1, (H * W * (g.CoV / LBN) + LBM - 1) / LBM);  // 64 x 128 tiles one sample fills
  int want = std::max(2, 18 / per_sample);  // measured at B = 32: 18 / 4 / 4 / 2 slices for the four layers of the 1024^2 net
  // plain convolutions of a few samples with long K (diffusion UNet at 16^2 / 8^2, 9 * 1024 ... 9 * 2048): aim at ~256
  // slice-tiles per sample instead, again from the layer shape only
  if (want_override) want = std::max(1, std::min(g.stages_total, want_override / per_sample));
  g.ksplit = g.stages_total;
  for (int k = want; k <= g.stages_total; k++)
    if (g.stages_total % k == 0) { g.ksplit = k; break; }
  if (want_override) {
    // The diffusion UNet's plain convolutions only: at large batches that many slices are thousands of two-stage workgroups and a
    // partial-sum plane each (36 planes at 8^2: 300 MB written and read again at batch 32) - take the fewest slices that still give
    // ~8 workgroups per CU.  This makes the slice count a function of the batch; the UNet's routing already is (LDS-direct or gather
    // kernel by workgroup count, unet.hip), its samples are equal across batch sizes to rounding, not to the bit
    // (test_full_size_unet_samples_across_batch_sizes).  The synthesis network's layers (no override) keep the shape-only rule above.
    const long mn = (long)((g.M + LBM - 1) / LBM) * (g.CoV / LBN);
    for (int k = 1; k < g.ksplit; k++)
      if (g.stages_total % k == 0 && mn * k >= 2048) { g.ksplit = k; break; }
  }
  return g;
}

template <typename T>
int launch_lowres_t(hipStream_t stream, const ConvArgs& a, void* xm, float* ws, int want = 0) {
  const LowresGeom g = lowres_geom((int)sizeof(T), a.B, a.H, a.W, a.Ci, a.Co, a.up, want);
  constexpr int EPC = 16 / (int)sizeof(T);
  const long pieces = (long)g.M * (a.Ci / EPC);
  if (a.s)  // (a plain convolution has no styles: the gather reads the dense input itself)
    hipLaunchKernelGGL(lowres_premod_kernel<T>, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const T*>(a.x), a.x_bstride, a.s, reinterpret_cast<T*>(xm), a.B, a.H * a.W, a.Ci);
  hipLaunchKernelGGL(lowres_conv_kernel<T>, dim3(cdiv(g.M, LBM), g.CoV / LBN, g.ksplit), dim3(256), 0, stream,
                     reinterpret_cast<const T*>(a.s ? xm : a.x), reinterpret_cast<const T*>(a.w), ws, a.H, a.W, a.Ci, g);
  const long total = (long)g.M * (g.CoV / 4);
  hipLaunchKernelGGL(lowres_epilogue_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, ws,
                     g.ksplit);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace

bool lowres_supported(int dtype, int Ci, int Co, int up, int H, int W) {
  if (dtype != MAUA_BF16 && dtype != MAUA_F32) return false;
  const int esize = dtype == MAUA_BF16 ? 2 : 4;
  return (up == 1 || up == 2) && H * W <= 64 && Ci % (LKCB / esize) == 0 && Co % 32 == 0 && (Co * up * up) % LBN == 0;
}

void lowres_workspace(int dtype, int B, int H, int W, int Ci, int Co, int up, size_t* xm_bytes, size_t* ws_bytes) {
  const int esize = dtype == MAUA_BF16 ? 2 : 4;
  const LowresGeom g = lowres_geom(esize, B, H, W, Ci, Co, up);
  *xm_bytes = (size_t)g.M * Ci * esize;
  *ws_bytes = (size_t)g.ksplit * g.M * g.CoV * sizeof(float);
}

// The same gather GEMM for PLAIN 3x3 convolutions (no styles, a.s == NULL; dense batch-major input) of any small grid:
// the 16^2 / 8^2 levels of the diffusion UNet (unet.hip), where a per-sample tiling leaves the chip empty.
bool gather_conv_supported(int dtype, int Ci, int Co, int H, int W) {
  if (dtype != MAUA_BF16 && dtype != MAUA_F32) return false;
  const int esize = dtype == MAUA_BF16 ? 2 : 4;
  return H * W <= 1024 && Ci % (LKCB / esize) == 0 && Co % LBN == 0;
}
constexpr int GATHER_TILES = 256;
size_t gather_conv_workspace(int dtype, int B, int H, int W, int Ci, int Co) {
  const LowresGeom g = lowres_geom(dtype == MAUA_BF16 ? 2 : 4, B, H, W, Ci, Co, 1, GATHER_TILES);
  return (size_t)g.ksplit * g.M * g.CoV * sizeof(float);
}
int launch_conv_gather(hipStream_t stream, int dtype, const ConvArgs& a, float* ws) {
  MAUA_REQUIRE(gather_conv_supported(dtype, a.Ci, a.Co, a.H, a.W) && a.up == 1 && !a.s && !a.d && !a.x_pstride,
               "conv_gather: unsupported shape / arguments");
  MAUA_REQUIRE(a.x_bstride == (long)a.H * a.W * a.Ci, "conv_gather: the input must be dense batch-major NHWC");
  if (a.B == 0) return MAUA_OK;
  MAUA_REQUIRE((long)a.B * a.H * a.W * std::max(a.Ci, a.Co) < (1L << 31), "conv_gather: 32-bit pixel indices");
  if (dtype == MAUA_BF16) return launch_lowres_t<bf16_t>(stream, a, nullptr, ws, GATHER_TILES);
  return launch_lowres_t<float>(stream, a, nullptr, ws, GATHER_TILES);
}

int launch_modconv_lowres(hipStream_t stream, int dtype, const ConvArgs& a, void* xm, float* ws) {
  MAUA_REQUIRE(lowres_supported(dtype, a.Ci, a.Co, a.up, a.H, a.W), "modconv_lowres: unsupported shape");
  if (a.B == 0) return MAUA_OK;
  MAUA_REQUIRE(xm && ws && a.s, "modconv_lowres: NULL workspace / styles");
  MAUA_REQUIRE((long)a.B * a.H * a.W * std::max(a.Ci, a.Co * a.up * a.up) < (1L << 31), "modconv_lowres: 32-bit pixel indices");
  if (dtype == MAUA_BF16) return launch_lowres_t<bf16_t>(stream, a, xm, ws);
  return launch_lowres_t<float>(stream, a, xm, ws);
}

}  // namespace maua
