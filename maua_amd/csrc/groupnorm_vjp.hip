// Input gradient of the UNet's fused GroupNorm pass (unet.hip: gn_apply_kernel) - one of the pieces of guidance speed "regular",
// which differentiates the loss through the diffusion UNet itself.
//
// Replaces (reference): the autograd of maua/diffusion/processors/guided.py:258-272 (`torch.autograd.grad(img, x, img_grad)` with
// img a function of the UNet's pred_xstart, :250-252) through guided_diffusion/unet.py's GroupNorm32 -> [scale-shift] -> SiLU ->
// [Upsample | Downsample] chains (ResBlock.in_layers / h_upd / x_upd, out_layers with use_scale_shift_norm, AttentionBlock.norm,
// UNetModel.out).
//
// Forward (per sample b, group g of C / 32 channels, n = HW * C / 32 values):
//     xh = (x - mean) * rstd,  u = xh * gamma + beta,  pre = u * (1 + scale) + shift,  a = silu(pre),  y = R(a)
// with R the identity, the 2 x 2 average (mode 1) or nearest x 2 (mode 2).  Given dy:
//     da   = R^T dy                      (mode 1: dy[p / 2] / 4;  mode 2: the sum of the four dy a pixel was copied to)
//     dxh  = da * silu'(pre) * (1 + scale) * gamma
//     dx   = rstd * (dxh - mean_n(dxh) - xh * mean_n(dxh * xh))
// Two passes over x and dy: the two group means (float64 partial sums, fixed order - like the forward statistics), then dx.
// The residual branches that meet at the GroupNorm's input are added in the second pass: `dres` (the gradient of x_upd = R(x), or
// of the identity skip: through R^T as well) and `add0 | add1` (the 1x1 skip_connection's input gradients, same shape as x0 | x1).
#include "common.h"
#include "internal.h"

namespace maua {

namespace {

template <typename T>
__device__ __forceinline__ void unpack_piece(const u32x4& v, float* f) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int k = 0; k < 4; k++) { f[2 * k] = bf2f((bf16_t)(v[k] & 0xffffu)); f[2 * k + 1] = bf2f((bf16_t)(v[k] >> 16)); }
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) f[k] = __uint_as_float(v[k]);
  }
}
template <typename T>
__device__ __forceinline__ u32x4 pack_piece(const float* f) {
  u32x4 o;
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = pack2bf(f[2 * k], f[2 * k + 1]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = __float_as_uint(f[k]);
  }
  return o;
}

// R^T applied to a dense [B][Ho][Wo][C] gradient at input pixel (iy, ix), one 16-byte channel piece
template <typename T>
__device__ __forceinline__ void adjoint_piece(const T* __restrict__ g, int b, int iy, int ix, int H, int W, int mode, int C, int c,
                                              float* out) {
  constexpr int EPC = 16 / (int)sizeof(T);
  if (mode == 0) {
    unpack_piece<T>(*reinterpret_cast<const u32x4*>(g + (((long)b * H + iy) * W + ix) * C + c), out);
  } else if (mode == 1) {
    const int Ho = H / 2, Wo = W / 2;
    unpack_piece<T>(*reinterpret_cast<const u32x4*>(g + (((long)b * Ho + (iy >> 1)) * Wo + (ix >> 1)) * C + c), out);
#pragma unroll
    for (int e = 0; e < EPC; e++) out[e] *= 0.25f;
  } else {
    const int Ho = H * 2, Wo = W * 2;
    float t[EPC];
#pragma unroll
    for (int e = 0; e < EPC; e++) out[e] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      unpack_piece<T>(*reinterpret_cast<const u32x4*>(g + (((long)b * Ho + 2 * iy + (k >> 1)) * Wo + 2 * ix + (k & 1)) * C + c), t);
#pragma unroll
      for (int e = 0; e < EPC; e++) out[e] += t[e];
    }
  }
}

// Per-thread constants of one 16-byte channel piece: xh = f * rs + nmr,  pre = xh * gs + bs,  dxh = da * silu'(pre) * gs
//   rs = rstd, nmr = -mean * rstd, gs = gamma * (1 + scale), bs = beta * (1 + scale) + shift
template <typename T>
struct PieceCoef {
  static constexpr int EPC = 16 / (int)sizeof(T);
  float rs[EPC], nmr[EPC], gs[EPC], bs[EPC];
  __device__ __forceinline__ void load(const GnVjpArgs& a, int b, int C, int c) {
    const int cpg = C / 32;
#pragma unroll
    for (int q4 = 0; q4 < EPC / 4; q4++) {
      const float4 gm = *reinterpret_cast<const float4*>(a.gamma + c + 4 * q4);
      const float4 bt = *reinterpret_cast<const float4*>(a.beta + c + 4 * q4);
      const float gv[4] = {gm.x, gm.y, gm.z, gm.w}, bv[4] = {bt.x, bt.y, bt.z, bt.w};
      float sv[4] = {1.f, 1.f, 1.f, 1.f}, hv[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.ss) {
        const float4 s4 = *reinterpret_cast<const float4*>(a.ss + (long)b * a.ss_ld + c + 4 * q4);
        const float4 h4 = *reinterpret_cast<const float4*>(a.ss + (long)b * a.ss_ld + C + c + 4 * q4);
        sv[0] = 1.f + s4.x; sv[1] = 1.f + s4.y; sv[2] = 1.f + s4.z; sv[3] = 1.f + s4.w;
        hv[0] = h4.x; hv[1] = h4.y; hv[2] = h4.z; hv[3] = h4.w;
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int e = 4 * q4 + k;
        gs[e] = gv[k] * sv[k];
        bs[e] = fmaf(bv[k], sv[k], hv[k]);
      }
    }
    if (cpg % EPC == 0) {   // the piece lies inside one group (every layer of the real network)
      const float2 mr = *reinterpret_cast<const float2*>(a.stats + ((long)b * 32 + c / cpg) * 2);
#pragma unroll
      for (int e = 0; e < EPC; e++) { rs[e] = mr.y; nmr[e] = -mr.x * mr.y; }
    } else {
#pragma unroll
      for (int e = 0; e < EPC; e++) {
        const float2 mr = *reinterpret_cast<const float2*>(a.stats + ((long)b * 32 + (c + e) / cpg) * 2);
        rs[e] = mr.y; nmr[e] = -mr.x * mr.y;
      }
    }
  }
  // x piece f, adjoint-resampled upstream gradient d -> dxh, xh (bf16 networks: hardware exp / reciprocal, like the forward)
  __device__ __forceinline__ void grad(const float* f, const float* d, int silu, float* dxh, float* xh) const {
#pragma unroll
    for (int e = 0; e < EPC; e++) {
      xh[e] = fmaf(f[e], rs[e], nmr[e]);
      float dd = d[e];
      if (silu) {
        const float pre = fmaf(xh[e], gs[e], bs[e]);
        float sg;
        if constexpr (sizeof(T) == 2) sg = __fdividef(1.f, 1.f + __expf(-pre));
        else sg = 1.f / (1.f + expf(-pre));
        dd *= sg * fmaf(pre, 1.f - sg, 1.f);
      }
      dxh[e] = dd * gs[e];
    }
  }
};

// pass 1: per (sample, pixel chunk) and channel: sum dxh, sum dxh * xh; a thread's own run in float32 (at most a few hundred
// terms), everything across threads and chunks in float64 in a fixed order
template <typename T>
__global__ void gn_vjp_partial_kernel(GnVjpArgs a, int ppc, double* __restrict__ part) {
  constexpr int EPC = 16 / (int)sizeof(T);
  extern __shared__ float redf[];   // [2][blockDim.x * EPC]
  const int C = a.C0 + a.C1, PPP = C / EPC;
  const int pc = threadIdx.x % PPP, ry = threadIdx.x / PPP, RY = blockDim.x / PPP;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int c = pc * EPC;
  const long HW = (long)a.H * a.W;
  const T* src;
  long stride;
  if (c < a.C0) { src = reinterpret_cast<const T*>(a.x0) + (long)b * HW * a.C0 + c; stride = a.C0; }
  else { src = reinterpret_cast<const T*>(a.x1) + (long)b * HW * a.C1 + (c - a.C0); stride = a.C1; }
  PieceCoef<T> k;
  k.load(a, b, C, c);
  float s1[EPC], s2[EPC];
#pragma unroll
  for (int e = 0; e < EPC; e++) s1[e] = s2[e] = 0.f;
  const long p0 = (long)chunk * ppc, p1 = p0 + ppc < HW ? p0 + ppc : HW;
  for (long p = p0 + ry; p < p1; p += RY) {
    const int iy = (int)(p / a.W), ix = (int)(p - (long)iy * a.W);
    float f[EPC], d[EPC], dxh[EPC], xh[EPC];
    unpack_piece<T>(*reinterpret_cast<const u32x4*>(src + p * stride), f);
    adjoint_piece<T>(reinterpret_cast<const T*>(a.dy), b, iy, ix, a.H, a.W, a.mode, C, c, d);
    k.grad(f, d, a.silu, dxh, xh);
#pragma unroll
    for (int e = 0; e < EPC; e++) { s1[e] += dxh[e]; s2[e] = fmaf(dxh[e], xh[e], s2[e]); }
  }
  const int nt = blockDim.x;
#pragma unroll
  for (int e = 0; e < EPC; e++) { redf[(long)threadIdx.x * EPC + e] = s1[e]; redf[((long)nt + threadIdx.x) * EPC + e] = s2[e]; }
  __syncthreads();
  if (ry == 0) {
    double* dst = part + (((long)b * gridDim.x + chunk) * C + c) * 2;
#pragma unroll
    for (int e = 0; e < EPC; e++) {
      double t1 = 0.0, t2 = 0.0;
      for (int r = 0; r < RY; r++) { t1 += (double)redf[((long)r * PPP + pc) * EPC + e]; t2 += (double)redf[((long)nt + r * PPP + pc) * EPC + e]; }
      dst[2 * e] = t1;
      dst[2 * e + 1] = t2;
    }
  }
}

// pass 1b: per (sample, group): the two means
__global__ __launch_bounds__(256) void gn_vjp_finalize_kernel(const double* __restrict__ part, int rows, int C, long HW,
                                                              float* __restrict__ m) {
  __shared__ double red[2][256];
  const int g = blockIdx.x, b = blockIdx.y;
  const int cpg = C / 32;
  const long n = (long)rows * cpg;
  double s = 0.0, ss = 0.0;
  for (long i = threadIdx.x; i < n; i += 256) {
    const long row = i / cpg;
    const int c = g * cpg + (int)(i - row * cpg);
    const double* p = part + (((long)b * rows + row) * C + c) * 2;
    s += p[0];
    ss += p[1];
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = ss;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double cnt = (double)HW * cpg;
    m[((long)b * 32 + g) * 2] = (float)(red[0][0] / cnt);
    m[((long)b * 32 + g) * 2 + 1] = (float)(red[1][0] / cnt);
  }
}

// pass 2: dx = rstd * (dxh - m1 - xh * m2) + R^T dres + add.  A thread owns one 16-byte channel piece and walks VJP_PX pixels of
// its input row with it (the per-piece constants are 100+ bytes of loads: amortised like the forward's gn_apply_group_kernel)
constexpr int VJP_PX = 8;
template <typename T>
__global__ __launch_bounds__(256) void gn_vjp_apply_kernel(GnVjpArgs a, const float* __restrict__ m, int b0) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const unsigned C = a.C0 + a.C1, PPP = C / EPC, cpg = C / 32;
  const unsigned li = blockIdx.x * 256u + threadIdx.x;
  const unsigned nxg = ((unsigned)a.W + VJP_PX - 1) / VJP_PX;
  if (li >= nxg * PPP) return;
  const unsigned xg = li / PPP, pc = li - xg * PPP;
  const unsigned bl = blockIdx.y / (unsigned)a.H, iy = blockIdx.y - bl * (unsigned)a.H;
  const unsigned b = (unsigned)b0 + bl;   // (the grid's y extent holds at most 65535 rows: large batches come in sample ranges)
  const unsigned c = pc * EPC;
  const long HW = (long)a.H * a.W;
  const bool first = c < (unsigned)a.C0;
  const T* src;
  long stride;
  if (first) { src = reinterpret_cast<const T*>(a.x0) + (long)b * HW * a.C0 + c; stride = a.C0; }
  else { src = reinterpret_cast<const T*>(a.x1) + (long)b * HW * a.C1 + (c - a.C0); stride = a.C1; }
  PieceCoef<T> k;
  k.load(a, (int)b, (int)C, (int)c);
  float m1[EPC], m2[EPC];
  if (cpg % EPC == 0) {
    const float2 mm = *reinterpret_cast<const float2*>(m + ((long)b * 32 + c / cpg) * 2);
#pragma unroll
    for (int e = 0; e < EPC; e++) { m1[e] = mm.x; m2[e] = mm.y; }
  } else {
#pragma unroll
    for (int e = 0; e < EPC; e++) {
      const float2 mm = *reinterpret_cast<const float2*>(m + ((long)b * 32 + (c + e) / cpg) * 2);
      m1[e] = mm.x; m2[e] = mm.y;
    }
  }
  const T* add = reinterpret_cast<const T*>(first ? a.add0 : a.add1);
  T* dst = reinterpret_cast<T*>(first ? a.dx0 : a.dx1);
  const unsigned cs = first ? c : c - a.C0;
  // the pixel group's x pieces are requested together
  u32x4 vin[VJP_PX];
#pragma unroll
  for (int q = 0; q < VJP_PX; q++) {
    const unsigned ix = min(xg * VJP_PX + q, (unsigned)a.W - 1);
    vin[q] = *reinterpret_cast<const u32x4*>(src + ((long)iy * a.W + ix) * stride);
  }
#pragma unroll
  for (int q = 0; q < VJP_PX; q++) {
    const unsigned ix = xg * VJP_PX + q;
    if (ix >= (unsigned)a.W) break;
    float f[EPC], d[EPC], dxh[EPC], xh[EPC], out[EPC];
    unpack_piece<T>(vin[q], f);
    adjoint_piece<T>(reinterpret_cast<const T*>(a.dy), (int)b, (int)iy, (int)ix, a.H, a.W, a.mode, (int)C, (int)c, d);
    k.grad(f, d, a.silu, dxh, xh);
#pragma unroll
    for (int e = 0; e < EPC; e++) out[e] = k.rs[e] * (dxh[e] - m1[e] - xh[e] * m2[e]);
    if (a.dres) {
      float r[EPC];
      adjoint_piece<T>(reinterpret_cast<const T*>(a.dres), (int)b, (int)iy, (int)ix, a.H, a.W, a.mode, (int)C, (int)c, r);
#pragma unroll
      for (int e = 0; e < EPC; e++) out[e] += r[e];
    }
    const long off = ((long)b * HW + (long)iy * a.W + ix) * (first ? a.C0 : a.C1) + cs;
    if (add) {
      float r[EPC];
      unpack_piece<T>(*reinterpret_cast<const u32x4*>(add + off), r);
#pragma unroll
      for (int e = 0; e < EPC; e++) out[e] += r[e];
    }
    *reinterpret_cast<u32x4*>(dst + off) = pack_piece<T>(out);
  }
}

struct VjpPlan { int RY, ppc; long nchunk; };
VjpPlan vjp_plan(int C, long HW, int esize) {
  VjpPlan p;
  const int PPP = C / (16 / esize);
  p.RY = std::max(1, 256 / PPP);
  long nchunk = HW / ((long)p.RY * 4);
  nchunk = std::max(1L, std::min(128L, nchunk));
  p.ppc = (int)((HW + nchunk - 1) / nchunk);
  p.nchunk = (HW + p.ppc - 1) / p.ppc;
  return p;
}

template <typename T>
int run(hipStream_t st, const GnVjpArgs& a, void* ws) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const int C = a.C0 + a.C1, PPP = C / EPC;
  const long HW = (long)a.H * a.W;
  const VjpPlan p = vjp_plan(C, HW, (int)sizeof(T));
  double* part = reinterpret_cast<double*>(ws);
  float* m = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + (size_t)a.B * p.nchunk * C * 16);
  const int nt = PPP * p.RY;
  hipLaunchKernelGGL(gn_vjp_partial_kernel<T>, dim3((unsigned)p.nchunk, a.B), dim3(nt), (size_t)nt * EPC * 8, st, a, p.ppc, part);
  hipLaunchKernelGGL(gn_vjp_finalize_kernel, dim3(32, a.B), dim3(256), 0, st, part, (int)p.nchunk, C, HW, m);
  const unsigned nxg = (unsigned)((a.W + VJP_PX - 1) / VJP_PX);
  const int bmax = std::max(1, 65535 / a.H);
  for (int b0 = 0; b0 < a.B; b0 += bmax)
    hipLaunchKernelGGL(gn_vjp_apply_kernel<T>, dim3((nxg * (unsigned)PPP + 255) / 256, (unsigned)(std::min(bmax, a.B - b0) * a.H)), dim3(256), 0,
                       st, a, m, b0);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace

size_t group_norm_vjp_workspace(int B, int C, long HW, int esize) {
  const VjpPlan p = vjp_plan(C, HW, esize);
  return (size_t)B * p.nchunk * C * 16 + (size_t)B * 32 * 2 * 4 + 256;
}

int launch_group_norm_vjp(hipStream_t stream, int dtype, const GnVjpArgs& a, void* workspace) {
  MAUA_REQUIRE(dtype == MAUA_BF16 || dtype == MAUA_F32, "group_norm_vjp: unsupported dtype");
  const int esize = dtype == MAUA_BF16 ? 2 : 4, EPC = 16 / esize, C = a.C0 + a.C1;
  MAUA_REQUIRE(a.x0 && a.stats && a.gamma && a.beta && a.dy && a.dx0 && workspace, "group_norm_vjp: NULL argument");
  MAUA_REQUIRE(C % 32 == 0 && C / EPC <= 1024 && a.C0 % EPC == 0 && a.C1 % EPC == 0 && (a.C1 == 0 || (a.x1 && a.dx1)),
               "group_norm_vjp: C % 32 == 0, at most 1024 16-byte pieces per pixel");
  MAUA_REQUIRE(a.mode >= 0 && a.mode <= 2 && (a.mode != 1 || (a.H % 2 == 0 && a.W % 2 == 0)), "group_norm_vjp: bad resample mode");
  MAUA_REQUIRE(a.H <= 65535 && a.B <= 65535, "group_norm_vjp: grid too large");
  if (a.B == 0) return MAUA_OK;
  return dtype == MAUA_BF16 ? run<bf16_t>(stream, a, workspace) : run<float>(stream, a, workspace);
}

}  // namespace maua
