// VGG perceptors of the guided sampler's image-prompt grad modules - forward AND the gradient of their losses with respect to the
// image, walked by hand on the transposed network (the library has no autograd).
//
// Replaces (reference):
//   maua/grad.py:73-93        VGGGrads: perceptor "kbc" (maua/perceptors/vgg_kbc.py:10-71 - torchvision vgg19.features up to relu5_1, first
//                             convolution with replicate padding, ImageNet Normalize of (img + 1) / 2), the style hooks of
//                             maua/perceptors/__init__.py:33-40 (Gram matrix of relu1_1 .. relu5_1, feature_loss = scaled MSE / numel,
//                             maua/loss.py:33-80) and torch.autograd.grad(loss, img)
//   maua/grad.py:178-196      LPIPSGrads: lpips.LPIPS(net="vgg") - ScalingLayer, vgg16.features' five ReLU taps, unit-normalised features,
//                             squared difference, 1x1 "lin" layers, spatial mean, summed - and torch.autograd.grad(dist.sum() * scale, img)
//
//   forward :  img (planar f32) -> conv0 (3 -> 64: a direct VALU kernel with the input affine + Normalize folded in, zero or replicate
//              padding) -> [conv3x3 + bias + ReLU | MaxPool2d(2)] ...; every activation is kept (NHWC, network dtype)
//   heads   :  style: G_b = F_b^T F_b per image (f32, fixed-order split over pixel slices), D = G - T, loss = strength sum D^2 /
//              (sum |D| + 1e-8) / numel, dL/dG in closed form, dL/dF = F (dG + dG^T)  (one GEMM per image and tap);
//              lpips: one wave per pixel - n = F / (|F| + 1e-10), d = n - n_target, val = sum w d^2, the gradient through the
//              normalisation in closed form
//   backward:  the same graph in reverse: ReLU's gradient is the mask of the STORED activation (fused with the sum of the tap's head
//              gradient and the gradient arriving from above), MaxPool's goes to the first maximal pixel of its window (torch's rule), a
//              convolution's is the convolution with the transposed, flipped kernel (prepared at load time), conv0's a direct kernel
//              that also folds the replicate padding's adjoint and the input affine.
// MI355X design: the convolutions run on the MFMA implicit-GEMM kernel of modconv.hip (unit styles, no demodulation; float32 mode =
// exact v_mfma_f32_32x32x2_f32 products, bf16 mode = v_mfma_f32_32x32x16_bf16), the head GEMM on gemm.hip; everything else is 16-byte
// streaming kernels.  A 256 x 256 image costs 94 GFLOP forward + backward through vgg19 - 4 % of one UNet evaluation.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"
#include "internal.h"

using namespace maua;

namespace {

struct PConv {
  int Ci = 0, Co = 0;
  void *wt = nullptr, *wt_t = nullptr;   // prepared [9][Co][Ci] and the transposed network's [9][Ci][Co] (not for conv0)
  float* bias = nullptr;                 // [Co]
  float* zero_bias = nullptr;            // [max(Ci, Co)]
  float* w0 = nullptr;                   // conv0 only: [27][Co] float32, row = c * 9 + ky * 3 + kx
};

struct POp {
  int kind = 0;        // 0 conv + ReLU, 1 MaxPool2d(2)
  int conv = -1;
  int C = 0;           // output channels
  int shift = 0;       // output resolution = input >> shift
  void* act = nullptr; // kept output of the last forward [B][h][w][C]
  void* hg = nullptr;  // head gradient of a tap [B][h][w][C]
  bool hg_set = false;
};

// ---------------------------------------------------------------------------------------------- conv0 (3 -> Co), direct
// thread = (pixel, 16 output channels); x' = (img * in_mul + in_add - mean_c) * istd_c; pad: 0 zeros (of x'), 1 replicate
template <typename T>
__global__ __launch_bounds__(256) void vgg_conv0_kernel(const float* __restrict__ img, const float* __restrict__ w0, const float* __restrict__ bias,
                                                        T* __restrict__ out, int B, int H, int W, int Co, int pad, float in_mul, float in_add,
                                                        float m0, float m1, float m2, float s0, float s1, float s2) {
  extern __shared__ float wsh[];   // [27][Co]
  for (int i = threadIdx.x; i < 27 * Co; i += 256) wsh[i] = w0[i];
  __syncthreads();
  const int groups = Co / 16;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long HW = (long)H * W;
  if (idx >= (long)B * HW * groups) return;
  const int cg = (int)(idx % groups);
  long p = idx / groups;
  const int x = (int)(p % W); p /= W;
  const int y = (int)(p % H);
  const int b = (int)(p / H);
  const float mean[3] = {m0, m1, m2}, istd[3] = {s0, s1, s2};
  float in[27];
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int ky = 0; ky < 3; ky++)
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        int yy = y + ky - 1, xx = x + kx - 1;
        const bool inside = yy >= 0 && yy < H && xx >= 0 && xx < W;
        yy = min(max(yy, 0), H - 1); xx = min(max(xx, 0), W - 1);
        const float v = (img[((long)b * 3 + c) * HW + (long)yy * W + xx] * in_mul + in_add - mean[c]) * istd[c];
        in[c * 9 + ky * 3 + kx] = (inside || pad == 1) ? v : 0.f;
      }
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; j++) acc[j] = bias[cg * 16 + j];
#pragma unroll
  for (int k = 0; k < 27; k++) {
    const float* wr = wsh + k * Co + cg * 16;
#pragma unroll
    for (int j = 0; j < 16; j++) acc[j] += in[k] * wr[j];
  }
  T* o = out + (((long)b * H + y) * W + x) * Co + cg * 16;
#pragma unroll
  for (int j = 0; j < 16; j++) Elem<T>::store(o + j, fmaxf(acc[j], 0.f));
}

// d loss / d img from the (masked) gradient of conv0's output: thread = input pixel.  With replicate padding an edge pixel also
// stands in for the virtual pixels outside the image that clamp to it.
template <typename T>
__global__ __launch_bounds__(256) void vgg_conv0_vjp_kernel(const T* __restrict__ g, const float* __restrict__ w0, float* __restrict__ grad, int B,
                                                            int H, int W, int Co, int pad, float f0, float f1, float f2) {
  extern __shared__ float wsh[];   // [27][Co]
  for (int i = threadIdx.x; i < 27 * Co; i += 256) wsh[i] = w0[i];
  __syncthreads();
  const long HW = (long)H * W;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * HW) return;
  const int x = (int)(idx % W);
  const int y = (int)((idx / W) % H);
  const int b = (int)(idx / HW);
  float acc[3] = {0.f, 0.f, 0.f};
  const int vy0 = (pad == 1 && y == 0) ? -1 : y, vy1 = (pad == 1 && y == H - 1) ? H : y;
  const int vx0 = (pad == 1 && x == 0) ? -1 : x, vx1 = (pad == 1 && x == W - 1) ? W : x;
  for (int vy = vy0; vy <= vy1; vy++)
    for (int vx = vx0; vx <= vx1; vx++)
      for (int ky = 0; ky < 3; ky++) {
        const int py = vy - (ky - 1);      // the output pixel whose tap (ky, kx) reads the virtual pixel (vy, vx)
        if (py < 0 || py >= H) continue;
        for (int kx = 0; kx < 3; kx++) {
          const int px = vx - (kx - 1);
          if (px < 0 || px >= W) continue;
          const T* gp = g + (((long)b * H + py) * W + px) * Co;
          const int k = ky * 3 + kx;
          for (int co = 0; co < Co; co++) {
            const float gv = Elem<T>::load(gp + co);
            acc[0] += gv * wsh[k * Co + co];
            acc[1] += gv * wsh[(9 + k) * Co + co];
            acc[2] += gv * wsh[(18 + k) * Co + co];
          }
        }
      }
  const long o = (long)b * 3 * HW + (long)y * W + x;
  grad[o] = acc[0] * f0;
  grad[o + HW] = acc[1] * f1;
  grad[o + 2 * HW] = acc[2] * f2;
}

// the same for Co == 64 at memory speed: 8 lanes share an input pixel, each owning 8 channels (one 16-byte load per tap and
// pixel: a wave reads 8 pixels x 128 contiguous bytes), 24 multiply-adds per tap and lane, then a 3-step lane reduction
template <typename T> __device__ __forceinline__ void load8(const T* p, float v[8]);
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float v[8]) {
  const u32x4 q = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; i++) { v[2 * i] = __uint_as_float(q[i] << 16); v[2 * i + 1] = __uint_as_float(q[i] & 0xffff0000u); }
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float v[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; i++) { v[i] = a[i]; v[4 + i] = b[i]; }
}
template <typename T>
__global__ __launch_bounds__(256) void vgg_conv0_vjp64_kernel(const T* __restrict__ g, const float* __restrict__ w0, float* __restrict__ grad, int B,
                                                              int H, int W, int pad, float f0, float f1, float f2) {
  __shared__ float wsh[27 * 64];
  for (int i = threadIdx.x; i < 27 * 64; i += 256) wsh[i] = w0[i];
  __syncthreads();
  const long HW = (long)H * W;
  const int cpart = threadIdx.x & 7;
  const long idx = (long)blockIdx.x * 32 + (threadIdx.x >> 3);
  const bool live = idx < (long)B * HW;
  const long id = live ? idx : 0;
  const int x = (int)(id % W);
  const int y = (int)((id / W) % H);
  const int b = (int)(id / HW);
  float acc[3] = {0.f, 0.f, 0.f};
  const int vy0 = (pad == 1 && y == 0) ? -1 : y, vy1 = (pad == 1 && y == H - 1) ? H : y;
  const int vx0 = (pad == 1 && x == 0) ? -1 : x, vx1 = (pad == 1 && x == W - 1) ? W : x;
  for (int vy = vy0; vy <= vy1; vy++)
    for (int vx = vx0; vx <= vx1; vx++)
#pragma unroll
      for (int ky = 0; ky < 3; ky++) {
        const int py = vy - (ky - 1);
        if (py < 0 || py >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
          const int px = vx - (kx - 1);
          if (px < 0 || px >= W) continue;
          float gv[8];
          load8<T>(g + (((long)b * H + py) * W + px) * 64 + cpart * 8, gv);
          const float* wr = wsh + (ky * 3 + kx) * 64 + cpart * 8;
#pragma unroll
          for (int j = 0; j < 8; j++) {
            acc[0] = fmaf(gv[j], wr[j], acc[0]);
            acc[1] = fmaf(gv[j], wr[9 * 64 + j], acc[1]);
            acc[2] = fmaf(gv[j], wr[18 * 64 + j], acc[2]);
          }
        }
      }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {
    acc[0] += __shfl_xor(acc[0], o); acc[1] += __shfl_xor(acc[1], o); acc[2] += __shfl_xor(acc[2], o);
  }
  if (live && cpart == 0) {
    const long o = (long)b * 3 * HW + (long)y * W + x;
    grad[o] = acc[0] * f0;
    grad[o + HW] = acc[1] * f1;
    grad[o + 2 * HW] = acc[2] * f2;
  }
}

// ---------------------------------------------------------------------------------------------- MaxPool2d(2) and its adjoint
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int H, int W, int C) {
  constexpr int E = 16 / (int)sizeof(T);
  const int ppp = C / E, h2 = H / 2, w2 = W / 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * h2 * w2 * ppp) return;
  const int pc = (int)(idx % ppp);
  long p = idx / ppp;
  const int x = (int)(p % w2); p /= w2;
  const int y = (int)(p % h2);
  const int b = (int)(p / h2);
  float v[E];
#pragma unroll
  for (int e = 0; e < E; e++) v[e] = -INFINITY;
#pragma unroll
  for (int dy = 0; dy < 2; dy++)
#pragma unroll
    for (int dx = 0; dx < 2; dx++) {
      const T* s = src + (((long)b * H + 2 * y + dy) * W + 2 * x + dx) * C + pc * E;
#pragma unroll
      for (int e = 0; e < E; e++) v[e] = fmaxf(v[e], Elem<T>::load(s + e));
    }
  T* d = dst + (((long)b * h2 + y) * w2 + x) * C + pc * E;
#pragma unroll
  for (int e = 0; e < E; e++) Elem<T>::store(d + e, v[e]);
}

// g_in[2y + dy][2x + dx] = g_out[y][x] at the FIRST maximal pixel of the window (row-major scan, strict >: torch's rule), 0 elsewhere
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_vjp_kernel(const T* __restrict__ src, const T* __restrict__ gout, T* __restrict__ gin, int B, int H,
                                                           int W, int C) {
  constexpr int E = 16 / (int)sizeof(T);
  const int ppp = C / E, h2 = H / 2, w2 = W / 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * h2 * w2 * ppp) return;
  const int pc = (int)(idx % ppp);
  long p = idx / ppp;
  const int x = (int)(p % w2); p /= w2;
  const int y = (int)(p % h2);
  const int b = (int)(p / h2);
  float v[4][E];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const T* s = src + (((long)b * H + 2 * y + (q >> 1)) * W + 2 * x + (q & 1)) * C + pc * E;
#pragma unroll
    for (int e = 0; e < E; e++) v[q][e] = Elem<T>::load(s + e);
  }
  const T* gp = gout + (((long)b * h2 + y) * w2 + x) * C + pc * E;
  float go[E];
  int am[E];
#pragma unroll
  for (int e = 0; e < E; e++) {
    go[e] = Elem<T>::load(gp + e);
    int a = 0;
    float m = v[0][e];
#pragma unroll
    for (int q = 1; q < 4; q++)
      if (v[q][e] > m) { m = v[q][e]; a = q; }
    am[e] = a;
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
    T* d = gin + (((long)b * H + 2 * y + (q >> 1)) * W + 2 * x + (q & 1)) * C + pc * E;
#pragma unroll
    for (int e = 0; e < E; e++) Elem<T>::store(d + e, am[e] == q ? go[e] : 0.f);
  }
}

// out = act > 0 ? g + hg : 0   (either gradient may be NULL)
template <typename T> __device__ __forceinline__ void store8(T* p, const float v[8]);
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float v[8]) {
  u32x4 q;
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] = pack2bf(v[2 * i], v[2 * i + 1]);
  *reinterpret_cast<u32x4*>(p) = q;
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float v[8]) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
// (8 elements per thread, 16-byte accesses; n is a multiple of 64 channels)
template <typename T>
__global__ __launch_bounds__(256) void mask_add_kernel(const T* __restrict__ g, const T* __restrict__ hg, const T* __restrict__ act, T* __restrict__ out,
                                                       long n) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i >= n) return;
  float a[8], v[8], h[8];
  load8<T>(act + i, a);
#pragma unroll
  for (int e = 0; e < 8; e++) v[e] = 0.f;
  if (g) load8<T>(g + i, v);
  if (hg) {
    load8<T>(hg + i, h);
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] += h[e];
  }
#pragma unroll
  for (int e = 0; e < 8; e++) v[e] = a[e] > 0.f ? v[e] : 0.f;
  store8<T>(out + i, v);
}

// NHWC T -> planar f32 [B][C][HW]
template <typename T>
__global__ __launch_bounds__(256) void features_out_kernel(const T* __restrict__ act, float* __restrict__ out, int B, long HW, int C) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * HW * C) return;
  const int c = (int)(idx % C);
  const long p = (idx / C) % HW;
  const long b = idx / ((long)C * HW);
  out[(b * C + c) * HW + p] = Elem<T>::load(act + idx);
}

// ---------------------------------------------------------------------------------------------- style head
// partial[b][s][c1][c2] = sum over the pixels of slice s of F[p][c1] F[p][c2]; a workgroup = one 64 x 64 tile of one slice
constexpr int GRAM_PX = 256;
template <typename T>
__global__ __launch_bounds__(256) void gram_partial_kernel(const T* __restrict__ F, float* __restrict__ partial, long HW, int C, int S) {
  __shared__ float As[32][64 + 4], Bs[32][64 + 4];
  const int tiles = C / 64;
  const int t1 = blockIdx.x / tiles, t2 = blockIdx.x % tiles;
  const int s = blockIdx.y, b = blockIdx.z;
  const long p0 = (long)s * GRAM_PX, p1 = std::min<long>(p0 + GRAM_PX, HW);
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  float acc[4][4] = {};
  const T* Fb = F + (long)b * HW * C;
  for (long pp = p0; pp < p1; pp += 32) {
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
      const int r = i >> 6, c = i & 63;
      const bool ok = pp + r < p1;
      As[r][c] = ok ? Elem<T>::load(Fb + (pp + r) * C + t1 * 64 + c) : 0.f;
      Bs[r][c] = ok ? Elem<T>::load(Fb + (pp + r) * C + t2 * 64 + c) : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < 32; r++) {
      float a[4], bb[4];
#pragma unroll
      for (int i = 0; i < 4; i++) { a[i] = As[r][ty * 4 + i]; bb[i] = Bs[r][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] += a[i] * bb[j];
    }
    __syncthreads();
  }
  float* out = partial + (((long)b * S + s) * C) * C;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) out[(long)(t1 * 64 + ty * 4 + i) * C + t2 * 64 + tx * 4 + j] = acc[i][j];
}

__global__ __launch_bounds__(256) void gram_reduce_kernel(const float* __restrict__ partial, float* __restrict__ G, long CC, int S) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= CC) return;
  float v = 0.f;
  for (int s = 0; s < S; s++) v += partial[((long)b * S + s) * CC + i];
  G[(long)b * CC + i] = v;
}

__device__ __forceinline__ float block_sum_1024(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += red[i];
  return t;
}

// D = G - T per image in two launches: (1) Q = sum D^2 and A0 = sum |D| as STYLE_CHUNKS partial sums per image (fixed order),
// (2) loss[b] += strength Q / A / numel (A = A0 + 1e-8) and Sym = dG + dG^T with dG = strength (2 D A - Q sign D) / A^2 / numel,
// stored in the network dtype (the head GEMM's W operand)
constexpr int STYLE_CHUNKS = 64;
__global__ __launch_bounds__(256) void style_dsum_kernel(const float* __restrict__ G, const float* __restrict__ Tg, long t_bstride, long CC,
                                                         float* __restrict__ part) {
  __shared__ float red[4];
  const int b = blockIdx.y, ch = blockIdx.x;
  const float* g = G + (long)b * CC;
  const float* t = Tg + (long)b * t_bstride;
  const long per = (CC + STYLE_CHUNKS - 1) / STYLE_CHUNKS, i0 = ch * per, i1 = std::min<long>(i0 + per, CC);
  float q = 0.f, a = 0.f;
  for (long i = i0 + threadIdx.x; i < i1; i += 256) {
    const float d = g[i] - t[i];
    q += d * d;
    a += fabsf(d);
  }
  const float Q = block_sum_1024(q, red);
  const float A = block_sum_1024(a, red);
  if (threadIdx.x == 0) {
    part[((long)b * STYLE_CHUNKS + ch) * 2] = Q;
    part[((long)b * STYLE_CHUNKS + ch) * 2 + 1] = A;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void style_dgram_kernel(const float* __restrict__ G, const float* __restrict__ Tg, long t_bstride, int C,
                                                          float strength, const float* __restrict__ part, T* __restrict__ Sym,
                                                          float* __restrict__ loss) {
  const int b = blockIdx.y;
  const long CC = (long)C * C;
  const float* g = G + (long)b * CC;
  const float* t = Tg + (long)b * t_bstride;
  float Q = 0.f, A = 0.f;
  for (int c = 0; c < STYLE_CHUNKS; c++) { Q += part[((long)b * STYLE_CHUNKS + c) * 2]; A += part[((long)b * STYLE_CHUNKS + c) * 2 + 1]; }
  A += 1e-8f;
  const float k = strength / (float)CC;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < CC) {
    const int r = (int)(i / C), c = (int)(i - (long)r * C);
    const float d1 = g[i] - t[i], d2 = g[(long)c * C + r] - t[(long)c * C + r];
    const float s1 = d1 > 0.f ? 1.f : d1 < 0.f ? -1.f : 0.f, s2 = d2 > 0.f ? 1.f : d2 < 0.f ? -1.f : 0.f;
    Elem<T>::store(Sym + (long)b * CC + i, k * ((2.f * d1 * A - Q * s1) + (2.f * d2 * A - Q * s2)) / (A * A));
  }
  if (loss && blockIdx.x == 0 && threadIdx.x == 0) loss[b] += k * Q / A;
}

// ---------------------------------------------------------------------------------------------- lpips head: one wave per pixel
// F [B][HW][C] (T), nt = unit-normalised target features [Bt][HW][C] f32 (nt_bstride 0: shared), lin [C]
// val[b][p] = sum_c lin_c (n_c - nt_c)^2;  hg = d (coef sum_p val) / dF  (coef = scale / HW)
template <typename T>
__global__ __launch_bounds__(256) void lpips_head_kernel(const T* __restrict__ F, const float* __restrict__ nt, long nt_bstride,
                                                         const float* __restrict__ lin, float* __restrict__ val, T* __restrict__ hg, int B, long HW,
                                                         int C, float coef) {
  const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (pix >= (long)B * HW) return;
  const long b = pix / HW, p = pix - b * HW;
  const T* f = F + pix * C;
  const float* t = nt + b * nt_bstride + p * C;
  float fv[8], sq = 0.f;   // C <= 512: up to 8 channels per lane
  const int per = C / 64;
#pragma unroll 8
  for (int i = 0; i < 8; i++) {
    fv[i] = i < per ? Elem<T>::load(f + i * 64 + lane) : 0.f;
    sq += fv[i] * fv[i];
  }
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  const float r = sqrtf(sq), re = r + 1e-10f;
  float gv[8], v = 0.f, proj = 0.f;
#pragma unroll 8
  for (int i = 0; i < 8; i++) {
    gv[i] = 0.f;
    if (i < per) {
      const int c = i * 64 + lane;
      const float d = fv[i] / re - t[c];
      const float w = lin[c];
      v += w * d * d;
      gv[i] = 2.f * w * d * coef;
      proj += gv[i] * fv[i];
    }
  }
  for (int o = 32; o > 0; o >>= 1) { v += __shfl_xor(v, o); proj += __shfl_xor(proj, o); }
  if (lane == 0) val[pix] = v;
  // d (f / (r + eps)) / d f = I / (r + eps) - f f^T / (r (r + eps)^2); at r == 0 the second term is 0 (the reference's autograd: NaN)
  const float k2 = r > 0.f ? proj / (r * re * re) : 0.f;
#pragma unroll 8
  for (int i = 0; i < 8; i++)
    if (i < per) Elem<T>::store(hg + pix * C + i * 64 + lane, gv[i] / re - fv[i] * k2);
}

// n = F / (|F| + 1e-10) per pixel -> f32 NHWC
template <typename T>
__global__ __launch_bounds__(256) void lpips_norm_kernel(const T* __restrict__ F, float* __restrict__ out, long n_pix, int C) {
  const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (pix >= n_pix) return;
  float sq = 0.f;
  for (int c = lane; c < C; c += 64) { const float v = Elem<T>::load(F + pix * C + c); sq += v * v; }
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  const float re = sqrtf(sq) + 1e-10f;
  for (int c = lane; c < C; c += 64) out[pix * C + c] = Elem<T>::load(F + pix * C + c) / re;
}

__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

// dist[b] += mul * sum_p val[b][p]  (one workgroup per image, fixed order)
__global__ __launch_bounds__(1024) void rows_sum_kernel(const float* __restrict__ val, long HW, float mul, float* __restrict__ dist) {
  __shared__ float red[16];
  const int b = blockIdx.x;
  float s = 0.f;
  for (long i = threadIdx.x; i < HW; i += 1024) s += val[(long)b * HW + i];
  const float t = block_sum_1024(s, red);
  if (threadIdx.x == 0) dist[b] += mul * t;
}

}  // namespace

struct maua_vgg {
  maua_ctx* ctx = nullptr;
  unsigned long long epoch = 0;    // moves whenever a workspace is freed or reallocated: a captured graph that points into them is stale
  int dtype = MAUA_BF16;
  size_t esize = 2;
  std::vector<POp> ops;
  std::vector<PConv> convs;
  int pad0 = 0;
  int use_dma = 1;                 // bf16 convolutions on modconv_dma.hip where the shape allows (MAUA_VGG_NO_DMA=1: the generic kernel)
  float in_mul = 1.f, in_add = 0.f, mean[3] = {0, 0, 0}, istd[3] = {1, 1, 1};
  float* ones = nullptr;
  int ones_b = 0;
  int B = 0, H = 0, W = 0;         // shape of the kept forward (0: none)
  size_t cap_key = 0;              // B * H * W the buffers were sized for
  void *ga = nullptr, *gb = nullptr;
  float* fbuf = nullptr;           // gram partials / per-pixel values
  size_t fbuf_bytes = 0;
  float* gram = nullptr;           // [B][512][512]
  void* sym = nullptr;             // [B][512][512] network dtype
  float* loss_dev = nullptr;       // [B]
  float* part = nullptr;           // [B][STYLE_CHUNKS][2]
  int loss_cap = 0;
};

namespace {

void free_ws(maua_vgg* n) {
  auto f = [](void*& p) { if (p) hipFree(p); p = nullptr; };
  for (auto& o : n->ops) { f(o.act); f(o.hg); o.hg_set = false; }
  f(n->ga); f(n->gb);
  { void* p = n->fbuf; f(p); n->fbuf = nullptr; n->fbuf_bytes = 0; }
  { void* p = n->gram; f(p); n->gram = nullptr; }
  f(n->sym);
  { void* p = n->loss_dev; f(p); n->loss_dev = nullptr; n->loss_cap = 0; }
  { void* p = n->part; f(p); n->part = nullptr; }
  n->cap_key = 0;
  n->B = n->H = n->W = 0;
  n->epoch++;
}

int max_c(const maua_vgg* n) {
  int m = 0;
  for (auto& o : n->ops) m = std::max(m, o.C);
  return m;
}

int ensure_ws(maua_vgg* n, int B, int H, int W) {
  hipStream_t st = n->ctx->stream;
  const size_t key = (size_t)B * H * W;
  if (key > n->cap_key || B > n->loss_cap) {
    MAUA_HIP_CHECK(hipStreamSynchronize(st));
    free_ws(n);
    size_t widest = 0;
    for (auto& o : n->ops) {
      const size_t e = ((size_t)B * (H >> o.shift) * (W >> o.shift)) * o.C;
      widest = std::max(widest, e);
      if (hipMalloc(&o.act, e * n->esize) != hipSuccess) return fail("maua_vgg: out of device memory (activations)");
      if (hipMalloc(&o.hg, e * n->esize) != hipSuccess) return fail("maua_vgg: out of device memory (head gradients)");
    }
    const int mc = max_c(n);
    // gram partials: B * ceil(hw / 256) * C^2 floats per tap (largest at the first tap or the widest one); per-pixel values: B * H * W
    size_t fb = (size_t)B * H * W * 4;
    for (auto& o : n->ops) {
      const long hw = (long)(H >> o.shift) * (W >> o.shift);
      fb = std::max(fb, (size_t)B * ((hw + GRAM_PX - 1) / GRAM_PX) * o.C * o.C * 4);
    }
    if (hipMalloc(&n->ga, widest * n->esize) != hipSuccess || hipMalloc(&n->gb, widest * n->esize) != hipSuccess ||
        hipMalloc((void**)&n->fbuf, fb) != hipSuccess || hipMalloc((void**)&n->gram, (size_t)B * mc * mc * 4) != hipSuccess ||
        hipMalloc(&n->sym, (size_t)B * mc * mc * n->esize) != hipSuccess || hipMalloc((void**)&n->loss_dev, (size_t)B * 4) != hipSuccess ||
        hipMalloc((void**)&n->part, (size_t)B * STYLE_CHUNKS * 2 * 4) != hipSuccess)
      return fail("maua_vgg: out of device memory (workspaces)");
    n->fbuf_bytes = fb;
    n->cap_key = key;
    n->loss_cap = B;
  }
  if (B > n->ones_b) {
    MAUA_HIP_CHECK(hipStreamSynchronize(st));
    if (n->ones) hipFree(n->ones);
    n->epoch++;
    std::vector<float> h((size_t)B * 512, 1.f);
    MAUA_HIP_CHECK(hipMalloc((void**)&n->ones, h.size() * 4));
    MAUA_HIP_CHECK(hipMemcpy(n->ones, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    n->ones_b = B;
  }
  return MAUA_OK;
}

// y = act(conv3x3(x) + bias) on dense NHWC tensors; transposed: the input-gradient convolution (Co -> Ci, no bias, no activation)
int run_conv(maua_vgg* n, const PConv& c, bool transposed, const void* x, void* y, int B, int h, int w) {
  ConvArgs a{};
  const int Ci = transposed ? c.Co : c.Ci, Co = transposed ? c.Ci : c.Co;
  a.x = x; a.x_bstride = (long)h * w * Ci; a.w = transposed ? c.wt_t : c.wt; a.s = n->ones; a.d = nullptr;
  a.noise = nullptr; a.bias = transposed ? c.zero_bias : c.bias; a.y = y;
  a.B = B; a.H = h; a.W = w; a.Ci = Ci; a.Co = Co; a.up = 1;
  a.act = transposed ? MAUA_ACT_LINEAR : MAUA_ACT_LRELU; a.alpha = transposed ? 1.f : 0.f; a.gain = 1.f; a.clamp = -1.f;
  // bf16: the LDS-direct kernel where its tiles fit (8 x 32 pixels; 128 / 256-channel N tiles, or the 64-channel narrow form)
  if (n->dtype == MAUA_BF16 && n->use_dma &&
      (dma_conv_supported(MAUA_BF16, Ci, Co, 1, h, w) || dma_conv_narrow_supported(MAUA_BF16, Ci, Co, h, w))) {
    a.s = nullptr;
    return launch_modconv_dma(n->ctx->stream, a);
  }
  return launch_modconv3x3(n->ctx->stream, n->dtype, a);
}

#define VGG_LAUNCH(KERNEL, TOTAL, ...)                                                                              \
  do {                                                                                                              \
    const long total_ = (TOTAL);                                                                                    \
    if (total_ > 0) hipLaunchKernelGGL(KERNEL, dim3((unsigned)((total_ + 255) / 256)), dim3(256), 0, st, __VA_ARGS__); \
    MAUA_HIP_CHECK(hipGetLastError());                                                                              \
  } while (0)

template <typename T>
int forward_t(maua_vgg* n, const float* img, int B, int H, int W) {
  hipStream_t st = n->ctx->stream;
  if (int rc = ensure_ws(n, B, H, W)) return rc;
  constexpr int E = 16 / (int)sizeof(T);
  for (auto& o : n->ops) o.hg_set = false;
  {
    const PConv& c = n->convs[0];
    const long total = (long)B * H * W * (c.Co / 16);
    hipLaunchKernelGGL(vgg_conv0_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), (size_t)27 * c.Co * 4, st, img, c.w0, c.bias,
                       (T*)n->ops[0].act, B, H, W, c.Co, n->pad0, n->in_mul, n->in_add, n->mean[0], n->mean[1], n->mean[2], n->istd[0],
                       n->istd[1], n->istd[2]);
    MAUA_HIP_CHECK(hipGetLastError());
  }
  for (size_t i = 1; i < n->ops.size(); i++) {
    POp& o = n->ops[i];
    const POp& prev = n->ops[i - 1];
    const int h = H >> prev.shift, w = W >> prev.shift;   // input grid
    if (o.kind == 0) {
      if (int rc = run_conv(n, n->convs[o.conv], false, prev.act, o.act, B, h, w)) return rc;
    } else {
      VGG_LAUNCH(maxpool2_kernel<T>, (long)B * (h / 2) * (w / 2) * (o.C / E), (const T*)prev.act, (T*)o.act, B, h, w, o.C);
    }
  }
  n->B = B; n->H = H; n->W = W;
  return MAUA_OK;
}

// walks the kept forward backwards from the deepest tap with a head gradient; grad = d (sum of the heads' losses) / d img
template <typename T>
int backward_t(maua_vgg* n, float* grad) {
  hipStream_t st = n->ctx->stream;
  constexpr int E = 16 / (int)sizeof(T);
  const int B = n->B, H = n->H, W = n->W;
  int last = -1;
  for (int i = (int)n->ops.size() - 1; i >= 0; i--)
    if (n->ops[i].hg_set) { last = i; break; }
  MAUA_REQUIRE(last >= 0, "maua_vgg: no head gradient to back-propagate");
  T* g = nullptr;              // gradient with respect to op i's output arriving from above (NULL at the deepest tap)
  T *bufa = (T*)n->ga, *bufb = (T*)n->gb;
  for (int i = last; i >= 0; i--) {
    POp& o = n->ops[i];
    const int h = H >> o.shift, w = W >> o.shift;
    const long ne = (long)B * h * w * o.C;
    if (o.kind == 0) {
      T* gpre = (g == bufa) ? bufb : bufa;
      VGG_LAUNCH(mask_add_kernel<T>, (ne + 7) / 8, (const T*)g, (const T*)(o.hg_set ? o.hg : nullptr), (const T*)o.act, gpre, ne);
      const PConv& c = n->convs[o.conv];
      if (i == 0) {
        const long total = (long)B * H * W;
        if (c.Co == 64)
          hipLaunchKernelGGL(vgg_conv0_vjp64_kernel<T>, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, (const T*)gpre, c.w0, grad, B, H, W,
                             n->pad0, n->in_mul * n->istd[0], n->in_mul * n->istd[1], n->in_mul * n->istd[2]);
        else
          hipLaunchKernelGGL(vgg_conv0_vjp_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), (size_t)27 * c.Co * 4, st, (const T*)gpre,
                             c.w0, grad, B, H, W, c.Co, n->pad0, n->in_mul * n->istd[0], n->in_mul * n->istd[1], n->in_mul * n->istd[2]);
        MAUA_HIP_CHECK(hipGetLastError());
      } else {
        T* gprev = (gpre == bufa) ? bufb : bufa;
        if (int rc = run_conv(n, c, true, gpre, gprev, B, h, w)) return rc;
        g = gprev;
      }
    } else {
      MAUA_REQUIRE(g != nullptr, "maua_vgg: a pooling layer cannot be the deepest tap");
      const POp& prev = n->ops[i - 1];
      const int hi = H >> prev.shift, wi = W >> prev.shift;
      T* gprev = (g == bufa) ? bufb : bufa;
      VGG_LAUNCH(maxpool2_vjp_kernel<T>, (long)B * (hi / 2) * (wi / 2) * (o.C / E), (const T*)prev.act, (const T*)g, gprev, B, hi, wi, o.C);
      g = gprev;
    }
  }
  return MAUA_OK;
}

template <typename T>
int gram_t(maua_vgg* n, int op, float* G_out) {
  hipStream_t st = n->ctx->stream;
  const POp& o = n->ops[op];
  const long hw = (long)(n->H >> o.shift) * (n->W >> o.shift);
  const int S = (int)((hw + GRAM_PX - 1) / GRAM_PX), C = o.C;
  MAUA_REQUIRE((size_t)n->B * S * C * C * 4 <= n->fbuf_bytes, "maua_vgg: gram workspace too small");
  const int tiles = C / 64;
  hipLaunchKernelGGL(gram_partial_kernel<T>, dim3(tiles * tiles, S, n->B), dim3(256), 0, st, (const T*)o.act, n->fbuf, hw, C, S);
  MAUA_HIP_CHECK(hipGetLastError());
  const long CC = (long)C * C;
  hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)((CC + 255) / 256), n->B), dim3(256), 0, st, (const float*)n->fbuf, G_out, CC, S);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int check_tap(const maua_vgg* n, int op, const char* who) {
  if (op < 0 || op >= (int)n->ops.size() || n->ops[op].kind != 0) return fail(std::string(who) + ": a tap must be a convolution (+ ReLU) entry of the plan");
  if (n->ops[op].C % 64 != 0) return fail(std::string(who) + ": tap channels must be a multiple of 64");
  return MAUA_OK;
}

template <typename T>
int style_grad_t(maua_vgg* n, const float* img, int B, int H, int W, const int* taps, int n_taps, const float* const* targets,
                 const long* t_bstride, float strength, float* grad, float* loss) {
  hipStream_t st = n->ctx->stream;
  if (int rc = forward_t<T>(n, img, B, H, W)) return rc;
  hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, n->loss_dev, (long)B);   // (no memset node: capturable)
  for (int k = 0; k < n_taps; k++) {
    if (int rc = check_tap(n, taps[k], "maua_vgg_style_grad")) return rc;
    POp& o = n->ops[taps[k]];
    const int C = o.C;
    const long hw = (long)(H >> o.shift) * (W >> o.shift);
    if (int rc = gram_t<T>(n, taps[k], n->gram)) return rc;
    const long CC = (long)C * C;
    hipLaunchKernelGGL(style_dsum_kernel, dim3(STYLE_CHUNKS, B), dim3(256), 0, st, (const float*)n->gram, targets[k], t_bstride ? t_bstride[k] : 0L,
                       CC, n->part);
    MAUA_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(style_dgram_kernel<T>, dim3((unsigned)((CC + 255) / 256), B), dim3(256), 0, st, (const float*)n->gram, targets[k],
                       t_bstride ? t_bstride[k] : 0L, C, strength, (const float*)n->part, (T*)n->sym, n->loss_dev);
    MAUA_HIP_CHECK(hipGetLastError());
    {                                      // d loss / d F_b = F_b (dG_b + dG_b^T): [hw][C] x [C][C]^T, all images in one launch
      GemmArgs g{};
      g.a0 = o.act; g.lda0 = C; g.K0 = C;
      g.w = n->sym;
      g.c = o.hg; g.ldc = C;
      g.M = hw; g.N = C;
      g.batch = B; g.a_bstride = hw * C; g.w_bstride = (long)C * C; g.c_bstride = hw * C;
      if (int rc = launch_gemm_nt(st, n->dtype, g)) return rc;
    }
    o.hg_set = true;
  }
  if (int rc = backward_t<T>(n, grad)) return rc;
  if (loss) MAUA_HIP_CHECK(hipMemcpyAsync(loss, n->loss_dev, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
  return MAUA_OK;
}

template <typename T>
int lpips_grad_t(maua_vgg* n, const float* img, int B, int H, int W, const int* taps, int n_taps, const float* const* targets,
                 const long* t_bstride, const float* const* lins, float scale, float* grad, float* dist) {
  hipStream_t st = n->ctx->stream;
  if (int rc = forward_t<T>(n, img, B, H, W)) return rc;
  hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, n->loss_dev, (long)B);   // (no memset node: capturable)
  for (int k = 0; k < n_taps; k++) {
    if (int rc = check_tap(n, taps[k], "maua_vgg_lpips_grad")) return rc;
    POp& o = n->ops[taps[k]];
    MAUA_REQUIRE(o.C <= 512, "maua_vgg_lpips_grad: at most 512 channels per tap");
    const long hw = (long)(H >> o.shift) * (W >> o.shift);
    const long n_pix = (long)B * hw;
    hipLaunchKernelGGL(lpips_head_kernel<T>, dim3((unsigned)((n_pix + 3) / 4)), dim3(256), 0, st, (const T*)o.act, targets[k],
                       t_bstride ? t_bstride[k] : 0L, lins[k], n->fbuf, (T*)o.hg, B, hw, o.C, scale / (float)hw);
    MAUA_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(rows_sum_kernel, dim3(B), dim3(1024), 0, st, (const float*)n->fbuf, hw, 1.f / (float)hw, n->loss_dev);
    MAUA_HIP_CHECK(hipGetLastError());
    o.hg_set = true;
  }
  if (int rc = backward_t<T>(n, grad)) return rc;
  if (dist) MAUA_HIP_CHECK(hipMemcpyAsync(dist, n->loss_dev, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
  return MAUA_OK;
}

int check_image(const maua_vgg* n, int B, int H, int W, const char* who) {
  int pools = 0;
  for (auto& o : n->ops) pools = std::max(pools, o.shift);
  if (B < 0 || H <= 0 || W <= 0 || (H % (1 << pools)) || (W % (1 << pools)))
    return fail(std::string(who) + ": H and W must be positive multiples of " + std::to_string(1 << pools));
  return MAUA_OK;
}

}  // namespace

namespace maua {
maua_ctx* vgg_ctx(maua_vgg* n) { return n ? n->ctx : nullptr; }
unsigned long long vgg_epoch(maua_vgg* n) { return n ? n->epoch : 0; }
}

extern "C" {

int maua_vgg_create(maua_ctx* ctx, int dtype, const int* plan, int n_ops, int replicate_first, float in_mul, float in_add, const float* mean,
                    const float* std3, maua_vgg** out) {
  MAUA_REQUIRE(ctx && plan && out && mean && std3, "maua_vgg_create: NULL argument");
  MAUA_REQUIRE(dtype == MAUA_F32 || dtype == MAUA_BF16, "maua_vgg_create: dtype must be MAUA_F32 or MAUA_BF16");
  MAUA_REQUIRE(n_ops >= 1 && plan[0] > 0, "maua_vgg_create: the plan starts with a convolution");
  maua_vgg* n = new maua_vgg();
  n->ctx = ctx; n->dtype = dtype; n->esize = dtype == MAUA_BF16 ? 2 : 4;
  n->pad0 = replicate_first ? 1 : 0; n->in_mul = in_mul; n->in_add = in_add;
  if (const char* e = getenv("MAUA_VGG_NO_DMA")) n->use_dma = !(e[0] == '1');
  for (int i = 0; i < 3; i++) { n->mean[i] = mean[i]; n->istd[i] = 1.f / std3[i]; }
  int cin = 3, shift = 0;
  for (int i = 0; i < n_ops; i++) {
    POp o;
    if (plan[i] > 0) {
      if (plan[i] % 64 != 0 || plan[i] > 512) { maua_vgg_destroy(n); return fail("maua_vgg_create: channels must be multiples of 64, at most 512"); }
      o.kind = 0; o.conv = (int)n->convs.size(); o.C = plan[i];
      PConv c; c.Ci = cin; c.Co = plan[i];
      n->convs.push_back(c);
      cin = plan[i];
    } else {
      if (i > 0 && n->ops.back().kind == 1) { maua_vgg_destroy(n); return fail("maua_vgg_create: two pooling layers in a row"); }
      o.kind = 1; o.C = cin; shift++;
    }
    o.shift = shift;
    n->ops.push_back(o);
  }
  for (size_t i = 0; i < n->convs.size(); i++) {
    PConv& c = n->convs[i];
    bool ok = hipMalloc((void**)&c.bias, (size_t)c.Co * 4) == hipSuccess &&
              hipMalloc((void**)&c.zero_bias, (size_t)std::max(c.Ci, c.Co) * 4) == hipSuccess;
    if (ok) { hipMemset(c.bias, 0, (size_t)c.Co * 4); hipMemset(c.zero_bias, 0, (size_t)std::max(c.Ci, c.Co) * 4); }
    if (ok && i == 0) {
      ok = hipMalloc((void**)&c.w0, (size_t)27 * c.Co * 4) == hipSuccess;
      if (ok) hipMemset(c.w0, 0, (size_t)27 * c.Co * 4);
    } else if (ok) {
      const size_t we = (size_t)9 * c.Co * c.Ci * n->esize;
      ok = hipMalloc(&c.wt, we) == hipSuccess && hipMalloc(&c.wt_t, we) == hipSuccess;
      if (ok) { hipMemset(c.wt, 0, we); hipMemset(c.wt_t, 0, we); }
    }
    if (!ok) { maua_vgg_destroy(n); return fail("maua_vgg_create: out of device memory"); }
  }
  *out = n;
  return MAUA_OK;
}

void maua_vgg_destroy(maua_vgg* n) {
  if (!n) return;
  hipStreamSynchronize(n->ctx->stream);
  free_ws(n);
  for (auto& c : n->convs) {
    if (c.wt) hipFree(c.wt);
    if (c.wt_t) hipFree(c.wt_t);
    if (c.bias) hipFree(c.bias);
    if (c.zero_bias) hipFree(c.zero_bias);
    if (c.w0) hipFree(c.w0);
  }
  if (n->ones) hipFree(n->ones);
  delete n;
}

int maua_vgg_conv_count(maua_vgg* n) { return n ? (int)n->convs.size() : 0; }

int maua_vgg_conv_shape(maua_vgg* n, int index, int* ci, int* co) {
  MAUA_REQUIRE(n && ci && co && index >= 0 && index < (int)n->convs.size(), "maua_vgg_conv_shape: bad argument");
  *ci = n->convs[index].Ci;
  *co = n->convs[index].Co;
  return MAUA_OK;
}

// what: 0 = convolution weight [Co][Ci][3][3] (torch layout; the transposed network's layout is derived here), 1 = bias [Co]
int maua_vgg_load(maua_vgg* n, int index, int what, const float* host, size_t count) {
  MAUA_REQUIRE(n && host, "maua_vgg_load: NULL argument");
  MAUA_REQUIRE(index >= 0 && index < (int)n->convs.size(), "maua_vgg_load: no such convolution");
  hipStream_t st = n->ctx->stream;
  PConv& c = n->convs[index];
  if (what == 1) {
    MAUA_REQUIRE(count == (size_t)c.Co, "maua_vgg_load: bias: wrong size");
    MAUA_HIP_CHECK(hipStreamSynchronize(st));
    MAUA_HIP_CHECK(hipMemcpy(c.bias, host, count * 4, hipMemcpyHostToDevice));
    return MAUA_OK;
  }
  MAUA_REQUIRE(what == 0, "maua_vgg_load: what must be 0 or 1");
  MAUA_REQUIRE(count == (size_t)c.Co * c.Ci * 9, "maua_vgg_load: weight: wrong size");
  MAUA_HIP_CHECK(hipStreamSynchronize(st));
  if (index == 0) {
    std::vector<float> w0((size_t)27 * c.Co);
    for (int o = 0; o < c.Co; o++)
      for (int i = 0; i < 3; i++)
        for (int k = 0; k < 9; k++) w0[((size_t)i * 9 + k) * c.Co + o] = host[((size_t)o * 3 + i) * 9 + k];
    MAUA_HIP_CHECK(hipMemcpy(c.w0, w0.data(), w0.size() * 4, hipMemcpyHostToDevice));
    return MAUA_OK;
  }
  // the input-gradient convolution: Wt[ci][co][ky][kx] = W[co][ci][2 - ky][2 - kx]
  std::vector<float> wt(count);
  for (int o = 0; o < c.Co; o++)
    for (int i = 0; i < c.Ci; i++)
      for (int k = 0; k < 9; k++) wt[((size_t)i * c.Co + o) * 9 + k] = host[((size_t)o * c.Ci + i) * 9 + (8 - k)];
  float* tmp;
  MAUA_HIP_CHECK(hipMalloc((void**)&tmp, count * 4));
  MAUA_HIP_CHECK(hipMemcpy(tmp, host, count * 4, hipMemcpyHostToDevice));
  int rc = launch_prep_weights(st, n->dtype, tmp, c.wt, nullptr, c.Co, c.Ci, 3, 1, 0, c.Co, c.Ci);
  hipStreamSynchronize(st);
  if (!rc) {
    hipMemcpy(tmp, wt.data(), count * 4, hipMemcpyHostToDevice);
    rc = launch_prep_weights(st, n->dtype, tmp, c.wt_t, nullptr, c.Ci, c.Co, 3, 1, 0, c.Ci, c.Co);
    hipStreamSynchronize(st);
  }
  hipFree(tmp);
  return rc;
}

int maua_vgg_forward(maua_vgg* n, const float* img, int B, int H, int W) {
  MAUA_REQUIRE(n && img, "maua_vgg_forward: NULL argument");
  if (int rc = check_image(n, B, H, W, "maua_vgg_forward")) return rc;
  if (B == 0) return MAUA_OK;
  return n->dtype == MAUA_BF16 ? forward_t<bf16_t>(n, img, B, H, W) : forward_t<float>(n, img, B, H, W);
}

// the kept activation of plan entry `op` as planar float32 [B][C][h][w]
int maua_vgg_features(maua_vgg* n, int op, float* out) {
  MAUA_REQUIRE(n && out, "maua_vgg_features: NULL argument");
  MAUA_REQUIRE(n->B > 0, "maua_vgg_features: call maua_vgg_forward first");
  MAUA_REQUIRE(op >= 0 && op < (int)n->ops.size(), "maua_vgg_features: no such plan entry");
  hipStream_t st = n->ctx->stream;
  const POp& o = n->ops[op];
  const long hw = (long)(n->H >> o.shift) * (n->W >> o.shift);
  if (n->dtype == MAUA_BF16) VGG_LAUNCH(features_out_kernel<bf16_t>, (long)n->B * hw * o.C, (const bf16_t*)o.act, out, n->B, hw, o.C);
  else VGG_LAUNCH(features_out_kernel<float>, (long)n->B * hw * o.C, (const float*)o.act, out, n->B, hw, o.C);
  return MAUA_OK;
}

// Gram matrices of the kept activation of plan entry `op`: out [B][C][C] float32
int maua_vgg_gram(maua_vgg* n, int op, float* out) {
  MAUA_REQUIRE(n && out, "maua_vgg_gram: NULL argument");
  MAUA_REQUIRE(n->B > 0, "maua_vgg_gram: call maua_vgg_forward first");
  if (int rc = check_tap(n, op, "maua_vgg_gram")) return rc;
  return n->dtype == MAUA_BF16 ? gram_t<bf16_t>(n, op, out) : gram_t<float>(n, op, out);
}

// unit-normalised features of the kept activation (what lpips compares): out [B][h * w][C] float32
int maua_vgg_lpips_features(maua_vgg* n, int op, float* out) {
  MAUA_REQUIRE(n && out, "maua_vgg_lpips_features: NULL argument");
  MAUA_REQUIRE(n->B > 0, "maua_vgg_lpips_features: call maua_vgg_forward first");
  if (int rc = check_tap(n, op, "maua_vgg_lpips_features")) return rc;
  hipStream_t st = n->ctx->stream;
  const POp& o = n->ops[op];
  const long n_pix = (long)n->B * (n->H >> o.shift) * (n->W >> o.shift);
  if (n->dtype == MAUA_BF16)
    hipLaunchKernelGGL(lpips_norm_kernel<bf16_t>, dim3((unsigned)((n_pix + 3) / 4)), dim3(256), 0, st, (const bf16_t*)o.act, out, n_pix, o.C);
  else
    hipLaunchKernelGGL(lpips_norm_kernel<float>, dim3((unsigned)((n_pix + 3) / 4)), dim3(256), 0, st, (const float*)o.act, out, n_pix, o.C);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_vgg_style_grad(maua_vgg* n, const float* img, int B, int H, int W, const int* taps, int n_taps, const float* const* targets,
                        const long* target_bstride, float strength, float* grad, float* loss) {
  MAUA_REQUIRE(n && img && taps && targets && grad && n_taps > 0, "maua_vgg_style_grad: NULL argument");
  if (int rc = check_image(n, B, H, W, "maua_vgg_style_grad")) return rc;
  if (B == 0) return MAUA_OK;
  return n->dtype == MAUA_BF16 ? style_grad_t<bf16_t>(n, img, B, H, W, taps, n_taps, targets, target_bstride, strength, grad, loss)
                               : style_grad_t<float>(n, img, B, H, W, taps, n_taps, targets, target_bstride, strength, grad, loss);
}

int maua_vgg_lpips_grad(maua_vgg* n, const float* img, int B, int H, int W, const int* taps, int n_taps, const float* const* targets,
                        const long* target_bstride, const float* const* lins, float scale, float* grad, float* dist) {
  MAUA_REQUIRE(n && img && taps && targets && lins && grad && n_taps > 0, "maua_vgg_lpips_grad: NULL argument");
  if (int rc = check_image(n, B, H, W, "maua_vgg_lpips_grad")) return rc;
  if (B == 0) return MAUA_OK;
  return n->dtype == MAUA_BF16 ? lpips_grad_t<bf16_t>(n, img, B, H, W, taps, n_taps, targets, target_bstride, lins, scale, grad, dist)
                               : lpips_grad_t<float>(n, img, B, H, W, taps, n_taps, targets, target_bstride, lins, scale, grad, dist);
}

}  // extern "C"
