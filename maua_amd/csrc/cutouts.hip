// Random cutouts of an image batch, resized to the perceptor's input size, and the gradient back to the images.
//
// Replaces (reference): maua/ops/cutouts.py:8-50 `random_cutouts` / `MauaCutouts` as CLIPGrads uses them (maua/grad.py:111-116,
// 148-150: `self.normalize(self.cutouts[c](img.add(1).div(2), t))`), and the part of `torch.autograd.grad(loss, img)` (:155) that
// runs back through them.  The resize inside is `resize_right.resize(cutout, out_shape=(cut_size, cut_size))` - a package absent from
// the reference tree and the image (parity unpinned; the published algorithm: 1-D passes, cubic kernel a = -0.5, kernel stretched by
// 1 / scale when shrinking, weights normalised per output sample, zero padding around the CUTOUT).
//
// DangoCutouts (cutouts.py:101-206, skip_augs): the same kernels; a rectangle's size entry carries a grey and a mirror flag
// (internal.h CUT_GREY / CUT_FLIP): luma of the three planes in the vertical pass, mirrored store; the adjoints mirror the load and
// spread the three planes' sum over the luma coefficients.
//
// Who draws the rectangles: the host (maua_amd/grad.py restates the reference's draws from torch's generator, pinned by
// tests/golden/g33_cutouts.npz); this file takes (size, top, left) per cutout in device memory, so a captured sampler loop reads
// its step's rectangles from a table uploaded before the loop.
//
// MI355X design: everything here is HBM / LDS bound and small beside the perceptor (~2 % of a text-guided step).
//   tables   one thread per (cutout, output coordinate): first source index + <= 16 weights, the float32 arithmetic of the
//            published algorithm in its order; square cutouts -> one table serves rows and columns.
//   forward  a workgroup = one (image, channel, band of 16 output rows): vertical pass straight from the f32 image (coalesced along
//            x, affine (img + 1) / 2 applied on the fly, zeros outside the cutout) into an LDS band [16][size], horizontal pass out
//            of LDS, Normalize(mean, std), stored either as a planar f32 image or - for the perceptor - as bf16 / f32 PATCH ROWS
//            [image * patches + patch][c * p * p + ky * p + kx]: the A operand of the patch-embedding GEMM, no im2col pass.
//   gradient two gather passes, no atomics (deterministic): (1) per (image, channel, output row) the horizontal adjoint
//            th[y][X] = sum_x w[x][X - left(x)] d[y][x] / std, candidates x enumerated from the monotone left() table;
//            (2) per image pixel (b, c, Y, X) the vertical adjoint summed over the cutouts that cover the pixel, scaled by the
//            affine's 1 / 2 - accumulated into the gradient image in a fixed cutout order.
#include "common.h"
#include "internal.h"

namespace maua {

namespace {

constexpr int CT_MAXT = 16;    // weights per output sample (cubic support 4 / scale: cutouts up to 4 x the perceptor's input size)
constexpr int CT_BAND = 16;    // output rows per forward workgroup

__device__ __forceinline__ float cubic_f(float x) {   // resize_right interp_methods.cubic, float32 like the tensor arithmetic there
  const float a = fabsf(x), a2 = __fmul_rn(a, a), a3 = __fmul_rn(a2, a);
  float r = 0.f;
  if (a <= 1.f) r = __fadd_rn(__fsub_rn(__fmul_rn(1.5f, a3), __fmul_rn(2.5f, a2)), 1.f);
  else if (a <= 2.f) r = __fadd_rn(__fsub_rn(__fadd_rn(__fmul_rn(-0.5f, a3), __fmul_rn(2.5f, a2)), __fmul_rn(4.f, a)), 2.f);
  return r;
}

// one dimension of resize(out_shape = cs) for a cutout of `size` samples
__global__ void cutout_tables_kernel(const int* __restrict__ rects, int n_cut, int cs, int* __restrict__ left, float* __restrict__ wts,
                                     int* __restrict__ taps_out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_cut * cs) return;
  const int n = idx / cs, o = idx - n * cs;
  const int size = rects[3 * n] & CUT_SIZE_MASK;
  const double scale_d = (double)cs / (double)size;
  const float scale = (float)scale_d;
  float pf = (float)o / scale;
  pf = __fadd_rn(pf, (float)((size - 1) / 2.0));
  pf = __fsub_rn(pf, (float)((cs - 1) / (2.0 * scale_d)));
  const bool aa = scale_d < 1.0;
  const double support = aa ? 4.0 / scale_d : 4.0;
  const double eps_d = 1.1920928955078125e-07;
  const float lf = ceilf(__fsub_rn(__fsub_rn(pf, (float)(support / 2.0)), (float)eps_d));
  const int l = (int)lf;
  int taps = (int)ceil(support - eps_d);
  if (taps > CT_MAXT) taps = CT_MAXT;   // (refused on the host before the launch)
  float w[CT_MAXT], sum = 0.f;
  for (int k = 0; k < taps; k++) {
    const float arg = __fsub_rn(pf, (float)(l + k));
    w[k] = aa ? __fmul_rn(scale, cubic_f(__fmul_rn(scale, arg))) : cubic_f(arg);
    sum = __fadd_rn(sum, w[k]);
  }
  if (sum == 0.f) sum = 1.f;
  left[idx] = l;
  for (int k = 0; k < CT_MAXT; k++) wts[(long)idx * CT_MAXT + k] = k < taps ? w[k] / sum : 0.f;
  if (o == 0) taps_out[n] = taps;
}

struct CutArgs {
  const float* img;      // [B][3][H][W]
  const int* rects;      // [n_cut][3] (size, top, left)
  const int* left;       // [n_cut][cs]
  const float* wts;      // [n_cut][cs][CT_MAXT]
  const int* taps;       // [n_cut]
  int B, H, W, n_cut, cs;
  float mul, add;        // the affine applied to the image before the cutouts ((img + 1) / 2: 0.5, 0.5)
  float mean[3], inv_std[3];
  void* out;             // planar f32 [n_cut * B][3][cs][cs] or patch rows (T) [n_cut * B * (cs / p)^2][3 p p]
  int patch;             // 0: planar f32; p: patch rows
  int smax;              // row stride of the horizontal-adjoint buffer = the largest cutout size
};

template <typename T>
__device__ __forceinline__ void store_px(const CutArgs& a, long image, int c, int y, int x, float v) {
  if (a.patch == 0) {
    reinterpret_cast<float*>(a.out)[((image * 3 + c) * a.cs + y) * a.cs + x] = v;
  } else {
    const int p = a.patch, g = a.cs / p;
    const long row = image * g * g + (long)(y / p) * g + x / p;
    Elem<T>::store(reinterpret_cast<T*>(a.out) + row * (3 * p * p) + c * p * p + (y % p) * p + x % p, v);
  }
}
template <typename T>
__device__ __forceinline__ float load_px(const CutArgs& a, const void* d, long image, int c, int y, int x) {
  if (a.patch == 0) return reinterpret_cast<const float*>(d)[((image * 3 + c) * a.cs + y) * a.cs + x];
  const int p = a.patch, g = a.cs / p;
  const long row = image * g * g + (long)(y / p) * g + x / p;
  return Elem<T>::load(reinterpret_cast<const T*>(d) + row * (3 * p * p) + c * p * p + (y % p) * p + x % p);
}

// grid (bands, 3, n_cut * B); LDS: [CT_BAND][size] floats
template <typename T>
__global__ __launch_bounds__(256) void cutouts_fwd_kernel(CutArgs a) {
  extern __shared__ float band[];
  const int image = blockIdx.z, c = blockIdx.y, y0 = blockIdx.x * CT_BAND;
  const int n = image / a.B, b = image - n * a.B;
  const int rs = a.rects[3 * n], size = rs & CUT_SIZE_MASK, oy = a.rects[3 * n + 1], ox = a.rects[3 * n + 2];
  const bool grey = (rs & CUT_GREY) != 0, flip = (rs & CUT_FLIP) != 0;
  const int taps = a.taps[n];
  const int rows = min(CT_BAND, a.cs - y0);
  const int* lt = a.left + (long)n * a.cs;
  const float* wt = a.wts + (long)n * a.cs * CT_MAXT;
  const long plane = (long)a.H * a.W;
  const float* src = a.img + ((long)b * 3 + (grey ? 0 : c)) * plane + (long)oy * a.W + ox;
  // vertical pass (the reference resizes rows first): band[yl][X] = sum_i w[y][i] * I[left(y) + i][X]; a grey cutout reads the luma
  // of the three planes instead of its own (resize and the luma are both linear: grey before or after the resize is the same image)
  for (int e = threadIdx.x; e < rows * size; e += 256) {
    const int yl = e / size, X = e - yl * size;
    const int y = y0 + yl, l = lt[y];
    float acc = 0.f;
    for (int i = 0; i < taps; i++) {
      const int Y = l + i;
      if (Y >= 0 && Y < size) {
        const float* sp = src + (long)Y * a.W + X;
        float v = fmaf(sp[0], a.mul, a.add);
        if (grey) v = 0.2989f * v + 0.587f * fmaf(sp[plane], a.mul, a.add) + 0.114f * fmaf(sp[2 * plane], a.mul, a.add);
        acc = fmaf(wt[(long)y * CT_MAXT + i], v, acc);
      }
    }
    band[yl * size + X] = acc;
  }
  __syncthreads();
  const float mean = a.mean[c], inv_std = a.inv_std[c];
  for (int e = threadIdx.x; e < rows * a.cs; e += 256) {
    const int yl = e / a.cs, x = e - yl * a.cs;
    const int l = lt[x];
    float acc = 0.f;
    for (int j = 0; j < taps; j++) {
      const int X = l + j;
      if (X >= 0 && X < size) acc = fmaf(wt[(long)x * CT_MAXT + j], band[yl * size + X], acc);
    }
    store_px<T>(a, image, c, y0 + yl, flip ? a.cs - 1 - x : x, (acc - mean) * inv_std);
  }
}

// horizontal adjoint: th[image][c][y][X] = sum_x w[x][X - left(x)] * d[y][x] / std   (X < size; the rest of the row is not read)
// grid (bands, 3, n_cut * B); LDS: [CT_BAND][cs] gradient rows + the cutout's left() table
template <typename T>
__global__ __launch_bounds__(256) void cutouts_bwd_h_kernel(CutArgs a, const void* __restrict__ d, float* __restrict__ th) {
  extern __shared__ float sm[];
  float* rows_s = sm;                                       // [CT_BAND][cs]
  int* left_s = reinterpret_cast<int*>(sm + CT_BAND * a.cs);  // [cs]
  const int image = blockIdx.z, c = blockIdx.y, y0 = blockIdx.x * CT_BAND;
  const int n = image / a.B;
  const int size = a.rects[3 * n] & CUT_SIZE_MASK;
  const bool flip = (a.rects[3 * n] & CUT_FLIP) != 0;
  const int taps = a.taps[n];
  const int rows = min(CT_BAND, a.cs - y0);
  const float* wt = a.wts + (long)n * a.cs * CT_MAXT;
  const float inv_std = a.inv_std[c];
  for (int e = threadIdx.x; e < a.cs; e += 256) left_s[e] = a.left[(long)n * a.cs + e];
  for (int e = threadIdx.x; e < rows * a.cs; e += 256) {
    const int yl = e / a.cs, x = e - yl * a.cs;
    rows_s[e] = load_px<T>(a, d, image, c, y0 + yl, flip ? a.cs - 1 - x : x) * inv_std;
  }
  __syncthreads();
  // left() is non-decreasing in x with slope 1 / scale: the candidates of X start near (X - taps - left(0)) * cs / size
  const float inv = (float)a.cs / (float)size;
  const int l0 = left_s[0];
  for (int e = threadIdx.x; e < rows * size; e += 256) {
    const int yl = e / size, X = e - yl * size;
    int x = (int)floorf((float)(X - taps - l0) * inv) - 2;
    x = max(x, 0);
    while (x > 0 && left_s[x] + taps > X) x--;   // (the estimate may start inside the candidate run: walk back to its start)
    float acc = 0.f;
    for (; x < a.cs && left_s[x] <= X; x++) {
      const int j = X - left_s[x];
      if (j < taps) acc = fmaf(wt[(long)x * CT_MAXT + j], rows_s[yl * a.cs + x], acc);
    }
    th[(((long)image * 3 + c) * a.cs + y0 + yl) * a.smax + X] = acc;
  }
}

// vertical adjoint + sum over the cutouts: grad[b][c][Y][X] (+)= mul * sum_n sum_y w_n[y][Y - top_n - left_n(y)] * th[n, b][c][y][X - left_n]
// grid (ceil(W / 256), H, B * 3)
__global__ __launch_bounds__(256) void cutouts_bwd_v_kernel(CutArgs a, const float* __restrict__ th, float* __restrict__ grad, int accumulate) {
  const int X = blockIdx.x * 256 + threadIdx.x, Yg = blockIdx.y;
  const int bc = blockIdx.z, b = bc / 3, c = bc - b * 3;
  float acc = 0.f;
  for (int n = 0; n < a.n_cut; n++) {
    const int rs = a.rects[3 * n], size = rs & CUT_SIZE_MASK, oy = a.rects[3 * n + 1], ox = a.rects[3 * n + 2];
    const bool grey = (rs & CUT_GREY) != 0;
    const int Y = Yg - oy;
    if (Y < 0 || Y >= size) continue;     // (block-uniform)
    const int taps = a.taps[n];
    const int* lt = a.left + (long)n * a.cs;
    const float* wt = a.wts + (long)n * a.cs * CT_MAXT;
    const float inv = (float)a.cs / (float)size;
    int y = (int)floorf((float)(Y - taps - lt[0]) * inv) - 2;
    y = max(y, 0);
    while (y > 0 && lt[y] + taps > Y) y--;
    const int Xc = X - ox;
    const bool in = X < a.W && Xc >= 0 && Xc < size;
    // a grey cutout's three output planes all came from the luma: plane c receives its luma coefficient x the sum of their adjoints
    const long tplane = (long)a.cs * a.smax;
    const float* tp = th + (((long)(n * a.B + b) * 3 + (grey ? 0 : c)) * a.cs) * a.smax + Xc;
    const float lum = c == 0 ? 0.2989f : c == 1 ? 0.587f : 0.114f;
    for (; y < a.cs && lt[y] <= Y; y++) {
      const int i = Y - lt[y];
      if (i < taps && in) {
        const float* q = tp + (long)y * a.smax;
        const float tv = grey ? lum * (q[0] + q[tplane] + q[2 * tplane]) : q[0];
        acc = fmaf(wt[(long)y * CT_MAXT + i], tv, acc);
      }
    }
  }
  if (X >= a.W) return;
  float* g = grad + ((long)bc * a.H + Yg) * a.W + X;
  *g = accumulate ? *g + acc * a.mul : acc * a.mul;
}

}  // namespace

size_t cutouts_table_bytes(int n_cut, int cs) { return (size_t)n_cut * cs * (4 + 4 * CT_MAXT) + (size_t)n_cut * 4 + 256; }
size_t cutouts_th_bytes(int n_cut, int B, int cs, int smax) { return (size_t)n_cut * B * 3 * cs * smax * 4; }

static int fill(CutArgs& a, const CutoutPlan& p, void* tables) {
  MAUA_REQUIRE(p.B >= 0 && p.n_cut > 0 && p.cs > 0 && p.H > 0 && p.W > 0 && p.rects && tables, "cutouts: bad arguments");
  MAUA_REQUIRE(p.patch == 0 || p.cs % p.patch == 0, "cutouts: the cut size must be a whole number of patches");
  MAUA_REQUIRE((long)p.n_cut * p.B <= 65535, "cutouts: at most 65535 cutout images per call");
  MAUA_REQUIRE(std::min(p.H, p.W) <= 4 * p.cs - 1, "cutouts: images of more than 4 x the cut size need more than 16 filter taps");
  a.img = p.img; a.rects = p.rects; a.B = p.B; a.H = p.H; a.W = p.W; a.n_cut = p.n_cut; a.cs = p.cs; a.mul = p.mul; a.add = p.add;
  for (int c = 0; c < 3; c++) { a.mean[c] = p.mean[c]; a.inv_std[c] = 1.f / p.std[c]; }
  a.patch = p.patch; a.smax = std::min(p.H, p.W);
  char* t = (char*)tables;
  a.left = (const int*)t;
  a.wts = (const float*)(t + (size_t)p.n_cut * p.cs * 4);
  a.taps = (const int*)(t + (size_t)p.n_cut * p.cs * (4 + 4 * CT_MAXT));
  return MAUA_OK;
}

int launch_cutout_tables(hipStream_t stream, const CutoutPlan& p, void* tables) {
  CutArgs a{};
  if (int rc = fill(a, p, tables)) return rc;
  const int total = p.n_cut * p.cs;
  hipLaunchKernelGGL(cutout_tables_kernel, dim3((total + 127) / 128), dim3(128), 0, stream, p.rects, p.n_cut, p.cs, (int*)a.left,
                     (float*)a.wts, (int*)a.taps);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int launch_cutouts_forward(hipStream_t stream, int dtype, const CutoutPlan& p, void* tables, void* out) {
  CutArgs a{};
  if (int rc = fill(a, p, tables)) return rc;
  if (p.B == 0) return MAUA_OK;
  a.out = out;
  const size_t smem = (size_t)CT_BAND * a.smax * 4;
  dim3 grid((unsigned)cdiv(p.cs, CT_BAND), 3, (unsigned)(p.n_cut * p.B));
  if (dtype == MAUA_BF16) {
    MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)cutouts_fwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(cutouts_fwd_kernel<bf16_t>, grid, dim3(256), smem, stream, a);
  } else {
    MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)cutouts_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(cutouts_fwd_kernel<float>, grid, dim3(256), smem, stream, a);
  }
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int launch_cutouts_vjp(hipStream_t stream, int dtype, const CutoutPlan& p, void* tables, const void* d_out, float* th, float* grad,
                       int accumulate) {
  CutArgs a{};
  if (int rc = fill(a, p, tables)) return rc;
  if (p.B == 0) return MAUA_OK;
  const size_t smem = (size_t)CT_BAND * p.cs * 4 + (size_t)p.cs * 4;
  dim3 grid((unsigned)cdiv(p.cs, CT_BAND), 3, (unsigned)(p.n_cut * p.B));
  if (dtype == MAUA_BF16) {
    MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)cutouts_bwd_h_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(cutouts_bwd_h_kernel<bf16_t>, grid, dim3(256), smem, stream, a, d_out, th);
  } else {
    MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)cutouts_bwd_h_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(cutouts_bwd_h_kernel<float>, grid, dim3(256), smem, stream, a, d_out, th);
  }
  MAUA_REQUIRE((long)p.B * 3 <= 65535 && p.H <= 65535, "cutouts: image batch / height too large for one launch");
  hipLaunchKernelGGL(cutouts_bwd_v_kernel, dim3((unsigned)cdiv(p.W, 256), (unsigned)p.H, (unsigned)(p.B * 3)), dim3(256), 0, stream, a, th,
                     grad, accumulate);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace maua
