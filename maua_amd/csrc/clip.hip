// Text-prompt guidance: the CLIP image tower forward, its gradient back to the images, and CLIPGrads' loss around it.
//
// Replaces (reference): maua/grad.py:96-165 `CLIPGrads` - per sampler step, `cutout_batches` times: cutouts of the current image
// estimate (cutouts.hip), `clip_model.encode_image` (OpenAI CLIP's VisionTransformer, clip/model.py - a pip dependency that is absent
// from the reference tree and this image: published architecture restated, **parity unpinned**), `spherical_dist_loss` to the target
// embeddings (maua/loss.py:22-25, pinned by g33), the weighted sum over prompts / mean over cutouts, and `torch.autograd.grad(loss, img)`.
// The library has no autograd: the gradient is the network walked backwards by hand, activations only (the perceptor's weights are
// frozen, `requires_grad_(False)`, :106).
//
//   forward   patch rows [N * G^2][3 p p]  --GEMM conv1-->  [N * G^2][w]  -> tokens [N][T = G^2 + 1][w] (class token, + positional
//             embedding) -> ln_pre -> L x { x += out_proj(attention(in_proj(ln_1(x))));  x += c_proj(QuickGELU(c_fc(ln_2(x)))) }
//             -> ln_post(class token) @ proj -> embedding [N][E]
//   head      per image, one workgroup, float32: ln_post, projection, F.normalize, distance 2 asin(|e - y| / 2)^2 to each target,
//             the weighted loss, and the whole way back to the class token's gradient (loss -> e -> ln_post -> x[0])
//   backward  the same chain in reverse on the transposed weights (prepared at load): every Linear's input gradient is a GEMM against
//             W^T, LayerNorm / QuickGELU have element-wise input-gradient kernels, attention the flash-style backward of
//             attention_vjp.hip (P rebuilt from the forward's log-sum-exp rows), residuals add.
//
// MI355X design.  A text-guided step is ~19 TFLOP per sample (256 cutouts x (35 forward + 37 backward) GFLOP) against 2.2 for the
// diffusion UNet, 95 % of it in eight plain GEMMs per layer with M = N * 197 ~ 200 000 rows: those run on gemm_dma.hip's LDS-direct
// 256 x 128 tiles; the in-projection's rows are permuted at load into the head-major [q | k | v] layout attention.hip reads, so the
// fused attention kernels of the diffusion UNet serve unchanged (T = 197 is masked in their last key block).  Activations are kept
// for the way back instead of recomputed (10 x [M][w] per layer: 37 GB at 1024 images - sized for 288 GB of HBM, not for 80);
// QuickGELU and its derivative ride on the GEMM epilogues (GemmArgs.epi) where the LDS-direct kernel takes the shape, and are
// separate 16-byte streaming kernels elsewhere (f32 parity mode, narrow test towers).  LayerNorm: one wave per token, values held in
// registers between the two statistics passes, float32 statistics kept for the gradient.
#include <atomic>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "internal.h"

using namespace maua;

namespace maua {
namespace {

// ------------------------------------------------------------------------------------------------ element-wise kernels
template <typename T> struct Pc;   // a 16-byte piece <-> floats
template <> struct Pc<bf16_t> {
  static constexpr int N = 8;
  __device__ static __forceinline__ void load(const bf16_t* p, float* v) {
    const u32x4 u = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int k = 0; k < 4; k++) { v[2 * k] = bf2f((bf16_t)(u[k] & 0xffffu)); v[2 * k + 1] = __uint_as_float(u[k] & 0xffff0000u); }
  }
  __device__ static __forceinline__ void store(bf16_t* p, const float* v) {
    u32x4 u;
#pragma unroll
    for (int k = 0; k < 4; k++) u[k] = pack2bf(v[2 * k], v[2 * k + 1]);
    *reinterpret_cast<u32x4*>(p) = u;
  }
};
template <> struct Pc<float> {
  static constexpr int N = 4;
  __device__ static __forceinline__ void load(const float* p, float* v) {
    const float4 u = *reinterpret_cast<const float4*>(p);
    v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
  }
  __device__ static __forceinline__ void store(float* p, const float* v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float sigmoid_f(float x, bool exact) { return 1.f / (1.f + (exact ? expf(-x) : __expf(-x))); }

constexpr int LN_MAXP = 8;   // 16-byte pieces per lane: C <= 64 * 8 * (16 / sizeof(T))

// y = LayerNorm(x) * g + b per row of C values (float32 statistics, eps 1e-5: clip/model.py LayerNorm); stats[row] = (mean, rstd)
// one wave per row, 4 rows per workgroup
template <typename T>
__global__ __launch_bounds__(256) void layer_norm_kernel(const T* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                         T* __restrict__ y, float* __restrict__ stats, long rows, int C) {
  constexpr int E = Pc<T>::N;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int pieces = C / E;
  float v[LN_MAXP][E];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXP; i++) {
    const int p = lane + 64 * i;
    if (p < pieces) {
      Pc<T>::load(x + row * C + (long)p * E, v[i]);
#pragma unroll
      for (int e = 0; e < E; e++) s += v[i][e];
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXP; i++)
    if (lane + 64 * i < pieces) {
#pragma unroll
      for (int e = 0; e < E; e++) { const float d = v[i][e] - mean; q = fmaf(d, d, q); }
    }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + 1e-5f);
  if (stats && lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
#pragma unroll
  for (int i = 0; i < LN_MAXP; i++) {
    const int p = lane + 64 * i;
    if (p < pieces) {
      float o[E];
#pragma unroll
      for (int e = 0; e < E; e++) o[e] = fmaf((v[i][e] - mean) * rstd, g[p * E + e], b[p * E + e]);
      Pc<T>::store(y + row * C + (long)p * E, o);
    }
  }
}

// dx = rstd * (g dy - mean(g dy) - xhat mean(g dy xhat)) (+ add): the input gradient of the kernel above.  drop_T > 0: rows are tokens
// [image][drop_T]; the class token's row (t == 0) is not written and the others are stored densely [image][drop_T - 1] (ln_pre: what
// flows on to the patch embedding)
template <typename T>
__global__ __launch_bounds__(256) void layer_norm_vjp_kernel(const T* __restrict__ x, const float* __restrict__ stats,
                                                             const float* __restrict__ g, const T* __restrict__ dy,
                                                             const T* __restrict__ add, T* __restrict__ dx, long rows, int C, int drop_T) {
  constexpr int E = Pc<T>::N;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  long orow = row;
  if (drop_T > 0) {
    const long im = row / drop_T;
    const int t = (int)(row - im * drop_T);
    if (t == 0) return;
    orow = im * (drop_T - 1) + t - 1;
  }
  const int pieces = C / E;
  const float mean = stats[2 * row], rstd = stats[2 * row + 1];
  float xh[LN_MAXP][E], gd[LN_MAXP][E];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXP; i++) {
    const int p = lane + 64 * i;
    if (p < pieces) {
      float xv[E], dv[E];
      Pc<T>::load(x + row * C + (long)p * E, xv);
      Pc<T>::load(dy + row * C + (long)p * E, dv);
#pragma unroll
      for (int e = 0; e < E; e++) {
        xh[i][e] = (xv[e] - mean) * rstd;
        gd[i][e] = dv[e] * g[p * E + e];
        s1 += gd[i][e];
        s2 = fmaf(gd[i][e], xh[i][e], s2);
      }
    }
  }
  const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
#pragma unroll
  for (int i = 0; i < LN_MAXP; i++) {
    const int p = lane + 64 * i;
    if (p < pieces) {
      float o[E];
#pragma unroll
      for (int e = 0; e < E; e++) o[e] = rstd * (gd[i][e] - m1 - xh[i][e] * m2);
      if (add) {
        float av[E];
        Pc<T>::load(add + row * C + (long)p * E, av);
#pragma unroll
        for (int e = 0; e < E; e++) o[e] += av[e];
      }
      Pc<T>::store(dx + orow * C + (long)p * E, o);
    }
  }
}

// a = h * sigmoid(1.702 h)   |   dh = da * (s + 1.702 h s (1 - s))
template <typename T>
__global__ __launch_bounds__(256) void quick_gelu_kernel(const T* __restrict__ h, T* __restrict__ a, long pieces) {
  constexpr int E = Pc<T>::N;
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= pieces) return;
  float v[E];
  Pc<T>::load(h + p * E, v);
#pragma unroll
  for (int e = 0; e < E; e++) v[e] = v[e] * sigmoid_f(1.702f * v[e], sizeof(T) == 4);
  Pc<T>::store(a + p * E, v);
}
template <typename T>
__global__ __launch_bounds__(256) void quick_gelu_vjp_kernel(const T* __restrict__ h, const T* da, T* dh, long pieces) {   // (dh may be da)
  constexpr int E = Pc<T>::N;
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= pieces) return;
  float v[E], d[E];
  Pc<T>::load(h + p * E, v);
  Pc<T>::load(da + p * E, d);
#pragma unroll
  for (int e = 0; e < E; e++) {
    const float s = sigmoid_f(1.702f * v[e], sizeof(T) == 4);
    d[e] *= s * (1.f + 1.702f * v[e] * (1.f - s));
  }
  Pc<T>::store(dh + p * E, d);
}

// tok[n][t][c] = (t == 0 ? class_embedding[c] : pe[n][t - 1][c]) + positional_embedding[t][c]
template <typename T>
__global__ __launch_bounds__(256) void tokens_kernel(const T* __restrict__ pe, const float* __restrict__ cls, const float* __restrict__ pos,
                                                     T* __restrict__ tok, long N, int Tk, int C) {
  constexpr int E = Pc<T>::N;
  const int ppr = C / E;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * Tk * ppr) return;
  const int pc = (int)(idx % ppr);
  const long nt = idx / ppr;
  const int t = (int)(nt % Tk);
  const long n = nt / Tk;
  float v[E];
  if (t == 0) {
#pragma unroll
    for (int e = 0; e < E; e++) v[e] = cls[pc * E + e];
  } else {
    Pc<T>::load(pe + (n * (Tk - 1) + t - 1) * C + (long)pc * E, v);
  }
#pragma unroll
  for (int e = 0; e < E; e++) v[e] += pos[(long)t * C + pc * E + e];
  Pc<T>::store(tok + nt * C + (long)pc * E, v);
}

// planar f32 images [N][3][R][R] <-> patch rows [N * G * G][3 p p] (k = c p p + ky p + kx: conv1.weight flattened); the patch
// convolution has stride = kernel, so this is a permutation and its transpose
template <typename T>
__global__ __launch_bounds__(256) void im2patch_kernel(const float* __restrict__ img, T* __restrict__ rows, long N, int R, int p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * 3 * R * R) return;
  const int x = (int)(idx % R), y = (int)((idx / R) % R), c = (int)((idx / ((long)R * R)) % 3);
  const long n = idx / ((long)3 * R * R);
  const int G = R / p;
  Elem<T>::store(rows + (n * G * G + (long)(y / p) * G + x / p) * (3 * p * p) + c * p * p + (y % p) * p + x % p, img[idx]);
}
template <typename T>
__global__ __launch_bounds__(256) void patch2im_kernel(const T* __restrict__ rows, float* __restrict__ img, long N, int R, int p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * 3 * R * R) return;
  const int x = (int)(idx % R), y = (int)((idx / R) % R), c = (int)((idx / ((long)R * R)) % 3);
  const long n = idx / ((long)3 * R * R);
  const int G = R / p;
  img[idx] = Elem<T>::load(rows + (n * G * G + (long)(y / p) * G + x / p) * (3 * p * p) + c * p * p + (y % p) * p + x % p);
}

// ------------------------------------------------------------------------------------------------ the head (per image, float32)
struct HeadArgs {
  const void* x;            // [N][Tk][w] (T): the last block's output; row 0 of each image is read
  const float *g, *b;       // ln_post
  const float* proj;        // [w][E]
  const float* tgt;         // [S][P][E] unit vectors
  const float* twt;         // [S][P] normalised prompt weights
  const int* sel;           // [B] target set of each sample, or NULL (set 0)
  float* embed;             // [N][E] or NULL
  float* loss;              // [N] or NULL: sum_p w_p dist_p of this image
  const float* d_embed;     // [N][E] or NULL: an upstream gradient instead of the loss's (maua_clip_encode_image_vjp)
  void* dx;                 // [N][Tk][w] (T) or NULL: row 0 of each image receives the gradient (the other rows are the caller's to zero)
  int Tk, w, E, P, B;
  float coef;               // d total / d (sum_p w_p dist_p) of one image: scale / cutn / cutout_batches
  const float* mult;        // [n_cut] or NULL: how many identical cutouts each image of cutout n stands for (coef is multiplied by it)
};

__device__ __forceinline__ float block_sum(float v, float* red) {   // 256 threads; red: 4 floats of LDS
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

template <typename T>
__global__ __launch_bounds__(256) void clip_head_kernel(HeadArgs a) {
  extern __shared__ float sm[];
  float* z = sm;               // [w] ln_post output, later d z
  float* xh = z + a.w;         // [w] normalised input
  float* e = xh + a.w;         // [E] embedding, later unit embedding
  float* de = e + a.E;         // [E]
  float* gp = de + a.E;        // [P] per-target factors
  float* red = gp + a.P;       // [4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long n = blockIdx.x;
  const T* xr = reinterpret_cast<const T*>(a.x) + n * a.Tk * a.w;
  float s = 0.f;
  for (int c = tid; c < a.w; c += 256) { const float v = Elem<T>::load(xr + c); xh[c] = v; s += v; }
  const float mean = block_sum(s, red) / (float)a.w;
  float q = 0.f;
  for (int c = tid; c < a.w; c += 256) { const float d = xh[c] - mean; q = fmaf(d, d, q); }
  const float rstd = rsqrtf(block_sum(q, red) / (float)a.w + 1e-5f);
  for (int c = tid; c < a.w; c += 256) {
    xh[c] = (xh[c] - mean) * rstd;
    z[c] = fmaf(xh[c], a.g[c], a.b[c]);
  }
  __syncthreads();
  for (int j = tid; j < a.E; j += 256) {
    float acc = 0.f;
    for (int c = 0; c < a.w; c++) acc = fmaf(z[c], a.proj[(long)c * a.E + j], acc);
    e[j] = acc;
    if (a.embed) a.embed[n * a.E + j] = acc;
  }
  __syncthreads();
  if (!a.dx) return;
  if (a.d_embed) {
    for (int j = tid; j < a.E; j += 256) de[j] = a.d_embed[n * a.E + j];
    __syncthreads();
  } else {
    // F.normalize (eps 1e-12), distances, and d loss / d e
    float ss = 0.f;
    for (int j = tid; j < a.E; j += 256) ss = fmaf(e[j], e[j], ss);
    const float ne = fmaxf(sqrtf(block_sum(ss, red)), 1e-12f);
    for (int j = tid; j < a.E; j += 256) e[j] /= ne;
    __syncthreads();
    const int set = a.sel ? a.sel[n % a.B] : 0;
    const float* tg = a.tgt + (long)set * a.P * a.E;
    const float* tw = a.twt + (long)set * a.P;
    float lsum = 0.f;
    for (int p = wave; p < a.P; p += 4) {
      float u2 = 0.f;
      for (int j = lane; j < a.E; j += 64) { const float d = e[j] - tg[(long)p * a.E + j]; u2 = fmaf(d, d, u2); }
      u2 = wave_sum(u2);
      const float u = sqrtf(u2), hu = fminf(0.5f * u, 1.f);
      const float as = asinf(hu);
      // d/du [2 asin(u / 2)^2] = 2 asin(u / 2) / sqrt(1 - u^2 / 4); d u / d e = (e - y) / u
      const float f = u > 1e-20f ? tw[p] * 2.f * as / (sqrtf(fmaxf(1.f - hu * hu, 1e-12f)) * u) : 0.f;
      if (lane == 0) gp[p] = f;
      lsum += tw[p] * 2.f * as * as;
    }
    __syncthreads();
    float lw = (lane == 0) ? lsum : 0.f;   // (a wave's lanes hold the same partial sum over the wave's targets)
    const float ltot = block_sum(lw, red);
    if (a.loss && tid == 0) a.loss[n] = ltot;
    float gsum = 0.f;
    for (int p = 0; p < a.P; p++) gsum += gp[p];
    float dot = 0.f;
    for (int j = tid; j < a.E; j += 256) {
      float acc = 0.f;
      for (int p = 0; p < a.P; p++) acc = fmaf(gp[p], tg[(long)p * a.E + j], acc);
      const float dj = a.coef * (a.mult ? a.mult[n / a.B] : 1.f) * (e[j] * gsum - acc);   // d / d ehat
      de[j] = dj;
      dot = fmaf(dj, e[j], dot);
    }
    dot = block_sum(dot, red);
    for (int j = tid; j < a.E; j += 256) de[j] = (de[j] - e[j] * dot) / ne;   // through x / |x|
    __syncthreads();
  }
  // d z = proj de (one wave per row of proj), then ln_post backwards
  for (int c = wave; c < a.w; c += 4) {
    float acc = 0.f;
    for (int j = lane; j < a.E; j += 64) acc = fmaf(a.proj[(long)c * a.E + j], de[j], acc);
    acc = wave_sum(acc);
    if (lane == 0) z[c] = acc * a.g[c];    // g dy
  }
  __syncthreads();
  float s1 = 0.f, s2 = 0.f;
  for (int c = tid; c < a.w; c += 256) { s1 += z[c]; s2 = fmaf(z[c], xh[c], s2); }
  const float m1 = block_sum(s1, red) / (float)a.w;
  const float m2 = block_sum(s2, red) / (float)a.w;
  T* dr = reinterpret_cast<T*>(a.dx) + n * a.Tk * a.w;
  for (int c = tid; c < a.w; c += 256) Elem<T>::store(dr + c, rstd * (z[c] - m1 - xh[c] * m2));
}

__global__ __launch_bounds__(256) void zero16_kernel(u32x4* __restrict__ p, long pieces) {   // (a kernel, not a memset node: the
  const long i = (long)blockIdx.x * 256 + threadIdx.x;                                        //  guided loop is captured as a hipGraph)
  if (i < pieces) p[i] = u32x4{0u, 0u, 0u, 0u};
}

// clamp_gradient (grad.py:156-158): magnitude = sqrt(mean(grad^2)); grad *= min(magnitude, clamp) / magnitude - two launches, fixed
// partial layout
__global__ __launch_bounds__(256) void sq_partial_kernel(const float* __restrict__ g, long n, double* __restrict__ part) {
  __shared__ double red[256];
  double s = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += (double)g[i] * g[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void clamp_scale_kernel(float* __restrict__ g, long n, const double* __restrict__ part, int nparts, float clamp,
                                                          float scale) {
  // the partial sums once per workgroup, in a fixed order (every thread used to walk all 1024 of them: 0.86 ms for a 6 M-value gradient)
  __shared__ double red[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const double s = red[0];
  float f = scale;
  if (clamp > 0.f) {
    const float mag = sqrtf((float)(s / (double)n)) * fabsf(scale);
    if (mag > clamp) f *= clamp / mag;
  }
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) g[i] = (s != s) ? 0.f : g[i] * f;   // (a gradient holding a NaN counts as zeros: guided.py:262-265)
}

struct Layer {
  void *w_qkv = nullptr, *w_qkv_t = nullptr, *w_out = nullptr, *w_out_t = nullptr, *w_fc = nullptr, *w_fc_t = nullptr, *w_pr = nullptr,
       *w_pr_t = nullptr;
  float *b_qkv = nullptr, *b_out = nullptr, *b_fc = nullptr, *b_pr = nullptr, *ln1_g = nullptr, *ln1_b = nullptr, *ln2_g = nullptr,
        *ln2_b = nullptr;
  // kept by the forward for the way back
  void *x = nullptr, *qkv = nullptr, *ao = nullptr, *xm = nullptr, *h = nullptr;
  float *st1 = nullptr, *st2 = nullptr, *lse = nullptr;
};

}  // namespace
}  // namespace maua

struct maua_clip {
  maua_ctx* ctx;
  int dtype;
  size_t esize;
  int res, patch, width, layers, heads, E, G, Tk, kp;   // G = res / patch, Tk = G * G + 1 tokens, kp = 3 * patch^2
  void *w_conv = nullptr, *w_conv_t = nullptr;          // [w][kp], [kp][w]
  float *cls = nullptr, *pos = nullptr, *lnpre_g = nullptr, *lnpre_b = nullptr, *lnpost_g = nullptr, *lnpost_b = nullptr, *proj = nullptr;
  std::vector<maua::Layer> L;
  // workspaces for `cap` images
  long cap = 0;
  int keep = 0;            // sized with the kept activations of a gradient pass
  void *patches = nullptr, *pe = nullptr, *tok = nullptr, *x_last = nullptr, *lno = nullptr, *act = nullptr;
  void *dxa = nullptr, *dxb = nullptr, *dqkv = nullptr, *dwide = nullptr, *dtmp = nullptr;
  float *st_pre = nullptr, *delta = nullptr, *embed = nullptr, *loss = nullptr;
  long kept_N = 0;         // images of the last kept forward (0: none)
  // targets
  float *tgt = nullptr, *twt = nullptr;
  int S = 0, P = 0;
  int* sel = nullptr; int sel_cap = 0; int sel_B = 0;
  // cutout scratch
  void* cut_tables = nullptr; size_t cut_tables_bytes = 0;
  float* cut_th = nullptr; size_t cut_th_bytes = 0;
  int* rects_dev = nullptr; size_t rects_cap = 0;   // + the multiplicities behind the rectangles (floats)
  double* parts = nullptr;
  unsigned long long uid = 0, epoch = 0;   // identity of this tower / generation of its buffers (a captured graph holds pointers into
};                                          // them: unet.hip compares both before a replay)

namespace maua {
namespace {

constexpr int NPARTS = 1024;

int dalloc(void** p, size_t bytes) {
  MAUA_HIP_CHECK(hipMalloc(p, bytes ? bytes : 16));
  return MAUA_OK;
}
void dfree(void* p) { if (p) hipFree(p); }

void free_ws(maua_clip* n) {
  for (void** p : {&n->patches, &n->pe, &n->tok, &n->x_last, &n->lno, &n->act, &n->dxa, &n->dxb, &n->dqkv, &n->dwide, &n->dtmp,
                   (void**)&n->st_pre, (void**)&n->delta, (void**)&n->embed, (void**)&n->loss}) {
    dfree(*p);
    *p = nullptr;
  }
  for (auto& l : n->L)
    for (void** p : {&l.x, &l.qkv, &l.ao, &l.xm, &l.h, (void**)&l.st1, (void**)&l.st2, (void**)&l.lse}) {
      dfree(*p);
      *p = nullptr;
    }
  n->cap = 0; n->keep = 0; n->kept_N = 0;
  n->epoch++;
}

// workspaces for N images; keep: every layer's activations stay (a gradient pass follows), otherwise the layers share one set
int ensure_ws(maua_clip* n, long N, int keep) {
  if (N <= n->cap && keep <= n->keep) return MAUA_OK;
  MAUA_HIP_CHECK(hipStreamSynchronize(n->ctx->stream));
  N = std::max(N, n->cap);
  keep = std::max(keep, n->keep);
  free_ws(n);
  const size_t es = n->esize, w = n->width, M = (size_t)N * n->Tk, Mp = (size_t)N * n->G * n->G;
  int rc = dalloc(&n->patches, Mp * n->kp * es);
  if (!rc) rc = dalloc(&n->pe, Mp * w * es);
  if (!rc) rc = dalloc(&n->tok, M * w * es);
  if (!rc) rc = dalloc(&n->x_last, M * w * es);
  if (!rc) rc = dalloc(&n->lno, M * w * es);
  if (!rc) rc = dalloc(&n->act, M * 4 * w * es);
  if (!rc) rc = dalloc((void**)&n->st_pre, M * 2 * 4);
  if (!rc) rc = dalloc((void**)&n->embed, (size_t)N * n->E * 4);
  if (!rc) rc = dalloc((void**)&n->loss, (size_t)N * 4);
  if (!rc && keep) {
    rc = dalloc(&n->dxa, M * w * es);
    if (!rc) rc = dalloc(&n->dxb, M * w * es);
    if (!rc) rc = dalloc(&n->dqkv, M * 3 * w * es);
    if (!rc) rc = dalloc(&n->dwide, M * 4 * w * es);
    if (!rc) rc = dalloc(&n->dtmp, M * w * es);
    if (!rc) rc = dalloc((void**)&n->delta, (size_t)N * n->heads * n->Tk * 4);
  }
  const int nl = keep ? n->layers : 1;
  for (int i = 0; i < nl && !rc; i++) {
    Layer& l = n->L[i];
    rc = dalloc(&l.x, M * w * es);
    if (!rc) rc = dalloc(&l.qkv, M * 3 * w * es);
    if (!rc) rc = dalloc(&l.ao, M * w * es);
    if (!rc) rc = dalloc(&l.xm, M * w * es);
    if (!rc) rc = dalloc(&l.h, M * 4 * w * es);
    if (!rc) rc = dalloc((void**)&l.st1, M * 2 * 4);
    if (!rc) rc = dalloc((void**)&l.st2, M * 2 * 4);
    if (!rc) rc = dalloc((void**)&l.lse, (size_t)N * n->heads * n->Tk * 4);
  }
  if (rc) { free_ws(n); return fail("maua_clip: out of device memory for the image tower's activations"); }
  n->cap = N; n->keep = keep;
  return MAUA_OK;
}

int gemm(maua_clip* n, const void* a, long M, int K, const void* w, int N, const float* bias, const void* res, void* c, int epi = 0,
         void* c2 = nullptr, const void* aux = nullptr) {
  GemmArgs g{};
  g.a0 = a; g.lda0 = K; g.K0 = K; g.w = w; g.bias = bias; g.res = res; g.ldr = N; g.c = c; g.ldc = N; g.M = M; g.N = N;
  g.prefer_dma = n->ctx->gemm_dma;
  if (epi && n->ctx->gemm_dma) {
    GemmArgs f = g;
    f.epi = epi; f.c2 = c2; f.ldc2 = N; f.aux = aux; f.ldaux = N;
    if (gemm_dma_supported(n->dtype, f)) return launch_gemm_nt(n->ctx->stream, n->dtype, f);
  }
  return launch_gemm_nt(n->ctx->stream, n->dtype, g);
}
bool gemm_fuses(maua_clip* n, long M, int K, int N) {
  GemmArgs g{};
  int dummy;
  g.a0 = &dummy; g.lda0 = K; g.K0 = K; g.w = &dummy; g.c = &dummy; g.ldc = N; g.M = M; g.N = N; g.epi = 1; g.c2 = &dummy; g.ldc2 = N;
  return n->ctx->gemm_dma && gemm_dma_supported(n->dtype, g);
}

template <typename T>
int run_forward(maua_clip* n, long N, bool keep) {
  hipStream_t st = n->ctx->stream;
  const int w = n->width, Tk = n->Tk;
  const long M = N * Tk, Mp = N * n->G * n->G;
  constexpr int E = 16 / (int)sizeof(T);
  // patch embedding (conv1, no bias) -> tokens -> ln_pre
  if (int rc = gemm(n, n->patches, Mp, n->kp, n->w_conv, w, nullptr, nullptr, n->pe)) return rc;
  {
    const long total = M * (w / E);
    hipLaunchKernelGGL(tokens_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const T*)n->pe, n->cls, n->pos, (T*)n->tok, N,
                       Tk, w);
  }
  const dim3 lgrid((unsigned)((M + 3) / 4));
  void* x = n->L[0].x;
  hipLaunchKernelGGL(layer_norm_kernel<T>, lgrid, dim3(256), 0, st, (const T*)n->tok, n->lnpre_g, n->lnpre_b, (T*)x, n->st_pre, M, w);
  for (int i = 0; i < n->layers; i++) {
    Layer& wl = n->L[i];                 // weights
    Layer& kl = n->L[keep ? i : 0];      // activations
    // x -> ln_1 -> in_proj -> attention -> out_proj (+ x) = xm
    hipLaunchKernelGGL(layer_norm_kernel<T>, lgrid, dim3(256), 0, st, (const T*)x, wl.ln1_g, wl.ln1_b, (T*)n->lno, kl.st1, M, w);
    if (int rc = gemm(n, n->lno, M, w, wl.w_qkv, 3 * w, wl.b_qkv, nullptr, kl.qkv)) return rc;
    AttnArgs a{};
    a.qkv = kl.qkv; a.out = kl.ao; a.B = (int)N; a.T = Tk; a.heads = n->heads; a.D = w / n->heads; a.ld_qkv = 3 * w; a.ld_out = w;
    a.scale = 1.f / std::sqrt((float)a.D); a.lse = kl.lse;
    if (int rc = launch_attention(st, n->dtype, a)) return rc;
    if (int rc = gemm(n, kl.ao, M, w, wl.w_out, w, wl.b_out, x, kl.xm)) return rc;
    // xm -> ln_2 -> c_fc -> QuickGELU -> c_proj (+ xm) = next x
    hipLaunchKernelGGL(layer_norm_kernel<T>, lgrid, dim3(256), 0, st, (const T*)kl.xm, wl.ln2_g, wl.ln2_b, (T*)n->lno, kl.st2, M, w);
    if (gemm_fuses(n, M, w, 4 * w)) {
      if (int rc = gemm(n, n->lno, M, w, wl.w_fc, 4 * w, wl.b_fc, nullptr, kl.h, 1, n->act)) return rc;
    } else {
      if (int rc = gemm(n, n->lno, M, w, wl.w_fc, 4 * w, wl.b_fc, nullptr, kl.h)) return rc;
      const long pieces = M * 4 * w / E;
      hipLaunchKernelGGL(quick_gelu_kernel<T>, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, (const T*)kl.h, (T*)n->act, pieces);
    }
    void* xn = i + 1 < n->layers ? (keep ? n->L[i + 1].x : (void*)((x == n->L[0].x) ? n->x_last : n->L[0].x)) : n->x_last;
    if (int rc = gemm(n, n->act, M, 4 * w, wl.w_pr, w, wl.b_pr, kl.xm, xn)) return rc;
    x = xn;
  }
  MAUA_HIP_CHECK(hipGetLastError());
  n->kept_N = keep ? N : 0;
  return MAUA_OK;
}

int run_head(maua_clip* n, long N, bool with_grad, const float* d_embed, int B, float coef, bool write_loss, const float* mult = nullptr) {
  HeadArgs h{};
  h.x = n->x_last; h.g = n->lnpost_g; h.b = n->lnpost_b; h.proj = n->proj; h.tgt = n->tgt; h.twt = n->twt;
  h.sel = n->sel_B > 0 ? n->sel : nullptr;
  h.embed = n->embed; h.loss = write_loss ? n->loss : nullptr; h.d_embed = d_embed; h.dx = with_grad ? n->dxa : nullptr;
  h.Tk = n->Tk; h.w = n->width; h.E = n->E; h.P = n->P; h.B = B > 0 ? B : 1; h.coef = coef; h.mult = mult;
  const size_t smem = ((size_t)2 * n->width + 2 * n->E + std::max(n->P, 1) + 8) * 4;
  MAUA_REQUIRE(smem <= 64 * 1024, "maua_clip: too many targets / too wide a tower for the head kernel's LDS");
  if (n->dtype == MAUA_BF16) hipLaunchKernelGGL(clip_head_kernel<bf16_t>, dim3((unsigned)N), dim3(256), smem, n->ctx->stream, h);
  else hipLaunchKernelGGL(clip_head_kernel<float>, dim3((unsigned)N), dim3(256), smem, n->ctx->stream, h);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// d x_last (in dxa: row 0 per image set by the head, the rest zero) -> d patches (in n->patches' layout, written to n->dwide)
template <typename T>
int run_backward(maua_clip* n, long N) {
  hipStream_t st = n->ctx->stream;
  const int w = n->width, Tk = n->Tk;
  const long M = N * Tk, Mp = N * n->G * n->G;
  constexpr int E = 16 / (int)sizeof(T);
  const dim3 lgrid((unsigned)((M + 3) / 4));
  void* dx = n->dxa;      // gradient of the current residual stream
  void* other = n->dxb;
  for (int i = n->layers - 1; i >= 0; i--) {
    Layer& l = n->L[i];
    // MLP: d act = dx W_pr ; d h = d act * QuickGELU'(h) ; d ln2 = d h W_fc ; dxm = dx + LN2'(d ln2)
    if (gemm_fuses(n, M, w, 4 * w)) {
      if (int rc = gemm(n, dx, M, w, l.w_pr_t, 4 * w, nullptr, nullptr, n->dwide, 2, nullptr, l.h)) return rc;
    } else {
      if (int rc = gemm(n, dx, M, w, l.w_pr_t, 4 * w, nullptr, nullptr, n->dwide)) return rc;
      const long pieces = M * 4 * w / E;
      hipLaunchKernelGGL(quick_gelu_vjp_kernel<T>, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, (const T*)l.h, (const T*)n->dwide,
                         (T*)n->dwide, pieces);
    }
    if (int rc = gemm(n, n->dwide, M, 4 * w, l.w_fc_t, w, nullptr, nullptr, n->dtmp)) return rc;
    hipLaunchKernelGGL(layer_norm_vjp_kernel<T>, lgrid, dim3(256), 0, st, (const T*)l.xm, l.st2, l.ln2_g, (const T*)n->dtmp, (const T*)dx,
                       (T*)other, M, w, 0);
    std::swap(dx, other);   // dx = d xm
    // attention: d ao = dxm W_out ; d qkv = attention'(d ao) ; d ln1 = d qkv W_qkv ; d x = dxm + LN1'(d ln1)
    if (int rc = gemm(n, dx, M, w, l.w_out_t, w, nullptr, nullptr, n->dtmp)) return rc;
    AttnVjpArgs v{};
    v.qkv = l.qkv; v.out = l.ao; v.d_out = n->dtmp; v.lse = l.lse; v.d_qkv = n->dqkv; v.delta = n->delta;
    v.B = (int)N; v.T = Tk; v.heads = n->heads; v.D = w / n->heads; v.ld_qkv = 3 * w; v.ld_out = w; v.scale = 1.f / std::sqrt((float)v.D);
    if (int rc = launch_attention_vjp(st, n->dtype, v)) return rc;
    if (int rc = gemm(n, n->dqkv, M, 3 * w, l.w_qkv_t, w, nullptr, nullptr, n->dtmp)) return rc;
    hipLaunchKernelGGL(layer_norm_vjp_kernel<T>, lgrid, dim3(256), 0, st, (const T*)l.x, l.st1, l.ln1_g, (const T*)n->dtmp, (const T*)dx,
                       (T*)other, M, w, 0);
    std::swap(dx, other);
  }
  // ln_pre backwards (dropping the class token: its inputs are parameters), then the patch embedding's transpose
  hipLaunchKernelGGL(layer_norm_vjp_kernel<T>, lgrid, dim3(256), 0, st, (const T*)n->tok, n->st_pre, n->lnpre_g, (const T*)dx, (const T*)nullptr,
                     (T*)n->dtmp, M, w, Tk);
  if (int rc = gemm(n, n->dtmp, Mp, w, n->w_conv_t, n->kp, nullptr, nullptr, n->dwide)) return rc;
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int zero_dx(maua_clip* n, long N) {
  const long pieces = N * n->Tk * n->width * (long)n->esize / 16;
  hipLaunchKernelGGL(zero16_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, n->ctx->stream, (u32x4*)n->dxa, pieces);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int forward_any(maua_clip* n, long N, bool keep) {
  return n->dtype == MAUA_BF16 ? run_forward<bf16_t>(n, N, keep) : run_forward<float>(n, N, keep);
}
int backward_any(maua_clip* n, long N) { return n->dtype == MAUA_BF16 ? run_backward<bf16_t>(n, N) : run_backward<float>(n, N); }

// host float matrix [R][C] -> device T [R][C] (and optionally its transpose [C][R]); rows permuted by `perm` when given
int upload_matrix(maua_clip* n, const float* h, int R, int C, const int* perm, void** dev, void** dev_t) {
  std::vector<float> a((size_t)R * C);
  for (int r = 0; r < R; r++) memcpy(&a[(size_t)r * C], h + (size_t)(perm ? perm[r] : r) * C, (size_t)C * 4);
  auto put = [&](const std::vector<float>& src, void** d) -> int {
    if (!*d)
      if (int rc = dalloc(d, src.size() * n->esize)) return rc;
    if (n->dtype == MAUA_F32) {
      MAUA_HIP_CHECK(hipMemcpy(*d, src.data(), src.size() * 4, hipMemcpyHostToDevice));
    } else {
      std::vector<uint16_t> b(src.size());
      for (size_t i = 0; i < src.size(); i++) {   // round to nearest even
        uint32_t u;
        memcpy(&u, &src[i], 4);
        if ((u & 0x7f800000u) == 0x7f800000u) b[i] = (uint16_t)(u >> 16);
        else b[i] = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
      }
      MAUA_HIP_CHECK(hipMemcpy(*d, b.data(), b.size() * 2, hipMemcpyHostToDevice));
    }
    return MAUA_OK;
  };
  if (int rc = put(a, dev)) return rc;
  if (dev_t) {
    std::vector<float> t((size_t)R * C);
    for (int r = 0; r < R; r++)
      for (int c = 0; c < C; c++) t[(size_t)c * R + r] = a[(size_t)r * C + c];
    if (int rc = put(t, dev_t)) return rc;
  }
  return MAUA_OK;
}
int upload_vec(const float* h, size_t count, const int* perm, float** dev) {
  std::vector<float> a(count);
  for (size_t i = 0; i < count; i++) a[i] = h[perm ? perm[i] : i];
  if (!*dev)
    if (int rc = dalloc((void**)dev, count * 4)) return rc;
  MAUA_HIP_CHECK(hipMemcpy(*dev, a.data(), count * 4, hipMemcpyHostToDevice));
  return MAUA_OK;
}

}  // namespace

void clip_stamp(maua_clip* n, unsigned long long* uid, unsigned long long* epoch) {
  *uid = n ? n->uid : 0;
  *epoch = n ? n->epoch : 0;
}
maua_ctx* clip_ctx(maua_clip* n) { return n ? n->ctx : nullptr; }

// One cutout batch group of CLIPGrads.forward on device rectangles: cutouts of img -> tower -> loss head -> back to d img, added into
// (or written to) grad.  rects_dev: [n_cut][3].  coef: the head's factor (scale excluded: applied at the end with the clamp).
int clip_grad_group(maua_clip* n, const float* img, int B, int H, int W, const int* rects_dev, const float* mult_dev, int n_cut, float coef,
                    float* grad, int accumulate) {
  hipStream_t st = n->ctx->stream;
  const long N = (long)n_cut * B;
  CutoutPlan p{};
  p.img = img; p.rects = rects_dev; p.B = B; p.H = H; p.W = W; p.n_cut = n_cut; p.cs = n->res; p.mul = 0.5f; p.add = 0.5f;
  const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f}, sd[3] = {0.26862954f, 0.26130258f, 0.27577711f};   // grad.py:110
  for (int c = 0; c < 3; c++) { p.mean[c] = mean[c]; p.std[c] = sd[c]; }
  p.patch = n->patch;
  if (int rc = launch_cutout_tables(st, p, n->cut_tables)) return rc;
  if (int rc = launch_cutouts_forward(st, n->dtype, p, n->cut_tables, n->patches)) return rc;
  if (int rc = forward_any(n, N, true)) return rc;
  if (int rc = zero_dx(n, N)) return rc;
  if (int rc = run_head(n, N, true, nullptr, B, coef, true, mult_dev)) return rc;
  if (int rc = backward_any(n, N)) return rc;
  return launch_cutouts_vjp(st, n->dtype, p, n->cut_tables, n->dwide, n->cut_th, grad, accumulate);
}

// workspaces of a guidance call (all allocations happen here, none inside clip_grad_group: a captured loop calls this first)
int clip_prepare_guide(maua_clip* n, int B, int H, int W, int n_cut_group) {
  MAUA_REQUIRE(n->tgt && n->P > 0, "maua_clip: no targets (maua_clip_set_targets)");
  if (int rc = ensure_ws(n, (long)n_cut_group * B, 1)) return rc;
  const size_t tb = cutouts_table_bytes(n_cut_group, n->res), thb = cutouts_th_bytes(n_cut_group, B, n->res, std::min(H, W));
  if (tb > n->cut_tables_bytes || thb > n->cut_th_bytes || !n->parts) {
    MAUA_HIP_CHECK(hipStreamSynchronize(n->ctx->stream));
    dfree(n->cut_tables); dfree(n->cut_th);
    n->cut_tables = nullptr; n->cut_th = nullptr;
    n->cut_tables_bytes = std::max(tb, n->cut_tables_bytes); n->cut_th_bytes = std::max(thb, n->cut_th_bytes);
    if (int rc = dalloc(&n->cut_tables, n->cut_tables_bytes)) return rc;
    if (int rc = dalloc((void**)&n->cut_th, n->cut_th_bytes)) return rc;
    if (!n->parts)
      if (int rc = dalloc((void**)&n->parts, NPARTS * 8)) return rc;
    n->epoch++;
  }
  return MAUA_OK;
}

// cutouts per pass through the tower: all of a cutout batch when they fit (32-bit byte offsets inside the GEMM operands, <= 65535 images)
int clip_group_size(maua_clip* n, int B, int cutn) {
  const long max_images = std::min<long>(65535, ((1L << 32) - 1) / ((long)n->Tk * 4 * n->width * (long)n->esize));
  return (int)std::max<long>(1, std::min<long>(cutn, max_images / std::max(B, 1)));
}

// the whole of CLIPGrads.forward (:145-159) on device rectangles [batches][cutn][3].  mult_dev: NULL, or [batches][cutn] multiplicities -
// cutout n stands for mult identical cutouts of the reference's list (on a square image its first cutn // 4 cutouts are the same
// rectangle, cutouts.py:16-27: one pass through the tower carries their weight) and cutn_total = what the multiplicities of a batch sum to
int clip_guide_grad(maua_clip* n, const float* img, int B, int H, int W, const int* rects_dev, const float* mult_dev, int cutn, int cutn_total,
                    int batches, float scale, float clamp_gradient, float* grad) {
  const int grp = clip_group_size(n, B, cutn);
  const float coef = 1.f / ((float)cutn_total * (float)batches);
  bool first = true;
  for (int k = 0; k < batches; k++)
    for (int c0 = 0; c0 < cutn; c0 += grp) {
      const int nc = std::min(grp, cutn - c0);
      if (int rc = clip_grad_group(n, img, B, H, W, rects_dev + ((long)k * cutn + c0) * 3, mult_dev ? mult_dev + (long)k * cutn + c0 : nullptr, nc,
                                   coef, grad, first ? 0 : 1))
        return rc;
      first = false;
    }
  const long cnt = (long)B * 3 * H * W;
  hipStream_t st = n->ctx->stream;
  hipLaunchKernelGGL(sq_partial_kernel, dim3(NPARTS), dim3(256), 0, st, grad, cnt, n->parts);
  hipLaunchKernelGGL(clamp_scale_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, grad, cnt, n->parts, NPARTS, clamp_gradient,
                     scale);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace maua

extern "C" {

int maua_clip_create(maua_ctx* ctx, int input_resolution, int patch_size, int width, int layers, int heads, int output_dim, int dtype,
                     maua_clip** out) {
  MAUA_REQUIRE(ctx && out, "maua_clip_create: NULL argument");
  MAUA_REQUIRE(dtype == MAUA_BF16 || dtype == MAUA_F32, "maua_clip_create: dtype must be MAUA_BF16 or MAUA_F32");
  MAUA_REQUIRE(input_resolution > 0 && patch_size > 0 && input_resolution % patch_size == 0, "maua_clip_create: the input is a whole number of patches");
  MAUA_REQUIRE(width > 0 && heads > 0 && width % heads == 0 && attention_supported(width / heads), "maua_clip_create: head width must be 32 or 64");
  const int epc = dtype == MAUA_BF16 ? 8 : 4, kc = dtype == MAUA_BF16 ? 32 : 16;
  MAUA_REQUIRE(width % 32 == 0 && width % kc == 0 && width / epc <= 64 * LN_MAXP, "maua_clip_create: width must be a multiple of 32 (at most 4096 / 2048)");
  MAUA_REQUIRE((3 * patch_size * patch_size) % 32 == 0, "maua_clip_create: 3 * patch_size^2 must be a multiple of 32");
  MAUA_REQUIRE(layers > 0 && output_dim > 0, "maua_clip_create: bad layer count / embedding size");
  MAUA_REQUIRE(3 * patch_size * patch_size <= 4 * width, "maua_clip_create: a patch row must not be wider than the MLP (shared workspace)");
  maua_clip* n = new maua_clip();
  static std::atomic<unsigned long long> next_uid{1};
  n->uid = next_uid.fetch_add(1);
  n->ctx = ctx; n->dtype = dtype; n->esize = dtype == MAUA_BF16 ? 2 : 4;
  n->res = input_resolution; n->patch = patch_size; n->width = width; n->layers = layers; n->heads = heads; n->E = output_dim;
  n->G = input_resolution / patch_size; n->Tk = n->G * n->G + 1; n->kp = 3 * patch_size * patch_size;
  n->L.resize(layers);
  *out = n;
  return MAUA_OK;
}

void maua_clip_destroy(maua_clip* n) {
  if (!n) return;
  hipStreamSynchronize(n->ctx->stream);
  free_ws(n);
  for (auto& l : n->L)
    for (void* p : {l.w_qkv, l.w_qkv_t, l.w_out, l.w_out_t, l.w_fc, l.w_fc_t, l.w_pr, l.w_pr_t, (void*)l.b_qkv, (void*)l.b_out, (void*)l.b_fc,
                    (void*)l.b_pr, (void*)l.ln1_g, (void*)l.ln1_b, (void*)l.ln2_g, (void*)l.ln2_b})
      dfree(p);
  for (void* p : {n->w_conv, n->w_conv_t, (void*)n->cls, (void*)n->pos, (void*)n->lnpre_g, (void*)n->lnpre_b, (void*)n->lnpost_g,
                  (void*)n->lnpost_b, (void*)n->proj, (void*)n->tgt, (void*)n->twt, (void*)n->sel, n->cut_tables, (void*)n->cut_th,
                  (void*)n->rects_dev, (void*)n->parts})
    dfree(p);
  delete n;
}

// name: a key of CLIP's state dict below "visual." ("conv1.weight", "class_embedding", "positional_embedding", "ln_pre.weight", ...,
// "transformer.resblocks.<i>.attn.in_proj_weight", ..., "ln_post.bias", "proj"); host float32 data in the checkpoint's layout
int maua_clip_load(maua_clip* n, const char* name, const float* host, size_t count) {
  MAUA_REQUIRE(n && name && host, "maua_clip_load: NULL argument");
  MAUA_HIP_CHECK(hipStreamSynchronize(n->ctx->stream));
  const std::string s(name);
  const int w = n->width;
  auto need = [&](size_t c) -> int {
    if (count != c) return fail("maua_clip_load: " + s + ": wrong size");
    return MAUA_OK;
  };
  if (s == "conv1.weight") { if (int rc = need((size_t)w * n->kp)) return rc; return upload_matrix(n, host, w, n->kp, nullptr, &n->w_conv, &n->w_conv_t); }
  if (s == "class_embedding") { if (int rc = need(w)) return rc; return upload_vec(host, w, nullptr, &n->cls); }
  if (s == "positional_embedding") { if (int rc = need((size_t)n->Tk * w)) return rc; return upload_vec(host, (size_t)n->Tk * w, nullptr, &n->pos); }
  if (s == "ln_pre.weight") { if (int rc = need(w)) return rc; return upload_vec(host, w, nullptr, &n->lnpre_g); }
  if (s == "ln_pre.bias") { if (int rc = need(w)) return rc; return upload_vec(host, w, nullptr, &n->lnpre_b); }
  if (s == "ln_post.weight") { if (int rc = need(w)) return rc; return upload_vec(host, w, nullptr, &n->lnpost_g); }
  if (s == "ln_post.bias") { if (int rc = need(w)) return rc; return upload_vec(host, w, nullptr, &n->lnpost_b); }
  if (s == "proj") { if (int rc = need((size_t)w * n->E)) return rc; return upload_vec(host, (size_t)w * n->E, nullptr, &n->proj); }
  const std::string pre = "transformer.resblocks.";
  if (s.compare(0, pre.size(), pre) == 0) {
    const size_t dot = s.find('.', pre.size());
    if (dot == std::string::npos) return fail("maua_clip_load: unknown parameter name: " + s);
    const int i = atoi(s.substr(pre.size(), dot - pre.size()).c_str());
    if (i < 0 || i >= n->layers) return fail("maua_clip_load: no such block: " + s);
    Layer& l = n->L[i];
    const std::string par = s.substr(dot + 1);
    // in_proj rows [q | k | v] x [head][d] -> head-major [head][q | k | v][d]: the layout attention.hip reads
    const int D = w / n->heads;
    std::vector<int> perm(3 * w);
    for (int h = 0; h < n->heads; h++)
      for (int part = 0; part < 3; part++)
        for (int d = 0; d < D; d++) perm[(h * 3 + part) * D + d] = part * w + h * D + d;
    if (par == "attn.in_proj_weight") { if (int rc = need((size_t)3 * w * w)) return rc; return upload_matrix(n, host, 3 * w, w, perm.data(), &l.w_qkv, &l.w_qkv_t); }
    if (par == "attn.in_proj_bias") { if (int rc = need((size_t)3 * w)) return rc; return upload_vec(host, 3 * w, perm.data(), &l.b_qkv); }
    if (par == "attn.out_proj.weight") { if (int rc = need((size_t)w * w)) return rc; return upload_matrix(n, host, w, w, nullptr, &l.w_out, &l.w_out_t); }
    if (par == "attn.out_proj.bias") { if (int rc = need(w)) return rc; return upload_vec(host, w, nullptr, &l.b_out); }
    if (par == "mlp.c_fc.weight") { if (int rc = need((size_t)4 * w * w)) return rc; return upload_matrix(n, host, 4 * w, w, nullptr, &l.w_fc, &l.w_fc_t); }
    if (par == "mlp.c_fc.bias") { if (int rc = need((size_t)4 * w)) return rc; return upload_vec(host, 4 * w, nullptr, &l.b_fc); }
    if (par == "mlp.c_proj.weight") { if (int rc = need((size_t)4 * w * w)) return rc; return upload_matrix(n, host, w, 4 * w, nullptr, &l.w_pr, &l.w_pr_t); }
    if (par == "mlp.c_proj.bias") { if (int rc = need(w)) return rc; return upload_vec(host, w, nullptr, &l.b_pr); }
    if (par == "ln_1.weight") { if (int rc = need(w)) return rc; return upload_vec(host, w, nullptr, &l.ln1_g); }
    if (par == "ln_1.bias") { if (int rc = need(w)) return rc; return upload_vec(host, w, nullptr, &l.ln1_b); }
    if (par == "ln_2.weight") { if (int rc = need(w)) return rc; return upload_vec(host, w, nullptr, &l.ln2_g); }
    if (par == "ln_2.bias") { if (int rc = need(w)) return rc; return upload_vec(host, w, nullptr, &l.ln2_b); }
  }
  return fail("maua_clip_load: unknown parameter name: " + s);
}

static int clip_loaded(maua_clip* n) {
  bool ok = n->w_conv && n->cls && n->pos && n->lnpre_g && n->lnpre_b && n->lnpost_g && n->lnpost_b && n->proj;
  for (auto& l : n->L)
    ok = ok && l.w_qkv && l.w_out && l.w_fc && l.w_pr && l.b_qkv && l.b_out && l.b_fc && l.b_pr && l.ln1_g && l.ln1_b && l.ln2_g && l.ln2_b;
  if (!ok) return fail("maua_clip: parameters missing (maua_clip_load every key of the image tower first)");
  return MAUA_OK;
}

// VisionTransformer.forward: images [N][3][R][R] f32 (already normalised), device -> embeds [N][E] f32.  keep != 0: the activations
// stay for maua_clip_encode_image_vjp
int maua_clip_encode_image(maua_clip* n, const float* images, int N, int keep, float* embeds) {
  MAUA_REQUIRE(n && images && embeds, "maua_clip_encode_image: NULL argument");
  MAUA_REQUIRE(N >= 0 && N <= 65535, "maua_clip_encode_image: 0 .. 65535 images per call");
  if (int rc = clip_loaded(n)) return rc;
  if (N == 0) return MAUA_OK;
  if (int rc = ensure_ws(n, N, keep ? 1 : 0)) return rc;
  hipStream_t st = n->ctx->stream;
  const long total = (long)N * 3 * n->res * n->res;
  if (n->dtype == MAUA_BF16)
    hipLaunchKernelGGL(im2patch_kernel<bf16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, images, (bf16_t*)n->patches, (long)N, n->res, n->patch);
  else
    hipLaunchKernelGGL(im2patch_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, images, (float*)n->patches, (long)N, n->res, n->patch);
  if (int rc = forward_any(n, N, keep != 0)) return rc;
  if (int rc = run_head(n, N, false, nullptr, 1, 0.f, false)) return rc;
  MAUA_HIP_CHECK(hipMemcpyAsync(embeds, n->embed, (size_t)N * n->E * 4, hipMemcpyDeviceToDevice, st));
  return MAUA_OK;
}

// (d embeds / d images)^T d_embeds for the images of the last maua_clip_encode_image(keep = 1): [N][E] -> [N][3][R][R] f32
int maua_clip_encode_image_vjp(maua_clip* n, const float* d_embeds, int N, float* d_images) {
  MAUA_REQUIRE(n && d_embeds && d_images, "maua_clip_encode_image_vjp: NULL argument");
  MAUA_REQUIRE(N > 0 && n->kept_N == N, "maua_clip_encode_image_vjp: no kept forward of that many images (maua_clip_encode_image with keep = 1)");
  hipStream_t st = n->ctx->stream;
  if (int rc = zero_dx(n, N)) return rc;
  if (int rc = run_head(n, N, true, d_embeds, 1, 0.f, false)) return rc;
  if (int rc = backward_any(n, N)) return rc;
  const long total = (long)N * 3 * n->res * n->res;
  if (n->dtype == MAUA_BF16)
    hipLaunchKernelGGL(patch2im_kernel<bf16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const bf16_t*)n->dwide, d_images, (long)N, n->res, n->patch);
  else
    hipLaunchKernelGGL(patch2im_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const float*)n->dwide, d_images, (long)N, n->res, n->patch);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// CLIPGrads.set_targets' result (:117-143): S sets of P target embeddings [S][P][E] (host or device floats are both copied with
// hipMemcpyDefault; normalised here like spherical_dist_loss does) and prompt weights [S][P] (already divided by |sum|).  sel: per
// sample of a batch the set it is guided towards ([B] host ints, audio-switched prompts) or NULL (set 0 for everybody)
int maua_clip_set_targets(maua_clip* n, const float* targets, const float* weights, int S, int P, const int* sel, int B) {
  MAUA_REQUIRE(n && targets && weights && S > 0 && P > 0, "maua_clip_set_targets: bad arguments");
  MAUA_HIP_CHECK(hipStreamSynchronize(n->ctx->stream));
  std::vector<float> t((size_t)S * P * n->E), wv((size_t)S * P);
  MAUA_HIP_CHECK(hipMemcpy(t.data(), targets, t.size() * 4, hipMemcpyDefault));
  MAUA_HIP_CHECK(hipMemcpy(wv.data(), weights, wv.size() * 4, hipMemcpyDefault));
  for (long r = 0; r < (long)S * P; r++) {
    double ss = 0;
    for (int j = 0; j < n->E; j++) ss += (double)t[r * n->E + j] * t[r * n->E + j];
    const float nn = std::max((float)std::sqrt(ss), 1e-12f);
    for (int j = 0; j < n->E; j++) t[r * n->E + j] /= nn;
  }
  if (S * P != n->S * n->P) {
    dfree(n->tgt); dfree(n->twt);
    n->tgt = nullptr; n->twt = nullptr;
    if (int rc = dalloc((void**)&n->tgt, t.size() * 4)) return rc;
    if (int rc = dalloc((void**)&n->twt, wv.size() * 4)) return rc;
    n->epoch++;
  }
  MAUA_HIP_CHECK(hipMemcpy(n->tgt, t.data(), t.size() * 4, hipMemcpyHostToDevice));
  MAUA_HIP_CHECK(hipMemcpy(n->twt, wv.data(), wv.size() * 4, hipMemcpyHostToDevice));
  n->S = S; n->P = P;
  n->sel_B = 0;
  if (sel) {
    MAUA_REQUIRE(B > 0, "maua_clip_set_targets: sel needs the batch size");
    for (int b = 0; b < B; b++) MAUA_REQUIRE(sel[b] >= 0 && sel[b] < S, "maua_clip_set_targets: sel out of range");
    if (B > n->sel_cap) {
      dfree(n->sel);
      n->sel = nullptr;
      if (int rc = dalloc((void**)&n->sel, (size_t)B * 4)) return rc;
      n->sel_cap = B;
      n->epoch++;
    }
    MAUA_HIP_CHECK(hipMemcpy(n->sel, sel, (size_t)B * 4, hipMemcpyHostToDevice));
    n->sel_B = B;
  }
  return MAUA_OK;
}

// CLIPGrads.forward (:145-159): img [B][3][H][W] f32 in [-1, 1] (device), rects: HOST ints [batches][cutn][3] = (size, top, left) of
// every cutout of every cutout batch (what MauaCutouts draws), -> grad [B][3][H][W] f32 = d (scale * sum_b loss_b) / d img, averaged
// over the cutout batches, with clamp_gradient (<= 0: none) applied
int maua_clip_guide_grad(maua_clip* n, const float* img, int B, int H, int W, const int* rects, const float* mult, int cutn, int batches,
                         float scale, float clamp_gradient, float* grad) {
  MAUA_REQUIRE(n && img && rects && grad, "maua_clip_guide_grad: NULL argument");
  MAUA_REQUIRE(B >= 0 && cutn > 0 && batches > 0, "maua_clip_guide_grad: bad sizes");
  if (int rc = clip_loaded(n)) return rc;
  if (B == 0) return MAUA_OK;
  MAUA_REQUIRE(n->sel_B == 0 || n->sel_B == B, "maua_clip_guide_grad: the per-sample target selection was set for another batch size");
  for (long i = 0; i < (long)batches * cutn; i++) {
    const int s = rects[3 * i] & CUT_SIZE_MASK, oy = rects[3 * i + 1], ox = rects[3 * i + 2];
    MAUA_REQUIRE(s > 0 && oy >= 0 && ox >= 0 && oy + s <= H && ox + s <= W, "maua_clip_guide_grad: a cutout leaves the image");
  }
  int cutn_total = cutn;
  if (mult) {
    double tot = 0;
    for (int i = 0; i < cutn; i++) tot += mult[i];
    cutn_total = (int)(tot + 0.5);
    for (int k = 1; k < batches; k++) {
      double t = 0;
      for (int i = 0; i < cutn; i++) t += mult[(long)k * cutn + i];
      MAUA_REQUIRE((int)(t + 0.5) == cutn_total, "maua_clip_guide_grad: every cutout batch must stand for the same number of cutouts");
    }
    MAUA_REQUIRE(cutn_total >= cutn, "maua_clip_guide_grad: multiplicities are >= 1");
  }
  const size_t cnt = (size_t)batches * cutn * 4;   // 3 ints + 1 float per cutout
  if (cnt > n->rects_cap) {
    MAUA_HIP_CHECK(hipStreamSynchronize(n->ctx->stream));
    dfree(n->rects_dev);
    n->rects_dev = nullptr;
    if (int rc = dalloc((void**)&n->rects_dev, cnt * 4)) return rc;
    n->rects_cap = cnt;
  }
  float* mult_dev = mult ? (float*)(n->rects_dev + (size_t)batches * cutn * 3) : nullptr;
  MAUA_HIP_CHECK(hipMemcpyAsync(n->rects_dev, rects, (size_t)batches * cutn * 12, hipMemcpyHostToDevice, n->ctx->stream));
  if (mult) MAUA_HIP_CHECK(hipMemcpyAsync(mult_dev, mult, (size_t)batches * cutn * 4, hipMemcpyHostToDevice, n->ctx->stream));
  MAUA_HIP_CHECK(hipStreamSynchronize(n->ctx->stream));   // (the caller's arrays may be temporaries)
  if (int rc = clip_prepare_guide(n, B, H, W, clip_group_size(n, B, cutn))) return rc;
  return clip_guide_grad(n, img, B, H, W, n->rects_dev, mult_dev, cutn, cutn_total, batches, scale, clamp_gradient, grad);
}

// sum_p w_p dist_p of each cutout image of the LAST pass through the tower ([n] floats, device; n <= cutn * B: cutout-major, what
// `dists.view(-1, B, P).mul(weights).sum(2)` holds in grad.py:153)
int maua_clip_last_image_losses(maua_clip* n, int count, float* out) {
  MAUA_REQUIRE(n && out && count >= 0 && count <= n->kept_N, "maua_clip_last_image_losses: more values asked for than the last pass had images");
  MAUA_HIP_CHECK(hipMemcpyAsync(out, n->loss, (size_t)count * 4, hipMemcpyDeviceToDevice, n->ctx->stream));
  return MAUA_OK;
}

// operator-level pieces (tests, other perceptors): the cutouts and their gradient on HOST rectangles [n_cut][3]
int maua_cutouts(maua_ctx* ctx, const float* img, int B, int H, int W, const int* rects, int n_cut, int cut_size, float mul, float add,
                 const float* mean3, const float* std3, float* out) {
  MAUA_REQUIRE(ctx && img && rects && out && mean3 && std3, "maua_cutouts: NULL argument");
  const size_t tb = cutouts_table_bytes(n_cut, cut_size);
  if (int rc = scratch_reserve(ctx, tb + (size_t)n_cut * 12 + 512)) return rc;
  int* rd = (int*)((char*)ctx->scratch + ((tb + 255) & ~(size_t)255));
  MAUA_HIP_CHECK(hipMemcpyAsync(rd, rects, (size_t)n_cut * 12, hipMemcpyHostToDevice, ctx->stream));
  MAUA_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  CutoutPlan p{};
  p.img = img; p.rects = rd; p.B = B; p.H = H; p.W = W; p.n_cut = n_cut; p.cs = cut_size; p.mul = mul; p.add = add;
  for (int c = 0; c < 3; c++) { p.mean[c] = mean3[c]; p.std[c] = std3[c]; }
  if (int rc = launch_cutout_tables(ctx->stream, p, ctx->scratch)) return rc;
  return launch_cutouts_forward(ctx->stream, MAUA_F32, p, ctx->scratch, out);
}

int maua_cutouts_vjp(maua_ctx* ctx, const float* d_out, int B, int H, int W, const int* rects, int n_cut, int cut_size, float mul,
                     const float* std3, float* d_img) {
  MAUA_REQUIRE(ctx && d_out && rects && d_img && std3, "maua_cutouts_vjp: NULL argument");
  const size_t tb = (cutouts_table_bytes(n_cut, cut_size) + 255) & ~(size_t)255;
  const size_t thb = (cutouts_th_bytes(n_cut, B, cut_size, std::min(H, W)) + 255) & ~(size_t)255;
  if (int rc = scratch_reserve(ctx, tb + thb + (size_t)n_cut * 12 + 512)) return rc;
  int* rd = (int*)((char*)ctx->scratch + tb + thb);
  MAUA_HIP_CHECK(hipMemcpyAsync(rd, rects, (size_t)n_cut * 12, hipMemcpyHostToDevice, ctx->stream));
  MAUA_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  CutoutPlan p{};
  p.img = nullptr; p.rects = rd; p.B = B; p.H = H; p.W = W; p.n_cut = n_cut; p.cs = cut_size; p.mul = mul; p.add = 0.f;
  for (int c = 0; c < 3; c++) { p.mean[c] = 0.f; p.std[c] = std3[c]; }
  if (int rc = launch_cutout_tables(ctx->stream, p, ctx->scratch)) return rc;
  return launch_cutouts_vjp(ctx->stream, MAUA_F32, p, ctx->scratch, d_out, (float*)((char*)ctx->scratch + tb), d_img, 0);
}

// nn.LayerNorm over the last dimension of [rows][C] (float32 statistics, eps 1e-5) and its input gradient; stats: [rows][2] f32
int maua_layer_norm(maua_ctx* ctx, const void* x, const float* gamma, const float* beta, long rows, int C, int dtype, void* y, float* stats) {
  MAUA_REQUIRE(ctx && x && gamma && beta && y, "maua_layer_norm: NULL argument");
  MAUA_REQUIRE(dtype == MAUA_BF16 || dtype == MAUA_F32, "maua_layer_norm: f32 / bf16");
  const int epc = dtype == MAUA_BF16 ? 8 : 4;
  MAUA_REQUIRE(C % epc == 0 && C / epc <= 64 * LN_MAXP, "maua_layer_norm: C must be a multiple of 16 bytes and at most 4096 (bf16) / 2048 (f32)");
  if (rows == 0) return MAUA_OK;
  const dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == MAUA_BF16) hipLaunchKernelGGL(layer_norm_kernel<bf16_t>, grid, dim3(256), 0, ctx->stream, (const bf16_t*)x, gamma, beta, (bf16_t*)y, stats, rows, C);
  else hipLaunchKernelGGL(layer_norm_kernel<float>, grid, dim3(256), 0, ctx->stream, (const float*)x, gamma, beta, (float*)y, stats, rows, C);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}
int maua_layer_norm_vjp(maua_ctx* ctx, const void* x, const float* stats, const float* gamma, const void* dy, const void* add, long rows, int C,
                        int dtype, void* dx) {
  MAUA_REQUIRE(ctx && x && stats && gamma && dy && dx, "maua_layer_norm_vjp: NULL argument");
  MAUA_REQUIRE(dtype == MAUA_BF16 || dtype == MAUA_F32, "maua_layer_norm_vjp: f32 / bf16");
  const int epc = dtype == MAUA_BF16 ? 8 : 4;
  MAUA_REQUIRE(C % epc == 0 && C / epc <= 64 * LN_MAXP, "maua_layer_norm_vjp: C must be a multiple of 16 bytes and at most 4096 (bf16) / 2048 (f32)");
  if (rows == 0) return MAUA_OK;
  const dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == MAUA_BF16)
    hipLaunchKernelGGL(layer_norm_vjp_kernel<bf16_t>, grid, dim3(256), 0, ctx->stream, (const bf16_t*)x, stats, gamma, (const bf16_t*)dy, (const bf16_t*)add, (bf16_t*)dx, rows, C, 0);
  else
    hipLaunchKernelGGL(layer_norm_vjp_kernel<float>, grid, dim3(256), 0, ctx->stream, (const float*)x, stats, gamma, (const float*)dy, (const float*)add, (float*)dx, rows, C, 0);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // extern "C"
