// Latent schedule kernels (once per clip, whole [T][num_ws*w_dim] tensors; HBM-bound streaming).
//
// Replaces (reference): maua/audiovisual/audioreactive/latent.py single_weighted :12-18, multi_weighted :21-31,
// select_modulo :34-43, slerp :54-65, slerp_loops :68-80, spline_loops :83-92; signal.py resample :5-24;
// selfsupervised/latent.py spline_loop_latents :7-13, latent_patch merges :57-78.
// The cubic spline of the un-vendored torchcubicspline is the natural cubic spline (unique interpolant):
// second derivatives by a tridiagonal (Thomas) solve per column in f64, piecewise-cubic evaluation.
#include <cmath>
#include <vector>

#include "common.h"
#include "internal.h"

namespace maua {

// ---- natural cubic spline -------------------------------------------------------------------------------------
// forward sweep / back substitution per column c; cp, inv_den, h depend on the knots only (host, f64).
__global__ __launch_bounds__(256) void spline_solve_kernel(const float* __restrict__ y, int n, long C,
                                                           const double* __restrict__ h, const double* __restrict__ cp,
                                                           const double* __restrict__ inv_den, double* __restrict__ M) {
  long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  M[c] = 0.0;  // row 0
  double dprev = 0.0;
  for (int i = 1; i < n - 1; i++) {
    double y0 = y[(long)(i - 1) * C + c], y1 = y[(long)i * C + c], y2 = y[(long)(i + 1) * C + c];
    double rhs = 6.0 * ((y2 - y1) / h[i] - (y1 - y0) / h[i - 1]);
    double d = (rhs - h[i - 1] * dprev) * inv_den[i];
    M[(long)i * C + c] = d;
    dprev = d;
  }
  M[(long)(n - 1) * C + c] = 0.0;
  double mnext = 0.0;
  for (int i = n - 2; i >= 1; i--) {
    double m = M[(long)i * C + c] - cp[i] * mnext;
    M[(long)i * C + c] = m;
    mnext = m;
  }
}

__global__ __launch_bounds__(256) void spline_eval_kernel(const float* __restrict__ y, const double* __restrict__ M,
                                                          long C, const int* __restrict__ idx,
                                                          const double* __restrict__ aw, const double* __restrict__ bw,
                                                          const double* __restrict__ hw, float* __restrict__ out) {
  long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int t = blockIdx.y;
  if (c >= C) return;
  int i = idx[t];
  double a = aw[t], b = bw[t], h = hw[t];
  double m0 = M[(long)i * C + c], m1 = M[(long)(i + 1) * C + c];
  double y0 = y[(long)i * C + c], y1 = y[(long)(i + 1) * C + c];
  double v = (m0 * a * a * a + m1 * b * b * b) / (6.0 * h) + (y0 / h - m0 * h / 6.0) * a + (y1 / h - m1 * h / 6.0) * b;
  out[(long)t * C + c] = (float)v;
}

// ---- envelope-weighted blends ---------------------------------------------------------------------------------
// out[t] = a[t] * (1 - e[t]) + b[t] * e[t]   (a/b time strides may be 0: constants)
__global__ __launch_bounds__(256) void blend_kernel(const float* __restrict__ a, long a_ts, const float* __restrict__ b,
                                                    long b_ts, const float* __restrict__ e, long C,
                                                    float* __restrict__ out) {
  long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int t = blockIdx.y;
  if (c >= C) return;
  float ev = e[t];
  float w0 = __fsub_rn(1.f, ev);
  out[(long)t * C + c] = __fadd_rn(__fmul_rn(a[(long)t * a_ts + c], w0), __fmul_rn(b[(long)t * b_ts + c], ev));
}

// multi_weighted: out[t][c] = sum_a (e[t][a] / sum_a' e[t][a']) * lat[a % n][c]
__global__ __launch_bounds__(256) void weighted_sum_kernel(const float* __restrict__ e, const float* __restrict__ lat,
                                                           int A, int n, long C, float* __restrict__ out) {
  extern __shared__ float es[];
  int t = blockIdx.y;
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int a = 0; a < A; a++) s += e[(long)t * A + a];
    for (int a = 0; a < A; a++) es[a] = e[(long)t * A + a] / s;
  }
  __syncthreads();
  long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float acc = 0.f;
  for (int a = 0; a < A; a++) acc += es[a] * lat[(long)(a % n) * C + c];
  out[(long)t * C + c] = acc;
}

// select_modulo index: idx = round_half_even(x * scale) as int64   (latent.py:38-40)
__global__ __launch_bounds__(256) void scale_round_index_kernel(const float* __restrict__ x, float scale, long n,
                                                                long long* __restrict__ idx) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  idx[i] = (long long)rintf(__fmul_rn(x[i], scale));
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const long long* __restrict__ idx,
                                                          int n_rows, long C, float* __restrict__ out) {
  long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int t = blockIdx.y;
  if (c >= C) return;
  long long r = idx[t];
  if (r < 0) r += n_rows;  // numpy-style negative index (latent.py:41 indexes a numpy array)
  out[(long)t * C + c] = src[r * C + c];
}

// F.interpolate(mode="linear", align_corners=False) along axis 0 of [n][C] -> [size][C]
__global__ __launch_bounds__(256) void resample_linear_kernel(const float* __restrict__ x, int n, long C, int size,
                                                              float* __restrict__ out) {
  long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int j = blockIdx.y;
  if (c >= C) return;
  const float scale = (float)n / (float)size;
  float src = __fsub_rn(__fmul_rn(scale, __fadd_rn((float)j, 0.5f)), 0.5f);
  if (src < 0.f) src = 0.f;
  int i0 = (int)src;
  if (i0 > n - 1) i0 = n - 1;
  int i1 = i0 + (i0 < n - 1 ? 1 : 0);
  float l1 = __fsub_rn(src, (float)i0), l0 = __fsub_rn(1.f, l1);
  out[(long)j * C + c] = __fadd_rn(__fmul_rn(l0, x[(long)i0 * C + c]), __fmul_rn(l1, x[(long)i1 * C + c]));
}

// slerp (latent.py:54-65): one workgroup per (t, segment, layer) vector of length D
__global__ __launch_bounds__(256) void slerp_kernel(const float* __restrict__ y, const float* __restrict__ tv, int n_seg,
                                                    int L, int D, float* __restrict__ out) {
  __shared__ float red[3][4];
  const int ti = blockIdx.z, seg = blockIdx.y, lay = blockIdx.x;
  const float* a = y + ((long)seg * L + lay) * D;
  const float* b = y + ((long)(seg + 1) * L + lay) * D;
  auto bsum = [&](float v, int slot) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[slot][threadIdx.x >> 6] = v;
    __syncthreads();
    float s = red[slot][0] + red[slot][1] + red[slot][2] + red[slot][3];
    __syncthreads();
    return s;
  };
  float na = 0.f, nb = 0.f;
  for (int i = threadIdx.x; i < D; i += blockDim.x) {
    na += a[i] * a[i];
    nb += b[i] * b[i];
  }
  na = sqrtf(bsum(na, 0));
  nb = sqrtf(bsum(nb, 1));
  float dot = 0.f;
  for (int i = threadIdx.x; i < D; i += blockDim.x) dot += (a[i] / na) * (b[i] / nb);
  dot = bsum(dot, 2);
  const float p = tv[ti] * acosf(dot);
  float nc = 0.f;
  for (int i = threadIdx.x; i < D; i += blockDim.x) {
    float cv = b[i] / nb - dot * (a[i] / na);
    nc += cv * cv;
  }
  nc = sqrtf(bsum(nc, 0));
  const float cp = cosf(p), sp = sinf(p);
  float nd = 0.f;
  for (int i = threadIdx.x; i < D; i += blockDim.x) {
    float av = a[i] / na, cv = (b[i] / nb - dot * av) / nc;
    float dv = av * cp + cv * sp;
    nd += dv * dv;
  }
  nd = sqrtf(bsum(nd, 1));
  float* o = out + (((long)ti * n_seg + seg) * L + lay) * D;
  for (int i = threadIdx.x; i < D; i += blockDim.x) {
    float av = a[i] / na, cv = (b[i] / nb - dot * av) / nc;
    o[i] = (av * cp + cv * sp) / nd;
  }
}

// latent_patch merges (selfsupervised/latent.py:57-78) on layers [l0, l1) of lat [T][L][D], in place
__global__ __launch_bounds__(256) void latent_merge_kernel(float* __restrict__ lat, const float* __restrict__ seq,
                                                           const float* __restrict__ mod, int mode, int L, int D,
                                                           int l0, int l1) {
  int t = blockIdx.y;
  long span = (long)(l1 - l0) * D;
  long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= span) return;
  long o = ((long)t * L + l0) * D + c;
  float a = lat[o], s = seq[o];
  float v;
  if (mode == 0) v = __fadd_rn(a, s) / 2.f;  // average: += then /= 2
  else if (mode == 1) {                      // modulate: *= (1 - m); += m * seq
    float m = mod[t];
    v = __fadd_rn(__fmul_rn(a, __fsub_rn(1.f, m)), __fmul_rn(m, s));
  } else v = s;                              // overwrite
  lat[o] = v;
}

}  // namespace maua

using namespace maua;

extern "C" {

int maua_spline_natural(maua_ctx* ctx, const double* t_knots_host, int n, const float* y, long C,
                        const double* t_eval_host, int T, float* out) {
  MAUA_REQUIRE(ctx && t_knots_host && y && t_eval_host && out, "maua_spline_natural: NULL argument");
  MAUA_REQUIRE(n >= 2 && T >= 0 && C >= 0, "maua_spline_natural: need at least two knots");
  if (T == 0 || C == 0) return MAUA_OK;
  std::vector<double> h(n), cp(n, 0.0), inv_den(n, 0.0);
  for (int i = 0; i < n - 1; i++) {
    h[i] = t_knots_host[i + 1] - t_knots_host[i];
    if (!(h[i] > 0)) return fail("maua_spline_natural: knots must be strictly increasing");
  }
  h[n - 1] = h[n - 2];
  // Thomas coefficients of the interior system (natural ends: M0 = Mn-1 = 0)
  double cprev = 0.0;
  for (int i = 1; i < n - 1; i++) {
    double diag = 2.0 * (h[i - 1] + h[i]);
    double den = diag - h[i - 1] * cprev;
    inv_den[i] = 1.0 / den;
    cp[i] = (i < n - 2) ? h[i] / den : 0.0;
    cprev = cp[i];
  }
  std::vector<int> idx(T);
  std::vector<double> aw(T), bw(T), hw(T);
  for (int t = 0; t < T; t++) {
    double te = t_eval_host[t];
    // last knot interval whose left end is <= te (searchsorted right - 1), clipped
    int lo = 0, hi = n;
    while (lo < hi) {
      int mid = (lo + hi) / 2;
      if (t_knots_host[mid] <= te) lo = mid + 1; else hi = mid;
    }
    int i = std::min(std::max(lo - 1, 0), n - 2);
    idx[t] = i;
    aw[t] = t_knots_host[i + 1] - te;
    bw[t] = te - t_knots_host[i];
    hw[t] = h[i];
  }
  // scratch layout: M [n][C] f64 | h | cp | inv_den | aw | bw | hw | idx
  size_t off = 0;
  auto carve = [&](size_t b) { size_t o = off; off += (b + 255) & ~(size_t)255; return o; };
  size_t oM = carve((size_t)n * C * 8), oh = carve(n * 8), ocp = carve(n * 8), oid = carve(n * 8);
  size_t oa = carve(T * 8), ob = carve(T * 8), ohw = carve(T * 8), oix = carve(T * 4);
  if (int rc = scratch_reserve(ctx, off)) return rc;
  char* base = (char*)ctx->scratch;
  hipStream_t s = ctx->stream;
  MAUA_HIP_CHECK(hipMemcpyAsync(base + oh, h.data(), n * 8, hipMemcpyHostToDevice, s));
  MAUA_HIP_CHECK(hipMemcpyAsync(base + ocp, cp.data(), n * 8, hipMemcpyHostToDevice, s));
  MAUA_HIP_CHECK(hipMemcpyAsync(base + oid, inv_den.data(), n * 8, hipMemcpyHostToDevice, s));
  MAUA_HIP_CHECK(hipMemcpyAsync(base + oa, aw.data(), T * 8, hipMemcpyHostToDevice, s));
  MAUA_HIP_CHECK(hipMemcpyAsync(base + ob, bw.data(), T * 8, hipMemcpyHostToDevice, s));
  MAUA_HIP_CHECK(hipMemcpyAsync(base + ohw, hw.data(), T * 8, hipMemcpyHostToDevice, s));
  MAUA_HIP_CHECK(hipMemcpyAsync(base + oix, idx.data(), T * 4, hipMemcpyHostToDevice, s));
  MAUA_HIP_CHECK(hipStreamSynchronize(s));  // host staging vectors go out of scope
  hipLaunchKernelGGL(spline_solve_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, s, y, n, C,
                     (const double*)(base + oh), (const double*)(base + ocp), (const double*)(base + oid),
                     (double*)(base + oM));
  hipLaunchKernelGGL(spline_eval_kernel, dim3((unsigned)((C + 255) / 256), T), dim3(256), 0, s, y,
                     (const double*)(base + oM), C, (const int*)(base + oix), (const double*)(base + oa),
                     (const double*)(base + ob), (const double*)(base + ohw), out);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_latent_blend(maua_ctx* ctx, const float* a, long a_tstride, const float* b, long b_tstride, const float* env,
                      int T, long C, float* out) {
  MAUA_REQUIRE(ctx, "maua_latent_blend: ctx is NULL");
  if (T == 0 || C == 0) return MAUA_OK;
  MAUA_REQUIRE(a && b && env && out, "maua_latent_blend: NULL argument");
  hipLaunchKernelGGL(blend_kernel, dim3((unsigned)((C + 255) / 256), T), dim3(256), 0, ctx->stream, a, a_tstride, b,
                     b_tstride, env, C, out);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_weighted_sum(maua_ctx* ctx, const float* env, const float* latents, int T, int A, int n, long C, float* out) {
  MAUA_REQUIRE(ctx, "maua_weighted_sum: ctx is NULL");
  if (T == 0 || C == 0) return MAUA_OK;
  MAUA_REQUIRE(env && latents && out && A > 0 && n > 0, "maua_weighted_sum: NULL argument");
  hipLaunchKernelGGL(weighted_sum_kernel, dim3((unsigned)((C + 255) / 256), T), dim3(256), (size_t)A * sizeof(float),
                     ctx->stream, env, latents, A, n, C, out);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_scale_round_index(maua_ctx* ctx, const float* x, float scale, long n, long long* idx) {
  MAUA_REQUIRE(ctx, "maua_scale_round_index: ctx is NULL");
  if (n == 0) return MAUA_OK;
  MAUA_REQUIRE(x && idx, "maua_scale_round_index: NULL argument");
  hipLaunchKernelGGL(scale_round_index_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, x, scale, n,
                     idx);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_gather_rows(maua_ctx* ctx, const float* src, const long long* idx, int n_rows, int T, long C, float* out) {
  MAUA_REQUIRE(ctx, "maua_gather_rows: ctx is NULL");
  if (T == 0 || C == 0) return MAUA_OK;
  MAUA_REQUIRE(src && idx && out, "maua_gather_rows: NULL argument");
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((C + 255) / 256), T), dim3(256), 0, ctx->stream, src, idx,
                     n_rows, C, out);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_resample_linear(maua_ctx* ctx, const float* x, int n, long C, int size, float* out) {
  MAUA_REQUIRE(ctx, "maua_resample_linear: ctx is NULL");
  if (size == 0 || C == 0) return MAUA_OK;
  MAUA_REQUIRE(x && out && n > 0, "maua_resample_linear: NULL argument");
  hipLaunchKernelGGL(resample_linear_kernel, dim3((unsigned)((C + 255) / 256), size), dim3(256), 0, ctx->stream, x, n, C,
                     size, out);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_slerp(maua_ctx* ctx, const float* y, const float* t, int k, int n_seg, int L, int D, float* out) {
  MAUA_REQUIRE(ctx, "maua_slerp: ctx is NULL");
  if (k == 0 || n_seg == 0) return MAUA_OK;
  MAUA_REQUIRE(y && t && out && L > 0 && D > 0, "maua_slerp: NULL argument");
  hipLaunchKernelGGL(slerp_kernel, dim3(L, n_seg, k), dim3(256), 0, ctx->stream, y, t, n_seg, L, D, out);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_latent_merge(maua_ctx* ctx, float* latents, const float* sequence, const float* mod, int mode, int T, int L,
                      int D, int l0, int l1) {
  MAUA_REQUIRE(ctx, "maua_latent_merge: ctx is NULL");
  if (T == 0 || l1 <= l0) return MAUA_OK;
  MAUA_REQUIRE(latents && sequence, "maua_latent_merge: NULL argument");
  MAUA_REQUIRE(mode >= 0 && mode <= 2 && (mode != 1 || mod), "maua_latent_merge: mode 0 average, 1 modulate (needs mod), 2 overwrite");
  MAUA_REQUIRE(l0 >= 0 && l1 <= L, "maua_latent_merge: layer slice out of range");
  long span = (long)(l1 - l0) * D;
  hipLaunchKernelGGL(latent_merge_kernel, dim3((unsigned)((span + 255) / 256), T), dim3(256), 0, ctx->stream, latents,
                     sequence, mod, mode, L, D, l0, l1);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // extern "C"
