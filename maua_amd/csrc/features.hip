// Remaining audio features (first batch) — once-per-clip kernels, HBM/latency-bound by nature.
//
// Replaces (reference, selfsupervised/features/audio.py): mfcc :65-70 (dct = rosa/spectral.py:35-56, as a cosine
// GEMM), tonnetz :50-62 (phi @ chroma), spectral_flatness :118-126, spectral_contrast :76-115 (sorted sub-band
// means), drop_strength :40-41 / processing.py:133-139 emphasize.
// Spectra arrive in this library's frame-major layout spec[frame][n_bins] (complex64), like audio.hip.
#include "common.h"

namespace maua {

// ---------------------------------------------------------------------------------------------- C = A x B^T
// A [M][K], B [N][K], C [M][N] (f32, row-major): 16 x 16 output tile per workgroup, K staged in 32-column chunks.
__global__ __launch_bounds__(256) void matmul_nt_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                        float* __restrict__ Cm, int M, int N, int K) {
  __shared__ float as[16][33], bs[16][33];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m = blockIdx.y * 16 + ty, n = blockIdx.x * 16 + tx;
  float acc = 0.f;
  for (int k0 = 0; k0 < K; k0 += 32) {
    for (int i = threadIdx.x; i < 16 * 32; i += 256) {
      const int r = i >> 5, c = i & 31;
      const int am = blockIdx.y * 16 + r, bn = blockIdx.x * 16 + r;
      as[r][c] = (am < M && k0 + c < K) ? A[(long)am * K + k0 + c] : 0.f;
      bs[r][c] = (bn < N && k0 + c < K) ? B[(long)bn * K + k0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 32; c++) acc = fmaf(as[ty][c], bs[tx][c], acc);
    __syncthreads();
  }
  if (m < M && n < N) Cm[(long)m * N + n] = acc;
}

// ---------------------------------------------------------------------------------------------- flatness
// out[f] = exp(mean_bins(log(max(amin, |z|^power)))) / mean_bins(max(amin, |z|^power)); one workgroup per frame.
__global__ __launch_bounds__(256) void flatness_kernel(const float2* __restrict__ spec, int n_bins, float amin,
                                                       float power, float* __restrict__ out) {
  __shared__ float red[2][256];
  const float2* row = spec + (long)blockIdx.x * n_bins;
  float sl = 0.f, sa = 0.f;
  for (int i = threadIdx.x; i < n_bins; i += 256) {
    const float2 z = row[i];
    const float mag = sqrtf(z.x * z.x + z.y * z.y);
    const float v = fmaxf(amin, power == 2.0f ? mag * mag : powf(mag, power));
    sl += logf(v);
    sa += v;
  }
  red[0][threadIdx.x] = sl;
  red[1][threadIdx.x] = sa;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {  // fixed tree: deterministic
    if ((int)threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = expf(red[0][0] / (float)n_bins) / (red[1][0] / (float)n_bins);
}

// ---------------------------------------------------------------------------------------------- sorted band means
// For frame f: sort |spec[f][lo..hi)| ascending (bitonic in LDS, <= 1024 bins), valley = mean of the first k,
// peak = mean of the last k (summed in ascending order like torch.mean over the sorted slice).
__global__ __launch_bounds__(512) void band_sorted_means_kernel(const float2* __restrict__ spec, int n_bins, int lo,
                                                                int hi, int k, float* __restrict__ valley,
                                                                float* __restrict__ peak) {
  __shared__ float v[1024];
  const float2* row = spec + (long)blockIdx.x * n_bins;
  const int n = hi - lo;
  for (int i = threadIdx.x; i < 1024; i += 512) {
    float x = __builtin_huge_valf();
    if (i < n) {
      const float2 z = row[lo + i];
      x = sqrtf(z.x * z.x + z.y * z.y);
    }
    v[i] = x;
  }
  __syncthreads();
  for (int size = 2; size <= 1024; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < 512; t += 512) {
        const int i = 2 * t - (t & (stride - 1));
        const int j = i + stride;
        const bool up = (i & size) == 0;
        const float a = v[i], b = v[j];
        if ((a > b) == up) { v[i] = b; v[j] = a; }
      }
      __syncthreads();
    }
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < k; i++) s += v[i];
    valley[blockIdx.x] = s / (float)k;
    s = 0.f;
    for (int i = n - k; i < n; i++) s += v[i];
    peak[blockIdx.x] = s / (float)k;
  }
}

// ---------------------------------------------------------------------------------------------- min / max, emphasize
__global__ __launch_bounds__(1024) void minmax_kernel(const float* __restrict__ x, long n, float* __restrict__ out2) {
  __shared__ float mn[1024], mx[1024];
  float a = __builtin_huge_valf(), b = -__builtin_huge_valf();
  for (long i = threadIdx.x; i < n; i += 1024) {
    const float v = x[i];
    a = fminf(a, v);
    b = fmaxf(b, v);
  }
  mn[threadIdx.x] = a;
  mx[threadIdx.x] = b;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      mn[threadIdx.x] = fminf(mn[threadIdx.x], mn[threadIdx.x + o]);
      mx[threadIdx.x] = fmaxf(mx[threadIdx.x], mx[threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out2[0] = mn[0]; out2[1] = mx[0]; }
}

// processing.py:133-139 with xn = (x - min) / max(x - min) already formed: y = xn * (1 + tanh(s * (xn - q))) * range + min
__global__ __launch_bounds__(256) void emphasize_kernel(const float* __restrict__ xn, long n, const float* __restrict__ mm,
                                                        const float* __restrict__ q, float strength, float* __restrict__ y) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float lo = mm[0], range = mm[1] - mm[0];
  const float v = xn[i];
  const float e = v * (1.0f + tanhf(strength * (v - q[0])));
  y[i] = e * range + lo;
}

// signal.py:84-100 compress / expand (before the normalise): y = x * ratio where x > threshold (invert: x < threshold)
__global__ __launch_bounds__(256) void threshold_scale_kernel(const float* __restrict__ x, long n, float threshold,
                                                              float ratio, int invert, float* __restrict__ y) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  const bool hit = invert ? v < threshold : v > threshold;
  y[i] = hit ? v * ratio : v;
}

// latent.py:46-51: mode 0 eerp a^(1-t) * b^t ; mode 1 copeerp a^t * (1 - b^t) / (1 - a^t + b^t); same-shape operands
__global__ __launch_bounds__(256) void eerp_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                   const float* __restrict__ t, long n, int mode, float* __restrict__ y) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float av = a[i], bv = b[i], tv = t[i];
  if (mode == 0) {
    y[i] = powf(av, 1.f - tv) * powf(bv, tv);
  } else {
    const float at = powf(av, tv), bt = powf(bv, tv);
    y[i] = at * (1.f - bt) / (1.f - at + bt);
  }
}

// ---------------------------------------------------------------------------------------------- tempo
// Windowed local autocorrelation of an onset envelope = librosa's autocorrelation tempogram (librosa.feature.tempogram,
// published algorithm; the reference calls it through rosa.beat.tempo in selfsupervised/mir.py:27-30):
//   ac[f][l] = sum_{n=0}^{W-1-l} (w[n] e[f+n]) (w[n+l] e[f+n+l]),  e = the envelope padded by W/2 linear-ramp samples.
// One workgroup per frame: the windowed frame sits in LDS, each thread owns lags l, l + 256, ...
__global__ __launch_bounds__(256) void autocorr_frames_kernel(const float* __restrict__ env_padded,
                                                              const float* __restrict__ window, int W, int n_lags,
                                                              float* __restrict__ ac) {
  extern __shared__ float fr[];
  const float* e = env_padded + blockIdx.x;
  for (int n = threadIdx.x; n < W; n += 256) fr[n] = window[n] * e[n];
  __syncthreads();
  for (int l = threadIdx.x; l < n_lags; l += 256) {
    float a0 = 0.f, a1 = 0.f;  // two chains; fixed order -> deterministic
    int n = 0;
    for (; n + 1 < W - l; n += 2) {
      a0 = fmaf(fr[n], fr[n + l], a0);
      a1 = fmaf(fr[n + 1], fr[n + 1 + l], a1);
    }
    if (n < W - l) a0 = fmaf(fr[n], fr[n + l], a0);
    ac[(long)blockIdx.x * n_lags + l] = a0 + a1;
  }
}

// ---------------------------------------------------------------------------------------------- constant-Q support
// out[i] = scale * sum_k taps[k] * x[i * stride + k - left]  (zero outside [0, n)): the polyphase form of a sinc
// resampler for integer down-sampling (rosa/constantq.py:92 resample(y, sr, sr / 2): torchaudio's kernel, one phase).
__global__ __launch_bounds__(256) void fir_decimate_kernel(const float* __restrict__ x, long n, const float* __restrict__ taps,
                                                           int ntaps, int stride, int left, float scale,
                                                           float* __restrict__ out, long n_out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  const long base = i * stride - left;
  float acc = 0.f;
  for (int k = 0; k < ntaps; k++) {
    const long j = base + k;
    if (j >= 0 && j < n) acc = fmaf(taps[k], x[j], acc);
  }
  out[i] = acc * scale;
}

// torch.istft's overlap-add for already windowed time frames [n_frames][W] at any hop: y[t] = sum_f frames[f][p - f hop] /
// sum_f win[p - f hop]^2 with p = t + start (rosa/spectral.py:24-32; used by the non-power-of-two tempogram of short clips,
// whose DFT runs as a GEMM).
__global__ __launch_bounds__(256) void overlap_add_kernel(const float* __restrict__ frames, int n_frames, int W, int hop,
                                                          const float* __restrict__ win, long start, long length,
                                                          float* __restrict__ y) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= length) return;
  const long p = t + start;
  long f_hi = p / hop, f_lo = p - W + 1 <= 0 ? 0 : (p - W + hop) / hop;
  if (f_hi > n_frames - 1) f_hi = n_frames - 1;
  float num = 0.f, den = 0.f;
  for (long f = f_lo; f <= f_hi; f++) {
    const long n = p - f * hop;
    num += frames[f * W + n];
    den = fmaf(win[n], win[n], den);
  }
  y[t] = den > 1e-11f ? num / den : 0.f;
}

// ------------------------------------------------------------------------------------ IIR second-order sections
// scipy.signal.sosfilt (audioreactive/audio.py:96-112: Butterworth low / high / band pass over the decoded clip) and the biquads of
// processing.py:142-151, as a blocked linear recurrence.  One section in scipy's direct form II transposed,
//   y = b0 x + z0;  z0' = b1 x - a1 y + z1;  z1' = b2 x - a2 y          (float64, the same operation order, no contraction)
// is linear in its state: after L samples z = P z_start + f with P = A^L, A = [[-a1, 1], [-a2, 0]] and f the end state of the same
// chunk started from zero.  (1) every chunk's f in parallel, (2) the chunk start states by a two-level scan, (3) every chunk
// again from its true start state - within a chunk the arithmetic is scipy's, sample for sample.
struct SosSection {
  double b0, b1, b2, a1, a2;
  double P[4];    // A^L
  double Pm[4];   // P^m: m chunks per scan thread
};

__device__ __forceinline__ double sos_step(const SosSection& c, double x, double& z0, double& z1) {
  const double y = __dadd_rn(__dmul_rn(c.b0, x), z0);
  z0 = __dadd_rn(__dsub_rn(__dmul_rn(c.b1, x), __dmul_rn(c.a1, y)), z1);
  z1 = __dsub_rn(__dmul_rn(c.b2, x), __dmul_rn(c.a2, y));
  return y;
}

__global__ __launch_bounds__(64) void sos_chunk_kernel(const double* __restrict__ x, long n, int L, long n_chunks, SosSection c,
                                                       double* __restrict__ f) {
  const long ch = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= n_chunks) return;
  const long lo = ch * L, hi = lo + L < n ? lo + L : n;
  double z0 = 0.0, z1 = 0.0;
  for (long i = lo; i < hi; i++) sos_step(c, x[i], z0, z1);
  f[2 * ch] = z0;
  f[2 * ch + 1] = z1;
}

// start state of every chunk: S[0] = 0, S[c + 1] = P S[c] + f[c].  256 threads, m consecutive chunks each.
__global__ __launch_bounds__(256) void sos_boundary_kernel(const double* __restrict__ f, long n_chunks, int m, SosSection c,
                                                           double* __restrict__ S) {
  __shared__ double run[256][2];
  const int t = threadIdx.x;
  const long lo = (long)t * m, hi = lo + m < n_chunks ? lo + m : n_chunks;
  double v0 = 0.0, v1 = 0.0;
  for (long k = lo; k < hi; k++) {
    const double w0 = c.P[0] * v0 + c.P[1] * v1 + f[2 * k], w1 = c.P[2] * v0 + c.P[3] * v1 + f[2 * k + 1];
    v0 = w0, v1 = w1;
  }
  run[t][0] = v0, run[t][1] = v1;
  __syncthreads();
  if (t == 0) {
    double r0 = 0.0, r1 = 0.0;
    for (int k = 0; k < 256; k++) {
      const double e0 = run[k][0], e1 = run[k][1];
      run[k][0] = r0, run[k][1] = r1;                       // state at the start of thread k's run
      const double w0 = c.Pm[0] * r0 + c.Pm[1] * r1 + e0, w1 = c.Pm[2] * r0 + c.Pm[3] * r1 + e1;
      r0 = w0, r1 = w1;
    }
  }
  __syncthreads();
  v0 = run[t][0], v1 = run[t][1];
  for (long k = lo; k < hi; k++) {
    S[2 * k] = v0, S[2 * k + 1] = v1;
    const double w0 = c.P[0] * v0 + c.P[1] * v1 + f[2 * k], w1 = c.P[2] * v0 + c.P[3] * v1 + f[2 * k + 1];
    v0 = w0, v1 = w1;
  }
}

__global__ __launch_bounds__(64) void sos_apply_kernel(const double* x, long n, int L, long n_chunks, SosSection c,
                                                       const double* __restrict__ S, double* y) {   // x may alias y
  const long ch = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= n_chunks) return;
  const long lo = ch * L, hi = lo + L < n ? lo + L : n;
  double z0 = S[2 * ch], z1 = S[2 * ch + 1];
  for (long i = lo; i < hi; i++) y[i] = sos_step(c, x[i], z0, z1);
}

// torchaudio.functional.contrast (processing.py:154-155 contrast_enhance): sin(t + amount / 750 sin(4 t)), t = x pi / 2.
__global__ __launch_bounds__(256) void contrast_kernel(const float* __restrict__ x, long n, float contrast, float* __restrict__ y) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float t1 = __fmul_rn(x[i], 1.57079632679489661923f);
  const float t2 = __fmul_rn(contrast, sinf(__fmul_rn(t1, 4.0f)));
  y[i] = sinf(__fadd_rn(t1, t2));
}

// y = step(spline(x)): the soft quantiser of chroma_cens (rosa/spectral.py:164-232).  spline = piecewise cubic
// a + f (b + f (c + f d)), f = x - knot[idx], idx = (number of knots < x) - 1 clamped to [0, nk - 2] (torch.bucketize);
// step(w) = h (floor(w - 0.5) + 1 / (2 m) / (1 + exp(-2 alpha r(w)))), r(w) = (w - 0.5) - floor(w - 0.5) - 0.5.
__global__ __launch_bounds__(256) void spline_step_kernel(const float* __restrict__ x, long n, const float* __restrict__ knots,
                                                          const float* __restrict__ coef /* [4][nk-1]: a, b, c, d */,
                                                          int nk, float h, float alpha, float inv2m, int apply_step,
                                                          float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float t = x[i];
  int lo = 0, hi = nk;  // first index with knots[idx] >= t  (bucketize, right = False)
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (knots[mid] < t) lo = mid + 1; else hi = mid;
  }
  int idx = lo - 1;
  idx = idx < 0 ? 0 : (idx > nk - 2 ? nk - 2 : idx);
  const float f = t - knots[idx];
  const int m = nk - 1;
  float w = coef[idx] + (coef[m + idx] + (coef[2 * m + idx] + coef[3 * m + idx] * f) * f) * f;
  if (apply_step) {
    const float fl = floorf(w - 0.5f);
    const float r = (w - 0.5f) - fl - 0.5f;
    w = h * (fl + inv2m / (1.f + expf(-2.f * alpha * r)));
  }
  out[i] = w;
}

// piptrack core (rosa/pitch.py:27-87) on the frame-major magnitude S [n_frames][n_bins]: per bin the parabolic
// interpolation shift, the local-max / threshold / band test, pitch (Hz) and interpolated magnitude (0 where rejected).
__global__ __launch_bounds__(256) void piptrack_kernel(const float* __restrict__ S, int n_frames, int n_bins,
                                                       const float* __restrict__ frame_max, float threshold,
                                                       const float* __restrict__ freqs, float bin_hz, float fmin,
                                                       float fmax, float* __restrict__ pitch, float* __restrict__ mag) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n_frames * n_bins) return;
  const int f = (int)(idx / n_bins), b = (int)(idx - (long)f * n_bins);
  const float* row = S + (long)f * n_bins;
  const float s0 = row[b];
  const float ref = threshold * frame_max[f];
  auto gated = [&](int k) { return (k < 0 || k >= n_bins) ? 0.f : (row[k] > ref ? row[k] : 0.f); };
  float p = 0.f, m = 0.f;
  const float freq = freqs[b];    // the caller's torch.linspace(0, sr / 2, n_bins): compared exactly like the reference
  const float g = gated(b);
  const bool lmax = g > gated(b - 1) && g >= gated(b + 1);
  if (lmax && fmin <= freq && freq < fmax) {
    float avg = 0.f, shift = 0.f;
    if (b >= 1 && b + 1 < n_bins) {
      avg = 0.5f * (row[b + 1] - row[b - 1]);
      float sh = 2.f * s0 - row[b + 1] - row[b - 1];
      sh = sh + (fabsf(sh) < 1.17549435e-38f ? 1.f : 0.f);
      shift = avg / sh;
    }
    p = (b + shift) * bin_hz;
    m = s0 + 0.5f * avg * shift;
  }
  pitch[idx] = p;
  mag[idx] = m;
}

}  // namespace maua

using namespace maua;

extern "C" {

int maua_matmul_nt(maua_ctx* ctx, const float* a, const float* b, float* c, int M, int N, int K) {
  MAUA_REQUIRE(ctx, "maua_matmul_nt: ctx is NULL");
  if (M == 0 || N == 0) return MAUA_OK;
  MAUA_REQUIRE(a && b && c && K > 0, "maua_matmul_nt: NULL argument");
  hipLaunchKernelGGL(matmul_nt_kernel, dim3((N + 15) / 16, (M + 15) / 16), dim3(256), 0, ctx->stream, a, b, c, M, N, K);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_spectral_flatness(maua_ctx* ctx, const float* spec, int n_frames, int n_bins, float amin, float power,
                           float* out) {
  MAUA_REQUIRE(ctx, "maua_spectral_flatness: ctx is NULL");
  if (n_frames == 0) return MAUA_OK;
  MAUA_REQUIRE(spec && out && n_bins > 0, "maua_spectral_flatness: NULL argument");
  hipLaunchKernelGGL(flatness_kernel, dim3(n_frames), dim3(256), 0, ctx->stream,
                     reinterpret_cast<const float2*>(spec), n_bins, amin, power, out);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_band_sorted_means(maua_ctx* ctx, const float* spec, int n_frames, int n_bins, int lo, int hi, int k,
                           float* valley, float* peak) {
  MAUA_REQUIRE(ctx, "maua_band_sorted_means: ctx is NULL");
  if (n_frames == 0) return MAUA_OK;
  MAUA_REQUIRE(spec && valley && peak, "maua_band_sorted_means: NULL argument");
  MAUA_REQUIRE(0 <= lo && lo < hi && hi <= n_bins && hi - lo <= 1024, "maua_band_sorted_means: band must hold 1..1024 bins");
  MAUA_REQUIRE(k >= 1 && k <= hi - lo, "maua_band_sorted_means: k out of range");
  hipLaunchKernelGGL(band_sorted_means_kernel, dim3(n_frames), dim3(512), 0, ctx->stream,
                     reinterpret_cast<const float2*>(spec), n_bins, lo, hi, k, valley, peak);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_minmax(maua_ctx* ctx, const float* x, long n, float* out2) {
  MAUA_REQUIRE(ctx && x && out2 && n > 0, "maua_minmax: NULL or empty argument");
  hipLaunchKernelGGL(minmax_kernel, dim3(1), dim3(1024), 0, ctx->stream, x, n, out2);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_emphasize(maua_ctx* ctx, const float* xn, long n, const float* minmax_dev, const float* q_dev, float strength,
                   float* y) {
  MAUA_REQUIRE(ctx, "maua_emphasize: ctx is NULL");
  if (n == 0) return MAUA_OK;
  MAUA_REQUIRE(xn && minmax_dev && q_dev && y, "maua_emphasize: NULL argument");
  hipLaunchKernelGGL(emphasize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, xn, n, minmax_dev,
                     q_dev, strength, y);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_threshold_scale(maua_ctx* ctx, const float* x, long n, float threshold, float ratio, int invert, float* y) {
  MAUA_REQUIRE(ctx, "maua_threshold_scale: ctx is NULL");
  if (n == 0) return MAUA_OK;
  MAUA_REQUIRE(x && y, "maua_threshold_scale: NULL argument");
  hipLaunchKernelGGL(threshold_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, x, n, threshold,
                     ratio, invert, y);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_eerp(maua_ctx* ctx, const float* a, const float* b, const float* t, long n, int mode, float* y) {
  MAUA_REQUIRE(ctx, "maua_eerp: ctx is NULL");
  if (n == 0) return MAUA_OK;
  MAUA_REQUIRE(a && b && t && y && (mode == 0 || mode == 1), "maua_eerp: NULL argument or unknown mode");
  hipLaunchKernelGGL(eerp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a, b, t, n, mode, y);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // extern "C"

extern "C" int maua_autocorr_frames(maua_ctx* ctx, const float* env_padded, const float* window, int n_frames, int win,
                                    int n_lags, float* ac) {
  MAUA_REQUIRE(ctx && env_padded && window && ac, "maua_autocorr_frames: NULL argument");
  MAUA_REQUIRE(win >= 1 && n_lags >= 1 && n_lags <= win && (size_t)win * 4 <= 64 * 1024, "maua_autocorr_frames: bad window");
  if (n_frames == 0) return MAUA_OK;
  hipLaunchKernelGGL(maua::autocorr_frames_kernel, dim3(n_frames), dim3(256), (size_t)win * 4, ctx->stream, env_padded,
                     window, win, n_lags, ac);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

extern "C" int maua_fir_decimate(maua_ctx* ctx, const float* x, long n, const float* taps, int ntaps, int stride, int left,
                                 float scale, float* out, long n_out) {
  MAUA_REQUIRE(ctx && x && taps && out, "maua_fir_decimate: NULL argument");
  MAUA_REQUIRE(ntaps >= 1 && stride >= 1 && n >= 0 && n_out >= 0, "maua_fir_decimate: bad sizes");
  if (n_out == 0) return MAUA_OK;
  hipLaunchKernelGGL(maua::fir_decimate_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, ctx->stream, x, n, taps,
                     ntaps, stride, left, scale, out, n_out);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

extern "C" int maua_overlap_add(maua_ctx* ctx, const float* frames, int n_frames, int W, int hop, const float* window,
                                long start, long length, float* y) {
  MAUA_REQUIRE(ctx && frames && window && y, "maua_overlap_add: NULL argument");
  MAUA_REQUIRE(n_frames >= 1 && W >= 1 && hop >= 1 && start >= 0 && length >= 0, "maua_overlap_add: bad sizes");
  if (length == 0) return MAUA_OK;
  hipLaunchKernelGGL(maua::overlap_add_kernel, dim3((unsigned)((length + 255) / 256)), dim3(256), 0, ctx->stream, frames,
                     n_frames, W, hop, window, start, length, y);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

static void mat2_mul(const double* a, const double* b, double* o) {
  const double r[4] = {a[0] * b[0] + a[1] * b[2], a[0] * b[1] + a[1] * b[3], a[2] * b[0] + a[3] * b[2], a[2] * b[1] + a[3] * b[3]};
  for (int i = 0; i < 4; i++) o[i] = r[i];
}
static void mat2_pow(const double* a, long e, double* o) {
  double base[4] = {a[0], a[1], a[2], a[3]}, acc[4] = {1.0, 0.0, 0.0, 1.0};
  for (; e > 0; e >>= 1) {
    if (e & 1) mat2_mul(acc, base, acc);
    mat2_mul(base, base, base);
  }
  for (int i = 0; i < 4; i++) o[i] = acc[i];
}

extern "C" int maua_sosfilt(maua_ctx* ctx, const double* sos_host, int n_sections, const double* x, long n, double* y) {
  MAUA_REQUIRE(ctx, "maua_sosfilt: ctx is NULL");
  MAUA_REQUIRE(n_sections >= 1 && sos_host, "maua_sosfilt: need at least one section");
  if (n == 0) return MAUA_OK;
  MAUA_REQUIRE(x && y && n > 0, "maua_sosfilt: NULL argument");
  const int L = 128;
  const long n_chunks = (n + L - 1) / L;
  const int m = (int)((n_chunks + 255) / 256);
  if (int rc = maua::scratch_reserve(ctx, (size_t)n_chunks * 4 * sizeof(double))) return rc;
  double* f = (double*)ctx->scratch;
  double* S = f + 2 * n_chunks;
  const unsigned grid = (unsigned)((n_chunks + 63) / 64);
  for (int k = 0; k < n_sections; k++) {
    const double* r = sos_host + 6 * k;
    MAUA_REQUIRE(r[3] != 0.0, "maua_sosfilt: a0 of a section is zero");
    maua::SosSection c;
    c.b0 = r[0] / r[3], c.b1 = r[1] / r[3], c.b2 = r[2] / r[3], c.a1 = r[4] / r[3], c.a2 = r[5] / r[3];
    const double A[4] = {-c.a1, 1.0, -c.a2, 0.0};
    mat2_pow(A, L, c.P);
    mat2_pow(c.P, m, c.Pm);
    const double* src = k == 0 ? x : y;
    hipLaunchKernelGGL(maua::sos_chunk_kernel, dim3(grid), dim3(64), 0, ctx->stream, src, n, L, n_chunks, c, f);
    hipLaunchKernelGGL(maua::sos_boundary_kernel, dim3(1), dim3(256), 0, ctx->stream, (const double*)f, n_chunks, m, c, S);
    hipLaunchKernelGGL(maua::sos_apply_kernel, dim3(grid), dim3(64), 0, ctx->stream, src, n, L, n_chunks, c, (const double*)S, y);
  }
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

extern "C" int maua_contrast(maua_ctx* ctx, const float* x, long n, float enhancement_amount, float* y) {
  MAUA_REQUIRE(ctx, "maua_contrast: ctx is NULL");
  if (n == 0) return MAUA_OK;
  MAUA_REQUIRE(x && y, "maua_contrast: NULL argument");
  MAUA_REQUIRE(enhancement_amount >= 0.f && enhancement_amount <= 100.f, "maua_contrast: enhancement_amount outside [0, 100]");
  hipLaunchKernelGGL(maua::contrast_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, x, n,
                     enhancement_amount / 750.0f, y);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

extern "C" int maua_spline_step(maua_ctx* ctx, const float* x, long n, const float* knots, const float* coef, int n_knots,
                                float h, float alpha, int apply_step, float* out) {
  MAUA_REQUIRE(ctx && x && knots && coef && out, "maua_spline_step: NULL argument");
  MAUA_REQUIRE(n_knots >= 2, "maua_spline_step: need at least two knots");
  if (n == 0) return MAUA_OK;
  const float m = 1.f / (1.f + expf(-alpha)) - 0.5f;
  hipLaunchKernelGGL(maua::spline_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, x, n, knots, coef,
                     n_knots, h, alpha, 1.f / (2.f * m), apply_step, out);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

extern "C" int maua_piptrack(maua_ctx* ctx, const float* mag_frames_bins, int n_frames, int n_bins, const float* frame_max,
                             float threshold, const float* freqs, float bin_hz, float fmin, float fmax, float* pitch,
                             float* mag) {
  MAUA_REQUIRE(ctx && mag_frames_bins && frame_max && freqs && pitch && mag, "maua_piptrack: NULL argument");
  const long n = (long)n_frames * n_bins;
  if (n == 0) return MAUA_OK;
  hipLaunchKernelGGL(maua::piptrack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, mag_frames_bins,
                     n_frames, n_bins, frame_max, threshold, freqs, bin_hz, fmin, fmax, pitch, mag);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// ---- mapping-network prologue / epilogue (inference/stylegan2.py:142-143, :183): rows of a [P][D] matrix ----------------
namespace {
// y[p][:] = x[p][:] / sqrt(mean(x[p][:]^2) + eps): one wave per row, fixed summation order (lane-strided partial sums,
// xor-butterfly)
__global__ __launch_bounds__(64) void normalize_2nd_moment_kernel(const float* __restrict__ x, float* __restrict__ y, int D,
                                                                   float eps) {
  const float* xr = x + (long)blockIdx.x * D;
  float s = 0.f;
  for (int i = threadIdx.x; i < D; i += 64) s += xr[i] * xr[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float inv = 1.0f / sqrtf(s / (float)D + eps);
  for (int i = threadIdx.x; i < D; i += 64) y[(long)blockIdx.x * D + i] = xr[i] * inv;
}
// out[p][k][:] = x[p][:] for k < n  (ws = w.unsqueeze(1).repeat(1, num_ws, 1))
__global__ __launch_bounds__(256) void repeat_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int D, int n,
                                                          long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const long row = idx / D;
  out[idx] = x[(row / n) * D + (idx - row * D)];
}
}  // namespace

extern "C" {

int maua_normalize_2nd_moment(maua_ctx* ctx, const float* x, int P, int D, float eps, float* y) {
  MAUA_REQUIRE(ctx && x && y && P >= 0 && D > 0, "maua_normalize_2nd_moment: bad argument");
  if (P == 0) return MAUA_OK;
  hipLaunchKernelGGL(normalize_2nd_moment_kernel, dim3(P), dim3(64), 0, ctx->stream, x, y, D, eps);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_repeat_rows(maua_ctx* ctx, const float* x, int P, int D, int n, float* out) {
  MAUA_REQUIRE(ctx && x && out && P >= 0 && D > 0 && n > 0, "maua_repeat_rows: bad argument");
  const long total = (long)P * n * D;
  if (total == 0) return MAUA_OK;
  hipLaunchKernelGGL(repeat_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, x, out, D, n, total);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // extern "C"
