// Beat tracking and Laplacian segmentation of the self-supervised front end (reference
// maua/audiovisual/audioreactive/selfsupervised/mir.py:24-45 retrieve_music_information):
//   mir.py:31  rosa.beat.beat_track(onset_envelope, trim=False, hop_length=1024, bpm=tempo)  -> maua_beat_dp
//   features/rosa/segment.py:152-209 laplacian_segmentation                                   -> the kernels below
// librosa is un-vendored and unpinned (setup.py:60): the dynamic program restates its published beat tracker (Ellis 2007;
// librosa.beat.__beat_local_score / __beat_track_dp, 0.8 - 0.10).  Once-per-clip work on a few hundred beats: small
// latency-bound kernels; what matters is that every reduction has a fixed order (the outputs are indices).
#include <cmath>
#include <vector>

#include "common.h"
#include "internal.h"

namespace maua {

// ---------------------------------------------------------------------------------------------- beat tracker
// localscore = scipy.signal.convolve(onsets / std, window, "same") in float64 (float32 envelope x float64 window)
__global__ __launch_bounds__(256) void beat_localscore_kernel(const float* __restrict__ env, const double* __restrict__ window,
                                                              int T, int period, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T) return;
  double acc = 0.0;
  for (int m = -period; m <= period; m++) {   // out[i] = sum_m x[i - m] w[m + period]
    const int j = i - m;
    if (j >= 0 && j < T) acc += (double)env[j] * window[m + period];
  }
  out[i] = acc;
}

__global__ __launch_bounds__(256) void max_f64_kernel(const double* __restrict__ x, int n, double* __restrict__ out) {
  __shared__ double sh[256];
  double m = -INFINITY;
  for (int i = threadIdx.x; i < n; i += 256) m = fmax(m, x[i]);
  sh[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sh[0];
}

// The dynamic program, one wave: frame i looks back over the predecessors i - 2 period .. i - round(period / 2), scores
// them txwt[j] + cumscore[prev] (txwt alone before time 0), takes the FIRST maximum (numpy argmax), and links to it.
constexpr int BEAT_RING = 4096;   // cumulative scores kept in LDS: a ring longer than the look-back window
__global__ __launch_bounds__(64) void beat_dp_kernel(const double* __restrict__ localscore, const double* __restrict__ txwt,
                                                     const double* __restrict__ score_max, int T, int period, int wlen,
                                                     double* __restrict__ cumscore, int* __restrict__ backlink) {
  __shared__ double ring[BEAT_RING];
  const int lane = threadIdx.x;
  const double small = 0.01 * *score_max;
  bool first_beat = true;
  for (int i = 0; i < T; i++) {
    double best = -INFINITY;
    int bj = 0x7fffffff;
    for (int j = lane; j < wlen; j += 64) {
      const int prev = i - 2 * period + j;
      const double c = txwt[j] + (prev >= 0 ? ring[prev & (BEAT_RING - 1)] : 0.0);
      if (c > best || bj == 0x7fffffff) { best = c; bj = j; }   // (a lane's j grow: strict > keeps its first maximum)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ob = __shfl_xor(best, o);
      const int oj = __shfl_xor(bj, o);
      if (oj != 0x7fffffff && (bj == 0x7fffffff || ob > best || (ob == best && oj < bj))) { best = ob; bj = oj; }
    }
    const double score_i = localscore[i];
    const double cs = score_i + best;
    __syncthreads();   // every lane has read the ring
    if (lane == 0) {
      ring[i & (BEAT_RING - 1)] = cs;
      cumscore[i] = cs;
      if (first_beat && score_i < small) {
        backlink[i] = -1;
      } else {
        backlink[i] = i - 2 * period + bj;
      }
    }
    if (!(first_beat && score_i < small)) first_beat = false;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------- segmentation
// beat-synchronous feature: lower median (torch.median) or mean of every channel over frames [bounds[s], bounds[s+1])
__global__ __launch_bounds__(64) void segment_reduce_kernel(const float* __restrict__ x, int C, const int* __restrict__ bounds,
                                                            int mode, float* __restrict__ out) {
  const int s = blockIdx.x, c = blockIdx.y * 64 + threadIdx.x;
  if (c >= C) return;
  const int lo = bounds[s], hi = bounds[s + 1], L = hi - lo;
  if (L <= 0) { out[(long)s * C + c] = NAN; return; }
  const float* col = x + (long)lo * C + c;
  if (mode == 1) {
    float acc = 0.f;
    for (int i = 0; i < L; i++) acc += col[(long)i * C];
    out[(long)s * C + c] = acc / (float)L;
    return;
  }
  const int want = (L - 1) / 2;   // rank of the lower median
  float res = col[0];
  for (int i = 0; i < L; i++) {
    const float v = col[(long)i * C];
    int rank = 0;
    for (int j = 0; j < L; j++) {
      const float u = col[(long)j * C];
      rank += (u < v || (u == v && j < i)) ? 1 : 0;
    }
    if (rank == want) res = v;
  }
  out[(long)s * C + c] = res;
}

// segment.py:23-45: column j keeps its k nearest rows (distances (sum (x - y)^2 + 1e-8)^(1/2); |i - j| < width excluded by
// the 1e20 the reference adds to zeroed entries); one workgroup per column, ranks by counting (ties: lower row first)
__global__ __launch_bounds__(256) void recurrence_topk_kernel(const float* __restrict__ data, int n, int d, int k, int width,
                                                              float* __restrict__ rec) {
  extern __shared__ float dist[];   // [n]
  const int j = blockIdx.x;
  for (int i = threadIdx.x; i < n; i += 256) {
    float acc = 0.f;
    for (int c = 0; c < d; c++) {
      const float t = data[(long)i * d + c] - data[(long)j * d + c];
      acc += t * t;
    }
    float v = sqrtf(acc + 1e-8f);
    const int off = i - j;
    if (off > -width && off < width) v = 0.f;
    if (v == 0.f) v += 1e20f;
    dist[i] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    const float v = dist[i];
    int rank = 0;
    for (int m = 0; m < n; m++) {
      const float u = dist[m];
      rank += (u < v || (u == v && m < i)) ? 1 : 0;
    }
    rec[(long)i * n + j] = rank < k ? v : 0.f;
  }
}

// rec = min(rec, rec^T) (in a second buffer) and the row maxima
__global__ __launch_bounds__(256) void recurrence_sym_kernel(const float* __restrict__ rec, int n, float* __restrict__ sym,
                                                             float* __restrict__ rowmax) {
  __shared__ float sh[256];
  const int i = blockIdx.x;
  float m = -INFINITY;
  for (int j = threadIdx.x; j < n; j += 256) {
    const float v = fminf(rec[(long)i * n + j], rec[(long)j * n + i]);
    sym[(long)i * n + j] = v;
    m = fmaxf(m, v);
  }
  sh[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) rowmax[i] = sh[0];
}

// lower median of x[0..n) (torch.median) by rank counting -> out[0].  `positive`: a zero median (more than half of the
// rows kept no neighbour: constant features) is replaced by the largest value, or 1 - the reference divides by it and
// carries NaNs into its eigensolver
__global__ __launch_bounds__(256) void lower_median_kernel(const float* __restrict__ x, int n, int positive,
                                                           float* __restrict__ out) {
  __shared__ float sh[256];
  const int want = (n - 1) / 2;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float v = x[i];
    mx = fmaxf(mx, v);
    int rank = 0;
    for (int m = 0; m < n; m++) {
      const float u = x[m];
      rank += (u < v || (u == v && m < i)) ? 1 : 0;
    }
    if (rank == want) *out = v;
  }
  sh[threadIdx.x] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0 && positive && !(*out > 0.f)) *out = sh[0] > 0.f ? sh[0] : 1.f;
}

// segment.py:51-57: negatives to zero, exp(rec / -bandwidth), entries that were zero (exp = 1) back to zero
__global__ __launch_bounds__(256) void recurrence_affinity_kernel(float* __restrict__ rec, long n2, const float* __restrict__ bw) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n2) return;
  float v = rec[i];
  if (v < 0.f) v = 0.f;
  v = expf(v / (-1.f * *bw));
  rec[i] = v >= 1.f ? 0.f : v;
}

__device__ __forceinline__ float median_small(float* v, int k) {   // k <= 15: insertion sort, middle element
  for (int a = 1; a < k; a++) {
    const float t = v[a];
    int b = a - 1;
    while (b >= 0 && v[b] > t) { v[b + 1] = v[b]; b--; }
    v[b + 1] = t;
  }
  return v[(k - 1) / 2];
}
__device__ __forceinline__ int reflect_index(int i, int n) {   // F.pad(mode="reflect")
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

// segment.py:75-83 timelag_median_filter: zero-pad the rows to 2n, roll column c up by c, median of 7 along the columns
// (reflect), roll back, crop: out[r][c] = median_dc P[(r - c + c') mod 2n][c'], c' = reflect(c + dc)
__global__ __launch_bounds__(256) void timelag_median_kernel(const float* __restrict__ rec, int n, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (c >= n) return;
  float v[7];
#pragma unroll
  for (int dc = -3; dc <= 3; dc++) {
    const int cc = reflect_index(c + dc, n);
    int row = (r - c + cc) % (2 * n);
    if (row < 0) row += 2 * n;
    v[dc + 3] = row < n ? rec[(long)row * n + cc] : 0.f;
  }
  out[(long)r * n + c] = median_small(v, 7);
}

// segment.py:60-64 median_filter1d along the ROWS of x [n][m] (the reference filters evecs.T along its columns), window k,
// reflect padding k / 2
__global__ __launch_bounds__(256) void median_rows_kernel(const float* __restrict__ x, int n, int m, int k,
                                                          float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (c >= m) return;
  float v[15];
  for (int t = 0; t < k; t++) v[t] = x[(long)reflect_index(r + t - k / 2, n) * m + c];
  out[(long)r * m + c] = median_small(v, k);
}

// segment.py:107-131 differentiable_k_means on unit-norm rows, one workgroup: `iters` updates mu = (r^T data) / sum r with
// r = softmax(temp * data mu^T), then the final responsibilities.  Fixed-order reductions (per-wave shuffles, then the
// waves in order).
constexpr int KM_MAXK = 16;
__global__ __launch_bounds__(256) void soft_kmeans_kernel(const float* __restrict__ data, int n, int k, const float* __restrict__ mu0,
                                                          int iters, float temp, float* __restrict__ r_out,
                                                          float* __restrict__ mu_out) {
  __shared__ float mu[KM_MAXK * KM_MAXK];
  __shared__ float part[4][KM_MAXK * (KM_MAXK + 1)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < k * k; i += 256) mu[i] = mu0[i];
  __syncthreads();
  for (int it = 0; it <= iters; it++) {
    float acc[KM_MAXK * (KM_MAXK + 1)];   // [j][c] weighted sums, then [k*k + j] total responsibility
    for (int i = 0; i < k * (k + 1); i++) acc[i] = 0.f;
    for (int row = tid; row < n; row += 256) {
      float x[KM_MAXK], dist[KM_MAXK];
      for (int c = 0; c < k; c++) x[c] = data[(long)row * k + c];
      float mx = -INFINITY;
      for (int j = 0; j < k; j++) {
        float dsum = 0.f;
        for (int c = 0; c < k; c++) dsum += x[c] * mu[j * k + c];
        dist[j] = temp * dsum;
        mx = fmaxf(mx, dist[j]);
      }
      float den = 0.f;
      for (int j = 0; j < k; j++) { dist[j] = expf(dist[j] - mx); den += dist[j]; }
      for (int j = 0; j < k; j++) {
        const float r = dist[j] / den;
        if (it == iters) r_out[(long)row * k + j] = r;
        acc[k * k + j] += r;
        for (int c = 0; c < k; c++) acc[j * k + c] += r * x[c];
      }
    }
    if (it == iters) break;
    for (int i = 0; i < k * (k + 1); i++) {
      float v = acc[i];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) part[wave][i] = v;
    }
    __syncthreads();
    for (int i = tid; i < k * k; i += 256) {
      const int j = i / k;
      const float num = part[0][i] + part[1][i] + part[2][i] + part[3][i];
      const float den = part[0][k * k + j] + part[1][k * k + j] + part[2][k * k + j] + part[3][k * k + j];
      mu[i] = (1.f / den) * num;
    }
    __syncthreads();
  }
  for (int i = tid; i < k * k; i += 256) mu_out[i] = mu[i];
}

}  // namespace maua

using namespace maua;

extern "C" {

int maua_beat_dp(maua_ctx* ctx, const float* onset_norm, int T, int period, double tightness, double* localscore,
                 double* cumscore, int* backlink) {
  MAUA_REQUIRE(ctx, "maua_beat_dp: ctx is NULL");
  if (T == 0) return MAUA_OK;
  MAUA_REQUIRE(onset_norm && localscore && cumscore && backlink, "maua_beat_dp: NULL argument");
  MAUA_REQUIRE(period >= 1, "maua_beat_dp: the beat period must be at least one frame (bpm too high for this frame rate)");
  MAUA_REQUIRE(tightness > 0, "maua_beat_dp: tightness must be strictly positive");
  MAUA_REQUIRE(2 * period + 2 <= BEAT_RING, "maua_beat_dp: beat period too long");
  // numpy: window = arange(-2 period, -round(period / 2) + 1) (round half to even), txwt = -tightness log(-window / period)^2
  const int half = (int)std::nearbyint(period / 2.0);
  const int wlen = 2 * period - half + 1;
  std::vector<double> gw(2 * period + 1), txwt(wlen);
  for (int m = -period; m <= period; m++) {
    const double a = m * 32.0 / period;
    gw[m + period] = std::exp(-0.5 * (a * a));
  }
  for (int j = 0; j < wlen; j++) {
    const double l = std::log(-(double)(-2 * period + j) / period);
    txwt[j] = -tightness * (l * l);
  }
  size_t off = 0;
  auto carve = [&](size_t b) { size_t o = off; off += (b + 255) & ~(size_t)255; return o; };
  const size_t og = carve(gw.size() * 8), ot = carve(txwt.size() * 8), om = carve(8);
  if (int rc = scratch_reserve(ctx, off)) return rc;
  char* base = (char*)ctx->scratch;
  hipStream_t s = ctx->stream;
  MAUA_HIP_CHECK(hipMemcpyAsync(base + og, gw.data(), gw.size() * 8, hipMemcpyHostToDevice, s));
  MAUA_HIP_CHECK(hipMemcpyAsync(base + ot, txwt.data(), txwt.size() * 8, hipMemcpyHostToDevice, s));
  MAUA_HIP_CHECK(hipStreamSynchronize(s));   // the host tables go out of scope
  hipLaunchKernelGGL(beat_localscore_kernel, dim3(cdiv(T, 256)), dim3(256), 0, s, onset_norm, (const double*)(base + og), T,
                     period, localscore);
  hipLaunchKernelGGL(max_f64_kernel, dim3(1), dim3(256), 0, s, localscore, T, (double*)(base + om));
  hipLaunchKernelGGL(beat_dp_kernel, dim3(1), dim3(64), 0, s, localscore, (const double*)(base + ot),
                     (const double*)(base + om), T, period, wlen, cumscore, backlink);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_segment_reduce(maua_ctx* ctx, const float* x, int T, int C, const int* bounds, int n_segments, int mode,
                        float* out) {
  MAUA_REQUIRE(ctx, "maua_segment_reduce: ctx is NULL");
  if (n_segments == 0 || C == 0) return MAUA_OK;
  MAUA_REQUIRE(x && bounds && out, "maua_segment_reduce: NULL argument");
  MAUA_REQUIRE(mode == 0 || mode == 1, "maua_segment_reduce: mode must be 0 (median) or 1 (mean)");
  MAUA_REQUIRE(T >= 0, "maua_segment_reduce: negative length");
  hipLaunchKernelGGL(segment_reduce_kernel, dim3(n_segments, cdiv(C, 64)), dim3(64), 0, ctx->stream, x, C, bounds, mode, out);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_recurrence_affinity(maua_ctx* ctx, const float* data, int n, int d, int k, int width, float* rec) {
  MAUA_REQUIRE(ctx, "maua_recurrence_affinity: ctx is NULL");
  if (n == 0) return MAUA_OK;
  MAUA_REQUIRE(data && rec, "maua_recurrence_affinity: NULL argument");
  MAUA_REQUIRE(k >= 1 && k <= n, "maua_recurrence_affinity: k must be in [1, n]");
  MAUA_REQUIRE(n <= 16384, "maua_recurrence_affinity: too many beats");
  size_t off = 0;
  auto carve = [&](size_t b) { size_t o = off; off += (b + 255) & ~(size_t)255; return o; };
  const size_t oraw = carve((size_t)n * n * 4), omax = carve((size_t)n * 4), obw = carve(4);
  if (int rc = scratch_reserve(ctx, off)) return rc;
  char* base = (char*)ctx->scratch;
  hipStream_t s = ctx->stream;
  hipLaunchKernelGGL(recurrence_topk_kernel, dim3(n), dim3(256), (size_t)n * 4, s, data, n, d, k, width, (float*)(base + oraw));
  hipLaunchKernelGGL(recurrence_sym_kernel, dim3(n), dim3(256), 0, s, (const float*)(base + oraw), n, rec,
                     (float*)(base + omax));
  hipLaunchKernelGGL(lower_median_kernel, dim3(1), dim3(256), 0, s, (const float*)(base + omax), n, 1, (float*)(base + obw));
  const long n2 = (long)n * n;
  hipLaunchKernelGGL(recurrence_affinity_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, rec, n2,
                     (const float*)(base + obw));
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_timelag_median(maua_ctx* ctx, const float* rec, int n, float* out) {
  MAUA_REQUIRE(ctx, "maua_timelag_median: ctx is NULL");
  if (n == 0) return MAUA_OK;
  MAUA_REQUIRE(rec && out && rec != out, "maua_timelag_median: NULL or aliased argument");
  MAUA_REQUIRE(n > 3, "maua_timelag_median: reflect padding of 3 needs more than 3 beats");
  hipLaunchKernelGGL(timelag_median_kernel, dim3(cdiv(n, 256), n), dim3(256), 0, ctx->stream, rec, n, out);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_median_filter_rows(maua_ctx* ctx, const float* x, int n, int m, int k, float* out) {
  MAUA_REQUIRE(ctx, "maua_median_filter_rows: ctx is NULL");
  if (n == 0 || m == 0) return MAUA_OK;
  MAUA_REQUIRE(x && out && x != out, "maua_median_filter_rows: NULL or aliased argument");
  MAUA_REQUIRE(k >= 1 && k <= 15 && (k & 1), "maua_median_filter_rows: the window must be odd and at most 15");
  MAUA_REQUIRE(n > k / 2, "maua_median_filter_rows: reflect padding needs more rows than half the window");
  hipLaunchKernelGGL(median_rows_kernel, dim3(cdiv(m, 256), n), dim3(256), 0, ctx->stream, x, n, m, k, out);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_soft_kmeans(maua_ctx* ctx, const float* data, int n, int k, const float* mu0, int iters, float temp, float* r_out,
                     float* mu_out) {
  MAUA_REQUIRE(ctx, "maua_soft_kmeans: ctx is NULL");
  MAUA_REQUIRE(data && mu0 && r_out && mu_out, "maua_soft_kmeans: NULL argument");
  MAUA_REQUIRE(k >= 1 && k <= KM_MAXK, "maua_soft_kmeans: at most 16 clusters");
  MAUA_REQUIRE(n >= 1 && iters >= 0, "maua_soft_kmeans: empty input");
  hipLaunchKernelGGL(soft_kmeans_kernel, dim3(1), dim3(256), 0, ctx->stream, data, n, k, mu0, iters, temp, r_out, mu_out);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // extern "C"
