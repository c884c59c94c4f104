// Audio pre-pass kernels (once per clip): STFT, iSTFT, HPSS (median-31 + soft masks), mel power -> dB ->
// onset envelope, RMS, temporal Gaussian filter, min/max, exact order statistics (radix select).
//
// Replaces (reference, maua/audiovisual/audioreactive/selfsupervised/features/...):
//   rosa/spectral.py:10-32 stft/istft, :59-70 spectrogram/melspectrogram, :113-161 magphase/softmask/hpss,
//   rosa/convert.py:7-12 power_to_db, rosa/beat.py:10-23 onset_strength, processing.py:11-49 gaussian_filter,
//   :53-56 normalize, :75-85 median_filter2d, audio.py:31-37 rms,
//   efficient_quantile/efficient_quantile.cpp:86-206 (midpoint quantile), audioreactive/signal.py:41-81.
// Spectra are stored FRAME-major: D[frame][bin] (float2), the natural layout for one-workgroup-per-frame FFTs;
// the Python mirror exposes them as the reference's [bin, frame] through a transposed view.
// All of this is HBM/LDS-bound integer/float streaming work: no GEMM reshaping, coalesced along bins.
#include <cmath>
#include <vector>

#include "common.h"
#include "internal.h"

namespace maua {

constexpr int NFFT = 2048, HOPL = 1024, NBIN = NFFT / 2 + 1;
constexpr int GFFT = 8192;   // the general-framing entry points (any power-of-two n_fft, caller's window) go up to this length: 2 x 64 KB of LDS

__device__ __forceinline__ int reflect_index(int i, int n) {
  // torch 'reflect' padding (no edge repeat); valid for pad < n
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

// Radix-2 Stockham (decimation-in-frequency, autosort: natural order in, natural order out) FFT of 2048 complex
// points held in LDS, ping-ponging between two buffers; 256 threads, 4 butterflies per thread per stage.
// Stage t (stride s = 2^t, sub-length n = N/s): butterfly i = p*s + q reads src[i], src[i + N/2] and writes
// dst[q + 2s*p] = a + b, dst[q + 2s*p + s] = (a - b) * exp(-2 pi i p / n).
// tw[k] = exp(-2 pi i k / 2048), k < 1024 (computed on the host in double).  dir = +1 forward, -1 inverse
// (conjugated twiddles, no 1/N scaling).  Returns the buffer holding the result.
// The same network runs any power-of-two length N <= 2048 (the tempogram of beat.py:33-39 uses 1024): the twiddle
// table is read with stride 2048 / N.
// twn: the length the twiddle table was built for (tw[k] = exp(-2 pi i k / twn), k < twn / 2)
__device__ float2* fft_pow2(float2* a, float2* b, const float2* __restrict__ tw, int dir, int N, int logN, int twn = NFFT) {
  float2* src = a;
  float2* dst = b;
  const int tws = twn / N;
  for (int t = 0; t < logN; t++) {
    const int s = 1 << t;
    for (int i = threadIdx.x; i < N / 2; i += blockDim.x) {
      const int p = i >> t, q = i & (s - 1);
      const float2 u = src[i];
      const float2 v = src[i + N / 2];
      float2 w = tw[p * s * tws];
      if (dir < 0) w.y = -w.y;
      const float2 d = make_float2(u.x - v.x, u.y - v.y);
      dst[q + 2 * s * p] = make_float2(u.x + v.x, u.y + v.y);
      dst[q + 2 * s * p + s] = make_float2(d.x * w.x - d.y * w.y, d.x * w.y + d.y * w.x);
    }
    __syncthreads();
    float2* tmp = src;
    src = dst;
    dst = tmp;
  }
  return src;
}
__device__ __forceinline__ float2* fft2048(float2* a, float2* b, const float2* __restrict__ tw, int dir) {
  return fft_pow2(a, b, tw, dir, NFFT, 11);
}

// ---- STFT: one workgroup per frame -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stft_kernel(const float* __restrict__ y, int n, const float* __restrict__ win,
                                                   const float2* __restrict__ tw, float2* __restrict__ out) {
  __shared__ float2 A[NFFT], B[NFFT];
  const int f = blockIdx.x;
  for (int i = threadIdx.x; i < NFFT; i += blockDim.x) {
    int src = reflect_index(f * HOPL - NFFT / 2 + i, n);
    A[i] = make_float2(y[src] * win[i], 0.f);
  }
  __syncthreads();
  float2* r = fft2048(A, B, tw, +1);
  for (int k = threadIdx.x; k < NBIN; k += blockDim.x) out[(long)f * NBIN + k] = r[k];
}

// ---- iSTFT: per frame inverse FFT * window -> frames[f][2048]; then overlap-add / window envelope --------------
__global__ __launch_bounds__(256) void istft_frames_kernel(const float2* __restrict__ spec, const float* __restrict__ win,
                                                           const float2* __restrict__ tw, float* __restrict__ frames) {
  __shared__ float2 A[NFFT], B[NFFT];
  const int f = blockIdx.x;
  for (int k = threadIdx.x; k < NFFT; k += blockDim.x) {
    float2 v;
    if (k < NBIN) {
      v = spec[(long)f * NBIN + k];
      if (k == 0 || k == NFFT / 2) v.y = 0.f;  // c2r ignores the imaginary part of DC / Nyquist
    } else {
      v = spec[(long)f * NBIN + (NFFT - k)];
      v.y = -v.y;
    }
    A[k] = v;
  }
  __syncthreads();
  float2* r = fft2048(A, B, tw, -1);
  for (int i = threadIdx.x; i < NFFT; i += blockDim.x) frames[(long)f * NFFT + i] = (r[i].x * (1.0f / NFFT)) * win[i];
}

__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, const float* __restrict__ win,
                                                        int n_frames, int length, float* __restrict__ y) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= length) return;
  int p = i + NFFT / 2;  // position in the padded signal
  int f1 = p / HOPL, f0 = f1 - 1;
  float acc = 0.f, env = 0.f;
  if (f0 >= 0 && f0 < n_frames) {
    int o = p - f0 * HOPL;
    acc += frames[(long)f0 * NFFT + o];
    env += win[o] * win[o];
  }
  if (f1 < n_frames) {
    int o = p - f1 * HOPL;
    acc += frames[(long)f1 * NFFT + o];
    env += win[o] * win[o];
  }
  y[i] = (f1 < n_frames || (f0 >= 0 && f0 < n_frames)) ? acc / env : 0.f;
}

// ---- general framing (n_fft a power of two <= 8192, any hop, caller's window): beat.py:33-39 fourier_tempogram, rosa/spectral.py:10-21
// stft at any n_fft (round 6: lengths above 2048 - dynamic LDS, 2 x n_fft complex values, and the twiddle table of that length) ----
__global__ __launch_bounds__(256) void stft_general_kernel(const float* __restrict__ y, int n, const float* __restrict__ win,
                                                           const float2* __restrict__ tw, float2* __restrict__ out,
                                                           int n_fft, int logn, int hop, int twn) {
  extern __shared__ float2 fft_sm[];
  float2* A = fft_sm;
  float2* B = fft_sm + n_fft;
  const int f = blockIdx.x;
  for (int i = threadIdx.x; i < n_fft; i += blockDim.x)
    A[i] = make_float2(y[reflect_index(f * hop - n_fft / 2 + i, n)] * win[i], 0.f);
  __syncthreads();
  float2* r = fft_pow2(A, B, tw, +1, n_fft, logn, twn);
  const int nb = n_fft / 2 + 1;
  for (int k = threadIdx.x; k < nb; k += blockDim.x) out[(long)f * nb + k] = r[k];
}

__global__ __launch_bounds__(256) void istft_general_frames_kernel(const float2* __restrict__ spec,
                                                                   const float* __restrict__ win,
                                                                   const float2* __restrict__ tw, float* __restrict__ frames,
                                                                   int n_fft, int logn, int twn) {
  extern __shared__ float2 fft_sm[];
  float2* A = fft_sm;
  float2* B = fft_sm + n_fft;
  const int f = blockIdx.x, nb = n_fft / 2 + 1;
  for (int k = threadIdx.x; k < n_fft; k += blockDim.x) {
    float2 v;
    if (k < nb) {
      v = spec[(long)f * nb + k];
      if (k == 0 || k == n_fft / 2) v.y = 0.f;
    } else {
      v = spec[(long)f * nb + (n_fft - k)];
      v.y = -v.y;
    }
    A[k] = v;
  }
  __syncthreads();
  float2* r = fft_pow2(A, B, tw, -1, n_fft, logn, twn);
  for (int i = threadIdx.x; i < n_fft; i += blockDim.x) frames[(long)f * n_fft + i] = (r[i].x * (1.0f / n_fft)) * win[i];
}

__global__ __launch_bounds__(256) void istft_general_ola_kernel(const float* __restrict__ frames, const float* __restrict__ win,
                                                                int n_frames, int n_fft, int hop, int length,
                                                                float* __restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= length) return;
  const int p = i + n_fft / 2;  // position in the padded signal; frame f covers [f*hop, f*hop + n_fft)
  int f_lo = (p - n_fft + hop) / hop;  // ceil((p - n_fft + 1) / hop) for p - n_fft + 1 > 0
  if (p - n_fft + 1 <= 0) f_lo = 0;
  const int f_hi = min(p / hop, n_frames - 1);
  float acc = 0.f, env = 0.f;
  for (int f = f_lo; f <= f_hi; f++) {  // ascending frame order: deterministic
    const int o = p - f * hop;
    acc += frames[(long)f * n_fft + o];
    env += win[o] * win[o];
  }
  y[i] = env > 0.f ? acc / env : 0.f;
}

// ---- predominant local pulse, steps 3-4 of beat.py:42-75 on one tempogram frame per workgroup -----------------------
// keep the tempo band, keep only the bins at the frame's peak of log1p(1e6 |z|), normalise by the largest magnitude
__global__ __launch_bounds__(256) void plp_select_kernel(float2* __restrict__ ft, const float* __restrict__ tempo_freq,
                                                         int nb, float tempo_min, float tempo_max) {
  __shared__ float red[256];
  float2* row = ft + (long)blockIdx.x * nb;
  float peak = 0.f;
  for (int k = threadIdx.x; k < nb; k += 256) {
    const float fq = tempo_freq[k];
    float2 z = row[k];
    if (fq < tempo_min || fq > tempo_max) z = make_float2(0.f, 0.f);
    row[k] = z;
    peak = fmaxf(peak, log1pf(1e6f * hypotf(z.x, z.y)));
  }
  red[threadIdx.x] = peak;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  peak = red[0];
  __syncthreads();
  float amax = 0.f;
  for (int k = threadIdx.x; k < nb; k += 256) {
    float2 z = row[k];
    if (log1pf(1e6f * hypotf(z.x, z.y)) < peak) z = make_float2(0.f, 0.f);
    row[k] = z;
    amax = fmaxf(amax, hypotf(z.x, z.y));
  }
  red[threadIdx.x] = amax;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  const float denom = 1.0842021724855044e-19f + red[0];  // finfo(float32).tiny ** 0.5 + max |z|
  for (int k = threadIdx.x; k < nb; k += 256) {
    const float2 z = row[k];
    row[k] = make_float2(z.x / denom, z.y / denom);
  }
}

// ---- magnitude ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void magnitude_kernel(const float2* __restrict__ d, float* __restrict__ mag, long n,
                                                        float power) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float2 v = d[i];
  float m = hypotf(v.x, v.y);  // torch.abs(complex)
  mag[i] = power == 1.f ? m : (power == 2.f ? m * m : powf(m, power));
}

// ---- median of 31 along time (axis 0 of [frames][bins]) or along bins; reflect padding --------------------------
template <int K>
__device__ __forceinline__ float median_of(const float* v) {
  // rank selection: the element with exactly (K/2) elements before it in the stable order
  float med = v[0];
#pragma unroll
  for (int i = 0; i < K; i++) {
    int rank = 0;
#pragma unroll
    for (int j = 0; j < K; j++) rank += (v[j] < v[i]) || (v[j] == v[i] && j < i);
    if (rank == K / 2) med = v[i];
  }
  return med;
}

template <int K>
__global__ __launch_bounds__(256) void median_time_kernel(const float* __restrict__ mag, int n_frames, int n_bins,
                                                          float* __restrict__ out) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  int f = blockIdx.y;
  if (k >= n_bins) return;
  float v[K];
#pragma unroll
  for (int j = 0; j < K; j++) v[j] = mag[(long)reflect_index(f + j - K / 2, n_frames) * n_bins + k];
  out[(long)f * n_bins + k] = median_of<K>(v);
}

template <int K>
__global__ __launch_bounds__(256) void median_freq_kernel(const float* __restrict__ mag, int n_frames, int n_bins,
                                                          float* __restrict__ out) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  int f = blockIdx.y;
  if (k >= n_bins) return;
  const float* row = mag + (long)f * n_bins;
  float v[K];
#pragma unroll
  for (int j = 0; j < K; j++) v[j] = row[reflect_index(k + j - K / 2, n_bins)];
  out[(long)f * n_bins + k] = median_of<K>(v);
}

// ---- soft masks + re-application (spectral.py:120-142, :158-161) ------------------------------------------------
__device__ __forceinline__ float softmask1(float X, float Xref, float power, int split_zeros) {
  float Z = fmaxf(X, Xref);
  bool bad = Z < 1.17549435e-38f;  // torch.finfo(float32).tiny
  if (bad) return split_zeros ? 0.5f : 0.f;
  float a = X / Z, b = Xref / Z;
  float m = power == 2.f ? a * a : powf(a, power);
  float r = power == 2.f ? b * b : powf(b, power);
  return m / (m + r);
}

__global__ __launch_bounds__(256) void hpss_apply_kernel(const float2* __restrict__ d, const float* __restrict__ mag,
                                                         const float* __restrict__ harm, const float* __restrict__ perc,
                                                         float margin, float power, float2* __restrict__ out_h,
                                                         float2* __restrict__ out_p, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int split = margin == 1.f;
  float h = harm[i], p = perc[i], S = mag[i];
  float mh = softmask1(h, p * margin, power, split);
  float mp = softmask1(p, h * margin, power, split);
  float2 v = d[i];
  // (S * mask) * phase, phase = exp(i angle(D)) = D / |D| (1 when D == 0)
  float2 ph = S > 0.f ? make_float2(v.x / S, v.y / S) : make_float2(1.f, 0.f);
  if (out_h) out_h[i] = make_float2((S * mh) * ph.x, (S * mh) * ph.y);
  if (out_p) out_p[i] = make_float2((S * mp) * ph.x, (S * mp) * ph.y);
}

// ---- mel power spectrogram: mel[m][f] = sum_k basis[m][k] * |D[f][k]|^2  (f < T: last frame dropped) ----------
// workgroup = 16 frames x 128 mels; K staged through LDS in chunks of 32 bins.
__global__ __launch_bounds__(256) void mel_power_kernel(const float2* __restrict__ d, const float* __restrict__ basis,
                                                        int T, int n_mels, float* __restrict__ mel) {
  __shared__ float P[16][33];
  __shared__ float Bs[128][33];
  const int f0 = blockIdx.x * 16;
  const int m = threadIdx.x & 127, fg = threadIdx.x >> 7;  // 2 groups of 8 frames
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int k0 = 0; k0 < NBIN; k0 += 32) {
    for (int i = threadIdx.x; i < 16 * 32; i += 256) {
      int ff = i >> 5, kk = i & 31;
      float v = 0.f;
      if (f0 + ff < T && k0 + kk < NBIN) {
        float2 c = d[(long)(f0 + ff) * NBIN + k0 + kk];
        float a = hypotf(c.x, c.y);
        v = a * a;  // |D| ** 2.0 as the reference computes it (abs, then power)
      }
      P[ff][kk] = v;
    }
    for (int i = threadIdx.x; i < 128 * 32; i += 256) {
      int mm = i >> 5, kk = i & 31;
      Bs[mm][kk] = (mm < n_mels && k0 + kk < NBIN) ? basis[(long)mm * NBIN + k0 + kk] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int kk = 0; kk < 32; kk++) {
      float bv = Bs[m][kk];
#pragma unroll
      for (int j = 0; j < 8; j++) acc[j] += bv * P[fg * 8 + j][kk];
    }
    __syncthreads();
  }
  if (m < n_mels)
    for (int j = 0; j < 8; j++) {
      int f = f0 + fg * 8 + j;
      if (f < T) mel[(long)m * T + f] = acc[j];
    }
}

// ---- generic reductions (deterministic: fixed partial layout) ---------------------------------------------------
__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* __restrict__ x, long n,
                                                             float* __restrict__ part /*[grid][2]*/) {
  __shared__ float smn[4], smx[4];
  float mn = INFINITY, mx = -INFINITY;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = x[i];
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o));
    mx = fmaxf(mx, __shfl_xor(mx, o));
  }
  if ((threadIdx.x & 63) == 0) {
    smn[threadIdx.x >> 6] = mn;
    smx[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; i++) {
      mn = fminf(mn, smn[i]);
      mx = fmaxf(mx, smx[i]);
    }
    part[blockIdx.x * 2] = fminf(mn, smn[0]);
    part[blockIdx.x * 2 + 1] = fmaxf(mx, smx[0]);
  }
}

__device__ __forceinline__ void reduce_partials(const float* part, int nparts, float& mn, float& mx) {
  mn = INFINITY;
  mx = -INFINITY;
  for (int i = 0; i < nparts; i++) {
    mn = fminf(mn, part[2 * i]);
    mx = fmaxf(mx, part[2 * i + 1]);
  }
}

// power_to_db, first half: db = 10 log10(max(amin, S)) (in place), partial maxima
__global__ __launch_bounds__(256) void power_to_db_kernel(float* __restrict__ s, long n, float amin) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  s[i] = 10.0f * log10f(fmaxf(amin, s[i]));
}

// onset envelope: apply the top_db floor, lag-1 difference, relu, mean over mels, left pad, crop (beat.py:13-21)
__global__ __launch_bounds__(256) void onset_env_kernel(const float* __restrict__ db, const float* __restrict__ part,
                                                        int nparts, int n_mels, int T, float top_db, int pad_width,
                                                        int aggregate, float* __restrict__ env) {
  int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= T) return;
  float mn, mx;
  reduce_partials(part, nparts, mn, mx);
  const float floor_db = mx - top_db;
  int c = f - pad_width;  // index into the diff sequence (length T-1): d[c] = db[c+1] - db[c]
  float acc = 0.f;
  if (c >= 0 && c + 1 < T) {
    auto val = [&](int m) {
      float a = fmaxf(db[(long)m * T + c + 1], floor_db), b = fmaxf(db[(long)m * T + c], floor_db);
      return fmaxf(a - b, 0.f);
    };
    if (aggregate == 0) {  // torch.mean
      for (int m = 0; m < n_mels; m++) acc += val(m);
      acc /= (float)n_mels;
    } else {  // torch.median(...).values: the lower middle element = rank (n - 1) / 2 in the stable order (beat.py:44)
      const int want = (n_mels - 1) / 2;
      for (int m = 0; m < n_mels; m++) {
        const float v = val(m);
        int rank = 0;
        for (int j = 0; j < n_mels; j++) {
          const float u = val(j);
          rank += (u < v) || (u == v && j < m);
        }
        if (rank == want) acc = v;
      }
    }
  }
  env[f] = acc;
}

// normalize (processing.py:53-56): (x - min) / ((max - min) + eps) ; classic signal.py:27-38 uses eps = 0
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ x, const float* __restrict__ part,
                                                        int nparts, float eps, long n, float* __restrict__ y) {
  float mn, mx;
  reduce_partials(part, nparts, mn, mx);
  const float denom = __fadd_rn(__fsub_rn(mx, mn), eps);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = __fdiv_rn(__fsub_rn(x[i], mn), denom);
}

// rms (audio.py:31-37): frame f covers reflect-padded samples [f*hop - n/2, f*hop + n/2)
__global__ __launch_bounds__(256) void rms_kernel(const float* __restrict__ y, int n, int frame_length, int hop,
                                                  float* __restrict__ out) {
  __shared__ float sh[4];
  const int f = blockIdx.x;
  float acc = 0.f;
  for (int i = threadIdx.x; i < frame_length; i += blockDim.x) {
    float v = y[reflect_index(f * hop - frame_length / 2 + i, n)];
    acc += v * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[f] = sqrtf((sh[0] + sh[1] + sh[2] + sh[3]) / (float)frame_length);
}

// ---- temporal Gaussian filter along axis 0 of [T][C] (processing.py:11-49 / signal.py:108-157) ------------------
// source index of padded position j (0 <= j < T + 2*radius): first `inner` = min(radius, T) samples each side use
// `mode`, anything further out replicates the edge of that padded array (the short-sequence fallback).
__device__ __forceinline__ int pad_src(int j, int T, int radius, int mode) {
  int inner = radius < T ? radius : T;   // width padded with `mode`
  int extra = radius - inner;            // replicate part
  int q = j - extra;                     // index into the [inner + T + inner] array
  int L = T + 2 * inner;
  if (q < 0) q = 0;
  if (q >= L) q = L - 1;
  int i = q - inner;                     // index relative to the signal
  if (i >= 0 && i < T) return i;
  if (mode == MAUA_PAD_CIRCULAR) return ((i % T) + T) % T;
  if (mode == MAUA_PAD_REFLECT) return i < 0 ? -i : 2 * (T - 1) - i;
  if (mode == MAUA_PAD_CONSTANT) return -1;  // zero padding (conv1d(padding="same"), chroma_cens smoothing)
  return i < 0 ? 0 : T - 1;
}

__global__ __launch_bounds__(256) void gauss1d_kernel(const float* __restrict__ x, const float* __restrict__ taps,
                                                      int radius, int T, long C, int mode, float* __restrict__ y) {
  long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int t = blockIdx.y;
  if (c >= C) return;
  float acc = 0.f;
  for (int k = 0; k <= 2 * radius; k++) {
    const int src = pad_src(t + k, T, radius, mode);
    if (src >= 0) acc += taps[k] * x[(long)src * C + c];
  }
  y[(long)t * C + c] = acc;
}

// ---- exact order statistics: 4-pass 8-bit radix select over the NaN-free (optionally masked) elements ----------
__device__ __forceinline__ uint32_t float_key(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone float -> uint
}
__device__ __forceinline__ float key_float(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

struct SelectState {
  unsigned long long count_valid;  // number of candidate elements
  unsigned long long k[2];         // remaining ranks (0-based) for the two order statistics
  uint32_t prefix[2];              // key prefix found so far
  uint32_t hist[2][256];
};

__global__ __launch_bounds__(256) void select_count_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask,
                                                           long n, SelectState* st) {
  unsigned long long c = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    c += (!(x[i] != x[i]) && (!mask || mask[i])) ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&st->count_valid, c);
}

__global__ __launch_bounds__(256) void select_hist_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask,
                                                          long n, int pass, SelectState* st) {
  __shared__ uint32_t h[2][256];
  h[0][threadIdx.x] = 0;
  h[1][threadIdx.x] = 0;
  __syncthreads();
  const int shift = 24 - 8 * pass;
  const uint32_t himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
  const uint32_t p0 = st->prefix[0], p1 = st->prefix[1];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = x[i];
    if (v != v || (mask && !mask[i])) continue;
    uint32_t key = float_key(v);
    uint32_t bin = (key >> shift) & 0xffu;
    if ((key & himask) == p0) atomicAdd(&h[0][bin], 1u);
    if ((key & himask) == p1) atomicAdd(&h[1][bin], 1u);
  }
  __syncthreads();
  if (h[0][threadIdx.x]) atomicAdd(&st->hist[0][threadIdx.x], h[0][threadIdx.x]);
  if (h[1][threadIdx.x]) atomicAdd(&st->hist[1][threadIdx.x], h[1][threadIdx.x]);
}

// rank spec -> the two 0-based ranks.  mode 0: midpoint quantile with float32 q (efficient_quantile.cpp:158-160);
// mode 1: kthvalue with 1-based k passed in kval (signal.py:51); mode 2: linear quantile ranks of torch.quantile
// (rank = q*(n-1) in float32, floor / ceil).
__global__ void select_init_kernel(SelectState* st, int mode, float q, long kval) {
  unsigned long long n = st->count_valid;
  unsigned long long lo = 0, hi = 0;
  if (n > 0) {
    if (mode == 0) {
      double pos = (double)q * (double)(n - 1);
      lo = (unsigned long long)pos;
      hi = (unsigned long long)ceil(pos);
    } else if (mode == 1) {
      lo = hi = (unsigned long long)(kval - 1);
    } else {
      float pos = __fmul_rn(q, (float)(n - 1));
      lo = (unsigned long long)floorf(pos);
      hi = (unsigned long long)ceilf(pos);
    }
    if (lo > n - 1) lo = n - 1;
    if (hi > n - 1) hi = n - 1;
  }
  st->k[0] = lo;
  st->k[1] = hi;
  st->prefix[0] = st->prefix[1] = 0;
  for (int i = 0; i < 256; i++) st->hist[0][i] = st->hist[1][i] = 0;
}

__global__ void select_step_kernel(SelectState* st, int pass) {
  const int shift = 24 - 8 * pass;
  for (int r = 0; r < 2; r++) {
    unsigned long long k = st->k[r], acc = 0;
    int b = 0;
    for (; b < 256; b++) {
      unsigned long long c = st->hist[r][b];
      if (acc + c > k) break;
      acc += c;
    }
    if (b > 255) b = 255;
    st->k[r] = k - acc;
    st->prefix[r] |= ((uint32_t)b) << shift;
  }
  for (int i = 0; i < 256; i++) st->hist[0][i] = st->hist[1][i] = 0;
}

// result: out[0] = value per mode, out[1] = x_(lo), out[2] = x_(hi); ranks[0..1] = lo, hi (or -1 if empty)
__global__ void select_finish_kernel(SelectState* st, int mode, float q, unsigned long long lo0, float* out,
                                     long long* ranks, unsigned long long* saved) {
  // saved[0..1] hold the original ranks (written by select_save_kernel)
  if (st->count_valid == 0) {
    out[0] = out[1] = out[2] = NAN;
    if (ranks) ranks[0] = ranks[1] = -1;
    return;
  }
  float a = key_float(st->prefix[0]), b = key_float(st->prefix[1]);
  unsigned long long lo = saved[0], hi = saved[1];
  float v;
  if (mode == 0) {
    double w = hi > lo ? 0.5 : 0.0;
    double r = (w < 0.5) ? (double)a + w * ((double)b - (double)a) : (double)b - ((double)b - (double)a) * (1.0 - w);
    v = (float)r;
  } else if (mode == 1) {
    v = a;
  } else {
    float pos = __fmul_rn(q, (float)(st->count_valid - 1));
    float w = __fsub_rn(pos, floorf(pos));
    // at::lerp for float: w < 0.5 ? a + w*(b-a) : b - (b-a)*(1-w)
    float diff = __fsub_rn(b, a);
    v = w < 0.5f ? __fadd_rn(a, __fmul_rn(w, diff)) : __fsub_rn(b, __fmul_rn(diff, __fsub_rn(1.f, w)));
  }
  out[0] = v;
  out[1] = a;
  out[2] = b;
  if (ranks) {
    ranks[0] = (long long)lo;
    ranks[1] = (long long)hi;
  }
}

__global__ void select_save_kernel(SelectState* st, unsigned long long* saved) {
  saved[0] = st->k[0];
  saved[1] = st->k[1];
}

// peak mask of signal.py:69-76: strictly greater than both neighbours (edge neighbours clamp to self)
__global__ __launch_bounds__(256) void peak_mask_kernel(const float* __restrict__ x, int n, uint8_t* __restrict__ m) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float c = x[i], p = x[i + 1 < n ? i + 1 : n - 1], q = x[i > 0 ? i - 1 : 0];
  m[i] = (c > p) && (c > q);
}

// clamp(lo, hi) with device-resident bounds, then optional division by a device scalar
__global__ __launch_bounds__(256) void clamp_kernel(const float* __restrict__ x, const float* __restrict__ lo,
                                                    const float* __restrict__ hi, float lo_c, float hi_add, long n,
                                                    float* __restrict__ y) {
  const float l = lo ? lo[0] : lo_c;
  const float h = __fadd_rn(hi[0], hi_add);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = x[i];
    // torch.clamp(min, max) = min(max(x, lo), hi), NaN-propagating
    y[i] = (v != v) ? v : fminf(fmaxf(v, l), h);
  }
}

static int grid_for(long n, int cap = 2048) { return (int)std::max<long>(1, std::min<long>((n + 255) / 256, cap)); }
constexpr int MM_PARTS = 256;

}  // namespace maua

using namespace maua;

// The twiddle / window tables are tiny constants; they are computed on the host in double and cached per ctx.
namespace {
struct AudioTables {
  float2* tw = nullptr;
  float* win = nullptr;
  float2* tw_big = nullptr;   // exp(-2 pi i k / GFFT), k < GFFT / 2: the general entry points above 2048 points
};
static thread_local std::vector<std::pair<maua_ctx*, AudioTables>> g_tables;

int get_tables(maua_ctx* ctx, AudioTables& t) {
  for (auto& p : g_tables)
    if (p.first == ctx) {
      t = p.second;
      return MAUA_OK;
    }
  std::vector<float2> tw(NFFT / 2);
  std::vector<float> win(NFFT);
  for (int k = 0; k < NFFT / 2; k++) {
    double a = -2.0 * M_PI * k / NFFT;
    tw[k] = make_float2((float)cos(a), (float)sin(a));
  }
  for (int i = 0; i < NFFT; i++) win[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / NFFT));  // periodic Hann
  MAUA_HIP_CHECK(hipMalloc((void**)&t.tw, sizeof(float2) * tw.size()));
  MAUA_HIP_CHECK(hipMalloc((void**)&t.win, sizeof(float) * win.size()));
  MAUA_HIP_CHECK(hipMemcpy(t.tw, tw.data(), sizeof(float2) * tw.size(), hipMemcpyHostToDevice));
  MAUA_HIP_CHECK(hipMemcpy(t.win, win.data(), sizeof(float) * win.size(), hipMemcpyHostToDevice));
  std::vector<float2> twb(GFFT / 2);
  for (int k = 0; k < GFFT / 2; k++) {
    double a = -2.0 * M_PI * k / GFFT;
    twb[k] = make_float2((float)cos(a), (float)sin(a));
  }
  MAUA_HIP_CHECK(hipMalloc((void**)&t.tw_big, sizeof(float2) * twb.size()));
  MAUA_HIP_CHECK(hipMemcpy(t.tw_big, twb.data(), sizeof(float2) * twb.size(), hipMemcpyHostToDevice));
  g_tables.push_back({ctx, t});
  return MAUA_OK;
}
}  // namespace

extern "C" {

int maua_stft_num_frames(int n_samples) { return 1 + n_samples / HOPL; }

int maua_stft(maua_ctx* ctx, const float* y, int n_samples, float* out_frames_bins_complex) {
  MAUA_REQUIRE(ctx && y && out_frames_bins_complex, "maua_stft: NULL argument");
  MAUA_REQUIRE(n_samples > NFFT / 2, "maua_stft: signal shorter than the reflect padding (n_fft/2 = 1024)");
  AudioTables t;
  if (int rc = get_tables(ctx, t)) return rc;
  int frames = 1 + n_samples / HOPL;
  hipLaunchKernelGGL(stft_kernel, dim3(frames), dim3(256), 0, ctx->stream, y, n_samples, t.win, t.tw,
                     (float2*)out_frames_bins_complex);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

static int log2_exact(int n) {
  int l = 0;
  while ((1 << l) < n) l++;
  return (1 << l) == n ? l : -1;
}

int maua_stft_general(maua_ctx* ctx, const float* y, int n_samples, int n_fft, int hop, const float* window_dev,
                      float* out_frames_bins_complex) {
  MAUA_REQUIRE(ctx && y && window_dev && out_frames_bins_complex, "maua_stft_general: NULL argument");
  const int logn = log2_exact(n_fft);
  MAUA_REQUIRE(logn >= 1 && n_fft <= GFFT, "maua_stft_general: n_fft must be a power of two in [2, 8192]");
  MAUA_REQUIRE(hop >= 1, "maua_stft_general: hop must be positive");
  MAUA_REQUIRE(n_samples > n_fft / 2, "maua_stft_general: signal shorter than the reflect padding (n_fft/2)");
  AudioTables t;
  if (int rc = get_tables(ctx, t)) return rc;
  const int frames = 1 + n_samples / hop;
  // (lengths up to 2048 keep the 2048-point table: the same twiddles, bit for bit, as before round 6)
  const bool big = n_fft > NFFT;
  const size_t smem = (size_t)2 * n_fft * sizeof(float2);
  if (smem > 64 * 1024)
    MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)stft_general_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(stft_general_kernel, dim3(frames), dim3(256), smem, ctx->stream, y, n_samples, window_dev, big ? t.tw_big : t.tw,
                     (float2*)out_frames_bins_complex, n_fft, logn, hop, big ? GFFT : NFFT);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_istft_general(maua_ctx* ctx, const float* spec_frames_bins_complex, int n_frames, int n_fft, int hop,
                       const float* window_dev, int length, float* y) {
  MAUA_REQUIRE(ctx && spec_frames_bins_complex && window_dev && y, "maua_istft_general: NULL argument");
  const int logn = log2_exact(n_fft);
  MAUA_REQUIRE(logn >= 1 && n_fft <= GFFT, "maua_istft_general: n_fft must be a power of two in [2, 8192]");
  MAUA_REQUIRE(hop >= 1 && n_frames > 0 && length > 0, "maua_istft_general: empty input");
  AudioTables t;
  if (int rc = get_tables(ctx, t)) return rc;
  if (int rc = scratch_reserve(ctx, (size_t)n_frames * n_fft * sizeof(float))) return rc;
  float* frames = (float*)ctx->scratch;
  const bool big = n_fft > NFFT;
  const size_t smem = (size_t)2 * n_fft * sizeof(float2);
  if (smem > 64 * 1024)
    MAUA_HIP_CHECK(hipFuncSetAttribute((const void*)istft_general_frames_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(istft_general_frames_kernel, dim3(n_frames), dim3(256), smem, ctx->stream,
                     (const float2*)spec_frames_bins_complex, window_dev, big ? t.tw_big : t.tw, frames, n_fft, logn, big ? GFFT : NFFT);
  hipLaunchKernelGGL(istft_general_ola_kernel, dim3(cdiv(length, 256)), dim3(256), 0, ctx->stream, frames, window_dev,
                     n_frames, n_fft, hop, length, y);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_plp_select(maua_ctx* ctx, float* tempogram_frames_bins_complex, int n_frames, int n_bins,
                    const float* tempo_freq_dev, float tempo_min, float tempo_max) {
  MAUA_REQUIRE(ctx, "maua_plp_select: ctx is NULL");
  if (n_frames == 0) return MAUA_OK;
  MAUA_REQUIRE(tempogram_frames_bins_complex && tempo_freq_dev && n_bins > 0, "maua_plp_select: NULL argument");
  hipLaunchKernelGGL(plp_select_kernel, dim3(n_frames), dim3(256), 0, ctx->stream, (float2*)tempogram_frames_bins_complex,
                     tempo_freq_dev, n_bins, tempo_min, tempo_max);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_istft(maua_ctx* ctx, const float* spec_frames_bins_complex, int n_frames, int length, float* y) {
  MAUA_REQUIRE(ctx && spec_frames_bins_complex && y, "maua_istft: NULL argument");
  MAUA_REQUIRE(n_frames > 0 && length > 0, "maua_istft: empty input");
  AudioTables t;
  if (int rc = get_tables(ctx, t)) return rc;
  if (int rc = scratch_reserve(ctx, (size_t)n_frames * NFFT * sizeof(float))) return rc;
  float* frames = (float*)ctx->scratch;
  hipLaunchKernelGGL(istft_frames_kernel, dim3(n_frames), dim3(256), 0, ctx->stream,
                     (const float2*)spec_frames_bins_complex, t.win, t.tw, frames);
  MAUA_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(istft_ola_kernel, dim3(cdiv(length, 256)), dim3(256), 0, ctx->stream, frames, t.win, n_frames,
                     length, y);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_magnitude(maua_ctx* ctx, const float* spec_complex, long n, float power, float* mag) {
  MAUA_REQUIRE(ctx && spec_complex && mag, "maua_magnitude: NULL argument");
  if (n == 0) return MAUA_OK;
  hipLaunchKernelGGL(magnitude_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const float2*)spec_complex, mag, n, power);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_median31(maua_ctx* ctx, const float* mag, int n_frames, int n_bins, int axis, float* out) {
  MAUA_REQUIRE(ctx && mag && out, "maua_median31: NULL argument");
  MAUA_REQUIRE(axis == 0 || axis == 1, "maua_median31: axis must be 0 (time) or 1 (frequency)");
  MAUA_REQUIRE((axis == 0 ? n_frames : n_bins) > 15, "maua_median31: axis shorter than the reflect padding (15)");
  dim3 grid(cdiv(n_bins, 256), n_frames);
  if (axis == 0)
    hipLaunchKernelGGL(median_time_kernel<31>, grid, dim3(256), 0, ctx->stream, mag, n_frames, n_bins, out);
  else
    hipLaunchKernelGGL(median_freq_kernel<31>, grid, dim3(256), 0, ctx->stream, mag, n_frames, n_bins, out);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_hpss(maua_ctx* ctx, const float* spec_complex, int n_frames, float margin, float power, float* harm_out,
              float* perc_out) {
  MAUA_REQUIRE(ctx && spec_complex && (harm_out || perc_out), "maua_hpss: NULL argument");
  const long n = (long)n_frames * NBIN;
  if (int rc = scratch_reserve(ctx, (size_t)3 * n * sizeof(float))) return rc;
  float* mag = (float*)ctx->scratch;
  float* harm = mag + n;
  float* perc = harm + n;
  if (int rc = maua_magnitude(ctx, spec_complex, n, 1.f, mag)) return rc;
  if (int rc = maua_median31(ctx, mag, n_frames, NBIN, 0, harm)) return rc;
  if (int rc = maua_median31(ctx, mag, n_frames, NBIN, 1, perc)) return rc;
  hipLaunchKernelGGL(hpss_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const float2*)spec_complex, mag, harm, perc, margin, power, (float2*)harm_out, (float2*)perc_out,
                     n);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_mel_power(maua_ctx* ctx, const float* spec_complex, int n_frames_used, const float* basis, int n_mels,
                   float* mel) {
  MAUA_REQUIRE(ctx && spec_complex && basis && mel, "maua_mel_power: NULL argument");
  MAUA_REQUIRE(n_mels > 0 && n_mels <= 128, "maua_mel_power: n_mels must be 1..128");
  if (n_frames_used == 0) return MAUA_OK;
  hipLaunchKernelGGL(mel_power_kernel, dim3(cdiv(n_frames_used, 16)), dim3(256), 0, ctx->stream,
                     (const float2*)spec_complex, basis, n_frames_used, n_mels, mel);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_onset_from_mel(maua_ctx* ctx, float* mel_inout, int n_mels, int T, float amin, float top_db, int pad_width,
                        int aggregate, float* env) {
  MAUA_REQUIRE(ctx && mel_inout && env, "maua_onset_from_mel: NULL argument");
  MAUA_REQUIRE(aggregate == 0 || aggregate == 1, "maua_onset_from_mel: aggregate must be 0 (mean) or 1 (median)");
  if (T == 0) return MAUA_OK;
  const long n = (long)n_mels * T;
  if (int rc = scratch_reserve(ctx, MM_PARTS * 2 * sizeof(float))) return rc;
  float* part = (float*)ctx->scratch;
  hipLaunchKernelGGL(power_to_db_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, mel_inout, n,
                     amin);
  const int parts = grid_for(n, MM_PARTS);
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(parts), dim3(256), 0, ctx->stream, mel_inout, n, part);
  hipLaunchKernelGGL(onset_env_kernel, dim3(cdiv(T, 256)), dim3(256), 0, ctx->stream, mel_inout, part, parts, n_mels, T,
                     top_db, pad_width, aggregate, env);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_normalize(maua_ctx* ctx, const float* x, long n, float eps, float* y) {
  MAUA_REQUIRE(ctx, "maua_normalize: ctx is NULL");
  if (n == 0) return MAUA_OK;
  MAUA_REQUIRE(x && y, "maua_normalize: NULL argument");
  if (int rc = scratch_reserve(ctx, MM_PARTS * 2 * sizeof(float))) return rc;
  float* part = (float*)ctx->scratch;
  const int parts = grid_for(n, MM_PARTS);
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(parts), dim3(256), 0, ctx->stream, x, n, part);
  hipLaunchKernelGGL(normalize_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, x, part, parts, eps, n, y);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_rms(maua_ctx* ctx, const float* y, int n_samples, int frame_length, int hop, int n_frames, float* out) {
  MAUA_REQUIRE(ctx && y && out, "maua_rms: NULL argument");
  MAUA_REQUIRE(frame_length > 0 && hop > 0 && n_samples > frame_length / 2, "maua_rms: bad framing");
  if (n_frames == 0) return MAUA_OK;
  hipLaunchKernelGGL(rms_kernel, dim3(n_frames), dim3(256), 0, ctx->stream, y, n_samples, frame_length, hop, out);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_gaussian_filter1d(maua_ctx* ctx, const float* x, const float* taps, int radius, int T, long C, int mode,
                           float* y) {
  MAUA_REQUIRE(ctx, "maua_gaussian_filter1d: ctx is NULL");
  if (T == 0 || C == 0) return MAUA_OK;
  MAUA_REQUIRE(x && taps && y && x != y, "maua_gaussian_filter1d: NULL or aliased argument");
  MAUA_REQUIRE(mode >= 0 && mode <= 3, "maua_gaussian_filter1d: unknown padding mode");
  // torch's F.pad(mode="reflect") rejects a pad >= the sequence length; the reference pads min(radius, T) with `mode`
  MAUA_REQUIRE(mode != MAUA_PAD_REFLECT || std::min(radius, T) < T,
               "maua_gaussian_filter1d: reflect padding must be smaller than the sequence (lower sigma)");
  dim3 grid((unsigned)((C + 255) / 256), T);
  hipLaunchKernelGGL(gauss1d_kernel, grid, dim3(256), 0, ctx->stream, x, taps, radius, T, C, mode, y);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_order_stat(maua_ctx* ctx, const float* x, const uint8_t* mask, long n, int mode, float q, long k,
                    float* out3, long long* ranks2) {
  MAUA_REQUIRE(ctx && out3, "maua_order_stat: NULL argument");
  MAUA_REQUIRE(mode >= 0 && mode <= 2, "maua_order_stat: mode must be 0 (midpoint quantile), 1 (kth value), 2 (linear quantile)");
  MAUA_REQUIRE(n == 0 || x, "maua_order_stat: x is NULL");
  MAUA_REQUIRE(mode != 1 || k >= 1, "maua_order_stat: k is 1-based");
  const size_t need = sizeof(SelectState) + 64;
  if (int rc = scratch_reserve(ctx, need)) return rc;
  SelectState* st = (SelectState*)ctx->scratch;
  unsigned long long* saved = (unsigned long long*)((char*)ctx->scratch + sizeof(SelectState));
  hipStream_t s = ctx->stream;
  MAUA_HIP_CHECK(hipMemsetAsync(st, 0, sizeof(SelectState), s));
  const int grid = grid_for(n, 1024);
  if (n > 0) hipLaunchKernelGGL(select_count_kernel, dim3(grid), dim3(256), 0, s, x, mask, n, st);
  hipLaunchKernelGGL(select_init_kernel, dim3(1), dim3(1), 0, s, st, mode, q, k);
  hipLaunchKernelGGL(select_save_kernel, dim3(1), dim3(1), 0, s, st, saved);
  for (int pass = 0; pass < 4 && n > 0; pass++) {
    hipLaunchKernelGGL(select_hist_kernel, dim3(grid), dim3(256), 0, s, x, mask, n, pass, st);
    hipLaunchKernelGGL(select_step_kernel, dim3(1), dim3(1), 0, s, st, pass);
  }
  hipLaunchKernelGGL(select_finish_kernel, dim3(1), dim3(1), 0, s, st, mode, q, 0ull, out3, ranks2, saved);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_peak_mask(maua_ctx* ctx, const float* x, int n, uint8_t* mask) {
  MAUA_REQUIRE(ctx, "maua_peak_mask: ctx is NULL");
  if (n == 0) return MAUA_OK;
  MAUA_REQUIRE(x && mask, "maua_peak_mask: NULL argument");
  hipLaunchKernelGGL(peak_mask_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, x, n, mask);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

int maua_clamp(maua_ctx* ctx, const float* x, const float* lo_dev, const float* hi_dev, float lo_const, float hi_add,
               long n, float* y) {
  MAUA_REQUIRE(ctx, "maua_clamp: ctx is NULL");
  if (n == 0) return MAUA_OK;
  MAUA_REQUIRE(x && hi_dev && y, "maua_clamp: NULL argument");
  hipLaunchKernelGGL(clamp_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, x, lo_dev, hi_dev, lo_const, hi_add, n,
                     y);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // extern "C"
