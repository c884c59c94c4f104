// Guided-diffusion UNet forward + DDIM step for BASELINE configs[3] ("guided-diffusion 256x256, 100-step DDIM").
//
// Replaces (reference): the model maua/diffusion/processors/guided.py:164-209 `create_models` builds (OpenAI
// guided-diffusion `UNetModel`: num_channels 256, num_res_blocks 2, attention at 32 / 16 / 8 with 64-channel heads,
// learn_sigma, resblock_updown, use_scale_shift_norm; fp16 there, bf16 operands / f32 accumulate here) and the sampler
// step `diffusion.ddim_sample(..., cond_fn=...)` that guided.py:277-339 `GuidedDiffusion.forward` loops over.  The network
// and sampler sources are the git submodule maua/submodules/guided_diffusion - EMPTY in the reference checkout, no pinned
// revision: the published architecture is restated (oracle/diffusion.py says the same), PARITY UNPINNED.
//
//   ResBlock : h = conv(resample(silu(gn(x))));  h = gn(h) * (1 + scale) + shift  [scale | shift = linear(silu(emb))];
//              out = skip(resample(x)) + conv(silu(h))          resample = avg_pool 2 / nearest x2 / identity
//   Attention: x + proj(attn(qkv(gn(x))))                        QKVAttentionLegacy, softmax in f32
//   UNet     : emb = linear(silu(linear(timestep_embedding(t)))); encoder blocks push their outputs, decoder blocks
//              read cat([h, hs.pop()]); out = conv(silu(gn(h)))
//
// MI355X design: activations NHWC in the network dtype (a 1x1 convolution is a plain row-major GEMM, GroupNorm's 32 groups
// are contiguous channel runs of a pixel); every 3x3 convolution runs on the MFMA implicit-GEMM kernels written for the
// StyleGAN2 path (LDS-direct-load kernel at >= 32-wide grids, the batch-wide split-K gather GEMM on the 16^2 / 8^2 levels,
// generic kernel elsewhere) with bias and the block's residual in the epilogue; GroupNorm + scale-shift + SiLU + the
// block's up / down resampling are ONE pass that writes the convolution's input; the decoder's channel concatenation is
// never materialised (GroupNorm and the 1x1 skip GEMM read both tensors); all ResBlocks' timestep projections are one
// batched GEMV at the top of the forward.  A forward is a fixed sequence of launches on one stream over a
// pre-planned arena (no allocation, no host sync), so maua_ddim_sample_loop can capture the whole sampler in a hipGraph.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"
#include "internal.h"

using namespace maua;

namespace {

// ------------------------------------------------------------------------------------------------ small f32 kernels
// nn.py timestep_embedding: e[b] = [cos(t_b * f_i) | sin(t_b * f_i)], f_i = exp(-ln(10000) * i / half)
// (freqs: the host's float32 table when it was uploaded - the frequencies multiply timesteps up to 1000, so one ulp of
//  difference between two exp implementations would show up as 6e-5 in the embedding)
__global__ void timestep_embedding_kernel(const float* __restrict__ t, const float* __restrict__ freqs, float* __restrict__ e,
                                          int B, int dim) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim / 2;
  if (idx >= B * half) return;
  const int b = idx / half, i = idx - b * half;
  const float freq = freqs ? freqs[i] : expf(-logf(10000.0f) * (float)i / (float)half);
  const float arg = t[b] * freq;
  e[(long)b * dim + i] = cosf(arg);
  e[(long)b * dim + half + i] = sinf(arg);
  if ((dim & 1) && i == 0) e[(long)b * dim + dim - 1] = 0.f;
}

// y[b][n] = act_out(W[n] . act_in(x[b]) + bias[n]); one wave per output row n (W streamed once, coalesced), all B
// samples per row (B <= 16 per pass).  act: 0 none, 1 SiLU.  The timestep MLP and every ResBlock's emb_layers (all in
// f32 like the reference, whose convert_to_fp16 leaves nn.Linear alone).
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ y, int B, int K,
                                                          int N, int act_in, int act_out) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float* wr = W + (long)n * K;
  for (int b0 = 0; b0 < B; b0 += 8) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0.f;
    for (int k = lane; k < K; k += 64) {
      const float w = wr[k];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        if (b0 + j < B) {
          float v = x[(long)(b0 + j) * K + k];
          if (act_in) v = v / (1.f + expf(-v));
          acc[j] = fmaf(w, v, acc[j]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      float v = acc[j];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0 && b0 + j < B) {
        v += bias ? bias[n] : 0.f;
        if (act_out) v = v / (1.f + expf(-v));
        y[(long)(b0 + j) * N + n] = v;
      }
    }
  }
}

// dst[i][o][k] = src[o][i][kk - 1 - k]: a convolution's (kk = 9) / linear layer's (kk = 1) weight for its input gradient
__global__ __launch_bounds__(256) void transpose_flip_kernel(const float* __restrict__ src, float* __restrict__ dst, int Co, int Ci, int kk) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)Co * Ci * kk) return;
  const int k = (int)(idx % kk);
  const long oi = idx / kk;
  const int i = (int)(oi % Ci), o = (int)(oi / Ci);
  dst[((long)i * Co + o) * kk + (kk - 1 - k)] = src[idx];
}

// ------------------------------------------------------------------------------------------------------ GroupNorm
// Statistics in float64 (sum and sum of squares of exactly representable f32 products): no cancellation whatever the
// mean / spread ratio; fixed summation order (bit-reproducible).  Pass 1: per (sample, pixel chunk, row slot) per-channel
// partial sums; pass 2: per (sample, group) mean and 1 / sqrt(var + eps).
template <typename T>
__global__ void gn_partial_kernel(const T* __restrict__ x0, int C0, const T* __restrict__ x1, int C1, long HW, int ppc,
                                  double* __restrict__ part) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const int C = C0 + C1, PPP = C / EPC;
  const int pc = threadIdx.x % PPP, ry = threadIdx.x / PPP, RY = blockDim.x / PPP;
  const int chunk = blockIdx.x, b = blockIdx.y;
  if (ry >= RY) return;
  const int c = pc * EPC;
  const T* src;
  long stride;
  if (c < C0) { src = x0 + (long)b * HW * C0 + c; stride = C0; }
  else { src = x1 + (long)b * HW * C1 + (c - C0); stride = C1; }
  const long p0 = (long)chunk * ppc, p1 = p0 + ppc < HW ? p0 + ppc : HW;
  double s[EPC], ss[EPC];
#pragma unroll
  for (int e = 0; e < EPC; e++) s[e] = ss[e] = 0.0;
  for (long p = p0 + ry; p < p1; p += RY) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(src + p * stride);
#pragma unroll
    for (int e = 0; e < EPC; e++) {
      float f;
      if constexpr (sizeof(T) == 2) f = bf2f((bf16_t)((v[e >> 1] >> ((e & 1) * 16)) & 0xffffu));
      else f = __uint_as_float(v[e]);
      const double d = (double)f;
      s[e] += d;
      ss[e] += d * d;
    }
  }
  double* dst = part + ((((long)b * gridDim.x + chunk) * RY + ry) * C + c) * 2;
#pragma unroll
  for (int e = 0; e < EPC; e++) { dst[2 * e] = s[e]; dst[2 * e + 1] = ss[e]; }
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ part, int rows, int C, long HW, float eps,
                                                          float* __restrict__ stats) {
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
    const double mean = red[0][0] / cnt;
    double var = red[1][0] / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[((long)b * 32 + g) * 2] = (float)mean;
    stats[((long)b * 32 + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// y = [silu]( gn(x) [* (1 + scale) + shift] ), optionally resampled (mode 1: 2x2 average of the activated values - the
// ResBlock's Downsample sits BEHIND norm + SiLU; mode 2: nearest x2), written dense NHWC as the next convolution's input.
// xr (optional, modes 1 / 2): the same resampling of the raw input (the block's x_upd, its residual branch).
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x0, int C0, const T* __restrict__ x1, int C1,
                                                       const float* __restrict__ stats, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ ss, long ss_ld,
                                                       int silu, int mode, T* __restrict__ y, T* __restrict__ xr, int B, int H,
                                                       int W) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const int C = C0 + C1, PPP = C / EPC, cpg = C / 32;
  const int Ho = mode == 1 ? H / 2 : (mode == 2 ? H * 2 : H), Wo = mode == 1 ? W / 2 : (mode == 2 ? W * 2 : W);
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * Ho * Wo * PPP) return;
  const int pc = (int)(idx % PPP);
  long p = idx / PPP;
  const int ox = (int)(p % Wo); p /= Wo;
  const int oy = (int)(p % Ho);
  const int b = (int)(p / Ho);
  const int c = pc * EPC;
  const T* src;
  long stride;
  if (c < C0) { src = x0 + (long)b * H * W * C0 + c; stride = C0; }
  else { src = x1 + (long)b * H * W * C1 + (c - C0); stride = C1; }
  float ca[EPC], cb[EPC], sc[EPC], sh[EPC];
#pragma unroll
  for (int e = 0; e < EPC; e++) {
    const int g = (c + e) / cpg;
    const float mean = stats[((long)b * 32 + g) * 2], rstd = stats[((long)b * 32 + g) * 2 + 1];
    ca[e] = rstd * gamma[c + e];
    cb[e] = beta[c + e] - mean * ca[e];
    sc[e] = ss ? 1.f + ss[(long)b * ss_ld + c + e] : 1.f;
    sh[e] = ss ? ss[(long)b * ss_ld + C + c + e] : 0.f;
  }
  float acc[EPC], raw[EPC];
#pragma unroll
  for (int e = 0; e < EPC; e++) acc[e] = raw[e] = 0.f;
  const int taps = mode == 1 ? 4 : 1;
  for (int t = 0; t < taps; t++) {
    int iy, ix;
    if (mode == 1) { iy = 2 * oy + (t >> 1); ix = 2 * ox + (t & 1); }
    else if (mode == 2) { iy = oy >> 1; ix = ox >> 1; }
    else { iy = oy; ix = ox; }
    const u32x4 v = *reinterpret_cast<const u32x4*>(src + ((long)iy * W + ix) * stride);
#pragma unroll
    for (int e = 0; e < EPC; e++) {
      float f;
      if constexpr (sizeof(T) == 2) f = bf2f((bf16_t)((v[e >> 1] >> ((e & 1) * 16)) & 0xffffu));
      else f = __uint_as_float(v[e]);
      raw[e] += f;
      float u = fmaf(f, ca[e], cb[e]);
      if (ss) u = fmaf(u, sc[e], sh[e]);
      if (silu) u = u / (1.f + expf(-u));
      acc[e] += u;
    }
  }
  const float norm = mode == 1 ? 0.25f : 1.f;
  u32x4 o, ro;
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      o[k] = pack2bf(acc[2 * k] * norm, acc[2 * k + 1] * norm);
      ro[k] = pack2bf(raw[2 * k] * norm, raw[2 * k + 1] * norm);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) { o[k] = __float_as_uint(acc[k] * norm); ro[k] = __float_as_uint(raw[k] * norm); }
  }
  const long opix = ((long)b * Ho + oy) * Wo + ox;
  *reinterpret_cast<u32x4*>(y + opix * C + c) = o;
  if (xr) *reinterpret_cast<u32x4*>(xr + opix * C + c) = ro;
}

// ---- fast path (C / 32 channels per group a multiple of the 16-byte piece: every real layer of the network; the
// per-channel kernels above serve narrow test networks).  Same arithmetic, organised for the memory system:
//   * partial sums per GROUP, reduced inside the workgroup (LDS, fixed order) -> one row of 32 (sum, sumsq) pairs per
//     pixel chunk; a finalize launch of B x 32 threads adds the <= 128 chunk rows;
//   * the apply pass indexes (sample, row) by blockIdx.y and (pixel, piece) by 32-bit arithmetic, loads its piece's
//     coefficients as float4s, and in bf16 mode uses the hardware exp / reciprocal for SiLU (exact in f32 mode).
// (Folding the finalize into this kernel - the last workgroup of a sample, found by an atomic ticket, adds the chunk rows - was
//  built and measured in round 3: 26.7 -> 35.3 ms per UNet step at B = 8.  A device-scope release fence per workgroup writes
//  the XCD's L2 back (8 XCDs, one L2 each): far dearer than the ~5 us launch it saves.  The finalize stays its own launch.)
template <typename T>
__global__ void gn_partial_group_kernel(const T* __restrict__ x0, int C0, const T* __restrict__ x1, int C1, long HW, int ppc,
                                        double* __restrict__ part) {
  constexpr int EPC = 16 / (int)sizeof(T);
  __shared__ double red[2][1024];
  const int C = C0 + C1, PPP = C / EPC;
  const int pc = threadIdx.x % PPP, ry = threadIdx.x / PPP, RY = blockDim.x / PPP;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int c = pc * EPC;
  const T* src;
  long stride;
  if (c < C0) { src = x0 + (long)b * HW * C0 + c; stride = C0; }
  else { src = x1 + (long)b * HW * C1 + (c - C0); stride = C1; }
  const long p0 = (long)chunk * ppc, p1 = p0 + ppc < HW ? p0 + ppc : HW;
  double s = 0.0, ss = 0.0;
  for (long p = p0 + ry; p < p1; p += RY) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(src + p * stride);
    float fs = 0.f, fq = 0.f;   // 8 (4) values: exact enough in f32 before they join the f64 sums
#pragma unroll
    for (int e = 0; e < EPC; e++) {
      float f;
      if constexpr (sizeof(T) == 2) f = bf2f((bf16_t)((v[e >> 1] >> ((e & 1) * 16)) & 0xffffu));
      else f = __uint_as_float(v[e]);
      if constexpr (sizeof(T) == 2) { fs += f; fq = fmaf(f, f, fq); }
      else { s += (double)f; ss += (double)f * (double)f; }
    }
    if constexpr (sizeof(T) == 2) { s += (double)fs; ss += (double)fq; }
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = ss;
  __syncthreads();
  if (threadIdx.x < 32) {
    const int g = threadIdx.x, ppg = PPP / 32;
    double ts = 0.0, tq = 0.0;
    for (int r = 0; r < RY; r++)
      for (int j = 0; j < ppg; j++) {
        ts += red[0][r * PPP + g * ppg + j];
        tq += red[1][r * PPP + g * ppg + j];
      }
    double* dst = part + (((long)b * gridDim.x + chunk) * 32 + g) * 2;
    dst[0] = ts;
    dst[1] = tq;
  }
}

__global__ __launch_bounds__(256) void gn_finalize_group_kernel(const double* __restrict__ part, int nchunk, double cnt,
                                                                float eps, float* __restrict__ stats) {
  // 32 groups x 8 lanes: lane j adds chunks j, j + 8, ... (independent loads in flight), then a fixed-order LDS tree
  __shared__ double red[2][256];
  const int g = threadIdx.x & 31, j = threadIdx.x >> 5, b = blockIdx.x;
  double s = 0.0, ss = 0.0;
  for (int k = j; k < nchunk; k += 8) {
    const double* p = part + (((long)b * nchunk + k) * 32 + g) * 2;
    s += p[0];
    ss += p[1];
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = ss;
  __syncthreads();
  if (j == 0) {
#pragma unroll
    for (int q = 1; q < 8; q++) { s += red[0][q * 32 + g]; ss += red[1][q * 32 + g]; }
    const double mean = s / cnt;
    double var = ss / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[((long)b * 32 + g) * 2] = (float)mean;
    stats[((long)b * 32 + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// Statistics from the piece sums a producing convolution left behind (ConvArgs.psum: per 8 x 32-pixel tile and 8-channel
// piece, sums and sums of squares of the stored values) - the tensor itself is not read again.  Two sources like everywhere
// (the decoder's virtual concatenation); 32 groups x 8 lanes, rows j, j + 8, ... per lane, fixed-order LDS tree.
__global__ __launch_bounds__(256) void gn_finalize_psum_kernel(const float* __restrict__ ps0, int rows0, int C0,
                                                               const float* __restrict__ ps1, int rows1, int C1, double cnt,
                                                               float eps, float* __restrict__ stats) {
  __shared__ double red[2][256];
  const int g = threadIdx.x & 31, j = threadIdx.x >> 5, b = blockIdx.x;
  const int np0 = C0 >> 3, np1 = C1 >> 3, ppg = (np0 + np1) >> 5;
  double s = 0.0, ss = 0.0;
  for (int pi = 0; pi < ppg; pi++) {
    const int piece = g * ppg + pi;
    const bool first = piece < np0;
    const float* src = first ? ps0 : ps1;
    const int rows = first ? rows0 : rows1, np = first ? np0 : np1, pc = first ? piece : piece - np0;
    for (int r = j; r < rows; r += 8) {
      const float4* p = reinterpret_cast<const float4*>(src + (((long)b * rows + r) * np + pc) * 16);
      const float4 a0 = p[0], a1 = p[1], q0 = p[2], q1 = p[3];
      s += (double)a0.x + (double)a0.y + (double)a0.z + (double)a0.w + (double)a1.x + (double)a1.y + (double)a1.z + (double)a1.w;
      ss += (double)q0.x + (double)q0.y + (double)q0.z + (double)q0.w + (double)q1.x + (double)q1.y + (double)q1.z + (double)q1.w;
    }
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = ss;
  __syncthreads();
  if (j == 0) {
#pragma unroll
    for (int q = 1; q < 8; q++) { s += red[0][q * 32 + g]; ss += red[1][q * 32 + g]; }
    const double mean = s / cnt;
    double var = ss / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[((long)b * 32 + g) * 2] = (float)mean;
    stats[((long)b * 32 + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// A thread owns one 16-byte channel piece and walks GN_PX output pixels of its row with it (round 5: one pixel per thread spent
// 136 bytes of parameter loads - gamma, beta, scale, shift, statistics - on 16 bytes of data and ran at 1.7 TB/s).
constexpr int GN_PX = 8;
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_group_kernel(const T* __restrict__ x0, int C0, const T* __restrict__ x1, int C1,
                                                             const float* __restrict__ stats, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ ss,
                                                             long ss_ld, int silu, int mode, T* __restrict__ y,
                                                             T* __restrict__ xr, int H, int W, int Ho, int Wo) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const unsigned C = C0 + C1, PPP = C / EPC, cpg = C / 32;
  const unsigned li = blockIdx.x * 256u + threadIdx.x;
  const unsigned nxg = ((unsigned)Wo + GN_PX - 1) / GN_PX;
  if (li >= nxg * PPP) return;
  const unsigned oxg = li / PPP, pc = li - oxg * PPP;
  const unsigned b = blockIdx.y / (unsigned)Ho, oy = blockIdx.y - b * (unsigned)Ho;
  const unsigned c = pc * EPC;
  const T* src;
  unsigned stride;
  if (c < (unsigned)C0) { src = x0 + (long)b * H * W * C0 + c; stride = C0; }
  else { src = x1 + (long)b * H * W * C1 + (c - C0); stride = C1; }
  const unsigned g = c / cpg;
  const float2 mr = *reinterpret_cast<const float2*>(stats + ((long)b * 32 + g) * 2);
  float ca[EPC], cb[EPC], sc[EPC], sh[EPC];
#pragma unroll
  for (int q4 = 0; q4 < EPC / 4; q4++) {
    const float4 gm = *reinterpret_cast<const float4*>(gamma + c + 4 * q4);
    const float4 bt = *reinterpret_cast<const float4*>(beta + c + 4 * q4);
    const float gv[4] = {gm.x, gm.y, gm.z, gm.w}, bv[4] = {bt.x, bt.y, bt.z, bt.w};
    float sv[4] = {1.f, 1.f, 1.f, 1.f}, hv[4] = {0.f, 0.f, 0.f, 0.f};
    if (ss) {
      const float4 s4 = *reinterpret_cast<const float4*>(ss + (long)b * ss_ld + c + 4 * q4);
      const float4 h4 = *reinterpret_cast<const float4*>(ss + (long)b * ss_ld + C + c + 4 * q4);
      sv[0] = 1.f + s4.x; sv[1] = 1.f + s4.y; sv[2] = 1.f + s4.z; sv[3] = 1.f + s4.w;
      hv[0] = h4.x; hv[1] = h4.y; hv[2] = h4.z; hv[3] = h4.w;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int e = 4 * q4 + k;
      ca[e] = mr.y * gv[k];
      cb[e] = bv[k] - mr.x * ca[e];
      sc[e] = sv[k];
      sh[e] = hv[k];
    }
  }
  const int taps = mode == 1 ? 4 : 1;
  const float norm = mode == 1 ? 0.25f : 1.f;
  // the pixel group's loads are requested together (mode 0 / 2: one per pixel; the 4-tap average pool walks pixel by pixel)
  u32x4 vin[GN_PX];
  if (taps == 1) {
#pragma unroll
    for (int k = 0; k < GN_PX; k++) {
      const unsigned ox = min(oxg * GN_PX + k, (unsigned)Wo - 1);
      const unsigned iy = mode == 2 ? oy >> 1 : oy, ix = mode == 2 ? ox >> 1 : ox;
      vin[k] = *reinterpret_cast<const u32x4*>(src + (long)(iy * (unsigned)W + ix) * stride);
    }
  }
#pragma unroll
  for (int k = 0; k < GN_PX; k++) {
    const unsigned ox = oxg * GN_PX + k;
    if (ox >= (unsigned)Wo) break;
    float acc[EPC], raw[EPC];
#pragma unroll
    for (int e = 0; e < EPC; e++) acc[e] = raw[e] = 0.f;
    for (int t = 0; t < taps; t++) {
      u32x4 v;
      if (taps == 1) v = vin[k];
      else v = *reinterpret_cast<const u32x4*>(src + (long)((2 * oy + (t >> 1)) * (unsigned)W + 2 * ox + (t & 1)) * stride);
#pragma unroll
      for (int e = 0; e < EPC; e++) {
        float f;
        if constexpr (sizeof(T) == 2) f = bf2f((bf16_t)((v[e >> 1] >> ((e & 1) * 16)) & 0xffffu));
        else f = __uint_as_float(v[e]);
        raw[e] += f;
        float u = fmaf(f, ca[e], cb[e]);
        if (ss) u = fmaf(u, sc[e], sh[e]);
        if (silu) {
          if constexpr (sizeof(T) == 2) u = __fdividef(u, 1.f + __expf(-u));
          else u = u / (1.f + expf(-u));
        }
        acc[e] += u;
      }
    }
    u32x4 o, ro;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        o[q] = pack2bf(acc[2 * q] * norm, acc[2 * q + 1] * norm);
        ro[q] = pack2bf(raw[2 * q] * norm, raw[2 * q + 1] * norm);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++) { o[q] = __float_as_uint(acc[q] * norm); ro[q] = __float_as_uint(raw[q] * norm); }
    }
    const long opix = ((long)b * Ho + oy) * Wo + ox;
    *reinterpret_cast<u32x4*>(y + opix * C + c) = o;
    if (xr) *reinterpret_cast<u32x4*>(xr + opix * C + c) = ro;
  }
}

// workspace (bytes) of one GroupNorm over [B][HW][C]: partial sums + the [B][32][2] statistics
struct GnPlan { int fast, RY, ppc; long nchunk; size_t part_bytes; };
static GnPlan gn_plan(int B, int C, long HW, int esize) {
  GnPlan p;
  const int EPC = 16 / esize, PPP = C / EPC, cpg = C / 32;
  p.fast = cpg % EPC == 0 && PPP <= 1024;
  p.RY = std::max(1, 512 / PPP);
  long nchunk = HW / ((long)p.RY * 4);
  nchunk = std::max(1L, std::min(128L, nchunk));
  p.ppc = (int)((HW + nchunk - 1) / nchunk);
  p.nchunk = (HW + p.ppc - 1) / p.ppc;
  p.part_bytes = p.fast ? (size_t)B * p.nchunk * 32 * 16 : (size_t)B * p.nchunk * p.RY * C * 16;
  return p;
}

// GroupNorm32 (+ scale-shift) (+ SiLU) (+ resample) of the virtually concatenated [x0 | x1]; part / stats: workspaces
template <typename T>
static int launch_group_norm(hipStream_t st, const T* x0, int C0, const T* x1, int C1, int B, int H, int W,
                             const float* gamma, const float* beta, const float* ss, long ss_ld, int silu, int mode, T* y, T* xr,
                             double* part, float* stats, const float* ps0 = nullptr, int rows0 = 0, const float* ps1 = nullptr,
                             int rows1 = 0) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const int C = C0 + C1, PPP = C / EPC;
  const long HW = (long)H * W;
  MAUA_REQUIRE(C % 32 == 0 && PPP <= 1024 && C0 % EPC == 0, "group_norm: C % 32 == 0, at most 1024 16-byte pieces per pixel");
  const GnPlan p = gn_plan(B, C, HW, (int)sizeof(T));
  const int Ho = mode == 1 ? H / 2 : (mode == 2 ? H * 2 : H), Wo = mode == 1 ? W / 2 : (mode == 2 ? W * 2 : W);
  if (p.fast && (long)B * Ho <= 65535) {
    if (ps0 && (!x1 || ps1) && sizeof(T) == 2) {
      // every source's producer left its piece sums: no statistics pass over the tensor
      hipLaunchKernelGGL(gn_finalize_psum_kernel, dim3(B), dim3(256), 0, st, ps0, rows0, C0, ps1, rows1, C1,
                         (double)HW * (C / 32), 1e-5f, stats);
    } else {
      hipLaunchKernelGGL(gn_partial_group_kernel<T>, dim3((unsigned)p.nchunk, B), dim3(PPP * p.RY), 0, st, x0, C0, x1, C1, HW,
                         p.ppc, part);
      hipLaunchKernelGGL(gn_finalize_group_kernel, dim3(B), dim3(256), 0, st, part, (int)p.nchunk, (double)HW * (C / 32), 1e-5f,
                         stats);
    }
    hipLaunchKernelGGL(gn_apply_group_kernel<T>, dim3((unsigned)(((long)((Wo + GN_PX - 1) / GN_PX) * PPP + 255) / 256), (unsigned)(B * Ho)),
                       dim3(256), 0, st, x0, C0, x1, C1, stats, gamma, beta, ss, ss_ld, silu, mode, y, xr, H, W, Ho, Wo);
  } else {
    const GnPlan q = p.fast ? GnPlan{0, p.RY, p.ppc, p.nchunk, 0} : p;
    hipLaunchKernelGGL(gn_partial_kernel<T>, dim3((unsigned)q.nchunk, B), dim3(PPP * q.RY), 0, st, x0, C0, x1, C1, HW, q.ppc,
                       part);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(32, B), dim3(256), 0, st, part, (int)(q.nchunk * q.RY), C, HW, 1e-5f, stats);
    const long total = (long)B * Ho * Wo * PPP;
    hipLaunchKernelGGL(gn_apply_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x0, C0, x1, C1, stats, gamma,
                       beta, ss, ss_ld, silu, mode, y, xr, B, H, W);
  }
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}
// the statistics passes alone (per-channel kernels: any C % 32 == 0)
template <typename T>
static int launch_group_norm_stats(hipStream_t st, const T* x, int C, int B, int H, int W, double* part, float* stats) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const int PPP = C / EPC;
  const long HW = (long)H * W;
  MAUA_REQUIRE(C % 32 == 0 && PPP <= 1024, "group_norm: C % 32 == 0, at most 1024 16-byte pieces per pixel");
  const GnPlan p = gn_plan(B, C, HW, (int)sizeof(T));
  hipLaunchKernelGGL(gn_partial_kernel<T>, dim3((unsigned)p.nchunk, B), dim3(PPP * p.RY), 0, st, x, C, (const T*)nullptr, 0, HW, p.ppc, part);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(32, B), dim3(256), 0, st, part, (int)(p.nchunk * p.RY), C, HW, 1e-5f, stats);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}
static size_t gn_part_bytes(int B, int C, long HW, int esize) {
  // (the per-channel layout is the larger one and also serves the fall-back of a fast-path shape with too many rows)
  const GnPlan p = gn_plan(B, C, HW, esize);
  return std::max(p.part_bytes, (size_t)B * p.nchunk * p.RY * C * 16);
}

// ------------------------------------------------------------------------------------------------------ DDIM step
// gaussian_diffusion.py ddim_sample for an epsilon-predicting model (clip_denoised False), in the reference's float32
// operation order; coefficients per sample: cf[b] = {sqrt_recip_ac, sqrt_recipm1_ac, sqrt(1 - ac), sqrt(ac_prev),
// sqrt(1 - ac_prev - sigma^2), sigma * nonzero_mask, 0, 0}.  model_out [B][Cm][HW] (eps = first C channels), x [B][C][HW].
__global__ __launch_bounds__(256) void ddim_step_kernel(const float* __restrict__ x, const float* __restrict__ model_out,
                                                        const float* __restrict__ grad, const float* __restrict__ noise,
                                                        const float* __restrict__ cf, int C, int Cm, long HW, long total,
                                                        float* __restrict__ sample, float* __restrict__ pred_out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long chw = (long)C * HW;
  const int b = (int)(idx / chw);
  const long rem = idx - (long)b * chw;
  const float* k = cf + b * 8;
  const float xv = x[idx];
  const float eps_m = model_out[(long)b * Cm * HW + rem];
  float pred = k[0] * xv - k[1] * eps_m;           // _predict_xstart_from_eps
  if (grad) {                                      // condition_score
    float eps = (k[0] * xv - pred) / k[1];
    eps = eps - k[2] * grad[idx];
    pred = k[0] * xv - k[1] * eps;
  }
  const float eps = (k[0] * xv - pred) / k[1];     // _predict_eps_from_xstart
  float s = pred * k[3] + k[4] * eps;
  if (noise) s += k[5] * noise[idx];
  sample[idx] = s;
  if (pred_out) pred_out[idx] = pred;
}

// gaussian_diffusion.py p_sample (ancestral step) for an epsilon model with learned-range variance (learn_sigma: the second half
// of the model's channels interpolates between the posterior and the beta log-variance), clip_denoised False, optional
// condition_mean.  cf[b] = {sqrt_recip_ac, sqrt_recipm1_ac, posterior_mean_coef1, posterior_mean_coef2,
// posterior_log_variance_clipped, log(beta), nonzero_mask, 0}.
__global__ __launch_bounds__(256) void p_sample_step_kernel(const float* __restrict__ x, const float* __restrict__ model_out,
                                                            const float* __restrict__ grad, const float* __restrict__ noise,
                                                            const float* __restrict__ cf, int C, long HW, long total,
                                                            float* __restrict__ sample, float* __restrict__ pred_out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long chw = (long)C * HW;
  const int b = (int)(idx / chw);
  const long rem = idx - (long)b * chw;
  const float* k = cf + b * 8;
  const float xv = x[idx];
  const float* mo = model_out + (long)b * 2 * chw;
  const float eps = mo[rem], vv = mo[chw + rem];
  const float frac = (vv + 1.f) / 2.f;
  const float logvar = frac * k[5] + (1.f - frac) * k[4];
  const float pred = k[0] * xv - k[1] * eps;                 // _predict_xstart_from_eps
  float mean = k[2] * pred + k[3] * xv;                      // q_posterior_mean_variance
  if (grad) mean = mean + expf(logvar) * grad[idx];          // condition_mean: mean + variance * gradient
  sample[idx] = mean + k[6] * expf(0.5f * logvar) * noise[idx];
  if (pred_out) pred_out[idx] = pred;
}

// One model evaluation of plms_sample (the pseudo linear multistep sampler of the guided-diffusion fork the reference
// vendors as a submodule): pred_orig = x0 from the network's epsilon; with a condition gradient the score is conditioned
// (condition_score) -> pred; eps = _predict_eps_from_xstart(x, t, pred).  cf[b] = maua_ddim_step's coefficients.
__global__ __launch_bounds__(256) void plms_eps_kernel(const float* __restrict__ x, const float* __restrict__ model_out,
                                                       const float* __restrict__ grad, const float* __restrict__ cf, int C,
                                                       int Cm, long HW, long total, float* __restrict__ eps_out,
                                                       float* __restrict__ pred_out, float* __restrict__ pred_orig_out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long chw = (long)C * HW;
  const int b = (int)(idx / chw);
  const long rem = idx - (long)b * chw;
  const float* k = cf + b * 8;
  const float xv = x[idx];
  const float eps_m = model_out[(long)b * Cm * HW + rem];
  const float pred_orig = k[0] * xv - k[1] * eps_m;
  float pred = pred_orig;
  if (grad) {
    float e = (k[0] * xv - pred) / k[1];
    e = e - k[2] * grad[idx];
    pred = k[0] * xv - k[1] * e;
  }
  eps_out[idx] = (k[0] * xv - pred) / k[1];
  if (pred_out) pred_out[idx] = pred;
  if (pred_orig_out) pred_orig_out[idx] = pred_orig;
}

// The multistep update: eps' = (sum_i w[i] * eps_i) / div (i < n: Adams-Bashforth weights, or (1, 1) / 2 for the improved-Euler start),
// pred' = _predict_xstart_from_eps(x, t, eps'), mean = pred' sqrt(ac_prev) + sqrt(1 - ac_prev) eps',
// sample = mean * nonzero + pred * (1 - nonzero).  cf[b] = {sqrt_recip_ac, sqrt_recipm1_ac, sqrt(ac_prev), sqrt(1 - ac_prev),
// nonzero_mask, 0, 0, 0}
struct PlmsEps { const float* e[4]; float w[4]; float div; int n; };
__global__ __launch_bounds__(256) void plms_update_kernel(const float* __restrict__ x, PlmsEps pe, const float* __restrict__ pred,
                                                          const float* __restrict__ cf, long chw, long total,
                                                          float* __restrict__ sample) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int b = (int)(idx / chw);
  const float* k = cf + b * 8;
  // (separately rounded products and sums, left to right, then the division: the order of the reference's expression)
  float ep = __fmul_rn(pe.w[0], pe.e[0][idx]);
  for (int i = 1; i < pe.n; i++) ep = __fadd_rn(ep, __fmul_rn(pe.w[i], pe.e[i][idx]));
  ep = ep / pe.div;
  const float pp = k[0] * x[idx] - k[1] * ep;
  const float mean = pp * k[2] + k[3] * ep;
  sample[idx] = mean * k[4] + pred[idx] * (1.f - k[4]);
}

// out = a[b] * x + c[b] * y (q_sample: sqrt(ac) * x_start + sqrt(1 - ac) * noise), per-sample coefficients
__global__ __launch_bounds__(256) void axpby_rows_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ ab, long row, long total,
                                                         float* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int b = (int)(idx / row);
  out[idx] = ab[2 * b] * x[idx] + ab[2 * b + 1] * y[idx];
}

// the image-MSE grad module of the guided sampler: g = (img - target) * k[b] (k = 2 scale / numel, one per sample; read from device
// memory so that a captured loop serves every scale), one target per sample or one for all (tstride 0); any NaN raises *flag
__global__ __launch_bounds__(256) void mse_guide_grad_kernel(const float* __restrict__ img, const float* __restrict__ target,
                                                             long tstride, const float* __restrict__ kdev, long row, long total,
                                                             float* __restrict__ out, int* __restrict__ flag) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  bool bad = false;
  if (idx < total) {
    const long b = idx / row, i = idx - b * row;
    const float v = (img[idx] - target[b * tstride + i]) * kdev[b];
    out[idx] = v;
    bad = v != v;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
// GradientGuidedConditioning.forward (guided.py:262-265): a grad module whose output holds a NaN contributes zeros
// eps = the first C channels of a model output [B][Cm][HW], made contiguous (speed "regular": pred_xstart is built from it with
// axpby_rows_kernel, exactly as the step-by-step path does - same kernel, same bits)
__global__ __launch_bounds__(256) void eps_rows_kernel(const float* __restrict__ model_out, long chw, long cmhw, long total,
                                                       float* __restrict__ eps) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long b = i / chw;
  eps[i] = model_out[b * cmhw + (i - b * chw)];
}

__global__ __launch_bounds__(256) void zero_if_flag_kernel(float* __restrict__ g, long total, const int* __restrict__ flag) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  // (an agent-scope load: served by L2, where the previous launch's atomicOr landed - not by the scalar / vector L1)
  if (idx < total && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) g[idx] = 0.f;
}

// ----------------------------------------------------------------------------------------------------- parameters
struct UGN { int C = 0; float* gamma = nullptr; float* beta = nullptr; };
struct UConv { int Ci = 0, Co = 0, Cip = 0, Cop = 0; void* wt = nullptr; float* bias = nullptr; void* wt_t = nullptr; };   // wt_t: the input-gradient convolution (option "vjp")
struct ULin { int K = 0, N = 0; void* w = nullptr; float* bias = nullptr; void* w_t = nullptr; };   // w_t [K][N]: the input-gradient GEMM
struct URes { int Cin, Cout, updown; UGN n1; UConv c1; int emb_off; UGN n2; UConv c2; bool skip; ULin sk; };
struct UAttn { int C, heads; UGN n; ULin qkv, proj; };
struct ULayer { int kind; int idx; };  // 0 conv_in, 1 res, 2 attn
struct UBlock { std::vector<ULayer> layers; int out_ch = 0; };

enum PKind { P_GN_G, P_GN_B, P_CONV_W, P_CONV_B, P_LIN_W, P_LIN_B, P_F32 };
struct PRef { PKind kind; void* obj; float* f32 = nullptr; size_t count = 0; };

// What a kept forward (maua_unet_forward_keep) leaves for maua_unet_vjp: per layer, the tensors its input gradient reads
struct TapeOp {
  int kind, idx;                  // 1 ResBlock, 2 AttentionBlock (index into res / attn)
  const void* x0; int C0;         // the layer's input (virtually concatenated [x0 | x1])
  const void* x1; int C1;
  int H, W;                       // input size
  void* out;                      // its output
  void* h1;                       // ResBlock: conv1's output (the second GroupNorm's input); Attention: qkv
  void* ao;                       // Attention: the attention result (proj_out's input)
  float* st1;                     // statistics of the first / only GroupNorm
  float* st2;                     // ... of the ResBlock's second
  float* lse;                     // Attention: log-sum-exp rows
};

struct Arena {
  char* base = nullptr;
  size_t cap = 0, top = 0, peak = 0;
  bool plan = true;
  void* get(size_t bytes) {
    const size_t o = (top + 255) & ~(size_t)255;
    top = o + bytes;
    if (top > peak) peak = top;
    return plan ? (void*)(uintptr_t)(o + 256) : (void*)(base + o);  // (planning: a non-NULL token, never dereferenced)
  }
};

}  // namespace

struct maua_unet {
  maua_ctx* ctx;
  int image_size, in_ch, mc, out_ch, nrb, head_ch, dtype;
  size_t esize;
  std::vector<float> mult;
  std::vector<int> attn_ds;
  int emb_dim;
  UConv conv_in, conv_out;
  UGN out_norm;
  std::vector<URes> res;
  std::vector<UAttn> attn;
  std::vector<UBlock> input, output;
  UBlock middle;
  int final_ch = 0;
  // f32 timestep path: time_embed.{0,2}, all emb_layers stacked [emb_total][emb_dim]
  float *te0_w = nullptr, *te0_b = nullptr, *te2_w = nullptr, *te2_b = nullptr, *embw = nullptr, *embb = nullptr;
  float* freqs = nullptr;  // [mc / 2] timestep-embedding frequencies (optional upload: "timestep_embedding.freqs")
  int freqs_loaded = 0;
  int emb_total = 0;
  std::unordered_map<std::string, PRef> params;
  std::vector<void*> owned;
  float* ones = nullptr;
  int ones_b = 0, max_ch = 0;
  Arena arena;
  size_t planned_key = 0;  // B, H, W the arena was planned for
  size_t gather_bytes = 0; // split-K workspace of the gather GEMM at that shape
  size_t out_cap = 0;      // bytes behind g_out + g_pred
  int route = 0;           // debugging / ablation: 1 = every 3x3 convolution on the generic kernel
  int psum_off = 0;        // 1: GroupNorm statistics always by their own pass (A/B of the convolution epilogues' piece sums)
  // sampler graph (maua_ddim_sample_loop)
  hipGraphExec_t graph_exec = nullptr;
  size_t graph_key = 0;
  float* emb_table = nullptr;        // [n_steps][emb_total]: every step's emb_layers outputs, computed once per sampler loop
  size_t emb_table_rows = 0;
  const float* emb_row = nullptr;    // non-NULL during a sampler-loop forward: this step's row (all samples share the timestep)
  hipStream_t cap_stream = nullptr;  // capture happens on a private stream (the caller's may be the legacy NULL stream)
  int graph_failed = 0;              // capture / instantiation failed once: the loop runs eagerly from then on
  float *g_x = nullptr, *g_out = nullptr, *g_pred = nullptr, *g_t = nullptr, *g_cf = nullptr;
  int g_steps = 0;
  // guided sampler graph (maua_ddim_guided_loop): its own executable; the sample, the target and every per-step constant live in
  // library buffers, so one capture serves every call of a shape
  hipGraphExec_t gd_exec = nullptr;
  size_t gd_key = 0;
  unsigned long long gd_sec_uid = 0, gd_sec_epoch = 0;   // the secondary model (and its buffers' generation) gd_exec points into
  // text-prompt guidance (maua_unet_set_clip_guide): CLIPGrads instead of the image-MSE module in the guided loop
  std::vector<maua_guide*> gd_guides;   // maua_unet_set_guides: grad modules evaluated (after CLIPGrads, if set) and summed per step
  std::vector<unsigned long long> gd_guide_uids, gd_guide_epochs;   // (epochs: as of the capture)
  int* gd_gflag = nullptr;           // the NaN screen's flag of the guides' sum
  maua_clip* gd_clip = nullptr;
  int* gd_rects = nullptr;           // device [n_steps][batches][cutn][3] (+ [n_steps][batches][cutn] float multiplicities behind them)
  float* gd_mult = nullptr;          // NULL: every cutout counts once
  int gd_last_graph = 0;             // the LAST guided loop replayed a captured graph (maua_unet_guided_graph_active)
  int gd_cutn_total = 0;
  size_t gd_rects_cap = 0;
  std::vector<int> gd_rects_host;
  int gd_rect_steps = 0, gd_cutn = 0, gd_batches = 0;
  float gd_clip_scale = 1.f, gd_clip_clamp = 0.f;
  unsigned long long gd_clip_uid = 0, gd_clip_epoch = 0, gd_guide_gen = 0, gd_guide_gen_seen = 0;
  int gd_failed = 0;
  float* gd_buf = nullptr;           // x | v | pred | eps | img | g | jv | grad | target, B * C * H * W floats each
  size_t gd_cap = 0;
  float* gd_tab = nullptr;           // cos_t [S][B] | (sigma, 1 - sigma) [S][B][2] | grad coefficients [S][B][2] | k [B]
  size_t gd_tab_cap = 0;
  int* gd_flag = nullptr;            // [gd_flags] one NaN flag per step, zeroed before every loop (outside the graph)
  int gd_flags = 0;
  // the guidance branch of a step (secondary forward, grad module, secondary VJP) depends on x only: it runs BESIDE the UNet forward
  // on a side stream (a parallel branch of the captured graph) and joins at the DDIM update; option "guided_fork" = 0: one stream
  int gd_fork = 1;
  hipStream_t side_stream = nullptr, cap_side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // input gradient (maua_unet_forward_keep + maua_unet_vjp; option "vjp" = 1 before the weights are loaded)
  int vjp = 0;
  float* zero_bias = nullptr;        // [max padded channels] zeros: the gradient convolutions have no bias
  std::vector<TapeOp> tape;
  bool tape_valid = false;           // the arena still holds the kept forward the tape describes
  int tape_B = 0, tape_H = 0, tape_W = 0;
  void* tape_h0 = nullptr;           // conv_in's output
  void* tape_hf = nullptr;           // the last block's output (out_norm's input)
  int tape_cf = 0;
  float* tape_stf = nullptr;         // out_norm's statistics
  float* tape_emb = nullptr;         // the emb_layers outputs of that forward
  long tape_emb_ld = 0;
  size_t tape_top = 0;               // arena top behind the kept forward
  float* tape_gather_ws = nullptr;   // the split-K workspace of that forward (the gradient convolutions share it)
};

namespace {

template <typename P>
int dev_alloc(maua_unet* n, P** p, size_t bytes, bool zero = true) {
  MAUA_HIP_CHECK(hipMalloc((void**)p, bytes));
  if (zero) MAUA_HIP_CHECK(hipMemset(*p, 0, bytes));
  n->owned.push_back(*p);
  return MAUA_OK;
}

int make_gn(maua_unet* n, UGN& g, int C, const std::string& name) {
  g.C = C;
  if (int rc = dev_alloc(n, &g.gamma, (size_t)C * 4)) return rc;
  if (int rc = dev_alloc(n, &g.beta, (size_t)C * 4)) return rc;
  n->params[name + ".weight"] = PRef{P_GN_G, &g};
  n->params[name + ".bias"] = PRef{P_GN_B, &g};
  return MAUA_OK;
}
int make_conv(maua_unet* n, UConv& c, int Ci, int Co, const std::string& name) {
  c.Ci = Ci; c.Co = Co; c.Cip = (Ci + 31) / 32 * 32; c.Cop = (Co + 31) / 32 * 32;
  if (int rc = dev_alloc(n, &c.wt, (size_t)9 * c.Cop * c.Cip * n->esize)) return rc;
  if (int rc = dev_alloc(n, &c.bias, (size_t)c.Cop * 4)) return rc;
  n->params[name + ".weight"] = PRef{P_CONV_W, &c};
  n->params[name + ".bias"] = PRef{P_CONV_B, &c};
  return MAUA_OK;
}
int make_lin(maua_unet* n, ULin& l, int K, int N, const std::string& name) {
  l.K = K; l.N = N;
  if (int rc = dev_alloc(n, &l.w, (size_t)N * K * n->esize)) return rc;
  if (int rc = dev_alloc(n, &l.bias, (size_t)N * 4)) return rc;
  n->params[name + ".weight"] = PRef{P_LIN_W, &l};
  n->params[name + ".bias"] = PRef{P_LIN_B, &l};
  return MAUA_OK;
}

// UNetModel.__init__: the module tree (same walk as oracle/diffusion.py unet_structure and maua_amd/diffusion.py)
int build_structure(maua_unet* n) {
  const int mc = n->mc;
  int ch = (int)(n->mult[0] * mc);
  int rc;
  if ((rc = make_conv(n, n->conv_in, n->in_ch, ch, "input_blocks.0.0"))) return rc;
  std::vector<int> chans{ch};
  {
    UBlock b;
    b.layers.push_back({0, 0});
    b.out_ch = ch;
    n->input.push_back(b);
  }
  // (vectors of parameter structs are referenced by pointer from the name table: reserve so they never move)
  const size_t nlev = n->mult.size();
  n->res.reserve(nlev * (2 * n->nrb + 3) + 4);
  n->attn.reserve(nlev * (2 * n->nrb + 1) + 2);
  auto is_attn = [&](int ds) {
    for (int a : n->attn_ds)
      if (a == ds) return true;
    return false;
  };
  auto add_res = [&](UBlock& b, const std::string& name, int cin, int cout, int updown) -> int {
    n->res.emplace_back();
    URes& r = n->res.back();
    r.Cin = cin; r.Cout = cout; r.updown = updown; r.skip = cin != cout;
    int rc2;
    if ((rc2 = make_gn(n, r.n1, cin, name + ".in_layers.0"))) return rc2;
    if ((rc2 = make_conv(n, r.c1, cin, cout, name + ".in_layers.2"))) return rc2;
    r.emb_off = n->emb_total;
    n->emb_total += 2 * cout;
    n->params[name + ".emb_layers.1.weight"] = PRef{P_F32, &r, nullptr, (size_t)2 * cout * n->emb_dim};
    n->params[name + ".emb_layers.1.bias"] = PRef{P_F32, &r, nullptr, (size_t)2 * cout};
    if ((rc2 = make_gn(n, r.n2, cout, name + ".out_layers.0"))) return rc2;
    if ((rc2 = make_conv(n, r.c2, cout, cout, name + ".out_layers.3"))) return rc2;
    if (r.skip && (rc2 = make_lin(n, r.sk, cin, cout, name + ".skip_connection"))) return rc2;
    b.layers.push_back({1, (int)n->res.size() - 1});
    b.out_ch = cout;
    return MAUA_OK;
  };
  auto add_attn = [&](UBlock& b, const std::string& name, int c) -> int {
    n->attn.emplace_back();
    UAttn& a = n->attn.back();
    a.C = c; a.heads = c / n->head_ch;
    int rc2;
    if ((rc2 = make_gn(n, a.n, c, name + ".norm"))) return rc2;
    if ((rc2 = make_lin(n, a.qkv, c, 3 * c, name + ".qkv"))) return rc2;
    if ((rc2 = make_lin(n, a.proj, c, c, name + ".proj_out"))) return rc2;
    b.layers.push_back({2, (int)n->attn.size() - 1});
    return MAUA_OK;
  };
  int ds = 1, bi = 1;
  for (size_t level = 0; level < nlev; level++) {
    for (int i = 0; i < n->nrb; i++) {
      UBlock b;
      const std::string pfx = "input_blocks." + std::to_string(bi);
      const int co = (int)(n->mult[level] * mc);
      if ((rc = add_res(b, pfx + ".0", ch, co, 0))) return rc;
      ch = co;
      if (is_attn(ds) && (rc = add_attn(b, pfx + ".1", ch))) return rc;
      n->input.push_back(b);
      chans.push_back(ch);
      bi++;
    }
    if (level != nlev - 1) {
      UBlock b;
      if ((rc = add_res(b, "input_blocks." + std::to_string(bi) + ".0", ch, ch, 1))) return rc;
      n->input.push_back(b);
      chans.push_back(ch);
      ds *= 2;
      bi++;
    }
  }
  if ((rc = add_res(n->middle, "middle_block.0", ch, ch, 0))) return rc;
  if ((rc = add_attn(n->middle, "middle_block.1", ch))) return rc;
  if ((rc = add_res(n->middle, "middle_block.2", ch, ch, 0))) return rc;
  bi = 0;
  for (int level = (int)nlev - 1; level >= 0; level--) {
    for (int i = 0; i <= n->nrb; i++) {
      UBlock b;
      const std::string pfx = "output_blocks." + std::to_string(bi);
      const int ich = chans.back();
      chans.pop_back();
      const int co = (int)(mc * n->mult[level]);
      int li = 0;
      if ((rc = add_res(b, pfx + "." + std::to_string(li++), ch + ich, co, 0))) return rc;
      ch = co;
      if (is_attn(ds) && (rc = add_attn(b, pfx + "." + std::to_string(li++), ch))) return rc;
      if (level && i == n->nrb) {
        if ((rc = add_res(b, pfx + "." + std::to_string(li++), ch, ch, 2))) return rc;
        ds /= 2;
      }
      n->output.push_back(b);
      bi++;
    }
  }
  n->final_ch = ch;
  if ((rc = make_gn(n, n->out_norm, ch, "out.0"))) return rc;
  if ((rc = make_conv(n, n->conv_out, ch, n->out_ch, "out.2"))) return rc;
  // f32 timestep path
  const int E = n->emb_dim;
  if ((rc = dev_alloc(n, &n->te0_w, (size_t)E * mc * 4))) return rc;
  if ((rc = dev_alloc(n, &n->te0_b, (size_t)E * 4))) return rc;
  if ((rc = dev_alloc(n, &n->te2_w, (size_t)E * E * 4))) return rc;
  if ((rc = dev_alloc(n, &n->te2_b, (size_t)E * 4))) return rc;
  if ((rc = dev_alloc(n, &n->embw, (size_t)n->emb_total * E * 4))) return rc;
  if ((rc = dev_alloc(n, &n->embb, (size_t)n->emb_total * 4))) return rc;
  if ((rc = dev_alloc(n, &n->freqs, (size_t)(mc / 2) * 4))) return rc;
  n->params["timestep_embedding.freqs"] = PRef{P_F32, nullptr, n->freqs, (size_t)(mc / 2)};
  n->params["time_embed.0.weight"] = PRef{P_F32, nullptr, n->te0_w, (size_t)E * mc};
  n->params["time_embed.0.bias"] = PRef{P_F32, nullptr, n->te0_b, (size_t)E};
  n->params["time_embed.2.weight"] = PRef{P_F32, nullptr, n->te2_w, (size_t)E * E};
  n->params["time_embed.2.bias"] = PRef{P_F32, nullptr, n->te2_b, (size_t)E};
  for (auto& kv : n->params) {
    if (kv.second.kind != P_F32 || kv.second.f32) continue;
    const URes* r = (const URes*)kv.second.obj;
    const bool is_w = kv.first.size() > 7 && kv.first.compare(kv.first.size() - 7, 7, ".weight") == 0;
    kv.second.f32 = is_w ? n->embw + (size_t)r->emb_off * E : n->embb + r->emb_off;
  }
  n->max_ch = 0;
  for (auto& r : n->res) n->max_ch = std::max(n->max_ch, std::max(r.Cin, r.Cout));
  n->max_ch = std::max(n->max_ch, 32);
  return MAUA_OK;
}

// ------------------------------------------------------------------------------------------------------ forward
template <typename T>
struct Runner {
  maua_unet* n;
  hipStream_t st;
  int B;
  bool plan;
  Arena& ar;
  float* emb_all = nullptr;  // [B][emb_total]
  // piece sums left by the LDS-direct convolution for the GroupNorm that reads its output: tensor -> (buffer, rows per sample)
  std::unordered_map<const void*, std::pair<const float*, int>> psums;
  float* gather_ws = nullptr;
  size_t gather_ws_bytes = 0;
  bool keep = false;         // record the tape and keep what the input gradient needs out of the per-layer scratch
  int g_channels = 0;        // backward: channels of the output gradient handed in (0: all of them)
  std::vector<TapeOp>* tape = nullptr;
  // gradients known so far, by forward tensor (backward pass)
  std::unordered_map<const void*, T*> grads;

  Runner(maua_unet* n_, hipStream_t s, int B_, bool plan_, bool keep_ = false) : n(n_), st(s), B(B_), plan(plan_), ar(n_->arena), keep(keep_) {}
  float* stats_alloc() { return keep ? (float*)ar.get((size_t)B * 32 * 2 * 4) : nullptr; }

  T* alloc(long px, int C) { return (T*)ar.get((size_t)px * C * sizeof(T)); }
  // room for the piece sums of a [B][H][W][C] tensor an LDS-direct convolution may produce (NULL where it cannot)
  float* psum_alloc(int H, int W, int C) {
    if (sizeof(T) != 2 || n->psum_off || H % 8 || W % 32 || C % 128) return nullptr;
    return (float*)ar.get((size_t)B * (H / 8) * (W / 32) * (C / 8) * 64);
  }

  // y = conv3x3(x) + bias (+ res); x, y, res dense NHWC [B][H][W][.]
  int conv(const UConv& c, const T* x, T* y, int H, int W, const T* res, float* psum = nullptr) {
    const size_t ws_need = gather_conv_supported(n->dtype, c.Cip, c.Cop, H, W) ? gather_conv_workspace(n->dtype, B, H, W, c.Cip, c.Cop) : 0;
    if (plan) {
      if (ws_need > gather_ws_bytes) gather_ws_bytes = ws_need;
      return MAUA_OK;
    }
    psums.erase(y);
    ConvArgs a{};
    a.x = x; a.x_bstride = (long)H * W * c.Cip; a.w = c.wt; a.s = nullptr; a.d = nullptr; a.noise = nullptr; a.bias = c.bias;
    a.y = y; a.B = B; a.H = H; a.W = W; a.Ci = c.Cip; a.Co = c.Cop; a.up = 1;
    a.act = MAUA_ACT_LINEAR; a.alpha = 1.f; a.gain = 1.f; a.clamp = -1.f;
    a.res = res; a.res_pstride = c.Cop; a.res_bstride = (long)H * W * c.Cop;
    const bool bf = n->dtype == MAUA_BF16;
    if (n->route != 1) {
      const bool dma_wide = bf && dma_conv_supported(n->dtype, c.Cip, c.Cop, 1, H, W);
      const bool dma_narrow = bf && dma_conv_narrow_supported(n->dtype, c.Cip, c.Cop, H, W);
      const bool gather = ws_need > 0;
      // enough workgroups for the chip on the LDS-direct kernel (8 x 32 pixel tiles x N tiles of 256 / 128 / Co channels)?
      const long tiles = (long)B * (H / 8) * (W / 32);
      long dma_wgs = dma_wide ? tiles * (c.Cop % 256 == 0 ? c.Cop / 256 : c.Cop / 128) : tiles;
      if (dma_wide && c.Cop % 256 == 0 && dma_wgs < 512) {  // fewer than two rounds of big tiles: 128-channel tiles, 2 per CU
        a.variant = 128;
        dma_wgs *= 2;
      }
      auto dma = [&]() -> int {
        if (dma_wide && psum) {   // the wide tiles can leave the statistics of what they store
          a.psum = psum;
          psums[y] = {psum, dma_psum_rows(a)};
        }
        return launch_modconv_dma(st, a);
      };
      if ((dma_wide || dma_narrow) && (dma_wgs >= 128 || !gather || n->route == 2)) return dma();
      if (gather && n->route != 2) return launch_conv_gather(st, n->dtype, a, gather_ws);
      if (dma_wide || dma_narrow) return dma();
    }
    a.s = n->ones;
    return launch_modconv3x3(st, n->dtype, a);
  }

  // GroupNorm (+ scale-shift) (+ SiLU) (+ resample) of [x0 | x1] -> y (dense, C0 + C1 channels); xr: resampled raw x0
  int gn(const UGN& g, const T* x0, int C0, const T* x1, int C1, int H, int W, const float* ss, int silu, int mode, T* y,
         T* xr, float* stats_keep = nullptr) {
    const size_t mark = ar.top;
    float* stats = stats_keep ? stats_keep : (float*)ar.get((size_t)B * 32 * 2 * 4);
    double* part = (double*)ar.get(gn_part_bytes(B, C0 + C1, (long)H * W, (int)sizeof(T)));
    int rc = MAUA_OK;
    if (!plan) {
      const float *ps0 = nullptr, *ps1 = nullptr;
      int rows0 = 0, rows1 = 0;
      auto it0 = psums.find(x0);
      if (it0 != psums.end()) { ps0 = it0->second.first; rows0 = it0->second.second; }
      if (x1) {
        auto it1 = psums.find(x1);
        if (it1 != psums.end()) { ps1 = it1->second.first; rows1 = it1->second.second; }
      }
      rc = launch_group_norm<T>(st, x0, C0, x1, C1, B, H, W, g.gamma, g.beta, ss, n->emb_row ? 0L : (long)n->emb_total, silu,
                                mode, y, xr, part, stats, ps0, rows0, ps1, rows1);
    }
    ar.top = mark;
    return rc;
  }

  // ResBlock on [x0 | x1] (H x W) -> out (dense, Cout channels at the resampled size)
  int resblock(const URes& r, const T* x0, int C0, const T* x1, int C1, int H, int W, T* out, float* out_psum) {
    const int Ho = r.updown == 1 ? H / 2 : (r.updown == 2 ? H * 2 : H), Wo = r.updown == 1 ? W / 2 : (r.updown == 2 ? W * 2 : W);
    const long opx = (long)B * Ho * Wo;
    // (kept forward: conv1's output and both GroupNorms' statistics outlive the block's scratch)
    T* h1_keep = keep ? alloc(opx, r.Cout) : nullptr;
    float *st1 = stats_alloc(), *st2 = stats_alloc();
    const size_t mark = ar.top;
    T* a1 = alloc(opx, r.Cin);
    T* xr = r.updown ? alloc(opx, r.Cin) : nullptr;
    int rc;
    if ((rc = gn(r.n1, x0, C0, x1, C1, H, W, nullptr, 1, r.updown, a1, xr, st1))) return rc;
    T* h1 = keep ? h1_keep : alloc(opx, r.Cout);
    float* ps1 = psum_alloc(Ho, Wo, r.Cout);
    if ((rc = conv(r.c1, a1, h1, Ho, Wo, nullptr, ps1))) return rc;
    T* a2 = alloc(opx, r.Cout);
    if ((rc = gn(r.n2, h1, r.Cout, nullptr, 0, Ho, Wo, emb_all + r.emb_off, 1, 0, a2, nullptr, st2))) return rc;
    if (keep) {
      TapeOp op{};
      op.kind = 1; op.idx = (int)(&r - n->res.data()); op.x0 = x0; op.C0 = C0; op.x1 = x1; op.C1 = C1; op.H = H; op.W = W; op.out = out;
      op.h1 = h1; op.st1 = st1; op.st2 = st2;
      tape->push_back(op);
    }
    const T* resid;
    if (r.skip) {  // 1x1 skip_connection over the (virtually concatenated) input
      T* sk = alloc(opx, r.Cout);
      if (!plan) {
        GemmArgs g{};
        g.a0 = x0; g.lda0 = C0; g.K0 = C0; g.a1 = x1; g.lda1 = C1; g.K1 = C1; g.w = r.sk.w; g.bias = r.sk.bias;
        g.c = sk; g.ldc = r.Cout; g.M = opx; g.N = r.Cout;
        if ((rc = launch_gemm_nt(st, n->dtype, g))) return rc;
      }
      resid = sk;
    } else {
      resid = r.updown ? xr : x0;  // (Cin == Cout: never a concatenated input)
    }
    if ((rc = conv(r.c2, a2, out, Ho, Wo, resid, out_psum))) return rc;
    ar.top = mark;
    return MAUA_OK;
  }

  // AttentionBlock, in place on x (dense [B][HW][C])
  int attention(const UAttn& at, T* x, int H, int W, T* out) {
    const long px = (long)B * H * W;
    T* qkv_keep = keep ? alloc(px, 3 * at.C) : nullptr;
    T* ao_keep = keep ? alloc(px, at.C) : nullptr;
    float* lse = keep ? (float*)ar.get((size_t)B * at.heads * H * W * 4) : nullptr;
    float* st1 = stats_alloc();
    const size_t mark = ar.top;
    T* a = alloc(px, at.C);
    int rc;
    if ((rc = gn(at.n, x, at.C, nullptr, 0, H, W, nullptr, 0, 0, a, nullptr, st1))) return rc;
    T* qkv = keep ? qkv_keep : alloc(px, 3 * at.C);
    T* ao = keep ? ao_keep : alloc(px, at.C);
    if (keep) {
      TapeOp op{};
      op.kind = 2; op.idx = (int)(&at - n->attn.data()); op.x0 = x; op.C0 = at.C; op.H = H; op.W = W; op.out = out; op.h1 = qkv; op.ao = ao;
      op.st1 = st1; op.lse = lse;
      tape->push_back(op);
    }
    if (!plan) {
      GemmArgs g{};
      g.a0 = a; g.lda0 = at.C; g.K0 = at.C; g.w = at.qkv.w; g.bias = at.qkv.bias; g.c = qkv; g.ldc = 3 * at.C; g.M = px;
      g.N = 3 * at.C;
      if ((rc = launch_gemm_nt(st, n->dtype, g))) return rc;
      AttnArgs aa{};
      aa.qkv = qkv; aa.out = ao; aa.B = B; aa.T = H * W; aa.heads = at.heads; aa.D = n->head_ch; aa.ld_qkv = 3 * at.C;
      aa.ld_out = at.C; aa.scale = 1.f / sqrtf((float)n->head_ch); aa.lse = lse;
      if ((rc = launch_attention(st, n->dtype, aa))) return rc;
      GemmArgs p{};
      p.a0 = ao; p.lda0 = at.C; p.K0 = at.C; p.w = at.proj.w; p.bias = at.proj.bias; p.res = x; p.ldr = at.C; p.c = out;
      p.ldc = at.C; p.M = px; p.N = at.C;
      if ((rc = launch_gemm_nt(st, n->dtype, p))) return rc;
    }
    ar.top = mark;
    return MAUA_OK;
  }

  // one TimestepEmbedSequential: layers applied to [x0 | x1]; returns the block's output tensor (allocated persistently)
  int block(const UBlock& b, const T* x0, int C0, const T* x1, int C1, int& H, int& W, T** outp) {
    const T* cur0 = x0; int c0 = C0; const T* cur1 = x1; int c1 = C1;
    T* out = nullptr;
    for (const ULayer& l : b.layers) {
      if (l.kind == 1) {
        const URes& r = n->res[l.idx];
        const int Ho = r.updown == 1 ? H / 2 : (r.updown == 2 ? H * 2 : H), Wo = r.updown == 1 ? W / 2 : (r.updown == 2 ? W * 2 : W);
        out = alloc((long)B * Ho * Wo, r.Cout);
        float* pso = psum_alloc(Ho, Wo, r.Cout);   // (lives as long as the tensor: later blocks' GroupNorms read it)
        if (int rc = resblock(r, cur0, c0, cur1, c1, H, W, out, pso)) return rc;
        H = Ho; W = Wo;
        cur0 = out; c0 = r.Cout; cur1 = nullptr; c1 = 0;
      } else {
        const UAttn& at = n->attn[l.idx];
        T* o2 = alloc((long)B * H * W, at.C);
        psums.erase(o2);
        if (int rc = attention(at, const_cast<T*>(cur0), H, W, o2)) return rc;
        out = o2;
        cur0 = out; c0 = at.C;
      }
    }
    *outp = out;
    return MAUA_OK;
  }

  int forward(const float* x_nchw, const float* t, int H, int W, float* out_nchw) {
    int rc;
    const int E = n->emb_dim, mc = n->mc;
    const long px = (long)B * H * W;
    // ---- timestep path (f32)
    float* e0 = (float*)ar.get((size_t)B * mc * 4);
    float* e1 = (float*)ar.get((size_t)B * E * 4);
    float* e2 = (float*)ar.get((size_t)B * E * 4);
    emb_all = (float*)ar.get((size_t)B * n->emb_total * 4);
    if (!plan && n->emb_row) {
      emb_all = const_cast<float*>(n->emb_row);   // (read-only; the samples' rows coincide: stride 0 below)
    } else if (!plan) {
      const int half = mc / 2;
      hipLaunchKernelGGL(timestep_embedding_kernel, dim3((B * half + 255) / 256), dim3(256), 0, st, t,
                         n->freqs_loaded ? n->freqs : nullptr, e0, B, mc);
      hipLaunchKernelGGL(linear_rows_kernel, dim3((E + 3) / 4), dim3(256), 0, st, e0, n->te0_w, n->te0_b, e1, B, mc, E, 0, 1);
      hipLaunchKernelGGL(linear_rows_kernel, dim3((E + 3) / 4), dim3(256), 0, st, e1, n->te2_w, n->te2_b, e2, B, E, E, 0, 0);
      hipLaunchKernelGGL(linear_rows_kernel, dim3((n->emb_total + 3) / 4), dim3(256), 0, st, e2, n->embw, n->embb, emb_all, B,
                         E, n->emb_total, 1, 0);
      MAUA_HIP_CHECK(hipGetLastError());
    }
    // split-K workspace of the gather GEMM: sized in the planning pass, placed here in the real one
    gather_ws = (float*)ar.get(plan ? 0 : n_gather_bytes);
    if (keep) n->tape_gather_ws = gather_ws;
    // ---- input: NCHW f32 -> NHWC T, channels padded to 32
    T* xin = alloc(px, n->conv_in.Cip);
    if (!plan && (rc = launch_nchw_to_nhwc<float, T>(st, x_nchw, xin, B, n->in_ch, H * W, n->conv_in.Cip))) return rc;
    std::vector<T*> hs;
    std::vector<int> hs_c;
    T* h = alloc(px, n->conv_in.Cop);
    if ((rc = conv(n->conv_in, xin, h, H, W, nullptr))) return rc;   // (3 -> C input convolution: generic kernel, no piece sums)
    int ch = n->conv_in.Co, hh = H, ww = W;
    T* const h0_first = h;
    hs.push_back(h); hs_c.push_back(ch);
    for (size_t i = 1; i < n->input.size(); i++) {
      T* o;
      if ((rc = block(n->input[i], h, ch, nullptr, 0, hh, ww, &o))) return rc;
      h = o; ch = n->input[i].out_ch;
      hs.push_back(h); hs_c.push_back(ch);
    }
    {
      T* o;
      if ((rc = block(n->middle, h, ch, nullptr, 0, hh, ww, &o))) return rc;
      h = o;
    }
    for (size_t i = 0; i < n->output.size(); i++) {
      T* skip = hs.back(); const int sc = hs_c.back();
      hs.pop_back(); hs_c.pop_back();
      T* o;
      if ((rc = block(n->output[i], h, ch, skip, sc, hh, ww, &o))) return rc;
      h = o; ch = n->output[i].out_ch;
    }
    // ---- out: GroupNorm -> SiLU -> conv -> NCHW f32
    float* stf = stats_alloc();
    if (keep) {
      n->tape_h0 = (void*)h0_first; n->tape_hf = h; n->tape_cf = ch; n->tape_stf = stf; n->tape_emb = emb_all;
      n->tape_emb_ld = n->emb_row ? 0L : (long)n->emb_total;
    }
    T* a = alloc(px, ch);
    if ((rc = gn(n->out_norm, h, ch, nullptr, 0, hh, ww, nullptr, 1, 0, a, nullptr, stf))) return rc;
    T* y = alloc(px, n->conv_out.Cop);
    if ((rc = conv(n->conv_out, a, y, hh, ww, nullptr))) return rc;
    if (!plan && (rc = launch_nhwc_to_nchw<T, float>(st, y, out_nchw, B, n->out_ch, H * W, n->conv_out.Cop))) return rc;
    return MAUA_OK;
  }

  size_t n_gather_bytes = 0;

  // ------------------------------------------------------------------------------------------ input gradient (guided.py:250-272)
  // dx = conv3x3(dy) with the transposed, spatially flipped kernel (UConv.wt_t), no bias, optional residual
  int conv_t(const UConv& c, const T* dy, T* dx, int H, int W, const T* res) {
    UConv v;
    v.Ci = c.Co; v.Co = c.Ci; v.Cip = c.Cop; v.Cop = c.Cip; v.wt = c.wt_t; v.bias = n->zero_bias;
    return conv(v, dy, dx, H, W, res);
  }
  // c [M][N] = a [M][K] w_t^T (+ res), w_t rows [n0, n0 + N) of a transposed linear weight [Kin][Nout]
  int gemm_t(const ULin& l, int n0, int N, const T* a, T* c, long M, const T* res) {
    if (plan) return MAUA_OK;
    GemmArgs g{};
    g.a0 = a; g.lda0 = l.N; g.K0 = l.N; g.w = (const char*)l.w_t + (size_t)n0 * l.N * sizeof(T); g.bias = nullptr;
    g.res = res; g.ldr = N; g.c = c; g.ldc = N; g.M = M; g.N = N;
    return launch_gemm_nt(st, n->dtype, g);
  }
  int gn_vjp(const UGN& g, const void* x0, int C0, const void* x1, int C1, int H, int W, const float* stats, const float* ss, int silu,
             int mode, const T* dy, const T* dres, const T* add0, const T* add1, T* dx0, T* dx1) {
    const size_t mark = ar.top;
    void* ws = ar.get(group_norm_vjp_workspace(B, C0 + C1, (long)H * W, (int)sizeof(T)));
    int rc = MAUA_OK;
    if (!plan) {
      GnVjpArgs a{};
      a.x0 = x0; a.C0 = C0; a.x1 = x1; a.C1 = C1; a.stats = stats; a.gamma = g.gamma; a.beta = g.beta; a.ss = ss; a.ss_ld = n->tape_emb_ld;
      a.silu = silu; a.mode = mode; a.dy = dy; a.dres = dres; a.add0 = add0; a.add1 = add1; a.dx0 = dx0; a.dx1 = dx1; a.B = B; a.H = H; a.W = W;
      rc = launch_group_norm_vjp(st, n->dtype, a, ws);
    }
    ar.top = mark;
    return rc;
  }
  // the gradient buffer of a forward tensor: the one already there (a second consumer adds to it) or a fresh one
  T* grad_of(const void* t, long px, int C, bool* had) {
    auto it = grads.find(t);
    *had = it != grads.end();
    if (*had) return it->second;
    T* g = alloc(px, C);
    grads[t] = g;
    return g;
  }

  int resblock_vjp(const TapeOp& op) {
    const URes& r = n->res[op.idx];
    const int H = op.H, W = op.W;
    const int Ho = r.updown == 1 ? H / 2 : (r.updown == 2 ? H * 2 : H), Wo = r.updown == 1 ? W / 2 : (r.updown == 2 ? W * 2 : W);
    const long ipx = (long)B * H * W, opx = (long)B * Ho * Wo;
    auto it = grads.find(op.out);
    if (it == grads.end()) return fail("maua_unet_vjp: a block's output has no gradient (tape out of order)");
    const T* dout = it->second;
    bool had0 = false, had1 = false;
    T* dx0 = grad_of(op.x0, ipx, op.C0, &had0);
    T* dx1 = op.x1 ? grad_of(op.x1, ipx, op.C1, &had1) : nullptr;
    const size_t mark = ar.top;
    int rc;
    T* d_a2 = alloc(opx, r.Cout);
    if ((rc = conv_t(r.c2, dout, d_a2, Ho, Wo, nullptr))) return rc;
    T* d_h1 = alloc(opx, r.Cout);
    if ((rc = gn_vjp(r.n2, op.h1, r.Cout, nullptr, 0, Ho, Wo, op.st2, n->tape_emb + r.emb_off, 1, 0, d_a2, nullptr, nullptr, nullptr, d_h1,
                     nullptr)))
      return rc;
    T* d_a1 = alloc(opx, r.Cin);
    if ((rc = conv_t(r.c1, d_h1, d_a1, Ho, Wo, nullptr))) return rc;
    if (r.skip) {
      // out = skip_connection([x0 | x1]) + ...: the 1x1 convolution's input gradients land in dx0 / dx1 first
      if ((rc = gemm_t(r.sk, 0, op.C0, dout, dx0, opx, had0 ? dx0 : nullptr))) return rc;
      if (op.x1 && (rc = gemm_t(r.sk, op.C0, op.C1, dout, dx1, opx, had1 ? dx1 : nullptr))) return rc;
      rc = gn_vjp(r.n1, op.x0, op.C0, op.x1, op.C1, H, W, op.st1, nullptr, 1, 0, d_a1, nullptr, dx0, dx1, dx0, dx1);
    } else {
      // identity (or resampled: x_upd) residual: through the same resampling's adjoint inside the pass
      rc = gn_vjp(r.n1, op.x0, op.C0, nullptr, 0, H, W, op.st1, nullptr, 1, r.updown, d_a1, dout, had0 ? dx0 : nullptr, nullptr, dx0, nullptr);
    }
    ar.top = mark;
    return rc;
  }

  int attention_vjp(const TapeOp& op) {
    const UAttn& at = n->attn[op.idx];
    const int H = op.H, W = op.W;
    const long px = (long)B * H * W;
    auto it = grads.find(op.out);
    if (it == grads.end()) return fail("maua_unet_vjp: a block's output has no gradient (tape out of order)");
    const T* dout = it->second;
    bool had = false;
    T* dx = grad_of(op.x0, px, at.C, &had);
    const size_t mark = ar.top;
    int rc;
    T* d_ao = alloc(px, at.C);
    if ((rc = gemm_t(at.proj, 0, at.C, dout, d_ao, px, nullptr))) return rc;
    T* d_qkv = alloc(px, 3 * at.C);
    float* delta = (float*)ar.get((size_t)B * at.heads * H * W * 4);
    if (!plan) {
      AttnVjpArgs a{};
      a.qkv = op.h1; a.out = op.ao; a.d_out = d_ao; a.lse = op.lse; a.d_qkv = d_qkv; a.delta = delta; a.B = B; a.T = H * W; a.heads = at.heads;
      a.D = n->head_ch; a.ld_qkv = 3 * at.C; a.ld_out = at.C; a.scale = 1.f / sqrtf((float)n->head_ch);
      if ((rc = launch_attention_vjp(st, n->dtype, a))) return rc;
    }
    T* d_a = alloc(px, at.C);
    if ((rc = gemm_t(at.qkv, 0, at.C, d_qkv, d_a, px, nullptr))) return rc;
    rc = gn_vjp(at.n, op.x0, at.C, nullptr, 0, H, W, op.st1, nullptr, 0, 0, d_a, dout, had ? dx : nullptr, nullptr, dx, nullptr);
    ar.top = mark;
    return rc;
  }

  // g_out [B][out_ch][H][W] f32 -> g_x [B][in_ch][H][W] f32; walks n->tape (or, planning, `ops`) backwards
  int backward(const std::vector<TapeOp>& ops, const float* g_out, int H, int W, float* g_x) {
    int rc;
    const long px = (long)B * H * W;
    T* dy = alloc(px, n->conv_out.Cop);
    if (!plan && (rc = launch_nchw_to_nhwc<float, T>(st, g_out, dy, B, g_channels ? g_channels : n->out_ch, H * W, n->conv_out.Cop))) return rc;
    T* d_a = alloc(px, n->tape_cf);
    if ((rc = conv_t(n->conv_out, dy, d_a, H, W, nullptr))) return rc;
    bool had = false;
    T* d_hf = grad_of(n->tape_hf, px, n->tape_cf, &had);
    if ((rc = gn_vjp(n->out_norm, n->tape_hf, n->tape_cf, nullptr, 0, H, W, n->tape_stf, nullptr, 1, 0, d_a, nullptr, nullptr, nullptr, d_hf,
                     nullptr)))
      return rc;
    for (size_t i = ops.size(); i-- > 0;) {
      const TapeOp& op = ops[i];
      if ((rc = op.kind == 1 ? resblock_vjp(op) : attention_vjp(op))) return rc;
    }
    auto it = grads.find(n->tape_h0);
    if (it == grads.end()) return fail("maua_unet_vjp: the input convolution's output has no gradient");
    T* d_xin = alloc(px, n->conv_in.Cip);
    if ((rc = conv_t(n->conv_in, it->second, d_xin, H, W, nullptr))) return rc;
    if (!plan && (rc = launch_nhwc_to_nchw<T, float>(st, d_xin, g_x, B, n->in_ch, H * W, n->conv_in.Cip))) return rc;
    return MAUA_OK;
  }
};

size_t shape_key(int B, int H, int W) { return ((size_t)B << 40) ^ ((size_t)H << 20) ^ (size_t)W; }

// the captured sampler loops hold pointers into the arena / the per-step tables: whatever moves those drops both executables
void drop_sampler_graphs(maua_unet* n) {
  if (n->graph_exec) { hipGraphExecDestroy(n->graph_exec); n->graph_exec = nullptr; n->graph_key = 0; }
  if (n->gd_exec) { hipGraphExecDestroy(n->gd_exec); n->gd_exec = nullptr; n->gd_key = 0; }
}

template <typename T>
int run_forward(maua_unet* n, const float* x, const float* t, int B, int H, int W, float* out, bool keep = false) {
  hipStream_t st = n->ctx->stream;
  const size_t key = shape_key(B, H, W) ^ (keep ? (size_t)1 << 62 : 0);
  n->tape_valid = false;
  if (key != n->planned_key) {
    // planning pass: the same walk with a counting arena (a kept forward: followed by the gradient's walk)
    n->arena.plan = true; n->arena.top = 0; n->arena.peak = 0;
    Runner<T> pr(n, st, B, true, keep);
    std::vector<TapeOp> ptape;
    pr.tape = &ptape;
    if (int rc = pr.forward(x, t, H, W, out)) return rc;
    if (keep)
      if (int rc = pr.backward(ptape, nullptr, H, W, nullptr)) return rc;
    const size_t need = n->arena.peak + pr.gather_ws_bytes + (1 << 20);
    if (need > n->arena.cap) {
      MAUA_HIP_CHECK(hipStreamSynchronize(st));
      if (n->arena.base) hipFree(n->arena.base);
      n->arena.base = nullptr; n->arena.cap = 0;
      MAUA_HIP_CHECK(hipMalloc((void**)&n->arena.base, need));
      n->arena.cap = need;
      drop_sampler_graphs(n);   // (their pointers were into the old arena)
    }
    // a replan inside the same arena keeps the captured loops: each was captured under its own mode's plan (the graphs' keys hold
    // shape and mode), its offsets are baked in and stay inside [base, base + cap) - alternating an ordinary and a kept forward at
    // one shape no longer destroys and recaptures both
    n->gather_bytes = pr.gather_ws_bytes;
    n->planned_key = key;
  }
  if (B > n->ones_b) {
    MAUA_HIP_CHECK(hipStreamSynchronize(st));
    std::vector<float> h((size_t)B * n->max_ch, 1.f);
    float* p;
    MAUA_HIP_CHECK(hipMalloc((void**)&p, h.size() * 4));
    MAUA_HIP_CHECK(hipMemcpy(p, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    if (n->ones) hipFree(n->ones);
    n->ones = p; n->ones_b = B;
  }
  n->arena.plan = false; n->arena.top = 0;
  Runner<T> r(n, st, B, false, keep);
  r.n_gather_bytes = n->gather_bytes;
  n->tape.clear();
  r.tape = &n->tape;
  if (int rc = r.forward(x, t, H, W, out)) return rc;
  if (keep) { n->tape_valid = true; n->tape_B = B; n->tape_H = H; n->tape_W = W; n->tape_top = n->arena.top; }
  return MAUA_OK;
}

// the gradient's walk continues on the arena where the kept forward stopped (its tensors stay where they are)
// g_channels: channels of g_out actually handed in ([B][g_channels][H][W]; the remaining output channels' gradient is zero)
template <typename T>
int run_vjp(maua_unet* n, const float* g_out, float* g_x, int g_channels = 0) {
  hipStream_t st = n->ctx->stream;
  n->arena.plan = false; n->arena.top = n->tape_top;
  Runner<T> r(n, st, n->tape_B, false, true);
  r.n_gather_bytes = n->gather_bytes;
  r.gather_ws = n->tape_gather_ws;
  r.g_channels = g_channels ? g_channels : n->out_ch;
  return r.backward(n->tape, g_out, n->tape_H, n->tape_W, g_x);
}

}  // namespace

extern "C" {

int maua_unet_create(maua_ctx* ctx, int image_size, int in_channels, int model_channels, int out_channels, int num_res_blocks,
                     const float* channel_mult, int n_mult, const int* attention_ds, int n_attn, int num_head_channels,
                     int dtype, maua_unet** out) {
  MAUA_REQUIRE(ctx && out && channel_mult && n_mult > 0 && (attention_ds || n_attn == 0), "maua_unet_create: NULL argument");
  MAUA_REQUIRE(dtype == MAUA_F32 || dtype == MAUA_BF16, "maua_unet_create: dtype must be MAUA_F32 or MAUA_BF16");
  MAUA_REQUIRE(in_channels > 0 && in_channels <= 32 && out_channels > 0 && out_channels <= 32,
               "maua_unet_create: image channels must be 1..32");
  MAUA_REQUIRE(model_channels > 0 && model_channels % 32 == 0 && num_res_blocks > 0, "maua_unet_create: model_channels % 32");
  MAUA_REQUIRE(attention_supported(num_head_channels), "maua_unet_create: num_head_channels must be 32 or 64");
  for (int i = 0; i < n_mult; i++) {
    const int c = (int)(channel_mult[i] * model_channels);
    MAUA_REQUIRE(c > 0 && c % 32 == 0, "maua_unet_create: every level's channel count must be a multiple of 32 (GroupNorm32)");
    MAUA_REQUIRE(c % num_head_channels == 0, "maua_unet_create: channels must divide into heads");
  }
  maua_unet* n = new maua_unet();
  n->ctx = ctx; n->image_size = image_size; n->in_ch = in_channels; n->mc = model_channels; n->out_ch = out_channels;
  n->nrb = num_res_blocks; n->head_ch = num_head_channels; n->dtype = dtype; n->esize = dtype == MAUA_BF16 ? 2 : 4;
  n->mult.assign(channel_mult, channel_mult + n_mult);
  n->attn_ds.assign(attention_ds, attention_ds + n_attn);
  n->emb_dim = 4 * model_channels;
  if (const char* e = getenv("MAUA_UNET_ROUTE")) n->route = atoi(e);
  if (int rc = build_structure(n)) {
    maua_unet_destroy(n);
    return rc;
  }
  *out = n;
  return MAUA_OK;
}

void maua_unet_destroy(maua_unet* n) {
  if (!n) return;
  hipStreamSynchronize(n->ctx->stream);
  if (n->graph_exec) hipGraphExecDestroy(n->graph_exec);
  if (n->gd_exec) hipGraphExecDestroy(n->gd_exec);
  if (n->cap_stream) hipStreamDestroy(n->cap_stream);
  if (n->side_stream) hipStreamDestroy(n->side_stream);
  if (n->cap_side) hipStreamDestroy(n->cap_side);
  if (n->ev_fork) hipEventDestroy(n->ev_fork);
  if (n->ev_join) hipEventDestroy(n->ev_join);
  for (void* p : {(void*)n->gd_buf, (void*)n->gd_tab, (void*)n->gd_flag, (void*)n->gd_rects, (void*)n->gd_gflag})
    if (p) hipFree(p);
  for (void* p : n->owned) hipFree(p);
  if (n->arena.base) hipFree(n->arena.base);
  if (n->ones) hipFree(n->ones);
  for (float* p : {n->g_out, n->g_pred, n->g_t, n->g_cf, n->emb_table})  // (g_x is the caller's tensor the graph was captured on)
    if (p) hipFree(p);
  delete n;
}

// "route": 0 = per-shape routing of the 3x3 convolutions (LDS-direct kernel / split-K gather GEMM / generic kernel),
// 1 = every convolution on the generic kernel, 2 = never the gather GEMM (parity tests compare the routes)
int maua_unet_set_option(maua_unet* n, const char* key, int value) {
  MAUA_REQUIRE(n && key, "maua_unet_set_option: NULL argument");
  if (!strcmp(key, "psum_off")) {
    n->psum_off = value;
    n->planned_key = 0;   // (the arena layout changes)
    drop_sampler_graphs(n);
    return MAUA_OK;
  }
  if (!strcmp(key, "route")) {
    n->route = value;
    drop_sampler_graphs(n);
    return MAUA_OK;
  }
  if (!strcmp(key, "guided_fork")) {   // 1 (default): the guidance branch of maua_ddim_guided_loop beside the UNet forward; 0: behind it
    n->gd_fork = value ? 1 : 0;
    drop_sampler_graphs(n);
    return MAUA_OK;
  }
  if (!strcmp(key, "vjp")) {   // 1: maua_unet_load also prepares the transposed weights maua_unet_vjp convolves / multiplies with
    MAUA_REQUIRE(value == 0 || value == 1, "maua_unet_set_option: vjp is 0 or 1");
    if (value && !n->zero_bias) {
      int mp = 32;
      for (auto& r : n->res) mp = std::max(mp, std::max(r.c1.Cip, r.c1.Cop));
      if (int rc = dev_alloc(n, &n->zero_bias, (size_t)mp * 4)) return rc;
    }
    n->vjp = value;
    return MAUA_OK;
  }
  return fail(std::string("maua_unet_set_option: unknown option ") + key);
}

// 1 when the last maua_ddim_sample_loop(use_graph = 1) replayed a captured hipGraph, 0 when it ran eagerly
int maua_unet_graph_active(maua_unet* n, int* active) {
  MAUA_REQUIRE(n && active, "maua_unet_graph_active: NULL argument");
  *active = n->graph_exec && !n->graph_failed ? 1 : 0;
  return MAUA_OK;
}

int maua_unet_param_count(maua_unet* n, long* count) {
  MAUA_REQUIRE(n && count, "maua_unet_param_count: NULL argument");
  *count = (long)n->params.size();
  return MAUA_OK;
}

// name: a key of guided-diffusion's UNetModel state dict; 3x3 conv weights [Co][Ci][3][3], 1x1 (conv1d / conv2d) weights
// [N][K][1(,1)], linear weights [N][K], GroupNorm weight / bias [C]; f32 on the host.
int maua_unet_load(maua_unet* n, const char* name, const float* host, size_t count) {
  MAUA_REQUIRE(n && name && host, "maua_unet_load: NULL argument");
  auto it = n->params.find(name);
  if (it == n->params.end()) return fail(std::string("maua_unet_load: unknown parameter name: ") + name);
  const PRef& p = it->second;
  hipStream_t st = n->ctx->stream;
  auto wrong = [&]() { return fail(std::string("maua_unet_load: ") + name + ": wrong size"); };
  switch (p.kind) {
    case P_GN_G: case P_GN_B: {
      UGN* g = (UGN*)p.obj;
      if (count != (size_t)g->C) return wrong();
      MAUA_HIP_CHECK(hipMemcpy(p.kind == P_GN_G ? g->gamma : g->beta, host, count * 4, hipMemcpyHostToDevice));
      return MAUA_OK;
    }
    case P_CONV_B: {
      UConv* c = (UConv*)p.obj;
      if (count != (size_t)c->Co) return wrong();
      MAUA_HIP_CHECK(hipMemcpy(c->bias, host, count * 4, hipMemcpyHostToDevice));
      return MAUA_OK;
    }
    case P_LIN_B: {
      ULin* l = (ULin*)p.obj;
      if (count != (size_t)l->N) return wrong();
      MAUA_HIP_CHECK(hipMemcpy(l->bias, host, count * 4, hipMemcpyHostToDevice));
      return MAUA_OK;
    }
    case P_F32: {
      if (count != p.count) return wrong();
      MAUA_HIP_CHECK(hipMemcpy(p.f32, host, count * 4, hipMemcpyHostToDevice));
      if (p.f32 == n->freqs) n->freqs_loaded = 1;
      return MAUA_OK;
    }
    case P_CONV_W: {
      UConv* c = (UConv*)p.obj;
      if (count != (size_t)c->Co * c->Ci * 9) return wrong();
      float* tmp;
      MAUA_HIP_CHECK(hipMalloc((void**)&tmp, count * 4));
      MAUA_HIP_CHECK(hipMemcpy(tmp, host, count * 4, hipMemcpyHostToDevice));
      MAUA_HIP_CHECK(hipMemsetAsync(c->wt, 0, (size_t)9 * c->Cop * c->Cip * n->esize, st));
      int rc = launch_prep_weights(st, n->dtype, tmp, c->wt, nullptr, c->Co, c->Ci, 3, 1, 0, c->Cop, c->Cip);
      if (!rc && n->vjp) {
        // the input-gradient convolution: Wt[ci][co][ky][kx] = W[co][ci][2 - ky][2 - kx]
        float* tt = nullptr;
        if (hipMalloc((void**)&tt, count * 4) != hipSuccess) { hipFree(tmp); return fail("maua_unet_load: out of device memory"); }
        if (!c->wt_t) rc = dev_alloc(n, &c->wt_t, (size_t)9 * c->Cop * c->Cip * n->esize);
        if (!rc) {
          hipLaunchKernelGGL(transpose_flip_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, tmp, tt, c->Co, c->Ci, 9);
          hipMemsetAsync(c->wt_t, 0, (size_t)9 * c->Cop * c->Cip * n->esize, st);
          rc = launch_prep_weights(st, n->dtype, tt, c->wt_t, nullptr, c->Ci, c->Co, 3, 1, 0, c->Cip, c->Cop);
        }
        hipStreamSynchronize(st);
        hipFree(tt);
      }
      hipStreamSynchronize(st);
      hipFree(tmp);
      return rc;
    }
    case P_LIN_W: {
      ULin* l = (ULin*)p.obj;
      if (count != (size_t)l->N * l->K) return wrong();
      float* tmp;
      MAUA_HIP_CHECK(hipMalloc((void**)&tmp, count * 4));
      MAUA_HIP_CHECK(hipMemcpy(tmp, host, count * 4, hipMemcpyHostToDevice));
      // a k = 1 "convolution": [N][K] rows, converted to the network dtype
      int rc = launch_prep_weights(st, n->dtype, tmp, l->w, nullptr, l->N, l->K, 1, 1, 0, l->N, l->K);
      if (!rc && n->vjp) {
        float* tt = nullptr;
        if (hipMalloc((void**)&tt, count * 4) != hipSuccess) { hipFree(tmp); return fail("maua_unet_load: out of device memory"); }
        if (!l->w_t) rc = dev_alloc(n, &l->w_t, (size_t)l->N * l->K * n->esize);
        if (!rc) {
          hipLaunchKernelGGL(transpose_flip_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, tmp, tt, l->N, l->K, 1);
          rc = launch_prep_weights(st, n->dtype, tt, l->w_t, nullptr, l->K, l->N, 1, 1, 0, l->K, l->N);
        }
        hipStreamSynchronize(st);
        hipFree(tt);
      }
      hipStreamSynchronize(st);
      hipFree(tmp);
      return rc;
    }
  }
  return MAUA_ERR;
}

// x: device f32 [B][in_channels][H][W]; timesteps: device f32 [B] (what the wrapped model passes: the ORIGINAL timestep,
// rescaled to 0..1000); out: device f32 [B][out_channels][H][W].  H, W: multiples of 2^(levels - 1).
int maua_unet_forward(maua_unet* n, const float* x, const float* timesteps, int B, int H, int W, float* out) {
  MAUA_REQUIRE(n && x && timesteps && out, "maua_unet_forward: NULL argument");
  MAUA_REQUIRE(B >= 0 && H > 0 && W > 0, "maua_unet_forward: bad shape");
  const int down = 1 << (n->mult.size() - 1);
  MAUA_REQUIRE(H % down == 0 && W % down == 0, "maua_unet_forward: H and W must be multiples of 2^(levels - 1)");
  if (B == 0) return MAUA_OK;
  return n->dtype == MAUA_BF16 ? run_forward<bf16_t>(n, x, timesteps, B, H, W, out) : run_forward<float>(n, x, timesteps, B, H, W, out);
}

// The forward again, keeping what its input gradient needs (GroupNorm inputs + statistics, qkv + the attention rows' log-sum-exp)
// on the arena; then maua_unet_vjp(g_out [B][out_channels][H][W]) -> g_x [B][in_channels][H][W] = (d out / d x)^T g_out, the
// vector-Jacobian product guided.py:258-272 asks autograd for (speed "regular": img = f(pred_xstart(UNet(x)))).  The pair must
// run back to back on one network (any other forward in between invalidates the kept tensors); option "vjp" = 1 before the
// weights are loaded.  Every 3x3 convolution's gradient is the same MFMA convolution with the transposed, flipped kernel.
int maua_unet_forward_keep(maua_unet* n, const float* x, const float* timesteps, int B, int H, int W, float* out) {
  MAUA_REQUIRE(n && x && timesteps && out, "maua_unet_forward_keep: NULL argument");
  MAUA_REQUIRE(n->vjp, "maua_unet_forward_keep: set option \"vjp\" = 1 before loading the weights");
  MAUA_REQUIRE(n->conv_in.wt_t && n->conv_out.wt_t, "maua_unet_forward_keep: the weights were loaded before option \"vjp\" was set");
  MAUA_REQUIRE(B > 0 && H > 0 && W > 0, "maua_unet_forward_keep: bad shape");
  const int down = 1 << (n->mult.size() - 1);
  MAUA_REQUIRE(H % down == 0 && W % down == 0, "maua_unet_forward_keep: H and W must be multiples of 2^(levels - 1)");
  return n->dtype == MAUA_BF16 ? run_forward<bf16_t>(n, x, timesteps, B, H, W, out, true) : run_forward<float>(n, x, timesteps, B, H, W, out, true);
}

int maua_unet_vjp(maua_unet* n, const float* g_out, int B, int H, int W, float* g_x) {
  MAUA_REQUIRE(n && g_out && g_x, "maua_unet_vjp: NULL argument");
  MAUA_REQUIRE(n->tape_valid && n->tape_B == B && n->tape_H == H && n->tape_W == W,
               "maua_unet_vjp: no kept forward of this shape (call maua_unet_forward_keep first, nothing in between)");
  return n->dtype == MAUA_BF16 ? run_vjp<bf16_t>(n, g_out, g_x) : run_vjp<float>(n, g_out, g_x);
}

// One DDIM update (gaussian_diffusion.py ddim_sample, epsilon model, clip_denoised False).  x [B][C][H][W], model_out
// [B][Cm][H][W] (Cm >= C: the learned-variance channels are not used by DDIM), cond_grad = cond_fn(x, t) or NULL, noise or
// NULL (eta = 0), coef: device f32 [B][8] = {sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod,
// sqrt(1 - alphas_cumprod), sqrt(alphas_cumprod_prev), sqrt(1 - alphas_cumprod_prev - sigma^2), sigma * (t != 0), 0, 0}.
int maua_ddim_step(maua_ctx* ctx, const float* x, const float* model_out, const float* cond_grad, const float* noise,
                   const float* coef, int B, int C, int Cm, long HW, float* sample, float* pred_xstart) {
  MAUA_REQUIRE(ctx, "maua_ddim_step: ctx is NULL");
  if (B == 0 || HW == 0) return MAUA_OK;
  MAUA_REQUIRE(x && model_out && coef && sample && C > 0 && Cm >= C, "maua_ddim_step: NULL argument");
  const long total = (long)B * C * HW;
  hipLaunchKernelGGL(ddim_step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, x, model_out,
                     cond_grad, noise, coef, C, Cm, HW, total, sample, pred_xstart);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// One p_sample update (gaussian_diffusion.py p_sample + p_mean_variance with learned-range variance, epsilon model,
// clip_denoised False; cond_grad = cond_fn(x, t) or NULL: condition_mean).  model_out [B][2 C][H][W]; noise [B][C][H][W];
// coef: device f32 [B][8] = {sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod, posterior_mean_coef1, posterior_mean_coef2,
// posterior_log_variance_clipped, log(betas), t != 0, 0}.
int maua_p_sample_step(maua_ctx* ctx, const float* x, const float* model_out, const float* cond_grad, const float* noise,
                       const float* coef, int B, int C, long HW, float* sample, float* pred_xstart) {
  MAUA_REQUIRE(ctx, "maua_p_sample_step: ctx is NULL");
  if (B == 0 || HW == 0) return MAUA_OK;
  MAUA_REQUIRE(x && model_out && noise && coef && sample && C > 0, "maua_p_sample_step: NULL argument");
  const long total = (long)B * C * HW;
  hipLaunchKernelGGL(p_sample_step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, x, model_out,
                     cond_grad, noise, coef, C, HW, total, sample, pred_xstart);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// plms_sample's get_model_output: eps (after the optional condition_score), its pred_xstart and the unconditioned one.
// coef: maua_ddim_step's table.
int maua_plms_eps(maua_ctx* ctx, const float* x, const float* model_out, const float* cond_grad, const float* coef, int B,
                  int C, int Cm, long HW, float* eps, float* pred_xstart, float* pred_xstart_orig) {
  MAUA_REQUIRE(ctx, "maua_plms_eps: ctx is NULL");
  if (B == 0 || HW == 0) return MAUA_OK;
  MAUA_REQUIRE(x && model_out && coef && eps && C > 0 && Cm >= C, "maua_plms_eps: NULL argument");
  const long total = (long)B * C * HW;
  hipLaunchKernelGGL(plms_eps_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, x, model_out,
                     cond_grad, coef, C, Cm, HW, total, eps, pred_xstart, pred_xstart_orig);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// plms_sample's update from n_eps (1..4) epsilon tensors: eps' = (sum_i weights[i] eps_list[i]) / divisor, accumulated left to
// right like the reference's expressions ((3 e1 - e2) / 2, (23 e1 - 16 e2 + 5 e3) / 12, ...); coef: device f32 [B][8] = {sqrt_recip_ac, sqrt_recipm1_ac, sqrt(ac_prev),
// sqrt(1 - ac_prev), t != 0, 0, 0, 0}.
int maua_plms_update(maua_ctx* ctx, const float* x, const float* const* eps_list, const float* weights, int n_eps,
                     float divisor, const float* pred_xstart, const float* coef, int B, long chw, float* sample) {
  MAUA_REQUIRE(ctx, "maua_plms_update: ctx is NULL");
  if (B == 0 || chw == 0) return MAUA_OK;
  MAUA_REQUIRE(x && eps_list && weights && pred_xstart && coef && sample && n_eps >= 1 && n_eps <= 4, "maua_plms_update: bad argument");
  MAUA_REQUIRE(divisor != 0.f, "maua_plms_update: divisor is zero");
  PlmsEps pe{};
  pe.n = n_eps; pe.div = divisor;
  for (int i = 0; i < n_eps; i++) {
    MAUA_REQUIRE(eps_list[i], "maua_plms_update: NULL epsilon tensor");
    pe.e[i] = eps_list[i]; pe.w[i] = weights[i];
  }
  const long total = (long)B * chw;
  hipLaunchKernelGGL(plms_update_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, x, pe, pred_xstart,
                     coef, chw, total, sample);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// out[b] = ab[b][0] * x[b] + ab[b][1] * y[b] over rows of `row` elements (q_sample, gaussian_diffusion.py)
int maua_axpby_rows(maua_ctx* ctx, const float* x, const float* y, const float* ab, int B, long row, float* out) {
  MAUA_REQUIRE(ctx, "maua_axpby_rows: ctx is NULL");
  if (B == 0 || row == 0) return MAUA_OK;
  MAUA_REQUIRE(x && y && ab && out, "maua_axpby_rows: NULL argument");
  const long total = (long)B * row;
  hipLaunchKernelGGL(axpby_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, x, y, ab, row,
                     total, out);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// ---- operator-level entry points of the UNet's building blocks (NHWC tensors in the network dtype) --------------------
// QKVAttentionLegacy.forward: qkv [B][T][3 * heads * head_ch] (channel = head * 3 ch + {q, k, v} * ch + c) -> out [B][T][heads * ch]
int maua_attention_legacy(maua_ctx* ctx, const void* qkv, void* out, int B, int T, int heads, int head_ch, int dtype) {
  MAUA_REQUIRE(ctx, "maua_attention_legacy: ctx is NULL");
  AttnArgs a{};
  a.qkv = qkv; a.out = out; a.B = B; a.T = T; a.heads = heads; a.D = head_ch; a.ld_qkv = 3L * heads * head_ch;
  a.ld_out = (long)heads * head_ch; a.scale = 1.f / sqrtf((float)head_ch);
  return launch_attention(ctx->stream, dtype, a);
}

// conv_nd(1, K, N, 1) / nn.Linear on rows: c[M][N] = a[M][K] x w[N][K]^T + bias (+ res[M][N]); a, w, res, c in dtype
int maua_linear_nt(maua_ctx* ctx, const void* a, const void* w, const float* bias, const void* res, void* c, long M, int N,
                   int K, int dtype) {
  MAUA_REQUIRE(ctx, "maua_linear_nt: ctx is NULL");
  GemmArgs g{};
  g.a0 = a; g.lda0 = K; g.K0 = K; g.w = w; g.bias = bias; g.res = res; g.ldr = N; g.c = c; g.ldc = N; g.M = M; g.N = N;
  g.prefer_dma = ctx->linear_dma;   // (option "linear_dma": large shapes on gemm_dma.hip - what the CLIP tower's projections run on)
  return launch_gemm_nt(ctx->stream, dtype, g);
}

// GroupNorm32(32, C) (+ optional per-sample scale-shift [B][2C]: y * (1 + scale) + shift) (+ SiLU) on NHWC x -> y
int maua_group_norm_nhwc(maua_ctx* ctx, const void* x, const float* gamma, const float* beta, const float* scale_shift,
                         int silu, int B, int H, int W, int C, int dtype, void* y) {
  MAUA_REQUIRE(ctx && x && gamma && beta && y, "maua_group_norm_nhwc: NULL argument");
  MAUA_REQUIRE(C % 32 == 0 && (dtype == MAUA_F32 || dtype == MAUA_BF16), "maua_group_norm_nhwc: C % 32, f32 / bf16");
  if (B == 0) return MAUA_OK;
  const size_t part_bytes = gn_part_bytes(B, C, (long)H * W, dtype == MAUA_BF16 ? 2 : 4);
  if (int rc = scratch_reserve(ctx, part_bytes + (size_t)B * 64 * 4 + 512)) return rc;
  double* part = (double*)ctx->scratch;
  float* stats = (float*)((char*)ctx->scratch + ((part_bytes + 255) & ~(size_t)255));
  if (dtype == MAUA_BF16)
    return launch_group_norm<bf16_t>(ctx->stream, (const bf16_t*)x, C, nullptr, 0, B, H, W, gamma, beta, scale_shift, 2L * C,
                                     silu, 0, (bf16_t*)y, nullptr, part, stats);
  return launch_group_norm<float>(ctx->stream, (const float*)x, C, nullptr, 0, B, H, W, gamma, beta, scale_shift, 2L * C, silu,
                                  0, (float*)y, nullptr, part, stats);
}

// Input gradients of the two operator-level blocks above (what maua_unet_vjp walks a network with).  Attention: qkv as the
// forward's, d_out [B][T][C] -> d_qkv [B][T][3C] (the forward is re-run here for the rows' log-sum-exp and the result).
int maua_attention_legacy_vjp(maua_ctx* ctx, const void* qkv, const void* d_out, void* d_qkv, int B, int T, int heads, int head_ch,
                              int dtype) {
  MAUA_REQUIRE(ctx && qkv && d_out && d_qkv, "maua_attention_legacy_vjp: NULL argument");
  MAUA_REQUIRE(dtype == MAUA_F32 || dtype == MAUA_BF16, "maua_attention_legacy_vjp: f32 / bf16");
  if (B == 0) return MAUA_OK;
  const size_t esize = dtype == MAUA_BF16 ? 2 : 4;
  const size_t out_bytes = ((size_t)B * T * heads * head_ch * esize + 255) & ~(size_t)255, row_bytes = ((size_t)B * heads * T * 4 + 255) & ~(size_t)255;
  if (int rc = scratch_reserve(ctx, out_bytes + 2 * row_bytes + 512)) return rc;
  char* ws = (char*)ctx->scratch;
  AttnArgs f{};
  f.qkv = qkv; f.out = ws; f.B = B; f.T = T; f.heads = heads; f.D = head_ch; f.ld_qkv = 3L * heads * head_ch;
  f.ld_out = (long)heads * head_ch; f.scale = 1.f / sqrtf((float)head_ch); f.lse = (float*)(ws + out_bytes);
  if (int rc = launch_attention(ctx->stream, dtype, f)) return rc;
  AttnVjpArgs a{};
  a.qkv = qkv; a.out = ws; a.d_out = d_out; a.lse = f.lse; a.d_qkv = d_qkv; a.delta = (float*)(ws + out_bytes + row_bytes); a.B = B; a.T = T;
  a.heads = heads; a.D = head_ch; a.ld_qkv = f.ld_qkv; a.ld_out = f.ld_out; a.scale = f.scale;
  return launch_attention_vjp(ctx->stream, dtype, a);
}

// GroupNorm32 (+ scale-shift) (+ SiLU) (+ resample: 0 none, 1 2x2 average, 2 nearest x2 - behind the activation, as the
// ResBlocks' h_upd): dy [B][Ho][Wo][C] -> dx [B][H][W][C]; dres (optional, like dy): the gradient of the resampled raw x, added.
int maua_group_norm_nhwc_vjp(maua_ctx* ctx, const void* x, const float* gamma, const float* beta, const float* scale_shift, int silu,
                             int resample, const void* dy, const void* dres, int B, int H, int W, int C, int dtype, void* dx) {
  MAUA_REQUIRE(ctx && x && gamma && beta && dy && dx, "maua_group_norm_nhwc_vjp: NULL argument");
  MAUA_REQUIRE(C % 32 == 0 && (dtype == MAUA_F32 || dtype == MAUA_BF16), "maua_group_norm_nhwc_vjp: C % 32, f32 / bf16");
  if (B == 0) return MAUA_OK;
  const int esize = dtype == MAUA_BF16 ? 2 : 4;
  const size_t part_bytes = (gn_part_bytes(B, C, (long)H * W, esize) + 255) & ~(size_t)255;
  const size_t vjp_bytes = group_norm_vjp_workspace(B, C, (long)H * W, esize);
  if (int rc = scratch_reserve(ctx, part_bytes + (size_t)B * 64 * 4 + 512 + vjp_bytes)) return rc;
  double* part = (double*)ctx->scratch;
  float* stats = (float*)((char*)ctx->scratch + part_bytes);
  void* ws = (char*)ctx->scratch + part_bytes + (size_t)B * 64 * 4 + 256;
  // the forward's statistics (its output goes nowhere: the statistics passes only)
  if (int rc = dtype == MAUA_BF16 ? launch_group_norm_stats<bf16_t>(ctx->stream, (const bf16_t*)x, C, B, H, W, part, stats)
                                  : launch_group_norm_stats<float>(ctx->stream, (const float*)x, C, B, H, W, part, stats))
    return rc;
  GnVjpArgs a{};
  a.x0 = x; a.C0 = C; a.stats = stats; a.gamma = gamma; a.beta = beta; a.ss = scale_shift; a.ss_ld = 2L * C; a.silu = silu; a.mode = resample;
  a.dy = dy; a.dres = dres; a.dx0 = dx; a.B = B; a.H = H; a.W = W;
  return launch_group_norm_vjp(ctx->stream, dtype, a, ws);
}

// The unconditioned sampler loop inside the library: n_steps x (UNet forward + DDIM update), x updated in place.
// model_t: host f32 [n_steps] (the timestep the network sees at each step, same for every sample); coef: host f32
// [n_steps][8] (maua_ddim_step's coefficients).  use_graph: capture the whole loop in ONE hipGraph on first use for a
// (B, H, W, n_steps) and replay it afterwards (a forward is ~400 short launches: the graph removes the launch gaps).
// pred_xstart (optional) receives the last step's prediction.
// what both sampler loops need on the device before their first step: the timesteps [n_steps][B], the DDIM coefficients
// [n_steps][B][8], the model-output / pred_xstart buffers and every step's timestep projections (emb_table)
static int prepare_sampler(maua_unet* n, int B, int H, int W, const float* model_t, const float* coef, int n_steps) {
  hipStream_t st = n->ctx->stream;
  const long chw = (long)n->in_ch * H * W;
  // per-step constants on the device: timesteps [n_steps][B], coefficients [n_steps][B][8]
  if (n->g_steps < n_steps * B || !n->g_t) {
    MAUA_HIP_CHECK(hipStreamSynchronize(st));
    for (float** p : {&n->g_t, &n->g_cf}) { if (*p) hipFree(*p); *p = nullptr; }
    MAUA_HIP_CHECK(hipMalloc((void**)&n->g_t, (size_t)n_steps * B * 4));
    MAUA_HIP_CHECK(hipMalloc((void**)&n->g_cf, (size_t)n_steps * B * 8 * 4));
    n->g_steps = n_steps * B;
    drop_sampler_graphs(n);
  }
  {
    std::vector<float> ht((size_t)n_steps * B), hc((size_t)n_steps * B * 8);
    for (int s = 0; s < n_steps; s++)
      for (int b = 0; b < B; b++) {
        ht[(size_t)s * B + b] = model_t[s];
        memcpy(&hc[((size_t)s * B + b) * 8], coef + (size_t)s * 8, 32);
      }
    MAUA_HIP_CHECK(hipMemcpyAsync(n->g_t, ht.data(), ht.size() * 4, hipMemcpyHostToDevice, st));
    MAUA_HIP_CHECK(hipMemcpyAsync(n->g_cf, hc.data(), hc.size() * 4, hipMemcpyHostToDevice, st));
    MAUA_HIP_CHECK(hipStreamSynchronize(st));
  }
  const size_t out_bytes = (size_t)B * n->out_ch * H * W * 4, pred_bytes = (size_t)B * chw * 4;
  if (!n->g_out || n->out_cap < out_bytes + pred_bytes) {
    MAUA_HIP_CHECK(hipStreamSynchronize(st));
    for (float** p : {&n->g_out, &n->g_pred}) { if (*p) hipFree(*p); *p = nullptr; }
    MAUA_HIP_CHECK(hipMalloc((void**)&n->g_out, out_bytes));
    MAUA_HIP_CHECK(hipMalloc((void**)&n->g_pred, pred_bytes));
    n->out_cap = out_bytes + pred_bytes;
    drop_sampler_graphs(n);
  }
  // every step's timestep projections at once: the timesteps are known up front and shared by the samples, so the stacked
  // emb_layers GEMV (51 k x 1024 f32 weights at the 256^2 configuration) runs once per loop with n_steps rows instead of
  // once per step with B rows
  if (n->emb_table_rows < (size_t)n_steps) {
    MAUA_HIP_CHECK(hipStreamSynchronize(st));
    if (n->emb_table) hipFree(n->emb_table);
    n->emb_table = nullptr; n->emb_table_rows = 0;
    MAUA_HIP_CHECK(hipMalloc((void**)&n->emb_table, (size_t)n_steps * n->emb_total * 4));
    n->emb_table_rows = n_steps;
    drop_sampler_graphs(n);
  }
  {
    const int E = n->emb_dim, mc = n->mc, half = mc / 2;
    float *tt, *e0, *e1, *e2;
    MAUA_HIP_CHECK(hipMalloc((void**)&tt, (size_t)n_steps * (1 + mc + 2 * E) * 4));
    e0 = tt + n_steps; e1 = e0 + (size_t)n_steps * mc; e2 = e1 + (size_t)n_steps * E;
    MAUA_HIP_CHECK(hipMemcpyAsync(tt, model_t, (size_t)n_steps * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n_steps * half + 255) / 256), dim3(256), 0, st, tt,
                       n->freqs_loaded ? n->freqs : nullptr, e0, n_steps, mc);
    hipLaunchKernelGGL(linear_rows_kernel, dim3((E + 3) / 4), dim3(256), 0, st, e0, n->te0_w, n->te0_b, e1, n_steps, mc, E, 0, 1);
    hipLaunchKernelGGL(linear_rows_kernel, dim3((E + 3) / 4), dim3(256), 0, st, e1, n->te2_w, n->te2_b, e2, n_steps, E, E, 0, 0);
    hipLaunchKernelGGL(linear_rows_kernel, dim3((n->emb_total + 3) / 4), dim3(256), 0, st, e2, n->embw, n->embb, n->emb_table,
                       n_steps, E, n->emb_total, 1, 0);
    MAUA_HIP_CHECK(hipGetLastError());
    MAUA_HIP_CHECK(hipStreamSynchronize(st));
    hipFree(tt);
  }
  return MAUA_OK;
}

int maua_ddim_sample_loop(maua_unet* n, float* x, int B, int H, int W, const float* model_t, const float* coef, int n_steps,
                          int use_graph, float* pred_xstart) {
  MAUA_REQUIRE(n && x && model_t && coef && n_steps > 0, "maua_ddim_sample_loop: NULL argument");
  if (B == 0) return MAUA_OK;
  hipStream_t st = n->ctx->stream;
  const long chw = (long)n->in_ch * H * W;
  const size_t key = shape_key(B, H, W) ^ ((size_t)n_steps << 52);
  if (int rc = prepare_sampler(n, B, H, W, model_t, coef, n_steps)) return rc;
  const size_t pred_bytes = (size_t)B * chw * 4;
  auto body = [&](int s) -> int {
    n->emb_row = n->emb_table + (size_t)s * n->emb_total;
    int rc = maua_unet_forward(n, x, n->g_t + (size_t)s * B, B, H, W, n->g_out);
    n->emb_row = nullptr;
    if (rc) return rc;
    return maua_ddim_step(n->ctx, x, n->g_out, nullptr, nullptr, n->g_cf + (size_t)s * B * 8, B, n->in_ch, n->out_ch,
                          (long)H * W, x, n->g_pred);
  };
  if (use_graph && !n->graph_failed) {
    if (!n->graph_exec || n->graph_key != key || n->g_x != x) {
      // one eager forward first: plans the arena, sets the kernels' attributes (nothing of that is capturable)
      // - on a scratch copy so that x is not advanced
      if (n->planned_key != shape_key(B, H, W) || B > n->ones_b) {
        float* tmp;
        MAUA_HIP_CHECK(hipMalloc((void**)&tmp, pred_bytes));
        MAUA_HIP_CHECK(hipMemcpyAsync(tmp, x, pred_bytes, hipMemcpyDeviceToDevice, st));
        int rc = maua_unet_forward(n, tmp, n->g_t, B, H, W, n->g_out);
        hipStreamSynchronize(st);
        hipFree(tmp);
        if (rc) return rc;
      }
      if (n->graph_exec) { hipGraphExecDestroy(n->graph_exec); n->graph_exec = nullptr; }
      if (!n->cap_stream) MAUA_HIP_CHECK(hipStreamCreateWithFlags(&n->cap_stream, hipStreamNonBlocking));
      MAUA_HIP_CHECK(hipStreamSynchronize(st));
      hipGraph_t graph = nullptr;
      hipError_t e = hipStreamBeginCapture(n->cap_stream, hipStreamCaptureModeThreadLocal);
      int rc = MAUA_OK;
      if (e == hipSuccess) {
        n->ctx->stream = n->cap_stream;   // the launchers read the context's stream
        for (int s = 0; s < n_steps && !rc; s++) rc = body(s);
        n->ctx->stream = st;
        e = hipStreamEndCapture(n->cap_stream, &graph);
      }
      if (!rc && e == hipSuccess) e = hipGraphInstantiate(&n->graph_exec, graph, nullptr, nullptr, 0);
      if (graph) hipGraphDestroy(graph);
      if (rc || e != hipSuccess) {
        (void)hipGetLastError();  // clear the sticky error; run eagerly from now on
        n->graph_exec = nullptr;
        n->graph_failed = 1;
        if (getenv("MAUA_VERBOSE"))
          fprintf(stderr, "[maua] ddim_sample_loop: graph capture unavailable (%s), running eagerly\n",
                  rc ? maua_last_error() : hipGetErrorString(e));
      } else {
        n->graph_key = key;
        n->g_x = x;
      }
    }
  }
  if (use_graph && n->graph_exec && !n->graph_failed) {
    MAUA_HIP_CHECK(hipGraphLaunch(n->graph_exec, st));
  } else {
    for (int s = 0; s < n_steps; s++)
      if (int rc = body(s)) return rc;
  }
  if (pred_xstart) MAUA_HIP_CHECK(hipMemcpyAsync(pred_xstart, n->g_pred, pred_bytes, hipMemcpyDeviceToDevice, st));
  return MAUA_OK;
}

// g = (img - target) * k over rows, zeros when any element is NaN (MSEGuide + the NaN rule of guided.py:262-265); k: device scalar
// (reset_flag = false: *flag was zeroed by the caller - the captured sampler loop keeps memset nodes out of its graph: replays of a
//  graph holding a 4-byte memset node were seen reading a non-zero flag after an eager run of the same calls, on ROCm 7.0.2)
static int mse_guide_grad(maua_ctx* ctx, const float* img, const float* target, long tstride, const float* kdev, int B, long row,
                          float* out, int* flag, bool reset_flag = true) {
  const long total = (long)B * row;
  if (reset_flag) MAUA_HIP_CHECK(hipMemsetAsync(flag, 0, 4, ctx->stream));
  hipLaunchKernelGGL(mse_guide_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, img, target, tstride,
                     kdev, row, total, out, flag);
  hipLaunchKernelGGL(zero_if_flag_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, out, total, flag);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

// The image-MSE grad module as an operator: out = (img - target) * k, zeros if that holds a NaN.  img, out [B][row]; target [B][row]
// (target_bstride = row) or one row for all samples (0).
int maua_mse_guide_grad(maua_ctx* ctx, const float* img, const float* target, long target_bstride, float k, int B, long row, float* out) {
  MAUA_REQUIRE(ctx, "maua_mse_guide_grad: ctx is NULL");
  if (B == 0 || row == 0) return MAUA_OK;
  MAUA_REQUIRE(img && target && out, "maua_mse_guide_grad: NULL argument");
  if (int rc = scratch_reserve(ctx, 256 + (size_t)B * 4)) return rc;
  int* flag = reinterpret_cast<int*>(ctx->scratch);
  float* kd = reinterpret_cast<float*>(reinterpret_cast<char*>(ctx->scratch) + 256);
  std::vector<float> hk((size_t)B, k);
  MAUA_HIP_CHECK(hipMemcpyAsync(kd, hk.data(), (size_t)B * 4, hipMemcpyHostToDevice, ctx->stream));
  MAUA_HIP_CHECK(hipStreamSynchronize(ctx->stream));   // (hk lives on this call's stack)
  return mse_guide_grad(ctx, img, target, target_bstride, kd, B, row, out, flag);
}

// configs[3] as BASELINE states it: the GUIDED DDIM loop inside the library, one hipGraph per shape.  Per step s (guided.py:302-311,
// 333-337 with cond_fn = GradientGuidedConditioning, speed "fast", :236-272, and the image-MSE grad module):
//   out  = unet(x, t_s)                                      pred = secondary(x, cos_t_s).pred          (:252-253)
//   img  = sigma_s pred + (1 - sigma_s) x                    g    = (img - target) k, zeros if NaN      (:254, :256-265)
//   grad = c0_s g + c1_s (dv/dx)^T g                         x, pred_xstart = ddim_step(x, out, grad)   (:266-268, ddim_sample)
// guide: host f32 [n_steps][5] = {cos_t, sigma, 1 - sigma, -(sigma a_c + 1 - sigma), sigma s_c} as GradientGuidedConditioning.forward
// evaluates them.  target: device [B][C][H][W] (target_bstride = C H W) or one image for all samples (0).  x is updated in place.
// sec == NULL: speed "regular" - out = forward_keep(x, t_s); eps = out[:, :C]; pred = ra x - rm eps; img, g as above;
// grad = c0_s g + c1_s (d eps / d x)^T g from maua_unet_vjp's walk; guide[s] = {-, sigma, 1 - sigma, -(sigma ra + 1 - sigma), sigma rm}.
int maua_ddim_guided_loop(maua_unet* n, maua_secondary* sec, float* x, int B, int H, int W, const float* model_t, const float* coef,
                          const float* guide, int n_steps, const float* target, long target_bstride, float mse_k, int use_graph,
                          float* pred_xstart) {
  MAUA_REQUIRE(n && x && model_t && coef && guide && n_steps > 0, "maua_ddim_guided_loop: NULL argument");
  // sec == NULL: speed "regular" (guided.py:214-218, 250-252) - the gradient goes through THIS network: a kept forward + its input
  // gradient per step, no secondary model
  const bool regular = sec == nullptr;
  if (regular) {
    MAUA_REQUIRE(n->vjp && n->conv_in.wt_t && n->conv_out.wt_t, "maua_ddim_guided_loop: speed \"regular\" needs option \"vjp\" = 1 before the weights are loaded");
    MAUA_REQUIRE(n->out_ch >= n->in_ch, "maua_ddim_guided_loop: the model output must hold an epsilon per image channel");
  } else {
    MAUA_REQUIRE(n->in_ch == 3, "maua_ddim_guided_loop: the secondary model guides 3-channel images");
    MAUA_REQUIRE(secondary_ctx(sec) == n->ctx, "maua_ddim_guided_loop: both networks must live on one context (one stream)");
  }
  const long chw = (long)n->in_ch * H * W;
  MAUA_REQUIRE(target_bstride == 0 || target_bstride == chw, "maua_ddim_guided_loop: target_bstride is 0 or C * H * W");
  if (B == 0) return MAUA_OK;
  hipStream_t st = n->ctx->stream;
  const size_t key = shape_key(B, H, W) ^ ((size_t)n_steps << 52) ^ ((size_t)(uintptr_t)sec << 1) ^ (target_bstride ? 1u : 0u) ^ (regular ? 2u : 0u);
  maua_clip* const clip = n->gd_clip;
  if (clip) {
    MAUA_REQUIRE(clip_ctx(clip) == n->ctx, "maua_ddim_guided_loop: the image tower must live on the networks' context");
    MAUA_REQUIRE(n->gd_rect_steps == n_steps, "maua_ddim_guided_loop: maua_unet_set_clip_guide was given another number of steps");
    MAUA_REQUIRE(n->in_ch == 3, "maua_ddim_guided_loop: CLIP guides 3-channel images");
    for (size_t i = 0; i < n->gd_rects_host.size(); i += 3)
      MAUA_REQUIRE((n->gd_rects_host[i] & CUT_SIZE_MASK) > 0 && n->gd_rects_host[i + 1] >= 0 && n->gd_rects_host[i + 2] >= 0 &&
                       n->gd_rects_host[i + 1] + (n->gd_rects_host[i] & CUT_SIZE_MASK) <= H &&
                       n->gd_rects_host[i + 2] + (n->gd_rects_host[i] & CUT_SIZE_MASK) <= W,
                   "maua_ddim_guided_loop: a cutout leaves the image");
    if (int rc = clip_prepare_guide(clip, B, H, W, clip_group_size(clip, B, n->gd_cutn))) return rc;
  } else if (n->gd_guides.empty()) {
    MAUA_REQUIRE(target, "maua_ddim_guided_loop: target is NULL");
  }
  const bool guides = !n->gd_guides.empty();
  if (guides) {
    MAUA_REQUIRE(n->in_ch == 3, "maua_ddim_guided_loop: the grad modules guide 3-channel images");
    for (maua_guide* g : n->gd_guides) {
      MAUA_REQUIRE(guide_ctx(g) == n->ctx, "maua_ddim_guided_loop: a grad module lives on another context");
      if (int rc = guide_prepare(g, B, H, W)) return rc;
    }
  }
  if (int rc = prepare_sampler(n, B, H, W, model_t, coef, n_steps)) return rc;
  const size_t tb = (size_t)B * chw, tab = (size_t)n_steps * B * 7 + B;
  if (n->gd_cap < 10 * tb || n->gd_tab_cap < tab || !n->gd_flag || n->gd_flags < n_steps) {
    MAUA_HIP_CHECK(hipStreamSynchronize(st));
    for (void* p : {(void*)n->gd_buf, (void*)n->gd_tab, (void*)n->gd_flag})
      if (p) hipFree(p);
    n->gd_buf = nullptr; n->gd_tab = nullptr; n->gd_flag = nullptr; n->gd_cap = n->gd_tab_cap = 0;
    MAUA_HIP_CHECK(hipMalloc((void**)&n->gd_buf, 10 * tb * 4));
    MAUA_HIP_CHECK(hipMalloc((void**)&n->gd_tab, tab * 4));
    MAUA_HIP_CHECK(hipMalloc((void**)&n->gd_flag, (size_t)n_steps * 4));
    n->gd_cap = 10 * tb; n->gd_tab_cap = tab; n->gd_flags = n_steps;
    drop_sampler_graphs(n);
  }
  float *bx = n->gd_buf, *bv = bx + tb, *bp = bv + tb, *be = bp + tb, *bimg = be + tb, *bg = bimg + tb, *bjv = bg + tb,
        *bgrad = bjv + tb, *btgt = bgrad + tb, *bsub = btgt + tb;
  float *t_ct = n->gd_tab, *t_img = t_ct + (size_t)n_steps * B, *t_grad = t_img + (size_t)n_steps * B * 2,
        *t_pred = t_grad + (size_t)n_steps * B * 2, *t_k = t_pred + (size_t)n_steps * B * 2;
  {
    std::vector<float> h(tab);
    for (int s = 0; s < n_steps; s++)
      for (int b = 0; b < B; b++) {
        const float* gs = guide + (size_t)s * 5;
        h[(size_t)s * B + b] = gs[0];
        h[(size_t)n_steps * B + ((size_t)s * B + b) * 2] = gs[1];
        h[(size_t)n_steps * B + ((size_t)s * B + b) * 2 + 1] = gs[2];
        h[(size_t)n_steps * B * 3 + ((size_t)s * B + b) * 2] = gs[3];
        h[(size_t)n_steps * B * 3 + ((size_t)s * B + b) * 2 + 1] = gs[4];
        h[(size_t)n_steps * B * 5 + ((size_t)s * B + b) * 2] = coef[(size_t)s * 8];        // (ra, -rm): pred_xstart from eps (regular)
        h[(size_t)n_steps * B * 5 + ((size_t)s * B + b) * 2 + 1] = -coef[(size_t)s * 8 + 1];
      }
    for (int b = 0; b < B; b++) h[(size_t)n_steps * B * 7 + b] = mse_k;
    MAUA_HIP_CHECK(hipMemcpyAsync(n->gd_tab, h.data(), tab * 4, hipMemcpyHostToDevice, st));
    MAUA_HIP_CHECK(hipStreamSynchronize(st));
  }
  MAUA_HIP_CHECK(hipMemsetAsync(n->gd_flag, 0, (size_t)n_steps * 4, st));
  MAUA_HIP_CHECK(hipMemcpyAsync(bx, x, tb * 4, hipMemcpyDeviceToDevice, st));
  if (!clip && !guides) MAUA_HIP_CHECK(hipMemcpyAsync(btgt, target, (target_bstride ? tb : (size_t)chw) * 4, hipMemcpyDeviceToDevice, st));
  // the grad module(s) of step s on the context's current stream: bimg -> bg
  auto guide_grad = [&](int s) -> int {
    int rc = MAUA_OK;
    if (clip)
      rc = clip_guide_grad(clip, bimg, B, H, W, n->gd_rects + (size_t)s * n->gd_batches * n->gd_cutn * 3,
                           n->gd_mult ? n->gd_mult + (size_t)s * n->gd_batches * n->gd_cutn : nullptr, n->gd_cutn, n->gd_cutn_total,
                           n->gd_batches, n->gd_clip_scale, n->gd_clip_clamp, bg);
    else if (!guides)
      return mse_guide_grad(n->ctx, bimg, btgt, target_bstride ? chw : 0, t_k, B, chw, bg, n->gd_flag + s, false);
    // guided.py:258-266: img_grad += sub_grad per module, a module whose gradient holds a NaN skipped
    for (size_t k = 0; k < n->gd_guides.size() && !rc; k++) {
      rc = guide_eval(n->gd_guides[k], bimg, B, H, W, bsub);
      if (!rc) rc = screened_accumulate(n->ctx->stream, bsub, bg, (long)tb, !clip && k == 0, n->gd_gflag);
    }
    return rc;
  };
  const long tstride = target_bstride ? chw : 0;
  if (n->gd_fork && !n->ev_fork) {
    MAUA_HIP_CHECK(hipEventCreateWithFlags(&n->ev_fork, hipEventDisableTiming));
    MAUA_HIP_CHECK(hipEventCreateWithFlags(&n->ev_join, hipEventDisableTiming));
    MAUA_HIP_CHECK(hipStreamCreateWithFlags(&n->side_stream, hipStreamNonBlocking));
    MAUA_HIP_CHECK(hipStreamCreateWithFlags(&n->cap_side, hipStreamNonBlocking));
  }
  // one step on (main, side): the UNet forward on main; the guidance branch - it reads x and nothing the forward writes - on side
  // (main itself when the fork is off); the DDIM update on main behind both.  Every launcher reads the context's stream.
  // speed "regular": no parallel branch (the gradient needs the forward it differentiates); everything on `main`
  auto step_regular = [&](int s, hipStream_t main) -> int {
    n->ctx->stream = main;
    n->emb_row = n->emb_table + (size_t)s * n->emb_total;
    int rc = n->dtype == MAUA_BF16 ? run_forward<bf16_t>(n, bx, n->g_t + (size_t)s * B, B, H, W, n->g_out, true)
                                   : run_forward<float>(n, bx, n->g_t + (size_t)s * B, B, H, W, n->g_out, true);
    if (!rc) {
      // pred_xstart = ra x - rm eps (gaussian_diffusion.py _predict_xstart_from_eps), img = sigma pred + (1 - sigma) x (:252)
      const long total = (long)B * chw;
      hipLaunchKernelGGL(eps_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, main, n->g_out, chw, (long)n->out_ch * H * W,
                         total, be);
      if (hipGetLastError() != hipSuccess) rc = fail("maua_ddim_guided_loop: launch failed");
      if (!rc) rc = maua_axpby_rows(n->ctx, bx, be, t_pred + (size_t)s * B * 2, B, chw, bp);
      if (!rc) rc = maua_axpby_rows(n->ctx, bp, bx, t_img + (size_t)s * B * 2, B, chw, bimg);
    }
    if (!rc) rc = guide_grad(s);
    if (!rc) rc = n->dtype == MAUA_BF16 ? run_vjp<bf16_t>(n, bg, bjv, n->in_ch) : run_vjp<float>(n, bg, bjv, n->in_ch);
    n->emb_row = nullptr;
    if (!rc) rc = maua_axpby_rows(n->ctx, bg, bjv, t_grad + (size_t)s * B * 2, B, chw, bgrad);
    if (rc) return rc;
    return maua_ddim_step(n->ctx, bx, n->g_out, bgrad, nullptr, n->g_cf + (size_t)s * B * 8, B, n->in_ch, n->out_ch, (long)H * W, bx,
                          n->g_pred);
  };
  auto step_on = [&](int s, hipStream_t main, hipStream_t side) -> int {
    if (regular) return step_regular(s, main);
    const bool fork = side != main;
    if (fork) {
      MAUA_HIP_CHECK(hipEventRecord(n->ev_fork, main));
      MAUA_HIP_CHECK(hipStreamWaitEvent(side, n->ev_fork, 0));
    }
    n->ctx->stream = main;
    n->emb_row = n->emb_table + (size_t)s * n->emb_total;
    int rc = maua_unet_forward(n, bx, n->g_t + (size_t)s * B, B, H, W, n->g_out);
    n->emb_row = nullptr;
    if (!rc) {
      n->ctx->stream = side;
      rc = maua_secondary_forward(sec, bx, t_ct + (size_t)s * B, B, H, W, bv, bp, be);
      if (!rc) rc = maua_axpby_rows(n->ctx, bp, bx, t_img + (size_t)s * B * 2, B, chw, bimg);
      if (!rc) rc = guide_grad(s);
      if (!rc) rc = maua_secondary_vjp(sec, bg, B, H, W, bjv);
      if (!rc) rc = maua_axpby_rows(n->ctx, bg, bjv, t_grad + (size_t)s * B * 2, B, chw, bgrad);
    }
    n->ctx->stream = main;
    if (fork) {   // (joined whatever happened: a capture must not end with an unjoined stream)
      hipError_t e1 = hipEventRecord(n->ev_join, side), e2 = hipStreamWaitEvent(main, n->ev_join, 0);
      if (!rc && (e1 != hipSuccess || e2 != hipSuccess)) rc = fail("maua_ddim_guided_loop: joining the guidance branch failed");
    }
    if (rc) return rc;
    return maua_ddim_step(n->ctx, bx, n->g_out, bgrad, nullptr, n->g_cf + (size_t)s * B * 8, B, n->in_ch, n->out_ch, (long)H * W, bx,
                          n->g_pred);
  };
  auto body = [&](int s) -> int {   // eager: on the caller's stream (+ the side stream)
    int rc = step_on(s, st, n->gd_fork ? n->side_stream : st);
    n->ctx->stream = st;
    return rc;
  };
  // the captured graph holds raw pointers into the secondary model's weights and workspaces: it is this model's, at this generation
  // of its buffers, or it is recaptured (the address alone does not identify a model: a freed one's can be handed out again)
  unsigned long long sec_uid = 0, sec_epoch = 0;
  unsigned long long clip_uid = 0, clip_ep = 0;
  auto sec_matches = [&]() {
    secondary_stamp(sec, &sec_uid, &sec_epoch);
    clip_stamp(clip, &clip_uid, &clip_ep);
    bool guides_same = n->gd_guide_epochs.size() == n->gd_guides.size();
    for (size_t k = 0; guides_same && k < n->gd_guides.size(); k++) guides_same = guide_epoch(n->gd_guides[k]) == n->gd_guide_epochs[k];
    return sec_uid == n->gd_sec_uid && sec_epoch == n->gd_sec_epoch && clip_uid == n->gd_clip_uid && clip_ep == n->gd_clip_epoch &&
           n->gd_guide_gen == n->gd_guide_gen_seen && guides_same;
  };
  if (use_graph && !n->gd_failed) {
    if (!n->gd_exec || n->gd_key != key || !sec_matches()) {
      // one eager step on scratch copies first: it plans both networks' workspaces and sets the kernels' attributes (none of that
      // can be captured); bx is restored afterwards
      {
        int rc = body(0);
        if (rc) return rc;
        MAUA_HIP_CHECK(hipMemcpyAsync(bx, x, tb * 4, hipMemcpyDeviceToDevice, st));
        MAUA_HIP_CHECK(hipMemsetAsync(n->gd_flag, 0, (size_t)n_steps * 4, st));
      }
      if (n->gd_exec) { hipGraphExecDestroy(n->gd_exec); n->gd_exec = nullptr; }
      if (!n->cap_stream) MAUA_HIP_CHECK(hipStreamCreateWithFlags(&n->cap_stream, hipStreamNonBlocking));
      MAUA_HIP_CHECK(hipStreamSynchronize(st));
      hipGraph_t graph = nullptr;
      hipError_t e = hipStreamBeginCapture(n->cap_stream, hipStreamCaptureModeThreadLocal);
      int rc = MAUA_OK;
      if (e == hipSuccess) {
        for (int s = 0; s < n_steps && !rc; s++) rc = step_on(s, n->cap_stream, n->gd_fork ? n->cap_side : n->cap_stream);
        n->ctx->stream = st;
        e = hipStreamEndCapture(n->cap_stream, &graph);
      }
      if (!rc && e == hipSuccess) e = hipGraphInstantiate(&n->gd_exec, graph, nullptr, nullptr, 0);
      if (graph) hipGraphDestroy(graph);
      if (rc || e != hipSuccess) {
        (void)hipGetLastError();
        n->gd_exec = nullptr;
        n->gd_failed = 1;
        if (getenv("MAUA_VERBOSE"))
          fprintf(stderr, "[maua] ddim_guided_loop: graph capture unavailable (%s), running launch by launch\n",
                  rc ? maua_last_error() : hipGetErrorString(e));
      } else {
        n->gd_key = key;
        secondary_stamp(sec, &n->gd_sec_uid, &n->gd_sec_epoch);   // (after the eager step: that is what sized the workspaces)
        clip_stamp(clip, &n->gd_clip_uid, &n->gd_clip_epoch);
        n->gd_guide_gen_seen = n->gd_guide_gen;
        n->gd_guide_epochs.clear();
        for (maua_guide* g : n->gd_guides) n->gd_guide_epochs.push_back(guide_epoch(g));
      }
    }
  }
  if (use_graph && n->gd_exec && !n->gd_failed) {
    // (the copies / memset above are stream-ordered before the graph anyway; one host wait per 100-step loop costs nothing and keeps
    //  the replay independent of how the runtime orders copy engines against graph launches)
    MAUA_HIP_CHECK(hipStreamSynchronize(st));
    MAUA_HIP_CHECK(hipGraphLaunch(n->gd_exec, st));
    n->gd_last_graph = 1;
  } else {
    n->gd_last_graph = 0;
    for (int s = 0; s < n_steps; s++)
      if (int rc = body(s)) return rc;
  }
  MAUA_HIP_CHECK(hipMemcpyAsync(x, bx, tb * 4, hipMemcpyDeviceToDevice, st));
  if (pred_xstart) MAUA_HIP_CHECK(hipMemcpyAsync(pred_xstart, n->g_pred, tb * 4, hipMemcpyDeviceToDevice, st));
  return MAUA_OK;
}

// CLIPGrads as the guided loop's grad module (maua/grad.py:96-165 in place of the image-MSE module); rects: host [n_steps][batches][cutn][3]
int maua_unet_set_clip_guide(maua_unet* n, maua_clip* clip, const int* rects, const float* mult, int n_steps, int cutn, int batches,
                             float scale, float clamp_gradient) {
  MAUA_REQUIRE(n, "maua_unet_set_clip_guide: net is NULL");
  if (!clip) {
    if (n->gd_clip) n->gd_guide_gen++;
    n->gd_clip = nullptr;
    return MAUA_OK;
  }
  MAUA_REQUIRE(rects && n_steps > 0 && cutn > 0 && batches > 0, "maua_unet_set_clip_guide: bad arguments");
  hipStream_t st = n->ctx->stream;
  const size_t per = (size_t)n_steps * batches * cutn, cnt = per * 4;   // 3 ints + 1 float per cutout
  int cutn_total = cutn;
  if (mult) {
    for (size_t b = 0; b < (size_t)n_steps * batches; b++) {
      double t = 0;
      for (int i = 0; i < cutn; i++) t += mult[b * cutn + i];
      if (b == 0) cutn_total = (int)(t + 0.5);
      MAUA_REQUIRE((int)(t + 0.5) == cutn_total && cutn_total >= cutn,
                   "maua_unet_set_clip_guide: every cutout batch must stand for the same number (>= cutn) of cutouts");
    }
  }
  if (cnt > n->gd_rects_cap) {
    MAUA_HIP_CHECK(hipStreamSynchronize(st));
    if (n->gd_rects) hipFree(n->gd_rects);
    n->gd_rects = nullptr; n->gd_rects_cap = 0;
    MAUA_HIP_CHECK(hipMalloc((void**)&n->gd_rects, cnt * 4));
    n->gd_rects_cap = cnt;
    n->gd_guide_gen++;
  }
  float* mult_dev = mult ? (float*)(n->gd_rects + per * 3) : nullptr;
  // (everything a captured loop bakes into its launches moves the generation; the rectangles themselves are data it reads)
  if (n->gd_clip != clip || n->gd_rect_steps != n_steps || n->gd_cutn != cutn || n->gd_batches != batches || n->gd_clip_scale != scale ||
      n->gd_clip_clamp != clamp_gradient || n->gd_mult != mult_dev || n->gd_cutn_total != cutn_total)
    n->gd_guide_gen++;
  n->gd_rects_host.assign(rects, rects + per * 3);
  MAUA_HIP_CHECK(hipMemcpyAsync(n->gd_rects, n->gd_rects_host.data(), per * 12, hipMemcpyHostToDevice, st));
  if (mult) MAUA_HIP_CHECK(hipMemcpyAsync(mult_dev, mult, per * 4, hipMemcpyHostToDevice, st));
  MAUA_HIP_CHECK(hipStreamSynchronize(st));
  n->gd_clip = clip; n->gd_rect_steps = n_steps; n->gd_cutn = cutn; n->gd_batches = batches; n->gd_clip_scale = scale;
  n->gd_clip_clamp = clamp_gradient; n->gd_mult = mult_dev; n->gd_cutn_total = cutn_total;
  return MAUA_OK;
}

// a list of grad modules (guides.hip) as the guided loop's conditioning: evaluated after CLIPGrads (if set) and summed
int maua_unet_set_guides(maua_unet* n, maua_guide* const* guides, int n_guides) {
  MAUA_REQUIRE(n && n_guides >= 0 && (n_guides == 0 || guides), "maua_unet_set_guides: bad arguments");
  std::vector<unsigned long long> uids;
  for (int k = 0; k < n_guides; k++) {
    MAUA_REQUIRE(guides[k], "maua_unet_set_guides: NULL guide");
    uids.push_back(guide_uid(guides[k]));
  }
  if (uids != n->gd_guide_uids) n->gd_guide_gen++;   // (a captured loop bakes the list into its launches)
  n->gd_guide_uids = uids;
  n->gd_guides.assign(guides, guides + n_guides);
  if (n_guides && !n->gd_gflag) MAUA_HIP_CHECK(hipMalloc((void**)&n->gd_gflag, 256));
  return MAUA_OK;
}

// 1 when the last maua_ddim_guided_loop(use_graph = 1) replayed a captured hipGraph
int maua_unet_guided_graph_active(maua_unet* n, int* active) {
  MAUA_REQUIRE(n && active, "maua_unet_guided_graph_active: NULL argument");
  *active = n->gd_exec && !n->gd_failed && n->gd_last_graph ? 1 : 0;
  return MAUA_OK;
}

}  // extern "C"
