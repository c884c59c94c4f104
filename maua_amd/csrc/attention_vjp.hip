// Input gradient of the fused attention kernel (attention.hip) - a piece of guidance speed "regular" (guided.py:250-272: autograd
// through the UNet), guided_diffusion/unet.py QKVAttentionLegacy backwards:
//     P = softmax(S),  S = scale * Q K^T,  O = P V
//     dV = P^T dO,   dP = dO V^T,   dS = P o (dP - delta),  delta_i = sum_j P_ij dP_ij = dO_i . O_i
//     dQ = scale * dS K,   dK = scale * dS^T Q
// Like the forward, the T x T matrices never leave the registers: P is rebuilt from the scores and the row's log-sum-exp the
// forward left behind (AttnArgs.lse), delta comes from the forward's result.  Two launches of ONE kernel skeleton - the
// forward's: a lane owns a row ("own": a query for dQ, a key for dK / dV), the other side is walked in blocks of 32 rows staged
// in LDS as rows (the contraction over the head's channels: scores and dP) and transposed (the contraction over the walked rows:
// the gradients), and the accumulator of the first product is the operand of the second:
//     MODE 0 (dQ):      s^T = K Q^T, dp^T = V dO^T  -> ds^T -> dQ^T += K^T ds^T
//     MODE 1 (dK, dV):  s   = Q K^T, dp   = dO V^T  -> p, ds -> dV^T += dO^T p,  dK^T += Q^T ds
// bf16: v_mfma_f32_32x32x16_bf16 with P / dS rounded to bf16 (as every mixed-precision attention backward does); f32: exact products.
#include "common.h"
#include "internal.h"

namespace maua {

namespace {

template <typename T> struct VMma;
template <> struct VMma<bf16_t> {
  __device__ static __forceinline__ void step(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
};
template <> struct VMma<float> {
  __device__ static __forceinline__ void step(f32x16& acc, const u32x4& a, const u32x4& b) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], acc, 0, 0, 0);
  }
};

// delta[b][head][t] = d_out[b][t][head] . out[b][t][head].  One thread per 16-byte piece of a row (coalesced: a wave streams whole
// rows of both tensors), the pieces of a head - D * sizeof(T) / 16 = 4 ... 16 neighbouring lanes - summed by a shuffle tree (round 6: the
// one-thread-per-(row, head) form walked 128 bytes per thread and ran at 0.9 TB/s, 4 % of a text-guided step)
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnVjpArgs a) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const int pph = a.D / EPC;                 // pieces per head: a power of two <= 16 (D 32 / 64)
  const int ppr = a.heads * pph;             // ... per row
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)a.B * a.T * ppr;
  float s = 0.f;
  long bt = 0;
  int pc = 0;
  if (idx < total) {
    bt = idx / ppr;
    pc = (int)(idx - bt * ppr);
    const u32x4 ov = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(a.out) + bt * a.ld_out + (long)pc * EPC);
    const u32x4 gv = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(a.d_out) + bt * a.ld_out + (long)pc * EPC);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if constexpr (sizeof(T) == 2) {
        s = fmaf(bf2f((bf16_t)(ov[k] & 0xffffu)), bf2f((bf16_t)(gv[k] & 0xffffu)), s);
        s = fmaf(bf2f((bf16_t)(ov[k] >> 16)), bf2f((bf16_t)(gv[k] >> 16)), s);
      } else {
        s = fmaf(__uint_as_float(ov[k]), __uint_as_float(gv[k]), s);
      }
    }
  }
  // (a head's pieces never straddle a wave: 256 % pph == 0 and rows start at multiples of pph)
  for (int o = 1; o < pph; o <<= 1) s += __shfl_xor(s, o);
  if (idx < total && (pc & (pph - 1)) == 0) {
    const int head = pc / pph;
    const int t = (int)(bt % a.T);
    const long b = bt / a.T;
    a.delta[(b * a.heads + head) * a.T + t] = s;
  }
}

template <typename T, int D, int MODE>
__global__ __launch_bounds__(256) void attention_vjp_kernel(AttnVjpArgs a) {
  constexpr int SZ = (int)sizeof(T), EPC = 16 / SZ;
  constexpr int RS = D * SZ + 16;         // staged rows: [32][D]
  constexpr int TS = 32 * SZ + 16;        // staged transposes: [D][32]
  constexpr int QS = D * SZ / 32;         // 32-byte k-steps of a contraction over the head's channels
  constexpr int PS = 32 * SZ / 32;        // ... over the 32 walked rows
  constexpr int PPR = D * SZ / 16;        // 16-byte pieces per row
  __shared__ __attribute__((aligned(16))) char y1_s[32 * RS];
  __shared__ __attribute__((aligned(16))) char y2_s[32 * RS];
  __shared__ __attribute__((aligned(16))) char y1t_s[D * TS];
  __shared__ __attribute__((aligned(16))) char y2t_s[MODE == 1 ? D * TS : 16];
  __shared__ float lse_s[32], delta_s[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int T_ = a.T;
  const int own = blockIdx.x * 128 + wave * 32 + r;
  const T* qbase = reinterpret_cast<const T*>(a.qkv) + (long)b * T_ * a.ld_qkv + head * 3 * D;
  const T* gbase = reinterpret_cast<const T*>(a.d_out) + (long)b * T_ * a.ld_out + head * D;
  const float* lse = a.lse + ((long)b * a.heads + head) * T_;
  const float* delta = a.delta + ((long)b * a.heads + head) * T_;

  // own row's operands: MODE 0: (Q, dO) of the query; MODE 1: (K, V) of the key
  u32x4 x1f[QS], x2f[QS];
#pragma unroll
  for (int ks = 0; ks < QS; ks++) {
    x1f[ks] = x2f[ks] = u32x4{0u, 0u, 0u, 0u};
    if (own < T_) {
      const int off = ks * (32 / SZ) + h * EPC;
      if (MODE == 0) {
        x1f[ks] = *reinterpret_cast<const u32x4*>(qbase + (long)own * a.ld_qkv + off);
        x2f[ks] = *reinterpret_cast<const u32x4*>(gbase + (long)own * a.ld_out + off);
      } else {
        x1f[ks] = *reinterpret_cast<const u32x4*>(qbase + (long)own * a.ld_qkv + D + off);
        x2f[ks] = *reinterpret_cast<const u32x4*>(qbase + (long)own * a.ld_qkv + 2 * D + off);
      }
    }
  }
  float own_lse = 0.f, own_delta = 0.f;
  if (MODE == 0 && own < T_) { own_lse = lse[own]; own_delta = delta[own]; }
  f32x16 g1[D / 32], g2[MODE == 1 ? D / 32 : 1];
#pragma unroll
  for (int i = 0; i < D / 32; i++)
#pragma unroll
    for (int e = 0; e < 16; e++) g1[i][e] = 0.f;
  if (MODE == 1) {
#pragma unroll
    for (int i = 0; i < D / 32; i++)
#pragma unroll
      for (int e = 0; e < 16; e++) g2[i][e] = 0.f;
  }

  // the walked block's rows travel HBM -> registers one block AHEAD of their use (round 6: a block's loads used to be issued between
  // the loop's two barriers and waited for there - 7 exposed round trips per 197-token head - now they fly under the previous block's
  // products), registers -> LDS (rows and transposes) between the barriers; same values, same order
  constexpr int NP = (32 * PPR + 255) / 256;   // pieces per thread and block (1 or 2)
  u32x4 n1[NP], n2[NP];
  float nl = 1.0e30f, nd = 0.f;
#define MAUA_AV_LOAD(YB_)                                                                            \
  {                                                                                                  \
    _Pragma("unroll") for (int it = 0; it < NP; it++) {                                             \
      const int p = tid + it * 256;                                                                  \
      const int kk = p / PPR, pc = p - kk * PPR;                                                     \
      n1[it] = n2[it] = u32x4{0u, 0u, 0u, 0u};                                                       \
      if (p < 32 * PPR && (YB_) + kk < T_) {                                                         \
        if (MODE == 0) {                                                                             \
          const T* row = qbase + (long)((YB_) + kk) * a.ld_qkv + pc * EPC;                           \
          n1[it] = *reinterpret_cast<const u32x4*>(row + D);                                         \
          n2[it] = *reinterpret_cast<const u32x4*>(row + 2 * D);                                     \
        } else {                                                                                     \
          n1[it] = *reinterpret_cast<const u32x4*>(qbase + (long)((YB_) + kk) * a.ld_qkv + pc * EPC); \
          n2[it] = *reinterpret_cast<const u32x4*>(gbase + (long)((YB_) + kk) * a.ld_out + pc * EPC); \
        }                                                                                            \
      }                                                                                              \
    }                                                                                                \
    if (MODE == 1 && tid < 32) {                                                                     \
      const bool in = (YB_) + tid < T_;                                                              \
      nl = in ? lse[(YB_) + tid] : 1.0e30f; /* exp(s - 1e30) = 0: rows past the end weigh nothing */ \
      nd = in ? delta[(YB_) + tid] : 0.f;                                                            \
    }                                                                                                \
  }
  MAUA_AV_LOAD(0)
  for (int yb = 0; yb < T_; yb += 32) {
    __syncthreads();  // the previous block's fragment reads are done
#pragma unroll
    for (int it = 0; it < NP; it++) {
      const int p = tid + it * 256;
      if (p >= 32 * PPR) break;
      const int kk = p / PPR, pc = p - kk * PPR;
      const u32x4 v1 = n1[it], v2 = n2[it];
      *reinterpret_cast<u32x4*>(y1_s + kk * RS + pc * 16) = v1;
      *reinterpret_cast<u32x4*>(y2_s + kk * RS + pc * 16) = v2;
      if constexpr (SZ == 2) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          *reinterpret_cast<bf16_t*>(y1t_s + (pc * 8 + 2 * e) * TS + kk * 2) = (bf16_t)(v1[e] & 0xffffu);
          *reinterpret_cast<bf16_t*>(y1t_s + (pc * 8 + 2 * e + 1) * TS + kk * 2) = (bf16_t)(v1[e] >> 16);
          if (MODE == 1) {
            *reinterpret_cast<bf16_t*>(y2t_s + (pc * 8 + 2 * e) * TS + kk * 2) = (bf16_t)(v2[e] & 0xffffu);
            *reinterpret_cast<bf16_t*>(y2t_s + (pc * 8 + 2 * e + 1) * TS + kk * 2) = (bf16_t)(v2[e] >> 16);
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          *reinterpret_cast<uint32_t*>(y1t_s + (pc * 4 + e) * TS + kk * 4) = v1[e];
          if (MODE == 1) *reinterpret_cast<uint32_t*>(y2t_s + (pc * 4 + e) * TS + kk * 4) = v2[e];
        }
      }
    }
    if (MODE == 1 && tid < 32) {
      lse_s[tid] = nl;
      delta_s[tid] = nd;
    }
    __syncthreads();
    if (yb + 32 < T_) MAUA_AV_LOAD(yb + 32)   // (flies during this block's products)

    // s, dp: rows = the walked block, columns = own rows
    f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; e++) s[e] = dp[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < QS; ks++) {
      const u32x4 f1 = *reinterpret_cast<const u32x4*>(y1_s + r * RS + ks * 32 + h * 16);
      const u32x4 f2 = *reinterpret_cast<const u32x4*>(y2_s + r * RS + ks * 32 + h * 16);
      VMma<T>::step(s, f1, x1f[ks]);
      VMma<T>::step(dp, f2, x2f[ks]);
    }
    // lane (own r, half h): element e belongs to walked row yb + 8 (e / 4) + 4 h + e % 4.  s becomes p, dp becomes ds.
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int li = 8 * (e >> 2) + 4 * h + (e & 3);
      float l, dl;
      if (MODE == 0) { l = own_lse; dl = own_delta; }
      else { l = lse_s[li]; dl = delta_s[li]; }
      float pv;
      if constexpr (SZ == 2) pv = __expf(s[e] * a.scale - l);
      else pv = expf(s[e] * a.scale - l);
      if (MODE == 0 && yb + li >= T_) pv = 0.f;
      s[e] = pv;
      dp[e] = pv * (dp[e] - dl) * a.scale;
    }
#pragma unroll
    for (int j = 0; j < PS; j++) {
      u32x4 pf, df;
      if constexpr (SZ == 2) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          pf[k] = pack2bf(s[8 * j + 2 * k], s[8 * j + 2 * k + 1]);
          df[k] = pack2bf(dp[8 * j + 2 * k], dp[8 * j + 2 * k + 1]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) { pf[k] = __float_as_uint(s[4 * j + k]); df[k] = __float_as_uint(dp[4 * j + k]); }
      }
#pragma unroll
      for (int i = 0; i < D / 32; i++) {
        u32x4 t1, t2 = u32x4{0u, 0u, 0u, 0u};
        const char* row1 = y1t_s + (i * 32 + r) * TS;
        const char* row2 = y2t_s + (i * 32 + r) * TS;
        if constexpr (SZ == 2) {
          // element e of half h <-> walked row 16 j + 8 (e >> 2) + 4 h + (e & 3)
          const uint2 lo = *reinterpret_cast<const uint2*>(row1 + (16 * j + 4 * h) * 2);
          const uint2 hi = *reinterpret_cast<const uint2*>(row1 + (16 * j + 8 + 4 * h) * 2);
          t1 = u32x4{lo.x, lo.y, hi.x, hi.y};
          if (MODE == 1) {
            const uint2 lo2 = *reinterpret_cast<const uint2*>(row2 + (16 * j + 4 * h) * 2);
            const uint2 hi2 = *reinterpret_cast<const uint2*>(row2 + (16 * j + 8 + 4 * h) * 2);
            t2 = u32x4{lo2.x, lo2.y, hi2.x, hi2.y};
          }
        } else {
          // element e of half h <-> walked row 8 j + 4 h + e
          t1 = *reinterpret_cast<const u32x4*>(row1 + (8 * j + 4 * h) * 4);
          if (MODE == 1) t2 = *reinterpret_cast<const u32x4*>(row2 + (8 * j + 4 * h) * 4);
        }
        VMma<T>::step(g1[i], t1, df);                  // MODE 0: dQ^T += K^T ds^T;  MODE 1: dK^T += Q^T ds
        if (MODE == 1) VMma<T>::step(g2[i], t2, pf);   //                            MODE 1: dV^T += dO^T p
      }
    }
  }
  if (own >= T_) return;
  T* orow = reinterpret_cast<T*>(a.d_qkv) + ((long)b * T_ + own) * a.ld_qkv + head * 3 * D + (MODE == 0 ? 0 : D);
#pragma unroll
  for (int i = 0; i < D / 32; i++)
#pragma unroll
    for (int qd = 0; qd < 4; qd++) {
      const int d = i * 32 + 8 * qd + 4 * h;
      if constexpr (SZ == 2) {
        *reinterpret_cast<uint2*>(orow + d) = make_uint2(pack2bf(g1[i][qd * 4], g1[i][qd * 4 + 1]), pack2bf(g1[i][qd * 4 + 2], g1[i][qd * 4 + 3]));
        if (MODE == 1)
          *reinterpret_cast<uint2*>(orow + D + d) =
              make_uint2(pack2bf(g2[i][qd * 4], g2[i][qd * 4 + 1]), pack2bf(g2[i][qd * 4 + 2], g2[i][qd * 4 + 3]));
      } else {
        *reinterpret_cast<float4*>(orow + d) = make_float4(g1[i][qd * 4], g1[i][qd * 4 + 1], g1[i][qd * 4 + 2], g1[i][qd * 4 + 3]);
        if (MODE == 1)
          *reinterpret_cast<float4*>(orow + D + d) = make_float4(g2[i][qd * 4], g2[i][qd * 4 + 1], g2[i][qd * 4 + 2], g2[i][qd * 4 + 3]);
      }
    }
}

#undef MAUA_AV_LOAD

}  // namespace

int launch_attention_vjp(hipStream_t stream, int dtype, const AttnVjpArgs& a) {
  MAUA_REQUIRE(dtype == MAUA_BF16 || dtype == MAUA_F32, "attention_vjp: unsupported dtype");
  MAUA_REQUIRE(attention_supported(a.D), "attention_vjp: head channels must be 32 or 64");
  MAUA_REQUIRE(a.qkv && a.out && a.d_out && a.lse && a.d_qkv && a.delta && a.T > 0 && a.heads > 0 && a.B <= 65535 && a.heads <= 65535,
               "attention_vjp: bad arguments");
  if (a.B == 0) return MAUA_OK;
  const long rows = (long)a.B * a.T * a.heads * (a.D * (dtype == MAUA_BF16 ? 2 : 4) / 16);   // 16-byte pieces of out / d_out
  dim3 grid((unsigned)((a.T + 127) / 128), (unsigned)a.heads, (unsigned)a.B);
#define MAUA_ATTN_VJP(TT, DD)                                                                                   \
  do {                                                                                                          \
    hipLaunchKernelGGL(attn_delta_kernel<TT>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, a);   \
    hipLaunchKernelGGL((attention_vjp_kernel<TT, DD, 0>), grid, dim3(256), 0, stream, a);                       \
    hipLaunchKernelGGL((attention_vjp_kernel<TT, DD, 1>), grid, dim3(256), 0, stream, a);                       \
  } while (0)
  if (dtype == MAUA_BF16) {
    if (a.D == 64) MAUA_ATTN_VJP(bf16_t, 64); else MAUA_ATTN_VJP(bf16_t, 32);
  } else {
    if (a.D == 64) MAUA_ATTN_VJP(float, 64); else MAUA_ATTN_VJP(float, 32);
  }
#undef MAUA_ATTN_VJP
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace maua
