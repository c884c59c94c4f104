// Multi-head self-attention over the pixels of a feature map (guided-diffusion's QKVAttentionLegacy) as one fused
// MFMA kernel: S = Q K^T / sqrt(ch), online softmax in float32, O = P V - the [T x T] weight matrix never leaves the
// registers.
//
// Replaces (reference): guided_diffusion/unet.py `QKVAttentionLegacy.forward` (the network maua/diffusion/processors/
// guided.py:164-209 builds with num_head_channels=64 and attention at 32 / 16 / 8; un-vendored submodule):
//     q, k, v = qkv.reshape(bs * n_heads, ch * 3, length).split(ch, dim=1)     # channel layout [head][q | k | v][ch]
//     weight  = softmax(einsum("bct,bcs->bts", q * scale, k * scale).float())   # scale = ch ** -0.25
//     a       = einsum("bts,bcs->bct", weight, v)
//
// Layout: qkv is the NHWC output of the qkv GEMM, [B][T][3C] with channel = head * 3 ch + {0, ch, 2 ch} + c; the result is
// [B][T][C] with channel = head * ch + c (what `a.reshape(bs, -1, length)` gives), ready for the proj_out GEMM.
//
// One workgroup = 4 waves = 128 queries of one (sample, head); a wave owns 32 queries.  Per 32-key block:
//   * K rows and V^T ([ch][32 keys]) are staged in LDS (V is transposed on the way in: the PV product contracts over
//     keys, and an MFMA operand holds consecutive k per lane);
//   * S^T = K Q^T (swapped operands: a lane then holds 16 of its query's 32 scores, the other 16 sit in lane ^ 32), so the
//     row maximum / sum are in-lane reductions plus ONE cross-half exchange;
//   * P goes straight from the accumulator registers into the PV MFMA's operand (the accumulator's key order
//     8 q + 4 h + k is matched by the order in which the V^T fragment is read).
// bf16: v_mfma_f32_32x32x16_bf16, P rounded to bf16; f32 (parity mode): v_mfma_f32_32x32x2_f32, exact products, expf.
// Attention is < 1 % of the UNet's FLOPs at 256^2 (T <= 1024, ch = 64): this kernel is built for exactness and for
// not materialising T x T, not for the MFMA roof.
#include "common.h"
#include "internal.h"

namespace maua {

namespace {

template <typename T> struct AMma;
template <> struct AMma<bf16_t> {
  __device__ static __forceinline__ void step(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0,
                                                  0, 0);
  }
};
template <> struct AMma<float> {
  __device__ static __forceinline__ void step(f32x16& acc, const u32x4& a, const u32x4& b) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], acc, 0, 0, 0);
  }
};

template <typename T, int D>
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs a) {
  constexpr int SZ = (int)sizeof(T), EPC = 16 / SZ;
  constexpr int KRS = D * SZ + 16;        // K rows: [32 keys][D]
  constexpr int VRS = 32 * SZ + 16;       // V^T rows: [D][32 keys]
  constexpr int QS = D * SZ / 32;         // 32-byte k-steps of the QK^T product
  constexpr int PS = 32 * SZ / 32;        // ... of the PV product per 32-key block (bf16: 2 x 16 keys, f32: 4 x 8 keys)
  constexpr int PPR = D * SZ / 16;        // 16-byte pieces per K / V row
  __shared__ __attribute__((aligned(16))) char k_s[32 * KRS];
  __shared__ __attribute__((aligned(16))) char v_s[D * VRS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int T_ = a.T;
  const int qrow = blockIdx.x * 128 + wave * 32 + r;
  const T* base = reinterpret_cast<const T*>(a.qkv) + (long)b * T_ * a.ld_qkv + head * 3 * D;

  u32x4 qf[QS];
#pragma unroll
  for (int ks = 0; ks < QS; ks++) {
    qf[ks] = u32x4{0u, 0u, 0u, 0u};
    if (qrow < T_) qf[ks] = *reinterpret_cast<const u32x4*>(base + (long)qrow * a.ld_qkv + ks * (32 / SZ) + h * EPC);
  }
  f32x16 o[D / 32];
#pragma unroll
  for (int i = 0; i < D / 32; i++)
#pragma unroll
    for (int e = 0; e < 16; e++) o[i][e] = 0.f;
  float m_i = -1.0e30f, l_i = 0.f;

  // K / V rows travel HBM -> registers one key block AHEAD of their use (round 6; see attention_vjp.hip), registers -> LDS between the
  // loop's barriers
  constexpr int NP = (32 * PPR + 255) / 256;
  u32x4 nk[NP], nv[NP];
#define MAUA_AT_LOAD(KB_)                                                       \
  _Pragma("unroll") for (int it = 0; it < NP; it++) {                          \
    const int p = tid + it * 256;                                               \
    const int kk = p / PPR, pc = p - kk * PPR;                                  \
    nk[it] = nv[it] = u32x4{0u, 0u, 0u, 0u};                                    \
    if (p < 32 * PPR && (KB_) + kk < T_) {                                      \
      const T* row = base + (long)((KB_) + kk) * a.ld_qkv + pc * EPC;           \
      nk[it] = *reinterpret_cast<const u32x4*>(row + D);                        \
      nv[it] = *reinterpret_cast<const u32x4*>(row + 2 * D);                    \
    }                                                                           \
  }
  MAUA_AT_LOAD(0)
  for (int kb = 0; kb < T_; kb += 32) {
    __syncthreads();  // the previous block's fragment reads are done
#pragma unroll
    for (int it = 0; it < NP; it++) {
      const int p = tid + it * 256;
      if (p >= 32 * PPR) break;
      const int kk = p / PPR, pc = p - kk * PPR;
      const u32x4 kv = nk[it], vv = nv[it];
      *reinterpret_cast<u32x4*>(k_s + kk * KRS + pc * 16) = kv;
      // V^T[d][key]
      if constexpr (SZ == 2) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          *reinterpret_cast<bf16_t*>(v_s + (pc * 8 + 2 * e) * VRS + kk * 2) = (bf16_t)(vv[e] & 0xffffu);
          *reinterpret_cast<bf16_t*>(v_s + (pc * 8 + 2 * e + 1) * VRS + kk * 2) = (bf16_t)(vv[e] >> 16);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) *reinterpret_cast<uint32_t*>(v_s + (pc * 4 + e) * VRS + kk * 4) = vv[e];
      }
    }
    __syncthreads();
    if (kb + 32 < T_) MAUA_AT_LOAD(kb + 32)   // (flies during this block's products)

    // S^T block: rows = keys, columns = queries
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; e++) s[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < QS; ks++) {
      const u32x4 kf = *reinterpret_cast<const u32x4*>(k_s + r * KRS + ks * 32 + h * 16);
      AMma<T>::step(s, kf, qf[ks]);
    }
    // lane (query r, half h): s[e] is the score of key kb + 8 (e / 4) + 4 h + e % 4
    float mx = -1.0e30f;
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int key = kb + 8 * (e >> 2) + 4 * h + (e & 3);
      s[e] = key < T_ ? s[e] * a.scale : -__builtin_huge_valf();
      mx = fmaxf(mx, s[e]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_i, mx);
    float alpha, ps = 0.f;
    if constexpr (SZ == 2) {
      alpha = __expf(m_i - m_new);
#pragma unroll
      for (int e = 0; e < 16; e++) { s[e] = __expf(s[e] - m_new); ps += s[e]; }
    } else {
      alpha = expf(m_i - m_new);
#pragma unroll
      for (int e = 0; e < 16; e++) { s[e] = expf(s[e] - m_new); ps += s[e]; }
    }
    ps += __shfl_xor(ps, 32);
    l_i = l_i * alpha + ps;
    m_i = m_new;
#pragma unroll
    for (int i = 0; i < D / 32; i++)
#pragma unroll
      for (int e = 0; e < 16; e++) o[i][e] *= alpha;
    // O^T += V^T P^T: rows = channels d, columns = queries; the k order of both operands is the accumulator's key order
#pragma unroll
    for (int j = 0; j < PS; j++) {
      u32x4 pf;
      if constexpr (SZ == 2) {
#pragma unroll
        for (int k = 0; k < 4; k++) pf[k] = pack2bf(s[8 * j + 2 * k], s[8 * j + 2 * k + 1]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) pf[k] = __float_as_uint(s[4 * j + k]);
      }
#pragma unroll
      for (int i = 0; i < D / 32; i++) {
        u32x4 vf;
        const char* vrow = v_s + (i * 32 + r) * VRS;
        if constexpr (SZ == 2) {
          // element e of half h <-> key 16 j + 8 (e >> 2) + 4 h + (e & 3)
          const uint2 lo = *reinterpret_cast<const uint2*>(vrow + (16 * j + 4 * h) * 2);
          const uint2 hi = *reinterpret_cast<const uint2*>(vrow + (16 * j + 8 + 4 * h) * 2);
          vf = u32x4{lo.x, lo.y, hi.x, hi.y};
        } else {
          // element e of half h <-> key 8 j + 4 h + e
          vf = *reinterpret_cast<const u32x4*>(vrow + (8 * j + 4 * h) * 4);
        }
        AMma<T>::step(o[i], vf, pf);
      }
    }
  }
  if (qrow >= T_) return;
  if (a.lse && h == 0) a.lse[((long)b * a.heads + head) * T_ + qrow] = m_i + logf(l_i);
  const float inv = 1.f / l_i;
  T* orow = reinterpret_cast<T*>(a.out) + ((long)b * T_ + qrow) * a.ld_out + head * D;
#pragma unroll
  for (int i = 0; i < D / 32; i++)
#pragma unroll
    for (int qd = 0; qd < 4; qd++) {
      const int d = i * 32 + 8 * qd + 4 * h;
      const float v0 = o[i][qd * 4] * inv, v1 = o[i][qd * 4 + 1] * inv, v2 = o[i][qd * 4 + 2] * inv, v3 = o[i][qd * 4 + 3] * inv;
      if constexpr (SZ == 2)
        *reinterpret_cast<uint2*>(orow + d) = make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
      else
        *reinterpret_cast<float4*>(orow + d) = make_float4(v0, v1, v2, v3);
    }
}

#undef MAUA_AT_LOAD

}  // namespace

bool attention_supported(int head_ch) { return head_ch == 32 || head_ch == 64; }

int launch_attention(hipStream_t stream, int dtype, const AttnArgs& a) {
  MAUA_REQUIRE(dtype == MAUA_BF16 || dtype == MAUA_F32, "attention: unsupported dtype");
  MAUA_REQUIRE(attention_supported(a.D), "attention: head channels must be 32 or 64");
  MAUA_REQUIRE(a.qkv && a.out && a.T > 0 && a.heads > 0 && a.B <= 65535 && a.heads <= 65535, "attention: bad arguments");
  if (a.B == 0) return MAUA_OK;
  dim3 grid((unsigned)((a.T + 127) / 128), (unsigned)a.heads, (unsigned)a.B);
#define MAUA_ATTN(TT, DD) hipLaunchKernelGGL((attention_kernel<TT, DD>), grid, dim3(256), 0, stream, a)
  if (dtype == MAUA_BF16) {
    if (a.D == 64) MAUA_ATTN(bf16_t, 64); else MAUA_ATTN(bf16_t, 32);
  } else {
    if (a.D == 64) MAUA_ATTN(float, 64); else MAUA_ATTN(float, 32);
  }
#undef MAUA_ATTN
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace maua
