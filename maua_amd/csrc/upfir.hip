// Second half of the minimal up-layer: 4x4 FIR over the transposed-conv tensor t + the layer epilogue.
//
// Replaces (reference): ops.py:225 upfirdn2d(x=t, f, padding=(1,1,1,1), gain=4) (via :87-114), :184-185 noise add,
// :65-84 bias_act — after modconv3x3_kernel (tmode) produced t = conv_transpose2d(x*s, W, stride 2) (ops.py:224).
//   out[Y,X,c] = act( d[b,c] * sum_{u,v<4} t[Y+u-1, X+v-1, c] * F[u][v] + noise[b,Y,X] + bias[c] ) * gain, clamp
// with F = outer(g,g), g = [1,3,3,1]/4 (= f * gain 4), zero outside t's (2H+1) x (2W+1) support.
// Streaming kernel: NHWC, 16-byte channel pieces.
#include "common.h"
#include "internal.h"

namespace maua {

// Each thread produces a 2 x 2 block of outputs for one 16-byte channel piece from the 5 x 5 neighbourhood of t
// (rows/cols outside t's support are skipped), accumulating separably in registers.  (Measured alternatives on
// MI355X at 512^2 x 64 ch, B = 16: this form 0.54 ms; unconditional clamped loads 0.67 ms; 2 x 4 blocks 0.84 ms;
// t staged through LDS tiles 0.78 ms.)
template <typename T, bool LRELU>
__global__ __launch_bounds__(256) void upfir_epilogue_kernel(UpfirArgs a) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const int Ht = 2 * a.H + 1, Wt = 2 * a.W + 1, Ho = 2 * a.H, Wo = 2 * a.W;
  const int pieces = a.Co / EPC;
  const int bw = Wo / 2, bh = Ho / 2;  // 2x2 output blocks
  const long total = (long)bh * bw * pieces;
  const int b = blockIdx.y;
  const T* tb = reinterpret_cast<const T*>(a.t) + (long)b * Ht * Wt * a.Co;
  T* yb = reinterpret_cast<T*>(a.y) + (long)b * Ho * Wo * a.Co;
  const float* nb = a.noise ? a.noise + (long)b * a.noise_bstride : nullptr;
  const float g4[4] = {0.25f, 0.75f, 0.75f, 0.25f};
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int pc = (int)(idx % pieces);
    const long pb = idx / pieces;
    const int bx = (int)(pb % bw), by = (int)(pb / bw);
    const int Y0 = 2 * by, X0 = 2 * bx;  // outputs (Y0..Y0+1, X0..X0+1) need t rows Y0-1..Y0+3, cols X0-1..X0+3
    float acc[2][2][EPC];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int e = 0; e < EPC; e++) acc[i][j][e] = 0.f;
#pragma unroll
    for (int ry = 0; ry < 5; ry++) {
      const int ty = Y0 - 1 + ry;
      if (ty < 0 || ty >= Ht) continue;
      float hrow[2][EPC];
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int e = 0; e < EPC; e++) hrow[j][e] = 0.f;
#pragma unroll
      for (int rx = 0; rx < 5; rx++) {
        const int tx = X0 - 1 + rx;
        if (tx < 0 || tx >= Wt) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(tb + ((long)ty * Wt + tx) * a.Co + pc * EPC);
        float f[EPC];
        if constexpr (sizeof(T) == 2) {
          f[0] = bf2f((bf16_t)(v.x & 0xffff)); f[1] = bf2f((bf16_t)(v.x >> 16));
          f[2] = bf2f((bf16_t)(v.y & 0xffff)); f[3] = bf2f((bf16_t)(v.y >> 16));
          f[4] = bf2f((bf16_t)(v.z & 0xffff)); f[5] = bf2f((bf16_t)(v.z >> 16));
          f[6] = bf2f((bf16_t)(v.w & 0xffff)); f[7] = bf2f((bf16_t)(v.w >> 16));
        } else {
          f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
          f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int vv = rx - j;  // filter column index for output X0+j
          if (vv >= 0 && vv < 4) {
#pragma unroll
            for (int e = 0; e < EPC; e++) hrow[j][e] += f[e] * g4[vv];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int uu = ry - i;  // filter row index for output Y0+i
        if (uu >= 0 && uu < 4) {
#pragma unroll
          for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < EPC; e++) acc[i][j][e] += hrow[j][e] * g4[uu];
        }
      }
    }
    // epilogue
    float dv[EPC], bv[EPC];
#pragma unroll
    for (int e4 = 0; e4 < EPC; e4 += 4) {  // 16-byte loads: the kernel is bound by vector-memory instruction issue
      const float4 d4 = a.d ? *reinterpret_cast<const float4*>(a.d + (long)b * a.Co + pc * EPC + e4)
                            : make_float4(1.f, 1.f, 1.f, 1.f);
      const float4 b4 = a.bias ? *reinterpret_cast<const float4*>(a.bias + pc * EPC + e4) : make_float4(0.f, 0.f, 0.f, 0.f);
      dv[e4] = d4.x; dv[e4 + 1] = d4.y; dv[e4 + 2] = d4.z; dv[e4 + 3] = d4.w;
      bv[e4] = b4.x; bv[e4 + 1] = b4.y; bv[e4 + 2] = b4.z; bv[e4 + 3] = b4.w;
    }
    if constexpr (LRELU) {  // lrelu is positively homogeneous: fold the gain into the coefficients
#pragma unroll
      for (int e = 0; e < EPC; e++) { dv[e] *= a.gain; bv[e] *= a.gain; }
    }
    const float nzs = LRELU ? a.noise_strength * a.gain : a.noise_strength;
    const float cl = a.clamp >= 0.f ? a.clamp : 3.0e38f;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const long pix = (long)(Y0 + i) * Wo + X0 + j;
        const float nz = nb ? nb[pix] * nzs : 0.f;
        float o[EPC];
#pragma unroll
        for (int e = 0; e < EPC; e++) {
          if constexpr (LRELU) {
            float t = fmaf(acc[i][j][e], dv[e], nz + bv[e]);
            t = fmaxf(t, t * a.alpha);  // 0 <= alpha <= 1
            o[e] = __builtin_amdgcn_fmed3f(t, -cl, cl);
          } else {
            float t = activate(acc[i][j][e] * dv[e] + nz + bv[e], a.act, a.alpha) * a.gain;
            o[e] = __builtin_amdgcn_fmed3f(t, -cl, cl);
          }
        }
        T* dst = yb + pix * a.Co + pc * EPC;
        if constexpr (sizeof(T) == 2)
          *reinterpret_cast<uint4*>(dst) = make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]),
                                                      pack2bf(o[6], o[7]));
        else
          *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
      }
  }
}

int launch_upfir_epilogue(hipStream_t stream, int dtype, const UpfirArgs& a) {
  if (a.B == 0) return MAUA_OK;
  const int epc = dtype == MAUA_BF16 ? 8 : 4;
  MAUA_REQUIRE(a.Co % epc == 0, "upfir_epilogue: Co must be a multiple of the 16-byte piece");
  MAUA_REQUIRE((!a.d || ((uintptr_t)a.d % 16) == 0) && (!a.bias || ((uintptr_t)a.bias % 16) == 0),
               "upfir_epilogue: d and bias must be 16-byte aligned");
  const long total = (long)a.H * a.W * (a.Co / epc);
  const dim3 grid((unsigned)std::min<long>((total + 255) / 256, 4096), a.B);
  const bool lr = a.act == MAUA_ACT_LRELU && a.alpha >= 0.f && a.alpha <= 1.f && a.gain > 0.f;  // folded-gain fast path
  if (dtype == MAUA_BF16) {
    if (lr) hipLaunchKernelGGL((upfir_epilogue_kernel<bf16_t, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((upfir_epilogue_kernel<bf16_t, false>), grid, dim3(256), 0, stream, a);
  } else if (dtype == MAUA_F32) {
    if (lr) hipLaunchKernelGGL((upfir_epilogue_kernel<float, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((upfir_epilogue_kernel<float, false>), grid, dim3(256), 0, stream, a);
  } else {
    return fail("upfir_epilogue: unsupported dtype");
  }
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

}  // namespace maua
