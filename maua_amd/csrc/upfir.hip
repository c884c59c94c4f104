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

// The kernel is bound by VALU and vector-memory instruction issue, not by bytes, so the FIR is evaluated separably
// with as much sharing as registers allow: a thread owns one 16-byte channel piece of a 2-column strip and walks
// down UPFIR_ROWS output rows.  Per t row it loads 5 pieces and forms the two horizontal sums
// (g = [1,3,3,1]/4 symmetric: .25*(c0+c3) + .75*(c1+c2)); the last four horizontal rows live in registers and
// every output row is their vertical combination.  2.5 loads and ~13 VALU ops per output value (a 2 x 2 block per
// thread costs 6.25 loads / 26 ops; measured 0.43 ms -> see DESIGN_LOG.md 4.2 at 512^2 x 64 ch, B = 16).
constexpr int UPFIR_ROWS = 32;

// channel pairs as float2 so the sums map to packed-f32 instructions (v_pk_add/mul/fma_f32)
template <typename T, int EP2>
__device__ __forceinline__ void upfir_hrow(const char* __restrict__ tb, uint32_t off, uint32_t pxb, bool rowok, bool lok,
                                           bool rok, f32x2_t (&h)[2][EP2]) {
  f32x2_t c[5][EP2];
#pragma unroll
  for (int rx = 0; rx < 5; rx++) {
    // strip ends: read a valid neighbour instead and zero it (no divergent branch around the load)
    const bool ok = rx == 0 ? lok : rx == 4 ? rok : true;
    const uint32_t o = off + (rx == 0 && !lok ? pxb : rx == 4 && !rok ? 3 * pxb : rx * pxb);
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (rowok) v = *reinterpret_cast<const u32x4*>(tb + o);
    if (rx == 0 || rx == 4) {
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = ok ? v[k] : 0u;
    }
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int k = 0; k < 4; k++) c[rx][k] = f32x2_t{Fmt16<T>::lo(v[k]), Fmt16<T>::hi(v[k])};
    } else {
      c[rx][0] = f32x2_t{__uint_as_float(v[0]), __uint_as_float(v[1])};
      c[rx][1] = f32x2_t{__uint_as_float(v[2]), __uint_as_float(v[3])};
    }
  }
  // un-normalised taps [1,3,3,1]: the 1/16 of the two passes is folded into the demodulation coefficients
#pragma unroll
  for (int e = 0; e < EP2; e++) {
    h[0][e] = (c[0][e] + c[3][e]) + 3.f * (c[1][e] + c[2][e]);
    h[1][e] = (c[1][e] + c[4][e]) + 3.f * (c[2][e] + c[3][e]);
  }
}

template <typename T, bool LRELU, int EP2>
__device__ __forceinline__ void upfir_emit(const UpfirArgs& a, char* __restrict__ yb, const float* __restrict__ nb,
                                           bool rowok, uint32_t yoff, uint32_t pxb, uint32_t noff,
                                           const f32x2_t (&dv)[EP2], const f32x2_t (&bv)[EP2],
                                           const f32x2_t (&sv)[EP2], float nzs, float cl,
                                           const f32x2_t (&r0)[2][EP2], const f32x2_t (&r1)[2][EP2],
                                           const f32x2_t (&r2)[2][EP2], const f32x2_t (&r3)[2][EP2]) {
  if (!rowok) return;
  float nz[2] = {0.f, 0.f};
  if (nb) {
    const float2 n2 = *reinterpret_cast<const float2*>(nb + noff);  // X0 even, Wo even
    nz[0] = n2.x * nzs;
    nz[1] = n2.y * nzs;
  }
#pragma unroll
  for (int j = 0; j < 2; j++) {
    f32x2_t o[EP2];
#pragma unroll
    for (int e = 0; e < EP2; e++) {
      const f32x2_t acc = (r0[j][e] + r3[j][e]) + 3.f * (r1[j][e] + r2[j][e]);
      f32x2_t t = acc * dv[e] + (bv[e] + nz[j]);
      if constexpr (LRELU) {
        const f32x2_t ta = t * a.alpha;  // 0 <= alpha <= 1
        t = f32x2_t{fmaxf(t[0], ta[0]), fmaxf(t[1], ta[1])};
      } else {
        t = f32x2_t{activate(t[0], a.act, a.alpha), activate(t[1], a.act, a.alpha)} * a.gain;
      }
      // (sv: the NEXT layer's styles when its kernel wants pre-modulated input, else 1)
      o[e] = f32x2_t{__builtin_amdgcn_fmed3f(t[0], -cl, cl), __builtin_amdgcn_fmed3f(t[1], -cl, cl)} * sv[e];
    }
    char* dst = yb + yoff + j * pxb;
    if constexpr (sizeof(T) == 2)
      *reinterpret_cast<u32x4*>(dst) = u32x4{Fmt16<T>::pack2(o[0][0], o[0][1]), Fmt16<T>::pack2(o[1][0], o[1][1]),
                                             Fmt16<T>::pack2(o[2][0], o[2][1]), Fmt16<T>::pack2(o[3][0], o[3][1])};
    else
      *reinterpret_cast<f32x4*>(dst) = f32x4{o[0][0], o[0][1], o[1][0], o[1][1]};
  }
}

template <typename T, bool LRELU>
__global__ __launch_bounds__(256) void upfir_epilogue_kernel(UpfirArgs a) {
  constexpr int EPC = 16 / (int)sizeof(T), EP2 = EPC / 2;
  const int Ht = 2 * a.H + 1, Wt = 2 * a.W + 1, Ho = 2 * a.H, Wo = 2 * a.W;
  const int pieces = a.Co / EPC;
  const int bw = Wo / 2;                                  // 2-column strips
  const int segs = (Ho + UPFIR_ROWS - 1) / UPFIR_ROWS;    // row segments
  const long total = (long)segs * bw * pieces;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int b = blockIdx.y;
  const int pc = (int)(idx % pieces);
  const int pb = (int)(idx / pieces);
  const int bx = pb % bw, seg = pb / bw;
  const int X0 = 2 * bx, Y0 = seg * UPFIR_ROWS, cho = pc * EPC;
  // per-sample bases are wave-uniform; everything below is 32-bit byte offsets from them (launcher checks the sizes)
  const char* tb = reinterpret_cast<const char*>(a.t) + (long)b * Ht * Wt * a.Co * (long)sizeof(T);
  char* yb = reinterpret_cast<char*>(a.y) + (long)b * Ho * Wo * a.Co * (long)sizeof(T);
  const float* nb = a.noise ? a.noise + (long)b * a.noise_bstride : nullptr;
  const uint32_t pxb = (uint32_t)a.Co * (uint32_t)sizeof(T);  // bytes per pixel
  const uint32_t trb = (uint32_t)Wt * pxb, yrb = (uint32_t)Wo * pxb;
  const bool lok = X0 > 0, rok = X0 + 3 < Wt;

  f32x2_t dv[EP2], bv[EP2], sv[EP2];
#pragma unroll
  for (int e4 = 0; e4 < EPC; e4 += 4) {
    const float4 s4 = a.out_scale ? *reinterpret_cast<const float4*>(a.out_scale + (long)b * a.Co + cho + e4)
                                  : make_float4(1.f, 1.f, 1.f, 1.f);
    sv[e4 / 2] = f32x2_t{s4.x, s4.y}; sv[e4 / 2 + 1] = f32x2_t{s4.z, s4.w};
    const float4 d4 = a.d ? *reinterpret_cast<const float4*>(a.d + (long)b * a.Co + cho + e4)
                          : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 b4 = a.bias ? *reinterpret_cast<const float4*>(a.bias + cho + e4) : make_float4(0.f, 0.f, 0.f, 0.f);
    dv[e4 / 2] = f32x2_t{d4.x, d4.y}; dv[e4 / 2 + 1] = f32x2_t{d4.z, d4.w};
    bv[e4 / 2] = f32x2_t{b4.x, b4.y}; bv[e4 / 2 + 1] = f32x2_t{b4.z, b4.w};
  }
  // FIR normalisation (1/4 per tap and pass -> 1/16) and, for lrelu (positively homogeneous), the gain are folded
  // into the coefficients
  const float gfold = LRELU ? a.gain : 1.f;
#pragma unroll
  for (int e = 0; e < EP2; e++) { dv[e] *= 0.0625f * gfold; bv[e] *= gfold; }
  const float nzs = a.noise_strength * gfold * (a.noise_scale ? a.noise_scale[b] : 1.f);
  const float cl = a.clamp >= 0.f ? a.clamp : 3.0e38f;

  // out row yo needs the horizontal sums of t rows yo-1 .. yo+2 (rows outside [0, Ht) are zero)
  f32x2_t h0[2][EP2], h1[2][EP2], h2[2][EP2], h3[2][EP2];
  uint32_t toff = (uint32_t)((Y0 * Wt + X0) * (int)pxb) - pxb + (uint32_t)cho * (uint32_t)sizeof(T);  // row Y0, col X0-1
  upfir_hrow<T, EP2>(tb, toff - trb, pxb, Y0 > 0, lok, rok, h0);
  upfir_hrow<T, EP2>(tb, toff, pxb, true, lok, rok, h1);
  upfir_hrow<T, EP2>(tb, toff + trb, pxb, true, lok, rok, h2);
  toff += 2 * trb;  // row yo + 2
  uint32_t yoff = (uint32_t)((Y0 * Wo + X0) * (int)pxb) + (uint32_t)cho * (uint32_t)sizeof(T);
  uint32_t noff = (uint32_t)(Y0 * Wo + X0);
  const int yend = min(Y0 + UPFIR_ROWS, Ho);
  for (int yo = Y0; yo < yend; yo += 4) {
    upfir_hrow<T, EP2>(tb, toff, pxb, yo + 2 < Ht, lok, rok, h3);
    upfir_emit<T, LRELU, EP2>(a, yb, nb, true, yoff, pxb, noff, dv, bv, sv, nzs, cl, h0, h1, h2, h3);
    upfir_hrow<T, EP2>(tb, toff + trb, pxb, yo + 3 < Ht, lok, rok, h0);
    upfir_emit<T, LRELU, EP2>(a, yb, nb, yo + 1 < Ho, yoff + yrb, pxb, noff + Wo, dv, bv, sv, nzs, cl, h1, h2, h3, h0);
    upfir_hrow<T, EP2>(tb, toff + 2 * trb, pxb, yo + 4 < Ht, lok, rok, h1);
    upfir_emit<T, LRELU, EP2>(a, yb, nb, yo + 2 < Ho, yoff + 2 * yrb, pxb, noff + 2 * Wo, dv, bv, sv, nzs, cl, h2, h3, h0, h1);
    upfir_hrow<T, EP2>(tb, toff + 3 * trb, pxb, yo + 5 < Ht, lok, rok, h2);
    upfir_emit<T, LRELU, EP2>(a, yb, nb, yo + 3 < Ho, yoff + 3 * yrb, pxb, noff + 3 * Wo, dv, bv, sv, nzs, cl, h3, h0, h1, h2);
    toff += 4 * trb;
    yoff += 4 * yrb;
    noff += 4 * Wo;
  }
}

int launch_upfir_epilogue(hipStream_t stream, int dtype, const UpfirArgs& a) {
  if (a.B == 0) return MAUA_OK;
  const int epc = dtype == MAUA_F32 ? 4 : 8;
  MAUA_REQUIRE(a.Co % epc == 0, "upfir_epilogue: Co must be a multiple of the 16-byte piece");
  MAUA_REQUIRE(!a.out_scale || ((uintptr_t)a.out_scale % 16) == 0, "upfir_epilogue: out_scale must be 16-byte aligned");
  MAUA_REQUIRE((!a.d || ((uintptr_t)a.d % 16) == 0) && (!a.bias || ((uintptr_t)a.bias % 16) == 0),
               "upfir_epilogue: d and bias must be 16-byte aligned");
  MAUA_REQUIRE(!a.noise || (((uintptr_t)a.noise % 8) == 0 && a.noise_bstride % 2 == 0),
               "upfir_epilogue: noise must be 8-byte aligned");
  MAUA_REQUIRE((long)(2 * a.H + 1) * (2 * a.W + 1) * a.Co * (dtype == MAUA_F32 ? 4 : 2) < (1L << 31),
               "upfir_epilogue: a sample of t must stay below 2 GiB (32-bit offsets)");
  const long total = (long)((2 * a.H + UPFIR_ROWS - 1) / UPFIR_ROWS) * a.W * (a.Co / epc);
  const dim3 grid((unsigned)((total + 255) / 256), a.B);
  const bool lr = a.act == MAUA_ACT_LRELU && a.alpha >= 0.f && a.alpha <= 1.f && a.gain > 0.f;  // folded-gain fast path
  if (dtype == MAUA_BF16) {
    if (lr) hipLaunchKernelGGL((upfir_epilogue_kernel<bf16_t, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((upfir_epilogue_kernel<bf16_t, false>), grid, dim3(256), 0, stream, a);
  } else if (dtype == MAUA_F16) {
    if (lr) hipLaunchKernelGGL((upfir_epilogue_kernel<f16_t, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((upfir_epilogue_kernel<f16_t, false>), grid, dim3(256), 0, stream, a);
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
