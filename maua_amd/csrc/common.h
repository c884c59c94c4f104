// Shared device/host helpers for the gfx950 kernels of libmaua_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/maua_hip.h"

namespace maua {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;  // native 16-byte register quad (stays in VGPRs)

typedef uint16_t bf16_t;  // raw bfloat16 bits

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even via the gfx950 hardware conversion (v_cvt_pk_bf16_f32: one instruction per two values)
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }

// [-1, 1] float -> u8 exactly as render/ffmpeg.py:72 + ops/io.py:47-70: (x + 1) / 2, clamp, * 255, round half to even
__device__ __forceinline__ uint32_t to_u8(float x) {
  float v = (x + 1.0f) / 2.0f;
  v = fminf(fmaxf(v, 0.f), 1.f);
  return (uint32_t)__float2int_rn(v * 255.0f);
}

// IEEE half (MAUA_F16, round 5: the reference's own render dtype - render/ffmpeg.py:45, wrappers/__init__.py fp16=True): raw
// bits in a type of its own - bf16_t is a plain uint16_t - so that kernels templated on the element type can tell the two 16-bit
// formats apart.  The fast bf16 kernels (LDS-direct convolutions, the walks) have no f16 form: an F16 network runs the generic
// templated kernels on v_mfma_f32_32x32x16_f16.
struct f16_t { uint16_t bits; };
typedef _Float16 hw_f16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ float h2f(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
__device__ __forceinline__ uint32_t pack2h(float lo, float hi) {   // round to nearest even (v_cvt_f16_f32)
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_f16x2));
}
// the two 16-bit storage formats behind one interface: a packed pair <-> two floats, and the value a float has once stored
template <typename T> struct Fmt16;
template <> struct Fmt16<bf16_t> {
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) { return pack2bf(lo, hi); }
  __device__ static __forceinline__ float lo(uint32_t u) { return bf2f((bf16_t)(u & 0xffffu)); }
  __device__ static __forceinline__ float hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
  __device__ static __forceinline__ float round(float f) { return bf2f(f2bf(f)); }
};
template <> struct Fmt16<f16_t> {
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) { return pack2h(lo, hi); }
  __device__ static __forceinline__ float lo(uint32_t u) { return h2f((uint16_t)(u & 0xffffu)); }
  __device__ static __forceinline__ float hi(uint32_t u) { return h2f((uint16_t)(u >> 16)); }
  __device__ static __forceinline__ float round(float f) { return h2f((uint16_t)(pack2h(f, 0.f) & 0xffffu)); }
};

// the 32x32x16 matrix instruction of a 16-bit storage format (operands as raw 16-byte register quads)
template <typename F> struct Mma16;
template <> struct Mma16<bf16_t> {
  __device__ static __forceinline__ void step(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
};
template <> struct Mma16<f16_t> {
  __device__ static __forceinline__ void step(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
  }
};

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kDtype = MAUA_F32;
  __device__ static __forceinline__ float load(const float* p) { return *p; }
  __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
// float32 storage whose PRODUCTS run on the bf16 matrix cores as three split products (MAUA_F32_SPLIT): same bytes as float
struct f32s_t { float v; };
template <> struct Elem<f32s_t> {
  static constexpr int kDtype = MAUA_F32_SPLIT;
  __device__ static __forceinline__ float load(const f32s_t* p) { return p->v; }
  __device__ static __forceinline__ void store(f32s_t* p, float v) { p->v = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int kDtype = MAUA_BF16;
  __device__ static __forceinline__ float load(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void store(bf16_t* p, float v) { *p = f2bf(v); }
};

template <> struct Elem<f16_t> {
  static constexpr int kDtype = MAUA_F16;
  __device__ static __forceinline__ float load(const f16_t* p) { return h2f(p->bits); }
  __device__ static __forceinline__ void store(f16_t* p, float v) { p->bits = (uint16_t)(pack2h(v, 0.f) & 0xffffu); }
};

// activation ids follow maua_act in the header (reference ops.py:44-62)
__device__ __forceinline__ float activate(float x, int act, float alpha) {
  switch (act) {
    case MAUA_ACT_LINEAR: return x;
    case MAUA_ACT_RELU: return x > 0.f ? x : 0.f;
    case MAUA_ACT_LRELU: return x > 0.f ? x : x * alpha;
    case MAUA_ACT_TANH: return tanhf(x);
    case MAUA_ACT_SIGMOID: return 1.f / (1.f + expf(-x));
    case MAUA_ACT_ELU: return x > 0.f ? x : expm1f(x);
    case MAUA_ACT_SELU: {
      const float a = 1.6732632423543772848170429916717f, s = 1.0507009873554804934193349852946f;
      return s * (x > 0.f ? x : a * expm1f(x));
    }
    case MAUA_ACT_SOFTPLUS: return x > 20.f ? x : log1pf(expf(x));
    case MAUA_ACT_SWISH: return x / (1.f + expf(-x));
  }
  return x;
}

// LDS-direct load, 16 bytes per lane: lane l of the wave fetches 16 bytes from its own global address into bytes
// [16 l, 16 l + 16) of the 1 KB LDS slot at `lds_wave_base` (wave-uniform) - no registers in flight, no ds_write.
// Issued from inline assembly on purpose: with the builtin the compiler's waitcnt insertion treats later LDS reads as
// aliasing and drains vmcnt(0) before the first of them, which also waits for every prefetch issued in between.  The
// caller orders the data itself: the request is OLDER than register loads whose consumption (counted vmcnt retires in
// order) precedes a barrier, and the LDS data are read only after that barrier.  M0 carries the LDS base; the kernels
// that use this do not use M0 for anything else.
__device__ __forceinline__ void lds_dma_b128(const void* g, void* lds_wave_base) {
  const unsigned base =
      __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(base) : "memory");
}

// same with a wave-uniform base pointer (SGPR pair) + a 32-bit per-lane byte offset: one address VGPR instead of two
__device__ __forceinline__ void lds_dma_b128(const void* sbase, unsigned voff_bytes, void* lds_wave_base) {
  const unsigned base =
      __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff_bytes), "s"(sbase), "s"(base)
               : "memory");
}

// ---- host side -------------------------------------------------------------------------------------
void set_error(const std::string& msg);
int fail(const std::string& msg);  // sets the thread-local error, returns MAUA_ERR
#define MAUA_HIP_CHECK(expr)                                                                  \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return ::maua::fail(std::string(#expr) + ": " + hipGetErrorString(_e));                 \
  } while (0)
#define MAUA_REQUIRE(cond, msg)                                   \
  do {                                                            \
    if (!(cond)) return ::maua::fail(std::string(msg));           \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace maua

struct maua_ctx {
  int device;
  hipStream_t stream;
  // grow-only scratch arena for the operator-level entry points (layout conversion, prepared weights)
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  int dma_conv = 1;  // operator-level modconv: eligible shapes run the LDS-direct-load kernel (maua_ctx_set_option)
  int gemm_dma = 1;  // clip.hip: large plain GEMMs on gemm_dma.hip (option "gemm_dma")
  int linear_dma = 0;  // maua_linear_nt: the same routing for operator-level callers (option "linear_dma")
};

namespace maua {
// returns a 256-byte aligned carve-out of ctx's scratch arena; grows (synchronising) when too small
int scratch_reserve(maua_ctx* ctx, size_t bytes);
}
