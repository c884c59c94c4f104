// Context, error reporting and version of libmaua_hip.so.
#include "common.h"

namespace maua {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(const std::string& msg) {
  g_err = msg;
  return MAUA_ERR;
}
int scratch_reserve(maua_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->scratch_bytes) return MAUA_OK;
  MAUA_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (ctx->scratch) MAUA_HIP_CHECK(hipFree(ctx->scratch));
  ctx->scratch = nullptr;
  ctx->scratch_bytes = 0;
  size_t want = bytes + bytes / 4;
  MAUA_HIP_CHECK(hipMalloc(&ctx->scratch, want));
  ctx->scratch_bytes = want;
  return MAUA_OK;
}
}  // namespace maua

namespace {
// stamp[0] = shader-cycle counter (s_memtime: ticks with the shader clock, so it follows the power-managed frequency),
// stamp[1] = constant-rate counter (s_memrealtime, 100 MHz)
__global__ void clock_stamp_kernel(unsigned long long* stamp) {
  stamp[0] = __builtin_amdgcn_s_memtime();
  stamp[1] = __builtin_amdgcn_s_memrealtime();
}
}  // namespace

extern "C" {

const char* maua_version(void) { return "maua_hip 0.1 (gfx950)"; }
const char* maua_last_error(void) { return maua::g_err.c_str(); }

int maua_ctx_create(int device, void* stream, maua_ctx** out) {
  MAUA_REQUIRE(out != nullptr, "maua_ctx_create: out is NULL");
  int n = 0;
  MAUA_HIP_CHECK(hipGetDeviceCount(&n));
  MAUA_REQUIRE(device >= 0 && device < n, "maua_ctx_create: no such HIP device (this library has no CPU fallback)");
  MAUA_HIP_CHECK(hipSetDevice(device));
  maua_ctx* c = new maua_ctx();
  c->device = device;
  c->stream = (hipStream_t)stream;
  *out = c;
  return MAUA_OK;
}

int maua_ctx_set_stream(maua_ctx* ctx, void* stream) {
  MAUA_REQUIRE(ctx != nullptr, "maua_ctx_set_stream: ctx is NULL");
  ctx->stream = (hipStream_t)stream;
  return MAUA_OK;
}

int maua_ctx_set_option(maua_ctx* ctx, const char* key, int value) {
  MAUA_REQUIRE(ctx && key, "maua_ctx_set_option: NULL argument");
  if (std::string(key) == "dma_conv") {
    ctx->dma_conv = value;
    return MAUA_OK;
  }
  if (std::string(key) == "linear_dma") {
    ctx->linear_dma = value;
    return MAUA_OK;
  }
  if (std::string(key) == "gemm_dma") {   // the CLIP tower's large GEMMs on the LDS-direct kernel (0: gemm.hip's register-staged kernels)
    ctx->gemm_dma = value;
    return MAUA_OK;
  }
  return maua::fail(std::string("maua_ctx_set_option: unknown option ") + key);
}

int maua_ctx_sync(maua_ctx* ctx) {
  MAUA_REQUIRE(ctx != nullptr, "maua_ctx_sync: ctx is NULL");
  MAUA_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return MAUA_OK;
}

int maua_ctx_clock_stamp(maua_ctx* ctx, unsigned long long* stamp_dev) {
  MAUA_REQUIRE(ctx && stamp_dev, "maua_ctx_clock_stamp: NULL argument");
  hipLaunchKernelGGL(clock_stamp_kernel, dim3(1), dim3(1), 0, ctx->stream, stamp_dev);
  MAUA_HIP_CHECK(hipGetLastError());
  return MAUA_OK;
}

void maua_ctx_destroy(maua_ctx* ctx) {
  if (!ctx) return;
  if (ctx->scratch) {
    hipStreamSynchronize(ctx->stream);
    hipFree(ctx->scratch);
  }
  delete ctx;
}

}  // extern "C"
