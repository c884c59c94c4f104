/* The C ABI from plain C: a host that is neither Python nor PyTorch creates the synthesis network, uploads a state dict
 * under the reference's parameter names (inference/stylegan2.py:195-436), renders one frame and writes it as a PPM.
 *
 *   gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/render_frame.c \
 *       -Lmaua_amd/csrc -lmaua_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/maua_amd/csrc -Wl,-rpath,/opt/rocm/lib -lm -o render_frame
 *   ./render_frame out.ppm [resolution = 64] [dtype: 0 = exact f32, 1 = bf16]
 *
 * The parameters are a fixed integer hash of (parameter index, element index), so that a test can rebuild the same network
 * in Python and compare the frames (tests/test_gpu_cabi_example.py). */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "maua_hip.h"

#define W_DIM 64
#define CHANNEL_BASE 2048
#define CHANNEL_MAX 64

#define CHECK(call)                                                                   \
  do {                                                                                \
    if ((call) != MAUA_OK) {                                                          \
      fprintf(stderr, "%s\n  -> %s\n", #call, maua_last_error());                     \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)
#define HIPCHECK(call)                                                                \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                      \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

static uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
/* uniform in [-1, 1) * scale + offset, element k of stream `seed` */
static void fill(float* dst, size_t n, uint32_t seed, float scale, float offset) {
  for (size_t k = 0; k < n; k++)
    dst[k] = ((float)(mix32(seed * 0x9E3779B9U + (uint32_t)k) >> 8) * (1.0f / 16777216.0f) * 2.0f - 1.0f) * scale + offset;
}

static maua_synth* g_net;
static uint32_t g_index;
static float* g_buf;
static int load(const char* name, size_t n, float scale, float offset) {
  fill(g_buf, n, ++g_index, scale, offset);
  if (maua_synth_load(g_net, name, g_buf, n) != MAUA_OK) {
    fprintf(stderr, "maua_synth_load(%s): %s\n", name, maua_last_error());
    return 1;
  }
  return 0;
}

int main(int argc, char** argv) {
  const char* out_path = argc > 1 ? argv[1] : "frame.ppm";
  const int res = argc > 2 ? atoi(argv[2]) : 64;
  const int dtype = argc > 3 ? atoi(argv[3]) : MAUA_BF16;
  maua_ctx* ctx = NULL;
  CHECK(maua_ctx_create(0, NULL, &ctx));
  CHECK(maua_synth_create(ctx, res, W_DIM, CHANNEL_BASE, CHANNEL_MAX, dtype, 0, &g_net));
  g_buf = (float*)malloc(sizeof(float) * CHANNEL_MAX * CHANNEL_MAX * 9);
  /* the state dict: blocks at 4, 8, ... res; channels min(channel_base / res, channel_max); uniform(-sqrt 3, sqrt 3) has the
   * unit variance of the reference's randn init */
  const float s3 = 1.7320508f;
  char name[96];
  int prev = 0, nblk = 0;
  for (int r = 4; r <= res; r *= 2, nblk++) {
    const int c = CHANNEL_BASE / r < CHANNEL_MAX ? CHANNEL_BASE / r : CHANNEL_MAX;
    if (nblk == 0) {
      snprintf(name, sizeof name, "bs.0.const");
      if (load(name, (size_t)c * 16, s3, 0.f)) return 1;
    }
    for (int which = nblk == 0 ? 1 : 0; which < 2; which++) {
      const int ci = which == 0 ? prev : c;
      snprintf(name, sizeof name, "bs.%d.conv%d.affine.weight", nblk, which);
      if (load(name, (size_t)ci * W_DIM, s3, 0.f)) return 1;
      snprintf(name, sizeof name, "bs.%d.conv%d.affine.bias", nblk, which);
      if (load(name, (size_t)ci, 0.f, 1.f)) return 1;
      snprintf(name, sizeof name, "bs.%d.conv%d.weight", nblk, which);
      if (load(name, (size_t)c * ci * 9, s3, 0.f)) return 1;
      snprintf(name, sizeof name, "bs.%d.conv%d.noise_const", nblk, which);
      if (load(name, (size_t)r * r, s3, 0.f)) return 1;
      snprintf(name, sizeof name, "bs.%d.conv%d.bias", nblk, which);
      if (load(name, (size_t)c, 0.1f, 0.f)) return 1;
    }
    snprintf(name, sizeof name, "bs.%d.torgb.affine.weight", nblk);
    if (load(name, (size_t)c * W_DIM, s3, 0.f)) return 1;
    snprintf(name, sizeof name, "bs.%d.torgb.affine.bias", nblk);
    if (load(name, (size_t)c, 0.f, 1.f)) return 1;
    snprintf(name, sizeof name, "bs.%d.torgb.weight", nblk);
    if (load(name, (size_t)3 * c, s3, 0.f)) return 1;
    snprintf(name, sizeof name, "bs.%d.torgb.bias", nblk);
    if (load(name, 3, 0.1f, 0.f)) return 1;
    prev = c;
  }
  const int num_ws = maua_synth_num_ws(g_net);
  /* one frame: ws [1][num_ws][w_dim] (a mapped latent would come from maua_mapping-style calls; here: the hash stream 999) */
  float* ws_h = (float*)malloc(sizeof(float) * num_ws * W_DIM);
  fill(ws_h, (size_t)num_ws * W_DIM, 999, 1.0f, 0.f);
  float* ws_d = NULL;
  uint8_t* rgb_d = NULL;
  HIPCHECK(hipMalloc((void**)&ws_d, sizeof(float) * num_ws * W_DIM));
  HIPCHECK(hipMalloc((void**)&rgb_d, (size_t)res * res * 3));
  HIPCHECK(hipMemcpy(ws_d, ws_h, sizeof(float) * num_ws * W_DIM, hipMemcpyHostToDevice));
  CHECK(maua_synth_render_rgb8(g_net, ws_d, NULL, NULL, 1, NULL, rgb_d));
  CHECK(maua_ctx_sync(ctx));
  uint8_t* rgb_h = (uint8_t*)malloc((size_t)res * res * 3);
  HIPCHECK(hipMemcpy(rgb_h, rgb_d, (size_t)res * res * 3, hipMemcpyDeviceToHost));
  FILE* f = fopen(out_path, "wb");
  if (!f) { perror(out_path); return 1; }
  fprintf(f, "P6\n%d %d\n255\n", res, res);
  fwrite(rgb_h, 1, (size_t)res * res * 3, f);
  fclose(f);
  printf("%s: %dx%d frame, %d synthesis layers, %d ws, library %s\n", out_path, res, res, maua_synth_num_layers(g_net), num_ws,
         maua_version());
  hipFree(ws_d); hipFree(rgb_d);
  maua_synth_destroy(g_net);
  maua_ctx_destroy(ctx);
  free(g_buf); free(ws_h); free(rgb_h);
  return 0;
}
