// Accuracy of the hardware sine (v_sin_f32, input in revolutions) against double precision, on the argument range of the noise Loop
// module's outer sine (|x| < 16).   hipcc --offload-arch=gfx950 -O3 scripts/ubench/hw_sin.hip -o scripts/ubench/_bin/hw_sin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* y, float* y2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  y[i] = __builtin_amdgcn_sinf(x[i] * 0.15915494309189535f);                        // v_mul + v_sin
  {  // reduction to |r| <= 0.5 revolutions with an exact product: k = rint(x C), r = fma(x, C, -k) + x C_lo
    const float C = 0.15915494309189535f, Clo = (float)(0.15915494309189533576888 - (double)0.15915494309189535f);
    const float kk = rintf(x[i] * C);
    float r = fmaf(x[i], C, -kk);
    r = fmaf(x[i], Clo, r);
    y2[i] = __builtin_amdgcn_sinf(r);
  }
}
int main() {
  const int n = 1 << 22;
  std::vector<float> hx(n), hy(n), hy2(n);
  for (int i = 0; i < n; i++) hx[i] = -16.f + 32.f * (float)i / (float)(n - 1);
  float *dx, *dy, *dy2;
  hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&dy2, n * 4);
  hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dy, dy2, n);
  hipMemcpy(hy.data(), dy, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hy2.data(), dy2, n * 4, hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0, es = 0;
  for (int i = 0; i < n; i++) {
    const double r = std::sin((double)hx[i]);
    e1 = std::fmax(e1, std::fabs(hy[i] - r));
    e2 = std::fmax(e2, std::fabs(hy2[i] - r));
    es = std::fmax(es, std::fabs((double)sinf(hx[i]) - r));
  }
  printf("max abs error on [-16, 16]: v_sin(x / 2pi) %.3e   v_sin(compensated reduction) %.3e   host sinf %.3e\n", e1, e2, es);
  return 0;
}
