/* Oracle restatement (plain C, test infrastructure only) of the reference's only native component:
 * maua/audiovisual/audioreactive/selfsupervised/features/efficient_quantile/efficient_quantile.cpp:86-206,
 * as the Python wrapper calls it (__init__.py:6-7): one float32 quantile q, NaNs ignored, interpolation
 * mode 3 ("midpoint").
 *
 *   qf  = (double)(float)q                      (.cpp:113 — the float32 tensor promoted to double)
 *   lo  = (int64) trunc(qf * (n-1)),  hi = (int64) ceil(qf * (n-1))        (.cpp:158-160)
 *   x_(lo), x_(hi) = order statistics of the NaN-free data                 (.cpp:8-34 nth_element)
 *   result = (float) lerp((double)x_(lo), (double)x_(hi), hi > lo ? 0.5 : 0.0)   (.cpp:73-83)
 * Returns NaN for empty / all-NaN input (.cpp:139-143).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int cmp_float(const void* a, const void* b) {
  float x = *(const float*)a, y = *(const float*)b;
  return (x > y) - (x < y);
}

/* also reports the two order-statistic indices (they must match the HIP path bit-for-bit) */
float maua_oracle_quantile_mid(const float* x, int64_t n, float q, int64_t* lo_out, int64_t* hi_out) {
  float* buf = (float*)malloc((size_t)(n > 0 ? n : 1) * sizeof(float));
  int64_t m = 0;
  for (int64_t i = 0; i < n; i++)
    if (!isnan(x[i])) buf[m++] = x[i];
  if (m <= 0) {
    free(buf);
    if (lo_out) *lo_out = -1;
    if (hi_out) *hi_out = -1;
    return NAN;
  }
  qsort(buf, (size_t)m, sizeof(float), cmp_float);
  double qf = (double)q;
  double pos = qf * (double)(m - 1);
  int64_t lo = (int64_t)pos; /* truncation toward zero, pos >= 0 */
  int64_t hi = (int64_t)ceil(pos);
  double ylo = (double)buf[lo], yhi = (double)buf[hi];
  double w = hi > lo ? 0.5 : 0.0;
  /* torch::lerp(a, b, w) for w < 0.5 ... : a + w * (b - a); for w >= 0.5: b - (b - a) * (1 - w) */
  double r = (w < 0.5) ? ylo + w * (yhi - ylo) : yhi - (yhi - ylo) * (1.0 - w);
  free(buf);
  if (lo_out) *lo_out = lo;
  if (hi_out) *hi_out = hi;
  return (float)r;
}
