// probe: absolute error of the hardware v_sin_f32 / v_cos_f32 (input in revolutions) against double precision (gfx950)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* t, float* s, float* c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { s[i] = __builtin_amdgcn_sinf(t[i]); c[i] = __builtin_amdgcn_cosf(t[i]); }
}
int main() {
  const int n = 1 << 22;
  const double ranges[] = {0.25, 0.5, 4.0, 64.0};
  for (double R : ranges) {
    std::vector<float> t(n), s(n), c(n);
    for (int i = 0; i < n; ++i) t[i] = (float)(R * (2.0 * (i + 0.5) / n - 1.0));
    float *dt, *ds, *dc;
    (void)hipMalloc(&dt, n * 4); (void)hipMalloc(&ds, n * 4); (void)hipMalloc(&dc, n * 4);
    (void)hipMemcpy(dt, t.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dt, ds, dc, n);
    (void)hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0, rs = 0;
    for (int i = 0; i < n; ++i) {
      const double x = 6.283185307179586476925 * (double)t[i];
      es = fmax(es, fabs(s[i] - sin(x)));
      ec = fmax(ec, fabs(c[i] - cos(x)));
      if (fabs(sin(x)) > 1e-3) rs = fmax(rs, fabs(s[i] - sin(x)) / fabs(sin(x)));
    }
    printf("|t| <= %-5g rev: max abs err sin %.3e  cos %.3e   max rel err sin (|sin|>1e-3) %.3e\n", R, es, ec, rs);
    (void)hipFree(dt); (void)hipFree(ds); (void)hipFree(dc);
  }
  return 0;
}
