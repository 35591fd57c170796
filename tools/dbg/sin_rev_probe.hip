// v_sin_f32 / v_cos_f32 take REVOLUTIONS.  Is feeding an unreduced phase x (|x| <= 256) the same as feeding the exactly
// reduced x - rint(x)?  (fract of an fp32 number is exact, so the hardware's own range reduction should lose nothing.)
// Also checks the v_fma_mix_f32 form of the fp16 hi/lo split against the cvt / sub form.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const float* x, float* s0, float* s1, float* c0, float* c1, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = x[i], r = v - __builtin_rintf(v);
  s0[i] = __builtin_amdgcn_sinf(v); s1[i] = __builtin_amdgcn_sinf(r);
  c0[i] = __builtin_amdgcn_cosf(v); c1[i] = __builtin_amdgcn_cosf(r);
}
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__global__ void k2(const float* x, unsigned* hi, unsigned* lo_ref, unsigned* lo_mix, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  float a = x[2 * i], b = x[2 * i + 1];
  f16x2 h = {(_Float16)a, (_Float16)b};
  f16x2 l = {(_Float16)(a - (float)h[0]), (_Float16)(b - (float)h[1])};
  unsigned hu = __builtin_bit_cast(unsigned, h);
  float la, lb;
  asm volatile("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
               "v_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(la), "=&v"(lb) : "v"(hu), "v"(a), "v"(b));
  f16x2 lm = {(_Float16)la, (_Float16)lb};
  hi[i] = hu; lo_ref[i] = __builtin_bit_cast(unsigned, l); lo_mix[i] = __builtin_bit_cast(unsigned, lm);
}
int main() {
  const int n = 1 << 22;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = ((rand() / (float)RAND_MAX) * 2 - 1) * ((i & 3) == 0 ? 250.f : (i & 3) == 1 ? 40.f : (i & 3) == 2 ? 4.f : 0.6f);
  float *x, *s0, *s1, *c0, *c1;
  hipMalloc(&x, n * 4); hipMalloc(&s0, n * 4); hipMalloc(&s1, n * 4); hipMalloc(&c0, n * 4); hipMalloc(&c1, n * 4);
  hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(x, s0, s1, c0, c1, n);
  std::vector<float> a(n), b(n), c(n), d(n);
  hipMemcpy(a.data(), s0, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), s1, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), c0, n * 4, hipMemcpyDeviceToHost); hipMemcpy(d.data(), c1, n * 4, hipMemcpyDeviceToHost);
  int diff = 0; double worst = 0, werr = 0;
  for (int i = 0; i < n; ++i) {
    if (a[i] != b[i] || c[i] != d[i]) ++diff;
    worst = fmax(worst, fmax(fabs((double)a[i] - b[i]), fabs((double)c[i] - d[i])));
    double xr = (double)h[i] - nearbyint((double)h[i]);
    werr = fmax(werr, fabs((double)a[i] - sin(2 * M_PI * xr)));
  }
  printf("unreduced vs reduced: %d of %d differ, worst |diff| %.3e; worst |v_sin(x) - sin(2 pi x)| %.3e\n", diff, n, worst, werr);
  unsigned *hi, *lr, *lm; hipMalloc(&hi, n * 2); hipMalloc(&lr, n * 2); hipMalloc(&lm, n * 2);
  k2<<<n / 512, 256>>>(x, hi, lr, lm, n);
  std::vector<unsigned> r1(n / 2), r2(n / 2);
  hipMemcpy(r1.data(), lr, n * 2, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), lm, n * 2, hipMemcpyDeviceToHost);
  int d2 = 0; for (int i = 0; i < n / 2; ++i) d2 += r1[i] != r2[i];
  printf("fma_mix lo limbs vs cvt/sub lo limbs: %d of %d dwords differ\n", d2, n / 2);
  return 0;
}
