// round trip of the 24-bit slot formats of csrc/mlp_bwd.hip (pack24f / unpack24f, pack24q / unpack24q)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
__device__ u32x3 pack24f(f32x4 v) {
  // (element copies first: __builtin_bit_cast applied to a vector-element lvalue read element 0 four times -- hipcc 7.2)
  const float x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3];
  const unsigned b0 = __builtin_bit_cast(unsigned, x0) + 0x80u, b1 = __builtin_bit_cast(unsigned, x1) + 0x80u;
  const unsigned b2 = __builtin_bit_cast(unsigned, x2) + 0x80u, b3 = __builtin_bit_cast(unsigned, x3) + 0x80u;
  return u32x3{__builtin_amdgcn_perm(b1, b0, 0x05030201u), __builtin_amdgcn_perm(b2, b1, 0x06050302u), __builtin_amdgcn_perm(b3, b2, 0x07060503u)};
}
__device__ f32x4 unpack24f(u32x3 d) {
  const unsigned b0 = __builtin_amdgcn_perm(0u, d[0], 0x0201000cu), b1 = __builtin_amdgcn_perm(d[1], d[0], 0x0504030cu);
  const unsigned b2 = __builtin_amdgcn_perm(d[2], d[1], 0x0403020cu), b3 = __builtin_amdgcn_perm(0u, d[2], 0x0302010cu);
  return f32x4{__builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1), __builtin_bit_cast(float, b2), __builtin_bit_cast(float, b3)};
}
__device__ u32x3 pack24q(f32x4 r) {
  const unsigned q0 = (unsigned)(r[0] * 16777216.f), q1 = (unsigned)(r[1] * 16777216.f), q2 = (unsigned)(r[2] * 16777216.f), q3 = (unsigned)(r[3] * 16777216.f);
  return u32x3{__builtin_amdgcn_perm(q1, q0, 0x04020100u), __builtin_amdgcn_perm(q2, q1, 0x05040201u), __builtin_amdgcn_perm(q3, q2, 0x06050402u)};
}
__device__ f32x4 unpack24q(u32x3 d) {
  const unsigned q0 = __builtin_amdgcn_perm(0u, d[0], 0x0c020100u), q1 = __builtin_amdgcn_perm(d[1], d[0], 0x0c050403u);
  const unsigned q2 = __builtin_amdgcn_perm(d[2], d[1], 0x0c040302u), q3 = __builtin_amdgcn_perm(0u, d[2], 0x0c030201u);
  constexpr float S = 1.0f / 16777216.f;
  return f32x4{(float)q0 * S, (float)q1 * S, (float)q2 * S, (float)q3 * S};
}
__global__ void k(const float* in, float* outf, float* outq, u32x3* buf) {
  const int i = threadIdx.x;
  f32x4 v = {in[4 * i], in[4 * i + 1], in[4 * i + 2], in[4 * i + 3]};
  buf[i] = pack24f(v);
  f32x4 u = unpack24f(buf[i]);
  f32x4 r = {v[0] - floorf(v[0]), v[1] - floorf(v[1]), v[2] - floorf(v[2]), v[3] - floorf(v[3])};
  f32x4 w = unpack24q(pack24q(r));
  for (int c = 0; c < 4; ++c) { outf[4 * i + c] = u[c]; outq[4 * i + c] = w[c] - r[c]; }
}
int main() {
  float h[256], *d, *of, *oq; u32x3* b;
  for (int i = 0; i < 256; ++i) h[i] = (i % 7 - 3) * 1.2345f * powf(1.7f, (float)(i % 23) - 11) + 0.001f * i;
  hipMalloc(&d, 1024); hipMalloc(&of, 1024); hipMalloc(&oq, 1024); hipMalloc(&b, 64 * 16);
  hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, of, oq, b);
  float hf[256], hq[256];
  hipMemcpy(hf, of, 1024, hipMemcpyDeviceToHost); hipMemcpy(hq, oq, 1024, hipMemcpyDeviceToHost);
  double worst = 0, worstq = 0;
  for (int i = 0; i < 256; ++i) { double e = fabs(hf[i] - h[i]) / fmax(1e-30, fabs(h[i])); if (e > worst) worst = e; if (fabs(hq[i]) > worstq) worstq = fabs(hq[i]); }
  printf("float24 worst relative error %.3e (expect <= 2^-17 = 7.6e-6)   phase24 worst abs error %.3e (expect < 6e-8)\n", worst, worstq);
  printf("samples: %g -> %g, %g -> %g, %g -> %g\n", h[5], hf[5], h[6], hf[6], h[7], hf[7]);
  return 0;
}
