// probe: two waves per SIMD, each alternating a block of 96 MFMAs (one layer GEMM of the F16X3 mode) with a block of
// NV vector instructions (the FiLM / sincos / split phase).  "lockstep": both waves of a SIMD run the same phase at
// the same time (what the per-layer workgroup barrier of sdf_mlp_kernel enforces); "anti-phase": the second wave of
// each SIMD starts with the vector block.  Same total work in both cases -> clock effects cancel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PK>
__device__ __forceinline__ void vblock(float (&q)[8], double (&p)[8], float m, float c, double M, double C) {
#pragma unroll
  for (int r = 0; r < (PK ? 60 : 120); ++r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(M), "v"(C));
      else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(q[i]) : "v"(m), "v"(c));
    }
  }
}
__device__ __forceinline__ void mblock(f32x16 (&c)[4], f16x8 A, f16x8 B) {
#pragma unroll
  for (int r = 0; r < 24; ++r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c[i], 0, 0, 0);
  }
}
// mode 0: lockstep, 1: anti-phase, 2: MFMA blocks only, 3: vector blocks only
template <int PK>
__global__ void __launch_bounds__(512) k(float* out, int iters, int mode, float a) {
  const int wave = threadIdx.x >> 6;
  f16x8 A, B;
  for (int i = 0; i < 8; ++i) { A[i] = (_Float16)(a + i); B[i] = (_Float16)(a - i); }
  f32x16 c[4] = {{0}, {0}, {0}, {0}};
  double p[8];
  float q[8];
  for (int i = 0; i < 8; ++i) q[i] = 0.01f * (a + i);
  for (int i = 0; i < 8; ++i) p[i] = __hiloint2double(__float_as_int(0.01f * (a + i)), __float_as_int(0.02f * (a + i)));
  const double M = __hiloint2double(__float_as_int(1.0001f), __float_as_int(0.9999f));
  const double C = __hiloint2double(__float_as_int(0.5f), __float_as_int(0.25f));
  const float m = 1.0001f * a, cc = 0.5f * a;
  const bool second = wave >= 4 && mode == 1;
  if (second) vblock<PK>(q, p, m, cc, M, C);
  for (int i = 0; i < iters; ++i) {
    if (mode != 3) mblock(c, A, B);
    if (mode != 2) vblock<PK>(q, p, m, cc, M, C);
  }
  if (second) mblock(c, A, B);
  float r = c[0][0] + c[1][1] + c[2][2] + c[3][3];
  for (int i = 0; i < 8; ++i) r += (float)__double2loint(p[i]) + (float)__double2hiint(p[i]) + q[i];
  out[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int PK>
void run(float* d) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000;
  const char* names[] = {"lockstep", "anti-phase", "MFMA blocks only", "vector blocks only"};
  for (int rep = 0; rep < 2; ++rep)
    for (int m = 0; m < 4; ++m) {
      k<PK><<<256, 512>>>(d, 10, m, 1.f);
      (void)hipEventRecord(e0);
      k<PK><<<256, 512>>>(d, iters, m, 1.f);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("%-14s %-20s %8.1f ns per (96 MFMA + %d %s) per wave pair\n", PK ? "v_pk_fma_f32" : "v_fma_f32", names[m],
             ms * 1e6 / iters, PK ? 480 : 960, PK ? "pk_fma" : "fma");
    }
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 512 * 4);
  run<0>(d); run<1>(d);
  return 0;
}
