// probe: can one wave's v_pk_fma stream issue while another wave of the same SIMD keeps the matrix pipe busy? (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
// mode bit0: waves 0..3 run MFMAs; bit1: waves 4..7 run VALU; bit2: same-wave interleave (all 8 waves do both, interleaved)
template <int V>
__device__ __forceinline__ void valu16(f2& p0, f2& p1, f2& p2, f2& p3, f2& p4, f2& p5, f2& p6, f2& p7, f2 M, f2 C) {
  if (V == 0) {
    p0 = __builtin_elementwise_fma(p0, M, C); p1 = __builtin_elementwise_fma(p1, M, C); p2 = __builtin_elementwise_fma(p2, M, C); p3 = __builtin_elementwise_fma(p3, M, C);
    p4 = __builtin_elementwise_fma(p4, M, C); p5 = __builtin_elementwise_fma(p5, M, C); p6 = __builtin_elementwise_fma(p6, M, C); p7 = __builtin_elementwise_fma(p7, M, C);
  } else if (V == 1) {
#define F(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(M[0]), "v"(C[0]))
    F(p0[0]); F(p1[0]); F(p2[0]); F(p3[0]); F(p4[0]); F(p5[0]); F(p6[0]); F(p7[0]);
#undef F
  } else {
#define X(x) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(M[0]))
    X(p0[0]); X(p1[0]); X(p2[0]); X(p3[0]); X(p4[0]); X(p5[0]); X(p6[0]); X(p7[0]);
#undef X
  }
}
template <int V>
__global__ void __launch_bounds__(512) k(float* out, int iters, int mode, float a) {
  const int wave = threadIdx.x >> 6;
  f16x8 A, B;
  for (int i = 0; i < 8; ++i) { A[i] = (_Float16)(a + i); B[i] = (_Float16)(a - i); }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  f2 p0 = {a, a + 1}, p1 = {a + 2, a + 3}, p2 = {a + 4, a + 5}, p3 = {a + 6, a + 7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
  const f2 M = {1.0001f, 0.9999f}, C = {0.5f, 0.25f};
  const bool do_m = (mode & 4) || ((mode & 1) && wave < 4);
  const bool do_v = (mode & 4) || ((mode & 2) && wave >= 4);
  if (mode & 4) {
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c0, 0, 0, 0);
      p0 = __builtin_elementwise_fma(p0, M, C); p1 = __builtin_elementwise_fma(p1, M, C); p2 = __builtin_elementwise_fma(p2, M, C); p3 = __builtin_elementwise_fma(p3, M, C);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c1, 0, 0, 0);
      p4 = __builtin_elementwise_fma(p4, M, C); p5 = __builtin_elementwise_fma(p5, M, C); p6 = __builtin_elementwise_fma(p6, M, C); p7 = __builtin_elementwise_fma(p7, M, C);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c2, 0, 0, 0);
      p0 = __builtin_elementwise_fma(p0, M, C); p1 = __builtin_elementwise_fma(p1, M, C); p2 = __builtin_elementwise_fma(p2, M, C); p3 = __builtin_elementwise_fma(p3, M, C);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c3, 0, 0, 0);
      p4 = __builtin_elementwise_fma(p4, M, C); p5 = __builtin_elementwise_fma(p5, M, C); p6 = __builtin_elementwise_fma(p6, M, C); p7 = __builtin_elementwise_fma(p7, M, C);
    }
  } else if (do_m) {
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c3, 0, 0, 0);
    }
  } else if (do_v) {
    for (int i = 0; i < iters; ++i) {
      valu16<V>(p0, p1, p2, p3, p4, p5, p6, p7, M, C);
      valu16<V>(p0, p1, p2, p3, p4, p5, p6, p7, M, C);
    }
  }
  float r = c0[0] + c1[1] + c2[2] + c3[3] + p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[1] + p6[0] + p7[1];
  out[blockIdx.x * 512 + threadIdx.x] = r;
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 512 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  const char* names[] = {"", "MFMA only (waves 0-3: 4 MFMA/iter)", "VALU only (waves 4-7: 16 pk_fma/iter)", "MFMA waves + VALU waves together", "same-wave interleave (4 MFMA + 16 pk_fma per iter, 8 waves)"};
  int modes[] = {1, 2, 3};
  for (int v = 0; v < 3; ++v) {
    printf("--- VALU flavour %s\n", v == 0 ? "v_pk_fma_f32" : (v == 1 ? "v_fma_f32" : "v_xor_b32"));
    for (int m : modes) {
      auto kk = v == 0 ? k<0> : (v == 1 ? k<1> : k<2>);
      kk<<<256, 512>>>(d, 10, m, 1.f);
      (void)hipEventRecord(e0);
      kk<<<256, 512>>>(d, iters, m, 1.f);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("%-62s %.3f ms  (%.1f ns/iter)\n", names[m], ms, ms * 1e6 / iters);
    }
  }
  return 0;
}
