// What does a launch boundary cost when the next kernel needs (almost) all of a CU's LDS, or the unified 512-register file?
// Sequence per trial: small, small, X, small, small (stream order); X = the same trivial kernel launched with 0 / 64 KiB /
// 156 KiB of dynamic LDS, or a kernel that touches AGPRs (one wave per SIMD).  Time per sequence with hipEvents over 2000 reps.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void small_k(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void lds_k(float* p) {
  extern __shared__ float sm[];
  sm[threadIdx.x] = p[threadIdx.x & 3];
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) p[1] = sm[5];
}
__global__ void __launch_bounds__(256) big_reg_k(float* p) {   // > 256 registers: AGPR half in use, one wave per SIMD
  float v[300];
#pragma unroll
  for (int i = 0; i < 300; ++i) v[i] = p[(i + threadIdx.x) & 7] * (float)i;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 300; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s) : "v"(v[i]));
  if (s == 12345.f) p[2] = s;
}
int main() {
  float* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipFuncSetAttribute((const void*)lds_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int lds[4] = {0, 64 * 1024 - 1024, 96 * 1024, 156 * 1024};
  for (int mode = 0; mode < 6; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a, st);
      for (int i = 0; i < 2000; ++i) {
        small_k<<<64, 256, 0, st>>>(d); small_k<<<64, 256, 0, st>>>(d);
        if (mode < 4) lds_k<<<256, 256, lds[mode], st>>>(d);
        else if (mode == 4) big_reg_k<<<256, 256, 0, st>>>(d);
        else small_k<<<256, 256, 0, st>>>(d);
        small_k<<<64, 256, 0, st>>>(d); small_k<<<64, 256, 0, st>>>(d);
      }
      hipEventRecord(b, st); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (rep) printf("mode %d (%s): %.2f us per 5-launch sequence\n", mode,
                      mode < 4 ? (mode == 0 ? "lds 0" : mode == 1 ? "lds 63 KiB" : mode == 2 ? "lds 96 KiB" : "lds 156 KiB") : mode == 4 ? "300 registers" : "small", ms * 1000.f / 2000.f);
    }
  }
  return 0;
}
