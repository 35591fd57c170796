// VALU issue rate of a wave that has the SIMD to itself vs two / four waves per SIMD (no MFMA in flight): 8 independent
// register chains, KIND 0 v_fma_f32, 1 v_mul_f32 (2 operand), 2 v_sin_f32, 3 v_pk_fma_f32, 4 mix of an epilogue
// (fma, fract, sin, cos, mul, mul).  Prints shader clocks per wave instruction and per SIMD instruction slot.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
  float f[8];
  f32x2 p[8];
  for (int i = 0; i < 8; ++i) { f[i] = threadIdx.x * 0.01f + i; p[i] = f32x2{f[i], f[i] + 1.f}; }
  float cst = 0.25f;
  asm volatile("" : "+v"(cst));
  extern __shared__ float dyn[];
  if (iters < 0) dyn[threadIdx.x] = cst;  // (keeps the allocation: it is what sets the occupancy)
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[q]) : "v"(cst));
        if (KIND == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[q]) : "v"(cst));
        if (KIND == 2) asm volatile("v_sin_f32 %0, %0" : "+v"(f[q]));
        if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[q]) : "v"(p[(q + 3) & 7]));
        if (KIND == 4) {
          asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fract_f32 %0, %0" : "+v"(f[q]) : "v"(cst));
          float s, c;
          asm volatile("v_sin_f32 %0, %2\n\tv_cos_f32 %1, %2" : "=&v"(s), "=&v"(c) : "v"(f[q]));
          asm volatile("v_mul_f32 %0, %1, %2\n\tv_fma_f32 %0, %0, %3, %1" : "+v"(f[q]) : "v"(s), "v"(c), "v"(cst));
        }
        if (KIND == 5) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[0]) : "v"(cst));  // ONE dependent chain
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += f[i] + p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND>
void run(const char* name, int per_slot, float* out, unsigned long long* cyc) {
  // occupancy through the LDS allocation: 256-thread workgroups, 160 KB / wps of LDS each -> wps waves per SIMD on every CU;
  // 16 workgroups per CU slot so that placement imbalance averages out
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int iters = 500;
    const int lds = (160 * 1024) / wps - 1024;
    const int blocks = 256 * wps * 8;
    hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), lds, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), lds, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 16 * 8 * per_slot;         // instructions per wave
    const double per_simd = n * blocks * 4 / 1024.0;              // instructions per SIMD
    printf("%-18s waves/SIMD %d: %.2f ticks per wave instruction (wave 0), %.3f ns per SIMD instruction slot\n", name, wps, c / n,
           ms * 1e6 / per_simd);
  }
}
int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 8192 * 256 * 4); (void)hipMalloc(&cyc, 8);
  run<0>("v_fma_f32", 1, out, cyc);
  run<1>("v_mul_f32", 1, out, cyc);
  run<2>("v_sin_f32", 1, out, cyc);
  run<3>("v_pk_fma_f32", 1, out, cyc);
  run<4>("epilogue mix x6", 6, out, cyc);
  run<5>("fma one chain", 1, out, cyc);
  return 0;
}
