// One wave per SIMD: 3 x v_mfma_f32_32x32x16_f16 + 2 x ds_read_b128 (the next step's A fragments) + NFILL VALU per step,
// with the accumulators in VGPRs (what hipcc picks for the MLP kernels) or pinned to AGPRs (inline asm, "+a").
// Question: do the LDS return and the VALU work hide behind the MFMAs, and does the accumulator file matter?
//   hipcc --offload-arch=gfx950 -O3 mfma_lds.hip -o bin/mfma_lds && bin/mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <bool AGPR>
__device__ __forceinline__ void mfma(f32x16& acc, f32x4 a, f32x4 b) {
  if constexpr (AGPR) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  } else {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  }
}

template <bool AGPR, int NREAD, int NFILL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float*>(lds)[i] = 0.001f * (i & 255);
  __syncthreads();
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  f32x4 b = {1.f, 2.f, 3.f, 4.f};
  f32x4 ah = {0.f, 0.f, 0.f, 0.f}, al = ah, nh = ah, nl = ah;
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = threadIdx.x + i;
  float cst = 1.0001f;
  asm volatile("" : "+v"(cst), "+v"(b));
  int addr = (threadIdx.x & 63) * 16;
  asm volatile("" : "+v"(addr));
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nh), "+v"(nl));
      ah = nh; al = nl;
      if (NREAD >= 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(nh) : "v"(addr), "n"(s * 2048));
      if (NREAD >= 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(nl) : "v"(addr), "n"(s * 2048 + 1024));
      mfma<AGPR>(acc, al, b);
#pragma unroll
      for (int q = 0; q < NFILL / 3; ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[q % 8]) : "v"(cst));
      mfma<AGPR>(acc, ah, b);
#pragma unroll
      for (int q = 0; q < NFILL / 3; ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[(q + 3) % 8]) : "v"(cst));
      mfma<AGPR>(acc, ah, b);
#pragma unroll
      for (int q = 0; q < NFILL - 2 * (NFILL / 3); ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[(q + 5) % 8]) : "v"(cst));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  if constexpr (AGPR) {
    f32x16 v;
    asm volatile("s_nop 7\n s_nop 7\n" ::);
    for (int r = 0; r < 16; ++r) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[r]) : "a"(acc[r]));
    for (int r = 0; r < 16; ++r) s += v[r];
  } else {
    for (int r = 0; r < 16; ++r) s += acc[r];
  }
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + nh[0] + nl[0];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <bool AGPR, int NREAD, int NFILL>
void run(float* out, unsigned long long* cyc) {
  const int iters = 1000, blocks = 256;
  auto kk = k<AGPR, NREAD, NFILL>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kk), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  kk<<<blocks, 256, 65536>>>(out, cyc, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  kk<<<blocks, 256, 65536>>>(out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < 256; ++i) mean += h[i]; mean /= 256;
  printf("acc in %s, %d ds_read_b128 + %2d VALU per 3 MFMAs: %6.1f ticks/step   %6.1f ns/step (wall)\n", AGPR ? "AGPR" : "VGPR", NREAD,
         NFILL, mean / (iters * 8.0), ms * 1e6 / (iters * 8.0));
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  run<false, 0, 0>(out, cyc); run<false, 2, 0>(out, cyc); run<false, 0, 12>(out, cyc); run<false, 2, 12>(out, cyc); run<false, 2, 18>(out, cyc);
  run<true, 0, 0>(out, cyc); run<true, 2, 0>(out, cyc); run<true, 0, 12>(out, cyc); run<true, 2, 12>(out, cyc); run<true, 2, 18>(out, cyc);
  run<false, 1, 0>(out, cyc); run<true, 1, 0>(out, cyc);
  return 0;
}
