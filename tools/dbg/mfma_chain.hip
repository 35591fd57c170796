// Cycles per v_mfma_f32_32x32x16_f16 for one wave per SIMD when the MFMAs rotate over NACC independent accumulators
// and NFILL independent VALU instructions sit between consecutive MFMAs (the in-wave software-pipelining pattern of
// mlp_fwd3.hip).  Answers: how far apart must two MFMAs on the SAME accumulator be once anything is issued between them?
//   hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o bin/mfma_chain && bin/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int NFILL>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  f16x8 va, vb;
  for (int i = 0; i < 8; ++i) { va[i] = (_Float16)(threadIdx.x * 0.001f + i); vb[i] = (_Float16)(i * 0.5f); }
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = threadIdx.x + i;
  float cst = 1.0001f;
  asm volatile("" : "+v"(cst));
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, vb, acc[u % NACC], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NFILL; ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[q % 8]) : "v"(cst));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int NFILL>
void run(float* out, unsigned long long* cyc) {
  const int iters = 2000, blocks = 256;
  k<NACC, NFILL><<<blocks, 256>>>(out, cyc, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<NACC, NFILL><<<blocks, 256>>>(out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < 256; ++i) mean += h[i]; mean /= 256;
  printf("acc=%d fill=%2d : %6.1f memtime-ticks/MFMA   %6.1f ns/MFMA (wall)\n", NACC, NFILL, mean / (iters * 12.0), ms * 1e6 / (iters * 12.0));
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  run<1, 0>(out, cyc); run<1, 1>(out, cyc); run<1, 4>(out, cyc); run<1, 8>(out, cyc); run<1, 12>(out, cyc);
  run<2, 0>(out, cyc); run<2, 1>(out, cyc); run<2, 4>(out, cyc); run<2, 8>(out, cyc); run<2, 12>(out, cyc);
  run<3, 0>(out, cyc); run<3, 1>(out, cyc); run<3, 4>(out, cyc); run<3, 8>(out, cyc); run<3, 12>(out, cyc);
  run<4, 4>(out, cyc); run<4, 8>(out, cyc); run<6, 8>(out, cyc);
  return 0;
}
