// What does an LDS read cost a single-wave-per-SIMD MFMA stream?  NREAD ds_read_b128 (lane-linear, conflict-free, each into
// its own 4 VGPRs) per v_mfma_f32_32x32x16_f16, one s_waitcnt lgkmcnt(0) per MFMA; WAVES waves per workgroup (= per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NREAD, int WIDTH, int PF>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = i;
  __syncthreads();
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  f16x8 va, vb;
  for (int i = 0; i < 8; ++i) { va[i] = (_Float16)(threadIdx.x * 0.001f + i); vb[i] = (_Float16)(i * 0.5f); }
  f32x4 d[8];
  for (int i = 0; i < 8; ++i) d[i] = f32x4{0, 0, 0, 0};
  unsigned addr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, vb, acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NREAD; ++q) {
        if (WIDTH == 16) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[(q + 4 * (u & 1)) % 8]) : "v"(addr), "n"((q % 4) * 1024 + 4096));
        if (WIDTH == 8) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(*reinterpret_cast<double*>(&d[(q + 4 * (u & 1)) % 8])) : "v"(addr), "n"((q % 4) * 1024 + 4096));
      }
      if (PF) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NREAD) : "memory");  // previous MFMA's reads back, this one's in flight
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int i = 0; i < 8; ++i) s += d[i][0] + d[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NREAD, int WIDTH, int PF>
void run(int waves, float* out, unsigned long long* cyc) {
  const int iters = 2000, blocks = 256;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<NREAD, WIDTH, PF>), hipFuncAttributeMaxDynamicSharedMemorySize, 140000);
  k<NREAD, WIDTH, PF><<<blocks, 64 * waves, 140000>>>(out, cyc, 10);   // 140 KB LDS: one workgroup per CU
  hipDeviceSynchronize();
  k<NREAD, WIDTH, PF><<<blocks, 64 * waves, 140000>>>(out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < 256; ++i) mean += h[i]; mean /= 256;
  printf("waves/CU=%d  %d x ds_read_b%-3d per MFMA %s: %6.1f ticks/MFMA\n", waves, NREAD, WIDTH * 8, PF ? "(prefetched one MFMA ahead)" : "(waited at once)            ", mean / (iters * 8.0));
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  for (int w : {1, 4, 8}) {
    run<0, 16, 0>(w, out, cyc); run<1, 16, 0>(w, out, cyc); run<2, 16, 0>(w, out, cyc);
    run<1, 16, 1>(w, out, cyc); run<2, 16, 1>(w, out, cyc); run<3, 16, 1>(w, out, cyc); run<4, 16, 1>(w, out, cyc);
    run<2, 8, 1>(w, out, cyc); run<4, 8, 1>(w, out, cyc);
  }
  return 0;
}
