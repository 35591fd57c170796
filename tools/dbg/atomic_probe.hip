// Throughput of fp32 global atomics on the pattern a fused weight-gradient flush would produce: every workgroup adds
// 64 KiB slabs (one 128x128 fp32 matrix) onto the same 8 x 64 KiB accumulator set, coalesced (1 KiB per wave-instruction).
//   hipcc --offload-arch=gfx950 -O3 atomic_probe.hip -o bin/atomic_probe && bin/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_atomic(float* buf, int reps, int nmat, int copies) {
  const int tid = threadIdx.x;
  float* base = buf + (size_t)(blockIdx.x % copies) * nmat * 16384;
  for (int r = 0; r < reps; ++r)
    for (int m = 0; m < nmat; ++m) {
      float* p = base + m * 16384;
#pragma unroll 8
      for (int i = 0; i < 64; ++i) atomicAdd(p + i * 256 + tid, 1.0f);
    }
}
__global__ void k_store(float* buf, int reps, int nmat) {  // same bytes as plain stores into private slabs, for scale
  const int tid = threadIdx.x;
  float* base = buf + (size_t)blockIdx.x * nmat * 16384;
  for (int r = 0; r < reps; ++r)
    for (int m = 0; m < nmat; ++m) {
      float* p = base + m * 16384;
#pragma unroll 8
      for (int i = 0; i < 64; ++i) p[i * 256 + tid] = (float)r;
    }
}
int main() {
  const int nmat = 8;
  float* buf;
  hipMalloc(&buf, (size_t)2048 * nmat * 16384 * 4);
  hipMemset(buf, 0, (size_t)2048 * nmat * 16384 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int copies : {1, 8, 64}) {
    for (int wgs : {256, 1024, 4096}) {
      const int reps = 4096 / wgs * 2 > 0 ? 4096 / wgs * 2 : 1;
      k_atomic<<<wgs, 256>>>(buf, 1, nmat, copies);
      hipEventRecord(e0);
      k_atomic<<<wgs, 256>>>(buf, reps, nmat, copies);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double n = (double)wgs * reps * nmat * 16384;
      printf("atomic copies=%d wgs=%d reps=%d: %.3f ms  %.1f G atomics/s  %.2f TB/s payload\n", copies, wgs, reps, ms, n / ms / 1e6, n * 4 / ms / 1e9);
    }
  }
  {
    const int wgs = 2048, reps = 4;
    k_store<<<wgs, 256>>>(buf, 1, nmat);
    hipEventRecord(e0);
    k_store<<<wgs, 256>>>(buf, reps, nmat);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)wgs * reps * nmat * 16384;
    printf("plain stores wgs=%d reps=%d: %.3f ms  %.2f TB/s\n", wgs, reps, ms, n * 4 / ms / 1e9);
  }
  // exactness of the accumulated sums (copies = 1 slab 0)
  hipMemset(buf, 0, (size_t)nmat * 16384 * 4);
  k_atomic<<<1000, 256>>>(buf, 1, nmat, 1);
  std::vector<float> h(nmat * 16384);
  hipMemcpy(h.data(), buf, h.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (float v : h) bad += v != 1000.0f;
  printf("sum check: %d wrong of %zu\n", bad, h.size());
  return 0;
}
