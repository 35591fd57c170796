// Read-only streaming ceiling: each wave reads 1 KiB per instruction (64 lanes x 16 B), DEPTH loads in flight, over a
// buffer far larger than the Infinity Cache.   hipcc --offload-arch=gfx950 -O3 hbm_read.hip -o bin/hbm_read && bin/hbm_read
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH, bool NT>
__global__ void __launch_bounds__(256) k(const f32x4* __restrict__ src, float* out, long long n_chunks_per_wave, long long stride_waves) {
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // wave w reads chunks w, w + stride, w + 2 stride ... (a chunk = 64 x 16 B = 1 KiB): neighbouring waves, neighbouring KiB
  for (long long c = 0; c < n_chunks_per_wave; c += DEPTH) {
    f32x4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const f32x4* p = src + ((c + d) * stride_waves + wave) * 64 + lane;
      v[d] = NT ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += v[d];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

template <int DEPTH, bool NT>
void run(const f32x4* src, float* out, size_t bytes, int wgs_per_cu) {
  const long long waves = 256LL * wgs_per_cu * 4;
  const long long chunks = bytes / 1024 / waves / DEPTH * DEPTH;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<DEPTH, NT><<<256 * wgs_per_cu, 256>>>(src, out, chunks, waves);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) k<DEPTH, NT><<<256 * wgs_per_cu, 256>>>(src, out, chunks, waves);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("depth %2d  %s  %d workgroups/CU : %.2f TB/s\n", DEPTH, NT ? "nt " : "   ", wgs_per_cu, chunks * waves * 1024.0 * 5 / (ms * 1e-3) / 1e12);
}

int main() {
  const size_t bytes = 8ull << 30;
  f32x4* src; float* out;
  (void)hipMalloc(&src, bytes); (void)hipMalloc(&out, 64);
  (void)hipMemset(src, 0, bytes);
  run<4, false>(src, out, bytes, 8); run<8, false>(src, out, bytes, 8); run<16, false>(src, out, bytes, 8); run<16, false>(src, out, bytes, 4);
  run<16, false>(src, out, bytes, 2); run<32, false>(src, out, bytes, 2); run<16, true>(src, out, bytes, 8); run<16, true>(src, out, bytes, 4);
  run<8, true>(src, out, bytes, 8); run<32, true>(src, out, bytes, 3);
  return 0;
}
