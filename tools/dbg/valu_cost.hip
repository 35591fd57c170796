// Marginal cost of one instruction of each kind when it sits between MFMAs in a single-wave-per-SIMD stream
// (the in-wave software-pipelining regime of mlp_fwd3.hip): 12 fillers per v_mfma_f32_32x32x16_f16, 8 independent
// registers round-robin.  cost = (ticks per MFMA - 32) / 12 once the MFMA shadow is exhausted.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define FILL(ASM) asm volatile(ASM : "+v"(f[q % 8]) : "v"(cst), "v"(f[(q + 3) % 8]))

template <int KIND, int NFILL>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  f16x8 va, vb;
  for (int i = 0; i < 8; ++i) { va[i] = (_Float16)(threadIdx.x * 0.001f + i); vb[i] = (_Float16)(i * 0.5f); }
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = threadIdx.x * 0.01f + i;
  float cst = 0.25f;
  asm volatile("" : "+v"(cst));
  __shared__ float sh[4096];
  sh[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const unsigned la = (unsigned)(size_t)sh + (threadIdx.x & 63) * 16;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, vb, acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NFILL; ++q) {
        if (KIND == 0) FILL("v_fma_f32 %0, %0, %1, %1");
        if (KIND == 1) FILL("v_sin_f32 %0, %0");
        if (KIND == 2) FILL("v_cvt_pk_f16_f32 %0, %0, %2");
        if (KIND == 3) FILL("v_cvt_f32_f16 %0, %0");
        if (KIND == 4) FILL("v_rndne_f32 %0, %0");
        if (KIND == 5) FILL("v_sub_f32 %0, %0, %1");
        if (KIND == 6) FILL("v_pk_add_f32 %0, %0, %1");  // placeholder: handled below with pairs
        if (KIND == 7) FILL("v_max3_f32 %0, %0, %1, %2");
        if (KIND == 8) FILL("v_accvgpr_write_b32 a100, %0");
        if (KIND == 9) FILL("v_accvgpr_read_b32 %0, a100");
        if (KIND == 10) FILL("v_cvt_f32_f16 %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1");
        if (KIND == 11) FILL("v_mul_f32 %0, %0, %1");
        if (KIND == 12) FILL("s_nop 0");
        if (KIND == 13) FILL("v_cos_f32 %0, %0\n\tv_fma_f32 %0, %0, %1, %1");  // trans -> dependent VALU
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + la;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int NFILL>
void run(const char* name, float* out, unsigned long long* cyc) {
  const int iters = 2000, blocks = 256;
  k<KIND, NFILL><<<blocks, 256>>>(out, cyc, 10);
  hipDeviceSynchronize();
  k<KIND, NFILL><<<blocks, 256>>>(out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < 256; ++i) mean += h[i]; mean /= 256;
  const double per = mean / (iters * 8.0);
  printf("%-28s fill=%2d : %6.1f ticks/MFMA  -> %5.2f ticks per filler beyond the shadow\n", name, NFILL, per, (per - 32.0) / NFILL);
}
#define BOTH(K, NAME) run<K, 6>(NAME, out, cyc); run<K, 12>(NAME, out, cyc); run<K, 24>(NAME, out, cyc);
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  BOTH(0, "v_fma_f32") BOTH(1, "v_sin_f32") BOTH(2, "v_cvt_pk_f16_f32") BOTH(3, "v_cvt_f32_f16") BOTH(10, "v_cvt_f32_f16 sdwa hi")
  BOTH(4, "v_rndne_f32") BOTH(5, "v_sub_f32") BOTH(11, "v_mul_f32") BOTH(7, "v_max3_f32") BOTH(8, "v_accvgpr_write_b32") BOTH(9, "v_accvgpr_read_b32")
  BOTH(12, "s_nop 0") BOTH(13, "v_cos + dependent v_fma (x2 instr)")
  return 0;
}
