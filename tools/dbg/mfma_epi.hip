// One wave per SIMD; per step 3 x v_mfma_f32_32x32x16_f16 + 2 ds_read_b128 (next A fragments) + ONE epilogue pair written
// the way mlp_fwd3.hip writes it, assembled piece by piece (template MODE bits) to find which piece breaks the overlap:
//   1  FiLM fma x2 (3 distinct source registers)          2  v_fract x2                 4  v_sin x2
//   8  v_cvt_pk + 2 v_fma_mix + v_cvt_pk (fp16 split)      16 the split results become B operands of later MFMAs
//   32 v_accvgpr_write x2 (park)                           64 the FiLM fma reads registers an EARLIER MFMA wrote
//   hipcc --offload-arch=gfx950 -O3 mfma_epi.hip -o bin/mfma_epi && bin/mfma_epi
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float*>(lds)[i] = 0.37f * ((i * 2654435761u) >> 20);
  __syncthreads();
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.01f * r + threadIdx.x; }
  f32x4 bh[8], bl[8];
  for (int s = 0; s < 8; ++s) { bh[s] = f32x4{1.f + s, 2.f, 3.f, 4.f}; bl[s] = f32x4{0.5f, 0.25f + s, 0.125f, 1.f}; }
  f32x4 nh = {0.f, 0.f, 0.f, 0.f}, nl = nh, ah, al;
  f32x4 ra = {1.1f, 1.2f, 1.3f, 1.4f}, rb = {0.1f, 0.2f, 0.3f, 0.4f};
  float park0 = 0.f, park1 = 0.f, src0 = threadIdx.x * 0.01f, src1 = threadIdx.x * 0.02f;
  asm volatile("" : "+v"(ra), "+v"(rb), "+v"(src0), "+v"(src1));
  int addr = (threadIdx.x & 63) * 16;
  asm volatile("" : "+v"(addr));
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nh), "+v"(nl));
      ah = nh; al = nl;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(nh) : "v"(addr), "n"(s * 2048));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(nl) : "v"(addr), "n"(s * 2048 + 1024));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(al), "v"(bh[s]));
      float x0 = (MODE & 64) ? acc1[2 * (s & 7)] : src0, x1 = (MODE & 64) ? acc1[2 * (s & 7) + 1] : src1;
      float r0 = x0, r1 = x1;
      if (MODE & 1) {
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(ra[s & 3]), "v"(x0), "v"(rb[s & 3]));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r1) : "v"(ra[(s + 1) & 3]), "v"(x1), "v"(rb[(s + 1) & 3]));
      }
      if (MODE & 2) {
        asm volatile("v_fract_f32 %0, %0" : "+v"(r0));
        asm volatile("v_fract_f32 %0, %0" : "+v"(r1));
      }
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(ah), "v"(bl[s]));
      float s0 = r0, s1 = r1;
      if (MODE & 4) {
        asm volatile("v_sin_f32 %0, %1" : "=v"(s0) : "v"(r0));
        asm volatile("v_sin_f32 %0, %1" : "=v"(s1) : "v"(r1));
      }
      if (MODE & 32) {
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(park0) : "v"(r0));
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(park1) : "v"(r1));
      }
      unsigned hi = 0, lo = 0;
      if (MODE & 8) {
        float q0, q1;
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(s0), "v"(s1));
        asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(q0) : "v"(hi), "v"(s0));
        asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(q1) : "v"(hi), "v"(s1));
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(q0), "v"(q1));
      }
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(ah), "v"(bh[s]));
      if (MODE & 16) {  // next iteration's B operands (different k-step than the one in flight)
        bh[(s + 4) & 7][s & 3] = __builtin_bit_cast(float, hi);
        bl[(s + 4) & 7][s & 3] = __builtin_bit_cast(float, lo);
      } else {
        src0 += __builtin_bit_cast(float, hi & 1u);
        src1 += __builtin_bit_cast(float, lo & 1u);
      }
      if (MODE & 32) asm volatile("" ::"a"(park0), "a"(park1));
    }
    if (MODE & 64) {  // refresh the "earlier" accumulator now and then, as a completed block would
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(ah), "v"(bh[0]));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sum = 0.f;
  for (int r = 0; r < 16; ++r) sum += acc0[r] + acc1[r];
  for (int s = 0; s < 8; ++s) sum += bh[s][0] + bl[s][1];
  out[blockIdx.x * 256 + threadIdx.x] = sum + nh[0] + nl[0] + src0 + src1;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// The same work with the fp16 split delayed by one step: a step splits the sines of the PREVIOUS step (all its inputs
// are a step old) and issues fma / fract / sin / park of its own pair -- no instruction consumes a result that is less
// than ~10 instructions old.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_staged(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float*>(lds)[i] = 0.37f * ((i * 2654435761u) >> 20);
  __syncthreads();
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.01f * r + threadIdx.x; }
  f32x4 bh[8], bl[8];
  for (int s = 0; s < 8; ++s) { bh[s] = f32x4{1.f + s, 2.f, 3.f, 4.f}; bl[s] = f32x4{0.5f, 0.25f + s, 0.125f, 1.f}; }
  f32x4 nh = {0.f, 0.f, 0.f, 0.f}, nl = nh, ah, al;
  f32x4 ra = {1.1f, 1.2f, 1.3f, 1.4f}, rb = {0.1f, 0.2f, 0.3f, 0.4f};
  float park0 = 0.f, park1 = 0.f, ps0 = 0.1f, ps1 = 0.2f;
  asm volatile("" : "+v"(ra), "+v"(rb), "+v"(ps0), "+v"(ps1));
  int addr = (threadIdx.x & 63) * 16;
  asm volatile("" : "+v"(addr));
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nh), "+v"(nl));
      ah = nh; al = nl;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(nh) : "v"(addr), "n"(s * 2048));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(nl) : "v"(addr), "n"(s * 2048 + 1024));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(al), "v"(bh[s]));
      unsigned hi, lo;
      float q0, q1, r0, r1, s0, s1;
      const float x0 = acc1[2 * (s & 7)], x1 = acc1[2 * (s & 7) + 1];
      asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(ps0), "v"(ps1));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(ra[s & 3]), "v"(x0), "v"(rb[s & 3]));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r1) : "v"(ra[(s + 1) & 3]), "v"(x1), "v"(rb[(s + 1) & 3]));
      asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(q0) : "v"(hi), "v"(ps0));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(ah), "v"(bl[s]));
      asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(q1) : "v"(hi), "v"(ps1));
      asm volatile("v_fract_f32 %0, %0" : "+v"(r0));
      asm volatile("v_fract_f32 %0, %0" : "+v"(r1));
      asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(park0) : "v"(r0));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(ah), "v"(bh[s]));
      asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(q0), "v"(q1));
      asm volatile("v_sin_f32 %0, %1" : "=v"(s0) : "v"(r0));
      asm volatile("v_sin_f32 %0, %1" : "=v"(s1) : "v"(r1));
      asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(park1) : "v"(r1));
      ps0 = s0; ps1 = s1;
      bh[(s + 4) & 7][s & 3] = __builtin_bit_cast(float, hi);
      bl[(s + 4) & 7][s & 3] = __builtin_bit_cast(float, lo);
      asm volatile("" ::"a"(park0), "a"(park1));
    }
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(ah), "v"(bh[0]));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sum = 0.f;
  for (int r = 0; r < 16; ++r) sum += acc0[r] + acc1[r];
  for (int s = 0; s < 8; ++s) sum += bh[s][0] + bl[s][1];
  out[blockIdx.x * 256 + threadIdx.x] = sum + nh[0] + nl[0] + ps0 + ps1;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

void run_staged(float* out, unsigned long long* cyc) {
  const int iters = 1000, blocks = 256;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_staged), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  k_staged<<<blocks, 256, 65536>>>(out, cyc, 10);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  k_staged<<<blocks, 256, 65536>>>(out, cyc, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < 256; ++i) mean += h[i]; mean /= 256;
  printf("staged   : %6.1f ticks/step   %6.1f ns/step (wall)\n", mean / (iters * 8.0), ms * 1e6 / (iters * 8.0));
}

template <int MODE>
void run(float* out, unsigned long long* cyc) {
  const int iters = 1000, blocks = 256;
  auto kk = k<MODE>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kk), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  kk<<<blocks, 256, 65536>>>(out, cyc, 10);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  kk<<<blocks, 256, 65536>>>(out, cyc, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < 256; ++i) mean += h[i]; mean /= 256;
  printf("mode %3d : %6.1f ticks/step   %6.1f ns/step (wall)\n", MODE, mean / (iters * 8.0), ms * 1e6 / (iters * 8.0));
}

int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
  run<0>(out, cyc); run<1>(out, cyc); run<3>(out, cyc); run<7>(out, cyc); run<15>(out, cyc); run<31>(out, cyc); run<63>(out, cyc);
  run<127>(out, cyc); run_staged(out, cyc); run<127>(out, cyc); run_staged(out, cyc); run<64 + 1>(out, cyc); run<8>(out, cyc); run<8 + 16>(out, cyc); run<32>(out, cyc); run<4>(out, cyc);
  return 0;
}
