// probe: issue cost of the conversion instructions the operand splits use (gfx950), one wave per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
template <int OP>
__global__ void k(float* out, int iters) {
  float a = threadIdx.x * 0.001f + 1.f, b = a + 1.f, c = a + 2.f, d = a + 3.f, e = a + 4.f, f = a + 5.f, g = a + 6.f, h = a + 7.f;
  unsigned u0 = 0, u1 = 0, u2 = 0, u3 = 0;
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { REP8(asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(u0) : "v"(a)); asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(u1) : "v"(b));) }
    if (OP == 1) { REP8(asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u0) : "v"(a), "v"(b)); asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u1) : "v"(c), "v"(d));) }
    if (OP == 2) { REP8(asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(e) : "v"(u2)); asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(f) : "v"(u3));) }
    if (OP == 3) { REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u0) : "v"(a), "v"(b)); asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u1) : "v"(c), "v"(d));) }
    if (OP == 4) { REP8(asm volatile("v_sub_f32 %0, %1, %2" : "=v"(g) : "v"(a), "v"(b)); asm volatile("v_sub_f32 %0, %1, %2" : "=v"(h) : "v"(c), "v"(d));) }
    if (OP == 5) { REP8(asm volatile("v_and_b32 %0, %1, %2" : "=v"(u2) : "v"(u0), "v"(u1)); asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u3) : "v"(u0));) }
    if (OP == 6) { REP8(asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(g) : "v"(u0), "v"(a)); asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(h) : "v"(u0), "v"(b));) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + u0 + u1 + u2 + u3;
}
template <int OP> void run(const char* n) {
  float* d; (void)hipMalloc(&d, 256 * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  k<OP><<<256, 256>>>(d, 10);
  (void)hipEventRecord(e0);
  k<OP><<<256, 256>>>(d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-22s %.2f ns per instruction per wave\n", n, ms * 1e6 / (iters * 16.0));
}
int main() {
  run<0>("v_cvt_f16_f32"); run<1>("v_cvt_pk_f16_f32"); run<2>("v_cvt_f32_f16"); run<3>("v_cvt_pk_bf16_f32");
  run<4>("v_sub_f32"); run<5>("v_and / v_lshlrev"); run<6>("v_fma_mix_f32");
  return 0;
}
