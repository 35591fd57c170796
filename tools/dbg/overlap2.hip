// probe: WHICH vector instructions of one wave can issue while another wave of the same SIMD keeps the matrix pipe
// busy (gfx950)?  For each flavour: time of the MFMA waves alone, the VALU waves alone, and both together.
// together ~= max(...) -> the instruction overlaps with MFMA; together ~= sum -> it shares the pipe.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define OPS(X)                                                                                      \
  X(0, "v_fma_f32", "v_fma_f32 %0, %0, %1, %2", 32)                                                     \
  X(1, "v_pk_fma_f32", "v_pk_fma_f32 %0, %0, %1, %2", 64)                                               \
  X(2, "v_mul_f32", "v_mul_f32 %0, %0, %1", 32)                                                         \
  X(3, "v_add_f32", "v_add_f32 %0, %0, %1", 32)                                                         \
  X(4, "v_pk_mul_f32", "v_pk_mul_f32 %0, %0, %1", 64)                                                   \
  X(5, "v_pk_add_f32", "v_pk_add_f32 %0, %0, %1", 64)                                                   \
  X(6, "v_max_f32", "v_max_f32 %0, %0, %1", 32)                                                         \
  X(7, "v_xor_b32", "v_xor_b32 %0, %0, %1", 32)                                                         \
  X(8, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 1, %1", 32)                                              \
  X(9, "v_add_u32", "v_add_u32 %0, %0, %1", 32)                                                         \
  X(10, "v_sin_f32", "v_sin_f32 %0, %0", 32)                                                            \
  X(11, "v_cos_f32", "v_cos_f32 %0, %0", 32)                                                            \
  X(12, "v_exp_f32", "v_exp_f32 %0, %0", 32)                                                            \
  X(13, "v_rcp_f32", "v_rcp_f32 %0, %0", 32)                                                            \
  X(14, "v_cvt_pk_f16_f32", "v_cvt_pk_f16_f32 %0, %0, %1", 32)                                          \
  X(15, "v_cvt_f32_f16", "v_cvt_f32_f16 %0, %0", 32)                                                    \
  X(16, "v_cvt_f16_f32", "v_cvt_f16_f32 %0, %0", 32)                                                    \
  X(17, "v_pk_fma_f16", "v_pk_fma_f16 %0, %0, %1, %2", 32)                                              \
  X(18, "v_pk_mul_f16", "v_pk_mul_f16 %0, %0, %1", 32)                                                  \
  X(19, "v_fract_f32", "v_fract_f32 %0, %0", 32)                                                        \
  X(20, "v_rndne_f32", "v_rndne_f32 %0, %0", 32)                                                        \
  X(21, "v_mov_b32", "v_mov_b32 %0, %1", 32)                                                            \
  X(22, "v_and_b32", "v_and_b32 %0, %0, %1", 32)                                                        \
  X(23, "v_fmac_f32", "v_fmac_f32 %0, %1, %2", 32)                                                      \
  X(24, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %2", 32)                                            \
  X(25, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %1, vcc", 32)                                           \
  X(26, "v_sub_f32", "v_sub_f32 %0, %0, %1", 32)                                                        \
  X(27, "v_mul_legacy_f32", "v_mul_legacy_f32 %0, %0, %1", 32)                                          \
  X(28, "v_ldexp_f32", "v_ldexp_f32 %0, %0, %1", 32)                                                    \
  X(29, "v_bfi_b32", "v_bfi_b32 %0, %0, %1, %2", 32)

template <int V>
__device__ __forceinline__ void valu8(double (&p)[8], float (&q)[8], double M, double C) {
  const float m = __int_as_float(__double2loint(M)), c = __int_as_float(__double2loint(C));
#define X(id, name, text, W)                                                                              \
  if (V == id) {                                                                                          \
    if (W == 64) {                                                                                        \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(text : "+v"(p[i]) : "v"(M), "v"(C));     \
    } else {                                                                                              \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(text : "+v"(q[i]) : "v"(m), "v"(c));     \
    }                                                                                                     \
  }
  OPS(X)
#undef X
}

// mode bit0: waves 0..3 run MFMAs; bit1: waves 4..7 run the vector instruction
template <int V>
__global__ void __launch_bounds__(512) k(float* out, int iters, int mode, float a) {
  const int wave = threadIdx.x >> 6;
  f16x8 A, B;
  for (int i = 0; i < 8; ++i) { A[i] = (_Float16)(a + i); B[i] = (_Float16)(a - i); }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  double p[8];
  float q[8];
  for (int i = 0; i < 8; ++i) q[i] = 0.01f * (a + i);
  for (int i = 0; i < 8; ++i) p[i] = __hiloint2double(__float_as_int(0.01f * (a + i)), __float_as_int(0.02f * (a + i)));
  const double M = __hiloint2double(__float_as_int(1.0001f), __float_as_int(0.9999f));
  const double C = __hiloint2double(__float_as_int(0.5f), __float_as_int(0.25f));
  const bool do_m = (mode & 1) && wave < 4;
  const bool do_v = (mode & 2) && wave >= 4;
  if (do_m) {
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c3, 0, 0, 0);
    }
  } else if (do_v) {
    for (int i = 0; i < iters; ++i) {
      valu8<V>(p, q, M, C);
      valu8<V>(p, q, M, C);
    }
  }
  float r = c0[0] + c1[1] + c2[2] + c3[3];
  for (int i = 0; i < 8; ++i) r += (float)__double2loint(p[i]) + (float)__double2hiint(p[i]) + q[i];
  out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int V>
void run(const char* name, float* d) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  float t[4] = {0, 0, 0, 0};
  for (int m = 1; m <= 3; ++m) {
    k<V><<<256, 512>>>(d, 10, m, 1.f);
    (void)hipEventRecord(e0);
    k<V><<<256, 512>>>(d, iters, m, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&t[m], e0, e1);
  }
  // per iteration: 4 MFMAs on the MFMA waves, 16 vector instructions on the VALU waves
  const double s = 1e6 / iters;
  const double ov = (t[1] + t[2] - t[3]) / (t[1] < t[2] ? t[1] : t[2]);
  printf("%-18s mfma %6.1f ns  valu %6.1f ns (%.2f ns/instr)  together %6.1f ns  overlap %.0f %%\n", name, t[1] * s, t[2] * s,
         t[2] * s / 16, t[3] * s, 100 * ov);
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 512 * 4);
#define X(id, name, text, W) run<id>(name, d);
  OPS(X)
#undef X
  return 0;
}
