// probe: does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float a, float b) {
  f16x8 A, B;
  for (int i = 0; i < 8; ++i) { A[i] = (_Float16)a; B[i] = (_Float16)b; }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc, 0, 0, 0);
  out[threadIdx.x] = acc[0];
  if (threadIdx.x == 0) { out[64] = (float)A[0]; out[65] = (float)B[0]; }
}
int main() {
  float* d; hipMalloc(&d, 66 * 4);
  float tests[][2] = {{1.f, 1.f}, {9.5367431640625e-07f, 1.f}, {1.f, 9.5367431640625e-07f}, {3.0517578125e-05f, 3.0517578125e-05f}, {6e-8f, 1024.f}};
  for (auto& t : tests) {
    k<<<1, 64>>>(d, t[0], t[1]);
    float h[66]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("a=%g b=%g  cvt a=%g b=%g  mfma sum16=%g expect=%g\n", t[0], t[1], h[64], h[65], h[0], 16.0 * h[64] * h[65]);
  }
  return 0;
}
