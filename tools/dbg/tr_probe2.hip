// ds_read_b64_tr_b16 semantics probe (gfx950): LDS halfword i holds the value i; every lane reads with address pattern `mode`
// and the four returned halfwords are printed per lane.   hipcc --offload-arch=gfx950 -O2 tr_probe2.hip -o bin/tr_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__global__ void k(int mode, int stride, float* out) {
  __shared__ _Float16 lds[4096];
  volatile _Float16* vl = lds;   // (the only reader is the asm below: without volatile hipcc deletes the stores)
  for (int i = threadIdx.x; i < 2048; i += 64) vl[i] = (_Float16)(float)i;
  for (int i = 2048 + threadIdx.x; i < 4096; i += 64) vl[i] = (_Float16)(-1.0f);
  __syncthreads();
  const int l = threadIdx.x, t = l & 15, grp = l >> 4;
  int addr;   // byte address
  if (mode == 0) addr = l * 8;                                             // lane l: halfwords 4l .. 4l+3
  else if (mode == 1) addr = ((t >> 2) * stride + 4 * (t & 3) + 16 * grp) * 2;  // hypothesis: row t/4, cols 4(t%4).. of block grp
  else addr = 0;
  h4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (float)v[j];
}
int main(int argc, char** argv) {
  float* d; (void)hipMalloc(&d, 256 * 4);
  float h[256];
  for (int mode = 0; mode < 2; ++mode) {
    const int stride = 64;   // halfwords per row in mode 1
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, mode, stride, d);
    hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
    if (e1 != hipSuccess || e2 != hipSuccess) printf("launch: %s / sync: %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d (stride %d halfwords)\n", mode, stride);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %5.0f %5.0f %5.0f %5.0f\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  }
  return 0;
}
