#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ float dpp_mov(float x, float old = 0.f) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, x), CTRL, ROWMASK, 0xf, false));
}
__global__ void k(const float* in, float* out) {
  float x = in[threadIdx.x];
  float a = x + dpp_mov<0xB1>(x);
  float b = a + dpp_mov<0x4E>(a);
  float c = b + dpp_mov<0x141>(b);
  float d = c + dpp_mov<0x140>(c);
  float e = d + dpp_mov<0x142, 0xa>(d);
  float f = e + dpp_mov<0x143, 0xc>(e);
  out[threadIdx.x] = a; out[64+threadIdx.x] = b; out[128+threadIdx.x] = c; out[192+threadIdx.x] = d; out[256+threadIdx.x] = e; out[320+threadIdx.x]=f;
}
int main() {
  float h[64], o[384]; for (int i=0;i<64;++i) h[i] = 1 << (i%16) ;
  for (int i=0;i<64;++i) h[i] = (float)(1 << (i%16)) * ((i<16)?1.f: (i<32?0.5f: (i<48?0.25f:0.125f)));
  float *di, *dout; hipMalloc(&di, 256); hipMalloc(&dout, 384*4); hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout); hipMemcpy(o, dout, 384*4, hipMemcpyDeviceToHost);
  for (int s=0;s<6;++s){ printf("step %d:", s); for (int i=0;i<64;i+=1) printf(" %g", o[s*64+i]); printf("\n"); }
  return 0;
}
