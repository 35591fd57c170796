// What ds_read_b64_tr_b16 returns: every lane reads 4 halves from its OWN 8-byte address; the LDS holds its own index
// (half h of the buffer = h), so the output shows which source lane / element each destination element comes from.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short u16;
__global__ void k(unsigned* out, int stride_halves) {
  __shared__ __attribute__((aligned(16))) u16 lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (u16)i;
  __syncthreads();
  const int lane = threadIdx.x;
  const unsigned addr = (unsigned)(size_t)(lds) + lane * stride_halves * 2;  // lane's 4 halves start at half index lane*stride
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  u2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr) : "memory");
  out[lane * 2 + 0] = r[0];
  out[lane * 2 + 1] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 64 * 8);
  for (int stride : {4, 16, 64}) {
    k<<<1, 64>>>(d, stride);
    unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("stride %d halves per lane: lane -> 4 elements as (source lane . element)\n", stride);
    for (int l = 0; l < 64; ++l) {
      printf("  l%2d:", l);
      for (int j = 0; j < 4; ++j) {
        unsigned v = (h[l * 2 + (j >> 1)] >> (16 * (j & 1))) & 0xffff;
        printf(" %2u.%u", v / stride, v % stride);
      }
      if (l % 4 == 3) printf("\n");
    }
  }
  return 0;
}
