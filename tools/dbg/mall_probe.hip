// probe: does a write-then-read-back scratch (LIFO per wave, like sdf_mlp_kernel's gamma*cos(phi) slots) run faster when
// every resident wave re-uses ONE private region (footprint = resident waves x region: can live in the 256 MB
// Infinity Cache) than when every tile gets fresh addresses (footprint = whole problem, streams through HBM)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int SLOT = 16384;  // bytes per slot per wave (64 lanes x 16 B x 16 fragments)
template <int NT>
__global__ void __launch_bounds__(512) k(char* scratch, float* out, int tiles, int slots, int persistent) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f4 acc = {0, 0, 0, 0};
  for (int t = 0; t < tiles; ++t) {
    const long long region = persistent ? ((long long)blockIdx.x * 8 + wave)
                                        : (((long long)t * gridDim.x + blockIdx.x) * 8 + wave);
    f4* base = reinterpret_cast<f4*>(scratch + region * (long long)slots * SLOT) + lane;
    for (int s = 0; s < slots; ++s)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        f4 v = {acc[0] + s, acc[1] + g, (float)t, (float)lane};
        if (NT) __builtin_nontemporal_store(v, base + (s * 16 + g) * 64);
        else base[(s * 16 + g) * 64] = v;
      }
    for (int s = slots - 1; s >= 0; --s)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        f4 v = NT ? __builtin_nontemporal_load(base + (s * 16 + g) * 64) : base[(s * 16 + g) * 64];
        acc += v * 1e-9f;
      }
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
int main() {
  const int tiles = 8, WG = 256;
  char* s; float* o;
  const size_t bytes = (size_t)tiles * WG * 8 * 9 * SLOT;  // 2.4 GB at 9 slots
  (void)hipMalloc(&s, bytes); (void)hipMalloc(&o, WG * 512 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int nt = 0; nt < 2; ++nt)
    for (int slots : {9, 7, 5, 3})
      for (int pers = 0; pers < 2; ++pers) {
        auto kk = nt ? k<1> : k<0>;
        float best = 1e9;
        for (int r = 0; r < 4; ++r) {
          (void)hipEventRecord(e0);
          kk<<<WG, 512>>>(s, o, tiles, slots, pers);
          (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
          float ms; (void)hipEventElapsedTime(&ms, e0, e1);
          if (r && ms < best) best = ms;
        }
        const double moved = 2.0 * tiles * WG * 8 * (double)slots * SLOT;
        printf("nt=%d slots=%d (%5.1f MB resident) %-10s  %.3f ms  %.2f TB/s (write+read)\n", nt, slots,
               WG * 8 * slots * SLOT / 1048576.0, pers ? "persistent" : "fresh", best, moved / best * 1e-9);
      }
  return 0;
}
