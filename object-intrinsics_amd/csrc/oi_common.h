// Shared helpers for liboi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdarg>
#include <cstdio>

#include "../../include/oi_hip.h"

namespace oi {

inline char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(OI_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return OI_OK;
}

#define OI_REQUIRE(cond, ...) \
  do {                        \
    if (!(cond)) return oi::fail(OI_ERR_INVALID_ARG, __VA_ARGS__); \
  } while (0)

inline hipStream_t as_stream(oi_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// DPP data movement (one VALU instruction, fused into the consuming op by hipcc); __shfl_xor would lower to
// ds_bpermute_b32 + s_waitcnt (an LDS round trip per step).
template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ float dpp_mov(float x, float old = 0.f) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, x),
                                                                CTRL, ROWMASK, 0xf, false));
}
// sum over each 32-lane half of the wavefront; the result is valid in lanes 16..31 of each half
__device__ __forceinline__ float half_sum32(float x) {
  x += dpp_mov<0xB1>(x);        // quad_perm [1,0,3,2]
  x += dpp_mov<0x4E>(x);        // quad_perm [2,3,0,1]
  x += dpp_mov<0x141>(x);       // row_half_mirror
  x += dpp_mov<0x140>(x);       // row_mirror
  x += dpp_mov<0x142, 0xa>(x);  // row_bcast15 -> rows 1, 3
  return x;
}
// 64-lane sum / max; every lane ends with the result (broadcast through an SGPR).
__device__ __forceinline__ float wave_sum(float v) {
  v = half_sum32(v);
  v += dpp_mov<0x143, 0xc>(v);  // row_bcast31 -> rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  const float ninf = -3.0e38f;
  v = fmaxf(v, dpp_mov<0xB1>(v, ninf));
  v = fmaxf(v, dpp_mov<0x4E>(v, ninf));
  v = fmaxf(v, dpp_mov<0x141>(v, ninf));
  v = fmaxf(v, dpp_mov<0x140>(v, ninf));
  v = fmaxf(v, dpp_mov<0x142, 0xa>(v, ninf));
  v = fmaxf(v, dpp_mov<0x143, 0xc>(v, ninf));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// Inclusive scan over the 64 lanes of a wavefront (log-step shuffles).
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(v, o, 64);
    if (lane >= o) v *= t;
  }
  return v;
}
__device__ __forceinline__ float wave_scan_add(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// 128-bit store through a buffer descriptor: per-lane byte offset `voff` + wave-uniform byte offset `uoff`.
// HAZARD (observed on gfx950 / ROCm 7.2): a VALU write to the data VGPRs of a >64-bit VMEM store in the
// wait states right after it corrupts the stored dwords (lanes 12..15 of every row here).  hipcc pads this
// hazard only when the store's soffset is NOT a register (an SI-era exemption in the hazard recogniser that
// does not hold on this part), so the uniform offset is folded into voffset (one v_add) and soffset stays 0;
// the compiler then inserts the required s_nop itself.  Loads keep the SGPR soffset (no such hazard).
// Cache policy of the scratch streams (buffer intrinsic aux operand, bit 1 = nt), measured A/B on one MI355X:
//   forward kernel  (9 slots, each written once and read back once, LIFO):  loads nt, stores ordinary  1.130 ms
//                    both nt 1.180 ms, stores only 1.168 ms, neither 1.204 ms -- a non-temporal STORE makes the line's
//                    only re-read miss, a non-temporal LOAD tells the caches the line is dead after its last use;
//   backward sweep  (46 slots, 25 GB, most of it consumed by the NEXT kernel):  both nt 68.0 it/s of training,
//                    loads only 67.5, stores only 67.5, neither 66.6.
#ifndef OI_FWD_NT_LD
#define OI_FWD_NT_LD 2
#endif
#ifndef OI_FWD_NT_ST
#define OI_FWD_NT_ST 0
#endif
#ifndef OI_BWD_NT_LD
#define OI_BWD_NT_LD 2
#endif
#ifndef OI_BWD_NT_ST
#define OI_BWD_NT_ST 2
#endif
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <int AUX>
__device__ __forceinline__ void buffer_store_b128(u32x4_t v, __amdgpu_buffer_rsrc_t rs, int voff, int uoff) {
  __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff + uoff, 0, AUX);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Zero fill as a KERNEL (grid-stride dwords).  The launchers used hipMemsetAsync; captured into a hipGraph (the
// discriminator step of oi_amd.graphed.GraphedDStep) those memset nodes left split-K / scatter-add outputs partly
// unfilled on ROCm 7.2 -- replay six of an otherwise bit-identical step summed onto stale memory and overflowed.  A kernel
// node is replayed like every other launch.
static __global__ void zero_fill_kernel(float* __restrict__ p, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = 0.f;
}
inline hipError_t zero_async(float* p, size_t n_floats, hipStream_t st) {
  if (n_floats == 0) return hipSuccess;
  const long long blocks = (long long)((n_floats + 255) / 256);
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, st, p, (long long)n_floats);
  return hipGetLastError();
}

// "Am I the last workgroup of this launch to get here?"  Called by ONE lane per workgroup after the workgroup's payload stores
// (agent-scope write-through: __hip_atomic_store relaxed / agent) have been acknowledged (s_waitcnt vmcnt(0)); the workgroup
// for which it returns true may read every other workgroup's payload with agent-scope loads.  Two levels of arrival counters
// -- groups of LA_GROUP workgroups, then one counter over the groups: a single word takes ~88 same-address atomics per
// microsecond on this part, i.e. 12 us for the 1024 workgroups of a C2 compositing launch (measured: the fused statistics
// cost 7 us more than the launch they saved until the counter was split).  Nobody waits for anybody; every counter is left
// at zero.  `ticket`: 1 + ceil(nblocks / LA_GROUP) zero-initialised words, not shared by launches that may overlap.
constexpr int LA_GROUP = 32;
constexpr int LA_WORDS = 1 + 4096;   // callers allocate this many words (up to 131,072 workgroups)
__device__ __forceinline__ bool last_arriver(unsigned* ticket, unsigned block, unsigned nblocks) {
  const unsigned ngroups = (nblocks + LA_GROUP - 1) / LA_GROUP, g = block / LA_GROUP;
  const unsigned gsize = g + 1 == ngroups ? nblocks - g * LA_GROUP : LA_GROUP;
  // Relaxed agent-scope atomics on purpose.  The hand-off is the guide's "write-through payload + drained flag" form
  // (MI355X_MICROARCH.md, valid forms: `sc1` payload stores -> asm s_waitcnt vmcnt(0) -> agent-scope atomic; the reader uses
  // agent-scope loads): visibility comes from the write-through stores being ACKNOWLEDGED before the arrival, not from a fence.
  // acq_rel orderings here lower to buffer_wbl2 + buffer_inv per arrival (1.7-3.5 us each on this part, price list row
  // "fence") -- more than the 4.5 us launch the last-arriver form replaces.  Both launchers check the grid against the ticket.
  if (__hip_atomic_fetch_add(ticket + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gsize - 1) return false;
  __hip_atomic_store(ticket + 1 + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ngroups - 1) return false;
  __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

// Accumulate-outputs (split-K sums, scatter-adds) are cleared by their launcher -- unless the caller has declared FOR THE
// STREAM OF THE LAUNCH that it hands every such output out of memory it has already zeroed (oi_outputs_prezeroed_stream: one
// fill per captured step instead of one per op).  Keyed by stream, not by thread and not process-wide: PyTorch runs a
// backward pass on its autograd worker thread but on the forward's stream, so the declaration covers those launches, while
// another thread working on its own stream ("re-entrant and thread-safe") keeps the default behaviour.
constexpr int PREZERO_SLOTS = 16;
inline std::atomic<uintptr_t>* prezeroed_streams() {
  static std::atomic<uintptr_t> v[PREZERO_SLOTS];  // 0 = free; otherwise (stream handle | 1): the null stream has the key 1
  return v;
}
inline bool stream_prezeroed(hipStream_t st) {
  const uintptr_t key = reinterpret_cast<uintptr_t>(st) | 1u;
  std::atomic<uintptr_t>* v = prezeroed_streams();
  for (int i = 0; i < PREZERO_SLOTS; ++i)
    if (v[i].load(std::memory_order_relaxed) == key) return true;
  return false;
}
// -> previous setting of that stream, or -1 when the table is full
inline int set_stream_prezeroed(hipStream_t st, bool on) {
  const uintptr_t key = reinterpret_cast<uintptr_t>(st) | 1u;
  std::atomic<uintptr_t>* v = prezeroed_streams();
  for (int i = 0; i < PREZERO_SLOTS; ++i) {
    uintptr_t cur = key;
    if (on ? v[i].load(std::memory_order_relaxed) == key : v[i].compare_exchange_strong(cur, 0)) return 1;  // was on
  }
  if (!on) return 0;
  for (int i = 0; i < PREZERO_SLOTS; ++i) {
    uintptr_t empty = 0;
    if (v[i].compare_exchange_strong(empty, key)) return 0;
  }
  return -1;
}
inline hipError_t zero_output_async(float* p, size_t n_floats, hipStream_t st) {
  return stream_prezeroed(st) ? hipSuccess : zero_async(p, n_floats, st);
}

}  // namespace oi
