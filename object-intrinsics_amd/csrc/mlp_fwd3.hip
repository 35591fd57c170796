// FiLM-SIREN forward with the analytic d sdf/dx and the albedo head, register-resident (gfx950, F16X3 operands).
//
// Same contract as sdf_mlp_kernel<F16X3, FULL> in mlp.hip (reference
// src/models/fields.py:49-77, 89-101, 104-122; src/third_party/stylesdf/volume_renderer.py:50-61), different
// residency: the gamma*cos(phi) fragments the reverse sweep needs never leave the register file.
//
//  * ONE wavefront per SIMD (4 waves = 128 points per workgroup and CU), the whole 512-entry VGPR+AGPR file per lane:
//    four 64-register "park" banks hold gamma*cos(phi) of up to four layers (hipcc places them in the AGPR half).
//  * The network is swept in two segments so that four banks suffice:
//       forward  layers 0..7        parks c3..c6;   layer 7 emits v7 = w_sigma * c7 directly (c7 is never stored)
//       reverse  layers 7..3        g3 -> bank 0
//       forward  layers 0..2 AGAIN  parks c0, c1;   layer 2 emits v2 = g3 * c2 directly (only cos is evaluated)
//       reverse  layers 2..0        -> d sdf/dx
//       albedo head
//    17 layer products instead of 15 (+13 % MFMAs), and the 4.6 KB/point scratch stream of the v2 kernel (4.95 GB per
//    524,288-point launch, profiles/r1_pmc_*_f16x3.txt) shrinks to the 512 B/point feature vector that the albedo head
//    reads back after the sweep.
//  * Every layer product is formed output block by output block; the FiLM / sin (or c-multiply) work of block t-1 is
//    issued between block t's MFMAs -- with one wave per SIMD nothing else can fill the matrix pipe's shadow.
#include <type_traits>

#include "mlp_common.h"
#include "f3_blob.h"

namespace {

using namespace oimlp;
static_assert(oif3::FC == C && oif3::HB_SIG == H_SIG && oif3::HB_TABS_END == H_TABS_END && oif3::HB_BIAS == H_BIAS &&
              oif3::HB_WSCALE == H_WSCALE && oif3::HB_NL_SDF == NL_SDF, "f3_blob.h mirrors the packed header's offsets");

constexpr int F3_WAVES = 4;
constexpr int F3_TILE = F3_WAVES * WAVE_PTS;  // 128 points per workgroup
// LDS: FiLM rows [10][A 128 | B 128 | G 128] floats, small tables, double-buffered image ring
using oif3::F3_FILM;       // 0
using oif3::F3_FILM_ROW;   // 3 * C * 4 bytes per FiLM layer
using oif3::F3_TABS;       // 15360
using oif3::F3_GMAX;       // [16] max |G_l| per FiLM layer (9: max |G7 w_sigma|)
// OI_F3_BLOB (round 5): everything in front of the image ring depends on the batch element only, so ONE launch per call
// (film_blob_f3_kernel, a block per element) writes it as a 22 KiB blob and a tile's prologue is 22 LDS-DMA copies of 1 KiB
// instead of ~25 dependent global loads per thread, 21 KB of ds_writes, the row maxima and two barriers -- 4,096 workgroups of
// a C2 launch repeated that work, 8.7k of a tile's 136k cycles with nothing to hide behind (one workgroup per CU).
#ifndef OI_F3_BLOB
#define OI_F3_BLOB 1
#endif
using oif3::F3_BLOB;       // 22528
constexpr int F3_WBUF = OI_F3_BLOB ? F3_BLOB : F3_GMAX + 64;  // 22,528 (21,696 without the blob)
constexpr int F3_LDS = F3_WBUF + 2 * 65536;                 // 153,600 of the CU's 163,840 bytes

// One parked 128-vector of this lane's point: [group g][k]  <->  act[4 g + k].  The values are pinned to the ACCUMULATOR
// half of the register file through the "a" constraint: left to itself hipcc's allocator treats them as ordinary
// VGPR candidates and, with 4 x 64 of them alive next to a ~240-register working set, spills hundreds of dwords to
// scratch memory (-Rpass-analysis=kernel-resource-usage); as AGPR-class values they cost one v_accvgpr_write and one
// v_accvgpr_read each and never compete with the VALU operands.
typedef float Bank[16][4];
// timing ablations of single epilogue pieces (results garbage): -DOI_F3_ABL_PIECE bit 1 = no v_sin / v_cos (a multiply
// instead), 2 = no v_fract, 4 = no AGPR parks, 8 = fp16 split without the residual (hi limb twice)
#ifndef OI_F3_ABL_PIECE
#define OI_F3_ABL_PIECE 0
#endif
__device__ __forceinline__ float f3_sin(float r) { return (OI_F3_ABL_PIECE & 1) ? r * 0.75f : __builtin_amdgcn_sinf(r); }
__device__ __forceinline__ float f3_cos(float r) { return (OI_F3_ABL_PIECE & 1) ? r * 0.85f : __builtin_amdgcn_cosf(r); }
__device__ __forceinline__ float to_acc(float v) {
  if (OI_F3_ABL_PIECE & 4) {
    asm volatile("" ::"v"(v));
    return 0.25f;
  }
  float a;
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v));
  return a;
}
__device__ __forceinline__ float from_acc(float a) {
  if (OI_F3_ABL_PIECE & 4) return a;
  float v;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
  return v;
}

constexpr int F3_CU_SLOTS = 4096;  // 16 XCC ids x 256 (SE, SH, CU) codes: an upper bound on distinct physical CUs
constexpr int F3_WG_SLOTS = 512;   // launches of up to this many workgroups index their scratch by workgroup
// index of the compute unit this wave runs on: HW_REG_XCC_ID[3:0] and HW_REG_HW_ID[15:8] = {SE_ID[2:0], SH_ID, CU_ID[3:0]}
__device__ __forceinline__ int cu_slot_id() {
  const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);  // 4 bits at offset 0 of register 20
  const unsigned cu = __builtin_amdgcn_s_getreg((8 - 1) << 11 | 8 << 6 | 4);    // 8 bits at offset 8 of register 4
  return (int)(xcc * 256 + cu);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// Packed fp32 arithmetic of the epilogues.  OI_F3_PK: 0 = scalar source, 1 = 2-vectors (hipcc unpacks v_pk_*_f32 it finds in
// the shadow of an MFMA on gfx950 again), 2 = the packed instruction as written (inline asm)
#ifndef OI_F3_PK
#define OI_F3_PK 1
#endif
#ifndef OI_F3_MAX3
#define OI_F3_MAX3 1
#endif
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
#if OI_F3_PK == 2
  f32x2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
#elif OI_F3_PK == 1
  return __builtin_elementwise_fma(a, b, c);
#else
  return f32x2{fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])};
#endif
}
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) {
#if OI_F3_PK == 2
  f32x2 r;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return a * b;
#endif
}
// a * s with the scalar s taken from the low dword of its register for both halves
__device__ __forceinline__ f32x2 pk_mul_s(f32x2 a, float s) {
#if OI_F3_PK == 2
  f32x2 r, sv;
  sv[0] = s;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(sv));
  return r;
#else
  return a * s;
#endif
}
// FiLM phase of an accumulator pair (the rows and the accumulators sit in aligned register pairs)
__device__ __forceinline__ f32x2 film_phase2(const f32x4& a, const f32x4& b, int k, float acc0, float acc1) {
  return pk_fma(f32x2{a[k], a[k + 1]}, f32x2{acc0, acc1}, f32x2{b[k], b[k + 1]});
}
// max(m, |x|, |y|) in one v_max3_f32
__device__ __forceinline__ float max3_abs(float m, float x, float y) {
#if OI_F3_MAX3
  float r;
  asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(m), "v"(x), "v"(y));
  return r;
#else
  return fmaxf(m, fmaxf(fabsf(x), fabsf(y)));
#endif
}

typedef unsigned Limbs[8][4];  // one fp16 limb plane of a B operand: [k-step][dword d] = act indices 8 s + 2 d, 8 s + 2 d + 1


// Adjoint vectors have no a-priori range: bring this point's 128-vector (64 entries here, 64 in lane ^ 32) to
// max |.| in [2^13, 2^14) with an exact power-of-two scale before the fp16 split; `run` accumulates the inverse scales
// (true vector = act * run).  `m` = this lane's max |act| (tracked by the epilogue that produced act).
__device__ __forceinline__ void split_limbs_normalised(const float (&act)[64], float m, float& run, Limbs& bh, Limbs& bl) {
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  int eb = (__builtin_bit_cast(int, m) >> 23) & 0xff;
  eb = eb < 14 ? 14 : (eb > 254 ? 254 : eb);
  const float sc = __builtin_bit_cast(float, (267 - eb) << 23);  // 2^(13 - (eb - 127))
  run *= __builtin_bit_cast(float, (eb - 13) << 23);             // 1 / sc
#pragma unroll
  for (int s = 0; s < 8; ++s) {
#pragma unroll
    for (int d = 0; d < 4; ++d) split_pair(act[8 * s + 2 * d] * sc, act[8 * s + 2 * d + 1] * sc, bh[s][d], bl[s][d]);
    __builtin_amdgcn_sched_barrier(0);
  }
}
__device__ __forceinline__ void split_limbs(const float (&act)[64], Limbs& bh, Limbs& bl) {
#pragma unroll
  for (int s = 0; s < 8; ++s) {
#pragma unroll
    for (int d = 0; d < 4; ++d) split_pair(act[8 * s + 2 * d], act[8 * s + 2 * d + 1], bh[s][d], bl[s][d]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

struct NoTail {
  __device__ __forceinline__ void operator()(int, int) const {}
};
template <class T> struct is_no_tail { static constexpr bool value = false; };
template <> struct is_no_tail<NoTail> { static constexpr bool value = true; };

// One layer product of the stream: acc = W_img . B (three fp16 MFMAs per product), output block t outer, one
// scheduling window per k-step = 3 MFMAs + the epilogue work that hides behind them:
//   block 0, k-steps 0..5   TAIL(3, rp): the PREVIOUS layer's block-3 epilogue pairs (8 of them, spread 2 1 1 2 1 1);
//                           they complete THIS layer's B limbs for k-steps 6 and 7 just before those are consumed
//   after block 0           MID(): runs once, when the previous layer's output vector is complete
//   blocks 1..3             EPI(t - 1, s): this layer's epilogue pair s of the block that has just completed
// This layer's own block-3 pairs are left to the caller: the next layer's TAIL, or run_tail().
// An epilogue pair (tb, rp) consumes accumulator slots 2 rp, 2 rp + 1 of block tb = act indices 16 tb + 2 rp (+1).
// With ONE wave per SIMD nothing but the wave's own independent instructions fills a dependency stall, and an MFMA that
// follows an MFMA waits for the matrix pipe with the whole wave behind it: sched_group_barrier pins every window to
// MFMA, n VALU, MFMA, n VALU, MFMA, rest.
#ifndef OI_F3_GROUPS
#define OI_F3_GROUPS 1
#endif
#ifndef OI_F3_ADIST
#define OI_F3_ADIST 1
#endif
#ifndef OI_F3_ABL_EPI
#define OI_F3_ABL_EPI 0
#endif
#ifndef OI_F3_WINSTEPS
#define OI_F3_WINSTEPS 1
#endif
#ifndef OI_F3_DSFIRST
#define OI_F3_DSFIRST 0
#endif
#ifndef OI_F3_VALU_PER_MFMA
#define OI_F3_VALU_PER_MFMA 6
#endif
struct NoMid {
  __device__ __forceinline__ void operator()() const {}
};
template <class TAIL, class MID, class EPI>
__device__ __forceinline__ void stream_layer(const char* lds, const LayOff& y, const Limbs& bh, const Limbs& bl,
                                             f32x16 (&acc)[4], TAIL&& tail, MID&& mid, EPI&& epi) {
  constexpr bool HAS_TAIL = !is_no_tail<std::remove_cv_t<std::remove_reference_t<TAIL>>>::value;
  // A fragments (hi / lo limb of the image) are requested OI_F3_ADIST k-steps ahead of their MFMAs: with one wave per
  // SIMD a ds_read_b128 issued only one window (~100 cycles) ahead is not back when its MFMA comes up
  constexpr int AD = OI_F3_ADIST;
  f32x4 ah[AD + 1], al[AD + 1];
#pragma unroll
  for (int i = 0; i < AD; ++i) {
    ah[i] = lds_f4(lds, i * 1024, y.wl);
    al[i] = lds_f4(lds, i * 1024, y.wh);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int cur = t * 8 + s, nxt = cur + AD;
      if (nxt < 32 && !(OI_F3_ABL_EPI == 2 && nxt >= 2)) {  // ablation 2: no A-fragment reads after the first two
        ah[nxt % (AD + 1)] = lds_f4(lds, nxt * 1024, y.wl);
        al[nxt % (AD + 1)] = lds_f4(lds, nxt * 1024, y.wh);
      }
      int npairs = 0;
#if OI_F3_ABL_EPI  // timing ablation: no epilogue work at all (the accumulators are only kept alive)
      if (t > 0) asm volatile("" ::"v"(acc[t - 1][2 * s]), "v"(acc[t - 1][2 * s + 1]));
#else
      if (t == 0) {
        if (HAS_TAIL && s < 6) {
          constexpr int first[7] = {0, 2, 3, 4, 6, 7, 8};
          for (int rp = first[s]; rp < first[s + 1]; ++rp) tail(3, rp);
          npairs = first[s + 1] - first[s];
        }
      } else {
        epi(t - 1, s);
        npairs = 1;
      }
#endif
      const int ci = (OI_F3_ABL_EPI == 2 ? cur & 1 : cur) % (AD + 1);
      const f16x8 wh = __builtin_bit_cast(f16x8, ah[ci]);
      const f16x8 wl = __builtin_bit_cast(f16x8, al[ci]);
      const u32x4 uh = {bh[s][0], bh[s][1], bh[s][2], bh[s][3]}, ul = {bl[s][0], bl[s][1], bl[s][2], bl[s][3]};
      const f16x8 vh = __builtin_bit_cast(f16x8, uh);
      const f16x8 vl = __builtin_bit_cast(f16x8, ul);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, vh, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, vl, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, vh, acc[t], 0, 0, 0);
      // OI_F3_WINSTEPS k-steps per scheduling window: with 2, two epilogue pairs (four independent chains) share a window
      // and hide each other's dependency stalls (v_sin -> cvt_pk -> fma_mix -> cvt_pk is a serial chain)
      if (OI_F3_WINSTEPS == 1 || (s % OI_F3_WINSTEPS) == OI_F3_WINSTEPS - 1) {
        if (OI_F3_GROUPS && (npairs >= 1 || OI_F3_WINSTEPS > 1)) {
          if (OI_F3_DSFIRST) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);  // every LDS read of the window first
#pragma unroll
          for (int q = 0; q < 3 * OI_F3_WINSTEPS - 1; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (npairs <= 1) __builtin_amdgcn_sched_group_barrier(0x002, OI_F3_VALU_PER_MFMA, 0);
            else __builtin_amdgcn_sched_group_barrier(0x002, 2 * OI_F3_VALU_PER_MFMA, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (OI_F3_ABL_PIECE & 16) asm volatile("" ::"v"(acc[t]));  // keeps the MFMAs alive when nothing reads them
    if (t == 0) {
      mid();  // the previous layer's output vector is complete here (its block-3 pairs ran above): scale decisions
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
// block 3's epilogue with nothing to hide behind: two pairs (four independent chains) per window
template <class EPI>
__device__ __forceinline__ void run_tail(EPI&& epi) {
  if (OI_F3_ABL_EPI) return;
#pragma unroll
  for (int rp = 0; rp < 8; ++rp) {
    epi(3, rp);
    if (rp & 1) __builtin_amdgcn_sched_barrier(0);
  }
}

// -DOI_F3_PROF: per-phase shader-clock accounting (tools/dbg/phase_prof3.py); -DOI_F3_ABL_NOPREFETCH: timing ablation
// without the weight stream (results are garbage)
#ifdef OI_F3_PROF
__device__ unsigned long long oi_prof3[16];
#define F3_T(i)                                                  \
  do {                                                           \
    const unsigned long long t_ = __builtin_readcyclecounter();  \
    pacc[i] += t_ - tprev;                                       \
    tprev = t_;                                                  \
  } while (0)
#else
#define F3_T(i)
#endif

template <bool FAST>
__global__ void __launch_bounds__(64 * F3_WAVES) __attribute__((amdgpu_waves_per_eu(1, 1)))
sdf_mlp_full3_kernel(const float* __restrict__ pts, const char* __restrict__ packed, const float* __restrict__ gamma,
                     const float* __restrict__ beta, float* __restrict__ sdf_out, float* __restrict__ grad_out,
                     float* __restrict__ rgb_out, float* __restrict__ feat_out, char* __restrict__ scratch,
                     const char* __restrict__ blob, long long n_per_elem) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int LB = 65536;
#ifdef OI_F3_PROF
  const unsigned long long t_entry = __builtin_readcyclecounter();
  const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();  // constant 100 MHz
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, j = lane & 31;
  const int e = blockIdx.y;
  const float* hdr = reinterpret_cast<const float*>(packed);
  const char* mats = packed + H_BYTES;

  LaneOff o;
  o.h16 = 16 * h;
  o.h64 = 64 * h;
  o.l16 = 16 * lane;
  o.l16hi = 16 * lane + 32768;
  asm volatile("" : "+v"(o.h16), "+v"(o.h64), "+v"(o.l16), "+v"(o.l16hi));

  // global point index of this lane (tail lanes re-read the element's last point); recomputed at every use from a
  // laundered lane id instead of living in two VGPRs for the whole kernel (the register budget is full: 256 + 256)
  auto point_of = [&](bool& valid) {
    int jj = lane & 31;
    asm volatile("" : "+v"(jj));
    const long long local = (long long)blockIdx.x * F3_TILE + wave * WAVE_PTS + jj;
    valid = local < n_per_elem;
    return (long long)e * n_per_elem + (valid ? local : n_per_elem - 1);
  };

  // The features of this wave tile wait in a 16 KiB scratch slot from layer 7 until the albedo head.  Small launches
  // (up to F3_WG_SLOTS workgroups) index the slot by workgroup; large ones by the PHYSICAL compute unit the workgroup runs on (XCC id + the SE / SH / CU
  // bits of HW_ID): a CU holds one workgroup of this kernel at a time (149 KiB of its 160 KiB LDS), so consecutive
  // workgroups of a CU overwrite the same 64 KiB and the whole launch touches 256 x 64 KiB = 16 MiB, which stays in the
  // XCD's L2 instead of streaming 512 B/point through HBM.  (Exclusive use of a slot follows from the LDS budget, not from
  // any dispatch-order assumption; tests/test_gpu_kernels.py::test_cu_slot_exclusive checks the id decoding on the device.)
  FwdScratch<false> ws;
  {
    long long wg = (long long)e * gridDim.x + blockIdx.x;
    if ((long long)gridDim.x * gridDim.y > F3_WG_SLOTS) wg = cu_slot_id();
    ws.rs = __builtin_amdgcn_make_buffer_rsrc(scratch + (wg * F3_WAVES + wave) * 16384ll, 0, 16384, 0x00020000);
  }

  // optional feature output through a buffer descriptor: an absent output (0 records) or a tail lane (offset past the
  // end) is dropped by the hardware's range check -- no branch inside the layer bodies (a branch splits the layer into
  // basic blocks and lets the compiler sink whole reductions across them, which costs hundreds of live registers)
  __amdgpu_buffer_rsrc_t feat_rs;
  int feat_off;
  {
    const long long base_pt = (long long)e * n_per_elem + (long long)blockIdx.x * F3_TILE + wave * WAVE_PTS;
    const long long left = n_per_elem - ((long long)blockIdx.x * F3_TILE + wave * WAVE_PTS);
    const int npts = feat_out == nullptr ? 0 : (left >= WAVE_PTS ? WAVE_PTS : (left > 0 ? (int)left : 0));
    feat_rs = __builtin_amdgcn_make_buffer_rsrc(feat_out + base_pt * C, 0, npts * C * 4, 0x00020000);
    feat_off = j * C * 4 + 16 * h;
  }

  // image sequence (ring slot = position & 1):
  //   0..6   forward layers 1..7          (mats 0..6)
  //   7..11  transposed layers 7..3       (mats 13..9)
  //   12,13  forward layers 1, 2          (mats 0, 1)
  //   14,15  transposed layers 2, 1       (mats 8, 7)
  //   16     albedo head                  (mat 14)
  // LDS-DMA through a buffer descriptor over the 16 images: SGPR image/chunk offset + 16 * lane -- no per-chunk 64-bit
  // address VGPRs (global_load_lds with flat pointers kept 16 address pairs alive across the whole kernel)
  const __amdgpu_buffer_rsrc_t img_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(mats), 0, NMAT * LB, 0x00020000);
  auto prefetch = [&](int pos) {
#ifdef OI_F3_ABL_NOPREFETCH
    if (pos >= 2) return;
#endif
    const int m = pos < 7 ? pos : (pos < 12 ? 20 - pos : (pos < 14 ? pos - 12 : (pos < 16 ? 22 - pos : 14)));
    // a wave copies 16 consecutive KiB, 4 KiB per (M0, soffset) setting: the instruction's immediate offset advances the
    // LDS and the global address alike, so four 1 KiB copies share one M0 / soffset pair (6 instead of 16 instructions per
    // 4 KiB; bit-identical results, -1.3 % kernel time)
#pragma unroll
    for (int q = 0; q < LB / 4096 / F3_WAVES; ++q) {
      const int c = (wave * (LB / 4096 / F3_WAVES) + q) * 4096;
      auto* dst = (__attribute__((address_space(3))) void*)(lds + F3_WBUF + (pos & 1) * LB + c);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(img_rs, dst, 16, o.l16, m * LB + c, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(img_rs, dst, 16, o.l16, m * LB + c, 1024, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(img_rs, dst, 16, o.l16, m * LB + c, 2048, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(img_rs, dst, 16, o.l16, m * LB + c, 3072, 0);
    }
  };
  auto lay = [&](int pos) {  // A-image lane bases of ring position pos
    LayOff y;
    y.wl = o.l16 + F3_WBUF + (pos & 1) * LB;
    y.wh = y.wl + 32768;
    y.wq = 0;
    y.f16 = 0;
    return y;
  };
  // lane base of FiLM layer l's rows: [A | B | G] x 128 floats (see the staging loop)
  auto film_base = [&](int l) { return o.h16 + F3_FILM + l * F3_FILM_ROW; };

#if OI_F3_BLOB
  float px, py, pz;
  {
    bool valid;
    const long long pt = point_of(valid);
    px = pts[pt * 3 + 0], py = pts[pt * 3 + 1], pz = pts[pt * 3 + 2];
  }
  {  // the element's blob (tables, FiLM rows in revolutions, row maxima: see film_blob_f3_kernel) -> LDS [0, F3_BLOB)
    const __amdgpu_buffer_rsrc_t blob_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(blob + (size_t)e * F3_BLOB), 0, F3_BLOB, 0x00020000);
#pragma unroll
    for (int q = 0; q < (F3_BLOB / 1024 + F3_WAVES - 1) / F3_WAVES; ++q) {
      const int c = q * F3_WAVES + wave;
      if (c < F3_BLOB / 1024)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(blob_rs, (__attribute__((address_space(3))) void*)(lds + c * 1024), 16, o.l16,
                                                 c * 1024, 0, 0);
    }
  }
  prefetch(0);
  // point + blob landed (everything older than image 0's 16 copies per wave: the counter retires in issue order), and visible
  // to every wave; image 0 stays in flight
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(LB / 1024 / F3_WAVES) : "memory");
#else
  prefetch(0);
  {  // small tables + the FiLM rows of all 9 layers, once.  The phase is formed in REVOLUTIONS so that the range reduction
     // is  r = phi - rint(phi)  (exact):  phi / 2pi = A * acc + B  with  A = gamma * 2^-k_image / 2pi  (2^-k: the power-of-
     // two scale baked into the layer's MFMA image) and  B = (gamma * bias + beta) / 2pi;  G = gamma * 2^-k_image is the
     // factor of cos(phi) in the reverse sweep.  Row 9: G7 * w_sigma (layer 7 emits the reverse sweep's first operand).
    float* tabs = reinterpret_cast<float*>(lds + F3_TABS);
    for (int i = tid; i < H_TABS_END; i += 64 * F3_WAVES) tabs[i] = hdr[i];
    float* film = reinterpret_cast<float*>(lds + F3_FILM);
    constexpr float INV_2PI = 0.15915494309189533577f;
    for (int i = tid; i < 9 * C; i += 64 * F3_WAVES) {
      const int l = i / C, f = i % C;
      const float gm = gamma[((size_t)e * 9 + l) * C + f];
      const float wsc = l == 0 ? 1.f : hdr[H_WSCALE + (l < NL_SDF ? l - 1 : 14)];
      const float G = gm * wsc;
      film[l * (F3_FILM_ROW / 4) + f] = G * INV_2PI;
      film[l * (F3_FILM_ROW / 4) + C + f] = fmaf(gm, hdr[H_BIAS + l * C + f], beta[((size_t)e * 9 + l) * C + f]) * INV_2PI;
      film[l * (F3_FILM_ROW / 4) + 2 * C + f] = G;
      if (l == 7) film[9 * (F3_FILM_ROW / 4) + f] = G * hdr[H_SIG + f];
    }
  }
  float px, py, pz;
  {
    bool valid;
    const long long pt = point_of(valid);
    px = pts[pt * 3 + 0], py = pts[pt * 3 + 1], pz = pts[pt * 3 + 2];
  }
  __syncthreads();  // tables visible (image 0 still in flight)
  {  // max |G_l| per layer (and of row 9): with the per-image bounds of the packed header they bound the growth of an
     // adjoint vector through one reverse layer, which is what lets its fp16 scale be chosen BEFORE it is complete
    const float* film = reinterpret_cast<const float*>(lds + F3_FILM);
    float* gmax = reinterpret_cast<float*>(lds + F3_GMAX);
    for (int l = wave; l < 10; l += F3_WAVES) {
      const int off = l * (F3_FILM_ROW / 4) + (l == 9 ? 0 : 2 * C);
      const float m = oi::wave_max(fmaxf(fabsf(film[off + lane]), fabsf(film[off + 64 + lane])));
      if (lane == 0) gmax[l] = m;
    }
  }
  __syncthreads();

#endif
#ifdef OI_F3_PROF
  unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_readcyclecounter();
  const unsigned long long tstart = tprev;
  pacc[4] = tstart - t_entry;  // prologue: tables, FiLM rows, row maxima, two barriers
#endif
  float act[64];                 // fp32 staging of an adjoint vector before its normalisation (reverse layers only)
  f32x16 acc[4];
#if OI_F3_ABL_PIECE & 16  // timing ablation: the epilogues read a plain VGPR instead of the accumulators
  float acc_dummy = 0.37f * lane;
  asm volatile("" : "+v"(acc_dummy));
#define F3_ACC(TB, I) acc_dummy
#else
#define F3_ACC(TB, I) acc[TB][I]
#endif
  Limbs AH, AL, BH, BL;          // two B-operand limb sets: a layer reads one and its epilogue fills the other
#if OI_F3_ABL_EPI
  for (int s_ = 0; s_ < 8; ++s_)
    for (int d_ = 0; d_ < 4; ++d_) {
      AH[s_][d_] = AL[s_][d_] = BH[s_][d_] = BL[s_][d_] = lane * 77u + s_;
      asm volatile("" : "+v"(AH[s_][d_]), "+v"(AL[s_][d_]), "+v"(BH[s_][d_]), "+v"(BL[s_][d_]));
    }
#endif
  Bank P0, P1, P2, P3;           // parked REDUCED PHASES r_l (revolutions, |r| <= 1/2) or one parked adjoint vector
  // FiLM / table rows of the epilogue group in flight, double-buffered by group parity: the rows of group g + 1 are
  // requested while group g is processed (an LDS round trip in front of every group's first FMA otherwise)
  struct Rows {
    f32x4 a, b, c, d;
  } rw[2];
  f32x4 fv;
  float run = 1.f, vmax = 0.f, sdf_part = 0.f;

  // phi (revolutions) -> reduced phase; v_sin_f32 / v_cos_f32 take revolutions and are specified on [-256, 256]: v_fract
  // (exact) keeps any phase inside that domain.  (Feeding the unreduced phase differs by at most 1 ulp inside the
  // domain, tools/dbg/sin_rev_probe.hip: the FAST flavour does that.)
  auto reduce = [&](float phi) { return (FAST || (OI_F3_ABL_PIECE & 2)) ? phi : __builtin_amdgcn_fractf(phi); };
  auto ld = [&](int imm, int base) { return lds_f4(lds, imm, base); };
#define ROW_A(FB, G) ld(grp_f0(G) * 4, FB)
#define ROW_B(FB, G) ld((C + grp_f0(G)) * 4, FB)
#define ROW_G(FB, G) ld((2 * C + grp_f0(G)) * 4, FB)
#define ROW_SIG(G) ld(F3_TABS + (H_SIG + grp_f0(G)) * 4, o.h16)

#define OI_PARK(BANK) [&](int g, int k, float r) { BANK[g][k] = to_acc(r); }
  auto no_park = [](int, int, float) {};

  // ---- layer 0 (K = 3) on the VALU: sin(phi_0) -> limb set (NH, NL); r_0 -> park
  auto layer0 = [&](Limbs& NH, Limbs& NL, auto&& park) {
    const int fb = film_base(0);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const f32x4 a4 = ROW_A(fb, g), b4 = ROW_B(fb, g);
      float sn[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 w = lds_f4(lds, F3_TABS + H_TAB0 * 4 + (grp_f0(g) + k) * 16, o.h64);
        const float u = fmaf(pz, w[2], fmaf(py, w[1], px * w[0]));
        const float r = reduce(fmaf(a4[k], u, b4[k]));
        sn[k] = f3_sin(r);
        park(g, k, r);
      }
      split_pair(sn[0], sn[1], NH[g >> 1][2 * (g & 1)], NL[g >> 1][2 * (g & 1)]);
      split_pair(sn[2], sn[3], NH[g >> 1][2 * (g & 1) + 1], NL[g >> 1][2 * (g & 1) + 1]);
      if (g & 1) __builtin_amdgcn_sched_barrier(0);
    }
  };

  // Epilogue pair (tb, rp) of a forward FiLM layer whose rows sit at lane base FB: sin(phi) -> next limb set, reduced
  // phase -> PARK.  NEXT: statement that requests the first group's rows of whatever epilogue follows this layer's last
  // group (into rw[0]).
#define OI_FWD_EPI(FB, NH, NL, PARK, NEXT)                                                                 \
  [&](int tb, int rp) {                                                                                    \
    const int g = tb * 4 + (rp >> 1), k = 2 * (rp & 1);                                                    \
    Rows& R = rw[g & 1];                                                                                   \
    if (k == 0) {                                                                                          \
      if (g < 15) {                                                                                        \
        rw[(g + 1) & 1].a = ROW_A(FB, g + 1);                                                              \
        rw[(g + 1) & 1].b = ROW_B(FB, g + 1);                                                              \
      } else {                                                                                             \
        NEXT;                                                                                              \
      }                                                                                                    \
    }                                                                                                      \
    const f32x2 ph = film_phase2(R.a, R.b, k, F3_ACC(tb, 2 * rp), F3_ACC(tb, 2 * rp + 1));                 \
    const float r0 = reduce(ph[0]), r1 = reduce(ph[1]);                                                    \
    PARK(g, k, r0);                                                                                        \
    PARK(g, k + 1, r1);                                                                                    \
    split_pair(f3_sin(r0), f3_sin(r1), NH[2 * tb + (rp >> 2)][rp & 3],       \
               NL[2 * tb + (rp >> 2)][rp & 3]);                                                            \
  }
  // Reverse sweep bookkeeping.  A reverse layer's input V_l = v_l * S_l is held as fp16 limbs with a power-of-two scale
  // S_l (run = 1 / S_l).  Its epilogue forms V_{l-1} = (W^T V_l) * G_{l-1} cos(phi_{l-1}) * sg and splits it into limbs
  // at once; sg is fixed BEFORE V_{l-1} exists from a bound:  |W^T V_l| <= bound(image) * max|V_l|  (H_BOUND: largest
  // absolute row sum of the scaled image) and |G cos| <= max|G_{l-1}|, so  U = bound * max|G| * max|V_l|  and
  // U * sg in [2^14, 2^15) can never overflow fp16.  max|V_l| is the MEASURED maximum (vmax, both lanes of the point),
  // so the slack of the bound (a few bits) does not compound from layer to layer: hi limb 11 bits, lo limb down to 2^-24
  // absolute = 2^-33 of the vector's maximum.
  float sg = 1.f;
  auto pow2_for = [&](float U) {  // -> sg with U * sg in [2^14, 2^15); run *= 1 / sg
    int eb = (__builtin_bit_cast(int, U) >> 23) & 0xff;
    eb = eb < 15 ? 15 : (eb > 253 ? 253 : eb);
    run *= __builtin_bit_cast(float, (eb - 14) << 23);
    sg = __builtin_bit_cast(float, (268 - eb) << 23);
  };
  const float* gmax = reinterpret_cast<const float*>(lds + F3_GMAX);
  // MID hook of reverse layer l (image mats[6 + l], parked layer l - 1): finalise max|V_l|, choose sg for V_{l-1}
#define OI_REV_MID(L)                                                          \
  [&]() {                                                                      \
    const float m = fmaxf(vmax, __shfl_xor(vmax, 32, 64));                     \
    pow2_for(m * (hdr[H_BOUND + 6 + (L)] * gmax[(L)-1]));                      \
    vmax = 0.f;                                                                \
  }
  // Epilogue pair of reverse layer l: V_{l-1} -> limb set (NH, NL); r_{l-1} from BANK, the G rows of layer l-1 at FB
#define OI_REV_EPI(FB, BANK, NH, NL, NEXT)                                                                 \
  [&](int tb, int rp) {                                                                                    \
    const int g = tb * 4 + (rp >> 1), k = 2 * (rp & 1);                                                    \
    Rows& R = rw[g & 1];                                                                                   \
    if (k == 0) {                                                                                          \
      if (g < 15) {                                                                                        \
        rw[(g + 1) & 1].c = ROW_G(FB, g + 1);                                                              \
      } else {                                                                                             \
        NEXT;                                                                                              \
      }                                                                                                    \
    }                                                                                                      \
    const f32x2 cs = {f3_cos(from_acc(BANK[g][k])), f3_cos(from_acc(BANK[g][k + 1]))};                     \
    const f32x2 v = pk_mul(pk_mul_s(f32x2{F3_ACC(tb, 2 * rp), F3_ACC(tb, 2 * rp + 1)}, sg),                \
                           pk_mul(f32x2{R.c[k], R.c[k + 1]}, cs));                                         \
    vmax = max3_abs(vmax, v[0], v[1]);                                                                     \
    split_pair(v[0], v[1], NH[2 * tb + (rp >> 2)][rp & 3], NL[2 * tb + (rp >> 2)][rp & 3]);                \
  }

  const LayOff y1 = lay(0), y2 = lay(1), y3 = lay(2), y4 = lay(3), y5 = lay(4), y6 = lay(5), y7 = lay(6);
  const int F0 = film_base(0), F1 = film_base(1), F2 = film_base(2), F3 = film_base(3), F4 = film_base(4),
            F5 = film_base(5), F6 = film_base(6), F7 = film_base(7), F8 = film_base(8), F9 = film_base(9);
#define NEXT_AB(FB) (rw[0].a = ROW_A(FB, 0), rw[0].b = ROW_B(FB, 0))
#define NEXT_G(FB) (rw[0].c = ROW_G(FB, 0))

  // ================= forward, layers 0..7 =================
  layer0(AH, AL, no_park);
  NEXT_AB(F1);
  F3_T(0);
  ring_sync();  // image 0 resident
  F3_T(2);
  prefetch(1);
  auto park0 = OI_PARK(P0);
  auto park1 = OI_PARK(P1);
  auto park2 = OI_PARK(P2);
  auto park3 = OI_PARK(P3);
  auto e1 = OI_FWD_EPI(F1, BH, BL, no_park, NEXT_AB(F2));
  stream_layer(lds, y1, AH, AL, acc, NoTail(), NoMid(), e1);
  F3_T(1);
  ring_sync();
  F3_T(2);
  prefetch(2);
  auto e2 = OI_FWD_EPI(F2, AH, AL, no_park, NEXT_AB(F3));
  stream_layer(lds, y2, BH, BL, acc, e1, NoMid(), e2);
  F3_T(1);
  ring_sync();
  F3_T(2);
  prefetch(3);
  auto e3 = OI_FWD_EPI(F3, BH, BL, park0, NEXT_AB(F4));
  stream_layer(lds, y3, AH, AL, acc, e2, NoMid(), e3);
  F3_T(1);
  ring_sync();
  F3_T(2);
  prefetch(4);
  auto e4 = OI_FWD_EPI(F4, AH, AL, park1, NEXT_AB(F5));
  stream_layer(lds, y4, BH, BL, acc, e3, NoMid(), e4);
  F3_T(1);
  ring_sync();
  F3_T(2);
  prefetch(5);
  auto e5 = OI_FWD_EPI(F5, BH, BL, park2, NEXT_AB(F6));
  stream_layer(lds, y5, AH, AL, acc, e4, NoMid(), e5);
  F3_T(1);
  ring_sync();
  F3_T(2);
  prefetch(6);
  auto e6 = OI_FWD_EPI(F6, AH, AL, park3,
                       (NEXT_AB(F7), rw[0].c = ROW_A(F9, 0), rw[0].d = ROW_SIG(0)));
  stream_layer(lds, y6, BH, BL, acc, e5, NoMid(), e6);
  F3_T(1);
  ring_sync();
  F3_T(2);
  prefetch(7);
  // layer 7: features a8 = sin(phi7) leave through the scratch slot (+ feat_out), sdf = a8 . wsig + bsig on the fly,
  // and the reverse sweep's first operand v7 = wsig * G7 * cos(phi7) is formed in place (r7 is never parked)
  // |v7| <= max|G7 w_sigma| (row 9): its fp16 scale is known before the first value exists
  pow2_for(gmax[9] * 2.0f);
  const float sg7 = sg;
  auto e7 = [&](int tb, int rp) {
    const int g = tb * 4 + (rp >> 1), k = 2 * (rp & 1);
    Rows& R = rw[g & 1];
    if (k == 0) {
      if (g < 15) {
        Rows& N = rw[(g + 1) & 1];
        N.a = ROW_A(F7, g + 1);
        N.b = ROW_B(F7, g + 1);
        N.c = ROW_A(F9, g + 1);
        N.d = ROW_SIG(g + 1);
      } else {
        NEXT_G(F6);
      }
    }
    const f32x2 ph = film_phase2(R.a, R.b, k, F3_ACC(tb, 2 * rp), F3_ACC(tb, 2 * rp + 1));
    f32x2 cs;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float r = reduce(ph[i]);
      const float sn = f3_sin(r);
      fv[k + i] = sn;
      sdf_part = fmaf(sn, R.d[k + i], sdf_part);
      cs[i] = f3_cos(r);
    }
    const f32x2 v = pk_mul(pk_mul_s(f32x2{R.c[k], R.c[k + 1]}, sg7), cs);
    vmax = max3_abs(vmax, v[0], v[1]);
    split_pair(v[0], v[1], BH[2 * tb + (rp >> 2)][rp & 3], BL[2 * tb + (rp >> 2)][rp & 3]);
    if (k == 2) {
      ws.store(0, g, o, fv);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, fv), feat_rs, feat_off + grp_f0(g) * 4, 0, 0);
    }
  };
  stream_layer(lds, y7, AH, AL, acc, e6, NoMid(), e7);
  F3_T(1);
  ring_sync();
  F3_T(2);
  prefetch(8);

  // ================= reverse, layers 7..3 =================
  // position 7..11 = transposed layers 7..3.  Every layer runs the previous layer's block-3 pairs during its own block 0.
  auto r7 = OI_REV_EPI(F6, P3, AH, AL, NEXT_G(F5));   // V6 = g7 * G6 cos(phi6)
  stream_layer(lds, lay(7), BH, BL, acc, e7, OI_REV_MID(7), r7);
  {
    sdf_part += __shfl_xor(sdf_part, 32, 64);  // complete since e7's last pair (inside the layer above)
    const float sdf_v = sdf_part + *reinterpret_cast<const float*>(lds + F3_TABS + (H_SIG + C) * 4);
    bool valid;
    const long long pt = point_of(valid);
    if (valid && h == 0) sdf_out[pt] = sdf_v;
  }
  F3_T(1);
  ring_sync();
  F3_T(2);
  prefetch(9);
  auto r6 = OI_REV_EPI(F5, P2, BH, BL, NEXT_G(F4));
  stream_layer(lds, lay(8), AH, AL, acc, r7, OI_REV_MID(6), r6);
  F3_T(1);
  ring_sync();
  F3_T(2);
  prefetch(10);
  auto r5 = OI_REV_EPI(F4, P1, AH, AL, NEXT_G(F3));
  stream_layer(lds, lay(9), BH, BL, acc, r6, OI_REV_MID(5), r5);
  F3_T(1);
  ring_sync();
  F3_T(2);
  prefetch(11);
  auto r4 = OI_REV_EPI(F3, P0, BH, BL, (void)0);      // V3 = g4 * G3 cos(phi3)
  stream_layer(lds, lay(10), AH, AL, acc, r5, OI_REV_MID(4), r4);
  F3_T(1);
  ring_sync();
  F3_T(2);
  prefetch(12);
  // layer 3: g3 (in units of 1 / run) is parked while layers 0..2 are recomputed.  The park is an asm statement: hipcc pads
  // no MFMA-result -> reader hazard for it, so the value first passes through a VALU instruction the compiler does know
  // (a multiply by an opaque 1.0).  max|g3| for the scale of V2 is tracked on the way.
  float one = 1.0f;
  asm volatile("" : "+v"(one));
  auto pg3 = [&](int tb, int rp) {
    const int g = tb * 4 + (rp >> 1), k = 2 * (rp & 1);
    const f32x2 a = pk_mul_s(f32x2{F3_ACC(tb, 2 * rp), F3_ACC(tb, 2 * rp + 1)}, one);
    vmax = max3_abs(vmax, a[0], a[1]);
    P0[g][k] = to_acc(a[0]);
    P0[g][k + 1] = to_acc(a[1]);
  };
  stream_layer(lds, lay(11), BH, BL, acc, r4, [&]() { vmax = 0.f; }, pg3);   // (V3's own maximum is not needed)
  F3_T(1);
  ring_sync();
  F3_T(2);
  prefetch(13);
  run_tail(pg3);
  F3_T(3);

  // ================= forward again, layers 0..2 =================
  const LayOff z1 = lay(12), z2 = lay(13);
  NEXT_AB(F1);
  layer0(AH, AL, park1);  // r0
  F3_T(0);
  auto f1 = OI_FWD_EPI(F1, BH, BL, park2, (NEXT_AB(F2), NEXT_G(F2)));  // r1
  stream_layer(lds, z1, AH, AL, acc, NoTail(), NoMid(), f1);
  F3_T(1);
  ring_sync();
  F3_T(2);
  prefetch(14);
  // layer 2: V2 = g3 * G2 cos(phi2) * sg (the sine is not needed any more); |g3| <= its measured maximum
  pow2_for(fmaxf(vmax, __shfl_xor(vmax, 32, 64)) * gmax[2]);
  vmax = 0.f;
  const float sg2 = sg;
  auto f2 = [&](int tb, int rp) {
    const int g = tb * 4 + (rp >> 1), k = 2 * (rp & 1);
    Rows& R = rw[g & 1];
    if (k == 0) {
      if (g < 15) {
        Rows& N = rw[(g + 1) & 1];
        N.a = ROW_A(F2, g + 1);
        N.b = ROW_B(F2, g + 1);
        N.c = ROW_G(F2, g + 1);
      } else {
        NEXT_G(F1);
      }
    }
    const f32x2 ph = film_phase2(R.a, R.b, k, F3_ACC(tb, 2 * rp), F3_ACC(tb, 2 * rp + 1));
    const f32x2 cs = {f3_cos(reduce(ph[0])), f3_cos(reduce(ph[1]))};
    const f32x2 v = pk_mul(pk_mul_s(f32x2{from_acc(P0[g][k]), from_acc(P0[g][k + 1])}, sg2), pk_mul(f32x2{R.c[k], R.c[k + 1]}, cs));
    vmax = max3_abs(vmax, v[0], v[1]);
    split_pair(v[0], v[1], AH[2 * tb + (rp >> 2)][rp & 3], AL[2 * tb + (rp >> 2)][rp & 3]);
  };
  stream_layer(lds, z2, BH, BL, acc, f1, NoMid(), f2);
  F3_T(1);
  ring_sync();
  F3_T(2);
  prefetch(15);

  // ================= reverse, layers 2..0 =================
  auto r2 = OI_REV_EPI(F1, P2, BH, BL, NEXT_G(F0));   // V1 = g2 * G1 cos(phi1)
  stream_layer(lds, lay(14), AH, AL, acc, f2, OI_REV_MID(2), r2);
  F3_T(1);
  ring_sync();
  F3_T(2);
  prefetch(16);
  // layer 1: v0 = g1 * G0 cos(phi0) stays fp32 (layer 0's transposed product runs on the VALU)
  auto r1 = [&](int tb, int rp) {
    const int g = tb * 4 + (rp >> 1), k = 2 * (rp & 1);
    Rows& R = rw[g & 1];
    if (k == 0 && g < 15) rw[(g + 1) & 1].c = ROW_G(F0, g + 1);
    const f32x2 cs = {f3_cos(from_acc(P1[g][k])), f3_cos(from_acc(P1[g][k + 1]))};
    const f32x2 v = pk_mul(f32x2{F3_ACC(tb, 2 * rp), F3_ACC(tb, 2 * rp + 1)}, pk_mul(f32x2{R.c[k], R.c[k + 1]}, cs));
    act[4 * g + k] = v[0];
    act[4 * g + k + 1] = v[1];
  };
  stream_layer(lds, lay(15), BH, BL, acc, r2, [&]() {}, r1);
  F3_T(1);
  run_tail(r1);
  F3_T(3);
  // features back for the albedo head (requested before the layer-0 gradient: their latency hides under it)
  f32x4 fback[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) fback[g] = ws.load(0, g, o);
  float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 w = lds_f4(lds, F3_TABS + H_TAB0 * 4 + (grp_f0(g) + k) * 16, o.h64);
      const float v = act[4 * g + k];
      gx = fmaf(v, w[0], gx);
      gy = fmaf(v, w[1], gy);
      gz = fmaf(v, w[2], gz);
    }
    if (g & 1) __builtin_amdgcn_sched_barrier(0);
  }
  gx += __shfl_xor(gx, 32, 64);
  gy += __shfl_xor(gy, 32, 64);
  gz += __shfl_xor(gz, 32, 64);
  gx *= run;  // identical in both lanes of a point (every max was taken over the pair)
  gy *= run;
  gz *= run;
  bool valid;
  const long long pt = point_of(valid);
  if (valid && h == 0) {
    grad_out[pt * 3 + 0] = gx;
    grad_out[pt * 3 + 1] = gy;
    grad_out[pt * 3 + 2] = gz;
  }

  {
    // ---- albedo head (always evaluated: a gradient-only variant of this kernel makes hipcc spill 150 registers, and the
    // only caller without the albedo is the stand-alone ShapeNetwork.gradient API):
    // sigmoid(Wrgb sin(gv * (Wv [feat, grad] + bv) + bv') + brgb)   (fields.py:89-101)
    ring_sync();  // image 16 resident
#pragma unroll
    for (int g = 0; g < 16; ++g)
#pragma unroll
      for (int k = 0; k < 4; ++k) act[4 * g + k] = fback[g][k];
    const LayOff yc = lay(16);
    // the accumulators carry the image scale 2^k: bring the rank-3 gradient term to the same scale
    const float cs3 = 1.0f / hdr[H_WSCALE + 14];
    const float vx = gx * cs3, vy = gy * cs3, vz = gz * cs3;
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    f32x4 w0, w1, w2;
    split_limbs(act, AH, AL);
    NEXT_AB(F8);
    auto ec = [&](int tb, int rp) {
      const int g = tb * 4 + (rp >> 1), k = 2 * (rp & 1);
      Rows& R = rw[g & 1];
      if (k == 0) {
        if (g < 15) {
          rw[(g + 1) & 1].a = ROW_A(F8, g + 1);
          rw[(g + 1) & 1].b = ROW_B(F8, g + 1);
        }
        w0 = lds_f4(lds, F3_TABS + (H_RGB + 0 * C + grp_f0(g)) * 4, o.h16);
        w1 = lds_f4(lds, F3_TABS + (H_RGB + 1 * C + grp_f0(g)) * 4, o.h16);
        w2 = lds_f4(lds, F3_TABS + (H_RGB + 2 * C + grp_f0(g)) * 4, o.h16);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f32x4 w = lds_f4(lds, F3_TABS + H_TABV * 4 + (grp_f0(g) + k + i) * 16, o.h64);
        const float u = F3_ACC(tb, 2 * rp + i) + fmaf(vz, w[2], fmaf(vy, w[1], vx * w[0]));
        const float sn = f3_sin(reduce(fmaf(R.a[k + i], u, R.b[k + i])));
        r0 = fmaf(sn, w0[k + i], r0);
        r1 = fmaf(sn, w1[k + i], r1);
        r2 = fmaf(sn, w2[k + i], r2);
      }
    };
    F3_T(5);
    stream_layer(lds, yc, AH, AL, acc, NoTail(), NoMid(), ec);
    F3_T(1);
    run_tail(ec);
    F3_T(3);
    r0 += __shfl_xor(r0, 32, 64);
    r1 += __shfl_xor(r1, 32, 64);
    r2 += __shfl_xor(r2, 32, 64);
    if (valid && h == 0 && rgb_out != nullptr) {
      const float* brgb = reinterpret_cast<const float*>(lds + F3_TABS + (H_RGB + 3 * C) * 4);
      rgb_out[pt * 3 + 0] = oi::sigmoidf_(r0 + brgb[0]);
      rgb_out[pt * 3 + 1] = oi::sigmoidf_(r1 + brgb[1]);
      rgb_out[pt * 3 + 2] = oi::sigmoidf_(r2 + brgb[2]);
    }
  }
#ifdef OI_F3_PROF
  F3_T(5);
  if (lane == 0) {
    for (int i = 0; i < 6; ++i) atomicAdd(&oi_prof3[i], pacc[i]);
    atomicAdd(&oi_prof3[6], __builtin_readcyclecounter() - tstart);
    atomicAdd(&oi_prof3[7], 1ull);
    atomicAdd(&oi_prof3[8], __builtin_readcyclecounter() - t_entry);          // with [9]: the shader clock in the kernel
    atomicAdd(&oi_prof3[9], __builtin_amdgcn_s_memrealtime() - rt_entry);
  }
#endif
#undef OI_PARK
#undef OI_FWD_EPI
#undef OI_REV_EPI
#undef OI_REV_MID
#undef NEXT_AB
#undef NEXT_G
#undef ROW_A
#undef ROW_B
#undef ROW_G
#undef ROW_SIG
}

// The per-element blob of sdf_mlp_full3_kernel: LDS bytes [0, F3_BLOB) as the kernel's own staging loop formed them (the
// phase in REVOLUTIONS: phi / 2pi = A * acc + B with A = gamma * 2^-k_image / 2pi, B = (gamma * bias + beta) / 2pi; G = gamma *
// 2^-k_image; row 9 = G7 * w_sigma; the header tables; max |G_l| per layer and of row 9).  One block per batch element.
__global__ void __launch_bounds__(128) film_blob_f3_kernel(const char* __restrict__ packed, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, char* __restrict__ blob) {
  __shared__ float red[8];
  const int e = blockIdx.x, l = blockIdx.y, tid = threadIdx.x;   // a workgroup per (element, FiLM layer)
  const float* hdr = reinterpret_cast<const float*>(packed);
  float* out = reinterpret_cast<float*>(blob + (size_t)e * F3_BLOB);
  // (oif3::blob_layer: the same code prep_render_kernel's FiLM workgroups run)
  const float gm = tid < C ? gamma[((size_t)e * 9 + l) * C + tid] : 0.f, bt = tid < C ? beta[((size_t)e * 9 + l) * C + tid] : 0.f;
  oif3::blob_layer(hdr, out, l, tid, gm, bt, red);
}

size_t full3_slot_bytes(int B, long long n_per_elem) {
  const long long wgs = (long long)B * oi::cdiv(n_per_elem, F3_TILE);
  return (size_t)(wgs > F3_WG_SLOTS ? F3_CU_SLOTS : wgs) * F3_WAVES * 16384;
}

template <bool FAST>
int launch_full3(const float* pts, const char* pk, const float* gamma, const float* beta, float* sdf, float* grad,
                 float* rgb, float* feat, char* scratch, int B, long long n, bool blob_ready, hipStream_t st) {
  dim3 grid(oi::cdiv(n, F3_TILE), B), block(64 * F3_WAVES);
  auto k = sdf_mlp_full3_kernel<FAST>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS);
  char* blob = scratch + full3_slot_bytes(B, n);   // behind the feature slots
#if OI_F3_BLOB
  if (!blob_ready) hipLaunchKernelGGL(film_blob_f3_kernel, dim3(B, 9), dim3(128), 0, st, pk, gamma, beta, blob);
#endif
  hipLaunchKernelGGL(k, grid, block, F3_LDS, st, pts, pk, gamma, beta, sdf, grad, rgb, feat, scratch, blob, n);
  return oi::check_launch("oi_sdf_mlp_fwd(full3)");
}

}  // namespace

namespace oimlp {

size_t full3_scratch_bytes(int B, long long n_per_elem) {   // feature slots + the per-element blobs
  return full3_slot_bytes(B, n_per_elem) + (size_t)B * F3_BLOB;
}

size_t full3_blob_offset(int B, long long n_per_elem) { return full3_slot_bytes(B, n_per_elem); }

int launch_full3_f16x3(const float* pts, const void* packed, const float* gamma, const float* beta, float* sdf,
                       float* grad, float* rgb, float* feat, void* scratch, int B, long long n, int fast_trig, bool blob_ready,
                       hipStream_t st) {
  const char* pk = reinterpret_cast<const char*>(packed);
  char* sc = reinterpret_cast<char*>(scratch);
  return fast_trig ? launch_full3<true>(pts, pk, gamma, beta, sdf, grad, rgb, feat, sc, B, n, blob_ready, st)
                   : launch_full3<false>(pts, pk, gamma, beta, sdf, grad, rgb, feat, sc, B, n, blob_ready, st);
}

}  // namespace oimlp

namespace {
// every workgroup marks its CU slot on entry and clears it on exit; a slot seen busy on entry = two workgroups shared it
__global__ void __launch_bounds__(256) cu_slot_selftest_kernel(int* busy, int* clashes, int* used, int spin) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int slot = cu_slot_id();
  if (threadIdx.x == 0) {
    if (atomicAdd(busy + slot, 1) != 0) atomicAdd(clashes, 1);
    atomicOr(used + slot, 1);
  }
  volatile float* l = reinterpret_cast<volatile float*>(lds);
  float acc = 0.f;
  for (int i = 0; i < spin; ++i) {
    l[threadIdx.x] = acc + i;
    acc += l[(threadIdx.x + 1) & 255];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (acc == 12345.678f) atomicAdd(clashes, 1 << 20);
    atomicAdd(busy + slot, -1);
  }
}
}  // namespace

extern "C" int oi_selftest_cu_slots(int* busy, int* clashes, int* used, int n_workgroups, int spin, oi_stream_t stream) {
  OI_REQUIRE(busy && clashes && used && n_workgroups > 0, "oi_selftest_cu_slots: bad argument");
  auto k = cu_slot_selftest_kernel;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS);
  hipLaunchKernelGGL(k, dim3(n_workgroups), dim3(256), F3_LDS, oi::as_stream(stream), busy, clashes, used, spin);
  return oi::check_launch("oi_selftest_cu_slots");
}

#ifdef OI_F3_PROF
extern "C" int oi_prof3_read(unsigned long long* out, int reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(oi_prof3), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(oi_prof3), z, sizeof(z));
  }
  return 0;
}
#endif
