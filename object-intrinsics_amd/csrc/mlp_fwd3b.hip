// FiLM-SIREN forward with the analytic d sdf/dx and the albedo head, register-resident, BF16 operands (gfx950).
//
// The operand mode BASELINE.json's configs[1] names (64x64, 128 samples/ray, bf16): ONE v_mfma_f32_32x32x16_bf16 per
// layer product and k-step, fp32 accumulation, FiLM phase / sin / cos in fp32.  Same contract as sdf_mlp_kernel<BF16, FULL>
// in mlp.hip (reference src/models/fields.py:49-77, 89-101, 104-122;
// src/third_party/stylesdf/volume_renderer.py:50-61) and the residency of mlp_fwd3.hip: nothing per point crosses HBM but
// the 12 B in and the 28 B (+ optional 512 B feature vector) out -- the round-1 kernel this replaces streamed 4.6 KB per
// point of gamma*cos(phi) through memory (2.47 GB per 524,288-point launch, profiles/r3_pmc_*_bf16.txt) and ran at the
// copy rate of the box.
//
//  * ONE wavefront per SIMD (4 waves = 128 points per workgroup and CU) with the whole 512-entry register file.
//  * cos(phi_l) of layers 0..6 is parked in the AGPR half as PACKED fp16 pairs (|cos| <= 1: 2^-12 absolute, below the
//    bf16 operand rounding): 7 x 32 = 224 registers -- all seven layers fit, so the reverse sweep needs NO recomputation
//    (15 layer products; the f16x3 kernel parks fp32 phases, four layers at a time, and runs 17).
//  * The features a_8 = sin(phi_7) wait for the albedo head as the bf16 B-operand limbs the head consumes (32 registers):
//    no scratch slot, no scratch buffer at all.
//  * 32 KiB images: a FOUR-slot LDS ring, three images in flight (a layer lasts ~0.6 us, an LDS-DMA round trip ~1 us).
//  * Every layer product is formed output block by output block; the FiLM / sin / cos (or cos-multiply) work of block
//    t-1 is issued between block t's MFMAs.  The first MFMA of a block takes an inline-constant zero accumulator.
//
// Two kernels share the machinery above:
//   sdf_mlp_full3p_kernel (round 5, the default)   per-element images diag(gamma) W built once per call by film_images_b_kernel
//                                                   (the accumulator IS the phase: no FiLM arithmetic in the epilogues) and a
//                                                   per-element table blob fetched by LDS-DMA -- see the section further down;
//   sdf_mlp_full3b_kernel (round 4)                 shared bf16(W) images + FiLM rows staged per tile; kept behind the run-time
//                                                   switch OI_BF16_PRESCALE=0 as the same-box A/B reference of the former.
#include <algorithm>
#include <type_traits>

#include "mlp_common.h"

namespace {

using namespace oimlp;

constexpr int B3_WAVES = 4;
constexpr int B3_TILE = B3_WAVES * WAVE_PTS;  // 128 points per workgroup
constexpr int LBB = 32768;                    // bytes of one bf16 image
constexpr int B3_NSLOT = 4;                   // ring slots
// LDS: FiLM rows [10][A 128 | B 128 | G 128] floats, small tables, image ring
constexpr int B3_FILM = 0;
constexpr int B3_FILM_ROW = 3 * C * 4;                      // bytes per FiLM layer
constexpr int B3_TABS = B3_FILM + 10 * B3_FILM_ROW;         // 15360
constexpr int B3_WBUF = B3_TABS + H_TABS_END * 4;           // 21632
// Two small A images for the K = 128 -> 3 products at the ends of the network (d sdf/dx = W0^T v0; rgb = Wrgb sin(phi_v)): rows
// 0..2 = the bf16 hi limb of the three output rows, 3..5 their lo limbs, 6..7 zero; [k-step 8][lane half 2][row 8][8 x bf16].
// (Lanes read row (lane & 7): MFMA output rows >= 8 then hold copies nobody reads.)
constexpr int B3_SIMG = B3_WBUF + B3_NSLOT * LBB;           // 152,704
constexpr int SIMG_BYTES = 8 * 2 * 8 * 16;                  // 2 KiB each
constexpr int B3_LDS = B3_SIMG + 2 * SIMG_BYTES;            // 156,800 of the CU's 163,840 bytes
// The three-input-column products (layer 0, the albedo head's gradient columns), the three-output-row products (d sdf/dx, rgb)
// on the matrix cores instead of the VALU: 1 = on (default), 0 = the round-4a VALU forms (A/B switch)
#ifndef OI_B3_MFMA_EDGES
#define OI_B3_MFMA_EDGES 1
#endif

typedef unsigned Limb[8][4];   // bf16 B operand of one layer: [k-step][dword d] = act indices 8 s + 2 d, 8 s + 2 d + 1
typedef unsigned BankB[16][2];  // one parked 128-vector of cos(phi) as fp16 pairs: [group g][pair] <-> act[4 g + 2 pair (+1)]

__device__ __forceinline__ unsigned to_acc_u(unsigned v) {
  unsigned a;
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v));
  return a;
}
__device__ __forceinline__ unsigned from_acc_u(unsigned a) {
  unsigned v;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
  return v;
}
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {  // v_cvt_pk_bf16_f32 (round to nearest even)
  const bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned pk_f16(float a, float b) {  // v_cvt_pk_f16_f32
  const f16x2 v = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, v);
}
// x * (fp16 half of c): v_fma_mix_f32 reads the half in place (hipcc's own choice is v_cvt_f32_f16 + v_mul_f32).  `x` must be
// the result of an instruction the compiler knows (it pads no MFMA-result hazard for an inline-asm reader).
__device__ __forceinline__ float mul_lo(float x, unsigned c) {
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(x), "v"(c));
  return r;
}
__device__ __forceinline__ float mul_hi(float x, unsigned c) {
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(x), "v"(c));
  return r;
}

struct NoTailB {
  __device__ __forceinline__ void operator()(int, int) const {}
};
template <class T> struct is_no_tail_b { static constexpr bool value = false; };
template <> struct is_no_tail_b<NoTailB> { static constexpr bool value = true; };

// A fragments are requested this many k-steps ahead of their MFMA (a window is ~40 cycles here, an LDS round trip of a lone
// wave 64-130)
#ifndef OI_B3_ADIST
#define OI_B3_ADIST 3
#endif
#ifndef OI_B3_GROUPS
#define OI_B3_GROUPS 1
#endif
#ifndef OI_B3_WINSTEPS
#define OI_B3_WINSTEPS 1
#endif
#ifndef OI_B3_VALU_PER_MFMA
#define OI_B3_VALU_PER_MFMA 8
#endif
// timing ablations (results garbage): 1 = no epilogue work, 2 = additionally no A-fragment reads after the first
#ifndef OI_B3_ABL
#define OI_B3_ABL 0
#endif

// One layer product of the stream: acc = W_img . B, output block t outer, one scheduling window per k-step = 1 MFMA + the
// epilogue pair that hides behind it:
//   block 0, k-steps 0..5   TAIL(3, rp): the PREVIOUS layer's block-3 epilogue pairs (8 of them, spread 2 1 1 2 1 1);
//                           they complete THIS layer's B limbs for k-steps 6 and 7 just before those are consumed
//   blocks 1..3             EPI(t - 1, s): this layer's epilogue pair s of the block that has just completed
// This layer's own block-3 pairs are left to the caller: the next layer's TAIL, or run_tail_b().
// An epilogue pair (tb, rp) consumes accumulator slots 2 rp, 2 rp + 1 of block tb = act indices 16 tb + 2 rp (+1).
struct NoPost {
  __device__ __forceinline__ void operator()(int, f32x16&) const {}
};
// POST(t, acc[t]) runs right after block t's eight MFMAs (the albedo head adds its three gradient columns there: one more MFMA)
// INIT (per-element images, sdf_mlp_full3p_kernel): the accumulators start from the FiLM offset row instead of zero, so that the
// finished accumulator IS the phase.  acc[0] must hold init(0) on entry; init(t + 1) is loaded straight into acc[t + 1] half a
// block ahead (its previous contents died with the previous layer's epilogue of that block, or with this layer's TAIL).
struct NoInit {
  __device__ __forceinline__ f32x16 operator()(int) const { return f32x16{}; }
};
template <class T> struct is_no_init_b { static constexpr bool value = false; };
template <> struct is_no_init_b<NoInit> { static constexpr bool value = true; };
template <class TAIL, class EPI, class POST = NoPost, class INIT = NoInit>
__device__ __forceinline__ void stream_layer_b(const char* lds, int wl, const Limb& bh, f32x16 (&acc)[4], TAIL&& tail,
                                               EPI&& epi, POST&& post = POST(), INIT&& init = INIT()) {
  constexpr bool HAS_TAIL = !is_no_tail_b<std::remove_cv_t<std::remove_reference_t<TAIL>>>::value;
  constexpr bool HAS_INIT = !is_no_init_b<std::remove_cv_t<std::remove_reference_t<INIT>>>::value;
  constexpr int AD = OI_B3_ADIST;
  f32x4 a[AD + 1];
#pragma unroll
  for (int i = 0; i < AD; ++i) a[i] = lds_f4(lds, i * 1024, wl);
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int cur = t * 8 + s, nxt = cur + AD;
      if (nxt < 32 && !(OI_B3_ABL == 2)) a[nxt % (AD + 1)] = lds_f4(lds, nxt * 1024, wl);
      int npairs = 0;
#if OI_B3_ABL
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
      const bf16x8 w = __builtin_bit_cast(bf16x8, a[(OI_B3_ABL == 2 ? cur % AD : cur) % (AD + 1)]);
      const u32x4 ub = {bh[s][0], bh[s][1], bh[s][2], bh[s][3]};
      const bf16x8 v = __builtin_bit_cast(bf16x8, ub);
      if (HAS_INIT && s == 4 && t < 3) acc[t + 1] = init(t + 1);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, v, (s == 0 && !HAS_INIT) ? zero : acc[t], 0, 0, 0);
      // OI_B3_WINSTEPS k-steps per scheduling window: with 2, two epilogue pairs (independent chains) share a window and fill
      // each other's wait states (accvgpr read -> use, v_fma_mix -> v_cvt_pk: an s_nop each when a pair is alone)
      if (OI_B3_WINSTEPS == 1 || (s % OI_B3_WINSTEPS) == OI_B3_WINSTEPS - 1) {
        if (OI_B3_GROUPS && npairs >= 1) {
#pragma unroll
          for (int q = 0; q + 1 < OI_B3_WINSTEPS; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // an MFMA opens each part, the epilogue fills its shadow
            __builtin_amdgcn_sched_group_barrier(0x002, OI_B3_VALU_PER_MFMA, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 64, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    post(t, acc[t]);
  }
}
// block 3's epilogue with nothing to hide behind: two pairs (four independent chains) per window
template <class EPI>
__device__ __forceinline__ void run_tail_b(EPI&& epi) {
  if (OI_B3_ABL) return;
#pragma unroll
  for (int rp = 0; rp < 8; ++rp) {
    epi(3, rp);
    if (rp & 1) __builtin_amdgcn_sched_barrier(0);
  }
}

// -DOI_B3_PROF: per-phase shader-clock accounting
#ifdef OI_B3_PROF
__device__ unsigned long long oi_prof3b[1024][16];  // replicated by workgroup: same-address atomics of 300k waves would
                                                     // dominate the memory system (and with it the prologue and the DMA waits)
#define B3_T(i)                                                  \
  do {                                                           \
    const unsigned long long t_ = __builtin_readcyclecounter();  \
    pacc[i] += t_ - tprev;                                       \
    tprev = t_;                                                  \
  } while (0)
#else
#define B3_T(i)
#endif

// wait until at most KEEP of this wave's vector-memory operations are outstanding (they retire in issue order: everything
// older -- the image this layer reads -- has landed), then rendezvous: image resident for every wave, the slot of the
// layer before free
// OI_B3_BARE_BARRIER (round 5): __syncthreads() carries workgroup-scope fences, and for those hipcc drains EVERY outstanding
// vector-memory operation in front of the s_barrier (`s_waitcnt vmcnt(0) lgkmcnt(0)` right behind the counted wait below: read off
// the ISA) -- the two younger images included, i.e. the ring never had more than the image it was waiting for in flight.  A bare
// s_barrier behind the counted wait keeps them in flight; LDS visibility of the awaited image follows from the wait itself plus
// the barrier (MI355X_MICROARCH.md: "nothing orders a ds_read behind a pending LDS-DMA except the issuing wave's covering vmcnt
// (plus a barrier, for other waves' reads)").
#ifndef OI_B3_BARE_BARRIER
#define OI_B3_BARE_BARRIER 1
#endif
template <int KEEP>
__device__ __forceinline__ void ring_sync_b() {
#if OI_B3_BARE_BARRIER
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(KEEP) : "memory");
#else
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
  __syncthreads();
#endif
}

template <bool FAST>
__global__ void __launch_bounds__(64 * B3_WAVES) __attribute__((amdgpu_waves_per_eu(1, 1)))
sdf_mlp_full3b_kernel(const float* __restrict__ pts, const char* __restrict__ packed, const float* __restrict__ gamma,
                      const float* __restrict__ beta, float* __restrict__ sdf_out, float* __restrict__ grad_out,
                      float* __restrict__ rgb_out, float* __restrict__ feat_out, long long n_per_elem) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
#ifdef OI_B3_PROF
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
  o.l16hi = 0;
  asm volatile("" : "+v"(o.h16), "+v"(o.h64), "+v"(o.l16));

  auto point_of = [&](bool& valid) {
    int jj = lane & 31;
    asm volatile("" : "+v"(jj));
    const long long local = (long long)blockIdx.x * B3_TILE + wave * WAVE_PTS + jj;
    valid = local < n_per_elem;
    return (long long)e * n_per_elem + (valid ? local : n_per_elem - 1);
  };

  // optional feature output through a buffer descriptor: an absent output (0 records) or a tail lane (offset past the
  // end) is dropped by the hardware's range check -- no branch inside the layer bodies
  __amdgpu_buffer_rsrc_t feat_rs;
  int feat_off;
  {
    const long long base_pt = (long long)e * n_per_elem + (long long)blockIdx.x * B3_TILE + wave * WAVE_PTS;
    const long long left = n_per_elem - ((long long)blockIdx.x * B3_TILE + wave * WAVE_PTS);
    const int npts = feat_out == nullptr ? 0 : (left >= WAVE_PTS ? WAVE_PTS : (left > 0 ? (int)left : 0));
    feat_rs = __builtin_amdgcn_make_buffer_rsrc(feat_out + base_pt * C, 0, npts * C * 4, 0x00020000);
    feat_off = j * C * 4 + 16 * h;
  }

  // image sequence (ring slot = position & 3):
  //   0..6    forward layers 1..7       (mats 0..6)
  //   7..13   transposed layers 7..1    (mats 13..7)
  //   14      albedo head               (mat 14)
  const __amdgpu_buffer_rsrc_t img_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(mats), 0, NMAT * LBB, 0x00020000);
  auto prefetch = [&](int pos) {  // 8 KiB per wave: 4 KiB per (M0, soffset) setting, the immediate advances both addresses
    const int m = pos < 7 ? pos : (pos < 14 ? 20 - pos : 14);
#pragma unroll
    for (int q = 0; q < LBB / 4096 / B3_WAVES; ++q) {
      const int c = (wave * (LBB / 4096 / B3_WAVES) + q) * 4096;
      auto* dst = (__attribute__((address_space(3))) void*)(lds + B3_WBUF + (pos & (B3_NSLOT - 1)) * LBB + c);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(img_rs, dst, 16, o.l16, m * LBB + c, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(img_rs, dst, 16, o.l16, m * LBB + c, 1024, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(img_rs, dst, 16, o.l16, m * LBB + c, 2048, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(img_rs, dst, 16, o.l16, m * LBB + c, 3072, 0);
    }
  };
  constexpr int DMA_PER_IMAGE = LBB / 1024 / B3_WAVES;  // vector-memory instructions per wave and image: 8
  // lane base of ring position pos, laundered: the A-fragment reads then are <this VGPR> + a 16-bit immediate (left to
  // itself hipcc folds the slot base into the immediate, overflows its 16 bits and forms a new address on the VALU per read)
  auto lay = [&](int pos) {
    int b = o.l16 + B3_WBUF + (pos & (B3_NSLOT - 1)) * LBB;
    asm volatile("" : "+v"(b));
    return b;
  };
  auto film_base = [&](int l) { return o.h16 + B3_FILM + l * B3_FILM_ROW; };

  float px, py, pz;
  {
    bool valid;
    const long long pt = point_of(valid);
    px = pts[pt * 3 + 0], py = pts[pt * 3 + 1], pz = pts[pt * 3 + 2];
  }
  {  // small tables + the FiLM rows of all 9 layers, once.  The phase is formed in REVOLUTIONS: phi / 2pi = A * acc + B with
     // A = gamma / 2pi and B = (gamma * bias + beta) / 2pi;  G = gamma is the factor of cos(phi) in the reverse sweep.
     // Row 9: G7 * w_sigma (layer 7 emits the reverse sweep's first operand).
    float* tabs = reinterpret_cast<float*>(lds + B3_TABS);
    for (int i = tid; i < H_TABS_END; i += 64 * B3_WAVES) tabs[i] = hdr[i];
    float* film = reinterpret_cast<float*>(lds + B3_FILM);
    constexpr float INV_2PI = 0.15915494309189533577f;
    for (int i = tid; i < 9 * C; i += 64 * B3_WAVES) {
      const int l = i / C, f = i % C;
      const float gm = gamma[((size_t)e * 9 + l) * C + f];
      film[l * (B3_FILM_ROW / 4) + f] = gm * INV_2PI;
      film[l * (B3_FILM_ROW / 4) + C + f] = fmaf(gm, hdr[H_BIAS + l * C + f], beta[((size_t)e * 9 + l) * C + f]) * INV_2PI;
      film[l * (B3_FILM_ROW / 4) + 2 * C + f] = gm;
      if (l == 7) film[9 * (B3_FILM_ROW / 4) + f] = gm * hdr[H_SIG + f];
    }
  }
#if OI_B3_MFMA_EDGES
  {  // the two 3-row A images: entry (image, k-step s, half hh, row i): 8 bf16 = M[i][feat_of(8 s + j, hh)], j = 0..7
    const int img = tid >> 7, s_ = (tid >> 4) & 7, hh = (tid >> 3) & 1, i = tid & 7;
    const int c = i < 3 ? i : i - 3;
    unsigned d[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[2];
#pragma unroll
      for (int e_ = 0; e_ < 2; ++e_) {
        const int f = feat_of(8 * s_ + 2 * q + e_, hh);
        const float w = i < 6 ? (img == 0 ? hdr[H_TAB0 + 4 * f + c] : hdr[H_RGB + c * C + f]) : 0.f;
        const float whi = (float)(__bf16)w;
        v[e_] = i < 3 ? whi : w - whi;
      }
      d[q] = pk_bf16(v[0], v[1]);
    }
    *reinterpret_cast<u32x4*>(lds + B3_SIMG + img * SIMG_BYTES + ((s_ * 2 + hh) * 8 + i) * 16) = u32x4{d[0], d[1], d[2], d[3]};
  }
#endif
  // (the prologue's loads are complete: hipcc waits with vmcnt(0) for them before the LDS writes above -- the image DMA is
  //  issued behind them so that those waits do not drain it)
  prefetch(0);
  prefetch(1);
  prefetch(2);
  __syncthreads();  // tables visible

#ifdef OI_B3_PROF
  unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_readcyclecounter();
  const unsigned long long tstart = tprev;
  pacc[4] = tstart - t_entry;  // prologue
#endif
  f32x16 acc[4];
  Limb AH, BH, CH;               // two B-operand limb sets in turn + the features a_8 for the albedo head
#if OI_B3_ABL
  for (int s_ = 0; s_ < 8; ++s_)
    for (int d_ = 0; d_ < 4; ++d_) {
      AH[s_][d_] = BH[s_][d_] = CH[s_][d_] = lane * 77u + s_;
      asm volatile("" : "+v"(AH[s_][d_]), "+v"(BH[s_][d_]), "+v"(CH[s_][d_]));
    }
#endif
  BankB P0, P1, P2, P3, P4, P5, P6;  // cos(phi_l), l = 0..6, fp16 pairs in the AGPR half
  // FiLM / table rows of the epilogue groups in flight: the rows of group g + 2 are requested while group g is processed
  struct Rows {
    f32x4 a, b, c, d;
  } rw[4];
  f32x4 fv;
  float sdf_part = 0.f;

  auto reduce = [&](float phi) { return FAST ? phi : __builtin_amdgcn_fractf(phi); };
  auto ld = [&](int imm, int base) { return lds_f4(lds, imm, base); };
#define ROW_A(FB, G) ld(grp_f0(G) * 4, FB)
#define ROW_B(FB, G) ld((C + grp_f0(G)) * 4, FB)
#define ROW_G(FB, G) ld((2 * C + grp_f0(G)) * 4, FB)
#define ROW_SIG(G) ld(B3_TABS + (H_SIG + grp_f0(G)) * 4, o.h16)
  // row requests by epilogue kind: the rows of group G of the layer whose rows sit at lane base FB -> rw[G & 3].  (A ring of
  // four: the block-3 pairs of a layer run inside the next layer, and while groups 14 / 15 are processed the NEXT epilogue's
  // groups 0 / 1 are requested -- 16 = 0 mod 4 keeps the two sequences on one ring.)
#define REQ_AB(FB) [&](int g_) { rw[g_ & 3].a = ROW_A(FB, g_); rw[g_ & 3].b = ROW_B(FB, g_); }
#define REQ_G(FB) [&](int g_) { rw[g_ & 3].c = ROW_G(FB, g_); }
  auto req_none = [](int) {};

  // ---- layer 0 (K = 3) on the VALU: sin(phi_0) -> limb set NH; cos(phi_0) -> P0
  auto layer0 = [&](Limb& NH) {
    const int fb = film_base(0);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const f32x4 a4 = ROW_A(fb, g), b4 = ROW_B(fb, g);
      float sn[4], cs[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 w = lds_f4(lds, B3_TABS + H_TAB0 * 4 + (grp_f0(g) + k) * 16, o.h64);
        const float u = fmaf(pz, w[2], fmaf(py, w[1], px * w[0]));
        const float r = reduce(fmaf(a4[k], u, b4[k]));
        sn[k] = __builtin_amdgcn_sinf(r);
        cs[k] = __builtin_amdgcn_cosf(r);
      }
      NH[g >> 1][2 * (g & 1)] = pk_bf16(sn[0], sn[1]);
      NH[g >> 1][2 * (g & 1) + 1] = pk_bf16(sn[2], sn[3]);
      P0[g][0] = to_acc_u(pk_f16(cs[0], cs[1]));
      P0[g][1] = to_acc_u(pk_f16(cs[2], cs[3]));
      if (g & 1) __builtin_amdgcn_sched_barrier(0);
    }
  };

  // B fragment of a three-column product: the 3-vector v as bf16 hi + lo limbs in the K slots
  //   half 0: (vh.x vh.y vh.z | vl.x vl.y vl.z | 0 0)   half 1: (vh.x vh.y vh.z | 0 ..)
  // against A rows  half 0: (wh.x wh.y wh.z | wh.x wh.y wh.z | 0 0)   half 1: (wl.x wl.y wl.z | 0 ..):  vh wh + vl wh + vh wl
  auto frag3_b = [&](float vx, float vy, float vz) {
    const float hx = (float)(__bf16)vx, hy = (float)(__bf16)vy, hz = (float)(__bf16)vz;
    const unsigned d0 = pk_bf16(hx, hy);
    const unsigned d1 = h == 0 ? pk_bf16(hz, vx - hx) : pk_bf16(hz, 0.f);
    const unsigned d2 = h == 0 ? pk_bf16(vy - hy, vz - hz) : 0u;
    return __builtin_bit_cast(bf16x8, u32x4{d0, d1, d2, 0u});
  };
  // A fragment of output block t from a [128][4] fp32 table (w.x w.y w.z 0 per feature) at LDS offset TAB
  auto frag3_a = [&](int tab, int t) {
    const f32x4 w = lds_f4(lds, tab + t * 32 * 16, 16 * (lane & 31));
    const float hx = (float)(__bf16)w[0], hy = (float)(__bf16)w[1], hz = (float)(__bf16)w[2];
    const float ax = h == 0 ? hx : w[0] - hx, ay = h == 0 ? hy : w[1] - hy, az = h == 0 ? hz : w[2] - hz;
    const unsigned d0 = pk_bf16(ax, ay);
    const unsigned d1 = h == 0 ? pk_bf16(az, hx) : pk_bf16(az, 0.f);
    const unsigned d2 = h == 0 ? pk_bf16(hy, hz) : 0u;
    return __builtin_bit_cast(bf16x8, u32x4{d0, d1, d2, 0u});
  };
  // A fragment (k-step s) of one of the two small 3-row images
  auto simg = [&](int img, int s_) {
    return __builtin_bit_cast(bf16x8, lds_f4(lds, B3_SIMG + img * SIMG_BYTES + s_ * 256, 128 * h + 16 * (lane & 7)));
  };
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // rows 0..2 (hi limb) + rows 3..5 (lo limb) of a 3-row product: row = (reg & 3) + 8 (reg >> 2) + 4 h
  auto rows3 = [&](const f32x16& a, float& r0, float& r1, float& r2) {
    const float u0 = __shfl_xor(a[0], 32, 64), u1 = __shfl_xor(a[1], 32, 64);  // rows 4, 5 sit in the other lane half
    r0 = a[0] + a[3];
    r1 = a[1] + u0;
    r2 = a[2] + u1;   // (valid in the half-0 lanes: they hold rows 0..3 and receive rows 4, 5)
  };

  // Epilogue pair (tb, rp) of a forward FiLM layer whose rows sit at lane base FB: sin(phi) -> next limb set NH,
  // cos(phi) -> BANK.  REQ: requester of this layer's rows; NEXT: requester of the rows of whatever epilogue follows.
#define OI_FWD_EPI(FB, NH, BANK, NEXT)                                                                     \
  [&](int tb, int rp) {                                                                                    \
    const int g = tb * 4 + (rp >> 1), k = 2 * (rp & 1);                                                    \
    const Rows& R = rw[g & 3];                                                                             \
    if (k == 0) {                                                                                          \
      if (g + 2 < 16) {                                                                                    \
        REQ_AB(FB)(g + 2);                                                                                 \
      } else {                                                                                             \
        NEXT(g + 2 - 16);                                                                                  \
      }                                                                                                    \
    }                                                                                                      \
    const float r0 = reduce(fmaf(R.a[k], acc[tb][2 * rp], R.b[k]));                                        \
    const float r1 = reduce(fmaf(R.a[k + 1], acc[tb][2 * rp + 1], R.b[k + 1]));                            \
    NH[2 * tb + (rp >> 2)][rp & 3] = pk_bf16(__builtin_amdgcn_sinf(r0), __builtin_amdgcn_sinf(r1));        \
    BANK[g][k >> 1] = to_acc_u(pk_f16(__builtin_amdgcn_cosf(r0), __builtin_amdgcn_cosf(r1)));              \
  }
  // Epilogue pair of the transposed product of layer l: V_{l-1} = g_l * G_{l-1} * cos(phi_{l-1}) -> limb set NH; the cosines
  // come from BANK (layer l - 1), the G rows of layer l - 1 sit at FB.  (acc * G is formed by an instruction the compiler
  // knows before the asm multiply reads it: see mul_lo.)
#define OI_REV_EPI(FB, BANK, NH, NEXT)                                                                     \
  [&](int tb, int rp) {                                                                                    \
    const int g = tb * 4 + (rp >> 1), k = 2 * (rp & 1);                                                    \
    const Rows& R = rw[g & 3];                                                                             \
    if (k == 0) {                                                                                          \
      if (g + 2 < 16) {                                                                                    \
        REQ_G(FB)(g + 2);                                                                                  \
      } else {                                                                                             \
        NEXT(g + 2 - 16);                                                                                  \
      }                                                                                                    \
    }                                                                                                      \
    const unsigned c2 = from_acc_u(BANK[g][k >> 1]);                                                       \
    const float v0 = mul_lo(acc[tb][2 * rp] * R.c[k], c2), v1 = mul_hi(acc[tb][2 * rp + 1] * R.c[k + 1], c2); \
    NH[2 * tb + (rp >> 2)][rp & 3] = pk_bf16(v0, v1);                                                      \
  }

  const int F0 = film_base(0), F1 = film_base(1), F2 = film_base(2), F3 = film_base(3), F4 = film_base(4),
            F5 = film_base(5), F6 = film_base(6), F7 = film_base(7), F8 = film_base(8), F9 = film_base(9);

  // ================= forward, layers 0..7 =================
#if OI_B3_MFMA_EDGES
  {  // layer 0 (K = 3) on the matrix cores: one MFMA per output block, then the ordinary forward epilogue
    REQ_AB(F0)(0);
    REQ_AB(F0)(1);
    const bf16x8 bp = frag3_b(px, py, pz);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag3_a(B3_TABS + H_TAB0 * 4, t), bp, zero16, 0, 0, 0);
    auto e0 = OI_FWD_EPI(F0, AH, P0, REQ_AB(F1));
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int rp = 0; rp < 8; ++rp) {
        e0(t, rp);
        if (rp & 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
#else
  layer0(AH);
  REQ_AB(F1)(0);
  REQ_AB(F1)(1);
#endif
  B3_T(0);
  ring_sync_b<2 * DMA_PER_IMAGE>();  // image 0 resident (1 and 2 may still be in flight)
  B3_T(2);
  prefetch(3);
  auto e1 = OI_FWD_EPI(F1, BH, P1, REQ_AB(F2));
  stream_layer_b(lds, lay(0), AH, acc, NoTailB(), e1);
  B3_T(1);
  ring_sync_b<2 * DMA_PER_IMAGE>();
  B3_T(2);
  prefetch(4);
  auto e2 = OI_FWD_EPI(F2, AH, P2, REQ_AB(F3));
  stream_layer_b(lds, lay(1), BH, acc, e1, e2);
  B3_T(1);
  ring_sync_b<2 * DMA_PER_IMAGE>();
  B3_T(2);
  prefetch(5);
  auto e3 = OI_FWD_EPI(F3, BH, P3, REQ_AB(F4));
  stream_layer_b(lds, lay(2), AH, acc, e2, e3);
  B3_T(1);
  ring_sync_b<2 * DMA_PER_IMAGE>();
  B3_T(2);
  prefetch(6);
  auto e4 = OI_FWD_EPI(F4, AH, P4, REQ_AB(F5));
  stream_layer_b(lds, lay(3), BH, acc, e3, e4);
  B3_T(1);
  ring_sync_b<2 * DMA_PER_IMAGE>();
  B3_T(2);
  prefetch(7);
  auto e5 = OI_FWD_EPI(F5, BH, P5, REQ_AB(F6));
  stream_layer_b(lds, lay(4), AH, acc, e4, e5);
  B3_T(1);
  ring_sync_b<2 * DMA_PER_IMAGE>();
  B3_T(2);
  prefetch(8);
  // layer 7's epilogue needs four rows per group: A7, B7, row 9 = G7 * w_sigma, w_sigma
  auto req7 = [&](int g_) {
    Rows& N = rw[g_ & 3];
    N.a = ROW_A(F7, g_);
    N.b = ROW_B(F7, g_);
    N.c = ROW_A(F9, g_);
    N.d = ROW_SIG(g_);
  };
  auto e6 = OI_FWD_EPI(F6, AH, P6, req7);
  stream_layer_b(lds, lay(5), BH, acc, e5, e6);
  B3_T(1);
  ring_sync_b<2 * DMA_PER_IMAGE>();
  B3_T(2);
  prefetch(9);
  // layer 7: features a8 = sin(phi7) -> limb set CH (+ feat_out), sdf = a8 . wsig + bsig on the fly, and the reverse
  // sweep's first operand v7 = wsig * G7 * cos(phi7) is formed in place (cos(phi7) is never parked)
  auto e7 = [&](int tb, int rp) {
    const int g = tb * 4 + (rp >> 1), k = 2 * (rp & 1);
    const Rows& R = rw[g & 3];
    if (k == 0) {
      if (g + 2 < 16) req7(g + 2);
      else REQ_G(F6)(g + 2 - 16);
    }
    float sn[2], v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float r = reduce(fmaf(R.a[k + i], acc[tb][2 * rp + i], R.b[k + i]));
      sn[i] = __builtin_amdgcn_sinf(r);
      fv[k + i] = sn[i];
      sdf_part = fmaf(sn[i], R.d[k + i], sdf_part);
      v[i] = R.c[k + i] * __builtin_amdgcn_cosf(r);
    }
    CH[2 * tb + (rp >> 2)][rp & 3] = pk_bf16(sn[0], sn[1]);
    BH[2 * tb + (rp >> 2)][rp & 3] = pk_bf16(v[0], v[1]);
    if (k == 2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, fv), feat_rs, feat_off + grp_f0(g) * 4, 0, 0);
  };
  stream_layer_b(lds, lay(6), AH, acc, e6, e7);
  B3_T(1);
  // (16 feature stores per wave were issued behind image 9's DMA: more younger operations than the count, never fewer)
  ring_sync_b<2 * DMA_PER_IMAGE>();
  B3_T(2);
  prefetch(10);

  // ================= reverse, layers 7..1 =================
  auto r7 = OI_REV_EPI(F6, P6, AH, REQ_G(F5));   // V6 = g7 * G6 cos(phi6)
  stream_layer_b(lds, lay(7), BH, acc, e7, r7);
  {
    sdf_part += __shfl_xor(sdf_part, 32, 64);  // complete since e7's last pair (inside the layer above)
    const float sdf_v = sdf_part + *reinterpret_cast<const float*>(lds + B3_TABS + (H_SIG + C) * 4);
    bool valid;
    const long long pt = point_of(valid);
    if (valid && h == 0) sdf_out[pt] = sdf_v;
  }
  B3_T(1);
  ring_sync_b<2 * DMA_PER_IMAGE>();
  B3_T(2);
  prefetch(11);
  auto r6 = OI_REV_EPI(F5, P5, BH, REQ_G(F4));
  stream_layer_b(lds, lay(8), AH, acc, r7, r6);
  B3_T(1);
  ring_sync_b<2 * DMA_PER_IMAGE>();
  B3_T(2);
  prefetch(12);
  auto r5 = OI_REV_EPI(F4, P4, AH, REQ_G(F3));
  stream_layer_b(lds, lay(9), BH, acc, r6, r5);
  B3_T(1);
  ring_sync_b<2 * DMA_PER_IMAGE>();
  B3_T(2);
  prefetch(13);
  auto r4 = OI_REV_EPI(F3, P3, BH, REQ_G(F2));
  stream_layer_b(lds, lay(10), AH, acc, r5, r4);
  B3_T(1);
  ring_sync_b<2 * DMA_PER_IMAGE>();
  B3_T(2);
  prefetch(14);
  auto r3 = OI_REV_EPI(F2, P2, AH, REQ_G(F1));
  stream_layer_b(lds, lay(11), BH, acc, r4, r3);
  B3_T(1);
  ring_sync_b<2 * DMA_PER_IMAGE>();  // image 12 resident; 13 and 14 in flight, nothing more to request
  B3_T(2);
  auto r2 = OI_REV_EPI(F1, P1, BH, REQ_G(F0));   // V1 = g2 * G1 cos(phi1)
  stream_layer_b(lds, lay(12), AH, acc, r3, r2);
  B3_T(1);
  ring_sync_b<DMA_PER_IMAGE>();      // image 13 resident
  B3_T(2);
#if OI_B3_MFMA_EDGES
  // transposed layer 1: V0 = g1 * G0 cos(phi0) as limbs; then d sdf/dx = W0^T V0 (K = 128 -> 3 rows) as eight MFMAs against
  // the small image
  auto r1 = OI_REV_EPI(F0, P0, AH, req_none);
  stream_layer_b(lds, lay(13), BH, acc, r2, r1);
  B3_T(1);
  run_tail_b(r1);
  B3_T(3);
  float gx, gy, gz;
  {
    f32x16 a0 = zero16;
#pragma unroll
    for (int s_ = 0; s_ < 8; ++s_) {
      const u32x4 ub = {AH[s_][0], AH[s_][1], AH[s_][2], AH[s_][3]};
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(simg(0, s_), __builtin_bit_cast(bf16x8, ub), a0, 0, 0, 0);
    }
    rows3(a0, gx, gy, gz);
    gx = __shfl(gx, lane & 31, 64);  // both halves of the point need the gradient (the albedo head's B fragment)
    gy = __shfl(gy, lane & 31, 64);
    gz = __shfl(gz, lane & 31, 64);
  }
#else
  // transposed layer 1: v0 = g1 * G0 cos(phi0) stays fp32 (layer 0's transposed product, K = 128 -> 3, runs on the VALU)
  float act[64];
  auto r1 = [&](int tb, int rp) {
    const int g = tb * 4 + (rp >> 1), k = 2 * (rp & 1);
    const Rows& R = rw[g & 3];
    if (k == 0 && g + 2 < 16) REQ_G(F0)(g + 2);
    const unsigned c2 = from_acc_u(P0[g][k >> 1]);
    act[4 * g + k] = mul_lo(acc[tb][2 * rp] * R.c[k], c2);
    act[4 * g + k + 1] = mul_hi(acc[tb][2 * rp + 1] * R.c[k + 1], c2);
  };
  stream_layer_b(lds, lay(13), BH, acc, r2, r1);
  B3_T(1);
  run_tail_b(r1);
  B3_T(3);
  float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 w = lds_f4(lds, B3_TABS + H_TAB0 * 4 + (grp_f0(g) + k) * 16, o.h64);
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
#endif
  bool valid;
  const long long pt = point_of(valid);
  if (valid && h == 0) {
    grad_out[pt * 3 + 0] = gx;
    grad_out[pt * 3 + 1] = gy;
    grad_out[pt * 3 + 2] = gz;
  }

  {
    // ---- albedo head: sigmoid(Wrgb sin(gv * (Wv [feat, grad] + bv) + bv') + brgb)   (fields.py:89-101)
    ring_sync_b<0>();  // image 14 resident
    float r0 = 0.f, r1c = 0.f, r2c = 0.f;
    REQ_AB(F8)(0);
    REQ_AB(F8)(1);
#if OI_B3_MFMA_EDGES
    // the head's three gradient columns are a ninth MFMA of every output block; its activations sin(phi_v) go on as bf16 limbs
    // (AH: free) and rgb = Wrgb sin(phi_v) is eight MFMAs against the second small image
    const bf16x8 bg3 = frag3_b(gx, gy, gz);
    auto post = [&](int t, f32x16& a) { a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag3_a(B3_TABS + H_TABV * 4, t), bg3, a, 0, 0, 0); };
    auto ec = [&](int tb, int rp) {
      const int g = tb * 4 + (rp >> 1), k = 2 * (rp & 1);
      const Rows& R = rw[g & 3];
      if (k == 0 && g + 2 < 16) REQ_AB(F8)(g + 2);
      const float s0 = __builtin_amdgcn_sinf(reduce(fmaf(R.a[k], acc[tb][2 * rp], R.b[k])));
      const float s1 = __builtin_amdgcn_sinf(reduce(fmaf(R.a[k + 1], acc[tb][2 * rp + 1], R.b[k + 1])));
      AH[2 * tb + (rp >> 2)][rp & 3] = pk_bf16(s0, s1);
    };
    B3_T(5);
    stream_layer_b(lds, lay(14), CH, acc, NoTailB(), ec, post);
    B3_T(1);
    run_tail_b(ec);
    {
      f32x16 a0 = zero16;
#pragma unroll
      for (int s_ = 0; s_ < 8; ++s_) {
        const u32x4 ub = {AH[s_][0], AH[s_][1], AH[s_][2], AH[s_][3]};
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(simg(1, s_), __builtin_bit_cast(bf16x8, ub), a0, 0, 0, 0);
      }
      rows3(a0, r0, r1c, r2c);
    }
    B3_T(3);
#else
    f32x4 w0, w1, w2;
    auto ec = [&](int tb, int rp) {
      const int g = tb * 4 + (rp >> 1), k = 2 * (rp & 1);
      const Rows& R = rw[g & 3];
      if (k == 0) {
        if (g + 2 < 16) REQ_AB(F8)(g + 2);
        w0 = lds_f4(lds, B3_TABS + (H_RGB + 0 * C + grp_f0(g)) * 4, o.h16);
        w1 = lds_f4(lds, B3_TABS + (H_RGB + 1 * C + grp_f0(g)) * 4, o.h16);
        w2 = lds_f4(lds, B3_TABS + (H_RGB + 2 * C + grp_f0(g)) * 4, o.h16);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f32x4 w = lds_f4(lds, B3_TABS + H_TABV * 4 + (grp_f0(g) + k + i) * 16, o.h64);
        const float u = acc[tb][2 * rp + i] + fmaf(gz, w[2], fmaf(gy, w[1], gx * w[0]));
        const float sn = __builtin_amdgcn_sinf(reduce(fmaf(R.a[k + i], u, R.b[k + i])));
        r0 = fmaf(sn, w0[k + i], r0);
        r1c = fmaf(sn, w1[k + i], r1c);
        r2c = fmaf(sn, w2[k + i], r2c);
      }
    };
    B3_T(5);
    stream_layer_b(lds, lay(14), CH, acc, NoTailB(), ec);
    B3_T(1);
    run_tail_b(ec);
    B3_T(3);
    r0 += __shfl_xor(r0, 32, 64);
    r1c += __shfl_xor(r1c, 32, 64);
    r2c += __shfl_xor(r2c, 32, 64);
#endif
    if (valid && h == 0 && rgb_out != nullptr) {
      const float* brgb = reinterpret_cast<const float*>(lds + B3_TABS + (H_RGB + 3 * C) * 4);
      rgb_out[pt * 3 + 0] = oi::sigmoidf_(r0 + brgb[0]);
      rgb_out[pt * 3 + 1] = oi::sigmoidf_(r1c + brgb[1]);
      rgb_out[pt * 3 + 2] = oi::sigmoidf_(r2c + brgb[2]);
    }
  }
#ifdef OI_B3_PROF
  B3_T(5);
  if (lane == 0) {
    unsigned long long* pr = oi_prof3b[(blockIdx.x * 4 + wave) & 1023];
    for (int i = 0; i < 6; ++i) atomicAdd(&pr[i], pacc[i]);
    atomicAdd(&pr[6], __builtin_readcyclecounter() - tstart);
    atomicAdd(&pr[7], 1ull);
    atomicAdd(&pr[8], __builtin_readcyclecounter() - t_entry);  // with [9]: the shader clock in the kernel
    atomicAdd(&pr[9], __builtin_amdgcn_s_memrealtime() - rt_entry);
  }
#endif
#undef OI_FWD_EPI
#undef OI_REV_EPI
#undef REQ_AB
#undef REQ_G
#undef ROW_A
#undef ROW_B
#undef ROW_G
#undef ROW_SIG
}


// =====================================================================================================================
// Per-element images (round 5, second half): sdf_mlp_full3p_kernel
//
// phi_l / 2pi = (gamma_l / 2pi) (W_l a + b_l) + beta_l / 2pi depends on gamma_l and W_l only through diag(gamma_l) W_l, and so does
// the reverse sweep: g_l = W_l^T (gamma_l * cos(phi_l) * g_{l+1}) = (diag(gamma_l) W_l)^T (cos(phi_l) * g_{l+1}).  One launch per
// MLP call (film_images_b_kernel: 15 x 32 KiB per batch element, built from the plain fp32 copy of the weights behind the
// packed images) rounds diag(gamma_l / 2pi) W_l and (diag(gamma_l) W_l)^T to bf16 ONCE -- the same 2^-9 per operand the mode
// carries, applied to gamma W instead of W.  In the kernel
//   * the accumulators of a forward product start from the row B2_l = (gamma_l b_l + beta_l) / 2pi (four ds_read_b128 per
//     output block, straight into the accumulator registers) instead of an inline zero: the finished accumulator IS the phase
//     in revolutions.  A forward epilogue pair is 2 sin + 2 cos + 2 packs + the AGPR write: no FMA, no FiLM row reads;
//   * a reverse epilogue pair is the AGPR read + 2 v_fma_mix (accumulator x parked fp16 cosine) + the pack: no gamma row, no
//     multiply (the mixes are C: see mul_pair_c).
// Image order in the per-element buffer = ring position: 0..6 forward layers 1..7, 7..13 transposed layers 7..1, 14 albedo head.
// =====================================================================================================================
constexpr int NIMG_P = 15;
// The per-element tables arrive as ONE blob the builder writes behind the 15 images (P_BLOB bytes, LDS layout = blob layout): a
// tile's prologue is 15 LDS-DMA copies of 1 KiB instead of ~40 dependent global loads, 15 KB of ds_writes and the small-image
// arithmetic per workgroup (4,096 workgroups per C2 launch did that work again and again).
constexpr int P_FILM = 0;                          // [9][128] B2_l
constexpr int P_FILM_ROW = C * 4;
constexpr int P_TABS = P_FILM + 9 * P_FILM_ROW;    // 4608: the header tables, TAB0 rows x gamma_0 / 2pi, TABV rows x gamma_v / 2pi
constexpr int P_SIMG = P_TABS + H_TABS_END * 4;    // 10880: the two 3-row A images
constexpr int P_BLOB = 15360;                      // 10880 + 4096 = 14976, padded to 15 KiB
constexpr int P_WBUF = P_BLOB;
constexpr int P_LDS = P_WBUF + B3_NSLOT * LBB;     // 146,432
constexpr size_t P_ELEM_BYTES = (size_t)NIMG_P * LBB + P_BLOB;   // scratch per batch element
static_assert(P_SIMG + 2 * SIMG_BYTES <= P_BLOB, "blob layout");

// one 16-byte A fragment (8 bf16) per thread: fragment u = lane + 64 (s + 8 t) of image `pos` of batch element e
__global__ void __launch_bounds__(256) film_images_b_kernel(const char* __restrict__ packed, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, char* __restrict__ out) {
  const int pos = blockIdx.y, e = blockIdx.z;
  const int u = blockIdx.x * 256 + threadIdx.x;
  const int lane = u & 63, s_ = (u >> 6) & 7, t = u >> 9;
  const int hh = lane >> 5, row = 32 * t + (lane & 31);
  const float* plain = reinterpret_cast<const float*>(packed + plain_off(OI_PREC_BF16));
  const float* gm = gamma + (size_t)e * 9 * C;
  constexpr float INV_2PI = 0.15915494309189533577f;
  if (pos == NIMG_P) {  // the table blob (P_BLOB bytes behind the images), two dwords per thread
    const float* hdr = reinterpret_cast<const float*>(packed);
    unsigned* blob = reinterpret_cast<unsigned*>(out + (size_t)e * P_ELEM_BYTES + (size_t)NIMG_P * LBB);
    for (int i = u; i < P_BLOB / 4; i += C * C / 8) {
      unsigned w = 0u;
      if (i < 9 * C) {                                  // B2_l = (gamma_l b_l + beta_l) / 2pi
        w = __builtin_bit_cast(unsigned, fmaf(gm[i], hdr[H_BIAS + i], beta[(size_t)e * 9 * C + i]) * INV_2PI);
      } else if (i < 9 * C + H_TABS_END) {              // header tables; the three-column tables carry gamma / 2pi per row
        const int k = i - 9 * C;
        float v = hdr[k];
        if (k < H_SIG) v *= gm[k >> 2] * INV_2PI;
        else if (k >= H_TABV && k < H_RGB) v *= gm[8 * C + ((k - H_TABV) >> 2)] * INV_2PI;
        w = __builtin_bit_cast(unsigned, v);
      } else if (i < P_SIMG / 4 + 2 * SIMG_BYTES / 4) {  // small images: [img][k-step 8][half 2][row 8][dword 4]
        const int k = i - P_SIMG / 4;
        const int q = k & 3, r = (k >> 2) & 7, h2 = (k >> 5) & 1, ks = (k >> 6) & 7, img = k >> 9;
        const int c = r < 3 ? r : r - 3;
        float v2[2];
#pragma unroll
        for (int e_ = 0; e_ < 2; ++e_) {
          const int f = feat_of(8 * ks + 2 * q + e_, h2);
          const float wv = r < 6 ? (img == 0 ? hdr[H_TAB0 + 4 * f + c] * gm[f] : hdr[H_RGB + c * C + f]) : 0.f;
          const float whi = (float)(__bf16)wv;
          v2[e_] = r < 3 ? whi : wv - whi;
        }
        w = pk_bf16(v2[0], v2[1]);
      }
      blob[i] = w;
    }
    return;
  }
  float v[8];
  if (pos < 7 || pos == 14) {  // diag(gamma_l / 2pi) W_l: element [row][k]
    const int m = pos < 7 ? pos : 7, l = pos < 7 ? pos + 1 : 8;
    const float sc = gm[l * C + row] * INV_2PI;
    const float* src = plain + ((size_t)m * C + row) * C;
#pragma unroll
    for (int ip = 0; ip < 8; ++ip) v[ip] = sc * src[feat_of(8 * s_ + ip, hh)];
  } else {                     // (diag(gamma_l) W_l)^T: element [row][k] = gamma_l[k] W_l[k][row]
    const int l = 14 - pos, m = l - 1;
#pragma unroll
    for (int ip = 0; ip < 8; ++ip) {
      const int k = feat_of(8 * s_ + ip, hh);
      v[ip] = gm[l * C + k] * plain[((size_t)m * C + k) * C + row];
    }
  }
  const u32x4 d = {pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
  *reinterpret_cast<u32x4*>(out + (size_t)e * P_ELEM_BYTES + (size_t)pos * LBB + (size_t)u * 16) = d;
}

// x * (fp16 half of the parked pair), written in C: with -fno-slp-vectorize (build.py, this file) hipcc selects v_fma_mix_f32
// itself (with the SLP vectoriser on it packs the pair into 2 x v_cvt_f32_f16 + v_pk_fma_f32) AND pads the MFMA-result hazard
// of the accumulator operand (11 wait states behind an 8-pass MFMA); an inline-asm reader gets no padding.
__device__ __forceinline__ void mul_pair_c(float x0, float x1, unsigned c, float& v0, float& v1) {
  const f16x2 hc = __builtin_bit_cast(f16x2, c);
  v0 = __builtin_fmaf((float)hc[0], x0, 0.0f);
  v1 = __builtin_fmaf((float)hc[1], x1, 0.0f);
}

template <bool FAST>
__global__ void __launch_bounds__(64 * B3_WAVES) __attribute__((amdgpu_waves_per_eu(1, 1)))
sdf_mlp_full3p_kernel(const float* __restrict__ pts, const char* __restrict__ packed, const char* __restrict__ fimg,
                      const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ sdf_out,
                      float* __restrict__ grad_out, float* __restrict__ rgb_out, float* __restrict__ feat_out,
                      long long n_per_elem) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
#ifdef OI_B3_PROF
  const unsigned long long t_entry = __builtin_readcyclecounter();
  const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();  // constant 100 MHz
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, j = lane & 31;
  const int e = blockIdx.y;
  (void)packed;  // (everything per element comes from the builder's blob)
  // PERSISTENT workgroups (OI_B3P_PERSIST, launch_full3p): a workgroup walks tiles blockIdx.x, + gridDim.x, ... of its batch
  // element.  The table blob is fetched once, and the image ring keeps turning across tiles: the last three layers of a tile
  // request images 0..2 of the next one (ring position 15 k + p of tile k: `rb` = 15 k mod 4 shifts the slots).
  const int ntiles = (int)((n_per_elem + B3_TILE - 1) / B3_TILE);
  int tile = blockIdx.x;
  int rb = 0;

  LaneOff o;
  o.h16 = 16 * h;
  o.h64 = 64 * h;
  o.l16 = 16 * lane;
  o.l16hi = 0;
  asm volatile("" : "+v"(o.h16), "+v"(o.h64), "+v"(o.l16));

  auto point_of = [&](bool& valid) {
    int jj = lane & 31;
    asm volatile("" : "+v"(jj));
    const long long local = (long long)tile * B3_TILE + wave * WAVE_PTS + jj;
    valid = local < n_per_elem;
    return (long long)e * n_per_elem + (valid ? local : n_per_elem - 1);
  };

  __amdgpu_buffer_rsrc_t feat_rs;
  const int feat_off = j * C * 4 + 16 * h;
  auto set_feat_rs = [&]() {
    const long long base_pt = (long long)e * n_per_elem + (long long)tile * B3_TILE + wave * WAVE_PTS;
    const long long left = n_per_elem - ((long long)tile * B3_TILE + wave * WAVE_PTS);
    const int npts = feat_out == nullptr ? 0 : (left >= WAVE_PTS ? WAVE_PTS : (left > 0 ? (int)left : 0));
    feat_rs = __builtin_amdgcn_make_buffer_rsrc(feat_out + base_pt * C, 0, npts * C * 4, 0x00020000);
  };

  const __amdgpu_buffer_rsrc_t img_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(fimg + (size_t)e * P_ELEM_BYTES), 0, (int)P_ELEM_BYTES, 0x00020000);
  // image (pos mod 15) of this or the next tile -> ring slot (pos + rb) & 3
  auto prefetch = [&](int pos) {
    const int img = pos < NIMG_P ? pos : pos - NIMG_P;
    const int slot = (pos + rb) & (B3_NSLOT - 1);
#pragma unroll
    for (int q = 0; q < LBB / 4096 / B3_WAVES; ++q) {
      const int c = (wave * (LBB / 4096 / B3_WAVES) + q) * 4096;
      auto* dst = (__attribute__((address_space(3))) void*)(lds + P_WBUF + slot * LBB + c);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(img_rs, dst, 16, o.l16, img * LBB + c, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(img_rs, dst, 16, o.l16, img * LBB + c, 1024, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(img_rs, dst, 16, o.l16, img * LBB + c, 2048, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(img_rs, dst, 16, o.l16, img * LBB + c, 3072, 0);
    }
  };
  constexpr int DMA_PER_IMAGE = LBB / 1024 / B3_WAVES;
  auto lay = [&](int pos) {
    int b = o.l16 + P_WBUF + ((pos + rb) & (B3_NSLOT - 1)) * LBB;
    asm volatile("" : "+v"(b));
    return b;
  };
  auto film_base = [&](int l) { return o.h16 + P_FILM + l * P_FILM_ROW; };

  float px, py, pz;
  {
    bool valid;
    const long long pt = point_of(valid);
    px = pts[pt * 3 + 0], py = pts[pt * 3 + 1], pz = pts[pt * 3 + 2];
  }
  // the per-element table blob -> LDS [0, P_BLOB): 15 copies of 1 KiB, wave w takes chunks w, w + 4, ...
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = q * 4 + wave;
    if (c < P_BLOB / 1024)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(img_rs, (__attribute__((address_space(3))) void*)(lds + c * 1024), 16, o.l16,
                                               NIMG_P * LBB + c * 1024, 0, 0);
  }
  prefetch(0);
  prefetch(1);
  prefetch(2);
  ring_sync_b<3 * DMA_PER_IMAGE>();  // point + tables landed (the three images may still be in flight), visible to every wave
  int tiles_done = 0;

#ifdef OI_B3_PROF
  unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_readcyclecounter();
  const unsigned long long tstart = tprev;
  pacc[4] = tstart - t_entry;  // prologue
#endif
  f32x16 acc[4];
  Limb AH, BH, CH;
  BankB P0, P1, P2, P3, P4, P5, P6;
  f32x4 sg[4];  // w_sigma rows of layer 7's epilogue, requested two groups ahead
  f32x4 fv;
  float sdf_part = 0.f;

  auto reduce = [&](float phi) { return FAST ? phi : __builtin_amdgcn_fractf(phi); };
  auto ld = [&](int imm, int base) { return lds_f4(lds, imm, base); };
  // the B2 rows of output block t of the layer whose rows sit at lane base FB, in accumulator order
  auto initrows = [&](int fb, int t) {
    f32x16 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 b = ld(grp_f0(4 * t + i) * 4, fb);
#pragma unroll
      for (int k = 0; k < 4; ++k) a[4 * i + k] = b[k];
    }
    return a;
  };
#define INIT_P(FB) [&](int t_) { return initrows(FB, t_); }

  auto frag3_b = [&](float vx, float vy, float vz) {
    const float hx = (float)(__bf16)vx, hy = (float)(__bf16)vy, hz = (float)(__bf16)vz;
    const unsigned d0 = pk_bf16(hx, hy);
    const unsigned d1 = h == 0 ? pk_bf16(hz, vx - hx) : pk_bf16(hz, 0.f);
    const unsigned d2 = h == 0 ? pk_bf16(vy - hy, vz - hz) : 0u;
    return __builtin_bit_cast(bf16x8, u32x4{d0, d1, d2, 0u});
  };
  auto frag3_a = [&](int tab, int t) {
    const f32x4 w = lds_f4(lds, tab + t * 32 * 16, 16 * (lane & 31));
    const float hx = (float)(__bf16)w[0], hy = (float)(__bf16)w[1], hz = (float)(__bf16)w[2];
    const float ax = h == 0 ? hx : w[0] - hx, ay = h == 0 ? hy : w[1] - hy, az = h == 0 ? hz : w[2] - hz;
    const unsigned d0 = pk_bf16(ax, ay);
    const unsigned d1 = h == 0 ? pk_bf16(az, hx) : pk_bf16(az, 0.f);
    const unsigned d2 = h == 0 ? pk_bf16(hy, hz) : 0u;
    return __builtin_bit_cast(bf16x8, u32x4{d0, d1, d2, 0u});
  };
  auto simg = [&](int img, int s_) {
    return __builtin_bit_cast(bf16x8, lds_f4(lds, P_SIMG + img * SIMG_BYTES + s_ * 256, 128 * h + 16 * (lane & 7)));
  };
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto rows3 = [&](const f32x16& a, float& r0, float& r1, float& r2) {
    const float u0 = __shfl_xor(a[0], 32, 64), u1 = __shfl_xor(a[1], 32, 64);
    r0 = a[0] + a[3];
    r1 = a[1] + u0;
    r2 = a[2] + u1;
  };

  // forward epilogue pair (tb, rp): the accumulator is the phase; sin -> next limb set NH, cos -> BANK
#define OI_FWD_EPI_P(NH, BANK)                                                                             \
  [&](int tb, int rp) {                                                                                    \
    const int g = tb * 4 + (rp >> 1), k = 2 * (rp & 1);                                                    \
    const float r0 = reduce(acc[tb][2 * rp]), r1 = reduce(acc[tb][2 * rp + 1]);                           \
    NH[2 * tb + (rp >> 2)][rp & 3] = pk_bf16(__builtin_amdgcn_sinf(r0), __builtin_amdgcn_sinf(r1));        \
    BANK[g][k >> 1] = to_acc_u(pk_f16(__builtin_amdgcn_cosf(r0), __builtin_amdgcn_cosf(r1)));              \
  }
  // reverse epilogue pair of the transposed product of layer l: V'_{l-1} = g_l * cos(phi_{l-1}) -> limb set NH
#define OI_REV_EPI_P(BANK, NH)                                                                             \
  [&](int tb, int rp) {                                                                                    \
    const int g = tb * 4 + (rp >> 1), k = 2 * (rp & 1);                                                    \
    const unsigned c2 = from_acc_u(BANK[g][k >> 1]);                                                       \
    float v0, v1;                                                                                          \
    mul_pair_c(acc[tb][2 * rp], acc[tb][2 * rp + 1], c2, v0, v1);                                          \
    NH[2 * tb + (rp >> 2)][rp & 3] = pk_bf16(v0, v1);                                                      \
  }

  const int F0 = film_base(0), F1 = film_base(1), F2 = film_base(2), F3 = film_base(3), F4 = film_base(4),
            F5 = film_base(5), F6 = film_base(6), F7 = film_base(7), F8 = film_base(8);

  for (;;) {  // ---- one tile per trip
  set_feat_rs();
  sdf_part = 0.f;
  // the next tile's point is requested now and used a tile later (a clamped re-read of this one on the last trip)
  float nx, ny, nz;
  const int ntile = tile + (int)gridDim.x;
  {
    const int keep = tile;
    if (ntile < ntiles) tile = ntile;
    bool v_;
    const long long pn = point_of(v_);
    nx = pts[pn * 3 + 0], ny = pts[pn * 3 + 1], nz = pts[pn * 3 + 2];
    tile = keep;
  }
  // ================= forward, layers 0..7 =================
  {  // layer 0 (K = 3): one MFMA per output block on top of the B2 row, then the ordinary forward epilogue
    const bf16x8 bp = frag3_b(px, py, pz);
#pragma unroll
    for (int t = 0; t < 4; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag3_a(P_TABS + H_TAB0 * 4, t), bp, initrows(F0, t), 0, 0, 0);
    auto e0 = OI_FWD_EPI_P(AH, P0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int rp = 0; rp < 8; ++rp) {
        e0(t, rp);
        if (rp & 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  B3_T(0);
  acc[0] = initrows(F1, 0);
  ring_sync_b<2 * DMA_PER_IMAGE>();  // image 0 resident (1 and 2 may still be in flight)
  B3_T(2);
  prefetch(3);
  auto e1 = OI_FWD_EPI_P(BH, P1);
  stream_layer_b(lds, lay(0), AH, acc, NoTailB(), e1, NoPost(), INIT_P(F1));
  B3_T(1);
  acc[0] = initrows(F2, 0);
  ring_sync_b<2 * DMA_PER_IMAGE>();
  B3_T(2);
  prefetch(4);
  auto e2 = OI_FWD_EPI_P(AH, P2);
  stream_layer_b(lds, lay(1), BH, acc, e1, e2, NoPost(), INIT_P(F2));
  B3_T(1);
  acc[0] = initrows(F3, 0);
  ring_sync_b<2 * DMA_PER_IMAGE>();
  B3_T(2);
  prefetch(5);
  auto e3 = OI_FWD_EPI_P(BH, P3);
  stream_layer_b(lds, lay(2), AH, acc, e2, e3, NoPost(), INIT_P(F3));
  B3_T(1);
  acc[0] = initrows(F4, 0);
  ring_sync_b<2 * DMA_PER_IMAGE>();
  B3_T(2);
  prefetch(6);
  auto e4 = OI_FWD_EPI_P(AH, P4);
  stream_layer_b(lds, lay(3), BH, acc, e3, e4, NoPost(), INIT_P(F4));
  B3_T(1);
  acc[0] = initrows(F5, 0);
  ring_sync_b<2 * DMA_PER_IMAGE>();
  B3_T(2);
  prefetch(7);
  auto e5 = OI_FWD_EPI_P(BH, P5);
  stream_layer_b(lds, lay(4), AH, acc, e4, e5, NoPost(), INIT_P(F5));
  B3_T(1);
  acc[0] = initrows(F6, 0);
  ring_sync_b<2 * DMA_PER_IMAGE>();
  B3_T(2);
  prefetch(8);
  auto e6 = OI_FWD_EPI_P(AH, P6);
  stream_layer_b(lds, lay(5), BH, acc, e5, e6, NoPost(), INIT_P(F6));
  B3_T(1);
  acc[0] = initrows(F7, 0);
  auto req_sig = [&](int g_) { sg[g_ & 3] = ld(P_TABS + (H_SIG + grp_f0(g_)) * 4, o.h16); };
  req_sig(0);
  req_sig(1);
  ring_sync_b<2 * DMA_PER_IMAGE>();
  B3_T(2);
  prefetch(9);
  // layer 7: features a8 = sin(phi7) -> limb set CH (+ feat_out), sdf = a8 . wsig + bsig on the fly, and the reverse sweep's
  // first operand V'7 = wsig * cos(phi7) is formed in place (cos(phi7) is never parked; gamma_7 sits in the transposed image)
  auto e7 = [&](int tb, int rp) {
    const int g = tb * 4 + (rp >> 1), k = 2 * (rp & 1);
    const f32x4& ws = sg[g & 3];
    if (k == 0 && g + 2 < 16) req_sig(g + 2);
    float sn[2], v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float r = reduce(acc[tb][2 * rp + i]);
      sn[i] = __builtin_amdgcn_sinf(r);
      fv[k + i] = sn[i];
      sdf_part = fmaf(sn[i], ws[k + i], sdf_part);
      v[i] = ws[k + i] * __builtin_amdgcn_cosf(r);
    }
    CH[2 * tb + (rp >> 2)][rp & 3] = pk_bf16(sn[0], sn[1]);
    BH[2 * tb + (rp >> 2)][rp & 3] = pk_bf16(v[0], v[1]);
    if (k == 2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, fv), feat_rs, feat_off + grp_f0(g) * 4, 0, 0);
  };
  stream_layer_b(lds, lay(6), AH, acc, e6, e7, NoPost(), INIT_P(F7));
  B3_T(1);
  ring_sync_b<2 * DMA_PER_IMAGE>();
  B3_T(2);
  prefetch(10);

  // ================= reverse, layers 7..1 =================
  auto r7 = OI_REV_EPI_P(P6, AH);   // V'6 = g7 * cos(phi6)
  stream_layer_b(lds, lay(7), BH, acc, e7, r7);
  B3_T(1);
  {
    sdf_part += __shfl_xor(sdf_part, 32, 64);  // complete since e7's last pair (inside the layer above)
    const float sdf_v = sdf_part + *reinterpret_cast<const float*>(lds + P_TABS + (H_SIG + C) * 4);
    bool valid;
    const long long pt = point_of(valid);
    if (valid && h == 0) sdf_out[pt] = sdf_v;
  }
  ring_sync_b<2 * DMA_PER_IMAGE>();
  B3_T(2);
  prefetch(11);
  auto r6 = OI_REV_EPI_P(P5, BH);
  stream_layer_b(lds, lay(8), AH, acc, r7, r6);
  B3_T(1);
  ring_sync_b<2 * DMA_PER_IMAGE>();
  B3_T(2);
  prefetch(12);
  auto r5 = OI_REV_EPI_P(P4, AH);
  stream_layer_b(lds, lay(9), BH, acc, r6, r5);
  B3_T(1);
  ring_sync_b<2 * DMA_PER_IMAGE>();
  B3_T(2);
  prefetch(13);
  auto r4 = OI_REV_EPI_P(P3, BH);
  stream_layer_b(lds, lay(10), AH, acc, r5, r4);
  B3_T(1);
  ring_sync_b<2 * DMA_PER_IMAGE>();
  B3_T(2);
  prefetch(14);
  auto r3 = OI_REV_EPI_P(P2, AH);
  stream_layer_b(lds, lay(11), BH, acc, r4, r3);
  B3_T(1);
  ring_sync_b<2 * DMA_PER_IMAGE>();  // image 12 resident; 13 and 14 in flight
  B3_T(2);
  prefetch(15);                      // the next tile's image 0 (requested again, unused, on a workgroup's last trip)
  auto r2 = OI_REV_EPI_P(P1, BH);   // V'1 = g2 * cos(phi1)
  stream_layer_b(lds, lay(12), AH, acc, r3, r2);
  B3_T(1);
  ring_sync_b<2 * DMA_PER_IMAGE>();  // image 13 resident
  B3_T(2);
  prefetch(16);
  // transposed layer 1: V'0 = g1 * cos(phi0) as limbs; then d sdf/dx = (diag(gamma_0) W0)^T V'0 (K = 128 -> 3 rows) as eight
  // MFMAs against the small image
  auto r1 = OI_REV_EPI_P(P0, AH);
  stream_layer_b(lds, lay(13), BH, acc, r2, r1);
  B3_T(1);
  run_tail_b(r1);
  B3_T(3);
  float gx, gy, gz;
  {
    f32x16 a0 = zero16;
#pragma unroll
    for (int s_ = 0; s_ < 8; ++s_) {
      const u32x4 ub = {AH[s_][0], AH[s_][1], AH[s_][2], AH[s_][3]};
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(simg(0, s_), __builtin_bit_cast(bf16x8, ub), a0, 0, 0, 0);
    }
    rows3(a0, gx, gy, gz);
    gx = __shfl(gx, lane & 31, 64);  // both halves of the point need the gradient (the albedo head's B fragment)
    gy = __shfl(gy, lane & 31, 64);
    gz = __shfl(gz, lane & 31, 64);
  }
  bool valid;
  const long long pt = point_of(valid);
  if (valid && h == 0) {
    grad_out[pt * 3 + 0] = gx;
    grad_out[pt * 3 + 1] = gy;
    grad_out[pt * 3 + 2] = gz;
  }

  {
    // ---- albedo head: sigmoid(Wrgb sin(gv * (Wv [feat, grad] + bv) + bv') + brgb)   (fields.py:89-101)
    acc[0] = initrows(F8, 0);
    ring_sync_b<2 * DMA_PER_IMAGE>();  // image 14 resident (the next tile's images 0 and 1 and this tile's gradient stores are younger)
  B3_T(2);
    prefetch(17);
    float r0 = 0.f, r1c = 0.f, r2c = 0.f;
    // the head's three gradient columns (rows of TABV x gamma_v / 2pi) are a ninth MFMA of every output block; its activations
    // sin(phi_v) go on as bf16 limbs (AH: free) and rgb = Wrgb sin(phi_v) is eight MFMAs against the second small image
    const bf16x8 bg3 = frag3_b(gx, gy, gz);
    auto post = [&](int t, f32x16& a) { a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag3_a(P_TABS + H_TABV * 4, t), bg3, a, 0, 0, 0); };
    auto ec = [&](int tb, int rp) {
      const float s0 = __builtin_amdgcn_sinf(reduce(acc[tb][2 * rp]));
      const float s1 = __builtin_amdgcn_sinf(reduce(acc[tb][2 * rp + 1]));
      AH[2 * tb + (rp >> 2)][rp & 3] = pk_bf16(s0, s1);
    };
    stream_layer_b(lds, lay(14), CH, acc, NoTailB(), ec, post, INIT_P(F8));
  B3_T(1);
    run_tail_b(ec);
  B3_T(3);
    {
      f32x16 a0 = zero16;
#pragma unroll
      for (int s_ = 0; s_ < 8; ++s_) {
        const u32x4 ub = {AH[s_][0], AH[s_][1], AH[s_][2], AH[s_][3]};
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(simg(1, s_), __builtin_bit_cast(bf16x8, ub), a0, 0, 0, 0);
      }
      rows3(a0, r0, r1c, r2c);
    }
    if (valid && h == 0 && rgb_out != nullptr) {
      const float* brgb = reinterpret_cast<const float*>(lds + P_TABS + (H_RGB + 3 * C) * 4);
      rgb_out[pt * 3 + 0] = oi::sigmoidf_(r0 + brgb[0]);
      rgb_out[pt * 3 + 1] = oi::sigmoidf_(r1c + brgb[1]);
      rgb_out[pt * 3 + 2] = oi::sigmoidf_(r2c + brgb[2]);
    }
  }
  ++tiles_done;
  if (ntile >= ntiles) break;
  tile = ntile;
  rb = (rb + NIMG_P) & (B3_NSLOT - 1);
  px = nx, py = ny, pz = nz;
  }  // tiles
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the three images requested for a tile that does not exist
#ifdef OI_B3_PROF
  B3_T(5);
  if (lane == 0) {
    unsigned long long* pr = oi_prof3b[(blockIdx.x * 4 + wave) & 1023];
    for (int i = 0; i < 6; ++i) atomicAdd(&pr[i], pacc[i]);
    atomicAdd(&pr[6], __builtin_readcyclecounter() - tstart);
    atomicAdd(&pr[7], (unsigned long long)tiles_done);
    atomicAdd(&pr[8], __builtin_readcyclecounter() - t_entry);  // with [9]: the shader clock in the kernel
    atomicAdd(&pr[9], __builtin_amdgcn_s_memrealtime() - rt_entry);
  }
#endif
#undef OI_FWD_EPI_P
#undef OI_REV_EPI_P
#undef INIT_P
}

size_t full3p_scratch(int B) { return (size_t)B * P_ELEM_BYTES; }

template <bool FAST>
int launch_full3p(const float* pts, const char* pk, const float* gamma, const float* beta, float* sdf, float* grad,
                  float* rgb, float* feat, void* scratch, int B, long long n, hipStream_t st) {
  char* fimg = reinterpret_cast<char*>(scratch);
  hipLaunchKernelGGL(film_images_b_kernel, dim3(C * C / 8 / 256, NIMG_P + 1, B), dim3(256), 0, st, pk, gamma, beta, fimg);
  // persistent workgroups: as many as the device runs at once (one per CU: the LDS), each walks its share of the tiles;
  // OI_B3P_PERSIST=0 (environment): one workgroup per tile, the same kernel (A/B switch)
  static const int per_dev = [] {
    const char* v = getenv("OI_B3P_PERSIST");
    if (v && v[0] == '0') return 0;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus > 0 ? cus : 256;
  }();
  const int tiles = oi::cdiv(n, B3_TILE);
  const int gx = per_dev > 0 ? std::min(tiles, std::max(1, per_dev / B)) : tiles;
  dim3 grid(gx, B), block(64 * B3_WAVES);
  auto k = sdf_mlp_full3p_kernel<FAST>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
  hipLaunchKernelGGL(k, grid, block, P_LDS, st, pts, pk, fimg, gamma, beta, sdf, grad, rgb, feat, n);
  return oi::check_launch("oi_sdf_mlp_fwd(full3p)");
}

template <bool FAST>
int launch_full3b(const float* pts, const char* pk, const float* gamma, const float* beta, float* sdf, float* grad,
                  float* rgb, float* feat, int B, long long n, hipStream_t st) {
  dim3 grid(oi::cdiv(n, B3_TILE), B), block(64 * B3_WAVES);
  auto k = sdf_mlp_full3b_kernel<FAST>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, B3_LDS);
  hipLaunchKernelGGL(k, grid, block, B3_LDS, st, pts, pk, gamma, beta, sdf, grad, rgb, feat, n);
  return oi::check_launch("oi_sdf_mlp_fwd(full3b)");
}

}  // namespace

namespace oimlp {

size_t full3_bf16_scratch_bytes(int B) { return full3p_scratch(B); }

// OI_BF16_PRESCALE=0 (environment, read once) keeps the round-4 kernel with shared images + FiLM rows: the same-box A/B switch
int launch_full3_bf16(const float* pts, const void* packed, const float* gamma, const float* beta, float* sdf, float* grad,
                      float* rgb, float* feat, void* scratch, int B, long long n, int fast_trig, hipStream_t st) {
  const char* pk = reinterpret_cast<const char*>(packed);
  static const bool prescale = [] { const char* v = getenv("OI_BF16_PRESCALE"); return !(v && v[0] == '0'); }();
  if (prescale)
    return fast_trig ? launch_full3p<true>(pts, pk, gamma, beta, sdf, grad, rgb, feat, scratch, B, n, st)
                     : launch_full3p<false>(pts, pk, gamma, beta, sdf, grad, rgb, feat, scratch, B, n, st);
  return fast_trig ? launch_full3b<true>(pts, pk, gamma, beta, sdf, grad, rgb, feat, B, n, st)
                   : launch_full3b<false>(pts, pk, gamma, beta, sdf, grad, rgb, feat, B, n, st);
}

}  // namespace oimlp

#ifdef OI_B3_PROF
extern "C" int oi_prof3b_read(unsigned long long* out, int reset) {
  (void)hipDeviceSynchronize();
  static unsigned long long host[1024][16];
  (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(oi_prof3b), sizeof(host));
  for (int i = 0; i < 16; ++i) {
    out[i] = 0;
    for (int r = 0; r < 1024; ++r) out[i] += host[r][i];
  }
  if (reset) {
    static unsigned long long z[1024][16];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(oi_prof3b), z, sizeof(z));
  }
  return 0;
}
#endif
