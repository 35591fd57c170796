// Backward of the FiLM-SIREN SDF / albedo MLP (csrc/mlp.hip) for gfx950 -- including the
// "double backward": the forward outputs d sdf/dx, so the parameter gradients of a loss that touches
// the normals (Phong shading, eikonal term, colour-head input) contain second-order terms.
//
// Replaces what the reference obtains from autograd with create_graph=True through
// ShapeNetwork.forward + ShapeNetwork.gradient + ColorNetwork.forward
// (src/models/fields.py:49-77, 89-101, 104-122; trainer call gan_pose_trainer.py:141).
//
// Per point, with a_0 = x, u_l = W_l a_l + b_l, phi_l = gamma_l u_l + beta_l, a_{l+1} = sin phi_l,
// c_l = gamma_l cos phi_l, g_8 = w_sigma, v_l = g_{l+1} * c_l, g_l = W_l^T v_l (grad = g_0):
//   colour             backward of the albedo head (needs a_8: saved by the forward) -> gbar_0 = dL/dgrad   [2 GEMMs]
//   up sweep   (l up)   TWO products per staged image:  vbar_l = W_l gbar_l  and  u_l = W_l a_l (recompute, flash-style);
//                       gbar_{l+1} = vbar_l * c_l;  parks phi_l and vbar_l                                    [14 GEMMs]
//   down sweep (l down) cbar_l = vbar_l * g_{l+1};  v_l = g_{l+1} * c_l;
//                       phibar_l = abar_{l+1} cos phi_l - cbar_l gamma_l sin phi_l;  ubar_l = phibar_l gamma_l;
//                       TWO products per staged image:  g_l = W_l^T v_l  and  abar_l = W_l^T ubar_l;
//                       parks v_l / gamma_l, phibar_l for the weight-gradient GEMM (round 4, see below)         [14 GEMMs]
// Round 3: the FiLM / bias gradients of the MFMA layers are NOT summed in the sweep any more.  phi_l = gamma_l (W_l a + b_l)
// + beta_l -- and with it every first- and second-order term -- depends on gamma_l, W_l, b_l only through gamma_l W_l and
// gamma_l b_l, hence per batch element
//       gamma_l[f] d gamma_l[f] = sum_i W_l[f][i] dW_l[f][i] + b_l[f] db_l[f],     gamma_l[f] d beta_l[f] = db_l[f],
// and db_l = sum_p ubar_l is a by-product of the weight-gradient GEMM (which reads ubar_l anyway).  That removes 72 of the
// 112 cross-lane point sums of a tile and the recomputation of u_l (one v_rcp + 4 VALU per element); the parked phase is
// the REDUCED phase in revolutions (FiLM rows staged pre-multiplied by 1/2pi, as in the forward kernel), so the down sweep
// and the GEMM feed it to v_sin / v_cos without a range reduction.
// Round 4: NO division by gamma.  Both weight-gradient operands carry gamma_l[f] as a factor of row f (v_l = gamma_l g_{l+1}
// cos phi_l, ubar_l = gamma_l phibar_l), so the sweep parks them WITHOUT it (v_l / gamma_l = g_{l+1} cos phi_l and phibar_l),
// the GEMM accumulates D_l[f][i] = sum_p (v_l/gamma_l gbar_l^T + phibar_l a_l^T)[f][i] and its epilogue writes
//       dW_l[f][i] = gamma_l[f] D_l[f][i],   db_l[f] = gamma_l[f] sum_p phibar_l[f],
//       d gamma_l[f] = sum_i W_l[f][i] D_l[f][i] + b_l[f] sum_p phibar_l[f],   d beta_l[f] = sum_p phibar_l[f]
// -- exact for gamma = 0 and free of the cancellation a small |gamma| caused (the reference, autograd through
// fields.py:104-122, has no restriction on gamma either; tests/test_gpu_backward.py::test_mlp_backward_gamma_through_zero).
// (Round 1 ran four sweeps -- phi up, g down, gbar up, abar down -- and parked 46 slots per point; forming g again in the
// last sweep instead of parking it, and gbar together with the recomputed activations, leaves 32 slots, 16 of which
// are read back here: 23.5 -> 16 KB written and 22 -> 9 KB read per point.)
// Both sweeps keep the point on the MFMA column (activations never leave registers inside a sweep);
// The weight gradients  dW_l = sum_p ( v_l gbar_l^T + ubar_l a_l^T )  are a second, split-K GEMM kernel over the
// parked operands (K = points); gbar_l = vbar_{l-1} c_{l-1} and a_l = sin phi_{l-1} are re-formed there from the SAME
// parked phi_{l-1}.
#include <algorithm>
#include <type_traits>

#include "mlp_common.h"

namespace {

using namespace oimlp;

// scratch slots of one wave tile (16 KiB each)
constexpr int S_PHI = 0;    // 8: phi_l, l = 1..7.  Layer 0 (three input columns, formed on the VALU) is NOT parked: the down
                            //    sweep and the weight-gradient GEMM form phi_0 / vbar_0 again from the point and dL/dgrad,
                            //    which sit in the first KiB of this slot: [32 points][x y z 0 | Gx Gy Gz 0]
constexpr int S_VB = 8;     // 8: gamma_l vbar_l, vbar_l = W_l gbar_l, l = 1..7   (down sweep: gamma_l cbar_l = . g_{l+1}; wgrad: gbar_{l+1} = . cos phi_l)
constexpr int S_V = 16;     // 7: v_l / gamma_l = g_{l+1} cos phi_l, l = 1..7   (wgrad operand)
constexpr int S_U = 23;     // 7: phibar_l = ubar_l / gamma_l,       l = 1..7   (wgrad operand)
constexpr int S_UV = 30;    // 1: phibar_v (colour head: gradient w.r.t. the phase, without gamma_v)
constexpr int S_AC = 31;    // 1: abar_8 contribution of the colour head
constexpr int NSLOT_BWD = 32;

// small-gradient buffer (floats)
constexpr int DS_W0 = 0;       // [128][3]
constexpr int DS_B = 384;      // [9][128]  b0..b7, bv
constexpr int DS_WSIG = 1536;  // [128]
constexpr int DS_BSIG = 1664;  // [1] (+3 pad)
constexpr int DS_WVX = 1668;   // [128][3]  Wv[:, 128:131]
constexpr int DS_WRGB = 2052;  // [3][128]
constexpr int DS_BRGB = 2436;  // [3] (+1 pad)
constexpr int DS_TOTAL = 2440;

// LDS of the sweep kernel: FiLM rows (slot 0) | small tables | image slot 0 | image slot 1 | reduction rows | FiLM slot 1.
// One workgroup (4 waves, one per SIMD, 512 registers each) per CU: the next layer's image and FiLM rows are requested
// while the current layer computes, so a layer boundary costs a barrier and not an LDS-DMA round trip + the drain of
// every scratch store in flight.
constexpr int L_RACC = L_WBUF + 2 * 65536;  // per-workgroup reduction scratch: [RACC_ROWS][128] floats
// Waves per workgroup of the sweep (one workgroup per CU either way: the two image slots fill the LDS).  4 = one wave per
// SIMD with 512 registers; 8 = two waves per SIMD with 256 registers each (the co-resident wave fills the other's LDS /
// MFMA-result / memory stalls and the VALU issues 1.45 x faster with two waves to pick from: tools/dbg/valu_issue.hip).
#ifndef OI_BWD_NW
#define OI_BWD_NW 8
#endif
constexpr int BW_NW = OI_BWD_NW, BW_THREADS = 64 * BW_NW, BW_TILE = WAVE_PTS * BW_NW;
// groups (of 16) of parked phi / vbar fragments the down sweep requests ahead of their use
// ... and how many of those stay in flight ACROSS the two products (the accumulators are dead during the epilogue: the
// ring can be deep there even with 256 registers, but only CARRY groups fit next to three point vectors)
#ifndef OI_BWD_PF
#define OI_BWD_PF (OI_BWD_NW == 4 ? 16 : 8)
#endif
#ifndef OI_BWD_CARRY
#define OI_BWD_CARRY (OI_BWD_NW == 4 ? 16 : 1)
#endif
#ifndef OI_BWD_PF_F32  // the fp32-MFMA product keeps more fragments live
#define OI_BWD_PF_F32 (OI_BWD_NW == 4 ? 16 : 4)
#endif
// the colour head's contribution to abar_8 waits for the down sweep in registers (1) or in scratch slot S_AC (0)
#ifndef OI_BWD_AC_REGS
#define OI_BWD_AC_REGS (OI_BWD_NW == 4)
#endif
// Point sums of the colour head, of w_sigma and of layer 0 in rows of their own (1): every row is flushed ONCE, behind the last
// barrier of the kernel -- the colour head's flush (two barriers and 900 atomics in the tile's first microseconds), the barrier
// of the turn and layer 0's re-zeroing go away.  0: one set of 8 rows, flushed and re-zeroed where each user ends (round 5).
// Persistent workgroups (1): one per CU (/ B) walks its share of the element's tiles -- tables, layer 0's FiLM rows and the zeroed
// reduction rows are set up once, the rows are flushed once per workgroup (needs OI_BWD_LATE_FLUSH).  0 (default): one workgroup
// per tile.  Measured, round 6: 2.78 / 2.83 ms persistent against 2.77 / 2.77 (same box, alternating) -- nothing sits between two
// tiles of a CU that a loop would remove (the per-CU timelines of tools/dbg/phase_prof_bwd.py: a CU's tiles follow each other
// within the profiling code's own cost), and tile lifetimes spread 0.55x .. 1.45x around their mean: the tiles wait on the
// memory system they share, not on their own start.
#ifndef OI_BWD_PERSIST
#define OI_BWD_PERSIST 0
#endif
#ifndef OI_BWD_LATE_FLUSH
#define OI_BWD_LATE_FLUSH 1
#endif
constexpr int RACC_ROWS = OI_BWD_LATE_FLUSH ? 16 : 8;
constexpr int RR_COL = OI_BWD_LATE_FLUSH ? 8 : 0;    // the colour head's rows 2..7 -> RR_COL + 2 .. RR_COL + 7
constexpr int RR_WSIG = OI_BWD_LATE_FLUSH ? 0 : 3;   // (layer 0: rows 1, 3, 4, 5)
constexpr int L_FILM2 = L_RACC + RACC_ROWS * C * 4;
constexpr int L_TOTAL_BWD = L_FILM2 + 1536;

// Cache policy per slot family (aux operand of the buffer instructions; measured in oi_common.h's table):
//   LOCAL  slots are re-read later in THIS kernel (phi, g, cbar, the colour-head pair),
//   WGRAD  slots are only written here and consumed by the weight-gradient GEMM (v, gbar, ubar).
#ifndef OI_BWD_ST_LOCAL
#define OI_BWD_ST_LOCAL OI_BWD_NT_ST
#endif
#ifndef OI_BWD_ST_WGRAD
#define OI_BWD_ST_WGRAD OI_BWD_NT_ST
#endif
#ifndef OI_BWD_LD_EARLY
#define OI_BWD_LD_EARLY 0  // phi read in phases B and C is read again in phase D: 8.59 vs 8.66 ms per training render
#endif
#ifndef OI_BWD_LD_LAST
#define OI_BWD_LD_LAST OI_BWD_NT_LD
#endif
// Round 4, measured and NOT adopted as the default (-DOI_BWD_PACK24=1 builds it): the parked vectors crossing HBM as 24-bit
// values (F16X3 mode, accurate trig) -- 12 instead of 16 KiB per slot, a quarter of the 19 GB a backward moves.
//   phase slots   the reduced phase r in [0, 1) as 24-bit FIXED point (2^-24 absolute: finer than fp32's own spacing near 1)
//   every other   fp32 with the mantissa rounded to 15 stored bits (2^-16 relative)
// Four values = three dwords, moved by v_perm_b32 (3 to pack + 4 rounding adds, 4 to unpack).  Same box, 30 training
// iterations: 9.68 -> 9.31 ms per iteration (render forward + backward 5.88 -> 5.57 ms).  The price: the worst parameter-
// gradient error of the seven f16x3 gradient tests goes from 0.5-1.2e-5 to 1.4-1.8e-5 against a bar of 2e-5 (tools/dbg/
// run_q24.sh: every slot family contributes -- gamma vbar 1.7e-5, v / phibar 1.5e-5, the colour pair 1.2e-5 on its own); a
// format that keeps the bar's margin (24-bit fixed point relative to a per-lane maximum) costs 11 + 8 instead of 7 + 4
// instructions per four values and needs the maxima before the stores: estimated to return less than half of the 0.37 ms.
#ifndef OI_BWD_PACK24
#define OI_BWD_PACK24 0
#endif
#ifndef OI_WGRAD_BF16
#define OI_WGRAD_BF16 1   // bf16 operand mode: the two-tiles-in-flight GEMM with bf16 operands (0: the generic fp32-MFMA GEMM)
#endif
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ u32x3 pack24f(f32x4 v) {
  // (element copies first: __builtin_bit_cast applied to a vector-element lvalue read element 0 four times -- hipcc 7.2)
  const float x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3];
  const unsigned b0 = __builtin_bit_cast(unsigned, x0) + 0x80u, b1 = __builtin_bit_cast(unsigned, x1) + 0x80u;
  const unsigned b2 = __builtin_bit_cast(unsigned, x2) + 0x80u, b3 = __builtin_bit_cast(unsigned, x3) + 0x80u;
  // bytes 1..3 of every value: {b0.1 b0.2 b0.3 b1.1} {b1.2 b1.3 b2.1 b2.2} {b2.3 b3.1 b3.2 b3.3}
  return u32x3{__builtin_amdgcn_perm(b1, b0, 0x05030201u), __builtin_amdgcn_perm(b2, b1, 0x06050302u),
               __builtin_amdgcn_perm(b3, b2, 0x07060503u)};
}
__device__ __forceinline__ f32x4 unpack24f(u32x3 d) {
  const unsigned b0 = __builtin_amdgcn_perm(0u, d[0], 0x0201000cu), b1 = __builtin_amdgcn_perm(d[1], d[0], 0x0504030cu);
  const unsigned b2 = __builtin_amdgcn_perm(d[2], d[1], 0x0403020cu), b3 = __builtin_amdgcn_perm(0u, d[2], 0x0302010cu);
  return f32x4{__builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1), __builtin_bit_cast(float, b2),
               __builtin_bit_cast(float, b3)};
}
__device__ __forceinline__ u32x3 pack24q(f32x4 r) {  // r in [0, 1)
  const unsigned q0 = (unsigned)(r[0] * 16777216.f), q1 = (unsigned)(r[1] * 16777216.f);
  const unsigned q2 = (unsigned)(r[2] * 16777216.f), q3 = (unsigned)(r[3] * 16777216.f);
  return u32x3{__builtin_amdgcn_perm(q1, q0, 0x04020100u), __builtin_amdgcn_perm(q2, q1, 0x05040201u),
               __builtin_amdgcn_perm(q3, q2, 0x06050402u)};
}
__device__ __forceinline__ f32x4 unpack24q(u32x3 d) {
  const unsigned q0 = __builtin_amdgcn_perm(0u, d[0], 0x0c020100u), q1 = __builtin_amdgcn_perm(d[1], d[0], 0x0c050403u);
  const unsigned q2 = __builtin_amdgcn_perm(d[2], d[1], 0x0c040302u), q3 = __builtin_amdgcn_perm(0u, d[2], 0x0c030201u);
  constexpr float S = 1.0f / 16777216.f;
  return f32x4{(float)q0 * S, (float)q1 * S, (float)q2 * S, (float)q3 * S};
}

// Round 5, the bf16 operand mode only (OI_BWD_PACK16, default on): 16-bit slots -- values as bf16 (v_cvt_pk_bf16_f32: fp32's
// range, 8 bits, what the mode's MFMA operands carry anyway), phases REDUCED to [0, 1) and stored as unorm16
// (v_cvt_pknorm_u16_f32: 1.5e-5 revolutions).  Half the bytes of a backward in the mode BASELINE's configs[1] names; four values
// = two dwords, 2 instructions to pack, 4 to unpack.
#ifndef OI_BWD_PACK16
#define OI_BWD_PACK16 1
#endif
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2 pack16f(f32x4 v) {
  const bf16x2 a = {(__bf16)v[0], (__bf16)v[1]}, b = {(__bf16)v[2], (__bf16)v[3]};
  return u32x2{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
}
__device__ __forceinline__ f32x4 unpack16f(u32x2 d) {
  return f32x4{__builtin_bit_cast(float, d[0] << 16), __builtin_bit_cast(float, d[0] & 0xffff0000u),
               __builtin_bit_cast(float, d[1] << 16), __builtin_bit_cast(float, d[1] & 0xffff0000u)};
}
__device__ __forceinline__ u32x2 pack16q(f32x4 r) {  // any phase in revolutions: reduced here
  const float r0 = __builtin_amdgcn_fractf(r[0]), r1 = __builtin_amdgcn_fractf(r[1]);
  const float r2 = __builtin_amdgcn_fractf(r[2]), r3 = __builtin_amdgcn_fractf(r[3]);
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  const u16x2 a = __builtin_amdgcn_cvt_pknorm_u16(r0, r1), b = __builtin_amdgcn_cvt_pknorm_u16(r2, r3);
  return u32x2{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
}
__device__ __forceinline__ f32x4 unpack16q(u32x2 d) {
  constexpr float S = 1.0f / 65535.f;
  return f32x4{(float)(d[0] & 0xffffu) * S, (float)(d[0] >> 16) * S, (float)(d[1] & 0xffffu) * S, (float)(d[1] >> 16) * S};
}

// Round 5 (second half), f16x3 mode: the two slot families only the weight-gradient GEMM reads (S_V, S_U: 14 of the 29 slots a
// sweep writes, 14 of the 28 the GEMM reads) as 24-bit FIXED point relative to a per-lane power-of-two scale (OI_BWD_XQ24).
//   y = x * s + 1.5 with |x| s < 1/4 puts a value into ONE binade: its 23 mantissa bits are fixed point (2^-23 of 4 max|x|:
//   2^-21 of the lane's largest value, the resolution of the GEMM's own fp16 hi + lo split), rounded to nearest by the FMA.
//   The low three bytes of y are kept; four values = three dwords [a0 a1 a2 d0] [b0 b1 b2 d1] [c0 c1 c2 d2]: 4 FMA + 3 v_perm to
//   pack (the round-4 float24 format: 4 adds + 3 v_perm), 3 v_and_or + 3 to unpack, and the de-scaling rides on the multiply by
//   the launch-wide scale the GEMM applies anyway (x * scx = y * k - 1.5 k, k = scx / s).  The lane's 1 / s sits behind the 16
//   groups of the slot (byte 12,288 + 4 lane).  The lane maximum exists before the first store because the down sweep stores
//   v / phibar AFTER its epilogue (OI_BWD_STORES_LAST): one v_max3 per two values, then reused for the launch-wide maxima.
#ifndef OI_BWD_XQ24
#define OI_BWD_XQ24 1
#endif
constexpr int XQ_SCALE_OFF = 16 * 768;  // byte offset of the per-lane inverse scales inside a slot
__device__ __forceinline__ u32x3 pack_q24(f32x4 v, float s) {
  const float y0 = fmaf(v[0], s, 1.5f), y1 = fmaf(v[1], s, 1.5f), y2 = fmaf(v[2], s, 1.5f), y3 = fmaf(v[3], s, 1.5f);
  const unsigned a = __builtin_bit_cast(unsigned, y0), b = __builtin_bit_cast(unsigned, y1);
  const unsigned c = __builtin_bit_cast(unsigned, y2), d = __builtin_bit_cast(unsigned, y3);
  return u32x3{__builtin_amdgcn_perm(d, a, 0x04020100u), __builtin_amdgcn_perm(d, b, 0x05020100u),
               __builtin_amdgcn_perm(d, c, 0x06020100u)};
}
// -> the four y in [1.25, 1.75] (the caller applies x = (y - 1.5) / s together with its own scale).  The exponent's lowest bit is
// OR-ed in with the upper ones (0x3f800000, not 0x3f000000): a PHASE within 2^-24 of 1 rounds to y = 2.0, whose low 24 bits are
// zero -- it then decodes to 1.0 = phase 0, the same angle (with 0x3f000000 it decoded to 0.5: half a revolution off, ~30 values
// per 524,288-point backward, 1e-2 in the weight gradients they hit)
__device__ __forceinline__ f32x4 unpack_q24(u32x3 q) {
  const unsigned a = (q[0] & 0x00ffffffu) | 0x3f800000u, b = (q[1] & 0x00ffffffu) | 0x3f800000u;
  const unsigned c = (q[2] & 0x00ffffffu) | 0x3f800000u;
  const unsigned t = __builtin_amdgcn_perm(q[1], q[0], 0x0c0c0703u);            // [q0.3 q1.3 0 0]
  const unsigned d = __builtin_amdgcn_perm(q[2], t, 0x0c070100u) | 0x3f800000u;  // [q0.3 q1.3 q2.3 3f]
  return f32x4{__builtin_bit_cast(float, a), __builtin_bit_cast(float, b), __builtin_bit_cast(float, c),
               __builtin_bit_cast(float, d)};
}
// OI_BWD_PHQ24: the phase slots the same way without a scale -- the parked phase is already reduced to [0, 1) revolutions, so
// y = r + 1 is its 23-bit fixed-point form (2^-23 revolutions: 7.5e-7 rad), and v_sin / v_cos take y AS IT IS (the period is 1).
#ifndef OI_BWD_PHQ24
#define OI_BWD_PHQ24 1
#endif
__device__ __forceinline__ u32x3 pack_q24_phase(f32x4 r) {
  const unsigned a = __builtin_bit_cast(unsigned, r[0] + 1.0f), b = __builtin_bit_cast(unsigned, r[1] + 1.0f);
  const unsigned c = __builtin_bit_cast(unsigned, r[2] + 1.0f), d = __builtin_bit_cast(unsigned, r[3] + 1.0f);
  return u32x3{__builtin_amdgcn_perm(d, a, 0x04020100u), __builtin_amdgcn_perm(d, b, 0x05020100u),
               __builtin_amdgcn_perm(d, c, 0x06020100u)};
}
// OI_BWD_VBQ24: the gamma vbar slots (S_VB: written by the up sweep, read by the down sweep and by the GEMM) in the X-slot format;
// the up sweep finishes the vector in its point registers, takes the lane maximum, then packs and stores
#ifndef OI_BWD_VBQ24
#define OI_BWD_VBQ24 1
#endif
// s with max|x| * s < 1/4 (mx < 2^(E - 126) for the biased exponent E of mx) and its inverse, both exact powers of two
__device__ __forceinline__ void q24_scale(float mx, float& s, float& inv_s) {
  int E = (__builtin_bit_cast(int, mx) >> 23) & 0xff;
  E = E > 250 ? 250 : E;
  s = __builtin_bit_cast(float, (251 - E) << 23);
  inv_s = __builtin_bit_cast(float, (E + 3) << 23);
}

// oi_selftest_q24: thread t owns values [64 t, 64 t + 64) -- one lane's vector
__global__ void selftest_q24_kernel(const float* __restrict__ x, float* __restrict__ y, long long nvec, int mode) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nvec) return;
  const float* xi = x + t * 64;
  float* yo = y + t * 64;
  if (mode == 1) {
    for (int g = 0; g < 16; ++g) {
      const f32x4 d = unpack_q24(pack_q24_phase(f32x4{xi[4 * g], xi[4 * g + 1], xi[4 * g + 2], xi[4 * g + 3]}));
      for (int k = 0; k < 4; ++k) yo[4 * g + k] = d[k];
    }
    return;
  }
  float mx = 0.f;
  for (int k = 0; k < 64; ++k) mx = fmaxf(mx, fabsf(xi[k]));
  float sq, inv_sq;
  q24_scale(mx, sq, inv_sq);
  const float c = -1.5f * inv_sq;
  for (int g = 0; g < 16; ++g) {
    const f32x4 d = unpack_q24(pack_q24(f32x4{xi[4 * g], xi[4 * g + 1], xi[4 * g + 2], xi[4 * g + 3]}, sq));
    for (int k = 0; k < 4; ++k) yo[4 * g + k] = fmaf(d[k], inv_sq, c);
  }
}

// PACK: 0 = fp32 slots (16 bytes per lane and group), 1 = 24-bit (12), 2 = 16-bit (8)
template <int PACK, bool PHQ = false>
struct WaveScratchT {
  __amdgpu_buffer_rsrc_t rs;
  int l12;  // 12 * lane (PACK 1) / 8 * lane (PACK 2)
  static constexpr int GROUP = PACK == 0 ? 1024 : (PACK == 1 ? 768 : 512);   // bytes of one group of a slot
  // (stores: the wave-uniform offset is folded into voffset, soffset = 0: the >64-bit store hazard of oi::buffer_store_b128)
  template <int AUX = OI_BWD_ST_LOCAL>
  __device__ __forceinline__ void store(int slot, int g, int l16, f32x4 v) const {
#ifdef OI_BWD_Q24_SET  // precision experiment (32-bit slots): round the families in the mask to the 24-bit float format
    {
      const int fam = slot >= S_UV ? 3 : (slot >= S_V ? 2 : 1);
      if ((OI_BWD_Q24_SET >> fam) & 1) v = unpack24f(pack24f(v));
    }
#endif
    if constexpr (PACK == 1) __builtin_amdgcn_raw_buffer_store_b96(pack24f(v), rs, l12 + (slot * 16384 + g * 768), 0, AUX);
    else if constexpr (PACK == 2) __builtin_amdgcn_raw_buffer_store_b64(pack16f(v), rs, l12 + (slot * 16384 + g * 512), 0, AUX);
    else oi::buffer_store_b128<AUX>(__builtin_bit_cast(u32x4, v), rs, l16, slot * 16384 + g * 1024);
  }
  template <int AUX = OI_BWD_ST_LOCAL>
  __device__ __forceinline__ void store_phase(int slot, int g, int l16, f32x4 v) const {
    if constexpr (PHQ) __builtin_amdgcn_raw_buffer_store_b96(pack_q24_phase(v), rs, l12 + (slot * 16384 + g * 768), 0, AUX);
    else if constexpr (PACK == 1) __builtin_amdgcn_raw_buffer_store_b96(pack24q(v), rs, l12 + (slot * 16384 + g * 768), 0, AUX);
    else if constexpr (PACK == 2) __builtin_amdgcn_raw_buffer_store_b64(pack16q(v), rs, l12 + (slot * 16384 + g * 512), 0, AUX);
    else oi::buffer_store_b128<AUX>(__builtin_bit_cast(u32x4, v), rs, l16, slot * 16384 + g * 1024);
  }
  // A parked fragment as it arrives (2, 3 or 4 dwords); unpacked where it is USED -- the ring of the down sweep requests
  // fragments a layer ahead, and an unpack next to the load would wait for it on the spot
  using Frag = std::conditional_t<PACK == 1, u32x3, std::conditional_t<PACK == 2, u32x2, f32x4>>;
  template <int AUX = OI_BWD_LD_LAST>
  __device__ __forceinline__ Frag load(int slot, int g, int l16) const {
    if constexpr (PACK == 1) return __builtin_amdgcn_raw_buffer_load_b96(rs, l12, slot * 16384 + g * 768, AUX);
    else if constexpr (PACK == 2) return __builtin_amdgcn_raw_buffer_load_b64(rs, l12, slot * 16384 + g * 512, AUX);
    else return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, l16, slot * 16384 + g * 1024, AUX));
  }
  static __device__ __forceinline__ f32x4 value(const Frag& f) {
    if constexpr (PACK == 1) return unpack24f(f); else if constexpr (PACK == 2) return unpack16f(f); else return f;
  }
  static __device__ __forceinline__ f32x4 phase(const Frag& f) {
    if constexpr (PACK == 1) return unpack24q(f); else if constexpr (PACK == 2) return unpack16q(f); else return f;
  }
  // phase slots under PHQ: three dwords per four values, whatever the other slots are
  using PFrag = std::conditional_t<PHQ, u32x3, Frag>;
  template <int AUX = OI_BWD_LD_LAST>
  __device__ __forceinline__ PFrag load_phase(int slot, int g, int l16) const {
    if constexpr (PHQ) return __builtin_amdgcn_raw_buffer_load_b96(rs, l12, slot * 16384 + g * 768, AUX);
    else return load<AUX>(slot, g, l16);
  }
  static __device__ __forceinline__ f32x4 phase_of(const PFrag& f) {   // (PHQ: phase + 1, which is what the trig wants anyway)
    if constexpr (PHQ) return unpack_q24(f); else return phase(f);
  }
};

// (`tid` goes through an empty asm: the destination address is then formed HERE -- hipcc otherwise computes the 64-bit
// per-lane addresses of every flush in the kernel's prologue and keeps, i.e. spills, them across both sweeps)
__device__ __forceinline__ int late(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// Sums over the points of a wave tile (parameter gradients of the FiLM rows, biases and heads).  The point sits on the
// lane (j = lane & 31), so every value needs a cross-lane sum; done one value at a time (5 DPP adds + a 2-lane LDS atomic
// behind a branch each) this was half of the instructions of the down sweep.  Instead 16 values (4 groups x 4 features of
// one 32-feature block) go through ONE transposed butterfly: at every step two registers become one -- each lane keeps
// the partial sum of the value its lane bit selects (2 selects + 1 DPP add) -- so after 4 steps lane r of every 16-lane
// row holds the row's sum of value r: 45 instructions per 16 values, no branch, and one full-wave ds_add_f32 (the two rows
// of a half hit the same address).
template <int CTRL>
__device__ __forceinline__ float fold_pair(float a, float b, bool bit) {
  const float keep = bit ? b : a, give = bit ? a : b;
  return keep + oi::dpp_mov<CTRL>(give);
}
__device__ __forceinline__ float row_transpose_sum16(const float (&v)[16], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
  float a[8], b[4], c[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = fold_pair<0xB1>(v[2 * i], v[2 * i + 1], b0);   // quad_perm [1,0,3,2]
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = fold_pair<0x4E>(a[2 * i], a[2 * i + 1], b1);   // quad_perm [2,3,0,1]
#pragma unroll
  for (int i = 0; i < 2; ++i) c[i] = fold_pair<0x124>(b[2 * i], b[2 * i + 1], b2);  // row_ror:4 (adjacent quads)
  return fold_pair<0x128>(c[0], c[1], b3);                                          // row_ror:8
}
// v[4 * rr + k] = value of feature 32 t + 8 rr + 4 h + k at this lane's point -> LDS accumulator row `row`
struct RowSum {
  float* racc;  // the workgroup's reduction rows (wave-uniform)
  int lane;
  __device__ __forceinline__ void add(int row, int t, const float (&v)[16]) const {
    // the lane's column, 8 * ((lane & 15) >> 2) + 4 * h + (lane & 3), is formed here (three instructions): kept in a
    // register from the prologue it was spilled across both sweeps
    const int l = late(lane);
    atomicAdd(racc + row * C + 32 * t + 8 * ((l & 15) >> 2) + 4 * (l >> 5) + (l & 3), row_transpose_sum16(v, lane));
  }
};

__device__ __forceinline__ void racc_zero(char* lds, int tid) {
  float* racc = reinterpret_cast<float*>(lds + L_RACC);
  for (int i = tid; i < RACC_ROWS * C; i += BW_THREADS) racc[i] = 0.f;
}
// flush `rows` accumulator rows: row r goes to dst[r] (a global base pointer per row)
__device__ __forceinline__ void racc_flush_row(char* lds, int row, float* dst, int stride, int tid) {
  const float* racc = reinterpret_cast<const float*>(lds + L_RACC) + row * C;
  tid = late(tid);
  if (tid < C) atomicAdd(dst + tid * stride, racc[tid]);
}
// the same row times a per-feature factor (sum_p ubar = gamma * sum_p phibar: the bias gradient needs no sum of its own)
__device__ __forceinline__ void racc_flush_row_scaled(char* lds, int row, const float* factor, float* dst, int stride, int tid) {
  const float* racc = reinterpret_cast<const float*>(lds + L_RACC) + row * C;
  tid = late(tid);
  if (tid < C) atomicAdd(dst + tid * stride, racc[tid] * factor[tid]);
}

// FiLM rows of the backward kernels (one 1536-byte slot): [gamma | G | B2] with the phase in REVOLUTIONS,
//   phi_l / 2pi = G * acc + B2,   G = gamma / 2pi * 2^-k_img (acc = scaled-image product without bias),
//   B2 = gamma / 2pi * b_l + beta / 2pi
constexpr float INV_2PI = 0.15915494309189533577f;
// In two halves: the global loads are issued BEFORE the layer image's LDS-DMA and the LDS writes happen after the epilogue.
// (vmcnt retires in order: with the loads issued behind the 16 DMA instructions, storing the rows drained the DMA right
// after it was requested -- one exposed round trip per layer.)
struct FilmRegs {
  float g, b, bias, inv_img;
};
__device__ __forceinline__ FilmRegs film_load(const float* __restrict__ gamma, const float* __restrict__ beta,
                                              const float* __restrict__ hdr, int e, int l, float inv_img, int tid) {
  // unconditional (the upper half of the workgroup re-reads the same rows): a branch here made hipcc finish the arithmetic
  // on the loaded values inside it, i.e. wait for the loads on the spot
  const int f = late(tid) & (C - 1);
  FilmRegs r;
  r.g = gamma[((size_t)e * 9 + l) * C + f];
  r.b = beta[((size_t)e * 9 + l) * C + f];
  r.bias = hdr[H_BIAS + l * C + f];
  r.inv_img = inv_img;
  return r;
}
__device__ __forceinline__ void film_store(char* lds, const FilmRegs& r, int tid) {
  float* film = reinterpret_cast<float*>(lds + L_FILM);
  tid = late(tid);
  if (tid < C) {
    const float gr = r.g * INV_2PI;
    film[tid] = r.g;
    film[C + tid] = gr * r.inv_img;
    film[2 * C + tid] = fmaf(gr, r.bias, r.b * INV_2PI);
  }
}
__device__ __forceinline__ void stage_film_bwd(char* lds, const float* __restrict__ gamma, const float* __restrict__ beta,
                                               const float* __restrict__ hdr, int e, int l, float inv_img, int tid) {
  film_store(lds, film_load(gamma, beta, hdr, e, l, inv_img, tid), tid);
}
// sin / cos of a phase in revolutions.  REDUCED: the value is already in [0, 1) (parked by the up sweep)
template <bool FAST, bool REDUCED>
__device__ __forceinline__ float rev_reduce(float x) {
  return (FAST || REDUCED) ? x : __builtin_amdgcn_fractf(x);
}
__device__ __forceinline__ void sincos_rev(float r, float& s, float& c) {
  s = __builtin_amdgcn_sinf(r);
  c = __builtin_amdgcn_cosf(r);
}

// this wave's LDS-DMA has landed (and so have its outstanding scratch loads), then rendezvous
__device__ __forceinline__ void dma_sync() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}
// the same, but the KEEP most recently issued vector-memory operations (scratch stores nobody waits for) stay in flight:
// the counter retires in issue order, so everything older -- the LDS-DMA -- has landed
template <int KEEP>
__device__ __forceinline__ void dma_sync_keep() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
  __syncthreads();
}
#ifndef OI_BWD_UP_KEEP
#define OI_BWD_UP_KEEP 32
#endif

// One layer product of the backward sweeps: acc = W_img . v with only ONE k-step of A fragments (4 output blocks x hi / lo
// limb = 32 VGPRs) live at a time -- two 64-register point vectors and the accumulators are live around every product of
// these sweeps, and two waves share a SIMD (256 registers each); the co-resident wave covers the LDS latency.
template <int PREC>
__device__ __forceinline__ void gemm_lean(const char* lds, const LaneOff& o, const float (&v)[64], f32x16 (&acc)[4]) {
  if constexpr (PREC == OI_PREC_F16X3) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      f16x8 bh, bl;
      split8_pairs(&v[8 * s], bh, bl);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f16x8 wh = __builtin_bit_cast(f16x8, lds_f4(lds, L_WBUF + (t * 8 + s) * 1024, o.l16));
        const f16x8 wl = __builtin_bit_cast(f16x8, lds_f4(lds, L_WBUF + (t * 8 + s) * 1024, o.l16hi));
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh, acc[t], 0, 0, 0);
#if OI_BWD_NW == 8
        if (t & 1) __builtin_amdgcn_sched_barrier(0);  // (256 registers: at most two output blocks' fragments in flight)
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    gemm_layer<PREC>(lds, o, v, acc);
  }
}
// acc = W_img . v (accumulators come in zeroed or holding the bias); returns the factor the accumulators still carry:
// 1 for the unscaled images, 2^-k_m (x 1 / normalisation of v, which is scaled in place) for F16X3
template <int PREC, bool NORM>
__device__ __forceinline__ float gemm2(const char* lds, const LaneOff& o, float (&v)[64], f32x16 (&acc)[4], float inv_img,
                                       float* lane_max = nullptr) {
  float f = 1.f;
  if constexpr (PREC == OI_PREC_F16X3) {
    f = inv_img;
    if constexpr (NORM) f *= pow2_normalise(v, lane_max);
  }
  gemm_lean<PREC>(lds, o, v, acc);
  return f;
}

// Launch-wide maxima of the weight-gradient operands (header of the scratch buffer, zeroed per launch).  The sweep
// already takes the per-point maximum of every adjoint vector it feeds to a product; the wave maximum of those is
// published here (one atomic per wave, layer and operand: non-negative floats order like their bit patterns), so that the
// weight-gradient GEMM can run on ONE power-of-two scale per operand instead of normalising every 32-point tile.
constexpr int OM_V = 0;    // 7: max |v_l / gamma_l|, l = 1..7   (the PARKED operands)
constexpr int OM_U = 7;    // 7: max |phibar_l|,      l = 1..7
constexpr int OM_G = 14;   // 7: max |gbar_l|, l = 1..7
constexpr int OM_UV = 21;  // 1: max |phibar_v|
// same-address atomics serialise in the L2 (16,384 waves x 22 slots on 22 addresses cost 0.7 ms): 64 replicas of the
// table, chosen by workgroup; the GEMM takes the maximum over the replicas
constexpr int OM_STRIDE = 32, OM_REPLICAS = 64;
constexpr int OM_FLOATS = OM_STRIDE * OM_REPLICAS;  // 8 KiB header: the tiles behind it stay 256-byte aligned
__device__ __forceinline__ void publish_max(float* op_max, int slot, float lane_max) {
  const float m = oi::wave_max(lane_max);
  const int rep = (blockIdx.x + blockIdx.y * gridDim.x) & (OM_REPLICAS - 1);
  if ((threadIdx.x & 63) == 0)
    atomicMax(reinterpret_cast<unsigned*>(op_max) + rep * OM_STRIDE + slot, __builtin_bit_cast(unsigned, m));
}

// timing ablations (results are garbage): -DOI_BWD_ABL=1 no v / ubar stores, 2 no phi / vbar reloads, 4 no up-sweep stores
#ifndef OI_BWD_ABL
#define OI_BWD_ABL 0
#endif
#ifndef OI_BWD_COL_FENCE
#define OI_BWD_COL_FENCE 1
#endif
#ifndef OI_BWD_STORES_LAST
#define OI_BWD_STORES_LAST 1
#endif
// -DOI_BWD_PROF: per-phase shader-clock accounting of the sweep (tools/dbg/phase_prof_bwd.py)
#ifdef OI_BWD_PROF
__device__ unsigned long long oi_prof_bwd[24];
__device__ unsigned long long oi_prof_bwd_wg[4 * 4096];   // per workgroup: start, end (100 MHz), HW_ID, XCC_ID
#define BW_T(i)                                                  \
  do {                                                           \
    const unsigned long long t_ = __builtin_readcyclecounter();  \
    pacc[i] += t_ - tprev;                                       \
    tprev = t_;                                                  \
  } while (0)
#else
#define BW_T(i)
#endif
// section markers in the assembly listing (comments only): tools/isa_mix.py --sections splits the instruction mix on them,
// the up / down layer bodies are run-time loops and execute 7 times per tile
#define OI_MARK(name) asm volatile("; OI_MARK " name)
// GFEAT: an upstream gradient of the FEATURE output a_8 (`g_feat`, [B n][128]) joins abar_8 at the turn -- a caller that keeps the
// reference's renderer.py:241-261 reads the features out (ShapeNetwork.forward) and feeds them to ColorNetwork.forward, so their
// gradient comes back from outside.  A separate instantiation: the fused path (GFEAT = false) compiles exactly as before.
template <int PREC, bool FAST, bool GFEAT = false>
__global__ void __launch_bounds__(BW_THREADS)
mlp_bwd_sweep_kernel(const float* __restrict__ pts, const char* __restrict__ packed, const float* __restrict__ gamma,
                     const float* __restrict__ beta, const float* __restrict__ grad_fwd,
                     const float* __restrict__ rgb_fwd, const float* __restrict__ feat_fwd,
                     const float* __restrict__ g_sdf, const float* __restrict__ g_grad,
                     const float* __restrict__ g_rgb, const float* __restrict__ g_feat, float* __restrict__ d_small, float* __restrict__ d_gamma,
                     float* __restrict__ d_beta, char* __restrict__ scratch, float* __restrict__ op_max,
                     long long n_per_elem, long long n_stride, long long pt_off) {
  // this launch covers points [pt_off, pt_off + n_per_elem) of every batch element; an element holds n_stride points
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int e = blockIdx.y;
  const float* hdr = reinterpret_cast<const float*>(packed);
  const char* mats = packed + H_BYTES;
  const bool has_col = rgb_fwd != nullptr && g_rgb != nullptr && feat_fwd != nullptr;
  const int n_tiles = (int)((n_per_elem + BW_TILE - 1) / BW_TILE);   // of this launch, per batch element
#if OI_BWD_PERSIST
  static_assert(OI_BWD_LATE_FLUSH == 1, "persistent workgroups keep their point sums in LDS across tiles");
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
  const bool first = tile == (int)blockIdx.x, last = tile + (int)gridDim.x >= n_tiles;
#else
  {
  const int tile = blockIdx.x;
  constexpr bool first = true, last = true;
#endif
  // (the thread index is taken anew per tile, through an empty asm: everything derived from it -- lane offsets, the per-lane
  // addresses of the FiLM rows and tables -- would otherwise be hoisted out of the tile loop and kept, i.e. spilled, across it)
  const int tid = late(threadIdx.x), lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, j = lane & 31;

  LaneOff o;
  o.h16 = 16 * h;
  o.h64 = 64 * h;
  o.l16 = 16 * lane;
  o.l16hi = 16 * lane + 32768;
  asm volatile("" : "+v"(o.h16), "+v"(o.h64), "+v"(o.l16), "+v"(o.l16hi));

  const RowSum rs{reinterpret_cast<float*>(lds + L_RACC), lane};
  const long long local = (long long)tile * BW_TILE + wave * WAVE_PTS + j;
  const bool valid = local < n_per_elem;
  const long long pt = (long long)e * n_stride + pt_off + (valid ? local : n_per_elem - 1);
  const float vmask = valid ? 1.f : 0.f;  // tail points contribute nothing
  const __amdgpu_buffer_rsrc_t img_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(mats), 0, NMAT * layer_bytes(PREC), 0x00020000);

  // (24-bit: accurate trig only -- fast trig parks unreduced phases; the 16-bit format reduces them as it packs)
  constexpr int PK = (OI_BWD_PACK16 && OI_WGRAD_BF16 && PREC == OI_PREC_BF16) ? 2 : ((OI_BWD_PACK24 && PREC == OI_PREC_F16X3 && !FAST) ? 1 : 0);
  constexpr bool PHQ = OI_BWD_PHQ24 && PK == 0 && PREC == OI_PREC_F16X3 && !FAST;   // phase slots as 24-bit fixed point
  WaveScratchT<PK, PHQ> ws;
  constexpr bool XQ = OI_BWD_XQ24 && PK == 0 && PREC == OI_PREC_F16X3 && OI_BWD_STORES_LAST;  // S_V / S_U as 24-bit fixed point
  constexpr bool VBQ = OI_BWD_VBQ24 && PK == 0 && PREC == OI_PREC_F16X3;                      // S_VB likewise
  ws.l12 = (PK == 2 ? 8 : 12) * lane;
  asm volatile("" : "+v"(ws.l12));
  {
    const long long wt = ((long long)e * n_tiles + tile) * BW_NW + wave;
    ws.rs = __builtin_amdgcn_make_buffer_rsrc(scratch + wt * (long long)(NSLOT_BWD * 16384), 0, NSLOT_BWD * 16384,
                                              0x00020000);
  }
  // image slot ws_ (0 / 1) and FiLM slot fs_ (0 / 1) as lane bases: every LDS access stays <per-lane VGPR> + immediate
  auto layer_off = [&](int ws_, int fs_) {
    LaneOff r = o;
    r.l16 += ws_ * 65536;
    r.l16hi += ws_ * 65536;
    r.h16 += fs_ * (L_FILM2 - L_FILM);
    return r;
  };
  auto stage_img = [&](int image, int ws_) {  // 1 KiB per instruction, the chunks dealt round-robin to the waves
    constexpr int NCHUNK = layer_bytes(PREC) / 1024;
#pragma unroll
    for (int c0 = 0; c0 < NCHUNK / BW_NW; ++c0) {
      const int c = c0 * BW_NW + wave;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(img_rs, (__attribute__((address_space(3))) void*)(lds + ws_ * 65536 + L_WBUF + c * 1024),
                                               16, o.l16, image * layer_bytes(PREC) + c * 1024, 0, 0);
    }
  };
  // FiLM rows of layer l_; the scale of the forward image that produces u_l (layer 0 runs on the VALU, colour head = image 14)
  auto film_scale = [&](int l_) {
    return (PREC == OI_PREC_F16X3 && l_ >= 1) ? hdr[H_WSCALE + (l_ == 8 ? 14 : l_ - 1)] : 1.f;
  };
  auto stage_flm = [&](int l_, int fs_) {
    stage_film_bwd(lds + fs_ * (L_FILM2 - L_FILM), gamma, beta, hdr, e, l_, film_scale(l_), tid);
  };
  auto load_flm = [&](int l_) { return film_load(gamma, beta, hdr, e, l_, film_scale(l_), tid); };
  auto store_flm = [&](const FilmRegs& r, int fs_) { film_store(lds + fs_ * (L_FILM2 - L_FILM), r, tid); };
  // Everything the tile needs from memory before its first product is requested up front (round 2 paid four exposed
  // round trips in the colour head: tables, image 14, the per-point forward values, image 15): both colour images go to
  // the two image slots, the colour head's FiLM rows to FiLM slot 1 and layer 0's to slot 0.
#ifdef OI_BWD_PROF
  unsigned long long pacc[20] = {0};
  unsigned long long tprev = __builtin_readcyclecounter();
  const unsigned long long tstart = tprev;
  const unsigned long long rstart = wall_clock64();   // (100 MHz, shared by the chip: workgroup lifetimes against the launch's span)
#endif
  float act[64];  // up sweep: a_l;     down sweep: abar_{l+1} -> ubar_l -> abar_l
  // The vector-memory counter retires in issue order, and the colour images are 128 KiB of LDS-DMA: every register load of the
  // prologue is issued AHEAD of them (tables, both sets of FiLM rows, the point, its upstream gradients and forward values), so
  // that the LDS copies of the tables wait for those loads only -- round 5 paid four dependent round trips here (DMA, rows, tables,
  // point) before the first product.
  // (tables, layer 0's rows and the zeroed reduction rows: the workgroup's first tile only -- the down sweep leaves layer 0's rows
  // in FiLM slot 0 again)
  FilmRegs fr8{}, fr1{}, fr0{};
  constexpr int NTAB = (H_TABS_END + BW_THREADS - 1) / BW_THREADS;
  float tabv[NTAB];
  if (first) {
    fr0 = load_flm(0);
#pragma unroll
    for (int i = 0; i < NTAB; ++i) tabv[i] = hdr[min(tid + i * BW_THREADS, H_TABS_END - 1)];
  }
  if (has_col) fr8 = load_flm(8);
  const float px = pts[pt * 3 + 0], py = pts[pt * 3 + 1], pz = pts[pt * 3 + 2];
  const float gs = (g_sdf ? g_sdf[pt] : 0.f) * vmask;
  float Gx = (g_grad ? g_grad[pt * 3 + 0] : 0.f) * vmask, Gy = (g_grad ? g_grad[pt * 3 + 1] : 0.f) * vmask,
        Gz = (g_grad ? g_grad[pt * 3 + 2] : 0.f) * vmask;
  float fx = 0.f, fy = 0.f, fz = 0.f, rho[3] = {0.f, 0.f, 0.f};
  if (has_col) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {  // a_8, as the forward wrote it (feat output): features grp_f0(g) + 4 h .. + 3
      const f32x4 v = *reinterpret_cast<const f32x4*>(feat_fwd + pt * C + grp_f0(g) + 4 * h);
#pragma unroll
      for (int k = 0; k < 4; ++k) act[4 * g + k] = v[k];
    }
    fx = grad_fwd[pt * 3 + 0]; fy = grad_fwd[pt * 3 + 1]; fz = grad_fwd[pt * 3 + 2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float rv = rgb_fwd[pt * 3 + k];
      rho[k] = g_rgb[pt * 3 + k] * rv * (1.0f - rv) * vmask;  // through the sigmoid
    }
    stage_img(14, 0);
    stage_img(15, 1);
  }
  if (first) {
    float* tabs = reinterpret_cast<float*>(lds + L_TABS);
#pragma unroll
    for (int i = 0; i < NTAB; ++i)
      if (tid + i * BW_THREADS < H_TABS_END) tabs[tid + i * BW_THREADS] = tabv[i];
    store_flm(fr0, 0);
    racc_zero(lds, tid);
  }
  if (has_col) store_flm(fr8, 1);
  __syncthreads();

  float gb[64];   // up sweep: gbar_l -> vbar_l -> gbar_{l+1};   down sweep: g_{l+1} -> v_l -> g_l
  f32x16 acc[4];
  // F16X3: the images carry a power-of-two scale 2^k_m (header H_WSCALE holds 2^-k_m); every product returns the factor
  // its accumulators still need (gemm2), adjoint vectors are normalised per point before the fp16 split.
  constexpr bool SC = PREC == OI_PREC_F16X3;

  // the colour head's point sums, rows 2 / 6 / 7 of sums of phibar_v (not of uvbar = gamma_v phibar_v):
  //   d gamma_v += sum_{j < 3} Wv[f][128 + j] R_j   (the GEMM adds the other 128 columns and bv),   dWv[f][128 + j] = gamma_v[f] R_j
  // and rows 3..5 = d Wrgb; gv_row = gamma_v (its LDS copy while the head runs, the caller's array at the end of the kernel)
  auto flush_colour = [&](const float* gv_row) {
    if (const int t_ = late(tid); t_ < C) {
      const float* racc = reinterpret_cast<const float*>(lds + L_RACC) + RR_COL * C;
      const f32x4 wx = *reinterpret_cast<const f32x4*>(lds + L_TABS + H_TABV * 4 + t_ * 16);
      atomicAdd(d_gamma + ((size_t)e * 9 + 8) * C + t_,
                fmaf(wx[0], racc[6 * C + t_], fmaf(wx[1], racc[7 * C + t_], wx[2] * racc[2 * C + t_])));
    }
    racc_flush_row_scaled(lds, RR_COL + 2, gv_row, d_small + DS_WVX + 2, 3, tid);
    racc_flush_row(lds, RR_COL + 3, d_small + DS_WRGB + 0 * C, 1, tid);
    racc_flush_row(lds, RR_COL + 4, d_small + DS_WRGB + 1 * C, 1, tid);
    racc_flush_row(lds, RR_COL + 5, d_small + DS_WRGB + 2 * C, 1, tid);
    racc_flush_row_scaled(lds, RR_COL + 6, gv_row, d_small + DS_WVX + 0, 3, tid);
    racc_flush_row_scaled(lds, RR_COL + 7, gv_row, d_small + DS_WVX + 1, 3, tid);
  };
  // ================= colour head backward (first: its dL/dgrad term is part of gbar_0) =================
  OI_MARK("colour x1");
  // abar_8 contribution of the colour head: waits for the down sweep in registers (one wave per SIMD: the AGPR half has
  // room) or in scratch slot S_AC
#if OI_BWD_AC_REGS
  float ac[64];
#pragma unroll
  for (int k = 0; k < 64; ++k) ac[k] = 0.f;
#endif
  if (has_col) {
    dma_sync();
    BW_T(14);
    const LaneOff oc = layer_off(0, 1);  // image slot 0, FiLM slot 1
    acc_zero(acc);
    (void)gemm2<PREC, false>(lds, oc, act, acc, 1.f);
    __syncthreads();   // every wave is done with image slot 0:
    stage_img(0, 0);   // the up sweep's first image travels under the epilogue
    BW_T(15);
    // uv -> phiv -> hv; then uvbar.  Point sums kept here: rows 3..5 dWrgb, rows 2 / 6 / 7 dWv[:, 130 / 128 / 129]; the FiLM and
    // bias gradients of the head come out of the weight-gradient GEMM (FiLM-scale identity), the part of it that belongs to
    // the three extra input columns is added at the flush below
    float dGx = 0.f, dGy = 0.f, dGz = 0.f, mx_pv = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float uv16[16], hv16[16];  // the six summed rows are products of these with per-point scalars: formed one at a time
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int g = 4 * t + rr;
        const f32x4 gm = lds_f4(lds, L_FILM + grp_f0(g) * 4, oc.h16);
        const f32x4 gr = lds_f4(lds, L_FILM + (C + grp_f0(g)) * 4, oc.h16);
        const f32x4 b2 = lds_f4(lds, L_FILM + (2 * C + grp_f0(g)) * 4, oc.h16);
        const f32x4 w0 = lds_f4(lds, L_TABS + (H_RGB + 0 * C + grp_f0(g)) * 4, o.h16);
        const f32x4 w1 = lds_f4(lds, L_TABS + (H_RGB + 1 * C + grp_f0(g)) * 4, o.h16);
        const f32x4 w2 = lds_f4(lds, L_TABS + (H_RGB + 2 * C + grp_f0(g)) * 4, o.h16);
        f32x4 uvb, pvb;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const f32x4 wx = lds_f4(lds, L_TABS + H_TABV * 4 + (grp_f0(g) + k) * 16, o.h64);
          // phase in revolutions: gamma/2pi (W a_8 + Wx grad + b) + beta/2pi
          const float dx = fmaf(fz, wx[2], fmaf(fy, wx[1], fx * wx[0]));
          const float ph = fmaf(gr[k], acc[t][4 * rr + k], fmaf(gm[k] * INV_2PI, dx, b2[k]));
          float hv, cv;
          sincos_rev(rev_reduce<FAST, false>(ph), hv, cv);
          const float hvb = w0[k] * rho[0] + w1[k] * rho[1] + w2[k] * rho[2];
          pvb[k] = hvb * cv;        // phibar_v: what is parked and summed (no gamma_v: see the header comment)
          uvb[k] = pvb[k] * gm[k];  // uvbar = gamma_v phibar_v: what travels on to abar_8 and dL/dgrad
          mx_pv = fmaxf(mx_pv, fabsf(pvb[k]));
          uv16[4 * rr + k] = pvb[k];
          hv16[4 * rr + k] = hv;
          dGx = fmaf(uvb[k], wx[0], dGx);
          dGy = fmaf(uvb[k], wx[1], dGy);
          dGz = fmaf(uvb[k], wx[2], dGz);
        }
        ws.store(S_UV, g, o.l16, pvb);
#pragma unroll
        for (int k = 0; k < 4; ++k) act[4 * g + k] = uvb[k];
#if OI_BWD_COL_FENCE
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
#pragma unroll
      for (int r = 2; r < 8; ++r) {
        const float sc = r == 2 ? fz : r == 6 ? fx : r == 7 ? fy : rho[r - 3];
        float row[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) row[i] = (r >= 3 && r <= 5) ? sc * hv16[i] : uv16[i] * sc;
        rs.add(RR_COL + r, t, row);
      }
    }
    BW_T(16);
    // contribution to dL/dgrad through the colour-head input
    dGx += __shfl_xor(dGx, 32, 64);
    dGy += __shfl_xor(dGy, 32, 64);
    dGz += __shfl_xor(dGz, 32, 64);
    Gx += dGx; Gy += dGy; Gz += dGz;
    {  // brgb
      float b0 = rho[0], b1 = rho[1], b2 = rho[2];
      if (h != 0) { b0 = 0.f; b1 = 0.f; b2 = 0.f; }
      b0 = oi::wave_sum(b0); b1 = oi::wave_sum(b1); b2 = oi::wave_sum(b2);
      if (lane == 0) {
        atomicAdd(d_small + DS_BRGB + 0, b0);
        atomicAdd(d_small + DS_BRGB + 1, b1);
        atomicAdd(d_small + DS_BRGB + 2, b2);
      }
    }
    if constexpr (!OI_BWD_LATE_FLUSH) {
      __syncthreads();
      flush_colour(reinterpret_cast<const float*>(lds + L_FILM2));  // the head's rows sit in FiLM slot 1
      __syncthreads();
      racc_zero(lds, tid);
    }
    BW_T(17);
    // abar_8 from the colour head: Wv[:, :128]^T uvbar   (transposed colour image, matrix 15: resident in slot 1 since the
    // prologue)
    acc_zero(acc);
    fr1 = load_flm(1);  // layer 1's FiLM rows travel under the product
    const float fT = gemm2<PREC, true>(lds, layer_off(1, 1), act, acc, SC ? hdr[H_WSCALE + 15] : 1.f);
    BW_T(18);
    if constexpr (SC) publish_max(op_max, OM_UV, mx_pv);  // (of the parked operand: phibar_v)
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = SC ? acc[g >> 2][4 * (g & 3) + k] * fT : acc[g >> 2][4 * (g & 3) + k];
      if constexpr (GFEAT) {
        const f32x4 gf = *reinterpret_cast<const f32x4*>(g_feat + pt * C + grp_f0(g) + 4 * h);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaf(gf[k], vmask, v[k]);
      }
#if OI_BWD_AC_REGS
#pragma unroll
      for (int k = 0; k < 4; ++k) ac[4 * g + k] = v[k];
#else
      ws.store(S_AC, g, o.l16, v);
#endif
    }
  } else if constexpr (GFEAT) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      f32x4 v = *reinterpret_cast<const f32x4*>(g_feat + pt * C + grp_f0(g) + 4 * h);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] *= vmask;
#if OI_BWD_AC_REGS
#pragma unroll
      for (int k = 0; k < 4; ++k) ac[4 * g + k] = v[k];
#else
      ws.store(S_AC, g, o.l16, v);
#endif
    }
  }

  BW_T(0);
  OI_MARK("up0 x1");
  // ================= up sweep: recompute phi_l, carry gbar_l =================
  // FiLM rows of layer l live in FiLM slot l & 1, layer l's forward image in image slot (l - 1) & 1: both are requested one
  // layer ahead
  if (!has_col) {
    fr1 = load_flm(1);
    stage_img(0, 0);  // (with a colour head: requested right after its first product)
  }
  __syncthreads();    // the colour head is done with FiLM slot 1
  store_flm(fr1, 1);
  // layer 0 on the VALU: phi_0; vbar_0 = W0 gbar_0 (gbar_0 = dL/dgrad, a 3-vector); gbar_1 = vbar_0 c_0
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const f32x4 gm = lds_f4(lds, L_FILM + grp_f0(g) * 4, o.h16);
    const f32x4 gr = lds_f4(lds, L_FILM + (C + grp_f0(g)) * 4, o.h16);
    const f32x4 b2 = lds_f4(lds, L_FILM + (2 * C + grp_f0(g)) * 4, o.h16);
    f32x4 ph, vb;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 w = lds_f4(lds, L_TABS + H_TAB0 * 4 + (grp_f0(g) + k) * 16, o.h64);
      const float u = fmaf(pz, w[2], fmaf(py, w[1], px * w[0]));
      ph[k] = rev_reduce<FAST, false>(fmaf(gr[k], u, b2[k]));
      float s, c;
      sincos_rev(ph[k], s, c);
      act[4 * g + k] = s;
      vb[k] = fmaf(Gz, w[2], fmaf(Gy, w[1], Gx * w[0])) * gm[k];  // gamma folded in
      gb[4 * g + k] = vb[k] * c;
      // (nothing with a side effect is left in this loop: without the pin hipcc reads all 64 table rows first and spills)
      asm volatile("" : "+v"(act[4 * g + k]), "+v"(gb[4 * g + k]));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (h == 0) {  // what the weight-gradient GEMM needs to form phi_0 / vbar_0 itself (lane j = point j)
    oi::buffer_store_b128<OI_BWD_ST_WGRAD>(__builtin_bit_cast(u32x4, f32x4{px, py, pz, 0.f}), ws.rs, 32 * j, S_PHI * 16384);
    oi::buffer_store_b128<OI_BWD_ST_WGRAD>(__builtin_bit_cast(u32x4, f32x4{Gx, Gy, Gz, 0.f}), ws.rs, 32 * j + 16, S_PHI * 16384);
  }
  BW_T(1);
  // The last layer is peeled: its phase and vbar are what the down sweep consumes FIRST, so they stay in the two point
  // vectors across the turn instead of making a round trip through the scratch (0.8 GB per launch at C2: vbar_7 is never
  // written, phi_7 is written for the colour head's weight gradient only); the w_sigma sums, which need a_8 and gbar_8, are
  // formed inside its epilogue.
  auto up_layer = [&](int l, auto is_last) {
    constexpr bool LAST = decltype(is_last)::value;
    if constexpr (LAST) OI_MARK("up_last x1"); else OI_MARK("up_body x6");
    // layer l's image and FiLM rows have landed; every wave is done with layer l - 1.  The 32 phi / vbar stores of the
    // previous layer were issued after that DMA and need not have drained.
    if constexpr (OI_BWD_ABL & 4) dma_sync(); else dma_sync_keep<OI_BWD_UP_KEEP>();
    BW_T(2);
    const FilmRegs fr = load_flm(LAST ? l : l + 1);  // next layer's FiLM rows: requested ahead of the image DMA
    stage_img(LAST ? 13 : l, l & 1);  // (the down sweep starts with the transposed image of layer 7)
    const LaneOff ol = layer_off((l - 1) & 1, l & 1);
    const float inv_img = SC ? hdr[H_WSCALE + l - 1] : 1.f;
    // vbar_l = W_l gbar_l
    acc_zero(acc);
    BW_T(3);
    float mx_g = 0.f;
    const float fA = gemm2<PREC, true>(lds, ol, gb, acc, inv_img, &mx_g);
    if constexpr (SC) publish_max(op_max, OM_G + l - 1, mx_g);
    BW_T(4);
#pragma unroll
    for (int g = 0; g < 16; ++g) {  // gamma_l vbar_l (gamma folded in: what both the down sweep and the GEMM consume)
      const f32x4 gm = lds_f4(lds, L_FILM + grp_f0(g) * 4, ol.h16);
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] = SC ? acc[g >> 2][4 * (g & 3) + k] * (fA * gm[k]) : acc[g >> 2][4 * (g & 3) + k] * gm[k];
        gb[4 * g + k] = v[k];
      }
      if constexpr (!VBQ)
        if (!LAST && !(OI_BWD_ABL & 4)) ws.store(S_VB + l, g, o.l16, v);
    }
    if constexpr (VBQ && !LAST && !(OI_BWD_ABL & 4)) {  // the vector is complete in gb: lane maximum, then pack and store
      float mx = 0.f;
#pragma unroll
      for (int k = 0; k < 64; k += 2) mx = fmaxf(mx, fmaxf(fabsf(gb[k]), fabsf(gb[k + 1])));
      float sq, inv_sq;
      q24_scale(mx, sq, inv_sq);
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const u32x3 q = pack_q24(f32x4{gb[4 * g], gb[4 * g + 1], gb[4 * g + 2], gb[4 * g + 3]}, sq);
        __builtin_amdgcn_raw_buffer_store_b96(q, ws.rs, ws.l12 + ((S_VB + l) * 16384 + g * 768), 0, OI_BWD_ST_LOCAL);
      }
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, inv_sq), ws.rs,
                                            4 * lane + ((S_VB + l) * 16384 + XQ_SCALE_OFF), 0, OI_BWD_ST_LOCAL);
    }
    if constexpr (!LAST) store_flm(fr, (l + 1) & 1);  // FiLM slot of layer l - 1: free since this layer's barrier
    // phi_l / 2pi = G (W_img a_l) + B2 (image scale and bias folded into the staged rows) -> a_{l+1};
    // gbar_{l+1} = vbar_l gamma_l cos phi_l
    acc_zero(acc);
    BW_T(5);
    (void)gemm2<PREC, false>(lds, ol, act, acc, 1.f);
    BW_T(4);
    float wsig_row[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const f32x4 gr = lds_f4(lds, L_FILM + (C + grp_f0(g)) * 4, ol.h16);
      const f32x4 b2 = lds_f4(lds, L_FILM + (2 * C + grp_f0(g)) * 4, ol.h16);
      f32x4 ph;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ph[k] = rev_reduce<FAST, false>(fmaf(gr[k], acc[g >> 2][4 * (g & 3) + k], b2[k]));
        float s, c;
        sincos_rev(ph[k], s, c);
        if constexpr (LAST) {  // act keeps phi_7, gb keeps gamma_7 vbar_7;  d w_sigma = sum_p (gbar_8 + gs a_8)  (row 3)
          wsig_row[4 * (g & 3) + k] = fmaf(gs, s, gb[4 * g + k] * c);
          act[4 * g + k] = ph[k];
        } else {
          act[4 * g + k] = s;
          gb[4 * g + k] *= c;
        }
      }
      if (!(OI_BWD_ABL & 4)) ws.store_phase(S_PHI + l, g, o.l16, ph);
      if constexpr (LAST)
        if ((g & 3) == 3) rs.add(RR_WSIG, g >> 2, wsig_row);
      __builtin_amdgcn_sched_barrier(0);
    }
    BW_T(5);
  };
  for (int l = 1; l < NL_SDF - 1; ++l) up_layer(l, std::false_type{});
  up_layer(NL_SDF - 1, std::true_type{});
  OI_MARK("mid x1");
  // d b_sigma = sum_p gs
  {
    float b = (h == 0) ? gs : 0.f;
    b = oi::wave_sum(b);
    if (lane == 0) atomicAdd(d_small + DS_BSIG, b);
  }
  if constexpr (!OI_BWD_LATE_FLUSH) {
    __syncthreads();
    racc_flush_row(lds, RR_WSIG, d_small + DS_WSIG, 1, tid);
  }

  // ================= down sweep: g_l and abar_l together =================
  // The top layer (7) takes phi_7 / vbar_7 from the point vectors, g_8 = w_sigma from the tables and abar_8 = gs * w_sigma
  // (+ the colour head's contribution: registers or slot S_AC, which then travels through the phi half of the ring).
  // the transposed image of layer l sits in image slot l & 1, its FiLM rows in FiLM slot l & 1; phi_l / vbar_l of the WHOLE
  // layer are requested one layer ahead (128 registers: the reason this kernel runs one wave per SIMD)
  // (a ring of PF groups: with 512 registers a whole layer, PF = 16, is in flight across the two products)
  constexpr int PF = PREC == OI_PREC_F32 ? OI_BWD_PF_F32 : OI_BWD_PF, CARRY = OI_BWD_CARRY;
  static_assert(PF >= 1 && PF <= 16 && (PF & (PF - 1)) == 0 && CARRY >= 1 && CARRY <= PF, "OI_BWD_PF / OI_BWD_CARRY");
  using Frag = typename WaveScratchT<PK, PHQ>::Frag;
  using PFrag = typename WaveScratchT<PK, PHQ>::PFrag;
  using VFrag = std::conditional_t<VBQ, u32x3, Frag>;
  PFrag phn[PF];
  VFrag vbn[PF];
  Frag acn[PF < 4 ? PF : 4];   // the colour head's slot S_AC, top layer only (the two rings above are idle there)
  [[maybe_unused]] float vbinv_next = 1.f;   // VBQ: 1 / s of this lane's gamma vbar of the layer about to be swept
  auto load_vb = [&](int slot, int g) -> VFrag {
    if constexpr (VBQ) return __builtin_amdgcn_raw_buffer_load_b96(ws.rs, ws.l12, slot * 16384 + g * 768, OI_BWD_LD_LAST);
    else return ws.load(slot, g, o.l16);
  };
  auto load_vb_scale = [&](int slot) {
    if constexpr (VBQ)
      vbinv_next = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ws.rs, 4 * lane, slot * 16384 + XQ_SCALE_OFF,
                                                                                  OI_BWD_LD_LAST));
  };
  f32x4 abl_sink;  // (OI_BWD_ABL & 8)
  constexpr int PFT = PF < 4 ? PF : 4;  // ring depth of the top layer (its point vectors carry phi_7 / vbar_7 as well)
#if !OI_BWD_AC_REGS
#pragma unroll
  for (int g = 0; g < PFT; ++g) acn[g] = ws.load(S_AC, g, o.l16);  // (no colour head: never written, never used)
#endif
  BW_T(6);
  // layer 0 is peeled (its extra d W0 rows and missing products are compile-time): no branch inside the unrolled epilogue
  // KIND 0: layers 6..1 (a run-time loop), 1: layer 0, 2: the top layer 7
  auto down_layer = [&](int l, auto kind) {
    constexpr bool L0 = decltype(kind)::value == 1, TOP = decltype(kind)::value == 2;
    if constexpr (L0) OI_MARK("down0 x1"); else if constexpr (TOP) OI_MARK("down_top x1"); else OI_MARK("down_body x6");
    dma_sync();  // layer l's transposed image and FiLM rows have landed
    BW_T(7);
    if constexpr (L0 && !OI_BWD_LATE_FLUSH) {  // the reduction rows still hold the w_sigma sums of the up sweep (flushed many barriers ago)
      racc_zero(lds, tid);
      __syncthreads();
    }
    const FilmRegs fr = load_flm(l >= 1 ? l - 1 : 0);  // ahead of the image DMA (see film_load)
    [[maybe_unused]] const float vbinv = vbinv_next;   // (requested with this layer's first ring groups, a layer ago)
    if constexpr (!L0 && !TOP && CARRY < PF && !(OI_BWD_ABL & 2)) {  // groups 0 .. CARRY-1 travelled under the products; fill the ring
#pragma unroll
      for (int g = CARRY; g < PF; ++g) {
        phn[g] = ws.load_phase(S_PHI + l, g, o.l16);
        vbn[g] = load_vb(S_VB + l, g);
      }
    }
    if (l >= 2) stage_img(7 + l - 2, (l - 1) & 1);
    const LaneOff ol = layer_off(l & 1, l & 1);
    // layer 0 only: the point and dL/dgrad come back from the slot the up sweep left them in for the weight-gradient GEMM
    // (six registers that would otherwise stay live through both sweeps)
    f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, G0 = x0;
    if constexpr (L0) {
      x0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ws.rs, 32 * j, S_PHI * 16384, 0));
      G0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ws.rs, 32 * j, S_PHI * 16384 + 16, 0));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int g = 4 * t + rr;
        const f32x4 gm = lds_f4(lds, L_FILM + grp_f0(g) * 4, ol.h16);
        f32x4 ph, vb;
        if constexpr (L0) {  // formed again exactly as the up sweep did (same operations in the same order: same bits)
          const f32x4 gr = lds_f4(lds, L_FILM + (C + grp_f0(g)) * 4, ol.h16);
          const f32x4 b2 = lds_f4(lds, L_FILM + (2 * C + grp_f0(g)) * 4, ol.h16);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const f32x4 w = lds_f4(lds, L_TABS + H_TAB0 * 4 + (grp_f0(g) + k) * 16, o.h64);
            const float u = fmaf(x0[2], w[2], fmaf(x0[1], w[1], x0[0] * w[0]));
            ph[k] = rev_reduce<FAST, false>(fmaf(gr[k], u, b2[k]));
            vb[k] = fmaf(G0[2], w[2], fmaf(G0[1], w[1], G0[0] * w[0])) * gm[k];
          }
        } else if constexpr (TOP) {  // left in the point vectors by the up sweep's last layer
          ph = f32x4{act[4 * g], act[4 * g + 1], act[4 * g + 2], act[4 * g + 3]};
          vb = f32x4{gb[4 * g], gb[4 * g + 1], gb[4 * g + 2], gb[4 * g + 3]};
        } else {
          ph = ws.phase_of(phn[g & (PF - 1)]);
          if constexpr (VBQ) {
            const f32x4 yv = unpack_q24(vbn[g & (PF - 1)]);
            const float cq = -1.5f * vbinv;
            vb = f32x4{fmaf(yv[0], vbinv, cq), fmaf(yv[1], vbinv, cq), fmaf(yv[2], vbinv, cq), fmaf(yv[3], vbinv, cq)};
          } else {
            vb = ws.value(vbn[g & (PF - 1)]);
          }
        }
        f32x4 gnx, abx;  // g_{l+1}, abar_{l+1}
        if constexpr (TOP) {
          gnx = lds_f4(lds, L_TABS + (H_SIG + grp_f0(g)) * 4, o.h16);  // g_8 = w_sigma
#if OI_BWD_AC_REGS
          const f32x4 a8 = {ac[4 * g], ac[4 * g + 1], ac[4 * g + 2], ac[4 * g + 3]};
#else
          const f32x4 a8 = ws.value(acn[g & (PFT - 1)]);
#endif
#pragma unroll
          for (int k = 0; k < 4; ++k) abx[k] = (has_col || GFEAT) ? fmaf(gs, gnx[k], a8[k]) : gs * gnx[k];  // abar_8
        } else {
          gnx = f32x4{gb[4 * g], gb[4 * g + 1], gb[4 * g + 2], gb[4 * g + 3]};
          abx = f32x4{act[4 * g], act[4 * g + 1], act[4 * g + 2], act[4 * g + 3]};
        }
        f32x4 ub, vv;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float s, c;
          sincos_rev(ph[k], s, c);                                     // parked reduced phase (revolutions)
          const float gn = gnx[k];                                     // g_{l+1}
          const float cb = vb[k] * gn;                                 // gamma_l cbar_l (gamma folded into the parked vbar)
          vv[k] = gn * c;                                              // v_l / gamma_l  (parked without gamma_l, see the header)
          ub[k] = fmaf(abx[k], c, -cb * s);                            // phibar_l = ubar_l / gamma_l
          gb[4 * g + k] = vv[k];
          act[4 * g + k] = ub[k];
#if OI_BWD_NW == 8
          // (pure arithmetic is not ordered against sched_barrier: without this pin LLVM sinks the whole epilogue below
          // the 16 groups' loads and spills the point vectors to make room for 32 in-flight fragments)
          asm volatile("" : "+v"(gb[4 * g + k]), "+v"(act[4 * g + k]));
#endif
        }
#if !OI_BWD_STORES_LAST
        if constexpr (!L0 && !(OI_BWD_ABL & 1)) {
          ws.template store<OI_BWD_ST_WGRAD>(S_V + l - 1, g, o.l16, vv);
          ws.template store<OI_BWD_ST_WGRAD>(S_U + l - 1, g, o.l16, ub);
        }
#endif
        // this group's fragments are consumed: request the same group of the NEXT layer into the same registers -- the loads
        // travel under the rest of the epilogue and both products, and there is no separate issue phase
        if constexpr (!(OI_BWD_ABL & 2)) {
          // ring slot g & (PF - 1) next holds group g + PF of this layer, or group g + PF - 16 of the layer below
          if constexpr (TOP) {
#if !OI_BWD_AC_REGS
            if (g + PFT < 16) acn[g & (PFT - 1)] = ws.load(S_AC, g + PFT, o.l16);
#endif
          } else if (!L0 && g + PF < 16) {
            phn[g & (PF - 1)] = ws.load_phase(S_PHI + l, g + PF, o.l16);
            vbn[g & (PF - 1)] = load_vb(S_VB + l, g + PF);
          } else if constexpr (!L0) {
            // (layer 1 requests CARRY groups of the un-parked layer 0 as well: nobody reads them, and a run-time test of `l`
            // here splits the unrolled epilogue into blocks the register allocator handles badly -- 220 spilled registers)
            if (g + PF - 16 < CARRY) {
              phn[g & (PF - 1)] = ws.load_phase(S_PHI + l - 1, g + PF - 16, o.l16);
              vbn[g & (PF - 1)] = load_vb(S_VB + l - 1, g + PF - 16);
              if (g + PF - 16 == 0) load_vb_scale(S_VB + l - 1);
            }
          }
        }
        if constexpr (!L0 && (OI_BWD_ABL & 8)) {  // timing ablation: the loads are ISSUED but nothing ever waits for them
          asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen nt\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen offset:1024 nt"
                       : "=&v"(abl_sink) : "v"(o.l16), "s"(ws.rs));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // layer 0 only (its "weight gradient" is three columns, summed here): rows 1 d b_0 = sum ubar_0, 3..5
      // d W0[:, 0..2] = sum_p (ubar_0 x^T + v_0 gbar_0^T).  Layers 1..7 need NO point sum in this kernel (FiLM-scale identity,
      // see the header comment).
      if constexpr (L0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float xs = r == 0 ? 0.f : x0[r - 1], gs_ = r == 0 ? 0.f : G0[r - 1];
          float row[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) row[i] = r == 0 ? act[16 * t + i] : fmaf(act[16 * t + i], xs, gb[16 * t + i] * gs_);
          rs.add(r == 0 ? 1 : r + 2, t, row);
        }
      }
    }
    if (l >= 1) store_flm(fr, (l - 1) & 1);  // FiLM slot of layer l + 1: free since this layer's barrier
    [[maybe_unused]] float mxq_v = 0.f, mxq_u = 0.f;  // lane maxima of the parked vectors (XQ: taken at the stores, reused below)
#if OI_BWD_STORES_LAST
    // v_l / ubar_l leave AFTER the next layer's phi / vbar have been requested: the vector memory pipe is one in-order
    // queue per wave, and a load issued behind two 1 KiB stores waited for their data to drain first (phase profile: the
    // reloads cost 100k of 470k ticks per tile, 67k of them gone when the stores are removed)
    if constexpr (!L0 && !(OI_BWD_ABL & 1)) {
      if constexpr (XQ) {
        auto store_q = [&](int slot, const float (&v)[64], float& mx) {
          mx = 0.f;
#pragma unroll
          for (int k = 0; k < 64; k += 2) mx = fmaxf(mx, fmaxf(fabsf(v[k]), fabsf(v[k + 1])));
          float sq, inv_sq;
          q24_scale(mx, sq, inv_sq);
          const int l12 = 12 * lane;
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            const u32x3 q = pack_q24(f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]}, sq);
            __builtin_amdgcn_raw_buffer_store_b96(q, ws.rs, l12 + (slot * 16384 + g * 768), 0, OI_BWD_ST_WGRAD);
          }
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, inv_sq), ws.rs,
                                                4 * lane + (slot * 16384 + XQ_SCALE_OFF), 0, OI_BWD_ST_WGRAD);
        };
        store_q(S_V + l - 1, gb, mxq_v);
        store_q(S_U + l - 1, act, mxq_u);
      } else {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        ws.template store<OI_BWD_ST_WGRAD>(S_V + l - 1, g, o.l16, f32x4{gb[4 * g], gb[4 * g + 1], gb[4 * g + 2], gb[4 * g + 3]});
        ws.template store<OI_BWD_ST_WGRAD>(S_U + l - 1, g, o.l16, f32x4{act[4 * g], act[4 * g + 1], act[4 * g + 2], act[4 * g + 3]});
      }
      }
    }
#endif
    BW_T(8);
    if constexpr (!L0) {
      // the point vectors hold v_l / gamma_l and phibar_l (what was parked); the products want v_l and ubar_l = gamma_l phibar_l.
      // The launch-wide maxima the weight-gradient GEMM scales its operands with are those of the PARKED vectors.
      auto times_gamma = [&](float (&v)[64], int om_slot, [[maybe_unused]] float mx_known) {
        float mx = (XQ && !(OI_BWD_ABL & 1)) ? mx_known : 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const f32x4 gm = lds_f4(lds, L_FILM + grp_f0(g) * 4, ol.h16);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if constexpr (SC && !(XQ && !(OI_BWD_ABL & 1))) mx = fmaxf(mx, fabsf(v[4 * g + k]));
            v[4 * g + k] *= gm[k];
            asm volatile("" : "+v"(v[4 * g + k]));  // (pins the group: see the epilogue above)
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (SC) publish_max(op_max, om_slot, mx);  // (at once: nothing scalar stays live across the next product)
      };
      const float inv_t = SC ? hdr[H_WSCALE + 7 + l - 1] : 1.f;
      times_gamma(gb, OM_V + l - 1, mxq_v);
      acc_zero(acc);
      BW_T(9);
      const float f1 = gemm2<PREC, true>(lds, ol, gb, acc, inv_t);   // g_l = W_l^T v_l
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) gb[16 * t + r] = SC ? acc[t][r] * f1 : acc[t][r];
      times_gamma(act, OM_U + l - 1, mxq_u);
      acc_zero(acc);
      const float f2 = gemm2<PREC, true>(lds, ol, act, acc, inv_t);  // abar_l = W_l^T ubar_l
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) act[16 * t + r] = SC ? acc[t][r] * f2 : acc[t][r];
    }
    if constexpr (TOP && !(OI_BWD_ABL & 2)) {  // (the top layer's products run at the register limit: its successor's first
#pragma unroll                                 // groups are requested after them, not under them)
      for (int q = 0; q < CARRY; ++q) {
        phn[q] = ws.load_phase(S_PHI + l - 1, q, o.l16);
        vbn[q] = load_vb(S_VB + l - 1, q);
      }
      load_vb_scale(S_VB + l - 1);
    }
    BW_T(10);
    if (L0 && last) {   // (a persistent workgroup: after its last tile)
      __syncthreads();
      // the rows hold sums of phibar_0 / (v_0 / gamma_0) terms, R1 = sum phibar_0, R_{3+j} = sum (phibar_0 x_j + v_0/gamma_0 G_j):
      //   d b_0 = gamma_0 R1,  d W0[:, j] = gamma_0 R_{3+j},  d beta_0 = R1,  d gamma_0 = sum_j W0[f][j] R_{3+j} + b_0[f] R1
      const float* g0_row = reinterpret_cast<const float*>(lds + L_FILM);  // layer 0's rows sit in FiLM slot 0
      racc_flush_row_scaled(lds, 1, g0_row, d_small + DS_B, 1, tid);
      racc_flush_row_scaled(lds, 3, g0_row, d_small + DS_W0 + 0, 3, tid);
      racc_flush_row_scaled(lds, 4, g0_row, d_small + DS_W0 + 1, 3, tid);
      racc_flush_row_scaled(lds, 5, g0_row, d_small + DS_W0 + 2, 3, tid);
      if (const int t_ = late(tid); t_ < C) {
        const float* racc = reinterpret_cast<const float*>(lds + L_RACC);
        const f32x4 w = *reinterpret_cast<const f32x4*>(lds + L_TABS + H_TAB0 * 4 + t_ * 16);
        const float r1 = racc[1 * C + t_];
        atomicAdd(d_beta + (size_t)e * 9 * C + t_, r1);
        atomicAdd(d_gamma + (size_t)e * 9 * C + t_,
                  fmaf(w[0], racc[3 * C + t_], fmaf(w[1], racc[4 * C + t_], fmaf(w[2], racc[5 * C + t_], hdr[H_BIAS + t_] * r1))));
      }
      if constexpr (OI_BWD_LATE_FLUSH) {
        racc_flush_row(lds, RR_WSIG, d_small + DS_WSIG, 1, tid);
        if (has_col) flush_colour(gamma + ((size_t)e * 9 + 8) * C);
      }
    }
    BW_T(11);
  };
  down_layer(NL_SDF - 1, std::integral_constant<int, 2>{});
  for (int l = NL_SDF - 2; l >= 1; --l) down_layer(l, std::integral_constant<int, 0>{});
  down_layer(0, std::integral_constant<int, 1>{});
#ifdef OI_BWD_PROF
  if (lane == 0) {
    for (int i = 0; i < 12; ++i) atomicAdd(&oi_prof_bwd[i], pacc[i]);
    for (int i = 14; i < 20; ++i) atomicAdd(&oi_prof_bwd[i], pacc[i]);
    atomicAdd(&oi_prof_bwd[12], __builtin_readcyclecounter() - tstart);
    atomicAdd(&oi_prof_bwd[13], 1ull);
    if (wave == 0) {
      const unsigned long long rend = wall_clock64();
      atomicAdd(&oi_prof_bwd[20], rend - rstart);
      atomicMax(&oi_prof_bwd[21], (1ull << 62) - rstart);
      atomicMax(&oi_prof_bwd[22], rend);
      const unsigned wg = tile + blockIdx.y * n_tiles;
      if (wg < 4096) {
        oi_prof_bwd_wg[4 * wg + 0] = rstart;
        oi_prof_bwd_wg[4 * wg + 1] = rend;
        oi_prof_bwd_wg[4 * wg + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
        oi_prof_bwd_wg[4 * wg + 3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
      }
    }
  }
#endif
  }  // tile
}

// Layer 0 is not parked by the sweep (see S_PHI): the weight-gradient kernels form phi_0 / (gamma vbar)_0 of matrix m = 0 from
// the point and dL/dgrad, with the operations of the sweep in the sweep's order -- the same bits the sweep consumed.
//   tab (LDS): [128 features][8] = (w0x, w0y, w0z, G = gamma_0 / 2pi | B2 = G b_0 + beta_0 / 2pi, gamma_0, -, -)
constexpr int L0TAB_FLOATS = C * 8;
__device__ __forceinline__ void l0tab_fill(float* tab, const float* __restrict__ hdr, const float* __restrict__ gamma,
                                           const float* __restrict__ beta, int e, int tid) {
  if (tid < C) {
    const float g = gamma[(size_t)e * 9 * C + tid], b = beta[(size_t)e * 9 * C + tid], bias = hdr[H_BIAS + tid];
    const float gr = g * INV_2PI;  // (film_store: gr * inv_img with inv_img = 1 for layer 0)
    reinterpret_cast<f32x4*>(tab)[2 * tid] = f32x4{hdr[H_TAB0 + 4 * tid], hdr[H_TAB0 + 4 * tid + 1], hdr[H_TAB0 + 4 * tid + 2], gr};
    reinterpret_cast<f32x4*>(tab)[2 * tid + 1] = f32x4{fmaf(gr, bias, b * INV_2PI), g, 0.f, 0.f};
  }
}
// features f0 .. f0+3 of one point: x = (x y z .), G = dL/dgrad (incl. the colour head's term)
template <bool FAST>
__device__ __forceinline__ void l0_phase_vb(const float* tab, f32x4 x, f32x4 G, int f0, f32x4& ph, f32x4& vb) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f32x4 a = reinterpret_cast<const f32x4*>(tab)[2 * (f0 + k)], c = reinterpret_cast<const f32x4*>(tab)[2 * (f0 + k) + 1];
    const float u = fmaf(x[2], a[2], fmaf(x[1], a[1], x[0] * a[0]));
    ph[k] = rev_reduce<FAST, false>(fmaf(a[3], u, c[0]));
    vb[k] = fmaf(G[2], a[2], fmaf(G[1], a[1], G[0] * a[0])) * c[1];
  }
}

// ------------------------------------------------------------------------------------------
// weight-gradient GEMM:  dW_m[o][i] += sum_p X[o][p] Y[i][p]  (K = points), fp32 MFMA, split over
// chunks of wave tiles; one workgroup = one (matrix, chunk), wave w owns output rows 32w..32w+31.
//   matrices m = 0..6  (layer l = m+1):  X1 = v_l, Y1 = gbar_l = vbar_{l-1} gamma_{l-1} cos(phi_{l-1});
//                                        X2 = ubar_l, Y2 = a_l = sin(phi_{l-1})      (both Y from the same parked phi)
//   matrix   m = 7     (colour head):     X  = uvbar,            Y  = a_8 = sin(phi_7)
// Operands are read from the sweep kernel's scratch slots (C-fragment order) and transposed via LDS.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int slot_index(int f, int p) {
  // float index of element (feature f, point p of the wave tile) in the LDS copy of a 16 KiB slot.  The copy is
  // skewed by 8 floats per 32-point block (t, rr, hh): the 32 features a wave reads for one point then fall into 32
  // distinct banks (unskewed, bank = 4p + k for every (rr, hh): an 8-way conflict).
  const int t = f >> 5, rr = (f >> 3) & 3, hh = (f >> 2) & 1, k = f & 3;
  const int blk = (4 * t + rr) * 2 + hh;
  return (blk * 32 + p) * 4 + k + 8 * blk;
}
constexpr int WG_SLOT_FLOATS = 4096 + 8 * 32;

// FiLM / bias gradients of layer `lrow` (1..7, 8 = colour head) of one batch element from a workgroup's partial results
// (everything is linear in the partial sums, so every workgroup adds its share):
//   dwv[t][rg]  partial D[oo][32 t + i], oo = 32 wave + (rg & 3) + 8 (rg >> 2) + 4 h   (MFMA accumulator layout)
//   sub         this lane's partial sum of phibar[fo] over the points its MFMA A-fragments cover (the other lane half holds
//               the other points), fo = 32 wave + i
//   d gamma = sum_i W[f][i] D[f][i] + b[f] s[f];   d beta = s;   db = gamma s;   dW[f][i] = gamma[f] D[f][i];   s = sum_p phibar
// DW(t, rg) returns the workgroup's partial D[oo][32 t + i] (operands parked without gamma: no division anywhere).
template <class DW>
__device__ __forceinline__ void film_identity_epilogue(DW dw, float* __restrict__ dst, float sub,
                                                       const float* __restrict__ wplain, const float* __restrict__ bias_row,
                                                       const float* __restrict__ gamma_row, float* __restrict__ d_gamma_row,
                                                       float* __restrict__ d_beta_row, float* __restrict__ d_bias_row, int tid) {
  const int lane = tid & 63, wave = tid >> 6, h = lane >> 5, i = lane & 31;
  float s[16];
#pragma unroll
  for (int rg = 0; rg < 16; ++rg) {
    const size_t row = (size_t)(32 * wave + (rg & 3) + 8 * (rg >> 2) + 4 * h) * C + i;
    const float grow = gamma_row[32 * wave + (rg & 3) + 8 * (rg >> 2) + 4 * h];
    float a = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float v = dw(t, rg);
      atomicAdd(dst + row + 32 * t, v * grow);
      a = fmaf(v, wplain[row + 32 * t], a);
    }
    s[rg] = a;
    __builtin_amdgcn_sched_barrier(0);  // one row at a time: keeps the 64 weight loads from being hoisted into live registers
  }
  float v = row_transpose_sum16(s, lane);  // lane r of every 16-lane row: the row's sum of value r
  v += __shfl_xor(v, 16, 64);              // both rows of the 32 columns
  const float w = sub + __shfl_xor(sub, 32, 64);
  if ((lane & 16) == 0) {
    const int r = lane & 15;
    const int oo = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
    atomicAdd(d_gamma_row + oo, v);
  }
  if (h == 0) {
    const int f = 32 * wave + i;
    atomicAdd(d_bias_row + f, w * gamma_row[f]);
    atomicAdd(d_beta_row + f, w);
    atomicAdd(d_gamma_row + f, bias_row[f] * w);
  }
}

template <bool FAST>
__global__ void __launch_bounds__(256)
mlp_wgrad_kernel(const char* __restrict__ scratch, const char* __restrict__ packed, size_t plain_offset,
                 const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ d_wmat,
                 float* __restrict__ d_gamma, float* __restrict__ d_beta, float* __restrict__ d_small, long long wt_per_elem,
                 int tiles_per_chunk, int has_col) {
  __shared__ __attribute__((aligned(16))) float sx[WG_SLOT_FLOATS], sy[WG_SLOT_FLOATS];
  __shared__ __attribute__((aligned(16))) float l0tab[L0TAB_FLOATS];
  const int m = blockIdx.y;
  if (m == 7 && !has_col) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, i = lane & 31;
  const int e = blockIdx.z;  // chunks never straddle batch elements: the FiLM gradients are per element
  if (m == 0) l0tab_fill(l0tab, reinterpret_cast<const float*>(packed), gamma, beta, e, tid);  // (first barrier below)
  const long long t_begin = e * wt_per_elem + (long long)blockIdx.x * tiles_per_chunk;
  const long long t_end = min((e + 1) * wt_per_elem, t_begin + tiles_per_chunk);
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float sub = 0.f;  // sum over this lane's points of ubar_l[fo] (uvbar for the colour head): d b_l
  const int npair = (m == 7) ? 1 : 2;
  for (long long wt = t_begin; wt < t_end; ++wt) {
    const char* base = scratch + wt * (long long)(NSLOT_BWD * 16384);
    for (int pr = 0; pr < npair; ++pr) {
      int sxi, syi;
      bool y_is_gbar;  // pair 0 of a layer matrix: Y = gbar_l, re-formed from vbar_{l-1} and phi_{l-1}
      if (m == 7) { sxi = S_UV; syi = S_PHI + 7; y_is_gbar = false; }
      else if (pr == 0) { sxi = S_V + m; syi = S_PHI + m; y_is_gbar = true; }
      else { sxi = S_U + m; syi = S_PHI + m; y_is_gbar = false; }  // a_l = sin(phi_{l-1}), l = m+1
      __syncthreads();
      const f32x4* gx4 = reinterpret_cast<const f32x4*>(base + (size_t)sxi * 16384);
      const f32x4* gy4 = reinterpret_cast<const f32x4*>(base + (size_t)syi * 16384);
      const f32x4* gv4 = reinterpret_cast<const f32x4*>(base + (size_t)(S_VB + (m < 7 ? m : 0)) * 16384);
      const bool x_is_ubar = m == 7 || pr == 1;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        // f32x4 number q = it*256 + tid of the slot: 32-point block q >> 5 = ((4t + rr) * 2 + hh)
        const int q = it * 256 + tid;
        const int dq = q + 2 * (q >> 5);  // + 8 floats per 32-point block
        reinterpret_cast<f32x4*>(sx)[dq] = gx4[q];
        f32x4 y, vb = {0.f, 0.f, 0.f, 0.f};
        if (m == 0) {  // slot S_PHI holds the points and dL/dgrad instead (lane j = point j)
          const f32x4* pd = reinterpret_cast<const f32x4*>(base + (size_t)S_PHI * 16384) + 2 * (tid & 31);
          l0_phase_vb<FAST>(l0tab, pd[0], pd[1], grp_f0(4 * it + wave) + 4 * h, y, vb);
        } else {
          y = gy4[q];
          if (y_is_gbar) vb = gv4[q];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float s, c;
          sincos_rev(y[k], s, c);  // parked reduced phase (revolutions)
          y[k] = y_is_gbar ? vb[k] * c : s;  // gamma_{l-1} is folded into the parked vbar
        }
        reinterpret_cast<f32x4*>(sy)[dq] = y;
      }
      __syncthreads();
      const int fo = 32 * wave + i;
#pragma unroll 4
      for (int s = 0; s < 16; ++s) {
        const int p = 2 * s + h;
        const float a = sx[slot_index(fo, p)];
        if (x_is_ubar) sub += a;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float b = sy[slot_index(32 * t + i, p)];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        }
      }
    }
  }
  // D[o][i]: column = lane & 31 (input feature i within tile t), row = (reg&3) + 8*(reg>>2) + 4h (o within the wave's strip)
  const int lrow = m + 1;
  film_identity_epilogue([&](int t, int rg) { return acc[t][rg]; }, d_wmat + (size_t)m * C * C, sub, reinterpret_cast<const float*>(packed + plain_offset) + (size_t)m * C * C,
                         reinterpret_cast<const float*>(packed) + H_BIAS + lrow * C, gamma + ((size_t)e * 9 + lrow) * C,
                         d_gamma + ((size_t)e * 9 + lrow) * C, d_beta + ((size_t)e * 9 + lrow) * C,
                         d_small + DS_B + lrow * C, tid);
}

// ---- F16X3 variant of the weight-gradient GEMM (3 fp16 MFMAs per product instead of 8 fp32-MFMA k-steps per 16
// points).  The operands are data with no a-priori range: they are scaled (exactly, by a power of two) with the launch-wide
// maximum the sweep published, before the split.  The fp32 slot copies sit in LDS with a 4-float skew per 32-point block: the 8 consecutive
// points of one feature that a lane needs for its MFMA operand are then conflict-free dword reads.  The B fragments
// (Y, all 128 columns) are the same for the four waves: each wave converts one column tile and shares it through LDS.
__device__ __forceinline__ int slot_index4(int f, int p) {
  const int t = f >> 5, rr = (f >> 3) & 3, hh = (f >> 2) & 1, k = f & 3;
  const int blk = (4 * t + rr) * 2 + hh;
  return (blk * 32 + p) * 4 + k + 4 * blk;
}
constexpr int WG16_SLOT_FLOATS = 4096 + 4 * 128;

__device__ __forceinline__ void pow2_scale_of(float m, float& sc, float& inv) {
  int eb = (__builtin_bit_cast(int, m) >> 23) & 0xff;
  eb = eb < 14 ? 14 : (eb > 254 ? 254 : eb);
  sc = __builtin_bit_cast(float, (267 - eb) << 23);
  inv = __builtin_bit_cast(float, (eb - 13) << 23);
}

// 8 consecutive points of feature f (points p0 .. p0+7) from the skewed fp32 LDS copy -> scaled fp16 hi / lo limbs
__device__ __forceinline__ float frag16(const float* sl, int f, int p0, f16x8& hi, f16x8& lo) {
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = sl[slot_index4(f, p0 + q)];
  split8_pairs(v, hi, lo);
  return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));  // used for the bias gradient (sum over points)
}

// the same 8 points as ONE bf16 fragment (bf16 operand mode: a single MFMA per product, no scale -- bf16 has fp32's range)
__device__ __forceinline__ float frag_bf16(const float* sl, int f, int p0, f16x8& out) {
  float v[8];
  bf16x8 b;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    v[q] = sl[slot_index4(f, p0 + q)];
    b[q] = (__bf16)v[q];
  }
  out = __builtin_bit_cast(f16x8, b);
  return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}

// the scratch operands are read exactly once by this kernel: non-temporal loads (streaming-read ceiling of the box
// 5.8 TB/s ordinary, 6.5-7.0 non-temporal: tools/dbg/hbm_read.hip)
#ifndef OI_WGRAD_NT
#define OI_WGRAD_NT 1
#endif
#ifndef OI_WG_TARGET
#define OI_WG_TARGET 2048  // workgroups of the weight-gradient GEMM (8 matrices x chunks)
#endif
#ifndef OI_WG_ABL
#define OI_WG_ABL 0
#endif
__device__ __forceinline__ f32x4 ld_once(const f32x4* p) { return OI_WGRAD_NT ? __builtin_nontemporal_load(p) : *p; }

// -DOI_WG_PROF: per-phase shader-clock accounting of the weight-gradient GEMM (read with oi_prof_bwd_read, slots 0..7)
#ifdef OI_WG_PROF
#ifndef OI_BWD_PROF
__device__ unsigned long long oi_prof_bwd[24];
#endif
#define WG_T(i)                                                  \
  do {                                                           \
    const unsigned long long t_ = __builtin_readcyclecounter();  \
    wacc[i] += t_ - wprev;                                       \
    wprev = t_;                                                  \
  } while (0)
#else
#define WG_T(i)
#endif
// OI_WG_TR (round 6, default on): the operands cross LDS as fp16 (bf16) limb planes [32 points][128 features] and come back as MFMA
// fragments through gfx950's transposing read.  The slots hold "point on the lane, four features per granule"; an MFMA operand is
// "feature on the lane, eight points per register group" -- a 32 x 128 transposition per operand and wave tile.  Until round 5:
// fp32 copies in LDS, eight ds_read_b32 per fragment, the fp16 split AFTER the read, the column fragments converted by one wave
// each and shared through a second LDS buffer -- 120 LDS instructions and five barriers per wave tile; the phase profile
// (profiles/r6_wgrad_phase_profile.txt) had the matrix cores busy for 13 % of a tile.  Now the split happens on the granule the
// thread holds anyway, the 8-byte piece (point p, features 4 c .. 4 c + 3) is written once per limb, and ds_read_b64_tr_b16
// (each lane of a 16-lane group supplies the address of ONE piece; the group receives the 4 x 16 block transposed: lane t gets
// column t, rows 0..3 -- tools/dbg/tr_probe2.hip) delivers four points of the lane's feature: two reads = one operand limb.  Both
// pairs of a tile are staged at once: 32 ds_write_b64 + 80 transposing reads and TWO barriers per wave tile.
//   piece (p, c) of a plane lives at byte 8 (32 p + (c ^ g(p))),  g(p) = 8 (p & 1) ^ 17 ((p >> 1) & 1) ^ 2 ((p >> 3) & 1):
//   the 32 pieces a half wave reads (4 points x 8 feature quads) fall on 32 distinct 8-byte bank pairs (conflict-free reads), the
//   16 a store group writes on 8 (2-way: 8 LDS cycles against the 6 the store's register transfer takes anyway).
#ifndef OI_WG_TR
#define OI_WG_TR 1
#endif
#ifndef OI_WG_TR_SPREAD
#define OI_WG_TR_SPREAD 1
#endif
constexpr int TR_PLANE = 8192;   // bytes of one limb plane: 32 points x 32 pieces of 8 bytes
constexpr int TR_PLANES = 8;     // X0h X0l Y0h Y0l X1h X1l Y1h Y1l  (bf16 operands: the four hi planes)
typedef __fp16 trh4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ int tr_g(int p) { return ((p & 1) ? 8 : 0) ^ ((p & 2) ? 17 : 0) ^ ((p & 8) ? 2 : 0); }
// one operand limb (8 consecutive points of the lane's feature) = two transposing reads, 4 points apart (1 KiB)
__device__ __forceinline__ f16x8 tr_frag(const char* planes, int addr, int imm) {
  const trh4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trh4*)(planes + addr + imm));
  const trh4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trh4*)(planes + addr + imm + 1024));
  const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
  return __builtin_bit_cast(f16x8, u32x4{ua[0], ua[1], ub[0], ub[1]});
}
typedef f16x8 (*SbPtr)[4][2][64];  // [hi|lo][column tile][k-step][lane]
// COL: the colour-head matrix (m = 7, one pair: X = uvbar, Y = a_8 = sin phi_7); otherwise a layer matrix (two pairs).  A
// compile-time split: with `m` tested at run time hipcc turned the per-element selects of the hot loop into branches.
// MODE 0: a layer matrix m = 1..6;  1 (COL): the colour head;  2 (FIRST): m = 0, whose Y operands come from layer 0 -- not
// parked, formed here from the point and dL/dgrad (l0tab)
// BF (round 5): the bf16 operand mode's GEMM -- operands rounded to bf16 as they are staged, one v_mfma_f32_32x32x16_bf16 per
// product, no operand scales.  (Until round 5 that mode fell through to the generic fp32-MFMA GEMM: 3.2 ms per backward against
// 1.45 ms for this body, the largest kernel of a bf16-mode training iteration.)
template <int MODE, bool FAST, bool BF = false>
__device__ __forceinline__ void wgrad_f16_body(float* sx, float* sy, SbPtr sb, char* planes, const float* l0tab, const int m, const char* __restrict__ scratch,
                                               const float* __restrict__ op_max, const char* __restrict__ packed,
                                               size_t plain_offset, const float* __restrict__ gamma,
                                               float* __restrict__ d_wmat, float* __restrict__ d_gamma,
                                               float* __restrict__ d_beta, float* __restrict__ d_small,
                                               long long wt_per_elem, int tiles_per_chunk) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, i = lane & 31;
  const int e = blockIdx.z;  // chunks never straddle batch elements: the FiLM gradients are per element
  const long long t_begin = e * wt_per_elem + (long long)blockIdx.x * tiles_per_chunk;
  const long long t_end = min((e + 1) * wt_per_elem, t_begin + tiles_per_chunk);
  constexpr bool COL = MODE == 1, FIRST = MODE == 2;
  constexpr int npair = COL ? 1 : 2;
  // ONE power-of-two scale per operand for the whole launch (maxima published by the sweep): the products of all tiles
  // then share a scale and accumulate straight in the MFMA accumulators -- no per-tile maximum, no per-tile merge.
  // pair 0: X = v_l (uvbar for the colour head), Y = gbar_l (a_8 = sin, |.| <= 1);  pair 1: X = ubar_l, Y = a_l = sin
  // ONE accumulator for both pairs (round 3: 64 instead of 128 accumulator registers -- the registers that hold the SECOND
  // tile in flight): the pair whose products are larger gets the optimal operand scales
  // (max |X|, max |Y| -> [2^13, 2^14)), the other pair's Y scale is lowered so that both products carry the same factor.
  // Nothing can overflow, and the smaller pair is resolved to 2^-22 of the larger one's terms -- the sum they form.
  float scx[2] = {1.f, 1.f}, scy[2] = {1.f, 1.f}, inv[2] = {1.f, 1.f}, inv_scx[2] = {1.f, 1.f};
  if constexpr (!BF) {
    // maximum over the replicas: lane r of every wave reads replica r (the table is 8 KiB and L2-resident)
    const float* rep = op_max + lane * OM_STRIDE;
    const float mx0 = oi::wave_max(rep[COL ? OM_UV : OM_V + m]);
    const float my0 = COL ? 1.0f : oi::wave_max(rep[OM_G + m]);
    const float mx1 = oi::wave_max(rep[OM_U + (COL ? 0 : m)]);
    float ivx, ivy;
    pow2_scale_of(mx0, scx[0], ivx);
    inv_scx[0] = ivx;
    pow2_scale_of(my0, scy[0], ivy);
    inv[0] = ivx * ivy;
    pow2_scale_of(mx1, scx[1], ivx);
    inv_scx[1] = ivx;
    pow2_scale_of(1.0f, scy[1], ivy);
    inv[1] = ivx * ivy;
    if constexpr (!COL) {  // common factor: the smaller of the two product scales (inv = 1 / (scx scy) is the larger)
      if (inv[1] > inv[0]) {
        scy[0] = scy[0] * (inv[0] / inv[1]);  // powers of two: exact
        inv[0] = inv[1];
      } else {
        scy[1] = scy[1] * (inv[1] / inv[0]);
        inv[1] = inv[0];
      }
    }
    // wave-uniform by construction (wave_max): keep them in SGPRs, the VGPR file holds two tiles of staging
    auto uni = [](float& v) { v = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    uni(scx[0]); uni(scx[1]); uni(scy[0]); uni(scy[1]); uni(inv[0]); uni(inv[1]); uni(inv_scx[0]); uni(inv_scx[1]);
  }
  f32x16 acc[4];
  acc_zero(acc);
  float sub = 0.f;  // sum over this lane's points of (scaled) ubar_l[fo] (uvbar for the colour head): d b_l
  const int fo = 32 * wave + i;
  // Software pipeline over the wave tiles: the five slots of tile wt + 1 are requested as soon as tile wt's values have
  // left the staging registers for LDS, and travel under both pairs' fragment extraction and MFMAs (round 2 issued them at
  // the top of the tile and waited: a full memory round trip exposed per tile and workgroup, 4.4 TB/s).
  //   both Y operands of a layer matrix come from the SAME parked phase (reduced, in revolutions): one read, one sin / cos
  //   pair 0: Y = gbar_l = (gamma vbar)_{l-1} cos(phi_{l-1})      pair 1: Y = a_l = sin(phi_{l-1})
  // what the sweep parked (see WaveScratchT): 24-bit values -> 60 staging registers, 16-bit (bf16 mode) -> 40
  constexpr int PK = BF ? (OI_BWD_PACK16 ? 2 : 0) : ((OI_BWD_PACK24 && !FAST) ? 1 : 0);
  constexpr bool PHQ = OI_BWD_PHQ24 && PK == 0 && !BF && !FAST;   // (the sweep's condition)
  using WS = WaveScratchT<PK, PHQ>;
  using Frag = typename WS::Frag;
  using PFrag = typename WS::PFrag;
  // X slots of the layer matrices (S_V, S_U) as 24-bit fixed point + a per-lane scale (see OI_BWD_XQ24; the sweep's condition)
  constexpr bool XQ = OI_BWD_XQ24 && PK == 0 && !BF && !COL && OI_BWD_STORES_LAST;
  using XFrag = std::conditional_t<XQ, u32x3, Frag>;
  constexpr bool VBQ = OI_BWD_VBQ24 && PK == 0 && !BF;   // (the sweep's condition)
  using VFrag = std::conditional_t<VBQ, u32x3, Frag>;
  struct Stage {  // the five slots of one wave tile as they arrive: 80 registers (48 for the colour head); 60 (36) packed
    XFrag xall[2][4];
    PFrag ph4[4];
    VFrag vb4[4];
    float vbinv;    // VBQ: 1 / s of this thread's lane of the gamma vbar slot
    f32x4 pt[2];  // FIRST: the point and dL/dgrad of this thread's lane (raw fp32)
    float xinv[2];  // XQ: 1 / s of this thread's lane of the two X slots
  };
  Stage stA, stB;
  // through a buffer descriptor over the wave tile (512 KiB): the tile base and the slot offsets travel in SGPRs, the lane
  // offset in ONE VGPR -- with flat pointers hipcc kept 20 address pairs live across the loop and spilled
  const int t16 = tid * (PK == 1 ? 12 : (PK == 2 ? 8 : 16));
  // granules [it0, it1) of the tile (the per-lane scales and layer 0's point ride with granule 0): the transposing-read build
  // issues a tile's requests in four parts BETWEEN its MFMA groups -- issued in one go after the stores they stalled the wave for
  // 14 % of a tile (the vector-memory queue is in order and 6 bits deep), with the matrix cores idle meanwhile
  auto request_part = [&](long long wt, Stage& st, int it0, int it1) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(scratch) + wt * (long long)(NSLOT_BWD * 16384), 0, NSLOT_BWD * 16384, 0x00020000);
    auto ld = [&](int slot, int it) -> Frag {   // granule q = it 256 + tid of the slot: group q / 64, lane q % 64
      if constexpr (PK == 1) return __builtin_amdgcn_raw_buffer_load_b96(rs, t16, slot * 16384 + it * 3072, OI_WGRAD_NT ? 2 : 0);
      else if constexpr (PK == 2) return __builtin_amdgcn_raw_buffer_load_b64(rs, t16, slot * 16384 + it * 2048, OI_WGRAD_NT ? 2 : 0);
      else return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, t16, slot * 16384 + it * 4096,
                                                                                  OI_WGRAD_NT ? 2 : 0));
    };
    auto ldx = [&](int slot, int it) -> XFrag {
      if constexpr (XQ) return __builtin_amdgcn_raw_buffer_load_b96(rs, tid * 12, slot * 16384 + it * 3072, OI_WGRAD_NT ? 2 : 0);
      else return ld(slot, it);
    };
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if (it < it0 || it >= it1) continue;
      if constexpr (!FIRST) {
        if constexpr (PHQ) st.ph4[it] = __builtin_amdgcn_raw_buffer_load_b96(rs, tid * 12, (S_PHI + m) * 16384 + it * 3072, OI_WGRAD_NT ? 2 : 0);
        else st.ph4[it] = ld(S_PHI + m, it);
      }
      st.xall[0][it] = ldx(COL ? S_UV : S_V + m, it);
      if constexpr (!COL) {
        if constexpr (!FIRST) {   // gamma_{l-1} vbar_{l-1}
          if constexpr (VBQ) st.vb4[it] = __builtin_amdgcn_raw_buffer_load_b96(rs, tid * 12, (S_VB + m) * 16384 + it * 3072, OI_WGRAD_NT ? 2 : 0);
          else st.vb4[it] = ld(S_VB + m, it);
        }
        st.xall[1][it] = ldx(S_U + m, it);
      }
    }
    if (it0 != 0) return;
    if constexpr (VBQ && !FIRST && !COL)
      st.vbinv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 4 * (tid & 63), (S_VB + m) * 16384 + XQ_SCALE_OFF,
                                                                                OI_WGRAD_NT ? 2 : 0));
    if constexpr (XQ) {
      st.xinv[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 4 * (tid & 63), (S_V + m) * 16384 + XQ_SCALE_OFF,
                                                                                  OI_WGRAD_NT ? 2 : 0));
      st.xinv[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 4 * (tid & 63), (S_U + m) * 16384 + XQ_SCALE_OFF,
                                                                                  OI_WGRAD_NT ? 2 : 0));
    }
    if constexpr (FIRST) {  // this thread's point (lane j = point j): (x y z .), (Gx Gy Gz .)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        st.pt[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, 32 * (tid & 31), S_PHI * 16384 + 16 * q,
                                                                                  OI_WGRAD_NT ? 2 : 0));
    }
  };
  auto request = [&](long long wt, Stage& st) { request_part(wt, st, 0, 4); };
  // fragments of one pair out of the fp32 LDS copies: this wave's column tile of Y -> shared fp16 fragments sb[pr], its
  // own row strip of X -> registers
  auto extract = [&](int pr, f16x8 (&ah)[2], f16x8 (&al)[2], bool last_pair) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8 bh, bl;
      if constexpr (BF) {
        frag_bf16(sy, fo, 16 * ks + 8 * h, bh);
        sb[0][wave][ks][lane] = bh;
      } else {
        frag16(sy, fo, 16 * ks + 8 * h, bh, bl);
        sb[0][wave][ks][lane] = bh;
        sb[1][wave][ks][lane] = bl;
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float xs;
      if constexpr (BF) xs = frag_bf16(sx, fo, 16 * ks + 8 * h, ah[ks]);
      else xs = frag16(sx, fo, 16 * ks + 8 * h, ah[ks], al[ks]);
      if (last_pair) sub += xs;  // the last pair's X is ubar_l / uvbar
    }
  };
  auto products = [&](const f16x8 (&ah)[2], const f16x8 (&al)[2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if constexpr (BF) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[ks]),
                                                           __builtin_bit_cast(bf16x8, sb[0][t][ks][lane]), acc[t], 0, 0, 0);
        } else {
          const f16x8 bh = sb[0][t][ks][lane], bl = sb[1][t][ks][lane];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh, acc[t], 0, 0, 0);
        }
      }
    }
  };
#ifdef OI_WG_PROF
  unsigned long long wacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long wprev = __builtin_readcyclecounter();
  const unsigned long long wstart = wprev;
#endif
  // X fragment `it` of pair pr, times the launch-wide scale scx[pr]
  auto xval = [&](const Stage& st, int pr, int it) -> f32x4 {
    if constexpr (XQ) {
      const float k = st.xinv[pr] * scx[pr], c = -1.5f * k;   // x scx = (y - 1.5) / s scx
      const f32x4 y = unpack_q24(st.xall[pr][it]);
      return f32x4{fmaf(y[0], k, c), fmaf(y[1], k, c), fmaf(y[2], k, c), fmaf(y[3], k, c)};
    } else {
      return WS::value(st.xall[pr][it]) * scx[pr];
    }
  };
  // one wave tile out of staging set `st`; the set is dead once pair 1 is staged, and the tile TWO ahead is requested into it
  auto tile = [&](long long wt, Stage& st) {
#if OI_WG_ABL & 1  // timing ablation: the loads alone (streaming rate of this access pattern)
    {
      float z = 0.f;
#pragma unroll
      for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          z += (FIRST ? st.pt[it & 1][k] : WS::phase_of(st.ph4[it])[k]) + xval(st, 0, it)[k] +
               (COL ? 0.f : (FIRST ? 0.f : (float)st.vb4[it][k & 1]) + xval(st, 1, it)[k]);
      sub += z;
      __builtin_amdgcn_sched_barrier(0);
      if (wt + 2 < t_end) request(wt + 2, st);
      return;
    }
#endif
    // pair 0 needs only the cosine (layer matrices) and pair 1 only the sine: the phase stays in its registers until pair 1
    // is staged, nothing else of the tile does
    f32x4 y0[4], ph4[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      f32x4 vb4;
      if constexpr (FIRST) {
        l0_phase_vb<FAST>(l0tab, st.pt[0], st.pt[1], grp_f0(4 * it + wave) + 4 * h, ph4[it], vb4);
      } else {
        ph4[it] = WS::phase_of(st.ph4[it]);
        if constexpr (!COL) {
          if constexpr (VBQ) vb4 = unpack_q24(st.vb4[it]); else vb4 = WS::value(st.vb4[it]);
        }
      }
      // (VBQ: vb4 holds y = vbar s + 1.5; the de-scaling joins the multiply by the launch-wide scale)
      const float kv = (VBQ && !FIRST && !COL) ? st.vbinv * scy[0] : scy[0], cv = (VBQ && !FIRST && !COL) ? -1.5f * kv : 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k)  // colour head (one pair): Y = a_8 = sin
        y0[it][k] = COL ? __builtin_amdgcn_sinf(ph4[it][k]) * scy[0] : fmaf(vb4[k], kv, cv) * __builtin_amdgcn_cosf(ph4[it][k]);
    }
    WG_T(0);
    __syncthreads();  // (1) the previous tile's readers of sx / sy / sb are done
    WG_T(1);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int q = it * 256 + tid;
      const int dq = q + (q >> 5);  // + 4 floats per 32-point block
      reinterpret_cast<f32x4*>(sx)[dq] = xval(st, 0, it);
      reinterpret_cast<f32x4*>(sy)[dq] = y0[it];
    }
    WG_T(2);
    __syncthreads();  // (2)
    f16x8 ah[2], al[2];
    extract(0, ah, al, npair == 1);
    WG_T(3);
    __syncthreads();  // (3) sx / sy are free again, sb is complete
    if constexpr (npair == 2) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int q = it * 256 + tid;
        const int dq = q + (q >> 5);
        reinterpret_cast<f32x4*>(sx)[dq] = xval(st, 1, it);
        f32x4 y1;
#pragma unroll
        for (int k = 0; k < 4; ++k) y1[k] = __builtin_amdgcn_sinf(ph4[it][k]) * scy[1];
        reinterpret_cast<f32x4*>(sy)[dq] = y1;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (wt + 2 < t_end) request(wt + 2, st);  // every register of the set is dead: two tiles are in flight from here on
    __builtin_amdgcn_sched_barrier(0);
    WG_T(4);
    products(ah, al);
    WG_T(5);
    if constexpr (npair == 2) {
      __syncthreads();  // (4)
      extract(1, ah, al, true);
      WG_T(6);
      __syncthreads();  // (5)
      products(ah, al);
      WG_T(7);
    }
  };
  // ---- OI_WG_TR: fp16 limb planes + transposing reads (see the flag) ----
  int wr[4], rdy[4], rdx = 0;
  if constexpr (OI_WG_TR) {
    const int p = lane & 31, gp = tr_g(p);
#pragma unroll
    for (int it = 0; it < 4; ++it) wr[it] = 8 * (32 * p + ((8 * it + 2 * wave + h) ^ gp));   // granule `it`: features 32 it + 8 wave + 4 h ..
    const int t = lane & 15, cb = (lane >> 4) & 1, hp = lane >> 5, r = t >> 2, tl = t & 3;
    const int gr = tr_g(8 * hp + r);   // (g reads bits 0, 1, 3 of the point: the k-step and half bits 4, 2 stay immediates)
#pragma unroll
    for (int T = 0; T < 4; ++T) rdy[T] = 8 * (32 * (8 * hp + r) + ((8 * T + 4 * cb + tl) ^ gr));
    rdx = wave == 0 ? rdy[0] : (wave == 1 ? rdy[1] : (wave == 2 ? rdy[2] : rdy[3]));
    asm volatile("" : "+v"(wr[0]), "+v"(wr[1]), "+v"(wr[2]), "+v"(wr[3]), "+v"(rdy[0]), "+v"(rdy[1]), "+v"(rdy[2]), "+v"(rdy[3]), "+v"(rdx));
  }
  auto put4 = [&](int plane, int it, const f32x4& v) {   // four features of this thread's point -> the hi (and lo) plane
    if constexpr (BF) {
      const bf16x2 a = {(__bf16)v[0], (__bf16)v[1]}, b = {(__bf16)v[2], (__bf16)v[3]};
      *reinterpret_cast<u32x2*>(planes + plane * TR_PLANE + wr[it]) = u32x2{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
    } else {
      unsigned h0, l0, h1, l1;
      split_pair(v[0], v[1], h0, l0);
      split_pair(v[2], v[3], h1, l1);
      *reinterpret_cast<u32x2*>(planes + plane * TR_PLANE + wr[it]) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(planes + (plane + 1) * TR_PLANE + wr[it]) = u32x2{l0, l1};
    }
  };
  auto tile_tr = [&](long long wt, Stage& st) {
    __syncthreads();  // (A) the previous tile's fragment reads are done
    WG_T(0);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      f32x4 ph, vb4 = {0.f, 0.f, 0.f, 0.f};
      if constexpr (FIRST) {
        l0_phase_vb<FAST>(l0tab, st.pt[0], st.pt[1], grp_f0(4 * it + wave) + 4 * h, ph, vb4);
      } else {
        ph = WS::phase_of(st.ph4[it]);
        if constexpr (!COL) {
          if constexpr (VBQ) vb4 = unpack_q24(st.vb4[it]); else vb4 = WS::value(st.vb4[it]);
        }
      }
      const float kv = (VBQ && !FIRST && !COL) ? st.vbinv * scy[0] : scy[0], cv = (VBQ && !FIRST && !COL) ? -1.5f * kv : 0.f;
      f32x4 y0, y1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        y0[k] = COL ? __builtin_amdgcn_sinf(ph[k]) * scy[0] : fmaf(vb4[k], kv, cv) * __builtin_amdgcn_cosf(ph[k]);
        if constexpr (npair == 2) y1[k] = __builtin_amdgcn_sinf(ph[k]) * scy[1];
      }
      put4(0, it, xval(st, 0, it));
      put4(2, it, y0);
      if constexpr (npair == 2) {
        put4(4, it, xval(st, 1, it));
        put4(6, it, y1);
      }
    }
    WG_T(1);
    __syncthreads();  // (B) the planes are complete
    WG_T(2);
    // every register of the set is dead: the tile two ahead is requested from here on, a part per MFMA group (OI_WG_TR_SPREAD=0:
    // all at once, in front of the groups)
    const bool more = wt + 2 < t_end;
    if (!OI_WG_TR_SPREAD) {
      __builtin_amdgcn_sched_barrier(0);
      if (more) request(wt + 2, st);
      __builtin_amdgcn_sched_barrier(0);
    }
    WG_T(3);
#pragma unroll
    for (int pr = 0; pr < npair; ++pr) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (OI_WG_TR_SPREAD) {
          const int part = pr * 2 + ks, nparts = 2 * npair;   // 4 granules over the 2 or 4 groups
          __builtin_amdgcn_sched_barrier(0);
          if (more) request_part(wt + 2, st, part * 4 / nparts, (part + 1) * 4 / nparts);
          __builtin_amdgcn_sched_barrier(0);
        }
        const int base = pr * 4 * TR_PLANE + ks * 4096;
        const f16x8 ah = tr_frag(planes, rdx, base);
        f16x8 al;
        if constexpr (!BF) al = tr_frag(planes, rdx, base + TR_PLANE);
        if (pr == npair - 1) {   // the last pair's X is phibar_l (uvbar): its sum over the points is the bias gradient
          const u32x4 uh = __builtin_bit_cast(u32x4, ah);
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const unsigned eh = uh[d];   // (element copies first: hipcc 7.2 reads element 0 for every d when a vector-element
            if constexpr (BF) {          //  lvalue is bit_cast directly -- see pack24f)
              sub = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, eh), bf16x2{(__bf16)1.0f, (__bf16)1.0f}, sub, false);
            } else {
              const u32x4 ul = __builtin_bit_cast(u32x4, al);
              const unsigned el = ul[d];
              sub = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, eh), f16x2{(_Float16)1.0f, (_Float16)1.0f}, sub, false);
              sub = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, el), f16x2{(_Float16)1.0f, (_Float16)1.0f}, sub, false);
            }
          }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f16x8 bh = tr_frag(planes, rdy[t], base + 2 * TR_PLANE);
          if constexpr (BF) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh), acc[t], 0, 0, 0);
          } else {
            const f16x8 bl = tr_frag(planes, rdy[t], base + 3 * TR_PLANE);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
          }
        }
      }
    }
    WG_T(4);
  };
  if (t_begin < t_end) request(t_begin, stA);
  if (t_begin + 1 < t_end) request(t_begin + 1, stB);
  for (long long wt = t_begin; wt < t_end; wt += 2) {
    if constexpr (OI_WG_TR) {
      tile_tr(wt, stA);
      if (wt + 1 < t_end) tile_tr(wt + 1, stB);
    } else {
      tile(wt, stA);
      if (wt + 1 < t_end) tile(wt + 1, stB);
    }
  }
#ifdef OI_WG_PROF
  if (lane == 0 && !COL) {
    for (int q = 0; q < 8; ++q) atomicAdd(&oi_prof_bwd[q], wacc[q]);
    atomicAdd(&oi_prof_bwd[12], __builtin_readcyclecounter() - wstart);
    atomicAdd(&oi_prof_bwd[13], (unsigned long long)(t_end - t_begin));
  }
#endif
  const int lrow = m + 1;
  film_identity_epilogue(
      [&](int t, int rg) { return acc[t][rg] * inv[0]; },
      d_wmat + (size_t)m * C * C, sub * (npair == 2 ? inv_scx[1] : inv_scx[0]), reinterpret_cast<const float*>(packed + plain_offset) + (size_t)m * C * C,
                         reinterpret_cast<const float*>(packed) + H_BIAS + lrow * C, gamma + ((size_t)e * 9 + lrow) * C,
                         d_gamma + ((size_t)e * 9 + lrow) * C, d_beta + ((size_t)e * 9 + lrow) * C,
                         d_small + DS_B + lrow * C, tid);
}

template <bool FAST, bool BF = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))  // VGPRs + AGPRs <= 256: two workgroups per CU
mlp_wgrad_f16_kernel(const char* __restrict__ scratch, const float* __restrict__ op_max, const char* __restrict__ packed,
                     size_t plain_offset, const float* __restrict__ gamma, const float* __restrict__ beta,
                     float* __restrict__ d_wmat, float* __restrict__ d_gamma, float* __restrict__ d_beta,
                     float* __restrict__ d_small, long long wt_per_elem, int tiles_per_chunk, int has_col) {
#if OI_WG_TR
  __shared__ __attribute__((aligned(16))) char planes[TR_PLANES * TR_PLANE];   // 64 KiB: two workgroups per CU
  __shared__ __attribute__((aligned(16))) float l0tab[L0TAB_FLOATS];
  float *sx = nullptr, *sy = nullptr;
  SbPtr sb = nullptr;
#else
  __shared__ __attribute__((aligned(16))) float sx[WG16_SLOT_FLOATS], sy[WG16_SLOT_FLOATS];
  __shared__ __attribute__((aligned(16))) float l0tab[L0TAB_FLOATS];
  // ONE copy for both pairs (barrier 4 separates pair 0's readers from pair 1's writers)
  __shared__ __attribute__((aligned(16))) f16x8 sb[2][4][2][64];
  char* planes = nullptr;
#endif
  const int m = blockIdx.y;
  // a compile-time split per kind of matrix: with `m` tested at run time hipcc turned per-element selects into branches
  if (m == 7) {
    if (has_col)
      wgrad_f16_body<1, FAST, BF>(sx, sy, sb, planes, l0tab, 7, scratch, op_max, packed, plain_offset, gamma, d_wmat, d_gamma, d_beta,
                              d_small, wt_per_elem, tiles_per_chunk);
  } else if (m == 0) {
    l0tab_fill(l0tab, reinterpret_cast<const float*>(packed), gamma, beta, blockIdx.z, threadIdx.x);
    __syncthreads();
    wgrad_f16_body<2, FAST, BF>(sx, sy, sb, planes, l0tab, 0, scratch, op_max, packed, plain_offset, gamma, d_wmat, d_gamma, d_beta,
                            d_small, wt_per_elem, tiles_per_chunk);
  } else {
    wgrad_f16_body<0, FAST, BF>(sx, sy, sb, planes, l0tab, m, scratch, op_max, packed, plain_offset, gamma, d_wmat, d_gamma, d_beta,
                            d_small, wt_per_elem, tiles_per_chunk);
  }
}

template <int PREC, bool FAST>
int launch_bwd(const float* pts, const void* packed, const float* gamma, const float* beta, const float* grad_fwd,
               const float* rgb_fwd, const float* feat_fwd, const float* g_sdf, const float* g_grad, const float* g_rgb,
               const float* g_feat, float* d_small,
               float* d_wmat, float* d_gamma, float* d_beta, void* scratch, size_t scratch_bytes, int B, long long n,
               hipStream_t st) {
  // The scratch is a bound, not a function of the problem: the points of every batch element are processed in chunks of
  // as many 128-point tiles as `scratch_bytes` holds; all outputs are accumulated, so chunks simply add up.
  const long long tile_bytes = (long long)BW_NW * NSLOT_BWD * 16384;
  const long long tiles_all = oi::cdiv(n, BW_TILE);
  OI_REQUIRE(scratch_bytes > OM_FLOATS * sizeof(float), "oi_sdf_mlp_bwd: scratch of %zu bytes", scratch_bytes);
  float* op_max = reinterpret_cast<float*>(scratch);             // header: launch-wide operand maxima
  char* tiles = reinterpret_cast<char*>(scratch) + OM_FLOATS * sizeof(float);
  scratch_bytes -= OM_FLOATS * sizeof(float);
  long long tiles_fit = (long long)(scratch_bytes / (size_t)(tile_bytes * B));
  if (tiles_fit < 1) return oi::fail(OI_ERR_INVALID_ARG, "oi_sdf_mlp_bwd: scratch of %zu bytes holds no tile (need >= %lld)",
                                     scratch_bytes, tile_bytes * B);
  if (tiles_fit > tiles_all) tiles_fit = tiles_all;
  tiles_fit = oi::cdiv(tiles_all, oi::cdiv(tiles_all, tiles_fit));  // equal chunks: no short last launch
  const int has_col = (rgb_fwd != nullptr && g_rgb != nullptr && feat_fwd != nullptr) ? 1 : 0;
  auto k = g_feat ? mlp_bwd_sweep_kernel<PREC, FAST, true> : mlp_bwd_sweep_kernel<PREC, FAST, false>;
  // per launch: the attribute is per device, and a process may drive several
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, L_TOTAL_BWD);
  for (long long t0 = 0; t0 < tiles_all; t0 += tiles_fit) {
    const long long nt = std::min(tiles_fit, tiles_all - t0);
    const long long off = t0 * BW_TILE, cn = std::min<long long>(nt * BW_TILE, n - off);
#if OI_BWD_PERSIST
    static const int cus = [] {
      int dev = 0, n_ = 256;
      (void)hipGetDevice(&dev);
      (void)hipDeviceGetAttribute(&n_, hipDeviceAttributeMultiprocessorCount, dev);
      return n_ > 0 ? n_ : 256;
    }();
    dim3 grid((unsigned)std::min<long long>(nt, std::max(1, cus / B)), B), block(256);
#else
    dim3 grid((unsigned)nt, B), block(256);
#endif
    if constexpr (PREC == OI_PREC_F16X3) {
      hipError_t e = oi::zero_async(op_max, OM_FLOATS, st);
      if (e != hipSuccess) return oi::fail(OI_ERR_LAUNCH, "oi_sdf_mlp_bwd: zero fill: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k, grid, dim3(BW_THREADS), L_TOTAL_BWD, st, pts, reinterpret_cast<const char*>(packed), gamma, beta, grad_fwd,
                       rgb_fwd, feat_fwd, g_sdf, g_grad, g_rgb, g_feat, d_small, d_gamma, d_beta, tiles, op_max, cn, n, off);
    int rc = oi::check_launch("oi_sdf_mlp_bwd(sweep)");
    if (rc != OI_OK) return rc;
    const long long wt_per_elem = nt * BW_NW, n_wt = (long long)B * wt_per_elem;
    // ~2048 workgroups in total; a chunk never straddles two batch elements (per-element FiLM gradients)
    const int chunk = (int)std::min<long long>(wt_per_elem, std::max<long long>(1, (n_wt * 8 + OI_WG_TARGET - 1) / OI_WG_TARGET));
    dim3 g2(oi::cdiv(wt_per_elem, chunk), 8, B);
    const char* pk = reinterpret_cast<const char*>(packed);
    if constexpr (PREC == OI_PREC_F16X3) {
      hipLaunchKernelGGL(mlp_wgrad_f16_kernel<FAST>, g2, block, 0, st, tiles, op_max, pk, plain_off(PREC), gamma, beta,
                         d_wmat, d_gamma, d_beta, d_small, wt_per_elem, chunk, has_col);
    } else if constexpr (PREC == OI_PREC_BF16 && OI_WGRAD_BF16) {
      hipLaunchKernelGGL((mlp_wgrad_f16_kernel<FAST, true>), g2, block, 0, st, tiles, op_max, pk, plain_off(PREC), gamma, beta,
                         d_wmat, d_gamma, d_beta, d_small, wt_per_elem, chunk, has_col);
    } else {
      hipLaunchKernelGGL(mlp_wgrad_kernel<FAST>, g2, block, 0, st, tiles, pk, plain_off(PREC), gamma, beta, d_wmat, d_gamma,
                         d_beta, d_small, wt_per_elem, chunk, has_col);
    }
    rc = oi::check_launch("oi_sdf_mlp_bwd(wgrad)");
    if (rc != OI_OK) return rc;
  }
  return OI_OK;
}

}  // namespace

#ifdef OI_BWD_PROF
extern "C" int oi_prof_bwd_read_wg(unsigned long long* out) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(oi_prof_bwd_wg), sizeof(unsigned long long) * 4 * 4096);
  return 0;
}
#endif
#ifdef OI_WG_PROF
extern "C" int oi_dbg_occupancy(int which) {
  int n = -1;
  if (which == 0) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, mlp_wgrad_f16_kernel<false>, 256, 0);
  if (which == 1) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, mlp_wgrad_kernel<false>, 256, 0);
  return n;
}
#endif
#if defined(OI_BWD_PROF) || defined(OI_WG_PROF)
extern "C" int oi_prof_bwd_read(unsigned long long* out, int reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(oi_prof_bwd), sizeof(unsigned long long) * 24);
  if (reset) {
    unsigned long long z[24] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(oi_prof_bwd), z, sizeof(z));
  }
  return 0;
}
#endif

extern "C" {

int oi_selftest_q24(const float* x, float* y, long long n, int mode, oi_stream_t stream) {
  OI_REQUIRE(x && y && n > 0 && n % 64 == 0 && (mode == 0 || mode == 1), "oi_selftest_q24: bad argument");
  const long long nvec = n / 64;
  hipLaunchKernelGGL(selftest_q24_kernel, dim3(oi::cdiv(nvec, 256)), dim3(256), 0, oi::as_stream(stream), x, y, nvec, mode);
  return oi::check_launch("oi_selftest_q24");
}


size_t oi_mlp_bwd_scratch_bytes(int B, long long n_per_elem) {
  const long long tiles = (n_per_elem + BW_TILE - 1) / BW_TILE;
  return (size_t)B * tiles * BW_NW * NSLOT_BWD * 16384 + OM_FLOATS * sizeof(float);
}

size_t oi_mlp_bwd_scratch_bytes_capped(int B, long long n_per_elem, size_t cap_bytes) {
  const size_t per_tile = (size_t)B * BW_NW * NSLOT_BWD * 16384, head = OM_FLOATS * sizeof(float);
  const size_t full = oi_mlp_bwd_scratch_bytes(B, n_per_elem);
  if (full <= cap_bytes) return full;
  const size_t tiles = cap_bytes > head ? (cap_bytes - head) / per_tile : 0;
  return (tiles < 1 ? 1 : tiles) * per_tile + head;
}

int oi_mlp_bwd_small_floats(void) { return DS_TOTAL; }

int oi_sdf_mlp_bwd_feat(const float* pts, const void* packed, const float* gamma, const float* beta, const float* grad_fwd,
                        const float* rgb_fwd, const float* feat_fwd, const float* g_sdf, const float* g_grad, const float* g_rgb,
                        const float* g_feat, float* d_small,
                   float* d_wmat, float* d_gamma, float* d_beta, void* scratch, size_t scratch_bytes, int B,
                   long long n_per_elem, int prec, int fast_trig, oi_stream_t stream) {
  OI_REQUIRE(pts && packed && gamma && beta && d_small && d_wmat && d_gamma && d_beta && scratch,
             "oi_sdf_mlp_bwd: null pointer");
  OI_REQUIRE(B > 0 && n_per_elem > 0, "oi_sdf_mlp_bwd: B=%d n=%lld", B, n_per_elem);
  OI_REQUIRE((rgb_fwd == nullptr) == (g_rgb == nullptr) || g_rgb == nullptr, "oi_sdf_mlp_bwd: g_rgb needs rgb_fwd");
  OI_REQUIRE(g_rgb == nullptr || (grad_fwd != nullptr && feat_fwd != nullptr),
             "oi_sdf_mlp_bwd: colour backward needs the forward gradient and the forward features");
  OI_REQUIRE(prec != OI_PREC_BF16X6, "oi_sdf_mlp_bwd: pass the OI_PREC_F32 image for the backward of the BF16X6 mode");
  OI_REQUIRE(prec != OI_PREC_BF16X3, "oi_sdf_mlp_bwd: pass the OI_PREC_F16X3 image for the backward of the BF16X3 mode");
  hipStream_t st = oi::as_stream(stream);
#define OI_BWD_CASE(P)                                                                                              \
  case P:                                                                                                           \
    return fast_trig ? launch_bwd<P, true>(pts, packed, gamma, beta, grad_fwd, rgb_fwd, feat_fwd, g_sdf, g_grad, g_rgb, g_feat, d_small, \
                                           d_wmat, d_gamma, d_beta, scratch, scratch_bytes, B, n_per_elem, st)      \
                     : launch_bwd<P, false>(pts, packed, gamma, beta, grad_fwd, rgb_fwd, feat_fwd, g_sdf, g_grad, g_rgb, g_feat, d_small, \
                                            d_wmat, d_gamma, d_beta, scratch, scratch_bytes, B, n_per_elem, st);
  switch (prec) {
    OI_BWD_CASE(OI_PREC_F32)
    OI_BWD_CASE(OI_PREC_BF16)
    OI_BWD_CASE(OI_PREC_F16X3)
    default:
      return oi::fail(OI_ERR_INVALID_ARG, "oi_sdf_mlp_bwd: bad precision %d", prec);
  }
#undef OI_BWD_CASE
}

int oi_sdf_mlp_bwd(const float* pts, const void* packed, const float* gamma, const float* beta, const float* grad_fwd,
                   const float* rgb_fwd, const float* feat_fwd, const float* g_sdf, const float* g_grad, const float* g_rgb, float* d_small,
                   float* d_wmat, float* d_gamma, float* d_beta, void* scratch, size_t scratch_bytes, int B,
                   long long n_per_elem, int prec, int fast_trig, oi_stream_t stream) {
  return oi_sdf_mlp_bwd_feat(pts, packed, gamma, beta, grad_fwd, rgb_fwd, feat_fwd, g_sdf, g_grad, g_rgb, nullptr, d_small, d_wmat, d_gamma,
                             d_beta, scratch, scratch_bytes, B, n_per_elem, prec, fast_trig, stream);
}

}  // extern "C"
