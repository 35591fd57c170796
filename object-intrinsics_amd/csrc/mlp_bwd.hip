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
//                       gamma / beta / bias gradients; parks v_l, ubar_l for the weight-gradient GEMM       [14 GEMMs]
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
constexpr int S_PHI = 0;    // 8: phi_l
constexpr int S_VB = 8;     // 8: vbar_l = W_l gbar_l           (down sweep: cbar_l = vbar_l g_{l+1}; wgrad: gbar_{l+1} = vbar_l c_l)
constexpr int S_V = 16;     // 7: v_l,    l = 1..7   (wgrad operand)
constexpr int S_U = 23;     // 7: ubar_l, l = 1..7   (wgrad operand)
constexpr int S_UV = 30;    // 1: uvbar (colour head pre-activation gradient)
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
constexpr int L_RACC = L_WBUF + 2 * 65536;  // per-workgroup reduction scratch: [8][128] floats
constexpr int L_FILM2 = L_RACC + 8 * C * 4;
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
struct WaveScratchB {
  __amdgpu_buffer_rsrc_t rs;
  template <int AUX = OI_BWD_ST_LOCAL>
  __device__ __forceinline__ void store(int slot, int g, int l16, f32x4 v) const {
    oi::buffer_store_b128<AUX>(__builtin_bit_cast(u32x4, v), rs, l16, slot * 16384 + g * 1024);
  }
  template <int AUX = OI_BWD_LD_LAST>
  __device__ __forceinline__ f32x4 load(int slot, int g, int l16) const {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, l16, slot * 16384 + g * 1024, AUX));
  }
};

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
  float* lane_base;  // racc + 8 * ((lane & 15) >> 2) + 4 * h + (lane & 3)
  int lane;
  __device__ __forceinline__ void add(int row, int t, const float (&v)[16]) const {
    atomicAdd(lane_base + row * C + 32 * t, row_transpose_sum16(v, lane));
  }
};

__device__ __forceinline__ void racc_zero(char* lds, int tid) {
  float* racc = reinterpret_cast<float*>(lds + L_RACC);
  for (int i = tid; i < 8 * C; i += 256) racc[i] = 0.f;
}
// flush `rows` accumulator rows: row r goes to dst[r] (a global base pointer per row)
__device__ __forceinline__ void racc_flush_row(char* lds, int row, float* dst, int stride, int tid) {
  const float* racc = reinterpret_cast<const float*>(lds + L_RACC) + row * C;
  if (tid < C) atomicAdd(dst + tid * stride, racc[tid]);
}
// the same row times a per-feature factor (sum_p ubar = gamma * sum_p phibar: the bias gradient needs no sum of its own)
__device__ __forceinline__ void racc_flush_row_scaled(char* lds, int row, const float* factor, float* dst, int tid) {
  const float* racc = reinterpret_cast<const float*>(lds + L_RACC) + row * C;
  if (tid < C) atomicAdd(dst + tid, racc[tid] * factor[tid]);
}

// this wave's LDS-DMA has landed (and so have its outstanding scratch loads), then rendezvous
__device__ __forceinline__ void dma_sync() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

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
constexpr int OM_V = 0;    // 7: max |v_l|,    l = 1..7
constexpr int OM_U = 7;    // 7: max |ubar_l|, l = 1..7
constexpr int OM_G = 14;   // 7: max |gbar_l|, l = 1..7
constexpr int OM_UV = 21;  // 1: max |uvbar|
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

// OI_BWD_WAVES_PER_SIMD = 1: one workgroup per CU with the whole 512-entry register file per wave (two 64-register point
// vectors, the accumulators and a full layer of prefetched phi / vbar fragments are live at once; with 256 registers hipcc
// spilled ~370 of them: same-box A/B 4.86 -> 4.29 ms before the double-buffered staging below)
#ifndef OI_BWD_WAVES_PER_SIMD
#define OI_BWD_WAVES_PER_SIMD 1
#endif
// timing ablations (results are garbage): -DOI_BWD_ABL=1 no v / ubar stores, 2 no phi / vbar reloads, 4 no up-sweep stores
#ifndef OI_BWD_ABL
#define OI_BWD_ABL 0
#endif
#ifndef OI_BWD_EARLY_RELOAD
#define OI_BWD_EARLY_RELOAD 1
#endif
// -DOI_BWD_PROF: per-phase shader-clock accounting of the sweep (tools/dbg/phase_prof_bwd.py)
#ifdef OI_BWD_PROF
__device__ unsigned long long oi_prof_bwd[16];
#define BW_T(i)                                                  \
  do {                                                           \
    const unsigned long long t_ = __builtin_readcyclecounter();  \
    pacc[i] += t_ - tprev;                                       \
    tprev = t_;                                                  \
  } while (0)
#else
#define BW_T(i)
#endif
template <int PREC, bool FAST>
__global__ void __launch_bounds__(256, OI_BWD_WAVES_PER_SIMD)
mlp_bwd_sweep_kernel(const float* __restrict__ pts, const char* __restrict__ packed, const float* __restrict__ gamma,
                     const float* __restrict__ beta, const float* __restrict__ grad_fwd,
                     const float* __restrict__ rgb_fwd, const float* __restrict__ feat_fwd,
                     const float* __restrict__ g_sdf, const float* __restrict__ g_grad,
                     const float* __restrict__ g_rgb, float* __restrict__ d_small, float* __restrict__ d_gamma,
                     float* __restrict__ d_beta, char* __restrict__ scratch, float* __restrict__ op_max,
                     long long n_per_elem, long long n_stride, long long pt_off) {
  // this launch covers points [pt_off, pt_off + n_per_elem) of every batch element; an element holds n_stride points
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, j = lane & 31;
  const int e = blockIdx.y;
  const float* hdr = reinterpret_cast<const float*>(packed);
  const char* mats = packed + H_BYTES;
  const bool has_col = rgb_fwd != nullptr && g_rgb != nullptr && feat_fwd != nullptr;

  LaneOff o;
  o.h16 = 16 * h;
  o.h64 = 64 * h;
  o.l16 = 16 * lane;
  o.l16hi = 16 * lane + 32768;
  asm volatile("" : "+v"(o.h16), "+v"(o.h64), "+v"(o.l16), "+v"(o.l16hi));

  const RowSum rs{reinterpret_cast<float*>(lds + L_RACC) + 8 * ((lane & 15) >> 2) + 4 * h + (lane & 3), lane};
  const long long local = (long long)blockIdx.x * TILE_PTS + wave * WAVE_PTS + j;
  const bool valid = local < n_per_elem;
  const long long pt = (long long)e * n_stride + pt_off + (valid ? local : n_per_elem - 1);
  const float vmask = valid ? 1.f : 0.f;  // tail points contribute nothing
  const __amdgpu_buffer_rsrc_t img_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(mats), 0, NMAT * layer_bytes(PREC), 0x00020000);

  WaveScratchB ws;
  {
    const long long wt = ((long long)e * gridDim.x + blockIdx.x) * 4 + wave;
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
  auto stage_img = [&](int image, int ws_) { stage_layer_rs<PREC>(lds + ws_ * 65536, img_rs, image, wave, o.l16); };
  auto stage_flm = [&](int l_, int fs_) { stage_film(lds + fs_ * (L_FILM2 - L_FILM), gamma, beta, hdr, e, l_, tid); };
  {
    float* tabs = reinterpret_cast<float*>(lds + L_TABS);
    for (int i = tid; i < H_TABS_END; i += 256) tabs[i] = hdr[i];
    stage_flm(0, 0);
    racc_zero(lds, tid);
  }
#ifdef OI_BWD_PROF
  unsigned long long pacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_readcyclecounter();
  const unsigned long long tstart = tprev;
#endif
  const float px = pts[pt * 3 + 0], py = pts[pt * 3 + 1], pz = pts[pt * 3 + 2];
  const float gs = (g_sdf ? g_sdf[pt] : 0.f) * vmask;
  float Gx = (g_grad ? g_grad[pt * 3 + 0] : 0.f) * vmask, Gy = (g_grad ? g_grad[pt * 3 + 1] : 0.f) * vmask,
        Gz = (g_grad ? g_grad[pt * 3 + 2] : 0.f) * vmask;
  __syncthreads();

  float act[64];  // up sweep: a_l;     down sweep: abar_{l+1} -> ubar_l -> abar_l
  float gb[64];   // up sweep: gbar_l -> vbar_l -> gbar_{l+1};   down sweep: g_{l+1} -> v_l -> g_l
  f32x16 acc[4];
  // F16X3: the images carry a power-of-two scale 2^k_m (header H_WSCALE holds 2^-k_m); every product returns the factor
  // its accumulators still need (gemm2), adjoint vectors are normalised per point before the fp16 split.
  constexpr bool SC = PREC == OI_PREC_F16X3;

  // ================= colour head backward (first: its dL/dgrad term is part of gbar_0) =================
  if (has_col) {
    __syncthreads();
    stage_img(14, 0);
    stage_flm(8, 0);
#pragma unroll
    for (int g = 0; g < 16; ++g) {  // a_8, as the forward wrote it (feat output): features grp_f0(g) + 4 h .. + 3
      const f32x4 v = *reinterpret_cast<const f32x4*>(feat_fwd + pt * C + grp_f0(g) + 4 * h);
#pragma unroll
      for (int k = 0; k < 4; ++k) act[4 * g + k] = v[k];
    }
    dma_sync();
    const float fx = grad_fwd[pt * 3 + 0], fy = grad_fwd[pt * 3 + 1], fz = grad_fwd[pt * 3 + 2];
    float rho[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float rv = rgb_fwd[pt * 3 + k];
      rho[k] = g_rgb[pt * 3 + k] * rv * (1.0f - rv) * vmask;  // through the sigmoid
    }
    if constexpr (SC) acc_zero(acc); else init_bias(lds, o, acc);
    const float fV = gemm2<PREC, false>(lds, o, act, acc, SC ? hdr[H_WSCALE + 14] : 1.f);
    // uv -> phiv -> hv; then uvbar.  Reductions: rows 0 gamma_v, 1 beta_v, 2 bv, 3..5 Wrgb, 6..7 dWv[:, 128 + (0, 1)]
    float dGx = 0.f, dGy = 0.f, dGz = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float R[8][16];  // rows: 0 gamma_v, 1 beta_v (x gamma_v = bv), 2 dWv[:, 130], 3..5 Wrgb, 6..7 dWv[:, 128 + (0, 1)]
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int g = 4 * t + rr;
        const f32x4 gm = lds_f4(lds, L_FILM + grp_f0(g) * 4, o.h16);
        const f32x4 bt = lds_f4(lds, L_FILM + (C + grp_f0(g)) * 4, o.h16);
        const f32x4 w0 = lds_f4(lds, L_TABS + (H_RGB + 0 * C + grp_f0(g)) * 4, o.h16);
        const f32x4 w1 = lds_f4(lds, L_TABS + (H_RGB + 1 * C + grp_f0(g)) * 4, o.h16);
        const f32x4 w2 = lds_f4(lds, L_TABS + (H_RGB + 2 * C + grp_f0(g)) * 4, o.h16);
        f32x4 uvb, bsv;
        if constexpr (SC) bsv = lds_f4(lds, L_FILM + (2 * C + grp_f0(g)) * 4, o.h16);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const f32x4 wx = lds_f4(lds, L_TABS + H_TABV * 4 + (grp_f0(g) + k) * 16, o.h64);
          const float ua = SC ? fmaf(acc[t][4 * rr + k], fV, bsv[k]) : acc[t][4 * rr + k];
          const float uv = ua + fmaf(fz, wx[2], fmaf(fy, wx[1], fx * wx[0]));
          const float phiv = fmaf(gm[k], uv, bt[k]);
          float hv, cv;
          sincos_<FAST>(phiv, hv, cv);
          const float hvb = w0[k] * rho[0] + w1[k] * rho[1] + w2[k] * rho[2];
          const float phb = hvb * cv;
          uvb[k] = phb * gm[k];
          R[0][4 * rr + k] = phb * uv;
          R[1][4 * rr + k] = phb;
          R[2][4 * rr + k] = uvb[k] * fz;
          R[3][4 * rr + k] = rho[0] * hv;
          R[4][4 * rr + k] = rho[1] * hv;
          R[5][4 * rr + k] = rho[2] * hv;
          R[6][4 * rr + k] = uvb[k] * fx;
          R[7][4 * rr + k] = uvb[k] * fy;
          dGx = fmaf(uvb[k], wx[0], dGx);
          dGy = fmaf(uvb[k], wx[1], dGy);
          dGz = fmaf(uvb[k], wx[2], dGz);
        }
        ws.store(S_UV, g, o.l16, uvb);
#pragma unroll
        for (int k = 0; k < 4; ++k) act[4 * g + k] = uvb[k];
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) rs.add(r, t, R[r]);
    }
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
    __syncthreads();
    racc_flush_row(lds, 0, d_gamma + ((size_t)e * 9 + 8) * C, 1, tid);
    racc_flush_row(lds, 1, d_beta + ((size_t)e * 9 + 8) * C, 1, tid);
    racc_flush_row_scaled(lds, 1, reinterpret_cast<const float*>(lds + L_FILM), d_small + DS_B + 8 * C, tid);
    racc_flush_row(lds, 2, d_small + DS_WVX + 2, 3, tid);
    racc_flush_row(lds, 3, d_small + DS_WRGB + 0 * C, 1, tid);
    racc_flush_row(lds, 4, d_small + DS_WRGB + 1 * C, 1, tid);
    racc_flush_row(lds, 5, d_small + DS_WRGB + 2 * C, 1, tid);
    racc_flush_row(lds, 6, d_small + DS_WVX + 0, 3, tid);
    racc_flush_row(lds, 7, d_small + DS_WVX + 1, 3, tid);
    __syncthreads();
    racc_zero(lds, tid);
    // abar_8 from the colour head: Wv[:, :128]^T uvbar   (transposed colour image, matrix 15)
    stage_img(15, 0);
    dma_sync();
    acc_zero(acc);
    float mx_uv = 0.f;
    const float fT = gemm2<PREC, true>(lds, o, act, acc, SC ? hdr[H_WSCALE + 15] : 1.f, &mx_uv);
    if constexpr (SC) publish_max(op_max, OM_UV, mx_uv);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = SC ? acc[g >> 2][4 * (g & 3) + k] * fT : acc[g >> 2][4 * (g & 3) + k];
      ws.store(S_AC, g, o.l16, v);
    }
    __syncthreads();
    stage_flm(0, 0);
    __syncthreads();
  }

  BW_T(0);
  // ================= up sweep: recompute phi_l, carry gbar_l =================
  // FiLM rows of layer l live in FiLM slot l & 1, layer l's forward image in image slot (l - 1) & 1: both are requested one
  // layer ahead
  stage_img(0, 0);
  stage_flm(1, 1);
  // layer 0 on the VALU: phi_0; vbar_0 = W0 gbar_0 (gbar_0 = dL/dgrad, a 3-vector); gbar_1 = vbar_0 c_0
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const f32x4 gm = lds_f4(lds, L_FILM + grp_f0(g) * 4, o.h16);
    const f32x4 bt = lds_f4(lds, L_FILM + (C + grp_f0(g)) * 4, o.h16);
    const f32x4 bs = lds_f4(lds, L_FILM + (2 * C + grp_f0(g)) * 4, o.h16);
    f32x4 ph, vb;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 w = lds_f4(lds, L_TABS + H_TAB0 * 4 + (grp_f0(g) + k) * 16, o.h64);
      const float u = fmaf(pz, w[2], fmaf(py, w[1], px * w[0])) + bs[k];
      ph[k] = fmaf(gm[k], u, bt[k]);
      float s, c;
      sincos_<FAST>(ph[k], s, c);
      act[4 * g + k] = s;
      vb[k] = fmaf(Gz, w[2], fmaf(Gy, w[1], Gx * w[0]));
      gb[4 * g + k] = vb[k] * gm[k] * c;
    }
    ws.store(S_PHI + 0, g, o.l16, ph);
    ws.store(S_VB + 0, g, o.l16, vb);
    __builtin_amdgcn_sched_barrier(0);
  }
  BW_T(1);
  for (int l = 1; l < NL_SDF; ++l) {
    dma_sync();  // layer l's image and FiLM rows have landed; every wave is done with layer l - 1
    BW_T(2);
    if (l < NL_SDF - 1) {
      stage_img(l, l & 1);
      stage_flm(l + 1, (l + 1) & 1);
    } else {
      stage_img(13, 1);  // the down sweep starts with the transposed image of layer 7 (FiLM rows 7 are resident)
    }
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
    for (int g = 0; g < 16; ++g) {
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] = SC ? acc[g >> 2][4 * (g & 3) + k] * fA : acc[g >> 2][4 * (g & 3) + k];
        gb[4 * g + k] = v[k];
      }
      if (!(OI_BWD_ABL & 4)) ws.store(S_VB + l, g, o.l16, v);
    }
    // u_l = W_l a_l + b_l -> phi_l, a_{l+1};  gbar_{l+1} = vbar_l gamma_l cos phi_l
    if constexpr (SC) acc_zero(acc); else init_bias(lds, ol, acc);
    BW_T(5);
    const float fB = gemm2<PREC, false>(lds, ol, act, acc, inv_img);
    BW_T(4);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const f32x4 gm = lds_f4(lds, L_FILM + grp_f0(g) * 4, ol.h16);
      const f32x4 bt = lds_f4(lds, L_FILM + (C + grp_f0(g)) * 4, ol.h16);
      f32x4 bs;
      if constexpr (SC) bs = lds_f4(lds, L_FILM + (2 * C + grp_f0(g)) * 4, ol.h16);
      f32x4 ph;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float u = SC ? fmaf(acc[g >> 2][4 * (g & 3) + k], fB, bs[k]) : acc[g >> 2][4 * (g & 3) + k];
        ph[k] = fmaf(gm[k], u, bt[k]);
        float s, c;
        sincos_<FAST>(ph[k], s, c);
        act[4 * g + k] = s;
        gb[4 * g + k] *= gm[k] * c;
      }
      if (!(OI_BWD_ABL & 4)) ws.store(S_PHI + l, g, o.l16, ph);
      __builtin_amdgcn_sched_barrier(0);
    }
    BW_T(5);
  }
  // d w_sigma = sum_p (gbar_8 + gs a_8)  (row 3);  d b_sigma = sum_p gs
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = fmaf(gs, act[16 * t + i], gb[16 * t + i]);
    rs.add(3, t, v);
  }
  {
    float b = (h == 0) ? gs : 0.f;
    b = oi::wave_sum(b);
    if (lane == 0) atomicAdd(d_small + DS_BSIG, b);
  }
  __syncthreads();
  racc_flush_row(lds, 3, d_small + DS_WSIG, 1, tid);

  // ================= down sweep: g_l and abar_l together =================
  // g_8 = w_sigma;  abar_8 = gs * w_sigma (+ the colour head's contribution)
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const f32x4 w = lds_f4(lds, L_TABS + (H_SIG + grp_f0(g)) * 4, o.h16);
    f32x4 ac = {0.f, 0.f, 0.f, 0.f};
    if (has_col) ac = ws.load(S_AC, g, o.l16);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      gb[4 * g + k] = w[k];
      act[4 * g + k] = fmaf(gs, w[k], ac[k]);
    }
  }
  // the transposed image of layer l sits in image slot l & 1, its FiLM rows in FiLM slot l & 1; phi_l / vbar_l of the WHOLE
  // layer are requested one layer ahead (128 registers: the reason this kernel runs one wave per SIMD)
  f32x4 phn[16], vbn[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    phn[g] = ws.load(S_PHI + 7, g, o.l16);
    vbn[g] = ws.load(S_VB + 7, g, o.l16);
  }
  BW_T(6);
  // layer 0 is peeled (its extra d W0 rows and missing products are compile-time): no branch inside the unrolled epilogue
  auto down_layer = [&](int l, auto is_layer0) {
    constexpr bool L0 = decltype(is_layer0)::value;
    dma_sync();  // layer l's transposed image and FiLM rows have landed; the previous layer's row flush is complete
    BW_T(7);
    racc_zero(lds, tid);
    if (l >= 2) stage_img(7 + l - 2, (l - 1) & 1);
    if (l >= 1) stage_flm(l - 1, (l - 1) & 1);
    const LaneOff ol = layer_off(l & 1, l & 1);
    __syncthreads();  // reduction rows zeroed
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float R[2][16], R0[3][16];  // rows 0 d gamma_l, 1 d beta_l (x gamma_l = d b_l); layer 0: rows 3..5 d W0[:, 0..2]
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int g = 4 * t + rr;
        const f32x4 ph = phn[g], vb = vbn[g];
        const f32x4 gm = lds_f4(lds, L_FILM + grp_f0(g) * 4, ol.h16);
        const f32x4 bt = lds_f4(lds, L_FILM + (C + grp_f0(g)) * 4, ol.h16);
        f32x4 ub, vv;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float s, c;
          sincos_<FAST>(ph[k], s, c);
          const float gn = gb[4 * g + k];                              // g_{l+1}
          const float cb = vb[k] * gn;                                 // cbar_l
          vv[k] = gn * gm[k] * c;                                      // v_l
          const float phb = act[4 * g + k] * c - cb * gm[k] * s;       // phibar_l
          const float u = (ph[k] - bt[k]) * __builtin_amdgcn_rcpf(gm[k]);  // u_l
          R[0][4 * rr + k] = fmaf(phb, u, cb * c);                     // d gamma_l
          R[1][4 * rr + k] = phb;                                      // d beta_l
          ub[k] = phb * gm[k];                                         // ubar_l
          gb[4 * g + k] = vv[k];
          act[4 * g + k] = ub[k];
          if constexpr (L0) {  // d W0 = sum_p (ubar_0 x^T + v_0 gbar_0^T)
            R0[0][4 * rr + k] = fmaf(ub[k], px, vv[k] * Gx);
            R0[1][4 * rr + k] = fmaf(ub[k], py, vv[k] * Gy);
            R0[2][4 * rr + k] = fmaf(ub[k], pz, vv[k] * Gz);
          }
        }
        if constexpr (!L0 && !(OI_BWD_ABL & 1)) {
          ws.store<OI_BWD_ST_WGRAD>(S_V + l - 1, g, o.l16, vv);
          ws.store<OI_BWD_ST_WGRAD>(S_U + l - 1, g, o.l16, ub);
        }
#if OI_BWD_EARLY_RELOAD
        // this group's fragments are consumed: request the same group of the NEXT layer into the same registers -- the loads
        // travel under the rest of the epilogue and both products, and there is no separate issue phase
        if constexpr (!L0 && !(OI_BWD_ABL & 2)) {
          phn[g] = ws.load(S_PHI + l - 1, g, o.l16);
          vbn[g] = ws.load(S_VB + l - 1, g, o.l16);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
      rs.add(0, t, R[0]);
      rs.add(1, t, R[1]);
      if constexpr (L0) {
        rs.add(3, t, R0[0]);
        rs.add(4, t, R0[1]);
        rs.add(5, t, R0[2]);
      }
    }
    BW_T(8);
    if constexpr (!OI_BWD_EARLY_RELOAD && !L0 && !(OI_BWD_ABL & 2)) {  // the next layer's fragments travel while this layer's two products run
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        phn[g] = ws.load(S_PHI + l - 1, g, o.l16);
        vbn[g] = ws.load(S_VB + l - 1, g, o.l16);
      }
    }
    if constexpr (!L0) {
      const float inv_t = SC ? hdr[H_WSCALE + 7 + l - 1] : 1.f;
      acc_zero(acc);
      BW_T(9);
      float mx_v = 0.f, mx_u = 0.f;
      const float f1 = gemm2<PREC, true>(lds, ol, gb, acc, inv_t, &mx_v);   // g_l = W_l^T v_l
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) gb[16 * t + r] = SC ? acc[t][r] * f1 : acc[t][r];
      acc_zero(acc);
      const float f2 = gemm2<PREC, true>(lds, ol, act, acc, inv_t, &mx_u);  // abar_l = W_l^T ubar_l
      if constexpr (SC) {
        publish_max(op_max, OM_V + l - 1, mx_v);
        publish_max(op_max, OM_U + l - 1, mx_u);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) act[16 * t + r] = SC ? acc[t][r] * f2 : acc[t][r];
    }
    BW_T(10);
    __syncthreads();
    racc_flush_row(lds, 0, d_gamma + ((size_t)e * 9 + l) * C, 1, tid);
    racc_flush_row(lds, 1, d_beta + ((size_t)e * 9 + l) * C, 1, tid);
    racc_flush_row_scaled(lds, 1, reinterpret_cast<const float*>(lds + L_FILM + (l & 1) * (L_FILM2 - L_FILM)),
                          d_small + DS_B + l * C, tid);
    if constexpr (L0) {
      racc_flush_row(lds, 3, d_small + DS_W0 + 0, 3, tid);
      racc_flush_row(lds, 4, d_small + DS_W0 + 1, 3, tid);
      racc_flush_row(lds, 5, d_small + DS_W0 + 2, 3, tid);
    }
    BW_T(11);
  };
  for (int l = NL_SDF - 1; l >= 1; --l) down_layer(l, std::false_type{});
  down_layer(0, std::true_type{});
#ifdef OI_BWD_PROF
  if (lane == 0) {
    for (int i = 0; i < 12; ++i) atomicAdd(&oi_prof_bwd[i], pacc[i]);
    atomicAdd(&oi_prof_bwd[12], __builtin_readcyclecounter() - tstart);
    atomicAdd(&oi_prof_bwd[13], 1ull);
  }
#endif
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

template <bool FAST>
__global__ void __launch_bounds__(256)
mlp_wgrad_kernel(const char* __restrict__ scratch, const float* __restrict__ gamma, float* __restrict__ d_wmat,
                 long long n_wave_tiles, long long wt_per_elem, int tiles_per_chunk, int has_col) {
  __shared__ __attribute__((aligned(16))) float sx[WG_SLOT_FLOATS], sy[WG_SLOT_FLOATS];
  const int m = blockIdx.y;
  if (m == 7 && !has_col) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, i = lane & 31;
  const long long t_begin = (long long)blockIdx.x * tiles_per_chunk;
  const long long t_end = min(n_wave_tiles, t_begin + tiles_per_chunk);
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
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
      const float* grow = gamma + ((wt / wt_per_elem) * 9 + (m < 7 ? m : 0)) * C;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        // f32x4 number q = it*256 + tid of the slot: 32-point block q >> 5 = ((4t + rr) * 2 + hh)
        const int q = it * 256 + tid;
        const int dq = q + 2 * (q >> 5);  // + 8 floats per 32-point block
        reinterpret_cast<f32x4*>(sx)[dq] = gx4[q];
        f32x4 y = gy4[q];
        f32x4 vb = {0.f, 0.f, 0.f, 0.f}, gm = {0.f, 0.f, 0.f, 0.f};
        if (y_is_gbar) {
          vb = gv4[q];
          gm = *reinterpret_cast<const f32x4*>(grow + grp_f0(q >> 6) + 4 * ((q >> 5) & 1));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float s, c;
          sincos_<FAST>(y[k], s, c);
          y[k] = y_is_gbar ? vb[k] * gm[k] * c : s;
        }
        reinterpret_cast<f32x4*>(sy)[dq] = y;
      }
      __syncthreads();
      const int fo = 32 * wave + i;
#pragma unroll 4
      for (int s = 0; s < 16; ++s) {
        const int p = 2 * s + h;
        const float a = sx[slot_index(fo, p)];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float b = sy[slot_index(32 * t + i, p)];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        }
      }
    }
  }
  // D[o][i]: column = lane & 31 (input feature i within tile t), row = (reg&3) + 8*(reg>>2) + 4h (o within the wave's strip)
  float* dst = d_wmat + (size_t)m * C * C;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int oo = 32 * wave + (rg & 3) + 8 * (rg >> 2) + 4 * h;
      atomicAdd(dst + (size_t)oo * C + 32 * t + i, acc[t][rg]);
    }
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
__device__ __forceinline__ void frag16(const float* sl, int f, int p0, f16x8& hi, f16x8& lo) {
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = sl[slot_index4(f, p0 + q)];
  split8_pairs(v, hi, lo);
}

// the scratch operands are read exactly once by this kernel: non-temporal loads (streaming-read ceiling of the box
// 5.8 TB/s ordinary, 6.5-7.0 non-temporal: tools/dbg/hbm_read.hip)
#ifndef OI_WGRAD_NT
#define OI_WGRAD_NT 1
#endif
__device__ __forceinline__ f32x4 ld_once(const f32x4* p) { return OI_WGRAD_NT ? __builtin_nontemporal_load(p) : *p; }

template <bool FAST>
__global__ void __launch_bounds__(256)
mlp_wgrad_f16_kernel(const char* __restrict__ scratch, const float* __restrict__ op_max, const float* __restrict__ gamma,
                     float* __restrict__ d_wmat, long long n_wave_tiles, long long wt_per_elem, int tiles_per_chunk,
                     int has_col) {
  __shared__ __attribute__((aligned(16))) float sx[WG16_SLOT_FLOATS], sy[WG16_SLOT_FLOATS];
  __shared__ __attribute__((aligned(16))) f16x8 sb[2][4][2][64];  // [hi|lo][column tile][k-step][lane]
  const int m = blockIdx.y;
  if (m == 7 && !has_col) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, i = lane & 31;
  const long long t_begin = (long long)blockIdx.x * tiles_per_chunk;
  const long long t_end = min(n_wave_tiles, t_begin + tiles_per_chunk);
  const int npair = (m == 7) ? 1 : 2;
  // ONE power-of-two scale per operand for the whole launch (maxima published by the sweep): the products of all tiles
  // then share a scale and accumulate straight in the MFMA accumulators -- no per-tile maximum, no per-tile merge.
  // pair 0: X = v_l (uvbar for the colour head), Y = gbar_l (a_8 = sin, |.| <= 1);  pair 1: X = ubar_l, Y = a_l = sin
  float scx[2], scy[2], inv[2];
  {
    // maximum over the replicas: lane r of every wave reads replica r (the table is 8 KiB and L2-resident)
    const float* rep = op_max + lane * OM_STRIDE;
    const float mx0 = oi::wave_max(rep[m == 7 ? OM_UV : OM_V + m]);
    const float my0 = m == 7 ? 1.0f : oi::wave_max(rep[OM_G + m]);
    const float mx1 = oi::wave_max(rep[OM_U + (m < 7 ? m : 0)]);
    float ivx, ivy;
    pow2_scale_of(mx0, scx[0], ivx);
    pow2_scale_of(my0, scy[0], ivy);
    inv[0] = ivx * ivy;
    pow2_scale_of(mx1, scx[1], ivx);
    pow2_scale_of(1.0f, scy[1], ivy);
    inv[1] = ivx * ivy;
  }
  f32x16 acc[2][4];
  acc_zero(acc[0]);
  acc_zero(acc[1]);
  const int fo = 32 * wave + i;
  for (long long wt = t_begin; wt < t_end; ++wt) {
    const char* base = scratch + wt * (long long)(NSLOT_BWD * 16384);
    // both Y operands of a layer matrix come from the SAME parked phi_{l-1}: one read, one sin / cos per element
    //   pair 0: Y = gbar_l = vbar_{l-1} gamma_{l-1} cos(phi_{l-1})      pair 1: Y = a_l = sin(phi_{l-1})
    const f32x4* gp4 = reinterpret_cast<const f32x4*>(base + (size_t)(S_PHI + m) * 16384);
    const f32x4* gv4 = reinterpret_cast<const f32x4*>(base + (size_t)(S_VB + (m < 7 ? m : 0)) * 16384);
    const float* grow = gamma + ((wt / wt_per_elem) * 9 + (m < 7 ? m : 0)) * C;
    // every load of the tile is issued before any of it is used: ONE memory round trip per tile
    const f32x4* gx0 = reinterpret_cast<const f32x4*>(base + (size_t)(m == 7 ? S_UV : S_V + m) * 16384);
    const f32x4* gx1 = reinterpret_cast<const f32x4*>(base + (size_t)(S_U + (m < 7 ? m : 0)) * 16384);
    f32x4 ysin[4], ygb[4], xall[2][4], ph4[4], vb4[4], gm4[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int q = it * 256 + tid;
      ph4[it] = ld_once(gp4 + q);
      xall[0][it] = ld_once(gx0 + q);
      vb4[it] = gm4[it] = f32x4{0.f, 0.f, 0.f, 0.f};
      xall[1][it] = xall[0][it];
      if (m < 7) {
        vb4[it] = ld_once(gv4 + q);
        gm4[it] = *reinterpret_cast<const f32x4*>(grow + grp_f0(q >> 6) + 4 * ((q >> 5) & 1));
        xall[1][it] = ld_once(gx1 + q);
      }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float sn, cs;
        sincos_<FAST>(ph4[it][k], sn, cs);
        ysin[it][k] = sn;
        ygb[it][k] = vb4[it][k] * (gm4[it][k] * scy[0]) * cs;
      }
    }
    for (int pr = 0; pr < npair; ++pr) {
      const bool y_is_gbar = (m < 7) && pr == 0;
      __syncthreads();  // previous pair's readers of sx / sy / sb are done
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int q = it * 256 + tid;
        const int dq = q + (q >> 5);  // + 4 floats per 32-point block
        reinterpret_cast<f32x4*>(sx)[dq] = xall[pr][it] * scx[pr];
        reinterpret_cast<f32x4*>(sy)[dq] = y_is_gbar ? ygb[it] : ysin[it] * scy[pr];
      }
      __syncthreads();
      // this wave's column tile of Y -> shared fp16 fragments
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        f16x8 bh, bl;
        frag16(sy, fo, 16 * ks + 8 * h, bh, bl);
        sb[0][wave][ks][lane] = bh;
        sb[1][wave][ks][lane] = bl;
      }
      f16x8 ah[2], al[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) frag16(sx, fo, 16 * ks + 8 * h, ah[ks], al[ks]);
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f16x8 bh = sb[0][t][ks][lane], bl = sb[1][t][ks][lane];
          acc[pr][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh, acc[pr][t], 0, 0, 0);
          acc[pr][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl, acc[pr][t], 0, 0, 0);
          acc[pr][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh, acc[pr][t], 0, 0, 0);
        }
      }
    }
  }
  float* dst = d_wmat + (size_t)m * C * C;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int oo = 32 * wave + (rg & 3) + 8 * (rg >> 2) + 4 * h;
      const float v = npair == 2 ? fmaf(acc[1][t][rg], inv[1], acc[0][t][rg] * inv[0]) : acc[0][t][rg] * inv[0];
      atomicAdd(dst + (size_t)oo * C + 32 * t + i, v);
    }
}

template <int PREC, bool FAST>
int launch_bwd(const float* pts, const void* packed, const float* gamma, const float* beta, const float* grad_fwd,
               const float* rgb_fwd, const float* feat_fwd, const float* g_sdf, const float* g_grad, const float* g_rgb, float* d_small,
               float* d_wmat, float* d_gamma, float* d_beta, void* scratch, size_t scratch_bytes, int B, long long n,
               hipStream_t st) {
  // The scratch is a bound, not a function of the problem: the points of every batch element are processed in chunks of
  // as many 128-point tiles as `scratch_bytes` holds; all outputs are accumulated, so chunks simply add up.
  const long long tile_bytes = 4LL * NSLOT_BWD * 16384;
  const long long tiles_all = oi::cdiv(n, TILE_PTS);
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
  auto k = mlp_bwd_sweep_kernel<PREC, FAST>;
  // per launch: the attribute is per device, and a process may drive several
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, L_TOTAL_BWD);
  for (long long t0 = 0; t0 < tiles_all; t0 += tiles_fit) {
    const long long nt = std::min(tiles_fit, tiles_all - t0);
    const long long off = t0 * TILE_PTS, cn = std::min<long long>(nt * TILE_PTS, n - off);
    dim3 grid((unsigned)nt, B), block(256);
    if constexpr (PREC == OI_PREC_F16X3) {
      hipError_t e = oi::zero_async(op_max, OM_FLOATS, st);
      if (e != hipSuccess) return oi::fail(OI_ERR_LAUNCH, "oi_sdf_mlp_bwd: zero fill: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k, grid, block, L_TOTAL_BWD, st, pts, reinterpret_cast<const char*>(packed), gamma, beta, grad_fwd,
                       rgb_fwd, feat_fwd, g_sdf, g_grad, g_rgb, d_small, d_gamma, d_beta, tiles, op_max, cn, n, off);
    int rc = oi::check_launch("oi_sdf_mlp_bwd(sweep)");
    if (rc != OI_OK) return rc;
    const long long n_wt = (long long)B * grid.x * 4;
    int chunk = (int)std::max<long long>(1, (n_wt * 8 + 2047) / 2048);  // ~2048 workgroups in total
    dim3 g2(oi::cdiv(n_wt, chunk), 8);
    if constexpr (PREC == OI_PREC_F16X3) {
      hipLaunchKernelGGL(mlp_wgrad_f16_kernel<FAST>, g2, block, 0, st, tiles, op_max, gamma, d_wmat, n_wt,
                         (long long)grid.x * 4, chunk, has_col);
    } else {
      hipLaunchKernelGGL(mlp_wgrad_kernel<FAST>, g2, block, 0, st, tiles, gamma, d_wmat, n_wt, (long long)grid.x * 4, chunk,
                         has_col);
    }
    rc = oi::check_launch("oi_sdf_mlp_bwd(wgrad)");
    if (rc != OI_OK) return rc;
  }
  return OI_OK;
}

}  // namespace

#ifdef OI_BWD_PROF
extern "C" int oi_prof_bwd_read(unsigned long long* out, int reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(oi_prof_bwd), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(oi_prof_bwd), z, sizeof(z));
  }
  return 0;
}
#endif

extern "C" {

size_t oi_mlp_bwd_scratch_bytes(int B, long long n_per_elem) {
  const long long tiles = (n_per_elem + TILE_PTS - 1) / TILE_PTS;
  return (size_t)B * tiles * 4 * NSLOT_BWD * 16384 + OM_FLOATS * sizeof(float);
}

size_t oi_mlp_bwd_scratch_bytes_capped(int B, long long n_per_elem, size_t cap_bytes) {
  const size_t per_tile = (size_t)B * 4 * NSLOT_BWD * 16384, head = OM_FLOATS * sizeof(float);
  const size_t full = oi_mlp_bwd_scratch_bytes(B, n_per_elem);
  if (full <= cap_bytes) return full;
  const size_t tiles = cap_bytes > head ? (cap_bytes - head) / per_tile : 0;
  return (tiles < 1 ? 1 : tiles) * per_tile + head;
}

int oi_mlp_bwd_small_floats(void) { return DS_TOTAL; }

int oi_sdf_mlp_bwd(const float* pts, const void* packed, const float* gamma, const float* beta, const float* grad_fwd,
                   const float* rgb_fwd, const float* feat_fwd, const float* g_sdf, const float* g_grad, const float* g_rgb, float* d_small,
                   float* d_wmat, float* d_gamma, float* d_beta, void* scratch, size_t scratch_bytes, int B,
                   long long n_per_elem, int prec, int fast_trig, oi_stream_t stream) {
  OI_REQUIRE(pts && packed && gamma && beta && d_small && d_wmat && d_gamma && d_beta && scratch,
             "oi_sdf_mlp_bwd: null pointer");
  OI_REQUIRE(B > 0 && n_per_elem > 0, "oi_sdf_mlp_bwd: B=%d n=%lld", B, n_per_elem);
  OI_REQUIRE((rgb_fwd == nullptr) == (g_rgb == nullptr) || g_rgb == nullptr, "oi_sdf_mlp_bwd: g_rgb needs rgb_fwd");
  OI_REQUIRE(g_rgb == nullptr || (grad_fwd != nullptr && feat_fwd != nullptr),
             "oi_sdf_mlp_bwd: colour backward needs the forward gradient and the forward features");
  OI_REQUIRE(prec != OI_PREC_BF16X6, "oi_sdf_mlp_bwd: pass the OI_PREC_F32 image for the backward of the BF16X6 mode");
  hipStream_t st = oi::as_stream(stream);
#define OI_BWD_CASE(P)                                                                                              \
  case P:                                                                                                           \
    return fast_trig ? launch_bwd<P, true>(pts, packed, gamma, beta, grad_fwd, rgb_fwd, feat_fwd, g_sdf, g_grad, g_rgb, d_small, \
                                           d_wmat, d_gamma, d_beta, scratch, scratch_bytes, B, n_per_elem, st)      \
                     : launch_bwd<P, false>(pts, packed, gamma, beta, grad_fwd, rgb_fwd, feat_fwd, g_sdf, g_grad, g_rgb, d_small, \
                                            d_wmat, d_gamma, d_beta, scratch, scratch_bytes, B, n_per_elem, st);
  switch (prec) {
    OI_BWD_CASE(OI_PREC_F32)
    OI_BWD_CASE(OI_PREC_BF16X3)
    OI_BWD_CASE(OI_PREC_BF16)
    OI_BWD_CASE(OI_PREC_F16X3)
    default:
      return oi::fail(OI_ERR_INVALID_ARG, "oi_sdf_mlp_bwd: bad precision %d", prec);
  }
#undef OI_BWD_CASE
}

}  // extern "C"
