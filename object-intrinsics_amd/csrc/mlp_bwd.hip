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
//   phase A  (l up)    recompute phi_l                                  [7 GEMMs]   flash-style recompute
//   colour             backward of the albedo head                      [2 GEMMs]
//   phase B  (l down)  g_{l+1}, v_l                                     [7 GEMMs]
//   phase C  (l up)    gbar_0 = dL/dgrad; vbar_l = W_l gbar_l; gbar_{l+1} = vbar_l * c_l; cbar_l = vbar_l * g_{l+1}
//   phase D  (l down)  phibar_l = abar_{l+1} cos phi_l - cbar_l gamma_l sin phi_l; ubar_l = phibar_l gamma_l;
//                      abar_l = W_l^T ubar_l; gamma/beta/bias gradients
// All four sweeps keep the point on the MFMA column (activations never leave registers inside a sweep);
// per-layer fragments that a later sweep needs are parked in a global scratch (buffer addressing).
// The weight gradients  dW_l = sum_p ( v_l gbar_l^T + ubar_l a_l^T )  are a second, split-K GEMM kernel
// over the parked operands (K = points).
#include <algorithm>

#include "mlp_common.h"

namespace {

using namespace oimlp;

// scratch slots of one wave tile (16 KiB each)
constexpr int S_PHI = 0;    // 8: phi_l
constexpr int S_G = 8;      // 7: g_{l+1}, l = 0..6   (g_8 = w_sigma is a constant)
constexpr int S_CB = 15;    // 8: cbar_l
constexpr int S_V = 23;     // 7: v_l,    l = 1..7   (wgrad operand)
constexpr int S_GB = 30;    // 7: gbar_l, l = 1..7   (wgrad operand)
constexpr int S_U = 37;     // 7: ubar_l, l = 1..7   (wgrad operand)
constexpr int S_UV = 44;    // 1: uvbar (colour head pre-activation gradient)
constexpr int S_AC = 45;    // 1: abar_8 contribution of the colour head
constexpr int NSLOT_BWD = 46;

// small-gradient buffer (floats)
constexpr int DS_W0 = 0;       // [128][3]
constexpr int DS_B = 384;      // [9][128]  b0..b7, bv
constexpr int DS_WSIG = 1536;  // [128]
constexpr int DS_BSIG = 1664;  // [1] (+3 pad)
constexpr int DS_WVX = 1668;   // [128][3]  Wv[:, 128:131]
constexpr int DS_WRGB = 2052;  // [3][128]
constexpr int DS_BRGB = 2436;  // [3] (+1 pad)
constexpr int DS_TOTAL = 2440;

constexpr int L_RACC = L_WBUF + 65536;  // per-workgroup reduction scratch: [8][128] floats
constexpr int L_TOTAL_BWD = L_RACC + 8 * C * 4;

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

// sum of v over the 32 points (lanes of one half) -> LDS accumulator row `row` at this lane's feature
__device__ __forceinline__ void reduce_group(char* lds, int row, int g, int h, int j, f32x4 v) {
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = oi::half_sum32(v[k]);  // 5 DPP adds; valid in lanes 16..31 of each half
  if (j == 16) {
    float* racc = reinterpret_cast<float*>(lds + L_RACC) + row * C + grp_f0(g) + 4 * h;
#pragma unroll
    for (int k = 0; k < 4; ++k) atomicAdd(racc + k, v[k]);
  }
}

__device__ __forceinline__ void racc_zero(char* lds, int tid) {
  float* racc = reinterpret_cast<float*>(lds + L_RACC);
  for (int i = tid; i < 8 * C; i += 256) racc[i] = 0.f;
}
// flush `rows` accumulator rows: row r goes to dst[r] (a global base pointer per row)
__device__ __forceinline__ void racc_flush_row(char* lds, int row, float* dst, int stride, int tid) {
  const float* racc = reinterpret_cast<const float*>(lds + L_RACC) + row * C;
  if (tid < C) atomicAdd(dst + tid * stride, racc[tid]);
}

// this wave's LDS-DMA has landed (and so have its outstanding scratch loads), then rendezvous
__device__ __forceinline__ void dma_sync() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

template <int PREC, bool FAST>
__global__ void __launch_bounds__(256, 2)
mlp_bwd_sweep_kernel(const float* __restrict__ pts, const char* __restrict__ packed, const float* __restrict__ gamma,
                     const float* __restrict__ beta, const float* __restrict__ grad_fwd,
                     const float* __restrict__ rgb_fwd, const float* __restrict__ g_sdf,
                     const float* __restrict__ g_grad, const float* __restrict__ g_rgb, float* __restrict__ d_small,
                     float* __restrict__ d_gamma, float* __restrict__ d_beta, char* __restrict__ scratch,
                     long long n_per_elem) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, j = lane & 31;
  const int e = blockIdx.y;
  const float* hdr = reinterpret_cast<const float*>(packed);
  const char* mats = packed + H_BYTES;
  const bool has_col = rgb_fwd != nullptr && g_rgb != nullptr;

  LaneOff o;
  o.h16 = 16 * h;
  o.h64 = 64 * h;
  o.l16 = 16 * lane;
  o.l16hi = 16 * lane + 32768;
  asm volatile("" : "+v"(o.h16), "+v"(o.h64), "+v"(o.l16), "+v"(o.l16hi));

  const long long local = (long long)blockIdx.x * TILE_PTS + wave * WAVE_PTS + j;
  const bool valid = local < n_per_elem;
  const long long pt = (long long)e * n_per_elem + (valid ? local : n_per_elem - 1);
  const float vmask = valid ? 1.f : 0.f;  // tail points contribute nothing

  WaveScratchB ws;
  {
    const long long wt = ((long long)e * gridDim.x + blockIdx.x) * 4 + wave;
    ws.rs = __builtin_amdgcn_make_buffer_rsrc(scratch + wt * (long long)(NSLOT_BWD * 16384), 0, NSLOT_BWD * 16384,
                                              0x00020000);
  }
  {
    float* tabs = reinterpret_cast<float*>(lds + L_TABS);
    for (int i = tid; i < H_TABS_END; i += 256) tabs[i] = hdr[i];
    stage_film(lds, gamma, beta, hdr, e, 0, tid);
    racc_zero(lds, tid);
  }
  const float px = pts[pt * 3 + 0], py = pts[pt * 3 + 1], pz = pts[pt * 3 + 2];
  const float gs = (g_sdf ? g_sdf[pt] : 0.f) * vmask;
  float Gx = (g_grad ? g_grad[pt * 3 + 0] : 0.f) * vmask, Gy = (g_grad ? g_grad[pt * 3 + 1] : 0.f) * vmask,
        Gz = (g_grad ? g_grad[pt * 3 + 2] : 0.f) * vmask;
  __syncthreads();

  float act[64];
  f32x16 acc[4];
  const float* film = reinterpret_cast<const float*>(lds + L_FILM);
  // F16X3: the images carry a power-of-two scale 2^k_m (header H_WSCALE holds 2^-k_m); every GEMM returns the factor
  // its accumulators still need (gemm_scaled), adjoint vectors are normalised per point before the fp16 split.
  constexpr bool SC = PREC == OI_PREC_F16X3;

  // ================= phase A: recompute phi_l (ascending) =================
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const f32x4 gm = lds_f4(lds, L_FILM + grp_f0(g) * 4, o.h16);
    const f32x4 bt = lds_f4(lds, L_FILM + (C + grp_f0(g)) * 4, o.h16);
    const f32x4 bs = lds_f4(lds, L_FILM + (2 * C + grp_f0(g)) * 4, o.h16);
    f32x4 ph;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 w = lds_f4(lds, L_TABS + H_TAB0 * 4 + (grp_f0(g) + k) * 16, o.h64);
      const float u = fmaf(pz, w[2], fmaf(py, w[1], px * w[0])) + bs[k];
      ph[k] = fmaf(gm[k], u, bt[k]);
      float s, c;
      sincos_<FAST>(ph[k], s, c);
      act[4 * g + k] = s;
    }
    ws.store(S_PHI + 0, g, o.l16, ph);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int l = 1; l < NL_SDF; ++l) {
    __syncthreads();
    stage_layer_dma<PREC>(lds, mats + (size_t)(l - 1) * layer_bytes(PREC), wave, lane);
    stage_film(lds, gamma, beta, hdr, e, l, tid);
    dma_sync();
    if constexpr (SC) acc_zero(acc); else init_bias(lds, o, acc);
    const float fA = gemm_scaled<PREC, false>(lds, o, act, acc, SC ? hdr[H_WSCALE + l - 1] : 1.f);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const f32x4 gm = lds_f4(lds, L_FILM + grp_f0(g) * 4, o.h16);
      const f32x4 bt = lds_f4(lds, L_FILM + (C + grp_f0(g)) * 4, o.h16);
      f32x4 bs;
      if constexpr (SC) bs = lds_f4(lds, L_FILM + (2 * C + grp_f0(g)) * 4, o.h16);
      f32x4 ph;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float u = SC ? fmaf(acc[g >> 2][4 * (g & 3) + k], fA, bs[k]) : acc[g >> 2][4 * (g & 3) + k];
        ph[k] = fmaf(gm[k], u, bt[k]);
        float s, c;
        sincos_<FAST>(ph[k], s, c);
        act[4 * g + k] = s;
      }
      ws.store(S_PHI + l, g, o.l16, ph);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // act = a_8 (features)

  // ================= colour head backward =================
  if (has_col) {
    __syncthreads();
    stage_layer_dma<PREC>(lds, mats + (size_t)14 * layer_bytes(PREC), wave, lane);
    stage_film(lds, gamma, beta, hdr, e, 8, tid);
    dma_sync();
    const float fx = grad_fwd[pt * 3 + 0], fy = grad_fwd[pt * 3 + 1], fz = grad_fwd[pt * 3 + 2];
    float rho[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float rv = rgb_fwd[pt * 3 + k];
      rho[k] = g_rgb[pt * 3 + k] * rv * (1.0f - rv) * vmask;  // through the sigmoid
    }
    if constexpr (SC) acc_zero(acc); else init_bias(lds, o, acc);
    const float fV = gemm_scaled<PREC, false>(lds, o, act, acc, SC ? hdr[H_WSCALE + 14] : 1.f);
    // uv -> phiv -> hv; then uvbar.  Reductions: rows 0 gamma_v, 1 beta_v, 2 bv, 3..5 Wrgb, (6,7 free)
    float dGx = 0.f, dGy = 0.f, dGz = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const f32x4 gm = lds_f4(lds, L_FILM + grp_f0(g) * 4, o.h16);
      const f32x4 bt = lds_f4(lds, L_FILM + (C + grp_f0(g)) * 4, o.h16);
      const f32x4 w0 = lds_f4(lds, L_TABS + (H_RGB + 0 * C + grp_f0(g)) * 4, o.h16);
      const f32x4 w1 = lds_f4(lds, L_TABS + (H_RGB + 1 * C + grp_f0(g)) * 4, o.h16);
      const f32x4 w2 = lds_f4(lds, L_TABS + (H_RGB + 2 * C + grp_f0(g)) * 4, o.h16);
      f32x4 uvb, r_g, r_b, r0, r1, r2, bsv;
      if constexpr (SC) bsv = lds_f4(lds, L_FILM + (2 * C + grp_f0(g)) * 4, o.h16);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 wx = lds_f4(lds, L_TABS + H_TABV * 4 + (grp_f0(g) + k) * 16, o.h64);
        const float ua = SC ? fmaf(acc[g >> 2][4 * (g & 3) + k], fV, bsv[k]) : acc[g >> 2][4 * (g & 3) + k];
        const float uv = ua + fmaf(fz, wx[2], fmaf(fy, wx[1], fx * wx[0]));
        const float phiv = fmaf(gm[k], uv, bt[k]);
        float hv, cv;
        sincos_<FAST>(phiv, hv, cv);
        const float hvb = w0[k] * rho[0] + w1[k] * rho[1] + w2[k] * rho[2];
        const float phb = hvb * cv;
        uvb[k] = phb * gm[k];
        r_g[k] = phb * uv;
        r_b[k] = phb;
        r0[k] = rho[0] * hv; r1[k] = rho[1] * hv; r2[k] = rho[2] * hv;
        dGx = fmaf(uvb[k], wx[0], dGx);
        dGy = fmaf(uvb[k], wx[1], dGy);
        dGz = fmaf(uvb[k], wx[2], dGz);
      }
      ws.store(S_UV, g, o.l16, uvb);
#pragma unroll
      for (int k = 0; k < 4; ++k) act[4 * g + k] = uvb[k];
      reduce_group(lds, 0, g, h, j, r_g);
      reduce_group(lds, 1, g, h, j, r_b);
      reduce_group(lds, 2, g, h, j, uvb);
      reduce_group(lds, 3, g, h, j, r0);
      reduce_group(lds, 4, g, h, j, r1);
      reduce_group(lds, 5, g, h, j, r2);
      __builtin_amdgcn_sched_barrier(0);
    }
    // dWv[:, 128+t] = sum_p uvbar * grad_t  (rows 6,7 then a second round for the third column)
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      f32x4 uvb;
#pragma unroll
      for (int k = 0; k < 4; ++k) uvb[k] = act[4 * g + k];
      reduce_group(lds, 6, g, h, j, uvb * fx);
      reduce_group(lds, 7, g, h, j, uvb * fy);
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
    racc_flush_row(lds, 2, d_small + DS_B + 8 * C, 1, tid);
    racc_flush_row(lds, 3, d_small + DS_WRGB + 0 * C, 1, tid);
    racc_flush_row(lds, 4, d_small + DS_WRGB + 1 * C, 1, tid);
    racc_flush_row(lds, 5, d_small + DS_WRGB + 2 * C, 1, tid);
    racc_flush_row(lds, 6, d_small + DS_WVX + 0, 3, tid);
    racc_flush_row(lds, 7, d_small + DS_WVX + 1, 3, tid);
    __syncthreads();
    racc_zero(lds, tid);
    // abar_8 from the colour head: Wv[:, :128]^T uvbar   (transposed colour image, matrix 15)
    stage_layer_dma<PREC>(lds, mats + (size_t)15 * layer_bytes(PREC), wave, lane);
    dma_sync();
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      f32x4 uvb;
#pragma unroll
      for (int k = 0; k < 4; ++k) uvb[k] = act[4 * g + k];
      reduce_group(lds, 0, g, h, j, uvb * fz);
    }
    acc_zero(acc);
    const float fT = gemm_scaled<PREC, true>(lds, o, act, acc, SC ? hdr[H_WSCALE + 15] : 1.f);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = SC ? acc[g >> 2][4 * (g & 3) + k] * fT : acc[g >> 2][4 * (g & 3) + k];
      ws.store(S_AC, g, o.l16, v);
    }
    __syncthreads();
    racc_flush_row(lds, 0, d_small + DS_WVX + 2, 3, tid);
    __syncthreads();
    racc_zero(lds, tid);
  }

  // ================= phase B: reverse sweep g_{l+1}, v_l (descending) =================
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const f32x4 w = lds_f4(lds, L_TABS + (H_SIG + grp_f0(g)) * 4, o.h16);
#pragma unroll
    for (int k = 0; k < 4; ++k) act[4 * g + k] = w[k];
  }
  for (int l = NL_SDF - 1; l >= 1; --l) {
    // the layer's phi fragments are requested before the image is staged: their HBM latency overlaps the staging and
    // the two barriers instead of being paid once per group of four features (the sweep is latency-, not math-bound)
    f32x4 phv[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) phv[g] = ws.load<OI_BWD_LD_EARLY>(S_PHI + l, g, o.l16);
    __syncthreads();
    stage_layer_dma<PREC>(lds, mats + (size_t)(7 + l - 1) * layer_bytes(PREC), wave, lane);
    stage_film(lds, gamma, beta, hdr, e, l, tid);
    dma_sync();
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const f32x4 ph = phv[g];
      const f32x4 gm = lds_f4(lds, L_FILM + grp_f0(g) * 4, o.h16);
      f32x4 gsv, vv;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float s, c;
        sincos_<FAST>(ph[k], s, c);
        gsv[k] = act[4 * g + k];
        vv[k] = gsv[k] * gm[k] * c;
        act[4 * g + k] = vv[k];
      }
      if (l < NL_SDF - 1) ws.store(S_G + l, g, o.l16, gsv);  // g_{l+1}; g_8 is the constant w_sigma
      ws.store<OI_BWD_ST_WGRAD>(S_V + l - 1, g, o.l16, vv);
      __builtin_amdgcn_sched_barrier(0);
    }
    acc_zero(acc);
    const float fB = gemm_scaled<PREC, true>(lds, o, act, acc, SC ? hdr[H_WSCALE + 7 + l - 1] : 1.f);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) act[16 * t + r] = SC ? acc[t][r] * fB : acc[t][r];
  }
#pragma unroll
  for (int g = 0; g < 16; ++g) {  // g_1
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = act[4 * g + k];
    ws.store(S_G + 0, g, o.l16, v);
  }

  // ================= phase C: gbar sweep (ascending) =================
  // layer 0: vbar_0 = W0 gbar_0 (gbar_0 = dL/dgrad, 3-vector); reductions rows 0..2: dW0 += v_0 gbar_0^T
  __syncthreads();
  stage_film(lds, gamma, beta, hdr, e, 0, tid);
  __syncthreads();
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const f32x4 ph = ws.load<OI_BWD_LD_EARLY>(S_PHI + 0, g, o.l16);
    const f32x4 g1 = ws.load(S_G + 0, g, o.l16);
    const f32x4 gm = lds_f4(lds, L_FILM + grp_f0(g) * 4, o.h16);
    f32x4 cb, v0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 w = lds_f4(lds, L_TABS + H_TAB0 * 4 + (grp_f0(g) + k) * 16, o.h64);
      float s, c;
      sincos_<FAST>(ph[k], s, c);
      const float cl = gm[k] * c;
      const float vb = fmaf(Gz, w[2], fmaf(Gy, w[1], Gx * w[0]));  // vbar_0
      act[4 * g + k] = vb * cl;                                   // gbar_1
      cb[k] = vb * g1[k];                                         // cbar_0
      v0[k] = g1[k] * cl;                                         // v_0
    }
    ws.store(S_CB + 0, g, o.l16, cb);
    reduce_group(lds, 0, g, h, j, v0 * Gx);
    reduce_group(lds, 1, g, h, j, v0 * Gy);
    reduce_group(lds, 2, g, h, j, v0 * Gz);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  racc_flush_row(lds, 0, d_small + DS_W0 + 0, 3, tid);
  racc_flush_row(lds, 1, d_small + DS_W0 + 1, 3, tid);
  racc_flush_row(lds, 2, d_small + DS_W0 + 2, 3, tid);
  for (int l = 1; l < NL_SDF; ++l) {
    __syncthreads();
    if (l == 1) racc_zero(lds, tid);
    stage_layer_dma<PREC>(lds, mats + (size_t)(l - 1) * layer_bytes(PREC), wave, lane);
    stage_film(lds, gamma, beta, hdr, e, l, tid);
    dma_sync();
#pragma unroll
    for (int g = 0; g < 16; ++g) {  // park gbar_l for the weight-gradient GEMM
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = act[4 * g + k];
      ws.store<OI_BWD_ST_WGRAD>(S_GB + l - 1, g, o.l16, v);
    }
    acc_zero(acc);
    const float fC = gemm_scaled<PREC, true>(lds, o, act, acc, SC ? hdr[H_WSCALE + l - 1] : 1.f);  // vbar_l = W_l gbar_l
    f32x4 phc[16], gnc[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      phc[g] = ws.load<OI_BWD_LD_EARLY>(S_PHI + l, g, o.l16);
      if (l < NL_SDF - 1) gnc[g] = ws.load(S_G + l, g, o.l16);
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const f32x4 ph = phc[g];
      const f32x4 gm = lds_f4(lds, L_FILM + grp_f0(g) * 4, o.h16);
      f32x4 gn;
      if (l < NL_SDF - 1) gn = gnc[g];
      else gn = lds_f4(lds, L_TABS + (H_SIG + grp_f0(g)) * 4, o.h16);
      f32x4 cb;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float s, c;
        sincos_<FAST>(ph[k], s, c);
        const float vb = SC ? acc[g >> 2][4 * (g & 3) + k] * fC : acc[g >> 2][4 * (g & 3) + k];
        act[4 * g + k] = vb * gm[k] * c;  // gbar_{l+1}
        cb[k] = vb * gn[k];               // cbar_l
      }
      ws.store(S_CB + l, g, o.l16, cb);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // d w_sigma += sum_p gbar_8   (row 3)
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = act[4 * g + k];
    reduce_group(lds, 3, g, h, j, v);
  }

  // ================= phase D: abar sweep (descending) =================
  // abar_8 = gs * w_sigma (+ colour head); also d w_sigma += gs * a_8 (row 3), d b_sigma += gs
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const f32x4 w = lds_f4(lds, L_TABS + (H_SIG + grp_f0(g)) * 4, o.h16);
    const f32x4 ph = ws.load(S_PHI + 7, g, o.l16);
    f32x4 ac = {0.f, 0.f, 0.f, 0.f};
    if (has_col) ac = ws.load(S_AC, g, o.l16);
    f32x4 a8;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float s, c;
      sincos_<FAST>(ph[k], s, c);
      a8[k] = s * gs;
      act[4 * g + k] = fmaf(gs, w[k], ac[k]);
    }
    reduce_group(lds, 3, g, h, j, a8);
    __builtin_amdgcn_sched_barrier(0);
  }
  {
    float b = (h == 0) ? gs : 0.f;
    b = oi::wave_sum(b);
    if (lane == 0) atomicAdd(d_small + DS_BSIG, b);
  }
  __syncthreads();
  racc_flush_row(lds, 3, d_small + DS_WSIG, 1, tid);
  for (int l = NL_SDF - 1; l >= 0; --l) {
    f32x4 phd[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) phd[g] = ws.load(S_PHI + l, g, o.l16);
    __syncthreads();
    racc_zero(lds, tid);
    if (l >= 1) stage_layer_dma<PREC>(lds, mats + (size_t)(7 + l - 1) * layer_bytes(PREC), wave, lane);
    stage_film(lds, gamma, beta, hdr, e, l, tid);
    f32x4 cbv[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) cbv[g] = ws.load(S_CB + l, g, o.l16);
    dma_sync();
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const f32x4 ph = phd[g];
      const f32x4 cb = cbv[g];
      const f32x4 gm = lds_f4(lds, L_FILM + grp_f0(g) * 4, o.h16);
      const f32x4 bt = lds_f4(lds, L_FILM + (C + grp_f0(g)) * 4, o.h16);
      f32x4 r_g, r_b, ub;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float s, c;
        sincos_<FAST>(ph[k], s, c);
        const float phb = act[4 * g + k] * c - cb[k] * gm[k] * s;   // phibar_l
        const float u = (ph[k] - bt[k]) * __builtin_amdgcn_rcpf(gm[k]);  // u_l
        r_g[k] = fmaf(phb, u, cb[k] * c);                           // d gamma_l
        r_b[k] = phb;                                               // d beta_l
        ub[k] = phb * gm[k];                                        // ubar_l
        act[4 * g + k] = ub[k];
      }
      reduce_group(lds, 0, g, h, j, r_g);
      reduce_group(lds, 1, g, h, j, r_b);
      reduce_group(lds, 2, g, h, j, ub);
      if (l >= 1) {
        ws.store<OI_BWD_ST_WGRAD>(S_U + l - 1, g, o.l16, ub);
      } else {  // d W0 += ubar_0 x^T
        reduce_group(lds, 3, g, h, j, ub * px);
        reduce_group(lds, 4, g, h, j, ub * py);
        reduce_group(lds, 5, g, h, j, ub * pz);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (l >= 1) {
      acc_zero(acc);
      const float fD = gemm_scaled<PREC, true>(lds, o, act, acc, SC ? hdr[H_WSCALE + 7 + l - 1] : 1.f);  // abar_l = W_l^T ubar_l
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) act[16 * t + r] = SC ? acc[t][r] * fD : acc[t][r];
    }
    __syncthreads();
    racc_flush_row(lds, 0, d_gamma + ((size_t)e * 9 + l) * C, 1, tid);
    racc_flush_row(lds, 1, d_beta + ((size_t)e * 9 + l) * C, 1, tid);
    racc_flush_row(lds, 2, d_small + DS_B + l * C, 1, tid);
    if (l == 0) {
      racc_flush_row(lds, 3, d_small + DS_W0 + 0, 3, tid);
      racc_flush_row(lds, 4, d_small + DS_W0 + 1, 3, tid);
      racc_flush_row(lds, 5, d_small + DS_W0 + 2, 3, tid);
    }
  }
}

// ------------------------------------------------------------------------------------------
// weight-gradient GEMM:  dW_m[o][i] += sum_p X[o][p] Y[i][p]  (K = points), fp32 MFMA, split over
// chunks of wave tiles; one workgroup = one (matrix, chunk), wave w owns output rows 32w..32w+31.
//   matrices m = 0..6  (layer l = m+1):  X1 = v_l, Y1 = gbar_l;  X2 = ubar_l, Y2 = a_l = sin(phi_{l-1})
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
mlp_wgrad_kernel(const char* __restrict__ scratch, float* __restrict__ d_wmat, long long n_wave_tiles,
                 int tiles_per_chunk, int has_col) {
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
      bool y_is_phi;
      if (m == 7) { sxi = S_UV; syi = S_PHI + 7; y_is_phi = true; }
      else if (pr == 0) { sxi = S_V + m; syi = S_GB + m; y_is_phi = false; }
      else { sxi = S_U + m; syi = S_PHI + m; y_is_phi = true; }  // a_l = sin(phi_{l-1}), l = m+1
      __syncthreads();
      const f32x4* gx4 = reinterpret_cast<const f32x4*>(base + (size_t)sxi * 16384);
      const f32x4* gy4 = reinterpret_cast<const f32x4*>(base + (size_t)syi * 16384);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        // f32x4 number q = it*256 + tid of the slot: 32-point block q >> 5 = ((4t + rr) * 2 + hh)
        const int q = it * 256 + tid;
        const int dq = q + 2 * (q >> 5);  // + 8 floats per 32-point block
        reinterpret_cast<f32x4*>(sx)[dq] = gx4[q];
        f32x4 y = gy4[q];
        if (y_is_phi) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float s, c;
            sincos_<FAST>(y[k], s, c);
            y[k] = s;
          }
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
// points).  Both operands are data with no a-priori range, so every staged 32-point tile is normalised by the power
// of two of its own max (block-wide), multiplied into a per-tile accumulator and merged into the running sum with the
// exact inverse scale.  The fp32 slot copies sit in LDS with a 4-float skew per 32-point block: the 8 consecutive
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
__device__ __forceinline__ void frag16(const float* sl, int f, int p0, float sc, f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float v = sl[slot_index4(f, p0 + q)] * sc;
    hi[q] = (_Float16)v;
    lo[q] = (_Float16)(v - (float)hi[q]);
  }
}

template <bool FAST>
__global__ void __launch_bounds__(256)
mlp_wgrad_f16_kernel(const char* __restrict__ scratch, float* __restrict__ d_wmat, long long n_wave_tiles,
                     int tiles_per_chunk, int has_col) {
  __shared__ __attribute__((aligned(16))) float sx[WG16_SLOT_FLOATS], sy[WG16_SLOT_FLOATS];
  __shared__ __attribute__((aligned(16))) f16x8 sb[2][4][2][64];  // [hi|lo][column tile][k-step][lane]
  __shared__ float smax[2][4];
  const int m = blockIdx.y;
  if (m == 7 && !has_col) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, i = lane & 31;
  const long long t_begin = (long long)blockIdx.x * tiles_per_chunk;
  const long long t_end = min(n_wave_tiles, t_begin + tiles_per_chunk);
  f32x16 acc[4];
  acc_zero(acc);
  const int npair = (m == 7) ? 1 : 2;
  const int fo = 32 * wave + i;
  for (long long wt = t_begin; wt < t_end; ++wt) {
    const char* base = scratch + wt * (long long)(NSLOT_BWD * 16384);
    for (int pr = 0; pr < npair; ++pr) {
      int sxi, syi;
      bool y_is_phi;
      if (m == 7) { sxi = S_UV; syi = S_PHI + 7; y_is_phi = true; }
      else if (pr == 0) { sxi = S_V + m; syi = S_GB + m; y_is_phi = false; }
      else { sxi = S_U + m; syi = S_PHI + m; y_is_phi = true; }
      const f32x4* gx4 = reinterpret_cast<const f32x4*>(base + (size_t)sxi * 16384);
      const f32x4* gy4 = reinterpret_cast<const f32x4*>(base + (size_t)syi * 16384);
      f32x4 xv[4], yv[4];
      float mx = 0.f, my = 0.f;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int q = it * 256 + tid;
        xv[it] = gx4[q];
        yv[it] = gy4[q];
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (y_is_phi) {
            float sn, cs;
            sincos_<FAST>(yv[it][k], sn, cs);
            yv[it][k] = sn;
          }
          mx = fmaxf(mx, fabsf(xv[it][k]));
          my = fmaxf(my, fabsf(yv[it][k]));
        }
      }
      mx = oi::wave_max(mx);
      my = oi::wave_max(my);
      __syncthreads();  // previous tile's readers of sx / sy / sb are done
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int q = it * 256 + tid;
        const int dq = q + (q >> 5);  // + 4 floats per 32-point block
        reinterpret_cast<f32x4*>(sx)[dq] = xv[it];
        reinterpret_cast<f32x4*>(sy)[dq] = yv[it];
      }
      if (lane == 0) {
        smax[0][wave] = mx;
        smax[1][wave] = my;
      }
      __syncthreads();
      float scx, ivx, scy, ivy;
      pow2_scale_of(fmaxf(fmaxf(smax[0][0], smax[0][1]), fmaxf(smax[0][2], smax[0][3])), scx, ivx);
      pow2_scale_of(fmaxf(fmaxf(smax[1][0], smax[1][1]), fmaxf(smax[1][2], smax[1][3])), scy, ivy);
      // this wave's column tile of Y -> shared fp16 fragments
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        f16x8 bh, bl;
        frag16(sy, fo, 16 * ks + 8 * h, scy, bh, bl);
        sb[0][wave][ks][lane] = bh;
        sb[1][wave][ks][lane] = bl;
      }
      f16x8 ah[2], al[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) frag16(sx, fo, 16 * ks + 8 * h, scx, ah[ks], al[ks]);
      __syncthreads();
      f32x16 tacc[4];
      acc_zero(tacc);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f16x8 bh = sb[0][t][ks][lane], bl = sb[1][t][ks][lane];
          tacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh, tacc[t], 0, 0, 0);
          tacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl, tacc[t], 0, 0, 0);
          tacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh, tacc[t], 0, 0, 0);
        }
      }
      const float inv = ivx * ivy;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = fmaf(tacc[t][r], inv, acc[t][r]);
    }
  }
  float* dst = d_wmat + (size_t)m * C * C;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int oo = 32 * wave + (rg & 3) + 8 * (rg >> 2) + 4 * h;
      atomicAdd(dst + (size_t)oo * C + 32 * t + i, acc[t][rg]);
    }
}

template <int PREC, bool FAST>
int launch_bwd(const float* pts, const void* packed, const float* gamma, const float* beta, const float* grad_fwd,
               const float* rgb_fwd, const float* g_sdf, const float* g_grad, const float* g_rgb, float* d_small,
               float* d_wmat, float* d_gamma, float* d_beta, void* scratch, int B, long long n, hipStream_t st) {
  dim3 grid(oi::cdiv(n, TILE_PTS), B), block(256);
  auto k = mlp_bwd_sweep_kernel<PREC, FAST>;
  // per launch: the attribute is per device, and a process may drive several
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, L_TOTAL_BWD);
  hipLaunchKernelGGL(k, grid, block, L_TOTAL_BWD, st, pts, reinterpret_cast<const char*>(packed), gamma, beta, grad_fwd,
                     rgb_fwd, g_sdf, g_grad, g_rgb, d_small, d_gamma, d_beta, reinterpret_cast<char*>(scratch), n);
  int rc = oi::check_launch("oi_sdf_mlp_bwd(sweep)");
  if (rc != OI_OK) return rc;
  const long long n_wt = (long long)B * grid.x * 4;
  int chunk = (int)std::max<long long>(1, (n_wt * 8 + 2047) / 2048);  // ~2048 workgroups in total
  dim3 g2(oi::cdiv(n_wt, chunk), 8);
  if constexpr (PREC == OI_PREC_F16X3) {
    hipLaunchKernelGGL(mlp_wgrad_f16_kernel<FAST>, g2, block, 0, st, reinterpret_cast<const char*>(scratch), d_wmat,
                       n_wt, chunk, (rgb_fwd != nullptr && g_rgb != nullptr) ? 1 : 0);
  } else {
    hipLaunchKernelGGL(mlp_wgrad_kernel<FAST>, g2, block, 0, st, reinterpret_cast<const char*>(scratch), d_wmat, n_wt,
                       chunk, (rgb_fwd != nullptr && g_rgb != nullptr) ? 1 : 0);
  }
  return oi::check_launch("oi_sdf_mlp_bwd(wgrad)");
}

}  // namespace

extern "C" {

size_t oi_mlp_bwd_scratch_bytes(int B, long long n_per_elem) {
  const long long tiles = (n_per_elem + TILE_PTS - 1) / TILE_PTS;
  return (size_t)B * tiles * 4 * NSLOT_BWD * 16384;
}

int oi_mlp_bwd_small_floats(void) { return DS_TOTAL; }

int oi_sdf_mlp_bwd(const float* pts, const void* packed, const float* gamma, const float* beta, const float* grad_fwd,
                   const float* rgb_fwd, const float* g_sdf, const float* g_grad, const float* g_rgb, float* d_small,
                   float* d_wmat, float* d_gamma, float* d_beta, void* scratch, int B, long long n_per_elem, int prec,
                   int fast_trig, oi_stream_t stream) {
  OI_REQUIRE(pts && packed && gamma && beta && d_small && d_wmat && d_gamma && d_beta && scratch,
             "oi_sdf_mlp_bwd: null pointer");
  OI_REQUIRE(B > 0 && n_per_elem > 0, "oi_sdf_mlp_bwd: B=%d n=%lld", B, n_per_elem);
  OI_REQUIRE((rgb_fwd == nullptr) == (g_rgb == nullptr) || g_rgb == nullptr, "oi_sdf_mlp_bwd: g_rgb needs rgb_fwd");
  OI_REQUIRE(g_rgb == nullptr || grad_fwd != nullptr, "oi_sdf_mlp_bwd: colour backward needs the forward gradient");
  OI_REQUIRE(prec != OI_PREC_BF16X6, "oi_sdf_mlp_bwd: pass the OI_PREC_F32 image for the backward of the BF16X6 mode");
  hipStream_t st = oi::as_stream(stream);
#define OI_BWD_CASE(P)                                                                                              \
  case P:                                                                                                           \
    return fast_trig ? launch_bwd<P, true>(pts, packed, gamma, beta, grad_fwd, rgb_fwd, g_sdf, g_grad, g_rgb, d_small, \
                                           d_wmat, d_gamma, d_beta, scratch, B, n_per_elem, st)                     \
                     : launch_bwd<P, false>(pts, packed, gamma, beta, grad_fwd, rgb_fwd, g_sdf, g_grad, g_rgb, d_small, \
                                            d_wmat, d_gamma, d_beta, scratch, B, n_per_elem, st);
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
