// FiLM-SIREN SDF / albedo MLP for gfx950 (MI355X, CDNA4).
//
// Replaces:
//   ShapeNetwork.style / .forward / .sdf / .gradient, ColorNetwork.forward
//     (reference src/models/fields.py:15-21, 49-77, 89-101, 104-122)
//   FiLMSiren.forward / LinearLayer.forward (src/third_party/stylesdf/volume_renderer.py:27-30, 50-61)
//
// Kernel design (see DESIGN.md "MLP kernel"):
//  * One wavefront owns 32 points for the whole network.  The contraction is computed
//    TRANSPOSED, out^T = W . act^T, so that the point index sits on the MFMA column (= lane & 31)
//    and never moves: the C/D fragment a lane receives from layer l (its 64 of the 128 output
//    features) is used AS IS as the B fragment of layer l+1.  The K-order this implies
//    (feature f(q,h) = 32*(q>>4) + 8*((q>>2)&3) + 4*h + (q&3) for accumulator slot q in lane-half h)
//    is absorbed into a one-time re-ordering of the weights (oi_mlp_pack_weights), so activations
//    never touch LDS or HBM between layers.
//  * Weights (the A operand) are streamed layer by layer through LDS as a lane-linear image read
//    with ds_read_b128; a 256-thread workgroup (4 waves, 128 points) shares each image.
//  * d sdf/dx is the analytic reverse sweep g_l = W_l^T (g_{l+1} * gamma_l * cos(phi_l)); the
//    per-layer gamma*cos(phi) fragments are parked in a global scratch (L2/MALL resident between
//    the write and the read by the same wave) instead of re-running the network as autograd does.
//  * MFMA operand precision is a template mode (oi_precision); accumulation, the FiLM phase and
//    sin/cos are always fp32.
#include <cstdlib>

#include <algorithm>

#include "mlp_common.h"
#include "f3_blob.h"

namespace {

using namespace oimlp;

// ------------------------------------------------------------------------------------------
// a1 + a2: style MLP + FiLM parameters.  grid (B, max(NL, 1)): one block of 128 threads per (batch element, FiLM
// layer); every block re-derives w from z itself (3 x 64 x 64 MACs) instead of waiting for another block.
// ------------------------------------------------------------------------------------------
__global__ void film_params_kernel(const float* __restrict__ style_w, const float* __restrict__ style_b,
                                   const float* __restrict__ z, float* __restrict__ w_out,
                                   const float* __restrict__ gw, const float* __restrict__ gb,
                                   const float* __restrict__ bw, const float* __restrict__ bb,
                                   float* __restrict__ gamma, float* __restrict__ beta, int NL) {
  __shared__ float h[2][64];
  const int e = blockIdx.x, t = threadIdx.x;
  if (z != nullptr) {
    if (t < 64) h[0][t] = z[e * 64 + t];
    __syncthreads();
    int cur = 0;
    for (int l = 0; l < 3; ++l) {
      if (t < 64) {
        const float* wr = style_w + (l * 64 + t) * 64;
        float acc = 0.f;
#pragma unroll 8
        for (int k = 0; k < 64; ++k) acc = fmaf(h[cur][k], wr[k], acc);
        acc += style_b[l * 64 + t];
        h[cur ^ 1][t] = acc > 0.f ? acc : 0.2f * acc;  // fused_bias_act: lrelu(x+b, 0.2) * scale(=1)
      }
      __syncthreads();
      cur ^= 1;
    }
    if (t < 64 && blockIdx.y == 0) w_out[e * 64 + t] = h[cur][t];
    if (cur != 0) {
      if (t < 64) h[0][t] = h[1][t];
    }
    __syncthreads();
  } else {
    if (t < 64) h[0][t] = w_out[e * 64 + t];
    __syncthreads();
  }
  const int l = blockIdx.y;
  if (l < NL) {
    const float* g = gw + ((size_t)l * C + t) * 64;
    const float* b = bw + ((size_t)l * C + t) * 64;
    float ag = 0.f, ab = 0.f;
#pragma unroll 8
    for (int k = 0; k < 64; ++k) {
      ag = fmaf(h[0][k], g[k], ag);
      ab = fmaf(h[0][k], b[k], ab);
    }
    gamma[((size_t)e * NL + l) * C + t] = 15.0f * (ag + gb[l * C + t]) + 30.0f;
    beta[((size_t)e * NL + l) * C + t] = 0.25f * (ab + bb[l * C + t]) + 0.0f;
  }
}

// a1 + a2 backward.  d_gamma / d_beta [B][NL][128] (from the MLP backward) ->
//   d_gw[l][f][k] = 15 sum_b d_gamma[b][l][f] w[b][k],   d_gb[l][f] = 15 sum_b d_gamma[b][l][f]
//   d_bw[l][f][k] = .25 sum_b d_beta[b][l][f] w[b][k],   d_bb[l][f] = .25 sum_b d_beta[b][l][f]
//   d_w[b][k]    += sum_f 15 d_gamma[b][l][f] gw[l][f][k] + .25 d_beta[b][l][f] bw[l][f][k]     (atomics over l)
// grid (NL), block 256.  The layer's two 128 x 64 head matrices are staged in LDS with coalesced loads; outer
// products are written with coalesced stores; the d_w contraction runs over LDS columns (conflict-free) and is
// reduced over four feature quarters.  All outputs are ASSIGNED except d_w (accumulated: the caller zeroes it or
// passes the upstream gradient of w).
__global__ void __launch_bounds__(256)
film_heads_bwd_kernel(const float* __restrict__ d_gamma, const float* __restrict__ d_beta,
                      const float* __restrict__ w, const float* __restrict__ gw, const float* __restrict__ bw,
                      float* __restrict__ d_gw, float* __restrict__ d_gb, float* __restrict__ d_bw,
                      float* __restrict__ d_bb, float* __restrict__ d_w, int B, int NL) {
  __shared__ float sg[C * 64], sb[C * 64];      // gw[l], bw[l]: [f][k]
  __shared__ float dg[C], db[C], ws[64];
  __shared__ float part[4][64];
  const int l = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < C * 64; i += 256) {
    sg[i] = gw[(size_t)l * C * 64 + i];
    sb[i] = bw[(size_t)l * C * 64 + i];
  }
  float ag[32], ab[32];                         // this thread's 32 outputs of each outer product: index i = tid + 256 j
#pragma unroll
  for (int j = 0; j < 32; ++j) { ag[j] = 0.f; ab[j] = 0.f; }
  float sgb = 0.f, sbb = 0.f;
  for (int b = 0; b < B; ++b) {
    __syncthreads();
    if (tid < C) {
      dg[tid] = 15.0f * d_gamma[((size_t)b * NL + l) * C + tid];
      db[tid] = 0.25f * d_beta[((size_t)b * NL + l) * C + tid];
    } else if (tid < C + 64) {
      ws[tid - C] = w[b * 64 + tid - C];
    }
    __syncthreads();
    if (tid < C) { sgb += dg[tid]; sbb += db[tid]; }
    const int k = tid & 63;                     // i = tid + 256 j  ->  f = (tid >> 6) + 4 j, k = tid & 63
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int f = (tid >> 6) + 4 * j;
      ag[j] = fmaf(dg[f], ws[k], ag[j]);
      ab[j] = fmaf(db[f], ws[k], ab[j]);
    }
    // d_w[b][k]: quarter q = tid >> 6 sums features 32 q .. 32 q + 31
    float acc = 0.f;
    const int q = tid >> 6;
#pragma unroll 8
    for (int f = 32 * q; f < 32 * q + 32; ++f) acc = fmaf(dg[f], sg[f * 64 + k], fmaf(db[f], sb[f * 64 + k], acc));
    part[q][k] = acc;
    __syncthreads();
    if (tid < 64) atomicAdd(d_w + b * 64 + tid, part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid]);
  }
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    d_gw[(size_t)l * C * 64 + tid + 256 * j] = ag[j];
    d_bw[(size_t)l * C * 64 + tid + 256 * j] = ab[j];
  }
  if (tid < C) {
    d_gb[l * C + tid] = sgb;
    d_bb[l * C + tid] = sbb;
  }
}

// Style MLP backward (3 x [64 -> 64, lrelu 0.2]): recompute the activations from z, back-propagate d_w.
// grid (B), block 256; parameter gradients accumulate over the batch with atomics (caller zeroes them).
// The three 16 KiB matrices are staged in LDS with coalesced loads (rows padded to 65 floats: a lane reads "its" row in
// the forward products and "its" column in the backward ones, both conflict-free), and the 64 x 64 outer products are
// written row by row, 64 consecutive floats per wave (round 2: one 64-thread block per element, every lane walking its
// own weight row in global memory and issuing 64 strided atomics per layer: 35 us; now 16).
__global__ void __launch_bounds__(256)
style_bwd_kernel(const float* __restrict__ style_w, const float* __restrict__ style_b,
                                 const float* __restrict__ z, const float* __restrict__ d_w,
                                 float* __restrict__ d_style_w, float* __restrict__ d_style_b,
                                 float* __restrict__ d_z) {
  __shared__ float W[3][64][65];
  __shared__ float h[4][64];
  __shared__ float pre_pos[3][64];
  __shared__ float dh[2][64];
  const int e = blockIdx.x, tid = threadIdx.x, t = tid & 63, part = tid >> 6;
  for (int i = tid; i < 3 * 64 * 64; i += 256) W[i >> 12][(i >> 6) & 63][i & 63] = style_w[i];
  if (tid < 64) h[0][t] = z[e * 64 + t];
  __syncthreads();
  for (int l = 0; l < 3; ++l) {
    if (tid < 64) {
      float acc = 0.f;
#pragma unroll 8
      for (int k = 0; k < 64; ++k) acc = fmaf(h[l][k], W[l][t][k], acc);
      acc += style_b[l * 64 + t];
      pre_pos[l][t] = acc > 0.f ? 1.0f : 0.2f;
      h[l + 1][t] = acc > 0.f ? acc : 0.2f * acc;
    }
    __syncthreads();
  }
  if (tid < 64) dh[0][t] = d_w[e * 64 + t];
  __syncthreads();
  int cur = 0;
  for (int l = 2; l >= 0; --l) {
    if (tid < 64) {
      const float dp = dh[cur][t] * pre_pos[l][t];  // gradient at the pre-activation of unit t
      atomicAdd(d_style_b + l * 64 + t, dp);
      dh[cur][t] = dp;  // reuse as the dp vector
    }
    __syncthreads();
    // d W_l[o][k] += dp[o] h_l[k]: wave `part` takes rows o = part, part + 4, ..: 64 consecutive addresses per instruction
    for (int o = part; o < 64; o += 4) atomicAdd(d_style_w + (l * 64 + o) * 64 + t, dh[cur][o] * h[l][t]);
    if (tid < 64) {
      float acc = 0.f;
#pragma unroll 8
      for (int o = 0; o < 64; ++o) acc = fmaf(dh[cur][o], W[l][o][t], acc);
      dh[cur ^ 1][t] = acc;
    }
    __syncthreads();
    cur ^= 1;
  }
  if (d_z != nullptr && tid < 64) d_z[e * 64 + t] = dh[cur][t];
}

// ------------------------------------------------------------------------------------------
// weight pre-pack
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float mat_elem(const float* wh, const float* wv, int m, int row, int k) {
  // m: 0..6 forward layer 1..7 -> W[row][k]; 7..13 transposed layer 1..7 -> W[k][row]; 14 colour -> Wv[row][k];
  // 15 colour transposed -> Wv[k][row] (first 128 input columns)
  if (m < 7) return wh[((size_t)m * C + row) * C + k];
  if (m < 14) return wh[((size_t)(m - 7) * C + k) * C + row];
  if (m == 14) return wv[(size_t)row * (C + 3) + k];
  return wv[(size_t)k * (C + 3) + row];
}

// F16X3: SIREN weights are O(1e-2), so the lo limb of an unscaled fp16 split would be subnormal (absolute step
// 2^-24, i.e. only ~2^-17 relative -- no better than a bf16 split).  Every image is therefore scaled by the power
// of two that puts max |W_m| in [2^13, 2^14); the forward kernel folds 2^-k_m into the FiLM gamma it stages (and
// thereby into the parked gamma*cos(phi)), which cancels against the scaled transposed image in the reverse sweep:
// exact, and free.  Returns the scale; *inv gets 2^-k_m.  Must be called by all 256 threads of the block.
template <int PREC>
__device__ float image_scale(const float* wh, const float* wv, int m, float* red, float* inv, float* bad,
                             float* rowsum = nullptr) {
  // rowsum (LDS, 128 zeroed floats, or null): also receives sum_k |M[row][k]| (the same pass over the matrix)
  *bad = 0.f;
  if (PREC != OI_PREC_F16X3) {
    *inv = 1.f;
    return 1.f;
  }
  float mx = 0.f;
  for (int i = threadIdx.x; i < C * C; i += 256) {
    const float a = fabsf(mat_elem(wh, wv, m, i >> 7, i & 127));
    mx = a <= 3.0e38f ? fmaxf(mx, a) : __builtin_inff();  // fmaxf would drop a NaN: non-finite weights must surface
    if (rowsum != nullptr) atomicAdd(rowsum + (i >> 7), a);
  }
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  mx = red[0];
  __syncthreads();
  *bad = mx <= 3.0e38f ? 0.f : 1.f;
  int eb = (__builtin_bit_cast(int, mx) >> 23) & 0xff;
  eb = eb < 14 ? 14 : (eb > 254 ? 254 : eb);
  *inv = __builtin_bit_cast(float, (eb - 13) << 23);
  return __builtin_bit_cast(float, (267 - eb) << 23);
}

constexpr int PK_PER = 8;  // image elements per thread of pack_weights_kernel

template <int PREC>
__global__ void pack_weights_kernel(const float* __restrict__ w0, const float* __restrict__ b0,
                                    const float* __restrict__ wh, const float* __restrict__ bh,
                                    const float* __restrict__ wsig, const float* __restrict__ bsig,
                                    const float* __restrict__ wv, const float* __restrict__ bv,
                                    const float* __restrict__ wrgb, const float* __restrict__ brgb,
                                    char* __restrict__ packed) {
  float* hdr = reinterpret_cast<float*>(packed);
  __shared__ float red[256];
  if (blockIdx.y == NMAT) {  // header (the image scales at H_WSCALE are written by block 0 of every matrix)
   for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < H_FLOATS; idx += gridDim.x * blockDim.x) {
    if ((idx >= H_WSCALE && idx < H_WSCALE + NMAT) || (idx >= H_BOUND && idx < H_BOUND + NMAT) ||
        (idx >= H_STATUS && idx < H_STATUS + NMAT))
      continue;
    float v = 0.f;
    if (idx < H_SIG) {
      int f = idx >> 2, j = idx & 3;
      v = j < 3 ? w0[f * 3 + j] : 0.f;
    } else if (idx < H_TABV) {
      int i = idx - H_SIG;
      v = i < C ? wsig[i] : (i == C ? bsig[0] : 0.f);
    } else if (idx < H_RGB) {
      int i = idx - H_TABV, f = i >> 2, j = i & 3;
      v = j < 3 ? wv[(size_t)f * (C + 3) + C + j] : 0.f;
    } else if (idx < H_TABS_END) {
      int i = idx - H_RGB;
      v = i < 3 * C ? wrgb[i] : (i < 3 * C + 3 ? brgb[i - 3 * C] : 0.f);
    } else if (idx < H_BIAS + 9 * C) {
      int i = idx - H_BIAS, l = i / C, f = i % C;
      v = l == 0 ? b0[f] : (l < 8 ? bh[(l - 1) * C + f] : bv[f]);
    }
    hdr[idx] = v;
   }
    return;
  }
  const int m = blockIdx.y;
  float inv, bad;
  // (every block of a matrix takes the maximum over the whole matrix: PK_PER elements per thread keep that to 8 passes
  // per matrix instead of 64, and the row sums of the growth bound ride on that pass -- the kernel runs once per optimiser step:
  // 54 -> 41 us; 2 / 4 / 16 / 64 elements per thread: 80 / 45 / 49 / 75 us)
  __shared__ float rowsum[C];
  if (threadIdx.x < C) rowsum[threadIdx.x] = 0.f;
  __syncthreads();
  const float wscale = image_scale<PREC>(wh, wv, m, red, &inv, &bad, blockIdx.x == 0 ? rowsum : nullptr);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    hdr[H_WSCALE + m] = inv;  // 2^-k_m for the kernels (1 in the unscaled modes)
    // the scaled image peaks in [2^13, 2^14): inside fp16 by construction for every finite weight; only inf / NaN is out
    hdr[H_STATUS + m] = bad;
  }
  if (blockIdx.x == 0) {
    // largest absolute row sum of the scaled image (induced infinity norm): an a-priori bound on the growth of a vector
    // through this layer product, used by mlp_fwd3.hip to pick the fp16 scale of an adjoint vector BEFORE it is complete
    float rs = 0.f;
    if (threadIdx.x < C) {
      if (PREC == OI_PREC_F16X3) rs = rowsum[threadIdx.x];   // (summed by the scale pass above)
      else for (int k = 0; k < C; ++k) rs += fabsf(mat_elem(wh, wv, m, threadIdx.x, k));
    }
    red[threadIdx.x] = rs;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
      __syncthreads();
    }
    if (threadIdx.x == 0) hdr[H_BOUND + m] = red[0] * wscale;
    __syncthreads();
  }
  for (int part = 0; part < PK_PER; ++part) {
  const int idx = (blockIdx.x * PK_PER + part) * blockDim.x + threadIdx.x;
  if (idx >= C * C) return;
  if (m < 7 || m == 14)  // plain fp32 copy [out][in] for the backward's FiLM-scale identity
    reinterpret_cast<float*>(packed + plain_off(PREC))[(size_t)(m < 7 ? m : 7) * C * C + idx] =
        mat_elem(wh, wv, m, idx >> 7, idx & 127);
  char* base = packed + H_BYTES + (size_t)m * layer_bytes(PREC);
  if (PREC == OI_PREC_F32) {
    // image [t(4)][g(16)][lane(64)][k(4)] fp32, q = 4g + k
    const int k = idx & 3, lane = (idx >> 2) & 63, g = (idx >> 8) & 15, t = idx >> 12;
    const int q = 4 * g + k, h = lane >> 5, i = lane & 31;
    reinterpret_cast<float*>(base)[idx] = mat_elem(wh, wv, m, 32 * t + i, feat_of(q, h));
  } else {
    // image [t(4)][s(8)][lane(64)][i'(8)] bf16 (hi; lo image 32 KiB later), q = 8s + i'
    const int ip = idx & 7, lane = (idx >> 3) & 63, s = (idx >> 9) & 7, t = idx >> 12;
    const int q = 8 * s + ip, h = lane >> 5, i = lane & 31;
    const float v = mat_elem(wh, wv, m, 32 * t + i, feat_of(q, h)) * wscale;
    if (PREC == OI_PREC_F16X3) {  // fp16 hi + lo limbs (22 mantissa bits), same image geometry as BF16X3
      const _Float16 fh = (_Float16)v;
      reinterpret_cast<_Float16*>(base)[idx] = fh;
      reinterpret_cast<_Float16*>(base + 32768)[idx] = (_Float16)(v - (float)fh);
      continue;
    }
    const __bf16 hi = (__bf16)v;
    reinterpret_cast<__bf16*>(base)[idx] = hi;
    if (PREC == OI_PREC_BF16X3) reinterpret_cast<__bf16*>(base + 32768)[idx] = (__bf16)(v - (float)hi);
    if (PREC == OI_PREC_BF16X6) {  // exact 3-way split: hi + mid + lo == v (24-bit mantissa)
      const float r1 = v - (float)hi;
      const __bf16 mid = (__bf16)r1;
      reinterpret_cast<__bf16*>(base + 32768)[idx] = mid;
      reinterpret_cast<__bf16*>(base + 65536)[idx] = (__bf16)(r1 - (float)mid);
    }
  }
  }
}

// ------------------------------------------------------------------------------------------
// Forward kernel, v2: 8 wavefronts (256 points) per workgroup, the 15 layer images stream through a
// DOUBLE-BUFFERED LDS ring filled by global_load_lds (LDS-DMA, no VGPR round trip): the image of layer
// i+1 is in flight while layer i's MFMAs run, and there is ONE barrier per layer.  FiLM vectors of all
// 9 layers are staged once.  (v1 staged synchronously with two barriers per layer: 56 % of the fp32
// MFMA peak, matrix pipe 64 % busy -- profiles/r1_*.)
// ------------------------------------------------------------------------------------------

// REV: the FiLM rows were staged in REVOLUTIONS (gamma / 2pi, beta' / 2pi; sdf-only F16X3 pass): the phase needs one
// v_fract (exact, keeps v_sin inside its [-256, 256] domain) instead of the four-operation reduction of sincos_, and no
// cosine exists in that pass -- the form the register-resident kernel (mlp_fwd3.hip) uses.
#ifndef OI_SDF_REV
#define OI_SDF_REV 1
#endif
template <bool FAST>
__device__ __forceinline__ float sin_rev(float phi_rev) {
  return __builtin_amdgcn_sinf(FAST ? phi_rev : __builtin_amdgcn_fractf(phi_rev));
}
template <bool FAST, bool FULL, int SRC, class SCR, bool REV = false>
__device__ __forceinline__ void film_sin2(const char* lds, const LaneOff& o, const LayOff& y, const f32x16 (&acc)[4],
                                          float (&act)[64], const SCR& ws, int slot, int tab_imm, float vx,
                                          float vy, float vz) {
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int t = g >> 2, rr = g & 3;
    const f32x4 gm = lds_f4(lds, grp_f0(g) * 4, y.f16);
    const f32x4 bt = lds_f4(lds, (C + grp_f0(g)) * 4, y.f16);
    f32x4 cv;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float u;
      if constexpr (SRC == 0) {
        u = acc[t][4 * rr + k];
      } else {
        const f32x4 w = lds_f4(lds, V2_TABS + tab_imm * 4 + (grp_f0(g) + k) * 16, o.h64);
        const float d = fmaf(vz, w[2], fmaf(vy, w[1], vx * w[0]));
        u = SRC == 1 ? d : acc[t][4 * rr + k] + d;
      }
      const float phi = fmaf(gm[k], u, bt[k]);
      if constexpr (REV) {
        act[4 * g + k] = sin_rev<FAST>(phi);
      } else {
        float s, c;
        sincos_<FAST>(phi, s, c);
        act[4 * g + k] = s;
        cv[k] = gm[k] * c;
      }
    }
    if constexpr (FULL) ws.store(slot, g, o, cv);
    // two groups (8 values = 4 packed sin/cos chains) per scheduling window: one group alone leaves the dependent
    // v_pk_fma chains latency-bound
    if ((g & 1) == 1) __builtin_amdgcn_sched_barrier(0);
  }
}

// F16X3 forward layer with the FiLM / sin phase software-pipelined INTO the GEMM (one wave, no extra point tile):
// the product is formed output-block by output-block (t outer, k inner) instead of k-step by k-step, so block t - 1 is
// complete while block t's 24 MFMAs are in flight -- its 16 FiLM / sin values per lane are issued two at a time between
// those MFMAs (independent instructions of the same wave run beside its own in-flight MFMAs); only block 3's FiLM
// phase is exposed.  The B operand is split into its fp16 limbs once per layer (all 8 k-steps stay in registers).
template <bool FAST, bool FULL, bool REV, class SCR>
__device__ __forceinline__ void layer_fwd_pipelined(const char* lds, const LaneOff& o, const LayOff& y,
                                                    f32x16 (&acc)[4], float (&act)[64], const SCR& ws, int slot) {
  f16x8 bh[8], bl[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) split8_pairs(&act[8 * s], bh[s], bl[s]);
  f32x4 gm, bt, cv;
  // one value (accumulator slot 2s + j of block tb) of the FiLM / sin phase; j = 0, 1
  auto film_val = [&](int tb, int s, int j) {
    const int g = tb * 4 + (s >> 1), k = 2 * (s & 1) + j;
    if (k == 0) {
      gm = lds_f4(lds, grp_f0(g) * 4, y.f16);
      bt = lds_f4(lds, (C + grp_f0(g)) * 4, y.f16);
    }
    const float phi = fmaf(gm[k], acc[tb][2 * s + j], bt[k]);
    if constexpr (REV) {
      act[4 * g + k] = sin_rev<FAST>(phi);
    } else {
      float sn, cs;
      sincos_<FAST>(phi, sn, cs);
      act[4 * g + k] = sn;
      cv[k] = gm[k] * cs;
    }
    if constexpr (FULL) {
      if (k == 3) ws.store(slot, g, o, cv);
    }
  };
  auto film_part = [&](int tb, int s) {
    film_val(tb, s, 0);
    film_val(tb, s, 1);
  };
  f32x4 ah = lds_f4(lds, 0, y.wl), al = lds_f4(lds, 0, y.wh), ahn = ah, aln = al;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int nxt = t * 8 + s + 1;
      if (nxt < 32) {
        ahn = lds_f4(lds, nxt * 1024, y.wl);
        aln = lds_f4(lds, nxt * 1024, y.wh);
      }
      const f16x8 wh = __builtin_bit_cast(f16x8, ah);
      const f16x8 wl = __builtin_bit_cast(f16x8, al);
      // a wave issues in order and the matrix pipe takes one MFMA at a time: the VALU work has to sit BETWEEN the MFMAs
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh[s], acc[t], 0, 0, 0);
      if (t > 0) {
        film_val(t - 1, s, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl[s], acc[t], 0, 0, 0);
      if (t > 0) {
        film_val(t - 1, s, 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh[s], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ah = ahn;
      al = aln;
    }
  }
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    film_part(3, s);
    if (s & 1) __builtin_amdgcn_sched_barrier(0);
  }
}

// the layer bias lives in beta' (phi = gamma * (W a + b) + beta = gamma * (W a) + beta'), accumulators start at 0
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[4]) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}

// 1: F16X3 forward layers run layer_fwd_pipelined (FiLM / sin of block t-1 between block t's MFMAs): same-box A/B
// sdf-only pass 1.61 -> 1.57 ms per 2^21 points, full kernel 1.040 -> 1.026 ms; 0: gemm_layer2 + film_sin2
#ifndef OI_PIPE_FWD
#define OI_PIPE_FWD 1
#endif
#ifdef OI_PROF
__device__ unsigned long long oi_prof[16];
#define PROF_T(i)                                                  \
  do {                                                             \
    const unsigned long long t_ = __builtin_readcyclecounter();    \
    pacc[i] += t_ - tprev;                                         \
    tprev = t_;                                                    \
  } while (0)
#else
#define PROF_T(i)
#endif

template <int PREC, bool FAST, bool FULL>
__global__ void __launch_bounds__(64 * v2_waves(PREC, FULL), 2)
sdf_mlp_kernel(const float* __restrict__ pts, const char* __restrict__ packed, const float* __restrict__ gamma,
               const float* __restrict__ beta, float* __restrict__ sdf_out, float* __restrict__ grad_out,
               float* __restrict__ rgb_out, float* __restrict__ feat_out, char* __restrict__ scratch,
               long long n_per_elem) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, j = lane & 31;
  const int e = blockIdx.y;
  const float* hdr = reinterpret_cast<const float*>(packed);
  const char* mats = packed + H_BYTES;
  constexpr int LB = layer_bytes(PREC);
  constexpr bool RING2 = v2_two_slots(PREC, FULL);
  constexpr int NWV = v2_waves(PREC, FULL);
  constexpr bool REV = OI_SDF_REV && PREC == OI_PREC_F16X3 && !FULL;  // FiLM rows in revolutions (see film_sin2)
  // double-buffered ring: the next image is requested at the START of a layer into the other slot.
  // single slot (BF16X6): it is requested right AFTER the layer's MFMAs, behind a barrier, and lands while the
  // FiLM/sin VALU phase runs.
  LaneOff o;
  o.h16 = 16 * h;
  o.h64 = 64 * h;
  o.l16 = 16 * lane;
  o.l16hi = 16 * lane + 32768;
  asm volatile("" : "+v"(o.h16), "+v"(o.h64), "+v"(o.l16), "+v"(o.l16hi));

  const __amdgpu_buffer_rsrc_t img_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(mats), 0, NMAT * LB, 0x00020000);
  auto stage_early = [&](int image, int slot) {
    if constexpr (RING2) prefetch_image<PREC, NWV>(lds, img_rs, image * LB, slot, wave, o.l16);
  };
  auto stage_late = [&](int image) {
    if constexpr (!RING2) {
      __syncthreads();
      prefetch_image<PREC, NWV>(lds, img_rs, image * LB, 0, wave, o.l16);
    }
  };

  constexpr int NW = NWV, NT = 64 * NW;
  // PERSISTENT workgroups for the sdf-only passes (round 5; launch_mlp_variant sizes the grid): a workgroup walks tiles
  // blockIdx.x, + gridDim.x, ... of its batch element -- tables and FiLM rows are staged once instead of once per tile, the
  // last layer of a tile requests image 0 of the next one (7 images on a two-slot ring: `rb` swaps the slots per tile).
  constexpr bool PERSIST = !FULL && RING2;
  const int ntiles = (int)((n_per_elem + NW * WAVE_PTS - 1) / (NW * WAVE_PTS));
  int tile = blockIdx.x, rb = 0;
  bool valid;
  long long pt;
  auto set_point = [&](int t_) {
    const long long local = (long long)t_ * (NW * WAVE_PTS) + wave * WAVE_PTS + j;
    valid = local < n_per_elem;
    pt = (long long)e * n_per_elem + (valid ? local : n_per_elem - 1);
  };
  set_point(tile);

  constexpr bool HALF_SCR = PREC == OI_PREC_BF16;
  constexpr int SLOT_B = HALF_SCR ? 8192 : 16384;
  FwdScratch<HALF_SCR> ws;
  {
    const long long wt = ((long long)e * gridDim.x + blockIdx.x) * NW + wave;
    char* wbase = FULL ? scratch + wt * (long long)(NSLOT * SLOT_B) : nullptr;
    ws.rs = __builtin_amdgcn_make_buffer_rsrc(wbase, 0, FULL ? NSLOT * SLOT_B : 0, 0x00020000);
  }

  // image sequence: i = 0..6 forward layers 1..7 (mats 0..6), i = 7..13 transposed layers 7..1 (mats 13..7),
  // i = 14 colour head (mat 14); image i lives in ring slot i & 1.
  prefetch_image<PREC, NWV>(lds, img_rs, 0, 0, wave, o.l16);
  {  // small tables + FiLM rows of all 9 layers (gamma | beta | bias), once
    float* tabs = reinterpret_cast<float*>(lds + V2_TABS);
    for (int i = tid; i < H_TABS_END; i += NT) tabs[i] = hdr[i];
    float* film = reinterpret_cast<float*>(lds + V2_FILM);
    for (int i = tid; i < 9 * C; i += NT) {
      const int l = i / C, f = i % C;
      const float gm = gamma[((size_t)e * 9 + l) * C + f];
      // image scale of the layer's MFMA operand folded into the multiplier of u (layer 0 runs on the VALU)
      const float ws = l == 0 ? 1.f : hdr[H_WSCALE + (l < NL_SDF ? l - 1 : 14)];
      constexpr float TO_REV = REV ? 0.15915494309189533577f : 1.f;
      film[l * 256 + f] = gm * ws * TO_REV;
      film[l * 256 + C + f] = fmaf(gm, hdr[H_BIAS + l * C + f], beta[((size_t)e * 9 + l) * C + f]) * TO_REV;
    }
  }
  float px = pts[pt * 3 + 0], py = pts[pt * 3 + 1], pz = pts[pt * 3 + 2];
  __syncthreads();  // tables visible (image 0 still in flight)

  float act[64];
  f32x16 acc[4];
#ifdef OI_PROF
  unsigned long long pacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_readcyclecounter();
  const unsigned long long tstart = tprev;
#endif
  for (;;) {  // ---- one tile per trip (exactly one unless PERSIST)
  const int ntile = tile + (int)gridDim.x;
  float nx = 0.f, ny = 0.f, nz = 0.f;
  if constexpr (PERSIST) {  // the next tile's point, used a tile later (a clamped re-read of this one on the last trip)
    set_point(ntile < ntiles ? ntile : tile);
    nx = pts[pt * 3 + 0], ny = pts[pt * 3 + 1], nz = pts[pt * 3 + 2];
    set_point(tile);
  }

  // ---- layer 0 (K = 3) on the VALU, overlapping the first image's DMA
  {
    const LayOff y = lay_off<PREC>(o, 0, 0);
    film_sin2<FAST, FULL, 1, FwdScratch<HALF_SCR>, REV>(lds, o, y, acc, act, ws, 0, H_TAB0, px, py, pz);
  }
  PROF_T(0);
  ring_sync();
  PROF_T(4);

  // ---- layers 1..7 on MFMA
  for (int l = 1; l < NL_SDF; ++l) {
    const int i = l - 1;
    // next image: forward layer l+1, or the first transposed image (layer 7) / nothing for the sdf-only variant
    const int next = (l < NL_SDF - 1) ? l : (FULL ? 13 : (PERSIST ? 0 : -1));  // image index, -1: none; PERSIST: the next tile's first
    if (next >= 0) stage_early(next, ((i + 1) & 1) ^ rb);
    const LayOff y = lay_off<PREC>(o, RING2 ? ((i & 1) ^ rb) : 0, l);
#if OI_PIPE_FWD
    if constexpr (PREC == OI_PREC_F16X3 && RING2) {
      layer_fwd_pipelined<FAST, FULL, REV>(lds, o, y, acc, act, ws, l);
      PROF_T(1);
    } else
#endif
    {
      zero_acc(acc);
      gemm_layer2<PREC>(lds, y, act, acc);
      PROF_T(1);
      if (next >= 0) stage_late(next);
      PROF_T(2);
      film_sin2<FAST, FULL, 0, FwdScratch<HALF_SCR>, REV>(lds, o, y, acc, act, ws, l, 0, 0.f, 0.f, 0.f);
    }
    PROF_T(3);
    ring_sync();
    PROF_T(4);
  }

  // ---- sdf = a8 . wsig + bsig   (fields.py:68; LinearLayer std_init=1, bias_init=0)
  float sdf_v;
  {
    float part = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const f32x4 w = lds_f4(lds, V2_TABS + (H_SIG + grp_f0(g)) * 4, o.h16);
#pragma unroll
      for (int k = 0; k < 4; ++k) part = fmaf(act[4 * g + k], w[k], part);
    }
    part += __shfl_xor(part, 32, 64);
    sdf_v = part + *reinterpret_cast<const float*>(lds + V2_TABS + (H_SIG + C) * 4);
  }
  if (valid && h == 0) sdf_out[pt] = sdf_v;

  if (feat_out != nullptr && valid) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = act[4 * g + k];
      *reinterpret_cast<f32x4*>(feat_out + pt * C + grp_f0(g) + 4 * h) = v;
    }
  }

  if (!PERSIST || ntile >= ntiles) break;
  tile = ntile;
  rb ^= 1;
  set_point(tile);
  px = nx, py = ny, pz = nz;
  }  // tiles

  if constexpr (FULL) {
    // park the features (slot 8) and start the reverse sweep with g8 = wsig
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = act[4 * g + k];
      ws.store(8, g, o, v);
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const f32x4 w = lds_f4(lds, V2_TABS + (H_SIG + grp_f0(g)) * 4, o.h16);
#pragma unroll
      for (int k = 0; k < 4; ++k) act[4 * g + k] = w[k];
    }
    float run = 1.f;
    // gamma * cos(phi) of the layer about to be swept: requested ONE GEMM AHEAD (inside the previous layer's MFMA loop,
    // into registers the consumed B operand has just freed) so that the HBM latency hides under the MFMAs instead of
    // stalling the top of every layer
    f32x4 cn[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) cn[g] = ws.load(NL_SDF - 1, g, o);
    for (int l = NL_SDF - 1; l >= 1; --l) {
      const int i = 14 - l;  // image index of transposed layer l
#pragma unroll
      for (int g = 0; g < 16; ++g) {
#pragma unroll
        for (int k = 0; k < 4; ++k) act[4 * g + k] *= cn[g][k];
      }
      if constexpr (PREC == OI_PREC_F16X3) {
        // adjoints have no a-priori range: bring this point's vector (its 128 entries live in lanes j and j+32)
        // to max |.| in [2^13, 2^14) with an exact power-of-two scale.  The scale is NOT undone layer by layer: `run`
        // carries the product of the inverse scales (true vector = act * run) and multiplies the final gradient once.
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < 64; k += 2) m = fmaxf(m, fmaxf(fabsf(act[k]), fabsf(act[k + 1])));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        int eb = (__builtin_bit_cast(int, m) >> 23) & 0xff;
        eb = eb < 14 ? 14 : (eb > 254 ? 254 : eb);
        float sc = __builtin_bit_cast(float, (267 - eb) << 23);  // 2^(13 - (eb - 127))
        // fence tied to sc (= to every c * g product): keeps the GEMM's A-fragment ds_reads from being scheduled
        // into the scratch-load phase while the 64 c registers are still live (140 spilled VGPRs otherwise)
        asm volatile("" : "+v"(sc) : : "memory");
        run *= __builtin_bit_cast(float, (eb - 13) << 23);             // 1 / sc
#pragma unroll
        for (int k = 0; k < 64; ++k) act[k] *= sc;
      }
      // next image after the scratch loads have been consumed (an in-flight LDS-DMA would otherwise be
      // drained by the vmcnt wait hipcc places in front of the first use of an ordinary load)
      const int next = (l > 1) ? 7 + l - 2 : (rgb_out != nullptr ? 14 : -1);
      if (next >= 0) stage_early(next, (i + 1) & 1);
      const LayOff y = lay_off<PREC>(o, RING2 ? (i & 1) : 0, l);
      zero_acc(acc);
      PROF_T(5);
      gemm_layer2<PREC>(lds, y, act, acc, [&](int s) {
        cn[2 * s] = ws.load(l - 1, 2 * s, o);
        cn[2 * s + 1] = ws.load(l - 1, 2 * s + 1, o);
      });
      PROF_T(6);
      if (next >= 0) stage_late(next);
      PROF_T(7);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) act[16 * t + r] = acc[t][r];
      ring_sync();
      PROF_T(8);
    }
    // layer 0: grad = W0^T (g1 * c0)
    float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const f32x4 c = cn[g];  // slot 0, requested during the last transposed GEMM
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float v = act[4 * g + k] * c[k];
        const f32x4 w = lds_f4(lds, V2_TABS + H_TAB0 * 4 + (grp_f0(g) + k) * 16, o.h64);
        gx = fmaf(v, w[0], gx);
        gy = fmaf(v, w[1], gy);
        gz = fmaf(v, w[2], gz);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    gx += __shfl_xor(gx, 32, 64);
    gy += __shfl_xor(gy, 32, 64);
    gz += __shfl_xor(gz, 32, 64);
    gx *= run;  // identical in both lanes of a point (the max was taken over the pair)
    gy *= run;
    gz *= run;
    if (valid && h == 0) {
      grad_out[pt * 3 + 0] = gx;
      grad_out[pt * 3 + 1] = gy;
      grad_out[pt * 3 + 2] = gz;
    }

    if (rgb_out != nullptr) {
      // ---- colour head: sigmoid(Wrgb sin(gv * (Wv [feat, grad] + bv) + bv') + brgb)   (fields.py:89-101)
      // image 14 (ring slot 0) was prefetched during transposed layer 1 and is resident after its ring_sync
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const f32x4 v = ws.load(8, g, o);
#pragma unroll
        for (int k = 0; k < 4; ++k) act[4 * g + k] = v[k];
        if ((g & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      const LayOff y = lay_off<PREC>(o, 0, 8);
      zero_acc(acc);
      gemm_layer2<PREC>(lds, y, act, acc);
      // the accumulators carry the image scale 2^k (1 unless F16X3): bring the rank-3 gradient term to the same scale
      const float cs = 1.0f / hdr[H_WSCALE + 14];
      film_sin2<FAST, false, 2>(lds, o, y, acc, act, ws, 0, H_TABV, gx * cs, gy * cs, gz * cs);
      float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const f32x4 w0 = lds_f4(lds, V2_TABS + (H_RGB + 0 * C + grp_f0(g)) * 4, o.h16);
        const f32x4 w1 = lds_f4(lds, V2_TABS + (H_RGB + 1 * C + grp_f0(g)) * 4, o.h16);
        const f32x4 w2 = lds_f4(lds, V2_TABS + (H_RGB + 2 * C + grp_f0(g)) * 4, o.h16);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          r0 = fmaf(act[4 * g + k], w0[k], r0);
          r1 = fmaf(act[4 * g + k], w1[k], r1);
          r2 = fmaf(act[4 * g + k], w2[k], r2);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      r0 += __shfl_xor(r0, 32, 64);
      r1 += __shfl_xor(r1, 32, 64);
      r2 += __shfl_xor(r2, 32, 64);
      if (valid && h == 0) {
        const float* brgb = reinterpret_cast<const float*>(lds + V2_TABS + (H_RGB + 3 * C) * 4);
        rgb_out[pt * 3 + 0] = oi::sigmoidf_(r0 + brgb[0]);
        rgb_out[pt * 3 + 1] = oi::sigmoidf_(r1 + brgb[1]);
        rgb_out[pt * 3 + 2] = oi::sigmoidf_(r2 + brgb[2]);
      }
    }
  }
#ifdef OI_PROF
  PROF_T(9);
  if (lane == 0 && FULL) {
    for (int i = 0; i < 10; ++i) atomicAdd(&oi_prof[i], pacc[i]);
    atomicAdd(&oi_prof[10], __builtin_readcyclecounter() - tstart);
    atomicAdd(&oi_prof[11], 1ull);
  }
#endif
}

template <int PREC, bool FAST, bool FULL>
int launch_mlp_variant(const float* pts, const char* pk, const float* gamma, const float* beta, float* sdf, float* grad,
                       float* rgb, float* feat, char* scratch, int B, long long n, hipStream_t st) {
  constexpr int NWV = v2_waves(PREC, FULL);
  constexpr int LDS_BYTES = v2_lds_total(PREC, FULL);
  // the sdf-only passes run persistent workgroups (one per CU: the LDS): as many as the device has CUs (/ B), each walks its
  // share of the tiles; OI_V2_PERSIST=0 (environment): one workgroup per tile, the same kernel (A/B switch)
  static const int per_dev = [] {
    const char* v = getenv("OI_V2_PERSIST");
    if (v && v[0] == '0') return 0;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus > 0 ? cus : 256;
  }();
  const int tiles = oi::cdiv(n, NWV * WAVE_PTS);
  const bool persist = !FULL && v2_two_slots(PREC, FULL) && per_dev > 0;
  dim3 grid(persist ? std::min(tiles, std::max(1, per_dev / B)) : tiles, B), block(64 * NWV);
  auto k = sdf_mlp_kernel<PREC, FAST, FULL>;
  // per launch: the attribute is per device, and a process may drive several
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipLaunchKernelGGL(k, grid, block, LDS_BYTES, st, pts, pk, gamma, beta, sdf, grad, rgb, feat, scratch, n);
  return oi::check_launch("oi_sdf_mlp_fwd");
}

template <int PREC, bool FAST>
int launch_mlp(const float* pts, const void* packed, const float* gamma, const float* beta, float* sdf,
               float* grad, float* rgb, float* feat, void* scratch, int B, long long n, hipStream_t st) {
  const char* pk = reinterpret_cast<const char*>(packed);
  if (grad != nullptr)
    return launch_mlp_variant<PREC, FAST, true>(pts, pk, gamma, beta, sdf, grad, rgb, feat,
                                                reinterpret_cast<char*>(scratch), B, n, st);
  return launch_mlp_variant<PREC, FAST, false>(pts, pk, gamma, beta, sdf, nullptr, nullptr, feat, nullptr, B, n, st);
}

// the device sin/cos of the MLP kernels, exposed for tests (include/oi_hip.h: oi_selftest_sincos)
template <bool FAST>
__global__ void selftest_sincos_kernel(const float* __restrict__ x, float* __restrict__ s, float* __restrict__ c,
                                       long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float sv, cv;
  sincos_<FAST>(x[i], sv, cv);
  s[i] = sv;
  c[i] = cv;
}

}  // namespace

extern "C" {

int oi_film_params(const float* style_w, const float* style_b, const float* z, float* w_out, const float* gw,
                   const float* gb, const float* bw, const float* bb, float* gamma, float* beta, int B, int NL,
                   oi_stream_t stream) {
  OI_REQUIRE(B > 0 && NL >= 0, "oi_film_params: B=%d NL=%d", B, NL);
  OI_REQUIRE(w_out && (NL == 0 || (gw && gb && bw && bb && gamma && beta)), "oi_film_params: null pointer");
  OI_REQUIRE(z == nullptr || (style_w && style_b), "oi_film_params: z given without style weights");
  hipLaunchKernelGGL(film_params_kernel, dim3(B, NL > 0 ? NL : 1), dim3(C), 0, oi::as_stream(stream), style_w, style_b, z, w_out,
                     gw, gb, bw, bb, gamma, beta, NL);
  return oi::check_launch("oi_film_params");
}

#ifdef OI_PROF
int oi_prof_read(unsigned long long* out, int reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(oi_prof), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(oi_prof), z, sizeof(z));
  }
  return 0;
}
#endif

int oi_film_params_bwd(const float* d_gamma, const float* d_beta, const float* w, const float* gw, const float* bw,
                       float* d_gw, float* d_gb, float* d_bw, float* d_bb, float* d_w, const float* style_w,
                       const float* style_b, const float* z, float* d_style_w, float* d_style_b, float* d_z, int B,
                       int NL, oi_stream_t stream) {
  // NL == 0: the style MLP alone (ShapeNetwork.style(z) under autograd, fields.py:15-21) -- d_w is then the upstream gradient
  OI_REQUIRE(B > 0 && NL >= 0 && (NL > 0 || z != nullptr), "oi_film_params_bwd: B=%d NL=%d", B, NL);
  OI_REQUIRE(d_w && (NL == 0 || (d_gamma && d_beta && w && gw && bw && d_gw && d_gb && d_bw && d_bb)),
             "oi_film_params_bwd: null pointer");
  OI_REQUIRE(z == nullptr || (style_w && style_b && d_style_w && d_style_b),
             "oi_film_params_bwd: style backward needs style_w, style_b, d_style_w, d_style_b");
  hipStream_t st = oi::as_stream(stream);
  int rc = OI_OK;
  if (NL > 0) {
    hipLaunchKernelGGL(film_heads_bwd_kernel, dim3(NL), dim3(256), 0, st, d_gamma, d_beta, w, gw, bw, d_gw, d_gb, d_bw,
                       d_bb, d_w, B, NL);
    rc = oi::check_launch("oi_film_params_bwd(heads)");
  }
  if (rc != OI_OK || z == nullptr) return rc;
  hipLaunchKernelGGL(style_bwd_kernel, dim3(B), dim3(256), 0, st, style_w, style_b, z, d_w, d_style_w, d_style_b, d_z);
  return oi::check_launch("oi_film_params_bwd(style)");
}

size_t oi_mlp_packed_bytes(int prec) { return packed_total_bytes(prec); }

int oi_mlp_pack_weights(const float* w0, const float* b0, const float* wh, const float* bh, const float* wsig,
                        const float* bsig, const float* wv, const float* bv, const float* wrgb, const float* brgb,
                        void* packed, int prec, oi_stream_t stream) {
  OI_REQUIRE(w0 && b0 && wh && bh && wsig && bsig && wv && bv && wrgb && brgb && packed,
             "oi_mlp_pack_weights: null pointer");
  dim3 grid(C * C / (256 * PK_PER), NMAT + 1), block(256);
  char* p = reinterpret_cast<char*>(packed);
  hipStream_t st = oi::as_stream(stream);
  switch (prec) {
    case OI_PREC_F32:
      hipLaunchKernelGGL(pack_weights_kernel<OI_PREC_F32>, grid, block, 0, st, w0, b0, wh, bh, wsig, bsig, wv, bv,
                         wrgb, brgb, p);
      break;
    case OI_PREC_BF16X3:
      hipLaunchKernelGGL(pack_weights_kernel<OI_PREC_BF16X3>, grid, block, 0, st, w0, b0, wh, bh, wsig, bsig, wv, bv,
                         wrgb, brgb, p);
      break;
    case OI_PREC_BF16:
      hipLaunchKernelGGL(pack_weights_kernel<OI_PREC_BF16>, grid, block, 0, st, w0, b0, wh, bh, wsig, bsig, wv, bv,
                         wrgb, brgb, p);
      break;
    case OI_PREC_BF16X6:
      hipLaunchKernelGGL(pack_weights_kernel<OI_PREC_BF16X6>, grid, block, 0, st, w0, b0, wh, bh, wsig, bsig, wv, bv,
                         wrgb, brgb, p);
      break;
    case OI_PREC_F16X3:
      hipLaunchKernelGGL(pack_weights_kernel<OI_PREC_F16X3>, grid, block, 0, st, w0, b0, wh, bh, wsig, bsig, wv, bv,
                         wrgb, brgb, p);
      break;
    default:
      return oi::fail(OI_ERR_INVALID_ARG, "oi_mlp_pack_weights: bad precision %d", prec);
  }
  return oi::check_launch("oi_mlp_pack_weights");
}

int oi_selftest_sincos(const float* x, float* s, float* c, long long n, int fast, oi_stream_t stream) {
  OI_REQUIRE(x && s && c && n > 0, "oi_selftest_sincos: bad argument");
  if (fast)
    hipLaunchKernelGGL(selftest_sincos_kernel<true>, dim3(oi::cdiv(n, 256)), dim3(256), 0, oi::as_stream(stream), x, s, c, n);
  else
    hipLaunchKernelGGL(selftest_sincos_kernel<false>, dim3(oi::cdiv(n, 256)), dim3(256), 0, oi::as_stream(stream), x, s, c, n);
  return oi::check_launch("oi_selftest_sincos");
}

int oi_mlp_pack_status(const void* packed, oi_stream_t stream) {
  OI_REQUIRE(packed, "oi_mlp_pack_status: null pointer");
  hipStream_t st = oi::as_stream(stream);
  static thread_local float host[H_FLOATS];
  if (hipMemcpyAsync(host, packed, sizeof(host), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return oi::fail(OI_ERR_LAUNCH, "oi_mlp_pack_status: reading the packed header failed");
  for (int m = 0; m < NMAT; ++m)
    if (host[H_STATUS + m] != 0.f)
      return oi::fail(OI_ERR_UNSUPPORTED, "oi_mlp_pack_status: weight image %d holds a non-finite value (inf / NaN)", m);
  for (int i = 0; i < H_BIAS + 9 * C; ++i)
    if (!(host[i] >= -3.0e38f && host[i] <= 3.0e38f))
      return oi::fail(OI_ERR_UNSUPPORTED, "oi_mlp_pack_status: non-finite first-layer / head weight or bias (header float %d)", i);
  return OI_OK;
}

size_t oi_mlp_scratch_bytes(int B, long long n_per_elem) {  // upper bound over every precision / kernel
  const long long tiles = (n_per_elem + V2_TILE - 1) / V2_TILE;
  return (size_t)B * tiles * V2_WAVES * NSLOT * 16 * 64 * 16;
}

size_t oi_mlp_scratch_bytes_prec(int B, long long n_per_elem, int prec) {
  static const bool use_v2 = [] { const char* v = getenv("OI_FWD_V2"); return v && v[0] == '1'; }();
  if (prec == OI_PREC_F16X3 && !use_v2) return oimlp::full3_scratch_bytes(B, n_per_elem);  // 512 B/point
  if (prec == OI_PREC_BF16 && !use_v2) return oimlp::full3_bf16_scratch_bytes(B);  // mlp_fwd3b.hip: 15 per-element images
  return oi_mlp_scratch_bytes(B, n_per_elem);                                               // 4.6 KB/point
}

int oi_sdf_mlp_fwd(const float* pts, const void* packed, const float* gamma, const float* beta, float* sdf,
                   float* grad, float* rgb, float* feat, void* scratch, int B, long long n_per_elem, int prec,
                   int fast_trig, oi_stream_t stream) {
  return oi_sdf_mlp_fwd_ex(pts, packed, gamma, beta, sdf, grad, rgb, feat, scratch, B, n_per_elem, prec, fast_trig, 0, stream);
}

size_t oi_mlp_f3_blob_offset(int B, long long n_per_elem) { return oimlp::full3_blob_offset(B, n_per_elem); }
size_t oi_mlp_f3_blob_bytes(void) { return (size_t)oif3::F3_BLOB; }

int oi_sdf_mlp_fwd_ex(const float* pts, const void* packed, const float* gamma, const float* beta, float* sdf,
                      float* grad, float* rgb, float* feat, void* scratch, int B, long long n_per_elem, int prec,
                      int fast_trig, int flags, oi_stream_t stream) {
  OI_REQUIRE(pts && packed && gamma && beta && sdf, "oi_sdf_mlp_fwd: null pointer");
  OI_REQUIRE((flags & ~OI_MLP_BLOB_READY) == 0, "oi_sdf_mlp_fwd_ex: unknown flag bits %d", flags);
  OI_REQUIRE(!(flags & OI_MLP_BLOB_READY) || (prec == OI_PREC_F16X3 && grad != nullptr),
             "oi_sdf_mlp_fwd_ex: OI_MLP_BLOB_READY applies to the OI_PREC_F16X3 pass with the gradient");
  OI_REQUIRE(B > 0 && n_per_elem > 0, "oi_sdf_mlp_fwd: B=%d n=%lld", B, n_per_elem);
  OI_REQUIRE(grad != nullptr || rgb == nullptr, "oi_sdf_mlp_fwd: rgb requires grad");
  OI_REQUIRE(grad == nullptr || scratch != nullptr, "oi_sdf_mlp_fwd: grad requires scratch");
  hipStream_t st = oi::as_stream(stream);
  // F16X3 with the gradient: the register-resident kernel (mlp_fwd3.hip); OI_FWD_V2=1 keeps the scratch-streaming v2
  static const bool use_v2 = [] { const char* v = getenv("OI_FWD_V2"); return v && v[0] == '1'; }();
  if (prec == OI_PREC_F16X3 && grad != nullptr && !use_v2)
    return oimlp::launch_full3_f16x3(pts, packed, gamma, beta, sdf, grad, rgb, feat, scratch, B, n_per_elem, fast_trig,
                                     (flags & OI_MLP_BLOB_READY) != 0, st);
  // BF16 with the gradient: the register-resident kernel of mlp_fwd3b.hip (no scratch stream)
  if (prec == OI_PREC_BF16 && grad != nullptr && !use_v2)
    return oimlp::launch_full3_bf16(pts, packed, gamma, beta, sdf, grad, rgb, feat, scratch, B, n_per_elem, fast_trig, st);
#define OI_MLP_CASE(P)                                                                                        \
  case P:                                                                                                     \
    return fast_trig ? launch_mlp<P, true>(pts, packed, gamma, beta, sdf, grad, rgb, feat, scratch, B, n_per_elem, st) \
                     : launch_mlp<P, false>(pts, packed, gamma, beta, sdf, grad, rgb, feat, scratch, B, n_per_elem, st);
  switch (prec) {
    OI_MLP_CASE(OI_PREC_F32)
    OI_MLP_CASE(OI_PREC_BF16X3)
    OI_MLP_CASE(OI_PREC_BF16)
    OI_MLP_CASE(OI_PREC_BF16X6)
    OI_MLP_CASE(OI_PREC_F16X3)
    default:
      return oi::fail(OI_ERR_INVALID_ARG, "oi_sdf_mlp_fwd: bad precision %d", prec);
  }
#undef OI_MLP_CASE
}

}  // extern "C"
