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
#include "oi_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int C = 128;           // hidden width (W in the reference config)
constexpr int NL_SDF = 8;        // FiLM layers of the SDF net
constexpr int NMAT = 15;         // 7 forward + 7 transposed + colour head
constexpr int TILE_PTS = 128;    // points per workgroup
constexpr int WAVE_PTS = 32;     // points per wavefront
constexpr int NSLOT = 9;         // scratch slots per wave tile: c_0..c_7, feat

// packed header (floats)
constexpr int H_TAB0 = 0;        // [128][4]  (w0x, w0y, w0z, 0)
constexpr int H_SIG = 512;       // [128] wsig, [128] = bsig
constexpr int H_TABV = 656;      // [128][4]  (wv[:,128], wv[:,129], wv[:,130], 0)
constexpr int H_RGB = 1168;      // [3][128] wrgb, then brgb[3]
constexpr int H_TABS_END = 1568; // tab0..rgb are copied to LDS as one block
constexpr int H_BIAS = 1568;     // [9][128]  b0, b1..b7, bv
constexpr int H_FLOATS = 2816;
constexpr size_t H_BYTES = H_FLOATS * 4;

__host__ __device__ constexpr int layer_bytes(int prec) { return prec == OI_PREC_BF16 ? 32768 : 65536; }

// LDS carve (bytes)
// (small tables first so that every table access is <lane-constant VGPR> + 16-bit immediate)
constexpr int L_FILM = 0;                 // gamma[128], beta[128], bias[128]
constexpr int L_TABS = L_FILM + 1536;     // H_TABS_END floats
constexpr int L_WBUF = L_TABS + H_TABS_END * 4;  // 7808, one layer image
constexpr int L_TOTAL = L_WBUF + 65536;

__host__ __device__ __forceinline__ int feat_of(int q, int h) {
  return 32 * (q >> 4) + 8 * ((q >> 2) & 3) + 4 * h + (q & 3);
}

// ------------------------------------------------------------------------------------------
// a1 + a2: style MLP + FiLM parameters.  One block of 128 threads per batch element.
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
    if (t < 64) w_out[e * 64 + t] = h[cur][t];
    if (cur != 0) {
      if (t < 64) h[0][t] = h[1][t];
    }
    __syncthreads();
  } else {
    if (t < 64) h[0][t] = w_out[e * 64 + t];
    __syncthreads();
  }
  for (int l = 0; l < NL; ++l) {
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

// ------------------------------------------------------------------------------------------
// weight pre-pack
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float mat_elem(const float* wh, const float* wv, int m, int row, int k) {
  // m: 0..6 forward layer 1..7 -> W[row][k]; 7..13 transposed layer 1..7 -> W[k][row]; 14 colour -> Wv[row][k]
  if (m < 7) return wh[((size_t)m * C + row) * C + k];
  if (m < 14) return wh[((size_t)(m - 7) * C + k) * C + row];
  return wv[(size_t)row * (C + 3) + k];
}

template <int PREC>
__global__ void pack_weights_kernel(const float* __restrict__ w0, const float* __restrict__ b0,
                                    const float* __restrict__ wh, const float* __restrict__ bh,
                                    const float* __restrict__ wsig, const float* __restrict__ bsig,
                                    const float* __restrict__ wv, const float* __restrict__ bv,
                                    const float* __restrict__ wrgb, const float* __restrict__ brgb,
                                    char* __restrict__ packed) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  float* hdr = reinterpret_cast<float*>(packed);
  if (blockIdx.y == NMAT) {  // header
    if (idx >= H_FLOATS) return;
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
    return;
  }
  const int m = blockIdx.y;
  if (idx >= C * C) return;
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
    const float v = mat_elem(wh, wv, m, 32 * t + i, feat_of(q, h));
    const __bf16 hi = (__bf16)v;
    reinterpret_cast<__bf16*>(base)[idx] = hi;
    if (PREC == OI_PREC_BF16X3) reinterpret_cast<__bf16*>(base + 32768)[idx] = (__bf16)(v - (float)hi);
  }
}

// ------------------------------------------------------------------------------------------
// the MLP kernel
// ------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// All LDS accesses are "<laundered per-lane VGPR> + compile-time immediate" so that hipcc emits
// ds_read_b128 v, vbase offset:imm and cannot hoist 60+ loop-invariant address registers out of the
// layer loop (that, not the data, is what overflowed the 256-VGPR budget in the first version).
struct LaneOff {
  int h16;    // 16 * (lane >> 5)        : selects the lane-half's 4 features inside a group of 8
  int h64;    // 64 * (lane >> 5)        : same for [128][4] tables
  int l16;    // 16 * lane               : lane-linear weight image, first 32 KiB
  int l16hi;  // 16 * lane + 32768       : second 32 KiB of the image
};

__device__ __forceinline__ f32x4 lds_f4(const char* lds, int imm, int var) {
  return *reinterpret_cast<const f32x4*>(lds + imm + var);
}
// 16 bytes of the staged layer image at byte offset `imm` (compile-time) for this lane
__device__ __forceinline__ f32x4 wimg_f4(const char* lds, const LaneOff& o, int imm) {
  return imm < 32768 ? lds_f4(lds, L_WBUF + imm, o.l16) : lds_f4(lds, L_WBUF + imm - 32768, o.l16hi);
}
// group g (0..15) of 4 consecutive features of this lane: first feature = 32*(g>>2) + 8*(g&3) + 4h
__device__ __forceinline__ constexpr int grp_f0(int g) { return 32 * (g >> 2) + 8 * (g & 3); }

template <int PREC>
__device__ __forceinline__ void stage_layer(char* lds, const char* __restrict__ src, int tid) {
  constexpr int N16 = layer_bytes(PREC) / 16;
  const f32x4* s = reinterpret_cast<const f32x4*>(src);
  f32x4* d = reinterpret_cast<f32x4*>(lds + L_WBUF);
#pragma unroll
  for (int i = 0; i < N16 / 256; ++i) d[i * 256 + tid] = s[i * 256 + tid];
}

__device__ __forceinline__ void stage_film(char* lds, const float* __restrict__ gamma,
                                           const float* __restrict__ beta, const float* __restrict__ hdr,
                                           int e, int l, int tid) {
  float* film = reinterpret_cast<float*>(lds + L_FILM);
  if (tid < C) {
    film[tid] = gamma[((size_t)e * 9 + l) * C + tid];
    film[C + tid] = beta[((size_t)e * 9 + l) * C + tid];
    film[2 * C + tid] = hdr[H_BIAS + l * C + tid];
  }
}

// acc[t][r] (+)= sum_k A[32t + row][k] * act[k]   with the packed A image in LDS.
// The A fragments are prefetched exactly one k-group ahead; sched_barrier pins that window.
template <int PREC>
__device__ __forceinline__ void gemm_layer(const char* lds, const LaneOff& o, const float (&act)[64],
                                           f32x16 (&acc)[4]) {
  if constexpr (PREC == OI_PREC_F32) {
    f32x4 a[4], an[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = wimg_f4(lds, o, (t * 16 + 0) * 1024);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      if (g < 15) {
#pragma unroll
        for (int t = 0; t < 4; ++t) an[t] = wimg_f4(lds, o, (t * 16 + g + 1) * 1024);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][k], act[4 * g + k], acc[t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] = an[t];
    }
  } else {
    f32x4 ah[4], ahn[4], al[4], aln[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      ah[t] = lds_f4(lds, L_WBUF + (t * 8 + 0) * 1024, o.l16);
      if constexpr (PREC == OI_PREC_BF16X3) al[t] = lds_f4(lds, L_WBUF + (t * 8 + 0) * 1024, o.l16hi);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < 7) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          ahn[t] = lds_f4(lds, L_WBUF + (t * 8 + s + 1) * 1024, o.l16);
          if constexpr (PREC == OI_PREC_BF16X3) aln[t] = lds_f4(lds, L_WBUF + (t * 8 + s + 1) * 1024, o.l16hi);
        }
      }
      bf16x8 bh, bl;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float v = act[8 * s + i];
        bh[i] = (__bf16)v;
        if constexpr (PREC == OI_PREC_BF16X3) bl[i] = (__bf16)(v - (float)bh[i]);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bf16x8 wh = __builtin_bit_cast(bf16x8, ah[t]);
        if constexpr (PREC == OI_PREC_BF16X3) {
          const bf16x8 wl = __builtin_bit_cast(bf16x8, al[t]);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, bh, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, bl, acc[t], 0, 0, 0);
        }
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, bh, acc[t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        ah[t] = ahn[t];
        if constexpr (PREC == OI_PREC_BF16X3) al[t] = aln[t];
      }
    }
  }
}

// sin and cos of one fp32 phase.  Accurate form: 2-constant Cody-Waite reduction by pi/2 with FMA
// (|phi| stays below a few hundred radians: gamma ~ 30 +- 15, |u| of order one) followed by the
// classic minimax kernels on [-pi/4, pi/4]; <1e-7 abs error, no stack, ~22 VALU ops for the pair.
// Fast form: v_sin_f32 / v_cos_f32 on phi/(2 pi) (used by the bf16 throughput mode).
template <bool FAST>
__device__ __forceinline__ void sincos_(float x, float& s, float& c) {
  if constexpr (FAST) {
    s = __sinf(x);
    c = __cosf(x);
  } else {
    const float n = rintf(x * 0.63661977236758134308f);
    float r = fmaf(n, -1.57079637050628662109375f, x);
    r = fmaf(n, 4.37113882867379e-08f, r);
    const float r2 = r * r;
    float ps = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(r2, ps, -1.6666654611e-1f);
    ps = fmaf(r2 * r, ps, r);
    float pc = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(r2, pc, 4.166664568298827e-2f);
    pc = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));
    const int q = (int)n;
    const float sa = (q & 1) ? pc : ps;
    const float ca = (q & 1) ? ps : pc;
    s = (q & 2) ? -sa : sa;
    c = ((q + 1) & 2) ? -ca : ca;
  }
}

// scratch of one wave tile, addressed through a buffer descriptor: voffset = 16*lane (VGPR),
// soffset = slot*16 KiB + g*1 KiB (SGPR / immediate) -> no per-access address VGPRs.
struct WaveScratch {
  __amdgpu_buffer_rsrc_t rs;
  __device__ __forceinline__ void store(int slot, int g, int l16, f32x4 v) const {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, l16, slot * 16384 + g * 1024, 0);
  }
  __device__ __forceinline__ f32x4 load(int slot, int g, int l16) const {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, l16, slot * 16384 + g * 1024, 0));
  }
};

// FiLM + sin; act <- sin(phi); optionally parks gamma*cos(phi) in scratch slot `slot`.
// SRC selects where the pre-activation u comes from, computed right where it is consumed so that no
// table value outlives its group of four features:
//   0: u = acc (MFMA layers, bias already in the accumulator)
//   1: u = tab[f].xyz . v + bias[f]         (layer 0: v = the point;  F.linear, volume_renderer.py:52)
//   2: u = acc + tab[f].xyz . v             (colour head: v = d sdf/dx, the 3 extra input columns)
template <bool FAST, bool FULL, int SRC>
__device__ __forceinline__ void film_sin(const char* lds, const LaneOff& o, const f32x16 (&acc)[4],
                                         float (&act)[64], const WaveScratch& ws, int slot, int tab_imm,
                                         float vx, float vy, float vz) {
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int t = g >> 2, rr = g & 3;
    const f32x4 gm = lds_f4(lds, L_FILM + grp_f0(g) * 4, o.h16);
    const f32x4 bt = lds_f4(lds, L_FILM + (C + grp_f0(g)) * 4, o.h16);
    f32x4 bs;
    if constexpr (SRC == 1) bs = lds_f4(lds, L_FILM + (2 * C + grp_f0(g)) * 4, o.h16);
    f32x4 cv;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float u;
      if constexpr (SRC == 0) {
        u = acc[t][4 * rr + k];
      } else {
        const f32x4 w = lds_f4(lds, L_TABS + tab_imm * 4 + (grp_f0(g) + k) * 16, o.h64);
        const float d = fmaf(vz, w[2], fmaf(vy, w[1], vx * w[0]));
        u = SRC == 1 ? d + bs[k] : acc[t][4 * rr + k] + d;
      }
      const float phi = fmaf(gm[k], u, bt[k]);
      float s, c;
      sincos_<FAST>(phi, s, c);
      act[4 * g + k] = s;
      cv[k] = gm[k] * c;
    }
    if constexpr (FULL) ws.store(slot, g, o.l16, cv);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ void init_bias(const char* lds, const LaneOff& o, f32x16 (&acc)[4]) {
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const f32x4 b = lds_f4(lds, L_FILM + (2 * C + grp_f0(g)) * 4, o.h16);
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[g >> 2][4 * (g & 3) + k] = b[k];
  }
}

template <int PREC, bool FAST, bool FULL>
__global__ void __launch_bounds__(256, 2)
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

  LaneOff o;
  o.h16 = 16 * h;
  o.h64 = 64 * h;
  o.l16 = 16 * lane;
  o.l16hi = 16 * lane + 32768;
  asm volatile("" : "+v"(o.h16), "+v"(o.h64), "+v"(o.l16), "+v"(o.l16hi));

  const long long local = (long long)blockIdx.x * TILE_PTS + wave * WAVE_PTS + j;
  const bool valid = local < n_per_elem;
  const long long pt = (long long)e * n_per_elem + (valid ? local : n_per_elem - 1);

  // scratch of this wave tile: [slot(9)][g(16)][lane(64)] x float4
  WaveScratch ws;
  {
    const long long wt = ((long long)e * gridDim.x + blockIdx.x) * 4 + wave;
    char* wbase = FULL ? scratch + wt * (long long)(NSLOT * 16384) : nullptr;
    ws.rs = __builtin_amdgcn_make_buffer_rsrc(wbase, 0, FULL ? NSLOT * 16384 : 0, 0x00020000);
  }

  // stage the small tables + layer-0 FiLM
  {
    float* tabs = reinterpret_cast<float*>(lds + L_TABS);
    for (int i = tid; i < H_TABS_END; i += 256) tabs[i] = hdr[i];
    stage_film(lds, gamma, beta, hdr, e, 0, tid);
  }
  const float px = pts[pt * 3 + 0], py = pts[pt * 3 + 1], pz = pts[pt * 3 + 2];
  __syncthreads();

  float act[64];
  f32x16 acc[4];

  // ---- layer 0 (K = 3) on the VALU
  film_sin<FAST, FULL, 1>(lds, o, acc, act, ws, 0, H_TAB0, px, py, pz);

  // ---- layers 1..7 on MFMA
  for (int l = 1; l < NL_SDF; ++l) {
    __syncthreads();
    stage_layer<PREC>(lds, mats + (size_t)(l - 1) * layer_bytes(PREC), tid);
    stage_film(lds, gamma, beta, hdr, e, l, tid);
    __syncthreads();
    init_bias(lds, o, acc);
    gemm_layer<PREC>(lds, o, act, acc);
    film_sin<FAST, FULL, 0>(lds, o, acc, act, ws, l, 0, 0.f, 0.f, 0.f);
  }

  // ---- sdf = a8 . wsig + bsig   (fields.py:68; LinearLayer std_init=1, bias_init=0)
  float sdf_v;
  {
    float part = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const f32x4 w = lds_f4(lds, L_TABS + (H_SIG + grp_f0(g)) * 4, o.h16);
#pragma unroll
      for (int k = 0; k < 4; ++k) part = fmaf(act[4 * g + k], w[k], part);
    }
    part += __shfl_xor(part, 32, 64);
    sdf_v = part + *reinterpret_cast<const float*>(lds + L_TABS + (H_SIG + C) * 4);
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

  if constexpr (FULL) {
    // park the features (slot 8) and start the reverse sweep with g8 = wsig
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = act[4 * g + k];
      ws.store(8, g, o.l16, v);
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const f32x4 w = lds_f4(lds, L_TABS + (H_SIG + grp_f0(g)) * 4, o.h16);
#pragma unroll
      for (int k = 0; k < 4; ++k) act[4 * g + k] = w[k];
    }
    for (int l = NL_SDF - 1; l >= 1; --l) {
      __syncthreads();
      stage_layer<PREC>(lds, mats + (size_t)(7 + l - 1) * layer_bytes(PREC), tid);
      __syncthreads();
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const f32x4 c = ws.load(l, g, o.l16);
#pragma unroll
        for (int k = 0; k < 4; ++k) act[4 * g + k] *= c[k];
        if ((g & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      gemm_layer<PREC>(lds, o, act, acc);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) act[16 * t + r] = acc[t][r];
    }
    // layer 0: grad = W0^T (g1 * c0)
    float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const f32x4 c = ws.load(0, g, o.l16);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float v = act[4 * g + k] * c[k];
        const f32x4 w = lds_f4(lds, L_TABS + H_TAB0 * 4 + (grp_f0(g) + k) * 16, o.h64);
        gx = fmaf(v, w[0], gx);
        gy = fmaf(v, w[1], gy);
        gz = fmaf(v, w[2], gz);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    gx += __shfl_xor(gx, 32, 64);
    gy += __shfl_xor(gy, 32, 64);
    gz += __shfl_xor(gz, 32, 64);
    if (valid && h == 0) {
      grad_out[pt * 3 + 0] = gx;
      grad_out[pt * 3 + 1] = gy;
      grad_out[pt * 3 + 2] = gz;
    }

    if (rgb_out != nullptr) {
      // ---- colour head: sigmoid(Wrgb sin(gv * (Wv [feat, grad] + bv) + bv') + brgb)   (fields.py:89-101)
      __syncthreads();
      stage_layer<PREC>(lds, mats + (size_t)14 * layer_bytes(PREC), tid);
      stage_film(lds, gamma, beta, hdr, e, 8, tid);
      __syncthreads();
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const f32x4 v = ws.load(8, g, o.l16);
#pragma unroll
        for (int k = 0; k < 4; ++k) act[4 * g + k] = v[k];
        if ((g & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      init_bias(lds, o, acc);
      gemm_layer<PREC>(lds, o, act, acc);
      film_sin<FAST, false, 2>(lds, o, acc, act, ws, 0, H_TABV, gx, gy, gz);
      float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const f32x4 w0 = lds_f4(lds, L_TABS + (H_RGB + 0 * C + grp_f0(g)) * 4, o.h16);
        const f32x4 w1 = lds_f4(lds, L_TABS + (H_RGB + 1 * C + grp_f0(g)) * 4, o.h16);
        const f32x4 w2 = lds_f4(lds, L_TABS + (H_RGB + 2 * C + grp_f0(g)) * 4, o.h16);
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
        const float* brgb = reinterpret_cast<const float*>(lds + L_TABS + (H_RGB + 3 * C) * 4);
        rgb_out[pt * 3 + 0] = oi::sigmoidf_(r0 + brgb[0]);
        rgb_out[pt * 3 + 1] = oi::sigmoidf_(r1 + brgb[1]);
        rgb_out[pt * 3 + 2] = oi::sigmoidf_(r2 + brgb[2]);
      }
    }
  }
}

template <int PREC, bool FAST>
int launch_mlp(const float* pts, const void* packed, const float* gamma, const float* beta, float* sdf,
               float* grad, float* rgb, float* feat, void* scratch, int B, long long n, hipStream_t st) {
  dim3 grid(oi::cdiv(n, TILE_PTS), B), block(256);
  const char* pk = reinterpret_cast<const char*>(packed);
  if (grad != nullptr) {
    auto k = sdf_mlp_kernel<PREC, FAST, true>;
    static thread_local bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, L_TOTAL);
      attr = true;
    }
    hipLaunchKernelGGL(k, grid, block, L_TOTAL, st, pts, pk, gamma, beta, sdf, grad, rgb, feat,
                       reinterpret_cast<char*>(scratch), n);
  } else {
    auto k = sdf_mlp_kernel<PREC, FAST, false>;
    static thread_local bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, L_TOTAL);
      attr = true;
    }
    hipLaunchKernelGGL(k, grid, block, L_TOTAL, st, pts, pk, gamma, beta, sdf, (float*)nullptr, (float*)nullptr,
                       feat, (char*)nullptr, n);
  }
  return oi::check_launch("oi_sdf_mlp_fwd");
}

}  // namespace

extern "C" {

int oi_film_params(const float* style_w, const float* style_b, const float* z, float* w_out, const float* gw,
                   const float* gb, const float* bw, const float* bb, float* gamma, float* beta, int B, int NL,
                   oi_stream_t stream) {
  OI_REQUIRE(B > 0 && NL >= 0, "oi_film_params: B=%d NL=%d", B, NL);
  OI_REQUIRE(w_out && (NL == 0 || (gw && gb && bw && bb && gamma && beta)), "oi_film_params: null pointer");
  OI_REQUIRE(z == nullptr || (style_w && style_b), "oi_film_params: z given without style weights");
  hipLaunchKernelGGL(film_params_kernel, dim3(B), dim3(C), 0, oi::as_stream(stream), style_w, style_b, z, w_out,
                     gw, gb, bw, bb, gamma, beta, NL);
  return oi::check_launch("oi_film_params");
}

size_t oi_mlp_packed_bytes(int prec) { return H_BYTES + (size_t)NMAT * layer_bytes(prec); }

int oi_mlp_pack_weights(const float* w0, const float* b0, const float* wh, const float* bh, const float* wsig,
                        const float* bsig, const float* wv, const float* bv, const float* wrgb, const float* brgb,
                        void* packed, int prec, oi_stream_t stream) {
  OI_REQUIRE(w0 && b0 && wh && bh && wsig && bsig && wv && bv && wrgb && brgb && packed,
             "oi_mlp_pack_weights: null pointer");
  dim3 grid(C * C / 256, NMAT + 1), block(256);
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
    default:
      return oi::fail(OI_ERR_INVALID_ARG, "oi_mlp_pack_weights: bad precision %d", prec);
  }
  return oi::check_launch("oi_mlp_pack_weights");
}

size_t oi_mlp_scratch_bytes(int B, long long n_per_elem) {
  const long long tiles = (n_per_elem + TILE_PTS - 1) / TILE_PTS;
  return (size_t)B * tiles * 4 * NSLOT * 16 * 64 * 16;
}

int oi_sdf_mlp_fwd(const float* pts, const void* packed, const float* gamma, const float* beta, float* sdf,
                   float* grad, float* rgb, float* feat, void* scratch, int B, long long n_per_elem, int prec,
                   int fast_trig, oi_stream_t stream) {
  OI_REQUIRE(pts && packed && gamma && beta && sdf, "oi_sdf_mlp_fwd: null pointer");
  OI_REQUIRE(B > 0 && n_per_elem > 0, "oi_sdf_mlp_fwd: B=%d n=%lld", B, n_per_elem);
  OI_REQUIRE(grad != nullptr || rgb == nullptr, "oi_sdf_mlp_fwd: rgb requires grad");
  OI_REQUIRE(grad == nullptr || scratch != nullptr, "oi_sdf_mlp_fwd: grad requires scratch");
  hipStream_t st = oi::as_stream(stream);
#define OI_MLP_CASE(P)                                                                                        \
  case P:                                                                                                     \
    return fast_trig ? launch_mlp<P, true>(pts, packed, gamma, beta, sdf, grad, rgb, feat, scratch, B, n_per_elem, st) \
                     : launch_mlp<P, false>(pts, packed, gamma, beta, sdf, grad, rgb, feat, scratch, B, n_per_elem, st);
  switch (prec) {
    OI_MLP_CASE(OI_PREC_F32)
    OI_MLP_CASE(OI_PREC_BF16X3)
    OI_MLP_CASE(OI_PREC_BF16)
    default:
      return oi::fail(OI_ERR_INVALID_ARG, "oi_sdf_mlp_fwd: bad precision %d", prec);
  }
#undef OI_MLP_CASE
}

}  // extern "C"
