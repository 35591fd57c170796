// Stand-alone albedo head: ColorNetwork.forward on CALLER-SUPPLIED features and normals
//   rgb = sigmoid(rgb_linear(sin(gamma_v * (cat[feature_vectors, normals] Wv^T + bv) + beta_v)))
// (src/models/fields.py:89-101 -> FiLMSiren.forward, stylesdf/volume_renderer.py:50-61), and its backward with respect to
// the features, the normals, the FiLM rows and every parameter.  The render path never comes here -- there the head is the
// tail of the fused MLP launch (mlp_fwd3.hip / mlp.hip) and its inputs never leave the registers; this file serves a caller
// that keeps the reference's renderer.py:241-261 and hands the features over in memory (SURVEY.md 8b "Field-network methods").
//
// Exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fma chain), point on the MFMA column as in the
// other MLP kernels: out^T = Wv in^T, so a lane keeps its point through both products and the C fragment of the first
// (du, after the epilogue) is the B fragment of the second as it stands.  The K order of each product is free and chosen for
// the operand that is hardest to move:
//   product 1 (u = Wv c):    lane-half kh supplies feature 64 kh + s at step s -> a lane loads ONE contiguous 256-byte half of
//                            its point's feature row; Wv sits in LDS with an odd row stride (133): A reads (feature on the
//                            lane) and the transposed reads of product 2 (input column on the lane) are both conflict-free.
//   product 2 (dc = Wv^T du): step (t, r) uses the feature the C layout gives slot r of block t in lane-half kh.
// Weight-side gradients: one GEMM over the points (K = points: both operands are read with the feature on the lane, straight
// from memory, coalesced) that accumulates D_e = dphi^T [c | 1] per batch element WITHOUT the FiLM scale, so that (no division
// by gamma, as in mlp_bwd.hip)
//   dWv[f][i] = sum_e gamma_e[f] D_e[f][i]      dbv[f] = sum_e gamma_e[f] D_e[f][131]
//   dgamma_e[f] = sum_i Wv[f][i] D_e[f][i] + bv[f] D_e[f][131]      dbeta_e[f] = D_e[f][131]
// and dWrgb = sum_p dpre a^T rides on the same pass (one more MFMA per step).  Per-workgroup partial sums go to the workspace
// and are added in a fixed order: no atomics, bit-reproducible.
#include "mlp_common.h"

namespace {

using oimlp::f32x16;
using oimlp::f32x4;

constexpr int C = 128, CIN = 131;
constexpr int SW = 133;                 // LDS row stride of Wv (floats); cols 131, 132 are zero
constexpr int TILE = 128;               // points per workgroup (4 waves x 32)
constexpr int LW = 0, LG = C * SW, LB = LG + C, LBI = LB + C, LR = LBI + C, LRB = LR + 3 * C, L_FLOATS = LRB + 4;
constexpr int L_BYTES = L_FLOATS * 4;   // 71,184 B: two workgroups per CU
constexpr int PCOLS = 144;              // workspace row: [0..130] D, [131] sum dphi, [132..134] sum a dpre_c
constexpr int WG_U = 8;                 // MFMA steps per register batch of the point GEMM (2 points per step)

__device__ __forceinline__ void stage(float* lds, const float* __restrict__ wv, const float* __restrict__ bv,
                                      const float* __restrict__ wrgb, const float* __restrict__ brgb,
                                      const float* __restrict__ gamma_e, const float* __restrict__ beta_e, int tid) {
  for (int idx = tid; idx < C * SW; idx += 256) {
    const int f = idx / SW, k = idx - f * SW;
    lds[LW + idx] = k < CIN ? wv[f * CIN + k] : 0.f;
  }
  if (tid < C) {
    lds[LG + tid] = gamma_e[tid];
    lds[LB + tid] = beta_e[tid];
    lds[LBI + tid] = bv[tid];
  }
  for (int idx = tid; idx < 3 * C; idx += 256) lds[LR + idx] = wrgb[idx];
  if (tid < 4) lds[LRB + tid] = tid < 3 ? brgb[tid] : 0.f;
}

// One 128-point tile: forward (BWD = false: rgb) or backward of the point side (BWD = true: d_feat, d_normals and the
// per-point vectors dphi, a = sin(phi), dpre the point GEMM consumes).
template <bool BWD>
__global__ void __launch_bounds__(256) color_head_kernel(
    const float* __restrict__ feat, const float* __restrict__ normals, const float* __restrict__ gamma,
    const float* __restrict__ beta, long long film_stride, const float* __restrict__ wv, const float* __restrict__ bv,
    const float* __restrict__ wrgb, const float* __restrict__ brgb, float* __restrict__ rgb, long long npe,
    const float* __restrict__ g_rgb, float* __restrict__ d_feat, float* __restrict__ d_normals, float* __restrict__ dphi_s,
    float* __restrict__ a_s, float* __restrict__ dpre_s, float* __restrict__ part_b) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, pl = lane & 31, h = lane >> 5;
  const int e = blockIdx.y;
  stage(lds, wv, bv, wrgb, brgb, gamma + e * film_stride, beta + e * film_stride, tid);
  const long long p = (long long)blockIdx.x * TILE + wave * 32 + pl;
  const bool valid = p < npe;
  const long long row = (long long)e * npe + (valid ? p : npe - 1);
  // this lane's half of its point's feature row, and the normal
  float x[64];
  {
    const f32x4* fr = reinterpret_cast<const f32x4*>(feat + row * C + 64 * h);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const f32x4 v = __builtin_nontemporal_load(fr + j);
      x[4 * j] = v[0], x[4 * j + 1] = v[1], x[4 * j + 2] = v[2], x[4 * j + 3] = v[3];
    }
  }
  const float n0 = normals[row * 3], n1 = normals[row * 3 + 1], n2 = normals[row * 3 + 2];
  const float xb64 = h ? n1 : n0, xb65 = h ? 0.f : n2;
  __syncthreads();

  // ---- product 1: u^T = Wv c^T (66 K-steps of 2; Wv column 131 is zero)
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  {
    const float* wa = lds + LW + pl * SW + 64 * h;
#pragma unroll
    for (int s = 0; s < 64; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[t * 32 * SW + s], x[s], acc[t], 0, 0, 0);
    const float* wn = lds + LW + pl * SW + 128 + h;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wn[t * 32 * SW], xb64, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wn[t * 32 * SW + 2], xb65, acc[t], 0, 0, 0);
    }
  }
  // ---- FiLM + sin; rgb head.  Slot r of block t in lane-half h is feature 32 t + 8 (r >> 2) + 4 h + (r & 3).
  float a[4][16], cs[4][16];
  float pre0 = 0.f, pre1 = 0.f, pre2 = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int f0 = 32 * t + 8 * g + 4 * h;
      const f32x4 gm = *reinterpret_cast<const f32x4*>(lds + LG + f0), bt = *reinterpret_cast<const f32x4*>(lds + LB + f0);
      const f32x4 bi = *reinterpret_cast<const f32x4*>(lds + LBI + f0);
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(lds + LR + f0), w1 = *reinterpret_cast<const f32x4*>(lds + LR + C + f0);
      const f32x4 w2 = *reinterpret_cast<const f32x4*>(lds + LR + 2 * C + f0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float u = acc[t][4 * g + k] + bi[k];
        float s_, c_;
        oimlp::sincos_<false>(fmaf(gm[k], u, bt[k]), s_, c_);
        a[t][4 * g + k] = s_;
        cs[t][4 * g + k] = c_;
        pre0 = fmaf(w0[k], s_, pre0), pre1 = fmaf(w1[k], s_, pre1), pre2 = fmaf(w2[k], s_, pre2);
      }
    }
  pre0 += __shfl_xor(pre0, 32, 64), pre1 += __shfl_xor(pre1, 32, 64), pre2 += __shfl_xor(pre2, 32, 64);
  const float o0 = oi::sigmoidf_(pre0 + lds[LRB]), o1 = oi::sigmoidf_(pre1 + lds[LRB + 1]), o2 = oi::sigmoidf_(pre2 + lds[LRB + 2]);
  if constexpr (!BWD) {
    if (valid && h == 0) rgb[row * 3] = o0, rgb[row * 3 + 1] = o1, rgb[row * 3 + 2] = o2;
    return;
  } else {
    // ---- backward of the head: dpre = g_rgb * rgb (1 - rgb)
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    if (valid) {
      d0 = g_rgb[row * 3] * o0 * (1.f - o0), d1 = g_rgb[row * 3 + 1] * o1 * (1.f - o1), d2 = g_rgb[row * 3 + 2] * o2 * (1.f - o2);
      if (h == 0) *reinterpret_cast<f32x4*>(dpre_s + row * 4) = f32x4{d0, d1, d2, 0.f};
    }
    {  // d brgb: this wave's 32 points (each point sits in both lane halves: count h == 0 only)
      const float s0 = oi::wave_sum(h ? 0.f : d0), s1 = oi::wave_sum(h ? 0.f : d1), s2 = oi::wave_sum(h ? 0.f : d2);
      if (lane == 0) {
        float* pb = part_b + (((long long)e * gridDim.x + blockIdx.x) * 4 + wave) * 4;
        pb[0] = s0, pb[1] = s1, pb[2] = s2, pb[3] = 0.f;
      }
    }
    // ---- dphi = (Wrgb^T dpre) cos(phi) (parked WITHOUT gamma for the point GEMM), du = gamma dphi
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int f0 = 32 * t + 8 * g + 4 * h;
        const f32x4 gm = *reinterpret_cast<const f32x4*>(lds + LG + f0);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(lds + LR + f0), w1 = *reinterpret_cast<const f32x4*>(lds + LR + C + f0);
        const f32x4 w2 = *reinterpret_cast<const f32x4*>(lds + LR + 2 * C + f0);
        f32x4 dp, av;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float da = fmaf(w2[k], d2, fmaf(w1[k], d1, w0[k] * d0));
          dp[k] = da * cs[t][4 * g + k];
          av[k] = a[t][4 * g + k];
          cs[t][4 * g + k] = gm[k] * dp[k];   // du, in the slot the cosine leaves
        }
        if (valid) {
          __builtin_nontemporal_store(dp, reinterpret_cast<f32x4*>(dphi_s + row * C + f0));
          __builtin_nontemporal_store(av, reinterpret_cast<f32x4*>(a_s + row * C + f0));
        }
      }
    // ---- product 2: dc^T = Wv^T du^T; column blocks 0..3 = features, block 4 = the three normal columns
    f32x16 acc2[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
    {
      const float* wt = lds + LW + 4 * h * SW + pl;
      const float* wt4 = lds + LW + 4 * h * SW + 128 + (pl < 4 ? pl : 4);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int fo = (32 * t + 8 * (r >> 2) + (r & 3)) * SW;
#pragma unroll
          for (int tb = 0; tb < 4; ++tb) acc2[tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[fo + 32 * tb], cs[t][r], acc2[tb], 0, 0, 0);
          acc2[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(wt4[fo], cs[t][r], acc2[4], 0, 0, 0);
        }
    }
    if (valid) {
#pragma unroll
      for (int tb = 0; tb < 4; ++tb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = {acc2[tb][4 * g], acc2[tb][4 * g + 1], acc2[tb][4 * g + 2], acc2[tb][4 * g + 3]};
          *reinterpret_cast<f32x4*>(d_feat + row * C + 32 * tb + 8 * g + 4 * h) = v;
        }
      if (h == 0) d_normals[row * 3] = acc2[4][0], d_normals[row * 3 + 1] = acc2[4][1], d_normals[row * 3 + 2] = acc2[4][2];
    }
  }
}

// D_e = dphi^T [feat | normals | 1] and E_e = a^T dpre over one chunk of one element's points.  Wave t owns output features
// 32 t .. 32 t + 31; a step consumes two points (lane-half kh = point parity).  Operands come straight from memory with the
// feature / column on the lane (128-byte rows per half wave), two register batches of WG_U steps in flight.
__global__ void __launch_bounds__(256) color_head_wgrad_kernel(
    const float* __restrict__ dphi_s, const float* __restrict__ a_s, const float* __restrict__ dpre_s,
    const float* __restrict__ feat, const float* __restrict__ normals, float* __restrict__ part, long long npe,
    long long chunk_pts) {
  const int tid = threadIdx.x, lane = tid & 63, t = tid >> 6, pl = lane & 31, kh = lane >> 5;
  const int e = blockIdx.y;
  const long long c0 = (long long)blockIdx.x * chunk_pts, c1 = c0 + chunk_pts < npe ? c0 + chunk_pts : npe;
  f32x16 acc[6];
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  struct Batch {
    float a1[WG_U], a2[WG_U], b[WG_U][5];
  };
  auto load = [&](long long p, Batch& q) {
#pragma unroll
    for (int s = 0; s < WG_U; ++s) {
      const long long pt = p + 2 * s + kh;
      const bool ok = pt < c1;
      const long long row = (long long)e * npe + (ok ? pt : c1 - 1);
      const float v1 = dphi_s[row * C + 32 * t + pl], v2 = a_s[row * C + 32 * t + pl];
      q.a1[s] = ok ? v1 : 0.f;
      q.a2[s] = ok ? v2 : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {   // (B zeroed on tail steps too: 0 * inf from the clamped row must not reach the sums)
        const float vb = feat[row * C + 32 * j + pl];
        q.b[s][j] = ok ? vb : 0.f;
      }
      // column block 4: normals (3) | 1 | dpre (3) | 0 ...
      float v4 = 0.f;
      if (pl < 3) v4 = normals[row * 3 + pl];
      else if (pl == 3) v4 = 1.f;
      else if (pl < 7) v4 = dpre_s[row * 4 + pl - 4];
      q.b[s][4] = ok ? v4 : 0.f;
    }
  };
  auto mma = [&](const Batch& q) {
#pragma unroll
    for (int s = 0; s < WG_U; ++s) {
#pragma unroll
      for (int j = 0; j < 5; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(q.a1[s], q.b[s][j], acc[j], 0, 0, 0);
      acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(q.a2[s], q.b[s][4], acc[5], 0, 0, 0);
    }
  };
  Batch qa, qb;
  if (c0 < c1) load(c0, qa);
  for (long long p = c0; p < c1; p += 4 * WG_U) {
    if (p + 2 * WG_U < c1) load(p + 2 * WG_U, qb);
    mma(qa);
    if (p + 2 * WG_U < c1) {
      if (p + 4 * WG_U < c1) load(p + 4 * WG_U, qa);
      mma(qb);
    }
  }
  float* out = part + ((long long)e * gridDim.x + blockIdx.x) * C * PCOLS;
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = 32 * t + 8 * (r >> 2) + 4 * kh + (r & 3);
      if (j < 4) out[f * PCOLS + 32 * j + pl] = acc[j][r];
      else if (j == 4) { if (pl < 4) out[f * PCOLS + 128 + pl] = acc[4][r]; }
      else if (pl >= 4 && pl < 7) out[f * PCOLS + 128 + pl] = acc[5][r];
    }
}

// One workgroup per output feature f: adds the chunk partials of every element in a fixed order and applies the identities
// of the file header.  blockDim = (PCOLS, 4): thread (col, y) walks chunks y, y + 4, ...
__global__ void __launch_bounds__(PCOLS * 4) color_head_finalize_kernel(
    const float* __restrict__ part, const float* __restrict__ part_b, const float* __restrict__ gamma, long long film_stride,
    const float* __restrict__ wv, const float* __restrict__ bv, float* __restrict__ d_wv, float* __restrict__ d_bv,
    float* __restrict__ d_wrgb, float* __restrict__ d_brgb, float* __restrict__ d_gamma, float* __restrict__ d_beta,
    long long d_film_stride, int B, int nchunk, long long n_part_b) {
  __shared__ float sm[4][PCOLS];
  __shared__ float red[PCOLS];
  const int f = blockIdx.x, col = threadIdx.x, y = threadIdx.y;
  float dw = 0.f;   // thread (col, 0): sum_e gamma_e D_e[col] (col < 132) or sum_e E_e (col 132..134)
  for (int e = 0; e < B; ++e) {
    float s = 0.f;
    if (col < 135)
      for (int c = y; c < nchunk; c += 4) s += part[(((long long)e * nchunk + c) * C + f) * PCOLS + col];
    sm[y][col] = s;
    __syncthreads();
    if (y == 0) {
      const float D = (sm[0][col] + sm[1][col]) + (sm[2][col] + sm[3][col]);
      const float ge = gamma[e * film_stride + f];
      dw += col < 132 ? ge * D : D;
      red[col] = col < CIN ? wv[f * CIN + col] * D : (col == CIN ? bv[f] * D : 0.f);
      if (col == CIN) d_beta[e * d_film_stride + f] = D;
    }
    __syncthreads();
    if (y == 0 && col == 0) {
      float g = 0.f;
      for (int i = 0; i <= CIN; ++i) g += red[i];
      d_gamma[e * d_film_stride + f] = g;
    }
    __syncthreads();
  }
  if (y == 0) {
    if (col < CIN) d_wv[f * CIN + col] = dw;
    else if (col == CIN) d_bv[f] = dw;
    else if (col < 135) d_wrgb[(col - 132) * C + f] = dw;
  }
  if (f == 0) {   // d brgb: the per-wave sums of dpre, fixed order
    const int tix = y * PCOLS + col;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (long long i = tix; i < n_part_b; i += PCOLS * 4) s0 += part_b[i * 4], s1 += part_b[i * 4 + 1], s2 += part_b[i * 4 + 2];
    __shared__ float rb[PCOLS * 4][3];
    rb[tix][0] = s0, rb[tix][1] = s1, rb[tix][2] = s2;
    __syncthreads();
    if (tix < 3) {
      float s = 0.f;
      for (int i = 0; i < PCOLS * 4; ++i) s += rb[i][tix];
      d_brgb[tix] = s;
    }
  }
}

struct BwdCarve {
  size_t dphi, a, dpre, part, part_b, total;
  int nchunk;
  long long chunk_pts, tiles;
};
BwdCarve carve(int B, long long npe) {
  BwdCarve w;
  const long long n = (long long)B * npe;
  w.tiles = (npe + TILE - 1) / TILE;
  // ~1024 workgroups for the point GEMM, chunks of whole register batches, never straddling a batch element
  long long cp = (n + 1023) / 1024;
  cp = (cp + 4 * WG_U - 1) / (4 * WG_U) * (4 * WG_U);
  if (cp < 128) cp = 128;
  w.chunk_pts = cp;
  w.nchunk = (int)((npe + cp - 1) / cp);
  auto al = [](size_t b) { return (b + 255) / 256 * 256; };
  w.dphi = 0;
  w.a = w.dphi + al((size_t)n * C * 4);
  w.dpre = w.a + al((size_t)n * C * 4);
  w.part = w.dpre + al((size_t)n * 16);
  w.part_b = w.part + al((size_t)B * w.nchunk * C * PCOLS * 4);
  w.total = w.part_b + al((size_t)B * w.tiles * 4 * 16);
  return w;
}

}  // namespace

extern "C" {

size_t oi_color_head_bwd_workspace_bytes(int B, long long n_per_elem) {
  if (B <= 0 || n_per_elem <= 0) return 0;
  return carve(B, n_per_elem).total;
}

int oi_color_head_fwd(const float* feat, const float* normals, const float* gamma, const float* beta, long long film_stride,
                      const float* wv, const float* bv, const float* wrgb, const float* brgb, float* rgb, int B,
                      long long n_per_elem, oi_stream_t stream) {
  OI_REQUIRE(feat && normals && gamma && beta && wv && bv && wrgb && brgb && rgb, "oi_color_head_fwd: null pointer");
  OI_REQUIRE(B > 0 && n_per_elem > 0 && film_stride >= C, "oi_color_head_fwd: B=%d n=%lld film_stride=%lld", B, n_per_elem, film_stride);
  const long long tiles = (n_per_elem + TILE - 1) / TILE;
  OI_REQUIRE(tiles < (1ll << 31) && B < 65536, "oi_color_head_fwd: grid too large");
  auto k = color_head_kernel<false>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, L_BYTES);
  hipLaunchKernelGGL(k, dim3((unsigned)tiles, B), dim3(256), L_BYTES, oi::as_stream(stream), feat, normals, gamma, beta, film_stride,
                     wv, bv, wrgb, brgb, rgb, n_per_elem, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  return oi::check_launch("oi_color_head_fwd");
}

int oi_color_head_bwd(const float* feat, const float* normals, const float* gamma, const float* beta, long long film_stride,
                      const float* wv, const float* bv, const float* wrgb, const float* brgb, const float* g_rgb, float* d_feat,
                      float* d_normals, float* d_gamma, float* d_beta, long long d_film_stride, float* d_wv, float* d_bv,
                      float* d_wrgb, float* d_brgb, void* workspace, size_t workspace_bytes, int B, long long n_per_elem,
                      oi_stream_t stream) {
  OI_REQUIRE(feat && normals && gamma && beta && wv && bv && wrgb && brgb && g_rgb, "oi_color_head_bwd: null input");
  OI_REQUIRE(d_feat && d_normals && d_gamma && d_beta && d_wv && d_bv && d_wrgb && d_brgb && workspace, "oi_color_head_bwd: null output");
  OI_REQUIRE(B > 0 && n_per_elem > 0 && film_stride >= C && d_film_stride >= C, "oi_color_head_bwd: B=%d n=%lld", B, n_per_elem);
  const BwdCarve w = carve(B, n_per_elem);
  OI_REQUIRE(workspace_bytes >= w.total, "oi_color_head_bwd: workspace of %zu bytes, need %zu", workspace_bytes, w.total);
  OI_REQUIRE(w.tiles < (1ll << 31) && B < 65536, "oi_color_head_bwd: grid too large");
  char* ws = reinterpret_cast<char*>(workspace);
  float* dphi_s = reinterpret_cast<float*>(ws + w.dphi);
  float* a_s = reinterpret_cast<float*>(ws + w.a);
  float* dpre_s = reinterpret_cast<float*>(ws + w.dpre);
  float* part = reinterpret_cast<float*>(ws + w.part);
  float* part_b = reinterpret_cast<float*>(ws + w.part_b);
  hipStream_t st = oi::as_stream(stream);
  auto k = color_head_kernel<true>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, L_BYTES);
  hipLaunchKernelGGL(k, dim3((unsigned)w.tiles, B), dim3(256), L_BYTES, st, feat, normals, gamma, beta, film_stride, wv, bv, wrgb,
                     brgb, nullptr, n_per_elem, g_rgb, d_feat, d_normals, dphi_s, a_s, dpre_s, part_b);
  int rc = oi::check_launch("oi_color_head_bwd(points)");
  if (rc != OI_OK) return rc;
  hipLaunchKernelGGL(color_head_wgrad_kernel, dim3(w.nchunk, B), dim3(256), 0, st, dphi_s, a_s, dpre_s, feat, normals, part,
                     n_per_elem, w.chunk_pts);
  rc = oi::check_launch("oi_color_head_bwd(weights)");
  if (rc != OI_OK) return rc;
  hipLaunchKernelGGL(color_head_finalize_kernel, dim3(C), dim3(PCOLS, 4), 0, st, part, part_b, gamma, film_stride, wv, bv, d_wv,
                     d_bv, d_wrgb, d_brgb, d_gamma, d_beta, d_film_stride, B, w.nchunk, (long long)B * w.tiles * 4);
  return oi::check_launch("oi_color_head_bwd(finalize)");
}

}  // extern "C"
