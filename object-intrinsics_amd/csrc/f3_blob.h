// The per-element blob of sdf_mlp_full3_kernel (mlp_fwd3.hip): LDS bytes [0, F3_BLOB) of a tile exactly as the kernel wants
// them -- FiLM rows of the 9 FiLM layers in REVOLUTIONS (A = gamma 2^-k_image / 2 pi, B = (gamma bias + beta) / 2 pi, G = gamma
// 2^-k_image), row 9 = G_7 w_sigma, the packed header's tables, max |G_l| per layer and of row 9.  Formed per batch element by
// film_blob_f3_kernel (first launch of an oi_sdf_mlp_fwd call) -- or, since round 6, by the FiLM workgroups of
// prep_render_kernel (render.hip), which have gamma_l / beta_l of their (element, layer) in registers anyway: one launch and its
// boundary less per render (oi_sdf_mlp_fwd_ex with OI_MLP_BLOB_READY).  One definition for both, so the bytes are the same.
// Own copies of the header offsets (render.hip does not include the MLP headers); mlp_fwd3.hip static_asserts them equal.
#pragma once
#include "oi_common.h"

namespace oif3 {
constexpr int FC = 128;
constexpr int HB_SIG = 512, HB_TABS_END = 1568, HB_BIAS = 1568, HB_WSCALE = 2720, HB_NL_SDF = 8;
constexpr int F3_FILM = 0;
constexpr int F3_FILM_ROW = 3 * FC * 4;                    // bytes per FiLM layer
constexpr int F3_TABS = F3_FILM + 10 * F3_FILM_ROW;        // 15360
constexpr int F3_GMAX = F3_TABS + HB_TABS_END * 4;         // [16] max |G_l| per FiLM layer (9: max |G7 w_sigma|)
constexpr int F3_BLOB = 22528;
static_assert(F3_GMAX + 64 <= F3_BLOB, "blob layout");

// Layer l (0..8) of element e, called by >= 128 threads of ONE workgroup (tid = threadIdx.x; all threads of the workgroup must
// call it: it contains a barrier).  gm / bt: gamma_l[f] / beta_l[f] of feature f = tid (tid < 128).  `red`: 8 floats of LDS.
// The workgroup that handles l == 0 also copies the header tables.
__device__ __forceinline__ void blob_layer(const float* __restrict__ hdr, float* __restrict__ out, int l, int tid, float gm, float bt,
                                           float* red) {
  constexpr float INV_2PI = 0.15915494309189533577f;
  float* film = out + F3_FILM / 4;
  float g_abs = 0.f, g9_abs = 0.f;
  if (tid < FC) {
    const int f = tid;
    const float wsc = l == 0 ? 1.f : hdr[HB_WSCALE + (l < HB_NL_SDF ? l - 1 : 14)];
    const float G = gm * wsc;
    film[l * (F3_FILM_ROW / 4) + f] = G * INV_2PI;
    film[l * (F3_FILM_ROW / 4) + FC + f] = fmaf(gm, hdr[HB_BIAS + l * FC + f], bt) * INV_2PI;
    film[l * (F3_FILM_ROW / 4) + 2 * FC + f] = G;
    g_abs = fabsf(G);
    if (l == 7) {
      const float g9 = G * hdr[HB_SIG + f];
      film[9 * (F3_FILM_ROW / 4) + f] = g9;
      g9_abs = fabsf(g9);
    }
  }
  if (l == 0) {
    float* tabs = out + F3_TABS / 4;
    for (int i = tid; i < HB_TABS_END; i += (int)blockDim.x) tabs[i] = hdr[i];
  }
  // max over the 128 features (two waves): a maximum is the same whatever the order
  const float m = oi::wave_max(g_abs), m9 = oi::wave_max(g9_abs);
  const int wave = tid >> 6;
  if ((tid & 63) == 0 && wave < 2) red[wave] = m, red[2 + wave] = m9;
  __syncthreads();
  if (tid == 0) {
    out[F3_GMAX / 4 + l] = fmaxf(red[0], red[1]);
    if (l == 7) out[F3_GMAX / 4 + 9] = fmaxf(red[2], red[3]);
  }
}
}  // namespace oif3
