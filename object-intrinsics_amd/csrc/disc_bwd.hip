// Backward kernels of the DC discriminator convolutions for gfx950.
//
// Replaces the cuDNN bwd-data / bwd-filter calls autograd issues for DCDiscriminator
// (reference src/models/discriminator.py:63-85) and, re-composed by oi_amd/autograd_conv.py, the
// double-backward the R1 penalty needs (src/loss/gan.py:5-14): the set {fwd, dgrad, wgrad} is closed
// under differentiation.
//
// Both are implicit GEMMs on v_mfma_f32_32x32x2_f32 in the same "pixel on the MFMA column" form as
// the forward kernel (csrc/disc.hip):
//   dgrad  D[c][m]      = sum_{n,tap} W[n][c][tap] * G[n][pix(m,tap)]      m = input pixel of ONE parity class
//   wgrad  D[n][(c,tap)] = sum_{pix}   G[n][pix]    * X[c][in(pix,tap)]
// K is split across wavefronts when the tile count is small; partial tiles are combined with fp32
// atomics into a zeroed output.
#include "oi_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------
// dgrad.  For stride S the taps that reach input row iy are ky = (iy + pad) % S + S*j; the input
// pixels are processed per parity class (a, b) = ((iy+pad)%S, (ix+pad)%S) so that a tile of 32 pixels
// shares its tap set: 4/S x 4/S taps per output channel.
// ------------------------------------------------------------------------------------------
struct DgradArgs {
  const float *g, *ref, *w;
  float* gx;
  float slope;
  const float* out_ref;  // non-null: the layer's input was a PRE-activation x; gx is returned with respect to it, i.e.
  float out_slope;       // multiplied by lrelu'(x) = (out_ref > 0 ? 1 : out_slope) (every split-K partial: the mask is linear)
  int B, Cin, H, W, Cout, Ho, Wo, pad, m_tiles, n_tiles, k_splits, couts_per_split, Hc, Wc;
  long long items;
};
struct WgradArgs {
  const float *g, *ref, *x;
  float* gw;
  float slope;
  float x_slope;  // != 1: x is a pre-activation, the operand is lrelu_{x_slope}(x), applied on load
  int accumulate, B, Cin, H, W, Cout, Ho, Wo, stride, pad, m_tiles, n_tiles, k_splits, pix_per_split;
  long long items;
};

// REF: compile-time "A.ref != null" -- tested at run time (a uniform branch per loaded value) it split the loop into blocks
// with a wait each, and the loads of a trip could not be issued together
template <int S, bool REF>
__device__ __forceinline__ void dgrad_body(const DgradArgs& A, long long block) {
  const float* __restrict__ g = A.g;
  const float* __restrict__ ref = A.ref;
  const float* __restrict__ w = A.w;
  float* __restrict__ gx = A.gx;
  const float slope = A.slope;
  const int B = A.B, Cin = A.Cin, H = A.H, W = A.W, Cout = A.Cout, Ho = A.Ho, Wo = A.Wo, pad = A.pad, m_tiles = A.m_tiles,
            n_tiles = A.n_tiles, k_splits = A.k_splits, couts_per_split = A.couts_per_split, Hc = A.Hc, Wc = A.Wc;
  // ref != null: the incoming gradient is taken through the LeakyReLU of the layer's forward output on load,
  // g_eff = ref > 0 ? g : slope * g (what lrelu_mask_mul_kernel would have written out first)
  constexpr int TJ = 4 / S;      // taps per axis
  const int lane = threadIdx.x & 63;
  const long long item = block * 4 + (threadIdx.x >> 6);
  const long long n_items = (long long)S * S * m_tiles * n_tiles * k_splits;
  // WG_SUM (k_splits % 4 == 0): the four waves of a workgroup are four consecutive K splits of ONE output tile (ks runs
  // fastest); their partial tiles are summed through LDS and leave as one atomic per element instead of four.  The
  // same-address atomics of the splits were the kernel: 19 of the 30 us of the 256 -> 512 layer's data gradient at two images
  // (plain stores instead: 11 us; a single loop trip instead of four: 28 us).
  const bool wg_sum = (k_splits & 3) == 0;
  if (item >= n_items) return;   // (with wg_sum a workgroup is complete or absent: n_items is a multiple of 4)
  const int ks = item % k_splits;
  const int nt = (item / k_splits) % n_tiles;
  const int mt = (item / ((long long)k_splits * n_tiles)) % m_tiles;
  const int cls = item / ((long long)k_splits * n_tiles * m_tiles);
  const int ca = cls / S, cb = cls % S;  // parity of (iy + pad), (ix + pad)

  const int h = lane >> 5, j = lane & 31;
  // pixels of this class: iy = S*yy + ((ca - pad) mod S), yy in [0, Hc)
  const int y_first = ((ca - pad) % S + S) % S, x_first = ((cb - pad) % S + S) % S;
  const int Mc = B * Hc * Wc;
  const int m = mt * 32 + j;
  const bool m_ok = m < Mc;
  const int mm = m_ok ? m : 0;
  const int xx = mm % Wc, yy = (mm / Wc) % Hc, b = mm / (Wc * Hc);
  const int iy = S * yy + y_first, ix = S * xx + x_first;
  const bool pix_ok = m_ok && iy < H && ix < W;
  // output pixel reached through tap (ky, kx): oy = (iy + pad - ky) / S
  const int oy_base = (iy + pad - ca) / S, ox_base = (ix + pad - cb) / S;  // for ky = ca, kx = cb; ky = ca + S*jy -> oy_base - jy

  const int c = nt * 32 + j;  // A side: input channel of this lane
  const bool c_ok = c < Cin;

  f32x16 acc = {0};
  const int n_begin = ks * couts_per_split, n_end = min(Cout, n_begin + couts_per_split);
  // U channel pairs per trip, all of their loads issued before the first MFMA (the loop is latency-bound: one wave per tile,
  // every load a dependent round trip -- 31 -> ~20 us for the largest layer of a 64^2 discriminator at two images)
  constexpr int U = S == 2 ? 4 : 2;
  for (int n0 = n_begin; n0 < n_end; n0 += 2 * U) {
    float a[U][TJ][TJ], bv[U][TJ][TJ];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int n = n0 + 2 * u + h;
      const bool n_ok = n < n_end;
      const float* wn = w + ((size_t)(n_ok ? n : 0) * Cin + (c_ok ? c : 0)) * 16;
      const size_t gn_off = ((size_t)b * Cout + (n_ok ? n : 0)) * Ho * Wo;
      const float* gn = g + gn_off;
#pragma unroll
      for (int jy = 0; jy < TJ; ++jy) {
        const int ky = ca + S * jy, oy = oy_base - jy;
#pragma unroll
        for (int jx = 0; jx < TJ; ++jx) {
          const int kx = cb + S * jx, ox = ox_base - jx;
          // loads are UNCONDITIONAL from clamped (always valid) addresses and the out-of-range lanes are zeroed by a select:
          // `cond ? p[i] : 0` compiles to a branch around every load (s_and_saveexec / s_cbranch_execz) with a wait every
          // third one -- the loop then is a chain of ~10 memory round trips per trip instead of one
          const float wt = wn[ky * 4 + kx];
          a[u][jy][jx] = (n_ok && c_ok) ? wt : 0.f;
          const bool ok = n_ok && pix_ok && oy >= 0 && oy < Ho && ox >= 0 && ox < Wo;
          const int gi = min(max(oy, 0), Ho - 1) * Wo + min(max(ox, 0), Wo - 1);
          float v = gn[gi];
          if constexpr (REF) {
            const float rv = ref[gn_off + gi];
            v = rv > 0.f ? v : v * slope;
          }
          bv[u][jy][jx] = ok ? v : 0.f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int jy = 0; jy < TJ; ++jy)
#pragma unroll
        for (int jx = 0; jx < TJ; ++jx) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][jy][jx], bv[u][jy][jx], acc, 0, 0, 0);
  }
  if (wg_sum) {  // (uniform per launch)
    __shared__ float part[3][16][64];
    const int wv = threadIdx.x >> 6;
    if (wv != 0)
#pragma unroll
      for (int rg = 0; rg < 16; ++rg) part[wv - 1][rg][lane] = acc[rg];
    __syncthreads();
    if (wv != 0) return;
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) acc[rg] += (part[0][rg][lane] + part[1][rg][lane]) + part[2][rg][lane];
  }
  if (!pix_ok) return;
#pragma unroll
  for (int rg = 0; rg < 16; ++rg) {
    const int cc = nt * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * h;
    if (cc >= Cin) continue;
    const size_t di = (((size_t)b * Cin + cc) * H + iy) * W + ix;
    float v = acc[rg];
    if (A.out_ref != nullptr && !(A.out_ref[di] > 0.f)) v *= A.out_slope;
    if (k_splits == 1 || (wg_sum && k_splits == 4)) gx[di] = v; else atomicAdd(gx + di, v);
  }
}

// ------------------------------------------------------------------------------------------
// wgrad.  A = G[n][pix] (rows = output channel), B = X gathered for ONE (c, ky) kernel row and the 4 kx
// taps spread over ... columns j = (c_local * 16 + ky*4 + kx): 32 columns = 2 input channels x 16 taps.
// ------------------------------------------------------------------------------------------
template <int S, bool REF>
__global__ void __launch_bounds__(256) conv4x4_dgrad_kernel(const DgradArgs A) { dgrad_body<S, REF>(A, blockIdx.x); }

template <bool REF>
__device__ __forceinline__ void wgrad_body(const WgradArgs& A, long long block) {
  const float* __restrict__ g = A.g;
  const float* __restrict__ ref = A.ref;
  const float* __restrict__ x = A.x;
  float* __restrict__ gw = A.gw;
  const float slope = A.slope;
  const int accumulate = A.accumulate, B = A.B, Cin = A.Cin, H = A.H, W = A.W, Cout = A.Cout, Ho = A.Ho, Wo = A.Wo,
            stride = A.stride, pad = A.pad, m_tiles = A.m_tiles, n_tiles = A.n_tiles, k_splits = A.k_splits,
            pix_per_split = A.pix_per_split;
  // ref / slope: as in the dgrad kernel;  accumulate: gw already holds a gradient (or zeros) that this one adds to
  const int lane = threadIdx.x & 63;
  const long long item = block * 4 + (threadIdx.x >> 6);
  const long long n_items = (long long)m_tiles * n_tiles * k_splits;
  if (item >= n_items) return;
  const int ks = item % k_splits;
  const int nt = (item / k_splits) % n_tiles;   // tile over (c, tap) columns: 2 channels per tile
  const int mt = item / ((long long)k_splits * n_tiles);  // tile over output channels
  const int h = lane >> 5, j = lane & 31;

  const int n = mt * 32 + j;  // A side: output channel
  const bool n_ok = n < Cout;
  const int c = nt * 2 + (j >> 4), tap = j & 15, ky = tap >> 2, kx = tap & 3;  // B side: (channel, tap)
  const bool c_ok = c < Cin;
  const int HW = Ho * Wo, P = B * HW;

  f32x16 acc = {0};
  const int p_begin = ks * pix_per_split, p_end = min(P, p_begin + pix_per_split);
  for (int p0 = p_begin; p0 < p_end; p0 += 16) {  // two 8-pixel k-steps per trip, their 16 loads before the 8 MFMAs
    float a[8], bv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int p = p0 + 8 * (q >> 2) + 4 * h + (q & 3);  // (half, u) -> k: both operands use the same map
      const bool p_ok = p < p_end;
      const int pp = p_ok ? p : 0;
      const int b = pp / HW, pl = pp % HW, oy = pl / Wo, ox = pl % Wo;
      const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
      // (unconditional loads from clamped addresses + selects: see the dgrad loop)
      const size_t gi = ((size_t)b * Cout + (n_ok ? n : 0)) * HW + pl;
      float av = g[gi];
      if constexpr (REF) {
        const float rv = ref[gi];
        av = rv > 0.f ? av : av * slope;
      }
      a[q] = (p_ok && n_ok) ? av : 0.f;
      float xv = x[(((size_t)b * Cin + (c_ok ? c : 0)) * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1)];
      if (A.x_slope != 1.0f) xv = xv > 0.f ? xv : xv * A.x_slope;
      bv[q] = (p_ok && c_ok && iy >= 0 && iy < H && ix >= 0 && ix < W) ? xv : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], bv[q], acc, 0, 0, 0);
  }
  const bool wg_sum = (k_splits & 3) == 0;  // four consecutive K splits of one tile per workgroup: summed in LDS first (see dgrad)
  if (wg_sum) {
    __shared__ float part[3][16][64];
    const int wv = threadIdx.x >> 6;
    if (wv != 0)
#pragma unroll
      for (int rg = 0; rg < 16; ++rg) part[wv - 1][rg][lane] = acc[rg];
    __syncthreads();
    if (wv != 0) return;
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) acc[rg] += (part[0][rg][lane] + part[1][rg][lane]) + part[2][rg][lane];
  }
  if (!c_ok) return;
  // D: column = lane & 31 = (c, tap), row = channel n
#pragma unroll
  for (int rg = 0; rg < 16; ++rg) {
    const int nn = mt * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * h;
    if (nn >= Cout) continue;
    float* dst = gw + ((size_t)nn * Cin + c) * 16 + tap;
    if ((k_splits == 1 || (wg_sum && k_splits == 4)) && !accumulate) *dst = acc[rg]; else atomicAdd(dst, acc[rg]);
  }
}

template <bool REF>
__global__ void __launch_bounds__(256) conv4x4_wgrad_kernel(const WgradArgs A) { wgrad_body<REF>(A, blockIdx.x); }

// Both gradients of one layer in ONE launch (they share the incoming gradient and nothing else): the first `blocks_d`
// workgroups run data-gradient items, the rest weight-gradient items.  At batch 1 a discriminator step is a chain of ~100
// launches of a few microseconds each; every launch less is ~6 us of a 0.7 ms step.
template <int S, bool REF>
__global__ void __launch_bounds__(256) conv4x4_bwd_kernel(const DgradArgs D, const WgradArgs Wg, int blocks_d) {
  if ((int)blockIdx.x < blocks_d) dgrad_body<S, REF>(D, blockIdx.x);
  else wgrad_body<REF>(Wg, (long long)blockIdx.x - blocks_d);
}

// out = ref > 0 ? v : slope * v     (LeakyReLU forward with ref = v, and its gradient with ref = output)
__global__ void lrelu_mask_mul_kernel(const float* __restrict__ v, const float* __restrict__ ref,
                                      float* __restrict__ out, long long n, float slope) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = v[i];
  out[i] = ref[i] > 0.f ? x : x * slope;
}

// bias gradient: gb[n] = sum_{b, pix} g[b][n][pix]
__global__ void channel_sum_kernel(const float* __restrict__ g, float* __restrict__ gb, int B, int C, int HW) {
  const int n = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < B * HW; i += blockDim.x) s += g[((size_t)(i / HW) * C + n) * HW + i % HW];
  s = oi::wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) gb[n] = red[0] + red[1] + red[2] + red[3];
}

}  // namespace

extern "C" {

int oi_conv4x4_dgrad(const float* g, const float* w, float* gx, int B, int Cin, int H, int W, int Cout, int stride,
                     int pad, oi_stream_t stream) {
  return oi_conv4x4_dgrad_masked(g, nullptr, 1.f, w, gx, B, Cin, H, W, Cout, stride, pad, stream);
}

static void launch_dgrad(const DgradArgs& A, int stride, hipStream_t st) {
  const dim3 grid(oi::cdiv(A.items, 4)), blk(256);
  if (stride == 2 && A.ref != nullptr) hipLaunchKernelGGL((conv4x4_dgrad_kernel<2, true>), grid, blk, 0, st, A);
  else if (stride == 2) hipLaunchKernelGGL((conv4x4_dgrad_kernel<2, false>), grid, blk, 0, st, A);
  else if (A.ref != nullptr) hipLaunchKernelGGL((conv4x4_dgrad_kernel<1, true>), grid, blk, 0, st, A);
  else hipLaunchKernelGGL((conv4x4_dgrad_kernel<1, false>), grid, blk, 0, st, A);
}

static int make_dgrad(DgradArgs& A, const float* g, const float* ref, float slope, const float* w, float* gx, int B, int Cin,
                      int H, int W, int Cout, int stride, int pad, const float* out_ref = nullptr, float out_slope = 1.f) {
  OI_REQUIRE(g && w && gx, "oi_conv4x4_dgrad: null pointer");
  OI_REQUIRE(stride == 1 || stride == 2, "oi_conv4x4_dgrad: stride %d (1 or 2 supported)", stride);
  const int Ho = (H + 2 * pad - 4) / stride + 1, Wo = (W + 2 * pad - 4) / stride + 1;
  OI_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Ho > 0 && Wo > 0, "oi_conv4x4_dgrad: bad shape");
  const int Hc = oi::cdiv(H, stride), Wc = oi::cdiv(W, stride);
  const int m_tiles = oi::cdiv((long long)B * Hc * Wc, 32), n_tiles = oi::cdiv(Cin, 32);
  int k_splits = 1;
  const long long tiles = (long long)stride * stride * m_tiles * n_tiles;
  while (tiles * k_splits < 2048 && Cout / (k_splits * 2) >= 16) k_splits *= 2;
  int cps = oi::cdiv(Cout, k_splits);
  cps += cps & 1;
  k_splits = oi::cdiv(Cout, cps);
  A = DgradArgs{g, ref, w, gx, slope, out_slope != 1.f ? out_ref : nullptr, out_slope, B, Cin, H, W, Cout, Ho, Wo, pad, m_tiles,
                n_tiles, k_splits, cps, Hc, Wc, tiles * k_splits};
  return OI_OK;
}

static int make_wgrad(WgradArgs& A, const float* g, const float* ref, float slope, const float* x, float* gw, int accumulate,
                      int B, int Cin, int H, int W, int Cout, int stride, int pad, float x_slope = 1.f) {
  OI_REQUIRE(g && x && gw, "oi_conv4x4_wgrad: null pointer");
  const int Ho = (H + 2 * pad - 4) / stride + 1, Wo = (W + 2 * pad - 4) / stride + 1;
  OI_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Ho > 0 && Wo > 0 && stride > 0, "oi_conv4x4_wgrad: bad shape");
  const int m_tiles = oi::cdiv(Cout, 32), n_tiles = oi::cdiv(Cin, 2);
  const int P = B * Ho * Wo;
  int k_splits = 1;
  const long long tiles = (long long)m_tiles * n_tiles;
  while (tiles * k_splits < 2048 && P / (k_splits * 2) >= 64) k_splits *= 2;
  int pps = oi::cdiv(P, k_splits);
  pps = (pps + 7) & ~7;
  k_splits = oi::cdiv(P, pps);
  A = WgradArgs{g, ref, x, gw, slope, x_slope, accumulate, B, Cin, H, W, Cout, Ho, Wo, stride, pad, m_tiles, n_tiles, k_splits, pps,
                tiles * k_splits};
  return OI_OK;
}

int oi_conv4x4_dgrad_masked(const float* g, const float* ref, float slope, const float* w, float* gx, int B, int Cin, int H,
                            int W, int Cout, int stride, int pad, oi_stream_t stream) {
  DgradArgs A;
  int rc = make_dgrad(A, g, ref, slope, w, gx, B, Cin, H, W, Cout, stride, pad);
  if (rc != OI_OK) return rc;
  hipStream_t st = oi::as_stream(stream);
  // rows/cols no tap reaches (e.g. the last row when (H + 2 pad - 4) % stride != 0) keep the zero fill
  hipError_t e = oi::zero_output_async(gx, (size_t)B * Cin * H * W, st);
  if (e != hipSuccess) return oi::fail(OI_ERR_LAUNCH, "oi_conv4x4_dgrad: memset: %s", hipGetErrorString(e));
  launch_dgrad(A, stride, st);
  return oi::check_launch("oi_conv4x4_dgrad");
}

int oi_conv4x4_wgrad(const float* g, const float* x, float* gw, int B, int Cin, int H, int W, int Cout, int stride,
                     int pad, oi_stream_t stream) {
  return oi_conv4x4_wgrad_masked(g, nullptr, 1.f, x, gw, 0, B, Cin, H, W, Cout, stride, pad, stream);
}

int oi_conv4x4_wgrad_masked(const float* g, const float* ref, float slope, const float* x, float* gw, int accumulate, int B,
                            int Cin, int H, int W, int Cout, int stride, int pad, oi_stream_t stream) {
  WgradArgs A;
  int rc = make_wgrad(A, g, ref, slope, x, gw, accumulate, B, Cin, H, W, Cout, stride, pad);
  if (rc != OI_OK) return rc;
  hipStream_t st = oi::as_stream(stream);
  if (A.k_splits > 1 && !accumulate) {
    hipError_t e = oi::zero_output_async(gw, (size_t)Cout * Cin * 16, st);
    if (e != hipSuccess) return oi::fail(OI_ERR_LAUNCH, "oi_conv4x4_wgrad: memset: %s", hipGetErrorString(e));
  }
  if (A.ref != nullptr) hipLaunchKernelGGL(conv4x4_wgrad_kernel<true>, dim3(oi::cdiv(A.items, 4)), dim3(256), 0, st, A);
  else hipLaunchKernelGGL(conv4x4_wgrad_kernel<false>, dim3(oi::cdiv(A.items, 4)), dim3(256), 0, st, A);
  return oi::check_launch("oi_conv4x4_wgrad");
}

int oi_conv4x4_bwd_masked(const float* g, const float* ref, float slope, const float* w, const float* x, float* gx, float* gw,
                          int accumulate, int B, int Cin, int H, int W, int Cout, int stride, int pad, oi_stream_t stream) {
  return oi_conv4x4_bwd_pre(g, ref, slope, w, x, 1.f, gx, gw, accumulate, B, Cin, H, W, Cout, stride, pad, stream);
}

int oi_conv4x4_dgrad_pre(const float* g, const float* w, const float* x, float x_slope, float* gx, int B, int Cin, int H, int W,
                         int Cout, int stride, int pad, oi_stream_t stream) {
  OI_REQUIRE(x != nullptr || x_slope == 1.f, "oi_conv4x4_dgrad_pre: x_slope without x");
  DgradArgs A;
  int rc = make_dgrad(A, g, nullptr, 1.f, w, gx, B, Cin, H, W, Cout, stride, pad, x, x_slope);
  if (rc != OI_OK) return rc;
  hipStream_t st = oi::as_stream(stream);
  hipError_t e = oi::zero_output_async(gx, (size_t)B * Cin * H * W, st);
  if (e != hipSuccess) return oi::fail(OI_ERR_LAUNCH, "oi_conv4x4_dgrad: memset: %s", hipGetErrorString(e));
  launch_dgrad(A, stride, st);
  return oi::check_launch("oi_conv4x4_dgrad_pre");
}

int oi_conv4x4_bwd_pre(const float* g, const float* ref, float slope, const float* w, const float* x, float x_slope, float* gx,
                       float* gw, int accumulate, int B, int Cin, int H, int W, int Cout, int stride, int pad,
                       oi_stream_t stream) {
  DgradArgs D;
  WgradArgs Wg;
  int rc = make_dgrad(D, g, ref, slope, w, gx, B, Cin, H, W, Cout, stride, pad, x, x_slope);
  if (rc == OI_OK) rc = make_wgrad(Wg, g, ref, slope, x, gw, accumulate, B, Cin, H, W, Cout, stride, pad, x_slope);
  if (rc != OI_OK) return rc;
  hipStream_t st = oi::as_stream(stream);
  hipError_t e = oi::zero_output_async(gx, (size_t)B * Cin * H * W, st);
  if (e == hipSuccess && !accumulate) e = oi::zero_output_async(gw, (size_t)Cout * Cin * 16, st);  // (the fused kernel always adds)
  if (e != hipSuccess) return oi::fail(OI_ERR_LAUNCH, "oi_conv4x4_bwd: memset: %s", hipGetErrorString(e));
  Wg.accumulate = 1;
  const int blocks_d = oi::cdiv(D.items, 4), blocks_w = oi::cdiv(Wg.items, 4);
  const dim3 grid(blocks_d + blocks_w), blk(256);
  if (stride == 2 && ref != nullptr) hipLaunchKernelGGL((conv4x4_bwd_kernel<2, true>), grid, blk, 0, st, D, Wg, blocks_d);
  else if (stride == 2) hipLaunchKernelGGL((conv4x4_bwd_kernel<2, false>), grid, blk, 0, st, D, Wg, blocks_d);
  else if (ref != nullptr) hipLaunchKernelGGL((conv4x4_bwd_kernel<1, true>), grid, blk, 0, st, D, Wg, blocks_d);
  else hipLaunchKernelGGL((conv4x4_bwd_kernel<1, false>), grid, blk, 0, st, D, Wg, blocks_d);
  return oi::check_launch("oi_conv4x4_bwd");
}

int oi_lrelu_mask_mul(const float* v, const float* ref, float* out, long long n, float slope, oi_stream_t stream) {
  OI_REQUIRE(v && ref && out && n > 0, "oi_lrelu_mask_mul: bad argument");
  hipLaunchKernelGGL(lrelu_mask_mul_kernel, dim3(oi::cdiv(n, 256)), dim3(256), 0, oi::as_stream(stream), v, ref, out,
                     n, slope);
  return oi::check_launch("oi_lrelu_mask_mul");
}

int oi_channel_sum(const float* g, float* gb, int B, int C, int HW, oi_stream_t stream) {
  OI_REQUIRE(g && gb && B > 0 && C > 0 && HW > 0, "oi_channel_sum: bad argument");
  hipLaunchKernelGGL(channel_sum_kernel, dim3(C), dim3(256), 0, oi::as_stream(stream), g, gb, B, C, HW);
  return oi::check_launch("oi_channel_sum");
}

}  // extern "C"
