// ADA-augmented DC discriminator kernels for gfx950.
//
// Replaces:
//   DCDiscriminator.forward: nn.Conv2d(4,2,1,bias=False)+LeakyReLU(0.2) blocks and the 4x4 valid
//     head (reference src/models/discriminator.py:63-85; cuDNN there)
//   upfirdn2d plugin (src/third_party/ada/torch_utils/ops/upfirdn2d.cu:29-200, upfirdn2d.cpp:16-94)
//   F.affine_grid + grid_sample (ada/augment.py:297-298, grid_sample_gradfix.py:33-97), reflect pad
//     (augment.py:286)
#include <cstdlib>

#include "oi_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------
// 4x4 convolution as an implicit GEMM on the fp32 matrix cores.
//   D[n][m] = sum_k W[n][k] * im2col[m][k],  k = (cin, ky, kx), m = (b, oy, ox)
// One wavefront = one 32(channel) x 32(pixel) tile of v_mfma_f32_32x32x2_f32 over one K split.
// The output pixel sits on the MFMA column (lane & 31): coalesced stores, and each lane gathers its
// own input window.  K is consumed one 4-tap kernel row per lane-half per 4 MFMAs (the A and B
// fragments use the same (half, tap) -> k map, so any K order is valid).
// Small-M layers (late blocks at batch 1-4) are weight-bandwidth bound: K is split across
// wavefronts (grid) and partial tiles are combined with fp32 atomics, activation in a second pass.
// (Measured alternatives for the combine, both slower on this multi-XCD part: a grid-level "last arriver" --
// device-scope release/acquire between workgroups costs an L2 write-back + invalidate per workgroup, 2-4x the extra
// launches; and splits as the waves of ONE workgroup with an LDS reduction -- a tile's whole K then streams through
// a single CU, 16 busy CUs instead of 256.)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
conv4x4_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                   float* __restrict__ y, int B, int Cin, int H, int W, int Cout, int Ho, int Wo, int stride,
                   int pad, float slope, int m_tiles, int n_tiles, int k_splits, int rows_per_split, float x_slope,
                   float* __restrict__ zero_ptr, long long zero_n) {
  // side job of the FIRST layer of a forward-only chain (OI_CONV_ZERO_TAIL): clear the part of the arena that the later
  // layers' split-K sums accumulate into -- one launch less than a separate fill
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < zero_n; i += (long long)gridDim.x * blockDim.x)
    zero_ptr[i] = 0.f;
  const int lane = threadIdx.x & 63;
  const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long n_items = (long long)m_tiles * n_tiles * k_splits;
  if (item >= n_items) return;
  const int ks = item % k_splits;
  const int nt = (item / k_splits) % n_tiles;
  const int mt = item / ((long long)k_splits * n_tiles);

  const int h = lane >> 5, j = lane & 31;
  const int M = B * Ho * Wo;
  // B-operand side: this lane's output pixel
  const int m = mt * 32 + j;
  const bool m_ok = m < M;
  const int mm = m_ok ? m : 0;
  const int ox = mm % Wo, oy = (mm / Wo) % Ho, b = mm / (Wo * Ho);
  const int ix0 = ox * stride - pad, iy0 = oy * stride - pad;
  // A-operand side: this lane's output channel
  const int n = nt * 32 + j;
  const bool n_ok = n < Cout;
  const float* wrow = w + (size_t)(n_ok ? n : 0) * Cin * 16;
  const float* xb = x + (size_t)b * Cin * H * W;

  f32x16 acc = {0};
  const int krows = Cin * 4;  // kernel rows (cin, ky)
  const int r_begin = ks * rows_per_split;
  const int r_end = min(krows, r_begin + rows_per_split);
  // four kernel-row pairs per trip: all 8 loads of a trip are issued before its 16 MFMAs (the loop is latency-bound)
  for (int r0 = r_begin; r0 < r_end; r0 += 8) {
    float a[4][4], bv[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + 2 * u + h;  // this lane-half's kernel row
      const bool r_ok = r < r_end;
      const int rs = r_ok ? r : r_begin;  // (a valid row for the lanes past the end: loaded, then zeroed by the selects)
      const int cin = rs >> 2, ky = rs & 3;
      // loads are UNCONDITIONAL from clamped addresses, out-of-range lanes are zeroed by selects: `cond ? load : 0` became a
      // branch around every load with a wait every few -- a chain of memory round trips per trip
      const float4 wv = *reinterpret_cast<const float4*>(wrow + rs * 4);
      const bool w_ok = r_ok && n_ok;
      a[u][0] = w_ok ? wv.x : 0.f; a[u][1] = w_ok ? wv.y : 0.f; a[u][2] = w_ok ? wv.z : 0.f; a[u][3] = w_ok ? wv.w : 0.f;
      const int iy = iy0 + ky;
      const bool row_ok = r_ok && m_ok && iy >= 0 && iy < H;
      const float* xr = xb + ((size_t)cin * H + min(max(iy, 0), H - 1)) * W;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xv = xr[min(max(ix0 + k, 0), W - 1)];
        bv[u][k] = (row_ok && ix0 + k >= 0 && ix0 + k < W) ? xv : 0.f;
      }
    }
    if (x_slope != 1.0f) {  // the producer left its LeakyReLU to this layer's loads (wave-uniform branch)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) bv[u][k] = bv[u][k] > 0.f ? bv[u][k] : bv[u][k] * x_slope;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][k], bv[u][k], acc, 0, 0, 0);
  }
  // The four waves of a workgroup are four consecutive K splits of ONE output tile when k_splits % 4 == 0 (ks runs fastest):
  // their partial tiles are summed through LDS and leave as one atomic per element instead of four -- the same-address
  // atomics of the splits, not the loop, were most of these kernels' time (csrc/disc_bwd.hip, dgrad)
  const bool wg_sum = (k_splits & 3) == 0;
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
  // D layout: column = lane & 31 (pixel), row = (reg & 3) + 8 * (reg >> 2) + 4 * h (channel)
  if (!m_ok) return;
#pragma unroll
  for (int rg = 0; rg < 16; ++rg) {
    const int nn = nt * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * h;
    if (nn >= Cout) continue;
    float* dst = y + (((size_t)b * Cout + nn) * Ho + oy) * Wo + ox;
    float v = acc[rg];
    if (k_splits == 1) {
      if (bias != nullptr) v += bias[nn];
      *dst = v > 0.f ? v : v * slope;
    } else if (wg_sum && k_splits == 4) {
      *dst = v;   // (all splits were in this workgroup; bias / activation follow in the launcher's pass, as for any split)
    } else {
      atomicAdd(dst, v);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Large-batch form of the same convolution (M = B * Ho * Wo >= 2048 pixels: discriminator batches of 16+, e.g. the
// B = 64 row of SURVEY.md 8d): an LDS-tiled implicit GEMM on the fp16 matrix cores with fp32 operands split into two
// fp16 limbs (hi + lo = 22 mantissa bits, three v_mfma_f32_32x32x16_f16 per product, fp32 accumulate -- the F16X3 mode of
// the MLP kernels; conv inputs are images / LeakyReLU activations of O(1), weights O(0.1): no scaling needed, and an
// |x| >= 65504 would surface as inf, not as a silent error).
//   workgroup tile   64 channels x TM pixels (TM = 128, or 64 when that is what fills the chip) x 64 taps per K chunk
//   wave (4)         32 channels x TM/2 pixels: TM/64 accumulator blocks, 3 MFMAs per block and 16-tap step
//   LDS              both operands in MFMA-fragment order [block][k-step][lane][8 x fp16], hi plane | lo plane: every
//                    operand read is one conflict-free ds_read_b128; every staged octet (8 consecutive taps = two kernel
//                    rows of one input channel) is one ds_write_b128 per plane
//   global           a thread gathers its octet's 2 x 4 input pixels (bounds-checked, LeakyReLU of the producer applied on
//                    the fly) / reads its 2 x float4 of weights, splits to limbs once, and every limb is then used by 32
//                    (weights: TM) MFMA columns: 16x (TM x) fewer global loads and conversions per MAC than the per-wave
//                    gather of conv4x4_fwd_kernel.  K is never split: summation order is fixed (reproducible).
// ------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split8(const float (&v)[8], f32x4& hi, f32x4& lo) {
  f16x8 h, l;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    h[i] = (_Float16)v[i];
    l[i] = (_Float16)(v[i] - (float)h[i]);
  }
  hi = __builtin_bit_cast(f32x4, h);
  lo = __builtin_bit_cast(f32x4, l);
}

template <int TM>
__global__ void __launch_bounds__(256)
conv4x4_tiled_f16x3_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                           float* __restrict__ y, int B, int Cin, int H, int W, int Cout, int Ho, int Wo, int stride,
                           int pad, float slope, float x_slope, int k_per_split) {
  constexpr int TN = 64, TK = 64, MB = TM / 32;        // MB pixel blocks per workgroup, MB / 2 per wave
  constexpr int A_PLANE = (TN / 32) * (TK / 16) * 1024;  // 8 KiB
  constexpr int B_PLANE = MB * (TK / 16) * 1024;         // 16 / 8 KiB
  __shared__ __attribute__((aligned(16))) char lds[2 * A_PLANE + 2 * B_PLANE];
  char* sa = lds;
  char* sb = lds + 2 * A_PLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wm = wave >> 1;
  const int n0 = blockIdx.y * TN;
  const long long m0 = (long long)blockIdx.x * TM;
  const long long M = (long long)B * Ho * Wo;
  const int K = Cin * 16;
  const int k_begin = blockIdx.z * k_per_split, k_end = min(K, k_begin + k_per_split);

  // ---- staging roles ----
  // weights: octet (n, q), q = 8-tap group of the chunk: 64 x 8 = 512 octets, 2 per thread (o = tid + 256 i: n = o & 63)
  // pixels:  octet (m, q): TM x 8 octets, TM / 32 per thread (o = tid + 256 i: pixel o % TM, octet o / TM); consecutive
  //          threads take consecutive pixels (coalesced input rows)
  int piy0[MB], pix0[MB];
  const float* pxb[MB];
  bool pok[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    const long long m = m0 + (tid + 256 * i) % TM;
    pok[i] = m < M;
    const long long mm = pok[i] ? m : 0;
    const int ox = mm % Wo, oy = (mm / Wo) % Ho;
    pxb[i] = x + (size_t)(mm / ((long long)Wo * Ho)) * Cin * H * W;
    piy0[i] = oy * stride - pad;
    pix0[i] = ox * stride - pad;
  }
  f32x16 acc[MB / 2];
#pragma unroll
  for (int i = 0; i < MB / 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // raw fp32 octets of one chunk in registers: requested one chunk ahead, so the gather's latency runs under the MFMAs
  float wv[2][8], xv[MB][8];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o = tid + 256 * i;
      const int n = n0 + (o & 63), k = k0 + 8 * (o >> 6);
#pragma unroll
      for (int e = 0; e < 8; ++e) wv[i][e] = 0.f;
      if (n < Cout && k < k_end) {
        const float4 a = *reinterpret_cast<const float4*>(w + (size_t)n * K + k);
        const float4 c = *reinterpret_cast<const float4*>(w + (size_t)n * K + k + 4);
        wv[i][0] = a.x; wv[i][1] = a.y; wv[i][2] = a.z; wv[i][3] = a.w;
        wv[i][4] = c.x; wv[i][5] = c.y; wv[i][6] = c.z; wv[i][7] = c.w;
      }
    }
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const int k = k0 + 8 * ((tid + 256 * i) / TM);
      const int cin = k >> 4, ky0 = (k >> 2) & 3;  // ky0 = 0 or 2: the octet is kernel rows ky0, ky0 + 1 of channel cin
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[i][e] = 0.f;
      if (pok[i] && k < k_end) {
        const float* xc = pxb[i] + (size_t)cin * H * W;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int iy = piy0[i] + ky0 + r;
          if (iy >= 0 && iy < H) {
            const float* xr = xc + (size_t)iy * W;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const int ix = pix0[i] + c;
              if (ix >= 0 && ix < W) xv[i][4 * r + c] = xr[ix];
            }
          }
        }
      }
    }
  };
  fetch(k_begin);
  for (int k0 = k_begin; k0 < k_end; k0 += TK) {
    __syncthreads();  // the previous chunk's readers are done
    // ---- registers -> limbs -> LDS, fragment order [block][k-step][lane = (row & 31) + 32 * (octet & 1)][8] ----
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o = tid + 256 * i;
      const int nl = o & 63, q = o >> 6;
      f32x4 hi, lo;
      split8(wv[i], hi, lo);
      const int off = (((nl >> 5) * (TK / 16) + (q >> 1)) * 64 + (nl & 31) + 32 * (q & 1)) * 16;
      *reinterpret_cast<f32x4*>(sa + off) = hi;
      *reinterpret_cast<f32x4*>(sa + A_PLANE + off) = lo;
    }
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const int o = tid + 256 * i;
      const int ml = o % TM, q = o / TM;
      if (x_slope != 1.0f) {  // the producer left its LeakyReLU to this layer's loads
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[i][e] = xv[i][e] > 0.f ? xv[i][e] : xv[i][e] * x_slope;
      }
      f32x4 hi, lo;
      split8(xv[i], hi, lo);
      const int off = (((ml >> 5) * (TK / 16) + (q >> 1)) * 64 + (ml & 31) + 32 * (q & 1)) * 16;
      *reinterpret_cast<f32x4*>(sb + off) = hi;
      *reinterpret_cast<f32x4*>(sb + B_PLANE + off) = lo;
    }
    __syncthreads();
    if (k0 + TK < k_end) fetch(k0 + TK);
    // ---- MFMAs: this wave's 32 channels x TM / 2 pixels ----
#pragma unroll
    for (int ks = 0; ks < TK / 16; ++ks) {
      const f16x8 ah = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(sa + ((wn * (TK / 16) + ks) * 64 + lane) * 16));
      const f16x8 al = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(sa + A_PLANE + ((wn * (TK / 16) + ks) * 64 + lane) * 16));
#pragma unroll
      for (int j = 0; j < MB / 2; ++j) {
        const int mb = wm * (MB / 2) + j;
        const f16x8 bh = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(sb + ((mb * (TK / 16) + ks) * 64 + lane) * 16));
        const f16x8 bl = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(sb + B_PLANE + ((mb * (TK / 16) + ks) * 64 + lane) * 16));
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[j], 0, 0, 0);
      }
    }
  }
  // ---- store: D column = lane & 31 (pixel), row = (reg & 3) + 8 (reg >> 2) + 4 h (channel) ----
  const int h = lane >> 5, jl = lane & 31;
  const bool split = gridDim.z > 1;  // exactly two K halves add into a zeroed output: a + b is order-independent
#pragma unroll
  for (int j = 0; j < MB / 2; ++j) {
    const long long m = m0 + (wm * (MB / 2) + j) * 32 + jl;
    if (m >= M) continue;
    const int ox = m % Wo, oy = (m / Wo) % Ho, b = m / ((long long)Wo * Ho);
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int nn = n0 + wn * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * h;
      if (nn >= Cout) continue;
      float* dst = y + (((size_t)b * Cout + nn) * Ho + oy) * Wo + ox;
      float v = acc[j][rg];
      if (split) {
        atomicAdd(dst, v);
      } else {
        if (bias != nullptr) v += bias[nn];
        *dst = v > 0.f ? v : v * slope;
      }
    }
  }
}

__global__ void bias_lrelu_kernel(float* __restrict__ y, const float* __restrict__ bias, long long n, int Cout,
                                  int hw, float slope) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = y[i];
  if (bias != nullptr) v += bias[(i / hw) % Cout];
  y[i] = v > 0.f ? v : v * slope;
}

// ------------------------------------------------------------------------------------------
// Stand-alone plugin ops for a caller that keeps the reference's own Python layers:
//   fused_bias_act(input, bias, refer, act, grad, alpha, scale)   stylesdf/op/fused_bias_act.cpp:11-20, kernel .cu:18-49
//   grid_sample(input, grid)  (bilinear, zeros padding, align_corners=False) + the aten::grid_sampler_2d_backward the
//   reference's autograd wrapper calls                             ada/torch_utils/ops/grid_sample_gradfix.py:33-66
// Both are HBM-bound elementwise / gather kernels: coalesced along the innermost dimension, nothing staged.
// ------------------------------------------------------------------------------------------
// y = (x + bias) * g * scale with a per-element gain g in {1, alpha, 0} selected without branches from the (act, grad) code:
//   value            (grad 0): gate = the biased input itself;        g = gate > 0 ? 1 : slope
//   first derivative (grad 1): gate = `refer` (the forward's output); g = gate > 0 ? 1 : slope
//   second derivative (grad 2) of either activation is zero;          g = 0
// slope = alpha for the leaky relu (act 3) and 1 for the linear activation (act 1, any other code).
__global__ void fused_bias_act_kernel(float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ bias,
                                      const float* __restrict__ refer, float slope, float zero_or_one, bool gate_on_refer,
                                      float scale, long long n, long long inner, int channels) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = in[i] + (bias != nullptr ? bias[(i / inner) % channels] : 0.f);
  const float gate = gate_on_refer ? (refer != nullptr ? refer[i] : 0.f) : v;
  out[i] = v * (gate > 0.f ? 1.f : slope) * zero_or_one * scale;
}

// unnormalise (align_corners = False): [-1, 1] -> [-0.5, size - 0.5]
__device__ __forceinline__ float gs_src(float g, int size) { return ((g + 1.f) * size - 1.f) * 0.5f; }

__global__ void grid_sample_fwd_kernel(const float* __restrict__ x, const float* __restrict__ grid, float* __restrict__ y,
                                       int N, int C, int Hi, int Wi, int Ho, int Wo) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = (long long)N * C * Ho * Wo;
  if (idx >= n) return;
  const int ox = idx % Wo, oy = (idx / Wo) % Ho, c = (idx / ((long long)Wo * Ho)) % C;
  const int b = idx / ((long long)Wo * Ho * C);
  const float* gp = grid + (((size_t)b * Ho + oy) * Wo + ox) * 2;
  const float ix = gs_src(gp[0], Wi), iy = gs_src(gp[1], Hi);
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float tx = ix - fx, ty = iy - fy;
  const float* xp = x + ((size_t)b * C + c) * Hi * Wi;
  float v = 0.f;
  if (y0 >= 0 && y0 < Hi) {
    if (x0 >= 0 && x0 < Wi) v += xp[y0 * Wi + x0] * (1.f - tx) * (1.f - ty);
    if (x0 + 1 >= 0 && x0 + 1 < Wi) v += xp[y0 * Wi + x0 + 1] * tx * (1.f - ty);
  }
  if (y0 + 1 >= 0 && y0 + 1 < Hi) {
    if (x0 >= 0 && x0 < Wi) v += xp[(y0 + 1) * Wi + x0] * (1.f - tx) * ty;
    if (x0 + 1 >= 0 && x0 + 1 < Wi) v += xp[(y0 + 1) * Wi + x0 + 1] * tx * ty;
  }
  y[idx] = v;
}

// one thread per output pixel, looping over the channels: the grid gradient is a sum over channels (no atomics for it)
__global__ void grid_sample_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                       const float* __restrict__ grid, float* __restrict__ gx, float* __restrict__ ggrid,
                                       int N, int C, int Hi, int Wi, int Ho, int Wo) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = (long long)N * Ho * Wo;
  if (idx >= n) return;
  const int ox = idx % Wo, oy = (idx / Wo) % Ho;
  const int b = idx / ((long long)Wo * Ho);
  const float* gp = grid + (size_t)idx * 2;
  const float ix = gs_src(gp[0], Wi), iy = gs_src(gp[1], Hi);
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float tx = ix - fx, ty = iy - fy;
  const bool vx0 = x0 >= 0 && x0 < Wi, vx1 = x0 + 1 >= 0 && x0 + 1 < Wi;
  const bool vy0 = y0 >= 0 && y0 < Hi, vy1 = y0 + 1 >= 0 && y0 + 1 < Hi;
  float dix = 0.f, diy = 0.f;
  for (int c = 0; c < C; ++c) {
    const float g = gy[(((size_t)b * C + c) * Ho + oy) * Wo + ox];
    const size_t plane = ((size_t)b * C + c) * Hi * Wi;
    if (gx != nullptr) {
      float* q = gx + plane;
      if (vy0 && vx0) atomicAdd(q + y0 * Wi + x0, g * (1.f - tx) * (1.f - ty));
      if (vy0 && vx1) atomicAdd(q + y0 * Wi + x0 + 1, g * tx * (1.f - ty));
      if (vy1 && vx0) atomicAdd(q + (y0 + 1) * Wi + x0, g * (1.f - tx) * ty);
      if (vy1 && vx1) atomicAdd(q + (y0 + 1) * Wi + x0 + 1, g * tx * ty);
    }
    if (ggrid != nullptr) {
      const float* xp = x + plane;
      const float v00 = vy0 && vx0 ? xp[y0 * Wi + x0] : 0.f, v01 = vy0 && vx1 ? xp[y0 * Wi + x0 + 1] : 0.f;
      const float v10 = vy1 && vx0 ? xp[(y0 + 1) * Wi + x0] : 0.f, v11 = vy1 && vx1 ? xp[(y0 + 1) * Wi + x0 + 1] : 0.f;
      dix += g * ((v01 - v00) * (1.f - ty) + (v11 - v10) * ty);
      diy += g * ((v10 - v00) * (1.f - tx) + (v11 - v01) * tx);
    }
  }
  if (ggrid != nullptr) {
    ggrid[(size_t)idx * 2 + 0] = dix * (0.5f * Wi);
    ggrid[(size_t)idx * 2 + 1] = diy * (0.5f * Hi);
  }
}

// ------------------------------------------------------------------------------------------
// upfirdn2d: zero-insert upsample -> pad/crop -> FIR -> decimate, direct form.
// ------------------------------------------------------------------------------------------
__global__ void upfirdn2d_kernel(const float* __restrict__ x, const float* __restrict__ f, float* __restrict__ y,
                                 int BC, int H, int W, int Ho, int Wo, int fh, int fw, int upx, int upy, int downx,
                                 int downy, int padx0, int pady0, int flip, float gain) {
  extern __shared__ float fs[];
  for (int i = threadIdx.x; i < fh * fw; i += blockDim.x) {
    const int fy = i / fw, fx = i % fw;
    // correlation taps: the op is a convolution unless flip_filter (upfirdn2d.py:194-196)
    fs[i] = (flip ? f[fy * fw + fx] : f[(fh - 1 - fy) * fw + (fw - 1 - fx)]) * gain;
  }
  __syncthreads();
  // grid = (row segments, output rows, planes): no integer division to find (plane, oy, ox) -- the three 64-bit
  // divisions of a flat index cost more than the six taps of an ADA pass
  const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y;
  const long long bc = blockIdx.z;
  if (ox >= Wo) return;
  const long long idx = (bc * Ho + oy) * Wo + ox;
  const float* xp = x + bc * H * W;
  // polyphase walk: only the taps that land on a real sample (every up-th one) are visited, input index + 1 per step; one
  // integer division per axis and OUTPUT instead of a modulo and a division per TAP (the ADA chain's four passes at B = 64:
  // 62.6 -> 36.3 us per call on average)
  const int ty = oy * downy - pady0, tx = ox * downx - padx0;  // zero-inserted coordinate of tap 0
  // first tap on a sample with input index >= 0, and that index (up = 1 and 2 -- every ADA pass -- without a division)
  auto first = [](int t, int up, int& f0, int& i0) {
    if (up == 1) {
      f0 = t < 0 ? -t : 0;
      i0 = t + f0;
    } else if (up == 2) {
      f0 = t >= 0 ? (t & 1) : -t;
      i0 = (t + f0) >> 1;
    } else {
      f0 = t >= 0 ? (up - t % up) % up : -t;
      i0 = (t + f0) / up;
    }
  };
  int fy0, fx0, iy, ix0;
  first(ty, upy, fy0, iy);
  first(tx, upx, fx0, ix0);
  float acc = 0.f;
  // separable passes with at most 12 visited taps (every ADA pass: 12-tap filters, up or down by 2): all loads of the
  // output are issued before the first FMA -- the rolled loop below waits out one memory round trip per tap
  if (fh == 1 && fy0 == 0 && iy < H && (fw - fx0 + upx - 1) / upx <= 12) {
    const float* xr = xp + (size_t)iy * W + ix0;
    float v[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) v[t] = (fx0 + t * upx < fw && ix0 + t < W) ? xr[t] : 0.f;
#pragma unroll
    for (int t = 0; t < 12; ++t) acc = fmaf(v[t], fx0 + t * upx < fw ? fs[fx0 + t * upx] : 0.f, acc);
  } else if (fw == 1 && fx0 == 0 && ix0 < W && (fh - fy0 + upy - 1) / upy <= 12) {
    const float* xc = xp + (size_t)iy * W + ix0;
    float v[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) v[t] = (fy0 + t * upy < fh && iy + t < H) ? xc[(size_t)t * W] : 0.f;
#pragma unroll
    for (int t = 0; t < 12; ++t) acc = fmaf(v[t], fy0 + t * upy < fh ? fs[fy0 + t * upy] : 0.f, acc);
  } else {
    for (int fy = fy0; fy < fh && iy < H; fy += upy, ++iy) {
      const float* xr = xp + (size_t)iy * W;
      const float* fr = fs + fy * fw;
      int ix = ix0;
      for (int fx = fx0; fx < fw && ix < W; fx += upx, ++ix) acc = fmaf(xr[ix], fr[fx], acc);
    }
  }
  y[idx] = acc;
}

// ------------------------------------------------------------------------------------------
// affine_grid + bilinear grid_sample (zeros padding, align_corners=False), fused
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void affine_src(const float* th, int ox, int oy, int Wo, int Ho, int Wi, int Hi, float& ix,
                                           float& iy) {
  const float xn = (2.0f * ox + 1.0f) / Wo - 1.0f;  // affine_grid base grid, align_corners=False
  const float yn = (2.0f * oy + 1.0f) / Ho - 1.0f;
  const float gx = th[0] * xn + th[1] * yn + th[2];
  const float gy = th[3] * xn + th[4] * yn + th[5];
  ix = ((gx + 1.0f) * Wi - 1.0f) * 0.5f;  // grid_sampler unnormalize, align_corners=False
  iy = ((gy + 1.0f) * Hi - 1.0f) * 0.5f;
}

__global__ void affine_grid_sample_fwd_kernel(const float* __restrict__ x, const float* __restrict__ theta,
                                              float* __restrict__ y, int B, int C, int Hi, int Wi, int Ho, int Wo) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = (long long)B * C * Ho * Wo;
  if (idx >= n) return;
  const int ox = idx % Wo, oy = (idx / Wo) % Ho, c = (idx / ((long long)Wo * Ho)) % C;
  const int b = idx / ((long long)Wo * Ho * C);
  float ix, iy;
  affine_src(theta + b * 6, ox, oy, Wo, Ho, Wi, Hi, ix, iy);
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float tx = ix - fx, ty = iy - fy;
  const float* xp = x + ((size_t)b * C + c) * Hi * Wi;
  float v = 0.f;
  if (y0 >= 0 && y0 < Hi) {
    if (x0 >= 0 && x0 < Wi) v += xp[y0 * Wi + x0] * (1.f - tx) * (1.f - ty);
    if (x0 + 1 >= 0 && x0 + 1 < Wi) v += xp[y0 * Wi + x0 + 1] * tx * (1.f - ty);
  }
  if (y0 + 1 >= 0 && y0 + 1 < Hi) {
    if (x0 >= 0 && x0 < Wi) v += xp[(y0 + 1) * Wi + x0] * (1.f - tx) * ty;
    if (x0 + 1 >= 0 && x0 + 1 < Wi) v += xp[(y0 + 1) * Wi + x0 + 1] * tx * ty;
  }
  y[idx] = v;
}

__global__ void affine_grid_sample_bwd_kernel(const float* __restrict__ gy_, const float* __restrict__ theta,
                                              float* __restrict__ gx_, int B, int C, int Hi, int Wi, int Ho, int Wo) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = (long long)B * C * Ho * Wo;
  if (idx >= n) return;
  const int ox = idx % Wo, oy = (idx / Wo) % Ho, c = (idx / ((long long)Wo * Ho)) % C;
  const int b = idx / ((long long)Wo * Ho * C);
  float ix, iy;
  affine_src(theta + b * 6, ox, oy, Wo, Ho, Wi, Hi, ix, iy);
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float tx = ix - fx, ty = iy - fy;
  float* gp = gx_ + ((size_t)b * C + c) * Hi * Wi;
  const float g = gy_[idx];
  if (y0 >= 0 && y0 < Hi) {
    if (x0 >= 0 && x0 < Wi) atomicAdd(gp + y0 * Wi + x0, g * (1.f - tx) * (1.f - ty));
    if (x0 + 1 >= 0 && x0 + 1 < Wi) atomicAdd(gp + y0 * Wi + x0 + 1, g * tx * (1.f - ty));
  }
  if (y0 + 1 >= 0 && y0 + 1 < Hi) {
    if (x0 >= 0 && x0 < Wi) atomicAdd(gp + (y0 + 1) * Wi + x0, g * (1.f - tx) * ty);
    if (x0 + 1 >= 0 && x0 + 1 < Wi) atomicAdd(gp + (y0 + 1) * Wi + x0 + 1, g * tx * ty);
  }
}

// ------------------------------------------------------------------------------------------
// reflect padding and its adjoint
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect_idx(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

template <bool BWD>
__global__ void reflect_pad_kernel(const float* __restrict__ src, float* __restrict__ dst, int BC, int H, int W,
                                   int px0, int py0, int Ho, int Wo) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = (long long)BC * Ho * Wo;
  if (idx >= n) return;
  const int ox = idx % Wo, oy = (idx / Wo) % Ho;
  const long long bc = idx / ((long long)Wo * Ho);
  const int ix = reflect_idx(ox - px0, W), iy = reflect_idx(oy - py0, H);
  if (BWD) atomicAdd(dst + (bc * H + iy) * W + ix, src[idx]);
  else dst[idx] = src[(bc * H + iy) * W + ix];
}

// ------------------------------------------------------------------------------------------
// The ADA geometric augmentation in TWO launches instead of six (augment.py:284-301):
//   K1  reflect pad + x2 up-FIR (12 taps, zero insertion, gain 4)       x [BC][H][W] -> canvas [BC][2 Hp][2 Wp]
//   K2  affine bilinear resample onto 2 (H + 6) x 2 (W + 6) + /2 down-FIR   canvas, theta [B][2][3] -> y [BC][H][W]
// Same arithmetic as the separate passes (upfirdn2d(up = 2, pad 6 / 5, gain 2 per axis), grid_sample(zeros, align_corners =
// False), upfirdn2d(down = 2, pad -1 / -1, flipped filter)); only the order of the fp32 sums inside a 6 x 6 / 12 x 12
// window differs.  The padded image, the row-upsampled image, the resampled grid and the row-filtered grid never reach
// memory.  Linear in x: the adjoint is the existing chain of adjoint kernels (oi_amd.autograd_disc).
// ------------------------------------------------------------------------------------------
constexpr int ADA_TAPS = 12, ADA_PAD = ADA_TAPS / 4 * 2;  // Hz_pad * 2 = 6: the grid is 2 (H + 6) x 2 (W + 6)

// K1 as a separable pass over an LDS tile: a workgroup produces ADA_UW x ADA_UH canvas pixels from the (ADA_UH / 2 + 6) x
// (ADA_UW / 2 + 6) padded-image pixels its taps reach (reflect indices resolved once per staged pixel), rows first, then
// columns -- 11.3 multiply-adds per canvas pixel instead of the 36 (+ 12 index computations) of one thread per pixel
// gathering its 6 x 6 window from memory (66 -> 25 us at B = 64).  Same association of the sums as before: bit-identical.
constexpr int ADA_UW = 64, ADA_UH = 16, ADA_SW = ADA_UW / 2 + 6, ADA_SH = ADA_UH / 2 + 6;

__global__ void __launch_bounds__(256)
ada_pad_up2_kernel(const float* __restrict__ x, const float* __restrict__ f, float* __restrict__ canvas, int BC, int H,
                   int W, int mx0, int my0, int Hp, int Wp) {
  __shared__ float fr[ADA_TAPS];  // correlation taps of the convolution: reversed, sqrt(gain) = 2 per axis
  __shared__ float src[ADA_SH][ADA_SW + 1];
  __shared__ float tmp[ADA_SH][ADA_UW + 1];
  const int tid = threadIdx.x;
  if (tid < ADA_TAPS) fr[tid] = 2.0f * f[ADA_TAPS - 1 - tid];
  const int Wc = 2 * Wp, Hc = 2 * Hp;
  const int u0 = blockIdx.y * ADA_UH, v0 = blockIdx.x * ADA_UW;  // even: out[n] = sum_k fr[k] xup[n + k - 6], xup[2 m] = P[m]
  const int pu0 = (u0 - 6) / 2, pv0 = (v0 - 6) / 2;             // P index of the first tap of the tile's first pixel (exact: even)
  const float* xp = x + (size_t)blockIdx.z * H * W;
  for (int i = tid; i < ADA_SH * ADA_SW; i += 256) {
    const int a = i / ADA_SW, b = i % ADA_SW;
    const int r = pu0 + a, c = pv0 + b;
    // zero outside the padded image (the FIR's own padding); inside it, the reflect-padded source
    src[a][b] = (r >= 0 && r < Hp && c >= 0 && c < Wp) ? xp[(size_t)reflect_idx(r - my0, H) * W + reflect_idx(c - mx0, W)] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < ADA_SH * ADA_UW; i += 256) {  // rows: the six taps k = kv, kv + 2, .. with (v + k) even
    const int a = i / ADA_UW, vl = i % ADA_UW;
    const int kv = vl & 1, pl = (vl + kv) / 2;        // (v + kv - 6) / 2 - pv0
    float s_ = 0.f;
#pragma unroll
    for (int b = 0; b < 6; ++b) s_ = fmaf(fr[kv + 2 * b], src[a][pl + b], s_);
    tmp[a][vl] = s_;
  }
  __syncthreads();
  for (int i = tid; i < ADA_UH * ADA_UW; i += 256) {
    const int ul = i / ADA_UW, vl = i % ADA_UW;
    const int u = u0 + ul, v = v0 + vl;
    if (u >= Hc || v >= Wc) continue;
    const int ku = ul & 1, pl = (ul + ku) / 2;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 6; ++a) acc = fmaf(fr[ku + 2 * a], tmp[pl + a][vl], acc);
    canvas[((size_t)blockIdx.z * Hc + u) * Wc + v] = acc;
  }
}

constexpr int ADA_T = 16;                          // output tile edge
constexpr int ADA_G = 2 * ADA_T + ADA_TAPS - 2;    // 42: grid points a tile's down-FIR windows cover per axis

__global__ void __launch_bounds__(256)
ada_resample_down2_kernel(const float* __restrict__ canvas, const float* __restrict__ theta, const float* __restrict__ f,
                          float* __restrict__ y, int C, int H, int W, int Hc, int Wc) {
  __shared__ float g[ADA_G][ADA_G + 1];     // resampled grid tile
  __shared__ float gy[ADA_T][ADA_G + 1];    // after the vertical pass
  __shared__ float fs[ADA_TAPS];
  const int tid = threadIdx.x;
  if (tid < ADA_TAPS) fs[tid] = f[tid];     // flip_filter = True: the correlation taps are the filter itself
  const int bc = blockIdx.z, b = bc / C;
  const int ox0 = blockIdx.x * ADA_T, oy0 = blockIdx.y * ADA_T;
  const int Ho = 2 * (H + ADA_PAD), Wo = 2 * (W + ADA_PAD);
  const float* cp = canvas + (size_t)bc * Hc * Wc;
  // y[oy][ox] = sum_{k, l} f[k] f[l] G[2 oy + k + 1][2 ox + l + 1]
  for (int i = tid; i < ADA_G * ADA_G; i += 256) {
    const int r = i / ADA_G, c = i % ADA_G;
    const int gyi = 2 * oy0 + 1 + r, gxi = 2 * ox0 + 1 + c;
    float v = 0.f;
    if (gyi < Ho && gxi < Wo) {
      float ix, iy;
      affine_src(theta + b * 6, gxi, gyi, Wo, Ho, Wc, Hc, ix, iy);
      const float fx = floorf(ix), fy = floorf(iy);
      const int x0 = (int)fx, y0 = (int)fy;
      const float tx = ix - fx, ty = iy - fy;
      if (y0 >= 0 && y0 < Hc) {
        if (x0 >= 0 && x0 < Wc) v += cp[(size_t)y0 * Wc + x0] * (1.f - tx) * (1.f - ty);
        if (x0 + 1 >= 0 && x0 + 1 < Wc) v += cp[(size_t)y0 * Wc + x0 + 1] * tx * (1.f - ty);
      }
      if (y0 + 1 >= 0 && y0 + 1 < Hc) {
        if (x0 >= 0 && x0 < Wc) v += cp[(size_t)(y0 + 1) * Wc + x0] * (1.f - tx) * ty;
        if (x0 + 1 >= 0 && x0 + 1 < Wc) v += cp[(size_t)(y0 + 1) * Wc + x0 + 1] * tx * ty;
      }
    }
    g[r][c] = v;
  }
  __syncthreads();
  for (int i = tid; i < ADA_T * ADA_G; i += 256) {  // vertical pass: row 2 t + k of the tile
    const int t = i / ADA_G, c = i % ADA_G;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < ADA_TAPS; ++k) s = fmaf(fs[k], g[2 * t + k][c], s);
    gy[t][c] = s;
  }
  __syncthreads();
  {
    const int t = tid / ADA_T, c = tid % ADA_T;
    float s = 0.f;
#pragma unroll
    for (int l = 0; l < ADA_TAPS; ++l) s = fmaf(fs[l], gy[t][2 * c + l], s);
    const int oy = oy0 + t, ox = ox0 + c;
    if (oy < H && ox < W) y[((size_t)bc * H + oy) * W + ox] = s;
  }
}


// ------------------------------------------------------------------------------------------
// The same augmentation for AXIS-ALIGNED sampling matrices (theta[1] = theta[3] = 0: integer / fractional translation, isotropic
// or anisotropic scale, flips -- everything but the rotations), ONE launch and no canvas.  Each of the four stages is linear and,
// without a rotation, separable (reflect pad, the two FIRs, and bilinear sampling with zero padding on an axis-aligned grid), so
// the whole map is  y_c = A_y x_c A_x^T  with two H x H matrices per image: A = D S(theta) U P, D = /2 down-FIR (12 taps per
// row), S = the 1-D bilinear sampling (2 per row), U = x2 up-FIR (6 per row), P = reflect pad (1 per row): 144 products per
// row.  At batch 64 the two-launch form writes and re-reads an 86 MB canvas for 6 MB of images (23 + 26 us); this one reads the
// images once.  A workgroup = (image, 16 output rows): builds A_x^T (64 x 64) and its 16 rows of A_y in LDS -- 1920 tasks (row,
// down-FIR tap, bilinear corner) of six products each, formed and added as FIXED POINT with 64-bit LDS atomics (integer sums: the
// order the threads arrive in does not matter, the result is bit-reproducible) -- then two small fp32 products on the matrix
// cores.  Same products as the two-launch form in a different association: equal to it within rounding, not bit-identical.
// ------------------------------------------------------------------------------------------
#ifndef OI_AS_ABL   // timing ablations (results garbage): 1 = no matrix build, 2 = no first product, 4 = no second product, 8 = no zero fill
#define OI_AS_ABL 0
#endif
#ifndef OI_AS_THREADS
#define OI_AS_THREADS 1024
#endif
constexpr int AS_RB = 16, AS_THREADS = OI_AS_THREADS, AS_NW = AS_THREADS / 64, AS_MAX_C = 3;
// Per image edge N (64: BASELINE's discriminators; 128: the shipped ones).  LDS row strides (floats) of the MFMA operands are
// chosen so that the 64 lanes of a fragment read 64 banks: B operands [k][n] N + 16, A operands [m][k] N + 4.
//   N = 64:  the matrices are accumulated as 2^-56 fixed point in 64 bits and converted into their own float arrays; the image
//            is staged in LDS (61 KB);
//   N = 128: 64-bit accumulators of A_x alone would be 128 KB: 32-bit, 2^-27 (each term rounded to nearest: 144 terms stay
//            near fp32's own 2^-24), converted IN PLACE; the image is not staged -- the first product's B fragments come
//            straight from memory (the eight row blocks of an image read the same 196 KB: L2).
template <int N>
struct AsCfg {
  static constexpr bool WIDE = N == 64;                 // 64-bit fixed point, separate float arrays, image in LDS
  static constexpr int XS = N + 16, ZS = N + 4;
  static constexpr int FXS = WIDE ? N + 1 : XS;         // row stride of the fixed-point A_x (WIDE adjoint: rows on the lanes -> other banks)
  static constexpr int FB = WIDE ? 8 : 4;               // bytes per accumulator
  static constexpr int O_FX = 0, O_FY = O_FX + N * FXS * FB, O_FEND = O_FY + N * AS_RB * FB;
  static constexpr int O_AX = WIDE ? O_FEND : O_FX, O_AY = WIDE ? O_AX + N * XS * 4 : O_FEND;
  static constexpr int O_X = O_AY + AS_RB * ZS * 4, O_Z = O_X + (WIDE ? AS_MAX_C * N * XS * 4 : 0);
  static constexpr int O_F = O_Z + AS_MAX_C * AS_RB * ZS * 4, LDS_BYTES = O_F + 2 * ADA_TAPS * 4;
  static_assert(LDS_BYTES <= 160 * 1024 && O_FEND % 16 == 0, "ada_sep_kernel: LDS");
};

// 1-D sampling coordinate on the canvas axis: affine_src with the other axis' coefficient zero, in ITS operations (the product
// and the sum round separately there: the zero term sits between them) -- the same bits, so the same bilinear weights
__device__ __forceinline__ float sep_coord(float ts, float tt, int g, int No, int Nc) {
  const float n = (2.0f * g + 1.0f) / No - 1.0f;
  float pr = ts * n;
  asm volatile("" : "+v"(pr));   // (hipcc fuses __fadd_rn(__fmul_rn()) as well: the product must round on its own)
  const float gg = pr + tt;
  return ((gg + 1.0f) * Nc - 1.0f) * 0.5f;
}
// The build in integers: taps and bilinear weights as 2^-28 fixed point, a term = ((f_k w + 2^27) >> 28) x fr_q at 2^-56 in 64 bits
// (each factor within 2^-29 of its fp32 value, rounded to nearest: 144 terms stay below fp32's own 2^-24).  Integer sums do not
// depend on the order the threads add in, and an entry of A is rounded to fp32 ONCE, after its terms have cancelled.
__device__ __forceinline__ int sep_fix28(float v) { return (int)rintf(v * 268435456.0f); }
__device__ __forceinline__ float sep_from_fix(long long v) { return (float)((double)v * (1.0 / 72057594037927936.0)); }
__device__ __forceinline__ float sep_from_fix(int v) { return (float)v * (1.0f / 134217728.0f); }   // (|v| < 2^31: two roundings of
                                                                                                    //  an entry, 2^-27 then fp32)

// the sampling matrices of up to AS_BYVAL images in the kernel arguments (a caller that drew them on the host: no upload)
constexpr int AS_BYVAL = 64;
struct AsTheta {
  float t[AS_BYVAL][6];
};

// ADJ: the adjoint map gx_c = A_y^T gy_c A_x (the backward of the forward map, and -- the map being linear -- what the R1
// penalty's second pass differentiates): the same kernel with the matrices stored the other way round; a workgroup's 16 rows of
// A_y^T are 16 COLUMNS of A_y, so all N rows of A_y are walked and the terms that land in the block are kept.
template <int N, int C, bool BYVAL, bool ADJ>
__global__ void __launch_bounds__(AS_THREADS)
ada_sep_kernel(const float* __restrict__ x, const float* __restrict__ theta, const AsTheta thv, const float* __restrict__ f,
               float* __restrict__ y, int mx0, int my0, int Wp, int Hp) {
  using K = AsCfg<N>;
  using FX = std::conditional_t<K::WIDE, unsigned long long, unsigned>;
  using FXS_ = std::conditional_t<K::WIDE, long long, int>;
  constexpr int XS = K::XS, ZS = K::ZS, LOGN = N == 64 ? 6 : 7;
  extern __shared__ __attribute__((aligned(16))) char as_lds[];
  FX* fxx = reinterpret_cast<FX*>(as_lds + K::O_FX);   // forward [source x][output x], adjoint [output x][source x]
  FX* fxy = reinterpret_cast<FX*>(as_lds + K::O_FY);   // [k of the first product][row of the block]
  float* axt = reinterpret_cast<float*>(as_lds + K::O_AX);
  float* ayb = reinterpret_cast<float*>(as_lds + K::O_AY);
  float* xs = reinterpret_cast<float*>(as_lds + K::O_X);
  float* zb = reinterpret_cast<float*>(as_lds + K::O_Z);
  int* fs = reinterpret_cast<int*>(as_lds + K::O_F);   // down-FIR taps (flip_filter: the filter itself), 2^-28 fixed point
  int* fr = fs + ADA_TAPS;                             // up-FIR taps: reversed, sqrt(gain) = 2 per axis
  const int tid = threadIdx.x, b = blockIdx.y, rb = blockIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const float* xb = x + (size_t)b * C * N * N;
  // WIDE: the image is requested first
  constexpr int NX = K::WIDE ? C * N * N / (4 * AS_THREADS) : 1;   // b128 loads per thread
  f32x4 xv[NX];
  if constexpr (K::WIDE) {
#pragma unroll
    for (int i = 0; i < NX; ++i) xv[i] = *reinterpret_cast<const f32x4*>(xb + 4 * (tid + i * AS_THREADS));
  }
  const float tsx = BYVAL ? thv.t[b][0] : theta[b * 6 + 0], ttx = BYVAL ? thv.t[b][2] : theta[b * 6 + 2];
  const float tsy = BYVAL ? thv.t[b][4] : theta[b * 6 + 4], tty = BYVAL ? thv.t[b][5] : theta[b * 6 + 5];
  if (tid < ADA_TAPS) {
    fs[tid] = sep_fix28(f[tid]);
    fr[tid] = sep_fix28(2.0f * f[ADA_TAPS - 1 - tid]);
  }
  for (int i = tid; i < ((OI_AS_ABL & 8) ? 0 : (K::O_FEND - K::O_FX) / 16); i += AS_THREADS)
    reinterpret_cast<f32x4*>(as_lds + K::O_FX)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  // the matrices: one task per (row, down-FIR tap k, bilinear corner c) = 6 products through the up-FIR and the reflect pad
  constexpr int NROW = N + (ADJ ? N : AS_RB), NTASK = NROW * ADA_TAPS * 2;
  for (int t_ = tid; t_ < ((OI_AS_ABL & 1) ? 0 : NTASK); t_ += AS_THREADS) {
    // (the row on the lane: the adds of a wave then go to different addresses -- with the taps on the lane up to 24 lanes
    // met in one)
    const int kc = t_ / NROW, row = t_ - kc * NROW, k = kc >> 1, c = kc & 1;
    const bool isx = row < N;
    const int a = isx ? row : (ADJ ? row - N : rb * AS_RB + (row - N));
    const int m0 = isx ? mx0 : my0, Np = isx ? Wp : Hp, Nc = 2 * Np;
    const float ic = sep_coord(isx ? tsx : tsy, isx ? ttx : tty, 2 * a + k + 1, 2 * (N + ADA_PAD), Nc);
    const float fl = floorf(ic), tt_ = ic - fl;
    const int u = (int)fl + c;
    if (u < 0 || u >= Nc) continue;
    const int wc = (int)(((long long)fs[k] * sep_fix28(c ? tt_ : 1.f - tt_) + (1ll << 27)) >> 28);
    const int ku = u & 1, base = (u + ku) / 2 - 3;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int r = base + q;
      if (r < 0 || r >= Np) continue;
      const int sidx = reflect_idx(r - m0, N);
      FX* dst;
      if constexpr (ADJ) {
        const int i = sidx - rb * AS_RB;
        if (!isx && (i < 0 || i >= AS_RB)) continue;
        dst = isx ? fxx + a * K::FXS + sidx : fxy + a * AS_RB + i;
      } else {
        dst = isx ? fxx + sidx * (K::WIDE ? N : XS) + a : fxy + sidx * AS_RB + (row - N);
      }
      const long long term = (long long)wc * fr[ku + 2 * q];   // 2^-56
      if constexpr (K::WIDE) atomicAdd(dst, (FX)term);
      else atomicAdd(dst, (FX)(int)((term + (1ll << 28)) >> 29));   // 2^-27, to nearest
    }
  }
  __syncthreads();
  // the second product's B operand [k][n]: forward k = source, adjoint k = output
  if constexpr (K::WIDE) {
    for (int i = tid; i < N * N; i += AS_THREADS)
      axt[(i >> LOGN) * XS + (i & (N - 1))] = sep_from_fix((FXS_)fxx[(i >> LOGN) * (ADJ ? K::FXS : N) + (i & (N - 1))]);
  } else {
    for (int i = tid; i < N * N; i += AS_THREADS) {   // in place: the thread that reads a word writes it
      const int o = (i >> LOGN) * XS + (i & (N - 1));
      axt[o] = sep_from_fix((FXS_)fxx[o]);
    }
  }
  for (int i = tid; i < AS_RB * N; i += AS_THREADS) ayb[(i & 15) * ZS + (i >> 4)] = sep_from_fix((FXS_)fxy[i]);
  if constexpr (K::WIDE) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = 4 * (tid + i * AS_THREADS);   // element of [C][N][N]: row e / N, column e % N
      *reinterpret_cast<f32x4*>(xs + (e >> LOGN) * XS + (e & (N - 1))) = xv[i];
    }
  }
  __syncthreads();
  // two small products on the matrix cores (exact fp32 operands, v_mfma_f32_16x16x4_f32: A lane = (row l % 16, k l / 16),
  // B lane = (k l / 16, column l % 16), D lane = (rows 4 (l / 16) + r, column l % 16)); a wave takes tiles wave, wave + AS_NW, ..
  const int l16 = lane & 15, kq = lane >> 4;
  constexpr int KS = N / 4, NT = N / 16;
  {  // Z_c = A_y[block] X_c: per channel 16 x N, K = N source rows
    float ay[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) ay[kk] = ayb[l16 * ZS + 4 * kk + kq];
    for (int p = wave; p < ((OI_AS_ABL & 2) ? 0 : C * NT); p += AS_NW) {
      const int c = p / NT, nt = p - c * NT;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if constexpr (K::WIDE) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ay[kk], xs[(c * N + 4 * kk + kq) * XS + 16 * nt + l16], acc, 0, 0, 0);
      } else {
        float bv[KS];   // every load of the tile in flight before the first product
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) bv[kk] = xb[(size_t)(c * N + 4 * kk + kq) * N + 16 * nt + l16];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ay[kk], bv[kk], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) zb[(c * AS_RB + 4 * kq + r) * ZS + 16 * nt + l16] = acc[r];
    }
  }
  __syncthreads();
  for (int p = wave; p < ((OI_AS_ABL & 4) ? 0 : C * NT); p += AS_NW) {   // Y = Z A_x^T: (C 16) x N, K = N
    const int c = p / NT, nt = p - c * NT;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zb[(c * AS_RB + l16) * ZS + 4 * kk + kq], axt[(4 * kk + kq) * XS + 16 * nt + l16],
                                                 acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      y[(((size_t)b * C + c) * N + rb * AS_RB + 4 * kq + r) * N + 16 * nt + l16] = acc[r];
  }
}

template <int N>
int ada_sep_launch_n(const char* who, bool adj, const float* x, const float* theta, const float* theta_host, const float* f,
                            float* y, int B, int C, int mx0, int my0, int Wp, int Hp, hipStream_t st) {
  constexpr int lds = AsCfg<N>::LDS_BYTES;
  auto launch = [&](auto k, const float* xb, const float* th, const AsTheta& tv, float* yb, int nb) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k, dim3(N / AS_RB, nb), dim3(AS_THREADS), lds, st, xb, th, tv, f, yb, mx0, my0, Wp, Hp);
  };
  auto pick = [&](auto byval, const float* xb, const float* th, const AsTheta& tv, float* yb, int nb) {
    constexpr bool BV = decltype(byval)::value;
    if (adj) {
      if (C == 1) launch(ada_sep_kernel<N, 1, BV, true>, xb, th, tv, yb, nb);
      else if (C == 2) launch(ada_sep_kernel<N, 2, BV, true>, xb, th, tv, yb, nb);
      else launch(ada_sep_kernel<N, 3, BV, true>, xb, th, tv, yb, nb);
    } else {
      if (C == 1) launch(ada_sep_kernel<N, 1, BV, false>, xb, th, tv, yb, nb);
      else if (C == 2) launch(ada_sep_kernel<N, 2, BV, false>, xb, th, tv, yb, nb);
      else launch(ada_sep_kernel<N, 3, BV, false>, xb, th, tv, yb, nb);
    }
  };
  if (theta != nullptr) {
    static const AsTheta none{};
    pick(std::false_type{}, x, theta, none, y, B);
    return oi::check_launch(who);
  }
  for (int b0 = 0; b0 < B; b0 += AS_BYVAL) {   // the matrices by value: AS_BYVAL images per launch
    const int nb = std::min(AS_BYVAL, B - b0);
    AsTheta tv;
    for (int i = 0; i < nb * 6; ++i) tv.t[i / 6][i % 6] = theta_host[(size_t)b0 * 6 + i];
    pick(std::true_type{}, x + (size_t)b0 * C * N * N, nullptr, tv, y + (size_t)b0 * C * N * N, nb);
    int rc = oi::check_launch(who);
    if (rc != OI_OK) return rc;
  }
  return OI_OK;
}

int ada_sep_launch(const char* who, bool adj, const float* x, const float* theta, const float* theta_host, const float* f,
                          float* y, int B, int C, int H, int W, int mx0, int mx1, int my0, int my1, oi_stream_t stream) {
  OI_REQUIRE(x && f && y, "%s: null pointer", who);
  OI_REQUIRE((theta != nullptr) != (theta_host != nullptr), "%s: the sampling matrices on the device OR on the host", who);
  OI_REQUIRE(B > 0 && B <= 65535, "%s: batch %d", who, B);
  OI_REQUIRE(oi_ada_geom_sep_supported(C, H, W), "%s: %d x %d x %d images (covered: 1..%d channels of 64 x 64 or 128 x 128)", who, C, H,
             W, AS_MAX_C);
  OI_REQUIRE(mx0 >= 0 && mx1 >= 0 && my0 >= 0 && my1 >= 0 && mx0 < W && mx1 < W && my0 < H && my1 < H,
             "%s: reflect margins must be in [0, size)", who);
  const int Hp = H + my0 + my1, Wp = W + mx0 + mx1;
  hipStream_t st = oi::as_stream(stream);
  return H == 64 ? ada_sep_launch_n<64>(who, adj, x, theta, theta_host, f, y, B, C, mx0, my0, Wp, Hp, st)
                 : ada_sep_launch_n<128>(who, adj, x, theta, theta_host, f, y, B, C, mx0, my0, Wp, Hp, st);
}

}  // namespace

extern "C" {

int oi_ada_pad_up2(const float* x, const float* f, float* canvas, int B, int C, int H, int W, int mx0, int mx1, int my0, int my1,
                   oi_stream_t stream) {
  OI_REQUIRE(x && f && canvas, "oi_ada_pad_up2: null pointer");
  OI_REQUIRE(B > 0 && C > 0 && H > 1 && W > 1, "oi_ada_pad_up2: bad shape");
  OI_REQUIRE(mx0 >= 0 && mx1 >= 0 && my0 >= 0 && my1 >= 0 && mx0 < W && mx1 < W && my0 < H && my1 < H,
             "oi_ada_pad_up2: reflect margins must be in [0, size)");
  const long long BC = (long long)B * C;
  const int Hp = H + my0 + my1, Wp = W + mx0 + mx1;
  OI_REQUIRE(BC <= 65535 && 2 * Hp <= 65535, "oi_ada_pad_up2: launch grid");
  hipLaunchKernelGGL(ada_pad_up2_kernel, dim3(oi::cdiv(2 * Wp, ADA_UW), oi::cdiv(2 * Hp, ADA_UH), (unsigned)BC), dim3(256), 0,
                     oi::as_stream(stream), x, f, canvas, (int)BC, H, W, mx0, my0, Hp, Wp);
  return oi::check_launch("oi_ada_pad_up2");
}

int oi_ada_geom_fwd(const float* x, const float* theta, const float* f, float* y, float* canvas, int B, int C, int H, int W,
                    int mx0, int mx1, int my0, int my1, oi_stream_t stream) {
  OI_REQUIRE(x && theta && f && y && canvas, "oi_ada_geom_fwd: null pointer");
  OI_REQUIRE(B > 0 && C > 0 && H > 1 && W > 1, "oi_ada_geom_fwd: bad shape");
  OI_REQUIRE(mx0 >= 0 && mx1 >= 0 && my0 >= 0 && my1 >= 0 && mx0 < W && mx1 < W && my0 < H && my1 < H,
             "oi_ada_geom_fwd: reflect margins must be in [0, size)");
  const long long BC = (long long)B * C;
  OI_REQUIRE(BC <= 65535, "oi_ada_geom_fwd: %lld planes exceed the launch grid", BC);
  const int Hp = H + my0 + my1, Wp = W + mx0 + mx1;
  OI_REQUIRE(2 * Hp <= 65535, "oi_ada_geom_fwd: canvas of %d rows", 2 * Hp);
  hipStream_t st = oi::as_stream(stream);
  hipLaunchKernelGGL(ada_pad_up2_kernel, dim3(oi::cdiv(2 * Wp, ADA_UW), oi::cdiv(2 * Hp, ADA_UH), (unsigned)BC), dim3(256), 0, st,
                     x, f, canvas, (int)BC, H, W, mx0, my0, Hp, Wp);
  int rc = oi::check_launch("oi_ada_geom_fwd(pad + upsample)");
  if (rc != OI_OK) return rc;
  hipLaunchKernelGGL(ada_resample_down2_kernel, dim3(oi::cdiv(W, ADA_T), oi::cdiv(H, ADA_T), (unsigned)BC), dim3(256), 0, st,
                     canvas, theta, f, y, C, H, W, 2 * Hp, 2 * Wp);
  return oi::check_launch("oi_ada_geom_fwd(resample + downsample)");
}

int oi_ada_geom_sep_supported(int C, int H, int W) { return (C >= 1 && C <= AS_MAX_C && H == W && (H == 64 || H == 128)) ? 1 : 0; }

int oi_ada_geom_sep_fwd(const float* x, const float* theta, const float* theta_host, const float* f, float* y, int B, int C, int H,
                        int W, int mx0, int mx1, int my0, int my1, oi_stream_t stream) {
  return ada_sep_launch("oi_ada_geom_sep_fwd", false, x, theta, theta_host, f, y, B, C, H, W, mx0, mx1, my0, my1, stream);
}

int oi_ada_geom_sep_adj(const float* gy, const float* theta, const float* theta_host, const float* f, float* gx, int B, int C, int H,
                        int W, int mx0, int mx1, int my0, int my1, oi_stream_t stream) {
  return ada_sep_launch("oi_ada_geom_sep_adj", true, gy, theta, theta_host, f, gx, B, C, H, W, mx0, mx1, my0, my1, stream);
}

int oi_conv4x4_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W, int Cout,
                   int stride, int pad, float slope, oi_stream_t stream) {
  return oi_conv4x4_fwd_into(x, w, bias, y, B, Cin, H, W, Cout, stride, pad, slope, 1.0f, 0, stream);
}

int oi_conv4x4_fwd_into(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W,
                        int Cout, int stride, int pad, float slope, float x_slope, int flags,
                        oi_stream_t stream) {
  return oi_conv4x4_fwd_arena(x, w, bias, y, B, Cin, H, W, Cout, stride, pad, slope, x_slope, flags, 0, stream);
}

int oi_conv4x4_fwd_arena(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W,
                         int Cout, int stride, int pad, float slope, float x_slope, int flags, long long zero_tail_floats,
                         oi_stream_t stream) {
  OI_REQUIRE(x && w && y, "oi_conv4x4_fwd: null pointer");
  OI_REQUIRE(zero_tail_floats >= 0, "oi_conv4x4_fwd_arena: zero_tail_floats %lld", zero_tail_floats);
  OI_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && stride > 0 && pad >= 0, "oi_conv4x4_fwd: bad shape");
  const int Ho = (H + 2 * pad - 4) / stride + 1, Wo = (W + 2 * pad - 4) / stride + 1;
  OI_REQUIRE(Ho > 0 && Wo > 0, "oi_conv4x4_fwd: input %dx%d too small", H, W);
  const long long M = (long long)B * Ho * Wo;
  hipStream_t st = oi::as_stream(stream);
  const bool y_is_zero = (flags & OI_CONV_Y_IS_ZERO) != 0;
  // OI_CONV_ANY_SCALE: an operand may be a gradient (the R1 double backward feeds d loss / d image and d loss / d w through
  // this entry): the tiled kernel's UNSCALED fp16 limbs lose precision below |v| ~ 3e-4 and flush below ~3e-8, so such
  // calls stay on the fp32 matrix cores
  const bool any_scale = (flags & OI_CONV_ANY_SCALE) != 0;
  // large batches: LDS-tiled F16X3 implicit GEMM, K never split (OI_CONV_TILED=0 keeps the per-wave fp32-MFMA path)
  static const bool tiled_on = [] { const char* e = getenv("OI_CONV_TILED"); return e == nullptr || e[0] != '0'; }();
  const int K = Cin * 16;
  const bool small = (long long)oi::cdiv(M, 128) * (Cout / 64) < 256;  // fewer workgroups than CUs: halve the pixel tile
  // still short of the CUs and a long K: two K halves.  Exactly two addends into a zeroed output commute, so the result
  // stays reproducible; the activation then needs its own pass (none when the caller defers it: slope 1, no bias)
  const int splits = (small && (long long)oi::cdiv(M, 64) * (Cout / 64) < 192 && K >= 1024) ? 2 : 1;
  // worth it from ~half a chip of workgroups; below that the per-wave split-K kernel spreads the K loop over more CUs
  const bool enough = (long long)oi::cdiv(M, small ? 64 : 128) * (Cout / 64) * splits >= 128;
  if (tiled_on && !any_scale && M >= 512 && enough && Cout % 64 == 0 && K % 64 == 0) {
    const int k_per_split = splits == 2 ? ((K / 2 + 63) / 64) * 64 : K;
    const long long total = (long long)B * Cout * Ho * Wo;
    if (splits == 2 && !y_is_zero) {
      hipError_t e = oi::zero_output_async(y, total, st);
      if (e != hipSuccess) return oi::fail(OI_ERR_LAUNCH, "oi_conv4x4_fwd: zero fill: %s", hipGetErrorString(e));
    }
    if (zero_tail_floats > 0) {
      hipError_t e = oi::zero_async(y + (total + 3) / 4 * 4, zero_tail_floats, st);
      if (e != hipSuccess) return oi::fail(OI_ERR_LAUNCH, "oi_conv4x4_fwd: zero fill: %s", hipGetErrorString(e));
    }
    dim3 grid(oi::cdiv(M, small ? 64 : 128), Cout / 64, splits), block(256);
    if (small)
      hipLaunchKernelGGL(conv4x4_tiled_f16x3_kernel<64>, grid, block, 0, st, x, w, bias, y, B, Cin, H, W, Cout, Ho, Wo, stride,
                         pad, slope, x_slope, k_per_split);
    else
      hipLaunchKernelGGL(conv4x4_tiled_f16x3_kernel<128>, grid, block, 0, st, x, w, bias, y, B, Cin, H, W, Cout, Ho, Wo, stride,
                         pad, slope, x_slope, k_per_split);
    int rc = oi::check_launch("oi_conv4x4_fwd(tiled)");
    if (rc == OI_OK && splits == 2 && (slope != 1.0f || bias != nullptr)) {
      hipLaunchKernelGGL(bias_lrelu_kernel, dim3(oi::cdiv(total, 256)), dim3(256), 0, st, y, bias, total, Cout, Ho * Wo, slope);
      rc = oi::check_launch("oi_conv4x4_fwd(bias_lrelu)");
    }
    return rc;
  }
  const int m_tiles = oi::cdiv(M, 32), n_tiles = oi::cdiv(Cout, 32);
  const int krows = Cin * 4;
  // split K until ~2 waves per SIMD are in flight, at least 16 kernel rows (64 taps) per split
  int k_splits = 1;
  const long long tiles = (long long)m_tiles * n_tiles;
  while (tiles * k_splits < 2048 && krows / (k_splits * 2) >= 16) k_splits *= 2;
  int rows_per_split = oi::cdiv(krows, k_splits);
  rows_per_split += rows_per_split & 1;  // whole lane-half pairs
  k_splits = oi::cdiv(krows, rows_per_split);
  const long long total = (long long)B * Cout * Ho * Wo;
  if (k_splits > 1 && !y_is_zero) {
    hipError_t e = oi::zero_output_async(y, total, st);
    if (e != hipSuccess) return oi::fail(OI_ERR_LAUNCH, "oi_conv4x4_fwd: memset: %s", hipGetErrorString(e));
  }
  const long long items = tiles * k_splits;
  // the tail of the arena (behind this layer's own output, rounded up to 4 floats as the caller lays it out) is cleared by
  // this launch when the layer itself does not accumulate (no split-K); otherwise by a fill of its own
  float* tail = y + (total + 3) / 4 * 4;
  if (zero_tail_floats > 0 && k_splits > 1) {
    hipError_t e = oi::zero_async(tail, zero_tail_floats, st);
    if (e != hipSuccess) return oi::fail(OI_ERR_LAUNCH, "oi_conv4x4_fwd: memset: %s", hipGetErrorString(e));
    zero_tail_floats = 0;
  }
  hipLaunchKernelGGL(conv4x4_fwd_kernel, dim3(oi::cdiv(items, 4)), dim3(256), 0, st, x, w, bias, y, B, Cin, H, W,
                     Cout, Ho, Wo, stride, pad, slope, m_tiles, n_tiles, k_splits, rows_per_split, x_slope, tail,
                     zero_tail_floats);
  int rc = oi::check_launch("oi_conv4x4_fwd");
  if (rc != OI_OK) return rc;
  if (k_splits > 1 && (slope != 1.0f || bias != nullptr)) {
    hipLaunchKernelGGL(bias_lrelu_kernel, dim3(oi::cdiv(total, 256)), dim3(256), 0, st, y, bias, total, Cout, Ho * Wo,
                       slope);
    rc = oi::check_launch("oi_conv4x4_fwd(bias_lrelu)");
  }
  return rc;
}

int oi_upfirdn2d(const float* x, const float* f, float* y, int BC, int H, int W, int fh, int fw, int upx, int upy,
                 int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip, float gain,
                 oi_stream_t stream) {
  OI_REQUIRE(x && f && y, "oi_upfirdn2d: null pointer");
  OI_REQUIRE(upx >= 1 && upy >= 1 && downx >= 1 && downy >= 1, "oi_upfirdn2d: up/down must be >= 1");
  OI_REQUIRE(BC > 0 && H > 0 && W > 0 && fh > 0 && fw > 0, "oi_upfirdn2d: bad shape");
  // upfirdn2d.cpp:38-40
  const int Wo = (W * upx + padx0 + padx1 - fw + downx) / downx;
  const int Ho = (H * upy + pady0 + pady1 - fh + downy) / downy;
  OI_REQUIRE(Wo >= 1 && Ho >= 1, "oi_upfirdn2d: output size %dx%d", Ho, Wo);
  OI_REQUIRE(Ho <= 65535, "oi_upfirdn2d: %d output rows exceed the launch grid", Ho);
  int bx = 64;  // row segment width with the least padding (ties: the wider one)
  for (int c = 128; c <= 256; c *= 2)
    if (oi::cdiv(Wo, c) * c <= oi::cdiv(Wo, bx) * bx) bx = c;
  // grid.z holds at most 65535 planes: more (e.g. 128 x 512 feature planes of a StyleGAN2 layer) go in several launches
  for (int p0 = 0; p0 < BC; p0 += 65535) {
    const int np = BC - p0 < 65535 ? BC - p0 : 65535;
    hipLaunchKernelGGL(upfirdn2d_kernel, dim3(oi::cdiv(Wo, bx), Ho, np), dim3(bx), fh * fw * sizeof(float),
                       oi::as_stream(stream), x + (size_t)p0 * H * W, f, y + (size_t)p0 * Ho * Wo, np, H, W, Ho, Wo, fh, fw,
                       upx, upy, downx, downy, padx0, pady0, flip, gain);
  }
  return oi::check_launch("oi_upfirdn2d");
}

int oi_affine_grid_sample_fwd(const float* x, const float* theta, float* y, int B, int C, int Hi, int Wi, int Ho,
                              int Wo, oi_stream_t stream) {
  OI_REQUIRE(x && theta && y, "oi_affine_grid_sample_fwd: null pointer");
  OI_REQUIRE(B > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "oi_affine_grid_sample_fwd: bad shape");
  const long long n = (long long)B * C * Ho * Wo;
  hipLaunchKernelGGL(affine_grid_sample_fwd_kernel, dim3(oi::cdiv(n, 256)), dim3(256), 0, oi::as_stream(stream), x,
                     theta, y, B, C, Hi, Wi, Ho, Wo);
  return oi::check_launch("oi_affine_grid_sample_fwd");
}

int oi_affine_grid_sample_bwd(const float* gy, const float* theta, float* gx, int B, int C, int Hi, int Wi, int Ho,
                              int Wo, oi_stream_t stream) {
  OI_REQUIRE(gy && theta && gx, "oi_affine_grid_sample_bwd: null pointer");
  OI_REQUIRE(B > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "oi_affine_grid_sample_bwd: bad shape");
  hipStream_t st = oi::as_stream(stream);
  hipError_t e = oi::zero_output_async(gx, (size_t)B * C * Hi * Wi, st);
  if (e != hipSuccess) return oi::fail(OI_ERR_LAUNCH, "oi_affine_grid_sample_bwd: memset: %s", hipGetErrorString(e));
  const long long n = (long long)B * C * Ho * Wo;
  hipLaunchKernelGGL(affine_grid_sample_bwd_kernel, dim3(oi::cdiv(n, 256)), dim3(256), 0, st, gy, theta, gx, B, C, Hi,
                     Wi, Ho, Wo);
  return oi::check_launch("oi_affine_grid_sample_bwd");
}

int oi_fused_bias_act(float* out, const float* x, const float* bias, const float* refer, int act, int grad, float alpha,
                      float scale, long long size_x, long long step_b, int size_b, oi_stream_t stream) {
  OI_REQUIRE(out && x, "oi_fused_bias_act: null pointer");
  OI_REQUIRE(size_x > 0 && (bias == nullptr || (step_b > 0 && size_b > 0)), "oi_fused_bias_act: bad shape");
  OI_REQUIRE((act == 1 || act == 3) && grad >= 0 && grad <= 2, "oi_fused_bias_act: act %d grad %d (linear = 1, lrelu = 3)", act,
             grad);
  hipLaunchKernelGGL(fused_bias_act_kernel, dim3(oi::cdiv(size_x, 256)), dim3(256), 0, oi::as_stream(stream), out, x, bias,
                     refer, act == 3 ? alpha : 1.0f, grad == 2 ? 0.0f : 1.0f, grad == 1, scale, size_x, step_b, size_b);
  return oi::check_launch("oi_fused_bias_act");
}

int oi_grid_sample_fwd(const float* x, const float* grid, float* y, int N, int C, int Hi, int Wi, int Ho, int Wo,
                       oi_stream_t stream) {
  OI_REQUIRE(x && grid && y, "oi_grid_sample_fwd: null pointer");
  OI_REQUIRE(N > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "oi_grid_sample_fwd: bad shape");
  const long long n = (long long)N * C * Ho * Wo;
  hipLaunchKernelGGL(grid_sample_fwd_kernel, dim3(oi::cdiv(n, 256)), dim3(256), 0, oi::as_stream(stream), x, grid, y, N, C, Hi,
                     Wi, Ho, Wo);
  return oi::check_launch("oi_grid_sample_fwd");
}

int oi_grid_sample_bwd(const float* gy, const float* x, const float* grid, float* gx, float* ggrid, int N, int C, int Hi,
                       int Wi, int Ho, int Wo, oi_stream_t stream) {
  OI_REQUIRE(gy && grid && (gx || ggrid), "oi_grid_sample_bwd: null pointer");
  OI_REQUIRE(ggrid == nullptr || x != nullptr, "oi_grid_sample_bwd: the grid gradient needs the input");
  OI_REQUIRE(N > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "oi_grid_sample_bwd: bad shape");
  hipStream_t st = oi::as_stream(stream);
  if (gx != nullptr) {
    hipError_t e = oi::zero_output_async(gx, (size_t)N * C * Hi * Wi, st);
    if (e != hipSuccess) return oi::fail(OI_ERR_LAUNCH, "oi_grid_sample_bwd: zero fill: %s", hipGetErrorString(e));
  }
  const long long n = (long long)N * Ho * Wo;
  hipLaunchKernelGGL(grid_sample_bwd_kernel, dim3(oi::cdiv(n, 256)), dim3(256), 0, st, gy, x, grid, gx, ggrid, N, C, Hi, Wi, Ho,
                     Wo);
  return oi::check_launch("oi_grid_sample_bwd");
}

int oi_reflect_pad_fwd(const float* x, float* y, int BC, int H, int W, int px0, int px1, int py0, int py1,
                       oi_stream_t stream) {
  OI_REQUIRE(x && y, "oi_reflect_pad_fwd: null pointer");
  OI_REQUIRE(px0 >= 0 && px1 >= 0 && py0 >= 0 && py1 >= 0 && px0 < W && px1 < W && py0 < H && py1 < H,
             "oi_reflect_pad_fwd: padding must be in [0, size)");
  const int Ho = H + py0 + py1, Wo = W + px0 + px1;
  const long long n = (long long)BC * Ho * Wo;
  hipLaunchKernelGGL(reflect_pad_kernel<false>, dim3(oi::cdiv(n, 256)), dim3(256), 0, oi::as_stream(stream), x, y, BC,
                     H, W, px0, py0, Ho, Wo);
  return oi::check_launch("oi_reflect_pad_fwd");
}

int oi_reflect_pad_bwd(const float* gy, float* gx, int BC, int H, int W, int px0, int px1, int py0, int py1,
                       oi_stream_t stream) {
  OI_REQUIRE(gy && gx, "oi_reflect_pad_bwd: null pointer");
  OI_REQUIRE(px0 >= 0 && px1 >= 0 && py0 >= 0 && py1 >= 0 && px0 < W && px1 < W && py0 < H && py1 < H,
             "oi_reflect_pad_bwd: padding must be in [0, size)");
  hipStream_t st = oi::as_stream(stream);
  hipError_t e = oi::zero_output_async(gx, (size_t)BC * H * W, st);
  if (e != hipSuccess) return oi::fail(OI_ERR_LAUNCH, "oi_reflect_pad_bwd: memset: %s", hipGetErrorString(e));
  const int Ho = H + py0 + py1, Wo = W + px0 + px1;
  const long long n = (long long)BC * Ho * Wo;
  hipLaunchKernelGGL(reflect_pad_kernel<true>, dim3(oi::cdiv(n, 256)), dim3(256), 0, st, gy, gx, BC, H, W, px0, py0,
                     Ho, Wo);
  return oi::check_launch("oi_reflect_pad_bwd");
}

int oi_outputs_prezeroed_stream(oi_stream_t stream, int on) {
  const int was = oi::set_stream_prezeroed(oi::as_stream(stream), on != 0);
  if (was < 0) return oi::fail(OI_ERR_INVALID_ARG, "oi_outputs_prezeroed_stream: more than %d streams hold the declaration", oi::PREZERO_SLOTS);
  return was;
}

int oi_version(void) { return 1; }
const char* oi_arch(void) { return "gfx950"; }
const char* oi_last_error(void) { return oi::err_buf(); }

}  // extern "C"
