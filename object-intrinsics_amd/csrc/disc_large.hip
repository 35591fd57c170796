// Large-batch forward of the DC discriminator (src/models/discriminator.py:57-85) without gradient: batch >= 16, the B = 64 row of
// SURVEY.md 8(d).  At these sizes the layers are ordinary GEMMs (M = B Ho Wo pixels, N = Cout, K = 16 Cin), and what limited the
// round-2 kernel (conv4x4_tiled_f16x3_kernel, disc.hip) was not the matrix cores but everything in front of them: every
// workgroup gathered its im2col tile element by element from NCHW fp32 and split it into fp16 limbs on the VALU -- once per
// output-channel tile and once per overlapping window, 8-32 times per input value -- about three VALU cycles per MFMA cycle.
// Here the operands are prepared ONCE and the inner loop is loads + MFMAs:
//   * activations travel between the layers as NHWC fp16 limb planes (hi, lo: 22 mantissa bits, the F16X3 form), written by the
//     producing layer's epilogue after its LeakyReLU.  With the K order (tap, cin) an MFMA B fragment -- 8 consecutive k of one
//     pixel -- is 8 consecutive channels at one input pixel: ONE 16-byte load per limb, straight into the registers the MFMA
//     reads (no LDS, no conversion; padding = the buffer descriptor's range check);
//   * weights are packed once per parameter version into fp16 limb images in MFMA A-fragment order and streamed through a
//     two-slot LDS ring by LDS-DMA, shared by the workgroup's four waves (pixel on the MFMA column, as in the MLP kernels);
//   * parallelism for the late layers (M = 1024 at 4 x 4) comes from splitting K across workgroups; the partial planes are added
//     in a FIXED order by the reduce kernel that also applies the LeakyReLU and writes the next layer's limb planes: no atomics,
//     bit-reproducible.
// Three fp16 MFMAs per product, fp32 accumulation: the arithmetic of the tiled kernel it replaces (2e-5 of the fp64 oracle).
#include <type_traits>

#include "oi_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int TN = 128;            // output channels per workgroup (4 MFMA row blocks)
constexpr int TK = 64;             // k per chunk = 64 consecutive input channels at one tap
constexpr int TM = 128;            // pixels per workgroup (4 waves x 32)
constexpr int IMG = TN * TK * 2;   // bytes of one limb plane of a chunk image (16 KiB); hi plane then lo plane
constexpr int MAX_BLOCKS = 8;
// timing ablations of dl_gemm_kernel (results garbage): bit 1 = B fragments loaded for the first two chunks only, 2 = no epilogue
// stores, 4 = weight image loaded / stored for the first two chunks only, 8 = no MFMAs
#ifndef OI_DL_ABL
#define OI_DL_ABL 0
#endif

__device__ __forceinline__ void split_half(float v, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;
  lo = (_Float16)(v - (float)hi);
}

// ---- weights of a GEMM layer: w [Cout][Cin][4][4] fp32 -> [n-tile][chunk][plane][row block t][k-step s][lane][8 x fp16]
__global__ void dl_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ img, int Cin, int Cout) {
  const int nchunk = Cin * 16 / TK;
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (ntile, chunk, t, s, lane)
  const long long total = (long long)(Cout / TN) * nchunk * 4 * 4 * 64;
  if (id >= total) return;
  const int lane = id & 63, s = (id >> 6) & 3, t = (id >> 8) & 3;
  const long long tc = id >> 10;
  const int chunk = tc % nchunk, ntile = tc / nchunk;
  const int n = ntile * TN + 32 * t + (lane & 31);
  f16x8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = chunk * TK + 16 * s + 8 * (lane >> 5) + e;
    const int tap = k / Cin, cin = k - tap * Cin;
    _Float16 h_, l_;
    split_half(w[((size_t)n * Cin + cin) * 16 + tap], h_, l_);
    hi[e] = h_, lo[e] = l_;
  }
  _Float16* base = img + (size_t)tc * (2 * IMG / 2);
  *reinterpret_cast<f16x8*>(base + ((t * 4 + s) * 64 + lane) * 8) = hi;
  *reinterpret_cast<f16x8*>(base + IMG / 2 + ((t * 4 + s) * 64 + lane) * 8) = lo;
}

// ---- first layer (Cin <= 4): x NCHW fp32 -> NHWC limb planes of lrelu(conv(x)), on the matrix cores as well: one k-step of 16
// = the 16 taps of one input channel.  A lane's B fragment is 2 kernel rows x 4 columns of its pixel's window (8 loads, bounds
// checked, split to limbs on the spot); the weights are packed as A fragments [row block][channel][lane] (12 KiB for 3 -> 64).
// (The first version ran on the VALU -- thread = (pixel, 8 channels): 670 instructions per thread, 30 us at batch 64.)
__global__ void dl_pack1_kernel(const float* __restrict__ w, _Float16* __restrict__ img, int Cin, int C1) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;   // (t, cin, lane)
  if (id >= (C1 / 32) * Cin * 64) return;
  const int lane = id & 63, cin = (id >> 6) % Cin, t = (id >> 6) / Cin;
  f16x8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    _Float16 h_, l_;
    split_half(w[((size_t)(32 * t + (lane & 31)) * Cin + cin) * 16 + 8 * (lane >> 5) + e], h_, l_);
    hi[e] = h_, lo[e] = l_;
  }
  *reinterpret_cast<f16x8*>(img + (size_t)id * 8) = hi;
  *reinterpret_cast<f16x8*>(img + ((size_t)(C1 / 32) * Cin * 64 + id) * 8) = lo;
}

template <int NT>   // C1 = 32 NT output channels
__global__ void __launch_bounds__(256) dl_conv1_kernel(const float* __restrict__ x, const _Float16* __restrict__ wimg,
                                                       _Float16* __restrict__ hi, _Float16* __restrict__ lo, int B, int Cin,
                                                       int H, int W, float slope) {
  constexpr int C1 = 32 * NT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, kg = lane >> 5;
  const int Ho = H / 2, Wo = W / 2;
  const long long M = (long long)B * Ho * Wo, m = ((long long)blockIdx.x * 4 + wave) * 32 + j;
  const bool valid = m < M;
  const long long mm = valid ? m : M - 1;
  const int ox = mm % Wo, oy = (mm / Wo) % Ho, b = mm / ((long long)Wo * Ho);
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const size_t plane = (size_t)NT * Cin * 64 * 8;
  for (int cin = 0; cin < Cin; ++cin) {
    const float* xc = x + ((size_t)b * Cin + cin) * H * W;
    f16x8 vh, vl;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int iy = 2 * oy - 1 + 2 * kg + r;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int ix = 2 * ox - 1 + c;
        const float v = (valid && iy >= 0 && iy < H && ix >= 0 && ix < W) ? xc[(size_t)iy * W + ix] : 0.f;
        _Float16 a_, c_;
        split_half(v, a_, c_);
        vh[4 * r + c] = a_, vl[4 * r + c] = c_;
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f16x8 ah = *reinterpret_cast<const f16x8*>(wimg + ((size_t)(t * Cin + cin) * 64 + lane) * 8);
      const f16x8 al = *reinterpret_cast<const f16x8*>(wimg + plane + ((size_t)(t * Cin + cin) * 64 + lane) * 8);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, vh, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, vl, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, vh, acc[t], 0, 0, 0);
    }
  }
  if (!valid) return;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f16x4 h_, l_;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[t][4 * g + e];
        v = v > 0.f ? v : v * slope;
        _Float16 a_, c_;
        split_half(v, a_, c_);
        h_[e] = a_, l_[e] = c_;
      }
      *reinterpret_cast<f16x4*>(hi + (size_t)m * C1 + 32 * t + 8 * g + 4 * kg) = h_;
      *reinterpret_cast<f16x4*>(lo + (size_t)m * C1 + 32 * t + 8 * g + 4 * kg) = l_;
    }
}

// ---- GEMM layer: out[m][n] = sum_k A[m][k] W[n][k], 4 x 4 stride 2 pad 1, NHWC limb planes in.
// SPLIT = false: writes lrelu(out) as limb planes; true: writes this K range's fp32 partial plane [split][M][Cout].
#ifndef OI_DL_WPE
#define OI_DL_WPE 2
#endif
#ifndef OI_DL_WGS
#define OI_DL_WGS 256
#endif
template <bool SPLIT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, OI_DL_WPE))) dl_gemm_kernel(const _Float16* __restrict__ in_hi, const _Float16* __restrict__ in_lo,
                                                      const char* __restrict__ wimg, _Float16* __restrict__ out_hi,
                                                      _Float16* __restrict__ out_lo, float* __restrict__ part, int B, int Cin,
                                                      int Hin, int Win, int Cout, int chunks_per_split, float slope) {
  __shared__ __attribute__((aligned(16))) char lds[2 * 2 * IMG];   // two ring slots of (hi | lo) = 64 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, kg = lane >> 5;
  const int Ho = Hin / 2, Wo = Win / 2;
  const long long M = (long long)B * Ho * Wo;
  const long long m = (long long)blockIdx.x * TM + wave * 32 + j;
  const bool valid = m < M;
  const long long mm = valid ? m : M - 1;
  const int ox = mm % Wo, oy = (mm / Wo) % Ho, b = mm / ((long long)Wo * Ho);
  const int nchunk = Cin * 16 / TK;
  const int c_beg = blockIdx.z * chunks_per_split, c_end = min(nchunk, c_beg + chunks_per_split);
  const size_t plane_bytes = (size_t)B * Hin * Win * Cin * 2;
  const __amdgpu_buffer_rsrc_t rs_hi = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(in_hi), 0, (int)plane_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_lo = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(in_lo), 0, (int)plane_bytes, 0x00020000);
  const int l16 = 16 * lane;

  // B fragments of a chunk: 4 k-steps x (hi, lo), one 16-byte load each; an out-of-image tap reads past the descriptor's range
  // and returns zeros (the convolution's padding)
  u32x4 bh[3][4], bl[3][4];
  auto load_b_steps = [&](int c, u32x4 (&h_)[4], u32x4 (&l_)[4]) {
    const int k0 = c * TK, tap = k0 / Cin, cin0 = k0 - tap * Cin;
    const int iy = 2 * oy - 1 + (tap >> 2), ix = 2 * ox - 1 + (tap & 3);
    const bool inb = valid && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
    const unsigned off = inb ? (unsigned)(((((size_t)b * Hin + iy) * Win + ix) * Cin + cin0 + 8 * kg) * 2) : 0xfffffe00u;
#pragma unroll
    for (int s = 0; s < 4; ++s) {   // k-step s: 16 channels = 32 bytes further on (folded into the immediate offset)
      h_[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_hi, (int)(off + 32u * s), 0, 0);
      l_[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_lo, (int)(off + 32u * s), 0, 0);
    }
  };
  // weight image of chunk c: 32 KiB = 128 bytes per thread, through registers (requested two chunks ahead, written to the ring
  // slot one chunk ahead).  Not LDS-DMA: with `buffer_load ... lds` in flight hipcc can no longer count vmcnt for the ordinary
  // loads beside it and waits for the NEWEST request in front of every use (measured: 2.9 us per chunk instead of 0.7).
  u32x4 wr[8];
  const char* wbase = wimg + (size_t)blockIdx.y * nchunk * (2 * IMG) + tid * 16;
  auto load_w = [&](int c) {
#pragma unroll
    for (int q = 0; q < 8; ++q) wr[q] = *reinterpret_cast<const u32x4*>(wbase + (size_t)c * (2 * IMG) + q * 4096);
  };
  auto store_w = [&](int slot) {
#pragma unroll
    for (int q = 0; q < 8; ++q) *reinterpret_cast<u32x4*>(lds + slot * (2 * IMG) + q * 4096 + tid * 16) = wr[q];
  };
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // The eight A fragments (4 row blocks x hi / lo) of k-step s + 1 are requested before the twelve MFMAs of k-step s are issued:
  // with one wave per SIMD an LDS round trip (~130 cycles with four waves reading) in front of every product otherwise -- a
  // distance of one product left 55 % of the matrix rate (measured: 58 cycles per MFMA).
#ifndef OI_DL_APF
#define OI_DL_APF 1
#endif
  auto compute = [&](int slot, const u32x4 (&h_)[4], const u32x4 (&l_)[4]) {
    const char* a0 = lds + slot * (2 * IMG) + l16;
    f32x4 ah[2][4], al[2][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      ah[0][t] = *reinterpret_cast<const f32x4*>(a0 + (t * 4) * 1024);
      al[0][t] = *reinterpret_cast<const f32x4*>(a0 + IMG + (t * 4) * 1024);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s + 1 < 4 && !(OI_DL_ABL & 16)) {   // (ablation 16: A fragments read for k-step 0 only)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          ah[(s + 1) & 1][t] = *reinterpret_cast<const f32x4*>(a0 + (t * 4 + s + 1) * 1024);
          al[(s + 1) & 1][t] = *reinterpret_cast<const f32x4*>(a0 + IMG + (t * 4 + s + 1) * 1024);
        }
      }
      const f16x8 vh = __builtin_bit_cast(f16x8, h_[s]), vl = __builtin_bit_cast(f16x8, l_[s]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f16x8 wh = __builtin_bit_cast(f16x8, ah[(OI_DL_ABL & 16) ? 0 : (s & 1)][t]), wl = __builtin_bit_cast(f16x8, al[(OI_DL_ABL & 16) ? 0 : (s & 1)][t]);
        if (OI_DL_ABL & 32) {   // (ablation 32: the reads stay, no MFMAs)
          asm volatile("" ::"v"(wh), "v"(wl));
          continue;
        }
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, vh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, vl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, vh, acc[t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // Pipeline: the image of chunk c + 2 and its B fragments are REQUESTED while chunk c is multiplied; the image of chunk c + 1
  // (in registers since the previous step) goes to the other ring slot first.  One barrier per chunk: behind it every wave has
  // finished reading the slot that is about to be overwritten and chunk c's image is visible.  A bare s_barrier behind
  // lgkmcnt(0): __syncthreads() carries workgroup fences for which hipcc drains vmcnt(0) -- the requests in flight included.
  auto step = [&](int c, auto pos) {
    constexpr int P = decltype(pos)::value;   // position of chunk c in the rotation of three B-fragment sets
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (c + 1 < c_end && !(OI_DL_ABL & 4)) store_w((c + 1 - c_beg) & 1);
    if (c + 2 < c_end) {
      if (!(OI_DL_ABL & 4)) load_w(c + 2);
      if (!(OI_DL_ABL & 1)) load_b_steps(c + 2, bh[(P + 2) % 3], bl[(P + 2) % 3]);
    }
    if (!(OI_DL_ABL & 8)) compute((c - c_beg) & 1, bh[P], bl[P]);
  };
  if (c_beg < c_end) {
    load_w(c_beg);
    load_b_steps(c_beg, bh[0], bl[0]);
    store_w(0);
  }
  if (c_beg + 1 < c_end) {
    load_w(c_beg + 1);
    load_b_steps(c_beg + 1, bh[1], bl[1]);
  }
  for (int c = c_beg; c < c_end; c += 6) {   // (6: ring-slot parity and B-set rotation are both compile-time inside the body)
    step(c, std::integral_constant<int, 0>());
    if (c + 1 < c_end) step(c + 1, std::integral_constant<int, 1>());
    if (c + 2 < c_end) step(c + 2, std::integral_constant<int, 2>());
    if (c + 3 < c_end) step(c + 3, std::integral_constant<int, 0>());
    if (c + 4 < c_end) step(c + 4, std::integral_constant<int, 1>());
    if (c + 5 < c_end) step(c + 5, std::integral_constant<int, 2>());
  }
  // ---- epilogue: slot r of block t in lane-half kg is channel 32 t + 8 (r >> 2) + 4 kg + (r & 3) of pixel m
  if (!valid) return;
  if ((OI_DL_ABL & 2) && acc[0][0] != 12345.f) return;
  const int n0 = blockIdx.y * TN;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + 32 * t + 8 * g + 4 * kg;
      if constexpr (SPLIT) {
        const f32x4 v = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
        *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.z * M + m) * Cout + n) = v;
      } else {
        f16x4 h_, l_;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[t][4 * g + e];
          v = v > 0.f ? v : v * slope;
          _Float16 a, c_;
          split_half(v, a, c_);
          h_[e] = a, l_[e] = c_;
        }
        *reinterpret_cast<f16x4*>(out_hi + (size_t)m * Cout + n) = h_;
        *reinterpret_cast<f16x4*>(out_lo + (size_t)m * Cout + n) = l_;
      }
    }
}

// ---- partial planes -> lrelu -> limb planes, split s = 0, 1, ... added in that order (thread = 8 channels of a pixel)
__global__ void __launch_bounds__(256) dl_reduce_kernel(const float* __restrict__ part, _Float16* __restrict__ hi,
                                                        _Float16* __restrict__ lo, long long MN8, long long MN, int S, float slope) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= MN8) return;
  f32x4 a = *reinterpret_cast<const f32x4*>(part + i * 8), c = *reinterpret_cast<const f32x4*>(part + i * 8 + 4);
  for (int s = 1; s < S; ++s) {
    a += *reinterpret_cast<const f32x4*>(part + s * MN + i * 8);
    c += *reinterpret_cast<const f32x4*>(part + s * MN + i * 8 + 4);
  }
  f16x8 h_, l_;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = e < 4 ? a[e] : c[e - 4];
    v = v > 0.f ? v : v * slope;
    _Float16 x_, y_;
    split_half(v, x_, y_);
    h_[e] = x_, l_[e] = y_;
  }
  *reinterpret_cast<f16x8*>(hi + i * 8) = h_;
  *reinterpret_cast<f16x8*>(lo + i * 8) = l_;
}

// ---- head: 4 x 4 valid convolution of the last 4 x 4 map (src/models/discriminator.py:74, 82-84) -> logits [B][out_dim].
// whp [out][p = ky 4 + kx][c] fp32 (packed); one workgroup per image, fixed summation order.
__global__ void __launch_bounds__(1024) dl_head_kernel(const _Float16* __restrict__ hi, const _Float16* __restrict__ lo,
                                                       const float* __restrict__ whp, const float* __restrict__ bias,
                                                       float* __restrict__ logits, int C, int out_dim) {
  __shared__ float red[8][16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, K = 16 * C;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k = tid * 8; k < K; k += 8192) {   // (one trip at C = 512: every load of the kernel is issued before the first FMA)
    const f16x8 h_ = *reinterpret_cast<const f16x8*>(hi + (size_t)b * K + k), l_ = *reinterpret_cast<const f16x8*>(lo + (size_t)b * K + k);
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = (float)h_[e] + (float)l_[e];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      if (o < out_dim) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(whp + (size_t)o * K + k), w1 = *reinterpret_cast<const f32x4*>(whp + (size_t)o * K + k + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[o] = fmaf(a[e], w0[e], acc[o]), acc[o] = fmaf(a[4 + e], w1[e], acc[o]);
      }
    }
  }
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    const float s_ = oi::wave_sum(acc[o]);   // fixed tree inside the wave, then the 16 waves in order
    if (lane == 0) red[o][wave] = s_;
  }
  __syncthreads();
  if (tid < out_dim) {
    float s_ = 0.f;
    for (int w_ = 0; w_ < 16; ++w_) s_ += red[tid][w_];
    logits[(size_t)b * out_dim + tid] = s_ + (bias != nullptr ? bias[tid] : 0.f);
  }
}

__global__ void dl_pack_head_kernel(const float* __restrict__ w, float* __restrict__ whp, int C, int out_dim) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // (o, p, c)
  if (i >= out_dim * 16 * C) return;
  const int c = i % C, p = (i / C) % 16, o = i / (16 * C);
  whp[i] = w[((size_t)o * C + c) * 16 + p];
}

struct Plan {
  int n_blocks, out_dim, chans[MAX_BLOCKS + 1];
  size_t w_off[MAX_BLOCKS + 1], head_off, packed_bytes;   // w_off[1]: the first layer's A fragments
};
// -> false when this network is not covered (the caller takes the general chain)
bool make_plan(const int* chans, int n_blocks, int out_dim, Plan& p) {
  if (n_blocks < 2 || n_blocks > MAX_BLOCKS || out_dim < 1 || out_dim > 8) return false;
  if (chans[0] < 1 || chans[0] > 4 || chans[1] % 32 != 0 || chans[1] > 128) return false;
  p.n_blocks = n_blocks, p.out_dim = out_dim;
  size_t off = 0;
  for (int l = 0; l <= n_blocks; ++l) p.chans[l] = chans[l];
  p.w_off[1] = 0;
  off += ((size_t)2 * (chans[1] / 32) * chans[0] * 64 * 16 + 255) / 256 * 256;
  for (int l = 2; l <= n_blocks; ++l) {   // block l - 1: chans[l - 1] -> chans[l]
    if (chans[l - 1] % TK != 0 || chans[l] % TN != 0) return false;
    p.w_off[l] = off;
    off += (size_t)(chans[l] / TN) * (chans[l - 1] * 16 / TK) * (2 * IMG);
  }
  p.head_off = off;
  off += (size_t)out_dim * 16 * chans[n_blocks] * 4;
  p.packed_bytes = off;
  return true;
}
int split_for(long long M, int Cout, int nchunk) {
  const long long tiles = ((M + TM - 1) / TM) * (Cout / TN);
  int S = 1;
  while (tiles * S < OI_DL_WGS && S * 2 <= nchunk && S < 32) S *= 2;
  return S;
}
struct Work {
  size_t act_off[MAX_BLOCKS + 1], part_off, total;
};
Work make_work(const Plan& p, int B, int H) {
  Work w;
  size_t off = 0, part = 0;
  int h = H;
  for (int l = 1; l <= p.n_blocks; ++l) {
    h /= 2;
    const size_t plane = ((size_t)B * h * h * p.chans[l] * 2 + 255) / 256 * 256;
    w.act_off[l] = off;
    off += 2 * plane;
    if (l >= 2) {
      const long long M = (long long)B * h * h;
      const int S = split_for(M, p.chans[l], p.chans[l - 1] * 16 / TK);
      if (S > 1) part = part > (size_t)S * M * p.chans[l] * 4 ? part : (size_t)S * M * p.chans[l] * 4;
    }
  }
  w.part_off = off;
  w.total = off + part + 256;
  return w;
}

}  // namespace

extern "C" {

size_t oi_disc_large_packed_bytes(const int* chans, int n_blocks, int out_dim) {
  Plan p;
  return (chans && make_plan(chans, n_blocks, out_dim, p)) ? p.packed_bytes : 0;
}

int oi_disc_large_pack(const float* const* w_blocks, const float* w_head, const int* chans, int n_blocks, int out_dim, void* packed,
                       oi_stream_t stream) {
  OI_REQUIRE(w_blocks && w_head && chans && packed, "oi_disc_large_pack: null pointer");
  Plan p;
  if (!make_plan(chans, n_blocks, out_dim, p)) return oi::fail(OI_ERR_UNSUPPORTED, "oi_disc_large_pack: network not covered");
  hipStream_t st = oi::as_stream(stream);
  char* base = reinterpret_cast<char*>(packed);
  {
    const int n1 = (chans[1] / 32) * chans[0] * 64;
    hipLaunchKernelGGL(dl_pack1_kernel, dim3((n1 + 255) / 256), dim3(256), 0, st, w_blocks[0], reinterpret_cast<_Float16*>(base + p.w_off[1]),
                       chans[0], chans[1]);
  }
  for (int l = 2; l <= n_blocks; ++l) {
    const long long total = (long long)(chans[l] / TN) * (chans[l - 1] * 16 / TK) * 1024;
    hipLaunchKernelGGL(dl_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w_blocks[l - 1],
                       reinterpret_cast<_Float16*>(base + p.w_off[l]), chans[l - 1], chans[l]);
  }
  const int nh = out_dim * 16 * chans[n_blocks];
  hipLaunchKernelGGL(dl_pack_head_kernel, dim3((nh + 255) / 256), dim3(256), 0, st, w_head, reinterpret_cast<float*>(base + p.head_off),
                     chans[n_blocks], out_dim);
  return oi::check_launch("oi_disc_large_pack");
}

size_t oi_disc_large_workspace_bytes(const int* chans, int n_blocks, int out_dim, int B, int H) {
  Plan p;
  if (!chans || !make_plan(chans, n_blocks, out_dim, p) || B < 1) return 0;
  int h = H;
  for (int l = 0; l < n_blocks; ++l) h /= 2;
  if (h != 4 || (H >> n_blocks) << n_blocks != H) return 0;
  return make_work(p, B, H).total;
}

int oi_disc_fwd_large(const float* x, const float* w1, const void* packed, const float* bhead, void* workspace, size_t workspace_bytes,
                      float* logits, const int* chans, int n_blocks, int out_dim, int B, int H, float slope, oi_stream_t stream) {
  (void)w1;   // (kept in the signature: the first layer's weights are part of the packed images since its move to the matrix cores)
  OI_REQUIRE(x && packed && workspace && logits && chans, "oi_disc_fwd_large: null pointer");
  Plan p;
  if (!make_plan(chans, n_blocks, out_dim, p)) return oi::fail(OI_ERR_UNSUPPORTED, "oi_disc_fwd_large: network not covered");
  const size_t need = oi_disc_large_workspace_bytes(chans, n_blocks, out_dim, B, H);
  if (need == 0) return oi::fail(OI_ERR_UNSUPPORTED, "oi_disc_fwd_large: %d x %d input with %d blocks", H, H, n_blocks);
  OI_REQUIRE(workspace_bytes >= need, "oi_disc_fwd_large: workspace of %zu bytes, need %zu", workspace_bytes, need);
  const Work wk = make_work(p, B, H);
  hipStream_t st = oi::as_stream(stream);
  char* ws = reinterpret_cast<char*>(workspace);
  const char* pk = reinterpret_cast<const char*>(packed);
  auto hi_of = [&](int l) { return reinterpret_cast<_Float16*>(ws + wk.act_off[l]); };
  auto lo_of = [&](int l, int h_) {
    const size_t plane = ((size_t)B * h_ * h_ * chans[l] * 2 + 255) / 256 * 256;
    return reinterpret_cast<_Float16*>(ws + wk.act_off[l] + plane);
  };
  int h = H / 2;
  {  // block 0
    const long long M = (long long)B * h * h;
    const dim3 grid((unsigned)((M + 127) / 128));
    const _Float16* w1img = reinterpret_cast<const _Float16*>(pk + p.w_off[1]);
    switch (chans[1] / 32) {
      case 1: hipLaunchKernelGGL(dl_conv1_kernel<1>, grid, dim3(256), 0, st, x, w1img, hi_of(1), lo_of(1, h), B, chans[0], H, H, slope); break;
      case 2: hipLaunchKernelGGL(dl_conv1_kernel<2>, grid, dim3(256), 0, st, x, w1img, hi_of(1), lo_of(1, h), B, chans[0], H, H, slope); break;
      case 3: hipLaunchKernelGGL(dl_conv1_kernel<3>, grid, dim3(256), 0, st, x, w1img, hi_of(1), lo_of(1, h), B, chans[0], H, H, slope); break;
      default: hipLaunchKernelGGL(dl_conv1_kernel<4>, grid, dim3(256), 0, st, x, w1img, hi_of(1), lo_of(1, h), B, chans[0], H, H, slope); break;
    }
    int rc = oi::check_launch("oi_disc_fwd_large(conv1)");
    if (rc != OI_OK) return rc;
  }
  for (int l = 2; l <= n_blocks; ++l) {
    const int hin = h;
    h /= 2;
    const long long M = (long long)B * h * h;
    const int nchunk = chans[l - 1] * 16 / TK, S = split_for(M, chans[l], nchunk), cps = (nchunk + S - 1) / S;
    OI_REQUIRE((size_t)B * hin * hin * chans[l - 1] * 2 < (1ull << 31), "oi_disc_fwd_large: activation plane too large");
    const dim3 grid((unsigned)((M + TM - 1) / TM), chans[l] / TN, S);
    float* part = reinterpret_cast<float*>(ws + wk.part_off);
    if (S == 1) {
      hipLaunchKernelGGL(dl_gemm_kernel<false>, grid, dim3(256), 0, st, hi_of(l - 1), lo_of(l - 1, hin), pk + p.w_off[l], hi_of(l),
                         lo_of(l, h), nullptr, B, chans[l - 1], hin, hin, chans[l], cps, slope);
    } else {
      hipLaunchKernelGGL(dl_gemm_kernel<true>, grid, dim3(256), 0, st, hi_of(l - 1), lo_of(l - 1, hin), pk + p.w_off[l], nullptr, nullptr,
                         part, B, chans[l - 1], hin, hin, chans[l], cps, slope);
      const long long MN = M * chans[l];
      hipLaunchKernelGGL(dl_reduce_kernel, dim3((unsigned)((MN / 8 + 255) / 256)), dim3(256), 0, st, part, hi_of(l), lo_of(l, h), MN / 8, MN,
                         S, slope);
    }
    int rc = oi::check_launch("oi_disc_fwd_large(gemm)");
    if (rc != OI_OK) return rc;
  }
  hipLaunchKernelGGL(dl_head_kernel, dim3(B), dim3(1024), 0, st, hi_of(n_blocks), lo_of(n_blocks, h),
                     reinterpret_cast<const float*>(pk + p.head_off), bhead, logits, chans[n_blocks], out_dim);
  return oi::check_launch("oi_disc_fwd_large(head)");
}

}  // extern "C"
