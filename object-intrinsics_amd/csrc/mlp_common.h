// Device-side building blocks shared by the forward (mlp.hip) and backward (mlp_bwd.hip) FiLM-SIREN kernels.
#pragma once
#include "oi_common.h"

namespace oimlp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int C = 128;           // hidden width (W in the reference config)
constexpr int NL_SDF = 8;        // FiLM layers of the SDF net
constexpr int NMAT = 16;         // 7 forward + 7 transposed + colour head + colour head transposed
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
constexpr int H_WSCALE = 2720;   // [16] 2^-k_m: inverse of the power-of-two scale baked into image m (1 unless F16X3)
constexpr int H_BOUND = 2736;    // [16] max_i sum_k |image_m[i][k]| of the (scaled) image m: |W_img x| <= bound * max|x|
constexpr int H_STATUS = 2752;   // [16] 1 if image m holds a non-finite weight (F16X3; read by oi_mlp_pack_status)
constexpr int H_FLOATS = 2816;
constexpr size_t H_BYTES = H_FLOATS * 4;

__host__ __device__ constexpr int layer_bytes(int prec) {
  return prec == OI_PREC_BF16 ? 32768 : (prec == OI_PREC_BF16X6 ? 98304 : 65536);
}
// Behind the 16 MFMA images: the 8 forward matrices once more as plain row-major fp32 [m][out][in] (W_1..W_7, Wv[:, :128]).
// The weight-gradient GEMM of the backward reads them for the FiLM-scale identity
//   gamma_l[f] * d gamma_l[f] = sum_i W_l[f][i] dW_l[f][i] + b_l[f] db_l[f]
// (phi_l = gamma_l (W_l a + b_l) + beta_l depends on gamma_l, W_l, b_l only through the products gamma_l W_l, gamma_l b_l).
constexpr int NPLAIN = 8;
__host__ __device__ constexpr size_t plain_off(int prec) { return H_BYTES + (size_t)NMAT * layer_bytes(prec); }
__host__ __device__ constexpr size_t packed_total_bytes(int prec) { return plain_off(prec) + (size_t)NPLAIN * C * C * 4; }

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
// the MLP kernel
// ------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// two fp32 values -> one dword of the hi plane and one of the lo plane: hi = v_cvt_pk_f16_f32 (round to nearest even); the
// residual a - float(hi) is exact in fp32, and v_fma_mixlo_f16 / v_fma_mixhi_f16 form it (fp16 operand read straight from
// the packed dword) and round it to fp16 into the low / high half of the lo dword: 3 instructions per pair, bit-identical
// to the cvt-back / subtract / cvt_pk form (tools/dbg/run_fwd_ab.sh compares whole forward passes bit for bit) and to
// round 2's v_fma_mix_f32 x 2 + v_cvt_pk (4 per pair, OI_SPLIT_MIXLO=0).
#ifndef OI_SPLIT_MIXLO
#define OI_SPLIT_MIXLO 1
#endif
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  const f16x2 hv = {(_Float16)a, (_Float16)b};
  hi = __builtin_bit_cast(unsigned, hv);
#ifdef OI_F3_ABL_PIECE
  if (OI_F3_ABL_PIECE & 8) {
    lo = hi;
    return;
  }
#endif
#if OI_SPLIT_MIXLO
  // the residual of each value, rounded to fp16 straight into its half of the lo dword (the residual is exact in fp32,
  // so this is the same single rounding v_cvt_pk_f16_f32 applies): 3 instructions per pair
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(lo)
      : "v"(hi), "v"(a), "v"(b));
#else
  float ra, rb;
  asm("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(ra), "=&v"(rb)
      : "v"(hi), "v"(a), "v"(b));
  const f16x2 lv = {(_Float16)ra, (_Float16)rb};
  lo = __builtin_bit_cast(unsigned, lv);
#endif
}
// eight fp32 values -> the hi and lo fp16 limb fragments (8 x fp16 each) of an MFMA operand: 16 instructions
__device__ __forceinline__ void split8_pairs(const float* v, f16x8& hi, f16x8& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) split_pair(v[2 * d], v[2 * d + 1], h[d], l[d]);
  hi = __builtin_bit_cast(f16x8, u32x4{h[0], h[1], h[2], h[3]});
  lo = __builtin_bit_cast(f16x8, u32x4{l[0], l[1], l[2], l[3]});
}


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

// Same copy through the LDS-DMA path (global_load_lds: 16 B per lane straight into LDS, no VGPR round trip, no
// ds_write): 4 wavefronts x N/4 instructions of 1 KiB.  The caller waits for vmcnt(0) before its barrier.
template <int PREC>
__device__ __forceinline__ void stage_layer_dma(char* lds, const char* __restrict__ src, int wave, int lane) {
  constexpr int NCHUNK = layer_bytes(PREC) / 1024;
#pragma unroll
  for (int c0 = 0; c0 < NCHUNK / 4; ++c0) {
    const int c = c0 * 4 + wave;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c * 1024 + lane * 16),
                                     (__attribute__((address_space(3))) void*)(lds + L_WBUF + c * 1024), 16, 0, 0);
  }
}

// The same LDS-DMA through a buffer descriptor over the 16 images: the image / chunk offset travels in an SGPR and the
// lane offset (16 * lane) in ONE VGPR.  With flat pointers hipcc keeps a 64-bit address pair per chunk alive across the
// whole kernel (16 pairs = 32 VGPRs, spilled and reloaded around every layer in the backward sweep).
template <int PREC>
__device__ __forceinline__ void stage_layer_rs(char* lds, __amdgpu_buffer_rsrc_t rs, int image, int wave, int l16) {
  constexpr int NCHUNK = layer_bytes(PREC) / 1024;
#pragma unroll
  for (int c0 = 0; c0 < NCHUNK / 4; ++c0) {
    const int c = c0 * 4 + wave;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + L_WBUF + c * 1024), 16, l16,
                                             image * layer_bytes(PREC) + c * 1024, 0, 0);
  }
}

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

struct LayOff {  // per-layer runtime VGPR bases (everything else is an immediate)
  int wl;   // 16*lane + ring slot base
  int wh;   // wl + 32768
  int wq;   // wl + 65536
  int f16;  // 16*h + 1024*layer  (FiLM rows of this layer)
};

struct NoHook {
  __device__ __forceinline__ void operator()(int) const {}
};

// HOOK(s), s = 0..7, runs after the MFMAs that consumed act[8s .. 8s+7]: the reverse sweep uses it to issue the NEXT
// layer's scratch loads into registers that have just died, one full GEMM ahead of their use.
template <int PREC, class HOOK = NoHook>
__device__ __forceinline__ void gemm_layer2(const char* lds, const LayOff& y, const float (&act)[64], f32x16 (&acc)[4],
                                            HOOK hook = HOOK()) {
  if constexpr (PREC == OI_PREC_F32) {
    f32x4 a[4], an[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = lds_f4(lds, ((t * 16 + 0) * 1024) & 32767, t < 2 ? y.wl : y.wh);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      if (g < 15) {
#pragma unroll
        for (int t = 0; t < 4; ++t) an[t] = lds_f4(lds, ((t * 16 + g + 1) * 1024) & 32767, t < 2 ? y.wl : y.wh);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][k], act[4 * g + k], acc[t], 0, 0, 0);
      }
      if (g & 1) hook(g >> 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] = an[t];
    }
  } else if constexpr (PREC == OI_PREC_BF16X6) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      bf16x8 bh, bm, bl;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float v = act[8 * s + i];
        bh[i] = (__bf16)v;
        const float r1 = v - (float)bh[i];
        bm[i] = (__bf16)r1;
        bl[i] = (__bf16)(r1 - (float)bm[i]);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bf16x8 wh = __builtin_bit_cast(bf16x8, lds_f4(lds, (t * 8 + s) * 1024, y.wl));
        const bf16x8 wm = __builtin_bit_cast(bf16x8, lds_f4(lds, (t * 8 + s) * 1024, y.wh));
        const bf16x8 wl = __builtin_bit_cast(bf16x8, lds_f4(lds, (t * 8 + s) * 1024, y.wq));
        // the six products of weight >= 2^-24, smallest first
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, bl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, bm, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, bm, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, bh, acc[t], 0, 0, 0);
      }
      hook(s);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (PREC == OI_PREC_F16X3) {
    f32x4 ah[4], ahn[4], al[4], aln[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      ah[t] = lds_f4(lds, (t * 8 + 0) * 1024, y.wl);
      al[t] = lds_f4(lds, (t * 8 + 0) * 1024, y.wh);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < 7) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          ahn[t] = lds_f4(lds, (t * 8 + s + 1) * 1024, y.wl);
          aln[t] = lds_f4(lds, (t * 8 + s + 1) * 1024, y.wh);
        }
      }
      f16x8 bh, bl;
      split8_pairs(&act[8 * s], bh, bl);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f16x8 wh = __builtin_bit_cast(f16x8, ah[t]);
        const f16x8 wl = __builtin_bit_cast(f16x8, al[t]);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh, acc[t], 0, 0, 0);
      }
      hook(s);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        ah[t] = ahn[t];
        al[t] = aln[t];
      }
    }
  } else {
    f32x4 ah[4], ahn[4], al[4], aln[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      ah[t] = lds_f4(lds, (t * 8 + 0) * 1024, y.wl);
      if constexpr (PREC == OI_PREC_BF16X3) al[t] = lds_f4(lds, (t * 8 + 0) * 1024, y.wh);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < 7) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          ahn[t] = lds_f4(lds, (t * 8 + s + 1) * 1024, y.wl);
          if constexpr (PREC == OI_PREC_BF16X3) aln[t] = lds_f4(lds, (t * 8 + s + 1) * 1024, y.wh);
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
      hook(s);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        ah[t] = ahn[t];
        if constexpr (PREC == OI_PREC_BF16X3) al[t] = aln[t];
      }
    }
  }
}

// acc[t][r] (+)= sum_k A[32t + row][k] * act[k]  with the packed A image in LDS: ONE implementation for every kernel
// (gemm_layer2); callers that address the image through a LaneOff go through this adapter.
template <int PREC>
__device__ __forceinline__ void gemm_layer(const char* lds, const LaneOff& o, const float (&act)[64],
                                           f32x16 (&acc)[4]) {
  LayOff y;
  y.wl = o.l16 + L_WBUF;
  y.wh = o.l16hi + L_WBUF;
  y.wq = y.wl + 65536;
  y.f16 = 0;
  gemm_layer2<PREC>(lds, y, act, acc);
}

// sin and cos of one fp32 phase.  Three accurate forms (OI_TRIG_HW), measured A/B on one MI355X (full f16x3 kernel):
//   2 (default)  whole-period reduction in revolutions + v_sin_f32 / v_cos_f32                      1.03 ms
//   1            n = round(phi/pi) by magic-number FMA, 2-constant Cody-Waite, then v_sin / v_cos     1.08 ms
//   0            the same reduction, weighted-minimax polynomials on [-pi/2, pi/2] (1.4e-8 / 7e-9), shared sign by
//                xor: 19 VALU ops per pair, all FP                                                     1.12 ms
// All three are 1e-7-class (hardware: 1.3e-7 absolute on [-1/2, 1/2] revolutions, tools/dbg/trans_acc.hip) and give the
// same parity margins; the transcendental instructions issue beside the FP VALU / MFMA work, the polynomial adds to it.
// Fast form (FAST = true, bf16 throughput mode): v_sin / v_cos on phi/(2 pi) with no reduction (error ~ |phi| * 6e-8).
#ifndef OI_TRIG_HW
#define OI_TRIG_HW 2
#endif
template <bool FAST>
__device__ __forceinline__ void sincos_(float x, float& s, float& c) {
  if constexpr (FAST) {
    s = __sinf(x);
    c = __cosf(x);
  } else {
#if OI_TRIG_HW == 2
    // Whole-period reduction in REVOLUTIONS, then the transcendental unit: n = round(x / 2pi); the fractional part
    // x/(2pi) - n comes out of ONE fma (exact product, single rounding of a value <= 1/2) plus the low word of 1/(2pi),
    // so it is good to ~3e-8 revolutions for the |phi| <= a few hundred radians of this network; v_sin / v_cos are
    // accurate to 1.3e-7 absolute on [-1/2, 1/2] (tools/dbg/trans_acc.hip).  No sign fix-up, four FP VALU operations,
    // and the two transcendental instructions issue beside the FP VALU / MFMA work instead of adding to it.
    const float q = x * 0.15915494309189533577f;
    const float n = __builtin_rintf(q);
    float r = fmaf(x, 0.15915494309189533577f, -n);
    r = fmaf(x, 6.4206383266e-09f, r);  // 1/(2 pi) - float(1/(2 pi)) = 0.15915494309189535 - 0.15915493667125702
    s = __builtin_amdgcn_sinf(r);
    c = __builtin_amdgcn_cosf(r);
#else
    constexpr float MAGIC = 12582912.f;  // 1.5 * 2^23: nf = MAGIC + round(x / pi), parity of n in mantissa bit 0
    const float nf = fmaf(x, 0.318309886183790671538f, MAGIC);
    const float n = nf - MAGIC;
    float r = fmaf(n, -3.1415927410125732f, x);
    r = fmaf(n, 8.742278000372485e-08f, r);  // float(pi) - pi
#if OI_TRIG_HW == 1
    // the transcendental unit on the exactly reduced argument (|r| <= pi/2 -> |t| <= 1/4 revolution)
    const float t = r * 0.15915494309189533577f;
    const float ps = __builtin_amdgcn_sinf(t);
    const float pc = __builtin_amdgcn_cosf(t);
#else
    const float t = r * r;
    float ps = fmaf(t, 2.5999420359e-06f, -1.9806565251e-04f);
    ps = fmaf(t, ps, 8.3330161870e-03f);
    ps = fmaf(t, ps, -1.6666656733e-01f);
    ps = fmaf(t * r, ps, r);
    float pc = fmaf(t, -2.6192776659e-07f, 2.4769255106e-05f);
    pc = fmaf(t, pc, -1.3888567919e-03f);
    pc = fmaf(t, pc, 4.1666656733e-02f);
    pc = fmaf(t * t, pc, fmaf(t, -0.5f, 1.0f));
#endif
    const unsigned sign = __builtin_bit_cast(unsigned, nf) << 31;
    s = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, ps) ^ sign);
    c = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, pc) ^ sign);
#endif
  }
}

// scratch of one wave tile, addressed through a buffer descriptor: voffset = 16*lane (VGPR),
// soffset = slot*16 KiB + g*1 KiB (SGPR / immediate) -> no per-access address VGPRs.
struct WaveScratch {
  __amdgpu_buffer_rsrc_t rs;
  __device__ __forceinline__ void store(int slot, int g, int l16, f32x4 v) const {
    oi::buffer_store_b128<OI_FWD_NT_ST>(__builtin_bit_cast(u32x4, v), rs, l16, slot * 16384 + g * 1024);
  }
  __device__ __forceinline__ f32x4 load(int slot, int g, int l16) const {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, l16, slot * 16384 + g * 1024, OI_FWD_NT_LD));
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

// Power-of-two normalisation of one point's 128-vector (64 entries in this lane, 64 in lane ^ 32) to max |.| in
// [2^13, 2^14): the fp16 split of the F16X3 mode then never leaves the normal range of its hi limb and keeps 22
// mantissa bits.  Scales `act` in place and returns 1/scale (exact).
__device__ __forceinline__ float pow2_normalise(float (&act)[64], float* lane_max = nullptr) {
  float m = 0.f;
#pragma unroll
  for (int k = 0; k < 64; k += 2) m = fmaxf(m, fmaxf(fabsf(act[k]), fabsf(act[k + 1])));
  if (lane_max != nullptr) *lane_max = m;  // max |.| of this lane's 64 entries, before the scaling
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  int eb = (__builtin_bit_cast(int, m) >> 23) & 0xff;
  eb = eb < 14 ? 14 : (eb > 254 ? 254 : eb);
  float sc = __builtin_bit_cast(float, (267 - eb) << 23);  // 2^(13 - (eb - 127))
  asm volatile("" : "+v"(sc) : : "memory");                // keeps the GEMM's ds_reads behind the producers of act
#pragma unroll
  for (int k = 0; k < 64; ++k) act[k] *= sc;
  return __builtin_bit_cast(float, (eb - 13) << 23);
}

// GEMM of the backward kernels: acc = W_img * act (accumulators must come in zeroed).  Returns the factor the
// accumulators still have to be multiplied by: 1 for the unscaled images, 2^-k_m (x 1/normalisation) for F16X3.
// NORM = false for inputs known to lie in [-1, 1] (sin activations).
template <int PREC, bool NORM>
__device__ __forceinline__ float gemm_scaled(const char* lds, const LaneOff& o, float (&act)[64], f32x16 (&acc)[4],
                                             float inv_img) {
  float f = 1.f;
  if constexpr (PREC == OI_PREC_F16X3) {
    f = inv_img;
    if constexpr (NORM) f *= pow2_normalise(act);
  }
  gemm_layer<PREC>(lds, o, act, acc);
  return f;
}

__device__ __forceinline__ void acc_zero(f32x16 (&acc)[4]) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}

__device__ __forceinline__ void init_bias(const char* lds, const LaneOff& o, f32x16 (&acc)[4]) {
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const f32x4 b = lds_f4(lds, L_FILM + (2 * C + grp_f0(g)) * 4, o.h16);
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[g >> 2][4 * (g & 3) + k] = b[k];
  }
}


// ------------------------------------------------------------------------------------------
// forward-kernel plumbing shared by mlp.hip (v2 kernels, all precisions) and mlp_fwd3.hip (register-resident F16X3)
// ------------------------------------------------------------------------------------------
// Scratch accessor of the forward kernel.  In the bf16 throughput mode the parked gamma*cos(phi) / feature
// fragments are stored as fp16 (|c| < 64, 2^-11 relative: far below the bf16 operand rounding), halving the
// one HBM stream that bounds that mode (profiles/r1_*: 4.9 GB per launch in fp32).
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <bool HALF>
struct FwdScratch {
  __amdgpu_buffer_rsrc_t rs;
  __device__ __forceinline__ void store(int slot, int g, const LaneOff& o, f32x4 v) const {
    if constexpr (HALF) {
      f16x4 hv;
#pragma unroll
      for (int k = 0; k < 4; ++k) hv[k] = (_Float16)v[k];
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hv), rs, o.l16 >> 1, slot * 8192 + g * 512, 0);
    } else {
      oi::buffer_store_b128<OI_FWD_NT_ST>(__builtin_bit_cast(u32x4, v), rs, o.l16, slot * 16384 + g * 1024);
    }
  }
  __device__ __forceinline__ f32x4 load(int slot, int g, const LaneOff& o) const {
    if constexpr (HALF) {
      const f16x4 hv = __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(rs, o.l16 >> 1, slot * 8192 + g * 512, 0));
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (float)hv[k];
      return v;
    } else {
      return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o.l16, slot * 16384 + g * 1024, OI_FWD_NT_LD));
    }
  }
};

constexpr int V2_WAVES = 8;                   // scratch is sized for 8-wave tiles (an upper bound for 4-wave tiles)
constexpr int V2_TILE = V2_WAVES * WAVE_PTS;  // 256 points
constexpr int V2_FILM = 0;                    // [9][gamma 128 | beta' 128],  beta' = gamma * bias + beta
constexpr int V2_TABS = 9 * 1024;             // 9216
constexpr int V2_WBUF = V2_TABS + H_TABS_END * 4;  // 15488
// Workgroup shape.  Every mode runs 8 wavefronts (256 points) per workgroup and CU with a double-buffered image
// ring, except BF16X6 (96 KiB images: one slot).  F16X3 can alternatively be built with 4-wave workgroups and ONE
// 64 KiB slot (79 KiB LDS, two independent workgroups per CU): measured 1.5 % slower (tools/bench_c5.py) -- on
// gfx950 FP VALU and MFMA time of co-resident waves add up, so running the two workgroups out of phase buys nothing.
#ifndef OI_F16X3_FULL_WAVES
#define OI_F16X3_FULL_WAVES 8
#endif
#ifndef OI_F16X3_SDF_WAVES
#define OI_F16X3_SDF_WAVES 8
#endif
__host__ __device__ constexpr int v2_waves(int prec, bool full) {
  return prec == OI_PREC_F16X3 ? (full ? OI_F16X3_FULL_WAVES : OI_F16X3_SDF_WAVES) : 8;
}
// BF16X6 images are 96 KiB: a single ring slot, refilled behind a barrier while the VALU phase runs
__host__ __device__ constexpr bool v2_two_slots(int prec, bool full) {
  return prec != OI_PREC_BF16X6 && !(prec == OI_PREC_F16X3 && v2_waves(prec, full) == 4);
}
__host__ __device__ constexpr int v2_lds_total(int prec, bool full) {
  return V2_WBUF + (v2_two_slots(prec, full) ? 2 : 1) * layer_bytes(prec);
}


// One layer image -> ring slot `slot` through the LDS-DMA path (16 B per lane straight into LDS).  `rs` spans the 16
// images, `src_off` = byte offset of the image in it.  A wave copies a contiguous share, 4 KiB per (M0, soffset) setting:
// the instruction's immediate offset advances the LDS and the global address alike, so four 1 KiB copies share one
// M0 / soffset pair and no per-chunk 64-bit address is formed on the VALU.
template <int PREC, int NWAVES>
__device__ __forceinline__ void prefetch_image(char* lds, __amdgpu_buffer_rsrc_t rs, int src_off, int slot, int wave, int l16) {
  constexpr int PER_WAVE = layer_bytes(PREC) / NWAVES;
  static_assert(PER_WAVE % 4096 == 0, "a wave's share of an image is a multiple of 4 KiB");
#pragma unroll
  for (int q = 0; q < PER_WAVE / 4096; ++q) {
    const int c = wave * PER_WAVE + q * 4096;
    auto* dst = (__attribute__((address_space(3))) void*)(lds + V2_WBUF + slot * layer_bytes(PREC) + c);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, l16, src_off + c, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, l16, src_off + c, 1024, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, l16, src_off + c, 2048, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, l16, src_off + c, 3072, 0);
  }
}

template <int PREC>
__device__ __forceinline__ LayOff lay_off(const LaneOff& o, int slot, int layer) {
  LayOff r;
  r.wl = o.l16 + V2_WBUF + slot * layer_bytes(PREC);
  r.wh = r.wl + 32768;
  r.wq = r.wl + 65536;
  r.f16 = o.h16 + V2_FILM + layer * 1024;
  return r;
}

// wait for this wave's LDS-DMA, then rendezvous: next image resident, previous ring slot free
// TRAILING > 0 would let that many younger VMEM operations (the FiLM phase's scratch stores) stay in flight; it relies
// on in-order vmcnt retirement between LDS-DMA loads and stores and measured no gain (the wait is for the slowest
// wave, not for write acknowledgements), so every call site waits for vmcnt(0).
template <int TRAILING = 0>
__device__ __forceinline__ void ring_sync() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TRAILING) : "memory");
  __syncthreads();
}


// mlp_fwd3.hip: register-resident F16X3 forward with gradient (+ albedo); scratch = one 16 KiB slot per wave tile
size_t full3_scratch_bytes(int B, long long n_per_elem);
size_t full3_blob_offset(int B, long long n_per_elem);   // byte offset of the per-element blobs inside that scratch
int launch_full3_f16x3(const float* pts, const void* packed, const float* gamma, const float* beta, float* sdf,
                       float* grad, float* rgb, float* feat, void* scratch, int B, long long n, int fast_trig, bool blob_ready,
                       hipStream_t st);

// mlp_fwd3b.hip: register-resident BF16 forward with gradient (+ albedo); no scratch
// (scratch: full3_bf16_scratch_bytes(B) = 15 per-element bf16 images of 32 KiB per batch element, built per call)
size_t full3_bf16_scratch_bytes(int B);
int launch_full3_bf16(const float* pts, const void* packed, const float* gamma, const float* beta, float* sdf, float* grad,
                      float* rgb, float* feat, void* scratch, int B, long long n, int fast_trig, hipStream_t st);

}  // namespace oimlp
