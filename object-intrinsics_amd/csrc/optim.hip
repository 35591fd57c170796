// Multi-tensor optimiser / EMA steps for gfx950: ONE launch updates every parameter of a network.
//
// Replaces the per-step parameter loops of the reference's training loop -- torch.optim.Adam for the generator,
// torch.optim.RMSprop for the two discriminators (configs/train.yaml:133-147, stepped at
// src/trainers/gan_pose_trainer.py:142,191) and the EMA lerp of the generator (src/utils/ema.py:26-32) -- which
// torch runs as 4-6 foreach launches per optimiser.  HBM-bound: reads 16 B and writes 12 B per parameter (Adam).
// Arithmetic follows torch's single-tensor formulas (torch/optim/adam.py, rmsprop.py; Tensor.lerp) in fp32.
#include <algorithm>

#include "oi_common.h"

namespace {

constexpr int CHUNK = 4096;  // elements per workgroup

__device__ __forceinline__ float lerp_(float a, float b, float w) {
  // at::lerp: the form that is exact at both ends
  return w < 0.5f ? a + w * (b - a) : b - (b - a) * (1.0f - w);
}

__global__ void __launch_bounds__(256)
multi_adam_kernel(const oi_mt_chunk* __restrict__ table, float lr_over_bc1, float beta1, float beta2, float eps,
                  float bc2_sqrt) {
  const oi_mt_chunk c = table[blockIdx.x];
  for (int i = threadIdx.x; i < c.n; i += 256) {
    const float g = c.g[i];
    const float m = lerp_(c.s0[i], g, 1.0f - beta1);
    const float v = fmaf(g * g, 1.0f - beta2, c.s1[i] * beta2);
    c.s0[i] = m;
    c.s1[i] = v;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    c.p[i] = c.p[i] - lr_over_bc1 * (m / denom);
  }
}

__global__ void __launch_bounds__(256)
multi_rmsprop_kernel(const oi_mt_chunk* __restrict__ table, float lr, float alpha, float eps) {
  const oi_mt_chunk c = table[blockIdx.x];
  for (int i = threadIdx.x; i < c.n; i += 256) {
    const float g = c.g[i];
    const float sq = fmaf(g * g, 1.0f - alpha, c.s0[i] * alpha);
    c.s0[i] = sq;
    c.p[i] = c.p[i] - lr * (g / (sqrtf(sq) + eps));
  }
}

__global__ void __launch_bounds__(256)
multi_lerp_kernel(const oi_mt_chunk* __restrict__ table, float beta) {
  const oi_mt_chunk c = table[blockIdx.x];
  for (int i = threadIdx.x; i < c.n; i += 256) c.p[i] = lerp_(c.g[i], c.p[i], beta);  // p_ema <- p.lerp(p_ema, beta)
}

__global__ void __launch_bounds__(256)
multi_copy_kernel(const oi_mt_chunk* __restrict__ table) {
  const oi_mt_chunk c = table[blockIdx.x];
  for (int i = threadIdx.x; i < c.n; i += 256) c.p[i] = c.g[i];
}

// The inputs of one captured step in ONE launch: up to 4 device-to-device copies into the graph's static buffers plus up to
// 64 floats that travel in the kernel arguments (augmentation matrices, scalar weights) -- instead of a copy launch per
// tensor, a pinned staging buffer + host-to-device copy for the matrices and a fill for every scalar.
struct StageArgs {
  const float* src[4];
  float* dst[4];
  long long n[4];
  float imm[64];
  float* imm_dst;
  int n_copies, n_imm;
};
__global__ void __launch_bounds__(256) stage_inputs_kernel(StageArgs a) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
  if (gid < a.n_imm) a.imm_dst[gid] = a.imm[gid];
  for (int c = 0; c < a.n_copies; ++c)
    for (long long i = gid; i < a.n[c]; i += stride) a.dst[c][i] = a.src[c][i];
}

}  // namespace

extern "C" {

int oi_stage_inputs(const float* const* srcs, float* const* dsts, const long long* counts, int n_copies, const float* imm,
                    int n_imm, float* imm_dst, oi_stream_t stream) {
  OI_REQUIRE(n_copies >= 0 && n_copies <= 4 && n_imm >= 0 && n_imm <= 64, "oi_stage_inputs: %d copies (<= 4), %d immediates (<= 64)",
             n_copies, n_imm);
  OI_REQUIRE(n_imm == 0 || (imm != nullptr && imm_dst != nullptr), "oi_stage_inputs: immediates without a destination");
  StageArgs a{};
  long long most = n_imm;
  for (int c = 0; c < n_copies; ++c) {
    OI_REQUIRE(srcs[c] != nullptr && dsts[c] != nullptr && counts[c] >= 0, "oi_stage_inputs: copy %d", c);
    a.src[c] = srcs[c];
    a.dst[c] = dsts[c];
    a.n[c] = counts[c];
    most = std::max(most, counts[c]);
  }
  for (int i = 0; i < n_imm; ++i) a.imm[i] = imm[i];
  a.imm_dst = imm_dst;
  a.n_copies = n_copies;
  a.n_imm = n_imm;
  if (most == 0) return OI_OK;
  const int blocks = (int)std::min<long long>(1024, (most + 1023) / 1024);
  hipLaunchKernelGGL(stage_inputs_kernel, dim3(blocks), dim3(256), 0, oi::as_stream(stream), a);
  return oi::check_launch("oi_stage_inputs");
}


int oi_zero_fill(float* p, long long n_floats, oi_stream_t stream) {
  OI_REQUIRE(n_floats >= 0 && (p != nullptr || n_floats == 0), "oi_zero_fill: null pointer");
  if (oi::zero_async(p, (size_t)n_floats, oi::as_stream(stream)) != hipSuccess) return oi::check_launch("oi_zero_fill");
  return OI_OK;
}

int oi_mt_chunk_elems(void) { return CHUNK; }

int oi_multi_adam(const oi_mt_chunk* table, int n_chunks, float lr, float beta1, float beta2, float eps,
                  float bias_correction1, float bias_correction2_sqrt, oi_stream_t stream) {
  OI_REQUIRE(table != nullptr || n_chunks == 0, "oi_multi_adam: null table");
  OI_REQUIRE(n_chunks >= 0 && bias_correction1 > 0.f && bias_correction2_sqrt > 0.f, "oi_multi_adam: bad arguments");
  if (n_chunks == 0) return OI_OK;
  hipLaunchKernelGGL(multi_adam_kernel, dim3(n_chunks), dim3(256), 0, oi::as_stream(stream), table,
                     lr / bias_correction1, beta1, beta2, eps, bias_correction2_sqrt);
  return oi::check_launch("oi_multi_adam");
}

int oi_multi_rmsprop(const oi_mt_chunk* table, int n_chunks, float lr, float alpha, float eps, oi_stream_t stream) {
  OI_REQUIRE(table != nullptr || n_chunks == 0, "oi_multi_rmsprop: null table");
  OI_REQUIRE(n_chunks >= 0, "oi_multi_rmsprop: n_chunks=%d", n_chunks);
  if (n_chunks == 0) return OI_OK;
  hipLaunchKernelGGL(multi_rmsprop_kernel, dim3(n_chunks), dim3(256), 0, oi::as_stream(stream), table, lr, alpha, eps);
  return oi::check_launch("oi_multi_rmsprop");
}

int oi_multi_lerp(const oi_mt_chunk* table, int n_chunks, float beta, oi_stream_t stream) {
  OI_REQUIRE(table != nullptr || n_chunks == 0, "oi_multi_lerp: null table");
  OI_REQUIRE(n_chunks >= 0, "oi_multi_lerp: n_chunks=%d", n_chunks);
  if (n_chunks == 0) return OI_OK;
  hipLaunchKernelGGL(multi_lerp_kernel, dim3(n_chunks), dim3(256), 0, oi::as_stream(stream), table, beta);
  return oi::check_launch("oi_multi_lerp");
}

int oi_multi_copy(const oi_mt_chunk* table, int n_chunks, oi_stream_t stream) {
  OI_REQUIRE(table != nullptr || n_chunks == 0, "oi_multi_copy: null table");
  OI_REQUIRE(n_chunks >= 0, "oi_multi_copy: n_chunks=%d", n_chunks);
  if (n_chunks == 0) return OI_OK;
  hipLaunchKernelGGL(multi_copy_kernel, dim3(n_chunks), dim3(256), 0, oi::as_stream(stream), table);
  return oi::check_launch("oi_multi_copy");
}

}  // extern "C"
