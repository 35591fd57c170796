// The scalar losses of one GAN training step for gfx950, one launch forward and one backward.
//
// Replaces what the reference composes from a dozen tiny tensor ops per term (src/loss/gan.py:5-22 compute_grad2 /
// GANLoss "bce" = F.binary_cross_entropy_with_logits, src/loss/position.py:4-18 PositionLoss "mse", summed with the weights
// of gan_pose_trainer.py:122-137, 163-190): at batch 1 the operands are 1..7 logits and one image-sized gradient, and
// every one of those ops is a launch of ~4 us -- 45 of the ~200 launches of a discriminator step.
//   real  = mean_b softplus(-d_real[b][0])                       BCE with logits against 1
//   fake  = mean_b softplus(+d_fake[b][0])                       ... against 0
//   reg   = mean_b sum_i gx[b][i]^2                              R1 (gx = d sum_b d_real[b][0] / d x_real)
//   aux   = mean_{b, k >= 1} (d_fake[b][k] - pose[b][k - 1])^2   pose regression on the remaining logits
//   total = real + fake + reg_w * reg + aux_w * aux              (every term optional: null pointer = 0)
#include <algorithm>

#include "oi_common.h"

namespace {

// F.binary_cross_entropy_with_logits against a constant target t (0 or 1), torch's stable form:
// (1 - t) x + m + log(exp(-m) + exp(-x - m)),  m = max(-x, 0)
__device__ __forceinline__ float bce_logits(float x, float t) {
  const float m = fmaxf(-x, 0.f);
  return (1.f - t) * x + m + logf(expf(-m) + expf(-x - m));
}

__global__ void __launch_bounds__(1024)
gan_losses_fwd_kernel(const float* __restrict__ d_real, const float* __restrict__ d_fake, const float* __restrict__ pose,
                      const float* __restrict__ gx, const float* __restrict__ aux_w, float reg_w, float* __restrict__ out,
                      int B, int K, long long N) {
  __shared__ float part[16];
  const int tid = threadIdx.x;
  float s = 0.f;
  if (gx != nullptr)
    for (long long i = tid; i < (long long)B * N; i += 1024) s = fmaf(gx[i], gx[i], s);
  s = oi::wave_sum(s);
  if ((tid & 63) == 0) part[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    float reg = 0.f;
    for (int w = 0; w < 16; ++w) reg += part[w];
    reg /= (float)B;
    float real = 0.f, fake = 0.f, aux = 0.f;
    for (int b = 0; b < B; ++b) {
      if (d_real != nullptr) real += bce_logits(d_real[(size_t)b * K], 1.f);
      if (d_fake != nullptr) {
        fake += bce_logits(d_fake[(size_t)b * K], 0.f);
        if (pose != nullptr)
          for (int k = 1; k < K; ++k) {
            const float d = d_fake[(size_t)b * K + k] - pose[(size_t)b * (K - 1) + k - 1];
            aux = fmaf(d, d, aux);
          }
      }
    }
    real /= (float)B;
    fake /= (float)B;
    if (pose != nullptr && K > 1) aux /= (float)(B * (K - 1));
    const float aw = (pose != nullptr && aux_w != nullptr) ? aux_w[0] : 0.f;
    out[0] = real + fake + reg_w * reg + aw * aux;
    out[1] = real + fake;
    out[2] = reg;
    out[3] = fake;
    out[4] = real;
    out[5] = aux;
  }
}

__global__ void __launch_bounds__(256)
gan_losses_bwd_kernel(const float* __restrict__ g_total, const float* __restrict__ d_real, const float* __restrict__ d_fake,
                      const float* __restrict__ pose, const float* __restrict__ gx, const float* __restrict__ aux_w,
                      float reg_w, float* __restrict__ g_real, float* __restrict__ g_fake, float* __restrict__ g_gx, int B,
                      int K, long long N) {
  const float go = g_total[0];
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gx != nullptr) {
    const float c = 2.f * reg_w * go / (float)B;
    for (long long i = gid; i < (long long)B * N; i += (long long)gridDim.x * 256) g_gx[i] = c * gx[i];
  }
  if (gid < (long long)B * K) {  // d/dx BCE = (sigmoid(x) - t) / B;  d/dx MSE = 2 (x - p) / count
    const int b = (int)(gid / K), k = (int)(gid % K);
    if (g_real != nullptr) g_real[gid] = k == 0 ? go * (oi::sigmoidf_(d_real[gid]) - 1.f) / (float)B : 0.f;
    if (g_fake != nullptr) {
      float g = 0.f;
      if (k == 0) g = go * oi::sigmoidf_(d_fake[gid]) / (float)B;
      else if (pose != nullptr) g = go * aux_w[0] * 2.f * (d_fake[gid] - pose[(size_t)b * (K - 1) + k - 1]) / (float)(B * (K - 1));
      g_fake[gid] = g;
    }
  }
}

// total = sum_i w_i * term_i of up to 8 device scalars (the generator's loss: GAN term per discriminator + the weighted
// regularisers, gan_pose_trainer.py:122-137): one launch instead of a multiply and an add per term, each way
struct WeightedTerms {
  const float* p[8];
  float w[8];
  int n;
};
__global__ void weighted_sum_fwd_kernel(WeightedTerms t, float* __restrict__ out) {
  float s = 0.f;
  for (int i = 0; i < t.n; ++i) s = fmaf(t.w[i], t.p[i][0], s);
  out[0] = s;
}
__global__ void weighted_sum_bwd_kernel(WeightedTerms t, const float* __restrict__ g_out, float* __restrict__ g_terms) {
  if ((int)threadIdx.x < t.n) g_terms[threadIdx.x] = t.w[threadIdx.x] * g_out[0];
}

// The two scalars the renderer derives from the compositing kernel's reductions r4 = (sum of eikonal errors inside the
// sphere, number of such samples, sum of |sdf| surface terms, .) -- reference renderer.py:430-446:
//   gradient_error = r4[0] / (r4[1] + 1e-5),   surface_loss = r4[2] * inv_nt
__global__ void render_scalars_fwd_kernel(const float* __restrict__ r4, float inv_nt, float* __restrict__ out2) {
  out2[0] = r4[0] / (r4[1] + 1e-5f);
  out2[1] = r4[2] * inv_nt;
}
__global__ void render_scalars_bwd_kernel(const float* __restrict__ r4, const float* __restrict__ g_err,
                                          const float* __restrict__ g_surf, float inv_nt, float* __restrict__ g_r4) {
  const float den = r4[1] + 1e-5f;
  const float ge = g_err != nullptr ? g_err[0] : 0.f;
  g_r4[0] = ge / den;
  g_r4[1] = -ge * r4[0] / (den * den);
  g_r4[2] = g_surf != nullptr ? g_surf[0] * inv_nt : 0.f;
  g_r4[3] = 0.f;
}

// The scalar glue of a render: what the reference derives from four 0-dim parameters with ~10 tensor ops per forward.
__global__ void scalar_glue_kernel(const float* __restrict__ variance, const float* __restrict__ ambient,
                                   const float* __restrict__ specular, const float* __restrict__ shininess,
                                   float* __restrict__ out5, float* __restrict__ packed3) {
  const float inv_s = fminf(fmaxf(expf(variance[0] * 10.0f), 1e-6f), 1e6f);   // renderer.py:404
  const float amb = 1.0f / (1.0f + expf(-ambient[0]));                          // lighting.py:50-60
  out5[0] = inv_s;
  out5[1] = 1.0f / inv_s;
  out5[2] = amb;
  out5[3] = 1.0f - amb;
  out5[4] = fmaxf(specular[0], 0.f);
  packed3[0] = ambient[0];
  packed3[1] = specular[0];
  packed3[2] = shininess[0];
}

}  // namespace

extern "C" {

int oi_scalar_glue(const float* variance, const float* param_ambient, const float* param_specular, const float* param_shininess,
                   float* out5, float* packed3, oi_stream_t stream) {
  OI_REQUIRE(variance != nullptr && param_ambient != nullptr && param_specular != nullptr && param_shininess != nullptr &&
             out5 != nullptr && packed3 != nullptr, "oi_scalar_glue: null pointer");
  hipLaunchKernelGGL(scalar_glue_kernel, dim3(1), dim3(1), 0, oi::as_stream(stream), variance, param_ambient, param_specular,
                     param_shininess, out5, packed3);
  return oi::check_launch("oi_scalar_glue");
}

int oi_render_scalars_fwd(const float* r4, float inv_nt, float* out2, oi_stream_t stream) {
  OI_REQUIRE(r4 != nullptr && out2 != nullptr, "oi_render_scalars_fwd: null pointer");
  hipLaunchKernelGGL(render_scalars_fwd_kernel, dim3(1), dim3(1), 0, oi::as_stream(stream), r4, inv_nt, out2);
  return oi::check_launch("oi_render_scalars_fwd");
}

int oi_render_scalars_bwd(const float* r4, const float* g_err, const float* g_surf, float inv_nt, float* g_r4,
                          oi_stream_t stream) {
  OI_REQUIRE(r4 != nullptr && g_r4 != nullptr, "oi_render_scalars_bwd: null pointer");
  hipLaunchKernelGGL(render_scalars_bwd_kernel, dim3(1), dim3(1), 0, oi::as_stream(stream), r4, g_err, g_surf, inv_nt, g_r4);
  return oi::check_launch("oi_render_scalars_bwd");
}

int oi_weighted_sum_fwd(const float* const* terms, const float* weights, int n, float* out, oi_stream_t stream) {
  OI_REQUIRE(terms != nullptr && weights != nullptr && out != nullptr && n >= 1 && n <= 8, "oi_weighted_sum_fwd: n=%d (1..8)", n);
  WeightedTerms t{};
  t.n = n;
  for (int i = 0; i < n; ++i) {
    OI_REQUIRE(terms[i] != nullptr, "oi_weighted_sum_fwd: term %d is null", i);
    t.p[i] = terms[i];
    t.w[i] = weights[i];
  }
  hipLaunchKernelGGL(weighted_sum_fwd_kernel, dim3(1), dim3(1), 0, oi::as_stream(stream), t, out);
  return oi::check_launch("oi_weighted_sum_fwd");
}

int oi_weighted_sum_bwd(const float* g_out, const float* weights, int n, float* g_terms, oi_stream_t stream) {
  OI_REQUIRE(g_out != nullptr && weights != nullptr && g_terms != nullptr && n >= 1 && n <= 8, "oi_weighted_sum_bwd: n=%d (1..8)", n);
  WeightedTerms t{};
  t.n = n;
  for (int i = 0; i < n; ++i) t.w[i] = weights[i];
  hipLaunchKernelGGL(weighted_sum_bwd_kernel, dim3(1), dim3(64), 0, oi::as_stream(stream), t, g_out, g_terms);
  return oi::check_launch("oi_weighted_sum_bwd");
}

int oi_gan_losses_fwd(const float* d_real, const float* d_fake, const float* pose, const float* gx, const float* aux_w,
                      float reg_w, float* out6, int B, int K, long long N, oi_stream_t stream) {
  OI_REQUIRE(out6 != nullptr && B > 0 && K > 0 && N >= 0, "oi_gan_losses_fwd: B=%d K=%d N=%lld", B, K, N);
  OI_REQUIRE(pose == nullptr || (d_fake != nullptr && aux_w != nullptr), "oi_gan_losses_fwd: the pose term needs d_fake and aux_w");
  hipLaunchKernelGGL(gan_losses_fwd_kernel, dim3(1), dim3(1024), 0, oi::as_stream(stream), d_real, d_fake, pose,
                     N > 0 ? gx : nullptr, aux_w, reg_w, out6, B, K, N);
  return oi::check_launch("oi_gan_losses_fwd");
}

int oi_gan_losses_bwd(const float* g_total, const float* d_real, const float* d_fake, const float* pose, const float* gx,
                      const float* aux_w, float reg_w, float* g_real, float* g_fake, float* g_gx, int B, int K, long long N,
                      oi_stream_t stream) {
  OI_REQUIRE(g_total != nullptr && B > 0 && K > 0 && N >= 0, "oi_gan_losses_bwd: B=%d K=%d N=%lld", B, K, N);
  OI_REQUIRE((g_real == nullptr || d_real != nullptr) && (g_fake == nullptr || d_fake != nullptr) &&
                 (g_gx == nullptr || gx != nullptr),
             "oi_gan_losses_bwd: a gradient output without its operand");
  OI_REQUIRE(pose == nullptr || (aux_w != nullptr && d_fake != nullptr && K >= 2),
             "oi_gan_losses_bwd: a pose target needs its weight (aux_w), the fake logits and K >= 2 (K=%d)", K);
  if (g_gx == nullptr || N == 0) gx = nullptr;  // (the R1 term's gradient is not wanted)
  const long long work = std::max<long long>((long long)B * K, gx != nullptr ? (long long)B * N : 0);
  const int blocks = (int)std::min<long long>(1024, std::max<long long>(1, (work + 255) / 256));
  hipLaunchKernelGGL(gan_losses_bwd_kernel, dim3(blocks), dim3(256), 0, oi::as_stream(stream), g_total, d_real, d_fake, pose,
                     gx, aux_w, reg_w, g_real, g_fake, g_gx, B, K, N);
  return oi::check_launch("oi_gan_losses_bwd");
}

}  // extern "C"
