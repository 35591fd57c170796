// Backward of the fused NeuS compositing + Phong shading kernel (csrc/render.hip) for gfx950.
//
// Replaces what autograd derives in the reference for NeuSRenderer.render_core after the network calls
// (src/third_party/neus/models/renderer.py:266-311) and Generator.render_maps / lighting.diffuse / specular
// (src/models/generator.py:107-172, src/models/lighting.py:126-225): ~200 elementwise / scan backward
// kernels over (N, T[, 3]) tensors become one launch.  One wavefront per ray: pass 1 re-runs the forward
// (product scan) and emits every gradient that does not depend on later samples; pass 2 walks the
// samples backwards with a suffix-sum scan for d w_j / d alpha_i (j > i).
#include "oi_common.h"

namespace {

using oi::sigmoidf_;
using oi::wave_scan_mul;
using oi::wave_sum;

constexpr int RAYS_PER_BLOCK = 4;

__device__ __forceinline__ float ld(const float* p, long long i) { return p ? p[i] : 0.f; }

// inclusive suffix sum over the 64 lanes
__device__ __forceinline__ float wave_suffix_add(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_down(v, o, 64);
    if (lane + o < 64) v += t;
  }
  return v;
}

__global__ void __launch_bounds__(256)
composite_bwd_kernel(const oi_composite_params p, const oi_composite_grads q) {
  extern __shared__ float smem[];
  __shared__ float red[RAYS_PER_BLOCK][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long r = (long long)blockIdx.x * RAYS_PER_BLOCK + wave;
  const bool live = r < p.N;
  if (!live) r = p.N - 1;
  const int T = p.T;
  const int e = (int)(r / (p.N / p.B));
  float* s_aw = smem + wave * 3 * T;  // A_i * w_i
  float* s_at = s_aw + T;             // A_i * T_i
  float* s_om = s_at + T;             // 1 - alpha_i + 1e-7

  const float ox = p.rays_o[r * 3 + 0], oy = p.rays_o[r * 3 + 1], oz = p.rays_o[r * 3 + 2];
  const float dx = p.rays_d[r * 3 + 0], dy = p.rays_d[r * 3 + 1], dz = p.rays_d[r * 3 + 2];
  float lx = p.light_dir[e * 3 + 0], ly = p.light_dir[e * 3 + 1], lz = p.light_dir[e * 3 + 2];
  {
    const float n = fmaxf(sqrtf(lx * lx + ly * ly + lz * lz), 1e-6f);
    lx /= n; ly /= n; lz /= n;
  }
  const float inv_s_raw = expf(p.variance[0] * 10.0f);
  const float inv_s = fminf(fmaxf(inv_s_raw, 1e-6f), 1e6f);
  const bool inv_s_free = inv_s_raw > 1e-6f && inv_s_raw < 1e6f;
  const float car = p.cos_anneal_ratio;
  const float l_amb = sigmoidf_(p.light[0]), l_dif = 1.0f - l_amb, l_spec = fmaxf(p.light[1], 0.f), l_shin = p.light[2];

  // ---- upstream per-ray gradients
  float W = 0.f;
  {  // weight_sum is needed for the mask clamp: recompute it first (cheap: one product scan)
    float carry = 1.0f;
    for (int c0 = 0; c0 < T; c0 += 64) {
      const int i = c0 + lane;
      const bool on = i < T;
      const long long k = r * T + (on ? i : T - 1);
      const float sdf = p.sdf[k], dist = p.dists[k];
      const float tc = dx * p.grad[k * 3] + dy * p.grad[k * 3 + 1] + dz * p.grad[k * 3 + 2];
      const float ic = -(fmaxf(-tc * 0.5f + 0.5f, 0.f) * (1.0f - car) + fmaxf(-tc, 0.f) * car);
      const float P = sigmoidf_((sdf - ic * dist * 0.5f) * inv_s), Nn = sigmoidf_((sdf + ic * dist * 0.5f) * inv_s);
      float alpha = fminf(fmaxf((P - Nn + 1e-5f) / (P + 1e-5f), 0.f), 1.f);
      if (!on) alpha = 0.f;
      const float incl = wave_scan_mul(on ? 1.0f - alpha + 1e-7f : 1.0f, lane);
      float excl = __shfl_up(incl, 1, 64);
      if (lane == 0) excl = 1.0f;
      W += wave_sum(on ? alpha * (excl * carry) : 0.f);
      carry *= __shfl(incl, 63, 64);
    }
  }
  // d image: [N][3], or the [B][3][N / B] map layout the forward wrote (p.image_planar)
  const long long hw_ = p.image_planar ? p.N / p.B : 1, gi_base = p.image_planar ? (long long)e * 3 * hw_ + (r - (long long)e * hw_) : r * 3;
  const float gi0 = ld(q.g_image, gi_base), gi1 = ld(q.g_image, gi_base + hw_), gi2 = ld(q.g_image, gi_base + 2 * hw_);
  const float b0 = p.bg ? p.bg[e * 3] : 0.f, b1 = p.bg ? p.bg[e * 3 + 1] : 0.f, b2 = p.bg ? p.bg[e * 3 + 2] : 0.f;
  const float gW = ld(q.g_weight_sum, r) + ((W > 1e-3f && W < 1.0f - 1e-3f) ? ld(q.g_mask, r) : 0.f) -
                   (gi0 * b0 + gi1 * b1 + gi2 * b2);
  const float gc0 = ld(q.g_color_fine, r * 3), gc1 = ld(q.g_color_fine, r * 3 + 1), gc2 = ld(q.g_color_fine, r * 3 + 2);
  const float gn0 = ld(q.g_image_no_bg, r * 3) + gi0, gn1 = ld(q.g_image_no_bg, r * 3 + 1) + gi1,
              gn2 = ld(q.g_image_no_bg, r * 3 + 2) + gi2;
  const float gsh = ld(q.g_shading, r), gz = ld(q.g_z_map, r), gsp = ld(q.g_specular_map, r), gdf = ld(q.g_diffuse_map, r);
  const float gm0 = ld(q.g_normal, r * 3), gm1 = ld(q.g_normal, r * 3 + 1), gm2 = ld(q.g_normal, r * 3 + 2);
  const float g_eik = q.g_reduce4 ? q.g_reduce4[0] : 0.f, g_surf = q.g_reduce4 ? q.g_reduce4[2] : 0.f;

  float acc_amb = 0.f, acc_cd = 0.f, acc_cs = 0.f, acc_sh = 0.f, acc_l0 = 0.f, acc_l1 = 0.f, acc_l2 = 0.f, acc_invs = 0.f;

  // ---- pass 1: forward order
  float carry = 1.0f;
  for (int c0 = 0; c0 < T; c0 += 64) {
    const int i = c0 + lane;
    const bool on = i < T;
    const long long k = r * T + (on ? i : T - 1);
    const float sdf = p.sdf[k], dist = p.dists[k], mz = p.mid_z[k];
    const float gx = p.grad[k * 3], gy = p.grad[k * 3 + 1], gzz = p.grad[k * 3 + 2];
    const float c_r = p.rgb[k * 3], c_g = p.rgb[k * 3 + 1], c_b = p.rgb[k * 3 + 2];
    const float tc = dx * gx + dy * gy + dz * gzz;
    const float ic = -(fmaxf(-tc * 0.5f + 0.5f, 0.f) * (1.0f - car) + fmaxf(-tc, 0.f) * car);
    const float P = sigmoidf_((sdf - ic * dist * 0.5f) * inv_s), Nn = sigmoidf_((sdf + ic * dist * 0.5f) * inv_s);
    float alpha = fminf(fmaxf((P - Nn + 1e-5f) / (P + 1e-5f), 0.f), 1.f);
    if (!on) alpha = 0.f;
    const float om = on ? 1.0f - alpha + 1e-7f : 1.0f;
    const float incl = wave_scan_mul(om, lane);
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 1.0f;
    const float Ti = excl * carry;
    const float w = alpha * Ti;
    carry *= __shfl(incl, 63, 64);

    const float px = ox + dx * mz, py = oy + dy * mz, pz = oz + dz * mz;
    const float pn = sqrtf(px * px + py * py + pz * pz);
    const float gn = sqrtf(gx * gx + gy * gy + gzz * gzz);
    const float gnc = fmaxf(gn, 1e-6f);
    const float nx = gx / gnc, ny = gy / gnc, nz = gzz / gnc;
    const float ndl = nx * lx + ny * ly + nz * lz;
    const float rndl = fmaxf(ndl, 0.f);
    const float diff = l_dif * rndl;
    float vx = ox - px, vy = oy - py, vz = oz - pz;
    {
      const float n = fmaxf(sqrtf(vx * vx + vy * vy + vz * vz), 1e-6f);
      vx /= n; vy /= n; vz /= n;
    }
    const float rx = -lx + 2.0f * (ndl * nx), ry = -ly + 2.0f * (ndl * ny), rz = -lz + 2.0f * (ndl * nz);
    const float vr = vx * rx + vy * ry + vz * rz;
    const float al = fmaxf(vr, 0.f) * (ndl > 0.f ? 1.f : 0.f);
    const float al_h = powf(al, l_shin);
    const float spec = l_spec * al_h;
    const float shade = l_amb + diff;

    // A_i = dL/dw_i
    const float gwi = (q.g_weights && on) ? q.g_weights[k] : 0.f;
    const float A = gW + gc0 * c_r + gc1 * c_g + gc2 * c_b + gn0 * (shade * c_r + spec) + gn1 * (shade * c_g + spec) +
                    gn2 * (shade * c_b + spec) + gsh * shade + gm0 * gx + gm1 * gy + gm2 * gzz + gz * mz + gsp * spec +
                    gdf * diff + gwi;
    if (on) {
      s_aw[i] = A * w;
      s_at[i] = A * Ti;
      s_om[i] = om;
    }
    if (on) {
      // direct terms
      const float d_shade = w * (gn0 * c_r + gn1 * c_g + gn2 * c_b + gsh);
      const float d_spec = w * (gn0 + gn1 + gn2 + gsp);
      const float d_diff = d_shade + w * gdf;
      acc_amb += d_shade;
      acc_cd += d_diff * rndl;
      acc_cs += d_spec * al_h;
      float d_al = 0.f;
      if (al > 0.f) {
        acc_sh += d_spec * l_spec * al_h * logf(al);
        d_al = d_spec * l_spec * l_shin * powf(al, l_shin - 1.0f);
      }
      float d_ndl = d_diff * l_dif * (ndl > 0.f ? 1.f : 0.f);
      const float d_vr = d_al * ((vr > 0.f && ndl > 0.f) ? 1.f : 0.f);
      const float drx = d_vr * vx, dry = d_vr * vy, drz = d_vr * vz;  // d rho
      d_ndl += 2.0f * (drx * nx + dry * ny + drz * nz);
      float dnx = 2.0f * ndl * drx + d_ndl * lx, dny = 2.0f * ndl * dry + d_ndl * ly, dnz = 2.0f * ndl * drz + d_ndl * lz;
      acc_l0 += -drx + d_ndl * nx;
      acc_l1 += -dry + d_ndl * ny;
      acc_l2 += -drz + d_ndl * nz;
      // n = g / max(|g|, eps)
      float dgx, dgy, dgz;
      if (gn > 1e-6f) {
        const float dot = nx * dnx + ny * dny + nz * dnz;
        dgx = (dnx - nx * dot) / gn; dgy = (dny - ny * dot) / gn; dgz = (dnz - nz * dot) / gn;
      } else {
        dgx = dnx / 1e-6f; dgy = dny / 1e-6f; dgz = dnz / 1e-6f;
      }
      dgx += w * gm0; dgy += w * gm1; dgz += w * gm2;  // normal map (raw gradient)
      const float m = pn < 1.2f ? 1.f : 0.f;
      if (m > 0.f && gn > 0.f) {
        const float ce = g_eik * 2.0f * (gn - 1.0f) / gn;  // d/dg of (|g|-1)^2
        dgx += ce * gx; dgy += ce * gy; dgz += ce * gzz;
      }
      if (live) {
        q.d_grad[k * 3] = dgx; q.d_grad[k * 3 + 1] = dgy; q.d_grad[k * 3 + 2] = dgz;
        q.d_rgb[k * 3] = w * (gc0 + gn0 * shade);
        q.d_rgb[k * 3 + 1] = w * (gc1 + gn1 * shade);
        q.d_rgb[k * 3 + 2] = w * (gc2 + gn2 * shade);
      }
    }
  }
  __syncthreads();

  // ---- pass 2: reverse order, suffix sums
  float suffix = 0.f;
  const int nchunk = (T + 63) / 64;
  for (int cc = nchunk - 1; cc >= 0; --cc) {
    const int i = cc * 64 + lane;
    const bool on = i < T;
    const long long k = r * T + (on ? i : T - 1);
    const float aw = on ? s_aw[i] : 0.f;
    const float sfx_incl = wave_suffix_add(aw, lane);
    const float S = sfx_incl - aw + suffix;  // sum_{j>i} A_j w_j
    suffix += __shfl(sfx_incl, 0, 64);
    if (!on) continue;
    float d_alpha = s_at[i] - S / s_om[i];
    const float sdf = p.sdf[k], dist = p.dists[k];
    const float gx = p.grad[k * 3], gy = p.grad[k * 3 + 1], gzz = p.grad[k * 3 + 2];
    const float tc = dx * gx + dy * gy + dz * gzz;
    const float ic = -(fmaxf(-tc * 0.5f + 0.5f, 0.f) * (1.0f - car) + fmaxf(-tc, 0.f) * car);
    const float eh = ic * dist * 0.5f;
    const float P = sigmoidf_((sdf - eh) * inv_s), Nn = sigmoidf_((sdf + eh) * inv_s);
    const float raw = (P - Nn + 1e-5f) / (P + 1e-5f);
    if (!(raw > 0.f && raw < 1.f)) d_alpha = 0.f;  // clip(0, 1)
    const float pe = P + 1e-5f;
    const float d_P = d_alpha * (Nn / (pe * pe));
    const float d_N = -d_alpha / pe;
    const float d_xp = d_P * P * (1.0f - P), d_xn = d_N * Nn * (1.0f - Nn);
    float d_sdf = (d_xp + d_xn) * inv_s;
    const float d_e = (d_xn - d_xp) * inv_s;
    acc_invs += d_xp * (sdf - eh) + d_xn * (sdf + eh);
    const float d_ic = d_e * dist * 0.5f;
    const float d_tc = d_ic * (0.5f * (1.0f - car) * (tc < 1.0f ? 1.f : 0.f) + car * (tc < 0.f ? 1.f : 0.f));
    if (g_surf != 0.f) d_sdf += g_surf * (-100.0f) * (sdf > 0.f ? 1.f : (sdf < 0.f ? -1.f : 0.f)) * expf(-100.0f * fabsf(sdf));
    if (live) {
      q.d_sdf[k] = d_sdf;
      q.d_grad[k * 3] += d_tc * dx;
      q.d_grad[k * 3 + 1] += d_tc * dy;
      q.d_grad[k * 3 + 2] += d_tc * dz;
    }
  }

  // ---- global accumulations
  float v[8] = {acc_amb, acc_cd, acc_cs, acc_sh, acc_l0, acc_l1, acc_l2, acc_invs};
#pragma unroll
  for (int t = 0; t < 8; ++t) v[t] = wave_sum(v[t]);
  if (q.ray_partials != nullptr) {  // parked per ray, reduced by composite_bwd_reduce_kernel
    if (lane == 0 && live) {
      float4* dst = reinterpret_cast<float4*>(q.ray_partials + r * 8);
      dst[0] = make_float4(v[0], v[1], v[2], v[3]);
      dst[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
    return;
  }
  // no workspace: wave -> block -> atomics (every block hits the same seven addresses)
  if (lane == 0) {
#pragma unroll
    for (int t = 0; t < 8; ++t) red[wave][t] = live ? v[t] : 0.f;
  }
  __syncthreads();
  // rays of one block may straddle two elements only when N/B is not a multiple of 4: keep light_dir per wave
  if (lane == 0 && live && q.d_light_dir) {
    atomicAdd(q.d_light_dir + e * 3 + 0, v[4]);
    atomicAdd(q.d_light_dir + e * 3 + 1, v[5]);
    atomicAdd(q.d_light_dir + e * 3 + 2, v[6]);
  }
  if (threadIdx.x == 0) {
    float s[8] = {0};
    for (int wv = 0; wv < RAYS_PER_BLOCK; ++wv)
      for (int t = 0; t < 8; ++t) s[t] += red[wv][t];
    if (q.d_light) {
      // ambient = sigmoid(a); diffuse = 1 - ambient; specular = max(s, 0); shininess
      atomicAdd(q.d_light + 0, (s[0] - s[1]) * l_amb * (1.0f - l_amb));
      atomicAdd(q.d_light + 1, p.light[1] > 0.f ? s[2] : 0.f);
      atomicAdd(q.d_light + 2, s[3]);
    }
    if (q.d_variance && inv_s_free) atomicAdd(q.d_variance, s[7] * inv_s * 10.0f);
  }
}

// One block per batch element: sums the parked per-ray partials in a fixed order; d_light / d_variance (shared by all
// elements, zeroed by the caller) receive one atomic per element.
__global__ void __launch_bounds__(256)
composite_bwd_reduce_kernel(const oi_composite_params p, const oi_composite_grads q) {
  __shared__ float red[4][8];
  const int e = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long per = p.N / p.B, r0 = (long long)e * per;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long long i = threadIdx.x; i < per; i += blockDim.x) {
    const float4* src = reinterpret_cast<const float4*>(q.ray_partials + (r0 + i) * 8);
    const float4 a = src[0], b = src[1];
    s[0] += a.x; s[1] += a.y; s[2] += a.z; s[3] += a.w;
    s[4] += b.x; s[5] += b.y; s[6] += b.z; s[7] += b.w;
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) s[t] = wave_sum(s[t]);
  if (lane == 0) {
#pragma unroll
    for (int t = 0; t < 8; ++t) red[wave][t] = s[t];
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
#pragma unroll
  for (int t = 0; t < 8; ++t) s[t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
  if (q.d_light_dir) {
    q.d_light_dir[e * 3 + 0] = s[4];
    q.d_light_dir[e * 3 + 1] = s[5];
    q.d_light_dir[e * 3 + 2] = s[6];
  }
  if (q.d_light) {
    const float l_amb = sigmoidf_(p.light[0]);
    atomicAdd(q.d_light + 0, (s[0] - s[1]) * l_amb * (1.0f - l_amb));
    atomicAdd(q.d_light + 1, p.light[1] > 0.f ? s[2] : 0.f);
    atomicAdd(q.d_light + 2, s[3]);
  }
  const float inv_s_raw = expf(p.variance[0] * 10.0f);
  if (q.d_variance && inv_s_raw > 1e-6f && inv_s_raw < 1e6f) atomicAdd(q.d_variance, s[7] * inv_s_raw * 10.0f);
}

// ------------------------------------------------------------------------------------------
// Unit light direction in every box frame and its gradient, one launch each way:
//   u = d / |d|,  v_b = R_b u  (R_b = w2b[b][:3][:3]),  n_b = v_b / max(|v_b|, 1e-6)
// = DirectionalLight.direction (lighting.py:35-39) -> batch_direction (lighting.py:115-119) -> the F.normalize in front of the
// Phong terms.  As tensor ops with autograd that is 8 launches forward and ~18 backward of 3..9 elements each.
// ------------------------------------------------------------------------------------------
__global__ void light_dir_fwd_kernel(const float* __restrict__ d, const float* __restrict__ w2b, float* __restrict__ n, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float inv = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  const float u[3] = {d[0] * inv, d[1] * inv, d[2] * inv};
  const float* R = w2b + (size_t)b * 16;
  float v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) v[i] = R[4 * i] * u[0] + R[4 * i + 1] * u[1] + R[4 * i + 2] * u[2];
  const float nv = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-6f);
#pragma unroll
  for (int i = 0; i < 3; ++i) n[b * 3 + i] = v[i] / nv;
}

__global__ void __launch_bounds__(64)
light_dir_bwd_kernel(const float* __restrict__ d, const float* __restrict__ w2b, const float* __restrict__ g_n,
                     float* __restrict__ g_d, int B) {
  const float nd = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  const float u[3] = {d[0] / nd, d[1] / nd, d[2] / nd};
  float gu[3] = {0.f, 0.f, 0.f};  // sum_b R_b^T (I - n n^T) g_n / |v|
  for (int b = threadIdx.x; b < B; b += 64) {
    const float* R = w2b + (size_t)b * 16;
    float v[3], g[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      v[i] = R[4 * i] * u[0] + R[4 * i + 1] * u[1] + R[4 * i + 2] * u[2];
      g[i] = g_n[b * 3 + i];
    }
    const float raw = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    float gv[3];
    if (raw > 1e-6f) {
      const float nn[3] = {v[0] / raw, v[1] / raw, v[2] / raw};
      const float dot = nn[0] * g[0] + nn[1] * g[1] + nn[2] * g[2];
#pragma unroll
      for (int i = 0; i < 3; ++i) gv[i] = (g[i] - nn[i] * dot) / raw;
    } else {  // the clamp is active: n = v / eps
#pragma unroll
      for (int i = 0; i < 3; ++i) gv[i] = g[i] * 1e6f;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) gu[j] += R[j] * gv[0] + R[4 + j] * gv[1] + R[8 + j] * gv[2];
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) gu[j] = oi::wave_sum(gu[j]);
  if (threadIdx.x == 0) {
    const float dot = u[0] * gu[0] + u[1] * gu[1] + u[2] * gu[2];
#pragma unroll
    for (int j = 0; j < 3; ++j) g_d[j] = (gu[j] - u[j] * dot) / nd;
  }
}

}  // namespace

extern "C" int oi_light_dir_fwd(const float* d, const float* w2b, float* n, int B, oi_stream_t stream) {
  OI_REQUIRE(d && w2b && n && B > 0, "oi_light_dir_fwd: bad argument");
  hipLaunchKernelGGL(light_dir_fwd_kernel, dim3(oi::cdiv(B, 64)), dim3(64), 0, oi::as_stream(stream), d, w2b, n, B);
  return oi::check_launch("oi_light_dir_fwd");
}

extern "C" int oi_light_dir_bwd(const float* d, const float* w2b, const float* g_n, float* g_d, int B, oi_stream_t stream) {
  OI_REQUIRE(d && w2b && g_n && g_d && B > 0, "oi_light_dir_bwd: bad argument");
  hipLaunchKernelGGL(light_dir_bwd_kernel, dim3(1), dim3(64), 0, oi::as_stream(stream), d, w2b, g_n, g_d, B);
  return oi::check_launch("oi_light_dir_bwd");
}

extern "C" int oi_composite_bwd(const oi_composite_params* p, const oi_composite_grads* g, oi_stream_t stream) {
  OI_REQUIRE(p && g, "oi_composite_bwd: null params");
  OI_REQUIRE(p->sdf && p->grad && p->rgb && p->dists && p->mid_z && p->rays_o && p->rays_d && p->light_dir &&
                 p->variance && p->light,
             "oi_composite_bwd: null input pointer");
  OI_REQUIRE(g->d_sdf && g->d_grad && g->d_rgb, "oi_composite_bwd: null output pointer");
  OI_REQUIRE(p->N > 0 && p->T > 0 && p->B > 0 && p->N % p->B == 0, "oi_composite_bwd: N=%lld T=%d B=%d", p->N, p->T,
             p->B);
  const size_t sh = (size_t)RAYS_PER_BLOCK * 3 * p->T * sizeof(float);
  OI_REQUIRE(sh <= 60 * 1024, "oi_composite_bwd: T=%d too large", p->T);
  hipLaunchKernelGGL(composite_bwd_kernel, dim3(oi::cdiv(p->N, RAYS_PER_BLOCK)), dim3(256), sh, oi::as_stream(stream),
                     *p, *g);
  int rc = oi::check_launch("oi_composite_bwd");
  if (rc != OI_OK || g->ray_partials == nullptr) return rc;
  hipLaunchKernelGGL(composite_bwd_reduce_kernel, dim3(p->B), dim3(256), 0, oi::as_stream(stream), *p, *g);
  return oi::check_launch("oi_composite_bwd(reduce)");
}
