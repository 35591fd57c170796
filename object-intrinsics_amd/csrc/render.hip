// Ray generation, hierarchical importance resampling and NeuS alpha compositing for gfx950.
//
// Replaces: the per-ray tensor programs of
//   NeuSRenderer.render / up_sample / sample_pdf / cat_z_vals / render_core
//     (reference src/third_party/neus/models/renderer.py:44-74, 137-197, 199-349, 351-473)
//   Generator.gen_rays_at / build_rays / near_far_from_sphere / render_maps
//     (src/models/generator.py:80-174, 255-279, 317-342) and lighting.diffuse / specular
//     (src/models/lighting.py:126-225)
// which the reference runs as ~150 separate elementwise/scan/sort/gather kernels.
//
// These are HBM-streaming kernels: one 64-lane wavefront owns one ray, samples are strided over
// lanes (coalesced), the transmittance product and the CDF are wavefront shuffle scans, the
// searchsorted / sort of the reference become binary searches over LDS-resident per-ray arrays.
#include "oi_common.h"
#include "f3_blob.h"

namespace {

using oi::sigmoidf_;
using oi::wave_max;
using oi::wave_scan_add;
using oi::wave_scan_mul;
using oi::wave_sum;

constexpr int RAYS_PER_BLOCK = 4;  // 256 threads
constexpr int MAX_SC = 1024;       // max samples per ray held in LDS by the resampling kernels

// torch.linspace(start, end, n)[i] in fp32: start + i*step below the midpoint, end - (n-1-i)*step above.
__device__ __forceinline__ float linspace_at(float start, float end, int n, int i) {
  if (n == 1) return start;
  const float step = (end - start) / (float)(n - 1);
  return i < n / 2 ? start + step * (float)i : end - step * (float)(n - 1 - i);
}

// Ray set-up arithmetic shared by the stand-alone kernels and the fused prep_render_kernel.  Floating-point contraction is OFF
// inside these helpers: left to the compiler, a*b + c becomes an fma in one kernel and a multiply + add in another (it
// depends on what surrounds the expression after inlining), and "the same expressions" would differ in the last bit.
// -DOI_RAY_CONTRACT_DEFAULT: the compiler's default contraction inside the helpers (debug A/B only)
#ifdef OI_RAY_CONTRACT_DEFAULT
#define OI_RAY_FP_CONTRACT
#else
#define OI_RAY_FP_CONTRACT _Pragma("clang fp contract(off)")
#endif
struct RayOD {
  float o[3], d[3], near_, far_;
};
__device__ __forceinline__ RayOD make_ray(const float* __restrict__ M /* c2b 4x4 */, const float* __restrict__ kinv, float offx,
                                          float offy, int R, int x, int y) {
OI_RAY_FP_CONTRACT
  RayOD r;
  // build_rays: pixels = linspace(0,1,R) * recp_size + offset   (generator.py:325-329)
  const float px = linspace_at(0.f, 1.f, R, x) * (float)R + offx;
  const float py = linspace_at(0.f, 1.f, R, y) * (float)R + offy;
  float p[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) p[i] = kinv[i * 3 + 0] * px + kinv[i * 3 + 1] * py + kinv[i * 3 + 2];
  const float nrm = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
  p[0] /= nrm;
  p[1] /= nrm;
  p[2] /= nrm;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    r.d[i] = M[i * 4 + 0] * p[0] + M[i * 4 + 1] * p[1] + M[i * 4 + 2] * p[2];
    r.o[i] = M[i * 4 + 3];
  }
  // near_far_from_sphere (generator.py:336-342)
  const float a = r.d[0] * r.d[0] + r.d[1] * r.d[1] + r.d[2] * r.d[2];
  const float bb = 2.0f * (r.o[0] * r.d[0] + r.o[1] * r.d[1] + r.o[2] * r.d[2]);
  const float mid = 0.5f * (-bb) / a;
  r.near_ = mid - 1.0f;
  r.far_ = mid + 1.0f;
  return r;
}
// coarse sample i of S on [near, far] (+ the per-ray jitter): renderer.py:359-360, 372-373
__device__ __forceinline__ float coarse_z_at(float nr, float fr, int S, int i, const float* __restrict__ jitter, long long r,
                                             bool jitter_normal = false) {
OI_RAY_FP_CONTRACT
  float zv = nr + (fr - nr) * linspace_at(0.f, 1.f, S, i);
  if (jitter != nullptr) {
    float u = jitter[r];
    // a standard normal draw (the caller's ONE generator call for latents + jitter) -> its CDF: uniform on (0, 1), kept below 1
    // like torch.rand's [0, 1)
    if (jitter_normal) u = fminf(0.5f * erfcf(-0.70710678118654752f * u), 0.99999994f);
    zv = zv + (u - 0.5f) * 2.0f / (float)S;
  }
  return zv;
}
__device__ __forceinline__ float along_ray(float o, float d, float z) {
OI_RAY_FP_CONTRACT
  return o + d * z;
}

// ------------------------------------------------------------------------------------------
// a13 + a14: rays of the object crop
// ------------------------------------------------------------------------------------------
__global__ void gen_rays_kernel(const float* __restrict__ c2b, const float* __restrict__ kinv,
                                const float* __restrict__ offs, int B, int R, float* __restrict__ rays_o,
                                float* __restrict__ rays_d, float* __restrict__ near_, float* __restrict__ far_,
                                const float* __restrict__ w2b, const float* __restrict__ light,
                                float* __restrict__ light_dir) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = (long long)B * R * R;
  if (idx >= n) return;
  const int x = idx % R, y = (idx / R) % R, b = idx / ((long long)R * R);
  if (light_dir != nullptr && x == 0 && y == 0) {
    // light direction in this box frame: w2b[:3,:3] (d / |d|)   (lighting.py:62-64, 115-119)
    const float l0 = light[0], l1 = light[1], l2 = light[2];
    const float ln = sqrtf(l0 * l0 + l1 * l1 + l2 * l2);
    const float* Wb = w2b + b * 16;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      light_dir[b * 3 + i] = Wb[i * 4 + 0] * (l0 / ln) + Wb[i * 4 + 1] * (l1 / ln) + Wb[i * 4 + 2] * (l2 / ln);
  }
  const RayOD ry = make_ray(c2b + b * 16, kinv, offs[b * 2 + 0], offs[b * 2 + 1], R, x, y);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    rays_d[idx * 3 + i] = ry.d[i];
    rays_o[idx * 3 + i] = ry.o[i];
  }
  near_[idx] = ry.near_;
  far_[idx] = ry.far_;
}

// ------------------------------------------------------------------------------------------
// Everything a render needs before its first MLP pass, ONE launch (round 4): the pose block arrives BY VALUE in the kernel
// arguments (no host-to-device copy on the stream: 4.5 us each on this part), the first workgroups evaluate the style MLP +
// the FiLM parameters of all layers (a1, a2: film_params_kernel's work), the others generate the crop's rays (a13, a14), their
// near / far and the coarse samples with their points (a8).  Replaces four dependent launches (copy, oi_gen_rays_light,
// oi_film_params, oi_coarse_samples) of ~5-10 us each; every value is computed by the same expressions as in those kernels.
// ------------------------------------------------------------------------------------------
constexpr int PREP_RAYS = 16;  // rays per ray workgroup (256 threads: 16 rays x S samples written coalesced)

__device__ __forceinline__ void style_film_block(const float* __restrict__ style_w, const float* __restrict__ style_b,
                                                 const float* __restrict__ z, float* __restrict__ w_out,
                                                 const float* __restrict__ gw, const float* __restrict__ gb,
                                                 const float* __restrict__ bw, const float* __restrict__ bb,
                                                 float* __restrict__ gamma, float* __restrict__ beta, int NL, int e, int l,
                                                 float (*h)[64], const float* __restrict__ f3_hdr = nullptr,
                                                 float* __restrict__ f3_blob = nullptr, float* f3_red = nullptr) {
  // (csrc/mlp.hip film_params_kernel, same operations in the same order; threads >= 128 only keep the barriers)
  const int t = threadIdx.x;
  if (z != nullptr) {
    if (t < 64) h[0][t] = z[e * 64 + t];
    __syncthreads();
    int cur = 0;
    for (int ly = 0; ly < 3; ++ly) {
      if (t < 64) {
        const float* wr = style_w + (ly * 64 + t) * 64;
        float acc = 0.f;
#pragma unroll 8
        for (int k = 0; k < 64; ++k) acc = fmaf(h[cur][k], wr[k], acc);
        acc += style_b[ly * 64 + t];
        h[cur ^ 1][t] = acc > 0.f ? acc : 0.2f * acc;
      }
      __syncthreads();
      cur ^= 1;
    }
    if (t < 64 && l == 0) w_out[e * 64 + t] = h[cur][t];
    if (cur != 0) {
      if (t < 64) h[0][t] = h[1][t];
    }
    __syncthreads();
  } else {
    if (t < 64) h[0][t] = w_out[e * 64 + t];
    __syncthreads();
  }
  float gm = 0.f, bt = 0.f;
  if (l < NL && t < 128) {
    const float* g = gw + ((size_t)l * 128 + t) * 64;
    const float* b = bw + ((size_t)l * 128 + t) * 64;
    float ag = 0.f, ab = 0.f;
#pragma unroll 8
    for (int k = 0; k < 64; ++k) {
      ag = fmaf(h[0][k], g[k], ag);
      ab = fmaf(h[0][k], b[k], ab);
    }
    gm = 15.0f * (ag + gb[l * 128 + t]) + 30.0f;
    bt = 0.25f * (ab + bb[l * 128 + t]) + 0.0f;
    gamma[((size_t)e * NL + l) * 128 + t] = gm;
    beta[((size_t)e * NL + l) * 128 + t] = bt;
  }
  // the per-element blob of the register-resident MLP kernel, this workgroup's layer of it (csrc/f3_blob.h): gamma_l / beta_l are
  // in registers here -- film_blob_f3_kernel would read them back from memory in a launch of its own
  if (f3_blob != nullptr && l < NL) oif3::blob_layer(f3_hdr, f3_blob + (size_t)e * (oif3::F3_BLOB / 4), l, t, gm, bt, f3_red);
}

__global__ void __launch_bounds__(256) prep_render_kernel(const oi_prep_params p) {
  __shared__ float h[2][64];
  __shared__ float ray[PREP_RAYS][8];  // o xyz, d xyz, near, far
  const int nl1 = p.NL > 0 ? p.NL : 1;
  const int n_film = p.B * nl1;
  if ((int)blockIdx.x < n_film) {
    __shared__ float f3_red[8];
    style_film_block(p.style_w, p.style_b, p.z, p.w_out, p.gw, p.gb, p.bw, p.bb, p.gamma, p.beta, p.NL, blockIdx.x / nl1,
                     blockIdx.x % nl1, h, reinterpret_cast<const float*>(p.f3_packed), reinterpret_cast<float*>(p.f3_blob), f3_red);
    return;
  }
  const int rb = blockIdx.x - n_film, t = threadIdx.x, R = p.R;
  const long long n = (long long)p.B * R * R;
  if (rb == 0) {  // the pose block for later consumers (pose loss, compositing): b2w [B][16] | w2b | c2b | offs [B][2] | bg [B][3]
    const int B = p.B;
    for (int i = t; i < B * 53; i += 256) {
      float v;
      if (i < 16 * B) v = p.b2w[i / 16][i % 16];
      else if (i < 32 * B) v = p.w2b[(i - 16 * B) / 16][(i - 16 * B) % 16];
      else if (i < 48 * B) v = p.c2b[(i - 32 * B) / 16][(i - 32 * B) % 16];
      else if (i < 50 * B) v = p.offs[(i - 48 * B) / 2][(i - 48 * B) % 2];
      else v = p.bg[(i - 50 * B) / 3][(i - 50 * B) % 3];
      p.pose_out[i] = v;
    }
    if (p.light_dir != nullptr && t < p.B) {  // light direction in each box frame: w2b[:3,:3] (d / |d|)   (lighting.py:62-64, 115-119)
      const float l0 = p.light_direction[0], l1 = p.light_direction[1], l2 = p.light_direction[2];
      const float ln = sqrtf(l0 * l0 + l1 * l1 + l2 * l2);
      const float* Wb = p.w2b[t];
#pragma unroll
      for (int i = 0; i < 3; ++i)
        p.light_dir[t * 3 + i] = Wb[i * 4 + 0] * (l0 / ln) + Wb[i * 4 + 1] * (l1 / ln) + Wb[i * 4 + 2] * (l2 / ln);
    }
  }
  const long long r0 = (long long)rb * PREP_RAYS;
  if (t < PREP_RAYS && r0 + t < n) {  // gen_rays_kernel's expressions
    const long long idx = r0 + t;
    const int x = idx % R, y = (idx / R) % R, b = idx / ((long long)R * R);
    const RayOD ry = make_ray(p.c2b[b], p.kinv, p.offs[b][0], p.offs[b][1], R, x, y);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      p.rays_d[idx * 3 + i] = ry.d[i];
      p.rays_o[idx * 3 + i] = ry.o[i];
      ray[t][i] = ry.o[i];
      ray[t][3 + i] = ry.d[i];
    }
    p.near_[idx] = ray[t][6] = ry.near_;
    p.far_[idx] = ray[t][7] = ry.far_;
  }
  __syncthreads();
  // coarse_samples_kernel's expressions, 16 rays x S samples, consecutive threads on consecutive samples
  const int S = p.S;
  for (int i = t; i < PREP_RAYS * S; i += 256) {
    const int lr = i / S, k = i % S;
    const long long r = r0 + lr;
    if (r >= n) break;
    const float zv = coarse_z_at(ray[lr][6], ray[lr][7], S, k, p.jitter, r, p.jitter_normal != 0);
    const long long o_ = r * S + k;
    p.z_coarse[o_] = zv;
#pragma unroll
    for (int c = 0; c < 3; ++c) p.pts_coarse[o_ * 3 + c] = along_ray(ray[lr][c], ray[lr][3 + c], zv);
  }
}

// ------------------------------------------------------------------------------------------
// a8: coarse samples + their points
// ------------------------------------------------------------------------------------------
__global__ void coarse_samples_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                      const float* __restrict__ near_, const float* __restrict__ far_,
                                      const float* __restrict__ jitter, long long N, int S,
                                      float* __restrict__ z, float* __restrict__ pts) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * S) return;
  const long long r = idx / S;
  const int i = idx % S;
  const float zv = coarse_z_at(near_[r], far_[r], S, i, jitter, r);
  z[idx] = zv;
#pragma unroll
  for (int k = 0; k < 3; ++k) pts[idx * 3 + k] = along_ray(rays_o[r * 3 + k], rays_d[r * 3 + k], zv);
}

// ------------------------------------------------------------------------------------------
// section mid-points of the merged z
// ------------------------------------------------------------------------------------------
__global__ void midpoints_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                 const float* __restrict__ z, long long N, int T, float last_dist,
                                 float* __restrict__ dists, float* __restrict__ mid_z, float* __restrict__ pts) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * T) return;
  const long long r = idx / T;
  const int i = idx % T;
  const float zi = z[idx];
  const float d = i + 1 < T ? z[idx + 1] - zi : last_dist;  // renderer.py:219-225
  const float m = zi + d * 0.5f;
  dists[idx] = d;
  mid_z[idx] = m;
#pragma unroll
  for (int k = 0; k < 3; ++k) pts[idx * 3 + k] = rays_o[r * 3 + k] + rays_d[r * 3 + k] * m;
}

// ------------------------------------------------------------------------------------------
// a9 + a10 (+ a11): importance resampling, one wavefront per ray
// ------------------------------------------------------------------------------------------
// number of elements of ascending a[0..n) that are <= v  (searchsorted right=True)
__device__ __forceinline__ int count_le(const float* a, int n, float v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] <= v) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// number of elements strictly < v
__device__ __forceinline__ int count_lt(const float* a, int n, float v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// alpha of section i of (zs, ss): renderer.py:143-173
__device__ __forceinline__ float upsample_alpha(const float* zs, const float* ss, int i, float ox, float oy,
                                                float oz, float dx, float dy, float dz, float inv_s) {
  const float z0 = zs[i], z1 = zs[i + 1], s0 = ss[i], s1 = ss[i + 1];
  const float ax = ox + dx * z0, ay = oy + dy * z0, az = oz + dz * z0;
  const float bx = ox + dx * z1, by = oy + dy * z1, bz = oz + dz * z1;
  const float r0 = sqrtf(ax * ax + ay * ay + az * az), r1 = sqrtf(bx * bx + by * by + bz * bz);
  const float inside = (r0 < 1.0f || r1 < 1.0f) ? 1.0f : 0.0f;
  const float cosv = (s1 - s0) / (z1 - z0 + 1e-5f);
  float prev_cos = 0.0f;
  if (i > 0) prev_cos = (s0 - ss[i - 1]) / (z0 - zs[i - 1] + 1e-5f);
  float c = fminf(prev_cos, cosv);
  c = fminf(fmaxf(c, -1e3f), 0.0f) * inside;
  const float dist = z1 - z0;
  const float mid = (s0 + s1) * 0.5f;
  const float prev_cdf = sigmoidf_((mid - c * dist * 0.5f) * inv_s);
  const float next_cdf = sigmoidf_((mid + c * dist * 0.5f) * inv_s);
  return (prev_cdf - next_cdf + 1e-5f) / (prev_cdf + 1e-5f);
}

__global__ void __launch_bounds__(256)
upsample_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ z,
                const float* __restrict__ sdf, long long N, int Sc, int n_new, float inv_s,
                float* __restrict__ z_new, float* __restrict__ pts_new, float* __restrict__ z_merged,
                float last_dist, float* __restrict__ mid_dists, float* __restrict__ mid_z, float* __restrict__ mid_pts) {
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long r = (long long)blockIdx.x * RAYS_PER_BLOCK + wave;
  const bool live = r < N;
  if (!live) r = N - 1;  // keep every wave in the block-wide barriers; stores are masked
  float* zs = smem + wave * (3 * Sc + n_new);
  float* ss = zs + Sc;
  float* cdf = ss + Sc;  // Sc entries: cdf[0] = 0, cdf[i+1] = cumsum(pdf)[i]
  float* zn = cdf + Sc;  // n_new

  const float ox = rays_o[r * 3 + 0], oy = rays_o[r * 3 + 1], oz = rays_o[r * 3 + 2];
  const float dx = rays_d[r * 3 + 0], dy = rays_d[r * 3 + 1], dz = rays_d[r * 3 + 2];
  for (int i = lane; i < Sc; i += 64) {
    zs[i] = z[r * Sc + i];
    ss[i] = sdf[r * Sc + i];
  }
  __syncthreads();

  // weights = alpha * exclusive-product(1 - alpha + 1e-7), then + 1e-5  (renderer.py:174-175, 47)
  const int nsec = Sc - 1;
  float carry = 1.0f, total = 0.0f;
  for (int c0 = 0; c0 < nsec; c0 += 64) {
    const int i = c0 + lane;
    const bool on = i < nsec;
    const float alpha = on ? upsample_alpha(zs, ss, i, ox, oy, oz, dx, dy, dz, inv_s) : 0.0f;
    const float incl = wave_scan_mul(on ? 1.0f - alpha + 1e-7f : 1.0f, lane);
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 1.0f;
    const float w5 = alpha * (excl * carry) + 1e-5f;
    if (on) cdf[i + 1] = w5;
    total += wave_sum(on ? w5 : 0.0f);
    carry *= __shfl(incl, 63, 64);
  }
  __syncthreads();
  float run = 0.0f;
  for (int c0 = 0; c0 < nsec; c0 += 64) {
    const int i = c0 + lane;
    const bool on = i < nsec;
    const float pdf = on ? cdf[i + 1] / total : 0.0f;
    const float incl = wave_scan_add(pdf, lane);
    if (on) cdf[i + 1] = incl + run;
    run += __shfl(incl, 63, 64);
  }
  if (lane == 0) cdf[0] = 0.0f;
  __syncthreads();

  // inverse CDF at u_j = linspace(0.5/n, 1 - 0.5/n, n)   (renderer.py:52-72)
  float zmaxrun = -3.0e38f;
  for (int c0 = 0; c0 < n_new; c0 += 64) {
    const int jn = c0 + lane;
    const bool on = jn < n_new;
    float s = -3.0e38f;
    if (on) {
      const float u = linspace_at(0.5f / (float)n_new, 1.0f - 0.5f / (float)n_new, n_new, jn);
      const int ind = count_le(cdf, Sc, u);
      const int below = max(ind - 1, 0), above = min(ind, Sc - 1);
      const float cb = cdf[below], ca = cdf[above], zb = zs[below], za = zs[above];
      float den = ca - cb;
      if (den < 1e-5f) den = 1.0f;
      s = zb + (u - cb) / den * (za - zb);
    }
    // keep the list non-decreasing (the reference sorts afterwards; this only matters at the ulp level)
    float m = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float t = __shfl_up(m, o, 64);
      if (lane >= o) m = fmaxf(m, t);
    }
    m = fmaxf(m, zmaxrun);
    zmaxrun = __shfl(m, 63, 64);
    if (on) {
      zn[jn] = m;
      if (live) {
        z_new[r * n_new + jn] = m;
        pts_new[(r * n_new + jn) * 3 + 0] = ox + dx * m;
        pts_new[(r * n_new + jn) * 3 + 1] = oy + dy * m;
        pts_new[(r * n_new + jn) * 3 + 2] = oz + dz * m;
      }
    }
  }
  if (z_merged == nullptr) return;
  __syncthreads();
  // a11: rank merge of the two ascending lists (== cat + sort, renderer.py:187-188)
  const int Tm = Sc + n_new;
  if (mid_dists == nullptr) {
    if (live) {
      for (int i = lane; i < Sc; i += 64) z_merged[r * Tm + i + count_lt(zn, n_new, zs[i])] = zs[i];
      for (int jn = lane; jn < n_new; jn += 64) z_merged[r * Tm + jn + count_le(zs, Sc, zn[jn])] = zn[jn];
    }
    return;
  }
  // ... and, on the last step, the section mid-points of the merged list in the same launch (midpoints_kernel: renderer.py
  // :219-235): the merged list is formed in LDS (zm: Tm floats behind the four arrays), every lane then has its neighbour
  float* zm = smem + RAYS_PER_BLOCK * (3 * Sc + n_new) + wave * Tm;
  for (int i = lane; i < Sc; i += 64) zm[i + count_lt(zn, n_new, zs[i])] = zs[i];
  for (int jn = lane; jn < n_new; jn += 64) zm[jn + count_le(zs, Sc, zn[jn])] = zn[jn];
  __syncthreads();
  if (!live) return;
  for (int i = lane; i < Tm; i += 64) {
    const float zi = zm[i];
    const float d = i + 1 < Tm ? zm[i + 1] - zi : last_dist;
    const float m = zi + d * 0.5f;
    const long long k = r * Tm + i;
    z_merged[k] = zi;
    mid_dists[k] = d;
    mid_z[k] = m;
    mid_pts[k * 3 + 0] = ox + dx * m;
    mid_pts[k * 3 + 1] = oy + dy * m;
    mid_pts[k * 3 + 2] = oz + dz * m;
  }
}

__global__ void __launch_bounds__(256)
merge_sorted_kernel(const float* __restrict__ z, const float* __restrict__ sdf, const float* __restrict__ z_new,
                    const float* __restrict__ sdf_new, long long N, int Sc, int n_new, float* __restrict__ z_out,
                    float* __restrict__ sdf_out) {
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long r = (long long)blockIdx.x * RAYS_PER_BLOCK + wave;
  const bool live = r < N;
  if (!live) r = N - 1;
  float* zs = smem + wave * (Sc + n_new);
  float* zn = zs + Sc;
  for (int i = lane; i < Sc; i += 64) zs[i] = z[r * Sc + i];
  for (int i = lane; i < n_new; i += 64) zn[i] = z_new[r * n_new + i];
  __syncthreads();
  const int Tm = Sc + n_new;
  if (!live) return;
  for (int i = lane; i < Sc; i += 64) {
    const int pos = i + count_lt(zn, n_new, zs[i]);
    z_out[r * Tm + pos] = zs[i];
    sdf_out[r * Tm + pos] = sdf[r * Sc + i];
  }
  for (int jn = lane; jn < n_new; jn += 64) {
    const int pos = jn + count_le(zs, Sc, zn[jn]);
    z_out[r * Tm + pos] = zn[jn];
    sdf_out[r * Tm + pos] = sdf_new[r * n_new + jn];
  }
}

// ------------------------------------------------------------------------------------------
// a12 (tail) + a15: alpha compositing + Phong shading + maps, one wavefront per ray
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void normalize3(float& x, float& y, float& z, float eps) {
  // F.normalize(v, eps): v / max(|v|, eps)
  const float n = fmaxf(sqrtf(x * x + y * y + z * z), eps);
  x /= n;
  y /= n;
  z /= n;
}

// Sums the per-block partials of composite_fwd_kernel in a fixed order (256 threads: block-strided partial sums, wave sums,
// four waves) and derives the scalars the reference gets from ~8 tiny tensor launches:  out[0..3] = reduce4 totals,
// out[4..7] = ray sums, out[8] = gradient_error = out[0] / (out[1] + 1e-5), out[9] = surface_loss = out[2] / (N T),
// out[10..12] = means over rays of cdf[:,0], weight_max, weight_sum.  AGENT: the partials were written by other workgroups
// of the SAME launch (agent-scope loads bypass this CU's L1).
template <bool AGENT>
__device__ __forceinline__ void stats_reduce(const float* __restrict__ block_partials, int n_blocks, float n_rays,
                                             float n_samples, float* __restrict__ out) {
  __shared__ float sred[4][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if constexpr (AGENT) {
    // four rows per trip, all 24 loads issued before the first add (one at a time they are a chain of L2 round trips: 10 us
    // for the 1024 rows of a C2 launch); the adds keep the row order of the plain loop below: same bits
    for (int i0 = threadIdx.x; i0 < n_blocks; i0 += 4 * blockDim.x) {
      float v[4][6];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * blockDim.x;
        const float* src = block_partials + (size_t)(i < n_blocks ? i : i0) * 8;
#pragma unroll
        for (int t = 0; t < 6; ++t) v[u][t] = __hip_atomic_load(src + t + (t >= 3 ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (i0 + u * (int)blockDim.x < n_blocks) {
#pragma unroll
          for (int t = 0; t < 6; ++t) s[t + (t >= 3 ? 1 : 0)] += v[u][t];
        }
      }
    }
  } else {
    for (int i = threadIdx.x; i < n_blocks; i += blockDim.x) {
      const float4* src = reinterpret_cast<const float4*>(block_partials + (size_t)i * 8);
      const float4 a = src[0], b = src[1];
      s[0] += a.x; s[1] += a.y; s[2] += a.z;
      s[4] += b.x; s[5] += b.y; s[6] += b.z;
    }
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) s[t] = oi::wave_sum(s[t]);
  if (lane == 0) {
#pragma unroll
    for (int t = 0; t < 8; ++t) sred[wave][t] = s[t];
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
#pragma unroll
  for (int t = 0; t < 8; ++t) out[t] = s[t] = sred[0][t] + sred[1][t] + sred[2][t] + sred[3][t];
  out[8] = s[0] / (s[1] + 1e-5f);
  out[9] = s[2] / n_samples;
  out[10] = s[4] / n_rays;
  out[11] = s[5] / n_rays;
  out[12] = s[6] / n_rays;
  out[13] = out[14] = out[15] = 0.f;
}

__global__ void __launch_bounds__(256)
composite_fwd_kernel(const oi_composite_params p) {
  __shared__ float red[RAYS_PER_BLOCK][6];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long r = (long long)blockIdx.x * RAYS_PER_BLOCK + wave;
  const bool live = r < p.N;
  if (!live) r = p.N - 1;
  const int T = p.T;
  const int e = (int)(r / (p.N / p.B));

  const float ox = p.rays_o[r * 3 + 0], oy = p.rays_o[r * 3 + 1], oz = p.rays_o[r * 3 + 2];
  const float dx = p.rays_d[r * 3 + 0], dy = p.rays_d[r * 3 + 1], dz = p.rays_d[r * 3 + 2];
  float lx = p.light_dir[e * 3 + 0], ly = p.light_dir[e * 3 + 1], lz = p.light_dir[e * 3 + 2];
  normalize3(lx, ly, lz, 1e-6f);
  // SingleVarianceNetwork + clip (neus/models/fields.py:267-268; renderer.py:266)
  const float inv_s = fminf(fmaxf(expf(p.variance[0] * 10.0f), 1e-6f), 1e6f);
  const float car = p.cos_anneal_ratio;
  const float l_amb = sigmoidf_(p.light[0]), l_dif = 1.0f - l_amb, l_spec = fmaxf(p.light[1], 0.f), l_shin = p.light[2];

  float carry = 1.0f;
  float a_wsum = 0.f, a_wmax = 0.f, a_c0 = 0.f, a_c1 = 0.f, a_c2 = 0.f, a_i0 = 0.f, a_i1 = 0.f, a_i2 = 0.f;
  float a_sh = 0.f, a_n0 = 0.f, a_n1 = 0.f, a_n2 = 0.f, a_z = 0.f, a_sp = 0.f, a_df = 0.f;
  float a_eik = 0.f, a_m = 0.f, a_surf = 0.f, cdf_first = 0.f;

  for (int c0 = 0; c0 < T; c0 += 64) {
    const int i = c0 + lane;
    const bool on = i < T;
    const long long k = r * T + (on ? i : T - 1);
    const float sdf = p.sdf[k], dist = p.dists[k], mz = p.mid_z[k];
    const float gx = p.grad[k * 3 + 0], gy = p.grad[k * 3 + 1], gz = p.grad[k * 3 + 2];
    const float c_r = p.rgb[k * 3 + 0], c_g = p.rgb[k * 3 + 1], c_b = p.rgb[k * 3 + 2];

    // renderer.py:269-286
    const float true_cos = dx * gx + dy * gy + dz * gz;
    const float iter_cos = -(fmaxf(-true_cos * 0.5f + 0.5f, 0.f) * (1.0f - car) + fmaxf(-true_cos, 0.f) * car);
    const float prev_cdf = sigmoidf_((sdf - iter_cos * dist * 0.5f) * inv_s);
    const float next_cdf = sigmoidf_((sdf + iter_cos * dist * 0.5f) * inv_s);
    float alpha = (prev_cdf - next_cdf + 1e-5f) / (prev_cdf + 1e-5f);
    alpha = fminf(fmaxf(alpha, 0.f), 1.f);
    if (!on) alpha = 0.f;

    // weights = alpha * cumprod([1, 1 - alpha + 1e-7])[:-1]   (renderer.py:300): wavefront product scan
    const float incl = wave_scan_mul(on ? 1.0f - alpha + 1e-7f : 1.0f, lane);
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 1.0f;
    const float w = alpha * (excl * carry);
    carry *= __shfl(incl, 63, 64);

    const float px = ox + dx * mz, py = oy + dy * mz, pz = oz + dz * mz;
    const float pn = sqrtf(px * px + py * py + pz * pz);
    const float gn = sqrtf(gx * gx + gy * gy + gz * gz);

    // Phong terms (lighting.py:167-170, 212-225; generator.py:128-152)
    const float gnc = fmaxf(gn, 1e-6f);
    const float nx = gx / gnc, ny = gy / gnc, nz = gz / gnc;
    const float ndl = nx * lx + ny * ly + nz * lz;
    const float diff = l_dif * fmaxf(ndl, 0.f);
    float vx = ox - px, vy = oy - py, vz = oz - pz;
    normalize3(vx, vy, vz, 1e-6f);
    const float rx = -lx + 2.0f * (ndl * nx), ry = -ly + 2.0f * (ndl * ny), rz = -lz + 2.0f * (ndl * nz);
    const float al = fmaxf(vx * rx + vy * ry + vz * rz, 0.f) * (ndl > 0.f ? 1.f : 0.f);
    const float spec = l_spec * powf(al, l_shin);
    const float shade = l_amb + diff;

    if (c0 == 0) cdf_first = __shfl(prev_cdf, 0, 64);
    if (on && live) {
      if (p.weights) p.weights[k] = w;
      if (p.cdf) p.cdf[k] = prev_cdf;
      if (p.alpha) p.alpha[k] = alpha;
      if (p.inside_sphere) p.inside_sphere[k] = pn < 1.0f ? 1.f : 0.f;
      if (p.pts_norm) p.pts_norm[k] = pn;
    }
    if (on) {
      a_wsum += w;
      a_wmax = fmaxf(a_wmax, w);
      a_c0 += w * c_r;
      a_c1 += w * c_g;
      a_c2 += w * c_b;
      a_i0 += w * (shade * c_r + spec);
      a_i1 += w * (shade * c_g + spec);
      a_i2 += w * (shade * c_b + spec);
      a_sh += w * shade;
      a_n0 += w * gx;
      a_n1 += w * gy;
      a_n2 += w * gz;
      a_z += w * mz;
      a_sp += w * spec;
      a_df += w * diff;
      const float m = pn < 1.2f ? 1.f : 0.f;  // relax_inside_sphere (renderer.py:290)
      a_eik += m * (gn - 1.0f) * (gn - 1.0f);
      a_m += m;
      a_surf += expf(-100.0f * fabsf(sdf));
    }
  }
  a_wsum = wave_sum(a_wsum);
  a_wmax = wave_max(a_wmax);
  a_c0 = wave_sum(a_c0); a_c1 = wave_sum(a_c1); a_c2 = wave_sum(a_c2);
  a_i0 = wave_sum(a_i0); a_i1 = wave_sum(a_i1); a_i2 = wave_sum(a_i2);
  a_sh = wave_sum(a_sh);
  a_n0 = wave_sum(a_n0); a_n1 = wave_sum(a_n1); a_n2 = wave_sum(a_n2);
  a_z = wave_sum(a_z); a_sp = wave_sum(a_sp); a_df = wave_sum(a_df);
  a_eik = wave_sum(a_eik); a_m = wave_sum(a_m); a_surf = wave_sum(a_surf);

  if (lane == 0 && live) {
    if (p.weight_sum) p.weight_sum[r] = a_wsum;
    if (p.weight_max) p.weight_max[r] = a_wmax;
    if (p.color_fine) { p.color_fine[r * 3] = a_c0; p.color_fine[r * 3 + 1] = a_c1; p.color_fine[r * 3 + 2] = a_c2; }
    if (p.image_no_bg) { p.image_no_bg[r * 3] = a_i0; p.image_no_bg[r * 3 + 1] = a_i1; p.image_no_bg[r * 3 + 2] = a_i2; }
    if (p.image) {
      const float t = 1.0f - a_wsum;  // generator.py:159
      const float b0 = p.bg ? p.bg[e * 3 + 0] : 0.f, b1 = p.bg ? p.bg[e * 3 + 1] : 0.f, b2 = p.bg ? p.bg[e * 3 + 2] : 0.f;
      if (p.image_planar) {  // [B][3][N / B]: the (B, 3, H, W) map the discriminator reads, no strided-to-contiguous copy
        const long long hw = p.N / p.B, px_ = r - (long long)e * hw;
        p.image[((long long)e * 3 + 0) * hw + px_] = a_i0 + b0 * t;
        p.image[((long long)e * 3 + 1) * hw + px_] = a_i1 + b1 * t;
        p.image[((long long)e * 3 + 2) * hw + px_] = a_i2 + b2 * t;
      } else {
        p.image[r * 3] = a_i0 + b0 * t;
        p.image[r * 3 + 1] = a_i1 + b1 * t;
        p.image[r * 3 + 2] = a_i2 + b2 * t;
      }
    }
    if (p.shading) p.shading[r] = a_sh;
    if (p.normal) { p.normal[r * 3] = a_n0; p.normal[r * 3 + 1] = a_n1; p.normal[r * 3 + 2] = a_n2; }
    if (p.mask) p.mask[r] = fminf(fmaxf(a_wsum, 1e-3f), 1.0f - 1e-3f);
    if (p.z_map) p.z_map[r] = a_z;
    if (p.specular_map) p.specular_map[r] = a_sp;
    if (p.diffuse_map) p.diffuse_map[r] = a_df;
  }
  if (p.block_partials != nullptr || p.reduce4 != nullptr) {
    // per-ray terms of the global reductions: [0..2] eikonal numerator / denominator / surface sum (renderer.py:306-311,
    // 459-461), [4..6] cdf of the first sample, weight_max, weight_sum (the logging means of generator.py:208-213)
    if (lane == 0) {
      red[wave][0] = live ? a_eik : 0.f;
      red[wave][1] = live ? a_m : 0.f;
      red[wave][2] = live ? a_surf : 0.f;
      red[wave][3] = live ? cdf_first : 0.f;
      red[wave][4] = live ? a_wmax : 0.f;
      red[wave][5] = live ? a_wsum : 0.f;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
      float v = 0.f;
      for (int w = 0; w < RAYS_PER_BLOCK; ++w) v += red[w][threadIdx.x];
      if (p.block_partials != nullptr) {
        // parked per block and summed by render_stats_kernel: 1024 blocks adding onto the same three addresses with
        // atomics cost more than the rest of this kernel (37 us vs 16 us at N = 4096)
        const int slot = threadIdx.x < 3 ? threadIdx.x : threadIdx.x + 1;
        // (stats16 given: the LAST block to arrive sums the partials in this launch -- the partials then are agent-scope
        //  write-through stores, which the ticket below may follow once they are acknowledged)
        if (p.stats16 != nullptr) __hip_atomic_store(p.block_partials + (size_t)blockIdx.x * 8 + slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else p.block_partials[(size_t)blockIdx.x * 8 + slot] = v;
      } else if (threadIdx.x < 3) {
        atomicAdd(p.reduce4 + threadIdx.x, v);
      }
    }
  }
  if (p.stats16 != nullptr && p.block_partials != nullptr) {
    // render_stats_kernel's work by the last block of THIS launch (one launch and ~4.5 us less per render).  Hand-off per
    // MI355X_MICROARCH.md ("valid forms"): write-through (sc1) payload stores -> s_waitcnt vmcnt(0) -> agent-scope ticket;
    // the last arriver reads the payload with agent-scope (L1-bypassing) loads.  No block waits for another one.
    __shared__ int is_last;
    if (wave == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) is_last = oi::last_arriver(p.stats_ticket, blockIdx.x, gridDim.x);
    }
    __syncthreads();
    if (!is_last) return;
    stats_reduce<true>(p.block_partials, (int)gridDim.x, (float)p.N, (float)p.N * (float)p.T, p.stats16);
  }
}

// stand-alone form of stats_reduce (callers that keep the two launches)
__global__ void __launch_bounds__(256)
render_stats_kernel(const float* __restrict__ block_partials, int n_blocks, float n_rays, float n_samples,
                    float* __restrict__ out) {
  stats_reduce<false>(block_partials, n_blocks, n_rays, n_samples, out);
}

}  // namespace

extern "C" {

int oi_gen_rays(const float* c2b, const float* kinv, const float* offs, int B, int R, float* rays_o, float* rays_d,
                float* near_, float* far_, oi_stream_t stream) {
  OI_REQUIRE(c2b && kinv && offs && rays_o && rays_d && near_ && far_, "oi_gen_rays: null pointer");
  OI_REQUIRE(B > 0 && R > 0, "oi_gen_rays: B=%d R=%d", B, R);
  const long long n = (long long)B * R * R;
  hipLaunchKernelGGL(gen_rays_kernel, dim3(oi::cdiv(n, 256)), dim3(256), 0, oi::as_stream(stream), c2b, kinv, offs, B,
                     R, rays_o, rays_d, near_, far_, nullptr, nullptr, nullptr);
  return oi::check_launch("oi_gen_rays");
}

int oi_gen_rays_light(const float* c2b, const float* kinv, const float* offs, int B, int R, float* rays_o,
                      float* rays_d, float* near_, float* far_, const float* w2b, const float* light_direction,
                      float* light_dir, oi_stream_t stream) {
  OI_REQUIRE(c2b && kinv && offs && rays_o && rays_d && near_ && far_ && w2b && light_direction && light_dir,
             "oi_gen_rays_light: null pointer");
  OI_REQUIRE(B > 0 && R > 0, "oi_gen_rays_light: B=%d R=%d", B, R);
  const long long n = (long long)B * R * R;
  hipLaunchKernelGGL(gen_rays_kernel, dim3(oi::cdiv(n, 256)), dim3(256), 0, oi::as_stream(stream), c2b, kinv, offs, B,
                     R, rays_o, rays_d, near_, far_, w2b, light_direction, light_dir);
  return oi::check_launch("oi_gen_rays_light");
}

int oi_prep_render(const oi_prep_params* p, oi_stream_t stream) {
  OI_REQUIRE(p != nullptr, "oi_prep_render: null params");
  OI_REQUIRE(p->B > 0 && p->B <= OI_PREP_MAX_B && p->R > 0 && p->S > 0 && p->NL >= 0, "oi_prep_render: B=%d (<= %d) R=%d S=%d NL=%d",
             p->B, OI_PREP_MAX_B, p->R, p->S, p->NL);
  OI_REQUIRE(p->kinv && p->pose_out && p->rays_o && p->rays_d && p->near_ && p->far_ && p->z_coarse && p->pts_coarse && p->w_out,
             "oi_prep_render: null pointer");
  OI_REQUIRE((p->light_dir == nullptr) == (p->light_direction == nullptr), "oi_prep_render: light_dir needs light_direction");
  OI_REQUIRE(p->NL == 0 || (p->gw && p->gb && p->bw && p->bb && p->gamma && p->beta), "oi_prep_render: null FiLM pointer");
  OI_REQUIRE(p->z == nullptr || (p->style_w && p->style_b), "oi_prep_render: z given without style weights");
  OI_REQUIRE((p->f3_packed == nullptr) == (p->f3_blob == nullptr), "oi_prep_render: f3_packed and f3_blob come together");
  OI_REQUIRE(p->f3_blob == nullptr || p->NL == 9, "oi_prep_render: the blob covers the 9 FiLM layers (NL=%d)", p->NL);
  const long long n = (long long)p->B * p->R * p->R;
  const int blocks = p->B * (p->NL > 0 ? p->NL : 1) + (int)oi::cdiv(n, PREP_RAYS);
  hipLaunchKernelGGL(prep_render_kernel, dim3(blocks), dim3(256), 0, oi::as_stream(stream), *p);
  return oi::check_launch("oi_prep_render");
}

int oi_coarse_samples(const float* rays_o, const float* rays_d, const float* near_, const float* far_,
                      const float* jitter, long long N, int S, float* z, float* pts, oi_stream_t stream) {
  OI_REQUIRE(rays_o && rays_d && near_ && far_ && z && pts, "oi_coarse_samples: null pointer");
  OI_REQUIRE(N > 0 && S > 0, "oi_coarse_samples: N=%lld S=%d", N, S);
  hipLaunchKernelGGL(coarse_samples_kernel, dim3(oi::cdiv(N * S, 256)), dim3(256), 0, oi::as_stream(stream), rays_o,
                     rays_d, near_, far_, jitter, N, S, z, pts);
  return oi::check_launch("oi_coarse_samples");
}

int oi_midpoints(const float* rays_o, const float* rays_d, const float* z, long long N, int T, float last_dist,
                 float* dists, float* mid_z, float* pts, oi_stream_t stream) {
  OI_REQUIRE(rays_o && rays_d && z && dists && mid_z && pts, "oi_midpoints: null pointer");
  OI_REQUIRE(N > 0 && T > 0, "oi_midpoints: N=%lld T=%d", N, T);
  hipLaunchKernelGGL(midpoints_kernel, dim3(oi::cdiv(N * T, 256)), dim3(256), 0, oi::as_stream(stream), rays_o,
                     rays_d, z, N, T, last_dist, dists, mid_z, pts);
  return oi::check_launch("oi_midpoints");
}

int oi_upsample(const float* rays_o, const float* rays_d, const float* z, const float* sdf, long long N, int Sc,
                int n_new, float inv_s, float* z_new, float* pts_new, float* z_merged, oi_stream_t stream) {
  OI_REQUIRE(rays_o && rays_d && z && sdf && z_new && pts_new, "oi_upsample: null pointer");
  OI_REQUIRE(N > 0 && Sc >= 2 && n_new > 0, "oi_upsample: N=%lld Sc=%d n_new=%d", N, Sc, n_new);
  OI_REQUIRE(Sc <= MAX_SC && n_new <= MAX_SC, "oi_upsample: at most %d samples per ray", MAX_SC);
  const size_t sh = (size_t)RAYS_PER_BLOCK * (3 * Sc + n_new) * sizeof(float);
  hipLaunchKernelGGL(upsample_kernel, dim3(oi::cdiv(N, RAYS_PER_BLOCK)), dim3(256), sh, oi::as_stream(stream), rays_o,
                     rays_d, z, sdf, N, Sc, n_new, inv_s, z_new, pts_new, z_merged, 0.f, nullptr, nullptr, nullptr);
  return oi::check_launch("oi_upsample");
}

int oi_upsample_mid(const float* rays_o, const float* rays_d, const float* z, const float* sdf, long long N, int Sc, int n_new,
                    float inv_s, float* z_new, float* pts_new, float* z_merged, float last_dist, float* dists, float* mid_z,
                    float* pts_mid, oi_stream_t stream) {
  OI_REQUIRE(rays_o && rays_d && z && sdf && z_new && pts_new && z_merged && dists && mid_z && pts_mid, "oi_upsample_mid: null pointer");
  OI_REQUIRE(N > 0 && Sc >= 2 && n_new > 0, "oi_upsample_mid: N=%lld Sc=%d n_new=%d", N, Sc, n_new);
  OI_REQUIRE(Sc <= MAX_SC && n_new <= MAX_SC, "oi_upsample_mid: at most %d samples per ray", MAX_SC);
  const size_t sh = (size_t)RAYS_PER_BLOCK * (3 * Sc + n_new + Sc + n_new) * sizeof(float);  // + the merged list
  auto k = upsample_kernel;
  if (sh > 65536) {
    // more than the default dynamic-LDS limit: opt in, and if this device cannot give the merged list its own rows (the
    // attribute call fails, or the footprint exceeds the CU's 160 KiB), take the two launches this one replaces
    // (bit-identical by construction: tests/test_gpu_kernels.py)
    if (sh > 160 * 1024 ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess) {
      (void)hipGetLastError();
      const int rc = oi_upsample(rays_o, rays_d, z, sdf, N, Sc, n_new, inv_s, z_new, pts_new, z_merged, stream);
      if (rc != OI_OK) return rc;
      return oi_midpoints(rays_o, rays_d, z_merged, N, Sc + n_new, last_dist, dists, mid_z, pts_mid, stream);
    }
  }
  hipLaunchKernelGGL(k, dim3(oi::cdiv(N, RAYS_PER_BLOCK)), dim3(256), sh, oi::as_stream(stream), rays_o, rays_d, z, sdf, N, Sc,
                     n_new, inv_s, z_new, pts_new, z_merged, last_dist, dists, mid_z, pts_mid);
  return oi::check_launch("oi_upsample_mid");
}

int oi_merge_sorted(const float* z, const float* sdf, const float* z_new, const float* sdf_new, long long N, int Sc,
                    int n_new, float* z_out, float* sdf_out, oi_stream_t stream) {
  OI_REQUIRE(z && sdf && z_new && sdf_new && z_out && sdf_out, "oi_merge_sorted: null pointer");
  OI_REQUIRE(N > 0 && Sc > 0 && n_new > 0, "oi_merge_sorted: N=%lld Sc=%d n_new=%d", N, Sc, n_new);
  OI_REQUIRE(Sc <= MAX_SC && n_new <= MAX_SC, "oi_merge_sorted: at most %d samples per ray", MAX_SC);
  const size_t sh = (size_t)RAYS_PER_BLOCK * (Sc + n_new) * sizeof(float);
  hipLaunchKernelGGL(merge_sorted_kernel, dim3(oi::cdiv(N, RAYS_PER_BLOCK)), dim3(256), sh, oi::as_stream(stream), z,
                     sdf, z_new, sdf_new, N, Sc, n_new, z_out, sdf_out);
  return oi::check_launch("oi_merge_sorted");
}

int oi_composite_fwd(const oi_composite_params* p, oi_stream_t stream) {
  OI_REQUIRE(p != nullptr, "oi_composite_fwd: null params");
  OI_REQUIRE(p->sdf && p->grad && p->rgb && p->dists && p->mid_z && p->rays_o && p->rays_d && p->light_dir &&
                 p->variance && p->light,
             "oi_composite_fwd: null input pointer");
  OI_REQUIRE(p->N > 0 && p->T > 0 && p->B > 0 && p->N % p->B == 0, "oi_composite_fwd: N=%lld T=%d B=%d", p->N, p->T,
             p->B);
  OI_REQUIRE(p->stats16 == nullptr || (p->stats_ticket != nullptr && p->block_partials != nullptr),
             "oi_composite_fwd: stats16 needs stats_ticket (OI_TICKET_WORDS zero-initialised words) and block_partials");
  // the two-level arrival counters cover LA_GROUP * (OI_TICKET_WORDS - 1) workgroups
  OI_REQUIRE(p->stats16 == nullptr || oi::cdiv(p->N, RAYS_PER_BLOCK) <= (long long)oi::LA_GROUP * (OI_TICKET_WORDS - 1),
             "oi_composite_fwd: stats16 supports at most %lld rays per launch", (long long)oi::LA_GROUP * (OI_TICKET_WORDS - 1) * RAYS_PER_BLOCK);
  hipLaunchKernelGGL(composite_fwd_kernel, dim3(oi::cdiv(p->N, RAYS_PER_BLOCK)), dim3(256), 0, oi::as_stream(stream),
                     *p);
  return oi::check_launch("oi_composite_fwd");
}

int oi_composite_num_blocks(long long N) { return N > 0 ? oi::cdiv(N, RAYS_PER_BLOCK) : 0; }

int oi_render_stats(const float* block_partials, int n_blocks, long long N, int T, float* out16, oi_stream_t stream) {
  OI_REQUIRE(block_partials && out16, "oi_render_stats: null pointer");
  OI_REQUIRE(N > 0 && T > 0 && n_blocks == oi::cdiv(N, RAYS_PER_BLOCK), "oi_render_stats: N=%lld T=%d n_blocks=%d", N,
             T, n_blocks);
  hipLaunchKernelGGL(render_stats_kernel, dim3(1), dim3(256), 0, oi::as_stream(stream), block_partials, n_blocks,
                     (float)N, (float)N * (float)T, out16);
  return oi::check_launch("oi_render_stats");
}

}  // extern "C"
