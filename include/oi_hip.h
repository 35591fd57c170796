/*
 * oi_hip.h -- C ABI of liboi_hip.so: the MI355X (gfx950 / CDNA4) implementation of the
 * object-intrinsics volumetric-render + GAN-discriminator hot path.
 *
 * This is the drop-in boundary.  The reference binds its native ops as
 * pybind/torch-extension modules built with torch.utils.cpp_extension.load
 *   - fused.fused_bias_act(...)          src/third_party/stylesdf/op/fused_bias_act.cpp:11-20
 *   - upfirdn2d_plugin.upfirdn2d(...)    src/third_party/ada/torch_utils/ops/upfirdn2d.cpp:16-94
 * and everything else on the path is ATen/cuBLAS/cuDNN called from Python
 * (renderer.py, fields.py, generator.py, discriminator.py, augment.py).  This library replaces
 * both groups by hand-written HIP kernels behind plain-C entry points: raw device pointers,
 * sizes and scalars only -- no torch types.  All buffers are owned by the caller (PyTorch
 * allocates them); the library allocates nothing user visible.  Every call is asynchronous and
 * strictly ordered on the hipStream_t passed as `stream` (an opaque void* here so that C
 * callers need no HIP headers); no call synchronises the device or touches the null stream.
 * Calls are re-entrant and thread safe (they may come from autograd worker threads).
 *
 * Return value: 0 (OI_OK) or a negative oi_status; oi_last_error() gives a thread-local message.
 * Tensors are dense row-major fp32 unless stated.  "element" = batch element of the generator;
 * point/ray rows are element-major: row r belongs to element r / (rows / B)  (fields.py:55).
 */
#ifndef OI_HIP_H_
#define OI_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* oi_stream_t; /* hipStream_t */

enum oi_status {
  OI_OK = 0,
  OI_ERR_INVALID_ARG = -1,
  OI_ERR_LAUNCH = -2,
  OI_ERR_UNSUPPORTED = -3,
};

/* MFMA operand precision of the MLP contractions (accumulation is always fp32, FiLM phase and
 * sin/cos always fp32).  F32 = v_mfma_f32_32x32x2_f32 (exact fp32, the 1e-4 parity path);
 * BF16X3 = three bf16 MFMAs on hi/lo splits of both operands (fp32-class accuracy at 3/16 of the
 * fp32 matrix cost); BF16 = one bf16 MFMA (throughput path, tolerance stated in DESIGN.md);
 * BF16X6 = fp32 operands split THREE ways (hi/mid/lo bf16 = the full 24-bit mantissa) and the six products
 * whose weight is >= 2^-24 accumulated in fp32: fp32-exact contractions on the bf16 matrix cores at 6/16 of the
 * fp32 MFMA cost (forward kernel only; the backward kernels take the F32 image);
 * F16X3 = fp32 operands split TWO ways into fp16 limbs (hi + lo = 22 mantissa bits; fp16 subnormals are kept by
 * v_cvt and by the MFMA on gfx950) and the three products hi*hi, hi*lo, lo*hi accumulated in fp32: 2^-22 relative
 * per product (fp32-class, inside the 1e-4 parity bar) at HALF the MFMA work and 2/3 of the LDS image of BF16X6.
 * Activations are sin() outputs in [-1,1]; the adjoint vectors of the analytic-gradient sweep are normalised per
 * point by a power of two before the split, so the fp16 range is never exceeded.  The weight images carry a
 * per-image power-of-two scale chosen from the image's own max |w| (scaled peak in [2^13, 2^14)), so EVERY finite
 * weight is inside the fp16 range by construction -- there is no |w| bound to respect; an image with an inf / NaN
 * weight is flagged in the packed header and reported by oi_mlp_pack_status (OI_ERR_UNSUPPORTED). */
enum oi_precision { OI_PREC_F32 = 0, OI_PREC_BF16X3 = 1, OI_PREC_BF16 = 2, OI_PREC_BF16X6 = 3, OI_PREC_F16X3 = 4 };

int oi_version(void);
const char* oi_arch(void);       /* "gfx950" */
const char* oi_last_error(void); /* thread local */

/* ---------------------------------------------------------------------------------------------
 * a1 + a2: style MLP and FiLM parameters.
 * Replaces ShapeNetwork.style (src/models/fields.py:15-21; MappingLinear stylesdf/model.py:49-54 ->
 * fused_bias_act) and the gamma/beta LinearLayers evaluated inside every FiLMSiren.forward
 * (stylesdf/volume_renderer.py:27-30, 47-48, 56-57) -- computed ONCE per render here.
 *   style_w [3][64][64], style_b [3][64]; z [B][64] -> w_out [B][64]   (skipped when z == NULL:
 *   w_out is then an input).  gw/bw [NL][128][64], gb/bb [NL][128]  ->  gamma, beta [B][NL][128]
 *   with gamma = 15*(w Wg^T + bg) + 30, beta = 0.25*(w Wb^T + bb).  NL = 9 (8 sdf layers + colour).
 */
int oi_film_params(const float* style_w, const float* style_b, const float* z, float* w_out,
                   const float* gw, const float* gb, const float* bw, const float* bb,
                   float* gamma, float* beta, int B, int NL, oi_stream_t stream);
/* Backward of oi_film_params (what autograd derives for the three MappingLinear layers and the 2 x 9 FiLM heads,
 * stylesdf/model.py:49-54, volume_renderer.py:27-30).  d_gamma, d_beta [B][NL][128] in; d_gw, d_bw [NL][128][64] and
 * d_gb, d_bb [NL][128] are ASSIGNED; d_w [B][64] is ACCUMULATED (pass zeros, or the upstream gradient of w).
 * With z != NULL the style MLP is back-propagated too: d_style_w [3][64][64] and d_style_b [3][64] are ACCUMULATED
 * (zero them), d_z [B][64] (optional) is assigned. */
int oi_film_params_bwd(const float* d_gamma, const float* d_beta, const float* w, const float* gw, const float* bw,
                       float* d_gw, float* d_gb, float* d_bw, float* d_bb, float* d_w, const float* style_w,
                       const float* style_b, const float* z, float* d_style_w, float* d_style_b, float* d_z, int B,
                       int NL, oi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Weight pre-pack for the MFMA kernels (run when weights change; a few microseconds).
 * Reorders the FiLM-SIREN matrices into the lane-linear A-operand images the MLP kernel streams
 * through LDS (DESIGN.md "MLP kernel").  Inputs carry the reference's shapes
 * (SURVEY.md 8b state_dict): w0 [128][3], b0 [128]; wh [7][128][128], bh [7][128] (layers 1..7);
 * wsig [128], bsig [1]; wv [128][131], bv [128]; wrgb [3][128], brgb [3].
 * `packed` must hold oi_mlp_packed_bytes(prec) bytes: header, 16 MFMA images (7 forward, 7 transposed, colour head both
 * ways) and the 8 forward matrices once more as plain fp32 (read by the backward for the FiLM-scale identity).
 */
size_t oi_mlp_packed_bytes(int prec);
int oi_mlp_pack_weights(const float* w0, const float* b0, const float* wh, const float* bh,
                        const float* wsig, const float* bsig, const float* wv, const float* bv,
                        const float* wrgb, const float* brgb, void* packed, int prec,
                        oi_stream_t stream);
/* Range / finiteness report of a packed image (the pre-pack itself stays asynchronous).  Synchronises `stream`, reads the
 * packed header back and returns OI_ERR_UNSUPPORTED (oi_last_error names the image) when a weight, bias or head value
 * is inf / NaN -- the only input the F16X3 images cannot represent; OI_OK otherwise.  Call it where a sync is free:
 * after loading a checkpoint, before an inference run (oi_amd.fields.FieldPack.check does). */
int oi_mlp_pack_status(const void* packed, oi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a3-a6: FiLM-SIREN SDF network (+ analytic d sdf/dx + colour head) at n points per element.
 * Replaces ShapeNetwork.forward/.sdf/.gradient and ColorNetwork.forward (src/models/fields.py:49-77,
 * 89-101, 104-122) and the F.linear/sin chains of FiLMSiren.forward (volume_renderer.py:50-61).
 *   pts [B*n][3]; gamma/beta from oi_film_params ([B][9][128]).
 *   sdf [B*n] (always).  If grad != NULL: grad [B*n][3] (raw d sdf/dx) and, if rgb != NULL,
 *   rgb [B*n][3] = colour head on [feat, grad].  feat [B*n][128] optional (NULL to skip).
 *   scratch: oi_mlp_scratch_bytes_prec(B, n, prec) bytes, needed when grad != NULL.  OI_PREC_F16X3 (the default mode)
 *   runs the register-resident kernel (csrc/mlp_fwd3.hip): the phases the reverse sweep needs stay in the register
 *   file and only the 128 features per point cross HBM (512 B/point); OI_PREC_BF16 (csrc/mlp_fwd3b.hip) keeps nothing per
 *   point in memory -- its scratch holds the 15 per-element images diag(gamma_l) W_l rounded to bf16 plus a 15 KiB table blob
 *   (15 * 32 KiB + 15 KiB = 495 KiB per batch element, built by a first launch of the same call); the other modes park
 *   gamma*cos(phase) of every layer (4.6 KB/point).  oi_mlp_scratch_bytes(B, n) is an upper bound over all modes.
 *   The call takes no scratch_bytes: the caller MUST size `scratch` with oi_mlp_scratch_bytes_prec for the same (B, n, prec) --
 *   the minimum differs per mode (OI_PREC_BF16: B * 495 KiB whatever n is) and an undersized buffer is written past its end.
 */
size_t oi_mlp_scratch_bytes(int B, long long n_per_elem);
size_t oi_mlp_scratch_bytes_prec(int B, long long n_per_elem, int prec);
int oi_sdf_mlp_fwd(const float* pts, const void* packed, const float* gamma, const float* beta,
                   float* sdf, float* grad, float* rgb, float* feat, void* scratch,
                   int B, long long n_per_elem, int prec, int fast_trig, oi_stream_t stream);
/* The same with `flags`.  OI_MLP_BLOB_READY (OI_PREC_F16X3 with grad != NULL only): the per-element table blobs the register-
 * resident kernel stages from -- oi_mlp_f3_blob_bytes() bytes per batch element at byte oi_mlp_f3_blob_offset(B, n) of `scratch` --
 * have already been written for THESE gamma / beta / packed by oi_prep_render (its `f3_packed` / `f3_blob` fields): the call then
 * skips its own blob launch (one launch and its boundary less per render; the bytes are the same: csrc/f3_blob.h). */
#define OI_MLP_BLOB_READY 1
size_t oi_mlp_f3_blob_offset(int B, long long n_per_elem);
size_t oi_mlp_f3_blob_bytes(void);
int oi_sdf_mlp_fwd_ex(const float* pts, const void* packed, const float* gamma, const float* beta,
                      float* sdf, float* grad, float* rgb, float* feat, void* scratch,
                      int B, long long n_per_elem, int prec, int fast_trig, int flags, oi_stream_t stream);

/* Backward of oi_sdf_mlp_fwd w.r.t. every parameter and the FiLM vectors -- including the second-order
 * terms that arise because d sdf/dx is a forward output (the reference: autograd with create_graph=True
 * through fields.py:104-122, backward at gan_pose_trainer.py:141).  Upstream gradients g_sdf [B*n],
 * g_grad [B*n][3], g_rgb [B*n][3] may be NULL.  grad_fwd / rgb_fwd / feat_fwd ([B*n][128]) are the forward outputs
 * (needed when g_rgb != NULL: the albedo head is differentiated first, from the features the forward wrote, because its
 * gradient with respect to d sdf/dx is an INPUT of the sweep that recomputes the network).  Results are ACCUMULATED (atomics) into caller-zeroed buffers:
 *   d_small  oi_mlp_bwd_small_floats() floats: dW0 [128][3] | db [9][128] (b0..b7, bv) | dwsig [128] | dbsig [1]+3 |
 *            dWv[:,128:131] [128][3] | dWrgb [3][128] | dbrgb [3]+1
 *   d_wmat   [8][128][128]: dW_1..dW_7, dWv[:, :128]
 *   d_gamma, d_beta [B][9][128]
 * scratch / scratch_bytes: working memory of the launch.  It is a BOUND, not a function of the problem size: the points of
 * every batch element are processed in chunks of as many 128-point tiles as the buffer holds (16 KiB per point and
 * 2 MiB x B per tile, behind an 8 KiB header that carries the launch-wide operand maxima from the sweep to the
 * weight-gradient GEMM), chunks accumulate into the same outputs.  oi_mlp_bwd_scratch_bytes(B, n) is the size that takes
 * one chunk; oi_mlp_bwd_scratch_bytes_capped(B, n, cap) the largest tile multiple <= cap (at least one tile). */
size_t oi_mlp_bwd_scratch_bytes(int B, long long n_per_elem);
size_t oi_mlp_bwd_scratch_bytes_capped(int B, long long n_per_elem, size_t cap_bytes);
int oi_mlp_bwd_small_floats(void);
/* Test hook for the CU-indexed feature scratch of the register-resident forward kernel: launches n_workgroups
 * workgroups with that kernel's LDS footprint; each marks busy[slot] (slot = XCC id x 256 + SE/SH/CU bits of HW_ID, < 4096)
 * on entry and clears it on exit.  *clashes counts workgroups that found their slot already busy (must stay 0);
 * used[slot] != 0 for every slot seen.  busy / used: 4096 zeroed ints. */
int oi_selftest_cu_slots(int* busy, int* clashes, int* used, int n_workgroups, int spin, oi_stream_t stream);

/* Test hook: the sin / cos the MLP kernels apply to a FiLM phase (fast != 0: the unreduced form of the bf16 throughput
 * mode).  x, s, c: n floats. */
int oi_selftest_sincos(const float* x, float* s, float* c, long long n, int fast, oi_stream_t stream);

/* Test hook: the 24-bit fixed-point slot format of the f16x3 backward (csrc/mlp_bwd.hip: pack_q24 / unpack_q24), one round trip
 * on the device.  x, y: n floats, n a multiple of 64; every run of 64 values is one lane's vector (shares one power-of-two scale).
 * mode 0: values -- y = what a consumer reconstructs; mode 1: phases in [0, 1) revolutions -- y = the decoded phase + 1 (what the
 * trig instructions are handed: the period is 1). */
int oi_selftest_q24(const float* x, float* y, long long n, int mode, oi_stream_t stream);

int oi_sdf_mlp_bwd(const float* pts, const void* packed, const float* gamma, const float* beta,
                   const float* grad_fwd, const float* rgb_fwd, const float* feat_fwd, const float* g_sdf, const float* g_grad,
                   const float* g_rgb, float* d_small, float* d_wmat, float* d_gamma, float* d_beta,
                   void* scratch, size_t scratch_bytes, int B, long long n_per_elem, int prec, int fast_trig,
                   oi_stream_t stream);
/* The same with an upstream gradient of the FEATURE output (feat of oi_sdf_mlp_fwd = the second to 129th column of
 * ShapeNetwork.forward, fields.py:72): g_feat [B*n][128] joins abar_8 at the turn of the sweep.  For a caller that keeps the
 * reference's renderer.py:241-261 (features read out, albedo head evaluated by oi_color_head_fwd): the head's d_feat comes back
 * through here.  g_feat == NULL is oi_sdf_mlp_bwd. */
int oi_sdf_mlp_bwd_feat(const float* pts, const void* packed, const float* gamma, const float* beta,
                        const float* grad_fwd, const float* rgb_fwd, const float* feat_fwd, const float* g_sdf,
                        const float* g_grad, const float* g_rgb, const float* g_feat, float* d_small, float* d_wmat,
                        float* d_gamma, float* d_beta, void* scratch, size_t scratch_bytes, int B, long long n_per_elem,
                        int prec, int fast_trig, oi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a6 stand-alone: the albedo head on CALLER-SUPPLIED features and normals.
 * Replaces ColorNetwork.forward(points, normals, view_dirs, feature_vectors, z, w) (src/models/fields.py:89-101: `points`,
 * `view_dirs` and `z` are ignored there) = FiLMSiren(131 -> 128) (stylesdf/volume_renderer.py:50-61) + rgb_linear + sigmoid,
 * for a caller that keeps the reference's renderer.py:241-261.  (NeuSRenderer.render of this library never calls it: there
 * the head is the tail of oi_sdf_mlp_fwd and its inputs stay in registers.)
 *   feat [B*n][128], normals [B*n][3] (raw d sdf/dx); gamma / beta: row e at gamma + e * film_stride (128 floats each; pass
 *   oi_film_params' [B][9][128] + 8 * 128 with film_stride = 9 * 128, or a dense [B][128] with 128);
 *   wv [128][131], bv [128], wrgb [3][128], brgb [3] in the reference's state_dict layout  ->  rgb [B*n][3].
 * Exact fp32 (v_mfma_f32_32x32x2_f32).
 * Backward: g_rgb [B*n][3] -> d_feat [B*n][128], d_normals [B*n][3], d_gamma / d_beta (row e at + e * d_film_stride), d_wv,
 * d_bv, d_wrgb, d_brgb -- all ASSIGNED; sums over the points are formed from per-workgroup partials in a fixed order (no
 * atomics: bit-reproducible).  workspace: oi_color_head_bwd_workspace_bytes(B, n) bytes (~1 KB per point). */
int oi_color_head_fwd(const float* feat, const float* normals, const float* gamma, const float* beta, long long film_stride,
                      const float* wv, const float* bv, const float* wrgb, const float* brgb, float* rgb, int B,
                      long long n_per_elem, oi_stream_t stream);
size_t oi_color_head_bwd_workspace_bytes(int B, long long n_per_elem);
int oi_color_head_bwd(const float* feat, const float* normals, const float* gamma, const float* beta, long long film_stride,
                      const float* wv, const float* bv, const float* wrgb, const float* brgb, const float* g_rgb, float* d_feat,
                      float* d_normals, float* d_gamma, float* d_beta, long long d_film_stride, float* d_wv, float* d_bv,
                      float* d_wrgb, float* d_brgb, void* workspace, size_t workspace_bytes, int B, long long n_per_elem,
                      oi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a13 + a14: crop rays.  Replaces Generator.gen_rays_at + build_rays + near_far_from_sphere
 * (src/models/generator.py:255-279, 317-333, 336-342).
 *   c2b [B][4][4] (camera->box), kinv [3][3] (row-major 3x3 of intrinsics_inv), offs [B][2]
 *   (x_offset, y_offset in scene pixels), R = crop resolution  ->
 *   rays_o, rays_d [B][R][R][3], near, far [B*R*R].
 */
int oi_gen_rays(const float* c2b, const float* kinv, const float* offs, int B, int R,
                float* rays_o, float* rays_d, float* near, float* far, oi_stream_t stream);
/* The same launch also writing the light direction of every box frame,
 *   light_dir[b] = w2b[b][:3,:3] (light_direction / |light_direction|)
 * i.e. DirectionalLightWithSpecularFixInit.direction + BatchDirectionalLight...direction
 * (src/models/lighting.py:62-64, 115-119): three tiny launches of the reference's tensor code folded into
 * the ray kernel.  Forward-only (the host mirror keeps the tensor path when the direction needs a gradient).
 *   w2b [B][4][4], light_direction [3] (the raw parameter), light_dir [B][3].
 */
int oi_gen_rays_light(const float* c2b, const float* kinv, const float* offs, int B, int R,
                      float* rays_o, float* rays_d, float* near, float* far, const float* w2b,
                      const float* light_direction, float* light_dir, oi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a8: coarse samples.  z[r][i] = near + (far-near) i/(S-1) (+ (jitter[r]-0.5)*2/S if jitter),
 * pts[r][i] = o + d z.   NeuSRenderer.render, renderer.py:359-373, 391.
 */
/* Everything a render needs before its first MLP pass, in ONE launch: oi_gen_rays_light + oi_film_params + oi_coarse_samples,
 * with the pose block passed BY VALUE (no host-to-device copy in front).  Host arrays: b2w / w2b / c2b [B][16] (row-major
 * 4x4), offs [B][2] (crop offsets, generator.py:262-268), bg [B][3]; B <= OI_PREP_MAX_B.  Device inputs: kinv [3][3],
 * light_direction [3] (raw parameter; NULL with light_dir NULL), jitter [B R R] or NULL, and oi_film_params' inputs.
 * Outputs: pose_out [53 B] = b2w [B][16] | w2b [B][16] | c2b [B][16] | offs [B][2] | bg [B][3] (device copies for the pose
 * loss / compositing), rays_o / rays_d
 * [B R R][3], near_ / far_ [B R R], light_dir [B][3], z_coarse [B R R][S], pts_coarse [B R R][S][3], w_out / gamma / beta as
 * oi_film_params.  Results are bit-identical to the three separate entry points (tests/test_gpu_kernels.py). */
#define OI_PREP_MAX_B 8
typedef struct oi_prep_params {
  float b2w[OI_PREP_MAX_B][16], w2b[OI_PREP_MAX_B][16], c2b[OI_PREP_MAX_B][16], offs[OI_PREP_MAX_B][2], bg[OI_PREP_MAX_B][3];
  int B, R, S, NL;
  const float *kinv, *light_direction, *jitter;
  const float *style_w, *style_b, *z, *gw, *gb, *bw, *bb;
  float *pose_out, *rays_o, *rays_d, *near_, *far_, *light_dir, *z_coarse, *pts_coarse, *w_out, *gamma, *beta;
  /* round 6.  jitter_normal != 0: `jitter` holds one STANDARD NORMAL draw per ray (the caller drew latents and jitter with one
   * generator call); the kernel maps it to the uniform of renderer.py:372 through the normal CDF.  f3_packed / f3_blob (both or
   * neither; NL must be 9): the OI_PREC_F16X3 packed image and where the FiLM workgroups write the per-element blobs of the
   * register-resident MLP kernel (B x oi_mlp_f3_blob_bytes(); see oi_sdf_mlp_fwd_ex / OI_MLP_BLOB_READY). */
  int jitter_normal;
  const void* f3_packed;
  void* f3_blob;
} oi_prep_params;
int oi_prep_render(const oi_prep_params* p, oi_stream_t stream);

int oi_coarse_samples(const float* rays_o, const float* rays_d, const float* near, const float* far,
                      const float* jitter, long long N, int S, float* z, float* pts,
                      oi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a9 + a10 (+ a11): SDF-guided importance resampling, one wavefront per ray.
 * Replaces NeuSRenderer.up_sample + sample_pdf(det=True) + cat_z_vals
 * (src/third_party/neus/models/renderer.py:137-181, 44-74, 183-197).
 *   z, sdf [N][Sc] (sorted z) -> z_new [N][n_new] and pts_new [N][n_new][3] (for the next SDF pass).
 *   If z_merged != NULL also writes the ascending merge of z and z_new ([N][Sc+n_new]).
 */
int oi_upsample(const float* rays_o, const float* rays_d, const float* z, const float* sdf,
                long long N, int Sc, int n_new, float inv_s, float* z_new, float* pts_new,
                float* z_merged, oi_stream_t stream);

/* a11 when more up-sampling steps follow: merge (z, sdf) with (z_new, sdf_new), ascending in z. */
/* oi_upsample with merge AND the section mid-points of the merged list (oi_midpoints) in the same launch: the last
 * up-sampling step of a render.  z_merged [N][Sc+n_new], dists / mid_z [N][Sc+n_new], pts_mid [N][Sc+n_new][3];
 * last_dist as in oi_midpoints.  Results are bit-identical to oi_upsample followed by oi_midpoints. */
int oi_upsample_mid(const float* rays_o, const float* rays_d, const float* z, const float* sdf, long long N, int Sc, int n_new,
                    float inv_s, float* z_new, float* pts_new, float* z_merged, float last_dist, float* dists, float* mid_z,
                    float* pts_mid, oi_stream_t stream);

int oi_merge_sorted(const float* z, const float* sdf, const float* z_new, const float* sdf_new,
                    long long N, int Sc, int n_new, float* z_out, float* sdf_out, oi_stream_t stream);

/* Section mid-points of the merged z: dists (last = 2/S), mid_z, pts = o + d*mid_z.
 * renderer.py:219-228. */
int oi_midpoints(const float* rays_o, const float* rays_d, const float* z, long long N, int T,
                 float last_dist, float* dists, float* mid_z, float* pts, oi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a12 (tail) + a15: NeuS alpha compositing with a wavefront transmittance scan, fused with the
 * Phong shading and the weighted sums to image-space maps.
 * Replaces NeuSRenderer.render_core after the network calls (renderer.py:266-311, 338) and
 * Generator.render_maps + lighting.diffuse/specular (generator.py:80-174; lighting.py:126-225).
 */
/* arrival counters of the launches that finish with a reduction by their last workgroup (two-level: csrc/oi_common.h) */
#define OI_TICKET_WORDS 4097

typedef struct oi_composite_params {
  /* per sample [N][T] / [N][T][3] */
  const float* sdf;
  const float* grad;
  const float* rgb;
  const float* dists;
  const float* mid_z;
  /* per ray [N][3] */
  const float* rays_o;
  const float* rays_d;
  /* per element */
  const float* light_dir; /* [B][3] light direction in the box frame (not normalised) */
  const float* bg;        /* [B][3] background colour, may be NULL (then image == image_no_bg) */
  /* scalars */
  const float* variance;  /* device pointer to the SingleVarianceNetwork parameter */
  const float* light;     /* device pointer to [param_ambient (logit), param_specular, param_shininess] of
                             DirectionalLightWithSpecularFixInit (lighting.py:33-52): ambient = sigmoid(l[0]),
                             diffuse = 1 - ambient, specular = max(l[1], 0), shininess = l[2].  No host sync. */
  float cos_anneal_ratio;
  long long N; /* rays = B*H*W */
  int T;       /* samples per ray */
  int B;       /* elements; N % B == 0 */
  /* per-sample outputs (any may be NULL): weights, cdf (prev_cdf), alpha, inside (<1.0), pts_norm */
  float* weights;
  float* cdf;
  float* alpha;
  float* inside_sphere;
  float* pts_norm;
  /* per-ray outputs [N] / [N][3] (any may be NULL) */
  float* weight_sum;
  float* weight_max;
  float* color_fine;   /* sum w*rgb */
  float* image_no_bg;  /* sum w*(shade*rgb + spec) */
  float* image;        /* + bg*(1-weight_sum) */
  float* shading;      /* sum w*shade (1 channel; the reference's 3 channels are identical) */
  float* normal;       /* sum w*grad (raw) */
  float* mask;         /* clamp(weight_sum, 1e-3, 1-1e-3) */
  float* z_map;        /* sum w*mid_z */
  float* specular_map; /* sum w*spec (1 channel) */
  float* diffuse_map;  /* sum w*diff (1 channel) */
  /* global reductions, accumulated with atomics; caller zeroes them: [0]=sum m*(|g|-1)^2,
   * [1]=sum m, [2]=sum exp(-100|sdf|), [3]=min mid_z is NOT here (see z_min) */
  float* reduce4;
  /* optional [oi_composite_num_blocks(N)][8] float workspace.  With it the kernel parks, per block and without
   * atomics, [0..2] = the three reduce4 terms and [4..6] = sum cdf[r][0], sum weight_max, sum weight_sum (the logging
   * means of generator.py:208-213 before the division); `reduce4` is then ignored and oi_render_stats turns the
   * workspace into totals + derived scalars.  (1024 blocks adding onto three addresses with atomics cost more than
   * the rest of the kernel: 37 us vs 16 us at N = 4096.) */
  float* block_partials;
  /* round 4 (appended: older callers that zero-initialise the struct keep the old behaviour)
   *   stats16 + stats_ticket: the launch ALSO does oi_render_stats' work -- the last workgroup to finish sums block_partials
   *     in oi_render_stats' order (bit-identical) and writes out16 to stats16.  stats_ticket: OI_TICKET_WORDS device words, zero
   *     before the first launch (the kernel leaves them zero); not to be shared by launches that may overlap (one per stream).
   *   image_planar != 0: `image` is written as [B][3][N / B] (the (B, 3, H, W) map itself) instead of [N][3]. */
  float* stats16;
  unsigned* stats_ticket;
  int image_planar;
} oi_composite_params;

int oi_composite_fwd(const oi_composite_params* p, oi_stream_t stream);
int oi_composite_num_blocks(long long N);

/* Reduces `block_partials` (fixed summation order) and derives the scalars the reference gets from ~8 tiny tensor
 * launches.  out16: [0..3] reduce4 totals, [4..7] ray sums,
 *   [8]  gradient_error = out[0] / (out[1] + 1e-5)                        (renderer.py:306-311)
 *   [9]  surface_loss   = out[2] / (N*T)                                  (renderer.py:459-461)
 *   [10..12] mean over rays of cdf[:,0], weight_max, weight_sum           (generator.py:208-213)
 * The host mirror feeds [8..12] to callers only when no gradient is recorded (autograd needs the tensor formulas). */
int oi_render_stats(const float* block_partials, int n_blocks, long long N, int T, float* out16,
                    oi_stream_t stream);

/* Backward of oi_composite_fwd (what autograd derives for renderer.py:266-311 + generator.py:107-172 in the
 * reference).  `fwd` repeats the forward inputs (outputs ignored).  Upstream gradients (any may be NULL):
 * per-ray [N] / [N][3] for the maps, g_weights [N][T], g_reduce4 [4] (device; grads of the three global sums).
 * Results: d_sdf [N][T], d_grad [N][T][3], d_rgb [N][T][3] (written), and accumulated with atomics into
 * caller-zeroed d_variance [1], d_light [3] (param_ambient, param_specular, param_shininess),
 * d_light_dir [B][3] (w.r.t. the normalised direction). */
typedef struct oi_composite_grads {
  const float* g_weights;
  const float* g_weight_sum;
  const float* g_color_fine;
  const float* g_image_no_bg;
  const float* g_image;
  const float* g_shading;
  const float* g_normal;
  const float* g_mask;
  const float* g_z_map;
  const float* g_specular_map;
  const float* g_diffuse_map;
  const float* g_reduce4;
  float* d_sdf;
  float* d_grad;
  float* d_rgb;
  float* d_variance;
  float* d_light;
  float* d_light_dir;
  /* optional [N][8] float workspace: with it the eight per-ray partial sums behind d_light / d_variance /
   * d_light_dir are parked per ray and reduced by a second tiny kernel (fixed order); without it (NULL) every wave
   * adds them with atomics onto the same seven addresses (measured 223 us vs 25 us at N = 4096). */
  float* ray_partials;
} oi_composite_grads;

int oi_composite_bwd(const oi_composite_params* fwd, const oi_composite_grads* grads, oi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a16: DC discriminator convolutions.  Replaces nn.Conv2d(4,2,1,bias=False)+LeakyReLU(0.2) blocks
 * and the 4x4 valid head of DCDiscriminator.forward (src/models/discriminator.py:63-85) (cuDNN in the
 * reference).  NCHW fp32; w [Cout][Cin][4][4].  stride/pad: (2,1) for blocks, (1,0) for the head.
 * y = lrelu_slope(conv(x)) when slope != 1 (slope == 1: linear).
 */
int oi_conv4x4_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin,
                   int H, int W, int Cout, int stride, int pad, float slope, oi_stream_t stream);
/* Chain form for forward-only passes:  y = lrelu_slope( conv( lrelu_x_slope(x) ) + bias ).
 *   x_slope != 1: the LeakyReLU of the PRODUCING layer is applied while x is loaded, so a split-K producer can hand
 *     over its pre-activation sums (slope = 1, no bias) and needs no activation pass of its own;
 *   flags & OI_CONV_Y_IS_ZERO: the caller promises that y already holds zeros; the split-K path then skips its zero-fill,
 *     so a chain of layers shares ONE fill of an arena holding all their outputs;
 *   flags & OI_CONV_ANY_SCALE: an operand may be of gradient scale (the R1 double backward runs d loss / d image through
 *     this entry): keeps the call on the fp32 matrix cores -- the large-batch kernel splits operands into UNSCALED fp16
 *     limbs, exact to 2^-22 only for |v| >~ 3e-4. */
#define OI_CONV_Y_IS_ZERO 1
#define OI_CONV_ANY_SCALE 2
int oi_conv4x4_fwd_into(const float* x, const float* w, const float* bias, float* y, int B, int Cin,
                        int H, int W, int Cout, int stride, int pad, float slope, float x_slope,
                        int flags, oi_stream_t stream);
/* The same for the first layer of a chain whose outputs share one arena: `y` is this layer's output at the start of the
 * arena region, and the `zero_tail_floats` floats behind it (from y + round_up(B * Cout * Ho * Wo, 4)) are cleared by the
 * same launch -- the later layers then run with OI_CONV_Y_IS_ZERO and the chain needs no fill launch at all. */
int oi_conv4x4_fwd_arena(const float* x, const float* w, const float* bias, float* y, int B, int Cin,
                         int H, int W, int Cout, int stride, int pad, float slope, float x_slope,
                         int flags, long long zero_tail_floats, oi_stream_t stream);

/* Backward of the convolution (cuDNN bwd-data / bwd-filter in the reference, issued by autograd for
 * discriminator.py:80-83).  g = dL/d(conv output, pre-activation) [B][Cout][Ho][Wo].
 *   dgrad: gx [B][Cin][H][W] = conv_transpose(g, w);  wgrad: gw [Cout][Cin][4][4] = correlate(x, g).
 * {fwd, dgrad, wgrad} is closed under differentiation, which is how the R1 double-backward
 * (src/loss/gan.py:5-14) is served.  oi_lrelu_mask_mul: out = ref > 0 ? v : slope*v (LeakyReLU and
 * its gradient through the saved output); oi_channel_sum: bias gradient. */
int oi_conv4x4_dgrad(const float* g, const float* w, float* gx, int B, int Cin, int H, int W, int Cout,
                     int stride, int pad, oi_stream_t stream);
int oi_conv4x4_wgrad(const float* g, const float* x, float* gw, int B, int Cin, int H, int W, int Cout,
                     int stride, int pad, oi_stream_t stream);
/* The same with the producing layer's LeakyReLU applied to the incoming gradient on load (ref = that layer's forward output,
 * same shape as g: g_eff = ref > 0 ? g : slope g; ref null = plain), and, for the weight gradient, accumulation into a buffer
 * that already holds a gradient or zeros (accumulate != 0: atomic adds, no clearing) -- what autograd otherwise does with one
 * oi_lrelu_mask_mul launch per layer and one `+=` launch per weight and contribution. */
int oi_conv4x4_dgrad_masked(const float* g, const float* ref, float slope, const float* w, float* gx, int B, int Cin, int H,
                            int W, int Cout, int stride, int pad, oi_stream_t stream);
int oi_conv4x4_wgrad_masked(const float* g, const float* ref, float slope, const float* x, float* gw, int accumulate, int B,
                            int Cin, int H, int W, int Cout, int stride, int pad, oi_stream_t stream);
/* Both of the above in ONE launch (the two gradients of a layer share the incoming gradient and nothing else). */
int oi_conv4x4_bwd_masked(const float* g, const float* ref, float slope, const float* w, const float* x, float* gx, float* gw,
                          int accumulate, int B, int Cin, int H, int W, int Cout, int stride, int pad, oi_stream_t stream);
/* The same for a layer fed with the PRE-activation of its predecessor, y = conv(lrelu_{x_slope}(x), w) (the chain form of
 * oi_conv4x4_fwd_into, where a layer's LeakyReLU is applied by its consumer on load -- no activation pass of its own): the
 * weight gradient takes lrelu_{x_slope}(x) on load, the data gradient is returned with respect to x, i.e. multiplied by
 * lrelu'(x) in the kernel's epilogue.  x_slope = 1: plain input.  _dgrad_pre: the data gradient alone (frozen weights). */
int oi_conv4x4_bwd_pre(const float* g, const float* ref, float slope, const float* w, const float* x, float x_slope, float* gx,
                       float* gw, int accumulate, int B, int Cin, int H, int W, int Cout, int stride, int pad,
                       oi_stream_t stream);
int oi_conv4x4_dgrad_pre(const float* g, const float* w, const float* x, float x_slope, float* gx, int B, int Cin, int H, int W,
                         int Cout, int stride, int pad, oi_stream_t stream);
int oi_lrelu_mask_mul(const float* v, const float* ref, float* out, long long n, float slope,
                      oi_stream_t stream);
int oi_channel_sum(const float* g, float* gb, int B, int C, int HW, oi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a18: upfirdn2d.  Same contract as the reference plugin entry
 * upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain)
 * (src/third_party/ada/torch_utils/ops/upfirdn2d.cpp:16-94): x [B*C][H][W], f [fh][fw] fp32,
 * y [B*C][Ho][Wo] with Ho = (H*upy + pady0 + pady1 - fh + downy) / downy (idem Wo).
 */
int oi_upfirdn2d(const float* x, const float* f, float* y, int BC, int H, int W, int fh, int fw,
                 int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1,
                 int flip, float gain, oi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a17/a19: affine_grid + bilinear grid_sample (zeros padding, align_corners=False) fused; the grid
 * is never materialised.  Replaces F.affine_grid + grid_sample_gradfix.grid_sample
 * (src/third_party/ada/augment.py:297-298; grid_sample_gradfix.py:33-66).
 *   x [B][C][Hi][Wi], theta [B][2][3] -> y [B][C][Ho][Wo].
 * The backward wrt x (aten::grid_sampler_2d_backward, grid_sample_gradfix.py:69-85) is the adjoint
 * scatter oi_affine_grid_sample_bwd; its own backward is oi_affine_grid_sample_fwd again.
 */
int oi_affine_grid_sample_fwd(const float* x, const float* theta, float* y, int B, int C, int Hi,
                              int Wi, int Ho, int Wo, oi_stream_t stream);
int oi_affine_grid_sample_bwd(const float* gy, const float* theta, float* gx, int B, int C, int Hi,
                              int Wi, int Ho, int Wo, oi_stream_t stream);

/* Unit light direction in every box frame, and the gradient with respect to the raw direction parameter, one launch each:
 *   n[b] = normalize(w2b[b][:3][:3] normalize(d)),  d [3], w2b [B][4][4], n / g_n [B][3], g_d [3] (summed over the batch)
 * = DirectionalLightWithSpecularFixInit.direction -> batch_direction (src/utils/lighting.py:35-39, 115-119) -> the
 * normalisation in front of the Phong terms (src/models/generator.py:84-100). */
int oi_light_dir_fwd(const float* d, const float* w2b, float* n, int B, oi_stream_t stream);
int oi_light_dir_bwd(const float* d, const float* w2b, const float* g_n, float* g_d, int B, oi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The scalar losses of a GAN training step, one launch each way (csrc/loss.hip).  Replaces GANLoss("bce") =
 * F.binary_cross_entropy_with_logits against a constant target (src/loss/gan.py:39-49), compute_grad2's
 * grad.pow(2).reshape(B, -1).sum(1).mean() (src/loss/gan.py:5-14), PositionLoss("mse") (src/loss/position.py:4-18) and their
 * weighted sum (src/trainers/gan_pose_trainer.py:122-137 generator step, 163-190 discriminator steps).
 *   d_real, d_fake [B][K] discriminator outputs (column 0 = the GAN logit; null = term absent): real = BCE(d_real[:, 0], 1),
 *   fake = BCE(d_fake[:, 0], 0);  pose [B][K-1] (null = absent): aux = mean (d_fake[:, 1:] - pose)^2;  gx [B][N] (null or
 *   N = 0 = absent): reg = mean_b sum_i gx^2;  aux_w: DEVICE scalar (changes per iteration inside a captured step).
 *   out6 = { real + fake + reg_w reg + aux_w aux,  real + fake,  reg,  fake,  real,  aux }
 * _bwd: g_total (device scalar) -> g_real, g_fake [B][K] (unused columns zero), g_gx [B][N]; any of them null = not wanted. */
int oi_gan_losses_fwd(const float* d_real, const float* d_fake, const float* pose, const float* gx, const float* aux_w,
                      float reg_w, float* out6, int B, int K, long long N, oi_stream_t stream);
int oi_gan_losses_bwd(const float* g_total, const float* d_real, const float* d_fake, const float* pose, const float* gx,
                      const float* aux_w, float reg_w, float* g_real, float* g_fake, float* g_gx, int B, int K, long long N,
                      oi_stream_t stream);

/* Zero fill of n_floats floats as a kernel launch on `stream` (the one fill a step's pool of accumulate-outputs gets,
 * oi_outputs_prezeroed_stream; a kernel node, not a memset node: see oi_common.h on memset nodes under hipGraph replay). */
int oi_zero_fill(float* p, long long n_floats, oi_stream_t stream);

/* The inputs of one captured (hipGraph) step in one launch: n_copies <= 4 device-to-device copies of counts[c] floats
 * (srcs / dsts / counts: HOST arrays) and n_imm <= 64 floats `imm` (HOST values, carried in the kernel arguments) written to
 * the device address imm_dst.  Replaces a copy launch per input tensor, the pinned-buffer upload of the augmentation
 * matrices and a fill per scalar in front of every replay (oi_amd.graphed). */
int oi_stage_inputs(const float* const* srcs, float* const* dsts, const long long* counts, int n_copies, const float* imm,
                    int n_imm, float* imm_dst, oi_stream_t stream);

/* The renderer's two derived scalars from the compositing reductions r4[4] (reference: src/models/renderer.py:430-446):
 * out2 = (r4[0] / (r4[1] + 1e-5), r4[2] * inv_nt), and the gradient w.r.t. r4 (g_err / g_surf: device scalars or NULL). */
int oi_render_scalars_fwd(const float* r4, float inv_nt, float* out2, oi_stream_t stream);

/* The scalar glue of a forward in ONE launch (round 4; ~10 tensor-op launches of 0-dim tensors per parameter version before):
 *   out5    = [inv_s, 1 / inv_s, sigmoid(param_ambient), 1 - sigmoid(param_ambient), max(param_specular, 0)]
 *             inv_s = clamp(exp(10 variance), 1e-6, 1e6) (renderer.py:404; `s_val` of :448 is out5[1]); the light colours are the
 *             logging scalars of generator.py:214-222 (lighting.py:50-60 before the expand(3))
 *   packed3 = [param_ambient, param_specular, param_shininess], the `light` block oi_composite_fwd reads.
 * All six pointers: single floats / small arrays on the device. */
int oi_scalar_glue(const float* variance, const float* param_ambient, const float* param_specular, const float* param_shininess,
                   float* out5, float* packed3, oi_stream_t stream);
int oi_render_scalars_bwd(const float* r4, const float* g_err, const float* g_surf, float inv_nt, float* g_r4,
                          oi_stream_t stream);

/* total = sum_i weights[i] * *terms[i] over n <= 8 device scalars, and its gradient g_terms[i] = weights[i] * *g_out: the
 * weighted sum of loss terms of a training step (reference: src/trainers/gan_pose_trainer.py:122-137, one multiply and one
 * add per term composed from tensor ops).  `terms` and `weights` are HOST arrays (n device pointers / n floats). */
int oi_weighted_sum_fwd(const float* const* terms, const float* weights, int n, float* out, oi_stream_t stream);
int oi_weighted_sum_bwd(const float* g_out, const float* weights, int n, float* g_terms, oi_stream_t stream);

/* Outputs that are ACCUMULATED into (the split-K sums of oi_conv4x4_fwd* / oi_conv4x4_wgrad, the scatter-adds of
 * oi_conv4x4_dgrad, oi_affine_grid_sample_bwd, oi_grid_sample_bwd, oi_reflect_pad_bwd) are cleared by their launcher with a
 * fill of their own.  A caller that takes every such output from memory it has already zeroed -- one fill per training step
 * instead of ~90 -- declares so FOR ONE STREAM: oi_outputs_prezeroed_stream(s, 1) ... oi_outputs_prezeroed_stream(s, 0).
 * Only launches on stream `s` skip their fill: a backward pass (issued from the framework's autograd thread, on the forward's
 * stream) is covered, a second thread working on its own stream is not affected (re-entrant, thread-safe;
 * tests/test_gpu_ddp.py::test_prezeroed_declaration_is_per_stream).  Returns the previous setting of that stream (0 / 1), or
 * a negative oi_status when more than 16 streams hold the declaration at once.
 * (oi_amd.ops.ZeroPool does this around a captured discriminator step.) */
int oi_outputs_prezeroed_stream(oi_stream_t stream, int on);

/* First half of oi_ada_geom_fwd alone: canvas [B C][2 (H + my0 + my1)][2 (W + mx0 + mx1)] = reflect pad + x2 up-FIR of x. */
int oi_ada_pad_up2(const float* x, const float* f, float* canvas, int B, int C, int H, int W, int mx0, int mx1, int my0, int my1,
                   oi_stream_t stream);

/* Discriminator forward at batch 1-4 in FOUR launches -- five when the matrices are in device memory or a tile's canvas
 * footprint does not fit the first kernel's LDS tile -- (csrc/disc_small.hip): AugmentPipe's geometric augmentation (given its
 * sampling matrix theta [B][2][3] and reflect margins, as oi_ada_geom_fwd) + the four 4x4 stride-2 LeakyReLU blocks + the 4x4
 * head of DCDiscriminator(img_size 64, n_feat 512) -- reference src/models/discriminator.py:57-85, ada/augment.py:284-301.
 *   x [B][C][64][64];  theta_host: HOST array, passed to the kernel by value (no copy on the stream) | theta_dev: device
 *   array (captured graphs) | both NULL: no augmentation;  f12: Hz_geom (12 taps);  w1..w4 [Cout][Cin][4][4] (C -> 64 -> 128 ->
 *   256 -> 512), whead [out_dim][512][4][4], bhead [out_dim] or NULL;  logits [B][out_dim].
  *   workspace: oi_disc_fwd_small_workspace_floats(...) floats;  ticket: OI_TICKET_WORDS zero-initialised device words (left
 *   zero; not to be shared by launches that may overlap).  Fixed summation order: results are bit-reproducible.
 * OI_ERR_UNSUPPORTED for any other shape (B > 4, other sizes): callers then take the general path. */
size_t oi_disc_fwd_small_workspace_floats(int B, int C, int mx0, int mx1, int my0, int my1);
int oi_disc_fwd_small(const float* x, const float* theta_host, const float* theta_dev, const float* f12, int mx0, int mx1, int my0,
                      int my1, const float* w1, const float* w2, const float* w3, const float* w4, const float* whead,
                      const float* bhead, float* workspace, unsigned* ticket, float* logits, int B, int C, int H, int W, int n_feat,
                      int out_dim, float slope, oi_stream_t stream);

/* The same forward as a PLAN OWNED BY THE LIBRARY (round 4): oi_disc_graph_create stores the arguments of oi_disc_fwd_small
 * (weights, workspace, ticket and logits are fixed addresses from then on; `aug` != 0: with the augmentation at the given STATIC
 * margins; no HIP call is made: an object may be created while its caller captures a stream).  oi_disc_graph_launch captures
 * oi_disc_fwd_small once (first call; private stream) and replays that hipGraph on `stream` for a new image pointer `x` and a new
 * HOST array of sampling matrices theta_host [B][2][3] (NULL iff aug == 0): both reach the kernels as updated kernel-node
 * parameters (hipGraphExecKernelNodeSetParams) -- no staging copy, no upload.  Results: bit-identical to oi_disc_fwd_small.
 * One graph object serves one stream at a time (its workspace and logits are shared by its launches, which the stream orders).
 * Re-create after anything that moves a weight tensor (in-place optimiser updates keep the addresses). */
typedef struct oi_disc_graph oi_disc_graph;
int oi_disc_graph_create(oi_disc_graph** out, int aug, const float* f12, int mx0, int mx1, int my0, int my1, const float* w1,
                         const float* w2, const float* w3, const float* w4, const float* whead, const float* bhead,
                         float* workspace, unsigned* ticket, float* logits, int B, int C, int H, int W, int n_feat, int out_dim,
                         float slope);
int oi_disc_graph_launch(oi_disc_graph* g, const float* x, const float* theta_host, oi_stream_t stream);
/* The shipped discriminators (configs/train.yaml:78-102: img_size 128 = FIVE blocks C -> 32 -> 64 -> 128 -> 256 -> 512 -> out_dim) the
 * same way (round 6): w0 [32][C][4][4] rides on the augmentation kernel, w1 [64][32][4][4] has a kernel of its own (32 input
 * channels: arithmetic, not a weight stream), w2..w4 + head are the 64 x 64 network's conv 2..4: five launches (six with the canvas
 * in memory).  x [B][C][128][128]; workspace: oi_disc_fwd_small128_workspace_floats.  The plan form is launched launch by launch
 * only (oi_disc_graph_launch_eager / _launch_ada with eager != 0; oi_disc_graph_launch returns OI_ERR_UNSUPPORTED for it). */
size_t oi_disc_fwd_small128_workspace_floats(int B, int C, int mx0, int mx1, int my0, int my1);
int oi_disc_fwd_small128(const float* x, const float* theta_host, const float* theta_dev, const float* f12, int mx0, int mx1, int my0,
                         int my1, const float* w0, const float* w1, const float* w2, const float* w3, const float* w4, const float* whead,
                         const float* bhead, float* workspace, unsigned* ticket, float* logits, int B, int C, int n_feat, int out_dim,
                         float slope, oi_stream_t stream);
int oi_disc_graph_create128(oi_disc_graph** out, int aug, const float* f12, int mx0, int mx1, int my0, int my1, const float* w0,
                            const float* w1, const float* w2, const float* w3, const float* w4, const float* whead, const float* bhead,
                            float* workspace, unsigned* ticket, float* logits, int B, int C, int n_feat, int out_dim, float slope);
/* the same launches issued one by one (no graph replay) from the object's stored arguments; `logits`: where this call's
 * [B][out_dim] result goes (NULL: the buffer given at creation) */
int oi_disc_graph_launch_eager(oi_disc_graph* g, const float* x, const float* theta_host, float* logits, oi_stream_t stream);
/* AugmentPipe's parameter draws (src/third_party/ada/augment.py:213-230: integer translation `xint`, isotropic `scale`) and the
 * sampling matrix (augment.py:285-297) formed INSIDE the library from one 64-bit seed per forward (splitmix64 counter stream; per
 * image: t_x, t_y, the translation gate, the scale's normal, the scale gate).  p_xint = xint * p, p_scale = scale * p (the gates'
 * probabilities).  theta_host [B][2][3] (HOST memory); ts_host [B][3] = (t_x, t_y, s) as drawn, or NULL.  No HIP call.
 * oi_disc_graph_launch_ada = these draws at the plan's static margins + oi_disc_graph_launch_eager (eager != 0; `logits` as there)
 * or oi_disc_graph_launch (eager == 0; logits must be NULL): the reference's own call, ADADiscriminator.forward
 * (src/models/discriminator.py:98-100), without numpy or Python arithmetic on the way. */
int oi_ada_theta_xint_scale(unsigned long long seed, int B, int H, int W, int mx0, int mx1, int my0, int my1, float p_xint,
                            float xint_max, float p_scale, float scale_std, float* theta_host, float* ts_host);
int oi_disc_graph_launch_ada(oi_disc_graph* g, const float* x, unsigned long long seed, float p_xint, float xint_max, float p_scale,
                             float scale_std, float* logits, int eager, oi_stream_t stream);
void oi_disc_graph_destroy(oi_disc_graph* g);

/* The whole geometric augmentation of AugmentPipe.forward (src/third_party/ada/augment.py:284-301) for a given sampling
 * grid, in two launches: reflect pad (margins mx0, mx1, my0, my1) + x2 up-FIR | affine bilinear resample + /2 down-FIR.
 *   x [B][C][H][W], theta [B][2][3] (the matrix F.affine_grid receives, augment.py:297), f: the 12 taps of Hz_geom
 *   -> y [B][C][H][W];  canvas: working memory of B * C * 2 (H + my0 + my1) * 2 (W + mx0 + mx1) floats.
 * Equal to oi_reflect_pad_fwd -> oi_upfirdn2d (up 2, x then y) -> oi_affine_grid_sample_fwd -> oi_upfirdn2d (down 2, x then
 * y) up to fp32 summation order; linear in x (the adjoint is the chain of the four adjoint entries). */
int oi_ada_geom_fwd(const float* x, const float* theta, const float* f, float* y, float* canvas, int B, int C, int H,
                    int W, int mx0, int mx1, int my0, int my1, oi_stream_t stream);

/* The same map for AXIS-ALIGNED sampling matrices (theta[b][0][1] == theta[b][1][0] == 0 for every b: translations, isotropic /
 * anisotropic scales, flips -- AugmentPipe without rotate / rotate90; the CALLER guarantees it, the kernel does not read those
 * two entries), in ONE launch and without a canvas: the four stages are then separable, y_c = A_y x_c A_x^T with two H x H
 * matrices per image that a workgroup builds in LDS (csrc/disc.hip, ada_sep_kernel).  Equal to oi_ada_geom_fwd up to fp32
 * summation order (not bit-identical to it; deterministic).  Covered shapes: oi_ada_geom_sep_supported(C, H, W) != 0
 * (1..3 channels of 64 x 64 or 128 x 128; at 128 the matrices are accumulated in 32-bit fixed point); anything else is
 * OI_ERR_INVALID_ARG -- use oi_ada_geom_fwd.
 * The matrices come from the device (theta) or from the HOST (theta_host, [B][2][3]: passed to the kernel by value, 64 images
 * per launch -- a caller that drew them on the host, e.g. with oi_ada_theta_xint_scale, uploads nothing); exactly one of the two. */
int oi_ada_geom_sep_supported(int C, int H, int W);
int oi_ada_geom_sep_fwd(const float* x, const float* theta, const float* theta_host, const float* f, float* y, int B, int C, int H,
                        int W, int mx0, int mx1, int my0, int my1, oi_stream_t stream);
/* The ADJOINT of that map, gx_c = A_y^T gy_c A_x (same arguments, same kernel with the matrices stored the other way round): the
 * gradient of a loss with respect to the un-augmented images, one launch instead of the six adjoint stages
 * (oi_upfirdn2d x 4, oi_affine_grid_sample_bwd, oi_reflect_pad_bwd); gx is written, not accumulated. */
int oi_ada_geom_sep_adj(const float* gy, const float* theta, const float* theta_host, const float* f, float* gx, int B, int C, int H,
                        int W, int mx0, int mx1, int my0, int my1, oi_stream_t stream);

/* Stand-alone plugin ops, for a caller that keeps the reference's own Python layers and only swaps the compiled ops
 * (INTEGRATION.md 3).
 * oi_fused_bias_act = fused_bias_act(input, bias, refer, act, grad, alpha, scale) of stylesdf/op/fused_bias_act.cpp:11-20
 * (kernel fused_bias_act_kernel.cu:18-49): out[i] = f(x[i] + bias[(i / step_b) % size_b]) * scale with act 1 = linear,
 * 3 = leaky relu (slope alpha); grad 0 = value, 1 = first derivative gated by sign(refer), 2 = second derivative (0).
 * bias / refer may be NULL (the reference passes empty tensors); step_b = product of the dimensions after the channel. */
int oi_fused_bias_act(float* out, const float* x, const float* bias, const float* refer, int act, int grad, float alpha,
                      float scale, long long size_x, long long step_b, int size_b, oi_stream_t stream);
/* grid_sample(input, grid) of ada/torch_utils/ops/grid_sample_gradfix.py:33-66: bilinear, zeros padding,
 * align_corners = False.  x [N][C][Hi][Wi], grid [N][Ho][Wo][2] (x, y in [-1, 1]), y [N][C][Ho][Wo].
 * oi_grid_sample_bwd is the aten::grid_sampler_2d_backward that wrapper calls: gx [N][C][Hi][Wi] (assigned; NULL to
 * skip) and ggrid [N][Ho][Wo][2] (assigned; NULL to skip: the output_mask of the reference call). */
int oi_grid_sample_fwd(const float* x, const float* grid, float* y, int N, int C, int Hi, int Wi, int Ho, int Wo,
                       oi_stream_t stream);
int oi_grid_sample_bwd(const float* gy, const float* x, const float* grid, float* gx, float* ggrid, int N, int C, int Hi,
                       int Wi, int Ho, int Wo, oi_stream_t stream);

/* Reflect padding (torch.nn.functional.pad(mode='reflect'), augment.py:286) and its adjoint. */
int oi_reflect_pad_fwd(const float* x, float* y, int BC, int H, int W, int px0, int px1, int py0,
                       int py1, oi_stream_t stream);
int oi_reflect_pad_bwd(const float* gy, float* gx, int BC, int H, int W, int px0, int px1, int py0,
                       int py1, oi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * multi-tensor optimiser / EMA steps -- one launch per network per step.
 * Replaces the torch.optim.Adam / torch.optim.RMSprop steps of the reference's training loop
 * (configs/train.yaml:133-147; src/trainers/gan_pose_trainer.py:142,191) and the EMA lerp
 * `p_ema.copy_(p.lerp(p_ema, beta))` (src/utils/ema.py:26-32).
 * `table` is a DEVICE array of n_chunks descriptors; each covers at most oi_mt_chunk_elems() contiguous fp32
 * elements of one parameter: p = parameter (EMA: the averaged copy), g = gradient (EMA: the live parameter),
 * s0/s1 = optimiser state (Adam: exp_avg / exp_avg_sq; RMSprop: square_avg / unused; EMA: unused).
 * Formulas (fp32, as torch's single-tensor path, no weight decay / amsgrad / momentum / centering):
 *   Adam:    m = lerp(m, g, 1-b1); v = b2 v + (1-b2) g^2; p -= lr/bc1 * m / (sqrt(v)/bc2_sqrt + eps)
 *   RMSprop: s = alpha s + (1-alpha) g^2; p -= lr * g / (sqrt(s) + eps)
 *   EMA:     p = lerp(g, p, beta)
 * bias_correction1 = 1 - b1^step, bias_correction2_sqrt = sqrt(1 - b2^step) are computed by the caller. */
typedef struct oi_mt_chunk {
  float* p;
  const float* g;
  float* s0;
  float* s1;
  int n;
  int reserved;
} oi_mt_chunk;
int oi_mt_chunk_elems(void);
int oi_multi_adam(const oi_mt_chunk* table, int n_chunks, float lr, float beta1, float beta2, float eps,
                  float bias_correction1, float bias_correction2_sqrt, oi_stream_t stream);
int oi_multi_rmsprop(const oi_mt_chunk* table, int n_chunks, float lr, float alpha, float eps, oi_stream_t stream);
int oi_multi_lerp(const oi_mt_chunk* table, int n_chunks, float beta, oi_stream_t stream);
/* p <- g for every chunk: many small tensors gathered into (slices of) one buffer in ONE launch -- the stacked parameter
 * layouts of oi_film_params / oi_mlp_pack_weights after an optimiser step (oi_amd.params.StackCache; the reference's modules
 * hold one nn.Parameter per layer, fields.py:49-77, and torch.stack would be one launch per stacked array). */
int oi_multi_copy(const oi_mt_chunk* table, int n_chunks, oi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a16 at batch >= 16 without gradient (round 5): DCDiscriminator.forward (src/models/discriminator.py:57-85) as GEMMs whose
 * operands are prepared once -- activations as NHWC fp16 limb planes (hi + lo = 22 bits) written by the producing layer, weights
 * as fp16 limb images in MFMA fragment order (packed per parameter version) -- three fp16 MFMAs per product, fp32 accumulation,
 * split-K partial planes added in a fixed order (no atomics: bit-reproducible).  csrc/disc_large.hip.
 *   chans [n_blocks + 1]: in_dim (<= 4), then the output channels of every 4x4 stride-2 block; covered when chans[1] % 32 == 0
 *   and chans[1] <= 128, every later block has Cin % 64 == 0 and Cout % 128 == 0, and H / 2^n_blocks == 4 (the 64 x 64 / n_feat 512 network of
 *   configs/train.yaml at 64 x 64: {3 | 1, 64, 128, 256, 512}).  oi_disc_large_packed_bytes / _workspace_bytes return 0 and the
 *   other two OI_ERR_UNSUPPORTED for anything else (the caller keeps the general chain).
 *   w_blocks: HOST array of n_blocks device pointers ([Cout][Cin][4][4] each, the reference's layout); w_head [out_dim][C][4][4].
 *   x [B][chans[0]][H][H] (after the augmentation), bhead [out_dim] or NULL -> logits [B][out_dim].  w1 is IGNORED (kept in the
 *   signature for ABI stability: layer 1 is read from the packed image like every other layer; pass w_blocks[0] or NULL). */
size_t oi_disc_large_packed_bytes(const int* chans, int n_blocks, int out_dim);
int oi_disc_large_pack(const float* const* w_blocks, const float* w_head, const int* chans, int n_blocks, int out_dim, void* packed,
                       oi_stream_t stream);
size_t oi_disc_large_workspace_bytes(const int* chans, int n_blocks, int out_dim, int B, int H);
int oi_disc_fwd_large(const float* x, const float* w1, const void* packed, const float* bhead, void* workspace, size_t workspace_bytes,
                      float* logits, const int* chans, int n_blocks, int out_dim, int B, int H, float slope, oi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* OI_HIP_H_ */
