"""Thin functional wrappers: torch tensors in/out, HIP kernels underneath (no autograd here).

Every function allocates its outputs with the torch caching allocator, passes raw device
pointers + the *current* HIP stream to liboi_hip.so, and returns immediately (stream ordered).
Autograd structure lives in `oi_amd.autograd`."""
import ctypes
import os

import numpy as np
import torch

from . import lib as _l

_vp = ctypes.c_void_p


def _stream():
    """Raw hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() builds a Python
    Stream object (~8 us); the two C accessors below cost ~0.3 us and this runs once per kernel launch."""
    return _vp(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _l.OiHipError("oi_amd ops need CUDA/HIP tensors (there is no CPU path)")
    if t.dtype not in (torch.float32, torch.uint8):
        raise _l.OiHipError(f"expected float32 tensor, got {t.dtype}")
    if not t.is_contiguous():
        raise _l.OiHipError("expected a contiguous tensor")
    return _vp(t.data_ptr())


def _c(t):
    return t.contiguous().float() if t is not None else None


def _new(ref, *shape):
    return torch.empty(shape, dtype=torch.float32, device=ref.device)


class ZeroPool:
    """Pre-zeroed memory for the accumulate-outputs (split-K sums, scatter-adds) of ONE captured step: a single fill at
    the start of the step instead of a fill launch in front of every such op (93 of 631 launches of a training iteration
    were fills).  Bump allocation in call order; while active, the library is told not to clear those outputs itself
    (oi_outputs_prezeroed_stream, for this stream only).  The first pass only measures (every request is served by its own torch.zeros); the buffer is
    allocated at the next `begin()`.  Slices stay valid until the next `begin()` -- the owner (oi_amd.graphed.GraphedDStep)
    guarantees that nothing outlives its step except what the optimiser reads before the next one starts."""

    _active = {}   # raw stream handle -> the pool that is active for launches on that stream

    def __init__(self):
        self.buf, self.off, self.need = None, 0, 0

    def begin(self, device, must_not_alias=()):
        """Start of a step: (re)allocate if the last pass asked for more, then ONE fill of what the step will hand out.
        `must_not_alias`: parameters whose `.grad` must not live in the pool any more (a gradient kept across steps --
        gradient accumulation, zero_grad(set_to_none=False) -- would be zeroed under its owner: fail loudly instead)."""
        if self.buf is not None and must_not_alias:
            lo, hi = self.buf.data_ptr(), self.buf.data_ptr() + self.buf.numel() * 4
            for p_ in must_not_alias:
                g_ = p_.grad
                if g_ is not None and lo <= g_.data_ptr() < hi:
                    raise RuntimeError("ZeroPool.begin: a parameter's .grad still points into the pool of the previous step; call "
                                       "zero_grad(set_to_none=True) before the step (or clone gradients that must outlive it)")
        if self.buf is None or self.buf.numel() < self.need:
            self.buf = torch.empty(self.need, dtype=torch.float32, device=device) if self.need else None
        self.off, self.need = 0, 0
        if self.buf is not None:
            _l.check(_l.load().oi_zero_fill(_p(self.buf), self.buf.numel(), _stream()), "oi_zero_fill")

    def take(self, ref, shape):
        n = int(torch.Size(shape).numel())
        n4 = (n + 63) // 64 * 64       # 256-byte granules
        self.need += n4
        if self.buf is None or self.off + n4 > self.buf.numel():
            dev = ref if isinstance(ref, torch.device) else ref.device
            return torch.zeros(shape, dtype=torch.float32, device=dev)   # measuring pass / overflow: its own fill
        out = self.buf[self.off:self.off + n].view(shape)
        self.off += n4
        return out

    def __enter__(self):
        # The declaration belongs to the CURRENT STREAM (oi_outputs_prezeroed_stream): the ops of this step -- forward here,
        # backward on the autograd thread but on the same stream -- take pool memory and skip their fills; any other thread /
        # stream keeps plain torch.empty outputs that the library clears itself.
        self._key = _stream().value or 0
        assert self._key not in ZeroPool._active, "nested ZeroPool on one stream"
        was = _l.load().oi_outputs_prezeroed_stream(_vp(self._key), 1)
        if was < 0:
            _l.check(was, "oi_outputs_prezeroed_stream")
        self._was = was
        ZeroPool._active[self._key] = self
        return self

    def __exit__(self, *exc):
        _l.load().oi_outputs_prezeroed_stream(_vp(self._key), self._was)
        del ZeroPool._active[self._key]
        return False


def _active_pool():
    if not ZeroPool._active:
        return None
    return ZeroPool._active.get(_stream().value or 0)


def _new_acc(ref, *shape):
    """Output buffer of an op that ACCUMULATES into it: plain memory (the launcher clears it) unless a ZeroPool is active."""
    pool = _active_pool()
    return torch.empty(shape, dtype=torch.float32, device=ref.device) if pool is None else pool.take(ref, shape)


def _zeros_split(dev, *shapes):
    """Several zero-initialised fp32 tensors from ONE fill launch (views of one flat buffer, each 16-byte aligned) -- or
    from the step's ZeroPool when one is active (no launch)."""
    pool = _active_pool()
    if pool is not None:
        return [pool.take(dev, sh) for sh in shapes]
    sizes = [int(torch.Size(sh).numel()) for sh in shapes]
    offs, tot = [], 0
    for n in sizes:
        offs.append(tot)
        tot += (n + 3) // 4 * 4
    flat = torch.zeros(tot, dtype=torch.float32, device=dev)
    return [flat[o:o + n].view(sh) for o, n, sh in zip(offs, sizes, shapes)]


# ------------------------------------------------------------------------------------------
# a6 stand-alone: the albedo head on caller-supplied features (csrc/color_head.hip)
# ------------------------------------------------------------------------------------------

def color_head_fwd(feat, normals, gamma, beta, wv, bv, wrgb, brgb, B):
    """feat (n,128), normals (n,3), gamma / beta (B,128) -> rgb (n,3); n = B * points per element (element-major rows)."""
    n = feat.shape[0]
    assert n % B == 0 and feat.shape[1] == 128 and normals.shape == (n, 3) and gamma.shape == (B, 128) == beta.shape
    rgb = _new(feat, n, 3)
    _l.check(_l.load().oi_color_head_fwd(_p(feat), _p(normals), _p(gamma), _p(beta), 128, _p(wv), _p(bv), _p(wrgb), _p(brgb),
                                         _p(rgb), B, n // B, _stream()), "oi_color_head_fwd")
    return rgb


def color_head_bwd(feat, normals, gamma, beta, wv, bv, wrgb, brgb, g_rgb, B):
    """-> d_feat, d_normals, d_gamma, d_beta, d_wv, d_bv, d_wrgb, d_brgb (all assigned; fixed summation order)."""
    L = _l.load()
    n = feat.shape[0]
    ws_bytes = int(L.oi_color_head_bwd_workspace_bytes(B, n // B))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=feat.device)
    d_feat, d_normals = _new(feat, n, 128), _new(feat, n, 3)
    d_gamma, d_beta = _new(feat, B, 128), _new(feat, B, 128)
    d_wv, d_bv, d_wrgb, d_brgb = _new(feat, 128, 131), _new(feat, 128), _new(feat, 3, 128), _new(feat, 3)
    _l.check(L.oi_color_head_bwd(_p(feat), _p(normals), _p(gamma), _p(beta), 128, _p(wv), _p(bv), _p(wrgb), _p(brgb), _p(g_rgb),
                                 _p(d_feat), _p(d_normals), _p(d_gamma), _p(d_beta), 128, _p(d_wv), _p(d_bv), _p(d_wrgb),
                                 _p(d_brgb), _p(ws), ws_bytes, B, n // B, _stream()), "oi_color_head_bwd")
    return d_feat, d_normals, d_gamma, d_beta, d_wv, d_bv, d_wrgb, d_brgb


# ------------------------------------------------------------------------------------------
# MLP
# ------------------------------------------------------------------------------------------

def film_params(style_w, style_b, gw, gb, bw, bb, z=None, w=None):
    """(w, gamma[B,NL,128], beta[B,NL,128]); give z (style MLP runs) or w."""
    L = _l.load()
    assert (z is None) != (w is None)
    src = z if z is not None else w
    B, NL = src.shape[0], (gw.shape[0] if gw is not None else 0)
    w_out = _new(src, B, 64) if w is None else _c(w)
    gamma, beta = _new(src, B, max(NL, 1), 128), _new(src, B, max(NL, 1), 128)
    z_ = _c(z)
    args = [_c(style_w), _c(style_b), z_, w_out, _c(gw), _c(gb), _c(bw), _c(bb)]
    _l.check(L.oi_film_params(*[_p(a) for a in args], _p(gamma), _p(beta), B, NL, _stream()), "oi_film_params")
    return w_out, gamma, beta


def prep_render(b2w, w2b, c2b, offs, bg, kinv, R, S, jitter, light_direction, film_P, z, jitter_normal=False, f3_packed=None,
                f3_blob=None):
    """ONE launch for everything a render needs before its first MLP pass (oi_prep_render): the pose block (numpy, host) goes
    BY VALUE in the kernel arguments; rays, near / far, light direction, coarse samples + points and the style MLP + FiLM
    parameters come back.  -> dict(pose (53 B: b2w | w2b | c2b | offs | bg blocks), rays_o, rays_d, near, far, light_dir, z_coarse, pts_coarse, w, gamma, beta)."""
    L = _l.load()
    B = b2w.shape[0]
    assert B <= _l.PREP_MAX_B and z is not None
    dev = z.device
    P = _l.PrepParams()
    for name, arr, n in (("b2w", b2w, 16), ("w2b", w2b, 16), ("c2b", c2b, 16), ("offs", offs, 2), ("bg", bg, 3)):
        a = np.ascontiguousarray(arr, dtype=np.float32).reshape(B, n)
        dst = np.frombuffer(getattr(P, name), dtype=np.float32).reshape(_l.PREP_MAX_B, n)
        dst[:B] = a
    NL = film_P["gw"].shape[0]
    N = B * R * R
    f = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
    out = {"pose": f(B * 53), "rays_o": f(B, R, R, 3), "rays_d": f(B, R, R, 3), "near": f(N, 1), "far": f(N, 1),
           "light_dir": f(B, 3), "z_coarse": f(N, S), "pts_coarse": f(N, S, 3), "w": f(B, 64), "gamma": f(B, NL, 128),
           "beta": f(B, NL, 128)}
    P.B, P.R, P.S, P.NL = B, int(R), int(S), NL
    keep = [_c(kinv), _c(light_direction), _c(jitter), _c(z)] + [_c(film_P[k]) for k in ("style_w", "style_b", "gw", "gb", "bw", "bb")]
    P.kinv, P.light_direction, P.jitter, P.z = (_p(t) for t in keep[:4])
    P.style_w, P.style_b, P.gw, P.gb, P.bw, P.bb = (_p(t) for t in keep[4:])
    P.pose_out, P.rays_o, P.rays_d, P.near_, P.far_ = (_p(out[k]) for k in ("pose", "rays_o", "rays_d", "near", "far"))
    P.light_dir, P.z_coarse, P.pts_coarse = _p(out["light_dir"]), _p(out["z_coarse"]), _p(out["pts_coarse"])
    P.w_out, P.gamma, P.beta = _p(out["w"]), _p(out["gamma"]), _p(out["beta"])
    P.jitter_normal = int(bool(jitter_normal))
    if f3_blob is not None:   # the f16x3 kernel's per-element blobs, formed by the FiLM workgroups (oi_sdf_mlp_fwd_ex / OI_MLP_BLOB_READY)
        P.f3_packed, P.f3_blob = _p(f3_packed), _p(f3_blob)
    _l.check(L.oi_prep_render(ctypes.byref(P), _stream()), "oi_prep_render")
    return out


def film_params_bwd(d_gamma, d_beta, w, gw, bw, style_w=None, style_b=None, z=None, d_w_in=None, want_dz=False):
    """Backward of film_params -> dict(d_gw, d_gb, d_bw, d_bb, d_w [, d_style_w, d_style_b, d_z])."""
    L = _l.load()
    B, NL = d_gamma.shape[0], d_gamma.shape[1]
    d_gamma, d_beta, w = _c(d_gamma), _c(d_beta), _c(w)
    out = {"d_gw": torch.empty_like(gw), "d_gb": _new(w, NL, 128), "d_bw": torch.empty_like(bw), "d_bb": _new(w, NL, 128),
           "d_w": _zeros_split(w.device, w.shape)[0] if d_w_in is None else d_w_in.contiguous().clone()}
    if z is not None:
        out["d_style_w"], out["d_style_b"] = _zeros_split(w.device, style_w.shape, style_b.shape)
        if want_dz:
            out["d_z"] = torch.empty_like(z)
    _l.check(L.oi_film_params_bwd(_p(d_gamma), _p(d_beta), _p(w), _p(_c(gw)), _p(_c(bw)), _p(out["d_gw"]), _p(out["d_gb"]),
                                  _p(out["d_bw"]), _p(out["d_bb"]), _p(out["d_w"]), _p(_c(style_w)), _p(_c(style_b)),
                                  _p(_c(z)), _p(out.get("d_style_w")), _p(out.get("d_style_b")), _p(out.get("d_z")), B, NL,
                                  _stream()), "oi_film_params_bwd")
    return out


def style_bwd(d_w, style_w, style_b, z, want_dz=False):
    """Backward of the style MLP alone (oi_film_params_bwd with no FiLM layer) -> dict(d_style_w, d_style_b [, d_z])."""
    L = _l.load()
    B = z.shape[0]
    out = dict(zip(("d_style_w", "d_style_b"), _zeros_split(z.device, style_w.shape, style_b.shape)))
    if want_dz:
        out["d_z"] = torch.empty_like(z)
    d_w = d_w.contiguous().clone()
    _l.check(L.oi_film_params_bwd(None, None, None, None, None, None, None, None, None, _p(d_w), _p(_c(style_w)),
                                  _p(_c(style_b)), _p(_c(z)), _p(out["d_style_w"]), _p(out["d_style_b"]), _p(out.get("d_z")),
                                  B, 0, _stream()), "oi_film_params_bwd")
    return out


def mlp_pack_weights(w0, b0, wh, bh, wsig, bsig, wv, bv, wrgb, brgb, prec):
    L = _l.load()
    packed = torch.empty(L.oi_mlp_packed_bytes(prec), dtype=torch.uint8, device=w0.device)
    args = [_c(t) for t in (w0, b0, wh, bh, wsig.reshape(-1), bsig.reshape(-1), wv, bv, wrgb, brgb)]
    _l.check(L.oi_mlp_pack_weights(*[_p(a) for a in args], _p(packed), prec, _stream()), "oi_mlp_pack_weights")
    return packed


def mlp_pack_status(packed):
    """Raises (OI_ERR_UNSUPPORTED) if the packed image was built from an inf / NaN weight; synchronises the stream."""
    _l.check(_l.load().oi_mlp_pack_status(_p(packed), _stream()), "oi_mlp_pack_status")


def mlp_scratch_bytes(B, n_per_elem, prec=None):
    L = _l.load()
    return L.oi_mlp_scratch_bytes(B, n_per_elem) if prec is None else L.oi_mlp_scratch_bytes_prec(B, n_per_elem, prec)


def sdf_mlp_fwd(pts, packed, gamma, beta, B, prec, fast_trig=False, want_grad=False, want_rgb=False,
                want_feat=False, scratch=None, blob_ready=False):
    """pts (B*n, 3) -> sdf (B*n,), grad (B*n,3)|None, rgb (B*n,3)|None, feat (B*n,128)|None, scratch.
    blob_ready: `scratch` (from f3_scratch_for) already holds the per-element blobs of the f16x3 kernel, written by
    prep_render for these gamma / beta (oi_sdf_mlp_fwd_ex, OI_MLP_BLOB_READY)."""
    L = _l.load()
    pts = _c(pts)
    n_tot = pts.shape[0]
    assert n_tot % B == 0
    n = n_tot // B
    sdf = _new(pts, n_tot)
    grad = _new(pts, n_tot, 3) if want_grad else None
    rgb = _new(pts, n_tot, 3) if want_rgb else None
    feat = _new(pts, n_tot, 128) if want_feat else None
    if want_grad and scratch is None:
        scratch = torch.empty(L.oi_mlp_scratch_bytes_prec(B, n, prec), dtype=torch.uint8, device=pts.device)
    assert not want_rgb or want_grad
    if blob_ready:
        # (== : the blob's offset inside the scratch depends on the point count the scratch was sized for -- f3_scratch_for(B, n))
        assert want_grad and prec == _l.OI_PREC_F16X3 and scratch.numel() == L.oi_mlp_scratch_bytes_prec(B, n, prec), \
            "blob_ready: the scratch was not made by f3_scratch_for for this (B, n)"
        _l.check(L.oi_sdf_mlp_fwd_ex(_p(pts), _p(packed), _p(gamma), _p(beta), _p(sdf), _p(grad), _p(rgb), _p(feat), _p(scratch), B, n,
                                     prec, int(bool(fast_trig)), _l.OI_MLP_BLOB_READY, _stream()), "oi_sdf_mlp_fwd_ex")
        return sdf, grad, rgb, feat, scratch
    _l.check(L.oi_sdf_mlp_fwd(_p(pts), _p(packed), _p(gamma), _p(beta), _p(sdf), _p(grad), _p(rgb), _p(feat),
                              _p(scratch) if want_grad else None, B, n, prec, int(bool(fast_trig)), _stream()),
             "oi_sdf_mlp_fwd")
    return sdf, grad, rgb, feat, scratch


def f3_scratch_for(B, n, device):
    """(scratch, blob view) of the f16x3 gradient pass over n points per element: the blob view is where oi_prep_render writes the
    per-element blobs (its f3_blob field) that OI_MLP_BLOB_READY then promises."""
    L = _l.load()
    scratch = torch.empty(L.oi_mlp_scratch_bytes_prec(B, n, _l.OI_PREC_F16X3), dtype=torch.uint8, device=device)
    off = L.oi_mlp_f3_blob_offset(B, n)
    return scratch, scratch[off:off + B * L.oi_mlp_f3_blob_bytes()]


def bwd_scratch_cap_bytes():
    """Upper bound of the MLP backward's working memory (OI_BWD_SCRATCH_MB, default 9216: the C2 training render fits one chunk): larger problems run in chunks."""
    return int(float(os.environ.get("OI_BWD_SCRATCH_MB", "9216")) * (1 << 20))


def sdf_mlp_bwd(pts, packed, gamma, beta, grad_fwd, rgb_fwd, feat_fwd, g_sdf, g_grad, g_rgb, B, prec, fast_trig=False,
                scratch_cap=None, g_feat=None):
    """-> d_small (flat), d_wmat (8,128,128), d_gamma (B,9,128), d_beta (B,9,128).  Working memory is bounded by
    `scratch_cap` bytes (default bwd_scratch_cap_bytes()); the library processes the points in chunks that fit.
    `g_feat` (n,128): upstream gradient of the feature output (oi_sdf_mlp_bwd_feat; None = the fused-path kernel)."""
    L = _l.load()
    pts = _c(pts)
    n = pts.shape[0] // B
    dev = pts.device
    d_small, d_wmat, d_gamma, d_beta = _zeros_split(dev, (L.oi_mlp_bwd_small_floats(),), (8, 128, 128), (B, 9, 128),
                                                    (B, 9, 128))
    nbytes = L.oi_mlp_bwd_scratch_bytes_capped(B, n, bwd_scratch_cap_bytes() if scratch_cap is None else int(scratch_cap))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    args = [_c(t) for t in (grad_fwd, rgb_fwd, feat_fwd, g_sdf, g_grad, g_rgb)]
    if g_feat is None:
        _l.check(L.oi_sdf_mlp_bwd(_p(pts), _p(packed), _p(_c(gamma)), _p(_c(beta)), *[_p(a) for a in args], _p(d_small),
                                  _p(d_wmat), _p(d_gamma), _p(d_beta), _p(scratch), nbytes, B, n, prec, int(bool(fast_trig)),
                                  _stream()), "oi_sdf_mlp_bwd")
    else:
        _l.check(L.oi_sdf_mlp_bwd_feat(_p(pts), _p(packed), _p(_c(gamma)), _p(_c(beta)), *[_p(a) for a in args], _p(_c(g_feat)),
                                       _p(d_small), _p(d_wmat), _p(d_gamma), _p(d_beta), _p(scratch), nbytes, B, n, prec,
                                       int(bool(fast_trig)), _stream()), "oi_sdf_mlp_bwd_feat")
    return d_small, d_wmat, d_gamma, d_beta


# ------------------------------------------------------------------------------------------
# rays / sampling / compositing
# ------------------------------------------------------------------------------------------

def gen_rays(c2b, kinv3, offs, R, w2b=None, light_direction=None):
    """-> rays_o, rays_d, near, far [, light_dir (B, 3) when `w2b` and the raw `light_direction` parameter are given]"""
    L = _l.load()
    B = c2b.shape[0]
    ro, rd = _new(c2b, B, R, R, 3), _new(c2b, B, R, R, 3)
    near, far = _new(c2b, B * R * R, 1), _new(c2b, B * R * R, 1)
    if w2b is None:
        _l.check(L.oi_gen_rays(_p(_c(c2b)), _p(_c(kinv3)), _p(_c(offs)), B, R, _p(ro), _p(rd), _p(near), _p(far),
                               _stream()), "oi_gen_rays")
        return ro, rd, near, far
    ldir = _new(c2b, B, 3)
    _l.check(L.oi_gen_rays_light(_p(_c(c2b)), _p(_c(kinv3)), _p(_c(offs)), B, R, _p(ro), _p(rd), _p(near), _p(far),
                                 _p(_c(w2b)), _p(_c(light_direction.detach())), _p(ldir), _stream()), "oi_gen_rays_light")
    return ro, rd, near, far, ldir


def coarse_samples(rays_o, rays_d, near, far, S, jitter=None):
    L = _l.load()
    N = rays_o.shape[0]
    z, pts = _new(rays_o, N, S), _new(rays_o, N, S, 3)
    _l.check(L.oi_coarse_samples(_p(_c(rays_o)), _p(_c(rays_d)), _p(_c(near)), _p(_c(far)), _p(_c(jitter)), N, S,
                                 _p(z), _p(pts), _stream()), "oi_coarse_samples")
    return z, pts


def upsample(rays_o, rays_d, z, sdf, n_new, inv_s, merge=True, mid_last_dist=None):
    """-> z_new, pts_new, z_merged [, (dists, mid_z, pts_mid) when `mid_last_dist` is given: the section mid-points of the
    merged list from the same launch (oi_upsample_mid = oi_upsample + oi_midpoints, bit-identical)]."""
    L = _l.load()
    N, Sc = z.shape
    z_new, pts_new = _new(z, N, n_new), _new(z, N, n_new, 3)
    if mid_last_dist is not None:
        T = Sc + n_new
        z_merged, dists, mid_z, pts = _new(z, N, T), _new(z, N, T), _new(z, N, T), _new(z, N, T, 3)
        _l.check(L.oi_upsample_mid(_p(_c(rays_o)), _p(_c(rays_d)), _p(_c(z)), _p(_c(sdf)), N, Sc, n_new, float(inv_s),
                                   _p(z_new), _p(pts_new), _p(z_merged), float(mid_last_dist), _p(dists), _p(mid_z), _p(pts),
                                   _stream()), "oi_upsample_mid")
        return z_new, pts_new, z_merged, (dists, mid_z, pts)
    z_merged = _new(z, N, Sc + n_new) if merge else None
    _l.check(L.oi_upsample(_p(_c(rays_o)), _p(_c(rays_d)), _p(_c(z)), _p(_c(sdf)), N, Sc, n_new, float(inv_s),
                           _p(z_new), _p(pts_new), _p(z_merged), _stream()), "oi_upsample")
    return z_new, pts_new, z_merged


def merge_sorted(z, sdf, z_new, sdf_new):
    L = _l.load()
    N, Sc = z.shape
    n_new = z_new.shape[1]
    zo, so = _new(z, N, Sc + n_new), _new(z, N, Sc + n_new)
    _l.check(L.oi_merge_sorted(_p(_c(z)), _p(_c(sdf)), _p(_c(z_new)), _p(_c(sdf_new)), N, Sc, n_new, _p(zo), _p(so),
                               _stream()), "oi_merge_sorted")
    return zo, so


def midpoints(rays_o, rays_d, z, last_dist):
    L = _l.load()
    N, T = z.shape
    dists, mid_z, pts = _new(z, N, T), _new(z, N, T), _new(z, N, T, 3)
    _l.check(L.oi_midpoints(_p(_c(rays_o)), _p(_c(rays_d)), _p(_c(z)), N, T, float(last_dist), _p(dists), _p(mid_z),
                            _p(pts), _stream()), "oi_midpoints")
    return dists, mid_z, pts


PER_SAMPLE_OUT = ("weights", "cdf", "alpha", "inside_sphere", "pts_norm")
PER_RAY_OUT = {"weight_sum": 1, "weight_max": 1, "color_fine": 3, "image_no_bg": 3, "image": 3, "shading": 1,
               "normal": 3, "mask": 1, "z_map": 1, "specular_map": 1, "diffuse_map": 1}


_STATS_TICKETS = {}
# oi_composite_fwd can do oi_render_stats' work itself (its last workgroup sums the partials: stats16 / stats_ticket).  Measured
# at C2 (1024 workgroups, rocprofv3): 25.7 us for the one launch against 14.4 + 4.7 us for the two -- the reduction then sits
# behind the slowest workgroup as a chain of round trips (store acknowledgement, two arrival counters, the loads) that costs
# more than the launch boundary it replaces.  Default: two launches; the one-launch form stays tested (bit-identical).
FUSED_STATS = False


def _stats_ticket(dev):
    """One zero-initialised device word per (device, stream): the arrival counter of the compositing launch's last-block
    reduction (the kernel leaves it at zero; launches that may overlap must not share one)."""
    key = (dev, _stream().value or 0)
    t = _STATS_TICKETS.get(key)
    if t is None:
        t = _STATS_TICKETS[key] = torch.zeros(4097, dtype=torch.int32, device=dev)   # OI_TICKET_WORDS
    return t


def composite_fwd(sdf, grad, rgb, dists, mid_z, rays_o, rays_d, light_dir, bg, variance, light, cos_anneal_ratio,
                  B, outputs=None, image_planar=False):
    """Returns dict of requested outputs (default: all) + 'reduce4' = [sum m*(|g|-1)^2, sum m, sum exp(-100|sdf|), 0].
    With 'reduce4' come two forward-only extras from the SAME launch (its last workgroup does oi_render_stats' sums):
    'ray_sums' = [sum cdf[:,0], sum weight_max, sum weight_sum, 0] and 'finals' = [gradient_error, surface_loss, the three
    means].  `image_planar`: 'image' comes back as (B, 3, N / B) -- the (B, 3, H, W) map itself -- instead of (N, 3)."""
    L = _l.load()
    N, T = dists.shape
    P = _l.CompositeParams()
    keep = []
    for name, t in (("sdf", sdf), ("grad", grad), ("rgb", rgb), ("dists", dists), ("mid_z", mid_z), ("rays_o", rays_o),
                    ("rays_d", rays_d), ("light_dir", light_dir), ("bg", bg), ("variance", variance.reshape(1)),
                    ("light", light)):
        t = _c(t)
        keep.append(t)
        setattr(P, name, _p(t))
    P.cos_anneal_ratio = float(cos_anneal_ratio)
    P.N, P.T, P.B = N, T, B
    want = set(outputs) if outputs is not None else set(PER_SAMPLE_OUT) | set(PER_RAY_OUT) | {"reduce4"}
    out = {}
    for name in PER_SAMPLE_OUT:
        if name in want:
            out[name] = _new(dists, N, T)
        setattr(P, name, _p(out.get(name)))
    for name, c in PER_RAY_OUT.items():
        if name in want:
            out[name] = _new(dists, B, 3, N // B) if (name == "image" and image_planar) else _new(dists, N, c)
        setattr(P, name, _p(out.get(name)))
    P.image_planar = int(bool(image_planar))
    partials = out16 = None
    if "reduce4" in want:  # per-block partial sums (no atomics), reduced by the launch's last workgroup in a fixed order
        partials = _new(dists, L.oi_composite_num_blocks(N), 8)
        out16 = _new(dists, 16)
        if FUSED_STATS:
            ticket = _stats_ticket(dists.device)
            P.stats16, P.stats_ticket = _p(out16), _vp(ticket.data_ptr())
    P.reduce4 = None
    P.block_partials = _p(partials)
    _l.check(L.oi_composite_fwd(ctypes.byref(P), _stream()), "oi_composite_fwd")
    if partials is not None:
        if not FUSED_STATS:  # the two-launch form (kept as the yardstick of tests/test_gpu_kernels.py)
            _l.check(L.oi_render_stats(_p(partials), partials.shape[0], N, T, _p(out16), _stream()), "oi_render_stats")
        out["reduce4"], out["ray_sums"], out["finals"] = out16[0:4], out16[4:8], out16[8:13]
    return out


GRAD_IN = ("weights", "weight_sum", "color_fine", "image_no_bg", "image", "shading", "normal", "mask", "z_map",
           "specular_map", "diffuse_map", "reduce4")


def composite_bwd(sdf, grad, rgb, dists, mid_z, rays_o, rays_d, light_dir, bg, variance, light, cos_anneal_ratio, B,
                  gouts, image_planar=False):
    """gouts: dict name -> upstream gradient (missing / None = zero).  -> d_sdf, d_grad, d_rgb, d_variance (1,),
    d_light (3,), d_light_dir (B,3).  `image_planar`: gouts["image"] is (B, 3, N / B), the layout the forward wrote."""
    L = _l.load()
    N, T = dists.shape
    P = _l.CompositeParams()
    keep = []
    for name, t in (("sdf", sdf), ("grad", grad), ("rgb", rgb), ("dists", dists), ("mid_z", mid_z), ("rays_o", rays_o),
                    ("rays_d", rays_d), ("light_dir", light_dir), ("bg", bg), ("variance", variance.reshape(1)),
                    ("light", light)):
        t = _c(t)
        keep.append(t)
        setattr(P, name, _p(t))
    P.cos_anneal_ratio = float(cos_anneal_ratio)
    P.N, P.T, P.B = N, T, B
    P.image_planar = int(bool(image_planar))
    G = _l.CompositeGrads()
    for name in GRAD_IN:
        t = _c(gouts.get(name))
        keep.append(t)
        setattr(G, "g_" + name, _p(t))
    dev = dists.device
    d_sdf, d_grad, d_rgb = _new(dists, N, T), _new(dists, N, T, 3), _new(dists, N, T, 3)
    d_var, d_light, d_ldir = _zeros_split(dev, (1,), (3,), (B, 3))
    G.d_sdf, G.d_grad, G.d_rgb = _p(d_sdf), _p(d_grad), _p(d_rgb)
    G.d_variance, G.d_light, G.d_light_dir = _p(d_var), _p(d_light), _p(d_ldir)
    partials = _new(dists, N, 8)  # per-ray partial sums of the global gradients, reduced by a second small kernel
    G.ray_partials = _p(partials)
    _l.check(L.oi_composite_bwd(ctypes.byref(P), ctypes.byref(G), _stream()), "oi_composite_bwd")
    return d_sdf, d_grad, d_rgb, d_var, d_light, d_ldir


# ------------------------------------------------------------------------------------------
# discriminator side
# ------------------------------------------------------------------------------------------

def conv4x4_out_shape(x_shape, Cout, stride, pad):
    B, _, H, W = x_shape
    return B, Cout, (H + 2 * pad - 4) // stride + 1, (W + 2 * pad - 4) // stride + 1


def conv4x4_fwd(x, w, bias=None, stride=2, pad=1, slope=0.2, out=None, x_slope=1.0, any_scale=False, zero_tail=0,
                out_is_zero=None):
    """y = lrelu_slope(conv(lrelu_x_slope(x)) + bias).  `out`: optional ZERO-FILLED contiguous output (e.g. a view of an
    arena shared by a chain of layers): the split-K path then needs no fill launch of its own.  `x_slope`: LeakyReLU
    applied to x while it is loaded (the producing layer handed over pre-activations).  `any_scale`: an operand may be a
    gradient (OI_CONV_ANY_SCALE: fp32 matrix cores only)."""
    L = _l.load()
    x, w = _c(x), _c(w)
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    assert w.shape[1:] == (Cin, 4, 4)
    shape = conv4x4_out_shape(x.shape, Cout, stride, pad)
    if out is None:
        y = _new_acc(x, *shape)
    else:
        assert tuple(out.shape) == shape and out.is_contiguous() and out.dtype == torch.float32, (out.shape, shape)
        y = out
    # `zero_tail`: floats of the arena behind `out` that this launch also clears (oi_conv4x4_fwd_arena);
    # `out_is_zero`: whether `out` already holds zeros (default: it does when given)
    zero = (out is not None) if out_is_zero is None else bool(out_is_zero)
    if out is not None and not zero and _active_pool() is not None:
        out.zero_()  # under a ZeroPool the library clears nothing: a caller-owned output that is not yet zero is cleared here
        zero = True
    _l.check(L.oi_conv4x4_fwd_arena(_p(x), _p(w), _p(_c(bias)), _p(y), B, Cin, H, W, Cout, stride, pad, float(slope),
                                    float(x_slope), int(zero) | (2 if any_scale else 0), int(zero_tail), _stream()),
             "oi_conv4x4_fwd")
    return y


def conv4x4_dgrad(g, w, H, W, stride=2, pad=1, mask_ref=None, slope=1.0):
    """mask_ref: the producing layer's forward output -- its LeakyReLU is applied to `g` on load (oi_conv4x4_dgrad_masked)."""
    L = _l.load()
    g, w = _c(g), _c(w)
    B, Cout = g.shape[:2]
    Cin = w.shape[1]
    gx = _new_acc(g, B, Cin, H, W)
    _l.check(L.oi_conv4x4_dgrad_masked(_p(g), _p(_c(mask_ref)), float(slope), _p(w), _p(gx), B, Cin, H, W, Cout, stride, pad,
                                       _stream()), "oi_conv4x4_dgrad")
    return gx


def conv4x4_wgrad(g, x, stride=2, pad=1, mask_ref=None, slope=1.0, acc=None):
    """acc: a [Cout, Cin, 4, 4] buffer that already holds a gradient (or zeros) -- this one is added to it and `acc` returned."""
    L = _l.load()
    g, x = _c(g), _c(x)
    B, Cin, H, W = x.shape
    Cout = g.shape[1]
    if acc is not None and (tuple(acc.shape) != (Cout, Cin, 4, 4) or not acc.is_contiguous()):
        raise _l.OiHipError(f"conv4x4_wgrad: accumulator of shape {tuple(acc.shape)} for a {(Cout, Cin, 4, 4)} gradient")
    gw = _new_acc(g, Cout, Cin, 4, 4) if acc is None else acc
    _l.check(L.oi_conv4x4_wgrad_masked(_p(g), _p(_c(mask_ref)), float(slope), _p(x), _p(gw), int(acc is not None), B, Cin, H, W,
                                       Cout, stride, pad, _stream()), "oi_conv4x4_wgrad")
    return gw


def conv4x4_dgrad_pre(g, w, x, x_slope, stride=2, pad=1):
    """Data gradient of y = conv(lrelu_{x_slope}(x), w) with respect to the pre-activation x (oi_conv4x4_dgrad_pre)."""
    L = _l.load()
    g, w, x = _c(g), _c(w), _c(x)
    B, Cin, H, W = x.shape
    gx = _new_acc(g, B, Cin, H, W)
    _l.check(L.oi_conv4x4_dgrad_pre(_p(g), _p(w), _p(x), float(x_slope), _p(gx), B, Cin, H, W, g.shape[1], stride, pad,
                                    _stream()), "oi_conv4x4_dgrad_pre")
    return gx


def conv4x4_bwd(g, w, x, stride=2, pad=1, mask_ref=None, slope=1.0, acc=None, x_slope=1.0):
    """Data and weight gradient of one layer in one launch (oi_conv4x4_bwd_pre) -> gx, gw (gw is `acc` when given).
    x_slope != 1: x is a pre-activation (the layer computed conv(lrelu(x), w)); gx is the gradient with respect to it."""
    L = _l.load()
    g, w, x = _c(g), _c(w), _c(x)
    B, Cin, H, W = x.shape
    Cout = g.shape[1]
    if acc is not None and (tuple(acc.shape) != (Cout, Cin, 4, 4) or not acc.is_contiguous()):
        raise _l.OiHipError(f"conv4x4_bwd: accumulator of shape {tuple(acc.shape)} for a {(Cout, Cin, 4, 4)} gradient")
    gx = _new_acc(g, B, Cin, H, W)
    gw = _new_acc(g, Cout, Cin, 4, 4) if acc is None else acc
    _l.check(L.oi_conv4x4_bwd_pre(_p(g), _p(_c(mask_ref)), float(slope), _p(w), _p(x), float(x_slope), _p(gx), _p(gw),
                                  int(acc is not None), B, Cin, H, W, Cout, stride, pad, _stream()), "oi_conv4x4_bwd")
    return gx, gw


class GradSink:
    """Weight-gradient accumulators for ONE plain backward pass (not create_graph): while active, the convolution Functions
    of oi_amd.autograd_conv add every weight-gradient contribution of a registered weight straight into its buffer
    (oi_conv4x4_wgrad_masked, accumulate) and hand autograd nothing for it -- instead of one fresh tensor per contribution
    that autograd then sums with one `+=` launch each (a discriminator step has four contributions per weight).  The owner
    zeroes the buffers beforehand and installs them as `.grad` afterwards (oi_amd.graphed.GraphedDStep)."""

    _active = None

    def __init__(self, buffers):
        self.buffers = {int(k): v for k, v in buffers.items()}   # weight.data_ptr() -> accumulator

    def __enter__(self):
        assert GradSink._active is None, "nested GradSink"
        GradSink._active = self
        return self

    def __exit__(self, *exc):
        GradSink._active = None

    @staticmethod
    def lookup(w):
        s = GradSink._active
        return None if s is None else s.buffers.get(w.data_ptr())


def lrelu_mask_mul(v, ref, slope):
    L = _l.load()
    v, ref = _c(v), _c(ref)
    out = torch.empty_like(v)
    _l.check(L.oi_lrelu_mask_mul(_p(v), _p(ref), _p(out), v.numel(), float(slope), _stream()), "oi_lrelu_mask_mul")
    return out


def light_dir_fwd(d, w2b):
    L = _l.load()
    d, w2b = _c(d), _c(w2b)
    n = _new(w2b, w2b.shape[0], 3)
    _l.check(L.oi_light_dir_fwd(_p(d), _p(w2b), _p(n), w2b.shape[0], _stream()), "oi_light_dir_fwd")
    return n


def light_dir_bwd(d, w2b, g_n):
    L = _l.load()
    d, w2b, g_n = _c(d), _c(w2b), _c(g_n)
    g_d = _new(d, 3)
    _l.check(L.oi_light_dir_bwd(_p(d), _p(w2b), _p(g_n), _p(g_d), w2b.shape[0], _stream()), "oi_light_dir_bwd")
    return g_d


def _gan_shapes(d_real, d_fake, pose, gx, aux_w):
    """(B, K, N) of the fused GAN-loss launches, with the shape checks the ATen composition they replace performed on its
    own (torch.split / mse_loss raise on a mismatch; the kernels index raw pointers)."""
    ref = d_real if d_real is not None else d_fake
    if ref is None or ref.dim() != 2:
        raise ValueError("gan_losses: logits must be [B, K]")
    B, K = ref.shape
    for name, t in (("d_real", d_real), ("d_fake", d_fake)):
        if t is not None and tuple(t.shape) != (B, K):
            raise ValueError(f"gan_losses: {name} is {tuple(t.shape)}, expected {(B, K)}")
    if pose is not None:
        if d_fake is None:
            raise ValueError("gan_losses: a pose target needs the fake logits")
        if tuple(pose.shape) != (B, K - 1):
            raise ValueError(f"gan_losses: pose is {tuple(pose.shape)}, expected {(B, K - 1)} (logits [:, 1:], position.py:4-12)")
        if aux_w is None or aux_w.numel() != 1:
            raise ValueError("gan_losses: a pose target needs its weight (one device scalar)")
    if gx is not None and (gx.dim() < 1 or gx.shape[0] != B):
        raise ValueError(f"gan_losses: the R1 gradient has leading dimension {tuple(gx.shape)[:1]}, expected {B}")
    return B, K, (0 if gx is None else gx.numel() // B)


def gan_losses_fwd(d_real, d_fake, pose, gx, aux_w, reg_w):
    """-> out6 = (total, real + fake, reg, fake, real, aux): see oi_gan_losses_fwd.  d_real / d_fake [B, K] (or None),
    pose [B, K-1] (or None), gx [B, ...] (or None), aux_w a device scalar tensor (or None)."""
    L = _l.load()
    B, K, N = _gan_shapes(d_real, d_fake, pose, gx, aux_w)
    out = _new(d_real if d_real is not None else d_fake, 6)
    _l.check(L.oi_gan_losses_fwd(_p(d_real), _p(d_fake), _p(pose), _p(gx), _p(aux_w), float(reg_w), _p(out), B, K, N, _stream()),
             "oi_gan_losses_fwd")
    return out


def stage_inputs(copies, imm=None, imm_dst=None):
    """One launch for the inputs of a captured step: `copies` = up to 4 (src, dst) pairs of equally sized contiguous float32
    device tensors; `imm` = up to 64 Python / numpy floats written to the device tensor `imm_dst` (oi_stage_inputs)."""
    L = _l.load()
    copies = [(s_, d_) for s_, d_ in copies if s_ is not None]
    n = len(copies)
    for s_, d_ in copies:
        assert s_.numel() == d_.numel(), (tuple(s_.shape), tuple(d_.shape))
    srcs = (ctypes.c_void_p * max(n, 1))(*[_p(_c(s_.detach())).value for s_, _ in copies])
    dsts = (ctypes.c_void_p * max(n, 1))(*[_p(d_).value for _, d_ in copies])
    cnts = (ctypes.c_longlong * max(n, 1))(*[s_.numel() for s_, _ in copies])
    vals = [] if imm is None else [float(v) for v in imm]
    assert len(vals) <= 64 and (not vals or imm_dst.numel() >= len(vals))
    immv = (ctypes.c_float * max(len(vals), 1))(*vals)
    _l.check(L.oi_stage_inputs(srcs, dsts, cnts, n, immv, len(vals), _p(imm_dst) if vals else None, _stream()), "oi_stage_inputs")


def scalar_glue(variance, ambient, specular, shininess):
    """-> (out5 = [inv_s, 1 / inv_s, ambient colour, diffuse colour, specular colour], packed3 = the compositing kernel's light
    block) from the four 0-dim parameters: one launch (oi_scalar_glue)."""
    L = _l.load()
    out5, packed = _new(variance, 5), _new(variance, 3)
    ps = [_p(t.detach().reshape(1)) for t in (variance, ambient, specular, shininess)]
    _l.check(L.oi_scalar_glue(*ps, _p(out5), _p(packed), _stream()), "oi_scalar_glue")
    return out5, packed


def upload_small(values, device):
    """numpy / Python floats (<= 64 of them) -> a new float32 device tensor of the same shape, through the ARGUMENTS of one launch
    (oi_stage_inputs): no pageable host-to-device copy -- which makes the host wait for everything queued on the stream -- and
    no pinned staging ring."""
    import numpy as np
    arr = np.ascontiguousarray(values, dtype=np.float32)
    out = torch.empty(arr.shape, dtype=torch.float32, device=device)
    stage_inputs([], arr.ravel(), out)
    return out


def render_scalars_fwd(r4, inv_nt):
    """(gradient_error, surface_loss) = (r4[0] / (r4[1] + 1e-5), r4[2] * inv_nt) as a [2] tensor."""
    L = _l.load()
    out = _new(r4, 2)
    _l.check(L.oi_render_scalars_fwd(_p(r4), float(inv_nt), _p(out), _stream()), "oi_render_scalars_fwd")
    return out


def render_scalars_bwd(r4, g_err, g_surf, inv_nt):
    L = _l.load()
    g = _new(r4, 4)
    _l.check(L.oi_render_scalars_bwd(_p(r4), _p(g_err), _p(g_surf), float(inv_nt), _p(g), _stream()), "oi_render_scalars_bwd")
    return g


def weighted_sum_fwd(terms, weights):
    """sum_i weights[i] * terms[i] over up to 8 device scalars (0-dim or 1-element float32 tensors) -> 0-dim tensor."""
    L = _l.load()
    n = len(terms)
    ptrs = (ctypes.c_void_p * n)(*[_p(t).value for t in terms])
    ws = (ctypes.c_float * n)(*[float(w) for w in weights])
    out = torch.empty((), dtype=torch.float32, device=terms[0].device)
    _l.check(L.oi_weighted_sum_fwd(ptrs, ws, n, _p(out), _stream()), "oi_weighted_sum_fwd")
    return out


def weighted_sum_bwd(g_out, weights, device):
    """-> [n] tensor of weights[i] * g_out (g_out: a device scalar)."""
    L = _l.load()
    n = len(weights)
    ws = (ctypes.c_float * n)(*[float(w) for w in weights])
    g = torch.empty(n, dtype=torch.float32, device=device)
    _l.check(L.oi_weighted_sum_bwd(_p(g_out), ws, n, _p(g), _stream()), "oi_weighted_sum_bwd")
    return g


def gan_losses_bwd(g_total, d_real, d_fake, pose, gx, aux_w, reg_w, want_real, want_fake, want_gx, out=None):
    """out: optional (g_real, g_fake, g_gx) destinations (contiguous; e.g. slices of one tensor)."""
    L = _l.load()
    B, K, N = _gan_shapes(d_real, d_fake, pose, gx, aux_w)
    if out is not None:
        g_real, g_fake, g_gx = out
    else:
        g_real = torch.empty_like(d_real) if want_real else None
        g_fake = torch.empty_like(d_fake) if want_fake else None
        g_gx = torch.empty_like(gx) if want_gx else None
    _l.check(L.oi_gan_losses_bwd(_p(g_total), _p(d_real), _p(d_fake), _p(pose), _p(gx), _p(aux_w), float(reg_w), _p(g_real),
                                 _p(g_fake), _p(g_gx), B, K, N, _stream()), "oi_gan_losses_bwd")
    return g_real, g_fake, g_gx


def channel_sum(g):
    L = _l.load()
    g = _c(g)
    B, C = g.shape[:2]
    gb = _new(g, C)
    _l.check(L.oi_channel_sum(_p(g), _p(gb), B, C, g[0, 0].numel(), _stream()), "oi_channel_sum")
    return gb


def upfirdn2d(x, f, upx=1, upy=1, downx=1, downy=1, padx0=0, padx1=0, pady0=0, pady1=0, flip=False, gain=1.0):
    """Same contract as the reference plugin op (f is 2-D [fh, fw])."""
    L = _l.load()
    x, f = _c(x), _c(f)
    B, C, H, W = x.shape
    fh, fw = f.shape
    Wo = (W * upx + padx0 + padx1 - fw + downx) // downx
    Ho = (H * upy + pady0 + pady1 - fh + downy) // downy
    y = _new(x, B, C, Ho, Wo)
    _l.check(L.oi_upfirdn2d(_p(x), _p(f), _p(y), B * C, H, W, fh, fw, upx, upy, downx, downy, padx0, padx1, pady0,
                            pady1, int(bool(flip)), float(gain), _stream()), "oi_upfirdn2d")
    return y


_DISC_TICKETS = {}


def disc_fwd_small(x, weights, whead, bhead, f12=None, theta_np=None, theta_dev=None, margins=(0, 0, 0, 0), slope=0.2):
    """DCDiscriminator(img_size 64, n_feat 512) forward at batch <= 4 in four / five launches (oi_disc_fwd_small): optional ADA
    geometry (theta_np: (B, 2, 3) numpy, passed by value | theta_dev: device tensor; margins (mx0, my0, mx1, my1) as
    AugmentPipe.margins_for returns them) + four conv blocks + head.  -> logits (B, out_dim)."""
    L = _l.load()
    x = _c(x)
    B, C, H, W = x.shape
    mx0, my0, mx1, my1 = (int(v) for v in margins)
    big = len(weights) == 5   # the shipped 128 x 128 / five-block network (oi_disc_fwd_small128)
    n = (L.oi_disc_fwd_small128_workspace_floats if big else L.oi_disc_fwd_small_workspace_floats)(B, C, mx0, mx1, my0, my1)
    ws = torch.empty(n, dtype=torch.float32, device=x.device)
    key = (x.device, _stream().value or 0)
    ticket = _DISC_TICKETS.get(key)
    if ticket is None:
        if torch.cuda.is_current_stream_capturing():
            # the zero fill would become a node of the caller's graph (and the memory part of its private pool): eager calls on
            # this stream would then reuse counters that were never cleared.  Warm the entry point up before capturing.
            raise _l.OiHipError("oi_disc_fwd_small: first call on this stream is inside a stream capture; call it once eagerly "
                                "(warm-up) before capturing")
        ticket = _DISC_TICKETS[key] = torch.zeros(4097, dtype=torch.int32, device=x.device)   # OI_TICKET_WORDS
    out_dim = whead.shape[0]
    logits = _new(x, B, out_dim)
    th_host = th_arr = None
    if theta_np is not None:
        th_arr = np.ascontiguousarray(theta_np, dtype=np.float32).reshape(-1)   # alive during the call: the C side copies the
        assert th_arr.size == 6 * B                                             # values into the kernel arguments
        th_host = th_arr.ctypes.data_as(ctypes.c_void_p)
    ws_ = [_c(w) for w in weights]
    if big:
        assert (H, W) == (128, 128)
        _l.check(L.oi_disc_fwd_small128(_p(x), th_host, _p(_c(theta_dev)), _p(_c(f12)) if f12 is not None else _p(x), mx0, mx1, my0, my1,
                                        *[_p(w) for w in ws_], _p(_c(whead)), _p(_c(bhead)), _p(ws), _vp(ticket.data_ptr()), _p(logits),
                                        B, C, int(ws_[4].shape[0]), int(out_dim), float(slope), _stream()), "oi_disc_fwd_small128")
        return logits
    _l.check(L.oi_disc_fwd_small(_p(x), th_host, _p(_c(theta_dev)), _p(_c(f12)) if f12 is not None else _p(x), mx0, mx1, my0, my1,
                                 *[_p(w) for w in ws_], _p(_c(whead)), _p(_c(bhead)), _p(ws), _vp(ticket.data_ptr()), _p(logits),
                                 B, C, H, W, int(ws_[3].shape[0]), int(out_dim), float(slope), _stream()), "oi_disc_fwd_small")
    return logits


class DiscLargePack:
    """Packed weights of the batch >= 16 no-grad discriminator forward (oi_disc_large_*, csrc/disc_large.hip).  `ok` is False when
    the network is not covered (the caller keeps the general chain)."""

    def __init__(self, weights, whead):
        L = _l.load()
        self.chans = [int(weights[0].shape[1])] + [int(w.shape[0]) for w in weights]
        self.nb, self.out_dim = len(weights), int(whead.shape[0])
        self.c_chans = (ctypes.c_int * len(self.chans))(*self.chans)
        n = int(L.oi_disc_large_packed_bytes(self.c_chans, self.nb, self.out_dim))
        self.ok = n > 0 and all(w.dtype == torch.float32 and w.is_contiguous() and w.is_cuda for w in list(weights) + [whead])
        if not self.ok:
            return
        self.packed = torch.empty(n, dtype=torch.uint8, device=weights[0].device)
        ptrs = (ctypes.c_void_p * self.nb)(*[w.data_ptr() for w in weights])
        _l.check(L.oi_disc_large_pack(ptrs, _p(whead), self.c_chans, self.nb, self.out_dim, _p(self.packed), _stream()),
                 "oi_disc_large_pack")
        self._ws = {}

    def workspace(self, B, H, device):
        key = (B, H, _stream().value or 0)
        ws = self._ws.get(key)
        if ws is None:
            n = int(_l.load().oi_disc_large_workspace_bytes(self.c_chans, self.nb, self.out_dim, B, H))
            if n == 0:
                return None
            if len(self._ws) >= 4:
                self._ws.clear()
            ws = self._ws[key] = torch.empty(n, dtype=torch.uint8, device=device)
        return ws


def disc_fwd_large(x, pack, w1, bhead, slope=0.2):
    """-> logits (B, out_dim), or None when the (shape, network) pair is not covered."""
    x = _c(x)
    B, C, H, W = x.shape
    if not pack.ok or H != W or C != pack.chans[0]:
        return None
    ws = pack.workspace(B, H, x.device)
    if ws is None:
        return None
    logits = _new(x, B, pack.out_dim)
    _l.check(_l.load().oi_disc_fwd_large(_p(x), _p(w1), _p(pack.packed), _p(_c(bhead)), _p(ws), ws.numel(), _p(logits), pack.c_chans,
                                         pack.nb, pack.out_dim, B, H, float(slope), _stream()), "oi_disc_fwd_large")
    return logits


class DiscGraph:
    """oi_disc_graph_*: the batch <= 4 discriminator forward as a plan owned by the library -- every argument but the image pointer
    and the sampling matrices is fixed at creation; a call passes those two and costs one short ctypes call.  `launch`:
    "eager" (default; OI_DISC_LAUNCH): the four launches issued one by one from the stored arguments | "graph": a hipGraph
    replay whose first nodes get the image pointer and the matrices as updated kernel-node parameters (no staging launch
    either way).  Measured at batch 1: 31.7 us per image eager, 36.6 us replayed -- a graph launch costs ~5 us of GPU time
    between two replays on this runtime, back-to-back eager launches have no gap, and the host needs ~16 us for them.
    `margins`: the STATIC ones (AugmentPipe.static_margins) or None for no augmentation.  The returned logits tensor is
    overwritten by the next call."""

    def __init__(self, shape, device, weights, whead, bhead, f12=None, margins=None, slope=0.2, launch=None):
        L = _l.load()
        B, C, H, W = shape
        if torch.cuda.is_current_stream_capturing():
            raise _l.OiHipError("ops.DiscGraph: created inside a stream capture (its arrival counters are zeroed at creation: the "
                                "fill must run, not be recorded); create the plan before capturing")
        self.shape, self.aug = (B, C, H, W), margins is not None
        launch = launch or os.environ.get("OI_DISC_LAUNCH", "eager")
        assert launch in ("eager", "graph"), launch
        self.eager = launch == "eager"
        mx0, my0, mx1, my1 = (int(v) for v in margins) if self.aug else (0, 0, 0, 0)
        self.big = len(weights) == 5   # the shipped 128 x 128 / five-block network: launch by launch only (oi_disc_graph_create128)
        if self.big and launch != "eager":
            raise _l.OiHipError("ops.DiscGraph: the 128 x 128 plan is launched launch by launch (launch='eager')")
        n = (L.oi_disc_fwd_small128_workspace_floats if self.big else L.oi_disc_fwd_small_workspace_floats)(B, C, mx0, mx1, my0, my1)
        self.ws = torch.empty(n, dtype=torch.float32, device=device)
        self.ticket = torch.zeros(4097, dtype=torch.int32, device=device)   # OI_TICKET_WORDS, this graph's own
        self.logits = torch.empty(B, whead.shape[0], dtype=torch.float32, device=device)
        # (the graph holds raw pointers: keep the tensors it was built from alive)
        nb = len(weights)
        self.keep = [_c(w) for w in weights] + [_c(whead), _c(bhead), _c(f12) if f12 is not None else None]
        self.handle = ctypes.c_void_p()
        if self.big:
            assert (H, W) == (128, 128)
            _l.check(L.oi_disc_graph_create128(ctypes.byref(self.handle), int(self.aug), _p(self.keep[nb + 2]), mx0, mx1, my0, my1,
                                               *[_p(w) for w in self.keep[:5]], _p(self.keep[5]), _p(self.keep[6]), _p(self.ws),
                                               _vp(self.ticket.data_ptr()), _p(self.logits), B, C, int(self.keep[4].shape[0]),
                                               int(whead.shape[0]), float(slope)), "oi_disc_graph_create128")
            return
        _l.check(L.oi_disc_graph_create(ctypes.byref(self.handle), int(self.aug), _p(self.keep[6]), mx0, mx1, my0, my1,
                                        *[_p(w) for w in self.keep[:4]], _p(self.keep[4]), _p(self.keep[5]), _p(self.ws),
                                        _vp(self.ticket.data_ptr()), _p(self.logits), B, C, H, W, int(self.keep[3].shape[0]),
                                        int(whead.shape[0]), float(slope)), "oi_disc_graph_create")

    def __call__(self, x, theta_np=None, fresh=False):
        """`fresh` (eager launches only): the result goes to a new tensor instead of the plan's own buffer."""
        assert tuple(x.shape) == self.shape and (theta_np is not None) == self.aug
        x = _c(x)
        th = None
        if self.aug:
            th_arr = np.ascontiguousarray(theta_np, dtype=np.float32).reshape(-1)   # (alive during the call: copied by value)
            assert th_arr.size == 6 * self.shape[0]
            th = th_arr.ctypes.data_as(ctypes.c_void_p)
        L = _l.load()
        if self.eager:
            out = torch.empty_like(self.logits) if fresh else self.logits
            _l.check(L.oi_disc_graph_launch_eager(self.handle, _p(x), th, _p(out), _stream()), "oi_disc_graph_launch_eager")
            return out
        assert not fresh, "a graph replay writes the buffer it was captured with"
        _l.check(L.oi_disc_graph_launch(self.handle, _p(x), th, _stream()), "oi_disc_graph_launch")
        return self.logits

    def call_ada(self, x, seed, p_xint, xint_max, p_scale, scale_std, fresh=False):
        """The forward with AugmentPipe's xint + scale draws made INSIDE the library from one 64-bit seed
        (oi_disc_graph_launch_ada): no numpy, no matrix algebra and no array marshalling on the way.  `x` must already be a
        contiguous fp32 CUDA tensor of the plan's shape (the caller's guard)."""
        L = _l.load()
        if self.eager:
            out = torch.empty_like(self.logits) if fresh else self.logits
            rc = L.oi_disc_graph_launch_ada(self.handle, x.data_ptr(), seed, p_xint, xint_max, p_scale, scale_std, out.data_ptr(), 1,
                                            _stream())
        else:
            out = self.logits
            rc = L.oi_disc_graph_launch_ada(self.handle, x.data_ptr(), seed, p_xint, xint_max, p_scale, scale_std, None, 0, _stream())
        if rc != 0:
            _l.check(rc, "oi_disc_graph_launch_ada")
        return out

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                _l.load().oi_disc_graph_destroy(h)
            except Exception:
                pass


def ada_theta_xint_scale(seed, B, H, W, margins, p_xint, xint_max, p_scale, scale_std, with_draws=False):
    """(B, 2, 3) float32 numpy: the sampling matrices the library forms from `seed` (oi_ada_theta_xint_scale; host only, no HIP
    call) -- what DiscGraph.call_ada uses for the same seed.  with_draws: also (B, 3) = (t_x, t_y, s) as drawn."""
    th = np.empty((B, 2, 3), np.float32)
    ts = np.empty((B, 3), np.float32) if with_draws else None
    mx0, my0, mx1, my1 = (int(v) for v in margins)
    _l.check(_l.load().oi_ada_theta_xint_scale(int(seed), B, H, W, mx0, mx1, my0, my1, p_xint, xint_max, p_scale, scale_std,
                                               th.ctypes.data_as(_vp), None if ts is None else ts.ctypes.data_as(_vp)),
             "oi_ada_theta_xint_scale")
    return (th, ts) if with_draws else th


ADA_SEPARABLE = os.environ.get("OI_ADA_SEP", "1") != "0"   # axis-aligned matrices: the one-launch form (oi_ada_geom_sep_fwd)


def ada_geom_sep_ok(x):
    """The one-launch separable augmentation covers this image batch (1..3 channels of 64 x 64, fp32, on the GPU)."""
    return (ADA_SEPARABLE and x.is_cuda and x.dim() == 4 and x.dtype is torch.float32 and x.shape[0] <= 65535
            and bool(_l.load().oi_ada_geom_sep_supported(x.shape[1], x.shape[2], x.shape[3])))


def ada_geom_sep_host(x, theta_np, f12, margins):
    """oi_ada_geom_sep_fwd with AXIS-ALIGNED sampling matrices that live on the host ((B, 2, 3) float32 numpy: they travel in the
    kernel arguments, nothing is uploaded).  No gradient."""
    x, f12 = _c(x), _c(f12)
    B, C, H, W = x.shape
    mx0, my0, mx1, my1 = margins
    th = np.ascontiguousarray(theta_np, np.float32)
    assert th.shape == (B, 2, 3) and f12.numel() == 12
    if np.any(th[:, 0, 1] != 0) or np.any(th[:, 1, 0] != 0):   # (the kernel does not read them: it would silently drop a rotation)
        raise ValueError("ada_geom_sep_host: a sampling matrix with off-diagonal entries (rotation) -- use ada_geom_fwd")
    y = torch.empty_like(x)
    _l.check(_l.load().oi_ada_geom_sep_fwd(_p(x), None, th.ctypes.data_as(_vp), _p(f12), _p(y), B, C, H, W, mx0, mx1, my0, my1,
                                           _stream()), "oi_ada_geom_sep_fwd")
    return y


def ada_geom_adj_sep(gy, theta, f12, margins):
    """A^T gy for AXIS-ALIGNED device matrices in one launch (oi_ada_geom_sep_adj); the caller checked ada_geom_sep_ok(gy)."""
    gy, theta, f12 = _c(gy), _c(theta), _c(f12)
    B, C, H, W = gy.shape
    mx0, my0, mx1, my1 = margins
    gx = torch.empty_like(gy)
    _l.check(_l.load().oi_ada_geom_sep_adj(_p(gy), _p(theta), None, _p(f12), _p(gx), B, C, H, W, mx0, mx1, my0, my1, _stream()),
             "oi_ada_geom_sep_adj")
    return gx


def ada_geom_fwd(x, theta, f12, margins, axis_aligned=False):
    """reflect pad + x2 up-FIR + affine resample + /2 down-FIR (AugmentPipe geometry); see oi_ada_geom_fwd (two launches) and
    oi_ada_geom_sep_fwd (one: `axis_aligned` is the caller's promise that no theta carries a rotation)."""
    L = _l.load()
    x, theta, f12 = _c(x), _c(theta), _c(f12)
    B, C, H, W = x.shape
    mx0, my0, mx1, my1 = margins
    assert f12.numel() == 12 and theta.shape == (B, 2, 3)
    y = torch.empty_like(x)
    if axis_aligned and ADA_SEPARABLE and B <= 65535 and L.oi_ada_geom_sep_supported(C, H, W):
        _l.check(L.oi_ada_geom_sep_fwd(_p(x), _p(theta), None, _p(f12), _p(y), B, C, H, W, mx0, mx1, my0, my1, _stream()),
                 "oi_ada_geom_sep_fwd")
        return y
    canvas = _new(x, B * C * 2 * (H + my0 + my1) * 2 * (W + mx0 + mx1))
    _l.check(L.oi_ada_geom_fwd(_p(x), _p(theta), _p(f12), _p(y), _p(canvas), B, C, H, W, mx0, mx1, my0, my1, _stream()),
             "oi_ada_geom_fwd")
    return y


def affine_grid_sample_fwd(x, theta, Ho, Wo):
    L = _l.load()
    x, theta = _c(x), _c(theta)
    B, C, Hi, Wi = x.shape
    y = _new(x, B, C, Ho, Wo)
    _l.check(L.oi_affine_grid_sample_fwd(_p(x), _p(theta), _p(y), B, C, Hi, Wi, Ho, Wo, _stream()),
             "oi_affine_grid_sample_fwd")
    return y


def affine_grid_sample_bwd(gy, theta, Hi, Wi):
    L = _l.load()
    gy, theta = _c(gy), _c(theta)
    B, C, Ho, Wo = gy.shape
    gx = _new_acc(gy, B, C, Hi, Wi)
    _l.check(L.oi_affine_grid_sample_bwd(_p(gy), _p(theta), _p(gx), B, C, Hi, Wi, Ho, Wo, _stream()),
             "oi_affine_grid_sample_bwd")
    return gx


def reflect_pad_fwd(x, px0, px1, py0, py1):
    L = _l.load()
    x = _c(x)
    B, C, H, W = x.shape
    y = _new(x, B, C, H + py0 + py1, W + px0 + px1)
    _l.check(L.oi_reflect_pad_fwd(_p(x), _p(y), B * C, H, W, px0, px1, py0, py1, _stream()), "oi_reflect_pad_fwd")
    return y


def reflect_pad_bwd(gy, H, W, px0, px1, py0, py1):
    L = _l.load()
    gy = _c(gy)
    B, C = gy.shape[:2]
    gx = _new_acc(gy, B, C, H, W)
    _l.check(L.oi_reflect_pad_bwd(_p(gy), _p(gx), B * C, H, W, px0, px1, py0, py1, _stream()), "oi_reflect_pad_bwd")
    return gx
