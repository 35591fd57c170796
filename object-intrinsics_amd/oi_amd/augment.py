"""StyleGAN2-ADA geometric augmentation -- drop-in for src.third_party.ada.augment.AugmentPipe
(augment.py:117-429) for the branches the path enables (configs/train.yaml:80-85: xint, scale; plus
xflip / rotate90 / rotate / aniso / xfrac which only change the 3x3 matrix).  Colour, image-filter,
noise and cutout branches are dead at this configuration and raise if enabled.

Execution (augment.py:270-301) = reflect pad -> x2 sym6 upsample -> affine bilinear resample -> /2
sym6 downsample, all HIP kernels (csrc/disc.hip) wrapped in autograd Functions whose backward is
built from the same kernels (arbitrary order, as the reference's upfirdn2d / grid_sample_gradfix)."""
import numpy as np
import torch

import os

from .autograd_disc import ada_geom, affine_grid_sample, reflect_pad, upfirdn2d_separable

# OI_ADA_FUSED=0: the four stages as separate kernels (the round-1 / round-2 path; kept for A/B and as the second opinion of
# tests/test_gpu_kernels.py::test_ada_geom_fused_matches_the_staged_chain)
FUSED = os.environ.get("OI_ADA_FUSED", "1") != "0"

SYM6 = [0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633,
        0.4910559419267466, 0.787641141030194, 0.3379294217276218, -0.07263752278646252,
        -0.021060292512300564, 0.04472490177066578, 0.0017677118642428036, -0.007800708325034148]


def _mat(rows, B):
    """(B,3,3) float32 matrices from python scalars / (B,) arrays.  numpy on the host: the augmentation
    parameters are O(B) scalars, so they are sampled and composed on the CPU (no kernel launches, and the
    data-dependent padding margins need no device->host read, unlike augment.py:283)."""
    out = np.zeros((B, 3, 3), dtype=np.float32)
    for i, row in enumerate(rows):
        for j, v in enumerate(row):
            out[:, i, j] = v
    return out


def translate2d(tx, ty, B):
    return _mat([[1, 0, tx], [0, 1, ty], [0, 0, 1]], B)


def scale2d(sx, sy, B):
    return _mat([[sx, 0, 0], [0, sy, 0], [0, 0, 1]], B)


def rotate2d(theta, B):
    return _mat([[np.cos(theta), np.sin(-theta), 0], [np.sin(theta), np.cos(theta), 0], [0, 0, 1]], B)


def _erfinv(x):
    return float(torch.erfinv(torch.tensor(float(x), dtype=torch.float32)))


class AugmentPipe(torch.nn.Module):
    def __init__(self, xflip=0, rotate90=0, xint=0, xint_max=0.125, scale=0, rotate=0, aniso=0, xfrac=0,
                 scale_std=0.2, rotate_max=1, aniso_std=0.2, xfrac_std=0.125, brightness=0, contrast=0, lumaflip=0,
                 hue=0, saturation=0, imgfilter=0, noise=0, cutout=0, **_unused):
        super().__init__()
        if any(float(v) > 0 for v in (brightness, contrast, lumaflip, hue, saturation, imgfilter, noise, cutout)):
            raise NotImplementedError("colour / filter / noise / cutout augmentations are disabled on the path "
                                      "(configs/train.yaml:80-85)")
        self.register_buffer("p", torch.ones([]))
        self.xflip, self.rotate90, self.xint, self.xint_max = float(xflip), float(rotate90), float(xint), float(xint_max)
        self.scale, self.rotate, self.aniso, self.xfrac = float(scale), float(rotate), float(aniso), float(xfrac)
        self.scale_std, self.rotate_max = float(scale_std), float(rotate_max)
        self.aniso_std, self.xfrac_std = float(aniso_std), float(xfrac_std)
        self._only_xint_scale = (self.xint > 0 and self.scale > 0 and
                                 not any(v > 0 for v in (self.xflip, self.rotate90, self.rotate, self.aniso, self.xfrac)))
        self._theta_ab = {}
        f = torch.tensor(SYM6, dtype=torch.float32)
        self.register_buffer("Hz_geom", f / f.sum())
        # kept for state_dict compatibility with the reference (augment.py:171-179); unused here
        self.register_buffer("Hz_fbank", torch.zeros(4, 1))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        k = prefix + "Hz_fbank"
        if k in state_dict and state_dict[k].shape != self.Hz_fbank.shape:
            self.Hz_fbank = torch.zeros_like(state_dict[k])
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _p_host(self):
        """`p` is a buffer the trainer sets once at construction (discriminator.py:91); mirror it on the host."""
        key = (self.p.data_ptr(), self.p._version)
        if getattr(self, "_p_cache", (None, None))[0] != key:
            self._p_cache = (key, float(self.p))
        return self._p_cache[1]

    def fast_draw_ok(self):
        """True when a forward's parameters may be drawn inside the library from one seed (ops.DiscGraph.call_ada): the shipped
        configuration (xint + scale only), this class's own `forward` / `sample_G_inv` (tests and the F13 replay pin
        `debug_percentile` by overriding them on the instance), the 12-tap filter."""
        d = self.__dict__
        return (self._only_xint_scale and "forward" not in d and "sample_G_inv" not in d and type(self).forward is AugmentPipe.forward
                and type(self).sample_G_inv is AugmentPipe.sample_G_inv and self.Hz_geom.shape[0] == 12)

    def fast_params(self):
        """(xint * p, xint_max, scale * p, scale_std): the gate probabilities and strengths oi_ada_theta_xint_scale takes."""
        key = (self.p.data_ptr(), self.p._version)
        hit = self.__dict__.get("_fast_params")
        if hit is None or hit[0] != key:
            p = self._p_host()
            hit = self.__dict__["_fast_params"] = (key, (self.xint * p, self.xint_max, self.scale * p, self.scale_std))
        return hit[1]

    @staticmethod
    def draw_seed():
        """One 64-bit seed per forward from numpy's GLOBAL stream (so `np.random.seed` still pins a forward's augmentation; the
        library expands it into the per-image draws).  53 random bits from one `np.random.random()` -- a Python float, 0.3 us."""
        return int(np.random.random() * 9007199254740992.0)

    def theta_fast(self, B, H, W, seed=None, with_draws=False):
        """The sampling matrices a library-drawn forward uses for `seed` (default: the next draw_seed()), at the static margins."""
        from . import ops
        if seed is None:
            seed = self.draw_seed()
        return ops.ada_theta_xint_scale(seed, B, H, W, self.static_margins(H, W), *self.fast_params(), with_draws=with_draws)

    def sample_G_inv(self, images, debug_percentile=None):
        """augment.py:191-268 for the geometric branches.  Returns a (B,3,3) float32 numpy array or None."""
        B, _, H, W = images.shape
        p = self._p_host()
        pct = None if debug_percentile is None else float(debug_percentile)
        f32 = np.float32
        G = None

        def mul(G, M):
            return M if G is None else (G @ M).astype(f32)

        rand = lambda *s: np.random.rand(*s).astype(f32)
        randn = lambda *s: np.random.randn(*s).astype(f32)
        if self._only_xint_scale:
            # the shipped configuration (train.yaml:80-85) in closed form -- same draws in the same order, same fp32 values as
            # the generic branches below (translate2d(-round(t W), -round(t H)) @ scale2d(1/s, 1/s)), a tenth of the host time:
            # at batch 1 the augmentation parameters cost more than the whole discriminator forward on the GPU
            if B <= 4 and pct is None:
                # tiny batches: the same fp32 operations on numpy SCALARS (an array op costs ~1 us whatever its size)
                r_t, r_tg = np.random.rand(B, 2), np.random.rand(B, 1)
                r_s, r_sg = np.random.randn(B), np.random.rand(B)
                G = np.zeros((B, 3, 3), f32)
                xm, ss, two, one = f32(self.xint_max), f32(self.scale_std), f32(2), f32(1)
                for i in range(B):
                    on = f32(r_tg[i, 0]) < self.xint * p
                    tx = (f32(r_t[i, 0]) * two - one) * xm if on else f32(0)
                    ty = (f32(r_t[i, 1]) * two - one) * xm if on else f32(0)
                    sc = np.exp2(f32(r_s[i]) * ss) if f32(r_sg[i]) < self.scale * p else one
                    G[i, 0, 0] = G[i, 1, 1] = one / sc
                    G[i, 0, 2] = -np.round(tx * f32(W))
                    G[i, 1, 2] = -np.round(ty * f32(H))
                    G[i, 2, 2] = 1
                return G
            t = (rand(B, 2) * 2 - 1) * f32(self.xint_max)
            t = np.where(rand(B, 1) < self.xint * p, t, 0).astype(f32)
            s = np.exp2(randn(B) * f32(self.scale_std))
            s = np.where(rand(B) < self.scale * p, s, 1).astype(f32)
            if pct is not None:
                t = np.full((B, 2), (f32(pct) * 2 - 1) * f32(self.xint_max), f32)
                s = np.full(B, np.exp2(f32(_erfinv(pct * 2 - 1)) * f32(self.scale_std)), f32)
            G = np.zeros((B, 3, 3), f32)
            G[:, 0, 0] = G[:, 1, 1] = f32(1) / s
            G[:, 0, 2] = -np.round(t[:, 0] * f32(W))
            G[:, 1, 2] = -np.round(t[:, 1] * f32(H))
            G[:, 2, 2] = 1
            return G
        if self.xflip > 0:
            i = np.floor(rand(B) * 2)
            i = np.where(rand(B) < self.xflip * p, i, 0).astype(f32)
            if pct is not None:
                i = np.full(B, np.floor(f32(pct) * 2), f32)
            G = mul(G, scale2d(1 / (1 - 2 * i), 1, B))
        if self.rotate90 > 0:
            i = np.floor(rand(B) * 4)
            i = np.where(rand(B) < self.rotate90 * p, i, 0).astype(f32)
            if pct is not None:
                i = np.full(B, np.floor(f32(pct) * 4), f32)
            G = mul(G, rotate2d((np.pi / 2 * i).astype(f32), B))
        if self.xint > 0:
            t = (rand(B, 2) * 2 - 1) * f32(self.xint_max)
            t = np.where(rand(B, 1) < self.xint * p, t, 0).astype(f32)
            if pct is not None:
                t = np.full((B, 2), (f32(pct) * 2 - 1) * f32(self.xint_max), f32)
            G = mul(G, translate2d(-np.round(t[:, 0] * f32(W)), -np.round(t[:, 1] * f32(H)), B))
        if self.scale > 0:
            s = np.exp2(randn(B) * f32(self.scale_std))
            s = np.where(rand(B) < self.scale * p, s, 1).astype(f32)
            if pct is not None:
                s = np.full(B, np.exp2(f32(_erfinv(pct * 2 - 1)) * f32(self.scale_std)), f32)
            G = mul(G, scale2d(1 / s, 1 / s, B))
        p_rot = 1 - np.sqrt(np.clip(1 - self.rotate * p, 0, 1))
        if self.rotate > 0:
            th = (rand(B) * 2 - 1) * f32(np.pi * self.rotate_max)
            th = np.where(rand(B) < p_rot, th, 0).astype(f32)
            if pct is not None:
                th = np.full(B, (f32(pct) * 2 - 1) * f32(np.pi * self.rotate_max), f32)
            G = mul(G, rotate2d(th, B))
        if self.aniso > 0:
            s = np.exp2(randn(B) * f32(self.aniso_std))
            s = np.where(rand(B) < self.aniso * p, s, 1).astype(f32)
            if pct is not None:
                s = np.full(B, np.exp2(f32(_erfinv(pct * 2 - 1)) * f32(self.aniso_std)), f32)
            G = mul(G, scale2d(1 / s, s, B))
        if self.rotate > 0:
            th = (rand(B) * 2 - 1) * f32(np.pi * self.rotate_max)
            th = np.where(rand(B) < p_rot, th, 0).astype(f32)
            if pct is not None:
                th = np.zeros(B, f32)
            G = mul(G, rotate2d(th, B))
        if self.xfrac > 0:
            t = randn(B, 2) * f32(self.xfrac_std)
            t = np.where(rand(B, 1) < self.xfrac * p, t, 0).astype(f32)
            if pct is not None:
                t = np.full((B, 2), f32(_erfinv(pct * 2 - 1)) * f32(self.xfrac_std), f32)
            G = mul(G, translate2d(-t[:, 0] * f32(W), -t[:, 1] * f32(H), B))
        return G

    def margins_for(self, G_inv, H, W):
        """Padding margins (mx0, my0, mx1, my1) of augment.py:272-283 from the host copy of G_inv."""
        f32 = np.float32
        cx, cy = (W - 1) / 2, (H - 1) / 2
        cp = np.array([[-cx, -cy, 1], [cx, -cy, 1], [cx, cy, 1], [-cx, cy, 1]], f32)
        cp = G_inv @ cp.T                                     # (B, 3, 4)
        Hz_pad = self.Hz_geom.shape[0] // 4
        m = cp[:, :2, :].transpose(1, 0, 2).reshape(2, -1)    # [xy, batch * idx]
        m = np.concatenate([-m, m]).max(axis=1)               # [x0, y0, x1, y1]
        m = m + np.array([Hz_pad * 2 - cx, Hz_pad * 2 - cy] * 2, f32)
        m = np.minimum(np.maximum(m, 0), np.array([W - 1, H - 1] * 2, f32))
        return tuple(int(v) for v in np.ceil(m))

    @staticmethod
    def static_margins(H, W):
        """The largest margins margins_for() can return (its own clamp): with them every intermediate shape is independent
        of the sampled transform -- what a captured hipGraph of the discriminator step needs.  The resampled pixels
        are the same ones (a larger reflect-padded canvas around the same image), up to the rounding of the normalised
        sampling coordinates."""
        return (W - 1, H - 1, W - 1, H - 1)

    def theta_for(self, G_inv, margins, H, W):
        """(B, 2, 3) float32 numpy: the affine sampling grid of augment.py:285-297 for given padding margins."""
        f32 = np.float32
        B = G_inv.shape[0]
        mx0, my0, mx1, my1 = margins
        Hz_pad = self.Hz_geom.shape[0] // 4
        if mx0 == mx1 and my0 == my1:
            # symmetric (e.g. static) margins: the first factor is the identity; the constant pre / post factors are composed
            # once per shape in fp64 (they are powers of two, +-0.5 shifts and one 2 / size scale), so theta = L @ G_inv @ R:
            # two products instead of seven (the fp32 chain below agrees to 1 ulp; at batch 1 this code, not the GPU, set the
            # discriminator's image rate)
            key = (mx0, my0, H, W)
            ab = self._theta_ab.get(key)
            if ab is None:
                d = lambda m_: m_[0].astype(np.float64)
                Ho, Wo = (H + Hz_pad * 2) * 2, (W + Hz_pad * 2) * 2
                Wp, Hp = (W + mx0 + mx1) * 2, (H + my0 + my1) * 2
                L = d(scale2d(2 / Wp, 2 / Hp, 1)) @ d(translate2d(-0.5, -0.5, 1)) @ d(scale2d(2, 2, 1))
                R = d(scale2d(0.5, 0.5, 1)) @ d(translate2d(0.5, 0.5, 1)) @ d(scale2d(Wo / 2, Ho / 2, 1))
                ab = self._theta_ab[key] = (L[:2], R)
            return np.ascontiguousarray((ab[0] @ G_inv.astype(np.float64) @ ab[1]).astype(f32))
        mm = lambda a, b: (a @ b).astype(f32)
        G_inv = mm(translate2d((mx0 - mx1) / 2, (my0 - my1) / 2, B), G_inv)
        G_inv = mm(mm(scale2d(2, 2, B), G_inv), scale2d(0.5, 0.5, B))
        G_inv = mm(mm(translate2d(-0.5, -0.5, B), G_inv), translate2d(0.5, 0.5, B))
        Ho, Wo = (H + Hz_pad * 2) * 2, (W + Hz_pad * 2) * 2
        Wp, Hp = (W + mx0 + mx1) * 2, (H + my0 + my1) * 2     # the padded, x2-upsampled canvas
        G_inv = mm(mm(scale2d(2 / Wp, 2 / Hp, B), G_inv), scale2d(Wo / 2, Ho / 2, B))
        return np.ascontiguousarray(G_inv[:, :2, :])

    def apply_theta(self, images, theta, margins):
        """reflect pad -> x2 sym6 upsample -> affine bilinear resample -> /2 sym6 downsample (augment.py:284-301) for a
        sampling grid `theta` that is already on the device."""
        B, C, H, W = images.shape
        mx0, my0, mx1, my1 = margins
        Hz_pad = self.Hz_geom.shape[0] // 4
        if FUSED and self.Hz_geom.shape[0] == 12 and B * C <= 65535:
            # the same four stages in two launches -- or in one, when this pipe cannot draw a rotation (every theta it forms is
            # then axis-aligned: xflip / xint / scale / aniso / xfrac only scale and shift the axes)
            return ada_geom(images, theta, self.Hz_geom, margins, axis_aligned=self.rotate90 == 0 and self.rotate == 0)
        x = reflect_pad(images, mx0, mx1, my0, my1)
        x = upfirdn2d_separable(x, self.Hz_geom, up=2, pad=(6, 5, 6, 5), flip=False, gain=4.0)  # upsample2d
        Ho, Wo = (H + Hz_pad * 2) * 2, (W + Hz_pad * 2) * 2
        x = affine_grid_sample(x, theta, Ho, Wo)
        # downsample2d(padding=-2*Hz_pad, flip_filter=True): pad = -6 + (12-2+1)//2 = -1, -6 + 5 = -1
        return upfirdn2d_separable(x, self.Hz_geom, down=2, pad=(-1, -1, -1, -1), flip=True, gain=1.0)

    def forward(self, images, debug_percentile=None, theta=None):
        """`theta`: (B, 2, 3) device tensor built by the caller with `theta_for(sample_G_inv(...), static_margins(H, W), H, W)`
        -- the shape-static form used under hipGraph capture (oi_amd.graphed.GraphedDStep); default: the reference's
        flow with margins fitted to the sampled transform.  A pipe without `rotate` / `rotate90` takes the one-launch separable
        kernel, which does not read theta[:, 0, 1] / theta[:, 1, 0]: a caller-built `theta` must come from THIS pipe's draws."""
        assert isinstance(images, torch.Tensor) and images.ndim == 4
        B, C, H, W = images.shape
        if theta is not None:
            return self.apply_theta(images, theta, self.static_margins(H, W))
        G_inv = self.sample_G_inv(images, debug_percentile)
        if G_inv is None:
            return images
        margins = self.margins_for(G_inv, H, W)
        th = self.theta_for(G_inv, margins, H, W)
        if images.is_cuda and th.size <= 64:   # (up to 10 images: in the arguments of one small launch, no pageable copy)
            from . import ops
            th = ops.upload_small(th, images.device)
        else:
            th = torch.from_numpy(th).to(images.device, non_blocking=True)
        return self.apply_theta(images, th, margins)
