"""StyleGAN2-ADA geometric augmentation -- drop-in for src.third_party.ada.augment.AugmentPipe
(augment.py:117-429) for the branches the path enables (configs/train.yaml:80-85: xint, scale; plus
xflip / rotate90 / rotate / aniso / xfrac which only change the 3x3 matrix).  Colour, image-filter,
noise and cutout branches are dead at this configuration and raise if enabled.

Execution (augment.py:270-301) = reflect pad -> x2 sym6 upsample -> affine bilinear resample -> /2
sym6 downsample, all HIP kernels (csrc/disc.hip) wrapped in autograd Functions whose backward is
built from the same kernels (arbitrary order, as the reference's upfirdn2d / grid_sample_gradfix)."""
import numpy as np
import torch

from .autograd_disc import affine_grid_sample, reflect_pad, upfirdn2d_separable

SYM6 = [0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633,
        0.4910559419267466, 0.787641141030194, 0.3379294217276218, -0.07263752278646252,
        -0.021060292512300564, 0.04472490177066578, 0.0017677118642428036, -0.007800708325034148]


def _mat(rows, ref):
    """3x3 matrices from python scalars / (B,) tensors, batched over ref's batch.  Host tensors: the
    augmentation parameters are O(B) scalars, so they are sampled and composed on the CPU (no launches,
    and the data-dependent padding margins need no device->host read, unlike augment.py:283)."""
    B = ref.shape[0]
    out = torch.zeros(B, 3, 3, dtype=torch.float32)
    for i, row in enumerate(rows):
        for j, v in enumerate(row):
            out[:, i, j] = v
    return out


def translate2d(tx, ty, ref):
    return _mat([[1, 0, tx], [0, 1, ty], [0, 0, 1]], ref)


def scale2d(sx, sy, ref):
    return _mat([[sx, 0, 0], [0, sy, 0], [0, 0, 1]], ref)


def rotate2d(theta, ref):
    return _mat([[torch.cos(theta), torch.sin(-theta), 0], [torch.sin(theta), torch.cos(theta), 0], [0, 0, 1]], ref)


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

    def sample_G_inv(self, images, debug_percentile=None):
        """augment.py:191-268 for the geometric branches.  Returns (B,3,3) or None (identity)."""
        B, _, H, W = images.shape
        dev = torch.device("cpu")
        ref = images
        p_host = float(self._p_host()) if not isinstance(self.p, float) else self.p
        pct = None if debug_percentile is None else torch.as_tensor(debug_percentile, dtype=torch.float32)
        G = None

        def mul(G, M):
            return M if G is None else G @ M

        rand = lambda *s: torch.rand(list(s), device=dev)
        randn = lambda *s: torch.randn(list(s), device=dev)
        if self.xflip > 0:
            i = torch.floor(rand(B) * 2)
            i = torch.where(rand(B) < self.xflip * p_host, i, torch.zeros_like(i))
            if pct is not None:
                i = torch.full_like(i, torch.floor(pct * 2))
            G = mul(G, scale2d(1 / (1 - 2 * i), 1, ref))
        if self.rotate90 > 0:
            i = torch.floor(rand(B) * 4)
            i = torch.where(rand(B) < self.rotate90 * p_host, i, torch.zeros_like(i))
            if pct is not None:
                i = torch.full_like(i, torch.floor(pct * 4))
            G = mul(G, rotate2d(np.pi / 2 * i, ref))
        if self.xint > 0:
            t = (rand(B, 2) * 2 - 1) * self.xint_max
            t = torch.where(rand(B, 1) < self.xint * p_host, t, torch.zeros_like(t))
            if pct is not None:
                t = torch.full_like(t, (pct * 2 - 1) * self.xint_max)
            G = mul(G, translate2d(-torch.round(t[:, 0] * W), -torch.round(t[:, 1] * H), ref))
        if self.scale > 0:
            s = torch.exp2(randn(B) * self.scale_std)
            s = torch.where(rand(B) < self.scale * p_host, s, torch.ones_like(s))
            if pct is not None:
                s = torch.full_like(s, torch.exp2(torch.erfinv(pct * 2 - 1) * self.scale_std))
            G = mul(G, scale2d(1 / s, 1 / s, ref))
        p_rot = 1 - torch.sqrt(torch.tensor(1 - self.rotate * p_host).clamp(0, 1))
        if self.rotate > 0:
            th = (rand(B) * 2 - 1) * np.pi * self.rotate_max
            th = torch.where(rand(B) < p_rot, th, torch.zeros_like(th))
            if pct is not None:
                th = torch.full_like(th, (pct * 2 - 1) * np.pi * self.rotate_max)
            G = mul(G, rotate2d(th, ref))
        if self.aniso > 0:
            s = torch.exp2(randn(B) * self.aniso_std)
            s = torch.where(rand(B) < self.aniso * p_host, s, torch.ones_like(s))
            if pct is not None:
                s = torch.full_like(s, torch.exp2(torch.erfinv(pct * 2 - 1) * self.aniso_std))
            G = mul(G, scale2d(1 / s, s, ref))
        if self.rotate > 0:
            th = (rand(B) * 2 - 1) * np.pi * self.rotate_max
            th = torch.where(rand(B) < p_rot, th, torch.zeros_like(th))
            if pct is not None:
                th = torch.zeros_like(th)
            G = mul(G, rotate2d(th, ref))
        if self.xfrac > 0:
            t = randn(B, 2) * self.xfrac_std
            t = torch.where(rand(B, 1) < self.xfrac * p_host, t, torch.zeros_like(t))
            if pct is not None:
                t = torch.full_like(t, torch.erfinv(pct * 2 - 1) * self.xfrac_std)
            G = mul(G, translate2d(-t[:, 0] * W, -t[:, 1] * H, ref))
        return G

    def forward(self, images, debug_percentile=None):
        assert isinstance(images, torch.Tensor) and images.ndim == 4
        B, C, H, W = images.shape
        G_inv = self.sample_G_inv(images, debug_percentile)
        if G_inv is None:
            return images
        ref = images
        # padding margins (augment.py:272-283), computed on the host copy of G_inv
        cx, cy = (W - 1) / 2, (H - 1) / 2
        cp = torch.tensor([[-cx, -cy, 1], [cx, -cy, 1], [cx, cy, 1], [-cx, cy, 1]])
        cp = G_inv @ cp.t()
        Hz_pad = self.Hz_geom.shape[0] // 4
        m = cp[:, :2, :].permute(1, 0, 2).flatten(1)
        m = torch.cat([-m, m]).max(dim=1).values
        m = m + torch.tensor([Hz_pad * 2 - cx, Hz_pad * 2 - cy] * 2)
        m = m.max(torch.zeros(4)).min(torch.tensor([W - 1, H - 1] * 2, dtype=torch.float32))
        mx0, my0, mx1, my1 = (int(v) for v in m.ceil().to(torch.int32).tolist())

        x = reflect_pad(images, mx0, mx1, my0, my1)
        G_inv = translate2d((mx0 - mx1) / 2, (my0 - my1) / 2, ref) @ G_inv
        x = upfirdn2d_separable(x, self.Hz_geom, up=2, pad=(6, 5, 6, 5), flip=False, gain=4.0)  # upsample2d
        G_inv = scale2d(2, 2, ref) @ G_inv @ scale2d(0.5, 0.5, ref)
        G_inv = translate2d(-0.5, -0.5, ref) @ G_inv @ translate2d(0.5, 0.5, ref)
        Ho, Wo = (H + Hz_pad * 2) * 2, (W + Hz_pad * 2) * 2
        G_inv = scale2d(2 / x.shape[3], 2 / x.shape[2], ref) @ G_inv @ scale2d(Wo / 2, Ho / 2, ref)
        theta = G_inv[:, :2, :].contiguous().to(images.device, non_blocking=True)
        x = affine_grid_sample(x, theta, Ho, Wo)
        # downsample2d(padding=-2*Hz_pad, flip_filter=True): pad = -6 + (12-2+1)//2 = -1, -6 + 5 = -1
        return upfirdn2d_separable(x, self.Hz_geom, down=2, pad=(-1, -1, -1, -1), flip=True, gain=1.0)
