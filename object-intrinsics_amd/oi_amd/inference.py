"""Inference driver of the path: camera walks and latent walks at a resolution / depth multiple of the
training configuration, rendered in ray chunks (the second caller of Generator.forward in the reference:
scripts/test.py:231-244, 274-281; src/utils/test.py:55-66, 131-155).  Returns frame tensors; writing
mp4/html is the reference's visualisation stack and out of scope."""
import math

import numpy as np
import torch

from . import generator as G


def scale_config(gen_kwargs, cfg_resolution, test_resolution=None, depth_multiplier=None):
    """update_config of the reference (src/utils/test.py:55-66): multiply n_samples / n_importance, set resolution.
    `gen_kwargs` is the generator's kwargs dict (mutated copy returned)."""
    import copy
    kw = copy.deepcopy(gen_kwargs)
    if depth_multiplier is not None:
        r = kw["renderer"]["kwargs"]
        r["n_importance"] = r["n_importance"] * depth_multiplier
        r["n_samples"] = r["n_samples"] * depth_multiplier
    if test_resolution is not None:
        ratio = test_resolution / cfg_resolution
        kw["resolution"] = int(cfg_resolution * ratio)
        kw["scene_resolution"] = int(kw["scene_resolution"] * ratio)
        kw["camera"]["kwargs"]["resolution"] = kw["scene_resolution"]
    return kw


def slerp(a, b, t):
    """Spherical interpolation of latent codes (src/utils/slerp.py)."""
    an, bn = a / a.norm(dim=-1, keepdim=True), b / b.norm(dim=-1, keepdim=True)
    omega = torch.acos((an * bn).sum(-1, keepdim=True).clamp(-1, 1))
    so = torch.sin(omega)
    return torch.where(so.abs() < 1e-6, (1 - t) * a + t * b, torch.sin((1 - t) * omega) / so * a + torch.sin(t * omega) / so * b)


def rotation_walk(b2w0, n_frames, axis=(0.0, -1.0, 0.0)):
    """b2w poses rotating the object about `axis` through 360 degrees (camera walk of scripts/test.py:231-244)."""
    from scipy.spatial.transform import Rotation as R
    out = []
    ax = np.asarray(axis, dtype=np.float64)
    for i in range(n_frames):
        rot = R.from_rotvec(ax * (2 * math.pi * i / n_frames)).as_matrix()
        m = b2w0.clone()
        m[:3, :3] = b2w0[:3, :3] @ torch.tensor(rot, dtype=torch.float32)
        out.append(m)
    return torch.stack(out)


@torch.no_grad()
def render_frames(gen, zs, b2ws, keys=("image", "mask", "normal_map", "shading_map"), max_ray_batch=None, graphed=False):
    """One frame per (z, b2w) pair, eval mode (perturb off, multi-chunk allowed: generator.py:286-305).
    graphed=True replays one captured hipGraph per frame (oi_amd.graphed.GraphedForward; background fixed to black)."""
    gen.eval()
    gen.renderer.pack.check()  # inf / NaN weights (a broken checkpoint) are reported here, once, not as NaN frames
    if graphed:
        from .graphed import GraphedForward
        old = G.MAX_RAY_BATCH_SIZE
        if max_ray_batch is not None:
            G.MAX_RAY_BATCH_SIZE = max_ray_batch
        try:
            gf = GraphedForward(gen, bs=1, it=gen.iteration(), return_raw=True, keys=keys).recapture()
            dev = gen.it.device
            frames = {k: [] for k in keys}
            for z, b2w in zip(zs, b2ws):
                out = gf(b2w[None].to(dev), z[None].to(dev))
                for k in keys:
                    frames[k].append(out[k][0].clone())
            return {k: torch.stack(v) for k, v in frames.items()}
        finally:
            G.MAX_RAY_BATCH_SIZE = old
    old = G.MAX_RAY_BATCH_SIZE
    if max_ray_batch is not None:
        G.MAX_RAY_BATCH_SIZE = max_ray_batch
    try:
        frames = {k: [] for k in keys}
        dev = gen.it.device
        for z, b2w in zip(zs, b2ws):
            blob = gen(bs=1, it=None, data={"z": z[None].to(dev), "b2w": b2w[None].to(dev)}, return_raw=True)["box"]
            for k in keys:
                frames[k].append(blob["render_out"][k][0])
        return {k: torch.stack(v) for k, v in frames.items()}
    finally:
        G.MAX_RAY_BATCH_SIZE = old


def camera_walk(gen, z, b2w0, n_frames=128, **kw):
    return render_frames(gen, [z] * n_frames, rotation_walk(b2w0, n_frames), **kw)


def latent_walk(gen, z0, z1, b2w, n_frames=128, **kw):
    ts = torch.linspace(0, 1, n_frames)
    return render_frames(gen, [slerp(z0, z1, t) for t in ts], [b2w] * n_frames, **kw)
