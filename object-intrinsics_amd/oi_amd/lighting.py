"""Learnable directional Phong light.  Parameter names/shapes mirror
DirectionalLightWithSpecularFixInit (src/models/lighting.py:6-76) so state_dicts interchange; the
shading arithmetic itself runs inside the fused compositing kernel (csrc/render.hip), which reads
`param_ambient/param_specular/param_shininess` straight from device memory (no .item() syncs)."""
import numpy as np
import torch
import torch.nn as nn

from .pose import look_at_rot


class DirectionalLightWithSpecularFixInit(nn.Module):
    def __init__(self, direction, ambient_color=0.33, diffuse_color=0.66, specular_color=0.01, shininess=10):
        super().__init__()
        ratio = ambient_color / (ambient_color + diffuse_color)
        self.param_ambient = nn.Parameter(torch.tensor(ratio, dtype=torch.float32).logit())
        self.param_direction = nn.Parameter(torch.as_tensor(np.asarray(direction), dtype=torch.float32))
        self.param_shininess = nn.Parameter(torch.tensor(float(shininess)))
        self.param_specular = nn.Parameter(torch.tensor(float(specular_color)))

    @property
    def specular_color(self):
        return self.param_specular.expand(3).clamp(min=0)

    @property
    def ambient_color(self):
        return torch.sigmoid(self.param_ambient).expand(3)

    @property
    def diffuse_color(self):
        return (1 - torch.sigmoid(self.param_ambient)).expand(3)

    @property
    def shininess(self):
        return self.param_shininess

    @property
    def direction(self):
        return self.param_direction / torch.linalg.norm(self.param_direction)

    def packed(self):
        """[param_ambient, param_specular, param_shininess] as the kernel reads them."""
        if torch.is_grad_enabled():
            return torch.stack([self.param_ambient, self.param_specular, self.param_shininess])
        return self._cached("packed", lambda: torch.stack([self.param_ambient, self.param_specular, self.param_shininess]))

    def stats(self):
        """(ambient, diffuse, specular) scalars for logging -- detached; lighting.py:50-60 up to the expand(3)."""
        def build():
            amb = torch.sigmoid(self.param_ambient.detach())
            return amb, 1 - amb, self.param_specular.detach().clamp(min=0)
        return self._cached("stats", build)

    def _cached(self, name, build):
        """Forward-only scalar glue, rebuilt when a parameter changes (in-place updates bump Tensor._version; the
        fused optimisers count their writes, oi_amd.optim._written): a handful of 5 us launches per render otherwise."""
        ps = (self.param_ambient, self.param_specular, self.param_shininess)
        key = tuple((p.data_ptr(), p._version) for p in ps)
        cache = self.__dict__.setdefault("_glue_cache", {})
        hit = cache.get(name)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = cache[name] = (key, build())
        return hit[1]

    def batch_direction(self, w2b):
        """(B,3) light direction in each box frame (lighting.py:115-119)."""
        return torch.einsum("bij,j->bi", w2b[:, :3, :3], self.direction)

    def batch_direction_unit(self, w2b):
        """normalize(batch_direction(w2b)) with the gradient back to `param_direction`, one launch each way
        (oi_light_dir_fwd / _bwd) -- what the compositing kernel consumes.  CUDA tensors; w2b carries no gradient (poses
        are sampled)."""
        return _LightDir.apply(self.param_direction, w2b.detach())

    def batch_transform(self, *, w2b):
        return BatchLight(self, w2b)


class PackLight(torch.autograd.Function):
    """[param_ambient, param_specular, param_shininess] -> the (3,) block the compositing kernel reads, WITHOUT a launch: the block
    was written by oi_scalar_glue (oi_amd.generator.Generator._glue); this node only routes the gradient back (torch.stack: a
    copy kernel per call)."""

    @staticmethod
    def forward(ctx, amb, spec, shin, packed):
        return packed.view(3)

    @staticmethod
    def backward(ctx, g):
        return g[0], g[1], g[2], None


class _LightDir(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d, w2b):
        from . import ops
        ctx.save_for_backward(d, w2b)
        return ops.light_dir_fwd(d, w2b)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_n):
        from . import ops
        d, w2b = ctx.saved_tensors
        return ops.light_dir_bwd(d, w2b, g_n).view_as(d), None


class BatchLight:
    """Counterpart of BatchDirectionalLightWithSpecularFixInit (lighting.py:79-119): just carries w2b."""

    def __init__(self, light, w2b):
        self.light, self.w2b = light, w2b

    @property
    def ambient_color(self):
        return self.light.ambient_color

    def direction(self):
        return self.light.batch_direction(self.w2b)


def build_directional_light_optimizable(cam_loc, light_loc, ambient_color=0.33, diffuse_color=0.66,
                                        specular_color=0, shininess=10):
    """src/utils/prior.py:32-49."""
    if cam_loc is None and light_loc is None:
        cam_loc, light_loc = [0, 0, -1], [0, 0, -1]
    dw = np.array(light_loc) / np.linalg.norm(light_loc)
    dc = look_at_rot(cam_loc).T @ dw
    return DirectionalLightWithSpecularFixInit(direction=dc, ambient_color=ambient_color, diffuse_color=diffuse_color,
                                               specular_color=specular_color, shininess=shininess)
