"""Generator -- drop-in for src.models.generator.Generator (generator.py:19-314): same constructor
(config dicts with `__target__`), same `forward(bs, it, data, return_raw)` contract and returned
structure {'box': {'loss', 'stats', 'render_out', 'prior_info' [, 'latent_info', 'rays_info',
'raw_render_out']}}.

MI355X-first differences: the whole image is rendered by ~8 kernel launches
with no host synchronisation -- `it` is mirrored on the host, the light scalars are read by the
compositing kernel from device memory, stats stay on the device -- and the Phong maps come out of
the same compositing launch instead of ~40 elementwise kernels over (N, T, 3) tensors."""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .config import build_from_config
from .lib import PREP_MAX_B
from .renderer import assemble_render_dict

import weakref

MAX_RAY_BATCH_SIZE = 128 * 128 * 1
# round 6, the no-grad fused forward (OI_STEP_TAIL=0 switches both off: the round-5 launch sequence, 13 per step):
ONE_DRAW = os.environ.get("OI_STEP_TAIL", "1") != "0"          # latents + per-ray jitter from ONE generator launch
F3_BLOB_IN_PREP = os.environ.get("OI_STEP_TAIL", "1") != "0"   # the f16x3 kernel's per-element blobs formed by the prep launch
_TORCH_RAND, _TORCH_RANDN = torch.rand, torch.randn           # (a caller that patches them replays recorded draws: see _prep_fused)
_STAGE_RINGS = weakref.WeakKeyDictionary()


def invert_rot_t(pose):
    """src/utils/pose.py:143-154 (rigid inverse)."""
    R = pose[..., :3, :3].transpose(-1, -2)
    t = -(R @ pose[..., :3, 3:4])
    out = torch.zeros_like(pose)
    out[..., :3, :3] = R
    out[..., :3, 3:4] = t
    out[..., 3, 3] = 1.0
    return out


class Generator(nn.Module):
    def __init__(self, color_network, sdf_network, deviation_network, light_network, camera, z_dim, resolution,
                 scene_resolution, renderer, anneal_end, pose_prior):
        super().__init__()
        self.resolution = resolution
        self.scene_resolution = scene_resolution
        self.z_dim = z_dim
        self.anneal_end = anneal_end
        self.register_buffer("it", torch.tensor(-1, dtype=torch.long))
        # The reference stores the iteration in the buffer on every forward (generator.py:187: one fill launch per
        # render).  Here the host copy is authoritative and the buffer is written when somebody can observe it:
        # state_dict() (checkpoints, EMA buffer copies go through `sync_it`).
        self._it_host, self._it_dirty, self._it_buf, self._it_ver = -1, False, self.it, self.it._version
        self.register_state_dict_pre_hook(lambda module, prefix, keep_vars: module.sync_it())
        self.camera = build_from_config(camera)
        self.light = build_from_config(light_network)
        self.pose_prior = build_from_config(pose_prior)
        self.color_network = build_from_config(color_network)
        self.sdf_network = build_from_config(sdf_network)
        self.deviation_network = build_from_config(deviation_network)
        self.renderer = build_from_config(renderer, nerf=None, sdf_network=self.sdf_network,
                                          deviation_network=self.deviation_network, color_network=self.color_network)

    def _it_external_write(self):
        buf = self.it
        if buf is not self._it_buf:  # .to() / .cuda() re-created the buffer (and it may have been written since)
            self._it_buf = buf
            if not self._it_dirty:
                return True
            self._it_ver = buf._version  # a value pending on the host wins
        return buf._version != self._it_ver

    def sync_it(self):
        """Write the host-side iteration counter into the `it` buffer (no-op when it is current)."""
        if self._it_dirty and not self._it_external_write():
            self.it.fill_(self._it_host)
            self._it_dirty, self._it_ver = False, self.it._version

    def iteration(self):
        """The current iteration: the host copy, unless the buffer was written from outside since (load_state_dict,
        EMA buffer copies, `gen.it.fill_`: all bump its version counter) -- then one D2H read."""
        if self._it_external_write():
            self._it_host, self._it_dirty, self._it_ver = int(self.it), False, self.it._version
        return self._it_host

    # -- host-side sampling (generator.py:65-78, 176-184; prior.py:11-29) -------------------------
    def _h2d(self, arr):
        """numpy -> device through a small ring of pinned staging buffers with a non-blocking copy.
        `torch.tensor(x, device='cuda')` (what the reference does, generator.py:71,161) is a pageable copy:
        it serialises the host behind everything already queued on the stream, once per render."""
        dev = self.it.device
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        if dev.type != "cuda":
            return torch.from_numpy(arr).to(dev)
        if arr.size <= 64:   # (batches of one: the whole pose block travels in the arguments of one small launch)
            return ops.upload_small(arr, dev)
        ring = _STAGE_RINGS.setdefault(self, {"bufs": [], "events": [], "i": 0})  # not in __dict__: deepcopy (EMA)
        n = arr.size
        if not ring["bufs"] or ring["bufs"][0].numel() < n:
            ring["bufs"] = [torch.empty(max(n, 256), dtype=torch.float32, pin_memory=True) for _ in range(8)]
            ring["events"] = [None] * 8
        i = ring["i"] = (ring["i"] + 1) % 8
        if ring["events"][i] is not None and not ring["events"][i].query():
            ring["events"][i].synchronize()  # the copy that last used this buffer has not executed yet (rare)
        buf = ring["bufs"][i][:n]
        buf.numpy()[:] = arr.reshape(-1)
        out = buf.to(dev, non_blocking=True).view(arr.shape)
        ev = torch.cuda.Event()
        ev.record()
        ring["events"][i] = ev
        return out

    def bg_color(self, bs):
        return np.random.uniform(low=0, high=1, size=(bs, 3))

    def sample_prior(self, bs, data):
        dev = self.it.device
        if "b2w" in data:
            assert not self.training
            b2w = data["b2w"].to(dev)
        else:
            # Poses come from the host RNG: do the 4x4 algebra that depends only on them (rigid inverse, camera-to-box,
            # crop offsets; ~20 tiny device kernels in the reference) on the host in fp32 and ship ONE staged buffer.
            b2w_h, w2b_h, c2b_h, xy_h, bg_h = self._sample_prior_host(bs, data)
            parts = [b2w_h.ravel(), w2b_h.ravel(), c2b_h.ravel(), xy_h.ravel()] + ([] if bg_h is None else [bg_h.ravel()])
            return self._prior_from_flat(self._h2d(np.concatenate(parts)), bs, bg_h is not None)
        self._xy_off = self._bg_dev = None
        w2b = invert_rot_t(b2w)
        c2b = torch.einsum("bij,jk->bik", w2b, self.camera.c2w)
        return {"c2b": c2b, "b2w": b2w, "w2b": w2b, "light": self.light.batch_transform(w2b=w2b)}

    def _sample_prior_host(self, bs, data):
        """The host half of sample_prior: poses from the numpy RNG and the 4x4 algebra that depends only on them."""
        b2w_h = np.ascontiguousarray(self.pose_prior(bs), dtype=np.float32)
        cam = self._camera_host()
        Rt = b2w_h[:, :3, :3].transpose(0, 2, 1)
        w2b_h = np.zeros_like(b2w_h)
        w2b_h[:, :3, :3] = Rt
        w2b_h[:, :3, 3:4] = -(Rt @ b2w_h[:, :3, 3:4])
        w2b_h[:, 3, 3] = 1.0
        c2b_h = w2b_h @ cam["c2w"]
        xy_h = self._crop_offsets_host(b2w_h, cam)
        # the background colour is the next numpy draw of the forward (generator.py:161): same order, same upload
        bg_h = None if "bg_color" in data else np.asarray(self.bg_color(bs), dtype=np.float32)
        return b2w_h, w2b_h, c2b_h, xy_h, bg_h

    def _prior_from_flat(self, flat, bs, has_bg):
        """Views of the device pose block  b2w [bs][16] | w2b | c2b | offs [bs][2] | bg [bs][3]  -> prior_info."""
        n = bs * 16
        b2w, w2b, c2b = flat[:n].view(bs, 4, 4), flat[n:2 * n].view(bs, 4, 4), flat[2 * n:3 * n].view(bs, 4, 4)
        self._xy_off = flat[3 * n:3 * n + 2 * bs].view(bs, 2)
        self._bg_dev = flat[3 * n + 2 * bs:3 * n + 5 * bs].view(bs, 3) if has_bg else None
        return {"c2b": c2b, "b2w": b2w, "w2b": w2b, "light": self.light.batch_transform(w2b=w2b)}

    def _glue(self):
        """Scalar glue of a forward -- inv_s / s_val, the three logging colours of the light, the light block the compositing
        kernel reads -- in ONE launch per parameter version (ops.scalar_glue; as tensor ops: exp, clamp, reciprocal, two muls,
        sigmoid, rsub, clamp and a stack).  -> (out5, packed3)"""
        lt = self.light
        ps = (self.deviation_network.variance, lt.param_ambient, lt.param_specular, lt.param_shininess)
        key = tuple((p.data_ptr(), p._version) for p in ps)
        hit = self.__dict__.get("_glue_cache")
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = self.__dict__["_glue_cache"] = (key,) + tuple(ops.scalar_glue(*ps))
        return hit[1], hit[2]

    def _kinv(self, dev):
        ki = self.camera.intrinsics_inv
        kkey = (ki.data_ptr(), ki._version)
        kinv = getattr(self, "_kinv33", None)
        if kinv is None or kinv.device != dev or getattr(self, "_kinv33_key", None) != kkey:
            kinv, self._kinv33_key = ki[:3, :3].contiguous(), kkey
            self._kinv33 = kinv
        return kinv

    def _prep_fused(self, bs, data):
        """No-grad forward with host-sampled poses: pose upload, rays, light direction, style MLP + FiLM parameters and the
        coarse samples in ONE launch (ops.prep_render); the torch draws keep their order (latent, then the per-ray jitter).
        -> (prior, latent, rays, film, coarse, f3_scratch)"""
        dev = self.it.device
        b2w_h, w2b_h, c2b_h, xy_h, bg_h = self._sample_prior_host(bs, data)
        R, S = self.resolution, self.renderer.n_samples
        perturb = self.renderer.perturb if self.training else 0
        jitter_normal = False
        if (ONE_DRAW and perturb > 0 and "sample_latent" not in self.__dict__ and type(self).sample_latent is Generator.sample_latent
                and torch.rand is _TORCH_RAND and torch.randn is _TORCH_RANDN):
            # ONE generator launch for the forward's two draws (latents, then one value per ray for the jitter -- the reference's
            # order, generator.py:234 / renderer.py:372): standard normals; the prep kernel maps a ray's value to its uniform
            # through the normal CDF.  (A caller that replays recorded draws -- F13 patches sample_latent / torch.rand -- keeps the
            # two separate calls.)
            buf = torch.randn(bs * self.z_dim + bs * R * R, device=dev)
            latent = {"z": buf[:bs * self.z_dim].view(bs, self.z_dim)}
            jitter, jitter_normal = buf[bs * self.z_dim:], True
        else:
            latent = self.sample_latent(bs, data)
            jitter = torch.rand([bs * R * R, 1], device=dev) if perturb > 0 else None   # renderer.py:372
        # f16x3: the per-element table blobs of the fine pass's MLP kernel are formed by this launch's FiLM workgroups (one
        # launch and its boundary less per render): the fine pass's working memory is allocated here
        pack, f3_scratch, f3_blob, f3_packed = self.renderer.pack, None, None, None
        if F3_BLOB_IN_PREP and pack.prec == ops._l.OI_PREC_F16X3 and pack.color_network is not None:
            I, K = self.renderer.n_importance, max(1, self.renderer.up_sample_steps)
            T = S + (I // K) * K if I > 0 else S
            f3_scratch, f3_blob = ops.f3_scratch_for(bs, R * R * T, dev)
            f3_packed = pack.packed()
        pre = ops.prep_render(b2w_h, w2b_h, c2b_h, xy_h, bg_h, self._kinv(dev), R, S, jitter, self.light.param_direction,
                              pack.film_stacked(differentiable=False), latent["z"], jitter_normal=jitter_normal,
                              f3_packed=f3_packed, f3_blob=f3_blob)
        prior = self._prior_from_flat(pre["pose"], bs, True)
        rays = {"x_offset": self._xy_off[:, 0], "y_offset": self._xy_off[:, 1], "light_dir": pre["light_dir"],
                "rays_o": pre["rays_o"], "rays_d": pre["rays_d"], "near": pre["near"], "far": pre["far"]}
        return prior, latent, rays, (pre["w"], pre["gamma"], pre["beta"]), (pre["z_coarse"], pre["pts_coarse"]), f3_scratch

    def _camera_host(self):
        # host copies of the camera buffers, keyed on (address, version) of the device buffers: load_state_dict / .to()
        # overwrite them (the reference's load path does, src/utils/test.py:update_legacy_state_dict) and the host-side
        # training geometry must follow
        key = tuple((t.data_ptr(), t._version) for t in (self.camera.c2w, self.camera.w2c))
        cam = getattr(self, "_cam_np", None)
        if cam is None or cam["key"] != key:
            cam = self._cam_np = {"key": key, "c2w": self.camera.c2w.detach().cpu().numpy().astype(np.float32),
                                  "w2c": self.camera.w2c.detach().cpu().numpy().astype(np.float32)}
        return cam

    def _crop_offsets_host(self, b2w_h, cam):
        """(x_offset, y_offset) of generator.py:262-268 in fp32 numpy, same operation order as the device path below."""
        R = np.float32(self.resolution)
        t = (cam["w2c"] @ b2w_h)[:, :3, 3]
        cd, half = np.float32(self.camera.cam_dist), np.float32(0.5 * self.scene_resolution)
        cx = cd / t[:, 2] * t[:, 0] * R / np.float32(2) + half
        cy = cd / t[:, 2] * t[:, 1] * R / np.float32(2) + half
        return np.stack([cx - R / np.float32(2), cy - R / np.float32(2)], -1).astype(np.float32)

    def sample_latent(self, bs, data):
        if "w" in data:
            assert not self.training
            return {"z": data["z"], "w": data["w"]}
        if "z" in data:
            assert not self.training
            return {"z": data["z"]}
        return {"z": torch.randn(bs, self.z_dim, device=self.it.device)}

    def gen_rays_at(self, data, prior_info, with_light=False):
        """generator.py:255-279 + build_rays :317-333 + near_far_from_sphere :336-342 (one kernel).  `with_light`: the
        same launch also evaluates prior_info["light"].direction() (lighting.py:115-119) -> "light_dir"."""
        b2w, R = prior_info["b2w"], self.resolution
        xy = getattr(self, "_xy_off", None)
        if xy is None:  # poses given on the device (eval / inference): the reference's tensor arithmetic
            b2c_t = torch.einsum("ij,bjk->bik", self.camera.w2c, b2w)[..., :3, 3]
            cx = self.camera.cam_dist / b2c_t[..., 2] * b2c_t[..., 0] * R / 2 + 0.5 * self.scene_resolution
            cy = self.camera.cam_dist / b2c_t[..., 2] * b2c_t[..., 1] * R / 2 + 0.5 * self.scene_resolution
            xy = torch.stack([cx - R / 2, cy - R / 2], -1)
        x_off, y_off = xy[:, 0], xy[:, 1]
        kinv = self._kinv(b2w.device)
        out = {"x_offset": x_off, "y_offset": y_off}
        if with_light:
            ro, rd, near, far, out["light_dir"] = ops.gen_rays(prior_info["c2b"], kinv, xy, R, w2b=prior_info["w2b"],
                                                               light_direction=self.light.param_direction)
        else:
            ro, rd, near, far = ops.gen_rays(prior_info["c2b"], kinv, xy, R)
        out.update({"rays_o": ro, "rays_d": rd, "near": near, "far": far})
        return out

    # -- forward --------------------------------------------------------------------------------
    def forward(self, bs, it, data, return_raw=False):
        pack = self.renderer.pack
        pack.hold(True)   # (parameters cannot change inside one forward: one version walk over ~65 parameters instead of four)
        try:
            return self._forward(bs, it, data, return_raw)
        finally:
            pack.hold(False)

    def _forward(self, bs, it, data, return_raw):
        if it is None:  # eval / inference callers only (one D2H read after a checkpoint load); training passes `it`
            it = self.iteration()
        if int(it) != self.iteration():
            self._it_host, self._it_dirty = int(it), True
        h = w = self.resolution
        n_rays = bs * h * w
        film = coarse = None
        fused = (not torch.is_grad_enabled() and self.it.is_cuda and bs <= PREP_MAX_B and bs * h * w <= MAX_RAY_BATCH_SIZE
                 and not any(k in data for k in ("b2w", "z", "w", "bg_color")))
        f3_scratch = None
        if fused:
            prior, latent, rays, film, coarse, f3_scratch = self._prep_fused(bs, data)
            grad_light = False
        else:
            prior = self.sample_prior(bs, data)
            latent = self.sample_latent(bs, data)
            # the light direction needs the tensor path only when a gradient has to reach param_direction
            grad_light = torch.is_grad_enabled() and self.light.param_direction.requires_grad
            rays = self.gen_rays_at(data, prior, with_light=not grad_light)
        cos_anneal_ratio = min(1.0, self._it_host / self.anneal_end)
        # "bg_color": optional (bs, 3) device tensor -- an extension used by the HIP-graph wrapper (oi_amd.graphed), whose
        # inputs must live at fixed device addresses; the reference always draws it from numpy (generator.py:161)
        if "bg_color" in data:
            bg = data["bg_color"]
        else:
            bg = getattr(self, "_bg_dev", None)
            if bg is None:
                bg = self._h2d(self.bg_color(bs))
        if grad_light and prior["w2b"].is_cuda:
            # unit direction per box frame with its own backward: one launch each way (tensor ops: 8 + ~18 launches)
            ldir = self.light.batch_direction_unit(prior["w2b"])
            ldir._oi_unit = True   # (CompositeFunction.run: already normalised, Jacobian owned by the producer)
        else:
            ldir = prior["light"].direction() if grad_light else rays["light_dir"]
        glue5, lpk = self._glue()
        if torch.is_grad_enabled() and any(p.requires_grad for p in (self.light.param_ambient, self.light.param_specular,
                                                                     self.light.param_shininess)):
            from .lighting import PackLight
            lpk = PackLight.apply(self.light.param_ambient, self.light.param_specular, self.light.param_shininess, lpk)

        ro_all, rd_all = rays["rays_o"].view(bs, h * w, 3), rays["rays_d"].view(bs, h * w, 3)
        near_all, far_all = rays["near"].view(bs, h * w, 1), rays["far"].view(bs, h * w, 1)
        chunk = int(MAX_RAY_BATCH_SIZE / bs)
        n_chunks = math.ceil(h * w / chunk)
        if n_chunks > 1:
            assert not self.training, (n_rays, chunk)
        # style MLP (generator.py:235-238) + FiLM parameters of all 9 layers: ONE launch, shared by every chunk
        if film is None:
            film = self.renderer.pack.film(z=None if "w" in latent else latent["z"], w=latent.get("w"))
        latent["w"] = film[0]
        # without return_raw the per-sample compositing outputs (weights, cdf, alpha, inside_sphere, pts_norm: 2 MB each at C2)
        # and the extra maps reach nobody: the launch is not asked for them
        want = None if (return_raw or n_chunks > 1) else ("weight_sum", "color_fine", "image_no_bg", "image", "shading", "mask", "reduce4")
        outs = []
        for ci in range(n_chunks):
            sl = slice(ci * chunk, (ci + 1) * chunk)
            flat = lambda t: t[:, sl].reshape(-1, t.shape[-1])
            s, c = self.renderer.render_full(flat(ro_all), flat(rd_all), flat(near_all), flat(far_all),
                                             perturb_overwrite=-1 if self.training else 0,
                                             cos_anneal_ratio=cos_anneal_ratio, z=latent["z"], w=latent["w"],
                                             light=lpk, light_dir=ldir, bg=bg, film=film, coarse=coarse,
                                             image_planar=(n_chunks == 1), outputs=want,
                                             f3_scratch=f3_scratch if n_chunks == 1 else None)
            outs.append((s, c))
        if n_chunks == 1:
            s, c = outs[0]
        else:  # eval only: (bs, chunk, ...) pieces back to (bs*h*w, ...) rows (generator.py:298-305)
            cat = lambda k, d: torch.cat([o[d][k].unflatten(0, (bs, -1)) for o in outs], 1).flatten(0, 1)
            s = {k: cat(k, 0) for k in outs[0][0]}
            c = {k: cat(k, 1) for k in outs[0][1] if k not in ("reduce4", "ray_sums", "finals")}
            c["reduce4"] = sum(o[1]["reduce4"] for o in outs)
        # no gradient recorded: gradient_error, surface_loss and the three per-ray logging means come out of the
        # compositing reduction itself (oi_render_stats); with autograd they are tensor expressions of reduce4
        finals = c.get("finals") if n_chunks == 1 else None
        # (with a gradient recorded gradient_error / surface_loss are autograd expressions of reduce4; the logging means still
        # come from the launch's own reduction)
        render_out = assemble_render_dict(s, c, self.deviation_network.variance,
                                          finals=None if torch.is_grad_enabled() else finals, s_val=glue5[1])
        if n_chunks > 1:
            render_out["gradient_error"] = None
            render_out["surface_loss"] = None

        def to_map(x):
            if x.dim() == 3:  # the compositing kernel wrote the (bs, 3, h * w) map itself (`image`, unchunked)
                return x.view(bs, -1, h, w)
            return x.reshape(bs, h, w, -1).permute(0, 3, 1, 2)

        new = {
            "weight_sum_map": to_map(c["weight_sum"]),
            "color_map": to_map(c["color_fine"]),
            "shading_map": to_map(c["shading"]).expand(bs, 3, h, w),
            "image_no_bg": to_map(c["image_no_bg"]),
            "image": to_map(c["image"]),
            "mask": to_map(c["mask"]),
        }
        if return_raw:
            amb = glue5[2]
            if torch.is_grad_enabled() and self.light.param_ambient.requires_grad:
                amb = torch.sigmoid(self.light.param_ambient)   # (keeps its graph, as in the reference; nothing trains on it)
            new.update({
                "amb_shading_map": (amb * to_map(c["weight_sum"])).expand(bs, 3, h, w),
                "diff_shading_map": to_map(c["diffuse_map"]).expand(bs, 3, h, w),
                "normal_map": to_map(c["normal"]),
                "no_specular_map": to_map(c["image_no_bg"]) - to_map(c["specular_map"]),
                "specular_map": to_map(c["specular_map"]).expand(bs, 3, h, w),
                "z_map": to_map(c["z_map"]),
                "z_min": s["mid_z_vals"].min(-1).values.reshape(bs, -1).min(-1).values,
            })
        # logging scalars (generator.py:208-223).  Four per-ray means in two launches instead of four; the light colours
        # are `expand(3)` of one scalar in the reference, so their means are that scalar: three launches instead of
        # nine.  All stay device tensors (the reference calls .item() on the light terms: four host syncs per forward).
        if finals is not None:
            ray_stats = finals[2:5]
        else:
            ray_stats = torch.cat([render_out["cdf_fine"][:, :1], render_out["weight_max"], render_out["weight_sum"]], 1).mean(0)
        amb, diff, spec = glue5[2], glue5[3], glue5[4]
        blob = {
            "loss": {"eikonal": render_out["gradient_error"]},
            "stats": {
                "surface": render_out["surface_loss"],
                "s_val": render_out["s_val"].detach()[0, 0],  # the mean of N copies of one scalar
                "cdf": ray_stats[0],
                "weight_max": ray_stats[1],
                "weight_sum": ray_stats[2],
                "light/ambient": amb,
                "light/diffuse": diff,
                "light/specular": spec,
                "material/shininess": self.light.shininess.detach(),
            },
            "render_out": new,
            "prior_info": prior,
        }
        if return_raw:
            blob.update({"latent_info": latent, "rays_info": rays, "raw_render_out": render_out})
        return {"box": blob}
