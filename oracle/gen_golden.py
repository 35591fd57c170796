"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by running the *reference* itself
(imported read-only from /root/reference as CPU PyTorch, see oracle/ref_import.py).

Run in the build container only:   python oracle/gen_golden.py
The output files are data (inputs, weights, expected outputs); no reference source travels.
Fixture ids follow SURVEY.md section 8(c).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_import import reference_on_cpu  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
NET_KW = dict(D=8, W=128, input_ch=3, input_ch_views=3, style_dim=64)


def npd(d):
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
            for k, v in d.items()}


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **npd(arrs))
    print(f"wrote {path}  {os.path.getsize(path) / 1024:.1f} KiB")


def sd_arrays(prefix, module):
    return {prefix + k: v for k, v in module.state_dict().items()}


def synth_rays(N, seed):
    g = torch.Generator().manual_seed(seed)
    o = torch.tensor([0.0, 0.0, -3.0]).expand(N, 3).contiguous()
    o = o + 0.05 * torch.randn(N, 3, generator=g)
    d = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, 1.0]) + 0.15 * torch.randn(N, 3, generator=g), dim=-1)
    return o, d


def example_cfg(resolution):
    """Config values scripts/train.py derives for data/example (train.py:25-47, 88-115)."""
    fov, img_size, img_size_scene = 10.0, 256, 1588
    cam_dist = float(1 / np.tan(0.5 * fov * np.pi / 180))
    scene_fov = float(2 * np.arctan(img_size_scene / img_size * np.tan(0.5 * fov * np.pi / 180)) * 180 / np.pi)
    scene_res = int(resolution * img_size_scene / img_size)
    pose_prior = {"__target__": "src.utils.pose_sampler.Plane",
                  "kwargs": dict(cam_loc=[0, -1, 0], rot_degree_range_scale=360,
                                 rot_roll_degree_range_scale=20, xy_range_scale=[6, 3.5])}
    return dict(cam_dist=cam_dist, scene_fov=scene_fov, scene_resolution=scene_res, pose_prior=pose_prior)


def build_generator(resolution, S, I, K, specular=0.0):
    from src.models.generator import Generator
    c = example_cfg(resolution)
    net = lambda t, **kw: {"__target__": t, "kwargs": kw}
    g = Generator(
        color_network=net("src.models.fields.ColorNetwork", **NET_KW),
        sdf_network=net("src.models.fields.ShapeNetwork", checkpoint_path="./checkpoints/sphere_init.pt", **NET_KW),
        deviation_network=net("src.third_party.neus.models.fields.SingleVarianceNetwork", init_val=0.3),
        light_network=net("src.utils.prior.build_directional_light_optimizable", cam_loc=None, light_loc=None,
                          ambient_color=0.33, diffuse_color=0.66, specular_color=specular, shininess=10),
        camera=net("src.models.camera_network.Camera", cam_dist=c["cam_dist"], resolution=c["scene_resolution"],
                   fov=c["scene_fov"]),
        z_dim=64, resolution=resolution, scene_resolution=c["scene_resolution"],
        renderer=net("src.third_party.neus.models.renderer.NeuSRenderer", n_importance=I, n_outside=0,
                     n_samples=S, perturb=1, up_sample_steps=K),
        anneal_end=50000, pose_prior=c["pose_prior"])
    return g, c


def main():
    with reference_on_cpu():
        from src.models.fields import ShapeNetwork, ColorNetwork
        from src.third_party.neus.models.fields import SingleVarianceNetwork
        from src.third_party.neus.models.renderer import NeuSRenderer
        from src.models.generator import near_far_from_sphere
        from src.models.discriminator import DCDiscriminator
        from src.third_party.ada.augment import AugmentPipe
        from src.third_party.ada.torch_utils.ops import upfirdn2d as ref_upfirdn2d
        from src.loss.gan import compute_grad2, GANLoss

        torch.manual_seed(0)
        np.random.seed(0)
        sdf = ShapeNetwork(checkpoint_path="./checkpoints/sphere_init.pt", **NET_KW)
        col = ColorNetwork(**NET_KW)
        dev = SingleVarianceNetwork(0.3)

        # weights (data): the sphere-initialised SDF net, and the seeded colour head
        save("weights_sdf", **sd_arrays("", sdf))
        save("weights_color", **sd_arrays("", col))

        # ---------------- F1: FiLM-SIREN sdf / feature / gradient ----------------
        g = torch.Generator().manual_seed(1)
        z = torch.randn(2, 64, generator=g)
        pts = (torch.rand(1024, 3, generator=g) * 2.4 - 1.2)
        w = sdf.style(z)
        out = sdf(pts, z=z, w=w)
        grad = sdf.gradient(pts.clone(), z=z, w=w)
        save("f1_film_siren", z=z, w=w, pts=pts, sdf=out[:, :1], feat=out[:, 1:], grad=grad)

        # ---------------- F2: colour head ----------------
        rgb = col(pts, grad.detach(), None, out[:, 1:].detach(), z=z, w=w)
        save("f2_color", w=w, feat=out[:, 1:], grad=grad, rgb=rgb)

        # ---------------- F3: up_sample + sample_pdf + cat_z_vals ----------------
        N = 64
        ro, rd = synth_rays(N, 3)
        near, far = near_far_from_sphere(ro, rd)
        z1 = torch.randn(1, 64, generator=g)
        w1 = sdf.style(z1)
        f3 = dict(rays_o=ro, rays_d=rd, near=near, far=far, w=w1)
        for K in (1, 4):
            r = NeuSRenderer(None, sdf, dev, col, n_samples=16, n_importance=16, n_outside=0,
                             up_sample_steps=K, perturb=0)
            with torch.no_grad():
                zv = near + (far - near) * torch.linspace(0, 1, 16)[None, :]
                p = ro[:, None, :] + rd[:, None, :] * zv[..., None]
                s = sdf.sdf(p.reshape(-1, 3), z=z1, w=w1).reshape(N, 16)
                if K == 1:
                    f3["z_coarse"], f3["sdf_coarse"] = zv, s
                    f3["z_new_k1"] = r.up_sample(ro, rd, zv, s, 16, 64)
                for i in range(K):
                    zn = r.up_sample(ro, rd, zv, s, 16 // K, 64 * 2 ** i)
                    zv, s = r.cat_z_vals(ro, rd, zv, zn, s, last=(i + 1 == K), z=z1, w=w1)
            f3[f"z_merged_k{K}"] = zv
        save("f3_upsample", **f3)

        # ---------------- F4: full render dict, 64 rays x (16+16) ----------------
        r = NeuSRenderer(None, sdf, dev, col, n_samples=16, n_importance=16, n_outside=0,
                         up_sample_steps=1, perturb=0)
        f4 = dict(rays_o=ro, rays_d=rd, near=near, far=far, w=w1, variance=dev.variance)
        for car in (0.0, 0.5, 1.0):
            o_ = r.render(ro, rd, near, far, perturb_overwrite=0, cos_anneal_ratio=car, z=z1, w=w1)
            tag = str(car).replace(".", "p")
            for k, v in o_.items():
                f4[f"c{tag}_{k}"] = v
        # two batch elements + training jitter (perturb): the jitter tensor is the first RNG draw
        z2 = torch.randn(2, 64, generator=g)
        w2 = sdf.style(z2)
        torch.manual_seed(77)
        jit = torch.rand([N, 1])
        torch.manual_seed(77)
        o_ = r.render(ro, rd, near, far, perturb_overwrite=1, cos_anneal_ratio=0.3, z=z2, w=w2)
        f4["b2_w"], f4["b2_jitter"] = w2, jit
        for k, v in o_.items():
            f4[f"b2_{k}"] = v
        save("f4_render", **f4)

        # ---------------- F5: Generator.forward (rays + render_maps), specular on ----------------
        torch.manual_seed(5)
        gen, cfg = build_generator(16, 16, 16, 1, specular=0.0)
        gen.eval()
        with torch.no_grad():
            gen.light.param_specular.fill_(0.35)
            gen.light.param_shininess.fill_(6.0)
            gen.light.param_direction.copy_(torch.tensor([0.3, -0.5, -0.8]))
        np.random.seed(11)
        b2w = torch.tensor(gen.pose_prior(2), dtype=torch.float32)
        zg = torch.randn(2, 64)
        np.random.seed(12)
        bg = torch.tensor(np.random.uniform(low=0, high=1, size=(2, 3)), dtype=torch.float32)
        np.random.seed(12)
        gen.it.fill_(20000)
        blob = gen(bs=2, it=None, data={"z": zg, "b2w": b2w}, return_raw=True)["box"]
        f5 = dict(b2w=b2w, z=zg, bg=bg, it=20000, anneal_end=50000, resolution=16,
                  scene_resolution=cfg["scene_resolution"], cam_dist=cfg["cam_dist"], scene_fov=cfg["scene_fov"],
                  intrinsics_inv=gen.camera.intrinsics_inv, c2w=gen.camera.c2w, w2c=gen.camera.w2c,
                  rays_o=blob["rays_info"]["rays_o"], rays_d=blob["rays_info"]["rays_d"],
                  c2b=blob["prior_info"]["c2b"], w=blob["latent_info"]["w"],
                  eikonal=blob["loss"]["eikonal"])
        f5.update(sd_arrays("color.", gen.color_network))
        f5.update(sd_arrays("light.", gen.light))
        for k, v in blob["render_out"].items():
            f5["map_" + k] = v
        for k in ("weights", "mid_z_vals", "weight_sum", "color_fine", "sdf"):
            f5["raw_" + k] = blob["raw_render_out"][k]
        for k, v in blob["stats"].items():
            f5["stat_" + k.replace("/", "_")] = torch.as_tensor(v)
        save("f5_generator", **f5)

        # ---------------- F6: first/second-order gradients of a scalar loss ----------------
        torch.manual_seed(6)
        gen, cfg = build_generator(8, 8, 8, 1, specular=0.0)
        gen.train()
        with torch.no_grad():
            gen.light.param_specular.fill_(0.2)
        rays = gen.gen_rays_at({}, gen.sample_prior(1, {"b2w": b2w[:1]}) if False else
                               {"b2w": b2w[:1], "c2b": torch.einsum("bij,jk->bik",
                                                                    torch.linalg.inv(b2w[:1]), gen.camera.c2w)})
        # (gen_rays_at only needs b2w and c2b)
        ro6 = rays["rays_o"].flatten(0, 2)
        rd6 = rays["rays_d"].flatten(0, 2)
        n6, f6_ = near_far_from_sphere(ro6, rd6)
        z6 = torch.randn(1, 64)
        w6 = gen.sdf_network.style(z6)
        torch.manual_seed(66)
        jit6 = torch.rand([ro6.shape[0], 1])
        torch.manual_seed(66)
        ro_ = gen.renderer.render(ro6, rd6, n6, f6_, perturb_overwrite=1, cos_anneal_ratio=0.4, z=z6, w=w6)
        from src.utils.pose import invert_rot_t
        prior = {"light": gen.light.batch_transform(w2b=invert_rot_t(b2w[:1]))}
        np.random.seed(13)
        bg6 = torch.tensor(np.random.uniform(low=0, high=1, size=(1, 3)), dtype=torch.float32)
        np.random.seed(13)
        eik = ro_["gradient_error"]
        maps = gen.render_maps(bs=1, render_out=ro_, rays_info=rays, prior_info=prior, return_raw=False)
        loss = maps["image"].sum() + 10.0 * eik + maps["shading_map"].sum() + 0.5 * maps["mask"].sum()
        params = dict(gen.named_parameters())
        grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True, retain_graph=True)  # F9 reuses the graph
        f6 = dict(b2w=b2w[:1], z=z6, bg=bg6, jitter=jit6, rays_o=ro6, rays_d=rd6, loss=loss, eikonal=eik,
                  image=maps["image"], shading_map=maps["shading_map"], mask=maps["mask"], cos_anneal_ratio=0.4)
        f6.update(sd_arrays("p.", gen))
        for (k, _), gr in zip(params.items(), grads):
            if gr is not None:
                f6["g." + k] = gr
        save("f6_grads", **f6)

        # ---------------- F9: one scripted training iteration assembled from the reference's pieces ----------------
        # (the Trainer class itself cannot be imported here: torchvision / omegaconf / tensorboard are absent).  Same
        # call pattern and loss composition as gan_pose_trainer.py:103-145 (G step) and :154-200 (D steps) with the
        # weights of configs/train.yaml:120-129; inputs are the F6 ones, augmentation probability 0 (identity).
        from src.models.discriminator import ADADiscriminatorView, ADADiscriminator
        from src.loss.position import PositionLoss, linear_increase
        torch.manual_seed(9)
        aug_cfg = {"__target__": "src.third_party.ada.augment.AugmentPipe", "kwargs": {"scale": 1, "xint": 1}}
        D9 = ADADiscriminatorView(out_dim_position=6, out_dim_latent=0, aug=aug_cfg, aug_p=0.0, in_dim=3, out_dim=7,
                                  n_feat=32, img_size=8, last_bias=False)
        M9 = ADADiscriminator(aug=aug_cfg, aug_p=0.0, in_dim=1, out_dim=1, n_feat=32, img_size=8, last_bias=False)
        gan9, pos9 = GANLoss("bce"), PositionLoss("mse")
        it9 = 500
        # G step
        ld = gan9(D9(maps["image"], it=it9)[:, :1], 1)
        lm = gan9(M9(maps["mask"], it=it9), 1)
        lg = ld * 1.0 + lm * 0.1 + 10.0 * eik
        gg = torch.autograd.grad(lg, list(params.values()), allow_unused=True, retain_graph=True)
        f9 = dict(it=it9, g_loss_disc=ld, g_loss_mask=lm, g_loss=lg)
        for (k, _), gr in zip(params.items(), gg):
            if gr is not None:
                f9["gg." + k] = gr
        # D step (real + R1 + fake + auxiliary pose regression) and mask-D step
        x_real = torch.rand(1, 3, 8, 8)
        m_real = (torch.rand(1, 1, 8, 8) > 0.5).float()
        c2b9 = torch.einsum("bij,jk->bik", torch.linalg.inv(b2w[:1]), gen.camera.c2w)
        f9.update(in_x_real=x_real, in_m_real=m_real, c2b=c2b9)
        f9.update(image=maps["image"], mask=maps["mask"], eikonal=eik)
        for tag, net, xr, xf, aux in (("d", D9, x_real, maps["image"], True), ("m", M9, m_real, maps["mask"], False)):
            xr = xr.clone().requires_grad_()
            d_real = net(xr, it=it9)[:, :1]
            l_real = gan9(d_real, 1)
            l_reg = compute_grad2(d_real, xr)
            xf = xf.detach().clone().requires_grad_()
            d_fake = net(xf, it=it9)
            l_aux = torch.zeros(())
            if aux:
                d_fake, d_aux = torch.split(d_fake, (1, 6), dim=1)
                l_aux = pos9(d_aux, c2b9[..., :2, :3].flatten(-2, -1))
            l_fake = gan9(d_fake, 0)
            loss9 = l_real + l_fake + l_reg * 10.0 + l_aux * linear_increase(1000, 1)(it9)
            gw9 = torch.autograd.grad(loss9, list(net.parameters()))
            f9.update({f"{tag}_real": l_real, f"{tag}_fake": l_fake, f"{tag}_reg": l_reg, f"{tag}_aux": l_aux,
                       f"{tag}_loss": loss9})
            f9.update(sd_arrays(f"{tag}_w.", net))
            for (k, _), gr in zip(net.named_parameters(), gw9):
                f9[f"{tag}_g." + k] = gr
        save("f9_train_step", **f9)

        # ---------------- F7: DC discriminator fwd, input-grad, R1 weight grads ----------------
        f7 = {}
        for res, nf, cin, cout in ((16, 32, 3, 7), (64, 64, 3, 7), (64, 32, 1, 1)):
            torch.manual_seed(res + cin)
            D = DCDiscriminator(in_dim=cin, out_dim=cout, n_feat=nf, img_size=res)
            x = torch.rand(2, cin, res, res, requires_grad=True)
            d = D(x)
            d1 = d[:, :1]
            reg = compute_grad2(d1, x)
            loss = GANLoss("bce")(d1, 1) + 10.0 * reg
            gw = torch.autograd.grad(loss, list(D.parameters()), retain_graph=True)
            (gx,) = torch.autograd.grad(d1.sum(), x, retain_graph=True)
            t = f"r{res}c{cin}_"
            f7.update({t + "x": x, t + "d": d, t + "reg": reg, t + "loss": loss, t + "gx": gx})
            f7.update(sd_arrays(t + "w.", D))
            for (k, _), gr in zip(D.named_parameters(), gw):
                f7[t + "g." + k] = gr
        save("f7_discriminator", **f7)

        # ---------------- F8: AugmentPipe(xint, scale) + upfirdn2d ----------------
        torch.manual_seed(8)
        aug = AugmentPipe(xint=1, scale=1)
        f8 = dict(Hz_geom=aug.Hz_geom)
        x32 = torch.rand(2, 3, 32, 32)
        x64 = torch.rand(2, 1, 64, 64)
        f8["x32"], f8["x64"] = x32, x64
        for pct in (0.1, 0.5, 0.9):
            tag = str(pct).replace(".", "p")
            f8[f"y32_{tag}"] = aug(x32, debug_percentile=pct)
            f8[f"y64_{tag}"] = aug(x64, debug_percentile=pct)
        xa = torch.rand(2, 3, 21, 17, requires_grad=True)
        up = ref_upfirdn2d.upsample2d(xa, aug.Hz_geom, up=2)
        dn = ref_upfirdn2d.downsample2d(up, aug.Hz_geom, down=2, padding=-2, flip_filter=True)
        (gxa,) = torch.autograd.grad((dn * dn).sum(), xa)
        f2d = torch.rand(3, 5)
        gen_ = ref_upfirdn2d.upfirdn2d(xa, f2d, up=[2, 1], down=[1, 3], padding=[1, 2, 0, 3], flip_filter=False, gain=1.7)
        f8.update(ufd_x=xa, ufd_up=up, ufd_down=dn, ufd_gx=gxa, ufd_f2d=f2d, ufd_general=gen_)
        save("f8_augment", **f8)


if __name__ == "__main__":
    main()
