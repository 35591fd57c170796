"""TEST INFRASTRUCTURE: two iterations of the reference's trainer (gan_pose_trainer.py:77-202: G step, D step, mask-D step, the
optimisers of configs/train.yaml:133-147) restated on the ORACLE (oracle/oi_oracle.py, CPU) for the F13 fixture.  Poses and
background colours come from the product's host-side sampler (numpy only; checked bit for bit against the fixture's b2w / c2b by
the caller), everything on the data path is oracle arithmetic + torch.optim.

`perturb_seed`: near / far of every render move by -1 / 0 / +1 ulp per ray (seeded) -- the size of difference any other
evaluation order of generator.py:336-342 produces.  tests/test_oracle_golden.py measures what that does to the reported
gradient norms; tests/test_gpu_trainer_f13.py takes its bar for `grad_stats/deviation_network` from that measurement."""
import numpy as np
import torch

import oi_oracle as O
from conftest import sub_sd


def run(g, perturb_seed=None, dt=torch.float32, iters=2, check_calls=False):
    import bench
    R, S, I = int(g["resolution"]), int(g["n_samples"]), int(g["n_importance"])
    cam_dist, scene_fov, scene_res = bench.example_cfg(R)
    gen, _ = bench.build_models(R, S, I, 1, "f16x3", torch.device("cpu"))   # host-side pose / background sampler only
    gen.load_state_dict(sub_sd(g, "g0."))
    c = lambda t: t.to(dt).clone()
    G0 = sub_sd(g, "g0.")
    leaf = lambda d: {k: c(v).requires_grad_(True) for k, v in d.items()}
    sd = leaf({k[12:]: v for k, v in G0.items() if k.startswith("sdf_network.")})
    csd = leaf({k[14:]: v for k, v in G0.items() if k.startswith("color_network.")})
    lsd = leaf({k[6:]: v for k, v in G0.items() if k.startswith("light.")})
    var = c(G0["deviation_network.variance"]).requires_grad_(True)
    dsd = leaf({k: v for k, v in sub_sd(g, "d0.").items() if "aug." not in k})
    msd = leaf({k: v for k, v in sub_sd(g, "m0.").items() if "aug." not in k})
    gparams = list(sd.values()) + list(csd.values()) + [var] + list(lsd.values())
    opt_g = torch.optim.Adam(gparams, lr=2e-5, betas=(0.0, 0.9))
    opt_d = torch.optim.RMSprop(list(dsd.values()), lr=1e-4)
    opt_m = torch.optim.RMSprop(list(msd.values()), lr=1e-4)
    K, K_inv, c2w, w2c = O.camera_matrices(cam_dist, scene_fov, scene_res)
    pct = torch.tensor(float(g["debug_percentile"]))
    Gaug = O.ada_G_inv(1, R, R, ((pct*2-1)*0.125).expand(1,2), torch.exp2(torch.erfinv(pct*2-1)*0.2).expand(1)).to(dt)
    aug = lambda x: O.ada_geometric(x, Gaug)[0]
    pg = torch.Generator().manual_seed(perturb_seed) if perturb_seed is not None else None
    calls = [0]
    np.random.seed(int(g["np_seed"]))
    def render(it):
        j = calls[0]; calls[0] += 1
        b2w_h, w2b_h, c2b_h, xy_h, bg_h = gen._sample_prior_host(1, {})
        b2w = torch.from_numpy(b2w_h.reshape(1,4,4))
        ro, rd, c2b, w2b = O.gen_rays(b2w, K_inv, c2w, w2c, cam_dist, R, scene_res)
        ro, rd = ro.reshape(-1,3), rd.reshape(-1,3)
        near, far = O.near_far_from_sphere(ro, rd)
        if pg is not None:   # +-1 ulp on a third of the rays each way
            def bump(x):
                s = torch.randint(0, 3, x.shape, generator=pg) - 1
                return (x.view(torch.int32) + s.to(torch.int32) * torch.sign(x).to(torch.int32)).view(torch.float32)
            near, far = bump(near), bump(far)
        w = O.style_mlp(sd, g["z_draws"][j].to(dt))
        car = min(1.0, it / 50000.0)
        out = O.render(sd, csd, var, ro.to(dt), rd.to(dt), near.to(dt), far.to(dt), w, S, I, 1, car, jitter=g["jitter_draws"][j].to(dt))
        maps = O.render_maps(out, ro.to(dt), lsd, w2b.to(dt), torch.from_numpy(bg_h).to(dt), 1, R, R)
        if check_calls:   # every render against what the reference's generator produced in the same call
            assert float((c2b - g[f"call{j}.c2b"]).abs().max()) == 0.0, j
            assert float((maps["image"] - g[f"call{j}.image"]).abs().max()) < 5e-6, j
            assert float((maps["mask"] - g[f"call{j}.mask"]).abs().max()) < 5e-6, j
        return maps, out["gradient_error"], c2b
    real_img, real_mask = g["data_image"].to(dt), g["data_mask"].to(dt)
    stats = {}
    for i in range(iters):
        it = int(g["it0"]) + i
        # ---- G step
        opt_g.zero_grad(set_to_none=True)
        maps, eik, _ = render(it)
        ld = O.bce_logits_const(O.dc_discriminator(dsd, aug(maps["image"]))[:, :1], 1)
        lm = O.bce_logits_const(O.dc_discriminator(msd, aug(maps["mask"])), 1)
        loss = ld + 0.1 * lm + 10.0 * eik
        grads = torch.autograd.grad(loss, gparams, allow_unused=True)
        for p_, g_ in zip(gparams, grads): p_.grad = g_
        stats[f"it{i}.generator/loss"], stats[f"it{i}.generator/loss_mask"], stats[f"it{i}.generator/eikonal"] = float(ld), float(lm), float(eik)
        stats[f"it{i}.grad_stats/deviation_network"] = float(var.grad.abs())
        for name, d in (("sdf_network", sd), ("color_network", csd), ("light", lsd)):
            ns = [torch.linalg.norm(p_.grad) for p_ in d.values() if p_.grad is not None]
            stats[f"it{i}.grad_stats/{name}"] = float(torch.stack(ns).mean())
        opt_g.step()
        # ---- D step, mask-D step
        for key, net, opt, real, mk in (("discriminator", dsd, opt_d, real_img, "image"), ("mask_discriminator", msd, opt_m, real_mask, "mask")):
            with torch.no_grad():
                maps, _, c2b = render(it)
            opt.zero_grad(set_to_none=True)
            xr = real.clone().requires_grad_(True)
            d_real = O.dc_discriminator(net, aug(xr))[:, :1]
            l_real = O.bce_logits_const(d_real, 1)
            l_reg = O.r1_penalty(d_real, xr)
            d_fake = O.dc_discriminator(net, aug(maps[mk].detach()))
            l_aux = torch.zeros((), dtype=dt)
            if d_fake.shape[1] > 1:
                l_aux = torch.nn.functional.mse_loss(d_fake[:, 1:7], O.pose_to_vec(c2b.to(dt)))
                d_fake = d_fake[:, :1]
            l_fake = O.bce_logits_const(d_fake, 0)
            loss = l_real + l_fake + 10.0 * l_reg + min(it / 1000.0, 1.0) * l_aux
            for p_, g_ in zip(net.values(), torch.autograd.grad(loss, list(net.values()))): p_.grad = g_
            opt.step()
            stats[f"it{i}.{key}/loss"], stats[f"it{i}.{key}/reg"] = float(loss), float(l_reg)
            stats[f"it{i}.{key}/fake"], stats[f"it{i}.{key}/real"], stats[f"it{i}.{key}/aux_pose"] = float(l_fake), float(l_real), float(l_aux)
    return stats
