"""F13: oi_amd.trainer.Trainer against TWO iterations of the reference's own `Trainer.train_step`
(src/trainers/gan_pose_trainer.py:77-202, recorded by oracle/gen_golden_r3.py from the imported class): ADA on with a
pinned `debug_percentile`, the three optimisers of configs/train.yaml:133-147.

What makes the two runs comparable:
  * initial weights, the real batch and the iteration counter come from the fixture;
  * poses and background colours are numpy draws -- both implementations draw them in the same order from the same seed
    (the reference's AugmentPipe draws from torch, ours from numpy: the test keeps our augmentation draws out of the
    numpy stream);
  * the generator's torch draws (latent, per-ray jitter) are replayed from the fixture.
"""
import functools
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden, sub_sd, F13_VARIANCE_GRAD_SENSITIVITY

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _build(g, graph_d_steps):
    import bench
    from oi_amd.config import build_from_config
    from oi_amd.optim import FusedAdam, FusedRMSprop
    from oi_amd.trainer import Trainer
    R, S, I = int(g["resolution"]), int(g["n_samples"]), int(g["n_importance"])
    dev = torch.device("cuda", 0)
    gen, _ = bench.build_models(R, S, I, 1, "f16x3", dev)
    gen.load_state_dict(sub_sd(g, "g0."))
    net = lambda t, **kw: {"__target__": t, "kwargs": kw}
    aug = net("src.third_party.ada.augment.AugmentPipe", scale=1, xint=1)
    D = build_from_config(net("src.models.discriminator.ADADiscriminatorView", out_dim_position=6, out_dim_latent=0, aug=aug,
                              aug_p=1, in_dim=3, out_dim=7, n_feat=32, img_size=R, last_bias=False))
    M = build_from_config(net("src.models.discriminator.ADADiscriminator", aug=aug, aug_p=1, in_dim=1, out_dim=1, n_feat=32,
                              img_size=R, last_bias=False))
    D.load_state_dict(sub_sd(g, "d0."))
    M.load_state_dict(sub_sd(g, "m0."))
    D, M = D.to(dev), M.to(dev)
    pct = float(g["debug_percentile"])
    for d in (D, M):
        # pinned percentile, and the augmentation's own (numpy) draws kept out of the stream the generator draws poses from
        inner = d.aug.sample_G_inv

        def sample(images, debug_percentile=None, _inner=inner):
            state = np.random.get_state()
            try:
                return _inner(images, pct)
            finally:
                np.random.set_state(state)

        d.aug.sample_G_inv = sample
    mods = {"generator": gen, "discriminator": D, "mask_discriminator": M,
            "opt_generator": FusedAdam(gen.parameters(), lr=2e-5, betas=(0.0, 0.9)),
            "opt_discriminator": FusedRMSprop(D.parameters(), lr=1e-4),
            "opt_mask_discriminator": FusedRMSprop(M.parameters(), lr=1e-4)}
    tr = Trainer(mods, it=int(g["it0"]) - 1, graph_d_steps=graph_d_steps)
    return tr, gen, D, M


@pytest.mark.parametrize("graph_d_steps", [False, True])
def test_trainer_two_iterations_match_the_reference_trainer_class_f13(graph_d_steps, monkeypatch):
    g = load_golden("f13_trainer")
    tr, gen, D, M = _build(g, graph_d_steps)
    zs = iter(g["z_draws"].cuda())
    jits = iter(g["jitter_draws"].cuda())
    gen.sample_latent = lambda bs, data: {"z": next(zs)}
    real_rand = torch.rand
    N = int(g["resolution"]) ** 2

    def rand(*a, **k):
        shape = tuple(a[0]) if len(a) == 1 and isinstance(a[0], (list, tuple, torch.Size)) else tuple(a)
        if shape == (N, 1):
            return next(jits)            # the reference's jitter draw of this render (renderer.py:372)
        return real_rand(*a, **k)

    monkeypatch.setattr(torch, "rand", rand)
    # every one of the six renders against what the reference's generator produced in the same call
    calls = []
    inner_forward = gen.forward

    def forward(*a, **k):
        r = inner_forward(*a, **k)
        j, blob = len(calls), r["box"]
        calls.append(j)
        for name, got in (("b2w", blob["prior_info"]["b2w"]), ("c2b", blob["prior_info"]["c2b"]),
                          ("image", blob["render_out"]["image"]), ("mask", blob["render_out"]["mask"]),
                          ("eikonal", blob["loss"]["eikonal"])):
            ref = g[f"call{j}.{name}"]
            err = float((got.detach().cpu().reshape(ref.shape) - ref).abs().max())
            assert err < 1e-4 * max(1.0, float(ref.abs().max())), (f"render {j}", name, err)
        return r

    gen.forward = forward
    data = {"image": g["data_image"].cuda(), "mask": g["data_mask"].cuda()}
    np.random.seed(int(g["np_seed"]))
    worst = {}
    for i in range(2):
        out = tr.train_step(data)
        torch.cuda.synchronize()
        for k, v in out.items():
            key = f"it{i}.{k}"
            if key not in g:
                continue
            ref = float(g[key])
            err = abs(float(v) - ref) / max(1.0, abs(ref))
            worst[key] = err
            assert err < 2e-4, (key, float(v), ref)
        # grad_stats/*: mean gradient norm per child module of the generator after the G step (tu/utils/training.py:24-41;
        # the generator's gradients survive the two discriminator steps)
        for name, child in gen.named_children():
            key = f"it{i}.grad_stats/{name}"
            norms = [torch.linalg.norm(p.grad) for p in child.parameters() if p.grad is not None]
            if key in g and float(g[key]) >= 0 and norms:
                got, ref = float(torch.stack(norms).mean()), float(g[key])
                # deviation_network = ONE scalar (d loss / d variance).  In iteration 1 it is BIMODAL under last-bit changes of
                # near / far: the oracle, which reproduces the reference's 0.0858335 to 3e-7, reports 0.08627 (+0.51 %) for 2 of 12
                # seeded +-1 ulp perturbations (one importance sample of one render switches bins; measured by
                # tests/test_oracle_golden.py::test_f13_two_iterations_on_the_oracle_and_last_bit_sensitivity) -- the HIP path sits
                # on that second mode (0.08628).  Not a wider bar: the value must sit on ONE of the two measured modes (the
                # reference's, or the reference's x (1 + the measured jump)) at the 2e-3 of every other gradient norm.
                modes = (ref, ref * (1 + F13_VARIANCE_GRAD_SENSITIVITY)) if name == "deviation_network" else (ref,)
                err = min(abs(got - m) for m in modes)
                assert err < 2e-3 * max(ref, 1e-3), (key, got, ref, modes)
                worst[key] = err / max(ref, 1e-3)
        for tag, net in (("g", gen), ("d", D), ("m", M)):
            flat = torch.cat([p.detach().double().reshape(-1) for p in net.parameters()]).cpu()
            for nm, val in (("sum", flat.sum()), ("abs", flat.abs().sum())):
                ref = float(g[f"it{i}.{tag}_{nm}"])
                assert abs(float(val) - ref) < 2e-6 * float(g[f"it{i}.{tag}_abs"]), (i, tag, nm, float(val), ref)
    with pytest.raises(StopIteration):
        next(zs)                       # all six latent draws consumed: three renders per iteration, as in the reference
    # the generator's weights after both iterations, tensor by tensor.  Adam with beta1 = 0 moves every weight by about
    # lr * sign(gradient) per step (2e-5): an entry whose gradient is at rounding level may take the other sign, so two
    # steps can differ by 4 lr = 8e-5 there while the check sums above (2e-6 of sum |w|) hold the bulk
    ref_sd = sub_sd(g, "g2.")
    for k, v in gen.state_dict().items():
        if k in ref_sd and v.dtype.is_floating_point and v.numel() > 1 and "camera" not in k:
            r = ref_sd[k]
            d = (v.cpu() - r).abs()
            assert float(d.max()) < 1e-4 and float(d.mean()) < 2e-6, (k, float(d.max()), float(d.mean()))
    assert len(worst) >= 20, sorted(worst)
