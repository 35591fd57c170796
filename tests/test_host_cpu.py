"""CPU tests of host-side pieces around the path: checkpoint format round trip (incl. DDP-prefix adaptation),
EMA update rule, pose prior / augmentation parameter distributions."""
import os

import numpy as np
import torch
import torch.nn as nn


def test_checkpoint_roundtrip_and_prefix_adaptation(tmp_path):
    from oi_amd.checkpoint import CheckpointIO
    from oi_amd.fields import ColorNetwork
    kw = dict(D=8, W=128, input_ch=3, input_ch_views=3, style_dim=64)
    a, b = ColorNetwork(**kw), ColorNetwork(**kw)
    opt = torch.optim.Adam(a.parameters(), lr=1e-3)
    io = CheckpointIO(str(tmp_path), color=a, opt=opt)
    path = io.save("model.pt", it=7, epoch=1)
    blob = torch.load(path, weights_only=False)
    assert set(blob) == {"color", "opt", "it", "epoch"} and "views_linears.weight" in blob["color"]
    scalars = CheckpointIO(str(tmp_path), color=b).load("model.pt")
    assert scalars["it"] == 7 and all(torch.equal(p, q) for p, q in zip(a.parameters(), b.parameters()))
    # a checkpoint written from a DDP-wrapped module ('module.' prefix) loads into a bare module
    blob["color"] = {"module." + k: v for k, v in blob["color"].items()}
    c = ColorNetwork(**kw)
    CheckpointIO(str(tmp_path), color=c).load(blob, strict=False)
    assert all(torch.equal(p, q) for p, q in zip(a.parameters(), c.parameters()))


def test_ema_matches_reference_rule():
    from oi_amd.ema import EMA
    torch.manual_seed(0)
    m = nn.Linear(4, 3)
    ema = EMA(m, 0.9)
    before = [p.clone() for p in ema.module.parameters()]
    with torch.no_grad():
        for p in m.parameters():
            p.add_(1.0)
    ema.update(0)
    for pe, b, p in zip(ema.module.parameters(), before, m.parameters()):
        assert torch.allclose(pe, p.lerp(b, 0.9), atol=1e-6)  # src/utils/ema.py:29-30


def test_plane_pose_prior_is_rigid_and_in_range():
    from oi_amd.pose import Plane
    np.random.seed(0)
    prior = Plane(cam_loc=[0, -1, 0], rot_degree_range_scale=360, rot_roll_degree_range_scale=20, xy_range_scale=[6, 3.5])
    m = prior(64)
    assert m.shape == (64, 4, 4)
    R = m[:, :3, :3]
    assert np.allclose(R @ R.transpose(0, 2, 1), np.eye(3)[None], atol=1e-6) and np.allclose(np.linalg.det(R), 1, atol=1e-6)
    assert np.abs(m[:, 0, 3]).max() <= 6 and np.abs(m[:, 1, 3]).max() <= 3.5
    assert prior.pose_to_vec_repr(torch.tensor(m)).shape == (64, 6)


def test_augment_parameter_distribution():
    from oi_amd.augment import AugmentPipe
    np.random.seed(0)
    aug = AugmentPipe(xint=1, scale=1)
    x = torch.zeros(4096, 1, 8, 8)
    G = aug.sample_G_inv(x)
    s = 1.0 / G[:, 0, 0]
    assert abs(np.log2(s).std() - 0.2) < 0.02 and np.allclose(G[:, 0, 0], G[:, 1, 1])  # isotropic scale, std 0.2 in log2
    t = -G[:, 0, 2]  # G = T(-t) S(1/s): the translation column is the integer pixel shift
    assert np.abs(t).max() <= 1.0 + 1e-5 and np.allclose(t, np.round(t), atol=1e-4)  # round(U(-.125,.125)*8) in {-1,0,1}
