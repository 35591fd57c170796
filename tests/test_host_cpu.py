"""CPU tests of host-side pieces around the path: checkpoint format round trip (incl. DDP-prefix adaptation),
EMA update rule, pose prior / augmentation parameter distributions."""
import os

import pytest

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


def test_ema_has_no_host_tensor_path():
    """Like every op of the path the EMA update is a HIP launch (oi_multi_lerp): a host-resident module is rejected, not
    silently updated by tensor ops.  (The rule itself, src/utils/ema.py:29-30, is pinned on the GPU:
    tests/test_gpu_modules.py::test_ema_update_matches_reference_rule.)"""
    from oi_amd.ema import EMA
    torch.manual_seed(0)
    m = nn.Linear(4, 3)
    ema = EMA(m, 0.9)
    assert not any(p.requires_grad for p in ema.module.parameters()) and not ema.module.training
    with pytest.raises(ValueError, match="CUDA"):
        ema.update(0)


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


# ---------------------------------------------------------------- dataset loader (no cv2)
def _write_rgba(path, arr):
    from PIL import Image
    Image.fromarray(arr, "RGBA").save(path)


def test_dataset_loader_matches_reference_semantics(tmp_path):
    """eval_dataset.py:13-52 / preprocess.py:5-20: sorted *.png, uint8 bilinear resize, mask = alpha >= 128,
    item = rgb * mask + random bg * (1 - mask) with one np.random.uniform(size=(1,3)) draw per item."""
    from oi_amd.dataset import Dataset, read_rgba, resize_linear_u8
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, size=(48, 40, 4), dtype=np.uint8) for _ in range(3)]
    for i, a in enumerate(imgs):
        _write_rgba(str(tmp_path / f"{2 - i:02d}.png"), a)   # written out of order: the loader sorts by name
    (tmp_path / "ignored.jpg").write_bytes(b"x")
    ds = Dataset(16, str(tmp_path))
    assert len(ds) == 3 and [os.path.basename(p) for p in ds.data["path"]] == ["00.png", "01.png", "02.png"]
    assert ds.data["rgb"].shape == (3, 3, 16, 16) and ds.data["alpha"].shape == (3, 1, 16, 16)
    assert ds.data["rgb"].dtype == torch.float32 and set(ds.data["alpha"].unique().tolist()) <= {0.0, 1.0}
    # resize: within half a grey level of float64 bilinear at cv2's half-pixel-centre sample positions
    src = imgs[2].astype(np.float64)   # file 00.png
    ys = np.clip((np.arange(16) + 0.5) * 48 / 16 - 0.5, 0, 47); xs = np.clip((np.arange(16) + 0.5) * 40 / 16 - 0.5, 0, 39)
    y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int)
    y1 = np.minimum(y0 + 1, 47); x1 = np.minimum(x0 + 1, 39)
    fy = (ys - y0)[:, None, None]; fx = (xs - x0)[None, :, None]
    ref = (src[y0][:, x0] * (1 - fx) + src[y0][:, x1] * fx) * (1 - fy) + (src[y1][:, x0] * (1 - fx) + src[y1][:, x1] * fx) * fy
    got = resize_linear_u8(imgs[2], (16, 16)).astype(np.float64)
    assert np.abs(got - ref).max() <= 0.5 + 1e-3
    assert np.array_equal(ds.data["rgb"][0].permute(1, 2, 0).numpy() * 255.0, got[:, :, :3].astype(np.float32))
    assert np.array_equal(ds.data["alpha"][0, 0].numpy() > 0.5, got[:, :, 3] >= 128)
    # identity size returns the pixels untouched; non-RGBA input is rejected like the reference's assert
    assert np.array_equal(resize_linear_u8(imgs[0], (40, 48)), imgs[0])
    from PIL import Image
    Image.fromarray(imgs[0][:, :, :3], "RGB").save(str(tmp_path / "rgb_only.png"))
    with pytest.raises(AssertionError):
        read_rgba(str(tmp_path / "rgb_only.png"))
    with pytest.raises(ValueError):
        read_rgba(str(tmp_path / "missing.png"))
    # item: composite over the random background drawn from numpy's global stream
    np.random.seed(7)
    item = ds[1]
    np.random.seed(7)
    bg = torch.tensor(np.random.uniform(low=0, high=1, size=(1, 3)), dtype=torch.float32)[0, :, None, None]
    exp = ds.data["rgb"][1] * ds.data["alpha"][1] + bg * (1 - ds.data["alpha"][1])
    assert torch.equal(item["image"], exp) and torch.equal(item["mask"], ds.data["alpha"][1])
    assert item["pose_indices"] == 1 and item["image_path"].endswith("01.png")
    # empty folder: a valid zero-length dataset
    os.makedirs(tmp_path / "empty")
    assert len(Dataset(16, str(tmp_path / "empty"))) == 0


def test_config_seam_redirects_dataset_and_ema():
    from oi_amd.config import get_obj_from_str
    from oi_amd.dataset import Dataset
    from oi_amd.ema import EMA
    assert get_obj_from_str("src.datasets.eval_dataset.Dataset") is Dataset
    assert get_obj_from_str("src.utils.ema.EMA") is EMA


def small_generator_cfg(R, S, I, K):
    """Reference-named generator config (configs/train.yaml layout) at a small size; shared with the fixture tests."""
    fov, img, img_scene = 10.0, 256, 1588
    cam_dist = float(1 / np.tan(0.5 * fov * np.pi / 180))
    scene_fov = float(2 * np.arctan(img_scene / img * np.tan(0.5 * fov * np.pi / 180)) * 180 / np.pi)
    scene_res = int(R * img_scene / img)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    kw = dict(D=8, W=128, input_ch=3, input_ch_views=3, style_dim=64)
    net = lambda t, **k: {"__target__": t, "kwargs": k}
    return net("src.models.generator.Generator",
               color_network=net("src.models.fields.ColorNetwork", **kw),
               sdf_network=net("src.models.fields.ShapeNetwork",
                               checkpoint_path=os.path.join(root, "tests", "golden", "weights_sdf.npz"), **kw),
               deviation_network=net("src.third_party.neus.models.fields.SingleVarianceNetwork", init_val=0.3),
               light_network=net("src.utils.prior.build_directional_light_optimizable", cam_loc=None, light_loc=None,
                                 ambient_color=0.33, diffuse_color=0.66, specular_color=0, shininess=10),
               camera=net("src.models.camera_network.Camera", cam_dist=cam_dist, resolution=scene_res, fov=scene_fov),
               z_dim=64, resolution=R, scene_resolution=scene_res,
               renderer=net("src.third_party.neus.models.renderer.NeuSRenderer", n_importance=I, n_outside=0,
                            n_samples=S, perturb=1, up_sample_steps=K),
               anneal_end=50000,
               pose_prior=net("src.utils.pose_sampler.Plane", cam_loc=[0, -1, 0], rot_degree_range_scale=360,
                              rot_roll_degree_range_scale=20, xy_range_scale=[6, 3.5]))


def test_field_pack_deepcopy_does_not_inherit_the_hold_depth():
    """FieldPack.hold is a depth counter: a deep copy taken while a forward is in flight (an EMA snapshot from a callback) starts
    released and with empty caches, or its parameter-version walk would be skipped for good (advisor, round 5)."""
    import copy
    from oi_amd.fields import ShapeNetwork, ColorNetwork, FieldPack
    kw = dict(D=8, W=128, input_ch=3, input_ch_views=3, style_dim=64)
    pack = FieldPack(ShapeNetwork(None, **kw), ColorNetwork(**kw))
    pack.hold(True)
    key0 = pack._key
    assert pack._held == 1 and key0 is not None
    twin = copy.deepcopy(pack)
    assert getattr(twin, "_held", 0) == 0 and twin._key is None and twin._packs == {}
    assert twin.sdf_network is not pack.sdf_network and twin.prec == pack.prec
    twin._refresh_key()
    with torch.no_grad():
        twin.sdf_network.sigma_linear.weight.add_(1.0)
    k1 = twin._key
    twin._refresh_key()
    assert twin._key != k1            # the copy's version walk is live
    pack.hold(False)
    assert pack._held == 0


def test_library_ada_draws_match_the_reference_distribution_and_matrix():
    """oi_ada_theta_xint_scale (host-only entry of the library; no GPU): the per-image draws it expands from one seed against
    what src/third_party/ada/augment.py:213-230 prescribes -- t ~ U(-xint_max, xint_max) per axis w.p. xint * p (else 0),
    s = 2^(N(0, 1) * scale_std) w.p. scale * p (else 1), independent gates -- by mean / variance / gate frequency over 2^16
    images, and the sampling matrix it forms against AugmentPipe.theta_for on the SAME draws (1 ulp), symmetric and
    asymmetric margins.  Same seed -> same draws; different seeds differ."""
    from oi_amd.augment import AugmentPipe
    from oi_amd import ops
    f32 = np.float32
    aug = AugmentPipe(xint=1, scale=1)
    n = 1 << 16
    np.random.seed(3)
    th, ts = aug.theta_fast(n, 64, 64, with_draws=True)
    t, s = ts[:, :2].astype(np.float64), ts[:, 2].astype(np.float64)
    # p = 1: every gate open.  U(-1/8, 1/8): mean 0, variance (1/4)^2 / 12; 5 sigma of the sample mean / variance
    var_t = 0.25 ** 2 / 12
    assert np.abs(t).max() < 0.125 and abs(t.mean()) < 5 * np.sqrt(var_t / (2 * n))
    assert abs(t.var() - var_t) < 5 * var_t * np.sqrt(0.8 / (2 * n))          # (kurtosis of a uniform: var of var = 0.8 s^4 / n)
    assert abs(np.corrcoef(t[:, 0], t[:, 1])[0, 1]) < 5 / np.sqrt(n)           # the two axes are independent draws
    ls = np.log2(s)
    assert abs(ls.mean()) < 5 * 0.2 / np.sqrt(n) and abs(ls.std() - 0.2) < 5 * 0.2 / np.sqrt(2 * n)
    assert abs(((ls / 0.2) ** 4).mean() - 3.0) < 0.15 and abs(((ls / 0.2) ** 3).mean()) < 0.06   # a normal: kurtosis 3, no skew
    assert abs(np.corrcoef(t[:, 0], ls)[0, 1]) < 5 / np.sqrt(n)
    # the matrix: augment.py:285-297 on the same draws
    for margins, H, W in ((aug.static_margins(64, 64), 64, 64), ((5, 9, 2, 11), 64, 48)):
        th2, ts2 = ops.ada_theta_xint_scale(77, 512, H, W, margins, *aug.fast_params(), with_draws=True)
        G = np.zeros((512, 3, 3), f32)
        G[:, 0, 0] = G[:, 1, 1] = f32(1) / ts2[:, 2]
        G[:, 0, 2], G[:, 1, 2], G[:, 2, 2] = -np.round(ts2[:, 0] * f32(W)), -np.round(ts2[:, 1] * f32(H)), 1
        ref = aug.theta_for(G, margins, H, W)
        assert np.abs(ref - th2).max() <= 1.2e-7 * max(1.0, np.abs(ref).max())
    # gates at p = 0.3: frequencies, independence of the two gates, closed gates give exactly t = 0 / s = 1
    aug.p.fill_(0.3)
    _, ts = aug.theta_fast(n, 64, 64, seed=12345, with_draws=True)
    on_t, on_s = ts[:, 0] != 0, ts[:, 2] != 1
    sd = np.sqrt(0.3 * 0.7 / n)
    assert abs(on_t.mean() - 0.3) < 5 * sd and abs(on_s.mean() - 0.3) < 5 * sd and abs((on_t & on_s).mean() - 0.09) < 5 * sd
    assert np.all(ts[~on_t][:, 1] == 0) and np.all(ts[~on_s][:, 2] == 1)     # ONE gate for both axes of the translation
    # determinism / seeding
    a = aug.theta_fast(4, 64, 64, seed=9)
    assert np.array_equal(a, aug.theta_fast(4, 64, 64, seed=9)) and not np.array_equal(a, aug.theta_fast(4, 64, 64, seed=10))
    np.random.seed(1); b = aug.theta_fast(4, 64, 64)
    np.random.seed(1); c = aug.theta_fast(4, 64, 64)
    assert np.array_equal(b, c)
    # an overridden sample_G_inv / forward (how tests pin debug_percentile) switches the library draws off
    assert aug.fast_draw_ok()
    aug.sample_G_inv = lambda *a, **k: None
    assert not aug.fast_draw_ok()
