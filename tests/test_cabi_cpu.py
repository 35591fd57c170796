"""CPU-only checks of the drop-in boundary: the C-ABI library builds/loads, exports every symbol that
include/oi_hip.h declares, and the host modules import and keep the reference's names (no compute)."""
import os
import re

import pytest
import torch

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "oi_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(oi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from oi_amd import lib
    L = lib.load()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/oi_hip.h but not exported"
    assert set(names) == set(lib.declared_symbols()), set(names) ^ set(lib.declared_symbols())
    assert L.oi_arch() == b"gfx950"
    assert L.oi_mlp_packed_bytes(0) == 2816 * 4 + 16 * 65536 + 8 * 65536  # header, 16 MFMA images, 8 plain fp32 matrices
    assert L.oi_mlp_packed_bytes(2) == 2816 * 4 + 16 * 32768 + 8 * 65536
    assert L.oi_mlp_packed_bytes(3) == 2816 * 4 + 16 * 98304 + 8 * 65536


def test_ops_fail_loudly_without_gpu_tensors():
    from oi_amd import ops, lib
    with pytest.raises(lib.OiHipError):
        ops.midpoints(torch.zeros(4, 3), torch.zeros(4, 3), torch.zeros(4, 8), 0.1)


def test_modules_mirror_reference_names():
    from oi_amd.config import TARGET_MAP, get_obj_from_str
    for ref_target in TARGET_MAP:
        assert get_obj_from_str(ref_target) is not None
    from oi_amd.fields import ShapeNetwork, ColorNetwork, SingleVarianceNetwork
    from oi_amd.discriminator import ADADiscriminatorView
    s = ShapeNetwork(None, D=8, W=128, input_ch=3, input_ch_views=3, style_dim=64)
    assert sum(p.numel() for p in s.parameters()) + sum(
        p.numel() for p in ColorNetwork(D=8, W=128, input_ch=3, input_ch_views=3, style_dim=64).parameters()) + 1 + 6 == 295755
    for m in ("style", "forward", "sdf", "gradient", "pts_linears", "sigma_linear"):
        assert hasattr(s, m)
    d = ADADiscriminatorView(out_dim_position=6, out_dim_latent=0,
                             aug={"__target__": "src.third_party.ada.augment.AugmentPipe", "kwargs": {"scale": 1, "xint": 1}},
                             aug_p=1, img_size=128, in_dim=3, last_bias=False, n_feat=512, out_dim=7)
    assert sum(p.numel() for p in d.parameters()) == 2844160
    import copy
    copy.deepcopy(s)  # EMA copies (src/utils/ema.py:11-12)


def test_unsupported_configurations_raise():
    from oi_amd.fields import ShapeNetwork
    from oi_amd.augment import AugmentPipe
    with pytest.raises(NotImplementedError):
        ShapeNetwork(None, D=8, W=256, input_ch=3, input_ch_views=3, style_dim=64)
    with pytest.raises(NotImplementedError):
        AugmentPipe(brightness=1)
