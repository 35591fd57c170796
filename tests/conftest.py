import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "object-intrinsics_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as f:
        return {k: torch.from_numpy(np.asarray(f[k])) for k in f.files}


def sub_sd(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


@pytest.fixture(scope="session")
def sdf_sd():
    return load_golden("weights_sdf")


@pytest.fixture(scope="session")
def col_sd():
    return load_golden("weights_color")


# Relative jump of |d loss / d variance| in F13's second iteration when near / far move in their last bit: MEASURED on the oracle
# (tests/test_oracle_golden.py::test_f13_two_iterations_on_the_oracle_and_last_bit_sensitivity: 0.08583 <-> 0.08627).  The GPU
# test accepts that one scalar on either mode (ref, ref x (1 + this)) at the ordinary 2e-3.
F13_VARIANCE_GRAD_SENSITIVITY = 5.2e-3


def maxdiff(a, b):
    return float((a.detach().double() - b.detach().double()).abs().max())


# ------------------------------------------------------------------------------------------------------------------
# Measured margins.  The gradient tests report the worst relative error of every tensor they check; with
# OI_MARGIN_OUT=<file.json> the collection is written at the end of the session (tools/grad_margin.py turns it into the
# table of DESIGN.md section 5).  The test tolerances are set from these measurements (<= 3x the native-fp32 error).
# ------------------------------------------------------------------------------------------------------------------
MARGINS = {}


def record_margin(case, name, err):
    d = MARGINS.setdefault(case, {})
    d[name] = max(float(err), d.get(name, 0.0))


def pytest_sessionfinish(session, exitstatus):
    out = os.environ.get("OI_MARGIN_OUT")
    if out and MARGINS:
        import json
        with open(out, "w") as fh:
            json.dump(MARGINS, fh, indent=1, sort_keys=True)


# ------------------------------------------------------------------------------------------------------------------
# F14 (tests/golden/f14_discriminator_128.npz, oracle/gen_golden_r5.py): the shipped 128 x 128 discriminators.  The fixture
# stores the weight recipe's check sums and strided samples of the two large gradient tensors instead of 4 x 11.4 MB.
# ------------------------------------------------------------------------------------------------------------------
F14_GRAD_STRIDE = {"blocks.2.weight": 5, "blocks.3.weight": 17, "blocks.4.weight": 61}
F14_NETS = {"v_": dict(view=True, in_dim=3, out_dim=7), "m_": dict(view=False, in_dim=1, out_dim=1)}


def f14_weights(g, tag):
    """The network's weights from the seeded recipe, verified against the fixture's check sums (a different torch whose
    generator drew other numbers must fail HERE, not as a parity error)."""
    import oi_oracle as O
    kw = F14_NETS[tag]
    chans = [kw["in_dim"], 32, 64, 128, 256, 512]
    shapes = {f"blocks.{i}.weight": (chans[i + 1], chans[i], 4, 4) for i in range(5)}
    shapes["conv_out.weight"] = (kw["out_dim"], 512, 4, 4)
    wsd = O.seeded_conv_weights(shapes, int(g[tag + "seed"]))
    for k, v in wsd.items():
        ref = g[tag + "wsum." + k].double()
        got = torch.stack([v.double().sum(), (v.double() ** 2).sum()])
        assert float((got - ref).abs().max()) <= 1e-9 * max(1.0, float(ref.abs().max())), ("weight recipe check sum", tag, k)
        assert torch.equal(v.flatten()[:: max(1, v.numel() // 64)][:64], g[tag + "wsample." + k]), ("weight recipe samples", tag, k)
    return wsd


def f14_grad_errors(g, t, named_grads):
    """{name: relative error} of weight gradients against the fixture (whole tensors, or strided samples + exact sums)."""
    errs = {}
    for k, gr in named_grads:
        gr = gr.detach().double().cpu()
        st = F14_GRAD_STRIDE.get(k)
        if st is None:
            ref = g[t + "g." + k].double()
            errs[k] = float((gr - ref).abs().max() / ref.abs().max())
        else:
            ref = g[t + "gs." + k].double()
            errs[k] = float((gr.flatten()[::st] - ref).abs().max() / ref.abs().max())
            sums = g[t + "gsum." + k].double()
            errs[k + "(sum)"] = float(abs(gr.sum() - sums[0]) / gr.abs().sum())
            errs[k + "(sum of squares)"] = float(abs((gr ** 2).sum() - sums[1]) / sums[1])
    return errs
