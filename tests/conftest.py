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
