"""TEST INFRASTRUCTURE ONLY — harness that imports the *reference* (read-only, /root/reference)
as CPU PyTorch so golden vectors can be generated in the build container.

Nothing under /root/reference is modified or copied.  This module is used only by
`oracle/gen_golden.py`; it never travels to the GPU box in a usable form (the reference tree
does not exist there) and is never imported by the product package.

Shims:
  1. `src.third_party.stylesdf.op` is pre-registered with the arithmetic of the reference's own
     CPU branch of `fused_leaky_relu` (stylesdf/op/fused_act.py:104-116) so that importing the
     package does not JIT-compile CUDA sources.
  2. A TorchFunctionMode rewrites device='cuda' factories to CPU; `.cuda()` is identity.
  3. `collections.MutableMapping` alias for python 3.10 (tu/configs.py:108).
"""
import collections
import collections.abc
import contextlib
import os
import sys
import types

import torch
import torch.nn.functional as F
from torch.overrides import TorchFunctionMode

REF_ROOT = os.environ.get("OI_REFERENCE_ROOT", "/root/reference")


class _CudaToCpu(TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if func is torch.Tensor.cuda:
            return args[0]
        dev = kwargs.get("device")
        if dev is not None and "cuda" in str(dev):
            kwargs["device"] = "cpu"
        return func(*args, **kwargs)


_installed = False


def install():
    """Make `import src....` resolve to the reference tree, CPU only."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"reference tree not found at {REF_ROOT}; golden vectors can only be "
                           f"regenerated in the build container")
    sys.dont_write_bytecode = True
    collections.MutableMapping = collections.abc.MutableMapping
    sys.path.insert(0, REF_ROOT)

    op = types.ModuleType("src.third_party.stylesdf.op")

    def fused_leaky_relu(input, bias=None, negative_slope=0.2, scale=2 ** 0.5):
        if bias is not None:
            rest = [1] * (input.ndim - bias.ndim - 1)
            return F.leaky_relu(input + bias.view(1, bias.shape[0], *rest), negative_slope=0.2) * scale
        return F.leaky_relu(input, negative_slope=0.2) * scale

    op.fused_leaky_relu = fused_leaky_relu
    op.FusedLeakyReLU = type("FusedLeakyReLU", (torch.nn.Module,), {})
    op.upfirdn2d = None
    sys.modules["src.third_party.stylesdf.op"] = op

    torch.nn.Module.cuda = lambda self, *a, **k: self
    _load = torch.load

    def _cpu_load(p, map_location=None, **k):
        k.setdefault("weights_only", False)
        return _load(p, map_location="cpu", **k)

    torch.load = _cpu_load
    _installed = True


@contextlib.contextmanager
def reference_on_cpu():
    """Context in which reference code may be constructed/called (cwd = reference root because
    configs/train.yaml:46 names './checkpoints/sphere_init.pt')."""
    install()
    cwd = os.getcwd()
    os.chdir(REF_ROOT)
    try:
        with _CudaToCpu():
            yield
    finally:
        os.chdir(cwd)
