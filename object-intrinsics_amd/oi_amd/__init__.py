"""oi_amd -- MI355X-native (gfx950) implementation of the object-intrinsics render + GAN hot path.

Host side is Python on PyTorch-ROCm (device memory, streams, autograd graph structure,
torch.distributed/RCCL); all math on the path runs in hand-written HIP kernels of liboi_hip.so,
reached through its C ABI (include/oi_hip.h) with ctypes.  Module names mirror the reference's
plugin seams so that a config only has to swap `__target__` strings:

    src.models.fields.ShapeNetwork / ColorNetwork            -> oi_amd.fields.*
    src.third_party.neus.models.fields.SingleVarianceNetwork -> oi_amd.fields.SingleVarianceNetwork
    src.third_party.neus.models.renderer.NeuSRenderer        -> oi_amd.renderer.NeuSRenderer
    src.models.generator.Generator                           -> oi_amd.generator.Generator
    src.models.discriminator.ADADiscriminator[View]          -> oi_amd.discriminator.*
    src.third_party.ada.augment.AugmentPipe                  -> oi_amd.augment.AugmentPipe
"""
__version__ = "0.1.0"
