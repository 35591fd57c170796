"""`build_from_config` -- the reference's plugin seam (tu/utils/config.py:11-25) with one addition:
dotted `__target__` paths that name the reference's hot-path classes are redirected to their
oi_amd drop-ins, so an unmodified reference config builds the MI355X implementation."""
import importlib

TARGET_MAP = {
    "src.models.generator.Generator": "oi_amd.generator.Generator",
    "src.models.fields.ShapeNetwork": "oi_amd.fields.ShapeNetwork",
    "src.models.fields.ColorNetwork": "oi_amd.fields.ColorNetwork",
    "src.third_party.neus.models.fields.SingleVarianceNetwork": "oi_amd.fields.SingleVarianceNetwork",
    "src.third_party.neus.models.renderer.NeuSRenderer": "oi_amd.renderer.NeuSRenderer",
    "src.models.discriminator.DCDiscriminator": "oi_amd.discriminator.DCDiscriminator",
    "src.models.discriminator.ADADiscriminator": "oi_amd.discriminator.ADADiscriminator",
    "src.models.discriminator.ADADiscriminatorView": "oi_amd.discriminator.ADADiscriminatorView",
    "src.third_party.ada.augment.AugmentPipe": "oi_amd.augment.AugmentPipe",
    "src.models.camera_network.Camera": "oi_amd.camera.Camera",
    "src.utils.prior.build_directional_light_optimizable": "oi_amd.lighting.build_directional_light_optimizable",
    "src.utils.pose_sampler.Plane": "oi_amd.pose.Plane",
    "src.loss.gan.GANLoss": "oi_amd.losses.GANLoss",
    "src.loss.gan.compute_grad2": "oi_amd.losses.compute_grad2",
    "src.loss.position.PositionLoss": "oi_amd.losses.PositionLoss",
    "src.loss.position.linear_increase": "oi_amd.losses.linear_increase",
    "src.datasets.eval_dataset.Dataset": "oi_amd.dataset.Dataset",
    "src.utils.ema.EMA": "oi_amd.ema.EMA",
}


def get_obj_from_str(string):
    string = TARGET_MAP.get(string, string)
    module, cls = string.rsplit(".", 1)
    return getattr(importlib.import_module(module), cls)


def build_from_config(config, **kwargs):
    if config is None:
        return None
    if isinstance(config, str):
        return get_obj_from_str(config)
    return get_obj_from_str(config["__target__"])(**config.get("kwargs", {}), **kwargs)
