"""ctypes binding of liboi_hip.so (C ABI declared in include/oi_hip.h).

The library handle is module-global (never stored on nn.Module instances, so modules stay
deepcopy-able for the EMA copies the reference trainer makes, src/utils/ema.py:11-12).
There is NO fallback: if the HIP library cannot be loaded every op raises."""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# OI_LIB: an alternative build of the same library (A/B experiments, tools/dbg/build_variants.sh); default = the in-tree build
LIB_PATH = os.environ.get("OI_LIB") or os.path.join(_HERE, "liboi_hip.so")
_lock = threading.Lock()
_lib = None

OI_PREC_F32, OI_PREC_BF16X3, OI_PREC_BF16, OI_PREC_BF16X6, OI_PREC_F16X3 = 0, 1, 2, 3, 4
OI_MLP_BLOB_READY = 1
PRECISIONS = {"f32": OI_PREC_F32, "fp32": OI_PREC_F32, "bf16x3": OI_PREC_BF16X3, "bf16": OI_PREC_BF16,
              "bf16x6": OI_PREC_BF16X6, "f16x3": OI_PREC_F16X3}

_vp, _i, _ll, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_size_t


class CompositeParams(ctypes.Structure):
    """Mirror of `oi_composite_params` (include/oi_hip.h)."""
    _fields_ = ([(n, _vp) for n in ("sdf", "grad", "rgb", "dists", "mid_z", "rays_o", "rays_d", "light_dir", "bg",
                                    "variance", "light")] +
                [("cos_anneal_ratio", _f), ("N", _ll), ("T", _i), ("B", _i)] +
                [(n, _vp) for n in ("weights", "cdf", "alpha", "inside_sphere", "pts_norm", "weight_sum", "weight_max",
                                    "color_fine", "image_no_bg", "image", "shading", "normal", "mask", "z_map",
                                    "specular_map", "diffuse_map", "reduce4", "block_partials", "stats16", "stats_ticket")] +
                [("image_planar", _i)])


PREP_MAX_B = 8


class PrepParams(ctypes.Structure):
    """Mirror of `oi_prep_params` (include/oi_hip.h)."""
    _fields_ = ([(n, (_f * 16) * PREP_MAX_B) for n in ("b2w", "w2b", "c2b")] +
                [("offs", (_f * 2) * PREP_MAX_B), ("bg", (_f * 3) * PREP_MAX_B)] +
                [(n, _i) for n in ("B", "R", "S", "NL")] +
                [(n, _vp) for n in ("kinv", "light_direction", "jitter", "style_w", "style_b", "z", "gw", "gb", "bw", "bb",
                                    "pose_out", "rays_o", "rays_d", "near_", "far_", "light_dir", "z_coarse", "pts_coarse",
                                    "w_out", "gamma", "beta")] +
                [("jitter_normal", _i), ("f3_packed", _vp), ("f3_blob", _vp)])


class CompositeGrads(ctypes.Structure):
    """Mirror of `oi_composite_grads` (include/oi_hip.h)."""
    _fields_ = [(n, _vp) for n in ("g_weights", "g_weight_sum", "g_color_fine", "g_image_no_bg", "g_image", "g_shading",
                                   "g_normal", "g_mask", "g_z_map", "g_specular_map", "g_diffuse_map", "g_reduce4",
                                   "d_sdf", "d_grad", "d_rgb", "d_variance", "d_light", "d_light_dir", "ray_partials")]


_SIGS = {
    "oi_version": (_i, []),
    "oi_arch": (ctypes.c_char_p, []),
    "oi_last_error": (ctypes.c_char_p, []),
    "oi_film_params": (_i, [_vp] * 10 + [_i, _i, _vp]),
    "oi_film_params_bwd": (_i, [_vp] * 16 + [_i, _i, _vp]),
    "oi_mlp_packed_bytes": (_sz, [_i]),
    "oi_mlp_pack_weights": (_i, [_vp] * 11 + [_i, _vp]),
    "oi_mlp_pack_status": (_i, [_vp, _vp]),
    "oi_mlp_scratch_bytes": (_sz, [_i, _ll]),
    "oi_mlp_scratch_bytes_prec": (_sz, [_i, _ll, _i]),
    "oi_sdf_mlp_fwd": (_i, [_vp] * 9 + [_i, _ll, _i, _i, _vp]),
    "oi_sdf_mlp_fwd_ex": (_i, [_vp] * 9 + [_i, _ll, _i, _i, _i, _vp]),
    "oi_mlp_f3_blob_offset": (_sz, [_i, _ll]),
    "oi_mlp_f3_blob_bytes": (_sz, []),
    "oi_selftest_sincos": (_i, [_vp, _vp, _vp, _ll, _i, _vp]),
    "oi_selftest_q24": (_i, [_vp, _vp, _ll, _i, _vp]),
    "oi_selftest_cu_slots": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "oi_mlp_bwd_scratch_bytes": (_sz, [_i, _ll]),
    "oi_mlp_bwd_scratch_bytes_capped": (_sz, [_i, _ll, _sz]),
    "oi_mlp_bwd_small_floats": (_i, []),
    "oi_sdf_mlp_bwd": (_i, [_vp] * 15 + [_sz, _i, _ll, _i, _i, _vp]),
    "oi_sdf_mlp_bwd_feat": (_i, [_vp] * 16 + [_sz, _i, _ll, _i, _i, _vp]),
    "oi_color_head_fwd": (_i, [_vp] * 4 + [_ll] + [_vp] * 5 + [_i, _ll, _vp]),
    "oi_color_head_bwd_workspace_bytes": (_sz, [_i, _ll]),
    "oi_color_head_bwd": (_i, [_vp] * 4 + [_ll] + [_vp] * 9 + [_ll] + [_vp] * 5 + [_sz, _i, _ll, _vp]),
    "oi_composite_bwd": (_i, [ctypes.POINTER(CompositeParams), ctypes.POINTER(CompositeGrads), _vp]),
    "oi_render_stats": (_i, [_vp, _i, _ll, _i, _vp, _vp]),
    "oi_composite_num_blocks": (_i, [_ll]),
    "oi_gen_rays": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "oi_gen_rays_light": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "oi_coarse_samples": (_i, [_vp] * 5 + [_ll, _i, _vp, _vp, _vp]),
    "oi_prep_render": (_i, [_vp, _vp]),
    "oi_upsample": (_i, [_vp] * 4 + [_ll, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "oi_upsample_mid": (_i, [_vp] * 4 + [_ll, _i, _i, _f, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "oi_merge_sorted": (_i, [_vp] * 4 + [_ll, _i, _i, _vp, _vp, _vp]),
    "oi_midpoints": (_i, [_vp] * 3 + [_ll, _i, _f, _vp, _vp, _vp, _vp]),
    "oi_composite_fwd": (_i, [ctypes.POINTER(CompositeParams), _vp]),
    "oi_conv4x4_fwd": (_i, [_vp] * 4 + [_i] * 7 + [_f, _vp]),
    "oi_conv4x4_fwd_into": (_i, [_vp] * 4 + [_i] * 7 + [_f, _f, _i, _vp]),
    "oi_conv4x4_fwd_arena": (_i, [_vp] * 4 + [_i] * 7 + [_f, _f, _i, ctypes.c_longlong, _vp]),
    "oi_conv4x4_dgrad": (_i, [_vp] * 3 + [_i] * 7 + [_vp]),
    "oi_conv4x4_wgrad": (_i, [_vp] * 3 + [_i] * 7 + [_vp]),
    "oi_conv4x4_dgrad_masked": (_i, [_vp, _vp, _f, _vp, _vp] + [_i] * 7 + [_vp]),
    "oi_conv4x4_wgrad_masked": (_i, [_vp, _vp, _f, _vp, _vp, _i] + [_i] * 7 + [_vp]),
    "oi_conv4x4_bwd_masked": (_i, [_vp, _vp, _f, _vp, _vp, _vp, _vp, _i] + [_i] * 7 + [_vp]),
    "oi_conv4x4_bwd_pre": (_i, [_vp, _vp, _f, _vp, _vp, _f, _vp, _vp, _i] + [_i] * 7 + [_vp]),
    "oi_conv4x4_dgrad_pre": (_i, [_vp, _vp, _vp, _f, _vp] + [_i] * 7 + [_vp]),
    "oi_lrelu_mask_mul": (_i, [_vp] * 3 + [_ll, _f, _vp]),
    "oi_channel_sum": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "oi_upfirdn2d": (_i, [_vp] * 3 + [_i] * 14 + [_f, _vp]),
    "oi_ada_geom_fwd": (_i, [_vp] * 5 + [_i] * 8 + [_vp]),
    "oi_ada_geom_sep_supported": (_i, [_i] * 3),
    "oi_ada_geom_sep_fwd": (_i, [_vp] * 5 + [_i] * 8 + [_vp]),
    "oi_ada_geom_sep_adj": (_i, [_vp] * 5 + [_i] * 8 + [_vp]),
    "oi_ada_pad_up2": (_i, [_vp] * 3 + [_i] * 8 + [_vp]),
    "oi_disc_fwd_small_workspace_floats": (_sz, [_i] * 6),
    "oi_disc_fwd_small": (_i, [_vp] * 4 + [_i] * 4 + [_vp] * 9 + [_i] * 6 + [_f, _vp]),
    "oi_disc_large_packed_bytes": (_sz, [_vp, _i, _i]),
    "oi_disc_large_pack": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "oi_disc_large_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "oi_disc_fwd_large": (_i, [_vp] * 5 + [_sz, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "oi_disc_graph_create": (_i, [_vp, _i, _vp] + [_i] * 4 + [_vp] * 9 + [_i] * 6 + [_f]),
    "oi_disc_graph_launch": (_i, [_vp, _vp, _vp, _vp]),
    "oi_disc_fwd_small128_workspace_floats": (_sz, [_i] * 6),
    "oi_disc_fwd_small128": (_i, [_vp] * 4 + [_i] * 4 + [_vp] * 10 + [_i] * 4 + [_f, _vp]),
    "oi_disc_graph_create128": (_i, [_vp, _i, _vp] + [_i] * 4 + [_vp] * 10 + [_i] * 4 + [_f]),
    "oi_disc_graph_launch_eager": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "oi_ada_theta_xint_scale": (_i, [ctypes.c_ulonglong] + [_i] * 7 + [_f] * 4 + [_vp, _vp]),
    "oi_disc_graph_launch_ada": (_i, [_vp, _vp, ctypes.c_ulonglong, _f, _f, _f, _f, _vp, _i, _vp]),
    "oi_disc_graph_destroy": (None, [_vp]),
    "oi_outputs_prezeroed_stream": (_i, [_vp, _i]),
    "oi_light_dir_fwd": (_i, [_vp, _vp, _vp, _i, _vp]),
    "oi_light_dir_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "oi_gan_losses_fwd": (_i, [_vp] * 5 + [_f, _vp, _i, _i, _ll, _vp]),
    "oi_gan_losses_bwd": (_i, [_vp] * 6 + [_f, _vp, _vp, _vp, _i, _i, _ll, _vp]),
    "oi_stage_inputs": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    "oi_render_scalars_fwd": (_i, [_vp, _f, _vp, _vp]),
    "oi_zero_fill": (_i, [_vp, _ll, _vp]),
    "oi_scalar_glue": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "oi_render_scalars_bwd": (_i, [_vp, _vp, _vp, _f, _vp, _vp]),
    "oi_weighted_sum_fwd": (_i, [_vp, _vp, _i, _vp, _vp]),
    "oi_weighted_sum_bwd": (_i, [_vp, _vp, _i, _vp, _vp]),
    "oi_affine_grid_sample_fwd": (_i, [_vp] * 3 + [_i] * 6 + [_vp]),
    "oi_affine_grid_sample_bwd": (_i, [_vp] * 3 + [_i] * 6 + [_vp]),
    "oi_fused_bias_act": (_i, [_vp] * 4 + [_i, _i, _f, _f, _ll, _ll, _i, _vp]),
    "oi_grid_sample_fwd": (_i, [_vp] * 3 + [_i] * 6 + [_vp]),
    "oi_grid_sample_bwd": (_i, [_vp] * 5 + [_i] * 6 + [_vp]),
    "oi_reflect_pad_fwd": (_i, [_vp, _vp] + [_i] * 7 + [_vp]),
    "oi_reflect_pad_bwd": (_i, [_vp, _vp] + [_i] * 7 + [_vp]),
    "oi_mt_chunk_elems": (_i, []),
    "oi_multi_adam": (_i, [_vp, _i, _f, _f, _f, _f, _f, _f, _vp]),
    "oi_multi_rmsprop": (_i, [_vp, _i, _f, _f, _f, _vp]),
    "oi_multi_lerp": (_i, [_vp, _i, _f, _vp]),
    "oi_multi_copy": (_i, [_vp, _i, _vp]),
}

# entry points added by later source files (backward kernels); bound when present in the .so
_OPTIONAL_SIGS = {}


class OiHipError(RuntimeError):
    pass


def declared_symbols():
    return sorted(_SIGS)


def load():
    """Load (once) and return the ctypes handle.  Raises OiHipError when the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        # PyTorch bundles its own libamdhip64.so.7; liboi_hip.so needs the same SONAME.  Importing torch FIRST makes the
        # dynamic loader resolve our dependency to the runtime torch already mapped -- one HIP runtime per process.
        # (Loaded the other way round, /opt/rocm's copy comes in through our RUNPATH, torch then maps its own by
        # path, and launches on torch's streams fail with "no ROCm-capable device is detected".)
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise OiHipError(
                f"{LIB_PATH} not found: build it with `python object-intrinsics_amd/build.py` (hipcc, gfx950). "
                "oi_amd has no CPU or PyTorch fallback for its kernels.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in {**_SIGS, **_OPTIONAL_SIGS}.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                if name in _OPTIONAL_SIGS:
                    continue
                raise OiHipError(f"{LIB_PATH} does not export {name}; rebuild the library")
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().oi_last_error()
        raise OiHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
