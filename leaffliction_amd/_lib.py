"""ctypes binding of libleafhip.so (the C ABI declared in include/leafhip.h).

The product path has no CPU fallback: if the HIP library is missing or a symbol is
absent, importing/using it raises.  `load()` only dlopens the library (no GPU needed, so
the CPU test-suite can check the export table); every call goes through `call()` which
turns a negative return code into a `LeafHipError` carrying `lf_last_error()`.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_LIB = None

LIB_PATH = Path(__file__).resolve().parent / "libleafhip.so"

c_void_p, c_int, c_size_t, c_u64, c_float, c_double = (
    C.c_void_p, C.c_int, C.c_size_t, C.c_uint64, C.c_float, C.c_double)
P = c_void_p  # every device/host buffer crosses the ABI as a raw pointer

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/leafhip.h
SIGNATURES = {
    "lf_version": [],
    "lf_last_error": [],
    "lf_pack_hwc_u8_to_nchw_f32": [P, P, c_int, c_int, c_int, P, P, P],
    "lf_hist_u8": [P, P, c_int, c_int, c_int, P],
    "lf_autocontrast_lut": [P, P, P, c_int, P],
    "lf_lut_apply_u8": [P, P, P, c_int, c_int, c_int, P],
    "lf_gather_rows_u8": [P, P, P, c_int, c_size_t, P],
    "lf_flip_u8": [P, P, P, c_int, c_int, c_int, P],
    "lf_noise_wrap_add_u8": [P, P, P, c_size_t, P],
    "lf_add_wrap_u8": [P, P, P, c_size_t, P],
    "lf_noise_philox_add_u8": [P, P, c_size_t, c_u64, c_float, P],
    "lf_noise_hist_u8": [P, P, P, P, c_int, c_int, c_int, c_u64, c_float, P],
    "lf_mask_composite_u8": [P, P, P, c_int, c_int, c_int, c_int, P],
    "lf_rgb2hsv_u8": [P, P, c_size_t, P],
    "lf_rgb2gray_u8": [P, P, c_size_t, P],
    "lf_gauss_blur_u8": [P, P, c_int, c_int, c_int, c_int, P, c_int, P],
    "lf_hsv_region_stats": [P, P, P, c_int, c_int, c_int, P],
    "lf_jpeg_fdct_quant_u8": [P, P, c_int, c_int, c_int, c_int, P],
    "lf_jpeg_quant_tables": [c_int, P, P],
    "lf_jpeg_file_bound": [c_int, c_int],
    "lf_jpeg_write_file": [P, c_int, c_int, c_int, P, c_size_t],
    "lf_jpeg_read_file": [P, c_size_t, P, c_size_t, P, P, P],
    "lf_jpeg_entropy_workspace": [c_int, c_size_t],
    "lf_jpeg_entropy_u8": [P, c_size_t, P, c_size_t, c_int, c_int, c_int, P, c_size_t, P],
    "lf_jpeg_wrap_scan": [P, c_size_t, c_int, c_int, c_int, P, c_size_t],
    "lf_jpeg_fdct_groups": [c_int, c_int],
    "lf_jpeg_fdct_quant_items_u8": [P, P, P, c_int, C.c_long, c_int, P],
    "lf_jpeg_entropy_items_u8": [P, P, P, c_size_t, c_int, P, c_size_t, P],
    "lf_legacy_normal_u8": [C.c_uint32, c_double, c_double, c_size_t, P, P],
    "lf_jpeg_decode_workspace": [c_int, c_int, c_int],
    "lf_copy_rows": [P, c_size_t, P, c_size_t, c_size_t, c_size_t, c_int, P],
    "lf_legacy_normal_batch_u8": [P, C.c_double, C.c_double, c_size_t, P, c_size_t, c_int, P, P],
    "lf_jpeg_scan_aux_offset": [c_int, c_int],
    "lf_jpeg_scan_prepare": [P, c_size_t, P, c_size_t, P, P, P],
    "lf_jpeg_huffman_u8": [P, c_size_t, c_int, c_int, c_int, P, c_int, P],
    "lf_jpeg_idct_rgb_u8": [P, c_size_t, P, c_size_t, P, c_int, c_int, c_int, P, c_size_t, P],
    "lf_inclusive_mask_workspace": [c_int, c_int, c_int],
    "lf_inclusive_mask_u8": [P, P, c_int, c_int, c_int, c_int, c_int, P, P, c_size_t, P],
    "lf_blur_saliency_workspace": [c_int, c_int, c_int],
    "lf_blur_saliency_u8": [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P,
                            c_size_t, P],
    "lf_warp_bicubic_u8": [P, P, P, c_int, c_int, c_int, c_int, P],
    "lf_affine_nearest_fixed_u8": [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P],
    "lf_resample_tile_u8": [P, P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P, P, c_int, c_int, P],
    "lf_resample_u8": [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P, P, c_int,
                       c_int, P],
    "lf_conv2d_f32": [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, c_int, P],
    "lf_conv2d_bf16_weight_elems": [c_int, c_int, c_int],
    "lf_conv2d_bf16_prep_weights": [P, P, c_int, c_int, c_int, P],
    "lf_conv2d_bf16_f32": [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P],
    "lf_conv2d_bf16_act": [P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int,
                           P, P, c_int, P],
    "lf_conv2d_bf16_act_mean_workspace": [c_int, c_int, c_int, c_int, c_int, c_int, c_int],
    "lf_conv2d_bf16_act_mean": [P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P, P, c_int, P, P,
                                c_size_t, P],
    "lf_gap_bf16": [P, P, c_int, c_int, c_int, P, P, c_int, P],
    "lf_block_tail_fwd_bf16": [P, P, P, P, P, P, P, c_int, P, c_int, c_int, c_int, c_int, P],
    "lf_conv2d_bf16_stats_tiles": [c_int, c_int, c_int, c_int, c_int, c_int, c_int],
    "lf_conv2d_bf16_train": [P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, c_int, P,
                             c_size_t, P, P, P, P, c_int, P],
    "lf_conv2d_wgrad_bf16_workspace": [c_int, c_int, c_int, c_int, c_int, c_int],
    "lf_conv2d_wgrad_bf16": [P, P, P, P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P,
                             c_int, P, c_size_t, P],
    "lf_gap_stats_bf16": [P, P, P, c_int, c_int, c_int, P, P, c_int, P],
    "lf_block_tail_fwd_train_bf16": [P, P, P, P, P, P, P, c_int, P, P, P, c_int, c_int, c_int, c_int, P],
    "lf_block_tail_bwd_bf16": [P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "lf_bcast_planes_bf16": [P, P, c_int, c_int, c_float, P],
    "lf_cast_f32_bf16": [P, P, c_size_t, P],
    "lf_cast_bf16_f32": [P, P, c_size_t, P],
    "lf_conv2d_variant": [c_int, c_int, c_int, c_int],
    "lf_conv2d_wgrad_variant": [c_int, c_int, c_int, c_int, c_int, c_int],
    "lf_conv2d_stats_tiles": [c_int, c_int, c_int, c_int, c_int, c_int],
    "lf_conv2d_stats_f32": [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P, P, c_size_t, P],
    "lf_conv2d_bnbwd_f32": [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, P,
                            c_size_t, P],
    "lf_conv2d_dgrad_weights_f32": [P, P, c_int, c_int, c_int, P],
    "lf_conv2d_wgrad_workspace": [c_int, c_int, c_int, c_int, c_int, c_int],
    "lf_conv2d_wgrad_f32": [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P,
                            c_size_t, P],
    "lf_conv2d_wgrad_bn_supported": [c_int, c_int, c_int, c_int, c_int, c_int],
    "lf_conv2d_wgrad_bn_f32": [P, P, P, P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P,
                               c_int, P, c_size_t, P],
    "lf_conv2d_wgrad_reduce_f32": [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P],
    "lf_input_stage_f32": [P, P, c_int, c_int, c_int, P, P, P, P, P],
    "lf_scale_shift_act_f32": [P, P, c_int, c_int, c_int, P, P, c_int, P],
    "lf_bn_workspace": [c_int],
    "lf_bn_train_stats_f32": [P, c_int, c_int, c_int, P, P, P, P, c_float, c_float, P, P, P, P,
                              P, c_size_t, P],
    "lf_bn_train_stats_tiles_f32": [P, C.c_longlong, c_int, c_int, c_int, P, P, P, P, c_float, c_float, P, P, P, P, P, c_size_t, P],
    "lf_bn_bwd_sums_f32": [P, P, P, P, P, P, P, P, c_int, P, P, P, P, P, P, c_int, c_int, c_int, P,
                           c_size_t, P],
    "lf_bn_bwd_sums_tiles_f32": [P, C.c_longlong, P, P, P, P, P, P, P, P, c_int, c_int, c_int, P,
                                 c_size_t, P],
    "lf_bn_infer_scale_shift_f32": [c_int, P, P, P, P, c_float, P, P, P],
    "lf_bn_bwd_f32": [P, P, P, P, P, P, P, P, c_int, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P,
                      c_size_t, P],
    "lf_gap_f32": [P, P, c_int, c_int, c_int, P, P, c_int, P, P],
    "lf_bcast_planes_f32": [P, P, c_int, c_int, c_float, P],
    "lf_se_fwd_f32": [P, P, P, P, P, P, P, c_int, c_int, c_int, P],
    "lf_se_bwd_workspace": [c_int, c_int, c_int],
    "lf_se_bwd_f32": [P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_float, P, c_size_t, P],
    "lf_block_tail_fwd_f32": [P, P, P, P, P, P, P, c_int, P, P, P, c_int, c_int, c_int, c_int, P],
    "lf_block_tail_bwd_f32": [P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "lf_head_fwd_f32": [P, P, P, P, P, P, c_int, c_int, c_int, P],
    "lf_head_bwd_f32": [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_float, P],
    "lf_mul_f32": [P, P, P, c_size_t, P],
    "lf_adamw_workspace": [c_int],
    "lf_adamw_step_f32": [P, P, P, P, P, P, P, c_int, C.c_longlong, c_float, c_float, c_float,
                          c_float, c_float, c_float, C.c_longlong, c_float, c_int, P, P, c_size_t, P],
    "lf_ema_update_f32": [P, P, c_size_t, c_float, c_int, P],
}
_RESTYPES = {"lf_last_error": C.c_char_p, "lf_conv2d_wgrad_workspace": c_size_t,
             "lf_bn_workspace": c_size_t, "lf_se_bwd_workspace": c_size_t,
             "lf_adamw_workspace": c_size_t, "lf_conv2d_stats_tiles": C.c_longlong,
             "lf_blur_saliency_workspace": c_size_t, "lf_inclusive_mask_workspace": c_size_t, "lf_conv2d_bf16_weight_elems": c_size_t,
             "lf_conv2d_bf16_act_mean_workspace": c_size_t,
             "lf_conv2d_bf16_stats_tiles": C.c_longlong, "lf_conv2d_wgrad_bf16_workspace": c_size_t,
             "lf_jpeg_file_bound": c_size_t, "lf_jpeg_scan_aux_offset": c_size_t, "lf_jpeg_entropy_workspace": c_size_t, "lf_jpeg_wrap_scan": C.c_long, "lf_jpeg_fdct_groups": C.c_long, "lf_jpeg_decode_workspace": c_size_t, "lf_jpeg_write_file": C.c_long, "lf_jpeg_quant_tables": None}


class LeafHipError(RuntimeError):
    pass


def load():
    """dlopen libleafhip.so and bind every declared symbol.  Raises if anything is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7 and libleafhip.so
    # depends on the same SONAME, so whichever is loaded first serves both.  Import torch
    # first, otherwise the system runtime gets loaded here and torch's device pointers and
    # streams (owned by the other runtime instance) are invalid inside the library.
    import torch  # noqa: F401
    path = Path(os.environ.get("LEAFHIP_LIB", LIB_PATH))
    if not path.exists():
        raise LeafHipError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C leaffliction_amd/csrc` (there is no CPU fallback)")
    lib = C.CDLL(str(path))
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise LeafHipError(f"libleafhip.so does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    _LIB = lib
    return lib


def call(name: str, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.lf_last_error()
        raise LeafHipError(f"{name} failed ({rc}): {msg.decode() if msg else ''}")
    return rc
