"""ctypes binding of libxpoint_hip.so (the C ABI declared in include/xpoint_hip.h).

There is no CPU fallback: if the library is missing or a call fails, this raises."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XP_LIB_PATH") or os.path.join(_HERE, "libxpoint_hip.so")      # XP_LIB_PATH: another build of the same library (A/B runs of compiler flags)

_lib = None

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_f = ctypes.c_float
c_l = ctypes.c_int64


class XPointHipError(RuntimeError):
    pass


# name -> argtypes; every function returns int (0 ok) except xp_last_error / xp_version.
c_sz = ctypes.c_size_t
_SIGNATURES = {
    "xp_device_info": [c_i, ctypes.POINTER(c_i), ctypes.POINTER(c_i), ctypes.c_char_p, c_i],
    "xp_knob_info": [c_i, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p)],
    "xp_selective_scan_fwd": [c_p] * 9 + [c_i] * 7 + [c_p],
    "xp_selective_scan_fwd_typed": [c_p] * 9 + [c_i] * 9 + [c_p],
    "xp_cross_scan": [c_p, c_p] + [c_i] * 9 + [c_p],
    "xp_cross_merge": [c_p, c_p] + [c_i] * 9 + [c_p],
    "xp_ss2d_core_fwd": [c_p] * 10 + [c_sz] + [c_i] * 6 + [c_f, c_p],
    "xp_ss2d_core_set_mode": [c_i],
    "xp_set_dense_products": [c_i],
    "xp_set_dense_engine": [c_i],
    "xp_set_dense_override": [ctypes.c_uint64],
    "xp_set_amp_mode": [c_i],
    "xp_round_f16": [c_p, c_p, c_l, c_p],
    "xp_gemm_nt": [c_p] * 7 + [c_i] * 7 + [c_p],
    "xp_conv3x3_nhwc": [c_p] * 6 + [c_i] * 8 + [c_p],
    "xp_split_weights_x3": [c_p, c_p, c_i, c_i, c_p],
    "xp_gemm_nt_x3": [c_p] * 7 + [c_i] * 7 + [c_p],
    "xp_conv3x3_nhwc_x3": [c_p] * 6 + [c_i] * 8 + [c_p],
    "xp_split_weights_h2": [c_p, c_p, c_i, c_i, c_p],
    "xp_gemm_nt_h2": [c_p] * 7 + [c_i] * 7 + [c_p],
    "xp_conv3x3_nhwc_h2": [c_p] * 6 + [c_i] * 8 + [c_p],
    "xp_split_activations_h2": [c_p, c_p, c_l, c_i, c_i, c_p],
    "xp_gemm_nt_h2s": [c_p, c_p, c_p, c_i] + [c_p] * 4 + [c_i] * 6 + [c_p],
    "xp_layernorm_p32": [c_p] * 4 + [c_l, c_i, c_f, c_p],
    "xp_ss2d_core_fwd_ex": [c_p] * 9 + [c_i, c_p, c_sz] + [c_i] * 6 + [c_f, c_p],
    "xp_f32_to_f16": [c_p, c_p, c_l, c_p],
    "xp_gemm_nt_f16": [c_p, c_p, c_p, c_i] + [c_p] * 4 + [c_i] * 7 + [c_p],
    "xp_conv3x3_nhwc_f16": [c_p, c_p, c_p, c_i] + [c_p] * 3 + [c_i] * 8 + [c_p],
    "xp_mlp_fused_h2": [c_p] * 10 + [c_i] * 3 + [c_f, c_p],
    "xp_mlp_fused_h2_pack": [c_p] * 4 + [c_i] * 2 + [c_p],
    "xp_ln_proj_h2_pack": [c_p] * 2 + [c_i] * 2 + [c_p],
    "xp_ln_proj_h2": [c_p] * 6 + [c_i] * 3 + [c_f, c_p],
    "xp_mlp_fused_x3": [c_p] * 7 + [c_i] * 3 + [c_f, c_p],
    "xp_mlp_fused_x3_pack": [c_p] * 4 + [c_i] * 2 + [c_p],
    "xp_ln_proj_x3_pack": [c_p] * 2 + [c_i] * 2 + [c_p],
    "xp_ln_proj_x3": [c_p] * 5 + [c_i] * 3 + [c_f, c_p],
    "xp_layernorm": [c_p] * 4 + [c_l, c_i, c_f, c_i, c_p],
    "xp_dwconv3x3_silu": [c_p] * 3 + [c_i] * 4 + [c_p],
    "xp_stem_conv_ln_gelu": [c_p] * 6 + [c_i] * 4 + [c_f, c_p],
    "xp_depth_to_space_nhwc": [c_p] * 2 + [c_i] * 5 + [c_p],
    "xp_softmax_shuffle": [c_p] * 2 + [c_i] * 6 + [c_p],
    "xp_l2norm_rows": [c_p] * 2 + [c_l, c_i, c_f, c_p],
    "xp_nhwc_to_nchw": [c_p] * 2 + [c_i] * 3 + [c_p],
    "xp_mul_mask": [c_p] * 3 + [c_l, c_p],
    "xp_stage_pair_batch": [c_p] * 6 + [c_l, c_p],
    "xp_maxpool2_nhwc": [c_p] * 2 + [c_i] * 4 + [c_p],
    "xp_costvolume_mean": [c_p] * 3 + [c_i] * 3 + [c_p],
    "xp_ingest_u8": [c_p] + [c_i] * 7 + [c_p] * 3,
    "xp_u8_to_unit_f32": [c_p, c_p, c_l, c_p],
    "xp_copy_to_mapped_host": [c_p, c_p, c_sz, c_p],
    "xp_ctx_create": [c_p, ctypes.POINTER(c_p)],
    "xp_ctx_destroy": [c_p],
    "xp_param_info": [c_p, c_i, ctypes.c_char_p, c_i, ctypes.POINTER(c_sz), ctypes.POINTER(c_sz)],
    "xp_forward_shapes": [c_p, c_i, c_i, c_i, ctypes.POINTER(c_i), ctypes.POINTER(c_i), ctypes.POINTER(c_i)],
    "xp_xpoint_forward": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_sz, c_p, c_p, c_p, c_p, c_p],
    "xp_xpoint_forward_ex": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_sz, c_p, c_p, c_p, c_p, c_p, c_p],
    "xp_xpoint_forward_f16": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_sz, c_p, c_p, c_p, c_p, c_p, c_p],
    "xp_prepare_f16_weights": [c_p, c_p, c_p, c_sz, c_p],
    "xp_mlp_fused_f16": [c_p] * 6 + [c_i] * 3 + [c_p],
    "xp_ln_proj_f16": [c_p, c_p, c_p, c_f, c_p, c_p, c_i, c_i, c_p],
    "xp_ln_mlp_fused_f16": [c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p] + [c_i] * 3 + [c_p],
    "xp_stem_conv_ln_gelu_f16": [c_p] * 6 + [c_i] * 4 + [c_f, c_p],
    "xp_layernorm_f16": [c_p] * 4 + [c_l, c_i, c_f, c_p],
    "xp_dwconv3x3_silu_f16": [c_p] * 4 + [c_i] * 4 + [c_p],
    "xp_depth_to_space_nhwc_f16": [c_p] * 3 + [c_i] * 5 + [c_p, c_p],
    "xp_ss2d_core_fwd_f16": [c_p] * 11 + [c_p, c_sz] + [c_i] * 6 + [c_f, c_p],
    "xp_prepare_split_weights": [c_p, c_p, c_p, c_sz, c_p],
    "xp_box_nms": [c_p, c_p, c_p, c_sz, c_i, c_i, c_i, c_f, c_f, c_f, c_i, c_i, c_i, ctypes.POINTER(c_i), c_p],
    "xp_box_nms_check": [c_p, c_i, c_i, c_i, ctypes.POINTER(c_i), c_p],
    "xp_extract_keypoints": [c_p, c_p, c_f, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_sz, c_p],
    "xp_sample_descriptors": [c_p] * 4 + [c_i] * 7 + [c_p],
    "xp_match_mnn": [c_p, c_p, c_p] + [c_i] * 8 + [c_p] * 9 + [c_sz, c_p],
    "xp_match_knn2": [c_p, c_p, c_p] + [c_i] * 7 + [c_p, c_p, c_p, c_sz, c_p],
    "xp_match_threshold": [c_p, c_p, c_p] + [c_i] * 7 + [ctypes.c_double, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_sz, c_p],
    "xp_match_stats": [c_p, c_p] + [c_i] * 7 + [c_p, c_p],
    "xp_points_min_dist": [c_p, c_i, c_p, c_i, c_p, c_p],
    "xp_gather_match_points": [c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p],
    "xp_find_homography": [c_p, c_p, c_p, c_i, c_i, c_f, c_i, ctypes.c_uint, c_p, c_p, c_p, c_p, c_sz, c_p],
    "xp_warp_perspective": [c_p, c_p, c_p] + [c_i] * 9 + [c_p],
    "xp_prof_enable": [c_i],
    "xp_prof_filter": [ctypes.c_char_p],
    "xp_prof_reset": [],
    "xp_prof_count": [],
    "xp_prof_get": [c_i, ctypes.c_char_p, c_i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i),
                    ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)],
}
# size queries: (restype size_t / int, argtypes)
_SIZE_QUERIES = {
    "xp_knob_count": (c_i, []),
    "xp_weights_numel": (c_sz, [c_p]),
    "xp_param_count": (c_i, [c_p]),
    "xp_forward_workspace_bytes": (c_sz, [c_p, c_i, c_i, c_i]),
    "xp_ss2d_core_workspace_bytes": (c_sz, [c_i] * 4),
    "xp_split_weights_x3_bytes": (c_sz, [c_i] * 2),
    "xp_split_weights_h2_bytes": (c_sz, [c_i] * 2),
    "xp_p32_bytes": (c_sz, [c_l, c_i]),
    "xp_gemm_nt_h2s_applies": (c_i, [c_i, c_i]),
    "xp_ss2d_core_p32_supported": (c_i, [c_i] * 4),
    "xp_split_weights_bytes": (c_sz, [c_p]),
    "xp_f16_weights_bytes": (c_sz, [c_p]),
    "xp_mlp_fused_f16_supported": (c_i, [c_i, c_i]),
    "xp_ss2d_core_f16_wants_f32_copies": (c_i, [c_i] * 4),
    "xp_mlp_fused_x3_supported": (c_i, [c_i, c_i]),
    "xp_get_dense_products": (c_i, []),
    "xp_get_dense_engine": (c_i, []),
    "xp_get_dense_override": (ctypes.c_uint64, []),
    "xp_get_amp_mode": (c_i, []),
    "xp_mlp_fused_x3_pack_bytes": (c_sz, [c_i, c_i, c_i]),
    "xp_mlp_fused_h2_pack_bytes": (c_sz, [c_i, c_i, c_i]),
    "xp_ln_proj_h2_pack_bytes": (c_sz, [c_i, c_i]),
    "xp_ln_proj_x3_pack_bytes": (c_sz, [c_i, c_i]),
    "xp_find_homography_workspace_bytes": (c_sz, [c_i]),
    "xp_box_nms_workspace_bytes": (c_sz, [c_i] * 4),
    "xp_match_workspace_bytes": (c_sz, [c_i] * 4),
    "xp_match_cand_cap": (c_i, []),
    "xp_extract_keypoints_workspace_bytes": (c_sz, [c_i] * 3),
}


def exported_symbols():
    """Symbols include/xpoint_hip.h declares (parsed from the header; used by the CPU load test)."""
    import re
    hdr = os.path.join(_HERE, "..", "include", "xpoint_hip.h")
    txt = open(hdr).read()
    return sorted(set(re.findall(r"\b(xp_[a-z0-9_]+)\s*\(", txt)))


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise XPointHipError(
            f"{LIB_PATH} not found: build it with `python -m xpoint_amd.build` (hipcc, gfx950). "
            "xpoint_amd has no CPU fallback.")
    # torch bundles its own libamdhip64: it must be the HIP runtime of this process, so import torch BEFORE
    # the dynamic loader resolves our library's libamdhip64.so dependency (otherwise two runtimes coexist and
    # every launch fails with "no ROCm-capable device").
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    lib.xp_last_error.restype = ctypes.c_char_p
    lib.xp_last_error.argtypes = []
    lib.xp_version.restype = c_i
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_i
    for name, (res, argtypes) in _SIZE_QUERIES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = res
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().xp_last_error().decode("utf-8", "replace")
        raise XPointHipError(f"{what} failed (rc={rc}): {msg}")


def call(name: str, *args):
    lib = load()
    check(getattr(lib, name)(*args), name)


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) float32/int32 torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "xpoint_amd ops need contiguous device tensors"
    return ctypes.c_void_p(t.data_ptr())


def current_stream(device=None):
    """The current HIP stream of `device` (a torch device / tensor; default: the current device).  Callers whose tensors
    may live on a device other than the current one pass it, or wrap the call in `torch.cuda.device(...)`."""
    import torch
    if device is not None and hasattr(device, "device"):
        device = device.device
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
