"""ctypes binding of libaffnet_hip.so (include/affnet_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol cannot be
resolved, importing the binding raises.  Build it with `python -c "import __graft_entry__ as g;
g.build()"` or `bash affnet_amd/csrc/build.sh`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AFFNET_HIP_LIB: the tuning tools under tools/ point this at libaffnet_hip_probes.so (the same sources + the probe kernels of
# include/affnet_hip_probes.h, `AFFNET_PROBES=1 bash affnet_amd/csrc/build.sh`); everything else loads the product library
LIB_PATH = os.environ.get("AFFNET_HIP_LIB") or os.path.join(_HERE, "libaffnet_hip.so")
PROBES_LIB_PATH = os.path.join(_HERE, "libaffnet_hip_probes.so")

MAX_OCTAVES, MAX_LEVELS, MAX_TAPS = 16, 8, 37
NET_AFFNET, NET_ORINET, NET_HARDNET, NET_AFFNET_FULLCONV = 0, 1, 2, 3
ARITH_FP32_MFMA, ARITH_FP32_SPLIT3, ARITH_FP32_SPLIT2H = 0, 1, 2        # include/affnet_hip.h AFFNET_ARITH_*
ARITH_NAMES = {"fp32": ARITH_FP32_MFMA, "fp32_mfma": ARITH_FP32_MFMA, "fp32_split3": ARITH_FP32_SPLIT3, "split3": ARITH_FP32_SPLIT3,
               "fp32_split2h": ARITH_FP32_SPLIT2H, "split2h": ARITH_FP32_SPLIT2H}


def arith_code(arith):
    """'fp32' (default: exact fp32 MFMA) / 'fp32_split3' (fp32 = 3 x bf16 split operands, six products) / 'fp32_split2h' (fp32 = 2 x fp16 split
    operands, three products) or an AFFNET_ARITH_* integer -> the integer."""
    if arith is None:
        return ARITH_FP32_MFMA
    if isinstance(arith, str):
        if arith.lower() not in ARITH_NAMES:
            raise ValueError("arith must be one of %s" % sorted(ARITH_NAMES))
        return ARITH_NAMES[arith.lower()]
    if int(arith) not in (ARITH_FP32_MFMA, ARITH_FP32_SPLIT3, ARITH_FP32_SPLIT2H):
        raise ValueError("arith must be AFFNET_ARITH_FP32_MFMA (0), AFFNET_ARITH_FP32_SPLIT3 (1) or AFFNET_ARITH_FP32_SPLIT2H (2)")
    return int(arith)
OK, ERR_INVALID, ERR_HIP, ERR_CAPACITY, ERR_EMPTY = 0, -1, -2, -3, -4


class Config(C.Structure):
    _fields_ = [
        ("height", C.c_int32), ("width", C.c_int32), ("n_octaves", C.c_int32), ("levels_per_octave", C.c_int32),
        ("oct_h", C.c_int32 * MAX_OCTAVES), ("oct_w", C.c_int32 * MAX_OCTAVES),
        ("level_sigma", (C.c_float * MAX_LEVELS) * MAX_OCTAVES),
        ("level_sigma4", (C.c_float * MAX_LEVELS) * MAX_OCTAVES),
        ("level_sigma_px", (C.c_double * MAX_LEVELS) * MAX_OCTAVES),
        ("first_blur_taps", C.c_int32), ("first_blur", C.c_float * (MAX_TAPS * MAX_TAPS)),
        ("level_blur_taps", C.c_int32 * MAX_LEVELS),
        ("level_blur", (C.c_float * (MAX_TAPS * MAX_TAPS)) * MAX_LEVELS),
        ("level_blur0_taps", C.c_int32 * MAX_LEVELS),
        ("level_blur0", (C.c_float * (MAX_TAPS * MAX_TAPS)) * MAX_LEVELS),
        ("lazy_shape_rows", C.c_int32),
        ("onepass", C.c_int32),
        ("mr_size", C.c_float), ("threshold", C.c_float),
        ("num_features", C.c_int32), ("num_prefilter", C.c_int32),
        ("max_raw_per_octave_div", C.c_int32), ("max_keep", C.c_int32), ("batch", C.c_int32), ("baum_iters", C.c_int32),
        ("arith", C.c_int32),
    ]


class Nets(C.Structure):
    _fields_ = [("d_affnet", C.c_void_p), ("d_orinet", C.c_void_p), ("d_hardnet", C.c_void_p),
                ("h_orientation_window", C.c_void_p), ("h_baumberg_window", C.c_void_p)]


# name -> (restype, argtypes); every symbol declared in include/affnet_hip.h
_P, _I, _SZ = C.c_void_p, C.c_int, C.c_size_t
SYMBOLS = {
    "affnet_config_fill": (_I, [C.POINTER(Config), _I, _I, _I, C.c_double, _I, C.c_double, C.c_double, _I, _I, _I, _I]),
    "affnet_ctx_create": (_I, [C.POINTER(_P), _I, C.POINTER(Config)]),
    "affnet_ctx_destroy": (None, [_P]),
    "affnet_last_error": (C.c_char_p, [_P]),
    "affnet_version": (C.c_char_p, []),
    "affnet_set_arith": (_I, [_P, _I]),
    "affnet_get_arith": (_I, [_P]),
    "affnet_workspace_bytes": (_SZ, [_P]),
    "affnet_bind_workspace": (_I, [_P, _P, _SZ]),
    "affnet_pyramid_level_offset": (C.c_int64, [_P, _I, _I]),
    "affnet_pyramid_image_stride": (C.c_int64, [_P]),
    "affnet_batch": (_I, [_P]),
    "affnet_capacity_prefilter": (_I, [_P]),
    "affnet_capacity_final": (_I, [_P]),
    "affnet_gauss_blur": (_I, [_P, _P, _P, _I, _I, C.POINTER(C.c_float), _I, _P]),
    "affnet_pyramid_build": (_I, [_P, _P, _P]),
    "affnet_hessian_response": (_I, [_P, _P, _P, _I, _I, C.c_float, _P]),
    "affnet_detect": (_I, [_P, _P, _P, _P, _P, _P]),
    "affnet_detect_responses": (_I, [_P, _P, _P, _P, _P, _P, _P]),
    "affnet_laf_grid_sample": (_I, [_P, _P, _I, _I, _P, _I, _I, _P, _P]),
    "affnet_pyr_grid_sample": (_I, [_P, _P, _P, _P, _I, _I, _P, _P]),
    "affnet_cnn32_packed_floats": (_SZ, [_I]),
    "affnet_cnn32_pack_weights": (_I, [_I, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), _P, _P, _P, _P, _P]),
    "affnet_cnn32_forward": (_I, [_P, _I, _P, _P, _P, _I, _P, _P, _P]),
    "affnet_cnn32_forward_pyr": (_I, [_P, _I, _P, _P, _P, _P, _I, _P, _P, _P]),
    "affnet_local_norm": (_I, [_P, _P, _P, _I, _I, _P]),
    "affnet_fullconv_scratch_bytes": (_SZ, [_I, _I]),
    "affnet_fullconv_forward": (_I, [_P, _P, _P, _I, _I, _P, _P, _P]),
    "affnet_nms2d": (_I, [_P, _P, _P, _I, _I, C.c_float, _P]),
    "affnet_shape_filter_select": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "affnet_apply_rotation": (_I, [_P, _P, _P, _P, _I, _P]),
    "affnet_scale_lafs": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "affnet_lafs_to_ellipses": (_I, [_P, _P, _P, _I, _P, _P]),
    "affnet_level_select": (_I, [_P, _P, _P, _I, _I, _P, _P, _P]),
    "affnet_handcrafted_forward": (_I, [_P, _I, _P, _I, _P, _P, _P, _P]),
    "affnet_handcrafted_forward_pyr": (_I, [_P, _I, _P, _P, _P, _I, _P, _P, _P]),
    "affnet_match_scratch_bytes": (_SZ, [_I, _I]),
    "affnet_distance_matrix": (_I, [_P, _P, _I, _P, _I, _I, _P, _P, _P]),
    "affnet_match_snn": (_I, [_P, _P, _I, _P, _I, _I, C.c_float, _P, _P, _P, _P, _P, _P, _P]),
    "affnet_reproject_lafs": (_I, [_P, _P, _I, C.POINTER(C.c_float), _P, _P]),
    "affnet_centre_nn": (_I, [_P, _P, _I, _P, _I, _P, _P, _P]),
    "affnet_extract_features": (_I, [_P, C.POINTER(Nets), _P, _I, _P, _P, _P, _P, _P, _P]),
    "affnet_detect_image": (_I, [_P, _P, _P]),
    "affnet_detect_image_responses": (_I, [_P, _P, _P]),
    "affnet_detect_image_onepass": (_I, [_P, _P, _P, _P]),
    "affnet_detect_image_onepass_responses": (_I, [_P, _P, _P, _P]),
    "affnet_detected_list": (_I, [_P, _P, _P, _P, _P, _P]),
    "affnet_shape_iterate": (_I, [_P, _P, _P, _P, _P, _I, _P, _P]),
    "affnet_affmap_offset": (C.c_int64, [_P, _I]),
    "affnet_affmap_image_stride": (C.c_int64, [_P]),
    "affnet_describe_detected": (_I, [_P, C.POINTER(Nets), _I, _P, _P, _P, _P, _P, _P]),
    "affnet_graph_capture_extract": (_I, [_P, C.POINTER(Nets), _P, _I, _P, _P, _P, _P, _P, _P]),
    "affnet_graph_launch": (_I, [_P, _P]),
    "affnet_profile_enable": (_I, [_P, _I]),
    "affnet_profile_read": (_I, [_P, C.POINTER(C.c_double * 8), C.POINTER(C.c_int32)]),
    "affnet_read_counts": (_I, [_P, C.POINTER(C.c_int32 * 4), _P]),
    "affnet_counter_offset": (C.c_int64, [_P, _I]),
    "affnet_counter_stride": (C.c_int64, [_P]),
    "affnet_host_base_grid": (_I, [_I, C.POINTER(C.c_float)]),
}

# include/affnet_hip_debug.h: parity / tuning aids, not part of the drop-in boundary
DEBUG_SYMBOLS = {
    "affnet_cnn32_debug_timing": (_I, [_P, _P]),
    "affnet_cnn32_debug_layer": (_I, [_P, _I, _P, _P, _I, _P, _P]),
    "affnet_selftest_mfma": (_I, [_P, _P, _P, _P]),
    "affnet_debug_split3_variant": (_I, [_P, _I]),
}

# include/affnet_hip_probes.h: probe kernels of the tuning tools - only in libaffnet_hip_probes.so (bound when present)
PROBE_SYMBOLS = {
    "affnet_cnn32_probe": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "affnet_split3_gemm": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "affnet_split3_rate": (_I, [_I, _I, _I, _P, _P]),
    "affnet_debug_stream": (_I, [_P, _P, _SZ, _I, _I, _I, _P]),
}


class AffnetHipError(RuntimeError):
    pass


class AffnetEmptyError(AffnetHipError):
    """AFFNET_ERR_EMPTY: no image of the call produced a detection (the reference raises in torch.cat([]),
    SparseImgRepresenter.py:100)."""


def _load():
    if not os.path.isfile(LIB_PATH):
        if os.path.basename(LIB_PATH) == os.path.basename(PROBES_LIB_PATH):          # a tuning tool asked for the probe build
            raise ImportError("affnet_amd: %s is missing - the tools under tools/ need the probe build of the library: "
                              "AFFNET_PROBES=1 bash affnet_amd/csrc/build.sh" % LIB_PATH)
        raise ImportError(
            "affnet_amd: %s is missing - the HIP extension is the only implementation of this path "
            "(no CPU fallback). Build it: bash affnet_amd/csrc/build.sh" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for table in (SYMBOLS, DEBUG_SYMBOLS):
        for name, (res, args) in table.items():
            fn = getattr(lib, name)  # raises AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
    for name, (res, args) in PROBE_SYMBOLS.items():
        if hasattr(lib, name):       # libaffnet_hip_probes.so
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


def check(rc, ctx=None, what=""):
    if rc != OK:
        msg = lib.affnet_last_error(ctx).decode() if ctx else ""
        if rc == ERR_EMPTY:
            raise AffnetEmptyError("%s: %s" % (what or "libaffnet_hip call", msg))
        raise AffnetHipError("%s failed (code %d): %s" % (what or "libaffnet_hip call", rc, msg))


def ptr(t):
    """Device (or host) pointer of a torch tensor as c_void_p; None -> NULL."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream(device):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
