"""ctypes binding of the C ABI in include/ptq4vit_hip.h (libptq4vit_hip.so, built in-tree).

The library is the product: if it cannot be loaded this module raises -- there is no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("P4V_LIB") or os.path.join(_HERE, "csrc", "libptq4vit_hip.so")   # P4V_LIB: tuning builds only

METRICS = {
    "L1_norm": 0,
    "L2_norm": 1,
    "linear_weighted_L2_norm": 2,
    "square_weighted_L2_norm": 3,
    "hessian": 4,
    "cosine": 5,
}


class LinearDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "batch", "tokens", "in_features", "out_features", "n_V", "n_H", "n_a", "w_bit", "a_bit", "metric",
        "eq_n", "search_round", "twin_postgelu", "init_layerwise", "has_bias", "reserved")]


class MatMulDesc(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("batch", "heads", "M", "K", "N")] + [("_pad0", C.c_int32)] +
                [("a_stride", C.c_int64 * 4), ("b_stride", C.c_int64 * 4)] +
                [(n, C.c_int32) for n in ("A_bit", "B_bit", "metric", "eq_n", "search_round", "sos",
                                          "init_layerwise", "reserved")])


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "batch", "in_channels", "height", "width", "out_channels", "kernel_h", "kernel_w", "stride_h", "stride_w",
        "pad_h", "pad_w", "dil_h", "dil_w", "w_bit", "a_bit", "metric", "eq_n", "search_round", "channelwise",
        "init_layerwise", "has_bias", "reserved")]


class KernelStats(C.Structure):
    _fields_ = [("sweep_i8_ms", C.c_double), ("sweep_i8_launches", C.c_int64), ("sweep_i8_macs", C.c_double),
                ("sweep_f32_ms", C.c_double), ("sweep_f32_launches", C.c_int64), ("sweep_f32_macs", C.c_double),
                ("sweep_i8_alg_macs", C.c_double), ("sweep_f32_alg_macs", C.c_double),
                ("sweep6_ms", C.c_double), ("sweep6_launches", C.c_int64), ("sweep6_macs", C.c_double), ("sweep6_alg_macs", C.c_double),
                ("memo_hits", C.c_int64), ("memo_misses", C.c_int64),
                ("sweep7_ms", C.c_double), ("sweep7_launches", C.c_int64), ("sweep7_macs", C.c_double), ("sweep7_alg_macs", C.c_double),
                ("sweep7_twin_ms", C.c_double), ("sweep7_twin_launches", C.c_int64), ("event_overhead_ms", C.c_double)]


class LaunchRecord(C.Structure):
    _fields_ = [("kind", C.c_int32), ("stage", C.c_int32), ("grid_x", C.c_int32), ("grid_z", C.c_int32),
                ("ms", C.c_double), ("ops", C.c_double), ("alg_ops", C.c_double), ("alg_bytes", C.c_double)]


LAUNCH_KINDS = {0: "k_sweep<int8>", 1: "k_sweep<float>", 2: "k_sweep6", 3: "k_sweep7", 4: "k_sweep7 (twin)", 5: "k_sweep4/5", 6: "k_sweep9",
                7: "k_sweep8", 8: "k_sweep2g", 9: "k_sweep2", 11: "k_sos_split", 12: "k_bound", 13: "k_slice_b", 14: "k_slice_a"}
LAUNCH_STAGES = {0: "full", 1: "A", 2: "B1", 3: "B2", 4: "A2"}


class PlaneDesc(C.Structure):
    _fields_ = [("rows", C.c_int64), ("cols", C.c_int64), ("cols_padded", C.c_int64), ("rows_per_scale", C.c_int64),
                ("mode", C.c_int32), ("lo", C.c_int32), ("hi", C.c_int32), ("qmax", C.c_int32),
                ("const_scale", C.c_float), ("reserved", C.c_int32)]


class ExportDesc(C.Structure):
    _fields_ = [("dims", C.c_int32 * 4), ("src_stride", C.c_int64 * 4),
                ("scale1_stride", C.c_int64 * 4), ("scale1_div", C.c_int32 * 4),
                ("scale2_stride", C.c_int64 * 4), ("scale2_div", C.c_int32 * 4),
                ("scale2_const", C.c_float),
                ("mode", C.c_int32), ("lo1", C.c_int32), ("hi1", C.c_int32), ("lo2", C.c_int32), ("hi2", C.c_int32),
                ("qmax", C.c_int32), ("reserved", C.c_int32)]


class GroupJob(C.Structure):
    """p4v_group_job: one member of a p4v_calibrate_group call (= one p4v_*_calibrate call)."""
    _fields_ = [("kind", C.c_int32), ("status", C.c_int32), ("desc", C.c_void_p), ("inp", C.c_void_p * 5), ("mult", C.c_void_p),
                ("out", C.c_void_p * 3), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


JOB_LINEAR, JOB_MATMUL, JOB_CONV = 0, 1, 2
PLANE_SYM, PLANE_SOS_HI, PLANE_SOS_LO, PLANE_TWIN = 1, 2, 3, 4
EXPORT_SYM_I8, EXPORT_SYM_F32, EXPORT_GELU_U8, EXPORT_SOS_U8 = 0, 1, 2, 3


EXPORTS = [
    "p4v_version", "p4v_last_error",
    "p4v_linear_workspace_bytes", "p4v_linear_calibrate",
    "p4v_matmul_workspace_bytes", "p4v_matmul_calibrate",
    "p4v_conv_workspace_bytes", "p4v_conv_calibrate",
    "p4v_calibrate_group", "p4v_launch_counters",
    "p4v_linear_quant_forward", "p4v_matmul_quant_forward",
    "p4v_amax_init_linear", "p4v_linear_search_w", "p4v_linear_search_a",
    "p4v_amax_init_matmul", "p4v_matmul_search_A", "p4v_sos_search_split", "p4v_matmul_search_B",
    "p4v_amax_init_conv", "p4v_conv_search_w_channelwise", "p4v_conv_search_w_layerwise", "p4v_conv_search_a",
    "p4v_score_argmax_gather",
    "p4v_quantize_i8", "p4v_pack_plane_i8", "p4v_fake_quant", "p4v_export_quantize", "p4v_multi_copy",
    "p4v_stats_enable", "p4v_stats_reset", "p4v_stats_get", "p4v_stats_launches", "p4v_prune_counters",
    "p4v_debug_set_variant", "p4v_debug_set_tuning", "p4v_debug_topk_rows",
]

_lib = None


def load():
    """Load libptq4vit_hip.so (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"ptq4vit_amd: HIP extension {LIB_PATH} is missing -- build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the calibration path.")
    # torch ships its own libamdhip64 (same soname as /opt/rocm's).  Import torch FIRST so that this library
    # binds to the HIP runtime torch already initialised: two runtimes in one process cannot both own the GPU.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, fp, ip = C.c_void_p, C.c_void_p, C.c_void_p
    lib.p4v_version.restype = C.c_int
    lib.p4v_last_error.restype = C.c_char_p
    lib.p4v_linear_workspace_bytes.restype = C.c_size_t
    lib.p4v_linear_workspace_bytes.argtypes = [C.POINTER(LinearDesc)]
    lib.p4v_linear_calibrate.restype = C.c_int
    lib.p4v_linear_calibrate.argtypes = [C.POINTER(LinearDesc), fp, fp, fp, fp, fp, fp, fp, fp, fp, ip, vp, C.c_size_t, vp]
    lib.p4v_matmul_workspace_bytes.restype = C.c_size_t
    lib.p4v_matmul_workspace_bytes.argtypes = [C.POINTER(MatMulDesc)]
    lib.p4v_matmul_calibrate.restype = C.c_int
    lib.p4v_matmul_calibrate.argtypes = [C.POINTER(MatMulDesc), fp, fp, fp, fp, fp, fp, fp, fp, fp, ip, vp, C.c_size_t, vp]
    lib.p4v_conv_workspace_bytes.restype = C.c_size_t
    lib.p4v_conv_workspace_bytes.argtypes = [C.POINTER(ConvDesc)]
    lib.p4v_conv_calibrate.restype = C.c_int
    lib.p4v_conv_calibrate.argtypes = [C.POINTER(ConvDesc), fp, fp, fp, fp, fp, fp, fp, fp, fp, ip, vp, C.c_size_t, vp]
    lib.p4v_linear_quant_forward.restype = C.c_int
    lib.p4v_linear_quant_forward.argtypes = [C.POINTER(LinearDesc), fp, fp, fp, fp, fp, fp, vp, C.c_size_t, vp]
    lib.p4v_matmul_quant_forward.restype = C.c_int
    lib.p4v_matmul_quant_forward.argtypes = [C.POINTER(MatMulDesc), fp, fp, fp, fp, fp, fp, vp, C.c_size_t, vp]
    # granular entry points: (desc, tensors..., workspace, bytes, stream)
    LD, MD, CD = C.POINTER(LinearDesc), C.POINTER(MatMulDesc), C.POINTER(ConvDesc)
    tail = [vp, C.c_size_t, vp]
    for name, args in (
            ("p4v_amax_init_linear", [LD] + [fp] * 4),
            ("p4v_linear_search_w", [LD] + [fp] * 9 + [ip]),
            ("p4v_linear_search_a", [LD] + [fp] * 9 + [ip]),
            ("p4v_amax_init_matmul", [MD] + [fp] * 4),
            ("p4v_matmul_search_A", [MD] + [fp] * 8 + [ip]),
            ("p4v_sos_search_split", [MD] + [fp] * 7 + [ip]),
            ("p4v_matmul_search_B", [MD] + [fp] * 9 + [ip]),
            ("p4v_amax_init_conv", [CD] + [fp] * 4),
            ("p4v_conv_search_w_channelwise", [CD] + [fp] * 9 + [ip]),
            ("p4v_conv_search_w_layerwise", [CD] + [fp] * 9 + [ip]),
            ("p4v_conv_search_a", [CD] + [fp] * 9 + [ip])):
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = args + tail
    lib.p4v_score_argmax_gather.restype = C.c_int
    lib.p4v_score_argmax_gather.argtypes = [fp, C.c_int32, C.c_int32, fp, fp, ip, vp]
    lib.p4v_quantize_i8.restype = C.c_int
    lib.p4v_quantize_i8.argtypes = [fp, C.c_int64, C.c_int64, C.c_int64, fp, C.c_int64, C.c_int32, C.c_int32, vp, vp]
    lib.p4v_fake_quant.restype = C.c_int
    lib.p4v_fake_quant.argtypes = [fp, C.c_int64, C.c_int64, fp, C.c_int64, C.c_int32, C.c_int32, fp, vp]
    lib.p4v_pack_plane_i8.restype = C.c_int
    lib.p4v_pack_plane_i8.argtypes = [C.POINTER(PlaneDesc), fp, fp, vp, vp]
    lib.p4v_export_quantize.restype = C.c_int
    lib.p4v_export_quantize.argtypes = [C.POINTER(ExportDesc), fp, fp, fp, vp, vp]
    lib.p4v_multi_copy.restype = C.c_int
    lib.p4v_multi_copy.argtypes = [vp, C.c_int32, C.c_int64, C.c_int64, vp]
    lib.p4v_debug_topk_rows.restype = C.c_int
    lib.p4v_debug_topk_rows.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.p4v_debug_set_variant.restype = C.c_int
    lib.p4v_debug_set_variant.argtypes = [C.c_int, C.c_int]
    lib.p4v_debug_set_tuning.restype = C.c_int
    lib.p4v_debug_set_tuning.argtypes = [C.c_int, C.c_int]
    lib.p4v_stats_enable.restype = C.c_int
    lib.p4v_stats_enable.argtypes = [C.c_int]
    lib.p4v_stats_reset.restype = C.c_int
    lib.p4v_stats_get.restype = C.c_int
    lib.p4v_stats_get.argtypes = [C.POINTER(KernelStats)]
    lib.p4v_stats_launches.restype = C.c_int
    lib.p4v_stats_launches.argtypes = [C.POINTER(LaunchRecord), C.c_int64, C.POINTER(C.c_int64)]
    lib.p4v_prune_counters.restype = C.c_int
    lib.p4v_prune_counters.argtypes = [C.POINTER(C.c_int64), C.c_int]
    lib.p4v_launch_counters.restype = C.c_int
    lib.p4v_launch_counters.argtypes = [C.POINTER(C.c_int64), C.c_int]
    lib.p4v_calibrate_group.restype = C.c_int
    lib.p4v_calibrate_group.argtypes = [C.POINTER(GroupJob), C.c_int32, vp, vp]
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().p4v_last_error().decode("utf-8", "replace")
        if rc == -2:
            raise NotImplementedError(f"{what}: {msg}")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def source_hash():
    """Short hash of the engine's sources (kernels, host side of the C ABI, the header): profiles committed under profiles/ carry
    it, and bench.py only quotes a profile's PMC traffic when it was measured on THIS code."""
    import hashlib
    root = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha256()
    for rel in ("csrc/p4v_kernels.h", "csrc/p4v_api.hip", "../include/ptq4vit_hip.h"):
        with open(os.path.join(root, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]
