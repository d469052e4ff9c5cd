"""ctypes binding of include/oatgpu.h (one declaration per exported symbol)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    """The product library.  Only with OATGPU_MEASURE_PY=1 in the environment does OATGPU_LIB redirect the binding to
    another build (the A/B tools under tools/ set both): nothing else in the environment changes what is loaded."""
    if os.environ.get("OATGPU_MEASURE_PY") == "1" and os.environ.get("OATGPU_LIB"):
        return os.environ["OATGPU_LIB"]
    return os.path.join(_HERE, "lib", "liboatgpu.so")


class OatGpuError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"oatgpu error {code}: {text}")
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("device", C.c_int32), ("n_streams", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32),
        ("ring_depth", C.c_int32), ("channels", C.c_int32),
        ("history", C.c_int32), ("nmixtures", C.c_int32),
        ("var_threshold", C.c_float), ("background_ratio", C.c_float), ("var_threshold_gen", C.c_float),
        ("var_init", C.c_float), ("var_min", C.c_float), ("var_max", C.c_float), ("ct", C.c_float),
        ("tau", C.c_float), ("detect_shadows", C.c_int32), ("shadow_value", C.c_int32),
        ("h_lo", C.c_int32), ("h_hi", C.c_int32), ("s_lo", C.c_int32), ("s_hi", C.c_int32),
        ("v_lo", C.c_int32), ("v_hi", C.c_int32), ("erode", C.c_int32), ("dilate", C.c_int32),
        ("min_area", C.c_double), ("max_area", C.c_double),
        ("diff_threshold", C.c_int32), ("blur", C.c_int32),
        ("mog_restore_nmodes", C.c_int32), ("reserved_", C.c_int32),
    ]


class Position(C.Structure):
    _fields_ = [("valid", C.c_int32), ("first_pixel", C.c_int32), ("x", C.c_double), ("y", C.c_double),
                ("area", C.c_double), ("a00", C.c_int64), ("a10", C.c_int64), ("a01", C.c_int64),
                ("velocity_valid", C.c_int32), ("raw_valid", C.c_int32), ("vx", C.c_double), ("vy", C.c_double),
                ("raw_x", C.c_double), ("raw_y", C.c_double)]


class Profile(C.Structure):
    _fields_ = [("steps", C.c_int64), ("mog_ms", C.c_double), ("morph_ms", C.c_double),
                ("blob_ms", C.c_double), ("total_ms", C.c_double), ("event_pair_ms", C.c_double),
                ("mog_frames", C.c_int64), ("dropped", C.c_int64)]


class Traffic(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("launches", "pixels", "lane_bytes_read", "lane_bytes_written",
                                         "sector32_bytes_read", "sector32_bytes_written",
                                         "sector64_bytes_read", "sector64_bytes_written")]


E_RING_FULL = -4
E_RING_EMPTY = -5
ABI_VERSION = 9          # must equal OATGPU_ABI_VERSION of include/oatgpu.h
TAP_THRESHOLD, TAP_MORPH, TAP_FINAL = 0, 1, 2

_u8p = C.POINTER(C.c_uint8)
_fp = C.POINTER(C.c_float)
_ctx = C.c_void_p

# name -> (restype, argtypes); mirrors include/oatgpu.h line by line
SIGNATURES = {
    "oatgpu_abi_version": (C.c_int, []),
    "oatgpu_default_config": (C.c_int, [C.POINTER(Config)]),
    "oatgpu_create": (_ctx, [C.POINTER(Config)]),
    "oatgpu_destroy": (None, [_ctx]),
    "oatgpu_last_error": (C.c_char_p, [_ctx]),
    "oatgpu_device_count": (C.c_int, []),
    "oatgpu_device_open_retries": (C.c_int, []),
    "oatgpu_device_numa_node": (C.c_int, [C.c_int32]),
    "oatgpu_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "oatgpu_host_unregister": (C.c_int, [C.c_void_p]),
    "oatgpu_host_alloc": (C.c_void_p, [C.c_size_t]),
    "oatgpu_host_free": (None, [C.c_void_p]),
    "oatgpu_set_stream": (C.c_int, [_ctx, C.c_void_p]),
    "oatgpu_get_stream": (C.c_void_p, [_ctx]),
    "oatgpu_synchronize": (C.c_int, [_ctx]),
    "oatgpu_set_detector": (C.c_int, [_ctx] + [C.c_int32] * 8 + [C.c_double, C.c_double]),
    "oatgpu_set_kalman": (C.c_int, [_ctx, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double]),
    "oatgpu_set_roi_mask": (C.c_int, [_ctx, C.c_int32, _u8p]),
    "oatgpu_bsub_filter": (C.c_int, [_ctx, C.c_int32, _u8p, _u8p, C.c_double]),
    "oatgpu_bsub_set_background": (C.c_int, [_ctx, C.c_int32, _u8p]),
    "oatgpu_mask_filter": (C.c_int, [_ctx, C.c_int32, _u8p, _u8p]),
    "oatgpu_thresh_filter": (C.c_int, [_ctx, _u8p, _u8p, C.c_int32, C.c_int32]),
    "oatgpu_mog_apply": (C.c_int, [_ctx, C.c_int32, _u8p, _u8p, C.c_double]),
    "oatgpu_mog_filter": (C.c_int, [_ctx, C.c_int32, _u8p, _u8p, C.c_double]),
    "oatgpu_bgr2hsv": (C.c_int, [_ctx, _u8p, _u8p]),
    "oatgpu_cvt_color": (C.c_int, [_ctx, C.c_int32, C.c_int32, _u8p, _u8p]),
    "oatgpu_set_fusion": (C.c_int, [_ctx, C.c_int32]),
    "oatgpu_set_deferred": (C.c_int, [_ctx, C.c_int32]),
    "oatgpu_fetch_frame": (C.c_int, [_ctx, _u8p]),
    "oatgpu_fetch_position": (C.c_int, [_ctx, C.POINTER(Position)]),
    "oatgpu_set_early_blob": (C.c_int, [_ctx, C.c_int32]),
    "oatgpu_early_blob_timeouts": (C.c_int64, [_ctx]),
    "oatgpu_set_k1_workgroup": (C.c_int, [_ctx, C.c_int32]),
    "oatgpu_last_step_shape": (C.c_int, [_ctx, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "oatgpu_set_homography": (C.c_int, [_ctx, C.c_int32, C.POINTER(C.c_double)]),
    "oatgpu_detect_hsv": (C.c_int, [_ctx, C.c_int32, _u8p, C.POINTER(Position)]),
    "oatgpu_detect_thresh": (C.c_int, [_ctx, C.c_int32, _u8p, C.POINTER(Position)]),
    "oatgpu_detect_diff": (C.c_int, [_ctx, C.c_int32, _u8p, C.POINTER(Position)]),
    "oatgpu_track_batch": (C.c_int, [_ctx, C.POINTER(_u8p), C.c_int32, C.c_double, C.POINTER(Position)]),
    "oatgpu_track_batch_dev": (C.c_int, [_ctx, C.c_void_p, C.c_double, C.POINTER(Position)]),
    "oatgpu_track_enqueue": (C.c_int, [_ctx, C.POINTER(_u8p), C.c_int32, C.c_double]),
    "oatgpu_track_enqueue_dev": (C.c_int, [_ctx, C.c_void_p, C.c_double]),
    "oatgpu_track_sequence_dev": (C.c_int, [_ctx, C.POINTER(C.c_void_p), C.c_int32, C.c_double, C.POINTER(Position)]),
    "oatgpu_track_sequence_dev_timed": (C.c_int, [_ctx, C.POINTER(C.c_void_p), C.c_int32, C.c_double, C.POINTER(Position),
                                                  C.POINTER(C.c_double)]),
    "oatgpu_track_sequence_dev_latency": (C.c_int, [_ctx, C.POINTER(C.c_void_p), C.c_int32, C.c_double, C.POINTER(Position),
                                                    C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "oatgpu_track_collect": (C.c_int, [_ctx, C.POINTER(Position)]),
    "oatgpu_track_outstanding": (C.c_int, [_ctx]),
    "oatgpu_track_input_consumed": (C.c_int, [_ctx]),
    "oatgpu_track_input_consumed_stream": (C.c_int, [_ctx, C.c_int32]),
    "oatgpu_track_stage": (C.c_int, [_ctx, C.c_int32, _u8p]),
    "oatgpu_track_enqueue_staged": (C.c_int, [_ctx, C.c_double]),
    "oatgpu_track_stage_abort": (C.c_int, [_ctx]),
    "oatgpu_set_stage_copy": (C.c_int, [_ctx, C.c_int32]),
    "oatgpu_track_ready": (C.c_int, [_ctx]),
    "oatgpu_read_mask": (C.c_int, [_ctx, C.c_int32, C.c_int32, _u8p]),
    "oatgpu_mog_get_state": (C.c_int, [_ctx, C.c_int32, _u8p, _fp, _fp, _fp, C.POINTER(C.c_int32)]),
    "oatgpu_mog_set_state": (C.c_int, [_ctx, C.c_int32, _u8p, _fp, _fp, _fp, C.c_int32]),
    "oatgpu_mog_save": (C.c_int, [_ctx, C.c_int32, C.c_char_p]),
    "oatgpu_mog_load": (C.c_int, [_ctx, C.c_int32, C.c_char_p]),
    "oatgpu_profile_enable": (C.c_int, [_ctx, C.c_int32]),
    "oatgpu_profile_read": (C.c_int, [_ctx, C.POINTER(Profile)]),
    "oatgpu_profile_reset": (C.c_int, [_ctx]),
    "oatgpu_traffic_audit": (C.c_int, [_ctx, C.c_int32]),
    "oatgpu_traffic_read": (C.c_int, [_ctx, C.POINTER(Traffic)]),
    "oatgpu_measure_hbm": (C.c_int, [_ctx, C.c_size_t, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib = None


def load():
    """Loads liboatgpu.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7.  If torch is
    # imported AFTER this library has pulled in /opt/rocm's runtime, torch finds no GPU; imported
    # first, both share torch's copy (same SONAME).  So when torch is installed, load it first.
    import sys
    if "torch" not in sys.modules and os.environ.get("OATGPU_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: the HIP extension has not been built "
            "(run `make` or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.oatgpu_abi_version() != ABI_VERSION:
        raise ImportError("liboatgpu.so ABI version mismatch")
    _lib = lib
    return lib


def check(lib, ctx, rc):
    if rc != 0:
        msg = lib.oatgpu_last_error(ctx)
        raise OatGpuError(rc, msg.decode() if msg else "")


def u8(a):
    return a.ctypes.data_as(_u8p)


def f32(a):
    return a.ctypes.data_as(_fp)
