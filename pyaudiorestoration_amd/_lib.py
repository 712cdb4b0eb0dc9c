"""ctypes binding of libpar_hip.so (the C ABI declared in include/par_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call
fails, an exception is raised (inside the reference's own backend chain,
util/fourier.py:67-75, that exception is what makes it try the next backend).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PAR_HIP_LIB: developer override to A/B two builds of the library inside one GPU session
LIB_PATH = os.environ.get("PAR_HIP_LIB") or os.path.join(_HERE, "libpar_hip.so")

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_vp = ctypes.c_void_p
c_dbl = ctypes.c_double
c_sz = ctypes.c_size_t
c_float = ctypes.c_float
c_u64 = ctypes.c_uint64

class FusedItem(ctypes.Structure):
    """par_fused_item of include/par_hip.h (one planned file of par_varispeed_fused_batch_f32)"""
    _fields_ = [("speeds", c_vp), ("m", c_i64), ("work", c_vp), ("aux", c_vp), ("max_out", c_i64), ("len_out", c_i64),
                ("sig0", c_vp), ("sig1", c_vp), ("sig_stride", c_i64), ("len_in", c_i64), ("out0", c_vp), ("out1", c_vp),
                ("out_stride", c_i64)]


# name -> (restype, argtypes); mirrors include/par_hip.h one to one
SIGNATURES = {
    "par_version": (c_int, []),
    "par_device_count": (c_int, []),
    "par_last_error": (c_int, [ctypes.c_char_p, c_int]),
    "par_last_plan_flags": (c_int, []),
    "par_event_create": (c_int, [ctypes.POINTER(c_vp)]),
    "par_event_destroy": (c_int, [c_vp]),
    "par_event_record": (c_int, [c_vp, c_vp]),
    "par_event_elapsed_ms": (c_int, [c_vp, c_vp, ctypes.POINTER(ctypes.c_float)]),
    "par_stream_sync": (c_int, [c_int, c_vp]),
    "par_stream_create": (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_vp)]),
    "par_stream_destroy": (c_int, [c_vp]),
    "par_stft_frames": (c_i64, [c_i64, c_int, c_int]),
    "par_stft_f32": (c_int, [c_int, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_int, c_i64, c_vp]),
    "par_istft_scratch_floats": (c_i64, [c_i64, c_int, c_int]),
    "par_istft_f32": (c_int, [c_int, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "par_spec_apply_gain_db_c64": (c_int, [c_int, c_vp, c_vp, c_i64, c_vp]),
    "par_copy_segments_f32": (c_int, [c_int, c_vp, c_i64, c_i64, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp]),
    "par_inpaint_gain_db_c64": (c_int, [c_int, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "par_spec_apply_gain_boxes_c64": (c_int, [c_int, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "par_flac_info": (c_int, [c_vp, c_sz, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                              ctypes.POINTER(c_i64), c_vp]),
    "par_flac_decode_f32": (c_int, [c_vp, c_sz, c_vp, c_i64, c_int, c_int, ctypes.POINTER(c_i64)]),
    "par_zero_crossings_work_len": (c_i64, [c_i64]),
    "par_zero_crossings_f64": (c_int, [c_int, c_vp, c_i64, c_vp, c_vp, c_i64, ctypes.POINTER(c_i64), c_vp]),
    "par_band_mean_db_f32": (c_int, [c_int, c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_i64, c_i64, c_vp, c_vp]),
    "par_lag_to_pos_f64": (c_int, [c_int, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, ctypes.POINTER(c_i64),
                                   ctypes.POINTER(c_int), c_vp]),
    "par_speed_plan_bytes": (c_sz, [c_i64]),
    "par_speed_to_pos_plan": (c_int, [c_int, c_vp, c_vp, c_i64, c_i64, c_vp, c_sz, ctypes.POINTER(c_i64),
                                      ctypes.POINTER(c_int), c_vp]),
    "par_speed_to_pos_plan_ex": (c_int, [c_int, c_vp, c_vp, c_i64, c_i64, c_vp, c_sz, ctypes.POINTER(c_i64),
                                         ctypes.POINTER(c_int), c_int, ctypes.POINTER(c_int), c_vp]),
    "par_speed_to_pos_fill": (c_int, [c_int, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp]),
    "par_speed_to_pos_fill_fused": (c_int, [c_int, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "par_sinc_resample_f32": (c_int, [c_int, c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_vp, c_i64, c_vp]),
    "par_varispeed_resample_f32": (c_int, [c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_int, c_vp, c_i64,
                                           c_int, c_vp]),
    "par_fused_aux_bytes": (c_sz, [c_i64, c_i64]),
    "par_speed_to_pos_plan_fused": (c_int, [c_int, c_vp, c_vp, c_i64, c_i64, c_vp, c_sz, c_vp, c_sz, c_i64,
                                            ctypes.POINTER(c_i64), ctypes.POINTER(c_int), c_int, ctypes.POINTER(c_int),
                                            ctypes.POINTER(c_int), c_vp]),
    "par_debug_sinc_kernel": (c_int, [c_int]),
    "par_fused_redo_tiles": (c_int, [c_int, c_vp, c_i64, c_i64, ctypes.POINTER(c_int), c_vp]),
    "par_fused_redo_list": (c_int, [c_int, c_vp, c_i64, c_i64, ctypes.POINTER(c_int), c_int, ctypes.POINTER(c_int), c_vp]),
    "par_varispeed_fused_f32": (c_int, [c_int, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_int, c_vp, c_i64,
                                        c_vp]),
    "par_varispeed_fused_batch_f32": (c_int, [c_int, c_int, ctypes.POINTER(FusedItem), c_int, c_vp]),
    "par_varispeed_fused_stereo_f32": (c_int, [c_int, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_i64, c_i64, c_int,
                                               c_vp, c_vp, c_i64, c_vp]),
    "par_profile_enable": (c_int, [c_int, c_int]),
    "par_profile_read": (c_int, [c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(c_int), ctypes.POINTER(c_i64)]),
    "par_linear_resample_f32": (c_int, [c_int, c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp]),
    "par_synth_signal_f32": (c_int, [c_int, c_vp, c_i64, c_i64, c_dbl, c_u64, c_vp]),
    "par_synth_speed_curve_f64": (c_int, [c_int, c_vp, c_vp, c_i64, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl, c_vp]),
    "par_track_peak_f64": (c_int, [c_int, c_vp, c_i64, c_int, c_i64, c_i64, c_i64, c_vp, c_int, c_dbl, c_dbl, c_int, c_vp, c_vp]),
    "par_track_peak_refined_f64": (c_int, [c_int, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_vp, c_int, c_i64, c_i64, c_i64, c_vp,
                                           c_dbl, c_dbl, c_int, c_vp, c_vp]),
    "par_sosfiltfilt_work_len": (c_i64, [c_i64, c_i64]),
    "par_sosfiltfilt_f64": (c_int, [c_int, c_vp, c_vp, c_int, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "par_curve_scale_f64": (c_int, [c_int, c_vp, c_i64, c_int, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "par_accumulate_f64_f32": (c_int, [c_int, c_vp, c_i64, c_int, c_i64, c_vp, c_vp]),
    "par_sosfiltfilt_batch_work_len": (c_i64, [c_i64, c_i64, c_int, c_int]),
    "par_sosfiltfilt_batch_f64": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_vp, c_i64, c_int, c_i64, c_i64, c_vp, c_i64, c_vp,
                                          c_i64, c_vp]),
    "par_track_cog_f64": (c_int, [c_int, c_vp, c_i64, c_int, c_i64, c_i64, c_i64, c_vp, c_int, c_dbl, c_dbl, c_vp, c_vp]),
    "par_stft_big_scratch_bytes": (ctypes.c_size_t, [c_i64, c_int, c_int, c_int]),
    "par_stft_big_f32": (c_int, [c_int, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_int, c_vp, ctypes.c_size_t, c_vp]),
    "par_xcorr_scratch_bytes": (ctypes.c_size_t, [c_i64, c_i64]),
    "par_xcorr_f64": (c_int, [c_int, c_vp, c_i64, c_vp, c_i64, c_vp, ctypes.c_size_t, c_vp, c_vp]),
    "par_find_delay_f64": (c_int, [c_int, c_vp, c_i64, c_vp, c_i64, c_int, c_vp, ctypes.c_size_t, c_vp, c_vp, c_vp]),
    "par_piptrack_f32": (c_int, [c_int, c_vp, c_i64, c_int, c_i64, c_float, c_float, c_int, c_dbl, c_dbl, c_dbl, c_float, c_vp, c_vp, c_vp]),
    "par_track_corr_work_len": (c_i64, [c_i64, c_int]),
    "par_track_corr_f64": (c_int, [c_int, c_vp, c_i64, c_int, c_i64, c_int, c_int, c_i64, c_vp, c_vp, c_int, c_dbl, c_dbl, c_vp, c_vp,
                                   c_vp, c_vp]),
}

_lib = None


class ParError(RuntimeError):
    """A libpar_hip call returned a non-zero status."""

    def __init__(self, code, msg):
        super().__init__(f"libpar_hip status {code}: {msg}")
        self.code = code


class ParUnsupported(ParError):
    """Valid for the reference but not implemented by the HIP path (status PAR_ERR_UNSUPPORTED)."""


class ParIndexError(ParError, IndexError):
    """The reference indexes past the end of an array at this point (status PAR_ERR_INDEX)."""


class ParShapeError(ParError, ValueError):
    """The reference multiplies arrays of different lengths at this point (status PAR_ERR_SHAPE)."""


class ParEmptyBand(ParError, ValueError):
    """A tracker band is empty (status PAR_ERR_EMPTY_BAND); also a ValueError, which is what the reference's
    argmax of an empty slice raises."""


def lib():
    """Load the shared library once.  Raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        # PyTorch-ROCm bundles its own libamdhip64 (soname libamdhip64.so.7, the same soname this library
        # needs).  Importing torch FIRST makes the dynamic loader hand that one runtime to both; loading
        # ours first would pull in /opt/rocm's copy as well and the second runtime then sees no device.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("PAR_HIP_LIB") and not hasattr(L, name):
                continue                   # developer A/B against an older build: newer entry points are simply absent
            fn = getattr(L, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error():
    buf = ctypes.create_string_buffer(512)
    lib().par_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(rc):
    if rc != 0:
        msg = last_error()
        raise {3: ParUnsupported, 5: ParEmptyBand, 6: ParIndexError, 7: ParShapeError}.get(rc, ParError)(rc, msg)
