"""ctypes loader for libhisstools_amd.so (the C ABI declared in include/hisstools_amd.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C hisstools_library_amd/csrc``.
There is no CPU fallback: if the shared object is missing this module raises ImportError, and if no
GPU is usable every ``*_create`` returns NULL, which the wrappers turn into RuntimeError.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (HCV_LIBRARY_PATH: a diagnostic build of the same library — the sanitizer builds of tools/sanitize/run.sh — in place of the product)
LIB_PATH = os.environ.get("HCV_LIBRARY_PATH") or os.path.join(_HERE, "libhisstools_amd.so")

f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
vp = C.c_void_p
usz = C.c_size_t
uptr = C.c_size_t      # uintptr_t
iptr = C.c_ssize_t     # intptr_t
u32 = C.c_uint32


class StageStats(C.Structure):
    _fields_ = [("fft_size", u32), ("partitions", u32), ("num_ins", u32), ("num_outs", u32),
                ("mac_launches", C.c_uint64), ("mac_hops", C.c_uint64), ("mac_ms", C.c_double),
                ("ksplit", u32), ("out_tile", u32), ("mac_steady_launches", C.c_uint64), ("hop_tile", u32), ("launch_partitions", u32), ("fused_launches", C.c_uint64), ("fused_stood_down", C.c_uint64), ("host_pre_launches", C.c_uint64)]


class RtStats(C.Structure):          # hcv_rt_stats
    _fields_ = [("start_collisions", C.c_uint64), ("mailbox_runs", C.c_uint64), ("mailbox_ns_max", C.c_uint64), ("mailbox_ns_total", C.c_uint64),
                ("ctl_sections", C.c_uint64), ("start_waits", C.c_uint64), ("arena_misses", C.c_uint64)]


class FFTCall(C.Structure):          # hcv_fft_call
    _fields_ = [("op", C.c_int), ("precision", C.c_int), ("log2n", C.c_uint), ("batch", usz),
                ("src_a", vp), ("src_b", vp), ("dst_a", vp), ("dst_b", vp),
                ("src_stride", usz), ("dst_stride", usz), ("in_length", usz)]


class IRCall(C.Structure):           # hcv_ir_call
    _fields_ = [("op", C.c_int), ("precision", C.c_int), ("log2n", C.c_uint), ("batch", usz),
                ("src_re", vp), ("src_im", vp), ("dst_re", vp), ("dst_im", vp),
                ("src_stride", usz), ("dst_stride", usz), ("value", C.c_double), ("zero_center", C.c_int)]


class IRProductCall(C.Structure):    # hcv_ir_product_call
    _fields_ = [("op", C.c_int), ("precision", C.c_int), ("size", usz), ("batch", usz),
                ("a_re", vp), ("a_im", vp), ("b_re", vp), ("b_im", vp), ("dst_re", vp), ("dst_im", vp),
                ("a_stride", usz), ("b_stride", usz), ("dst_stride", usz), ("b_broadcast", C.c_int), ("scale", C.c_double)]


class AudioFileInfo(C.Structure):    # hcv_audiofile_info
    _fields_ = [("file_type", C.c_int), ("pcm_format", C.c_int), ("header_endianness", C.c_int), ("audio_endianness", C.c_int),
                ("sampling_rate", C.c_double), ("channels", C.c_uint), ("frames", C.c_uint), ("bit_depth", C.c_uint),
                ("error_flags", C.c_int)]


# name -> (restype, argtypes); must list every symbol include/hisstools_amd.h declares
SIGNATURES = {
    "hcv_version": (C.c_char_p, []),
    "hcv_device_count": (C.c_int, []),
    "hcv_set_default_device": (C.c_int, [C.c_int]),
    "hcv_ctl_reserve": (C.c_int, [C.c_int, C.c_size_t]),
    "hcv_ctl_reserved": (C.c_size_t, [C.c_int]),
    "hcv_order_check_violations": (C.c_longlong, []),
    "hcv_debug_native_backtrace_on_crash": (None, []),
    "hcv_get_default_device": (C.c_int, []),
    "hcv_last_error": (C.c_char_p, []),
    "hcv_rfft_f32": (C.c_int, [f32p, usz, usz, usz, C.c_uint, f32p, f32p]),
    "hcv_rifft_f32": (C.c_int, [f32p, f32p, usz, C.c_uint, f32p]),
    "hcv_partitioned_create": (vp, [uptr, uptr, uptr, uptr]),
    "hcv_partitioned_destroy": (None, [vp]),
    "hcv_partitioned_set_fft_size": (C.c_int, [vp, uptr]),
    "hcv_partitioned_set_length": (C.c_int, [vp, uptr]),
    "hcv_partitioned_set_offset": (None, [vp, uptr]),
    "hcv_partitioned_set_reset_offset": (None, [vp, iptr]),
    "hcv_partitioned_set": (C.c_int, [vp, f32p, uptr]),
    "hcv_partitioned_reset": (None, [vp]),
    "hcv_partitioned_process": (C.c_int, [vp, f32p, f32p, uptr]),
    "hcv_timedomain_create": (vp, [uptr, uptr]),
    "hcv_timedomain_destroy": (None, [vp]),
    "hcv_timedomain_set_length": (C.c_int, [vp, uptr]),
    "hcv_timedomain_set_offset": (None, [vp, uptr]),
    "hcv_timedomain_set": (C.c_int, [vp, f32p, uptr]),
    "hcv_timedomain_reset": (None, [vp]),
    "hcv_timedomain_process": (C.c_int, [vp, f32p, f32p, uptr]),
    "hcv_mono_create": (vp, [uptr, C.c_int]),
    "hcv_mono_create_custom": (vp, [uptr, C.c_int, u32, u32, u32, u32, C.c_char_p, usz]),
    "hcv_mono_destroy": (None, [vp]),
    "hcv_mono_set_reset_offset": (None, [vp, iptr]),
    "hcv_mono_resize": (C.c_int, [vp, uptr]),
    "hcv_mono_set": (C.c_int, [vp, f32p, uptr, C.c_int]),
    "hcv_mono_reset": (C.c_int, [vp]),
    "hcv_mono_process": (C.c_int, [vp, f32p, f32p, f32p, uptr, C.c_int]),
    "hcv_ntomono_create": (vp, [u32, uptr, C.c_int]),
    "hcv_ntomono_destroy": (None, [vp]),
    "hcv_ntomono_resize": (C.c_int, [vp, u32, uptr]),
    "hcv_ntomono_set": (C.c_int, [vp, u32, f32p, uptr, C.c_int]),
    "hcv_ntomono_reset": (C.c_int, [vp, u32]),
    "hcv_ntomono_process": (C.c_int, [vp, C.POINTER(f32p), f32p, f32p, usz, usz]),
    "hcv_convolver_create": (vp, [u32, u32, C.c_int]),
    "hcv_convolver_create_parallel": (vp, [u32, C.c_int]),
    "hcv_convolver_destroy": (None, [vp]),
    "hcv_convolver_clear": (None, [vp, C.c_int]),
    "hcv_convolver_clear_chan": (None, [vp, u32, u32, C.c_int]),
    "hcv_convolver_reset": (None, [vp]),
    "hcv_convolver_reset_chan": (C.c_int, [vp, u32, u32]),
    "hcv_convolver_resize": (C.c_int, [vp, u32, u32, uptr]),
    "hcv_convolver_set_f32": (C.c_int, [vp, u32, u32, f32p, uptr, C.c_int]),
    "hcv_convolver_set_f64": (C.c_int, [vp, u32, u32, f64p, uptr, C.c_int]),
    "hcv_convolver_process_f32": (C.c_int, [vp, C.POINTER(f32p), C.POINTER(f32p), usz, usz, usz]),
    "hcv_convolver_process_f64": (C.c_int, [vp, C.POINTER(f64p), C.POINTER(f64p), usz, usz, usz]),
    "hcv_convolver_create_on": (vp, [u32, u32, C.c_int, C.c_int, u32]),
    "hcv_convolver_create_custom": (vp, [u32, u32, C.c_int, uptr, C.c_int, u32, u32, u32, u32, C.c_int, u32]),
    "hcv_convolver_create_extended": (vp, [u32, u32, C.c_int, uptr, C.c_int, u32, u32, u32, u32, C.c_int, u32, u32]),
    "hcv_convolver_set_f32_dev": (C.c_int, [vp, u32, u32, vp, uptr, C.c_int]),
    "hcv_convolver_process_f32_dev": (C.c_int, [vp, vp, usz, vp, usz, usz, usz, usz, C.c_int]),
    "hcv_convolver_synchronize": (C.c_int, [vp]),
    "hcv_convolver_device": (C.c_int, [vp]),
    "hcv_convolver_create_sharded": (vp, [u32, u32, C.c_int, uptr, C.c_int, u32, u32, u32, u32, C.POINTER(C.c_int), C.c_int, u32]),
    "hcv_convolver_num_shards": (C.c_int, [vp]),
    "hcv_rccl_unique_id": (C.c_int, [vp]),
    "hcv_convolver_comm_init": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "hcv_convolver_process_f32_dev_allreduce": (C.c_int, [vp, vp, usz, vp, usz, usz, usz, usz, C.c_int]),
    "hcv_convolver_rt_stats": (C.c_int, [vp, C.POINTER(RtStats)]),
    "hcv_host_register": (C.c_int, [vp, usz]),
    "hcv_host_unregister": (C.c_int, [vp]),
    "hcv_convolver_set_profiling": (None, [vp, C.c_int]),
    "hcv_convolver_num_stages": (C.c_int, [vp]),
    "hcv_convolver_stage_stats": (C.c_int, [vp, C.c_int, C.POINTER(StageStats)]),
    "hcv_convolver_clear_stats": (None, [vp]),
    "hcv_spectral_size": (usz, [usz, usz, C.c_int]),
    "hcv_spectral_convolve_f32": (C.c_int, [f32p, usz, f32p, usz, C.c_int, f32p]),
    "hcv_spectral_correlate_f32": (C.c_int, [f32p, usz, f32p, usz, C.c_int, f32p]),
    "hcv_spectral_convolve_f64": (C.c_int, [f64p, usz, f64p, usz, C.c_int, f64p]),
    "hcv_spectral_correlate_f64": (C.c_int, [f64p, usz, f64p, usz, C.c_int, f64p]),
    "hcv_spectral_convolve_complex_f32": (C.c_int, [f32p, usz, f32p, usz, f32p, usz, f32p, usz, C.c_int, f32p, f32p]),
    "hcv_spectral_correlate_complex_f32": (C.c_int, [f32p, usz, f32p, usz, f32p, usz, f32p, usz, C.c_int, f32p, f32p]),
    "hcv_spectral_convolve_complex_f64": (C.c_int, [f64p, usz, f64p, usz, f64p, usz, f64p, usz, C.c_int, f64p, f64p]),
    "hcv_spectral_correlate_complex_f64": (C.c_int, [f64p, usz, f64p, usz, f64p, usz, f64p, usz, C.c_int, f64p, f64p]),
    "hcv_spectral_convolve_f32_dev": (C.c_int, [vp, usz, vp, usz, C.c_int, vp, vp, C.c_int]),
    "hcv_spectral_correlate_f32_dev": (C.c_int, [vp, usz, vp, usz, C.c_int, vp, vp, C.c_int]),
    "hcv_fft_exec": (C.c_int, [C.POINTER(FFTCall)]),
    "hcv_fft_exec_dev": (C.c_int, [C.POINTER(FFTCall), vp, C.c_int]),
    "hcv_ir_exec": (C.c_int, [C.POINTER(IRCall)]),
    "hcv_ir_exec_dev": (C.c_int, [C.POINTER(IRCall), vp, C.c_int]),
    "hcv_ir_product_exec": (C.c_int, [C.POINTER(IRProductCall)]),
    "hcv_ir_product_exec_dev": (C.c_int, [C.POINTER(IRProductCall), vp, C.c_int]),
    "hcv_spectral_phase_size": (usz, [usz, C.c_double]),
    "hcv_spectral_change_phase_f32": (C.c_int, [f32p, usz, C.c_double, C.c_double, f32p]),
    "hcv_spectral_change_phase_f64": (C.c_int, [f64p, usz, C.c_double, C.c_double, f64p]),
    "hcv_iaudiofile_open": (vp, [C.c_char_p]),
    "hcv_oaudiofile_open": (vp, [C.c_char_p, C.c_int, C.c_int, C.c_uint, C.c_double, C.c_int]),
    "hcv_audiofile_close": (None, [vp]),
    "hcv_audiofile_is_open": (C.c_int, [vp]),
    "hcv_audiofile_get_info": (C.c_int, [vp, C.POINTER(AudioFileInfo)]),
    "hcv_audiofile_seek": (None, [vp, u32]),
    "hcv_audiofile_position": (u32, [vp]),
    "hcv_iaudiofile_read_raw": (None, [vp, vp, u32]),
    "hcv_iaudiofile_read_interleaved_f32": (None, [vp, f32p, u32]),
    "hcv_iaudiofile_read_interleaved_f64": (None, [vp, f64p, u32]),
    "hcv_iaudiofile_read_channel_f32": (None, [vp, f32p, u32, C.c_uint]),
    "hcv_iaudiofile_read_channel_f64": (None, [vp, f64p, u32, C.c_uint]),
    "hcv_oaudiofile_write_raw": (None, [vp, vp, u32]),
    "hcv_oaudiofile_write_interleaved_f32": (None, [vp, f32p, u32]),
    "hcv_oaudiofile_write_interleaved_f64": (None, [vp, f64p, u32]),
    "hcv_oaudiofile_write_channel_f32": (None, [vp, f32p, u32, C.c_uint]),
    "hcv_oaudiofile_write_channel_f64": (None, [vp, f64p, u32, C.c_uint]),
}

_lib = None


def load():
    """Load the shared library (once) and bind every declared symbol.  Raises ImportError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C hisstools_library_amd/csrc).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lenient = bool(os.environ.get("HCV_AB_OLD_LIBRARY"))      # tools/ab.sh only: time an older build that lacks newer entry points
    for name, (res, args) in SIGNATURES.items():
        if lenient and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)      # AttributeError here means header and library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return (load().hcv_last_error() or b"").decode()
