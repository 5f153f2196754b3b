"""Python mirror of the reference's spectral IR functions (SpectralFunctions.hpp:365-436) over ``hcv_ir_exec`` / ``hcv_ir_product_exec``.

Spectra are packed half spectra as ``hisstools_rfft`` produces them: ``fft_size / 2`` values per array, bin 0 =
(DC, Nyquist).  1-D arrays are one spectrum, 2-D arrays a batch (one spectrum per row, one launch).  The numpy dtype
(float32 / float64) selects the precision.  Everything runs on the GPU; results are returned, nothing is modified in place.
"""
from __future__ import annotations

import ctypes as C
from enum import IntEnum

import numpy as np

from . import _lib


class IrOp(IntEnum):               # hcv_ir_call.op
    COPY = 0
    SPIKE = 1
    DELAY = 2
    TIME_REVERSE = 3
    PHASE = 4


def _log2(fft_size: int) -> int:
    if fft_size < 2 or fft_size & (fft_size - 1):
        raise ValueError("fft_size must be a power of two >= 2")
    return fft_size.bit_length() - 1


def _run(op: IrOp, realp, imagp, fft_size: int, value: float = 0.0, zero_center: bool = False, dtype=None, batch=None):
    log2n = _log2(fft_size)
    half = fft_size >> 1
    if op == IrOp.SPIKE:
        dt = np.dtype(np.float64 if dtype is None else dtype)
        rows = 1 if batch is None else batch
        re = im = None
        one = batch is None
    else:
        re = np.ascontiguousarray(realp)
        if re.dtype not in (np.float32, np.float64):
            re = re.astype(np.float64)
        im = np.ascontiguousarray(imagp, dtype=re.dtype)
        one = re.ndim == 1
        if one:
            re, im = re.reshape(1, -1), im.reshape(1, -1)
        if re.shape != im.shape or re.shape[1] < half:
            raise ValueError(f"{half} values per array expected")
        dt, rows = re.dtype, re.shape[0]
    if dt not in (np.float32, np.float64):
        raise TypeError("float32 or float64 expected")
    out_re, out_im = np.zeros((rows, half), dt), np.zeros((rows, half), dt)
    call = _lib.IRCall(op=int(op), precision=0 if dt == np.float32 else 1, log2n=log2n, batch=rows,
                       src_re=None if re is None else re.ctypes.data, src_im=None if im is None else im.ctypes.data,
                       dst_re=out_re.ctypes.data, dst_im=out_im.ctypes.data, src_stride=0 if re is None else re.shape[1], dst_stride=half,
                       value=float(value), zero_center=int(bool(zero_center)))
    if _lib.load().hcv_ir_exec(C.byref(call)) != 0:
        raise RuntimeError(f"ir_{op.name.lower()}: {_lib.last_error()}")
    return (out_re[0], out_im[0]) if one else (out_re, out_im)


def ir_copy(realp, imagp, fft_size: int):
    """ir_copy(out, in, fft_size) (SpectralFunctions.hpp:365-369)."""
    return _run(IrOp.COPY, realp, imagp, fft_size)


def ir_spike(fft_size: int, spike_position: float, dtype=np.float32):
    """ir_spike(out, fft_size, spike_position) (:371-375): spectrum of a unit impulse at a (fractional) sample position."""
    return _run(IrOp.SPIKE, None, None, fft_size, spike_position, dtype=dtype)


def ir_delay(realp, imagp, fft_size: int, delay: float):
    """ir_delay(out, in, fft_size, delay) (:377-384): circular delay by a (fractional) number of samples."""
    return _run(IrOp.DELAY, realp, imagp, fft_size, delay)


def ir_time_reverse(realp, imagp, fft_size: int):
    """ir_time_reverse(out, in, fft_size) (:386-390)."""
    return _run(IrOp.TIME_REVERSE, realp, imagp, fft_size)


def ir_phase(realp, imagp, fft_size: int, phase: float, zero_center: bool = False):
    """ir_phase(setup, out, in, fft_size, phase, zero_center) (:392-413): 0 minimum, 0.5 linear, 1 maximum phase."""
    return _run(IrOp.PHASE, realp, imagp, fft_size, phase, zero_center)


class IrProductOp(IntEnum):        # hcv_ir_product_call.op
    CONVOLVE_COMPLEX = 0
    CONVOLVE_REAL = 1
    CORRELATE_COMPLEX = 2
    CORRELATE_REAL = 3


def _product(op: IrProductOp, r1, i1, r2, i2, fft_size: int, scale: float):
    if fft_size < 1 or fft_size & (fft_size - 1):
        raise ValueError("fft_size must be a power of two")
    n = fft_size >> 1 if int(op) & 1 else fft_size
    a = np.ascontiguousarray(r1)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    dt = a.dtype
    b = np.ascontiguousarray(i1, dtype=dt)
    c, d = np.ascontiguousarray(r2, dtype=dt), np.ascontiguousarray(i2, dtype=dt)
    one = a.ndim == 1
    if one:
        a, b = a.reshape(1, -1), b.reshape(1, -1)
    broadcast = c.ndim == 1
    if broadcast:
        c, d = c.reshape(1, -1), d.reshape(1, -1)
    if a.shape != b.shape or c.shape != d.shape or a.shape[1] < n or c.shape[1] < n or (not broadcast and c.shape[0] != a.shape[0]):
        raise ValueError(f"{n} values per array expected; the second operand one spectrum or one per row")
    out_re, out_im = np.zeros((a.shape[0], n), dt), np.zeros((a.shape[0], n), dt)
    call = _lib.IRProductCall(op=int(op), precision=0 if dt == np.float32 else 1, size=fft_size, batch=a.shape[0], a_re=a.ctypes.data, a_im=b.ctypes.data,
                              b_re=c.ctypes.data, b_im=d.ctypes.data, dst_re=out_re.ctypes.data, dst_im=out_im.ctypes.data, a_stride=a.shape[1],
                              b_stride=c.shape[1], dst_stride=n, b_broadcast=int(broadcast and a.shape[0] > 1), scale=float(scale))
    if _lib.load().hcv_ir_product_exec(C.byref(call)) != 0:
        raise RuntimeError(f"ir_{op.name.lower()}: {_lib.last_error()}")
    return (out_re[0], out_im[0]) if one else (out_re, out_im)


def ir_convolve_complex(r1, i1, r2, i2, fft_size: int, scale: float = 1.0):
    """ir_convolve_complex(out, in1, in2, fft_size, scale) (:414-418): scale * in1 * in2 on fft_size values per array."""
    return _product(IrProductOp.CONVOLVE_COMPLEX, r1, i1, r2, i2, fft_size, scale)


def ir_convolve_real(r1, i1, r2, i2, fft_size: int, scale: float = 1.0):
    """ir_convolve_real (:420-424): the same on packed half spectra of fft_size real samples (bin 0 = (DC, Nyquist))."""
    return _product(IrProductOp.CONVOLVE_REAL, r1, i1, r2, i2, fft_size, scale)


def ir_correlate_complex(r1, i1, r2, i2, fft_size: int, scale: float = 1.0):
    """ir_correlate_complex (:426-430): scale * in1 * conj(in2)."""
    return _product(IrProductOp.CORRELATE_COMPLEX, r1, i1, r2, i2, fft_size, scale)


def ir_correlate_real(r1, i1, r2, i2, fft_size: int, scale: float = 1.0):
    """ir_correlate_real (:432-436)."""
    return _product(IrProductOp.CORRELATE_REAL, r1, i1, r2, i2, fft_size, scale)


def exec_dev(op: IrOp, precision: int, log2n: int, batch: int, src_re: int, src_im: int, dst_re: int, dst_im: int,
             src_stride: int = 0, dst_stride: int = 0, value: float = 0.0, zero_center: bool = False, stream: int = 0, sync: bool = True):
    """hcv_ir_exec_dev: the same operations on device pointers (integers), enqueued on a HIP stream."""
    call = _lib.IRCall(op=int(op), precision=int(precision), log2n=log2n, batch=batch, src_re=src_re or None, src_im=src_im or None,
                       dst_re=dst_re or None, dst_im=dst_im or None, src_stride=src_stride, dst_stride=dst_stride, value=float(value),
                       zero_center=int(bool(zero_center)))
    if _lib.load().hcv_ir_exec_dev(C.byref(call), stream or None, int(sync)) != 0:
        raise RuntimeError(f"hcv_ir_exec_dev: {_lib.last_error()}")
