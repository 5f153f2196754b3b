"""Python mirror of the whole hisstools_* FFT surface (HISSTools_FFT.h:87-369) over ``hcv_fft_exec``.

The reference overloads on the setup / split types; here the numpy dtype (float32 / float64) selects the precision.
Every function takes one transform (1-D arrays) or a batch (2-D arrays, one transform per row) and runs on the GPU —
there is no host implementation.  Like the reference the transforms are unnormalised and the real spectra are doubled
with DC / Nyquist packed into bin 0.  Unlike the reference nothing is modified in place: results are returned.
"""
from __future__ import annotations

import ctypes as C
from enum import IntEnum

import numpy as np

from . import _lib


class Op(IntEnum):                 # hcv_fft_call.op
    FFT = 0
    IFFT = 1
    RFFT = 2
    RIFFT = 3
    RFFT_ZIP = 4
    RIFFT_ZIP = 5
    UNZIP = 6
    ZIP = 7


class Precision(IntEnum):          # hcv_fft_call.precision
    F32 = 0
    F64 = 1
    F32_TO_F64 = 2


def _prec(dtype) -> Precision:
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return Precision.F32
    if dtype == np.float64:
        return Precision.F64
    raise TypeError(f"hisstools FFT: float32 or float64 data expected, got {dtype}")


def _exec(call: _lib.FFTCall, what: str):
    if _lib.load().hcv_fft_exec(C.byref(call)) != 0:
        raise RuntimeError(f"{what}: {_lib.last_error()}")


def _split_pair(realp, imagp):
    re = np.ascontiguousarray(realp)
    if re.dtype not in (np.float32, np.float64):
        re = re.astype(np.float64)
    im = np.ascontiguousarray(imagp, dtype=re.dtype)
    if re.shape != im.shape:
        raise ValueError("realp and imagp must have the same shape")
    one = re.ndim == 1
    return (re.reshape(1, -1), im.reshape(1, -1), one) if one else (re, im, one)


def _split_op(op: Op, realp, imagp, log2n: int, length: int, what: str):
    re, im, one = _split_pair(realp, imagp)
    if re.shape[1] < length:
        raise ValueError(f"{what}: {length} values per transform expected, got {re.shape[1]}")
    out_re, out_im = re.copy(), im.copy()
    call = _lib.FFTCall(op=int(op), precision=int(_prec(re.dtype)), log2n=log2n, batch=re.shape[0],
                        src_a=out_re.ctypes.data, src_b=out_im.ctypes.data, dst_a=out_re.ctypes.data, dst_b=out_im.ctypes.data,
                        src_stride=re.shape[1], dst_stride=re.shape[1], in_length=0)
    _exec(call, what)
    return (out_re[0], out_im[0]) if one else (out_re, out_im)


def hisstools_fft(realp, imagp, log2n: int):
    """hisstools_fft(setup, split, log2n) (HISSTools_FFT.h:130,142): complex forward transform of 2^log2n points."""
    return _split_op(Op.FFT, realp, imagp, log2n, 1 << log2n, "hisstools_fft")


def hisstools_ifft(realp, imagp, log2n: int):
    """hisstools_ifft (HISSTools_FFT.h:220,232): complex inverse, unnormalised (ifft(fft(z)) = N z)."""
    return _split_op(Op.IFFT, realp, imagp, log2n, 1 << log2n, "hisstools_ifft")


def hisstools_rfft_split(realp, imagp, log2n: int):
    """hisstools_rfft(setup, split, log2n) (HISSTools_FFT.h:154,166): real forward transform of 2^log2n samples that
    were unzipped into realp (even) / imagp (odd)."""
    return _split_op(Op.RFFT, realp, imagp, log2n, (1 << log2n) >> 1, "hisstools_rfft")


def hisstools_rifft_split(realp, imagp, log2n: int):
    """hisstools_rifft(setup, split, log2n) (HISSTools_FFT.h:244,256): packed spectrum -> unzipped samples, unnormalised."""
    return _split_op(Op.RIFFT, realp, imagp, log2n, (1 << log2n) >> 1, "hisstools_rifft")


def _samples(x, dtype=None):
    x = np.ascontiguousarray(x)
    if dtype is not None:
        x = x.astype(dtype, copy=False)
    elif x.dtype not in (np.float32, np.float64):
        x = x.astype(np.float64)
    one = x.ndim == 1
    return (x.reshape(1, -1), one) if one else (x, one)


def _to_split(op: Op, x, log2n: int, in_length, out_dtype, what: str):
    x, one = _samples(x)
    out_dtype = np.dtype(x.dtype if out_dtype is None else out_dtype)
    if x.dtype == np.float32 and out_dtype == np.float64:
        prec = Precision.F32_TO_F64
    elif x.dtype == out_dtype:
        prec = _prec(out_dtype)
    else:
        raise TypeError(f"{what}: {x.dtype} samples cannot produce a {out_dtype} spectrum")
    in_length = x.shape[1] if in_length is None else min(in_length, x.shape[1])
    half = (1 << log2n) >> 1
    re = np.zeros((x.shape[0], half), out_dtype)
    im = np.zeros((x.shape[0], half), out_dtype)
    if half and in_length:
        call = _lib.FFTCall(op=int(op), precision=int(prec), log2n=log2n, batch=x.shape[0], src_a=x.ctypes.data, src_b=None,
                            dst_a=re.ctypes.data, dst_b=im.ctypes.data, src_stride=x.shape[1], dst_stride=half, in_length=in_length)
        _exec(call, what)
    return (re[0], im[0]) if one else (re, im)


def hisstools_rfft(x, log2n: int, in_length=None, out_dtype=None):
    """hisstools_rfft(setup, in, out, in_length, log2n) (HISSTools_FFT.h:180,194,208): zero-padding unzip + real transform.
    float32 samples with out_dtype=float64 select the converting overload (:208)."""
    return _to_split(Op.RFFT_ZIP, x, log2n, in_length, out_dtype, "hisstools_rfft")


def hisstools_unzip_zero(x, in_length: int, log2n: int, out_dtype=None):
    """hisstools_unzip_zero (HISSTools_FFT.h:295,308,321)."""
    return _to_split(Op.UNZIP, x, log2n, in_length, out_dtype, "hisstools_unzip_zero")


def hisstools_unzip(x, log2n: int):
    """hisstools_unzip (HISSTools_FFT.h:333,345): 2^log2n samples -> (even, odd)."""
    return _to_split(Op.UNZIP, x, log2n, 1 << log2n, None, "hisstools_unzip")


def _from_split(op: Op, realp, imagp, log2n: int, what: str):
    re, im, one = _split_pair(realp, imagp)
    n = 1 << log2n
    half = n >> 1
    if re.shape[1] < half:
        raise ValueError(f"{what}: {half} values per transform expected, got {re.shape[1]}")
    out = np.zeros((re.shape[0], n if half else 0), re.dtype)
    if half:
        call = _lib.FFTCall(op=int(op), precision=int(_prec(re.dtype)), log2n=log2n, batch=re.shape[0], src_a=re.ctypes.data,
                            src_b=im.ctypes.data, dst_a=out.ctypes.data, dst_b=None, src_stride=re.shape[1], dst_stride=n, in_length=0)
        _exec(call, what)
    return out[0] if one else out


def hisstools_rifft(realp, imagp, log2n: int):
    """hisstools_rifft(setup, in, out, log2n) (HISSTools_FFT.h:269,282): packed spectrum -> 2^log2n samples, unnormalised."""
    return _from_split(Op.RIFFT_ZIP, realp, imagp, log2n, "hisstools_rifft")


def hisstools_zip(realp, imagp, log2n: int):
    """hisstools_zip (HISSTools_FFT.h:357,369)."""
    return _from_split(Op.ZIP, realp, imagp, log2n, "hisstools_zip")


def exec_dev(op: Op, precision: Precision, log2n: int, batch: int, src_a: int, src_b: int, dst_a: int, dst_b: int,
             src_stride: int = 0, dst_stride: int = 0, in_length: int = 0, stream: int = 0, sync: bool = True):
    """hcv_fft_exec_dev: the same operations on device pointers (integers, e.g. ``tensor.data_ptr()``), enqueued on a HIP stream."""
    call = _lib.FFTCall(op=int(op), precision=int(precision), log2n=log2n, batch=batch, src_a=src_a or None, src_b=src_b or None,
                        dst_a=dst_a or None, dst_b=dst_b or None, src_stride=src_stride, dst_stride=dst_stride, in_length=in_length)
    if _lib.load().hcv_fft_exec_dev(C.byref(call), stream or None, int(sync)) != 0:
        raise RuntimeError(f"hcv_fft_exec_dev: {_lib.last_error()}")
