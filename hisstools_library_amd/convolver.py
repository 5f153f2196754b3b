"""Python mirror of the reference's convolution classes, bound to the HIP engine through the C ABI.

Same class names, method names, argument meaning and error codes as
``HIRT_Multichannel_Convolution/{Convolver,NToMonoConvolve,MonoConvolve,PartitionedConvolve,
TimeDomainConvolve}.h`` so tests read like code written against the reference.  All arithmetic runs on
the GPU (libhisstools_amd.so); nothing here computes audio.
"""
from __future__ import annotations

import ctypes as C
from enum import IntEnum

import numpy as np

from . import _lib
from ._lib import f32p, f64p


class LatencyMode(IntEnum):            # MonoConvolve.h:14-19
    kLatencyZero = 0
    kLatencyShort = 1
    kLatencyMedium = 2


kLatencyZero, kLatencyShort, kLatencyMedium = LatencyMode.kLatencyZero, LatencyMode.kLatencyShort, LatencyMode.kLatencyMedium


class ConvolveError(IntEnum):          # ConvolveErrors.h:4-19
    CONVOLVE_ERR_NONE = 0
    CONVOLVE_ERR_IN_CHAN_OUT_OF_RANGE = 1
    CONVOLVE_ERR_OUT_CHAN_OUT_OF_RANGE = 2
    CONVOLVE_ERR_MEM_UNAVAILABLE = 3
    CONVOLVE_ERR_MEM_ALLOC_TOO_SMALL = 4
    CONVOLVE_ERR_TIME_IMPULSE_TOO_LONG = 5
    CONVOLVE_ERR_TIME_LENGTH_OUT_OF_RANGE = 6
    CONVOLVE_ERR_PARTITION_LENGTH_TOO_LARGE = 7
    CONVOLVE_ERR_FFT_SIZE_MAX_TOO_SMALL = 8
    CONVOLVE_ERR_FFT_SIZE_MAX_TOO_LARGE = 9
    CONVOLVE_ERR_FFT_SIZE_MAX_NON_POWER_OF_TWO = 10
    CONVOLVE_ERR_FFT_SIZE_OUT_OF_RANGE = 11
    CONVOLVE_ERR_FFT_SIZE_NON_POWER_OF_TWO = 12


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a: np.ndarray):
    return a.ctypes.data_as(f32p)


def _need(handle, what: str):
    if not handle:
        raise RuntimeError(f"{what}: {_lib.last_error() or 'creation failed'}")
    return handle


def _ptr_array(rows, ptr_t):
    arr = (ptr_t * max(len(rows), 1))()
    for i, r in enumerate(rows):
        arr[i] = r.ctypes.data_as(ptr_t)
    return arr


def _blocks(total: int, block):
    pos, k = 0, 0
    sizes = [block] if isinstance(block, int) else list(block)
    while pos < total:
        n = min(sizes[k % len(sizes)], total - pos)
        yield pos, n
        pos += n
        k += 1


def _check(rc: int, what: str) -> int:
    if rc < 0:
        raise RuntimeError(f"{what}: {_lib.last_error()}")
    return rc


# ------------------------------------------------------------------------------------------- FFT plumbing

def hisstools_rfft(x, log2n: int, in_length=None):
    """hisstools_rfft(setup, in, out, in_length, log2n) (HISSTools_FFT.cpp:226-230).  x: [n] or [batch][n].
    Returns (realp, imagp) with 2^(log2n-1) entries per row."""
    L = _lib.load()
    x = _f32(x)
    one = x.ndim == 1
    x2 = x.reshape(1, -1) if one else x
    batch, stride = x2.shape
    n_in = stride if in_length is None else in_length
    half = 1 << (log2n - 1)
    re = np.zeros((batch, half), np.float32)
    im = np.zeros((batch, half), np.float32)
    _check(L.hcv_rfft_f32(_fp(x2), n_in, stride, batch, log2n, _fp(re), _fp(im)), "hcv_rfft_f32")
    return (re[0], im[0]) if one else (re, im)


def hisstools_rifft(realp, imagp, log2n: int):
    """hisstools_rifft(setup, in, out, log2n) (HISSTools_FFT.cpp:244-248); unnormalised, input left intact."""
    L = _lib.load()
    re, im = _f32(realp), _f32(imagp)
    one = re.ndim == 1
    re2, im2 = (re.reshape(1, -1), im.reshape(1, -1)) if one else (re, im)
    batch = re2.shape[0]
    out = np.zeros((batch, 1 << log2n), np.float32)
    _check(L.hcv_rifft_f32(_fp(re2), _fp(im2), batch, log2n, _fp(out)), "hcv_rifft_f32")
    return out[0] if one else out


# ------------------------------------------------------------------------------------------- spectral_processor (next row)

class EdgeMode(IntEnum):               # spectral_processor::EdgeMode, SpectralProcessor.hpp:22
    Linear = 0
    Wrap = 1
    WrapCentre = 2
    Fold = 3
    FoldRepeat = 4


class spectral_processor:
    """spectral_processor<T>::convolve / correlate (SpectralProcessor.hpp:164-184): the real overloads (float32 or float64 by
    the inputs' dtype) and the complex overloads (``convolve_complex`` / ``correlate_complex``), plus change_phase."""

    def __init__(self, max_fft_size=1 << 22):
        self.L = _lib.load()
        self._max = max_fft_size

    def convolved_size(self, size1, size2, mode):
        return self.L.hcv_spectral_size(size1, size2, int(mode))

    correlated_size = convolved_size

    @staticmethod
    def _typed(*arrays):
        """float64 when any operand is float64, float32 otherwise; returns (dtype, pointer type, contiguous arrays)"""
        arrays = [np.ascontiguousarray(a) for a in arrays]
        dbl = any(a.dtype == np.float64 for a in arrays)
        dt = np.float64 if dbl else np.float32
        return dt, (f64p if dbl else f32p), [np.ascontiguousarray(a, dt) for a in arrays]

    def _run(self, op, in1, in2, mode, what):
        dt, pt, (a, b) = self._typed(in1, in2)
        fn = getattr(self.L, f"hcv_spectral_{op}_{'f64' if dt == np.float64 else 'f32'}")
        n = self.L.hcv_spectral_size(a.size, b.size, int(mode))
        out = np.zeros(n, dt)
        if n:
            _check(fn(a.ctypes.data_as(pt), a.size, b.ctypes.data_as(pt), b.size, int(mode), out.ctypes.data_as(pt)), what)
        return out

    def convolve(self, in1, in2, mode=EdgeMode.Linear):
        return self._run("convolve", in1, in2, mode, "spectral_processor.convolve")

    def correlate(self, in1, in2, mode=EdgeMode.Linear):
        return self._run("correlate", in1, in2, mode, "spectral_processor.correlate")

    def _run_complex(self, op, r1, i1, r2, i2, mode, what):
        dt, pt, ins = self._typed(r1, i1, r2, i2)
        fn = getattr(self.L, f"hcv_spectral_{op}_complex_{'f64' if dt == np.float64 else 'f32'}")
        n = self.L.hcv_spectral_size(max(ins[0].size, ins[1].size), max(ins[2].size, ins[3].size), int(mode))
        r_out, i_out = np.zeros(n, dt), np.zeros(n, dt)
        if n:
            args = []
            for a in ins:
                args += [a.ctypes.data_as(pt), a.size]
            _check(fn(*args, int(mode), r_out.ctypes.data_as(pt), i_out.ctypes.data_as(pt)), what)
        return r_out, i_out

    def convolve_complex(self, r_in1, i_in1, r_in2, i_in2, mode=EdgeMode.Linear):
        """convolve(r_out, i_out, r_in1, i_in1, r_in2, i_in2, mode) (SpectralProcessor.hpp:164-167); returns (r_out, i_out)"""
        return self._run_complex("convolve", r_in1, i_in1, r_in2, i_in2, mode, "spectral_processor.convolve_complex")

    def correlate_complex(self, r_in1, i_in1, r_in2, i_in2, mode=EdgeMode.Linear):
        """correlate(r_out, i_out, …) (SpectralProcessor.hpp:176-179): in1 x conj(in2) in the spectral domain"""
        return self._run_complex("correlate", r_in1, i_in1, r_in2, i_in2, mode, "spectral_processor.correlate_complex")

    def convolve_dev(self, in1_ptr: int, size1: int, in2_ptr: int, size2: int, out_ptr: int, mode=EdgeMode.Linear, correlate=False, stream: int = 0, sync=True):
        """hcv_spectral_convolve_f32_dev / _correlate_f32_dev: float32 operands and result resident in HBM (device pointers as
        integers, e.g. ``tensor.data_ptr()``); ``out`` must hold ``convolved_size(size1, size2, mode)`` floats."""
        fn = self.L.hcv_spectral_correlate_f32_dev if correlate else self.L.hcv_spectral_convolve_f32_dev
        _check(fn(in1_ptr, size1, in2_ptr, size2, int(mode), out_ptr, stream or None, int(sync)), "spectral_processor.convolve_dev")

    def change_phase(self, x, phase: float, time_multiplier: float = 1.0):
        """spectral_processor<T>::change_phase (SpectralProcessor.hpp:188-208); float32 or float64 by the input's dtype.
        Returns the fft_size output samples (fft_size = the power of two covering round(size * time_multiplier))."""
        x = np.ascontiguousarray(x)
        if x.dtype != np.float64:
            x = x.astype(np.float32, copy=False)
        n = self.L.hcv_spectral_phase_size(x.size, float(time_multiplier))
        if not n:
            raise ValueError("spectral_processor.change_phase: size out of range")
        out = np.zeros(n, x.dtype)
        if x.dtype == np.float32:
            rc = self.L.hcv_spectral_change_phase_f32(_fp(x), x.size, float(phase), float(time_multiplier), _fp(out))
        else:
            rc = self.L.hcv_spectral_change_phase_f64(x.ctypes.data_as(f64p), x.size, float(phase), float(time_multiplier), out.ctypes.data_as(f64p))
        _check(rc, "spectral_processor.change_phase")
        return out


def host_register(a: np.ndarray):
    """Pin and map a numpy array (hcv_host_register): process() calls on [channels][n] views of it skip the staging copies.
    Keep the array alive and call host_unregister(a) before dropping it."""
    _check(_lib.load().hcv_host_register(a.ctypes.data, a.nbytes), "host_register")


def ctl_reserve(device: int, nbytes: int):
    """hcv_ctl_reserve: the least the control arena of `device` holds while the device has an object (set / resize beside a running stream
    then never has the driver map memory: include/hisstools_amd.h).  0 = what the objects ask for themselves."""
    _check(_lib.load().hcv_ctl_reserve(device, nbytes), "ctl_reserve")


def ctl_reserved(device: int) -> int:
    return int(_lib.load().hcv_ctl_reserved(device))


def host_unregister(a: np.ndarray):
    _check(_lib.load().hcv_host_unregister(a.ctypes.data), "host_unregister")


def rccl_unique_id() -> bytes:
    """128 bytes (ncclUniqueId) for hcv_convolver_comm_init, made on ONE rank of a row group and handed to the others."""
    buf = C.create_string_buffer(128)
    _check(_lib.load().hcv_rccl_unique_id(buf), "rccl_unique_id")
    return buf.raw


# ------------------------------------------------------------------------------------------- classes

class PartitionedConvolve:
    def __init__(self, maxFFTSize, maxLength, offset, length):
        self.L = _lib.load()
        self.h = _need(self.L.hcv_partitioned_create(maxFFTSize, maxLength, offset, length), "PartitionedConvolve")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.hcv_partitioned_destroy(self.h)
            self.h = None

    def setFFTSize(self, FFTSize): return self.L.hcv_partitioned_set_fft_size(self.h, FFTSize)
    def setLength(self, length): return self.L.hcv_partitioned_set_length(self.h, length)
    def setOffset(self, offset): self.L.hcv_partitioned_set_offset(self.h, offset)
    def setResetOffset(self, offset=-1): self.L.hcv_partitioned_set_reset_offset(self.h, offset)
    def reset(self): self.L.hcv_partitioned_reset(self.h)

    def set(self, input, length=None):
        if input is None:
            return self.L.hcv_partitioned_set(self.h, None, 0 if length is None else length)
        ir = _f32(input)
        return self.L.hcv_partitioned_set(self.h, _fp(ir), ir.size if length is None else length)

    def process(self, x, out=None):
        """Returns (wrote, out): wrote False means `out` was left untouched (no IR loaded)."""
        x = _f32(x)
        if out is None:
            out = np.full(x.size, np.nan, np.float32)
        rc = _check(self.L.hcv_partitioned_process(self.h, _fp(x), _fp(out), x.size), "PartitionedConvolve.process")
        return bool(rc), out

    def run(self, x, block=512):
        x = _f32(x)
        y = np.zeros_like(x)
        for pos, n in _blocks(x.size, block):
            _check(self.L.hcv_partitioned_process(self.h, _fp(x[pos:pos + n]), _fp(y[pos:pos + n]), n), "process")
        return y


class TimeDomainConvolve:
    def __init__(self, offset, length):
        self.L = _lib.load()
        self.h = _need(self.L.hcv_timedomain_create(offset, length), "TimeDomainConvolve")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.hcv_timedomain_destroy(self.h)
            self.h = None

    def setLength(self, length): return self.L.hcv_timedomain_set_length(self.h, length)
    def setOffset(self, offset): self.L.hcv_timedomain_set_offset(self.h, offset)
    def reset(self): self.L.hcv_timedomain_reset(self.h)

    def set(self, input, length=None):
        if input is None:
            return self.L.hcv_timedomain_set(self.h, None, 0 if length is None else length)
        ir = _f32(input)
        return self.L.hcv_timedomain_set(self.h, _fp(ir), ir.size if length is None else length)

    def process(self, x, out=None):
        x = _f32(x)
        if out is None:
            out = np.full(x.size, np.nan, np.float32)
        rc = _check(self.L.hcv_timedomain_process(self.h, _fp(x), _fp(out), x.size), "TimeDomainConvolve.process")
        return bool(rc), out

    def run(self, x, block=512):
        x = _f32(x)
        y = np.zeros_like(x)
        for pos, n in _blocks(x.size, block):
            _check(self.L.hcv_timedomain_process(self.h, _fp(x[pos:pos + n]), _fp(y[pos:pos + n]), n), "process")
        return y


class MonoConvolve:
    """MonoConvolve(maxLength, latency) or MonoConvolve(maxLength, zeroLatency, A, B=0, C=0, D=0)."""

    def __init__(self, maxLength, latency=None, zeroLatency=None, A=0, B=0, C_=0, D=0):
        self.L = _lib.load()
        if latency is not None:
            self.h = _need(self.L.hcv_mono_create(maxLength, int(latency)), "MonoConvolve")
        else:
            buf = C.create_string_buffer(128)
            self.h = self.L.hcv_mono_create_custom(maxLength, int(bool(zeroLatency)), A, B, C_, D, buf, 128)
            if not self.h:
                # the reference throws std::runtime_error with this text (MonoConvolve.cpp:207-229)
                raise RuntimeError(buf.value.decode() or _lib.last_error())

    def __del__(self):
        if getattr(self, "h", None):
            self.L.hcv_mono_destroy(self.h)
            self.h = None

    def setResetOffset(self, offset=-1): self.L.hcv_mono_set_reset_offset(self.h, offset)
    def resize(self, length): return self.L.hcv_mono_resize(self.h, length)
    def reset(self): return self.L.hcv_mono_reset(self.h)

    def set(self, input, requestResize, length=None):
        if input is None:
            return self.L.hcv_mono_set(self.h, None, 0 if length is None else length, int(requestResize))
        ir = _f32(input)
        return self.L.hcv_mono_set(self.h, _fp(ir), ir.size if length is None else length, int(requestResize))

    def process(self, x, out=None, accumulate=False):
        x = _f32(x)
        if out is None:
            out = np.full(x.size, np.nan, np.float32)
        temp = np.zeros(x.size, np.float32)
        _check(self.L.hcv_mono_process(self.h, _fp(x), _fp(temp), _fp(out), x.size, int(accumulate)), "MonoConvolve.process")
        return out

    def run(self, x, block=512):
        x = _f32(x)
        y = np.zeros_like(x)
        temp = np.zeros(x.size, np.float32)
        for pos, n in _blocks(x.size, block):
            _check(self.L.hcv_mono_process(self.h, _fp(x[pos:pos + n]), _fp(temp), _fp(y[pos:pos + n]), n, 0), "process")
        return y


class NToMonoConvolve:
    def __init__(self, inChans, maxLength, latency):
        self.L = _lib.load()
        self.h = _need(self.L.hcv_ntomono_create(inChans, maxLength, int(latency)), "NToMonoConvolve")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.hcv_ntomono_destroy(self.h)
            self.h = None

    def resize(self, inChan, length): return self.L.hcv_ntomono_resize(self.h, inChan, length)
    def reset(self, inChan): return self.L.hcv_ntomono_reset(self.h, inChan)

    def set(self, inChan, input, resize, length=None):
        if input is None:
            return self.L.hcv_ntomono_set(self.h, inChan, None, 0 if length is None else length, int(resize))
        ir = _f32(input)
        return self.L.hcv_ntomono_set(self.h, inChan, _fp(ir), ir.size if length is None else length, int(resize))

    def run(self, ins, block=512, activeIns=None):
        ins = _f32(ins)
        nin, total = ins.shape
        y = np.zeros(total, np.float32)
        temp = np.zeros(total, np.float32)
        act = nin if activeIns is None else activeIns
        for pos, n in _blocks(total, block):
            rows = [ins[i, pos:pos + n] for i in range(nin)]
            _check(self.L.hcv_ntomono_process(self.h, _ptr_array(rows, f32p), _fp(y[pos:pos + n]), _fp(temp), n, act), "process")
        return y


class Convolver:
    """Convolver(numIns, numOuts, latency) — N x M matrix; Convolver(numIO, latency=...) — parallel (diagonal)."""

    def __init__(self, numIns, numOuts=None, latency=kLatencyZero, device=-1, maxBlock=0, custom=None, tailRatio=0, devices=None):
        self.L = _lib.load()
        if devices is not None:
            # MI355X extension: ONE object sharded over several devices (hcv_convolver_create_sharded) — output rows first,
            # inputs too when there are fewer rows than devices
            parallel = numOuts is None
            if custom is not None:
                maxLength, zero, A, B, C_, D = custom
            else:
                maxLength, (zero, A, B, C_, D) = 16384, {0: (True, 256, 1024, 4096, 16384), 1: (False, 256, 1024, 4096, 16384)}.get(
                    int(latency), (False, 1024, 4096, 16384, 0))
            devs = (C.c_int * len(devices))(*[int(d) for d in devices])
            self.h = _need(self.L.hcv_convolver_create_sharded(numIns, numIns if parallel else numOuts, int(parallel), maxLength, int(bool(zero)),
                                                               A, B, C_, D, devs, len(devices), maxBlock), "Convolver (sharded)")
        elif custom is not None:
            # MI355X extension: custom partitioning / capacity (maxLength, zeroLatency, A, B, C, D) and, with tailRatio,
            # the extended far-tail ladder (hcv_convolver_create_extended)
            maxLength, zero, A, B, C_, D = custom
            parallel = numOuts is None
            self.h = _need(self.L.hcv_convolver_create_extended(numIns, numIns if parallel else numOuts, int(parallel), maxLength,
                                                                int(bool(zero)), A, B, C_, D, device, maxBlock, tailRatio), "Convolver")
        elif numOuts is None:
            self.h = _need(self.L.hcv_convolver_create_parallel(numIns, int(latency)), "Convolver")
        elif device >= 0 or maxBlock:
            self.h = _need(self.L.hcv_convolver_create_on(numIns, numOuts, int(latency), device, maxBlock), "Convolver")
        else:
            self.h = _need(self.L.hcv_convolver_create(numIns, numOuts, int(latency)), "Convolver")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.hcv_convolver_destroy(self.h)
            self.h = None

    def clear(self, *args):
        if len(args) == 1:
            self.L.hcv_convolver_clear(self.h, int(args[0]))
        else:
            self.L.hcv_convolver_clear_chan(self.h, args[0], args[1], int(args[2]))

    def reset(self, *args):
        if not args:
            self.L.hcv_convolver_reset(self.h)
            return None
        return self.L.hcv_convolver_reset_chan(self.h, args[0] & 0xFFFFFFFF, args[1] & 0xFFFFFFFF)

    def resize(self, inChan, outChan, length):
        return self.L.hcv_convolver_resize(self.h, inChan & 0xFFFFFFFF, outChan & 0xFFFFFFFF, length)

    def set(self, inChan, outChan, input, resize, length=None):
        if input is None:
            return self.L.hcv_convolver_set_f32(self.h, inChan, outChan, None, 0 if length is None else length, int(resize))
        ir = np.ascontiguousarray(input)
        if ir.dtype == np.float64:
            return self.L.hcv_convolver_set_f64(self.h, inChan, outChan, ir.ctypes.data_as(f64p), ir.size if length is None else length, int(resize))
        ir = _f32(ir)
        return self.L.hcv_convolver_set_f32(self.h, inChan, outChan, _fp(ir), ir.size if length is None else length, int(resize))

    @staticmethod
    def _rows(a, what):
        """Validate a [channels][n] block handed to the C ABI as raw row pointers: float32 or float64, 2-D, rows contiguous."""
        if not isinstance(a, np.ndarray) or a.ndim != 2:
            raise ValueError(f"Convolver.process: {what} must be a 2-D numpy array [channels][samples]")
        if a.dtype not in (np.float32, np.float64):
            raise TypeError(f"Convolver.process: {what} must be float32 or float64, not {a.dtype}")
        if a.shape[1] > 1 and a.strides[1] != a.itemsize:
            raise ValueError(f"Convolver.process: the rows of {what} must be contiguous (stride {a.strides[1]} bytes between samples)")
        return a

    def process(self, ins, outs, numIns=None, numOuts=None):
        """ins: [numIns][n], outs: [numOuts][n] written in place; both float32 or both float64 (Convolver.cpp:138-183).
        numIns / numOuts beyond the arrays' row counts are clamped: the C side reads that many row pointers."""
        ins, outs = self._rows(ins, "ins"), self._rows(outs, "outs")
        if ins.dtype != outs.dtype:
            raise TypeError(f"Convolver.process: ins ({ins.dtype}) and outs ({outs.dtype}) must share a dtype")
        if outs.shape[1] < ins.shape[1]:
            raise ValueError("Convolver.process: outs holds fewer samples per row than ins")
        if not outs.flags.writeable:
            raise ValueError("Convolver.process: outs is read-only")
        n = ins.shape[1]
        ni = ins.shape[0] if numIns is None else min(int(numIns), ins.shape[0])
        no = outs.shape[0] if numOuts is None else min(int(numOuts), outs.shape[0])
        if ins.dtype == np.float64:
            rc = self.L.hcv_convolver_process_f64(self.h, _ptr_array(list(ins), f64p), _ptr_array(list(outs), f64p), ni, no, n)
        else:
            rc = self.L.hcv_convolver_process_f32(self.h, _ptr_array(list(ins), f32p), _ptr_array(list(outs), f32p), ni, no, n)
        _check(rc, "Convolver.process")

    def run(self, ins, numOuts, block=512):
        ins = np.ascontiguousarray(ins)
        if ins.dtype != np.float64:
            ins = ins.astype(np.float32, copy=False)
        if ins.ndim != 2:
            raise ValueError("Convolver.run: ins must be [channels][samples]")
        nin, total = ins.shape
        outs = np.zeros((numOuts, total), ins.dtype)
        dbl = ins.dtype == np.float64
        pt = f64p if dbl else f32p
        fn = self.L.hcv_convolver_process_f64 if dbl else self.L.hcv_convolver_process_f32
        for pos, n in _blocks(total, block):
            i_rows = [ins[i, pos:pos + n] for i in range(nin)]
            o_rows = [outs[o, pos:pos + n] for o in range(numOuts)]
            _check(fn(self.h, _ptr_array(i_rows, pt), _ptr_array(o_rows, pt), nin, numOuts, n), "process")
        return outs

    # ---- MI355X extensions (HBM-resident data, profiling) ----

    def set_dev(self, inChan, outChan, dev_ptr: int, length: int, resize=True):
        return self.L.hcv_convolver_set_f32_dev(self.h, inChan, outChan, dev_ptr, length, int(resize))

    def process_dev(self, ins_ptr: int, in_stride: int, outs_ptr: int, out_stride: int, numIns: int, numOuts: int, n: int, sync=False):
        _check(self.L.hcv_convolver_process_f32_dev(self.h, ins_ptr, in_stride, outs_ptr, out_stride, numIns, numOuts, n, int(sync)),
               "Convolver.process_dev")

    def synchronize(self):
        _check(self.L.hcv_convolver_synchronize(self.h), "Convolver.synchronize")

    def num_shards(self) -> int:
        return self.L.hcv_convolver_num_shards(self.h)

    def comm_init(self, unique_id: bytes, rank: int, nranks: int):
        """One process per GPU: join this object to its row group's RCCL communicator (hcv_convolver_comm_init).  `unique_id` =
        the 128 bytes rccl_unique_id() returned on one rank of the group, distributed by the caller."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _check(self.L.hcv_convolver_comm_init(self.h, buf, rank, nranks), "Convolver.comm_init")

    def process_dev_allreduce(self, ins_ptr: int, in_stride: int, outs_ptr: int, out_stride: int, numIns: int, numOuts: int, n: int, sync=False):
        """process_dev + ONE in-place ncclAllReduce(sum) of the output block over the row group, on the engine's stream."""
        _check(self.L.hcv_convolver_process_f32_dev_allreduce(self.h, ins_ptr, in_stride, outs_ptr, out_stride, numIns, numOuts, n, int(sync)),
               "Convolver.process_dev_allreduce")

    def rt_stats(self):
        """Audio-thread contract counters since the last clear_stats(): {start_collisions, mailbox_runs, mailbox_ns_max, mailbox_ns_total,
        ctl_sections} (include/hisstools_amd.h: hcv_rt_stats)"""
        st = _lib.RtStats()
        _check(self.L.hcv_convolver_rt_stats(self.h, C.byref(st)), "Convolver.rt_stats")
        return {k: getattr(st, k) for k, _ in _lib.RtStats._fields_}

    def device(self) -> int:
        return self.L.hcv_convolver_device(self.h)

    def set_profiling(self, on: bool):
        self.L.hcv_convolver_set_profiling(self.h, int(on))

    def clear_stats(self):
        self.L.hcv_convolver_clear_stats(self.h)

    def stage_stats(self):
        out = []
        for s in range(self.L.hcv_convolver_num_stages(self.h)):
            st = _lib.StageStats()
            if self.L.hcv_convolver_stage_stats(self.h, s, C.byref(st)) == 0:
                out.append({k: getattr(st, k) for k, _ in _lib.StageStats._fields_})
        return out
