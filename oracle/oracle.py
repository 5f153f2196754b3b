"""TEST INFRASTRUCTURE — ctypes bindings for the CPU oracle.

Two back ends with the same Python surface (mirroring the reference classes):

* ``port`` — oracle/libhcv_oracle.so, the plain-C restatement (hcv_oracle.c).  Travels in-tree.
* ``ref``  — oracle/_ref/libhisstools_ref.so, the *unmodified* reference compiled from
  /root/reference by oracle/Makefile.  Exists wherever it was prebuilt (it is git-ignored but
  travels to the GPU box with the snapshot).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (hisstools_library_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PORT_PATH = os.path.join(_HERE, "libhcv_oracle.so")
REF_PATH = os.path.join(_HERE, "_ref", "libhisstools_ref.so")
REF_WIDE_PATH = os.path.join(_HERE, "_ref", "libhisstools_ref_wide.so")      # the same sources under -O3 -mavx2 -mfma (CPU-baseline leg only)
REF_ROOT = "/root/reference"

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_sz = C.c_size_t
_ip = C.c_ssize_t


def build(ref: bool = True) -> None:
    """Compile the oracle (and the reference build where /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    if ref and os.path.isdir(REF_ROOT):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def have_ref() -> bool:
    return os.path.exists(REF_PATH)


def have_ref_wide() -> bool:
    return os.path.exists(REF_WIDE_PATH)


def _fp(a: np.ndarray):
    return a.ctypes.data_as(_f32p)


def _dp(a: np.ndarray):
    return a.ctypes.data_as(_f64p)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


_libs: dict = {}


def _decl(lib, name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


def lib(backend: str):
    if backend in _libs:
        return _libs[backend]
    if backend == "port":
        if not os.path.exists(PORT_PATH):
            build(ref=False)
        L = C.CDLL(PORT_PATH)
        s = "_f32"
        n = {k: "hcvo_" + k + s for k in (
            "part_new part_delete part_set_fft_size part_set_length part_set_offset part_set_reset_offset "
            "part_set part_reset part_process td_new td_delete td_set_length td_set_offset td_set td_reset "
            "td_process mono_new mono_new_custom mono_delete mono_set_reset_offset mono_resize mono_set "
            "mono_reset mono_process n2m_new n2m_delete n2m_resize n2m_set n2m_reset n2m_process conv_new "
            "conv_new_parallel conv_delete conv_clear conv_clear_chan conv_reset conv_reset_chan conv_resize "
            "conv_stream").split()}
        n.update(conv_set_f32="hcvo_conv_set_f32", conv_set_f64="hcvo_conv_set_f64",
                 conv_process_f32="hcvo_conv_process_f32", conv_process_f64="hcvo_conv_process_f64",
                 rfft="hcvo_rfft_f32", rifft="hcvo_rifft_f32", fft="hcvo_fft_f32")
    elif backend in ("ref", "ref_wide"):
        path = REF_PATH if backend == "ref" else REF_WIDE_PATH
        if not os.path.exists(path):
            if os.path.isdir(REF_ROOT):
                build(ref=True)
            else:
                raise FileNotFoundError(path + " (reference build) is not available here")
        L = C.CDLL(path)
        n = {k: "ref_" + k for k in (
            "part_new part_delete part_set_fft_size part_set_length part_set_offset part_set_reset_offset "
            "part_set part_reset part_process td_new td_delete td_set_length td_set_offset td_set td_reset "
            "td_process mono_new mono_new_custom mono_delete mono_set_reset_offset mono_resize mono_set "
            "mono_reset mono_process n2m_new n2m_delete n2m_resize n2m_set n2m_reset n2m_process conv_new "
            "conv_new_parallel conv_delete conv_clear conv_clear_chan conv_reset conv_reset_chan conv_resize "
            "conv_set_f32 conv_set_f64 conv_process_f32 conv_process_f64").split()}
        n.update(conv_stream="ref_conv_stream_f32", rfft="ref_rfft_f32", rifft="ref_rifft_f32", fft="ref_fft_f32")
    else:
        raise ValueError(backend)

    vp = C.c_void_p
    f = {}
    f["part_new"] = _decl(L, n["part_new"], vp, _sz, _sz, _sz, _sz)
    f["part_delete"] = _decl(L, n["part_delete"], None, vp)
    f["part_set_fft_size"] = _decl(L, n["part_set_fft_size"], C.c_int, vp, _sz)
    f["part_set_length"] = _decl(L, n["part_set_length"], C.c_int, vp, _sz)
    f["part_set_offset"] = _decl(L, n["part_set_offset"], None, vp, _sz)
    f["part_set_reset_offset"] = _decl(L, n["part_set_reset_offset"], None, vp, _ip)
    f["part_set"] = _decl(L, n["part_set"], C.c_int, vp, _f32p, _sz)
    f["part_reset"] = _decl(L, n["part_reset"], None, vp)
    f["part_process"] = _decl(L, n["part_process"], C.c_int, vp, _f32p, _f32p, _sz)
    f["td_new"] = _decl(L, n["td_new"], vp, _sz, _sz)
    f["td_delete"] = _decl(L, n["td_delete"], None, vp)
    f["td_set_length"] = _decl(L, n["td_set_length"], C.c_int, vp, _sz)
    f["td_set_offset"] = _decl(L, n["td_set_offset"], None, vp, _sz)
    f["td_set"] = _decl(L, n["td_set"], C.c_int, vp, _f32p, _sz)
    f["td_reset"] = _decl(L, n["td_reset"], None, vp)
    f["td_process"] = _decl(L, n["td_process"], C.c_int, vp, _f32p, _f32p, _sz)
    f["mono_new"] = _decl(L, n["mono_new"], vp, _sz, C.c_int)
    if backend == "port":
        f["mono_new_custom"] = _decl(L, n["mono_new_custom"], vp, _sz, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_char_p))
    else:
        f["mono_new_custom"] = _decl(L, n["mono_new_custom"], vp, _sz, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_char_p, _sz)
    f["mono_delete"] = _decl(L, n["mono_delete"], None, vp)
    f["mono_set_reset_offset"] = _decl(L, n["mono_set_reset_offset"], None, vp, _ip)
    f["mono_resize"] = _decl(L, n["mono_resize"], C.c_int, vp, _sz)
    f["mono_set"] = _decl(L, n["mono_set"], C.c_int, vp, _f32p, _sz, C.c_int)
    f["mono_reset"] = _decl(L, n["mono_reset"], C.c_int, vp)
    f["mono_process"] = _decl(L, n["mono_process"], None, vp, _f32p, _f32p, _f32p, _sz, C.c_int)
    f["n2m_new"] = _decl(L, n["n2m_new"], vp, C.c_uint32, _sz, C.c_int)
    f["n2m_delete"] = _decl(L, n["n2m_delete"], None, vp)
    f["n2m_resize"] = _decl(L, n["n2m_resize"], C.c_int, vp, C.c_uint32, _sz)
    f["n2m_set"] = _decl(L, n["n2m_set"], C.c_int, vp, C.c_uint32, _f32p, _sz, C.c_int)
    f["n2m_reset"] = _decl(L, n["n2m_reset"], C.c_int, vp, C.c_uint32)
    f["n2m_process"] = _decl(L, n["n2m_process"], None, vp, C.POINTER(_f32p), _f32p, _f32p, _sz, _sz)
    f["conv_new"] = _decl(L, n["conv_new"], vp, C.c_uint32, C.c_uint32, C.c_int)
    f["conv_new_parallel"] = _decl(L, n["conv_new_parallel"], vp, C.c_uint32, C.c_int)
    f["conv_delete"] = _decl(L, n["conv_delete"], None, vp)
    f["conv_clear"] = _decl(L, n["conv_clear"], None, vp, C.c_int)
    f["conv_clear_chan"] = _decl(L, n["conv_clear_chan"], None, vp, C.c_uint32, C.c_uint32, C.c_int)
    f["conv_reset"] = _decl(L, n["conv_reset"], None, vp)
    f["conv_reset_chan"] = _decl(L, n["conv_reset_chan"], C.c_int, vp, C.c_uint32, C.c_uint32)
    f["conv_resize"] = _decl(L, n["conv_resize"], C.c_int, vp, C.c_uint32, C.c_uint32, _sz)
    f["conv_set_f32"] = _decl(L, n["conv_set_f32"], C.c_int, vp, C.c_uint32, C.c_uint32, _f32p, _sz, C.c_int)
    f["conv_set_f64"] = _decl(L, n["conv_set_f64"], C.c_int, vp, C.c_uint32, C.c_uint32, _f64p, _sz, C.c_int)
    f["conv_process_f32"] = _decl(L, n["conv_process_f32"], None, vp, C.POINTER(_f32p), C.POINTER(_f32p), _sz, _sz, _sz)
    f["conv_process_f64"] = _decl(L, n["conv_process_f64"], None, vp, C.POINTER(_f64p), C.POINTER(_f64p), _sz, _sz, _sz)
    f["conv_stream"] = _decl(L, n["conv_stream"], C.c_double, vp, _f32p, _f32p, _sz, _sz, _sz, _sz)
    if backend == "port":
        f["rfft"] = _decl(L, n["rfft"], None, _f32p, _sz, C.c_uint, _f32p, _f32p)
        f["rifft"] = _decl(L, n["rifft"], None, _f32p, _f32p, C.c_uint, _f32p)
        f["fft"] = _decl(L, n["fft"], None, _f32p, _f32p, C.c_uint, C.c_int)
        f["synth_audio"] = _decl(L, "hcvo_synth_audio_f32", None, C.c_uint32, _f32p, _sz)
        f["synth_ir"] = _decl(L, "hcvo_synth_ir_f32", None, C.c_uint32, C.c_uint32, _f32p, _sz)
        f["conv_set_reset_offset"] = _decl(L, "hcvo_conv_set_reset_offset_f32", None, vp, _ip)
        f["n2m_set_reset_offset"] = _decl(L, "hcvo_n2m_set_reset_offset_f32", None, vp, _ip)
        f["rfft_f64"] = _decl(L, "hcvo_rfft_f64", None, _f64p, _sz, C.c_uint, _f64p, _f64p)
        f["rifft_f64"] = _decl(L, "hcvo_rifft_f64", None, _f64p, _f64p, C.c_uint, _f64p)
    else:
        f["rfft"] = _decl(L, n["rfft"], None, _f32p, _sz, _sz, _f32p, _f32p)
        f["rifft"] = _decl(L, n["rifft"], None, _f32p, _f32p, _sz, _f32p)
        f["fft"] = _decl(L, n["fft"], None, _f32p, _f32p, _sz, C.c_int)
        f["part_stream"] = _decl(L, "ref_part_stream_f32", C.c_double, vp, _f32p, _f32p, _sz, _sz)
        f["mono_stream"] = _decl(L, "ref_mono_stream_f32", C.c_double, vp, _f32p, _f32p, _sz, _sz)
        f["unzip"] = _decl(L, "ref_unzip_f32", None, _f32p, _f32p, _f32p, _sz)
        f["zip"] = _decl(L, "ref_zip_f32", None, _f32p, _f32p, _f32p, _sz)
        f["unzip_zero"] = _decl(L, "ref_unzip_zero_f32", None, _f32p, _f32p, _f32p, _sz, _sz)
    ns = type("OracleLib", (), f)
    ns.backend = backend
    ns.cdll = L
    _libs[backend] = ns
    return ns


# --------------------------------------------------------------------------------------- FFT

def rfft(x, log2n: int, backend: str = "port"):
    """hisstools_rfft 5-arg: returns (realp, imagp) of length 2^(log2n-1)."""
    L = lib(backend)
    x = _f32(x)
    half = 1 << (log2n - 1)
    re = np.zeros(half, np.float32)
    im = np.zeros(half, np.float32)
    L.rfft(_fp(x), x.size, log2n, _fp(re), _fp(im))
    return re, im


def rifft(re, im, log2n: int, backend: str = "port"):
    """hisstools_rifft 4-arg: returns 2^log2n real samples (unnormalised)."""
    L = lib(backend)
    re, im = _f32(re).copy(), _f32(im).copy()
    out = np.zeros(1 << log2n, np.float32)
    L.rifft(_fp(re), _fp(im), log2n, _fp(out))
    return out


def fft(re, im, log2n: int, inverse: bool = False, backend: str = "port"):
    L = lib(backend)
    re, im = _f32(re).copy(), _f32(im).copy()
    L.fft(_fp(re), _fp(im), log2n, int(inverse))
    return re, im


# --------------------------------------------------------------------------------------- the full hisstools_* FFT surface

FFT_OPS = ("fft", "ifft", "rfft", "rifft", "rfft_zip", "rifft_zip", "unzip", "zip")
FFT_PRECISIONS = ("f32", "f64", "f32_to_f64")
_surface = {}


def _surface_lib(backend):
    """Bind the FFT-surface entry points of the port / the compiled reference (same call shapes for both)."""
    if backend in _surface:
        return _surface[backend]
    L = lib(backend).cdll
    u = C.c_uint if backend == "port" else _sz
    f = {}
    for sfx, fp in (("f32", _f32p), ("f64", _f64p)):
        if backend == "port":
            f["fft_" + sfx] = _decl(L, "hcvo_fft_" + sfx, None, fp, fp, u, C.c_int)
            rf = _decl(L, "hcvo_rfft_inplace_" + sfx, None, fp, fp, u)
            ri = _decl(L, "hcvo_rifft_inplace_" + sfx, None, fp, fp, u)
            f["rfft_inplace_" + sfx] = (lambda rf, ri: (lambda re, im, l2, inv: (ri if inv else rf)(re, im, l2)))(rf, ri)
            f["rfft_" + sfx] = _decl(L, "hcvo_rfft_" + sfx, None, fp, _sz, u, fp, fp)
            f["rifft_" + sfx] = _decl(L, "hcvo_rifft_" + sfx, None, fp, fp, u, fp)
            f["unzip_zero_" + sfx] = _decl(L, "hcvo_unzip_zero_" + sfx, None, fp, fp, fp, _sz, u)
            f["unzip_" + sfx] = _decl(L, "hcvo_unzip_" + sfx, None, fp, fp, fp, u)
            f["zip_" + sfx] = _decl(L, "hcvo_zip_" + sfx, None, fp, fp, fp, u)
        else:
            f["fft_" + sfx] = _decl(L, "ref_fft_" + sfx, None, fp, fp, u, C.c_int)
            f["rfft_inplace_" + sfx] = _decl(L, "ref_rfft_inplace_" + sfx, None, fp, fp, u, C.c_int)
            f["rfft_" + sfx] = _decl(L, "ref_rfft_" + sfx, None, fp, _sz, u, fp, fp)
            f["rifft_" + sfx] = _decl(L, "ref_rifft_" + sfx, None, fp, fp, u, fp)
            f["unzip_zero_" + sfx] = _decl(L, "ref_unzip_zero_" + sfx, None, fp, fp, fp, _sz, u)
            f["unzip_" + sfx] = _decl(L, "ref_unzip_" + sfx, None, fp, fp, fp, u)
            f["zip_" + sfx] = _decl(L, "ref_zip_" + sfx, None, fp, fp, fp, u)
    pre = "hcvo_" if backend == "port" else "ref_"
    f["rfft_f32_to_f64"] = _decl(L, pre + "rfft_f32_f64", None, _f32p, _sz, u, _f64p, _f64p)
    f["unzip_zero_f32_to_f64"] = _decl(L, pre + "unzip_zero_f32_f64", None, _f32p, _f64p, _f64p, _sz, u)
    _surface[backend] = f
    return f


def fft_surface(op: str, precision: str, log2n: int, a, b=None, in_length=None, backend: str = "port"):
    """One transform of the hisstools_* surface (HISSTools_FFT.h:87-369) on the CPU oracle.

    op / operands / result:
      fft, ifft          (realp, imagp) of 2^log2n values        -> (realp, imagp)
      rfft, rifft        (realp, imagp) of 2^(log2n-1) values    -> (realp, imagp)      in-place real transforms
      rfft_zip           samples a (in_length of them)           -> (realp, imagp)
      rifft_zip          (realp, imagp)                          -> 2^log2n samples
      unzip              samples a (in_length, default 2^log2n)  -> (realp, imagp)      unzip / unzip_zero
      zip                (realp, imagp)                          -> samples
    precision: "f32", "f64" or "f32_to_f64" (float samples in, double split out; rfft_zip and unzip only).
    """
    f = _surface_lib(backend)
    n = 1 << log2n
    half = n >> 1
    dt = np.float32 if precision == "f32" else np.float64
    sfx = "f32" if precision == "f32" else "f64"
    ptr = (lambda v: v.ctypes.data_as(_f32p)) if dt == np.float32 else (lambda v: v.ctypes.data_as(_f64p))
    if op in ("fft", "ifft"):
        re, im = np.array(a, dt).copy(), np.array(b, dt).copy()
        if log2n >= 1:
            f["fft_" + sfx](ptr(re), ptr(im), log2n, int(op == "ifft"))
        return re, im
    if op in ("rfft", "rifft"):
        re, im = np.array(a, dt).copy(), np.array(b, dt).copy()
        if log2n >= 1:
            f["rfft_inplace_" + sfx](ptr(re), ptr(im), log2n, int(op == "rifft"))
        return re, im
    if op in ("rfft_zip", "unzip"):
        src_dt = np.float32 if precision in ("f32", "f32_to_f64") else np.float64
        x = np.ascontiguousarray(a, src_dt)
        in_length = x.size if in_length is None else in_length
        re, im = np.zeros(max(half, 1), dt), np.zeros(max(half, 1), dt)
        xp = x.ctypes.data_as(_f32p if src_dt == np.float32 else _f64p)
        if half and in_length:
            if op == "rfft_zip":
                name = "rfft_f32_to_f64" if precision == "f32_to_f64" else "rfft_" + sfx
                f[name](xp, in_length, log2n, ptr(re), ptr(im))
            else:
                name = "unzip_zero_f32_to_f64" if precision == "f32_to_f64" else "unzip_zero_" + sfx
                f[name](xp, ptr(re), ptr(im), in_length, log2n)
        return re[:half], im[:half]
    if op in ("rifft_zip", "zip"):
        re, im = np.array(a, dt).copy(), np.array(b, dt).copy()
        out = np.zeros(max(n, 2), dt)
        if half:
            if op == "zip":
                f["zip_" + sfx](ptr(re), ptr(im), ptr(out), log2n)
            else:
                f["rifft_" + sfx](ptr(re), ptr(im), log2n, ptr(out))
        return out[:n] if half else out[:0]
    raise ValueError(op)


# --------------------------------------------------------------------------------------- synthetic data

def synth_audio(ch: int, n: int) -> np.ndarray:
    x = np.zeros(n, np.float32)
    lib("port").synth_audio(ch, _fp(x), n)
    return x


def synth_ir(i: int, o: int, length: int) -> np.ndarray:
    h = np.zeros(length, np.float32)
    lib("port").synth_ir(i, o, _fp(h), length)
    return h


# --------------------------------------------------------------------------------------- classes

def _blocks(total: int, block):
    """Yield (pos, n) for a fixed block size or a cyclic list of ragged sizes."""
    pos, k = 0, 0
    sizes = [block] if isinstance(block, int) else list(block)
    while pos < total:
        n = min(sizes[k % len(sizes)], total - pos)
        yield pos, n
        pos += n
        k += 1


class PartitionedConvolve:
    def __init__(self, maxFFTSize, maxLength, offset, length, backend="port"):
        self.L = lib(backend)
        self.h = self.L.part_new(maxFFTSize, maxLength, offset, length)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.part_delete(self.h)
            self.h = None

    def setFFTSize(self, n): return self.L.part_set_fft_size(self.h, n)
    def setLength(self, n): return self.L.part_set_length(self.h, n)
    def setOffset(self, n): self.L.part_set_offset(self.h, n)
    def setResetOffset(self, n=-1): self.L.part_set_reset_offset(self.h, n)
    def reset(self): self.L.part_reset(self.h)

    def set(self, ir, length=None):
        if ir is None:
            return self.L.part_set(self.h, None, 0 if length is None else length)
        ir = _f32(ir)
        return self.L.part_set(self.h, _fp(ir), ir.size if length is None else length)

    def process(self, x, out=None):
        x = _f32(x)
        if out is None:
            out = np.full(x.size, np.nan, np.float32)
        wrote = self.L.part_process(self.h, _fp(x), _fp(out), x.size)
        return bool(wrote), out

    def run(self, x, block=512):
        x = _f32(x)
        y = np.zeros_like(x)
        for pos, n in _blocks(x.size, block):
            self.L.part_process(self.h, _fp(x[pos:pos + n]), _fp(y[pos:pos + n]), n)
        return y


class TimeDomainConvolve:
    def __init__(self, offset, length, backend="port"):
        self.L = lib(backend)
        self.h = self.L.td_new(offset, length)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.td_delete(self.h)
            self.h = None

    def setLength(self, n): return self.L.td_set_length(self.h, n)
    def setOffset(self, n): self.L.td_set_offset(self.h, n)
    def reset(self): self.L.td_reset(self.h)

    def set(self, ir, length=None):
        if ir is None:
            return self.L.td_set(self.h, None, 0 if length is None else length)
        ir = _f32(ir)
        return self.L.td_set(self.h, _fp(ir), ir.size if length is None else length)

    def process(self, x, out=None):
        x = _f32(x)
        if out is None:
            out = np.full(x.size, np.nan, np.float32)
        wrote = self.L.td_process(self.h, _fp(x), _fp(out), x.size)
        return bool(wrote), out

    def run(self, x, block=512):
        x = _f32(x)
        y = np.zeros_like(x)
        for pos, n in _blocks(x.size, block):
            self.L.td_process(self.h, _fp(x[pos:pos + n]), _fp(y[pos:pos + n]), n)
        return y


class MonoConvolve:
    def __init__(self, maxLength, latency=None, zeroLatency=None, A=0, B=0, C_=0, D=0, backend="port"):
        self.L = lib(backend)
        self.error = None
        if latency is not None:
            self.h = self.L.mono_new(maxLength, int(latency))
        elif backend == "port":
            err = C.c_char_p()
            self.h = self.L.mono_new_custom(maxLength, int(bool(zeroLatency)), A, B, C_, D, C.byref(err))
            if not self.h:
                self.error = err.value.decode()
        else:
            buf = C.create_string_buffer(128)
            self.h = self.L.mono_new_custom(maxLength, int(bool(zeroLatency)), A, B, C_, D, buf, 128)
            if not self.h:
                self.error = buf.value.decode()
        if not self.h:
            raise RuntimeError(self.error or "construction failed")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.mono_delete(self.h)
            self.h = None

    def setResetOffset(self, n=-1): self.L.mono_set_reset_offset(self.h, n)
    def resize(self, n): return self.L.mono_resize(self.h, n)
    def reset(self): return self.L.mono_reset(self.h)

    def set(self, ir, requestResize, length=None):
        if ir is None:
            return self.L.mono_set(self.h, None, 0 if length is None else length, int(requestResize))
        ir = _f32(ir)
        return self.L.mono_set(self.h, _fp(ir), ir.size if length is None else length, int(requestResize))

    def process(self, x, out=None, accumulate=False):
        x = _f32(x)
        if out is None:
            out = np.full(x.size, np.nan, np.float32)
        temp = np.zeros(x.size, np.float32)
        self.L.mono_process(self.h, _fp(x), _fp(temp), _fp(out), x.size, int(accumulate))
        return out

    def run(self, x, block=512):
        x = _f32(x)
        y = np.zeros_like(x)
        temp = np.zeros(x.size, np.float32)
        for pos, n in _blocks(x.size, block):
            self.L.mono_process(self.h, _fp(x[pos:pos + n]), _fp(temp), _fp(y[pos:pos + n]), n, 0)
        return y


def _ptr_array(rows, ptr_t):
    arr = (ptr_t * len(rows))()
    for i, r in enumerate(rows):
        arr[i] = r.ctypes.data_as(ptr_t)
    return arr


class NToMonoConvolve:
    def __init__(self, inChans, maxLength, latency, backend="port"):
        self.L = lib(backend)
        self.h = self.L.n2m_new(inChans, maxLength, int(latency))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.n2m_delete(self.h)
            self.h = None

    def resize(self, inChan, n): return self.L.n2m_resize(self.h, inChan, n)
    def reset(self, inChan): return self.L.n2m_reset(self.h, inChan)

    def setResetOffset(self, n):          # port only
        self.L.n2m_set_reset_offset(self.h, n)

    def set(self, inChan, ir, resize, length=None):
        if ir is None:
            return self.L.n2m_set(self.h, inChan, None, 0 if length is None else length, int(resize))
        ir = _f32(ir)
        return self.L.n2m_set(self.h, inChan, _fp(ir), ir.size if length is None else length, int(resize))

    def run(self, ins, block=512, activeIns=None):
        ins = _f32(ins)
        nin, total = ins.shape
        y = np.zeros(total, np.float32)
        temp = np.zeros(total, np.float32)
        act = nin if activeIns is None else activeIns
        for pos, n in _blocks(total, block):
            rows = [ins[i, pos:pos + n] for i in range(nin)]
            self.L.n2m_process(self.h, _ptr_array(rows, _f32p), _fp(y[pos:pos + n]), _fp(temp), n, act)
        return y


class Convolver:
    def __init__(self, numIns, numOuts=None, latency=0, backend="port"):
        self.L = lib(backend)
        if numOuts is None:
            self.h = self.L.conv_new_parallel(numIns, int(latency))
        else:
            self.h = self.L.conv_new(numIns, numOuts, int(latency))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.conv_delete(self.h)
            self.h = None

    def clear(self, *args):
        if len(args) == 1:
            self.L.conv_clear(self.h, int(args[0]))
        else:
            self.L.conv_clear_chan(self.h, args[0], args[1], int(args[2]))

    def reset(self, *args):
        if not args:
            self.L.conv_reset(self.h)
            return None
        return self.L.conv_reset_chan(self.h, args[0] & 0xFFFFFFFF, args[1] & 0xFFFFFFFF)

    def resize(self, inChan, outChan, n): return self.L.conv_resize(self.h, inChan & 0xFFFFFFFF, outChan & 0xFFFFFFFF, n)

    def setResetOffset(self, n):          # port only
        self.L.conv_set_reset_offset(self.h, n)

    def set(self, inChan, outChan, ir, resize, length=None):
        if ir is None:
            return self.L.conv_set_f32(self.h, inChan, outChan, None, 0 if length is None else length, int(resize))
        ir = np.ascontiguousarray(ir)
        if ir.dtype == np.float64:
            return self.L.conv_set_f64(self.h, inChan, outChan, _dp(ir), ir.size if length is None else length, int(resize))
        ir = _f32(ir)
        return self.L.conv_set_f32(self.h, inChan, outChan, _fp(ir), ir.size if length is None else length, int(resize))

    def process(self, ins, outs, numIns=None, numOuts=None):
        """ins: [numIns][n], outs: [numOuts][n] (written in place); dtype float32 or float64."""
        n = ins.shape[1]
        ni = ins.shape[0] if numIns is None else numIns
        no = outs.shape[0] if numOuts is None else numOuts
        if ins.dtype == np.float64:
            self.L.conv_process_f64(self.h, _ptr_array(list(ins), _f64p), _ptr_array(list(outs), _f64p), ni, no, n)
        else:
            self.L.conv_process_f32(self.h, _ptr_array(list(ins), _f32p), _ptr_array(list(outs), _f32p), ni, no, n)

    def run(self, ins, numOuts, block=512):
        ins = np.ascontiguousarray(ins)
        nin, total = ins.shape
        outs = np.zeros((numOuts, total), ins.dtype)
        for pos, n in _blocks(total, block):
            i_rows = [ins[i, pos:pos + n] for i in range(nin)]
            o_rows = [outs[o, pos:pos + n] for o in range(numOuts)]
            pt = _f64p if ins.dtype == np.float64 else _f32p
            fn = self.L.conv_process_f64 if ins.dtype == np.float64 else self.L.conv_process_f32
            fn(self.h, _ptr_array(i_rows, pt), _ptr_array(o_rows, pt), nin, numOuts, n)
        return outs

    def stream_timed(self, ins, numOuts, block=512):
        """C-side streaming loop; returns (outs, seconds)."""
        ins = _f32(ins)
        nin, total = ins.shape
        outs = np.zeros((numOuts, total), np.float32)
        secs = self.L.conv_stream(self.h, _fp(ins), _fp(outs), nin, numOuts, total, block)
        return outs, secs


# --------------------------------------------------------------------------------------- spectral_processor (next row)

REF_SPECTRAL_PATH = os.path.join(_HERE, "_ref", "libhisstools_ref_spectral.so")
_spectral = {}


def have_ref_spectral() -> bool:
    return os.path.exists(REF_SPECTRAL_PATH)


def _spectral_lib(backend):
    """dict of callables keyed by (op, complex, dtype): op in {"convolve", "correlate"}; plus "size" """
    if backend in _spectral:
        return _spectral[backend]
    fns = {}
    if backend == "port":
        L = lib("port").cdll
        size = _decl(L, "hcvo_spectral_size", _sz, _sz, _sz, C.c_int, _sz)
        fns["size"] = lambda n1, n2, mode: size(n1, n2, mode, 1 << 24)
        for op in ("convolve", "correlate"):
            for suf, pt in (("f32", _f32p), ("f64", _f64p)):
                f = _decl(L, f"hcvo_spectral_{op}_{suf}", None, pt, _sz, pt, _sz, C.c_int, pt, _sz)
                fns[(op, False, suf)] = lambda *a, f=f: f(*a, 1 << 24)
                g = _decl(L, f"hcvo_spectral_{op}_complex_{suf}", None, pt, _sz, pt, _sz, pt, _sz, pt, _sz, C.c_int, pt, pt, _sz)
                fns[(op, True, suf)] = lambda *a, g=g: g(*a, 1 << 24)
    else:
        if not have_ref_spectral():
            raise RuntimeError("oracle/_ref/libhisstools_ref_spectral.so missing: run `make -C oracle ref` where /root/reference exists")
        L = C.CDLL(REF_SPECTRAL_PATH)
        fns["size"] = _decl(L, "ref_spectral_size", _sz, _sz, _sz, C.c_int)
        for op in ("convolve", "correlate"):
            for suf, pt in (("f32", _f32p), ("f64", _f64p)):
                fns[(op, False, suf)] = _decl(L, f"ref_spectral_{op}_{suf}", None, pt, _sz, pt, _sz, C.c_int, pt)
                fns[(op, True, suf)] = _decl(L, f"ref_spectral_{op}_complex_{suf}", None, pt, _sz, pt, _sz, pt, _sz, pt, _sz, C.c_int, pt, pt)
    _spectral[backend] = fns
    return fns


def spectral_size(n1, n2, mode, backend="port"):
    return _spectral_lib(backend)["size"](n1, n2, int(mode))


def _spectral_typed(arrays):
    arrays = [np.ascontiguousarray(a) for a in arrays]
    dbl = any(a.dtype == np.float64 for a in arrays)
    dt = np.float64 if dbl else np.float32
    return dt, ("f64" if dbl else "f32"), (_f64p if dbl else _f32p), [np.ascontiguousarray(a, dt) for a in arrays]


def spectral_convolve(in1, in2, mode, backend="port", correlate=False):
    """spectral_processor<T>::convolve / correlate, real overloads; float32 or float64 by the inputs' dtype."""
    fns = _spectral_lib(backend)
    dt, suf, pt, (a, b) = _spectral_typed([in1, in2])
    n = fns["size"](a.size, b.size, int(mode))
    out = np.zeros(n, dt)
    if n:
        fns[("correlate" if correlate else "convolve", False, suf)](a.ctypes.data_as(pt), a.size, b.ctypes.data_as(pt), b.size, int(mode), out.ctypes.data_as(pt))
    return out


def spectral_correlate(in1, in2, mode, backend="port"):
    return spectral_convolve(in1, in2, mode, backend, correlate=True)


def spectral_convolve_complex(r1, i1, r2, i2, mode, backend="port", correlate=False):
    """the complex overloads (SpectralProcessor.hpp:164-167, 176-179); returns (r_out, i_out).  NOTE backend="ref" reads past
    its result in the two wrap modes (a reference defect, see hcv_oracle_spectral.inc)."""
    fns = _spectral_lib(backend)
    dt, suf, pt, ins = _spectral_typed([r1, i1, r2, i2])
    n = fns["size"](max(ins[0].size, ins[1].size), max(ins[2].size, ins[3].size), int(mode))
    r_out, i_out = np.zeros(n, dt), np.zeros(n, dt)
    if n:
        args = []
        for a in ins:
            args += [a.ctypes.data_as(pt), a.size]
        fns[("correlate" if correlate else "convolve", True, suf)](*args, int(mode), r_out.ctypes.data_as(pt), i_out.ctypes.data_as(pt))
    return r_out, i_out


# --------------------------------------------------------------------------------------- spectral IR functions (fourth "next" row)

IR_OPS = ("copy", "spike", "delay", "time_reverse", "phase")
_ir = {}


def _ir_lib(backend):
    if backend in _ir:
        return _ir[backend]
    f = {}
    if backend == "port":
        L = lib("port").cdll
        for sfx, fp in (("f32", _f32p), ("f64", _f64p)):
            f["copy_" + sfx] = _decl(L, "hcvo_ir_copy_" + sfx, None, fp, fp, fp, fp, _sz)
            f["time_reverse_" + sfx] = _decl(L, "hcvo_ir_time_reverse_" + sfx, None, fp, fp, fp, fp, _sz)
            f["spike_" + sfx] = _decl(L, "hcvo_ir_spike_" + sfx, None, fp, fp, _sz, C.c_double)
            f["delay_" + sfx] = _decl(L, "hcvo_ir_delay_" + sfx, None, fp, fp, fp, fp, _sz, C.c_double)
            f["phase_" + sfx] = _decl(L, "hcvo_ir_phase_" + sfx, None, fp, fp, fp, fp, _sz, C.c_double, C.c_int)
            f["change_phase_" + sfx] = _decl(L, "hcvo_change_phase_" + sfx, _sz, fp, _sz, C.c_double, C.c_double, fp)
            f["product_" + sfx] = _decl(L, "hcvo_ir_product_" + sfx, None, C.c_int, fp, fp, fp, fp, fp, fp, _sz, C.c_double)
    else:
        if not have_ref_spectral():
            raise FileNotFoundError(REF_SPECTRAL_PATH)
        L = C.CDLL(REF_SPECTRAL_PATH)
        for sfx, fp in (("f32", _f32p), ("f64", _f64p)):
            f["ir_" + sfx] = _decl(L, "ref_ir_" + sfx, None, C.c_int, fp, fp, fp, fp, _sz, C.c_double, C.c_int)
            f["change_phase_" + sfx] = _decl(L, "ref_change_phase_" + sfx, _sz, fp, _sz, C.c_double, C.c_double, fp)
            f["product_" + sfx] = _decl(L, "ref_ir_product_" + sfx, None, C.c_int, fp, fp, fp, fp, fp, fp, _sz, C.c_double)
    _ir[backend] = f
    return f


def ir_op(op: str, realp, imagp, fft_size: int, value: float = 0.0, zero_center: bool = False, precision: str = "f32", backend: str = "port"):
    """ir_copy / ir_spike / ir_delay / ir_time_reverse / ir_phase (SpectralFunctions.hpp:365-413) on one packed half
    spectrum of fft_size/2 values per array.  Returns (realp, imagp).  spike ignores the inputs."""
    f = _ir_lib(backend)
    dt = np.float32 if precision == "f32" else np.float64
    half = fft_size >> 1
    ptr = (lambda v: v.ctypes.data_as(_f32p)) if dt == np.float32 else (lambda v: v.ctypes.data_as(_f64p))
    ri = np.zeros(half, dt) if realp is None else np.ascontiguousarray(realp, dt).copy()
    ii = np.zeros(half, dt) if imagp is None else np.ascontiguousarray(imagp, dt).copy()
    ro, io = np.zeros(half, dt), np.zeros(half, dt)
    if backend == "ref":
        if op == "phase":                       # the reference's callers run it in place (SpectralProcessor.hpp:204)
            f["ir_" + precision](4, ptr(ri), ptr(ii), ptr(ri), ptr(ii), fft_size, value, int(zero_center))
            return ri, ii
        f["ir_" + precision](IR_OPS.index(op), ptr(ro), ptr(io), ptr(ri), ptr(ii), fft_size, value, int(zero_center))
        return ro, io
    if op == "copy":
        f["copy_" + precision](ptr(ro), ptr(io), ptr(ri), ptr(ii), fft_size)
    elif op == "time_reverse":
        f["time_reverse_" + precision](ptr(ro), ptr(io), ptr(ri), ptr(ii), fft_size)
    elif op == "spike":
        f["spike_" + precision](ptr(ro), ptr(io), fft_size, value)
    elif op == "delay":
        f["delay_" + precision](ptr(ro), ptr(io), ptr(ri), ptr(ii), fft_size, value)
    elif op == "phase":
        f["phase_" + precision](ptr(ro), ptr(io), ptr(ri), ptr(ii), fft_size, value, int(zero_center))
    else:
        raise ValueError(op)
    return ro, io


IR_PRODUCTS = ("convolve_complex", "convolve_real", "correlate_complex", "correlate_real")


def ir_product(op: str, r1, i1, r2, i2, fft_size: int, scale: float = 1.0, precision: str = "f32", backend: str = "port"):
    """ir_convolve_complex / ir_convolve_real / ir_correlate_complex / ir_correlate_real (SpectralFunctions.hpp:415-436):
    out = scale * in1 * in2 (or * conj(in2)); the complex forms on fft_size values per array, the real forms on fft_size / 2 packed
    values with bin 0 = (DC, Nyquist).  Returns (realp, imagp)."""
    f = _ir_lib(backend)
    dt = np.float32 if precision == "f32" else np.float64
    if fft_size < 1 or fft_size & (fft_size - 1):
        raise ValueError("power-of-two sizes (the reference's vector loops drop the remainder of any other size, SpectralFunctions.hpp:44)")
    n = fft_size if op.endswith("complex") else fft_size >> 1
    ptr = (lambda v: v.ctypes.data_as(_f32p)) if dt == np.float32 else (lambda v: v.ctypes.data_as(_f64p))

    def aligned(v=None):
        # (the reference reads and writes whole SIMD vectors through reinterpret_cast: 32-byte aligned arrays, :32-41)
        raw = np.zeros(n * np.dtype(dt).itemsize + 64, np.uint8)
        off = (-raw.ctypes.data) % 64
        out = raw[off:off + n * np.dtype(dt).itemsize].view(dt)
        if v is not None:
            out[:] = np.asarray(v, dt)[:n]
        return out

    a, b, c, d = (aligned(v) for v in (r1, i1, r2, i2))
    ro, io = aligned(), aligned()
    f["product_" + precision](IR_PRODUCTS.index(op), ptr(ro), ptr(io), ptr(a), ptr(b), ptr(c), ptr(d), fft_size, float(scale))
    return ro, io


def change_phase(x, phase: float, time_multiplier: float = 1.0, precision: str = "f32", backend: str = "port"):
    """spectral_processor<T>::change_phase (SpectralProcessor.hpp:188-208): returns the fft_size output samples."""
    f = _ir_lib(backend)
    dt = np.float32 if precision == "f32" else np.float64
    x = np.ascontiguousarray(x, dt)
    want = int(round(x.size * time_multiplier))
    log2n = 0
    while want >> log2n:
        log2n += 1
    if log2n and want == 1 << (log2n - 1):
        log2n -= 1
    out = np.zeros(max(1 << log2n, 1), dt)
    ptr = (lambda v: v.ctypes.data_as(_f32p)) if dt == np.float32 else (lambda v: v.ctypes.data_as(_f64p))
    n = f["change_phase_" + precision](ptr(x), x.size, phase, time_multiplier, ptr(out))
    return out[:n]


# --------------------------------------------------------------------------------------- audio files (third "next" row): the reference itself

REF_AUDIO_PATH = os.path.join(_HERE, "_ref", "libhisstools_ref_audio.so")
_audio = {}


class RefAudioInfo(C.Structure):
    _fields_ = [("file_type", C.c_int), ("pcm_format", C.c_int), ("header_endianness", C.c_int), ("audio_endianness", C.c_int),
                ("sampling_rate", C.c_double), ("channels", C.c_uint), ("frames", C.c_uint), ("bit_depth", C.c_uint), ("error_flags", C.c_int)]


def have_ref_audio() -> bool:
    return os.path.exists(REF_AUDIO_PATH)


def _audio_lib():
    if "L" not in _audio:
        L = C.CDLL(REF_AUDIO_PATH)
        _audio["write"] = _decl(L, "ref_audio_write", C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_uint, C.c_double, C.c_int, _f64p, C.c_uint, C.c_int, C.c_int)
        _audio["info"] = _decl(L, "ref_audio_info", C.c_int, C.c_char_p, C.POINTER(RefAudioInfo))
        _audio["read64"] = _decl(L, "ref_audio_read_f64", C.c_int, C.c_char_p, _f64p, C.c_uint, C.c_uint, C.c_int)
        _audio["read32"] = _decl(L, "ref_audio_read_f32", C.c_int, C.c_char_p, _f32p, C.c_uint, C.c_uint, C.c_int)
        _audio["L"] = L
    return _audio


def ref_audio_write(path, file_type, pcm_format, data, rate, endianness=-1, mode=0, as_float=False):
    """Write [frames][channels] float64 data with the reference's OAudioFile.  mode 0 interleaved, 1 per channel, 2 two calls."""
    a = np.ascontiguousarray(data, np.float64)
    frames, channels = a.shape
    return _audio_lib()["write"](str(path).encode(), int(file_type), int(pcm_format), channels, float(rate), int(endianness),
                                 a.ctypes.data_as(_f64p), frames, int(mode), int(as_float))


def ref_audio_info(path):
    info = RefAudioInfo()
    rc = _audio_lib()["info"](str(path).encode(), C.byref(info))
    return rc, {k: getattr(info, k) for k, _ in RefAudioInfo._fields_}


def ref_audio_read(path, frames, channels, first=0, channel=-1, dtype=np.float64):
    out = np.zeros((frames, channels) if channel < 0 else frames, dtype)
    fn = _audio_lib()["read64" if out.dtype == np.float64 else "read32"]
    ptr = out.ctypes.data_as(_f64p if out.dtype == np.float64 else _f32p)
    rc = fn(str(path).encode(), ptr, first, frames, channel)
    return rc, out
