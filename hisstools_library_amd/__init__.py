"""hisstools_library_amd — MI355X-native partitioned-convolution engine behind the HISSTools Convolver API.

The package holds only what the hot path needs:
  csrc/          HIP kernels (gfx950), the device engine and the C ABI (include/hisstools_amd.h)
  _lib.py        ctypes loader of the in-tree shared library (no CPU fallback)
  convolver.py   Python mirror of the reference classes over the C ABI
  fft.py, spectral_functions.py   Python mirror of the hisstools_* FFT functions and the spectral IR functions
  sharded.py     one-process-per-GPU sharding (torch.distributed plumbing; the single-process form is Convolver(devices=[...]))
"""
from ._lib import LIB_PATH, load, last_error  # noqa: F401
from .convolver import (  # noqa: F401
    Convolver, NToMonoConvolve, MonoConvolve, PartitionedConvolve, TimeDomainConvolve,
    LatencyMode, kLatencyZero, kLatencyShort, kLatencyMedium, ConvolveError,
    hisstools_rfft, hisstools_rifft, spectral_processor, EdgeMode, rccl_unique_id, host_register, host_unregister, ctl_reserve, ctl_reserved,
)
