"""The C++ drop-in headers (include/hisstools_amd/*.h): a caller written against the reference's class names compiles
and links against libhisstools_amd.so; on a GPU it must produce the exact impulse-IR answer."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "dropin_smoke.cpp")
OUT_DIR = os.path.join(ROOT, "tests", "cpp", "build")
EXE = os.path.join(OUT_DIR, "dropin_smoke")
LIBDIR = os.path.join(ROOT, "hisstools_library_amd")


FFT_SRC = os.path.join(ROOT, "tests", "cpp", "fft_tester.cpp")
FFT_EXE = os.path.join(OUT_DIR, "fft_tester")


def build(src=SRC, exe=EXE):
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["g++", "-std=c++14", "-O2", "-Wall", "-Werror", f"-I{os.path.join(ROOT, 'include')}", src, "-o", exe, f"-L{LIBDIR}", "-lhisstools_amd",
           f"-Wl,-rpath,{LIBDIR}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"]
    subprocess.check_call(cmd)


def test_dropin_headers_compile_and_link():
    build()
    rc = subprocess.call([EXE])
    assert rc in (0, 2)          # 2 = no GPU here: compile/link check only


@pytest.mark.gpu
def test_dropin_convolver_runs_on_gpu():
    build()
    out = subprocess.run([EXE], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_fft_header_compiles_and_links():
    """Every overload of HISSTools_FFT.h:87-369 resolves against the drop-in header."""
    build(FFT_SRC, FFT_EXE)
    assert subprocess.call([FFT_EXE]) in (0, 2)


@pytest.mark.gpu
def test_fft_tester_programme_runs_on_gpu():
    """The FFT_Tester-style programme: zip round trips log2 1..23, fft/ifft/rfft/rifft log2 0..21, double and float."""
    build(FFT_SRC, FFT_EXE)
    out = subprocess.run([FFT_EXE], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Finished Running" in out.stdout


AUDIO_SRC = os.path.join(ROOT, "tests", "cpp", "audiofile_smoke.cpp")
AUDIO_EXE = os.path.join(OUT_DIR, "audiofile_smoke")


def test_audiofile_header_round_trip(tmp_path):
    """HISSTools::IAudioFile / OAudioFile drop-ins: per-channel 24-bit AIFC write, read back (needs no GPU)."""
    build(AUDIO_SRC, AUDIO_EXE)
    out = subprocess.run([AUDIO_EXE, str(tmp_path / "smoke.aifc")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_audiofile_loads_an_ir_into_the_convolver(tmp_path):
    build(AUDIO_SRC, AUDIO_EXE)
    out = subprocess.run([AUDIO_EXE, str(tmp_path / "smoke.aifc")], capture_output=True, text=True)
    assert out.returncode == 0 and "convolver load ok" in out.stdout, out.stdout + out.stderr


GOLDEN_SRC = os.path.join(ROOT, "tests", "cpp", "dropin_golden.cpp")
GOLDEN_EXE = os.path.join(OUT_DIR, "dropin_golden")


def _export_golden(path):
    """golden_v1.npz (made by the unmodified reference) as a flat binary a C++ programme reads without a zip / npy parser"""
    import struct

    import numpy as np
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"))
    with open(path, "wb") as f:
        f.write(struct.pack("<I", len(g.files)))
        for k in g.files:
            a = np.ascontiguousarray(g[k], dtype=np.float32)
            name = k.encode()
            f.write(struct.pack("<I", len(name)) + name + struct.pack("<I", a.ndim) + struct.pack(f"<{a.ndim}I", *a.shape))
            f.write(a.tobytes())


def test_golden_programme_compiles_and_links(tmp_path):
    build(GOLDEN_SRC, GOLDEN_EXE)
    path = str(tmp_path / "golden_v1.bin")
    _export_golden(path)
    assert subprocess.call([GOLDEN_EXE, path]) in (0, 2)


@pytest.mark.gpu
def test_reference_golden_vectors_through_the_cpp_headers(tmp_path):
    """Every end-to-end golden vector of the reference (PartitionedConvolve x 3, TimeDomainConvolve x 4, MonoConvolve x 4 + moved
    elements of a std::vector, NToMonoConvolve 3 -> 1, Convolver 2 x 3 float and double, 3-channel parallel) through
    include/hisstools_amd/*.h, as a C++ caller of the reference would drive its classes: 2e-6 / 1e-5 of the peak."""
    build(GOLDEN_SRC, GOLDEN_EXE)
    path = str(tmp_path / "golden_v1.bin")
    _export_golden(path)
    out = subprocess.run([GOLDEN_EXE, path], capture_output=True, text=True)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all golden vectors within tolerance" in out.stdout and out.stdout.count(" ok") >= 22


CONTRACT_SRC = os.path.join(ROOT, "tests", "cpp", "audio_contract.cpp")
CONTRACT_EXE = os.path.join(OUT_DIR, "audio_contract")


def test_audio_contract_programme_compiles_and_links():
    build(CONTRACT_SRC, CONTRACT_EXE)
    assert subprocess.call([CONTRACT_EXE]) in (0, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("block,calls", [(128, 1400), (32, 4200)])
def test_audio_thread_contract_from_cpp(block, calls):
    """tests/cpp/audio_contract.cpp: HISSTools::Convolver, a SCHED_FIFO audio thread (where permitted) making paced host-pointer calls,
    a control thread looping set(resize) with growing 10 s impulse responses and stalled 300 us inside every control call's hand-over.
    Engine-exact criteria at any load (no start collision, every section on the audio thread, the untouched rows unchanged); no call
    over its budget and the slowest call inside it on a quiet host (MemorySwap.h:182-185, MonoConvolve.cpp:181-183)."""
    build(CONTRACT_SRC, CONTRACT_EXE)
    out = subprocess.run([CONTRACT_EXE, str(block), str(calls), "300"], capture_output=True, text=True, timeout=600)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr


UTIL_SRC = os.path.join(ROOT, "tests", "cpp", "utility_types.cpp")
UTIL_EXE = os.path.join(OUT_DIR, "utility_types")


def test_utility_types_behave_like_the_references():
    """MemorySwap<T> / Ptr, thread_lock, lock_hold (MemorySwap.h:19-290, ThreadLocks.hpp:51-120 — visible to every caller of the reference's
    Convolver.h) and FloatVector / ALIGNED_MALLOC (ConvolveSIMD.h:62-107) through the drop-in headers: attempt() fails empty while another
    thread holds a Ptr, grow / equal reallocate on > / !=, custom allocators are balanced, swap() never frees the caller's memory, moves,
    mutual exclusion of thread_lock.  Host-only."""
    build(UTIL_SRC, UTIL_EXE)
    out = subprocess.run([UTIL_EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "utility types ok" in out.stdout, out.stdout + out.stderr
