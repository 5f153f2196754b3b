"""The full hisstools_* FFT surface (second "next" row of SURVEY.md §8f; HISSTools_FFT.h:87-369).

CPU: the oracle restatement against golden vectors produced by the unmodified reference (float: bit-identical; double:
1e-14 of the peak — the reference's SIMD passes order the double arithmetic differently), and against numpy.
GPU (-m gpu): every operation / precision through hcv_fft_exec against the oracle, the golden vectors and float64
numpy, plus the reference's own FFT_Tester programme ("- Test/FFT_Tester/FFT_Tester/main.cpp"): the zip / unzip
integer round trip for log2 1..23 and the fft / ifft / rfft / rifft sweep over log2 0..21 in both precisions."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = (0, 1, 2, 3, 4, 5, 6, 8, 10)
TOL32 = 2e-6            # of the output peak (float); long transforms: see tol()
TOL64 = 1e-13           # of the output peak (double)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_fft_v1.npz"))


def tol(prec, log2n):
    # rounding grows with the number of butterfly passes; 2e-6 holds to 2^16, allow sqrt-like growth above
    if prec == "f32":
        return TOL32 if log2n <= 16 else TOL32 * (1.0 + 0.25 * (log2n - 16))
    return TOL64


def err(got, want):
    got = got if isinstance(got, tuple) else (got,)
    want = want if isinstance(want, tuple) else (want,)
    peak = max(float(np.abs(np.asarray(w, np.float64)).max()) if w.size else 0.0 for w in want) or 1.0
    return max(float(np.abs(np.asarray(g, np.float64) - np.asarray(w, np.float64)).max()) if w.size else 0.0 for g, w in zip(got, want)) / peak


def cases(gold_keys=None):
    from oracle import oracle as O
    for prec in O.FFT_PRECISIONS:
        for l2 in SIZES:
            for op in O.FFT_OPS:
                if prec == "f32_to_f64" and op not in ("rfft_zip", "unzip"):
                    continue
                if op not in ("fft", "ifft") and l2 == 0:
                    continue
                yield op, prec, l2


def gold_case(gold, op, prec, l2):
    key = f"{op}_{prec}_{l2}"
    a = gold[key + "_a"]
    b = gold[key + "_b"] if key + "_b" in gold else None
    want = (gold[key + "_o0"], gold[key + "_o1"]) if key + "_o1" in gold else gold[key + "_o0"]
    return a, b, want


def numpy_truth(op, log2n, a, b, in_length=None):
    """float64 statement of the reference conventions (forward real spectra doubled, bin 0 = (DC, Nyquist))."""
    n = 1 << log2n
    half = n >> 1
    if op == "fft":
        z = np.fft.fft(a.astype(np.float64) + 1j * b.astype(np.float64))
        return z.real, z.imag
    if op == "ifft":
        z = np.fft.ifft(a.astype(np.float64) + 1j * b.astype(np.float64)) * n
        return z.real, z.imag
    if op in ("rfft", "rfft_zip"):
        if op == "rfft":
            x = np.empty(n)
            x[0::2], x[1::2] = a[:half], b[:half]
        else:
            x = np.zeros(n)
            x[:in_length] = a[:in_length]
        z = 2.0 * np.fft.rfft(x)
        re, im = z.real[:half].copy(), z.imag[:half].copy()
        im[0] = z.real[half]
        return re, im
    if op in ("rifft", "rifft_zip"):
        z = np.zeros(half + 1, complex)
        z[:half] = a[:half].astype(np.float64) + 1j * b[:half].astype(np.float64)
        z[0], z[half] = a[0], b[0]
        x = np.fft.irfft(z, n) * n
        return (x[0::2], x[1::2]) if op == "rifft" else x
    raise ValueError(op)


# ------------------------------------------------------------------------------------------------ CPU: oracle pinned

def test_oracle_matches_reference_vectors(oracle, gold):
    checked = 0
    for op, prec, l2 in cases():
        a, b, want = gold_case(gold, op, prec, l2)
        got = oracle.fft_surface(op, prec, l2, a, b, in_length=a.size if b is None else None)
        if prec == "f32" or op in ("unzip", "zip"):
            ok = all(np.array_equal(g, w) for g, w in zip(got if isinstance(got, tuple) else (got,), want if isinstance(want, tuple) else (want,)))
            assert ok, (op, prec, l2)
        else:
            assert err(got, want) <= 1e-14, (op, prec, l2, err(got, want))
        checked += 1
    assert checked == 148


def test_oracle_matches_numpy(oracle):
    rng = np.random.default_rng(5)
    for prec, dt, t in (("f32", np.float32, 2e-6), ("f64", np.float64, 1e-13)):
        for l2 in (3, 4, 7, 10, 13):
            n = 1 << l2
            half = n >> 1
            a, b = rng.uniform(-1, 1, n).astype(dt), rng.uniform(-1, 1, n).astype(dt)
            for op in ("fft", "ifft"):
                assert err(oracle.fft_surface(op, prec, l2, a, b), numpy_truth(op, l2, a, b)) <= t
            for op in ("rfft", "rifft", "rifft_zip"):
                assert err(oracle.fft_surface(op, prec, l2, a[:half], b[:half]), numpy_truth(op, l2, a, b)) <= t
            assert err(oracle.fft_surface("rfft_zip", prec, l2, a[: n - 3], in_length=n - 3), numpy_truth("rfft_zip", l2, a, None, n - 3)) <= t


# ------------------------------------------------------------------------------------------------ GPU

def hip_surface(op, prec, l2, a, b=None, in_length=None):
    import hisstools_library_amd.fft as F
    dt = np.float32 if prec == "f32" else np.float64
    if op == "fft":
        return F.hisstools_fft(np.asarray(a, dt), np.asarray(b, dt), l2)
    if op == "ifft":
        return F.hisstools_ifft(np.asarray(a, dt), np.asarray(b, dt), l2)
    if op == "rfft":
        return F.hisstools_rfft_split(np.asarray(a, dt), np.asarray(b, dt), l2)
    if op == "rifft":
        return F.hisstools_rifft_split(np.asarray(a, dt), np.asarray(b, dt), l2)
    if op == "rifft_zip":
        return F.hisstools_rifft(np.asarray(a, dt), np.asarray(b, dt), l2)
    if op == "zip":
        return F.hisstools_zip(np.asarray(a, dt), np.asarray(b, dt), l2)
    src = np.asarray(a, np.float64 if prec == "f64" else np.float32)
    if op == "rfft_zip":
        return F.hisstools_rfft(src, l2, in_length, out_dtype=dt)
    return F.hisstools_unzip_zero(src, src.size if in_length is None else in_length, l2, out_dtype=dt)


@pytest.mark.gpu
def test_gpu_matches_reference_vectors(gold):
    for op, prec, l2 in cases():
        a, b, want = gold_case(gold, op, prec, l2)
        got = hip_surface(op, prec, l2, a, b, in_length=a.size if b is None else None)
        if op in ("unzip", "zip"):
            assert all(np.array_equal(g, w) for g, w in zip(got if isinstance(got, tuple) else (got,), want if isinstance(want, tuple) else (want,)))
        else:
            assert err(got, want) <= tol(prec, l2), (op, prec, l2, err(got, want))


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("l2", list(range(0, 19)))
def test_gpu_matches_oracle_every_size(oracle, prec, l2):
    """All eight operations at every size from 1 point up to the first four-step sizes, with ragged input lengths."""
    rng = np.random.default_rng(1000 + l2)
    dt = np.float32 if prec == "f32" else np.float64
    n = 1 << l2
    half = n >> 1
    a, b = rng.uniform(-1, 1, n).astype(dt), rng.uniform(-1, 1, n).astype(dt)
    for op in ("fft", "ifft"):
        assert err(hip_surface(op, prec, l2, a, b), oracle.fft_surface(op, prec, l2, a, b)) <= tol(prec, l2), op
    if not half:
        return
    for op in ("rfft", "rifft", "rifft_zip", "zip"):
        want = oracle.fft_surface(op, prec, l2, a[:half], b[:half])
        got = hip_surface(op, prec, l2, a[:half], b[:half])
        assert err(got, want) <= (0 if op == "zip" else tol(prec, l2)), op
    for in_len in sorted({n, max(1, n - 1), max(1, n // 2 + 1), 1}):
        for op in ("rfft_zip", "unzip"):
            want = oracle.fft_surface(op, prec, l2, a[:in_len], in_length=in_len)
            got = hip_surface(op, prec, l2, a[:in_len], in_length=in_len)
            assert err(got, want) <= (0 if op == "unzip" else tol(prec, l2)), (op, in_len)
    if prec == "f64":
        x32 = a.astype(np.float32)
        for op in ("rfft_zip", "unzip"):
            want = oracle.fft_surface(op, "f32_to_f64", l2, x32[: n - 1 if n > 1 else 1], in_length=max(1, n - 1))
            got = hip_surface(op, "f32_to_f64", l2, x32[: max(1, n - 1)], in_length=max(1, n - 1))
            assert err(got, want) <= (0 if op == "unzip" else TOL64), op


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_gpu_batched_rows_match_single_transforms(oracle, prec):
    """2-D inputs are one launch; every row must equal the single transform of that row (no cross-talk between groups)."""
    import hisstools_library_amd.fft as F
    rng = np.random.default_rng(9)
    dt = np.float32 if prec == "f32" else np.float64
    for l2, batch in ((2, 700), (5, 300), (9, 37), (13, 5), (15, 3)):
        n = 1 << l2
        a, b = rng.uniform(-1, 1, (batch, n)).astype(dt), rng.uniform(-1, 1, (batch, n)).astype(dt)
        re, im = F.hisstools_fft(a, b, l2)
        for r in (0, batch // 2, batch - 1):
            assert err((re[r], im[r]), oracle.fft_surface("fft", prec, l2, a[r], b[r])) <= tol(prec, l2)
        x = F.hisstools_rifft(*F.hisstools_rfft(a, l2), l2)
        assert err(x, a * (2 * n)) <= tol(prec, l2) * 4


@pytest.mark.gpu
def test_fft_tester_zip_correctness():
    """zip_correctness_test (FFT_Tester main.cpp:201-250): integers survive unzip -> zip exactly for log2 1..23, both precisions."""
    import hisstools_library_amd.fft as F
    for dt in (np.float64, np.float32):
        for i in range(1, 24):
            ptr = np.arange(1 << i).astype(dt)                                   # exact in float up to 2^24
            re, im = F.hisstools_unzip(ptr, i)
            j = np.arange(1 << (i - 1))
            assert np.array_equal(re, (j << 1).astype(dt)), ("zip error", i)
            assert np.array_equal(im, ((j << 1) + 1).astype(dt)), ("zip error", i)
            assert np.array_equal(F.hisstools_zip(re, im, i), ptr), ("unzip error", i)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_fft_tester_crash_sweep(prec):
    """crash_test (FFT_Tester main.cpp:87-139): fft, ifft, rfft, rifft for every log2 in [0, 22) on uniform noise.  Beyond
    "does not crash" the sweep checks the size-independent properties: ifft(fft(z)) = N z, rifft(rfft(x)) = 2N x, Parseval."""
    import hisstools_library_amd.fft as F
    rng = np.random.default_rng(77)
    dt = np.float32 if prec == "f32" else np.float64
    for i in range(0, 22):
        n = 1 << i
        re, im = (1.0 - 2.0 * rng.random(n)).astype(dt), (1.0 - 2.0 * rng.random(n)).astype(dt)
        fr, fi = F.hisstools_fft(re, im, i)
        assert np.isfinite(fr).all() and np.isfinite(fi).all()
        e_t = float((re.astype(np.float64) ** 2 + im.astype(np.float64) ** 2).sum())
        e_f = float((fr.astype(np.float64) ** 2 + fi.astype(np.float64) ** 2).sum()) / n
        assert abs(e_f - e_t) <= 1e-4 * e_t if prec == "f32" else abs(e_f - e_t) <= 1e-11 * e_t, ("parseval", i)
        br, bi = F.hisstools_ifft(fr, fi, i)
        assert err((br, bi), (re * n, im * n)) <= 4 * tol(prec, i), ("ifft(fft)", i)
        if i >= 1:
            half = n >> 1
            sr, si = F.hisstools_rfft_split(re[:half], im[:half], i)
            xr, xi = F.hisstools_rifft_split(sr, si, i)
            assert err((xr, xi), (re[:half] * (2 * n), im[:half] * (2 * n))) <= 4 * tol(prec, i), ("rifft(rfft)", i)


@pytest.mark.gpu
@pytest.mark.parametrize("prec,l2", [("f32", 20), ("f32", 22), ("f64", 20), ("f64", 22), ("f32", 23), ("f64", 23)])
def test_gpu_largest_sizes_against_float64(oracle, prec, l2):
    """Largest supported transforms (complex 2^22, real 2^23) against the double oracle / numpy."""
    rng = np.random.default_rng(l2)
    dt = np.float32 if prec == "f32" else np.float64
    t = 1e-5 if prec == "f32" else 1e-12
    if l2 <= 22:
        n = 1 << l2
        a, b = rng.uniform(-1, 1, n).astype(dt), rng.uniform(-1, 1, n).astype(dt)
        assert err(hip_surface("fft", prec, l2, a, b), numpy_truth("fft", l2, a, b)) <= t
        assert err(hip_surface("ifft", prec, l2, a, b), numpy_truth("ifft", l2, a, b)) <= t
    n = 1 << l2
    x = rng.uniform(-1, 1, n - 5).astype(dt)
    spec = hip_surface("rfft_zip", prec, l2, x, in_length=n - 5)
    assert err(spec, numpy_truth("rfft_zip", l2, x, None, n - 5)) <= t
    back = hip_surface("rifft_zip", prec, l2, spec[0], spec[1])
    xp = np.zeros(n)
    xp[: n - 5] = x
    assert err(back, xp * (2.0 * n)) <= 4 * t


@pytest.mark.gpu
def test_gpu_device_pointer_entry(oracle):
    """hcv_fft_exec_dev on HBM-resident tensors, strided batch, on a side stream."""
    torch = pytest.importorskip("torch")
    import hisstools_library_amd.fft as F
    l2, batch, stride = 11, 6, (1 << 11) + 64
    n = 1 << l2
    rng = np.random.default_rng(3)
    a, b = rng.uniform(-1, 1, (batch, n)).astype(np.float32), rng.uniform(-1, 1, (batch, n)).astype(np.float32)
    re = torch.zeros(batch, stride, device="cuda"); im = torch.zeros(batch, stride, device="cuda")
    re[:, :n] = torch.from_numpy(a).cuda(); im[:, :n] = torch.from_numpy(b).cuda()
    re[:, n:] = 7.0
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    F.exec_dev(F.Op.FFT, F.Precision.F32, l2, batch, re.data_ptr(), im.data_ptr(), re.data_ptr(), im.data_ptr(), stride, stride, 0, st.cuda_stream, True)
    for r in range(batch):
        want = oracle.fft_surface("fft", "f32", l2, a[r], b[r])
        assert err((re[r, :n].cpu().numpy(), im[r, :n].cpu().numpy()), want) <= TOL32
    assert bool((re[:, n:] == 7.0).all())                       # the gaps between strided rows are untouched


@pytest.mark.gpu
@pytest.mark.parametrize("op", ["fft", "rfft"])
def test_gpu_four_step_batch_larger_than_one_scratch_chunk(op):
    """The four-step passes work through a batch in chunks of 1 GiB of scratch (hcv_fftx.hip: run_big): 17 double transforms of 2^22
    complex points (64 MiB each) are two chunks, 16 + 1.  The transforms either side of the chunk boundary and the last one against
    torch.fft in float64 on the same inputs (rfft: the packed, doubled spectrum of HISSTools_FFT_Core.h:934-988)."""
    torch = pytest.importorskip("torch")
    import hisstools_library_amd.fft as F
    lm, nb = 22, 17
    m = 1 << lm
    g = torch.Generator(device="cuda").manual_seed(11)
    a = torch.rand(nb * m, device="cuda", dtype=torch.float64, generator=g) * 2 - 1
    b = torch.rand(nb * m, device="cuda", dtype=torch.float64, generator=g) * 2 - 1
    rows = (0, 15, 16)
    keep = {r: (a[r * m:(r + 1) * m].clone(), b[r * m:(r + 1) * m].clone()) for r in rows}
    st = torch.cuda.current_stream().cuda_stream
    if op == "fft":
        F.exec_dev(F.Op.FFT, F.Precision.F64, lm, nb, a.data_ptr(), b.data_ptr(), a.data_ptr(), b.data_ptr(), m, m, 0, st, True)
    else:
        F.exec_dev(F.Op.RFFT, F.Precision.F64, lm + 1, nb, a.data_ptr(), b.data_ptr(), a.data_ptr(), b.data_ptr(), m, m, 0, st, True)
    for r in rows:
        x, y = keep[r]
        if op == "fft":
            want = torch.fft.fft(torch.complex(x, y))
            wre, wim = want.real, want.imag
        else:
            t = torch.stack((x, y), dim=1).reshape(-1)           # even samples in the real array, odd ones in the imaginary
            spec = torch.fft.rfft(t) * 2
            wre, wim = spec.real[:m].clone(), spec.imag[:m].clone()
            wim[0] = spec.real[m]
        peak = float(torch.maximum(wre.abs().max(), wim.abs().max()))
        e = max(float((a[r * m:(r + 1) * m] - wre).abs().max()), float((b[r * m:(r + 1) * m] - wim).abs().max())) / peak
        assert e <= 1e-12, (op, r, e)


@pytest.mark.gpu
def test_gpu_bad_descriptors_fail_loudly():
    import hisstools_library_amd.fft as F
    with pytest.raises(RuntimeError):
        F.hisstools_fft(np.zeros(1 << 23, np.float32), np.zeros(1 << 23, np.float32), 23)      # complex log2 > 22
    with pytest.raises(TypeError):
        F.hisstools_rfft(np.zeros(8, np.float64), 3, out_dtype=np.float32)


@pytest.mark.gpu
def test_gpu_random_batches_with_padded_rows(oracle):
    """Random operation / precision / size / batch, rows wider than the transform (stride > length, so the strided
    addressing of every kernel family is exercised): each row must equal the single transform of that row and the padding
    columns must come back untouched."""
    import hisstools_library_amd.fft as F
    rng = np.random.default_rng(2026)
    for case in range(120):
        prec = "f32" if rng.random() < 0.5 else "f64"
        dt = np.float32 if prec == "f32" else np.float64
        op = ["fft", "ifft", "rfft", "rifft", "rifft_zip", "rfft_zip"][int(rng.integers(0, 6))]
        l2 = int(rng.integers(1, 17))
        n = 1 << l2
        length = n if op in ("fft", "ifft") else n >> 1
        if op == "rfft_zip":
            length = int(rng.integers(1, n + 1))
        batch = int(rng.integers(1, 40 if l2 < 12 else 4))
        pad = int(rng.integers(0, 9))
        a = rng.uniform(-1, 1, (batch, length + pad)).astype(dt)
        b = rng.uniform(-1, 1, (batch, length + pad)).astype(dt)
        if op == "fft":
            got = F.hisstools_fft(a, b, l2)
        elif op == "ifft":
            got = F.hisstools_ifft(a, b, l2)
        elif op == "rfft":
            got = F.hisstools_rfft_split(a, b, l2)
        elif op == "rifft":
            got = F.hisstools_rifft_split(a, b, l2)
        elif op == "rifft_zip":
            got = F.hisstools_rifft(a, b, l2)
        else:
            got = F.hisstools_rfft(a, l2, in_length=length)
        for r in {0, batch // 2, batch - 1}:
            if op == "rfft_zip":
                want = oracle.fft_surface(op, prec, l2, a[r, :length], in_length=length)
                row = (got[0][r], got[1][r])
            elif op == "rifft_zip":
                want = oracle.fft_surface(op, prec, l2, a[r, :length], b[r, :length])
                row = got[r]
            else:
                want = oracle.fft_surface(op, prec, l2, a[r, :length], b[r, :length])
                row = (got[0][r, :length], got[1][r, :length])
                if pad:
                    assert np.array_equal(got[0][r, length:], a[r, length:]) and np.array_equal(got[1][r, length:], b[r, length:]), (case, op, "padding")
            assert err(row, want) <= tol(prec, l2), (case, op, prec, l2, batch, pad, err(row, want))


@pytest.mark.gpu
def test_gpu_vector_and_element_paths_agree_on_misaligned_rows(oracle):
    """The small and mid-size kernels move split arrays as 16-byte vectors when rows are 16-byte aligned and fall back to
    element access otherwise (hcv_fftx.hip: the staged paths, the real pre / post passes in LDS, the four-step tiles).  The
    same batch is transformed from an aligned base and from bases shifted by 1 ... 3 elements: every variant must match the
    oracle, for complex, real forward and real inverse transforms across the kernel families."""
    torch = pytest.importorskip("torch")
    import hisstools_library_amd.fft as F
    rng = np.random.default_rng(77)
    for prec, tdt, fprec, ndt in (("f32", torch.float32, F.Precision.F32, np.float32), ("f64", torch.float64, F.Precision.F64, np.float64)):
        for op, fop in (("fft", F.Op.FFT), ("rfft", F.Op.RFFT), ("rifft", F.Op.RIFFT)):
            for l2 in (4, 6, 8, 9, 10, 12, 13, 15, 16):
                n = 1 << l2
                length = n if op == "fft" else n >> 1
                batch = 5 if l2 < 12 else 2
                stride = length + 8                                  # a multiple of the vector width: alignment decides the path
                a = rng.uniform(-1, 1, (batch, length)).astype(ndt)
                b = rng.uniform(-1, 1, (batch, length)).astype(ndt)
                for shift in (0, 1, 2, 3):
                    re = torch.full((batch * stride + 8,), 7.0, dtype=tdt, device="cuda")
                    im = torch.full((batch * stride + 8,), 7.0, dtype=tdt, device="cuda")
                    rv = re[shift: shift + batch * stride].view(batch, stride)
                    iv = im[shift: shift + batch * stride].view(batch, stride)
                    rv[:, :length] = torch.from_numpy(a).cuda()
                    iv[:, :length] = torch.from_numpy(b).cuda()
                    torch.cuda.synchronize()
                    esz = re.element_size()
                    F.exec_dev(fop, fprec, l2, batch, re.data_ptr() + shift * esz, im.data_ptr() + shift * esz,
                               re.data_ptr() + shift * esz, im.data_ptr() + shift * esz, stride, stride, 0, 0, True)
                    gr, gi = rv.cpu().numpy(), iv.cpu().numpy()
                    for r in (0, batch - 1):
                        want = oracle.fft_surface(op, prec, l2, a[r], b[r])
                        e = err((gr[r, :length], gi[r, :length]), want)
                        assert e <= tol(prec, l2), (op, prec, l2, shift, r, e)
                    assert bool((rv[:, length:] == 7.0).all()) and bool((iv[:, length:] == 7.0).all()), (op, prec, l2, shift, "padding")
