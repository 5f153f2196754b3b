#!/usr/bin/env python3
"""Correctness of the four-step transforms (2^15 ... 2^22 complex points, every operation of the surface) against torch.fft in
float64 on the same box: max error relative to the spectrum's RMS.  Used when the passes change (HCV_FX_TILE=0 runs the LDS-staged
passes for comparison).

    python tools/micro/fx_tile_check.py [--sizes 15,16,...] [--batch 3]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hisstools_library_amd.fft as F  # noqa: E402


def rel(a, b):
    return float((a - b).abs().max() / b.abs().pow(2).mean().sqrt())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="15,16,17,18,19,20,21,22")
    ap.add_argument("--batch", type=int, default=3)
    ap.add_argument("--tol", type=float, default=2e-6)
    ap.add_argument("--min-tiles", type=int, default=0, help="batch of at least this many 16384-point tiles (the one-launch batch takes >= 4 per CU)")
    args = ap.parse_args()
    st = torch.cuda.current_stream().cuda_stream
    P = F.Precision.F32
    worst = 0.0
    g = torch.Generator(device="cuda").manual_seed(5)
    for lm in [int(x) for x in args.sizes.split(",")]:
        m = 1 << lm
        nb = max(args.batch, -(-args.min_tiles // max(1, m >> 14)))
        a = torch.rand(nb * m, device="cuda", generator=g) * 2 - 1
        b = torch.rand(nb * m, device="cuda", generator=g) * 2 - 1
        z = torch.complex(a.double(), b.double()).view(nb, m)
        # complex forward / inverse, in place
        for op, ref in ((F.Op.FFT, torch.fft.fft(z, dim=1)), (F.Op.IFFT, torch.fft.ifft(z, dim=1) * m)):
            x, y = a.clone(), b.clone()
            F.exec_dev(op, P, lm, nb, x.data_ptr(), y.data_ptr(), x.data_ptr(), y.data_ptr(), m, m, 0, st, True)
            e = rel(torch.complex(x.double(), y.double()).view(nb, m), ref)
            worst = max(worst, e)
            print(f"2^{lm} {op.name:10s} rel {e:.2e}", flush=True)
        # real forward, n = 2m samples: split input (even/odd samples), spectrum doubled with DC/Nyquist packed (Core.h:934-988)
        n = 2 * m
        xs = torch.rand(nb, n, device="cuda", generator=g) * 2 - 1
        spec = torch.fft.rfft(xs.double(), dim=1) * 2
        want_re, want_im = spec.real[:, :m].clone(), spec.imag[:, :m].clone()
        want_im[:, 0] = spec.real[:, m]
        ev, od = xs[:, 0::2].contiguous().view(-1), xs[:, 1::2].contiguous().view(-1)
        x, y = ev.clone(), od.clone()
        F.exec_dev(F.Op.RFFT, P, lm + 1, nb, x.data_ptr(), y.data_ptr(), x.data_ptr(), y.data_ptr(), m, m, 0, st, True)
        e = max(rel(x.double().view(nb, m), want_re), rel(y.double().view(nb, m), want_im))
        worst = max(worst, e)
        print(f"2^{lm + 1} RFFT       rel {e:.2e}", flush=True)
        xz = xs.contiguous().view(-1)
        x, y = torch.empty(nb * m, device="cuda"), torch.empty(nb * m, device="cuda")
        F.exec_dev(F.Op.RFFT_ZIP, P, lm + 1, nb, xz.data_ptr(), 0, x.data_ptr(), y.data_ptr(), n, m, n, st, True)
        e = max(rel(x.double().view(nb, m), want_re), rel(y.double().view(nb, m), want_im))
        worst = max(worst, e)
        print(f"2^{lm + 1} RFFT_ZIP   rel {e:.2e}", flush=True)
        # real inverse of that spectrum: n * x * 2 (unnormalised, Core.h:1364-1374), split and zipped
        sr, si = want_re.float().contiguous().view(-1), want_im.float().contiguous().view(-1)
        want = xs.double() * (2.0 * n)
        x, y = sr.clone(), si.clone()
        F.exec_dev(F.Op.RIFFT, P, lm + 1, nb, x.data_ptr(), y.data_ptr(), x.data_ptr(), y.data_ptr(), m, m, 0, st, True)
        got = torch.stack((x.view(nb, m), y.view(nb, m)), dim=2).reshape(nb, n).double()
        e = rel(got, want)
        worst = max(worst, e)
        print(f"2^{lm + 1} RIFFT      rel {e:.2e}", flush=True)
        out = torch.empty(nb * n, device="cuda")
        F.exec_dev(F.Op.RIFFT_ZIP, P, lm + 1, nb, sr.data_ptr(), si.data_ptr(), out.data_ptr(), 0, m, n, 0, st, True)
        e = rel(out.view(nb, n).double(), want)
        worst = max(worst, e)
        print(f"2^{lm + 1} RIFFT_ZIP  rel {e:.2e}", flush=True)
    print(f"worst {worst:.2e} (tolerance {args.tol:.1e}) {'OK' if worst < args.tol else 'FAIL'}")
    return 0 if worst < args.tol else 1


if __name__ == "__main__":
    sys.exit(main())
