#!/usr/bin/env python
"""The K x K solve of the host-driven adaptive loop (more than 256 states) on its own: mbar_host_newton_direction (blocked, threaded
Cholesky factorisation + substitutions, csrc/mbar_host.cpp) at 511 / 767 / 1023 unknowns for a range of team sizes, best of 7,
next to numpy's LAPACK on the same matrix.  Host only (no device call); MBAR_DEBUG_TIMING=1 adds the library's own phase times."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymbar_amd import _lib  # noqa: E402

rng = np.random.default_rng(0)


def hessian(m):
    W = rng.random((4 * m, m))
    W /= W.sum(1, keepdims=True)
    return np.diag(W.sum(0)) - W.T @ W


print(f"host: {os.cpu_count()} logical CPUs", flush=True)
if "--small" in sys.argv:  # where the blocked form starts to pay: threads = -1 is the unblocked panels-of-4 factorisation
    for m in (65, 129, 193, 257, 321, 385, 449):
        H = hessian(m)
        g = rng.normal(size=m) * 1e-2
        g -= g.mean()
        out = []
        for th in (-1, 1, 2, 4):
            ts = []
            for _ in range(9):
                t0 = time.perf_counter()
                _lib.host_newton_direction(H, g, threads=th)
                ts.append(time.perf_counter() - t0)
            out.append(f"{'panels of 4' if th < 0 else str(th) + ' threads'} {1e3 * min(ts):6.3f} ms")
        print(f"{m - 1} unknowns: " + " | ".join(out), flush=True)
    sys.exit(0)
if "--cadence" in sys.argv:  # one call every ~6 ms, like the loop (the team asleep, its caches cold): median of 15
    for m in (512, 1024):
        H = hessian(m)
        g = rng.normal(size=m) * 1e-2
        g -= g.mean()
        out = []
        for th in (1, 2, 5, 8, 10, 16):
            ts = []
            for _ in range(15):
                time.sleep(0.006)
                t0 = time.perf_counter()
                _lib.host_newton_direction(H, g, threads=th)
                ts.append(time.perf_counter() - t0)
            out.append(f"{th} threads {1e3 * sorted(ts)[7]:6.3f} ms")
        print(f"{m - 1} unknowns, one call per 6 ms: " + " | ".join(out), flush=True)
    sys.exit(0)
for m in (512, 768, 1024):
    H = hessian(m)
    g = rng.normal(size=m) * 1e-2
    g -= g.mean()
    ref = np.linalg.lstsq(H, g, rcond=-1)[0]
    ref -= ref[0]
    out = []
    for th in (1, 2, 4, 5, 8, 10, 16):
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            x = _lib.host_newton_direction(H, g, threads=th)
            ts.append(time.perf_counter() - t0)
        assert np.max(np.abs(x - ref)) <= 1e-10 * np.max(np.abs(ref))
        out.append(f"{th} threads {1e3 * min(ts):6.3f} ms")
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        x = _lib.host_newton_direction(H, g)
        ts.append(time.perf_counter() - t0)
    out.append(f"library's choice {1e3 * min(ts):6.3f} ms")
    ts = []
    Hs = H[1:, 1:].copy()
    for _ in range(7):
        t0 = time.perf_counter()
        np.linalg.solve(Hs, g[1:])
        ts.append(time.perf_counter() - t0)
    out.append(f"numpy.linalg.solve {1e3 * min(ts):6.3f} ms")
    print(f"{m - 1} unknowns: " + " | ".join(out), flush=True)
