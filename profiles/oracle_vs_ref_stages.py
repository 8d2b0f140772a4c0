#!/usr/bin/env python
"""How much slower the oracle (plain C restatement, gcc -O2, scalar) is than the reference's own SSE kernels, stage by stage,
on one 1080p plane — so that bench.py's cpu_baseline (kind "port") can say by how much it understates the reference.
Runs in the build container (needs /root/reference through oracle/_ref/libacfref.so); writes profiles/r04_oracle_vs_ref.json.

    python profiles/oracle_vs_ref_stages.py
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def best(fn, n=7):
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return min(t)


def main():
    from acf_amd import synth
    from oracle import binding as ob
    o, r = ob.lib(), ob.ref()
    H, W = 1080, 1920
    I = synth.make_frame(1, H, W, "luv")[0].copy()
    F = ob.F
    out = {}
    a, b = np.zeros_like(I), np.zeros_like(I)
    M, O, S = np.zeros_like(I), np.zeros_like(I), np.zeros_like(I)
    Hh = np.zeros((6, W // 4, H // 4), np.float32)
    fl = lambda p: p.ctypes.data_as(C.POINTER(C.c_float))
    # signatures of the reference wrappers (oracle/ref_api.cpp) and of the oracle
    cases = {
        "convTri1 (r=1, p=2)": (lambda: r.ref_convTri1(fl(I), fl(a), H, W, 1, C.c_float(2.0), 1), lambda: o.acfo_conv_tri1(fl(I), fl(b), H, W, 1, C.c_float(2.0), 1)),
        "gradMag": (lambda: r.ref_gradMag(fl(I), fl(M), fl(O), H, W, 1, 0), lambda: o.acfo_grad_mag(fl(I), fl(M), fl(O), H, W, 1, 0)),
        "convTri (r=5)": (lambda: r.ref_convTri(fl(M), fl(S), H, W, 1, 5, 1), lambda: o.acfo_conv_tri(fl(M), fl(S), H, W, 1, 5, 1)),
        "gradHist (bin 4, 6 orientations)": (lambda: r.ref_gradHist(fl(M), fl(O), fl(Hh), H, W, 4, 6, 0, 0), lambda: o.acfo_grad_hist(fl(M), fl(O), fl(Hh), H, W, 4, 6, 0, 0)),
    }
    for k, (fr, fo) in cases.items():
        tr, to = best(fr), best(fo)
        out[k] = {"reference_ms": round(tr * 1e3, 2), "oracle_ms": round(to * 1e3, 2), "oracle_over_reference": round(to / tr, 2)}
        print(k, out[k])
    json.dump({"plane": "1080x1920 f32", "host": "build container, 1 thread, best of 7", "stages": out}, open(os.path.join(ROOT, "profiles", "r04_oracle_vs_ref.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
