#!/usr/bin/env python
"""The build host's `_mm_rcp_ps` / `_mm_rsqrt_ps` as tables (run in the build container; any x86 host works).

    python tests/golden/make_x86_tables.py

The reference's SSE kernels use the two instructions at three sites (T/gradientMex.cpp:209-219,266; T/rgbConvertMex.cpp:161;
macros T/sse.hpp:185-192).  Their results are ~12-bit approximations whose bits belong to the CPU, so "the reference's output"
exists only per CPU.  This script shows that on THIS CPU each instruction is a pure function of few input bits and writes that
function down:

  rcp(x)   : sign, exponent 127 - e (+ the table entry's), mantissa RCP[m >> 11]              (4096 entries, x in [1, 2))
  rsqrt(x) : exponent -(e - 127 - odd) / 2 (+ the entry's), mantissa RSQ[odd][m >> 11]        (2 x 4096 entries, x in [1, 4))
  zero / subnormal -> inf of the input's sign, inf -> 0, NaN -> quiet NaN, underflow -> 0, rsqrt(negative) -> 0xffc00000

(12 mantissa bits cover both parts probed so far: the Intel Xeon of the build host decides on 11 / 10 of them — its entries repeat
in pairs / fours —, the AMD EPYC 9575F of the GPU boxes on all 12: profiles/r06_ab/probe_x86_structure.c.)

(1) probes the tables from the live instructions (oracle/acf_oracle.c: acfo_x86_probe), (2) checks the table functions
(acfo_x86_rcp_bits / acfo_x86_rsqrt_bits) against the live instructions for ALL 2^32 inputs — 0 mismatches or the script fails —,
(3) writes tests/golden/x86_rcp_rsqrt.npz {rcp, rsqrt, cpu, checked}.  The oracle's table tier (acfo_set_approx(3)) and the HIP
path's reference-arithmetic option (acf_hip_set_x86_tables + option "arith") evaluate these tables; with them the three sites are
bit-exact against the reference's own compiled kernels on this host, and tests/golden/tref_study.npz's T-ref hits (made on the
same host by make_tref.py) are reproduced bit for bit.
"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import binding as ob  # noqa: E402


def cpu_name():
    try:
        rows = {}
        for l in open("/proc/cpuinfo"):
            if ":" in l:
                k, v = l.split(":", 1)
                rows.setdefault(k.strip(), v.strip())
        return "%s | family %s model %s stepping %s" % (rows.get("model name", "?"), rows.get("cpu family", "?"), rows.get("model", "?"), rows.get("stepping", "?"))
    except Exception:
        import platform
        return platform.processor()


def main():
    t = ob.x86_probe()
    assert t is not None, "the oracle was not built for an SSE host"
    rcp, rsq = t
    ob.set_x86_tables(rcp, rsq)
    chunks = 256
    per = (1 << 32) // chunks
    with ThreadPoolExecutor(os.cpu_count() or 4) as ex:  # (ctypes releases the GIL)
        bad = list(ex.map(lambda k: ob.x86_verify(k * per, per, 1), range(chunks)))
    bad_rcp, bad_rsq = sum(b[0] for b in bad), sum(b[1] for b in bad)
    print("inputs checked: 2^32; mismatches rcp %d, rsqrt %d" % (bad_rcp, bad_rsq))
    assert bad_rcp == 0 and bad_rsq == 0, "this CPU's rcpps / rsqrtps are not the table functions: no fixture written"
    # structure: 12 significant result bits
    assert not (rcp & 0x7ff).any() and not (rsq & 0x7ff).any()
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "x86_rcp_rsqrt.npz")   # (another path: a probe of another host)
    np.savez_compressed(out, rcp=rcp, rsqrt=rsq, cpu=np.asarray(cpu_name()),
                        checked=np.asarray([1 << 32, bad_rcp, bad_rsq], np.int64))
    print("cpu:", cpu_name())
    print("rcp[0..2] %08x %08x %08x  rsqrt[0] %08x rsqrt[4096] %08x" % (rcp[0], rcp[1], rcp[2], rsq[0], rsq[4096]))


if __name__ == "__main__":
    main()
