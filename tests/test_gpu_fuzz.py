"""A fixed-seed slice of tests/fuzz_parity.py: random frame sizes, model shapes (depth 1-4, trees, scales per octave,
pads, gradient plane, full orientation) and option sets (fused_grad, segments, streams, rank cells, graph, tiles) against the
oracle — pyramid bits, hits and boxes.  (Five seeds x 60 cases ran clean when this was added; the slice keeps that alive.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [7, 23])
def test_random_cases_match_the_oracle(seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), str(seed), "20"], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    assert last.startswith("cases 20") and last.endswith("mismatches 0"), out.stdout[-2000:]
    # skipped cases (geometry the plan refuses, more hits than a buffer holds) do not count as clean ones
    words = last.split()
    ran, checked = int(words[words.index("ran") + 1]), int(words[words.index("frames_checked") + 1])
    assert ran >= 10 and checked >= ran, last


def test_random_ldcf_cases_match_the_oracle():
    """tests/fuzz_ldcf.py: the LDCF post-stage (k_ldcf_tile) on random frame sizes, k = 1 .. 5, strides, scales per octave."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_ldcf.py"), "11", "16"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    last = out.stdout.strip().splitlines()[-1]
    assert last.startswith("cases 16") and last.endswith("mismatches 0"), out.stdout[-2000:]
    words = last.split()
    ran, checked = int(words[words.index("ran") + 1]), int(words[words.index("frames_checked") + 1])
    assert ran >= 8 and checked >= ran, last
