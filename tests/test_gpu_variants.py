"""The kernel forms that only an environment variable selects (A/B knobs) ship in the library: each runs here once,
in its own process (the variables are read once per process), against the oracle — tests/variant_case.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANTS = [
    "",  # the defaults, as the baseline of this file
    "ACF_HIP_CASC_BOUNDS=8,24,24,96",
    "ACF_HIP_CASC_BOUNDS=32,32,32,128",
    "ACF_HIP_TILE_PERSIST=0",
    "ACF_HIP_TILE_PERSIST=24",
    "ACF_HIP_RTILE_TR=16",
    "ACF_HIP_RTILE_NW=4",
    "ACF_HIP_RTILE_NW=16",
    "ACF_HIP_RTILE_WG=4",
    "ACF_HIP_RTILE_WG=2",
    "ACF_HIP_NO_RANK=1",
    "ACF_HIP_NO_RANK=1 ACF_HIP_TILE_TR=16",
    "ACF_HIP_NO_RANK=1 ACF_HIP_TILE_NW=4",
    "ACF_HIP_TAIL3=1",
    "ACF_HIP_NO_TAIL_CODES=1",
    "ACF_HIP_TILE_PAD_KB=16",
    "ACF_HIP_SMOOTH_SEGMENTS=1",
    "ACF_HIP_SMOOTH_SEGMENTS=5 ACF_HIP_SMOOTH_WARM=16",
    "ACF_HIP_GRAD_SEGMENTS=3 ACF_HIP_FUSED_GRAD=2",
    "ACF_HIP_FUSED_GRAD=2",
    "ACF_HIP_NO_FUSED_GRAD=1",
    "ACF_HIP_FUSED_GRAD=1 ACF_HIP_FUSED_GRAD_MINPX=1000 ACF_HIP_FUSED_GRAD_MINF=1",
    "ACF_HIP_FUSED_GRAD=2 ACF_HIP_FUSED_TRI=2",                                                  # k_smooth_grad_tri at every scale it fits
    "ACF_HIP_FUSED_GRAD=2 ACF_HIP_SHARED_DEVICE=1 ACF_HIP_FUSED_TRI_MINF=1",                     # the pools' option as the process default
    "ACF_HIP_GMV_BLOCKS=64",
    "ACF_HIP_RESAMPLE_NO_PAIR=1",
    "ACF_HIP_RESAMPLE_NO_STRIP=1",
    "ACF_HIP_SCALES_SERIAL=1 ACF_HIP_RESAMPLE_NO_PAIR=1",
    "ACF_HIP_RESAMPLE_NO_UP=1",
    "ACF_HIP_RESAMPLE_GENERIC=1",
    "ACF_HIP_TRIY_UNFUSED=1",
    "ACF_HIP_MOU_PLAIN=1",
    "ACF_HIP_LEVEL_GROUPS=1",
    "ACF_HIP_LEVEL_SEGMENTS=4 ACF_HIP_LEVEL_WARM=16",
    "ACF_HIP_SCALES_SERIAL=1",
    "ACF_HIP_CASCADE_TURNS=5",
    "ACF_HIP_GRAPH=1",
]
LDCF_VARIANTS = ["", "ACF_HIP_LDCF_UNFUSED=1", "ACF_HIP_LDCF_UNFUSED=1 ACF_HIP_RESAMPLE_GENERIC=1", "ACF_HIP_NO_DEDUP=1"]
# fixed depths other than 2: the pooled tile kernel (k_cascade_tile3D) and the forms an environment variable selects instead
DEPTH_VARIANTS = [(1, ""), (1, "ACF_HIP_NO_RANK=1"), (1, "ACF_HIP_TILED_STAGED=1"), (1, "ACF_HIP_NO_RANK=1 ACF_HIP_TILED_POOLED1=1"), (1, "ACF_HIP_TILE_PERSIST=0"),
                  (3, "ACF_HIP_NO_RANK=1"), (4, "ACF_HIP_NO_RANK=1"), (3, ""), (3, "ACF_HIP_TILED_STAGED=1"),
                  (3, "ACF_HIP_TILE_PERSIST=0"), (3, "ACF_HIP_TILE_NW=4"), (3, "ACF_HIP_NO_TAIL_CODES=1"), (4, ""), (4, "ACF_HIP_TILED_STAGED=1"), (4, "ACF_HIP_TILE_PERSIST=0")]


def _run(setting, *args):
    env = dict(os.environ)
    for kv in setting.split():
        k, v = kv.split("=", 1)
        env[k] = v
    return subprocess.run([sys.executable, os.path.join(ROOT, "tests", "variant_case.py"), *args], cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=300)


@pytest.mark.parametrize("setting", VARIANTS)
def test_environment_selected_form_matches_the_oracle(setting):
    r = _run(setting)
    assert r.returncode == 0 and r.stdout.strip().startswith("ok"), (setting, r.stdout[-1500:], r.stderr[-1500:])


@pytest.mark.parametrize("setting", LDCF_VARIANTS)
def test_environment_selected_ldcf_form_matches_the_oracle(setting):
    r = _run(setting, "ldcf")
    assert r.returncode == 0 and r.stdout.strip().startswith("ok"), (setting, r.stdout[-1500:], r.stderr[-1500:])


@pytest.mark.parametrize("depth,setting", DEPTH_VARIANTS)
def test_fixed_depth_forms_match_the_oracle(depth, setting):
    r = _run(setting, "depth%d" % depth)
    assert r.returncode == 0 and r.stdout.strip().startswith("ok"), (depth, setting, r.stdout[-1500:], r.stderr[-1500:])
