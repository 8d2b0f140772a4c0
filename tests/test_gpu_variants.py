"""Every kernel form that only an option or the environment selects (DESIGN.md "Switches") ships in the library: each runs here once,
in its own process (the environment is read once per process), against the oracle — tests/variant_case.py.  A name that
ACF_HIP_FORCE_FALLBACK does not know aborts the process: a renamed form cannot pass on the default form."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# a setting = space-separated tokens: NAME=value is an environment variable (ACF_HIP_FORCE_FALLBACK: a stage's fallback form runs
# where its default would; ACF_HIP_CASC_BOUNDS / ACF_HIP_TILE_TR / ACF_HIP_TILE_NW: the tile kernels' tuning knobs), opt:key=value an
# acf_hip_set_option call made before the plan
VARIANTS = [
    "",  # the defaults, as the baseline of this file
    "ACF_HIP_CASC_BOUNDS=8,24,24,96",
    "ACF_HIP_CASC_BOUNDS=32,32,32,128",
    "opt:tile_persist=0",
    "opt:tile_persist=24",
    "ACF_HIP_TILE_TR=16",
    "ACF_HIP_TILE_NW=4",
    "ACF_HIP_TILE_NW=16",
    "opt:rank_cells=0",
    "opt:rank_cells=0 ACF_HIP_TILE_TR=16",
    "opt:rank_cells=0 ACF_HIP_TILE_NW=4",
    "ACF_HIP_FORCE_FALLBACK=tail3",
    "ACF_HIP_FORCE_FALLBACK=no_tail_codes",
    "opt:smooth_segments=1",
    "opt:smooth_segments=5 opt:smooth_warm=16",
    "opt:fused_grad=2",
    "opt:fused_grad=0",
    "opt:fused_grad=2 opt:fused_tri=2",                                                    # k_smooth_grad_tri at every scale it fits
    "opt:fused_grad=2 opt:shared_device=1",                                                # the pools' option
    "ACF_HIP_FORCE_FALLBACK=resample_no_pair",
    "ACF_HIP_FORCE_FALLBACK=resample_no_strip",
    "opt:scale_streams=0 ACF_HIP_FORCE_FALLBACK=resample_no_pair",
    "ACF_HIP_FORCE_FALLBACK=resample_no_up",
    "ACF_HIP_FORCE_FALLBACK=resample_generic",
    "ACF_HIP_FORCE_FALLBACK=triy_unfused",
    "ACF_HIP_FORCE_FALLBACK=mou_plain",
    "ACF_HIP_FORCE_FALLBACK=level_groups",
    "ACF_HIP_FORCE_FALLBACK=triy_unfused,mou_plain,level_groups,resample_generic",       # every pyramid fallback at once
    "opt:level_segments=4 opt:level_warm=16",
    "opt:scale_streams=0",
    "opt:cascade_turns=5",
    "opt:graph=1",
]

LDCF_VARIANTS = ["", "ACF_HIP_FORCE_FALLBACK=ldcf_unfused", "ACF_HIP_FORCE_FALLBACK=ldcf_unfused,resample_generic", "ACF_HIP_FORCE_FALLBACK=no_dedup"]

DEPTH_VARIANTS = [(1, ""), (1, "opt:rank_cells=0"), (1, "ACF_HIP_FORCE_FALLBACK=tiled_staged"), (1, "opt:rank_cells=0 ACF_HIP_FORCE_FALLBACK=tiled_pooled1"), (1, "opt:tile_persist=0"),
                  (3, "opt:rank_cells=0"), (4, "opt:rank_cells=0"), (3, ""), (3, "ACF_HIP_FORCE_FALLBACK=tiled_staged"),
                  (3, "opt:tile_persist=0"), (3, "ACF_HIP_TILE_NW=4"), (3, "ACF_HIP_FORCE_FALLBACK=no_tail_codes"), (4, ""), (4, "ACF_HIP_FORCE_FALLBACK=tiled_staged"), (4, "opt:tile_persist=0")]


def _run(setting, *args):
    env = dict(os.environ)
    opts = []
    for kv in setting.split():
        k, v = kv.split("=", 1)
        if k.startswith("opt:"):
            opts.append("%s=%s" % (k[4:], v))
        else:
            env[k] = v
    env["VARIANT_OPTIONS"] = ",".join(opts)
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


def test_unknown_fallback_name_is_refused():
    r = _run("ACF_HIP_FORCE_FALLBACK=no_such_form")
    assert r.returncode != 0 and "unknown name" in r.stderr
