"""`python bench.py --gpus N` must itself become N ranks (the flag was parsed and ignored).  The launch
plumbing — spawn under torch.distributed.run, rendezvous on 127.0.0.1, max-over-ranks clock, one JSON line from rank 0 —
is exercised here without a GPU through --dry-launch (gloo); the detector work of the real run needs a GPU per rank and
refuses to start without one."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_plain_gpus_2_spawns_two_ranks():
    r = _run(["--gpus", "2", "--dry-launch", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["config"]["rccl_ranks"] == 2
    assert j["config"]["per_rank"] == [1.0, 2.0]  # both ranks contributed
    assert j["seconds_max_over_ranks"] >= 0.02     # the slower rank's clock (rank 1 sleeps 20 ms)
    assert j["steps"] == 3 and j["warmup"] == 1    # the command line reached the ranks


def test_gpus_1_does_not_spawn():
    j = _json_line(_run(["--dry-launch"]).stdout)
    assert j["n_gpus"] == 1


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--dry-launch"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_real_run_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is visible")
    r = _run(["--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode != 0 and "no CPU form" in (r.stderr + r.stdout)


import pytest


@pytest.mark.gpu
def test_two_ranks_on_a_one_gpu_box_fail_with_the_missing_device():
    """Multi-GPU readiness without a node: `bench.py --gpus 2 --frames-total 64` on a box with ONE device spawns both ranks;
    rank 0 gets as far as the RCCL rendezvous, rank 1 stops with the explicit message and the launcher reports the failure —
    nothing hangs, nothing falls back to one GPU."""
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs exactly one visible GPU")
    r = _run(["--gpus", "2", "--frames-total", "64", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-latency"])
    out = r.stderr + r.stdout
    assert r.returncode != 0
    assert "rank 1 needs GPU 1 but 1 device(s) are visible" in out, out[-3000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]  # no bench line from a broken launch
