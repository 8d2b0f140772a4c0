"""The deployment bench.py times, as a test: three detector contexts on their own streams (acf_amd.detector.DetectorPool: turns for
the VALU-bound kernels, one tile workgroup per tile), batches of 1080p frames resident in HBM, the detection-only call
(keep_pyramid = 0: the levels leave as 16-bit rank cells), the device bbNms + prune, the fixed-capacity record export — run
concurrently for several steps, then six sampled frames of the LAST step checked against the oracle (chnsPyramid + acfDetect +
bbNms + prune on the CPU): count, boxes, levels, score bits.  bench.py's own self-check does the same after the clock stops;
this keeps the configuration green in the test suite."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_three_contexts_detection_only_with_device_nms():
    import torch
    import bench
    from acf_amd import capi, synth
    from acf_amd.detector import DetectorPool
    from acf_amd.dist import RecordGather
    H, W, C, B, cap = 1080, 1920, 3, 16, 32
    model = synth.make_model(seed=1, name="FACE80")
    dev = torch.device("cuda", 0)
    base = torch.from_numpy(np.stack([synth.make_frame(4000 + i, H, W, "luv") for i in range(3)])).to(dev)
    frames = torch.empty((C * B, 3, W, H), dtype=torch.float32, device=dev)
    for i in range(C * B):
        frames[i] = torch.roll(base[i % 3], shifts=(41 * (i // 3), 29 * (i // 3)), dims=(1, 2))
    pool = DetectorPool(C, model, H, W, 3, max_batch=B, max_hits=8192, device=0)
    nms = capi.make_nms(type="maxg", overlap=0.65, ovrDnm="min", prune=True, maxCount=10, pruneRatio=0.0)
    for det in pool.dets:
        det.set_option("scale_streams", 0)
        det.set_option("keep_pyramid", 0)
        det.set_nms(nms)
    pipes = [RecordGather(B, 1 + 6 * cap, 1, 0, dev) for _ in range(C)]
    for _ in range(3):
        for i in range(C):
            with torch.cuda.stream(pool.streams[i]):
                rec = pipes[i].buffer()
                pool.dets[i].run(frames[i * B:(i + 1) * B], B)
                pool.dets[i].export_detections(rec, cap)
                pipes[i].submit()
    for i in range(C):
        with torch.cuda.stream(pool.streams[i]):
            pipes[i].finish()
    torch.cuda.synchronize()
    recs = [pipes[i].rec[(pipes[i].k - 1) & 1].cpu().numpy() for i in range(C)]
    picks = [(i % C, (3 + 5 * i) % B) for i in range(6)]
    n = bench.verify_frames(model, H, W, [frames[i * B:(i + 1) * B] for i in range(C)], recs, cap, nms, picks)
    assert n == 6
    assert sum(int(r[:, 0].sum()) for r in recs) > 0
    pool.close()


def test_the_timed_configuration_as_it_is_timed():
    """bench.py's default run, in the suite: THREE contexts x 96 resident 1080p frames (memory: 3 x 96 x 24.9 MB = 7.2 GB of frames), the
    pool's own options (cascade_turns, one tile workgroup per tile, shared_device), keep_pyramid = 0, device bbNms + prune.  At 96
    frames per launch shared_device switches scale 0 to the one-chain-per-frame smoothing with convTri's x pass on it
    (k_smooth_grad_tri) and leaves every smoothing chain uncut: asserted from the profile (acf_hip_profile_get), not assumed.  Nine
    frames spread over the three contexts and over the batch are then checked against the oracle, score bits and all."""
    import torch
    import bench
    from acf_amd import capi, synth
    from acf_amd.detector import DetectorPool
    from acf_amd.dist import RecordGather
    H, W, C, B, cap = 1080, 1920, 3, 96, 32
    model = synth.make_model(seed=1, name="FACE80")
    dev = torch.device("cuda", 0)
    base = torch.from_numpy(np.stack([synth.make_frame(4100 + i, H, W, "luv") for i in range(4)])).to(dev)
    frames = torch.empty((C * B, 3, W, H), dtype=torch.float32, device=dev)
    for i in range(C * B):
        frames[i] = torch.roll(base[i % 4], shifts=(37 * (i // 4), 23 * (i // 4)), dims=(1, 2))
    pool = DetectorPool(C, model, H, W, 3, max_batch=B, max_hits=8192, device=0)   # (shared_device: the pool's default)
    nms = capi.make_nms(type="maxg", overlap=0.65, ovrDnm="min", prune=True, maxCount=10, pruneRatio=0.0)
    for det in pool.dets:
        det.set_option("scale_streams", 0)
        det.set_option("keep_pyramid", 0)
        det.set_option("profile", 1)
        det.set_nms(nms)
    pipes = [RecordGather(B, 1 + 6 * cap, 1, 0, dev) for _ in range(C)]
    for _ in range(2):
        for i in range(C):
            with torch.cuda.stream(pool.streams[i]):
                rec = pipes[i].buffer()
                pool.dets[i].run(frames[i * B:(i + 1) * B], B)
                pool.dets[i].export_detections(rec, cap)
                pipes[i].submit()
    for i in range(C):
        with torch.cuda.stream(pool.streams[i]):
            pipes[i].finish()
    torch.cuda.synchronize()
    for det in pool.dets:
        prof = det.profile()
        assert prof.get("k_smooth_grad_tri", (0, 0))[1] >= 2, sorted(prof)      # scale 0 of both steps
        assert "k_smooth_grad" not in prof and "k_tri_x" in prof, sorted(prof)  # scales 1-3 keep the separate x pass
        assert any(k.startswith("k_cascade_tile") for k in prof), sorted(prof)
        assert "k_level(fused)" in prof and "k_nms" in prof, sorted(prof)
    recs = [pipes[i].rec[(pipes[i].k - 1) & 1].cpu().numpy() for i in range(C)]
    picks = [(i % C, (5 + 11 * i) % B) for i in range(9)]
    n = bench.verify_frames(model, H, W, [frames[i * B:(i + 1) * B] for i in range(C)], recs, cap, nms, picks)
    assert n == 9
    assert sum(int(r[:, 0].sum()) for r in recs) > 0
    pool.close()


def test_bench_line_contract():
    """`python bench.py` (a small step: 2 contexts x 8 frames, no CPU leg) prints ONE JSON line with the fields the driver reads: the
    metric and unit, value = frames of the timed steps / time, the repeats, the roofline block with its kernel, the as-worded figures."""
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--contexts", "2", "--batch", "8", "--no-cpu-baseline"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["metric"].startswith("detector FPS @1080p") and d["unit"] == "frames/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert d["value"] > 0 and abs(d["value"] - 2 * 8 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    assert len(d["value_repeats"]) == 3 and d["value_min"] <= d["value_median"] <= d["value_max"]
    assert d["verified_frames"] == 3
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1
    cfg = d["config"]
    assert "workload" in cfg and cfg["contexts"] == 2 and cfg["frames_per_launch"] == 8
    assert cfg["batch8_fps_1gpu"] > 0 and cfg["batch8_pipelined_fps_1gpu"] > 0 and d["latency_ms_batch1"] > 0
    assert cfg["ref_arith_fps"] > 0 and cfg["ref_arith_verified_frames"] == 2   # the reference-arithmetic tier beside the headline, self-checked
    assert d["cpu_baseline"] is None or "value" in d["cpu_baseline"]
