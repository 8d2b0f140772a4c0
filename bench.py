#!/usr/bin/env python
"""bench.py — detector FPS @1080p, 8 scales/octave, FACE80 model on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--contexts C] [--config {2,4,5}] [--frames-total T]

One "step" = one pass of the hot path (acf_hip_run: chnsPyramid + acfDetect, then the device-side export of the
detection records) over batches of synthetic frames that are already resident in HBM.  With N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank runs its own frames (weak scaling, no data-path collective) and the
fixed-capacity detection records are gathered to rank 0 over RCCL inside the timed step.  --frames-total T is
BASELINE.json's cfg 3 as worded: T frames per step shared by the N GPUs (T/N per GPU: strong scaling).

Prints ONE JSON line on rank 0 (contract in the task description) with
 - value: whole-job frames/s,
 - roofline: the dominant kernel's algorithmic bytes per launch over its average launch duration INSIDE the timed region
   (HIP events on the launch stream), the same kernel alone on the GPU under `solo`, the whole path under `path_*`,
 - latency_ms_batch1: one frame, one context, submit to synchronise,
 - cpu_baseline: the oracle (CPU restatement) on a bounded sample of the same workload, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

# BASELINE.json configs that are bench lines (cfg 1 is the CPU plumbing case, cfg 3 is cfg 2 batched over GPUs)
CONFIGS = {
    2: dict(H=1080, W=1920, kind="luv", model=dict(name="FACE80"),
            what="1920x1080 planar f32 LUV frames, synthetic FACE80-shaped model (80x80, 10 ch, depth 2, 2048 trees), nPerOct 8, nApprox 7, shrink 4"),
    4: dict(H=480, W=640, kind="rgb", model=dict(name="INRIA"),
            what="640x480 planar f32 RGB frames (RGB->LUV on the device), synthetic INRIA-shaped model (128x64 padded window, pad [16 12], nOctUp 1, 2048 trees)"),
    5: dict(H=2160, W=3840, kind="luv", model=dict(name="FACE80", nPerOct=12, nApprox=11, ldcfK=4),
            what="3840x2160 planar f32 LUV frames, 12 scales/octave, FACE80-shaped model over LDCF channels (k = 4 5x5 filters per channel, 40 channels at shrink 8)"),
}


def kernel_bytes_per_frame(det, model):
    """Algorithmic (compulsory in+out) bytes per frame of each kernel, from the plan."""
    lv = det.levels
    nC = det.nChns
    d = 1 if model["colorSpace"] == 0 else 3
    sh = model["shrink"]
    real = [l for l in lv if l.isReal]
    np_real = [l.hC * sh * l.wC * sh for l in real]
    raw_real = 4 * nC * sum(l.hC * l.wC for l in real)
    pyr = 4 * det.pyr_floats
    b = {}
    b["k_smooth_tri1(image)"] = sum(2 * d * n * 4 for n in np_real)
    # fused smoothing: every plane read; the gradient plane written at every scale, all planes at the scale later scales are
    # resampled from (index 1 here); the colour channels (1/16) and, from scale 0, the half-resolution next image (1/4)
    b["k_smooth_vec"] = sum(4 * (d * n + n + d * n // 16) for n in np_real) + (4 * (d - 1) * np_real[1] if len(np_real) > 1 else 0) + \
        (4 * d * np_real[0] // 4 if len(np_real) > 1 else 0)
    # scale 0's gradient plane as its own launch (profile names k_smooth_grad / k_smooth_grad_tri): the plane read; M, O (and U with
    # convTri's x pass on the chain) written instead of the smoothed plane; its colour cells (1/16) and its quarter of the half-size
    # image.  "k_smooth_vec" is then the other planes' launch of scale 0 plus every plane of the other scales.
    n0 = np_real[0]
    own = 4 * (n0 + n0 // 16 + (n0 // 4 if len(np_real) > 1 else 0))
    b["k_smooth_grad"] = own + 4 * 2 * n0
    b["k_smooth_grad_tri"] = own + 4 * 3 * n0
    b["k_grad_mag"] = sum(3 * n * 4 for n in np_real)
    b["k_tri_x"] = sum(2 * n * 4 for n in np_real)
    b["k_tri_y"] = sum(2 * n * 4 for n in np_real)
    b["k_chns"] = sum(3 * n * 4 + (nC - d) * (n // (sh * sh)) * 4 for n in np_real)  # M, S, O in; magnitude + histogram channels out
    b["k_triy_chns"] = b["k_chns"]  # the fused y pass: U (instead of S), M, O in; the same channels out
    b["k_resample(image)"] = sum(d * 4 * np_real[1] + d * 4 * np_real[i] for i in range(2, len(np_real))) if len(np_real) > 2 else 0
    b["k_level(fused)"] = raw_real + pyr          # real levels' raw channels in, padded pyramid out
    b["k_level(smooth)"] = 2 * pyr
    b["k_resample(approx)"] = raw_real + pyr
    b["k_smooth_tri1(levels)"] = 2 * pyr
    casc = pyr
    if det.ldcf_levels:
        k = int(model.get("ldcfK", 0))
        casc = 4 * sum(nC * k * l.hP * l.wP for l in det.ldcf_levels)
        b["k_ldcf_conv"] = pyr + 4 * casc  # the plain pyramid in, k filtered full-resolution levels out
        b["k_resample(ldcf)"] = 4 * casc + casc
        b["k_ldcf_tile"] = pyr + casc             # the fused form: the plain pyramid in, the k filtered and halved levels out
    b["k_cascade_tile"] = casc                    # every cell of the cascade's pyramid read once at SURVEY §8d's 4 bytes per cell (halo re-reads are L2 hits); the rank-cell form moves half of it
    b["k_cascade"] = casc
    b["k_tail_scan"] = 0                          # data dependent: (windows alive after tree 128) x (tail trees) code bytes
    b["k_cascade_tail3"] = 0
    b["k_sort_map"] = 0
    b["k_nms"] = 0
    return b


PMC_FILES = ("profiles/r06_pmc_traffic.json", "profiles/r05_pmc_traffic.json", "profiles/r04_pmc_traffic.json")  # the newest committed pass that exists
PMC_FILE = next((f for f in PMC_FILES if os.path.exists(os.path.join(ROOT, f))), PMC_FILES[-1])


def pmc_traffic(kernel, batch):
    """HBM bytes per launch of `kernel` from the COMMITTED rocprofv3 PMC summary (PMC_FILE, produced by profiles/r04_profile.sh:
    FETCH_SIZE x 2 (gfx950 wide-load correction, MI355X_MICROARCH.md §HBM) + WRITE_SIZE, KB -> bytes), or None.  Not measured
    in this run: the line says so (roofline.traffic_source)."""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            t = json.load(f)
        if t.get("frames_per_step") != batch:
            return None
        return t["kernels"].get(kernel)
    except Exception:
        return None


def pmc_path_bytes_per_frame(batch):
    """HBM bytes per frame of the WHOLE path (every kernel of a step) from the same committed PMC passes, or None."""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            t = json.load(f)
        if t.get("frames_per_step") != batch:
            return None
        return sum(t["kernels"].values()) / float(batch)
    except Exception:
        return None


def verify_frames(model, H, W, frames, recs, cap, nms, picks, ref_arith=False):
    """Outside the clock: the records the TIMED contexts exported for `picks` = [(context, frame in batch), ...] against the
    oracle on the very same frames (pulled back from HBM): chnsPyramid + acfDetect (+ bbNms + prune) on the CPU, boxes and
    levels exact, score bits exact.  Returns the number of frames checked; raises on the first difference."""
    from acf_amd import capi
    from acf_amd.dist import records_to_detections
    from oracle import binding as ob
    plan = ob.Plan(model, H, W, 3)
    n = 0
    for (ci, fi) in picks:
        frame = frames[ci][fi].cpu().numpy()
        if ref_arith:
            ob.set_x86_tables(*ob.x86_fixture())
            ob.set_approx(3)
        try:
            pyr, _, _ = ob.chns_pyramid(plan, frame)
        finally:
            ob.set_approx(0)
        want, _ = ob.detect(plan, pyr)
        if nms is not None:
            keep = ob.nms(np.stack([want["x"], want["y"], want["w"], want["h"]], axis=1), want["score"].astype(np.float64), nms)
            want = want[keep]
        rec = recs[ci][fi]
        if int(rec[0]) != len(want) or len(want) > cap:
            raise SystemExit("bench self-check: context %d frame %d: %d detections exported, oracle has %d (cap %d)" % (ci, fi, int(rec[0]), len(want), cap))
        got = records_to_detections(rec, cap)
        for g, w_ in zip(got, want):
            same = (g[0], g[1], g[2], g[3], g[5]) == (int(w_["x"]), int(w_["y"]), int(w_["w"]), int(w_["h"]), int(w_["scale"])) and \
                np.float32(g[4]).view(np.uint32) == np.float32(w_["score"]).view(np.uint32)
            if not same:
                raise SystemExit("bench self-check: context %d frame %d differs from the oracle: %r vs %r" % (ci, fi, g, w_))
        n += 1
    return n


def cpu_baseline(model, frames_np, H, W, budget_s=12.0):
    """Oracle (oracle/acf_oracle.c, the CPU restatement) timed on a bounded sample of the same workload: first one frame
    single-threaded, then for `budget_s` seconds on every host core with frames dealt to threads (the reference
    parallelises over images with one detector per thread, src/app/acf/acf.cpp:255-320; the C calls release the GIL).
    Reported baseline, not the thing measured."""
    import threading
    from oracle import binding as ob
    plan = ob.Plan(model, H, W, 3)
    ldcf = int(model.get("ldcfK", 0)) > 0

    def one(i):
        pyr, _, _ = ob.chns_pyramid(plan, frames_np[i % len(frames_np)])
        if ldcf:
            lvL, pyrL, _ = ob.ldcf(plan, pyr)
            ob.detect_ldcf(plan, lvL, pyrL, cap=1 << 17)
        else:
            ob.detect(plan, pyr)

    t0 = time.perf_counter()
    one(0)  # also initialises the oracle's lazily built tables before any thread starts
    single = 1.0 / (time.perf_counter() - t0)
    cores = max(1, min(os.cpu_count() or 1, 64))
    done = [0] * cores
    stop = time.perf_counter() + budget_s

    def work(k):
        i = k
        while time.perf_counter() < stop:
            one(i)
            i += cores
            done[k] += 1

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    n = sum(done)
    out = {"value": n / dt, "unit": "frames/s", "cores": cores, "kind": "port", "single_thread_value": single,
           "sample": "%d synthetic frames of the benchmarked workload, oracle/acf_oracle.c (gcc -O3, no FMA, no fast-math), %d threads "
                     "(one frame each at a time), %.1f s" % (n, cores, dt)}
    if ob.have_ref() and not ldcf:
        # the same sample with the REFERENCE'S OWN compiled SSE kernels (oracle/_ref/libacfref.so, shipped as a binary: convTri1, convTri,
        # gradMag, gradMagNorm, gradHist, resample, rgbConvert — rsqrtps / rcpps and all) under the restated orchestration and cascade,
        # on this box's cores: the nearest thing to "the reference's CPU path" that can run here (the OpenCV-typed rest cannot be built)
        done2 = [0] * cores
        stop2 = time.perf_counter() + budget_s / 2

        def work2(k):
            ob.set_tref(True)  # (thread-local)
            try:
                i = k
                while time.perf_counter() < stop2:
                    one(i)
                    i += cores
                    done2[k] += 1
            finally:
                ob.set_tref(False)

        t0 = time.perf_counter()
        th = [threading.Thread(target=work2, args=(k,)) for k in range(cores)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt2 = time.perf_counter() - t0
        # ... and that is the baseline the line reports (`kind`: "reference"); the port's own figure stays beside it
        out = {"value": sum(done2) / dt2, "unit": "frames/s", "cores": cores, "kind": "reference", "single_thread_value_port": single,
               "sample": "%d synthetic frames of the benchmarked workload; the reference's own compiled toolbox kernels (oracle/_ref/libacfref.so: SSE, rsqrtps / "
                         "rcpps) under the restated chnsPyramid / acfDetect orchestration (the OpenCV-typed rest of the reference cannot be built), %d threads "
                         "(one frame each at a time), %.1f s" % (sum(done2), cores, dt2),
               "port": {"value": out["value"], "kind": "port", "sample": out["sample"]}}
    try:
        # how the port compares with the reference's own SSE kernels where those compile here (profiles/oracle_vs_ref_stages.py,
        # run in the build container): > 1 means the port is slower, i.e. this baseline understates the reference by about that
        with open(os.path.join(ROOT, "profiles", "r04_oracle_vs_ref.json")) as f:
            st = json.load(f)["stages"]
        out["port_over_reference_kernels"] = {k: v["oracle_over_reference"] for k, v in st.items()}
    except Exception:
        pass
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line as N ranks under torch.distributed.run
    (one process per GPU, LOCAL_RANK -> device, rendezvous on 127.0.0.1 at a free port).  Rank 0's JSON line is this
    process's output; returns the launcher's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_launch(args, world):
    """The N-rank protocol of the bench with the detector work left out: rendezvous, barrier-bracketed clock, MAX over
    ranks, per-rank figures summed into one vector, one JSON line from rank 0.  gloo when no GPU is visible."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo", rank=rank, world_size=world)
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    if world > 1:
        dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    per_rank = torch.zeros(world, dtype=torch.float64)
    per_rank[rank] = float(rank + 1)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"metric": "launch plumbing only (no detector work)", "value": None, "dry_launch": True, "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "seconds_max_over_ranks": float(dt.item()),
                          "config": {"rccl_ranks": world, "per_rank": [float(x) for x in per_rank]}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json configuration (2: the metric's; 4, 5: extra bench lines)")
    ap.add_argument("--batch", type=int, default=0, help="frames per detector context per launch (default: 96 at cfg 2, 192 at cfg 4, 24 at cfg 5)")
    ap.add_argument("--contexts", type=int, default=3, help="detector contexts per GPU, each on its own HIP stream; a step runs one batch on each "
                    "(the cascade of one batch overlaps the pyramid of another)")
    ap.add_argument("--frames-total", type=int, default=0, help="cfg 3 as worded: this many frames per step shared by all GPUs (strong scaling); "
                    "overrides --batch (frames per GPU = total / gpus, split over the contexts)")
    ap.add_argument("--cap", type=int, default=0, help="detections exported per frame (gather record capacity; default 32 with the device NMS, 1024 without)")
    ap.add_argument("--no-nms", action="store_true", help="export the raw detections (scale, column, row order) instead of the survivors of the "
                    "device bbNms + prune (Detector::operator()'s default: maxg, overlap .65 / min, at most 10 per frame)")
    ap.add_argument("--streams", type=int, default=1, help="sub-batch contexts per GPU (acf_hip_set_option streams): chunks of the batch run concurrently")
    ap.add_argument("--turns", type=int, default=-1, help="A/B: option cascade_turns of every context (default: what DetectorPool sets, 5 with several contexts)")
    ap.add_argument("--persist", type=int, default=-1, help="A/B: option tile_persist of every context (default: what DetectorPool sets, 0 with several contexts)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT", help="A/B: acf_hip_set_option(KEY, INT) on every context (repeatable)")
    ap.add_argument("--ref-arith", action="store_true", help="the reference-arithmetic tier (option arith = 1 with the committed build-host tables, "
                    "tests/golden/x86_rcp_rsqrt.npz): the three rsqrtps / rcpps sites return that CPU's bits; the self-check then uses the oracle's table tier")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency launches (PMC passes: every launch in the trace is then a full batch)")
    ap.add_argument("--keep-pyramid", action="store_true", help="also materialise the float pyramid (acf_hip option keep_pyramid = 1); by default the timed "
                    "call is detection only — Detector::operator()(image) — and the levels leave the level kernel as 16-bit threshold-rank cells")
    ap.add_argument("--no-repeats", action="store_true", help="skip the two extra timed regions behind `value_repeats` (profiling passes)")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-run check of three timed frames against the oracle")
    ap.add_argument("--dry-launch", action="store_true", help="launch plumbing only (spawn, rendezvous, max-over-ranks clock, per-rank gather) "
                    "with no detector work: runs without a GPU over gloo (tests/test_bench_launch.py); never a measurement")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "0"))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if world == 0 and args.gpus > 1:
        # plain `python bench.py --gpus N`: become N ranks (one per GPU) under torch.distributed.run, rendezvous on 127.0.0.1
        sys.exit(spawn_ranks(args.gpus))
    world = max(world, 1)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one rank per GPU (python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ...)"
                         % (args.gpus, world, args.gpus, args.gpus))
    if args.dry_launch:
        return dry_launch(args, world)

    import torch
    import torch.distributed as dist
    from acf_amd import synth
    from acf_amd.detector import DetectorPool
    from acf_amd.dist import RecordGather

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local:
        raise SystemExit("bench.py: rank %d needs GPU %d but %d device(s) are visible (the hot path has no CPU form)"
                         % (rank, local, torch.cuda.device_count() if torch.cuda.is_available() else 0))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = CONFIGS[args.config]
    H, W, C = cfg["H"], cfg["W"], max(1, args.contexts)
    B = args.batch or {2: 96, 4: 192, 5: 24}[args.config]
    scaling = "weak"
    if args.frames_total:
        per_gpu = args.frames_total // world
        if per_gpu * world != args.frames_total or per_gpu < 1:
            raise SystemExit("--frames-total must be a positive multiple of the number of GPUs")
        C = min(C, per_gpu)
        while per_gpu % C:
            C -= 1
        B = per_gpu // C
        scaling = "strong"

    model = synth.make_model(seed=1, **cfg["model"])
    # distinct base frames per rank (8 seeds), expanded to C*B distinct frames by cyclic shifts (cheap, on device)
    nbase = 8
    base_np = [synth.make_frame(1000 * rank + i + 1, H, W, cfg["kind"]) for i in range(nbase)]
    base = torch.from_numpy(np.stack(base_np)).to(dev)
    frames = torch.empty((C * B, 3, W, H), dtype=torch.float32, device=dev)
    for i in range(C * B):
        frames[i] = torch.roll(base[i % nbase], shifts=(37 * (i // nbase), 53 * (i // nbase)), dims=(1, 2))
    del base
    torch.cuda.synchronize()

    # C detector contexts per GPU, each with its own HIP stream, plan and buffers (the reference runs one detector per thread the
    # same way, src/app/acf/acf.cpp:255-320).  The path alternates HBM-bound kernels (smoothing, gradMag, running sums) with
    # latency-bound ones (level chains, cascade tiles): with independent streams the cascade of one batch fills the machine while
    # another batch's pyramid waits on memory.  Every context has its own record gather (the only exchange of the path, issued
    # asynchronously on the context's stream over two record buffers); everything in flight is waited for inside the timed region.
    pool = DetectorPool(C, model, H, W, 3, max_batch=B, max_hits=8192, device=local, streams=args.streams)
    streams, dets = pool.streams, pool.dets
    from acf_amd import capi
    nms_params = None
    args.cap = args.cap or (1024 if args.no_nms else 32)
    for det in dets:
        if C > 1:
            # several contexts side by side already fill each other's gaps: a context's real scales stay on its one stream
            # (acf_hip.h, scale_streams; measured: 3 contexts 12.1k frames/s with 0, 11.6k with 1; one context 9.8k / 10.5k)
            det.set_option("scale_streams", 0)
        if args.turns >= 0:
            det.set_option("cascade_turns", args.turns)
        if args.persist >= 0:
            det.set_option("tile_persist", args.persist)
        for kv in args.opt:
            det.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        if args.ref_arith:
            z_ = np.load(os.path.join(ROOT, "tests", "golden", "x86_rcp_rsqrt.npz"))
            det.set_x86_tables(z_["rcp"], z_["rsqrt"])
            det.set_option("arith", 1)
        if not args.keep_pyramid and args.config != 5:
            det.set_option("keep_pyramid", 0)
        if not args.no_profile:
            det.set_option("profile", 1)
        if not args.no_nms:
            # ACF.cpp:332-353 + ObjectDetector.cpp:28-44 with acf::Detector's defaults, on the device: the gather record shrinks
            # from 24 KB to < 1 KB per frame
            nms_params = capi.make_nms(type="maxg", overlap=0.65, ovrDnm="min", prune=True, maxCount=10, pruneRatio=0.0)
            det.set_nms(nms_params)
    pipes = [RecordGather(B, 1 + 6 * args.cap, world, rank, dev) for _ in range(C)]
    if args.no_nms:
        nms_params = None

    def step():
        for i in range(C):
            with torch.cuda.stream(streams[i]):
                rec = pipes[i].buffer()
                dets[i].run(frames[i * B:(i + 1) * B], B)
                dets[i].export_detections(rec, args.cap)
                pipes[i].submit()

    def finish():
        out = None
        for i in range(C):
            with torch.cuda.stream(streams[i]):
                out = pipes[i].finish()
        torch.cuda.synchronize()
        return out

    for _ in range(args.warmup):
        step()
    finish()
    if not args.no_profile:
        for det in dets:
            det.profile()  # drop warm-up events
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    gathered = finish()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt_rank = dt
    dt = float(tmax.item())
    rank_fps = torch.zeros(world, dtype=torch.float64, device=dev)
    rank_fps[rank] = C * B * args.steps / dt_rank
    if world > 1:
        dist.all_reduce(rank_fps, op=dist.ReduceOp.SUM)

    # the first region above is `value` (the contract's K steps, MAX over ranks).  Boxes of the pool differ by +-4 %, and a region
    # is ~0.4 s: two more regions of the same K steps, timed the same way, say how far a single region of THIS box moves
    prof_first = None
    if not args.no_profile:
        prof_first = [d_.profile() for d_ in dets]  # (the per-kernel figures stay those of the first region)
    repeat_fps = [C * B * world * args.steps / dt]
    for _ in range(0 if args.no_repeats else 2):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        finish()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        tr = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        repeat_fps.append(C * B * world * args.steps / float(tr.item()))
    if not args.no_profile:
        for d_ in dets:
            d_.profile()  # (drop the repeats' events)

    det = dets[0]
    prof, solo = {}, {}
    latency_ms = None
    if not args.no_profile:
        for pf in prof_first:
            for k_, (ms_, n_) in pf.items():
                a_ = prof.get(k_, (0.0, 0))
                prof[k_] = (a_[0] + ms_, a_[1] + n_)
        # outside the timed region: the same launches with one context alone on the machine, so that a kernel's own speed
        # can be read next to its speed while sharing the chip with the other contexts' kernels
        with torch.cuda.stream(streams[0]):
            for _ in range(3):
                dets[0].run(frames[:B], B)
        dets[0].synchronize()
        solo = dets[0].profile()
    keep_fps = None
    if world == 1 and not args.keep_pyramid and args.config != 5 and not args.no_latency:
        # the chnsPyramid-returning call beside the headline's detection-only call: the same steps with the float pyramid
        # materialised as well (option keep_pyramid = 1), timed the same way; reported in config, never `value`
        for d_ in dets:
            d_.set_option("keep_pyramid", 1)
        step()
        finish()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        finish()
        torch.cuda.synchronize()
        keep_fps = C * B * args.steps / (time.perf_counter() - t1)
        for d_ in dets:
            d_.set_option("keep_pyramid", 0)
        step()  # (the records of the detection-only call are what the self-check below reads)
        finish()
        if not args.no_profile:
            for d_ in dets:
                d_.profile()  # these launches are not the timed region's
    ref_fps, ref_verified = None, None
    if world == 1 and not args.ref_arith and not args.keep_pyramid and args.config != 5 and not args.no_latency:
        # the reference-arithmetic tier beside the headline (option arith = 1 with the committed build-host tables: the reference's own
        # rsqrtps / rcpps bits): the same steps, timed the same way, one frame per context checked against the
        # oracle's table tier; reported in config, never `value`
        z_ = np.load(os.path.join(ROOT, "tests", "golden", "x86_rcp_rsqrt.npz"))
        for d_ in dets:
            d_.set_x86_tables(z_["rcp"], z_["rsqrt"])
            d_.set_option("arith", 1)
        step()
        finish()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        finish()
        torch.cuda.synchronize()
        ref_fps = C * B * args.steps / (time.perf_counter() - t1)
        if rank == 0 and not args.no_verify:
            recs_ = [pipes[i].rec[(pipes[i].k - 1) & 1].cpu().numpy() for i in range(C)]
            ref_verified = verify_frames(model, H, W, [frames[i * B:(i + 1) * B] for i in range(C)], recs_, args.cap, nms_params,
                                         [(i % C, (11 + 17 * i) % B) for i in range(min(C, 3))], ref_arith=True)
        for d_ in dets:
            d_.set_option("arith", 0)
        step()  # (the records of the default tier are what the self-check below reads)
        finish()
        if not args.no_profile:
            for d_ in dets:
                d_.profile()
    if rank == 0 and not args.no_latency:
        # one frame through one context, submit -> results on the device (cfg 2 is worded "single frame")
        lat = []
        dets[0].set_option("scale_streams", 1)  # one frame alone: the scales' chains run beside each other
        dets[0].set_option("profile", 0)
        dets[0].set_option("cascade_turns", 0)  # (one context: nobody to take turns with)
        dets[0].set_option("graph", 1)          # the call's ~45 launches replayed as one captured HIP graph (same input buffer)
        with torch.cuda.stream(streams[0]):
            for _ in range(14):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                dets[0].run(frames[:1], 1)
                dets[0].synchronize()
                lat.append(time.perf_counter() - t1)
        latency_ms = 1e3 * float(np.median(lat[4:]))  # (the first call runs plainly, the second captures)
        dets[0].set_option("graph", 0)
        dets[0].set_option("profile", 0 if args.no_profile else 1)
        if not args.no_profile:
            dets[0].profile()
    counts = pipes[0].rec[(pipes[0].k - 1) & 1][:, 0].cpu().numpy()
    verified = None
    if rank == 0 and not args.no_verify and args.config != 5:
        # the timed contexts' own last records (still in their record buffers), three frames of three different contexts
        recs = [pipes[i].rec[(pipes[i].k - 1) & 1].cpu().numpy() for i in range(C)]
        picks = [(i % C, (7 + 31 * i) % B) for i in range(3)]
        verified = verify_frames(model, H, W, [frames[i * B:(i + 1) * B] for i in range(C)], recs, args.cap, nms_params, picks, ref_arith=args.ref_arith)
    strong64 = None
    if rank == 0 and world == 1 and args.config == 2 and not args.no_latency and not args.frames_total and C * B >= 64:
        # BASELINE cfg 3 as worded (64 frames per step) on this one GPU: 2 contexts x 32 frames, same contexts and buffers
        c2 = min(C, 2)
        per = 64 // c2
        if per <= B:
            for d_ in dets:
                d_.set_option("scale_streams", 0 if c2 > 1 else 1)
            def step64():
                for i in range(c2):
                    with torch.cuda.stream(streams[i]):
                        dets[i].run(frames[i * B:i * B + per], per)
                        dets[i].export_detections(pipes[i].rec[0], args.cap)
            for _ in range(2):
                step64()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                step64()
            torch.cuda.synchronize()
            strong64 = 64 * 10 / (time.perf_counter() - t1)
            if not args.no_profile:
                for d_ in dets:
                    d_.profile()
    batch8 = batch8_pipe = None
    if rank == 0 and world == 1 and args.config == 2 and not args.no_latency and not args.frames_total and B >= 8:
        # one GPU's share of BASELINE cfg 3 AS WORDED (64 frames per step over 8 GPUs = 8 frames per GPU and step): one context,
        # 8 frames per call, the scales' chains beside each other, the call replayed as one HIP graph; submit -> synchronise per step
        dets[0].set_option("scale_streams", 1)
        dets[0].set_option("profile", 0)
        dets[0].set_option("cascade_turns", 0)
        dets[0].set_option("graph", 1)
        with torch.cuda.stream(streams[0]):
            for _ in range(4):
                dets[0].run(frames[:8], 8)
                dets[0].export_detections(pipes[0].rec[0], args.cap)
            dets[0].synchronize()
            t1 = time.perf_counter()
            for _ in range(20):
                dets[0].run(frames[:8], 8)
                dets[0].export_detections(pipes[0].rec[0], args.cap)
                dets[0].synchronize()
            batch8 = 8 * 20 / (time.perf_counter() - t1)
        # the same 8-frame calls with consecutive steps in flight at once (every context its own call, no synchronisation between
        # steps, the headline's settings: plain launches, a context's scales on its one stream, the cascades taking turns): what a
        # rank sustains when step k+1 is submitted while step k still runs
        dets[0].set_option("graph", 0)
        for d_ in dets:
            d_.set_option("scale_streams", 0 if C > 1 else 1)
            d_.set_option("profile", 0)
            d_.set_option("cascade_turns", args.turns if args.turns >= 0 else (5 if C > 1 else 0))
        def step8():
            for i in range(C):
                with torch.cuda.stream(streams[i]):
                    dets[i].run(frames[i * B:i * B + 8], 8)
                    dets[i].export_detections(pipes[i].rec[0], args.cap)
        for _ in range(4):
            step8()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            step8()
        torch.cuda.synchronize()
        batch8_pipe = 8 * C * 20 / (time.perf_counter() - t1)
        for d_ in dets:
            d_.set_option("profile", 0 if args.no_profile else 1)
        dets[0].set_option("graph", 0)
        dets[0].set_option("profile", 0 if args.no_profile else 1)
        if not args.no_profile:
            dets[0].profile()
    if rank == 0:
        if world > 1:
            # the first real multi-GPU run must not silently report one rank: every rank's frames of the last step arrived
            assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
            assert gathered is not None and int(gathered.shape[0]) == world * B, \
                "gather: %s rows on rank 0, expected %d ranks x %d frames" % (None if gathered is None else int(gathered.shape[0]), world, B)
            assert all(float(x) > 0 for x in rank_fps.cpu().numpy()), "a rank reported no frames"
        frames_total = C * B * world * args.steps
        fps = frames_total / dt
        b_in = 3 * 4 * H * W
        b_pyr = 4 * det.pyr_floats
        b_frame = b_in + 2 * b_pyr
        # the same sum with the pyramid at the 2 bytes per cell the detection-only call really writes and reads (16-bit threshold-rank
        # cells, columns padded to 8 cells): what THIS data flow makes compulsory, beside SURVEY 8d's 4 bytes per cell
        b_rank = 2 * sum(det.nChns * l.wP * ((l.hP + 7) // 8 * 8) for l in det.levels)
        b_frame_rank = b_in + 2 * b_rank
        metric = "detector FPS @1080p, 8 scales/octave, FACE80 model" if args.config == 2 else \
            "detector FPS, BASELINE.json cfg %d (%dx%d)" % (args.config, W, H)
        out = {
            "metric": metric,
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # the same K steps timed three times back to back on this box (the first is `value`): box-to-box differences of the pool
            # are +-4 %, a region is short; min / median / max say what a single region is worth
            "value_repeats": [round(v, 1) for v in repeat_fps], "value_median": float(np.median(repeat_fps)),
            "value_min": float(min(repeat_fps)), "value_max": float(max(repeat_fps)),
            "latency_ms_batch1": latency_ms,
            "verified_frames": verified,
            "config": {"workload": "%s; %d frames resident in HBM per GPU per step (%d detector contexts x %d frames)" % (cfg["what"], C * B, C, B),
                       "baseline_config": args.config,
                       "frames_per_gpu_per_step": C * B, "contexts": C, "frames_per_launch": B, "levels": len(det.levels),
                       "windows_per_frame": int(sum(l.nWinR * l.nWinC for l in (det.ldcf_levels or det.levels))),
                       "mean_detections_per_frame": float(counts.mean()), "nms": "none (raw detections)" if args.no_nms else "device bbNms maxg .65/min + prune(10)",
                       "gather_record_bytes_per_frame": 4 * (1 + 6 * args.cap), "parallelism": "frames sharded, %d rank(s)" % world,
                       "rccl_ranks": world, "per_rank_fps": [float(x) for x in rank_fps.cpu().numpy()],
                       "pyramid_output": ("float pyramid + 16-bit threshold-rank cells" if args.keep_pyramid else
                                          "16-bit threshold-rank cells only (detection-only call; --keep-pyramid also writes the float levels)") if args.config != 5 else "float (LDCF)",
                       # BASELINE cfg 2 as worded ("single frame") and cfg 3 as worded (64 frames per step, here on ONE GPU)
                       "cfg2_single_frame_latency_ms": latency_ms, "cfg2_single_frame_fps": (1e3 / latency_ms) if latency_ms else None,
                       "cfg3_64_frames_per_step_fps_1gpu": strong64,
                       # one GPU's share of cfg 3 as worded (8 frames per GPU and step), measured on this GPU: 8 x this figure bounds what
                       # eight GPUs give for 64 frames per step (the gather of 8 x 772 bytes per rank comes on top)
                       "batch8_fps_1gpu": batch8,
                       "batch8_pipelined_fps_1gpu": batch8_pipe,   # C contexts x 8 frames in flight, no synchronisation between steps
                       "cfg3_as_worded_8gpu_estimate_fps": (8 * batch8) if batch8 else None,
                       # the Pyramid-returning call (float levels written as well as the rank cells), same steps, timed the same way
                       "keep_pyramid_fps": keep_fps,
                       # the reference-arithmetic tier (the reference's own rsqrtps / rcpps bits, build-host tables), same steps on this box,
                       # and the frames of it checked against the oracle's table tier
                       "ref_arith_fps": ref_fps, "ref_arith_verified_frames": ref_verified,
                       "cfg3_scaling_expectation": "weak scaling (--gpus N: every GPU its own 3 x 96 frames, one 772-byte record gather per frame) is "
                                                   "expected near-linear; strong scaling of cfg 3 as worded (64 frames per step over 8 GPUs = 8 per GPU) is bound by "
                                                   "per-launch floors: 8 x batch8_fps_1gpu (measured on one GPU) against this line's value.  No multi-GPU node was "
                                                   "available: neither curve is measured."},
        }
        if args.frames_total:
            out["config"]["frames_total_per_step"] = args.frames_total
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "traffic_source": "%s (committed rocprofv3 PMC passes of this workload; not measured in this run)" % PMC_FILE}
        if prof:
            tot_ms = sum(v[0] for v in prof.values())
            kb = kernel_bytes_per_frame(det, model)
            # the kernel the roofline block is about: the largest one ALONE on the machine (a property of the kernels, stable from run
            # to run); which kernel has the largest launch-to-finish time beside the other contexts flips between runs and is
            # reported next to it (region_dominant_kernel)
            region_dom = max(prof, key=lambda k: prof[k][0])
            # (with the smoothing launches under their own names this is the cascade's tile kernel at cfg 2: the fat kernel whose
            # time alone bounds a step, DESIGN.md "Speed of light")
            dom = max((k for k in solo if k in prof), key=lambda k: solo[k][0]) if solo else region_dom
            launches = max(prof[dom][1], 1)
            # a kernel may take several launches per batch (one per real scale): its figures are per BATCH of B frames — the
            # summed duration of the launches one batch needs — so that bytes and time cover the same frames
            batches = C * args.steps
            avg_ms = prof[dom][0] / batches
            frames_per_launch = C * B * args.steps / launches
            # dominant kernel of the TIMED REGION: its algorithmic bytes per batch / its duration per batch there (HIP
            # events on the launch stream; with C contexts the launches share the machine with other contexts' kernels, which
            # is how it runs in the product and what produced `value`)
            ach = kb.get(dom, 0) * B / (avg_ms * 1e-3) / 1e9
            # whole hot path: B = B_in + 2*B_pyr per frame over the wall clock of the timed region (kernel times of
            # concurrent contexts overlap, so their sum is not elapsed time)
            path = b_frame * C * B * args.steps / dt / 1e9

            def kernel_line(k):
                ms_b = prof[k][0] / batches
                a_ = kb.get(k, 0) * B / (ms_b * 1e-3) / 1e9
                tr = pmc_traffic(k, B)
                return {"kernel": k, "ms_per_batch": ms_b, "launches_per_batch": prof[k][1] / batches, "achieved": a_, "frac": a_ / HBM_PEAK_GBS,
                        "bytes_moved_frac": (tr / (ms_b * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr else None}
            roof.update({
                "achieved": ach, "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(dom, B),
                # what the kernel really moves (PMC bytes of the committed pass, all its launches of a batch) over its duration per
                # batch here, as a fraction of peak: e.g. the rank-cell tile kernel reads half of the 4 bytes per cell that `achieved`
                # charges it (SURVEY.md 8d's figure)
                "kernel_bytes_moved_frac": (pmc_traffic(dom, B) / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if pmc_traffic(dom, B) else None,
                "kernel": dom, "region_dominant_kernel": region_dom, "kernel_avg_ms": avg_ms, "kernel_launches_per_batch": launches / batches, "kernel_share": prof[dom][0] / tot_ms,
                "kernel_bytes_per_launch": kb.get(dom, 0) * B,
                "measured": "inside the timed region (%d contexts sharing the GPU); per batch of %d frames" % (C, B),
                "path_bytes_per_frame": b_frame, "path_achieved": path, "path_frac": path / HBM_PEAK_GBS,
                "path_rank_bytes_per_frame": b_frame_rank, "path_frac_rank_bytes": path * b_frame_rank / b_frame / HBM_PEAK_GBS,
                # what the whole path really moves per frame (every kernel of a step, same committed PMC passes as `traffic`) and how
                # that compares with SURVEY 8d's algorithmic bytes: > 1 = intermediates and re-reads, the first thing to cut
                "traffic_path_bytes_per_frame": pmc_path_bytes_per_frame(B),
                "wasted_traffic_ratio": (pmc_path_bytes_per_frame(B) / b_frame) if pmc_path_bytes_per_frame(B) else None,
                "path_bytes_moved_frac": (pmc_path_bytes_per_frame(B) * C * B * args.steps / dt / 1e9 / HBM_PEAK_GBS) if pmc_path_bytes_per_frame(B) else None,
                # the three largest kernels of the region, same figures (two of them are within a few percent of each other)
                "top_kernels": [kernel_line(k) for k in sorted(prof, key=lambda k: -prof[k][0])[:3]],
                # summed over the C contexts of a step (they run concurrently: the sum exceeds ms_per_step when C > 1)
                "kernels_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])},
            })
            if solo.get(dom):
                # the same kernel with the GPU to itself (same process, buffers and event mechanism; 3 launches right after the
                # timed region): a property of the kernel, next to the figure of the configuration that produced `value`
                s_ms = solo[dom][0] / 3  # per batch: 3 solo runs
                s_ach = kb.get(dom, 0) * B / (s_ms * 1e-3) / 1e9
                roof["solo"] = {"note": "same kernel, one context alone on the GPU, 3 batches after the timed region; per batch",
                                "kernel_avg_ms": s_ms, "achieved": s_ach, "frac": s_ach / HBM_PEAK_GBS,
                                "kernel_bytes_moved_frac": (pmc_traffic(dom, B) / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if pmc_traffic(dom, B) else None,
                                "kernels_ms_per_launch": {k: round(v[0] / max(v[1], 1), 4) for k, v in sorted(solo.items(), key=lambda kv: -kv[1][0])},
                                "kernels_ms_per_batch": {k: round(v[0] / 3, 4) for k, v in sorted(solo.items(), key=lambda kv: -kv[1][0])}}
                # every kernel's HBM rate alone: committed PMC traffic of all its launches in a step (profiles/r02_pmc_traffic.json,
                # measured with this batch size) over the time of those launches here; GB/s (fraction of the 8 TB/s peak)
                rates = {}
                for k, v in solo.items():
                    tb = pmc_traffic(k, B)
                    if tb and v[0] > 0:
                        r = tb * 3 / (v[0] * 1e-3) / 1e9  # 3 solo steps
                        rates[k] = [round(r, 1), round(r / HBM_PEAK_GBS, 3)]
                if rates:
                    roof["solo"]["kernels_hbm_GBps_frac"] = dict(sorted(rates.items(), key=lambda kv: -solo[kv[0]][0]))
        else:
            path = b_frame * fps / world / 1e9
            roof.update({"path_bytes_per_frame": b_frame, "path_achieved": path, "path_frac": path / HBM_PEAK_GBS,
                         "path_rank_bytes_per_frame": b_frame_rank, "path_frac_rank_bytes": path * b_frame_rank / b_frame / HBM_PEAK_GBS,
                         "note": "per-kernel events disabled: whole path on the wall clock only"})
        # what this box delivers to a plain stream: a device-to-device copy of 256 MiB (read + write), outside the timed region —
        # the practical ceiling beside the 8 TB/s of `peak` (profiles/ubench/copy_bw.py: ~4.8 TB/s)
        try:
            cx = torch.empty(1 << 26, dtype=torch.float32, device=dev)
            cy = torch.empty_like(cx)
            for _ in range(2):
                cy.copy_(cx)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                cy.copy_(cx)
            e1.record()
            torch.cuda.synchronize()
            roof["stream_copy_GBps"] = round(8 * 2 * cx.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
            del cx, cy
        except Exception:  # (a reported extra, never a reason to lose the line)
            roof["stream_copy_GBps"] = None
        out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, base_np, H, W)
        else:
            out["cpu_baseline"] = None
        if gathered is not None and world > 1:
            out["config"]["gathered_records"] = int(gathered.shape[0])
        print(json.dumps(out))
    pool.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
