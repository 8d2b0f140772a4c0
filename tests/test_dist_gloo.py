"""The N>1 path on CPU: two gloo ranks each own a contiguous shard of frames
(acf_amd.dist.shard_range), build the fixed-capacity detection records that
acf_hip_export_detections writes on a GPU (here filled from the oracle's
detections — the checker stands in for the device, the plumbing under test is
the sharding + gather), gather them on rank 0 and decode.  The gathered list
must equal the unsharded run frame by frame."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAP = 64
N_FRAMES = 5  # odd on purpose: shards of 3 and 2


def _frame_records():
    from acf_amd import synth
    from acf_amd.dist import detections_to_record
    from oracle import binding as ob
    model = synth.make_model(seed=3, name="TINY", nTrees=96)
    H, W = 64, 80
    plan = ob.Plan(model, H, W, 3)
    recs = []
    for i in range(N_FRAMES):
        pyr, _, _ = ob.chns_pyramid(plan, synth.make_frame(17 + i, H, W, "luv"))
        det, _ = ob.detect(plan, pyr)
        recs.append(detections_to_record(det, CAP))
    return np.stack(recs)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from acf_amd.dist import gather_records, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        allrec = _frame_records()
        a, b = shard_range(N_FRAMES, world, rank)
        # ranks may own different frame counts; the gather needs equal shapes -> pad to the largest shard
        per = (N_FRAMES + world - 1) // world
        mine = np.zeros((per, allrec.shape[1]), dtype=np.int32)
        mine[:, 0] = -1  # count -1 marks a padding row
        mine[:b - a] = allrec[a:b]
        got = gather_records(torch.from_numpy(mine), world, rank)
        if rank == 0:
            g = got.numpy()
            g = g[g[:, 0] >= 0]
            q.put(("ok", g.tobytes(), allrec.tobytes()))
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_equals_unsharded():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    tag, got, want = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tag == "ok" and got == want


def test_shard_range_partitions():
    from acf_amd.dist import shard_range
    for n in (1, 5, 8, 64, 67):
        for world in (1, 2, 3, 8):
            cuts = [shard_range(n, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_record_roundtrip():
    from acf_amd import capi
    from acf_amd.dist import detections_to_record, records_to_detections
    d = np.zeros(3, dtype=capi.DET_DTYPE)
    d["x"], d["y"], d["w"], d["h"], d["scale"] = [1, 2, 3], [4, 5, 6], [80, 90, 100], [80, 90, 100], [0, 3, 7]
    d["score"] = np.asarray([1.5, -0.25, 26.0038], dtype=np.float32)
    rec = detections_to_record(d, 8)
    out = records_to_detections(rec, 8)
    assert len(out) == 3 and out[2][:4] == (3, 6, 100, 100) and out[2][5] == 7
    assert np.float32(out[2][4]) == np.float32(26.0038)


def _pipe_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from acf_amd.dist import RecordGather
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frames, width = 3, 1 + 6 * 4
        pipe = RecordGather(frames, width, world, rank, torch.device("cpu"))
        seen = []
        for batch in range(4):
            rec = pipe.buffer()          # waits for this buffer's previous gather
            if rank == 0 and batch >= 2:
                seen.append(pipe.out[batch & 1].clone())  # the gather issued two batches ago has landed here
            rec.copy_(torch.full((frames, width), 1000 * rank + 10 * batch, dtype=torch.int32) + torch.arange(frames, dtype=torch.int32)[:, None])
            pipe.submit()
        last = pipe.finish()
        if rank == 0:
            seen.append(pipe.out[0].clone())  # batch 2
            seen.append(last.clone())         # batch 3
            q.put(("ok", [s.numpy().tolist() for s in seen]))
        else:
            assert last is None
    finally:
        dist.destroy_process_group()


def test_pipelined_gather_keeps_every_batch_in_order():
    """RecordGather (bench.py's N>1 exchange): asynchronous gathers over two buffers deliver batches 0..3 complete and in
    rank-major frame order on rank 0."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipe_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    tag, seen = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tag == "ok" and len(seen) == 4
    for batch, g in enumerate(seen):
        g = np.asarray(g)
        assert g.shape == (6, 25)
        for r in range(2):
            for f in range(3):
                assert (g[3 * r + f] == 1000 * r + 10 * batch + f).all(), (batch, r, f)
