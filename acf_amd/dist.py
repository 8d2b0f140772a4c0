"""Multi-GPU plumbing: frames shard across ranks (independent frames, no
data-path collective); the only exchange is one gather of the fixed-capacity
detection records to rank 0.  Payload is KBs per frame, so this
is latency- not bandwidth-bound: a single gather, no ring all-reduce.

Record layout per frame (written by acf_hip_export_detections):
int32 [count, cap x {x, y, w, h, score bits, level}].
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_frames, world, rank):
    """Contiguous block of frames owned by `rank` (blocks differ by at most one frame)."""
    base, extra = divmod(n_frames, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_records(rec, world, rank, dst=0):
    """Gather every rank's [frames, 1+6*cap] int32 record tensor on `dst`.

    Returns the [world*frames, 1+6*cap] tensor on dst (rank-major order = frame
    order when frames are sharded with shard_range), None elsewhere."""
    if world == 1:
        return rec
    out = [torch.empty_like(rec) for _ in range(world)] if rank == dst else None
    dist.gather(rec, out, dst=dst)
    return torch.cat(out, dim=0) if rank == dst else None


def records_to_detections(rec_row, cap):
    """Decode one frame's record (numpy int32 row) into a list of (x, y, w, h, score, level)."""
    n = min(int(rec_row[0]), cap)
    body = np.asarray(rec_row[1:1 + 6 * cap]).reshape(cap, 6)[:n]
    scores = body[:, 4].astype(np.int32).view(np.float32)
    return [(int(b[0]), int(b[1]), int(b[2]), int(b[3]), float(s), int(b[5])) for b, s in zip(body, scores)]


def detections_to_record(dets, cap):
    """Inverse of records_to_detections for host-side tests: dets is a structured array (capi.DET_DTYPE)."""
    rec = np.zeros(1 + 6 * cap, dtype=np.int32)
    rec[0] = len(dets)
    n = min(len(dets), cap)
    body = rec[1:].reshape(cap, 6)
    for k, name in enumerate(("x", "y", "w", "h")):
        body[:n, k] = dets[name][:n]
    body[:n, 4] = dets["score"][:n].view(np.int32)
    body[:n, 5] = dets["scale"][:n]
    return rec
