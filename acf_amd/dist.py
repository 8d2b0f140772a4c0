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


class RecordGather:
    """The same gather, pipelined: buffers are allocated once (the destination is one [world*frames, ...] tensor whose
    per-rank slices are the gather outputs, so nothing is concatenated), the collective is issued asynchronously and
    waited for when the next one is issued (or at `finish`).  With two record buffers per rank the gather of batch k
    overlaps the kernels of batch k+1; rank `dst` still sees every batch, in order."""

    def __init__(self, frames, width, world, rank, device, dst=0, dtype=torch.int32):
        self.world, self.rank, self.dst = world, rank, dst
        self.rec = [torch.zeros((frames, width), dtype=dtype, device=device) for _ in range(2)]
        self.out = [torch.zeros((world * frames, width), dtype=dtype, device=device) if rank == dst and world > 1 else None for _ in range(2)]
        self.work = [None, None]
        self.k = 0

    def buffer(self):
        """Record buffer the next batch's export should write (its previous gather has completed)."""
        i = self.k & 1
        if self.work[i] is not None:
            self.work[i].wait()
            self.work[i] = None
        return self.rec[i]

    def submit(self):
        """Start gathering the buffer returned by the last buffer() call."""
        i = self.k & 1
        self.k += 1
        if self.world == 1:
            return
        frames = self.rec[i].shape[0]
        outs = list(self.out[i].split(frames, dim=0)) if self.rank == self.dst else None
        self.work[i] = dist.gather(self.rec[i], outs, dst=self.dst, async_op=True)

    def finish(self):
        """Wait for everything in flight; returns the last gathered tensor on dst (the record buffer itself when world == 1)."""
        for i in (0, 1):
            if self.work[i] is not None:
                self.work[i].wait()
                self.work[i] = None
        last = (self.k - 1) & 1
        if self.world == 1:
            return self.rec[last]
        return self.out[last] if self.rank == self.dst else None


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
