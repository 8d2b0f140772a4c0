#!/usr/bin/env python
"""Pairwise co-run matrix of the path's kernels (cfg 2, 96 frames per launch).

For every pair (X, Y) of kernel classes of `acf_hip_run` — image smoothing (+ fused gradMag), gradMag of the small scales, the
convTri x pass, the y pass + channel cells, the image resamples, the level kernel, the cascade's tile kernel — X runs in a loop on
one context's stream and Y on another context's, alone and side by side.  Per pair: each side's slowdown beside the other and
the summed rate relative to running them one after the other (2.0 = the two do not see each other, 1.0 = they share the machine
like one stream would).

How a class is run on its own without touching the library: `acf_hip_run` of a context is CAPTURED into a HIP graph from the
outside (hipStreamBeginCapture on the context's stream; the call makes ~45 launches and no host round trip), the graph's kernel
nodes are named with hipKernelNameRefByPtr, every kernel node that is not of the class is removed from the graph and the
survivors (and every memset node: counters, flags) are chained in their original order.  Replaying that graph launches exactly the
library's kernels of that class with the library's own arguments, on buffers that hold a finished run's data.

    python profiles/ubench/corun_matrix.py > gpurun_out/corun_matrix.json
"""
import ctypes as C
import json
import os
import re
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acf_amd import capi, synth  # noqa: E402
from acf_amd.detector import HipDetector  # noqa: E402

hip = C.CDLL("libamdhip64.so")
hip.hipKernelNameRefByPtr.restype = C.c_char_p
hip.hipKernelNameRefByPtr.argtypes = [C.c_void_p, C.c_void_p]


class Dim3(C.Structure):
    _fields_ = [("x", C.c_uint), ("y", C.c_uint), ("z", C.c_uint)]


class KernelNodeParams(C.Structure):
    _fields_ = [("blockDim", Dim3), ("extra", C.c_void_p), ("func", C.c_void_p), ("gridDim", Dim3), ("kernelParams", C.c_void_p),
                ("sharedMemBytes", C.c_uint)]


def chk(rc, what):
    if rc:
        raise RuntimeError("%s: hip error %d" % (what, rc))


CLASSES = [
    ("smooth", r"k_smooth_(vec|grad|verify)"),
    ("gradmag", r"k_grad_mag_vec"),
    ("tri_x", r"k_tri_x5v"),
    ("triy_chns", r"k_triy_chns"),
    ("resample", r"k_resample_(strip|half)"),
    ("level", r"k_level_all"),
    ("tile", r"k_cascade_tile3"),
]

H, W = 1080, 1920
B = int(os.environ.get("BATCH", "96"))
PERSIST = int(os.environ.get("PERSIST", "0"))
TARGET_MS = float(os.environ.get("TARGET_MS", "80"))


def make_frames(dev, seed0):
    base = torch.from_numpy(np.stack([synth.make_frame(seed0 + i, H, W, "luv") for i in range(2)])).to(dev)
    frames = torch.empty((B, 3, W, H), dtype=torch.float32, device=dev)
    for i in range(B):
        frames[i] = torch.roll(base[i % 2], shifts=(37 * (i // 2), 53 * (i // 2)), dims=(1, 2))
    return frames


def capture(det, stream, frames):
    g = C.c_void_p()
    chk(hip.hipStreamBeginCapture(C.c_void_p(stream.cuda_stream), 2), "hipStreamBeginCapture")  # 2: relaxed
    det.run(frames, B)
    chk(hip.hipStreamEndCapture(C.c_void_p(stream.cuda_stream), C.byref(g)), "hipStreamEndCapture")
    return g


def graph_nodes(g):
    n = C.c_size_t(0)
    chk(hip.hipGraphGetNodes(g, None, C.byref(n)), "hipGraphGetNodes")
    nodes = (C.c_void_p * n.value)()
    chk(hip.hipGraphGetNodes(g, nodes, C.byref(n)), "hipGraphGetNodes")
    ne = C.c_size_t(0)
    chk(hip.hipGraphGetEdges(g, None, None, C.byref(ne)), "hipGraphGetEdges")
    fr, to = (C.c_void_p * max(ne.value, 1))(), (C.c_void_p * max(ne.value, 1))()
    if ne.value:
        chk(hip.hipGraphGetEdges(g, fr, to, C.byref(ne)), "hipGraphGetEdges")
    # topological order (a one-stream capture is a chain; be general)
    ids = [nodes[i] for i in range(n.value)]
    indeg = {v: 0 for v in ids}
    out = {v: [] for v in ids}
    for i in range(ne.value):
        out[fr[i]].append(to[i])
        indeg[to[i]] += 1
    order, ready = [], [v for v in ids if indeg[v] == 0]
    while ready:
        v = ready.pop(0)
        order.append(v)
        for w in out[v]:
            indeg[w] -= 1
            if indeg[w] == 0:
                ready.append(w)
    assert len(order) == len(ids)
    return order


def node_info(node, stream):
    t = C.c_int(-1)
    chk(hip.hipGraphNodeGetType(C.c_void_p(node), C.byref(t)), "hipGraphNodeGetType")
    name = ""
    if t.value == 0:
        p = KernelNodeParams()
        chk(hip.hipGraphKernelNodeGetParams(C.c_void_p(node), C.byref(p)), "hipGraphKernelNodeGetParams")
        nm = hip.hipKernelNameRefByPtr(C.c_void_p(p.func), C.c_void_p(stream.cuda_stream))
        name = (nm or b"?").decode()
    return t.value, name


def class_graph(det, stream, frames, pattern):
    """Capture a run, keep the kernel nodes matching `pattern` and every memset node, chain them in order -> (exec, kernels kept)."""
    g = capture(det, stream, frames)
    order = graph_nodes(g)
    keep, names = [], []
    for v in order:
        t, name = node_info(v, stream)
        if t == 2 or (t == 0 and re.search(pattern, name)):
            keep.append(v)
            if t == 0:
                names.append(re.search(r"k_[a-z0-9_]+", name).group(0))
        else:
            chk(hip.hipGraphDestroyNode(C.c_void_p(v)), "hipGraphDestroyNode")
    # re-chain the survivors (removing a node removed its edges); edges that survived are removed first to avoid duplicates
    ne = C.c_size_t(0)
    chk(hip.hipGraphGetEdges(g, None, None, C.byref(ne)), "hipGraphGetEdges")
    if ne.value:
        fr, to = (C.c_void_p * ne.value)(), (C.c_void_p * ne.value)()
        chk(hip.hipGraphGetEdges(g, fr, to, C.byref(ne)), "hipGraphGetEdges")
        chk(hip.hipGraphRemoveDependencies(g, fr, to, ne), "hipGraphRemoveDependencies")
    if len(keep) > 1:
        fr = (C.c_void_p * (len(keep) - 1))(*keep[:-1])
        to = (C.c_void_p * (len(keep) - 1))(*keep[1:])
        chk(hip.hipGraphAddDependencies(g, fr, to, C.c_size_t(len(keep) - 1)), "hipGraphAddDependencies")
    ex = C.c_void_p()
    chk(hip.hipGraphInstantiate(C.byref(ex), g, None, None, C.c_size_t(0)), "hipGraphInstantiate")
    return ex, names, g


def main():
    dev = torch.device("cuda", 0)
    model = synth.make_model(seed=1, name="FACE80")
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    frames = [make_frames(dev, 1), make_frames(dev, 11)]
    dets = [HipDetector(model, H, W, 3, max_batch=B, max_hits=8192, device=0, stream=s.cuda_stream) for s in streams]
    nms = capi.make_nms(type="maxg", overlap=0.65, ovrDnm="min", prune=True, maxCount=10, pruneRatio=0.0)
    for d in dets:
        d.set_option("scale_streams", 0)
        d.set_option("keep_pyramid", 0)
        d.set_option("tile_persist", PERSIST)
        d.set_option("cascade_turns", 0)
        d.set_option("graph", 0)
        d.set_nms(nms)
    for i, d in enumerate(dets):
        with torch.cuda.stream(streams[i]):
            d.run(frames[i], B)
            d.run(frames[i], B)
    torch.cuda.synchronize()
    want = [d.detections(0)[0].tobytes() for d in dets]

    execs = [{}, {}]
    kept = {}
    graphs = []
    for i in range(2):
        for cname, pat in CLASSES:
            ex, names, g = class_graph(dets[i], streams[i], frames[i], pat)
            execs[i][cname] = ex
            graphs.append(g)
            kept[cname] = names
        # the whole run as one graph too (its replay must reproduce the detections: the capture itself is sound)
        g = capture(dets[i], streams[i], frames[i])
        ex = C.c_void_p()
        chk(hip.hipGraphInstantiate(C.byref(ex), g, None, None, C.c_size_t(0)), "hipGraphInstantiate")
        execs[i]["(whole run)"] = ex
        graphs.append(g)
    torch.cuda.synchronize()

    def launch(i, cname, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(streams[i])
        for _ in range(n):
            chk(hip.hipGraphLaunch(execs[i][cname], C.c_void_p(streams[i].cuda_stream)), "hipGraphLaunch")
        e1.record(streams[i])
        return e0, e1

    # sanity: replaying the whole-run graph leaves the same detections
    for i in range(2):
        launch(i, "(whole run)", 1)
    torch.cuda.synchronize()
    for i, d in enumerate(dets):
        assert d.detections(0)[0].tobytes() == want[i], "replayed graph differs from the plain run"

    names = [c for c, _ in CLASSES]
    alone = {}
    for cname in names + ["(whole run)"]:
        for i in range(2):
            launch(i, cname, 2)
        torch.cuda.synchronize()
        e0, e1 = launch(0, cname, 8)
        torch.cuda.synchronize()
        alone[cname] = e0.elapsed_time(e1) / 8
    reps = {c: max(2, int(round(TARGET_MS / max(alone[c], 1e-3)))) for c in alone}
    # alone again with the pair runs' repetition counts (same launch-queue depth as beside another class)
    for cname in alone:
        e0, e1 = launch(0, cname, reps[cname])
        torch.cuda.synchronize()
        alone[cname] = e0.elapsed_time(e1) / reps[cname]

    def launch_marked(i, cname, n, ref):
        """n replays with an event after each; returns the completion times in ms after `ref` (rep 0's start first)."""
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        evs[0].record(streams[i])
        for k in range(n):
            chk(hip.hipGraphLaunch(execs[i][cname], C.c_void_p(streams[i].cuda_stream)), "hipGraphLaunch")
            evs[k + 1].record(streams[i])
        return lambda: np.array([ref.elapsed_time(e) for e in evs])

    def progress(t, at):
        """replays completed at time `at` (fractional: linear inside a replay); t[0] = start of replay 0, t[k] = end of replay k - 1"""
        return float(np.interp(at, t, np.arange(len(t))))

    # Side by side: both loops are given twice the work of TARGET_MS, so that neither runs dry inside the window that is
    # analysed — from the later of the two starts to the earlier of the two ends.  Inside it: replays completed by each side
    # x that class's time alone = machine time's worth of work done; divided by the window = rate_sum (1 = as if one stream ran
    # the two loops in turn, 2 = the two do not see each other); slowdown = time per replay in the window / alone.
    pairs = {}
    for a in names:
        for b in names:
            if names.index(b) < names.index(a):
                continue
            torch.cuda.synchronize()
            ref = torch.cuda.Event(enable_timing=True)
            ref.record(streams[0])
            streams[1].wait_event(ref)
            ta_f = launch_marked(0, a, 2 * reps[a], ref)
            tb_f = launch_marked(1, b, 2 * reps[b], ref)
            torch.cuda.synchronize()
            ta, tb = ta_f(), tb_f()
            w0, w1 = max(ta[0], tb[0]) + 0.5, min(ta[-1], tb[-1])
            na, nb = progress(ta, w1) - progress(ta, w0), progress(tb, w1) - progress(tb, w0)
            win = w1 - w0
            pairs["%s|%s" % (a, b)] = dict(window_ms=round(win, 2), replays=[round(na, 2), round(nb, 2)],
                                           slowdown=[round(win / max(na, 1e-6) / alone[a], 3), round(win / max(nb, 1e-6) / alone[b], 3)],
                                           rate=[round(na * alone[a] / win, 3), round(nb * alone[b] / win, 3)],
                                           rate_sum=round((na * alone[a] + nb * alone[b]) / win, 3))
    whole = launch(0, "(whole run)", 6), launch(1, "(whole run)", 6)
    torch.cuda.synchronize()
    tw = [w[0].elapsed_time(w[1]) / 6 for w in whole]
    out = dict(what="cfg 2, %d frames per launch, tile_persist=%d: every kernel class of acf_hip_run replayed alone and beside another class on a second context "
                    "(graphs cut out of a captured run); rate_sum: 2 = no interference, 1 = like one stream" % (B, PERSIST),
               kernels_in_class=kept, alone_ms_per_batch={k: round(v, 4) for k, v in alone.items()},
               alone_us_per_frame={k: round(v * 1000 / B, 2) for k, v in alone.items()}, pairs=pairs,
               whole_run_two_contexts_ms=[round(t, 3) for t in tw], whole_run_rate_sum=round(alone["(whole run)"] / tw[0] + alone["(whole run)"] / tw[1], 3))
    print(json.dumps(out, indent=1))
    for d in dets:
        d.close()


if __name__ == "__main__":
    main()
