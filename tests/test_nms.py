"""bbNms (max / maxg) + ObjectDetector::prune (bbNms.cpp:111-192,229-304, ObjectDetector.cpp:28-44).

CPU: the oracle's restatement (oracle/acf_oracle.c:acfo_nms) against the properties the reference code implies.
GPU: the device kernel (k_nms: acf_hip_op_nms on a host list, acf_hip_set_nms inside the pipeline) bit-for-bit against
the oracle, ties included (equal scores keep their input order in both; the reference's std::sort leaves it open)."""
import numpy as np
import pytest

from acf_amd import capi, synth


def boxes_scores(seed, n, quant=0.5, span=(300, 200)):
    u = synth.uniform(seed, n * 5, 3).reshape(n, 5)
    boxes = np.stack([(u[:, 0] * span[0]).astype(np.int32), (u[:, 1] * span[1]).astype(np.int32),
                      20 + (u[:, 2] * 60).astype(np.int32), 20 + (u[:, 3] * 60).astype(np.int32)], axis=1).astype(np.int32)
    scores = np.floor(u[:, 4] * 30 / quant) * quant - 5.0 if quant else u[:, 4] * 30 - 5.0
    return boxes, scores.astype(np.float64)


def iou(a, b, union):
    iw = min(a[0] + a[2], b[0] + b[2]) - max(a[0], b[0])
    ih = min(a[1] + a[3], b[1] + b[3]) - max(a[1], b[1])
    if iw <= 0 or ih <= 0:
        return 0.0
    o = float(iw * ih)
    return o / ((a[2] * a[3] + b[2] * b[3] - o) if union else min(a[2] * a[3], b[2] * b[3]))


@pytest.mark.parametrize("typ", ["max", "maxg"])
@pytest.mark.parametrize("ovr", ["union", "min"])
def test_oracle_nms_properties(oracle, typ, ovr):
    boxes, scores = boxes_scores(11, 300)
    q = capi.make_nms(type=typ, overlap=0.4, ovrDnm=ovr)
    keep = list(oracle.nms(boxes, scores, q))
    assert len(set(keep)) == len(keep) and 0 < len(keep) < 300
    ks = scores[keep]
    assert np.all(ks[:-1] >= ks[1:])                                   # score order
    for a, b in zip(keep[:-1], keep[1:]):
        assert scores[a] > scores[b] or a < b                          # ties keep input order
    kept = set(keep)
    order = sorted(range(300), key=lambda i: (-scores[i], i))
    rank = {i: r for r, i in enumerate(order)}
    for j in range(300):
        above = [i for i in order[:rank[j]] if iou(boxes[i], boxes[j], ovr == "union") > 0.4]
        if typ == "maxg":
            above = [i for i in above if i in kept]                    # greedy: only survivors suppress
        assert (j in kept) == (len(above) == 0), j
    # the best box always survives; "none" returns the input
    assert keep[0] == order[0]
    assert list(oracle.nms(boxes, scores, capi.make_nms(type="none"))) == list(range(300))


def test_oracle_threshold_and_prune(oracle):
    boxes, scores = boxes_scores(12, 200)
    base = list(oracle.nms(boxes, scores, capi.make_nms(type="maxg", overlap=0.3, ovrDnm="union")))
    thr = float(np.median(scores))
    kt = list(oracle.nms(boxes, scores, capi.make_nms(type="maxg", overlap=0.3, ovrDnm="union", thr=thr)))
    assert all(scores[i] >= thr for i in kt) and kt == [i for i in base if scores[i] >= thr][:len(kt)]
    # prune (ObjectDetector.cpp:30-42): at most maxCount, and ONE box past the first score below scores[0] * ratio
    s0 = scores[base[0]]
    for mc, ratio in ((7, 0.0), (7, 0.9), (1, 0.5), (1000, 0.8), (2, 2.0)):
        kp = list(oracle.nms(boxes, scores, capi.make_nms(type="maxg", overlap=0.3, ovrDnm="union", prune=True, maxCount=mc, pruneRatio=ratio)))
        L = min(mc, len(base))
        low = [i for i in range(1, L) if scores[base[i]] < s0 * ratio]
        want = 1 if L < 2 else (low[0] + 1 if low else L)
        assert kp == base[:want], (mc, ratio)
    # a single box is never pruned, an empty list stays empty
    assert list(oracle.nms(boxes[:1], scores[:1], capi.make_nms(prune=True, maxCount=0))) == [0]
    assert list(oracle.nms(boxes[:0], scores[:0], capi.make_nms())) == []


@pytest.fixture(scope="module")
def dev():
    from acf_amd.detector import HipDetector
    return HipDetector()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 63, 300, 512, 513, 1025, 2048])
@pytest.mark.parametrize("typ,ovr,overlap,quant", [("maxg", "min", 0.65, 0.5), ("max", "union", 0.5, 0.5), ("maxg", "union", 0.3, 0), ("max", "min", 0.2, 2.0)])
def test_gpu_op_nms_matches_oracle(dev, oracle, n, typ, ovr, overlap, quant):
    boxes, scores = boxes_scores(100 + n, n, quant, span=(1900, 1060))
    for prune in (False, True):
        q = capi.make_nms(type=typ, overlap=overlap, ovrDnm=ovr, thr=float(np.percentile(scores, 10)) if n > 2 else -1e300,
                          prune=prune, maxCount=25, pruneRatio=0.6)
        want = oracle.nms(boxes, scores, q)
        got = dev.op_nms(boxes, scores, q)
        assert np.array_equal(got, want), (n, typ, ovr, prune, len(got), len(want))


@pytest.mark.gpu
def test_gpu_op_nms_edges(dev, oracle):
    from acf_amd.detector import HipError
    b, s = boxes_scores(5, 10)
    assert list(dev.op_nms(b[:0], s[:0], capi.make_nms())) == []
    assert list(dev.op_nms(b, s, capi.make_nms(type="none"))) == list(range(10))
    # identical boxes, identical scores: only the first survives (tie order = input order)
    bb = np.tile(b[:1], (50, 1))
    assert list(dev.op_nms(bb, np.zeros(50), capi.make_nms(type="maxg", overlap=0.5))) == [0]
    # disjoint boxes: everything survives, in score order, stable on ties
    grid = np.array([[100 * (i % 10), 100 * (i // 10), 50, 50] for i in range(60)], np.int32)
    sc = np.array([float(i % 7) for i in range(60)])
    assert np.array_equal(dev.op_nms(grid, sc, capi.make_nms(type="max", overlap=0.1)), oracle.nms(grid, sc, capi.make_nms(type="max", overlap=0.1)))
    big_b, big_s = boxes_scores(6, 2049)
    with pytest.raises(HipError):
        dev.op_nms(big_b, big_s, capi.make_nms())


@pytest.mark.gpu
def test_gpu_pipeline_nms_before_export(oracle):
    """acf_hip_set_nms: detections() and the exported records are the survivors of bbNms + prune of the raw detections."""
    import torch
    from acf_amd.detector import HipDetector
    from acf_amd.dist import records_to_detections
    H, W = 240, 320
    model = synth.make_model(seed=3, name="TINY", nTrees=96, cascThr=-2.0)
    frames = np.stack([synth.make_frame(70 + i, H, W, "luv") for i in range(3)])
    det = HipDetector(model, H, W, 3, max_batch=3, max_hits=1 << 14)
    fr = torch.from_numpy(frames).cuda()
    det.run(fr)
    raw = [det.detections(f)[0] for f in range(3)]
    assert all(50 < len(r) <= 2048 for r in raw)
    q = capi.make_nms(type="maxg", overlap=0.65, ovrDnm="min", prune=True, maxCount=40, pruneRatio=0.1)
    det.set_nms(q)
    det.run(fr)
    cap = 64
    rec = torch.zeros((3, 1 + 6 * cap), dtype=torch.int32, device="cuda")
    det.export_detections(rec, cap)
    det.synchronize()
    rec = rec.cpu().numpy()
    for f in range(3):
        r = raw[f]
        boxes = np.stack([r["x"], r["y"], r["w"], r["h"]], axis=1)
        keep = oracle.nms(boxes, r["score"].astype(np.float64), q)
        got, _ = det.detections(f)
        assert 1 <= len(keep) < len(r)
        assert got.tobytes() == r[keep].tobytes(), f
        out = records_to_detections(rec[f], cap)
        assert rec[f][0] == len(keep) and [(d[0], d[1], d[2], d[3], d[5]) for d in out] == [(int(r[i]["x"]), int(r[i]["y"]), int(r[i]["w"]), int(r[i]["h"]), int(r[i]["scale"])) for i in keep]
    det.set_nms(None)
    det.run(fr)
    assert det.detections(1)[0].tobytes() == raw[1].tobytes()


@pytest.mark.gpu
def test_gpu_nms_with_sub_batch_streams_and_replan(oracle):
    """Device NMS state reaches the sub-batch child contexts (option streams > 1) whether acf_hip_set_nms comes before the
    plan, after it, or the context is re-planned (children are destroyed and recreated)."""
    import torch
    from acf_amd.detector import HipDetector
    H, W = 120, 160
    model = synth.make_model(seed=3, name="TINY", nTrees=96, cascThr=-2.0)
    frames = np.stack([synth.make_frame(90 + i, H, W, "luv") for i in range(4)])
    fr = torch.from_numpy(frames).cuda()
    q = capi.make_nms(type="maxg", overlap=0.65, ovrDnm="min", prune=True, maxCount=15, pruneRatio=0.1)
    ref = HipDetector(model, H, W, 3, max_batch=4, max_hits=1 << 14)
    ref.run(fr)
    raw = [ref.detections(f)[0] for f in range(4)]
    want = []
    for r in raw:
        keep = oracle.nms(np.stack([r["x"], r["y"], r["w"], r["h"]], axis=1), r["score"].astype(np.float64), q)
        assert 1 <= len(keep) < len(r)
        want.append(r[keep].tobytes())
    det = HipDetector(streams=2)
    det.set_model(model)
    det.set_nms(q)                      # before any plan: no children exist yet
    det.plan(H, W, 3, max_batch=4, max_hits=1 << 14)
    det.run(fr)
    assert [det.detections(f)[0].tobytes() for f in range(4)] == want
    assert [det.raw_detections(f).tobytes() for f in range(4)] == [r.tobytes() for r in raw]
    det.plan(H, W, 3, max_batch=4, max_hits=1 << 13)   # re-plan: new children
    det.run(fr)
    assert [det.detections(f)[0].tobytes() for f in range(4)] == want
    det.set_nms(None)
    det.run(fr)
    assert [det.detections(f)[0].tobytes() for f in range(4)] == [r.tobytes() for r in raw]


@pytest.mark.gpu
def test_gpu_hits_follow_the_survivors(oracle):
    """With the device NMS on, hit i of acf_hip_get_hits is the window behind detection i (the survivors' order)."""
    import torch
    from acf_amd.detector import HipDetector
    H, W = 240, 320
    model = synth.make_model(seed=3, name="TINY", nTrees=96, cascThr=-2.0)
    fr = torch.from_numpy(np.stack([synth.make_frame(70 + i, H, W, "luv") for i in range(2)])).cuda()
    det = HipDetector(model, H, W, 3, max_batch=2, max_hits=1 << 14)
    det.run(fr)
    raw = [det.detections(f) for f in range(2)]
    q = capi.make_nms(type="maxg", overlap=0.65, ovrDnm="min", prune=True, maxCount=40, pruneRatio=0.1)
    det.set_nms(q)
    det.run(fr)
    for f in range(2):
        d, h = det.detections(f)
        rd, rh = raw[f]
        keep = oracle.nms(np.stack([rd["x"], rd["y"], rd["w"], rd["h"]], axis=1), rd["score"].astype(np.float64), q)
        assert len(d) == len(h) == len(keep)
        assert d.tobytes() == rd[keep].tobytes() and h.tobytes() == rh[keep].tobytes()
        assert np.array_equal(h["score"].view(np.uint32), d["score"].view(np.uint32))


@pytest.mark.gpu
def test_gpu_nms_capacity_falls_back_to_the_raw_list():
    """More than ACF_HIP_NMS_CAP raw detections in a frame: get_detections reports E_CAPACITY for it, the raw list stays
    available (acf::HipDetector suppresses it on the host: tests/test_host_cpp.py)."""
    import torch
    from acf_amd.detector import HipDetector, HipError
    H, W = 240, 320
    model = synth.make_model(seed=3, name="TINY", nTrees=8, cascThr=-1e6)  # every window is a detection
    fr = torch.from_numpy(synth.make_frame(70, H, W, "luv")[None]).cuda()
    det = HipDetector(model, H, W, 3, max_batch=1, max_hits=1 << 16)
    det.run(fr)
    raw = det.detections(0)[0]
    assert len(raw) > 2048
    det.set_nms(capi.make_nms(type="maxg", overlap=0.65, ovrDnm="min", prune=True, maxCount=10))
    det.run(fr)
    with pytest.raises(HipError) as e:
        det.detections(0)
    assert e.value.code == capi.E_CAPACITY
    assert det.raw_detections(0).tobytes() == raw.tobytes()
