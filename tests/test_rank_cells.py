"""Threshold-rank cells (acf_amd/csrc/host_plan.h): the cascade's 16-bit pyramid.

The cascade only evaluates `chns[cid] < thrs[node]` (acfDetect1.cpp:102-104,157-166).  With t_0 < ... < t_{m-1} the distinct
thresholds of a channel, rank(v) = #{t_j <= v} and index(t_k) = k + 1 satisfy  v < t_k  <=>  rank(v) < index(t_k)  for every
float v — so a pyramid of ranks gives the cascade the float pyramid's decisions.  CPU tests: the host table builder and its
bucket function against numpy's searchsorted and against the float compare itself, on the values where it could go wrong
(the thresholds, their float neighbours, zeros of both signs, negatives, huge values).  GPU tests: the device's rank cells
against ranks of the oracle's float pyramid, and detections with rank cells on == off == oracle."""
import ctypes as C

import numpy as np
import pytest

from acf_amd import capi, synth


def host_ranks(model, nChns, chn, v):
    lib = capi.load()
    params, keep = capi.make_params(model)
    v = np.ascontiguousarray(v, np.float32)
    cells = np.zeros(len(v), np.uint16)
    idx = np.zeros(len(model["thrs"].ravel()), np.uint32)
    info = (C.c_int32 * 4)()
    rc = lib.acf_hip_rank_cells_host(C.byref(params), nChns, chn, capi.fptr(v), len(v), cells.ctypes.data_as(C.POINTER(C.c_uint16)),
                                     idx.ctypes.data_as(C.POINTER(C.c_uint32)), info)
    return rc, cells, idx, list(info)


def channel_thresholds(model, nChns, chn):
    mH, mW = model["modelDsPad_h"] // model["shrink"], model["modelDsPad_w"] // model["shrink"]
    fids = model["fids"][:, :3].ravel()
    thrs = model["thrs"][:, :3].ravel()
    return np.unique(thrs[fids // (mH * mW) == chn].astype(np.float32))


def probes(t):
    """values where an inexact rank would show: every threshold, its two float neighbours, and the usual suspects"""
    t = t.astype(np.float32)
    v = [t, np.nextafter(t, np.float32(-np.inf)), np.nextafter(t, np.float32(np.inf)),
         np.asarray([0.0, -0.0, 1e-45, -1e-45, -1.0, -3e38, 3e38, 1.0, 0.5, 1e-20, 65504.0], np.float32),
         (synth.uniform(5, 4000, 1) * 1.2 - 0.1).astype(np.float32)]
    return np.concatenate(v)


@pytest.mark.parametrize("name,kw,nChns", [("FACE80", {}, 10), ("INRIA", {}, 10), ("TINY", dict(nTrees=96), 10), ("FACE64", {}, 7)])
def test_host_rank_tables_are_exact(name, kw, nChns):
    model = synth.make_model(seed=1, name=name, **kw)
    for chn in range(nChns):
        t = channel_thresholds(model, nChns, chn)
        v = probes(t) if len(t) else probes(np.asarray([0.5], np.float32))
        rc, cells, idx, info = host_ranks(model, nChns, chn, v)
        assert rc == 0 and info[0] == 1 and info[3] == len(t), (chn, info)
        assert np.array_equal(cells, np.searchsorted(t, v, side="right").astype(np.uint16)), chn
    # every node: the rank compare IS the float compare, for every probe value
    mH, mW = model["modelDsPad_h"] // model["shrink"], model["modelDsPad_w"] // model["shrink"]
    fids = model["fids"]
    thrs = model["thrs"]
    for chn in range(nChns):
        t = channel_thresholds(model, nChns, chn)
        if not len(t):
            continue
        v = probes(t)
        _, cells, idx, _ = host_ranks(model, nChns, chn, v)
        idx = idx.reshape(model["thrs"].shape)
        sel = (fids[:, :3] // (mH * mW)) == chn
        th, ix = thrs[:, :3][sel], idx[:, :3][sel]
        assert (ix >= 1).all() and (ix <= len(t)).all()
        assert np.array_equal(v[:, None] < th[None, :], cells[:, None].astype(np.uint32) < ix[None, :])
    assert (idx.reshape(model["thrs"].shape)[:, 3:] == 0).all()  # leaves carry no index


def test_host_rank_tables_dense_and_degenerate_models():
    model = synth.make_model(seed=2, name="TINY", nTrees=64)
    nChns = 10
    mH, mW = model["modelDsPad_h"] // model["shrink"], model["modelDsPad_w"] // model["shrink"]
    thrs = model["thrs"].copy()
    fids = model["fids"].copy()
    # all nodes on channel 0; thresholds = 192 CONSECUTIVE floats: a bucket of 8 float patterns would hold more than RANK_WINDOW = 7
    fids[:, :3] = fids[:, :3] % (mH * mW)
    base = np.float32(0.25)
    seq = [base]
    for _ in range(thrs[:, :3].size - 1):
        seq.append(np.nextafter(seq[-1], np.float32(1)))
    thrs[:, :3] = np.asarray(seq, np.float32).reshape(-1, 3)
    m2 = dict(model, thrs=thrs.reshape(model["thrs"].shape), fids=fids.reshape(model["fids"].shape))
    v = probes(np.asarray(seq, np.float32))
    rc, cells, idx, info = host_ranks(m2, nChns, 0, v)
    assert rc == 0 and info[1] <= 2, info  # buckets of at most four float patterns (<= 7 thresholds each): still exact
    assert np.array_equal(cells, np.searchsorted(np.asarray(seq, np.float32), v, side="right").astype(np.uint16))
    # the other channels have no thresholds at all: every cell ranks 0
    rc, cells, _, info = host_ranks(m2, nChns, 3, v)
    assert rc == 0 and info[3] == 0 and not cells.any()
    # thresholds spread over so many octaves at that density that no table of RANK_MAX_BUCKETS fits: reported, not mis-ranked
    wide = np.concatenate([np.float32(2.0) ** -k * np.asarray(seq[:9], np.float32) for k in range(20)])  # 9 consecutive floats per octave
    wide = np.concatenate([wide, np.repeat(wide[-1:], thrs[:, :3].size - len(wide))])
    thrs[:, :3] = wide.reshape(-1, 3)
    m3 = dict(m2, thrs=thrs.reshape(model["thrs"].shape))
    rc, _, _, info = host_ranks(m3, nChns, 0, v)
    assert rc == capi.E_UNSUPPORTED and info[0] == 0
    # duplicates and zeros of both signs
    thrs[:, :3] = 0.5
    thrs[0, 0], thrs[0, 1], thrs[1, 0] = 0.0, -0.0, 0.75
    m4 = dict(m2, thrs=thrs.reshape(model["thrs"].shape))
    t = np.asarray([0.0, 0.5, 0.75], np.float32)
    v = probes(t)
    rc, cells, idx, info = host_ranks(m4, nChns, 0, v)
    assert rc == 0 and info[3] == 3
    assert np.array_equal(cells, np.searchsorted(t, v, side="right").astype(np.uint16))
    idx = idx.reshape(model["thrs"].shape)
    assert np.array_equal(v[:, None] < thrs[:2, :3].ravel()[None, :], cells[:, None].astype(np.uint32) < idx[:2, :3].ravel()[None, :])
    # a negative threshold: negative cells could no longer rank 0 -> the model keeps the float cascade
    thrs[0, 2] = -0.25
    rc, _, _, info = host_ranks(dict(m2, thrs=thrs.reshape(model["thrs"].shape)), nChns, 0, v)
    assert rc == capi.E_UNSUPPORTED and info[0] == 0


# ------------------------------------------------------------------ GPU

def _ranks_of(model, nChns, pyr_level):
    out = np.zeros(pyr_level.shape, np.uint16)
    for z in range(nChns):
        t = channel_thresholds(model, nChns, z)
        out[z] = np.searchsorted(t, pyr_level[z].ravel(), side="right").reshape(pyr_level[z].shape)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["luv", "rgb_pad", "gray"])
def test_gpu_rank_pyramid_and_detections(oracle, cfg):
    """Rank cells on the device == ranks of the oracle's float pyramid (every level, every cell); detections with rank cells
    == detections with float cells == oracle's, bit for bit."""
    import torch
    from acf_amd.detector import HipDetector
    if cfg == "luv":
        H, W, kind, d_in, kw = 200, 264, "luv", 3, dict(name="TINY", nTrees=256, cascThr=-1.0)
    elif cfg == "rgb_pad":
        H, W, kind, d_in, kw = 264, 200, "rgb", 3, dict(name="INRIA", nTrees=64, cascThr=-1.5)
    else:
        H, W, kind, d_in, kw = 144, 192, "gray", 1, dict(name="FACE64", nTrees=160, modelDs_h=32, modelDs_w=32, modelDsPad_h=32, modelDsPad_w=32,
                                                         minDs_h=32, minDs_w=32, cascThr=-2.0)
    model = synth.make_model(seed=3, **kw)
    frames = np.stack([synth.make_frame(11 + i, H, W, kind) for i in range(3)])
    fr = torch.from_numpy(frames).cuda()
    det = HipDetector(model, H, W, d_in, max_batch=3, max_hits=1 << 15)
    plan = oracle.Plan(model, H, W, d_in)
    det.run(fr)
    got = [det.detections(f) for f in range(3)]
    for f in range(3):
        pyr, _, _ = oracle.chns_pyramid(plan, frames[f])
        want, whits = oracle.detect(plan, pyr)
        assert len(want) > 0
        assert got[f][0].tobytes() == want.tobytes() and got[f][1].tobytes() == whits.tobytes(), (cfg, f)
        for i, l in enumerate(det.levels):
            lvl = pyr[l.offset:l.offset + det.nChns * l.hP * l.wP].reshape(det.nChns, l.wP, l.hP)
            assert np.array_equal(det.read_rank_level(f, i), _ranks_of(model, det.nChns, lvl)), (cfg, f, i)
    det.set_option("rank_cells", 0)
    det.run(fr)
    for f in range(3):
        d, h = det.detections(f)
        assert d.tobytes() == got[f][0].tobytes() and h.tobytes() == got[f][1].tobytes()
    from acf_amd.detector import HipError
    with pytest.raises(HipError):
        det.read_rank_level(0, 0)
    det.close()


@pytest.mark.gpu
def test_gpu_rank_cells_with_sub_batch_streams_and_nms(oracle):
    import torch
    from acf_amd.detector import HipDetector
    H, W = 120, 160
    model = synth.make_model(seed=3, name="TINY", nTrees=96, cascThr=-2.0)
    frames = np.stack([synth.make_frame(90 + i, H, W, "luv") for i in range(4)])
    fr = torch.from_numpy(frames).cuda()
    a = HipDetector(model, H, W, 3, max_batch=4, max_hits=1 << 14)
    a.set_option("rank_cells", 0)
    a.run(fr)
    b = HipDetector(streams=2)
    b.set_model(model)
    b.plan(H, W, 3, max_batch=4, max_hits=1 << 14)
    b.run(fr)
    for f in range(4):
        assert a.detections(f)[0].tobytes() == b.detections(f)[0].tobytes()
        assert np.array_equal(b.read_rank_level(f, 0).shape, (b.nChns, b.levels[0].wP, b.levels[0].hP))


@pytest.mark.gpu
def test_gpu_detections_without_the_float_pyramid(oracle):
    """Option keep_pyramid = 0: the level kernels write rank cells only; detections stay the oracle's, the float level is
    reported as absent (not silently stale)."""
    import torch
    from acf_amd.detector import HipDetector, HipError
    H, W = 200, 264
    model = synth.make_model(seed=3, name="TINY", nTrees=256, cascThr=-1.0)
    frames = np.stack([synth.make_frame(11 + i, H, W, "luv") for i in range(5)])   # 5: one full group of 4 frames + 1
    fr = torch.from_numpy(frames).cuda()
    det = HipDetector(model, H, W, 3, max_batch=5, max_hits=1 << 15)
    det.set_option("keep_pyramid", 0)
    det.run(fr)
    plan = oracle.Plan(model, H, W, 3)
    for f in range(5):
        pyr, _, _ = oracle.chns_pyramid(plan, frames[f])
        want, whits = oracle.detect(plan, pyr)
        d, h = det.detections(f)
        assert len(want) > 0 and d.tobytes() == want.tobytes() and h.tobytes() == whits.tobytes(), f
        l = det.levels[3]
        lvl = pyr[l.offset:l.offset + det.nChns * l.hP * l.wP].reshape(det.nChns, l.wP, l.hP)
        assert np.array_equal(det.read_rank_level(f, 3), _ranks_of(model, det.nChns, lvl))
    with pytest.raises(HipError):
        det.read_level(0, 0)
    det.set_option("keep_pyramid", 1)
    det.run(fr)
    pyr, _, _ = oracle.chns_pyramid(plan, frames[4])
    assert np.array_equal(det.read_pyramid(4).view(np.uint32), pyr.view(np.uint32))
    det.close()


@pytest.mark.gpu
def test_gpu_tail_queue_overflow_on_rank_cells(oracle):
    """More windows reach the tail than stage E has code rows for (cascThr far below every score): the overflow handler of
    the rank form (k_cascade_tail_rank) finishes them from the rank pyramid; with and without the float pyramid, == oracle."""
    import torch
    from acf_amd.detector import HipDetector
    H, W = 160, 200
    model = synth.make_model(seed=5, name="TINY", nTrees=200, cascThr=-1e6)
    frames = np.stack([synth.make_frame(31 + i, H, W, "luv") for i in range(2)])
    fr = torch.from_numpy(frames).cuda()
    plan = oracle.Plan(model, H, W, 3)
    want = []
    for f in range(2):
        pyr, _, _ = oracle.chns_pyramid(plan, frames[f])
        want.append(oracle.detect(plan, pyr, cap=1 << 16) if False else oracle.detect(plan, pyr))
    assert all(len(w[0]) > 1500 for w in want)
    for keep in (1, 0):
        det = HipDetector(model, H, W, 3, max_batch=2, max_hits=1 << 16)
        det.set_option("keep_pyramid", keep)
        det.run(fr)
        for f in range(2):
            d, h = det.detections(f)
            assert d.tobytes() == want[f][0].tobytes() and h.tobytes() == want[f][1].tobytes(), (keep, f)
        det.close()
