"""GPU parity of the whole hot path through the C ABI: chnsPyramid stage taps,
every level of the fused pyramid, cascade hits and mapped boxes — all compared
bit-for-bit with the oracle on the same seeded frames and models."""
import numpy as np
import pytest

from acf_amd import capi, synth

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


CONFIGS = {
    # name: (H, W, frame kind, d_in, model kwargs)
    "tiny_luv": (96, 128, "luv", 3, dict(name="TINY", nTrees=256)),
    "odd_luv": (130, 175, "luv", 3, dict(name="TINY", nTrees=128, cascThr=-2.5)),          # sz != sz1 at scale 1: image is resampled first
    "rgb_inria_pad": (240, 320, "rgb", 3, dict(name="INRIA", nTrees=256)),  # RGB->LUV, pad [16 12], nOctUp 1 (up-sampled real scale)
    "gray_face64": (240, 320, "gray", 1, dict(name="FACE64", nTrees=256)),  # 7 channels, colour disabled, 1-plane input
    "luv_nperoct4": (200, 260, "luv", 3, dict(name="TINY", nTrees=128, nPerOct=4, nApprox=3, minDs_h=24, minDs_w=24, cascThr=-2.5)),
    "all_real": (120, 160, "luv", 3, dict(name="TINY", nTrees=64, nApprox=0, minDs_h=32, minDs_w=32)),
    "shrink2": (96, 128, "luv", 3, dict(name="TINY", nTrees=64, shrink=2, modelDsPad_h=16, modelDsPad_w=16, minDs_h=32, minDs_w=32)),
    "face80_vga": (480, 640, "luv", 3, dict(name="FACE80", nTrees=512)),
    # gradHist's other even-softBin branch (nearest orientation bin, gradientMex.cpp:391-450) and a positive even value (same branch as 0)
    "hardbin_vga": (480, 640, "luv", 3, dict(name="FACE80", nTrees=256, softBin=-2, cascThr=-2.0)),
    "softbin2": (200, 260, "luv", 3, dict(name="TINY", nTrees=128, softBin=2, cascThr=-2.5)),
    # rgbConvert to HSV (rgbConvertMex.cpp:194-238)
    "rgb_hsv": (240, 320, "rgb", 3, dict(name="INRIA", nTrees=128, colorSpace=capi.CS_HSV, cascThr=-3.0)),
}


def build(cfg, batch=1, taps=True):
    from acf_amd.detector import HipDetector
    H, W, kind, d_in, kw = CONFIGS[cfg]
    model = synth.make_model(seed=3, **kw)
    det = HipDetector(model, H, W, d_in, max_batch=batch, max_hits=1 << 16, taps=taps)
    return det, model, (H, W, kind, d_in)


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("cfg", list(CONFIGS))
def test_pipeline_bit_exact(oracle, cfg, fused):
    """fused=1: approximated-scale resample + final smoothing in one wave-per-plane kernel (k_level_fused),
    tiled cascade; fused=0: the separate resample / smoothing launches and the global-memory staged cascade."""
    import torch
    det, model, (H, W, kind, d_in) = build(cfg)
    det.set_option("fused_levels", fused)
    det.set_option("cascade_tiles", fused)
    frame = synth.make_frame(17, H, W, kind)
    plan = oracle.Plan(model, H, W, d_in)
    # plan geometry agrees (host_plan.cpp vs the oracle's independent restatement)
    assert len(det.levels) == plan.nScales and det.nChns == plan.nChns
    for a, b in zip(det.levels, list(plan.levels)[:plan.nScales]):
        for f, _ in capi.Level._fields_:
            assert getattr(a, f) == getattr(b, f), f
    pyr, taps, chns = oracle.chns_pyramid(plan, frame, want_taps=True, want_chns=True)
    det.run(torch.from_numpy(frame[None]).cuda())
    d = 1 if model["colorSpace"] == capi.CS_GRAY else 3
    for k, lvl in enumerate(plan.real):
        t = taps[k]
        w1, h1 = t["M"].shape
        for name, tap, shape in (("image", capi.TAP_IMAGE, (d, w1, h1)), ("smoothed", capi.TAP_SMOOTHED, (d, w1, h1)),
                                 ("M", capi.TAP_M, (w1, h1)), ("O", capi.TAP_O, (w1, h1)), ("S", capi.TAP_S, (w1, h1)),
                                 ("Mnorm", capi.TAP_MNORM, (w1, h1))):
            got = det.read_tap(0, tap, k, shape)
            assert np.array_equal(bits(got), bits(t[name])), (cfg, "real scale", k, name, float(np.abs(got - t[name]).max()))
    for i in range(plan.nScales):
        l = plan.levels[i]
        got = det.read_tap(0, capi.TAP_CHNS, i, (plan.nChns, l.wC, l.hC))
        assert np.array_equal(bits(got), bits(chns[i])), (cfg, "raw channels level", i, float(np.abs(got - chns[i]).max()))
        gl = det.read_level(0, i)
        assert np.array_equal(bits(gl), bits(plan.level_view(pyr, i))), (cfg, "pyramid level", i)
    want, want_hits = oracle.detect(plan, pyr)
    got, got_hits = det.detections(0)
    assert len(got) == len(want), (len(got), len(want))
    assert len(want) > 0, "config produces no detections: the cascade comparison would be vacuous"
    for k in ("scale", "c", "r"):
        assert np.array_equal(got_hits[k], want_hits[k]), k
    for k in ("x", "y", "w", "h", "scale"):
        assert np.array_equal(got[k], want[k]), k
    # north_star tolerance is 1e-4 on scores; we require bit equality
    assert np.array_equal(bits(got["score"]), bits(want["score"]))
    det.close()


def test_batch_frames_independent(oracle):
    """A batch is processed frame by frame identically: frame i of a batch of 5 == the oracle on frame i."""
    import torch
    det, model, (H, W, kind, d_in) = build("tiny_luv", batch=5, taps=False)
    frames = np.stack([synth.make_frame(100 + i, H, W, kind) for i in range(5)])
    det.run(torch.from_numpy(frames).cuda())
    plan = oracle.Plan(model, H, W, d_in)
    for i in range(5):
        pyr, _, _ = oracle.chns_pyramid(plan, frames[i])
        assert np.array_equal(bits(det.read_pyramid(i)), bits(pyr))
        want, _ = oracle.detect(plan, pyr)
        got, _ = det.detections(i)
        assert len(got) == len(want)
        assert np.array_equal(bits(got["score"]), bits(want["score"]))
        assert np.array_equal(got["x"], want["x"]) and np.array_equal(got["y"], want["y"])
    # host-pointer entry gives the same result as the device-pointer entry
    det.run_host(frames[:3])
    for i in range(3):
        pyr, _, _ = oracle.chns_pyramid(plan, frames[i])
        assert np.array_equal(bits(det.read_pyramid(i)), bits(pyr))
    det.close()


def test_input_frame_not_modified():
    """The reference smooths the caller's planes in place when isLuv (shallow copies,
    chnsPyramid.cpp:247,305); the HIP path is out-of-place and must leave the input intact."""
    import torch
    det, model, (H, W, kind, d_in) = build("tiny_luv", taps=False)
    frame = synth.make_frame(4, H, W, kind)
    t = torch.from_numpy(frame[None]).cuda()
    det.run(t)
    det.synchronize()
    assert np.array_equal(t.cpu().numpy()[0], frame)
    det.close()


def test_capacity_error_is_reported():
    import torch
    from acf_amd.detector import HipDetector, HipError
    H, W, kind, d_in, kw = CONFIGS["tiny_luv"]
    model = synth.make_model(seed=3, **dict(kw, cascThr=-1e9))  # nothing is ever rejected
    det = HipDetector(model, H, W, d_in, max_batch=1, max_hits=16)
    det.run(torch.from_numpy(synth.make_frame(1, H, W, kind)[None]).cuda())
    with pytest.raises(HipError) as e:
        det.detections(0)
    assert e.value.code == 7
    det.close()


@pytest.mark.parametrize("max_hits", [128, 1 << 16])
def test_windows_sharing_an_offset(oracle, max_hits):
    """stride < shrink: the cascade runs once per distinct cell offset and k_expand_hits writes every window of a surviving
    offset (T/acfDetect1.cpp:88-96: r * stride / shrink).  With room for them the detections equal the oracle's, one hit per
    WINDOW; with max_hits between the distinct survivors and the windows they stand for the call reports the overflow."""
    import torch
    from acf_amd.detector import HipDetector, HipError
    H, W, kind, d_in, kw = CONFIGS["tiny_luv"]
    model = synth.make_model(seed=3, **dict(kw, stride=2, cascThr=-1.2))
    frame = synth.make_frame(1, H, W, kind)
    plan = oracle.Plan(model, H, W, d_in)
    pyr, _, _ = oracle.chns_pyramid(plan, frame)
    want, want_hits = oracle.detect(plan, pyr)
    distinct = len({(int(h["scale"]), int(h["r"]) // 2, int(h["c"]) // 2) for h in want_hits})
    assert distinct <= 128 < len(want), (distinct, len(want))  # (the small capacity holds the offsets, not the windows)
    det = HipDetector(model, H, W, d_in, max_batch=1, max_hits=max_hits)
    det.run(torch.from_numpy(frame[None]).cuda())
    if max_hits < len(want):
        with pytest.raises(HipError) as e:
            det.detections(0)
        assert e.value.code == 7
    else:
        got, got_hits = det.detections(0)
        assert got.tobytes() == want.tobytes() and got_hits.tobytes() == want_hits.tobytes()
    det.close()


def test_export_detections_layout(oracle):
    import torch
    det, model, (H, W, kind, d_in) = build("tiny_luv", batch=2, taps=False)
    frames = np.stack([synth.make_frame(50 + i, H, W, kind) for i in range(2)])
    det.run(torch.from_numpy(frames).cuda())
    cap = 64
    dst = torch.full((2, 1 + 6 * cap), -7, dtype=torch.int32, device="cuda")
    det.export_detections(dst, cap)
    det.synchronize()
    rec = dst.cpu().numpy()
    for i in range(2):
        got, _ = det.detections(i)
        n = min(len(got), cap)
        assert rec[i, 0] == len(got)
        body = rec[i, 1:].reshape(cap, 6)
        assert np.array_equal(body[:n, 0], got["x"][:n]) and np.array_equal(body[:n, 3], got["h"][:n])
        assert np.array_equal(body[:n, 4].view(np.float32), got["score"][:n])
        assert np.all(body[n:] == 0)
    det.close()


@pytest.mark.parametrize("streams", [2, 3])
def test_sub_batch_streams(oracle, streams):
    """Option "streams": the batch is cut into chunks run by child contexts on their own streams.  Every frame's
    pyramid, hits, boxes and export record must equal the single-stream (oracle) result, including an uneven split."""
    import torch
    from acf_amd.detector import HipDetector
    from acf_amd.dist import records_to_detections
    H, W = 96, 128
    model = synth.make_model(seed=3, name="TINY", nTrees=160)
    n = 7
    det = HipDetector(model, H, W, 3, max_batch=n, max_hits=1 << 14, streams=streams)
    frames = np.stack([synth.make_frame(60 + i, H, W, "luv") for i in range(n)])
    fr = torch.from_numpy(frames).cuda()
    plan = oracle.Plan(model, H, W, 3)
    cap = 256
    rec = torch.zeros((n, 1 + 6 * cap), dtype=torch.int32, device="cuda")
    for nrun in (n, n - 2):  # a full batch, then a shorter one that leaves the last child with fewer frames
        det.run(fr, nrun)
        det.export_detections(rec, cap)
        torch.cuda.synchronize()
        r = rec.cpu().numpy()
        for f in range(nrun):
            pyr, _, _ = oracle.chns_pyramid(plan, frames[f])
            want, wh = oracle.detect(plan, pyr)
            assert np.array_equal(bits(det.read_pyramid(f)), bits(pyr)), f
            got, gh = det.detections(f)
            assert got.tobytes() == want.tobytes() and gh.tobytes() == wh.tobytes(), f
            dec = records_to_detections(r[f], cap)
            assert len(dec) == min(len(want), cap) and r[f, 0] == len(want)
            for a, b in zip(dec, want):
                assert a[:4] == (int(b["x"]), int(b["y"]), int(b["w"]), int(b["h"])) and np.float32(a[4]) == b["score"]
    det.close()


@pytest.mark.parametrize("fused_smooth", [2, 1, 0])   # 2: with convTri's x pass on the gradient plane's chain as well (k_smooth_grad_tri)
@pytest.mark.parametrize("H,W,kw", [
    (256, 384, dict(name="TINY", nTrees=96)),                       # fused: exact-half next scale, colour channels from registers
    (256, 384, dict(name="TINY", nTrees=96, full=1, colorChn=1)),   # orientation over 2 pi, gradient plane 1 (k_smooth_grad takes that plane)
    (480, 640, dict(name="FACE80", nTrees=256)),
    (200, 264, dict(name="TINY", nTrees=96)),                       # sz != sz1 at scale 1 (image resampled first), generic next-scale resample
    (96, 132, dict(name="TINY", nTrees=96)),                        # w % 8 != 0 at some scale: falls back per scale
    (240, 320, dict(name="TINY", nTrees=96, nApprox=0, minDs_h=32, minDs_w=32)),  # every scale real
    # k_smooth_vec's waves own 60 row quads each and shadow two of each neighbour: 60 quads = one wave exactly, 61 = a second
    # wave with a single owned quad, 122 = three waves, 480 = eight (the most), 482 = the separate-kernel path for that scale;
    # the tall planes also take k_level's R = 5..8 specialisations (their own launches, ring of 5 slots)
    (240, 128, dict(name="TINY", nTrees=96)),
    (244, 128, dict(name="TINY", nTrees=96)),
    (488, 136, dict(name="TINY", nTrees=96)),
    (1920, 128, dict(name="TINY", nTrees=96)),
    (1928, 136, dict(name="TINY", nTrees=96)),
])
def test_fused_smoothing_paths(oracle, H, W, kw, fused_smooth):
    """k_smooth_vec (smoothing + colour channels + exact-half next image, taps off) against the oracle, and the
    separate-kernel path on the same inputs."""
    import torch
    from acf_amd.detector import HipDetector
    model = synth.make_model(seed=3, **kw)
    frame = synth.make_frame(23, H, W, "luv")
    det = HipDetector(model, H, W, 3, max_batch=2, max_hits=1 << 15)
    det.set_option("fused_smooth", min(fused_smooth, 1))
    det.set_option("fused_tri", 2 if fused_smooth == 2 else 0)
    det.set_option("fused_grad", 2 if fused_smooth else 0)  # gradMag inside the gradient plane's smoothing chain at every scale it applies to / its own kernel
    det.set_option("scale_streams", fused_smooth)  # real scales on their own streams / all on the context's stream
    det.run(torch.from_numpy(np.stack([frame, frame])).cuda())
    plan = oracle.Plan(model, H, W, 3)
    pyr, _, _ = oracle.chns_pyramid(plan, frame)
    want, wh = oracle.detect(plan, pyr)
    for f in (0, 1):
        assert np.array_equal(bits(det.read_pyramid(f)), bits(pyr))
        got, gh = det.detections(f)
        assert got.tobytes() == want.tobytes() and gh.tobytes() == wh.tobytes()
    det.close()


def test_detector_pool_concurrent_contexts_match_single(oracle):
    """bench.py's deployment: several contexts on their own streams working on different batches at the same time.
    Every context must return exactly what a lone context returns for the same frames (no shared mutable state)."""
    import torch
    from acf_amd.detector import DetectorPool, HipDetector
    H, W, kind, d_in, kw = CONFIGS["face80_vga"]
    model = synth.make_model(seed=3, **kw)
    nF, C = 4, 3
    batches = [np.stack([synth.make_frame(200 + 10 * c + i, H, W, kind) for i in range(nF)]) for c in range(C)]
    ref = HipDetector(model, H, W, d_in, max_batch=nF, max_hits=1 << 14)
    want = []
    for b in batches:
        ref.run(torch.from_numpy(b).cuda())
        want.append([ref.detections(f)[0] for f in range(nF)])
    ref.close()
    pool = DetectorPool(C, model, H, W, d_in, max_batch=nF, max_hits=1 << 14)
    dev_batches = [torch.from_numpy(b).cuda() for b in batches]
    torch.cuda.synchronize()
    for rep in range(3):  # several rounds in flight before anything is read back
        for (det, stream), x in zip(pool, dev_batches):
            with torch.cuda.stream(stream):
                det.run(x)
    pool.synchronize()
    total = 0
    for c, (det, _) in enumerate(pool):
        for f in range(nF):
            got = det.detections(f)[0]
            assert len(got) == len(want[c][f])
            for k in ("x", "y", "w", "h", "scale"):
                assert np.array_equal(got[k], want[c][f][k]), (c, f, k)
            assert np.array_equal(bits(got["score"]), bits(want[c][f]["score"]))
            total += len(got)
    assert total > 0
    # and the oracle agrees with the lone context on one of them
    plan = oracle.Plan(model, H, W, d_in)
    pyr, _, _ = oracle.chns_pyramid(plan, batches[1][2])
    odet, _ = oracle.detect(plan, pyr)
    assert np.array_equal(bits(odet["score"]), bits(want[1][2]["score"]))
    pool.close()
    # the pool's contexts take turns with their level and tile kernels (option cascade_turns: events shared through the library);
    # a second pool after the first one is gone must not wait on a destroyed context's event
    pool = DetectorPool(2, model, H, W, d_in, max_batch=nF, max_hits=1 << 14)
    for rep in range(2):
        for (det, stream), x in zip(pool, dev_batches):
            with torch.cuda.stream(stream):
                det.run(x)
    pool.synchronize()
    for c, (det, _) in enumerate(pool):
        for f in range(nF):
            assert det.detections(f)[0].tobytes() == want[c][f].tobytes()
    pool.close()


LDCF_CASES = {
    # name: (H, W, model kwargs) — BASELINE cfg 5's post-stage on small frames: k 5x5 filters per channel, halved levels, cascade at shrink*2
    "tiny32_k3": (120, 160, dict(name="TINY", nTrees=64, ldcfK=3, modelDs_h=32, modelDs_w=32, modelDsPad_h=32, modelDsPad_w=32,
                                 minDs_h=32, minDs_w=32, cascThr=-2.0)),
    "tiny32_k2_stride8_odd": (131, 175, dict(name="TINY", nTrees=64, ldcfK=2, stride=8, modelDs_h=32, modelDs_w=32, modelDsPad_h=32,
                                             modelDsPad_w=32, minDs_h=32, minDs_w=32, cascThr=-2.0)),
    "face80_k4_vga": (480, 640, dict(name="FACE80", nTrees=256, ldcfK=4, cascThr=-2.0)),
}


@pytest.mark.parametrize("case", list(LDCF_CASES))
def test_ldcf_post_stage_bit_exact(oracle, case):
    """LDCF: no reference counterpart, so the oracle's restatement of the toolbox definition
    (oracle/acf_oracle.c acfo_ldcf_*) is the only yardstick: filtered + halved levels and the detections on them."""
    import torch
    from acf_amd.detector import HipDetector
    H, W, kw = LDCF_CASES[case]
    model = synth.make_model(seed=3, **kw)
    nF = 2
    det = HipDetector(model, H, W, 3, max_batch=nF, max_hits=1 << 16)
    frames = np.stack([synth.make_frame(70 + i, H, W, "luv") for i in range(nF)])
    det.run(torch.from_numpy(frames).cuda())
    plan = oracle.Plan(model, H, W, 3)
    total = 0
    for f in range(nF):
        pyr, _, _ = oracle.chns_pyramid(plan, frames[f])
        lvL, pyrL, k = oracle.ldcf(plan, pyr)
        assert len(det.ldcf_levels) == plan.nScales
        for i in range(plan.nScales):
            l = lvL[i]
            for fld in ("hP", "wP", "nWinR", "nWinC", "offset"):
                assert getattr(det.ldcf_levels[i], fld) == getattr(l, fld), (i, fld)
            n = plan.nChns * k * l.hP * l.wP
            want = pyrL[l.offset:l.offset + n].reshape(plan.nChns * k, l.wP, l.hP)
            got = det.read_tap(f, capi.TAP_LDCF, i, (plan.nChns * k, l.wP, l.hP))
            assert np.array_equal(bits(got), bits(want)), (case, f, "LDCF level", i, float(np.abs(got - want).max()))
        want, want_hits = oracle.detect_ldcf(plan, lvL, pyrL)
        got, got_hits = det.detections(f)
        assert len(got) == len(want), (len(got), len(want))
        for key in ("scale", "c", "r"):
            assert np.array_equal(got_hits[key], want_hits[key]), key
        for key in ("x", "y", "w", "h", "scale"):
            assert np.array_equal(got[key], want[key]), key
        assert np.array_equal(bits(got["score"]), bits(want["score"]))
        total += len(want)
    assert total > 0
    det.close()


@pytest.mark.parametrize("scale_streams", [1, 0])
def test_graph_replay_matches_plain_launches(oracle, scale_streams):
    """Option graph: acf_hip_run captures its own launches (second call with the same input pointer and batch size) and replays the
    HIP graph afterwards.  Every call — plain, capturing, replaying, re-capturing for another input — returns the oracle's result."""
    import torch
    from acf_amd import capi
    from acf_amd.detector import HipDetector
    H, W = 480, 640
    model = synth.make_model(seed=3, name="FACE80", nTrees=256)
    frames = np.stack([synth.make_frame(301 + i, H, W, "luv") for i in range(2)])
    dev = torch.from_numpy(frames).cuda()
    det = HipDetector(model, H, W, 3, max_batch=1, max_hits=1 << 14)
    det.set_option("scale_streams", scale_streams)
    det.set_option("graph", 1)
    plan = oracle.Plan(model, H, W, 3)
    want = []
    for f in range(2):
        pyr, _, _ = oracle.chns_pyramid(plan, frames[f])
        want.append(oracle.detect(plan, pyr))
    for k, f in enumerate([0, 0, 0, 0, 1, 1, 0, 1]):
        det.run(dev[f:f + 1], 1)
        got, gh = det.detections(0)
        assert got.tobytes() == want[f][0].tobytes() and gh.tobytes() == want[f][1].tobytes(), (k, f)
        assert np.array_equal(bits(det.read_pyramid(0)), bits(oracle.chns_pyramid(plan, frames[f])[0])), (k, f)
    det.close()


def test_graph_replay_sees_new_content_in_the_same_buffer(oracle):
    """The documented use of option graph: new frame content copied into the SAME device buffer between replays.  Frames with
    different detection counts, detections() after every run (which caches the counts on the host): a replay must not hand out
    the previous run's counts.  Then two buffers alternating (double buffering): one cached graph per buffer, no re-capture."""
    import torch
    from acf_amd.detector import HipDetector
    H, W = 480, 640
    model = synth.make_model(seed=3, name="FACE80", nTrees=256)
    frames = np.stack([synth.make_frame(311 + i, H, W, "luv") for i in range(3)])
    src = torch.from_numpy(frames).cuda()
    plan = oracle.Plan(model, H, W, 3)
    want = []
    for f in range(3):
        pyr, _, _ = oracle.chns_pyramid(plan, frames[f])
        want.append(oracle.detect(plan, pyr))
    assert len({len(w[0]) for w in want}) > 1, "the frames must differ in their detection counts"
    for keep in (1, 0):
        det = HipDetector(model, H, W, 3, max_batch=1, max_hits=1 << 14)
        det.set_option("graph", 1)
        det.set_option("keep_pyramid", keep)
        buf = [torch.empty_like(src[0:1]), torch.empty_like(src[0:1])]
        for k, (b, f) in enumerate([(0, 0), (0, 0), (0, 1), (0, 2), (0, 0), (1, 1), (0, 2), (1, 0), (0, 1), (1, 2)]):
            buf[b].copy_(src[f:f + 1])
            det.run(buf[b], 1)
            got, gh = det.detections(0)
            assert got.tobytes() == want[f][0].tobytes() and gh.tobytes() == want[f][1].tobytes(), (keep, k, b, f)
        det.close()


def test_detection_only_call_with_the_staged_cascade(oracle):
    """keep_pyramid = 0 (the level kernel may leave rank cells only) together with cascade_tiles = 0 (the staged cascade reads
    floats): the levels must still leave as floats; switching the cascade between acf_hip_pyramid and acf_hip_detect to one
    whose cells were not written is an error, not garbage."""
    import torch
    from acf_amd.detector import HipDetector, HipError
    H, W = 480, 640
    model = synth.make_model(seed=3, name="FACE80", nTrees=256)
    frame = synth.make_frame(321, H, W, "luv")
    dev = torch.from_numpy(frame[None]).cuda()
    plan = oracle.Plan(model, H, W, 3)
    pyr, _, _ = oracle.chns_pyramid(plan, frame)
    want, wh = oracle.detect(plan, pyr)
    assert len(want) > 0
    det = HipDetector(model, H, W, 3, max_batch=1, max_hits=1 << 14)
    det.set_option("keep_pyramid", 0)
    det.set_option("cascade_tiles", 0)
    det.run(dev, 1)
    got, gh = det.detections(0)
    assert got.tobytes() == want.tobytes() and gh.tobytes() == wh.tobytes()
    det.set_option("cascade_tiles", 1)
    det.run(dev, 1)
    got, gh = det.detections(0)
    assert got.tobytes() == want.tobytes() and gh.tobytes() == wh.tobytes()
    # pyramid with rank cells only, then the float-reading cascade selected: refused
    det.pyramid(dev, 1)
    det.set_option("cascade_tiles", 0)
    with pytest.raises(HipError):
        det.detect()
    det.close()


@pytest.mark.parametrize("nTrees", [129, 144, 145, 193])
@pytest.mark.parametrize("depth", [1, 3, 4])
def test_fixed_depth_tail_codes_tree_counts(oracle, depth, nTrees):
    """Depths 1, 3, 4: trees [0, 32) on LDS tiles (k_cascade_tileD), [32, 128) on the staged queue, the rest as leaf codes + ordered
    scan (k_tail_codesD / k_tail_scanD).  Tree counts just past 128: one tail tree, exactly one / one more than one group of 16
    (the scan adds whole groups, padded with rows of -0.0f), one more than a 64-tree code batch."""
    import torch
    from acf_amd.detector import HipDetector
    H, W = 200, 264
    model = synth.make_model(seed=5 + depth, name="TINY", nTrees=nTrees, treeDepth=depth, cascThr=-2.0)
    frame = synth.make_frame(77, H, W, "luv")
    det = HipDetector(model, H, W, 3, max_batch=1, max_hits=1 << 15)
    det.run(torch.from_numpy(frame[None]).cuda())
    plan = oracle.Plan(model, H, W, 3)
    pyr, _, _ = oracle.chns_pyramid(plan, frame)
    want, wh = oracle.detect(plan, pyr)
    got, gh = det.detections(0)
    assert len(want) > 0
    assert got.tobytes() == want.tobytes() and gh.tobytes() == wh.tobytes()
    det.close()
