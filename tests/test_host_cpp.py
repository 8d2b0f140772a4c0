"""The C++ host side (acf_amd/host: acf::HipDetector + dlopen loader + CLI).

CPU tests: it builds with g++ alone, its NMS / prune host logic matches the
oracle's restatement of the reference's nmsMax (bbNms.cpp:111-192) and prune
(ObjectDetector.cpp:28-44), and it fails loudly when libacf_hip.so is missing.  GPU tests: the CLI's detections through every entry (image, batch,
host-pyramid round trip, NMS) equal the oracle's.
"""
import os
import subprocess

import numpy as np
import pytest

from acf_amd import capi, synth
from acf_amd.modelio import write_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "acf_amd", "host")
CLI = os.path.join(HOST, "acf_hip_detect")


@pytest.fixture(scope="module")
def cli():
    subprocess.check_call(["make", "-s", "-C", HOST])
    return CLI


def run(cli, args, env=None, ok=True):
    e = dict(os.environ)
    e["ACF_HIP_LIBRARY"] = capi.LIB_PATH
    if env:
        e.update(env)
    p = subprocess.run([cli] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    if ok:
        assert p.returncode == 0, p.stderr
    return p


def parse(out):
    frames = []
    cur = None
    for line in out.strip().splitlines():
        t = line.split()
        if t[0] == "frame":
            cur = []
            frames.append(cur)
        else:
            cur.append((int(t[0]), int(t[1]), int(t[2]), int(t[3]), int(t[5], 16)))
    return frames


@pytest.mark.parametrize("typ,ovr,overlap", [("maxg", "min", 0.65), ("max", "union", 0.5), ("maxg", "union", 0.3), ("none", "min", 0.5)])
def test_nms_and_prune_match_the_oracle(cli, oracle, tmp_path, typ, ovr, overlap):
    """Host bbNms + prune of acf::HipDetector (own formulation over the score ranking) against oracle/acf_oracle.c:acfo_nms,
    the restatement of bbNms.cpp:111-192,229-304 + ObjectDetector.cpp:28-44.  Scores repeat (quantised), so ties occur."""
    u = synth.uniform(5, 400 * 5, 3).reshape(400, 5)
    boxes = [(int(r[0] * 300), int(r[1] * 200), 20 + int(r[2] * 60), 20 + int(r[3] * 60)) for r in u]
    scores = [float(np.float32(np.floor(r[4] * 60) / 2)) for r in u]
    path = tmp_path / "boxes.txt"
    path.write_text("\n".join("%d %d %d %d %.9g" % (b + (s,)) for b, s in zip(boxes, scores)))
    p = run(cli, ["--nms-only", str(path), "--type", typ, "--ovrdnm", ovr, "--overlap", str(overlap), "--prune", "--max-count", "7", "--prune-ratio", "0.5"])
    got = parse(p.stdout)[0]
    keep = oracle.nms(boxes, scores, capi.make_nms(type=typ, overlap=overlap, ovrDnm=ovr, prune=True, maxCount=7, pruneRatio=0.5))
    assert len(set(scores)) < len(scores)
    assert [g[:4] for g in got] == [tuple(boxes[i]) for i in keep]
    assert [g[4] for g in got] == [int(np.float32(scores[i]).view(np.uint32)) for i in keep]


def test_missing_library_fails_loudly(cli, tmp_path):
    m = synth.make_model(seed=3, name="TINY", nTrees=8)
    write_model(str(tmp_path / "m.acfm"), m)
    (tmp_path / "f.raw").write_bytes(np.zeros((3, 80, 64), np.float32).tobytes())
    p = run(cli, ["--model", str(tmp_path / "m.acfm"), "--frames", str(tmp_path / "f.raw"), "--rows", "80", "--cols", "64", "--channels", "3", "--count", "1"],
            env={"ACF_HIP_LIBRARY": "/nonexistent/libacf_hip.so"}, ok=False)
    assert p.returncode != 0 and "dlopen" in p.stderr


# ------------------------------------------------------------------ GPU

def _oracle_dets(oracle, model, frames, H, W, d_in):
    plan = oracle.Plan(model, H, W, d_in)
    out = []
    for f in frames:
        pyr, _, _ = oracle.chns_pyramid(plan, f)
        det, _ = oracle.detect(plan, pyr)
        out.append([(int(d["x"]), int(d["y"]), int(d["w"]), int(d["h"]), int(np.float32(d["score"]).view(np.uint32))) for d in det])
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [[], ["--batch"], ["--via-pyramid"]])
@pytest.mark.parametrize("cfg", ["luv", "rgb_pad"])
def test_cli_matches_oracle(cli, oracle, tmp_path, mode, cfg):
    if cfg == "luv":
        H, W, kind, kw, extra = 96, 128, "luv", dict(name="TINY", nTrees=96), ["--luv"]
    else:
        H, W, kind, kw, extra = 112, 96, "rgb", dict(name="INRIA", nTrees=64, cascThr=-1.5), []
    model = synth.make_model(seed=3, **kw)
    frames = [synth.make_frame(40 + i, H, W, kind) for i in range(3)]
    write_model(str(tmp_path / "m.acfm"), model)
    (tmp_path / "f.raw").write_bytes(np.stack(frames).tobytes())
    p = run(cli, ["--model", str(tmp_path / "m.acfm"), "--frames", str(tmp_path / "f.raw"), "--rows", str(W), "--cols", str(H),
                  "--channels", "3", "--count", "3"] + extra + mode)
    got = parse(p.stdout)
    want = _oracle_dets(oracle, model, frames, H, W, 3)
    assert sum(len(w) for w in want) > 0
    assert [len(g) for g in got] == [len(w) for w in want], p.stderr
    for g, w in zip(got, want):
        assert g == w, [(a, b) for a, b in zip(g, w) if a != b][:5]


@pytest.mark.gpu
def test_cli_nms_and_calibration(cli, oracle, tmp_path):
    H, W = 96, 128
    model = synth.make_model(seed=3, name="TINY", nTrees=96)
    frames = [synth.make_frame(40, H, W, "luv")]
    write_model(str(tmp_path / "m.acfm"), model)
    (tmp_path / "f.raw").write_bytes(np.stack(frames).tobytes())
    p = run(cli, ["--model", str(tmp_path / "m.acfm"), "--frames", str(tmp_path / "f.raw"), "--rows", str(W), "--cols", str(H),
                  "--channels", "3", "--count", "1", "--luv", "--nms", "--max-count", "5", "--casc-cal", "0.01"])
    got = parse(p.stdout)[0]
    m2 = dict(model)
    m2["hs"] = (model["hs"] + np.float32(0.01)).astype(np.float32)  # acfModify.cpp:143
    want = _oracle_dets(oracle, m2, frames, H, W, 3)[0]
    scores = [float(np.uint32(w[4]).view(np.float32)) for w in want]
    # HipDetector's default pNms (maxg, overlap .65, ovrDnm min) + prune(5, 0): on the device before the records leave it
    keep = oracle.nms([w[:4] for w in want], scores, capi.make_nms(type="maxg", overlap=0.65, ovrDnm="min", prune=True, maxCount=5, pruneRatio=0.0))
    assert len(got) > 0
    assert [g[:4] for g in got] == [tuple(want[i][:4]) for i in keep]
    assert [g[4] for g in got] == [want[i][4] for i in keep]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [[], ["--stream", "2"], ["--stream", "2", "--nms"]])
def test_cli_packed_u8_and_stream(cli, oracle, tmp_path, mode):
    """The CV_8UC3 image entry and the streaming front end of acf::HipDetector against the oracle's
    restated entry (acfo_ingest_u8 -> chnsPyramid -> acfDetect [-> restated NMS + prune])."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_ingest import make_u8, _oracle_planar
    H, W, pix = 120, 160, capi.PIX_BGR
    model = synth.make_model(seed=3, name="INRIA", nTrees=64, cascThr=-3.0, nOctUp=0)
    write_model(str(tmp_path / "m.acfm"), model)
    bufs = [make_u8(60 + i, H, W, pix)[0] for i in range(5)]
    (tmp_path / "f.u8").write_bytes(np.stack(bufs).tobytes())
    p = run(cli, ["--model", str(tmp_path / "m.acfm"), "--frames", str(tmp_path / "f.u8"), "--u8", "bgr", "--rows", str(H), "--cols", str(W),
                  "--count", "5", "--max-count", "6"] + mode)
    got = parse(p.stdout)
    plan = oracle.Plan(model, H, W, 3)
    assert len(got) == 5
    total = 0
    for f in range(5):
        pyr, _, _ = oracle.chns_pyramid(plan, _oracle_planar(oracle, bufs[f], W * 3, H, W, pix))
        det, _ = oracle.detect(plan, pyr)
        want = [(int(d["x"]), int(d["y"]), int(d["w"]), int(d["h"]), int(np.float32(d["score"]).view(np.uint32))) for d in det]
        if "--nms" in mode:
            scores = [float(np.uint32(w[4]).view(np.float32)) for w in want]
            keep = oracle.nms([w[:4] for w in want], scores, capi.make_nms(type="maxg", overlap=0.65, ovrDnm="min", prune=True, maxCount=6, pruneRatio=0.0))
            assert got[f] == [want[i] for i in keep], f
        else:
            assert got[f] == want, f
        total += len(want)
    assert total > 0


def test_default_options_match_the_references_own_tests(cli):
    """The only values the reference's test-suite pins for this path are the default parameters
    (src/test/test-acf-api.cpp:354-371 testChnsDefault, :444-452 pyramid defaults): acf::HipDetector::Options
    starts from the same ones.  Known, deliberate difference: the reference leaves `lambdas` unset (estimated from the
    image, chnsPyramid.cpp:341-374) — the device path wants them given, so Options carries the toolbox's values."""
    p = run(cli, ["--dump-defaults"])
    kv = dict(line.split(" ", 1) for line in p.stdout.strip().splitlines())
    want = {
        "shrink": "4", "color.enabled": "1", "color.smooth": "1", "color.colorSpace": "luv",
        "gradMag.enabled": "1", "gradMag.colorChn": "0", "gradMag.normRad": "5", "gradMag.full": "0",
        "gradHist.enabled": "1", "gradHist.binSize": "0",  # unset in the reference = shrink (chnsCompute.cpp:186-188); 0 means that here
        "gradHist.nOrients": "6", "gradHist.softBin": "0",
        "nPerOct": "8", "nOctUp": "0", "nApprox": "7", "pad": "0 0", "minDs": "16 16", "smooth": "1",
        "good": "0",  # a default-constructed detector has no model (ACF.h:59, good() false)
    }
    for k, v in want.items():
        assert kv[k] == v, (k, kv[k], v)
    assert float(kv["gradMag.normConst"]) == 0.005  # toolbox default (chnsCompute.m)


def _fnv1a(b):
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.gpu
def test_chns_pyramid_logger_taps(cli, oracle, tmp_path):
    """chnsPyramid(..., MatLoggerType): every real scale reports L,U,V / M / Mnorm / O / H with the reference's tags
    (chnsCompute.cpp:241-250,285-300,322-329; gradientMag.cpp:119-123) and the oracle's bytes."""
    H, W = 96, 128
    model = synth.make_model(seed=3, name="TINY", nTrees=32)
    frame = synth.make_frame(41, H, W, "luv")
    write_model(str(tmp_path / "m.acfm"), model)
    (tmp_path / "f.raw").write_bytes(frame.tobytes())
    p = run(cli, ["--model", str(tmp_path / "m.acfm"), "--frames", str(tmp_path / "f.raw"), "--rows", str(W), "--cols", str(H),
                  "--channels", "3", "--count", "1", "--luv", "--log-taps"])
    got = [tuple(l.split()[1:]) for l in p.stdout.splitlines() if l.startswith("tap ")]
    plan = oracle.Plan(model, H, W, 3)
    pyr, taps, chns = oracle.chns_pyramid(plan, frame, want_taps=True, want_chns=True)
    want = []
    nO = model["nOrients"]
    for k, lvl in enumerate(plan.real):
        t = taps[k]
        w1, h1 = t["M"].shape
        for z, name in enumerate("LUV"):
            want.append(("%s:%dx%d" % (name, h1, w1), "%016x" % _fnv1a(np.ascontiguousarray(t["smoothed"][z]).tobytes())))
        for name in ("M", "Mnorm", "O"):
            want.append(("%s:%dx%d" % (name, h1, w1), "%016x" % _fnv1a(np.ascontiguousarray(t[name]).tobytes())))
        c = chns[lvl]                                   # [nChns][wC][hC] raw channels of the real level
        hc = np.concatenate([c[4 + b] for b in range(nO)], axis=1)  # cv::hconcat of the histogram planes
        want.append(("H:%dx%d" % (hc.shape[1], hc.shape[0]), "%016x" % _fnv1a(np.ascontiguousarray(hc).tobytes())))
    assert len(want) >= 7 and got == want


@pytest.mark.gpu
def test_cli_ldcf_model(cli, oracle, tmp_path):
    """acf::HipDetector with Options::ldcfK / ldcfFilters (the LDCF stage of BASELINE cfg 5) through the CLI."""
    H, W = 120, 160
    model = synth.make_model(seed=3, name="TINY", nTrees=64, ldcfK=3, modelDs_h=32, modelDs_w=32, modelDsPad_h=32, modelDsPad_w=32,
                             minDs_h=32, minDs_w=32, cascThr=-2.0)
    frames = [synth.make_frame(70 + i, H, W, "luv") for i in range(2)]
    write_model(str(tmp_path / "m.acfm"), model)
    (tmp_path / "f.raw").write_bytes(np.stack(frames).tobytes())
    p = run(cli, ["--model", str(tmp_path / "m.acfm"), "--frames", str(tmp_path / "f.raw"), "--rows", str(W), "--cols", str(H),
                  "--channels", "3", "--count", "2", "--luv", "--batch"])
    got = parse(p.stdout)
    plan = oracle.Plan(model, H, W, 3)
    total = 0
    for f in range(2):
        pyr, _, _ = oracle.chns_pyramid(plan, frames[f])
        lvL, pyrL, _ = oracle.ldcf(plan, pyr)
        det, _ = oracle.detect_ldcf(plan, lvL, pyrL)
        want = [(int(d["x"]), int(d["y"]), int(d["w"]), int(d["h"]), int(np.float32(d["score"]).view(np.uint32))) for d in det]
        assert got[f] == want
        total += len(want)
    assert total > 0


@pytest.mark.gpu
def test_cli_nms_beyond_device_capacity_is_suppressed_on_the_host(cli, oracle, tmp_path):
    """A frame with more raw detections than ACF_HIP_NMS_CAP (bbNms.cpp has no limit): acf::HipDetector takes the raw list
    and runs its host bbNms + prune instead of failing; the result equals the checker's NMS of the raw list."""
    H, W = 240, 320
    model = synth.make_model(seed=3, name="TINY", nTrees=8, cascThr=-1e6)  # every window is a detection
    frames = [synth.make_frame(70, H, W, "luv")]
    write_model(str(tmp_path / "m.acfm"), model)
    (tmp_path / "f.raw").write_bytes(np.stack(frames).tobytes())
    common = ["--model", str(tmp_path / "m.acfm"), "--frames", str(tmp_path / "f.raw"), "--rows", str(W), "--cols", str(H),
              "--channels", "3", "--count", "1", "--luv"]
    want = _oracle_dets(oracle, model, frames, H, W, 3)[0]
    assert len(want) > capi.NMS_CAP
    for mode in ([], ["--batch"]):
        got = parse(run(cli, common + ["--nms", "--max-count", "12"] + mode).stdout)[0]
        scores = [float(np.uint32(w[4]).view(np.float32)) for w in want]
        keep = oracle.nms([w[:4] for w in want], scores, capi.make_nms(type="maxg", overlap=0.65, ovrDnm="min", prune=True, maxCount=12, pruneRatio=0.0))
        assert 0 < len(keep) <= 12
        assert [g[:4] for g in got] == [tuple(want[i][:4]) for i in keep]
        assert [g[4] for g in got] == [want[i][4] for i in keep]


@pytest.mark.gpu
def test_cli_pyramid_as_roi_atlas(cli, oracle, tmp_path):
    """Pyramid::rois (ACF.h:377-378): every level handed over as one atlas plane with a roi per channel — the form the GL
    backend's read-back has — goes through computeChannelIndex's addressing (acfDetect1.cpp:346-366) to the same detections."""
    H, W = 96, 128
    model = synth.make_model(seed=3, name="TINY", nTrees=96)
    frames = [synth.make_frame(40 + i, H, W, "luv") for i in range(2)]
    write_model(str(tmp_path / "m.acfm"), model)
    (tmp_path / "f.raw").write_bytes(np.stack(frames).tobytes())
    p = run(cli, ["--model", str(tmp_path / "m.acfm"), "--frames", str(tmp_path / "f.raw"), "--rows", str(W), "--cols", str(H),
                  "--channels", "3", "--count", "2", "--luv", "--via-atlas"])
    got = parse(p.stdout)
    want = _oracle_dets(oracle, model, frames, H, W, 3)
    assert sum(len(w) for w in want) > 0 and got == want


@pytest.mark.gpu
@pytest.mark.parametrize("devices,count", [("0", 5), ("0,0", 5), ("0,0,0", 2)])
def test_cli_detector_pool_shards_frames(cli, oracle, tmp_path, devices, count):
    """acf::HipDetectorPool: one detector per listed device (the same device several times = independent contexts: what a
    one-GPU box can exercise), frames in contiguous blocks that differ by at most one, results in frame order."""
    H, W = 96, 128
    model = synth.make_model(seed=3, name="TINY", nTrees=96)
    frames = [synth.make_frame(60 + i, H, W, "luv") for i in range(count)]
    write_model(str(tmp_path / "m.acfm"), model)
    (tmp_path / "f.raw").write_bytes(np.stack(frames).tobytes())
    p = run(cli, ["--model", str(tmp_path / "m.acfm"), "--frames", str(tmp_path / "f.raw"), "--rows", str(W), "--cols", str(H),
                  "--channels", "3", "--count", str(count), "--luv", "--pool", "--pool-devices", devices])
    assert "pool: %d detector(s)" % len(devices.split(",")) in p.stderr
    assert parse(p.stdout) == _oracle_dets(oracle, model, frames, H, W, 3)
