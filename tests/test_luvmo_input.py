"""Frames that arrive with their own gradient magnitude and orientation planes: an input of FIVE planes — the GL pipeline's LUVMO —
whose planes 3, 4 replace gradientMag (and its normalisation) at the first real scale (L/chnsPyramid.cpp:248-255,318-322,
L/chnsCompute.cpp:219-226,265-269).  No reference vectors exist (restated); what can be checked independently of the restatement:
if the two planes ARE what chnsCompute would have computed — the normalised magnitude and the orientation of the smoothed gradient
plane at scale 1 — the pyramid must be the three-plane pyramid bit for bit.  Then the HIP path against the oracle on arbitrary M, O."""
import numpy as np
import pytest

from acf_amd import synth


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


CASES = [("luv", dict(name="TINY", nTrees=64, cascThr=-2.0), 120, 160),
         ("rgb", dict(name="TINY", nTrees=64, cascThr=-2.0, isLuv=0), 96, 128),
         ("luv", dict(name="FACE80", nTrees=128, minDs_h=80, minDs_w=80), 272, 480)]


def _with_own_mo(oracle, model, frame, H, W):
    plan3 = oracle.Plan(model, H, W, 3)
    pyr3, taps, _ = oracle.chns_pyramid(plan3, frame, want_taps=True)
    assert plan3.levels[0].scale == 1.0
    return pyr3, np.concatenate([frame, taps[0]["Mnorm"][None], taps[0]["O"][None]]).astype(np.float32)


@pytest.mark.parametrize("kind,kw,H,W", CASES)
def test_oracle_five_planes_with_the_images_own_gradients_reproduce_three_planes(oracle, kind, kw, H, W):
    model = synth.make_model(seed=3, **kw)
    frame = synth.make_frame(31, H, W, kind)
    pyr3, frame5 = _with_own_mo(oracle, model, frame, H, W)
    plan5 = oracle.Plan(model, H, W, 5)
    pyr5, _, _ = oracle.chns_pyramid(plan5, frame5)
    assert np.array_equal(bits(pyr5), bits(pyr3))
    # and other M, O planes do change the magnitude / histogram channels of the levels that hang off the first real scale
    f2 = frame5.copy()
    f2[3] *= np.float32(0.5)
    other, _, _ = oracle.chns_pyramid(plan5, f2)
    assert not np.array_equal(bits(other), bits(pyr3))
    # single scale: chnsCompute with five planes
    c3 = oracle.chns_compute(model, frame)
    c5 = oracle.chns_compute(model, frame5)
    assert np.array_equal(bits(c5), bits(c3))


def test_oracle_refuses_five_planes_when_the_first_real_scale_is_resampled(oracle):
    model = synth.make_model(seed=3, name="INRIA", nTrees=16)      # nOctUp = 1: the first real scale is the up-sampled image
    H, W = 120, 160
    plan = oracle.Plan(model, H, W, 5)
    with pytest.raises(RuntimeError):
        oracle.chns_pyramid(plan, np.zeros((5, W, H), np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("kind,kw,H,W", CASES)
def test_hip_five_plane_frames_match_the_oracle(oracle, kind, kw, H, W):
    import torch
    from acf_amd.detector import HipDetector
    model = synth.make_model(seed=3, **kw)
    frames5 = []
    for f in range(3):
        frame = synth.make_frame(31 + f, H, W, kind)
        _, f5 = _with_own_mo(oracle, model, frame, H, W)
        if f:   # arbitrary planes: another frame's magnitude, a shifted orientation
            f5[3] = np.roll(f5[3], 7 * f, axis=0) * np.float32(1.0 + 0.25 * f)
            f5[4] = np.roll(f5[4], 5 * f, axis=1)
        frames5.append(f5)
    frames5 = np.stack(frames5)
    plan5 = oracle.Plan(model, H, W, 5)
    det = HipDetector(model, H, W, 5, max_batch=3, max_hits=1 << 15)
    det.run(torch.from_numpy(frames5).cuda())
    total = 0
    for f in range(3):
        pyr, _, _ = oracle.chns_pyramid(plan5, frames5[f])
        want, whits = oracle.detect(plan5, pyr)
        assert np.array_equal(bits(det.read_pyramid(f)), bits(pyr)), f
        d, h = det.detections(f)
        assert d.tobytes() == want.tobytes() and h.tobytes() == whits.tobytes(), f
        total += len(want)
    # frame 0 carries the image's own M, O: the three-plane result
    plan3 = oracle.Plan(model, H, W, 3)
    pyr3, _, _ = oracle.chns_pyramid(plan3, np.ascontiguousarray(frames5[0][:3]))
    assert np.array_equal(bits(det.read_pyramid(0)), bits(pyr3))
    # the single-scale entry
    got = det.chns_compute(frames5[1], model)
    assert np.array_equal(bits(got), bits(oracle.chns_compute(model, frames5[1])))
    det.close()


@pytest.mark.gpu
def test_hip_refuses_what_the_reference_cannot_do_with_five_planes():
    from acf_amd.detector import HipDetector, HipError
    model = synth.make_model(seed=3, name="INRIA", nTrees=16)      # nOctUp = 1
    with pytest.raises(HipError):
        HipDetector(model, 120, 160, 5, max_batch=1)
    ok = synth.make_model(seed=3, name="TINY", nTrees=16)
    with pytest.raises(HipError):
        HipDetector(ok, 120, 160, 4, max_batch=1)                    # 4 planes: neither an image nor image + M, O


@pytest.mark.gpu
def test_cpp_host_takes_five_plane_matp(oracle, tmp_path):
    """acf::HipDetector::operator()(const MatP&) with a five-plane MatP (image + M, O) through the CLI, and the static chnsCompute."""
    import os
    import subprocess
    from acf_amd import capi
    from acf_amd.modelio import write_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = os.path.join(root, "acf_amd", "host")
    subprocess.check_call(["make", "-s", "-C", host])
    H, W = 120, 160
    model = synth.make_model(seed=3, name="TINY", nTrees=64, cascThr=-2.0)
    frames5 = []
    for f in range(2):
        _, f5 = _with_own_mo(oracle, model, synth.make_frame(31 + f, H, W, "luv"), H, W)
        f5[3] = np.roll(f5[3], 9, axis=0)
        frames5.append(f5)
    write_model(str(tmp_path / "m.acfm"), model)
    (tmp_path / "f.raw").write_bytes(np.stack(frames5).tobytes())
    e = dict(os.environ)
    e["ACF_HIP_LIBRARY"] = capi.LIB_PATH
    base = [os.path.join(host, "acf_hip_detect"), "--model", str(tmp_path / "m.acfm"), "--frames", str(tmp_path / "f.raw"), "--rows", str(W), "--cols", str(H),
            "--channels", "5", "--count", "2", "--luv"]
    p = subprocess.run(base, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert p.returncode == 0, p.stderr
    got, cur = [], None
    for line in p.stdout.splitlines():
        t = line.split()
        if t[0] == "frame":
            cur = []
            got.append(cur)
        else:
            cur.append((int(t[0]), int(t[1]), int(t[2]), int(t[3]), int(t[5], 16)))
    plan5 = oracle.Plan(model, H, W, 5)
    total = 0
    for f in range(2):
        pyr, _, _ = oracle.chns_pyramid(plan5, frames5[f])
        det, _ = oracle.detect(plan5, pyr)
        want = [(int(d["x"]), int(d["y"]), int(d["w"]), int(d["h"]), int(np.float32(d["score"]).view(np.uint32))) for d in det]
        assert got[f] == want, f
        total += len(want)
    assert total > 0
    out = tmp_path / "c.raw"
    p = subprocess.run(base + ["--chns", str(out), "--log-taps"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert p.returncode == 0, p.stderr
    chn = np.fromfile(str(out), np.float32).reshape(2, 10, W // 4, H // 4)
    for f in range(2):
        assert np.array_equal(bits(chn[f]), bits(oracle.chns_compute(model, frames5[f]))), f
    taps = [l.split()[1].split(":")[0] for l in p.stdout.splitlines() if l.startswith("tap ")]
    assert taps[:4] == ["L", "U", "V", "H"]      # (M, Mnorm, O are not logged when they came with the image)
