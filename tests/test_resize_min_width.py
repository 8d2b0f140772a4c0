"""f4: the apps' resize to a minimum object width (src/app/acf/acf.cpp:117-148 `Resizer`; GPUDetectionPipeline.cpp:250-266) on the
device, in front of the 8-bit entries.  cv::resize is OpenCV's and absent from the reference tree and from this image: the oracle
restates the published CV_8U algorithm (oracle/acf_oracle.c:acfo_resize_u8) — PARITY UNPINNED; what is tested is (CPU) the
restatement against the properties the algorithm implies and (GPU) k_resize_u8 and the whole resized detection against it, bit for bit."""
import os
import sys

import numpy as np
import pytest

from acf_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _img(seed, rows, cols, cn):
    return (synth.uniform(seed, rows * cols * cn, 5).reshape(rows, cols, cn) * 255).astype(np.uint8)


def test_oracle_resize_properties(oracle):
    rng = np.random.RandomState(3)
    # sizes: cvRound (half to even) of rows * scale
    assert oracle.resize_dims(101, 50, 0.5) == (50, 25) and oracle.resize_dims(103, 51, 0.5) == (52, 26)
    # a constant image stays constant under every form (weights sum to 1; the fixed-point weights to 2048)
    for scale in (0.5, 1.0 / 3.0, 0.48, 0.8, 1.0, 1.25, 2.0):
        for v in (0, 1, 127, 255):
            img = np.full((37, 53, 3), v, np.uint8)
            out = oracle.resize_u8(img, scale)
            assert out.shape[:2] == oracle.resize_dims(37, 53, scale) and (out == v).all(), (scale, v)
    # exactly 1/2: (a + b + c + d + 2) >> 2 of each 2 x 2 cell
    img = rng.randint(0, 256, (40, 64, 3)).astype(np.uint8)
    out = oracle.resize_u8(img, 0.5)
    s = img.astype(np.int32)
    want = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    assert np.array_equal(out, want.astype(np.uint8))
    # exactly 1/3 and 1/4: cvRound(int sum * float(1 / area)), half to even
    for k in (3, 4):
        img = rng.randint(0, 256, (12 * k, 10 * k, 1)).astype(np.uint8)
        out = oracle.resize_u8(img, 1.0 / k)
        sums = img.astype(np.int64).reshape(12, k, 10, k).sum(axis=(1, 3))
        want = np.rint((sums.astype(np.float32) * np.float32(1.0 / (k * k))).astype(np.float64)).astype(np.uint8)
        assert np.array_equal(out[..., 0], want), k
    # scale 1 under INTER_LINEAR is the identity; fractional area keeps the mean within rounding
    img = rng.randint(0, 256, (33, 47, 4)).astype(np.uint8)
    assert np.array_equal(oracle.resize_u8(img, 1.0), img)
    img = rng.randint(0, 256, (120, 160, 3)).astype(np.uint8)
    out = oracle.resize_u8(img, 0.6)
    assert abs(out.mean() - img.mean()) < 0.6
    # channels are independent: resizing one channel alone gives that channel of the interleaved result
    for scale in (0.37, 1.6):
        out = oracle.resize_u8(img, scale)
        for c in range(3):
            assert np.array_equal(oracle.resize_u8(img[..., c:c + 1].copy(), scale)[..., 0], out[..., c]), (scale, c)
    # Resizer::operator()(objects): float products, cvRound
    assert oracle.unscale_rect(0.5, [3, 5, 40, 41]) == [6, 10, 80, 82]
    sc = float(np.float32(80) / np.float32(150))
    inv = np.float32(1.0) / np.float32(sc)
    assert oracle.unscale_rect(sc, [7, 9, 80, 80]) == [int(np.rint(float(np.float32(v) * inv))) for v in (7, 9, 80, 80)]


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [0.5, 1.0 / 3.0, 0.25, 80.0 / 150.0, 0.37, 0.9, 1.0, 1.3, 2.0])
@pytest.mark.parametrize("cn", [1, 3, 4])
def test_k_resize_u8_equals_the_restated_cv_resize(oracle, scale, cn):
    from acf_amd.detector import HipDetector
    scale = float(np.float32(scale))  # Resizer's scale is a float
    model = synth.make_model(seed=3, name="TINY", nTrees=8)
    det = HipDetector(model, 96, 128, 3, max_batch=1)
    for (rows, cols) in ((61, 83), (128, 96), (240, 321)):
        img = _img(rows + cn, rows, cols, cn)
        got = det.op_resize_u8(img, scale)
        want = oracle.resize_u8(img, scale)
        assert got.shape == want.shape and np.array_equal(got, want), (scale, cn, rows, cols, int(np.abs(got.astype(int) - want.astype(int)).max()))
    det.close()


@pytest.mark.gpu
@pytest.mark.parametrize("min_width", [160, 107, 64])
def test_detection_with_min_object_width(oracle, min_width):
    """Python mirror: reduce on the device, detect, map the boxes back == the oracle doing the same three steps on the CPU."""
    import torch
    from acf_amd.detector import HipDetector
    from test_gpu_ingest import make_u8, _oracle_planar
    H, W, pix = 360, 480, capi.PIX_RGB
    model = synth.make_model(seed=3, name="FACE80", nTrees=64, cascThr=-4.0)
    scale = HipDetector.resize_scale(model["modelDs_w"], min_width)
    dr, dc = HipDetector.resize_dims(H, W, scale)
    assert (dr, dc) == oracle.resize_dims(H, W, scale)
    nF = 2
    det = HipDetector(model, dr, dc, 3, max_batch=nF, max_hits=1 << 17)
    det.set_input_resize(H, W, scale)
    bufs = [make_u8(80 + i, H, W, pix)[0] for i in range(nF)]
    det.run_u8(torch.from_numpy(np.stack(bufs)).cuda(), pix, W * 3)
    plan = oracle.Plan(model, dr, dc, 3)
    total = 0
    for f in range(nF):
        red = oracle.resize_u8(bufs[f].reshape(H, W, 3), scale)
        planar = _oracle_planar(oracle, red.reshape(dr, dc * 3), dc * 3, dr, dc, pix)
        pyr, _, _ = oracle.chns_pyramid(plan, planar)
        want, _ = oracle.detect(plan, pyr)
        got, _ = det.detections(f)
        assert np.array_equal(det.read_pyramid(f).view(np.uint32), pyr.view(np.uint32)), (min_width, f)
        assert got.tobytes() == want.tobytes()
        back = HipDetector.unscale_boxes(got, scale)
        for g, w_ in zip(back, want):
            assert [int(g["x"]), int(g["y"]), int(g["w"]), int(g["h"])] == oracle.unscale_rect(scale, [w_["x"], w_["y"], w_["w"], w_["h"]])
        total += len(want)
    assert total > 0
    # switching it off again: frames of the plan's own size
    det.set_input_resize(0, 0, 1.0)
    det.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [[], ["--stream", "2", "--nms"]])
def test_cli_min_width(oracle, tmp_path, mode):
    """acf::HipDetector::setMinObjectWidth through the C++ CLI (operator()(packed) and the streaming entries)."""
    from test_gpu_ingest import make_u8, _oracle_planar
    from test_host_cpp import parse, run
    from acf_amd.modelio import write_model
    clip = os.path.join(ROOT, "acf_amd", "host", "acf_hip_detect")
    H, W, pix, min_width = 240, 320, capi.PIX_BGR, 150
    model = synth.make_model(seed=3, name="FACE80", nTrees=64, cascThr=-4.0)
    write_model(str(tmp_path / "m.acfm"), model)
    bufs = [make_u8(90 + i, H, W, pix)[0] for i in range(3)]
    (tmp_path / "f.u8").write_bytes(np.stack(bufs).tobytes())
    p = run(clip, ["--model", str(tmp_path / "m.acfm"), "--frames", str(tmp_path / "f.u8"), "--u8", "bgr", "--rows", str(H), "--cols", str(W),
                   "--count", "3", "--max-count", "6", "--min-width", str(min_width), "--luv"] + mode)  # (FACE80 takes its planes as they are)
    got = parse(p.stdout)
    scale = float(np.float32(model["modelDs_w"]) / np.float32(min_width))
    dr, dc = oracle.resize_dims(H, W, scale)
    plan = oracle.Plan(model, dr, dc, 3)
    total = 0
    for f in range(3):
        red = oracle.resize_u8(bufs[f].reshape(H, W, 3), scale)
        pyr, _, _ = oracle.chns_pyramid(plan, _oracle_planar(oracle, red.reshape(dr, dc * 3), dc * 3, dr, dc, pix))
        det, _ = oracle.detect(plan, pyr)
        want = [(int(d["x"]), int(d["y"]), int(d["w"]), int(d["h"]), int(np.float32(d["score"]).view(np.uint32))) for d in det]
        if "--nms" in mode:
            scores = [float(np.uint32(w[4]).view(np.float32)) for w in want]
            keep = oracle.nms([w[:4] for w in want], scores, capi.make_nms(type="maxg", overlap=0.65, ovrDnm="min", prune=True, maxCount=6, pruneRatio=0.0))
            want = [want[i] for i in keep]
        want = [tuple(oracle.unscale_rect(scale, w[:4])) + (w[4],) for w in want]
        assert got[f] == want, f
        total += len(want)
    assert total > 0
