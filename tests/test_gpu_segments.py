"""k_smooth_vec's speculative column segments (acf_amd/csrc/kernels.hip.h): the image smoothing's recursion along image-x
(convTri1 called in place, chnsCompute.cpp:239: SURVEY.md H2) is cut into segments that start `smooth_warm` columns early,
and every hand-over is checked bit for bit on the device; a plane with a difference is recomputed as one chain.  Whatever
the segmentation, the pyramid must be the oracle's, bit for bit — including when the warm-up is too short to converge
(the repair launch does the work), when every plane is forced through the repair, and on frames with exactly-zero regions."""
import numpy as np
import pytest

from acf_amd import synth

pytestmark = pytest.mark.gpu


def _check(oracle, det, frames, model, H, W):
    import torch
    det.run(torch.from_numpy(frames).cuda())
    plan = oracle.Plan(model, H, W, 3)
    for f in range(len(frames)):
        pyr, _, _ = oracle.chns_pyramid(plan, frames[f])
        want, whits = oracle.detect(plan, pyr)
        assert np.array_equal(det.read_pyramid(f).view(np.uint32), pyr.view(np.uint32)), f
        d, h = det.detections(f)
        assert d.tobytes() == want.tobytes() and h.tobytes() == whits.tobytes()


# gradMag as its own kernel / inside the gradient plane's chain (k_smooth_grad) / with convTri's x pass there too (k_smooth_grad_tri: that plane is one segment)
@pytest.mark.parametrize("fused_grad,fused_tri", [(0, 0), (2, 0), (2, 2)])
@pytest.mark.parametrize("segments,warm,force", [(0, 48, 0), (1, 48, 0), (4, 48, 0), (7, 32, 0), (16, 16, 0), (5, 48, 1)])
def test_segmented_smoothing_is_bit_exact(oracle, segments, warm, force, fused_grad, fused_tri):
    from acf_amd.detector import HipDetector
    H, W = 256, 512   # w % 8 == 0, h % 4 == 0: the vector smoothing kernel; every level goes through the fused level kernel
    model = synth.make_model(seed=3, name="TINY", nTrees=96, cascThr=-1.0)
    frames = np.stack([synth.make_frame(51 + i, H, W, "luv") for i in range(3)])
    # a black band and a black box: regions where the recursion's state is exactly zero
    frames[1, :, 160:290, :] = 0.0
    frames[2, :, 330:, 100:180] = 0.0
    det = HipDetector(model, H, W, 3, max_batch=3, max_hits=1 << 15)
    det.set_option("smooth_segments", segments)
    det.set_option("smooth_warm", warm)
    det.set_option("smooth_force_redo", force)
    det.set_option("fused_grad", fused_grad)
    det.set_option("fused_tri", fused_tri)
    # the level chains' segments too (off by default): as many as the smoothing's, with a warm-up short enough to miss sometimes
    det.set_option("level_segments", 0 if segments == 0 else min(segments, 8))
    det.set_option("level_warm", min(warm, 32))
    det.set_option("count_repairs", 1)
    _check(oracle, det, frames, model, H, W)
    r = det.repairs()
    assert r[0] > 0 or segments == 1
    if force:
        # every smoothing plane and every level plane with more than one segment was repaired (k_smooth_grad_tri's plane is one chain)
        assert (r[1] == r[0] if not fused_tri else r[0] * 2 // 3 <= r[1] < r[0]) and r[3] > 0
    if segments > 1:
        assert r[2] > 0                            # the level chains were cut too
    if segments == 16:
        assert r[1] > 0                            # 16-column warm-ups DO miss: the repair is what makes the result right
    det.close()


def test_shared_device_forms_are_bit_exact(oracle):
    """Option shared_device (what the pools set on contexts that share a device): batches of >= 64 frames keep their smoothing chains
    uncut — nothing is verified or repaired — and convTri's x pass rides on the gradient plane's chain (k_smooth_grad_tri)."""
    import torch
    from acf_amd.detector import HipDetector
    H, W = 256, 512
    model = synth.make_model(seed=3, name="TINY", nTrees=96, cascThr=-1.0)
    base = [synth.make_frame(51 + i, H, W, "luv") for i in range(4)]
    base[1][:, 160:290, :] = 0.0
    frames = np.stack([base[i % 4] for i in range(64)])
    det = HipDetector(model, H, W, 3, max_batch=64, max_hits=1 << 15)
    det.set_option("shared_device", 1)
    det.set_option("fused_grad", 2)       # (planes this small are below the default's 2^20 pixels)
    det.set_option("level_segments", 1)
    det.set_option("count_repairs", 1)
    det.run(torch.from_numpy(frames).cuda())
    assert det.repairs()[0] == 0          # no smoothing plane had a hand-over to check
    plan = oracle.Plan(model, H, W, 3)
    for f in (0, 1, 62, 63):
        pyr, _, _ = oracle.chns_pyramid(plan, frames[f])
        want, whits = oracle.detect(plan, pyr)
        assert np.array_equal(det.read_pyramid(f).view(np.uint32), pyr.view(np.uint32)), f
        d, h = det.detections(f)
        assert d.tobytes() == want.tobytes() and h.tobytes() == whits.tobytes()
    # the same batch with the lone context's forms: segments, k_tri_x5v
    det.set_option("shared_device", 0)
    det.run(torch.from_numpy(frames).cuda())
    assert det.repairs()[0] > 0
    det.close()


@pytest.mark.parametrize("fused_tri", [0, 2])
@pytest.mark.parametrize("H,W", [(256, 512), (136, 484)])
def test_one_plane_input_with_its_colour_channel(oracle, H, W, fused_tri):
    """A grey frame whose one plane is both the colour channel and the gradient plane (8 channels): the scale's smoothing is then
    k_smooth_grad's launch alone (no k_smooth_vec beside it) — with convTri's x pass on the chain as well."""
    import torch
    from acf_amd import capi
    from acf_amd.detector import HipDetector
    model = synth.make_model(seed=3, name="TINY", nTrees=96, cascThr=-1.0, colorSpace=capi.CS_GRAY, isLuv=0)
    frames = np.stack([synth.make_frame(91 + i, H, W, "gray") for i in range(2)])
    det = HipDetector(model, H, W, 1, max_batch=2, max_hits=1 << 15)
    det.set_option("fused_grad", 2)
    det.set_option("fused_tri", fused_tri)
    det.set_option("smooth_segments", 3)
    det.set_option("smooth_warm", 32)
    det.run(torch.from_numpy(frames).cuda())
    plan = oracle.Plan(model, H, W, 1)
    for f in range(2):
        pyr, _, _ = oracle.chns_pyramid(plan, frames[f])
        want, whits = oracle.detect(plan, pyr)
        assert np.array_equal(det.read_pyramid(f).view(np.uint32), pyr.view(np.uint32)), f
        d, h = det.detections(f)
        assert d.tobytes() == want.tobytes() and h.tobytes() == whits.tobytes()
    det.close()


def test_segmented_smoothing_rgb_and_sub_batches(oracle):
    """RGB input (the smoothing follows rgb2luv), sub-batch contexts (option streams) and the rank-cell cascade together."""
    import torch
    from acf_amd.detector import HipDetector
    H, W = 200, 512
    model = synth.make_model(seed=5, name="TINY", nTrees=96, cascThr=-1.0)
    frames = np.stack([synth.make_frame(71 + i, H, W, "rgb") for i in range(4)])
    det = HipDetector(streams=2)
    det.set_model(model)
    det.set_option("smooth_segments", 6)
    det.set_option("smooth_warm", 32)
    det.set_option("fused_grad", 2)
    det.plan(H, W, 3, max_batch=4, max_hits=1 << 15)
    _check(oracle, det, frames, model, H, W)
    det.close()


@pytest.mark.parametrize("fused_grad,fused_tri", [(0, 0), (2, 0), (2, 2)])
@pytest.mark.parametrize("W,segments", [(484, 0), (484, 3), (492, 1), (488, 1), (496, 0), (48, 0), (52, 1), (20, 0), (36, 2)])
def test_widths_that_are_not_multiples_of_eight(oracle, W, segments, fused_grad, fused_tri):
    """w % 8 == 4: the vector smoothing kernel's four-column tail (the third real scale of a 1080p frame is 484 columns wide)."""
    from acf_amd.detector import HipDetector
    H = 136 if W > 100 else 64
    model = synth.make_model(seed=3, name="TINY", nTrees=64, cascThr=-1.0)
    frames = np.stack([synth.make_frame(81 + i, H, W, "luv") for i in range(2)])
    det = HipDetector(model, H, W, 3, max_batch=2, max_hits=1 << 15)
    det.set_option("smooth_segments", segments)
    det.set_option("smooth_warm", 16)
    det.set_option("fused_grad", fused_grad)
    det.set_option("fused_tri", fused_tri)
    import torch
    det.run(torch.from_numpy(frames).cuda())
    plan = oracle.Plan(model, H, W, 3)
    for f in range(2):
        pyr, _, _ = oracle.chns_pyramid(plan, frames[f])
        assert np.array_equal(det.read_pyramid(f).view(np.uint32), pyr.view(np.uint32)), (W, f)
    det.close()


def test_default_level_segments_on_a_single_frame_with_flat_bands(oracle, capsys):
    """The defaults at batch 1 (level_segments 0 = auto: the level chains are cut, level_warm 32) on frames with black and flat
    bands, where channel planes are exactly zero and a segment's warm-up may not reproduce the chain's state: results stay the
    oracle's, and the number of planes the repair launch had to recompute is counted so that the cost of that path stays visible."""
    from acf_amd.detector import HipDetector
    H, W = 480, 640
    model = synth.make_model(seed=3, name="TINY", nTrees=96, cascThr=-1.0)
    total = [0, 0, 0, 0]
    for k in range(3):
        frame = synth.make_frame(91 + k, H, W, "luv")[None].copy()
        if k == 0:
            frame[0, :, 200:330, :] = 0.0            # a black band of columns
        elif k == 1:
            frame[0, :, :, 100:260] = 0.25           # a flat band of rows (no gradient: zero magnitude / histogram cells)
        det = HipDetector(model, H, W, 3, max_batch=1, max_hits=1 << 15)
        det.set_option("count_repairs", 1)
        _check(oracle, det, frame, model, H, W)
        r = det.repairs()
        assert r[2] > 0, "batch 1 should cut the level chains into segments by default"
        total = [a + b for a, b in zip(total, r)]
        det.close()
    with capsys.disabled():
        print("\nlevel-chain planes repaired at batch 1 with the defaults: %d of %d (image smoothing: %d of %d)" % (total[3], total[2], total[1], total[0]))
    assert total[3] <= total[2] // 4, "more than a quarter of the level planes needed the repair launch: the default warm-up is too short"


def test_repair_flags_are_taken_down_between_calls(oracle):
    """The repair flags are cleared at plan time (on the context's stream) and by the repair launch that reads them — not per call.
    A run with every plane forced through the repair, then a normal run on the SAME context: the second run must repair nothing
    (warm-ups of 64 columns converge on these frames) and still be the oracle's."""
    import torch
    from acf_amd.detector import HipDetector
    H, W = 256, 512
    model = synth.make_model(seed=3, name="TINY", nTrees=96, cascThr=-1.0)
    frames = np.stack([synth.make_frame(51 + i, H, W, "luv") for i in range(3)])
    det = HipDetector(model, H, W, 3, max_batch=3, max_hits=1 << 15)
    det.set_option("smooth_segments", 4)
    det.set_option("smooth_warm", 64)
    det.set_option("level_segments", 4)
    det.set_option("level_warm", 32)
    det.set_option("count_repairs", 1)
    det.set_option("smooth_force_redo", 1)
    det.run(torch.from_numpy(frames).cuda())
    r1 = det.repairs()
    assert r1[1] > 0 and r1[3] > 0
    det.set_option("smooth_force_redo", 0)
    _check(oracle, det, frames, model, H, W)
    r2 = det.repairs()
    assert r2[0] > r1[0] and r2[1] == r1[1], (r1, r2)    # planes were checked again, none recomputed
    assert r2[3] == r1[3], (r1, r2)
    det.close()


def test_run_and_pyramid_plus_detect_alternate_on_one_context(oracle):
    """acf_hip_run clears the tiled cascade's counters in front of the pyramid's launches and tells runCascade so; pyramid() + detect()
    clear them in the cascade itself.  Both orders on one context, twice over: the hits stay the oracle's (no stale counter adds up)."""
    import torch
    from acf_amd.detector import HipDetector
    H, W = 272, 480
    model = synth.make_model(seed=3, name="FACE80", nTrees=256, minDs_h=80, minDs_w=80)
    frames = np.stack([synth.make_frame(71 + i, H, W, "luv") for i in range(4)])
    det = HipDetector(model, H, W, 3, max_batch=4, max_hits=1 << 15)
    fr = torch.from_numpy(frames).cuda()
    plan = oracle.Plan(model, H, W, 3)
    want = []
    for f in range(4):
        pyr, _, _ = oracle.chns_pyramid(plan, frames[f])
        want.append(oracle.detect(plan, pyr))
    assert sum(len(w[1]) for w in want) > 0
    for order in ("run", "split", "run", "split", "split", "run"):
        if order == "run":
            det.run(fr)
        else:
            det.pyramid(fr)
            det.detect()
        for f in range(4):
            d, h = det.detections(f)
            assert d.tobytes() == want[f][0].tobytes() and h.tobytes() == want[f][1].tobytes(), (order, f)
    det.close()
