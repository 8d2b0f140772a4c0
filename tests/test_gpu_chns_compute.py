"""Detector::chnsCompute / computeChannels / setLogger (ACF.h:327-349,419-420,578-581; chnsCompute.cpp:146-338) — the boundary
entries that compute the channels of ONE image at its own scale, without a pyramid plan:

 * C ABI acf_hip_chns_compute against the oracle's acfo_chns_compute (crop, every colour space, channel subsets, both shrinks,
   sizes that are not multiples of shrink), bit for bit; in the reference's arithmetic too (option "arith");
 * the oracle's single-scale function against its own pyramid (scale 1 of chnsPyramid IS chnsCompute of the converted frame):
   CPU, so the checker is not a second opinion of itself only;
 * C++ acf::HipDetector::chnsCompute / computeChannels / setLogger / setReferenceArithmetic through the CLI.
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


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


CASES = [
    ("luv_in", 96, 128, "luv", 3, dict(name="TINY", isLuv=1)),
    ("luv_in_crop", 97, 131, "luv", 3, dict(name="TINY", isLuv=1)),          # 97 x 131 -> 96 x 128
    ("rgb_to_luv", 120, 150, "rgb", 3, dict(name="INRIA")),                  # 150 % 4 != 0
    ("rgb_to_luv_odd_n", 61, 45, "rgb", 3, dict(name="INRIA")),              # cropped to 60 x 44
    ("gray", 240, 320, "gray", 1, dict(name="FACE64")),
    ("gray_from_rgb", 64, 80, "rgb", 3, dict(name="FACE64", colorEnabled=1)),
    ("hsv", 72, 88, "rgb", 3, dict(name="TINY", colorSpace=capi.CS_HSV, isLuv=0)),
    ("orig", 72, 88, "rgb", 3, dict(name="TINY", colorSpace=capi.CS_ORIG, isLuv=0)),
    ("no_hist", 96, 128, "luv", 3, dict(name="TINY", isLuv=1, gradHistEnabled=0)),
    ("hist_only", 96, 128, "luv", 3, dict(name="TINY", isLuv=1, gradMagEnabled=0, colorEnabled=0)),
    ("no_norm", 96, 128, "luv", 3, dict(name="TINY", isLuv=1, normRad=0)),
    ("shrink2", 96, 130, "luv", 3, dict(name="TINY", isLuv=1, shrink=2, stride=2)),
    ("orients9_full", 96, 128, "luv", 3, dict(name="TINY", isLuv=1, nOrients=9, full=1)),
    ("hardbin", 96, 128, "luv", 3, dict(name="TINY", isLuv=1, softBin=-2)),
    ("vga", 480, 640, "rgb", 3, dict(name="INRIA")),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_single_scale_equals_its_pyramids_scale_one(oracle, case):
    """CPU: acfo_chns_compute (through oracle.chns_compute's crop + conversion) == the raw channels chnsPyramid keeps for scale 1."""
    _, H, W, kind, d, kw = case
    if H % kw.get("shrink", 4) or W % kw.get("shrink", 4):
        pytest.skip("chnsPyramid's scale 1 is the whole image only when it needs no crop")
    model = synth.make_model(seed=3, nTrees=16, **kw)
    frame = synth.make_frame(9, H, W, kind)
    if model["minDs_h"] > H or model["minDs_w"] > W:
        pytest.skip("image smaller than the model")
    plan = oracle.Plan(model, H, W, d)
    if plan.levels[0].scale != 1.0:
        pytest.skip("first scale is not 1 (nOctUp)")
    _, _, chns = oracle.chns_pyramid(plan, frame, want_chns=True)
    got = oracle.chns_compute(model, frame)
    assert np.array_equal(bits(got), bits(chns[0]))


@pytest.fixture(scope="module")
def dev():
    from acf_amd.detector import HipDetector
    d = HipDetector()
    yield d
    d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_chns_compute_equals_the_oracle(oracle, dev, case):
    _, H, W, kind, d, kw = case
    model = synth.make_model(seed=3, nTrees=16, **kw)
    frame = synth.make_frame(9, H, W, kind)
    want = oracle.chns_compute(model, frame)
    got = dev.chns_compute(frame, model)
    assert got.shape == want.shape
    assert np.array_equal(bits(got), bits(want))


@pytest.mark.gpu
@pytest.mark.parametrize("case", [CASES[2], CASES[3], CASES[4], CASES[0]], ids=lambda c: c[0])
def test_chns_compute_in_reference_arithmetic(oracle, case):
    from acf_amd.detector import HipDetector
    _, H, W, kind, d, kw = case
    model = synth.make_model(seed=3, nTrees=16, **kw)
    frame = synth.make_frame(9, H, W, kind)
    oracle.set_x86_tables(*oracle.x86_fixture())
    oracle.set_approx(3)
    try:
        want = oracle.chns_compute(model, frame)
    finally:
        oracle.set_approx(0)
    dev = HipDetector()
    dev.set_x86_tables(*oracle.x86_fixture())
    dev.set_option("arith", 1)
    got = dev.chns_compute(frame, model)
    assert np.array_equal(bits(got), bits(want))
    assert not np.array_equal(bits(got), bits(oracle.chns_compute(model, frame)))
    dev.close()


@pytest.mark.gpu
def test_chns_compute_uses_the_contexts_model_and_reports_sizes(oracle):
    from acf_amd.detector import HipDetector, HipError
    import ctypes as C
    model = synth.make_model(seed=3, name="TINY", nTrees=16, isLuv=1)
    frame = synth.make_frame(9, 96, 128, "luv")
    dev = HipDetector()
    with pytest.raises(HipError):
        dev.chns_compute(frame)                 # no parameters, no model
    dev.set_model(model)
    got = dev.chns_compute(frame)
    assert np.array_equal(bits(got), bits(oracle.chns_compute(model, frame)))
    out = np.zeros(10, np.float32)
    n, hc, wc = C.c_int(), C.c_int(), C.c_int()
    rc = dev.lib.acf_hip_chns_compute(dev.ctx, None, capi.fptr(frame), 96, 128, 3, capi.fptr(out), out.size, C.byref(n), C.byref(hc), C.byref(wc))
    assert rc == capi.E_CAPACITY and (n.value, hc.value, wc.value) == (10, 24, 32)   # sizes still reported
    bad = dict(model)
    bad["softBin"] = 1
    with pytest.raises(HipError):
        dev.chns_compute(frame, bad)
    dev.close()


# ------------------------------------------------------------------ C++ host (CLI)

@pytest.fixture(scope="module")
def cli():
    subprocess.check_call(["make", "-s", "-C", HOST])
    return CLI


def run(cli, args):
    e = dict(os.environ)
    e["ACF_HIP_LIBRARY"] = capi.LIB_PATH
    p = subprocess.run([cli] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert p.returncode == 0, p.stderr
    return p


def _fnv1a(b):
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.gpu
@pytest.mark.parametrize("logger", [False, True])
def test_cli_static_chns_compute(cli, oracle, tmp_path, logger):
    """acf::HipDetector::chnsCompute (static): Channels {data per type, info names, padWith}; with a MatLoggerType the stage-by-stage
    path reports chnsCompute's planes and returns the same floats."""
    H, W = 98, 133   # cropped to 96 x 132
    model = synth.make_model(seed=3, name="INRIA", nTrees=16)
    frames = [synth.make_frame(50 + i, H, W, "rgb") for i in range(2)]
    write_model(str(tmp_path / "m.acfm"), model)
    (tmp_path / "f.raw").write_bytes(np.stack(frames).tobytes())
    out = tmp_path / "c.raw"
    p = run(cli, ["--model", str(tmp_path / "m.acfm"), "--frames", str(tmp_path / "f.raw"), "--rows", str(W), "--cols", str(H), "--channels", "3",
                  "--count", "2", "--chns", str(out)] + (["--log-taps"] if logger else []))
    lines = [l for l in p.stdout.splitlines() if l.startswith("chns ")]
    assert lines[0] == "chns 0 types 3 [color channels|3|replicate|24x33] [gradient magnitude|1||24x33] [gradient histogram|6||24x33]"
    got = np.fromfile(str(out), np.float32).reshape(2, 10, 33, 24)
    for f in range(2):
        assert np.array_equal(bits(got[f]), bits(oracle.chns_compute(model, frames[f]))), f
    taps = [l.split()[1] for l in p.stdout.splitlines() if l.startswith("tap ")]
    if logger:
        assert taps[:7] == ["L:96x132", "U:96x132", "V:96x132", "M:96x132", "Mnorm:96x132", "O:96x132", "H:144x33"]
    else:
        assert not taps


@pytest.mark.gpu
def test_cli_compute_channels_defaults(cli, oracle, tmp_path):
    """Detector::computeChannels: the toolbox defaults whatever the model says, fused into one plane stack (ACF.cpp:183-240)."""
    H, W = 96, 128
    model = synth.make_model(seed=3, name="FACE64", nTrees=16)        # a gray model: must not matter
    frame = synth.make_frame(52, H, W, "rgb")
    write_model(str(tmp_path / "m.acfm"), model)
    (tmp_path / "f.raw").write_bytes(frame.tobytes())
    out = tmp_path / "c.raw"
    p = run(cli, ["--model", str(tmp_path / "m.acfm"), "--frames", str(tmp_path / "f.raw"), "--rows", str(W), "--cols", str(H), "--channels", "3",
                  "--count", "1", "--chns", str(out), "--defaults"])
    assert "chns 0 fused 24x320" in p.stdout
    dfl = synth.make_model(seed=3, name="TINY", nTrees=16, isLuv=0)    # default_options = the toolbox defaults
    want = oracle.chns_compute(dfl, frame)
    got = np.fromfile(str(out), np.float32).reshape(10, 32, 24)
    assert np.array_equal(bits(got), bits(want))


@pytest.mark.gpu
def test_cli_reference_arithmetic_and_level_logger(cli, oracle, tmp_path):
    """setReferenceArithmetic(tables): detections == the oracle's table tier; setLogger: one normalised level per scale."""
    H, W = 112, 96
    model = synth.make_model(seed=3, name="INRIA", nTrees=64, cascThr=-1.5)
    frames = [synth.make_frame(40 + i, H, W, "rgb") for i in range(3)]
    write_model(str(tmp_path / "m.acfm"), model)
    (tmp_path / "f.raw").write_bytes(np.stack(frames).tobytes())
    rcp, rsq = oracle.x86_fixture()
    (tmp_path / "t.bin").write_bytes(rcp.tobytes() + rsq.tobytes())
    p = run(cli, ["--model", str(tmp_path / "m.acfm"), "--frames", str(tmp_path / "f.raw"), "--rows", str(W), "--cols", str(H), "--channels", "3",
                  "--count", "3", "--ref-arith", str(tmp_path / "t.bin"), "--log-levels"])
    plan = oracle.Plan(model, H, W, 3)
    oracle.set_x86_tables(rcp, rsq)
    got, cur = [], None
    for line in p.stdout.splitlines():
        t = line.split()
        if t[0] == "frame":
            cur = []
            got.append(cur)
        elif t[0] != "level":
            cur.append((int(t[0]), int(t[1]), int(t[2]), int(t[3]), int(t[5], 16)))
    total = 0
    for f in range(3):
        oracle.set_approx(3)
        try:
            pyr, _, _ = oracle.chns_pyramid(plan, frames[f])
        finally:
            oracle.set_approx(0)
        det, _ = oracle.detect(plan, pyr)
        want = [(int(d["x"]), int(d["y"]), int(d["w"]), int(d["h"]), int(np.float32(d["score"]).view(np.uint32))) for d in det]
        assert got[f] == want, f
        total += len(want)
    assert total > 0
    levels = [l.split() for l in p.stdout.splitlines() if l.startswith("level ")]
    assert len(levels) == 3 * plan.nScales
    assert [l[1] for l in levels[:plan.nScales]] == ["%06d" % i for i in range(plan.nScales)]
    l0 = plan.levels[0]
    assert levels[0][2] == "%dx%d" % (plan.nChns * l0.wP, l0.hP)     # transposed: cols x rows of the canvas


@pytest.mark.gpu
def test_cli_reference_arithmetic_probed_from_this_host(cli, oracle, tmp_path):
    """setReferenceArithmetic(true): the C++ host probes the CPU it runs on (whatever the GPU box has) and the detector then returns
    what the reference would return HERE: the oracle's table tier with tables probed from the same CPU.  A CPU whose rcpps / rsqrtps
    are not table functions must be refused with an exception, not emulated wrongly."""
    H, W = 112, 96
    model = synth.make_model(seed=3, name="INRIA", nTrees=64, cascThr=-1.5)
    frames = [synth.make_frame(40 + i, H, W, "rgb") for i in range(3)]
    write_model(str(tmp_path / "m.acfm"), model)
    (tmp_path / "f.raw").write_bytes(np.stack(frames).tobytes())
    e = dict(os.environ)
    e["ACF_HIP_LIBRARY"] = capi.LIB_PATH
    p = subprocess.run([cli, "--model", str(tmp_path / "m.acfm"), "--frames", str(tmp_path / "f.raw"), "--rows", str(W), "--cols", str(H), "--channels", "3",
                        "--count", "3", "--ref-arith", "host"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    t = oracle.x86_probe()
    table_cpu = False
    if t is not None:
        oracle.set_x86_tables(*t)
        table_cpu = oracle.x86_verify(0, 1 << 22, 1021) == (0, 0) and oracle.x86_verify(0x3f800000, 1 << 23, 7) == (0, 0)
    if not table_cpu:
        assert p.returncode != 0 and "rcpps" in p.stderr, (p.returncode, p.stderr[-300:])
        return
    assert p.returncode == 0, p.stderr[-500:]
    got, cur = [], None
    for line in p.stdout.splitlines():
        w = line.split()
        if w[0] == "frame":
            cur = []
            got.append(cur)
        else:
            cur.append((int(w[0]), int(w[1]), int(w[2]), int(w[3]), int(w[5], 16)))
    plan = oracle.Plan(model, H, W, 3)
    total = 0
    for f in range(3):
        oracle.set_approx(3)
        try:
            pyr, _, _ = oracle.chns_pyramid(plan, frames[f])
        finally:
            oracle.set_approx(0)
        det, _ = oracle.detect(plan, pyr)
        want = [(int(d["x"]), int(d["y"]), int(d["w"]), int(d["h"]), int(np.float32(d["score"]).view(np.uint32))) for d in det]
        assert got[f] == want, f
        total += len(want)
    assert total > 0
