"""CPU-only checks of the boundary: the C-ABI library loads, exports every
symbol include/acf_hip.h declares, fails loudly without a GPU, and its host
planning (getScales, level geometry) agrees with the oracle's independent
restatement.  No device compute is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from acf_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "acf_hip.h")).read()
    declared = sorted(set(re.findall(r"ACF_HIP_API\s+[\w\s\*]+?\b(acf_hip_\w+)\s*\(", hdr)))
    assert len(declared) >= 28
    lib = capi.load()
    for name in declared:
        assert hasattr(lib, name), name
    # the ctypes binding covers the whole header
    assert sorted(capi.DECLARED_SYMBOLS) == declared
    assert lib.acf_hip_abi_version() == 9


def test_every_option_the_library_accepts_is_documented_in_the_header():
    """acf_hip_set_option's keys (the strcmp chain of acf_hip.hip) against include/acf_hip.h: an option without a description
    there is an undocumented part of the boundary."""
    import re
    src = open(os.path.join(ROOT, "acf_amd", "csrc", "acf_hip.hip")).read()
    body = src[src.index("int acf_hip_set_option"):]
    body = body[:body.index("unknown option")]
    keys = sorted(set(re.findall(r'strcmp\(key, "([a-z_0-9]+)"\)', body)))
    hdr = open(os.path.join(ROOT, "include", "acf_hip.h")).read()
    assert len(keys) >= 20
    assert [k for k in keys if '"%s"' % k not in hdr] == []


def test_no_silent_cpu_fallback():
    """Without a GPU the product must refuse, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = capi.load()
    ctx = C.c_void_p()
    assert lib.acf_hip_create(0, None, C.byref(ctx)) == 6  # ACF_HIP_E_NODEVICE
    from acf_amd.detector import HipDetector, HipError
    with pytest.raises(HipError):
        HipDetector(synth.make_model(name="TINY", nTrees=4), 64, 64)


def test_product_does_not_reference_oracle():
    """Nothing under acf_amd/ may import, include or link the oracle."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "acf_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", ".map")) or f == "Makefile":
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"(from|import)\s+oracle|oracle/|liboracle|acfo_", txt):
                    bad.append(os.path.join(base, f))
    assert not bad, bad


SIZES = [(1080, 1920), (480, 640), (2160, 3840), (720, 1280), (130, 175), (96, 128), (175, 130), (333, 517), (64, 48)]


@pytest.mark.parametrize("H,W", SIZES)
@pytest.mark.parametrize("nPerOct,nOctUp,minDs", [(8, 0, (80, 80)), (8, 1, (100, 41)), (12, 0, (80, 80)), (4, 0, (16, 16)), (1, 0, (32, 32))])
def test_get_scales_matches_oracle(oracle, H, W, nPerOct, nOctUp, minDs):
    lib = capi.load()
    cap = 512
    a = [(C.c_double * cap)() for _ in range(3)]
    n = C.c_int()
    assert lib.acf_hip_get_scales(nPerOct, nOctUp, minDs[0], minDs[1], 4, H, W, a[0], a[1], a[2], cap, C.byref(n)) == 0
    b = [(C.c_double * cap)() for _ in range(3)]
    m = oracle.lib().acfo_get_scales(nPerOct, nOctUp, minDs[0], minDs[1], 4, H, W, b[0], b[1], b[2], cap)
    assert n.value == m
    for k in range(3):
        assert list(a[k])[:m] == list(b[k])[:m]  # exact doubles


def test_get_scales_known_values():
    """The numbers SURVEY.md §8d records for the headline configuration."""
    lib = capi.load()
    cap = 128
    s, sh, sw = (C.c_double * cap)(), (C.c_double * cap)(), (C.c_double * cap)()
    n = C.c_int()
    lib.acf_hip_get_scales(8, 0, 80, 80, 4, 1080, 1920, s, sh, sw, cap, C.byref(n))
    assert n.value == 31
    assert s[0] == 1.0 and s[8] == 0.5
    assert abs(s[16] - 0.252) < 1e-3 and abs(s[24] - 0.12533) < 1e-4 and abs(s[30] - 0.0747) < 1e-4
    lib.acf_hip_get_scales(12, 0, 80, 80, 4, 2160, 3840, s, sh, sw, cap, C.byref(n))
    assert n.value == 58
    lib.acf_hip_get_scales(8, 0, 64, 64, 4, 480, 640, s, sh, sw, cap, C.byref(n))
    assert n.value == 24


@pytest.mark.parametrize("H,W,name,d", [(1080, 1920, "FACE80", 3), (480, 640, "FACE64", 1), (480, 640, "INRIA", 3), (130, 175, "TINY", 3)])
def test_plan_levels_match_oracle(oracle, H, W, name, d):
    model = synth.make_model(name=name, nTrees=4)
    params, keep = capi.make_params(model)
    lib = capi.load()
    lv = (capi.Level * 256)()
    n, nc = C.c_int(), C.c_int()
    assert lib.acf_hip_plan_levels(C.byref(params), H, W, d, lv, 256, C.byref(n), C.byref(nc)) == 0
    plan = oracle.Plan(model, H, W, d)
    assert n.value == plan.nScales and nc.value == plan.nChns
    for i in range(n.value):
        for f, _ in capi.Level._fields_:
            assert getattr(lv[i], f) == getattr(plan.levels[i], f), (i, f)


def test_headline_plan_numbers(oracle):
    """31 levels, 4 real, 662,799 windows, 32.47 MB pyramid, 89.8 MB algorithmic bytes per frame."""
    plan = oracle.Plan(synth.make_model(name="FACE80", nTrees=4), 1080, 1920, 3)
    assert plan.nScales == 31 and plan.real == [0, 8, 16, 24]
    assert [(plan.levels[i].hC * 4, plan.levels[i].wC * 4) for i in plan.real] == [(1080, 1920), (540, 960), (272, 484), (136, 240)]
    assert sum(plan.levels[i].nWinR * plan.levels[i].nWinC for i in range(31)) == 662799
    assert abs(plan.total * 4 / 1e6 - 32.47) < 0.01
    assert abs((3 * 4 * 1080 * 1920 + 2 * 4 * plan.total) / 1e6 - 89.8) < 0.1


def test_plan_rejects_unsupported():
    lib = capi.load()
    lv = (capi.Level * 8)()
    n, nc = C.c_int(), C.c_int()
    # odd softBin = trilinear HOG binning (not built); an unknown colour space; hsv on planes declared LUV (rgbConvert.cpp:150-155: CV_Assert)
    for over, code in ((dict(softBin=1), 2), (dict(softBin=-1), 2), (dict(shrink=3, modelDsPad_h=15, modelDsPad_w=15), 2), (dict(lambdas=[0.1, 0.1]), 1),
                       (dict(colorSpace=7), 2), (dict(colorSpace=capi.CS_HSV, isLuv=1), 1), (dict(colorChn=5), 1)):
        params, keep = capi.make_params(synth.make_model(name="TINY", nTrees=4, **over))
        assert lib.acf_hip_plan_levels(C.byref(params), 96, 128, 3, lv, 8, C.byref(n), C.byref(nc)) == code, over
    for over in (dict(softBin=2), dict(softBin=-2), dict(colorSpace=capi.CS_HSV, isLuv=0)):  # built since round 4
        params, keep = capi.make_params(synth.make_model(name="TINY", nTrees=4, **over))
        assert lib.acf_hip_plan_levels(C.byref(params), 96, 128, 3, lv, 8, C.byref(n), C.byref(nc)) == 0, over
    params, keep = capi.make_params(synth.make_model(name="FACE80", nTrees=4))
    assert lib.acf_hip_plan_levels(C.byref(params), 40, 40, 3, lv, 8, C.byref(n), C.byref(nc)) == 1  # smaller than minDs: no scales


def test_thrs_u8_host_entry_matches_oracle(oracle):
    """acf_hip_thrs_u8 (host only) vs the oracle's restatement of thrs.convertTo(thrsU8, CV_8UC1, 255.0f)
    (ACFIOArchive.h:96-99): halves round to even, values saturate, NaN -> 0."""
    lib = capi.load()
    halves = (np.arange(0, 256, dtype=np.float32) + np.float32(0.5)) / np.float32(255.0)
    t = np.concatenate([synth.uniform(3, 4000, 0).astype(np.float32) * 1.4 - 0.2, halves,
                        np.array([0.0, 1.0, -1.0, 2.0, 1e30, -1e30, np.nan, 0.5 / 255, 1.5 / 255, 2.5 / 255], np.float32)]).astype(np.float32)
    got = np.zeros(t.size, np.uint8)
    assert lib.acf_hip_thrs_u8(capi.fptr(t), t.size, got.ctypes.data) == 0
    want = oracle.thrs_u8(t)
    assert np.array_equal(got, want)
    assert want[-4] == 0 and want[-5] == 0 and want[-6] == 0 and want[-7] == 255  # NaN, -1e30, 1e30 (cvtss2si overflow -> INT_MIN -> 0), 2.0
    assert list(oracle.thrs_u8(np.array([0.5, 1.5, 2.5, 3.5], np.float32))) == [128, 255, 255, 255]
