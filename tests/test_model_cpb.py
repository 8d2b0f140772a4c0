"""`.cpb` model files (acf::Detector through cereal's PortableBinary archive).

No sample file ships with the reference, so the layout (acf_amd/host/ModelIO.h) is pinned the only way
available: two independent restatements — C++ (acf_amd/host/ModelIO.cpp, one schema for both directions)
and Python (acf_amd/modelio.py, explicit byte stream) — must produce byte-identical files and read each
other's output; the GPU test then runs detection from a `.cpb` model.
"""
import os
import struct
import subprocess

import numpy as np
import pytest

from acf_amd import capi, synth
from acf_amd.modelio import read_cpb, write_cpb, write_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "acf_amd", "host")
CLI = os.path.join(HOST, "acf_hip_detect")


@pytest.fixture(scope="module")
def cli():
    subprocess.check_call(["make", "-s", "-C", HOST])
    return CLI


MODELS = [dict(name="INRIA", nTrees=64), dict(name="FACE64", nTrees=32), dict(name="TINY", nTrees=16, treeDepth=0),
          dict(name="FACE80", nTrees=8, treeDepth=3, nPerOct=12, nApprox=11)]


@pytest.mark.parametrize("kw", MODELS, ids=[m["name"] + str(m["nTrees"]) for m in MODELS])
def test_two_restatements_agree_byte_for_byte(cli, tmp_path, kw):
    m = synth.make_model(seed=5, **kw)
    write_model(str(tmp_path / "m.acfm"), m)
    write_cpb(str(tmp_path / "py.cpb"), m)
    subprocess.check_call([cli, "--convert", str(tmp_path / "m.acfm"), "--out", str(tmp_path / "cc.cpb")])
    py, cc = (tmp_path / "py.cpb").read_bytes(), (tmp_path / "cc.cpb").read_bytes()
    assert py == cc
    # the C++ reader on the Python file, written back
    subprocess.check_call([cli, "--convert", str(tmp_path / "py.cpb"), "--out", str(tmp_path / "cc2.cpb")])
    assert (tmp_path / "cc2.cpb").read_bytes() == py
    # the Python reader on the C++ file
    got, nms, cal = read_cpb(str(tmp_path / "cc.cpb"))
    for k, v in got.items():
        assert np.array_equal(np.asarray(v), np.asarray(m[k])), k
    assert nms == ("maxg", 0.65, "min") and cal == 0.0


def test_layout_landmarks(tmp_path):
    """The first bytes follow cereal's rules: endian flag, Detector version 1 (CEREAL_CLASS_VERSION,
    ACFIOArchiveCereal.cpp:7), Classifier version 0, cv::Mat version 0 written once, then rows/cols/type/continuous."""
    m = synth.make_model(seed=5, name="TINY", nTrees=16)
    write_cpb(str(tmp_path / "m.cpb"), m)
    b = (tmp_path / "m.cpb").read_bytes()
    assert b[0] == 1
    assert struct.unpack_from("<III", b, 1) == (1, 0, 0)
    assert struct.unpack_from("<iiiB", b, 13) == (16, 7, 4, 1)           # fids: [nTrees][nTreeNodes] CV_32S continuous
    off = 13 + 13 + 16 * 7 * 4
    assert struct.unpack_from("<iiiB", b, off) == (16, 7, 5, 1)          # thrs follows WITHOUT another version word
    assert np.array_equal(np.frombuffer(b, "<f4", 16 * 7, off + 13).reshape(16, 7), m["thrs"])


def test_reader_rejects_malformed(cli, tmp_path):
    m = synth.make_model(seed=5, name="TINY", nTrees=16)
    write_cpb(str(tmp_path / "m.cpb"), m)
    b = (tmp_path / "m.cpb").read_bytes()
    cases = {"trunc.cpb": b[:len(b) // 2], "flag.cpb": bytes([7]) + b[1:], "type.cpb": b[:13 + 8] + struct.pack("<i", 6) + b[13 + 12:],
             "empty.cpb": b""}
    for name, data in cases.items():
        (tmp_path / name).write_bytes(data)
        p = subprocess.run([cli, "--convert", str(tmp_path / name), "--out", str(tmp_path / "x.cpb")], stderr=subprocess.PIPE, universal_newlines=True)
        assert p.returncode != 0, name


def test_absent_field_keeps_default(cli, tmp_path):
    """Field<T>::has == false (ACFField.h:123-130): the stored value is not taken."""
    m = synth.make_model(seed=5, name="TINY", nTrees=16)
    write_cpb(str(tmp_path / "m.cpb"), m)
    b = bytearray((tmp_path / "m.cpb").read_bytes())
    # find Field<int> "stride": value i32, then name "stride", has, isLeaf
    key = struct.pack("<Q", 6) + b"stride"
    at = b.index(key)
    b[at - 4:at] = struct.pack("<i", 77)   # a value that must be ignored ...
    b[at + len(key)] = 0                   # ... because has = false
    (tmp_path / "nohas.cpb").write_bytes(bytes(b))
    subprocess.check_call([cli, "--convert", str(tmp_path / "nohas.cpb"), "--out", str(tmp_path / "out.cpb")])
    got, _, _ = read_cpb(str(tmp_path / "out.cpb"))
    assert got["stride"] == 4  # HipDetector::Options default, not 77


@pytest.mark.gpu
def test_detect_from_cpb_model(cli, oracle, tmp_path):
    H, W = 96, 128
    model = synth.make_model(seed=3, name="TINY", nTrees=96)
    frames = [synth.make_frame(40 + i, H, W, "luv") for i in range(2)]
    write_cpb(str(tmp_path / "m.cpb"), model)
    (tmp_path / "f.raw").write_bytes(np.stack(frames).tobytes())
    e = dict(os.environ)
    e["ACF_HIP_LIBRARY"] = capi.LIB_PATH
    p = subprocess.run([cli, "--model", str(tmp_path / "m.cpb"), "--frames", str(tmp_path / "f.raw"), "--rows", str(W), "--cols", str(H),
                        "--channels", "3", "--count", "2", "--luv"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    assert p.returncode == 0, p.stderr
    plan = oracle.Plan(model, H, W, 3)
    lines = [l.split() for l in p.stdout.strip().splitlines()]
    f = -1
    got = [[], []]
    for t in lines:
        if t[0] == "frame":
            f += 1
        else:
            got[f].append((int(t[0]), int(t[1]), int(t[2]), int(t[3]), int(t[5], 16)))
    for f in range(2):
        pyr, _, _ = oracle.chns_pyramid(plan, frames[f])
        det, _ = oracle.detect(plan, pyr)
        want = [(int(d["x"]), int(d["y"]), int(d["w"]), int(d["h"]), int(np.float32(d["score"]).view(np.uint32))) for d in det]
        assert len(want) > 0 and got[f] == want
