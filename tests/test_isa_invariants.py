"""Invariants of the built code object that hand-counted waits rest on (CPU test: disassembles acf_amd/libacf_hip.so).

k_resample_strip (kernels_ldcf_strip.hip.h) lets the next tile's LDS-DMA requests stay in flight behind a step's y pass with a COUNTED
wait, `s_waitcnt vmcnt(RS_ITEMS * (HAVEB ? 2 : 1))`: correct only if exactly that many buffer stores are issued per wave and step after
the requests (requests complete in order).  A toolchain that merged, split or dropped stores would let a tile be read before it has
arrived, and only the bit-exact GPU tests would notice.  Here the number is read off the ISA: every instantiation holds exactly
RS_ITEMS * (HAVEB ? 2 : 1) `buffer_store_dword` instructions and a wait with that count."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "acf_amd", "libacf_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not os.path.exists(OBJDUMP) or not os.path.exists(LIB):
        pytest.skip("llvm-objdump or the built library is missing")
    d = tmp_path_factory.mktemp("isa")
    lib = os.path.join(str(d), "lib.so")
    shutil.copy(LIB, lib)
    subprocess.run([OBJDUMP, "--offloading", lib], cwd=str(d), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    co = [f for f in os.listdir(str(d)) if "gfx950" in f]
    assert len(co) == 1, os.listdir(str(d))
    out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", os.path.join(str(d), co[0])], check=True, stdout=subprocess.PIPE, universal_newlines=True).stdout
    kernels = {}
    cur = None
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
        elif cur is not None:
            kernels[cur].append(line)
    return kernels


def test_strip_march_store_count_matches_its_counted_wait(isa):
    src = open(os.path.join(ROOT, "acf_amd", "csrc", "kernels_ldcf_strip.hip.h")).read()
    items = int(re.search(r"constexpr int RS_ITEMS = (\d+);", src).group(1))
    assert re.search(r'"n"\(RS_ITEMS \* \(HAVEB \? 2 : 1\)\)', src), "the counted wait's operand changed: update this test with it"
    seen = 0
    for name, body in isa.items():
        m = re.match(r"_ZN6acfhip16k_resample_stripILb([01])ELb([01])EEEvNS_9StripArgsE$", name)
        if not m:
            continue
        seen += 1
        want = items * (2 if m.group(1) == "1" else 1)
        text = "\n".join(body)
        stores = len(re.findall(r"\bbuffer_store_dword\b", text))
        assert stores == want, (name, stores, want)
        assert re.search(r"s_waitcnt vmcnt\(%d\)" % want, text), (name, want)
        assert not re.search(r"\bglobal_store|\bflat_store", text), name   # (stores of another kind do not complete in order with the requests)
    assert seen == 4
