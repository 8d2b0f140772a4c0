#!/usr/bin/env python3
"""Loop bodies of a kernel in hipcc's device assembly: instruction, load, store, LDS, branch and barrier counts and the
`s_waitcnt vmcnt(N)` immediates in program order.  A column-sequential kernel whose loads are really in flight across steps
shows large N (the wave waits until at most N younger memory operations are outstanding); N near 0 in a loop with many
branches means every step waits for a full memory round trip (MI355X: completion is in order).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize --offload-device-only -S \
          -x hip acf_amd/csrc/acf_hip.hip -o /tmp/k.s
    python profiles/isa_loops.py /tmp/k.s k_level_all [min_instructions]
"""
import re
import sys


def main():
    txt = open(sys.argv[1]).read().split("\n")
    name = sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    starts = [i for i, l in enumerate(txt) if re.match(r"^_ZN6acfhip\d+" + name + r".*:", l)]
    for S in starts:
        E = next(i for i in range(S + 1, len(txt)) if ".amdhsa_kernel" in txt[i] or txt[i].startswith(".Lfunc_end"))
        lines = txt[S:E]
        print("==", txt[S].split(":")[0][:100])
        labels = {}
        for i, l in enumerate(lines):
            m = re.match(r"^(\.LBB[0-9_]+):", l)
            if m:
                labels[m.group(1)] = i
        seen = set()
        for i, l in enumerate(lines):
            m = re.search(r"s_cbranch_\w+\s+(\.LBB[0-9_]+)", l) or re.search(r"s_branch\s+(\.LBB[0-9_]+)", l)
            if not (m and m.group(1) in labels and labels[m.group(1)] < i):
                continue
            a = labels[m.group(1)]
            if a in seen:
                continue  # nested back edges to the same header: report the innermost only
            seen.add(a)
            body = lines[a:i + 1]
            n = len([x for x in body if x.startswith("\t") and not x.strip().startswith((";", "."))])
            if n < min_n:
                continue
            vm = [re.search(r"vmcnt\((\d+)\)", x).group(1) for x in body if "vmcnt" in x]
            dma = len([x for x in body if re.search(r"(global|buffer)_load", x) and " lds" in x])
            ld = len([x for x in body if re.search(r"(global|buffer)_load", x)]) - dma
            st = len([x for x in body if re.search(r"(global|buffer)_store", x)])
            ds = len([x for x in body if "\tds_" in x])
            br = len([x for x in body if "s_cbranch" in x])
            bar = len([x for x in body if "s_barrier" in x])
            print("  loop @%d: %d instructions, %d loads, %d lds-dma, %d stores, %d ds, %d branches, %d barriers; vmcnt waits: %s"
                  % (a, n, ld, dma, st, ds, br, bar, " ".join(vm)))


if __name__ == "__main__":
    main()
