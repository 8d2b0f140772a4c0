#!/usr/bin/env python
"""T-ref against T-exact, END TO END (build container only: needs oracle/_ref, i.e. /root/reference).

    python tests/golden/make_tref.py [--frames N]

The oracle (and the HIP path, which equals it bit for bit) computes `1/sqrt` and `1/x` exactly where the reference's SSE
kernels use `_mm_rsqrt_ps` / `_mm_rcp_ps` (T/gradientMex.cpp:209-219,266; T/rgbConvertMex.cpp:161; T/sse.hpp:185-192):
the "T-exact" tier.  tests/test_oracle_vs_ref.py bounds the difference per stage.  This script measures what that
difference does to DETECTIONS: the same restated orchestration (oracle/acf_oracle.c: chnsPyramid, chnsCompute, acfDetect,
box mapping, bbNms + prune) is run twice on the same frames and models, once on the T-exact kernels and once on the
reference's OWN compiled kernels (oracle/_ref/libacfref.so through acfo_set_ref_kernels: convTri1, convTri, gradMag,
gradMagNorm, gradHist, resample, rgbConvert), for BASELINE.json's cfg 1 / 2 / 4 shapes at full size.

Outputs (both data only):
  tests/golden/tref_study.npz    per configuration and frame: the cascade's hits {level, c, r, score} of BOTH tiers
                                 (the T-ref ones are the bits of THIS host's rsqrtps / rcpps), the frame / model seeds.
  profiles/r05_tref_study.json   the table DESIGN.md section 2 quotes.
"""
import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from acf_amd import capi, synth  # noqa: E402
from oracle import binding as ob  # noqa: E402

# name -> (H, W, frame kind, d_in, model preset): BASELINE.json configs 1, 2, 4 as tests/test_gpu_configs.py builds them
CFG = {
    "cfg1_vga_gray_face64": (480, 640, "gray", 1, "FACE64"),
    "cfg2_1080p_luv_face80": (1080, 1920, "luv", 3, "FACE80"),
    "cfg4_vga_rgb_inria": (480, 640, "rgb", 3, "INRIA"),
}
FRAME_SEED0 = 500
MODEL_SEED = 1
NMS = dict(type="maxg", overlap=0.65, ovrDnm="min", prune=True, maxCount=10, pruneRatio=0.0)  # Detector::operator()'s defaults


def run_tier(plan, frame, tref, approx=0):
    """tref: the reference's compiled kernels; approx 1 / 2: the oracle with its three rsqrt / rcp sites as 12-bit approximations
    (rounded / truncated: acfo_set_approx) — two more implementations inside _mm_rsqrt_ps's documented error bound."""
    ob.set_tref(tref)
    ob.set_approx(approx)
    try:
        pyr, _, _ = ob.chns_pyramid(plan, frame)
    finally:
        ob.set_tref(False)
        ob.set_approx(0)
    det, hits = ob.detect(plan, pyr)
    return pyr, det, hits


def hit_key(h):
    return (h["scale"].astype(np.int64) << 40) | (h["c"].astype(np.int64) << 20) | h["r"].astype(np.int64)


def compare_hits(he, hr):
    """he: T-exact hits, hr: T-ref hits -> (#common, #only exact, #only ref, max |dscore| over common)."""
    ke, kr = hit_key(he), hit_key(hr)
    common, ie, ir = np.intersect1d(ke, kr, return_indices=True)
    d = np.abs(he["score"][ie].astype(np.float64) - hr["score"][ir].astype(np.float64))
    return len(common), len(ke) - len(common), len(kr) - len(common), float(d.max()) if len(d) else 0.0, d


def final_boxes(det, nms):
    keep = ob.nms(np.stack([det["x"], det["y"], det["w"], det["h"]], axis=1), det["score"].astype(np.float64), nms)
    return det[keep]


def study(nframes):
    assert ob.have_ref(), "oracle/_ref/libacfref.so missing: run `make -C oracle` with /root/reference present"
    nms = capi.make_nms(**NMS)
    table, store = {}, {}
    for name, (H, W, kind, d_in, preset) in CFG.items():
        model = synth.make_model(seed=MODEL_SEED, name=preset)
        plan = ob.Plan(model, H, W, d_in)
        rows = []
        alld = []
        yard = {}
        for f in range(nframes):
            frame = synth.make_frame(FRAME_SEED0 + f, H, W, kind)
            pe, de, he = run_tier(plan, frame, False)
            pr, dr, hr = run_tier(plan, frame, True)
            _, _, h1 = run_tier(plan, frame, False, 1)
            _, _, h2 = run_tier(plan, frame, False, 2)
            for pn, (ha_, hb_) in (("approx_round_vs_ref", (h1, hr)), ("approx_trunc_vs_ref", (h2, hr)), ("approx_round_vs_approx_trunc", (h1, h2)),
                                   ("exact_vs_ref", (he, hr))):
                nc_, oa_, ob_, _, d_ = compare_hits(ha_, hb_)
                t_ = yard.setdefault(pn, dict(common=0, only_first=0, only_second=0, gt=0, dmax=0.0))
                t_["common"] += nc_
                t_["only_first"] += oa_
                t_["only_second"] += ob_
                t_["gt"] += int((d_ > 1e-4).sum())
                t_["dmax"] = max(t_["dmax"], float(d_.max()) if len(d_) else 0.0)
            nc, oe, orr, dmax, d = compare_hits(he, hr)
            alld.append(d)
            fe, fr = final_boxes(de, nms), final_boxes(dr, nms)
            same_boxes = sum(1 for a in fe if any((a["x"], a["y"], a["w"], a["h"]) == (b["x"], b["y"], b["w"], b["h"]) for b in fr))
            rel = np.abs(pe - pr) / np.maximum(np.abs(pe), 1e-3)
            rows.append(dict(frame_seed=FRAME_SEED0 + f, hits_exact=len(he), hits_ref=len(hr), common=nc, only_exact=oe, only_ref=orr,
                             max_abs_dscore_common=dmax, pyramid_cells_differing_frac=float((pe != pr).mean()),
                             pyramid_max_abs_diff=float(np.abs(pe - pr).max()), pyramid_max_rel_diff=float(rel.max()),
                             final_exact=len(fe), final_ref=len(fr), final_same_box=same_boxes))
            store["%s_f%d_hits_exact" % (name, f)] = he
            store["%s_f%d_hits_ref" % (name, f)] = hr
            print(name, rows[-1], flush=True)
        alld = np.concatenate(alld) if alld else np.zeros(0)
        tot = {k: int(sum(r[k] for r in rows)) for k in ("hits_exact", "hits_ref", "common", "only_exact", "only_ref", "final_exact", "final_ref", "final_same_box")}
        tot["max_abs_dscore_common"] = max(r["max_abs_dscore_common"] for r in rows)
        tot["common_with_dscore_gt_1e-4"] = int((alld > 1e-4).sum())
        tot["median_abs_dscore_common"] = float(np.median(alld)) if len(alld) else 0.0
        tot["p99_abs_dscore_common"] = float(np.quantile(alld, 0.99)) if len(alld) else 0.0
        tot["pyramid_max_abs_diff"] = max(r["pyramid_max_abs_diff"] for r in rows)
        # the yardstick: the same comparison between pairs of CONFORMING approximations (the reference's own kernels on this host, and
        # the oracle's arithmetic with 12-bit rsqrt / rcp results, rounded or truncated): what one conforming CPU differs from another by
        tot["yardstick_pairs"] = {k: dict(common=v["common"], only_first=v["only_first"], only_second=v["only_second"],
                                          frac_common_gt_1e4=round(v["gt"] / max(v["common"], 1), 3), max_abs_dscore=round(v["dmax"], 4)) for k, v in yard.items()}
        table[name] = dict(frames=nframes, totals=tot, per_frame=rows)
        store[name + "_meta"] = np.asarray([H, W, d_in, nframes, FRAME_SEED0, MODEL_SEED], np.int64)
    return table, store


SMALL = {
    # name -> (H, W, kind, d_in, model kwargs): whole T-ref PYRAMIDS at sizes small enough to commit ("every scale of the fused
    # pyramid, both T-ref bytes and T-exact bytes"; the T-exact ones are tests/golden/pipeline_*.npz)
    "luv_tiny_160x120": (120, 160, "luv", 3, dict(name="TINY", nTrees=64, cascThr=-3.0)),
    "rgb_inria_160x120": (120, 160, "rgb", 3, dict(name="INRIA", nTrees=64, cascThr=-1.5)),
    "gray_face64_320x240": (240, 320, "gray", 1, dict(name="FACE64", nTrees=96, cascThr=-1.0)),
}


def small_pyramids():
    """tests/golden/tref_pyramids.npz: frame + model seeds, the fused pyramid and the hits computed with the reference's OWN compiled
    kernels (this host's rsqrtps / rcpps) under the restated orchestration.  What the table tier (oracle: acfo_set_approx(3) with
    tests/golden/x86_rcp_rsqrt.npz; device: option arith) must reproduce bit for bit, on any box."""
    assert ob.have_ref()
    store = {}
    for name, (H, W, kind, d_in, kw) in SMALL.items():
        model = synth.make_model(seed=3, **kw)
        frame = synth.make_frame(77, H, W, kind)
        plan = ob.Plan(model, H, W, d_in)
        pr, dr, hr = run_tier(plan, frame, True)
        pe, de, he = run_tier(plan, frame, False)
        assert not np.array_equal(pr, pe)
        store[name + "_pyramid_ref"] = pr
        store[name + "_hits_ref"] = hr
        store[name + "_det_ref"] = dr
        store[name + "_meta"] = np.asarray([H, W, d_in, 77, 3], np.int64)
        print(name, pr.size, "floats,", len(hr), "hits (T-exact:", len(he), "), cells differing from T-exact: %.1f %%" % (100 * float((pr != pe).mean())))
    np.savez_compressed(os.path.join(HERE, "tref_pyramids.npz"), **store)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--small-only", action="store_true", help="only tests/golden/tref_pyramids.npz (the small whole-pyramid fixtures)")
    a = ap.parse_args()
    small_pyramids()
    if a.small_only:
        sys.exit(0)
    table, store = study(a.frames)
    import platform
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        cpu = platform.processor()
    out = dict(what="T-ref (the reference's own compiled SSE kernels under the restated orchestration) against T-exact (the oracle = the HIP path, "
                    "bit for bit), end to end; hits = windows that pass the cascade (acfDetect1), final = after bbNms maxg .65/min + prune(10)",
               host_cpu=cpu, nms=NMS, configs=table)
    np.savez_compressed(os.path.join(HERE, "tref_study.npz"), **store)
    with open(os.path.join(ROOT, "profiles", "r05_tref_study.json"), "w") as f:
        json.dump(out, f, indent=1)
    for name, t in table.items():
        print(name, json.dumps(t["totals"]))
