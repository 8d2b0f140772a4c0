#!/usr/bin/env python
"""Build profiles/rNN_pmc_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

usage: make_traffic_json.py <fetch counter_collection.csv> <write counter_collection.csv> <frames_per_step> > profiles/r04_pmc_traffic.json
Bytes per launch = FETCH_SIZE*f + WRITE_SIZE (KB -> bytes); f = 2 for kernels whose reads are 16 B/lane streams (gfx950's
FETCH_SIZE counts 64 B per 128-B request, MI355X_MICROARCH.md §HBM), 1 otherwise (uncalibrated).  Launches of one bench
"kernel" (prof name) are summed: k_level(fused) = all k_level<R,mode> launches of a step, etc.
"""
import csv, json, sys, collections

# (k_triy_chns<.., true> — M, O, U in blocks, four 16-byte loads per lane and plane — was missing from this list until the end of round 5: its reads
# were reported at half their size, 17 MB per 1080p frame; totals of earlier files are that much too small)
WIDE = ("k_cascade_tile", "k_cascade_tail3", "k_grad_mag_vec", "k_tri_x5v", "k_tri_y5(", "k_chns", "k_resample_half", "k_resample_strip", "k_smooth_vec", "k_smooth_grad", "k_tail_scan",
        "k_triy_chns<6, true>", "k_triy_chns<12, true>")
GROUP = [("k_cascade_tile", "k_cascade_tile"), ("k_cascade_tail", "k_cascade_tail3"), ("k_tail_scan", "k_tail_scan"), ("k_level", "k_level(fused)"),
         ("k_triy_chns", "k_triy_chns"), ("k_sort_map", "k_sort_map"), ("k_nms", "k_nms"), ("k_export", "k_export"),
         ("k_chns", "k_chns"), ("k_smooth_vec", "k_smooth_vec"), ("k_smooth_grad_tri", "k_smooth_grad_tri"), ("k_smooth_grad", "k_smooth_grad"), ("k_smooth_tri1", "k_smooth_tri1(image)"), ("k_grad_mag", "k_grad_mag"), ("k_tri_x", "k_tri_x"),
         ("k_tri_y", "k_tri_y"), ("k_resample", "k_resample(image)")]


def per_dispatch(path, counter):
    d = collections.defaultdict(float)
    name = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[r["Dispatch_Id"]] += float(r["Counter_Value"])
            name[r["Dispatch_Id"]] = r["Kernel_Name"]
    return d, name


def main():
    fetch, fn = per_dispatch(sys.argv[1], "FETCH_SIZE")
    write, wn = per_dispatch(sys.argv[2], "WRITE_SIZE")
    frames = int(sys.argv[3])

    def agg(vals, names, scale_wide):
        tot = collections.defaultdict(float)
        cnt = collections.Counter()
        for k, v in vals.items():
            kn = names[k]
            for pat, g in GROUP:
                if pat in kn:
                    f = 2.0 if (scale_wide and any(w in kn for w in WIDE)) else 1.0
                    tot[g] += v * f * 1024.0
                    cnt[g] += 1
                    break
        return tot, cnt
    ft, fc = agg(fetch, fn, True)
    wt, wc = agg(write, wn, False)
    # number of bench steps seen = launches of the tile kernel
    steps = max(fc.get("k_cascade_tile", 1), 1)
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, profiles/run_pmc.sh), MI355X",
           "note": __doc__.split("\n\n")[1].strip().replace("\n", " "),
           "frames_per_step": frames,
           "kernels": {g: int((ft.get(g, 0) + wt.get(g, 0)) / steps) for g in sorted(set(ft) | set(wt))}}
    out["MB_per_frame"] = {g: round(v / frames / 1e6, 2) for g, v in out["kernels"].items()}
    out["fetch_MB_per_frame"] = {g: round(v / steps / frames / 1e6, 2) for g, v in sorted(ft.items())}  # (after the wide-load correction)
    out["write_MB_per_frame"] = {g: round(v / steps / frames / 1e6, 2) for g, v in sorted(wt.items())}
    out["total_MB_per_frame"] = round(sum(out["MB_per_frame"].values()), 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
