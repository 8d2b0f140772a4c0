"""Seeded synthetic frames and models for parity tests and bench.py.

No model file or test image of the reference is available, so
every configuration runs on synthetic data of the named shape.  Everything is
derived from a counter-based splitmix64 generator written out here with numpy
integer ops and IEEE +,-,*,/ only, so this container and the GPU box produce
identical bytes.
"""
import numpy as np

from . import capi

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n, stream=0):
    """n uint64 values: splitmix64 of counters 1..n on (seed, stream)."""
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + np.uint64(stream) * np.uint64(0xA0761D6478BD642F)
        x = base + (np.arange(1, n + 1, dtype=np.uint64)) * np.uint64(0x9E3779B97F4A7C15)
        z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(seed, n, stream=0):
    """n float64 in [0,1) with 24 random bits (exact in float32)."""
    return (splitmix64(seed, n, stream) >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def _upsample_bilinear(g, h, w):
    """Bilinear upsample of grid g (gh, gw) to (h, w), float64, fixed op order."""
    gh, gw = g.shape
    ys = (np.arange(h, dtype=np.float64) + 0.5) * gh / h - 0.5
    xs = (np.arange(w, dtype=np.float64) + 0.5) * gw / w - 0.5
    y0 = np.clip(np.floor(ys), 0, gh - 1).astype(np.int64)
    x0 = np.clip(np.floor(xs), 0, gw - 1).astype(np.int64)
    y1 = np.minimum(y0 + 1, gh - 1)
    x1 = np.minimum(x0 + 1, gw - 1)
    fy = np.clip(ys - y0, 0.0, 1.0)[:, None]
    fx = np.clip(xs - x0, 0.0, 1.0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x1]
    c = g[y1][:, x0]
    d = g[y1][:, x1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def band_noise(seed, h, w, stream=0, octaves=(64, 32, 16, 8, 4, 2)):
    """Band-limited noise in [0,1]: octave cells of the given pixel sizes, amplitude ~ cell size."""
    acc = np.zeros((h, w), dtype=np.float64)
    tot = 0.0
    for k, cell in enumerate(octaves):
        gh, gw = max(2, h // cell + 1), max(2, w // cell + 1)
        g = uniform(seed, gh * gw, stream * 16 + k).reshape(gh, gw)
        amp = float(cell) ** 0.75
        acc += amp * _upsample_bilinear(g, h, w)
        tot += amp
    return acc / tot


def make_frame(seed, h, w, kind="luv"):
    """One synthetic frame in the reference's transposed planar layout.

    Returns float32 [d][w][h] (h contiguous).  kind: 'luv' (3 planes with the
    value ranges of rgbConvert.cpp:34-37), 'rgb' (3 planes in [0,1]) or 'gray'
    (1 plane in [0,1]).
    """
    def plane(stream):
        n = band_noise(seed, h, w, stream)           # upright (h, w)
        # a few blobs so that gradients have structure beyond noise
        yy = (np.arange(h, dtype=np.float64)[:, None] + 0.5) / h
        xx = (np.arange(w, dtype=np.float64)[None, :] + 0.5) / w
        u = uniform(seed, 12, 900 + stream)
        for b in range(4):
            cy, cx, rad = u[3 * b], u[3 * b + 1], 0.05 + 0.1 * u[3 * b + 2]
            d2 = ((yy - cy) * (yy - cy) * (h * h) + (xx - cx) * (xx - cx) * (w * w)) / (rad * rad * h * w)
            n = n + 0.25 / (1.0 + d2)
        n = (n - 0.2) / 0.9
        return np.clip(n, 0.0, 1.0)

    if kind == "gray":
        planes = [plane(0)]
    elif kind == "rgb":
        planes = [plane(0), plane(1), plane(2)]
    elif kind == "luv":
        planes = [0.37 * plane(0), 0.15 + 0.7 * plane(1), 0.15 + 0.6 * plane(2)]
    else:
        raise ValueError(kind)
    # upright (h, w) -> transposed planar [w][h]
    return np.ascontiguousarray(np.stack([p.T for p in planes]).astype(np.float32))


# Typical value ranges (q25, q75) of the ten ACF channels on make_frame('luv')
# data after the pyramid: L,U,V,M,H0..H5.  Frozen from a one-off calibration run of
# the oracle (quartiles of each channel over a few frames; the script was not kept) so that
# thresholds split the windows.
_CHN_Q = {
    "luv": [(0.13, 0.20), (0.38, 0.50), (0.35, 0.46)],
    "gray": [(0.3, 0.6)],
    "mag": [(0.355, 0.44)],
    "hist": [(0.034, 0.088)],
}


# Leaf-value schedule of the synthetic cascade (calibrated with the oracle on
# make_frame(1, 1080, 1920): 13.9 trees per window on average, 335 of the
# 662,799 windows (5e-4) survive all 2048 trees).
HS_AMP, HS_T0, HS_POW, HS_FLOOR, HS_DRIFT, HS_DPOW = 0.36, 36.0, 2.5, 0.001, 0.45, 0.3


def default_options(**over):
    """Options tree defaults (chnsCompute.cpp:156-197, chnsPyramid.cpp:177-205)."""
    m = dict(
        treeDepth=2, modelDs_h=80, modelDs_w=80, modelDsPad_h=80, modelDsPad_w=80, stride=4, cascThr=-1.0,
        nPerOct=8, nOctUp=0, nApprox=7, lambdas=[0.0, 0.1105, 0.1083], pad_h=0, pad_w=0, minDs_h=80, minDs_w=80,
        smooth=1.0, shrink=4, colorEnabled=1, colorSmooth=1.0, colorSpace=capi.CS_LUV, gradMagEnabled=1, colorChn=0,
        normRad=5, normConst=0.005, full=0, gradHistEnabled=1, binSize=0, nOrients=6, softBin=0, isLuv=1,
    )
    m.update(over)
    return m


def n_channels(m):
    d = 1 if m["colorSpace"] == capi.CS_GRAY else 3
    return (d if m["colorEnabled"] else 0) + (1 if m["gradMagEnabled"] else 0) + \
        (m["nOrients"] if m["gradHistEnabled"] else 0)


def make_model(seed=1, nTrees=2048, name="FACE80", **over):
    """Synthetic boosted-tree cascade of the named shape.

    FACE80: 80x80, 10 channels (LUV+M+6H), depth 2, 2048 trees, stride 4,
    cascThr -1 (README.rst:195-199 names the model; its file is absent).
    fids ~ U[0, nC*mh*mw); thrs drawn between the channel quartiles above; leaf
    values decay like a boosted classifier's and carry a negative drift so that
    most windows are rejected after 10-20 trees and a small fraction survives.
    """
    presets = {
        "FACE80": dict(modelDs_h=80, modelDs_w=80, modelDsPad_h=80, modelDsPad_w=80, minDs_h=80, minDs_w=80),
        "FACE64": dict(modelDs_h=64, modelDs_w=64, modelDsPad_h=64, modelDsPad_w=64, minDs_h=64, minDs_w=64,
                       colorEnabled=0, colorSpace=capi.CS_GRAY, isLuv=0),
        "INRIA": dict(modelDs_h=100, modelDs_w=41, modelDsPad_h=128, modelDsPad_w=64, minDs_h=100, minDs_w=41,
                      pad_h=16, pad_w=12, nOctUp=1, isLuv=0),
        "TINY": dict(modelDs_h=16, modelDs_w=16, modelDsPad_h=16, modelDsPad_w=16, minDs_h=16, minDs_w=16),
    }
    m = default_options(**presets[name])
    m.update(over)
    depth = m["treeDepth"]
    nC = n_channels(m)
    mh, mw = m["modelDsPad_h"] // m["shrink"], m["modelDsPad_w"] // m["shrink"]
    ldcf_k = int(m.get("ldcfK", 0))
    if ldcf_k > 0:
        # LDCF (BASELINE cfg 5): the cascade sees nC*k channels at shrink*2 (include/acf_hip.h, acf_hip_params::ldcfK)
        mh, mw = m["modelDsPad_h"] // (2 * m["shrink"]), m["modelDsPad_w"] // (2 * m["shrink"])
        fu = uniform(seed, ldcf_k * nC * 25, 31).reshape(ldcf_k, nC, 5, 5)
        filt = (fu - 0.5) * 0.4
        filt[0, :, 2, 2] += 1.0  # first filter close to identity, the others zero-mean-ish band filters
        m["ldcfFilters"] = filt.astype(np.float32)
    nCe = nC * max(ldcf_k, 1)
    nF = nCe * mh * mw
    if depth > 0:
        nNodes = (1 << (depth + 1)) - 1
        nInternal = (1 << depth) - 1
    else:
        nNodes, nInternal = 7, 3  # variable-depth models use the child[] walk; built below
    u = uniform(seed, nTrees * nNodes * 4, 7).reshape(4, nTrees, nNodes)
    fids = np.minimum((u[0] * nF).astype(np.int64), nF - 1).astype(np.uint32)
    # channel quartiles per feature id
    qs = []
    if m["colorEnabled"]:
        qs += _CHN_Q["gray"] if m["colorSpace"] == capi.CS_GRAY else _CHN_Q["luv"]
    if m["gradMagEnabled"]:
        qs += _CHN_Q["mag"]
    if m["gradHistEnabled"]:
        qs += _CHN_Q["hist"] * m["nOrients"]
    qs = np.asarray(qs, dtype=np.float64)
    if ldcf_k > 0:
        # filtered channels: filter 0 keeps the channel's range, the others are centred near 0 with a fraction of its spread
        base = qs
        spread = (base[:, 1] - base[:, 0])[:, None] * np.array([-0.6, 0.6])
        qs = np.concatenate([base] + [spread for _ in range(ldcf_k - 1)], axis=0)
    ch = fids // (mh * mw)
    lo, hi = qs[ch, 0], qs[ch, 1]
    thrs = (lo + u[1] * (hi - lo)).astype(np.float32)
    t = np.arange(nTrees, dtype=np.float64)[:, None]
    amp = HS_AMP / (1.0 + t / HS_T0) ** HS_POW + HS_FLOOR
    sign = np.where(u[2] < 0.5, -1.0, 1.0)
    drift = HS_DRIFT * HS_AMP / (1.0 + t / HS_T0) ** (HS_POW + HS_DPOW)
    hs = (amp * sign * (0.5 + 0.5 * u[3]) - drift).astype(np.float32)
    child = np.zeros((nTrees, nNodes), dtype=np.uint32)
    if depth == 0:
        # 3 internal nodes + 4 leaves, same topology as depth 2 but walked
        # through child[] (1-based index of the right-hand child, 0 = leaf):
        # k_next = child[k] - (ftr < thr) (acfDetect1.cpp:146-155).
        child[:, 0] = 2 + 1 - 1  # children of node 0 are nodes 1,2 -> child = 2 (k - 1 -> 1 if less)
        child[:, 0] = 2
        child[:, 1] = 4
        child[:, 2] = 6
    else:
        thrs[:, nInternal:] = 0
        fids[:, nInternal:] = 0
        hs[:, :nInternal] = 0
    m.update(fids=fids, thrs=thrs, hs=hs, child=child, name=name)
    return m
