// cascade_plan.hip.h — part of acf_hip.hip (included there, inside its extern "C" block, and nowhere else): the cascade's plan-time
// tables.  buildTileSet: tile geometry, the tile list and the node records in tile-layout offsets of the LDS-tiled cascade
// (k_cascade_tile3 / tile3D / tileD), on float cells or on 16-bit threshold-rank cells.  buildCascadeTables: everything
// acf_hip_plan and acf_hip_op_acf_detect1 need for a list of level geometries — cids (acfDetect1.cpp:390-406), packed node
// records, rank tables (host_plan.h), tile sets, tail tables, leaf-code buffers.
#pragma once

// The cascade code reads the cell size from c->p.shrink; the LDCF cascade works on cells of 2*shrink pixels.
struct ShrinkScope
{
    acf_hip_ctx* c;
    int saved;
    ShrinkScope(acf_hip_ctx* ctx, int factor) : c(ctx), saved(ctx->p.shrink) { c->p.shrink = saved * factor; }
    ~ShrinkScope() { c->p.shrink = saved; }
};

// Build the cascade tables for a list of level geometries (hP, wP) into the
// context.  Shared by acf_hip_plan and acf_hip_op_acf_detect1.
// One set of tile tables of the LDS-tiled cascade: geometry, tile list, node records with tile-layout offsets.  `rank` ==
// nullptr: float cells, thresholds as float bits; else 16-bit threshold-rank cells, thresholds as rank indices.
struct TileSet
{
    bool ok = false;
    TileGeom g{};
    int nTiles = 0, aTB = 4;
    CascTile* d_tiles = nullptr;
    TreeNode* d_tileNodes = nullptr;
    uint32_t* d_tileNodesS = nullptr;
    // depths other than 2 (k_cascade_tileD): records of trees [0, t1D) in batches of tbD
    uint32_t* d_nodesD = nullptr;
    int tbD = 0, t1D = 0;
    uint32_t* d_tileOffD = nullptr; // k_cascade_tile3D: tile offsets of every node, [tree][nTreeNodes]
    uint32_t* d_thrsRankD = nullptr; // ... and, on rank cells, the thresholds' rank indices in the same layout
};

static int buildTileSet(acf_hip_ctx* c, const std::vector<acf_hip_level>& lv, int nChns, const RankTables* rank, TileSet& out, bool allowPooledD = true)
{
    const acf_hip_params& p = c->p;
    const int mH = p.modelDsPad_h / p.shrink, mW = p.modelDsPad_w / p.shrink;
    const int cellBytes = rank ? 2 : 4, CPB = 16 / cellBytes; // cells per 16-byte fill chunk
    int rc;
    TileGeom g{};
    g.step = p.stride / p.shrink;
    g.TR = 32;
    g.winFloats = nChns * mW * mH;
    // Stage boundaries.  k_cascade_tile3 (pooled survivors, the default for depth 2): dense [0,16) on every window, dense
    // [16,32) on the workgroup's pooled survivors, sparse [32,128) as leaf codes + one ordered chain.
    // (depths 3, 4, and 1 on rank cells: k_cascade_tile3D, the same stages, for models of at least 32 trees.  Depth 1 on FLOAT cells keeps
    // k_cascade_tileD + the staged queue: stumps reject slowly, half of a tile's windows are still alive at tree 32, and with the float
    // tile's two workgroups per CU the queue's lanes = windows form beats items = windows x trees (26 against 37 us per 1080p frame; on
    // rank cells the pooled kernel takes 19) — ACF_HIP_TILED_POOLED1 pools it there too; ACF_HIP_TILED_STAGED keeps the staged form everywhere)
    const bool pooledD = allowPooledD && ((p.treeDepth == 1 && (rank || fallbackForced(FB_TILED_POOLED1))) || p.treeDepth == 3 || p.treeDepth == 4) && p.nTrees >= 32 &&
        !fallbackForced(FB_TILED_STAGED);
    const bool pooled = p.treeDepth == 2 || pooledD;
    int bounds[5] = { 0, 32, 32, 64, 128 };
    if (pooled)
    {
        bounds[1] = 16;
        bounds[3] = 32;
    }
    if (const char* e = getenv("ACF_HIP_CASC_BOUNDS")) // tuning knob: "b1,b2,b3,b4"
    {
        int v1, v2, v3, v4;
        if (sscanf(e, "%d,%d,%d,%d", &v1, &v2, &v3, &v4) == 4 && 0 < v1 && v1 <= v2 && v2 <= v3 && v3 <= v4)
        {
            bounds[1] = v1;
            bounds[2] = v2;
            bounds[3] = v3;
            bounds[4] = v4;
        }
    }
    if (pooled)
    {
        // the dense stages run whole batches of four trees (unless the model ends inside one); at most 128 sparse trees
        bounds[1] = std::max(4, bounds[1] / 4 * 4);
        bounds[2] = std::max(bounds[1], bounds[2] / 4 * 4);
        bounds[3] = bounds[2];
        bounds[4] = std::min(std::max(bounds[4], bounds[2]), bounds[2] + 128);
    }
    for (int i = 0; i < 5; i++)
    {
        g.b[i] = std::min(bounds[i], p.nTrees);
    }
    g.pooled = pooled ? 1 : 0;
    if (pooled)
    {
        const int tsPad = (g.b[4] - g.b[2] + 15) / 16 * 16;
        g.pitchC = (tsPad / 4) | 1; // dwords per window, odd: the chain's lanes (windows) read conflict-free
        g.pitchC *= 4;
    }
    // waves per tile: the largest of 8/4/2/1 whose footprint + survivor lists leave room for several workgroups per CU
    // (160 KiB LDS: three with rank cells, two with floats), else one
    const int W = 1; // windows per lane in stage A
    auto ldsBytes = [&](int nw, int passW) {
        const int tc = nw * W * (64 / g.TR);
        const int64_t rows = int64_t(g.TR - 1) * g.step + mH, cols = int64_t(tc - 1) * g.step + mW;
        const int64_t rowsP = (rows + CPB - 1) / CPB * CPB;
        // k_cascade_tileD (depths other than 2 on float cells): footprint + one survivor list segment per wave (+ its few static words)
        if (!g.pooled)
        {
            return int64_t(nChns) * rowsP * cols * cellBytes + int64_t(nw) * 64 * 8 + 64;
        }
        // k_cascade_tile3: leaf table + footprint + list 1 (later the codes of 64 windows) + list 2
        const int64_t nwin = int64_t(nw) * 64;
        const int64_t leafBytes = pooledD ? int64_t(128) * 4 * (int64_t(1) << p.treeDepth) : int64_t(TILE3_LEAF_BYTES);
        if (int64_t(nChns) * rowsP * cols > 65535)
        {
            return int64_t(1) << 40; // (its list entries hold a window's first cell in 16 bits)
        }
        return leafBytes + int64_t(nChns) * rowsP * cols * cellBytes + ((std::max<int64_t>(nwin * 8, passW * int64_t(g.pitchC)) + 15) / 16 * 16) + nwin * 8 + 64;
    };
    int nw = 0;
    if (const char* e = getenv("ACF_HIP_TILE_TR")) // tuning knobs: rows of windows per tile, waves per tile
    {
        const int v = atoi(e);
        g.TR = (v >= 8 && v <= 64) ? v : g.TR; // (a value that does not divide 64 leaves 64 % TR lanes of a wave idle in stage A)
    }
    const int wgPerCu = 3; // the footprint + lists must fit three times into a CU's LDS with rank cells
    const char* nwEnv = getenv("ACF_HIP_TILE_NW");
    const int nwForce = nwEnv ? atoi(nwEnv) : 0;
    for (int64_t limit : { rank ? int64_t(160 * 1024 / wgPerCu / 1280 * 1280) : int64_t(80) * 1024, int64_t(80) * 1024, int64_t(159) * 1024 })
    {
        for (int cand : { nwForce == 16 && pooled ? 16 : 8, 8 / W, 4 / W, 2 / W, 1 })
        {
            // (k_cascade_tile3's sparse stage: one thread per tree of a window)
            const int tlp = g.b[4] - g.b[2] <= 32 ? 32 : (g.b[4] - g.b[2] <= 64 ? 64 : 128);
            for (int passW : { 64, 32 }) // (k_cascade_tile3: windows per pass of the sparse stage)
            {
                if (!nw && cand >= 1 && ldsBytes(cand, passW) <= limit && (!nwForce || cand == nwForce) && cand * 64 >= g.TR && (!g.pooled || cand * 64 >= tlp))
                {
                    nw = cand;
                    g.passW = passW;
                }
            }
        }
    }
    if (!nw)
    {
        return pooledD ? buildTileSet(c, lv, nChns, rank, out, false) : ACF_HIP_OK;
    }
    g.NW = nw;
    g.W = W;
    g.TC = nw * W * (64 / g.TR);
    g.rowsT = (g.TR - 1) * g.step + mH;
    g.colsT = (g.TC - 1) * g.step + mW;
    g.rowsP = (g.rowsT + CPB - 1) / CPB * CPB;
    g.tileFloats = nChns * g.rowsP * g.colsT; // cells
    g.cpsMagic = uint32_t(((uint64_t(1) << 32) + uint32_t(g.rowsP / CPB) - 1) / uint32_t(g.rowsP / CPB));
    g.colsMagic = uint32_t(((uint64_t(1) << 32) + uint32_t(g.colsT) - 1) / uint32_t(g.colsT));
    std::vector<CascTile> tiles;
    bool ok = true;
    {
        // the fill kernel divides chunk indices by mulhi with these magics: check every index it will see
        const uint32_t cps = uint32_t(g.rowsP / CPB), nSeg = uint32_t(nChns * g.colsT);
        for (uint32_t q = 0; q < nSeg * cps && ok; q++)
        {
            const uint32_t seg = uint32_t((uint64_t(q) * g.cpsMagic) >> 32); // a divisor of 1 has magic 2^32 = 0 in 32 bits: caught here
            ok = seg == q / cps && uint32_t((uint64_t(seg) * g.colsMagic) >> 32) == seg / uint32_t(g.colsT);
        }
    }
    for (size_t i = 0; i < lv.size() && ok; i++)
    {
        for (int c0 = 0; c0 < lv[i].nWinC; c0 += g.TC)
        {
            for (int r0 = 0; r0 < lv[i].nWinR; r0 += g.TR)
            {
                if (r0 > 32767 || c0 > 32767)
                {
                    ok = false;
                    break;
                }
                CascTile t{};
                t.level = int16_t(i);
                t.r0 = int16_t(r0);
                t.c0 = int16_t(c0);
                tiles.push_back(t);
            }
        }
    }
    if (!ok)
    {
        return ACF_HIP_OK;
    }
    if (p.treeDepth != 2)
    {
        // k_cascade_tileD: stage 0 of the staged path, trees [0, 32) (the staged path's second boundary), on float tiles.
        // Records per batch of TB trees: {off[TB][NN], thr[TB][NN], hs[TB][NL]}, nodes in heap order, leaves left to right.
        const int D = p.treeDepth;
        if ((rank && !g.pooled) || D < 1 || D > 4 || D == 2)
        {
            return ACF_HIP_OK; // (rank cells: k_cascade_tile3D only)
        }
        const int NN = (1 << D) - 1, NL = 1 << D, TB = D == 1 ? 4 : (D == 3 ? 2 : 1);
        const int t1 = std::min(32, p.nTrees) / TB * TB;
        if (t1 <= 0 || (t1 != p.nTrees && t1 != 32) || p.nTreeNodes < NN + NL)
        {
            return ACF_HIP_OK; // (a model shorter than 32 trees whose length is not a multiple of the batch: staged path)
        }
        std::vector<uint32_t> nd(size_t(t1 / TB) * TB * (2 * NN + NL), 0u);
        for (int t = 0; t < t1; t++)
        {
            const size_t q = size_t(t) * p.nTreeNodes;
            uint32_t* d = nd.data() + size_t(t / TB) * TB * (2 * NN + NL);
            const int tq = t % TB;
            for (int k = 0; k < NN; k++)
            {
                const uint32_t f = c->fids[q + k];
                const uint32_t z = f / (mW * mH), cc = (f / mH) % mW, rr = f % mH; // computeChannelIndexColMajor, acfDetect1.cpp:390-406
                d[tq * NN + k] = (z * uint32_t(g.colsT) + cc) * uint32_t(g.rowsP) + rr;
                if (rank)
                {
                    d[TB * NN + tq * NN + k] = rank->rankOfThreshold(int(z), c->thrs[q + k]);
                }
                else
                {
                    memcpy(&d[TB * NN + tq * NN + k], &c->thrs[q + k], 4);
                }
            }
            for (int j = 0; j < NL; j++)
            {
                memcpy(&d[2 * TB * NN + tq * NL + j], &c->hs[q + NN + j], 4);
            }
        }
        if (g.pooled)
        {
            if (t1 != 32 || p.nTreeNodes < NN + NL || g.tileFloats > 65535)
            {
                return buildTileSet(c, lv, nChns, rank, out, false); // (k_cascade_tileD + the staged queue)
            }
            std::vector<uint32_t> to(size_t(p.nTrees) * p.nTreeNodes, 0u), tr(rank ? size_t(p.nTrees) * p.nTreeNodes : 0, 0u);
            for (int t = 0; t < p.nTrees; t++)
            {
                for (int k = 0; k < NN; k++)
                {
                    const uint32_t f = c->fids[size_t(t) * p.nTreeNodes + k];
                    const uint32_t z = f / (mW * mH), cc = (f / mH) % mW, rr = f % mH;
                    to[size_t(t) * p.nTreeNodes + k] = (z * uint32_t(g.colsT) + cc) * uint32_t(g.rowsP) + rr;
                    if (rank)
                    {
                        tr[size_t(t) * p.nTreeNodes + k] = rank->rankOfThreshold(int(z), c->thrs[size_t(t) * p.nTreeNodes + k]);
                    }
                }
            }
            if ((rc = devUpload(c, &out.d_tileOffD, to)) || (rank && (rc = devUpload(c, &out.d_thrsRankD, tr))))
            {
                return rc;
            }
        }
        out.g = g;
        out.nTiles = int(tiles.size());
        out.tbD = TB;
        out.t1D = t1;
        if ((rc = devUpload(c, &out.d_nodesD, nd)) || (rc = devUpload(c, &out.d_tiles, tiles)))
        {
            return rc;
        }
        out.ok = true;
        return ACF_HIP_OK;
    }
    std::vector<TreeNode> tileNodes(size_t(std::max(p.nTrees, 1)));
    for (int t = 0; t < p.nTrees; t++)
    {
        const size_t q = size_t(t) * p.nTreeNodes;
        TreeNode a{};
        for (int k = 0; k < 3; k++)
        {
            const uint32_t f = c->fids[q + k];
            const uint32_t z = f / (mW * mH), cc = (f / mH) % mW, rr = f % mH; // computeChannelIndexColMajor, acfDetect1.cpp:390-406
            a.off[k] = (z * uint32_t(g.colsT) + cc) * uint32_t(g.rowsP) + rr;
            if (rank)
            {
                const uint32_t rk = rank->rankOfThreshold(int(z), c->thrs[q + k]);
                memcpy(&a.thr[k], &rk, 4);
            }
            else
            {
                a.thr[k] = c->thrs[q + k];
            }
        }
        for (int k = 0; k < 4; k++)
        {
            a.hs[k] = c->hs[q + 3 + k];
        }
        tileNodes[size_t(t)] = a;
    }
    // stage A of the tile kernels reads its trees four at a time through the scalar unit
    // (batches of 4 measured 4 % faster than batches of 8 once the leaf add went under EXEC)
    const int aTB = 4;
    const int nTreesS = (g.pooled ? g.b[2] : g.b[1]) / aTB * aTB; // (k_cascade_tile3: both dense stages read batches)
    std::vector<uint32_t> nodesS(size_t(std::max(nTreesS / aTB, 1)) * 10 * aTB, 0u);
    for (int t = 0; t + aTB - 1 < nTreesS; t += aTB)
    {
        uint32_t* d = nodesS.data() + size_t(t / aTB) * 10 * aTB;
        for (int q = 0; q < aTB; q++)
        {
            const TreeNode& nd = tileNodes[size_t(t + q)];
            for (int k = 0; k < 3; k++)
            {
                d[3 * q + k] = nd.off[k];
                memcpy(&d[3 * aTB + 3 * q + k], &nd.thr[k], 4);
            }
            for (int k = 0; k < 4; k++)
            {
                memcpy(&d[6 * aTB + 4 * q + k], &nd.hs[k], 4);
            }
        }
    }
    out.aTB = aTB;
    out.g = g;
    out.nTiles = int(tiles.size());
    if ((rc = devUpload(c, &out.d_tileNodesS, nodesS)) || (rc = devUpload(c, &out.d_tiles, tiles)) || (rc = devUpload(c, &out.d_tileNodes, tileNodes)))
    {
        return rc;
    }
    out.ok = true;
    return ACF_HIP_OK;
}

static int buildCascadeTables(acf_hip_ctx* c, const std::vector<acf_hip_level>& lv, int nChns, CascState& cs, bool wantRank = false)
{
    CascLevel** d_levels = &cs.d_cascLevels;
    int32_t** d_blockLevel = &cs.d_blockLevel;
    int* blocksPerFrame = &cs.blocksPerFrame;
    uint32_t** d_cidAll = &cs.d_cidAll;
    CascNode2** d_nodes2 = &cs.d_nodes2;
    const acf_hip_params& p = c->p;
    const int mH = p.modelDsPad_h / p.shrink, mW = p.modelDsPad_w / p.shrink;
    const uint32_t nF = uint32_t(nChns) * mH * mW;
    const size_t nNodes = size_t(p.nTrees) * p.nTreeNodes;
    const bool packed = p.treeDepth == 2;
    std::vector<CascLevel> cl(lv.size());
    std::vector<int32_t> bl;
    std::vector<uint32_t> cidAll;
    std::vector<CascNode2> nodes2;
    // which nodes carry a feature test
    std::vector<uint8_t> internal(nNodes, 0);
    for (int t = 0; t < p.nTrees; t++)
    {
        for (int k = 0; k < p.nTreeNodes; k++)
        {
            const size_t q = size_t(t) * p.nTreeNodes + k;
            internal[q] = p.treeDepth > 0 ? (k < (1 << p.treeDepth) - 1) : (c->child[q] != 0);
            if (internal[q] && c->fids[q] >= nF)
            {
                return fail(c, ACF_HIP_E_INVALID, "model: feature id out of range for modelDsPad/shrink/channels");
            }
            if (p.treeDepth == 0 && c->child[q] != 0)
            {
                // next node = child[k] - (ftr < thr) in 0-based terms child[k] - 1 or child[k] (acfDetect1.cpp:146-155): both must
                // stay inside the tree AND lie after k — trees are stored parent-first, and a backward or self reference in
                // an untrusted model file would make the walk `while (child[k])` spin forever on the GPU
                if (c->child[q] >= uint32_t(p.nTreeNodes) || c->child[q] - 1 <= uint32_t(k))
                {
                    return fail(c, ACF_HIP_E_INVALID, "model: child index out of range or not after its parent");
                }
            }
        }
    }
    int block = 0;
    cs.dedupQ = (p.stride < p.shrink && p.shrink % p.stride == 0 && !fallbackForced(FB_NO_DEDUP)) ? p.shrink / p.stride : 1;
    std::vector<int2> realWin(lv.size());
    for (size_t i = 0; i < lv.size(); i++)
    {
        CascLevel& L = cl[i];
        L.hP = lv[i].hP;
        L.wP = lv[i].wP;
        L.nWinR = lv[i].nWinR;
        L.nWinC = lv[i].nWinC;
        realWin[i] = make_int2(L.nWinR, L.nWinC);
        if (cs.dedupQ > 1)
        {
            // distinct offsets r * stride / shrink of the windows r = 0 .. nWinR - 1
            L.nWinR = L.nWinR > 0 ? (L.nWinR - 1) / cs.dedupQ + 1 : 0;
            L.nWinC = L.nWinC > 0 ? (L.nWinC - 1) / cs.dedupQ + 1 : 0;
        }
        L.nWin = L.nWinR * L.nWinC;
        L.off = lv[i].offset;
        L.firstBlock = block;
        const int nb = cdiv(L.nWin, 256);
        for (int b = 0; b < nb; b++)
        {
            bl.push_back(int32_t(i));
        }
        block += nb;
        const int64_t area = int64_t(L.hP) * L.wP;
        if (area * nChns >= (int64_t(1) << 31) || L.nWin >= (1 << 24))
        {
            return fail(c, ACF_HIP_E_UNSUPPORTED, "level too large for 32-bit channel offsets / 24-bit window ids");
        }
        if (packed)
        {
            L.nodeOff = 0;
        }
        else
        {
            L.nodeOff = int64_t(cidAll.size());
            for (size_t q = 0; q < nNodes; q++)
            {
                uint32_t v = 0;
                if (internal[q])
                {
                    const uint32_t f = c->fids[q];
                    const uint32_t z = f / (mW * mH), cc = (f / mH) % mW, rr = f % mH;
                    v = uint32_t(z * area + int64_t(cc) * L.hP + rr);
                }
                cidAll.push_back(v);
            }
        }
    }
    if (packed)
    {
        // one level-independent table: feature ids kept as (z, c, r) of
        // computeChannelIndexColMajor (acfDetect1.cpp:390-406); the kernel rebuilds
        // z*area + c*hP + r from the lane's level geometry
        if (mW > 4095 || mH > 4095 || nChns > 255 || lv.size() > 255)
        {
            return fail(c, ACF_HIP_E_UNSUPPORTED, "model window / channel count too large for the packed node table");
        }
        for (int t = 0; t < p.nTrees; t++)
        {
            CascNode2 nd{};
            const size_t q = size_t(t) * p.nTreeNodes;
            for (int k = 0; k < 3; k++)
            {
                const uint32_t f = c->fids[q + k];
                const uint32_t z = f / (mW * mH), cc = (f / mH) % mW, rr = f % mH;
                nd.zcr[k] = (z << 24) | (cc << 12) | rr;
                nd.thr[k] = c->thrs[q + k];
            }
            for (int k = 0; k < 4; k++)
            {
                nd.hs[k] = c->hs[q + 3 + k];
            }
            nodes2.push_back(nd);
        }
    }
    *blocksPerFrame = block;
    int rc;
    if (cs.dedupQ > 1 && (rc = devUpload(c, &cs.d_realWin, realWin)))
    {
        return rc;
    }
    if ((rc = devUpload(c, d_levels, cl)))
    {
        return rc;
    }
    if ((rc = devUpload(c, d_blockLevel, bl)))
    {
        return rc;
    }
    if ((rc = devUpload(c, d_cidAll, cidAll)))
    {
        return rc;
    }
    if ((rc = devUpload(c, d_nodes2, nodes2)))
    {
        return rc;
    }
    // ---- LDS-tiled path (kernels.hip.h, k_cascade_tile3 + stage E + k_tail_scan, k_cascade_tail3 for queue overflow)
    cs.useTiles = false;
    cs.useRank = false;
    cs.useTileD = false;
    cs.useRankD = false;
    cs.codeCapD = 0;
    if (!packed && p.treeDepth >= 1 && p.treeDepth <= 4 && p.nTrees > 128 && !fallbackForced(FB_NO_TAIL_CODES))
    {
        // the staged path's last stage [128, nTrees) as leaf codes + ordered scan: the first codeCapD queue entries of a frame
        // (sized like the depth-2 path's), when the scan's leaf table fits a workgroup's LDS
        const int nT = p.nTrees - 128, NL = 1 << p.treeDepth;
        if (int64_t((nT + 15) / 16 * 16) * NL * 4 <= 150 * 1024)
        {
            cs.codePitchD = (nT + 63) / 64 * 64;
            int64_t nWinTotal = 0;
            for (const auto& l : lv)
            {
                nWinTotal += int64_t(std::max(l.nWinR, 0)) * std::max(l.nWinC, 0);
            }
            int64_t cap = std::min<int64_t>(std::max<int64_t>(nWinTotal / 64, 1024), 8192);
            cap = std::min(cap, std::max<int64_t>((int64_t(1) << 28) / (int64_t(std::max(c->maxBatch, 1)) * cs.codePitchD), 256));
            cap = std::min<int64_t>(cap, std::max<int64_t>(nWinTotal, 1));
            void* codes = nullptr;
            if (hipMalloc(&codes, size_t(std::max(c->maxBatch, 1)) * size_t(cap) * size_t(cs.codePitchD)) == hipSuccess)
            {
                c->allocs.push_back(codes);
                cs.d_codesD = static_cast<uint8_t*>(codes);
                cs.codeCapD = int(cap);
            }
            else
            {
                (void)hipGetLastError();
            }
        }
    }
    // the rank pyramid of a plan: per level nChns planes [wP][pitchR], pitchR = hP rounded up to 8 cells (16 bytes); bucket tables on
    // the device; the level table again with the rank layout
    auto setupRankPyramid = [&](const RankTables& rt) -> int {
        std::vector<RankJob> jobs(lv.size());
        int64_t off = 0;
        cs.rankMaxWP = 0;
        for (size_t i = 0; i < lv.size(); i++)
        {
            const int pitch = rankPitch(lv[i].hP);
            cl[i].offR = off;
            cl[i].pitchR = pitch;
            jobs[i].src_off = lv[i].offset;
            jobs[i].dst_off = off;
            jobs[i].hP = lv[i].hP;
            jobs[i].wP = lv[i].wP;
            jobs[i].pitchR = pitch;
            off += int64_t(nChns) * pitch * lv[i].wP;
            cs.rankMaxWP = std::max(cs.rankMaxWP, lv[i].wP);
        }
        cs.pyrRCells = off;
        cs.rankMaxRec = rt.maxRec;
        int rcl;
        // + slack: a tile's 16-byte fill chunks run up to rowsP cells past the last column of the last plane
        if ((rcl = devUpload(c, &cs.d_rankChan, rt.chan)) || (rcl = devUpload(c, &cs.d_rankRec, rt.rec)) || (rcl = devUpload(c, &cs.d_rankJobs, jobs)) ||
            (rcl = devAlloc(c, &cs.d_pyrR, size_t(std::max(c->maxBatch, 1)) * size_t(off) + 4096)))
        {
            return rcl;
        }
        HIPCHK(c, hipMemset(cs.d_pyrR, 0, (size_t(std::max(c->maxBatch, 1)) * size_t(off) + 4096) * sizeof(uint16_t))); // pitch padding cells: defined
        HIPCHK(c, hipMemcpy(*d_levels, cl.data(), cl.size() * sizeof(CascLevel), hipMemcpyHostToDevice));
        return ACF_HIP_OK;
    };
    if (!packed && p.treeDepth >= 1 && p.treeDepth <= 4 && p.stride % p.shrink == 0 && p.stride >= p.shrink)
    {
        TileSet tsD;
        if ((rc = buildTileSet(c, lv, nChns, nullptr, tsD)))
        {
            return rc;
        }
        if (tsD.ok && (tsD.g.NW == 8 || tsD.g.NW == 4))
        {
            cs.useTileD = true;
            cs.d_tilesD = tsD.d_tiles;
            cs.nTilesD = tsD.nTiles;
            cs.tbD = tsD.tbD;
            cs.t1D = tsD.t1D;
            cs.geomD = tsD.g;
            cs.d_nodesD = tsD.d_nodesD;
            cs.d_tileOffD = tsD.d_tileOffD;
            // ---- the pooled kernel on threshold-rank cells (the rank tables do not depend on the depth): half the fill, three
            // workgroups per CU.  The float pyramid is still written for these depths (the queue's overflow path reads it).
            if ((tsD.g.pooled || p.treeDepth == 1) && wantRank && !c->noRank && !fallbackForced(FB_TILED_STAGED))
            {
                const int mHc = p.modelDsPad_h / p.shrink, mWc = p.modelDsPad_w / p.shrink;
                std::vector<int32_t> chnOfNode(nNodes, -1);
                for (size_t q = 0; q < nNodes; q++)
                {
                    if (internal[q])
                    {
                        chnOfNode[q] = int32_t(c->fids[q] / uint32_t(mWc * mHc));
                    }
                }
                RankTables rt;
                buildRankTables(c->thrs.data(), chnOfNode.data(), nNodes, nChns, rt);
                TileSet tsR;
                if (rt.ok && (rc = buildTileSet(c, lv, nChns, &rt, tsR)))
                {
                    return rc;
                }
                if (rt.ok && tsR.ok && tsR.g.pooled && tsR.d_thrsRankD && (tsR.g.NW == 8 || tsR.g.NW == 4))
                {
                    if ((rc = setupRankPyramid(rt)))
                    {
                        return rc;
                    }
                    cs.geomDR = tsR.g;
                    cs.d_tilesDR = tsR.d_tiles;
                    cs.nTilesDR = tsR.nTiles;
                    cs.d_nodesDR = tsR.d_nodesD;
                    cs.d_tileOffDR = tsR.d_tileOffD;
                    cs.d_thrsRankD = tsR.d_thrsRankD;
                    cs.useRank = true;
                    cs.useRankD = true;
                }
            }
        }
    }
    if (packed && p.stride % p.shrink == 0 && p.stride >= p.shrink)
    {
        TileSet tsF;
        if ((rc = buildTileSet(c, lv, nChns, nullptr, tsF)))
        {
            return rc;
        }
        const int mHc = p.modelDsPad_h / p.shrink, mWc = p.modelDsPad_w / p.shrink;
        int tw = 0;
        const int winFloats = nChns * mWc * mHc;
        const int tailSlab = (std::max(winFloats, TAIL_G * TAIL_PITCH) + 3) / 4 * 4; // footprint, reused as phase 2's transposition tile (16-byte rows)
        for (int64_t limit : { int64_t(80) * 1024, int64_t(159) * 1024 })
        {
            for (int cand : { 2, 1 })
            {
                if (!tw && int64_t(cand) * tailSlab * 4 <= limit)
                {
                    tw = cand;
                }
            }
        }
        if (tsF.ok && tw)
        {
            const TileGeom& g = tsF.g;
            std::vector<TreeNode> tailNodes(static_cast<size_t>(p.nTrees));
            for (int t = 0; t < p.nTrees; t++)
            {
                const size_t q = size_t(t) * p.nTreeNodes;
                TreeNode b{};
                for (int k = 0; k < 3; k++)
                {
                    b.off[k] = c->fids[q + k];
                    b.thr[k] = c->thrs[q + k];
                }
                for (int k = 0; k < 4; k++)
                {
                    b.hs[k] = c->hs[q + 3 + k];
                }
                tailNodes[size_t(t)] = b;
            }
            cs.aTB = tsF.aTB;
            cs.d_tileNodesS = tsF.d_tileNodesS;
            cs.d_tileNodes = tsF.d_tileNodes;
            cs.d_tiles = tsF.d_tiles;
            cs.nTiles = tsF.nTiles;
            cs.geom = g;
            cs.tailWaves = tw;
            cs.tailSlab = tailSlab;
            cs.tailPad = (std::max(p.nTrees - g.b[4], 1) + 63) / 64 * 64;
            cs.tailBlocks = std::max(512, c->maxBatch);
            // k_cascade_tail3 only takes queue overflow now: its LDS stays small (footprint slabs only, node table from
            // L2) so that its blocks — which leave at once in the normal case — never wait for a whole CU's LDS while
            // other streams' kernels are resident (measured: up to 2.7 ms of queueing per launch with 156 KB blocks)
            cs.tailNodesLds = 0;
            if (g.b[4] < p.nTrees && (rc = devAlloc(c, &cs.d_tailScratch, size_t(cs.tailBlocks) * tw * TAIL_G * cs.tailPad)))
            {
                return rc;
            }
            cs.codeCap = 0;
            if (g.b[4] < p.nTrees && !fallbackForced(FB_TAIL3)) // (forced: every tail window goes to k_cascade_tail3)
            {
                const int nT = p.nTrees - g.b[4];
                cs.codePitch = (nT + 63) / 64 * 64; // stage E writes whole 64-tree batches
                int64_t nWinTotal = 0;
                for (const auto& l : lv)
                {
                    nWinTotal += int64_t(std::max(l.nWinR, 0)) * std::max(l.nWinC, 0);
                }
                // entries per frame with codes: 1/64 of the windows (the tail sees ~1/700 of them on natural
                // images), at most 256 MB for the batch (the rows are touched per survivor: ~1k of them per 1080p
                // frame); whatever is beyond goes to k_cascade_tail3, and so does everything if the buffer cannot be had
                int64_t cap = std::min<int64_t>(std::max<int64_t>(nWinTotal / 64, 1024), 8192);
                cap = std::min(cap, std::max<int64_t>((int64_t(1) << 28) / (int64_t(std::max(c->maxBatch, 1)) * cs.codePitch), 256));
                cap = std::min<int64_t>(cap, std::max<int64_t>(nWinTotal, 1));
                cs.codeCap = int(cap);
                void* codes = nullptr;
                if (hipMalloc(&codes, size_t(std::max(c->maxBatch, 1)) * size_t(cs.codeCap) * size_t(cs.codePitch)) == hipSuccess)
                {
                    c->allocs.push_back(codes);
                    cs.d_tailCodes = static_cast<uint8_t*>(codes);
                }
                else
                {
                    (void)hipGetLastError();
                    cs.codeCap = 0;
                }
            }
            if ((rc = devUpload(c, &cs.d_tailNodes, tailNodes)))
            {
                return rc;
            }
            cs.useTiles = true;
            // ---- the same tiles over threshold-rank cells (16 bits per cell): half the fill, half the LDS
            if (wantRank && !c->noRank)
            {
                std::vector<int32_t> chnOfNode(nNodes, -1);
                for (size_t q = 0; q < nNodes; q++)
                {
                    if (internal[q])
                    {
                        chnOfNode[q] = int32_t(c->fids[q] / uint32_t(mWc * mHc));
                    }
                }
                RankTables rt;
                buildRankTables(c->thrs.data(), chnOfNode.data(), nNodes, nChns, rt);
                TileSet tsR;
                if (rt.ok && (rc = buildTileSet(c, lv, nChns, &rt, tsR)))
                {
                    return rc;
                }
                if (rt.ok && tsR.ok)
                {
                    if ((rc = setupRankPyramid(rt)))
                    {
                        return rc;
                    }
                    std::vector<TreeNode> tailR(static_cast<size_t>(p.nTrees));
                    for (int t = 0; t < p.nTrees; t++)
                    {
                        const size_t q = size_t(t) * p.nTreeNodes;
                        TreeNode b{};
                        for (int k = 0; k < 3; k++)
                        {
                            const uint32_t f = c->fids[q + k];
                            const uint32_t z = f / uint32_t(mWc * mHc), cc = (f / uint32_t(mHc)) % uint32_t(mWc), rr = f % uint32_t(mHc);
                            b.off[k] = (z << 24) | (cc << 12) | rr; // (mW, mH <= 4095 and nChns <= 255: checked for the packed node table above)
                            const uint32_t rk = rt.rankOfThreshold(int(z), c->thrs[q + k]);
                            memcpy(&b.thr[k], &rk, 4);
                        }
                        for (int k = 0; k < 4; k++)
                        {
                            b.hs[k] = c->hs[q + 3 + k];
                        }
                        tailR[size_t(t)] = b;
                    }
                    if ((rc = devUpload(c, &cs.d_tailNodesR, tailR)))
                    {
                        return rc;
                    }
                    cs.geomR = tsR.g;
                    cs.d_tilesR = tsR.d_tiles;
                    cs.nTilesR = tsR.nTiles;
                    cs.d_tileNodesR = tsR.d_tileNodes;
                    cs.d_tileNodesSR = tsR.d_tileNodesS;
                    cs.useRank = true;
                }
            }
        }
    }
    return ACF_HIP_OK;
}
