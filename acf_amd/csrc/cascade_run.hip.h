// cascade_run.hip.h — part of acf_hip.hip (included there and nowhere else): acf_hip_detect's launch logic.
//   runCascadeTiled   the LDS-tiled cascade (depth 2: k_cascade_tile3; depths 1, 3, 4: tile3D / tileD) + tail (leaf codes +
//                     k_tail_scan, or k_cascade_tail3 / _tail_rank for what does not fit the code buffers)
//   runCascade        acfDetect1 on every level of a batch (tiled or staged), k_expand_hits for stride < shrink, k_sort_map
//                     (scale-ordered output + box mapping, ACF.cpp:302-329), launchNms (bbNms + prune on the device)
#pragma once

static int allowLds(acf_hip_ctx* c, const void* kernel, size_t bytes)
{
    if (bytes > 64 * 1024)
    {
        HIPCHK(c, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
    }
    return ACF_HIP_OK;
}

// the ONE statement of "the cascade runs on LDS tiles": runCascade's branch and acf_hip_run's early counter clear both ask it
static inline bool tiledCascadeSelected(const acf_hip_ctx* c)
{
    return c->cs.useTiles && !c->noTiles;
}

static int runCascadeTiled(acf_hip_ctx* c, const float* pyr, int64_t pyr_fs, int nF, int nChns)
{
    const acf_hip_params& p = c->p;
    const CascState& cs = c->cs;
    const TileGeom& g = cs.geom;
    if (!c->countersZeroed)
    {
        HIPCHK(c, hipMemsetAsync(cs.d_qcounts, 0, sizeof(int32_t) * (2 * size_t(c->maxBatch) + 8), c->stream));
    }
    TileArgs a{};
    a.pyr = pyr;
    a.pyr_fs = pyr_fs;
    a.levels = cs.d_cascLevels;
    a.tiles = cs.d_tiles;
    a.nTiles = cs.nTiles;
    a.nFrames = nF;
    a.nChns = nChns;
    a.mH = p.modelDsPad_h / p.shrink;
    a.mW = p.modelDsPad_w / p.shrink;
    a.nTrees = p.nTrees;
    a.g = g;
    a.tileNodes = cs.d_tileNodes;
    a.tileNodesS = cs.d_tileNodesS;
    a.aTB = cs.aTB;
    a.tailNodes = cs.d_tailNodes;
    a.cascThr = float(p.cascThr); // DetectionParams::cascThr is a float (acfDetect1.cpp:63,323)
    a.q = cs.d_queue[0];
    a.qcount = cs.d_qcounts;
    a.qhead = cs.d_qcounts + c->maxBatch;
    a.tileNext = cs.d_qcounts + 2 * size_t(c->maxBatch);
    a.qcap = cs.qcap;
    a.hits = cs.d_hits;
    a.counts = cs.d_counts;
    a.maxHits = c->maxHits;
    a.tailScratch = cs.d_tailScratch;
    a.tailPad = cs.tailPad;
    a.tailSlab = cs.tailSlab;
    a.tailNodesLds = cs.tailNodesLds;
    a.tailCodes = cs.d_tailCodes;
    a.codeCap = cs.codeCap;
    a.codePitch = cs.codePitch;
    a.debug = 0;
#ifdef ACF_HIP_STAMPS
    static const int cascDebug = getenv("ACF_HIP_CASC_DEBUG") ? atoi(getenv("ACF_HIP_CASC_DEBUG")) : 0; // timing experiments (profiles/ab_*.sh)
    a.debug = cascDebug;
#endif
    if (a.debug & 12)
    {
        a.debug |= 4;
        const int64_t total = int64_t(std::max(cs.nTiles, cs.nTilesR)) * nF;
        HIPCHK(c, hipMalloc(&a.stamps, size_t((total + 7) / 8 * 8) * 8 * sizeof(long long)));
        HIPCHK(c, hipMemsetAsync(a.stamps, 0, size_t((total + 7) / 8 * 8) * 8 * sizeof(long long), c->stream));
    }
    const bool rank = cs.useRank && !c->noRank && pyr == c->d_pyr;
    if (rank && !c->ranksValid)
    {
        // the float pyramid -> threshold-rank cells (levels whose kernels did not emit them)
        int rc = 0;
        const size_t ldsR = size_t(cs.rankMaxRec) * sizeof(RankRec);
        if ((rc = allowLds(c, reinterpret_cast<const void*>(&k_rank), ldsR)))
        {
            return rc;
        }
        prof(c, "k_rank");
        hipLaunchKernelGGL(k_rank, dim3(cdiv(cs.rankMaxWP, RANK_CHUNK_COLS), int(c->plan.levels.size()) * nChns, nF), dim3(256), ldsR, c->stream, pyr, pyr_fs,
            cs.d_pyrR, cs.pyrRCells, (const RankJob*)cs.d_rankJobs, nChns, (const RankChan*)cs.d_rankChan, (const RankRec*)cs.d_rankRec);
        LAUNCHCHK(c, "k_rank");
        c->ranksValid = true;
    }
    if (cs.nTiles > 0)
    {
        TileArgs at = a; // the tile kernel's view: the rank form has its own tile geometry, tile list and node records
        if (rank)
        {
            at.pyrR = cs.d_pyrR;
            at.pyrR_fs = cs.pyrRCells;
            at.g = cs.geomR;
            at.tiles = cs.d_tilesR;
            at.nTiles = cs.nTilesR;
            at.tileNodes = cs.d_tileNodesR;
            at.tileNodesS = cs.d_tileNodesSR;
        }
        const TileGeom& gt = at.g;
        const int64_t total = int64_t(at.nTiles) * nF;
        const int64_t perX = (total + 7) / 8;
        const size_t nwin = size_t(gt.NW) * 64;
        const size_t lds = gt.pooled ? size_t(TILE3_LEAF_BYTES) + size_t(gt.tileFloats) * (rank ? 2 : 4) + ((std::max(nwin * 8, size_t(gt.passW) * size_t(gt.pitchC)) + 15) / 16 * 16) + nwin * 8
                                     : size_t(gt.tileFloats) * (rank ? 2 : 4) + nwin * 8;
        // k_cascade_tile3: persistent workgroups (as many as the CUs hold at once) that draw their tiles from one counter per XCD
        // (tilePersist: 0 one workgroup per tile, 1 as many workgroups as the device's CUs hold at once, n > 1 that many, rounded up
        // to a multiple of 8 = the tile counters)
        const int64_t resident = int64_t(c->numCus) * std::max<int64_t>(1, int64_t(c->ldsPerCu) / int64_t((lds + 1279) / 1280 * 1280));
        const int64_t gridP = c->tilePersist > 1 ? int64_t(c->tilePersist) : resident;
        const bool persist = gt.pooled && c->tilePersist > 0 && (gridP + 7) / 8 * 8 < perX * 8;
        if (!persist)
        {
            at.tileNext = nullptr;
        }
        dim3 grid((unsigned int)(persist ? (gridP + 7) / 8 * 8 : perX * 8)), block(gt.NW * 64);
        int rc = 0;
        if ((c->cascTurns & 1) && (rc = turnBegin(c, 0, 0))) // (before the profile event: the wait for the turn is not the kernel's time)
        {
            return rc;
        }
        prof(c, "k_cascade_tile");
#define TILE2_LAUNCH(N, CT)                                                                           \
    {                                                                                                 \
        if ((rc = allowLds(c, reinterpret_cast<const void*>(&k_cascade_tile3<N, CT>), lds)))          \
            return rc;                                                                                \
        hipLaunchKernelGGL((k_cascade_tile3<N, CT>), grid, block, lds, c->stream, at);                \
    }
#define TILE3_LAUNCH16(CT)                                                                        \
    if ((rc = allowLds(c, reinterpret_cast<const void*>(&k_cascade_tile3<16, CT>), lds)))             \
        return rc;                                                                                    \
    hipLaunchKernelGGL((k_cascade_tile3<16, CT>), grid, block, lds, c->stream, at);
#define TILE2_NW(CT)                              \
    switch (gt.NW)                                \
    {                                             \
        case 16: TILE3_LAUNCH16(CT); break;       \
        case 8: TILE2_LAUNCH(8, CT); break;       \
        case 4: TILE2_LAUNCH(4, CT); break;       \
        case 2: TILE2_LAUNCH(2, CT); break;       \
        default: TILE2_LAUNCH(1, CT); break;      \
    }
        if (rank)
        {
            TILE2_NW(CellRank);
        }
        else
        {
            TILE2_NW(CellF32);
        }
#undef TILE2_NW
#undef TILE2_LAUNCH
        LAUNCHCHK(c, "k_cascade_tile");
        if ((c->cascTurns & 1) && (rc = turnEnd(c, 0, 0)))
        {
            return rc;
        }
#ifdef ACF_HIP_STAMPS
        if (a.debug & 4)
        {
            // debug only: mean cycles per phase of thread 0 of every block
            HIPCHK(c, hipStreamSynchronize(c->stream));
            std::vector<long long> st(size_t(grid.x) * 8);
            HIPCHK(c, hipMemcpy(st.data(), a.stamps, st.size() * 8, hipMemcpyDeviceToHost));
            if (gt.pooled)
            {
                double acc[6] = { 0, 0, 0, 0, 0, 0 }, n1 = 0, n2 = 0, nE = 0, accE = 0;
                long long nb = 0, nbE = 0;
                for (size_t b = 0; b < size_t(grid.x); b++)
                {
                    if (st[b * 8 + 5] > st[b * 8])
                    {
                        for (int k = 0; k < 5; k++)
                        {
                            acc[k] += double(st[b * 8 + k + 1] - st[b * 8 + k]);
                        }
                        n1 += double(st[b * 8 + 7] & 0xffff);
                        n2 += double((st[b * 8 + 7] >> 16) & 0xffff);
                        nE += double(st[b * 8 + 7] >> 32);
                        nb++;
                        if ((st[b * 8 + 7] >> 32) > 0 && st[b * 8 + 6] > st[b * 8 + 5]) // (tiles with windows in the tail queue: stage E ran)
                        {
                            accE += double(st[b * 8 + 6] - st[b * 8 + 5]);
                            nbE++;
                        }
                    }
                }
                fprintf(stderr, "[casc stamps, pooled] blocks %lld  fill %.0f  A1 (thread 0) %.0f  barrier %.0f  A2 + barrier %.0f  S %.0f cycles;  survivors per tile: A1 %.1f  A2 %.1f  S %.2f;  E %.0f cycles in %.1f %% of the tiles\n",
                    nb, acc[0] / nb, acc[1] / nb, acc[2] / nb, acc[3] / nb, acc[4] / nb, n1 / nb, n2 / nb, nE / nb, nbE ? accE / nbE : 0.0, 100.0 * nbE / std::max<long long>(nb, 1));
            }
            else
            {
                double acc[4] = { 0, 0, 0, 0 }, sub[3] = { 0, 0, 0 };
                long long nb = 0;
                for (size_t b = 0; b < size_t(grid.x); b++)
                {
                    if (st[b * 8 + 4] > st[b * 8])
                    {
                        for (int k = 0; k < 4; k++)
                        {
                            acc[k] += double(st[b * 8 + k + 1] - st[b * 8 + k]);
                        }
                        for (int k = 0; k < 3; k++)
                        {
                            sub[k] += double(st[b * 8 + 5 + k]);
                        }
                        nb++;
                    }
                }
                fprintf(stderr, "[casc stamps] blocks %lld (those with tail windows)  fill %.0f  wave 0: A + sparse pieces %.0f  barrier %.0f  E %.0f cycles   (since the fill barrier: A starts %.0f, A evaluated %.0f, compacted %.0f)\n", nb,
                    acc[0] / nb, acc[1] / nb, acc[2] / nb, acc[3] / nb, sub[0] / nb, sub[1] / nb, sub[2] / nb);
            }
            (void)hipFree(a.stamps);
        }
#endif
        if (g.b[4] < p.nTrees && cs.codeCap > 0)
        {
            // the first codeCap queue entries of every frame carry leaf codes (stage E of the tile kernel): the ordered scan
            const size_t ldsS = size_t(p.nTrees - g.b[4]) * 16;
            if ((rc = allowLds(c, reinterpret_cast<const void*>(&k_tail_scan), ldsS)))
            {
                return rc;
            }
            prof(c, "k_tail_scan");
            hipLaunchKernelGGL(k_tail_scan, dim3(nF * ((cs.codeCap + 255) / 256)), dim3(256), ldsS, c->stream, a);
            LAUNCHCHK(c, "k_tail_scan");
        }
        if (g.b[4] < p.nTrees && rank)
        {
            // queue entries without codes (beyond codeCap) from the rank cells: a wave per entry; its blocks leave at once
            // when k_tail_scan has taken the whole queue
            prof(c, "k_cascade_tail3");
            hipLaunchKernelGGL(k_cascade_tail_rank, dim3(std::max(1, 1024 / nF) * nF), dim3(64), 0, c->stream, at, (const TreeNode*)cs.d_tailNodesR);
            LAUNCHCHK(c, "k_cascade_tail_rank");
        }
        else if (g.b[4] < p.nTrees)
        {
            // queue entries without codes (beyond codeCap, or ACF_HIP_TAIL3): k_cascade_tail3; its blocks leave at once when
            // k_tail_scan has taken the whole queue
            dim3 tgrid(std::min(cs.tailBlocks, std::max(1, 512 / nF) * nF)), tblock(cs.tailWaves * 64);
            const size_t tl = (size_t(cs.tailWaves) * a.tailSlab + size_t(a.tailNodesLds)) * 4;
            prof(c, "k_cascade_tail3");
#define TAIL_LAUNCH(N)                                                                                              \
    if (a.tailNodesLds)                                                                                             \
    {                                                                                                               \
        if ((rc = allowLds(c, reinterpret_cast<const void*>(&k_cascade_tail3<N, true>), tl)))                       \
            return rc;                                                                                              \
        hipLaunchKernelGGL((k_cascade_tail3<N, true>), tgrid, tblock, tl, c->stream, a);                            \
    }                                                                                                               \
    else                                                                                                            \
    {                                                                                                               \
        if ((rc = allowLds(c, reinterpret_cast<const void*>(&k_cascade_tail3<N, false>), tl)))                      \
            return rc;                                                                                              \
        hipLaunchKernelGGL((k_cascade_tail3<N, false>), tgrid, tblock, tl, c->stream, a);                           \
    }
            switch (cs.tailWaves)
            {
                case 4:
                    TAIL_LAUNCH(4);
                    break;
                case 2:
                    TAIL_LAUNCH(2);
                    break;
                default:
                    TAIL_LAUNCH(1);
                    break;
            }
#undef TAIL_LAUNCH
            LAUNCHCHK(c, "k_cascade_tail3");
        }
    }
    return ACF_HIP_OK;
}

static const size_t kNmsLds = size_t(NMS_CAP) * (8 + 16 + 4 + 1);

static void fillNmsArgs(NmsArgs& a, const acf_hip_nms_params& q)
{
    a.greedy = q.type == 2;
    a.ovrUnion = q.ovrDnmUnion != 0;
    a.doPrune = q.prune != 0;
    a.maxCount = q.maxCount;
    a.overlap = q.overlap;
    a.thr = q.thr;
    a.pruneRatio = q.pruneRatio;
}

// bbNms + prune of every frame's detections (k_nms, one workgroup per frame)
static int launchNms(acf_hip_ctx* c, int nF)
{
    int rc;
    if (!c->d_nmsKeep)
    {
        if ((rc = devAlloc(c, &c->d_nmsKeep, size_t(c->maxBatch) * NMS_CAP)) || (rc = devAlloc(c, &c->d_nmsN, size_t(c->maxBatch))) ||
            (rc = devAlloc(c, &c->d_nmsCounts, size_t(c->maxBatch))) || (rc = devAlloc(c, &c->d_nmsDets, size_t(c->maxBatch) * c->maxHits)))
        {
            return rc;
        }
    }
    NmsArgs a{};
    a.dets = c->cs.d_dets;
    a.counts = c->cs.d_counts;
    a.maxHits = c->maxHits;
    fillNmsArgs(a, c->nms);
    a.keep = c->d_nmsKeep;
    a.nKeep = c->d_nmsN;
    a.outDets = c->d_nmsDets;
    a.outCounts = c->d_nmsCounts;
    if ((rc = allowLds(c, reinterpret_cast<const void*>(&k_nms), kNmsLds)))
    {
        return rc;
    }
    prof(c, "k_nms");
    hipLaunchKernelGGL(k_nms, dim3(nF), dim3(1024), kNmsLds, c->stream, a);
    LAUNCHCHK(c, "k_nms");
    return ACF_HIP_OK;
}

static inline bool nmsActive(const acf_hip_ctx* c)
{
    return c->nmsOn && c->nms.type != 0 && c->d_nmsDets;
}

static int runCascade(acf_hip_ctx* c, const float* pyr, int64_t pyr_fs, const BoxLevel* d_box, int nF, int nChns)
{
    const acf_hip_params& p = c->p;
    const CascLevel* d_levels = c->cs.d_cascLevels;
    const int32_t* d_blockLevel = c->cs.d_blockLevel;
    const int blocksPerFrame = c->cs.blocksPerFrame;
    const uint32_t* d_cidAll = c->cs.d_cidAll;
    const CascNode2* d_nodes2 = c->cs.d_nodes2;
    prof(c, "k_cascade");
    const bool rankPath = c->cs.useTiles && !c->noTiles && c->cs.useRank && !c->noRank;
    if (pyr == c->d_pyr && !c->floatPyramid && !(rankPath && c->ranksValid))
    {
        // (options changed between acf_hip_pyramid and acf_hip_detect: the float cells this path reads were never written)
        return fail(c, ACF_HIP_E_INVALID, "detect: the float pyramid of this batch was not written (keep_pyramid = 0) and the selected cascade reads floats");
    }
    if (!c->countersZeroed)
    {
        HIPCHK(c, hipMemsetAsync(c->cs.d_counts, 0, sizeof(int32_t) * nF, c->stream));
    }
    if (tiledCascadeSelected(c))
    {
        int rc = runCascadeTiled(c, pyr, pyr_fs, nF, nChns);
        if (rc)
        {
            return rc;
        }
    }
    else if (blocksPerFrame > 0)
    {
        // stage boundaries (see kernels.hip.h): [0,16) [16,32) [32,128) [128,nTrees)
        std::vector<int> bounds;
        for (int b : { 16, 32, 128 })
        {
            if (b < p.nTrees)
            {
                bounds.push_back(b);
            }
        }
        bounds.push_back(p.nTrees);
        const int nStages = int(bounds.size());
        HIPCHK(c, hipMemsetAsync(c->cs.d_qcounts, 0, sizeof(int32_t) * size_t(nStages) * c->maxBatch, c->stream));
        CascArgs a{};
        a.pyr = pyr;
        a.pyr_fs = pyr_fs;
        a.levels = d_levels;
        a.blockLevel = d_blockLevel;
        a.blocksPerFrame = blocksPerFrame;
        a.nFrames = nF;
        a.mH = p.modelDsPad_h / p.shrink;
        a.mW = p.modelDsPad_w / p.shrink;
        a.nChns = nChns;
        a.fids = c->cs.d_fids;
        a.nTrees = p.nTrees;
        a.nTreeNodes = p.nTreeNodes;
        a.treeDepth = p.treeDepth;
        a.stride = c->cs.dedupQ > 1 ? p.shrink : p.stride; // (the grid of distinct offsets: its windows are one cell apart)
        a.shrink = p.shrink;
        a.cascThr = float(p.cascThr); // DetectionParams::cascThr is a float (acfDetect1.cpp:63,323)
        a.cidAll = d_cidAll;
        a.thrs = c->cs.d_thrs;
        a.hs = c->cs.d_hs;
        a.child = c->cs.d_child;
        a.nodes2 = d_nodes2;
        a.qcap = c->cs.qcap;
        a.hits = c->cs.d_hits;
        a.counts = c->cs.d_counts;
        a.maxHits = c->maxHits;
        const int mode = p.treeDepth == 2 ? 2 : (p.treeDepth > 0 ? 1 : 0);
        // later stages see a shrinking survivor set; grid-stride loops cover any count
        const int qGrid[3] = { std::max(1, blocksPerFrame / 2), std::max(1, blocksPerFrame / 8), std::max(1, std::min(blocksPerFrame, 64)) };
        int firstStage = 0;
        bool pooledTail = false; // k_cascade_tile3D has written the tail's leaf codes: no k_tail_codesD
        if (c->cs.useTileD && !c->noTiles)
        {
            // depths 1, 3, 4: trees [0, t1D) of every window from LDS tiles (k_cascade_tileD) instead of the first stages'
            // per-lane gathers from the pyramid; its survivors enter the queue of the stage that ends at t1D
            const auto& cs = c->cs;
            int sD = -1;
            const bool rankD = cs.useRankD && !c->noRank && pyr == c->d_pyr;
            const bool pooledRun = rankD || cs.geomD.pooled;
            const int endD = rankD ? cs.geomDR.b[4] : (cs.geomD.pooled ? cs.geomD.b[4] : cs.t1D); // last tree the tile kernel evaluates
            for (int i = 0; i < nStages; i++)
            {
                if (bounds[size_t(i)] == endD)
                {
                    sD = i;
                }
            }
            if (sD >= 0)
            {
                TileDArgs at{};
                at.pyr = pyr;
                at.pyr_fs = pyr_fs;
                at.levels = d_levels;
                at.tiles = cs.d_tilesD;
                at.nTiles = cs.nTilesD;
                at.nFrames = nF;
                at.nChns = nChns;
                at.nBatches = cs.t1D / cs.tbD;
                at.g = cs.geomD;
                at.nodesD = cs.d_nodesD;
                at.cascThr = float(p.cascThr);
                at.last = sD == nStages - 1;
                at.qout = cs.d_queue[sD & 1];
                at.qoutCount = cs.d_qcounts + size_t(sD) * c->maxBatch;
                at.qcap = cs.qcap;
                at.hits = cs.d_hits;
                at.counts = cs.d_counts;
                at.maxHits = c->maxHits;
                int rcl = 0;
                const int64_t total = int64_t(rankD ? cs.nTilesDR : at.nTiles) * nF;
                const int64_t perX = (total + 7) / 8;
                if (rankD && !c->ranksValid)
                {
                    // the float pyramid -> threshold-rank cells (levels whose kernels did not emit them)
                    const size_t ldsR = size_t(cs.rankMaxRec) * sizeof(RankRec);
                    if ((rcl = allowLds(c, reinterpret_cast<const void*>(&k_rank), ldsR)))
                    {
                        return rcl;
                    }
                    prof(c, "k_rank");
                    hipLaunchKernelGGL(k_rank, dim3(cdiv(cs.rankMaxWP, RANK_CHUNK_COLS), int(c->plan.levels.size()) * nChns, nF), dim3(256), ldsR, c->stream, pyr, pyr_fs,
                        cs.d_pyrR, cs.pyrRCells, (const RankJob*)cs.d_rankJobs, nChns, (const RankChan*)cs.d_rankChan, (const RankRec*)cs.d_rankRec);
                    LAUNCHCHK(c, "k_rank");
                    c->ranksValid = true;
                    prof(c, "k_cascade");
                }
                if (pooledRun)
                {
                    // k_cascade_tile3D: everything up to tree b[4]; the tail's codes come from its stage E
                    at.tileOff = cs.d_tileOffD;
                    at.thrs = c->cs.d_thrs;
                    if (rankD)
                    {
                        at.g = cs.geomDR;
                        at.tiles = cs.d_tilesDR;
                        at.nTiles = cs.nTilesDR;
                        at.nodesD = cs.d_nodesDR;
                        at.tileOff = cs.d_tileOffDR;
                        at.thrs = reinterpret_cast<const float*>(cs.d_thrsRankD);
                        at.pyrR = cs.d_pyrR;
                        at.pyrR_fs = cs.pyrRCells;
                    }
                    at.hs = c->cs.d_hs;
                    at.nTrees = p.nTrees;
                    at.nTreeNodes = p.nTreeNodes;
                    pooledTail = !at.last && cs.codeCapD > 0;
                    at.codes = pooledTail ? cs.d_codesD : nullptr;
                    at.codeCap = pooledTail ? cs.codeCapD : 0;
                    at.codePitch = cs.codePitchD;
                    const size_t nwin = size_t(at.g.NW) * 64;
                    const size_t lds = size_t(128) * 4 * (size_t(1) << p.treeDepth) + size_t(at.g.tileFloats) * (rankD ? 2 : 4) +
                        ((std::max(nwin * 8, size_t(at.g.passW) * size_t(at.g.pitchC)) + 15) / 16 * 16) + nwin * 8;
                    const int64_t resident = int64_t(c->numCus) * std::max<int64_t>(1, int64_t(c->ldsPerCu) / int64_t((lds + 1279) / 1280 * 1280));
                    const int64_t gridP = c->tilePersist > 1 ? int64_t(c->tilePersist) : resident;
                    const bool persist = c->tilePersist > 0 && (gridP + 7) / 8 * 8 < perX * 8;
                    // (the staged path's counters [stage][frame] start at d_qcounts; the tile counters sit behind them)
                    at.tileNext = persist ? cs.d_qcounts + size_t(8) * c->maxBatch : nullptr;
                    if (persist)
                    {
                        HIPCHK(c, hipMemsetAsync(at.tileNext, 0, sizeof(int32_t) * 8, c->stream));
                    }
                    dim3 grid((unsigned int)(persist ? (gridP + 7) / 8 * 8 : perX * 8)), block(at.g.NW * 64);
                    prof(c, "k_cascade_tile");
#define TILE3D_LAUNCH(N, DD, TT)                                                                                   \
    if (rankD)                                                                                                     \
    {                                                                                                              \
        if ((rcl = allowLds(c, reinterpret_cast<const void*>(&k_cascade_tile3D<N, DD, TT, CellRank>), lds)))       \
            return rcl;                                                                                            \
        hipLaunchKernelGGL((k_cascade_tile3D<N, DD, TT, CellRank>), grid, block, lds, c->stream, at);              \
    }                                                                                                              \
    else                                                                                                           \
    {                                                                                                              \
        if ((rcl = allowLds(c, reinterpret_cast<const void*>(&k_cascade_tile3D<N, DD, TT, CellF32>), lds)))        \
            return rcl;                                                                                            \
        hipLaunchKernelGGL((k_cascade_tile3D<N, DD, TT, CellF32>), grid, block, lds, c->stream, at);               \
    }
#define TILE3D_DEPTH(N)                                    \
    switch (p.treeDepth)                                   \
    {                                                      \
        case 1: TILE3D_LAUNCH(N, 1, 4); break;             \
        case 3: TILE3D_LAUNCH(N, 3, 2); break;             \
        default: TILE3D_LAUNCH(N, 4, 1); break;            \
    }
                    if (at.g.NW == 8)
                    {
                        TILE3D_DEPTH(8)
                    }
                    else
                    {
                        TILE3D_DEPTH(4)
                    }
#undef TILE3D_DEPTH
#undef TILE3D_LAUNCH
                    LAUNCHCHK(c, "k_cascade_tile3D");
                    prof(c, "k_cascade");
                }
                else
                {
                const size_t lds = size_t(at.g.tileFloats) * 4;
                dim3 grid((unsigned int)(perX * 8)), block(at.g.NW * 64);
#define TILED_LAUNCH(N, DD, TT)                                                                          \
    {                                                                                                    \
        if ((rcl = allowLds(c, reinterpret_cast<const void*>(&k_cascade_tileD<N, DD, TT>), lds)))        \
            return rcl;                                                                                  \
        hipLaunchKernelGGL((k_cascade_tileD<N, DD, TT>), grid, block, lds, c->stream, at);               \
    }
#define TILED_DEPTH(N)                                    \
    switch (p.treeDepth)                                  \
    {                                                     \
        case 1: TILED_LAUNCH(N, 1, 4); break;             \
        case 3: TILED_LAUNCH(N, 3, 2); break;             \
        default: TILED_LAUNCH(N, 4, 1); break;            \
    }
                if (at.g.NW == 8)
                {
                    TILED_DEPTH(8)
                }
                else
                {
                    TILED_DEPTH(4)
                }
#undef TILED_DEPTH
#undef TILED_LAUNCH
                LAUNCHCHK(c, "k_cascade_tileD");
                }
                firstStage = sD + 1;
            }
        }
        for (int sidx = firstStage; sidx < nStages; sidx++)
        {
            a.t0 = sidx == 0 ? 0 : bounds[sidx - 1];
            a.t1 = bounds[sidx];
            a.last = sidx == nStages - 1;
            a.qin = sidx > 0 ? c->cs.d_queue[(sidx - 1) & 1] : nullptr;
            a.qinCount = sidx > 0 ? c->cs.d_qcounts + size_t(sidx - 1) * c->maxBatch : nullptr;
            a.qout = c->cs.d_queue[sidx & 1];
            a.qoutCount = c->cs.d_qcounts + size_t(sidx) * c->maxBatch;
            dim3 block(256);
            const size_t winBytes = sizeof(float) * size_t(nChns) * a.mH * a.mW;
            const bool tail = a.last && sidx > 0 && a.t0 >= 128 && winBytes <= 64 * 1024;
            if (sidx == 0)
            {
                dim3 grid(blocksPerFrame * nF);
                if (mode == 2)
                {
                    hipLaunchKernelGGL(k_cascade_first<2>, grid, block, 0, c->stream, a);
                }
                else if (mode == 1)
                {
                    hipLaunchKernelGGL(k_cascade_first<1>, grid, block, 0, c->stream, a);
                }
                else
                {
                    hipLaunchKernelGGL(k_cascade_first<0>, grid, block, 0, c->stream, a);
                }
            }
            else if (tail)
            {
                // one wave per surviving window; enough waves per frame to fill the chip
                dim3 grid(std::max(1, 8192 / nF) * nF);
                if (mode == 1 && c->cs.codeCapD > 0 && a.t0 == 128)
                {
                    // leaf codes of the first codeCapD entries, then their ordered sums with lanes = windows; k_cascade_tail
                    // (below) takes the entries beyond and leaves at once when there are none
                    a.codes = c->cs.d_codesD;
                    a.codeCap = c->cs.codeCapD;
                    a.codePitch = c->cs.codePitchD;
                    if (!pooledTail)
                    {
                        hipLaunchKernelGGL(k_tail_codesD, grid, dim3(64), winBytes, c->stream, a);
                        LAUNCHCHK(c, "k_tail_codesD");
                    }
                    const int nT = a.t1 - a.t0, NL = 1 << p.treeDepth;
                    const size_t ldsS = size_t((nT + 15) / 16 * 16) * NL * sizeof(float);
                    dim3 gridS(nF * ((a.codeCap + 255) / 256));
                    int rcl = 0;
#define TSD_LAUNCH(DD)                                                                              \
    {                                                                                               \
        if ((rcl = allowLds(c, reinterpret_cast<const void*>(&k_tail_scanD<DD>), ldsS)))            \
            return rcl;                                                                             \
        hipLaunchKernelGGL(k_tail_scanD<DD>, gridS, dim3(256), ldsS, c->stream, a);                 \
    }
                    switch (p.treeDepth)
                    {
                        case 1: TSD_LAUNCH(1); break;
                        case 3: TSD_LAUNCH(3); break;
                        case 4: TSD_LAUNCH(4); break;
                        default: TSD_LAUNCH(2); break;
                    }
#undef TSD_LAUNCH
                    LAUNCHCHK(c, "k_tail_scanD");
                    a.qskip = a.codeCap;
                }
                if (mode == 2)
                {
                    hipLaunchKernelGGL(k_cascade_tail<2>, grid, dim3(64), winBytes, c->stream, a);
                }
                else if (mode == 1)
                {
                    hipLaunchKernelGGL(k_cascade_tail<1>, grid, dim3(64), winBytes, c->stream, a);
                }
                else
                {
                    hipLaunchKernelGGL(k_cascade_tail<0>, grid, dim3(64), winBytes, c->stream, a);
                }
            }
            else
            {
                dim3 grid(qGrid[std::min(sidx - 1, 2)] * nF);
                if (mode == 2)
                {
                    hipLaunchKernelGGL(k_cascade_queue<2>, grid, block, 0, c->stream, a);
                }
                else if (mode == 1)
                {
                    hipLaunchKernelGGL(k_cascade_queue<1>, grid, block, 0, c->stream, a);
                }
                else
                {
                    hipLaunchKernelGGL(k_cascade_queue<0>, grid, block, 0, c->stream, a);
                }
            }
            LAUNCHCHK(c, "k_cascade stage");
        }
    }
    // shift = (modelDsPad - modelDs)/2 - pad (ACF.cpp:275; cv::Size integer arithmetic)
    const int shift_h = (p.modelDsPad_h - p.modelDs_h) / 2 - p.pad_h;
    const int shift_w = (p.modelDsPad_w - p.modelDs_w) / 2 - p.pad_w;
    const acf_hip_hit* hitsForSort = c->cs.d_hits;
    if (c->cs.dedupQ > 1)
    {
        if (!c->cs.d_hitsX || !c->cs.d_realWin)
        {
            return fail(c, ACF_HIP_E_INVALID, "detect: no buffer for the windows that share an offset (stride < shrink)");
        }
        hipLaunchKernelGGL(k_expand_hits, dim3(nF), dim3(256), 0, c->stream, (const acf_hip_hit*)c->cs.d_hits, c->cs.d_counts, c->cs.d_hitsX, c->maxHits,
            (const int2*)c->cs.d_realWin, c->cs.dedupQ);
        LAUNCHCHK(c, "k_expand_hits");
        hitsForSort = c->cs.d_hitsX;
    }
    prof(c, "k_sort_map");
    hipLaunchKernelGGL(k_sort_map, dim3(SM_BLOCKS, nF), dim3(256), 0, c->stream, hitsForSort, (const int32_t*)c->cs.d_counts, c->maxHits,
        d_box, p.stride, shift_h, shift_w, c->cs.d_sorted, c->cs.d_dets);
    LAUNCHCHK(c, "k_sort_map");
    if (c->nmsOn && c->nms.type != 0)
    {
        int rc = launchNms(c, nF);
        if (rc)
        {
            return rc;
        }
    }
    prof(c, "(end)");
    c->countsFetched = false;
    return ACF_HIP_OK;
}
