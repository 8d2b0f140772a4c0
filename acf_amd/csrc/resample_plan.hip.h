// resample_plan.hip.h — part of acf_hip.hip (included there and nowhere else): plans of the LDS-tiled image resamples.
// stripPlan / launchStrip: k_resample_strip's row tiles and steps (the march over strips of output columns, one or two outputs
// from one pass over their common source); resampleTilePlan: the tile geometry of a down-sampling descriptor (k_ldcf_tile).
#pragma once

// k_resample_strip's plan (kernels.hip.h): row tiles of `yt` output rows of A (yt / 2 of B), steps of RS_XO output columns of A
// (RS_XO / 2 of B); per row tile / per step the union of the two outputs' source ranges, appended to the int arena.
static StripPlan stripPlan(const ResampleDesc& da, const ResampleDesc* db, TableArena& arena)
{
    StripPlan sp;
    auto down = [](const ResampleDesc& d) { return (d.xmode == RS_DOWN || d.xmode == RS_EXACT) && (d.ymode == RS_DOWN || d.ymode == RS_EXACT); };
    if (!down(da) || (db && !down(*db)) || da.ha % 4 || da.ha < 8 || da.src_frame_stride % 4 || da.src_off % 4 ||
        (db && (db->ha != da.ha || db->wa != da.wa || db->nplanes != da.nplanes || db->src_frame_stride != da.src_frame_stride || db->src_off != da.src_off)))
    {
        return sp;
    }
    const int32_t* it = arena.ints.data();
    auto rowRange = [&](const ResampleDesc& d, int yb0, int yb1, int& lo, int& hi) {
        if (d.ymode == RS_EXACT)
        {
            lo = d.yk * yb0;
            hi = d.yk * (yb1 - 1) + d.yk - 1;
        }
        else
        {
            lo = it[d.y_src + it[d.y_start + yb0]];
            hi = std::max(it[d.y_src + it[d.y_start + yb1 - 1]] + d.ybd0 - 1, it[d.y_src + it[d.y_start + yb1] - 1]);
        }
    };
    auto colRange = [&](const ResampleDesc& d, int xb0, int xb1, int& lo, int& hi) {
        lo = it[d.x_col + 8 * xb0];
        hi = lo;
        for (int x = xb0; x < xb1; x++)
        {
            lo = std::min(lo, it[d.x_col + 8 * x]);
            hi = std::max(hi, it[d.x_col + 8 * x] + it[d.x_col + 8 * x + 1] - 1);
        }
    };
    // the y pass's slow form (more than four taps) keeps a row's taps in registers: at most 8, each within 15 rows of the first
    auto slowOk = [&](const ResampleDesc& d) {
        if (!(d.ymode == RS_DOWN && d.ybd0 > 4))
        {
            return true;
        }
        for (int yb = 0; yb < d.hb; yb++)
        {
            const int q0 = it[d.y_start + yb], q1 = it[d.y_start + yb + 1];
            if (q1 - q0 > 8)
            {
                return false;
            }
            for (int q = q0; q < q1; q++)
            {
                const int off = it[d.y_src + q] - it[d.y_src + q0];
                if (off < 0 || off > 15)
                {
                    return false;
                }
            }
        }
        return true;
    };
    if (!slowOk(da) || (db && !slowOk(*db)))
    {
        return sp;
    }
    auto fourTaps = [&](const ResampleDesc& d) {
        for (int x = 0; x < d.wb; x++)
        {
            if (it[d.x_col + 8 * x + 1] > 4)
            {
                return false;
            }
        }
        return true;
    };
    if (!fourTaps(da) || (db && !fourTaps(*db)))
    {
        return sp; // (x ratios above 4: the generic kernels)
    }
    const int nSteps = cdiv(da.wb, RS_XO), nStepsB = db ? cdiv(db->wb, RS_XO / 2) : 0;
    if (nStepsB > nSteps)
    {
        return sp;
    }
    std::vector<int32_t> tx;
    int maxC = 0;
    for (int st = 0; st < nSteps; st++)
    {
        int lo, hi;
        colRange(da, st * RS_XO, std::min((st + 1) * RS_XO, da.wb), lo, hi);
        if (st < nStepsB)
        {
            int lob, hib;
            colRange(*db, st * (RS_XO / 2), std::min((st + 1) * (RS_XO / 2), db->wb), lob, hib);
            lo = std::min(lo, lob);
            hi = std::max(hi, hib);
        }
        tx.push_back(lo);
        tx.push_back(hi - lo + 1);
        maxC = std::max(maxC, hi - lo + 1);
    }
    for (int nty = 1; nty <= 64; nty++)
    {
        const int yt = (cdiv(da.hb, nty) + 1) / 2 * 2;
        const int ntyA = cdiv(da.hb, yt), ntyB = db ? cdiv(db->hb, yt / 2) : 0;
        if (ntyA != nty || ntyB > nty)
        {
            continue;
        }
        std::vector<int32_t> tyv;
        int maxR = 0;
        for (int t = 0; t < nty; t++)
        {
            int lo, hi;
            rowRange(da, t * yt, std::min((t + 1) * yt, da.hb), lo, hi);
            if (t < ntyB)
            {
                int lob, hib;
                rowRange(*db, t * (yt / 2), std::min((t + 1) * (yt / 2), db->hb), lob, hib);
                lo = std::min(lo, lob);
                hi = std::max(hi, hib);
            }
            lo = lo / 4 * 4;
            tyv.push_back(lo);
            tyv.push_back(hi - lo + 1);
            maxR = std::max(maxR, hi - lo + 1);
        }
        const int rowsP = (maxR + 3) / 4 * 4;
        const int64_t items = int64_t(RS_XO) * std::min(yt, da.hb) + (db ? int64_t(RS_XO / 2) * (yt / 2) : 0);
        const bool slowA = da.ymode == RS_DOWN && da.ybd0 > 4, slowB = db && db->ymode == RS_DOWN && db->ybd0 > 4;
        const int slowRows = (slowA ? yt : 0) + (slowB ? yt / 2 : 0);
        // a tile's requests: whole rounds of RS_NT chunks of 16 bytes (the kernel issues a fixed number per wave)
        const int fillRounds = cdiv(int64_t(maxC) * (rowsP / 4), RS_NT);
        const int tileFloats = fillRounds * RS_NT * 4;
        const size_t lds = (size_t(2) * tileFloats + size_t(RS_XO + RS_XO / 2) * RS_CP) * sizeof(float) + size_t(2) * RS_REC * 4 +
            size_t(std::max(slowRows, 1)) * 8 * sizeof(float);
        if (rowsP > 64 * RS_KCH || fillRounds > 4 || items > int64_t(RS_ITEMS) * RS_NT || lds + size_t(8) * (nSteps + 2) > size_t(52) * 1024)
        {
            continue;
        }
        const uint32_t cps = uint32_t(rowsP / 4);
        const uint32_t magic = uint32_t(((uint64_t(1) << 32) + cps - 1) / cps);
        bool ok = cps > 1;
        for (uint32_t q = 0; q < uint32_t(maxC) * cps && ok; q++)
        {
            ok = uint32_t((uint64_t(q) * magic) >> 32) == q / cps;
        }
        if (!ok)
        {
            continue;
        }
        sp.ok = true;
        sp.yt = yt;
        sp.nty = nty;
        sp.ntyB = ntyB;
        sp.nSteps = nSteps;
        sp.nStepsB = nStepsB;
        sp.rowsP = rowsP;
        sp.maxCols = maxC;
        sp.magic = magic;
        sp.lds = lds;
        sp.slowRows = slowRows;
        sp.fillRounds = fillRounds;
        sp.tileFloats = tileFloats;
        sp.tileY = int(arena.ints.size());
        arena.ints.insert(arena.ints.end(), tyv.begin(), tyv.end());
        sp.tileX = int(arena.ints.size());
        arena.ints.insert(arena.ints.end(), tx.begin(), tx.end());
        return sp;
    }
    return sp;
}

// k_resample_strip for one output (descB < 0) or two outputs of one source
static void launchStrip(acf_hip_ctx* c, const StripPlan& sp, const ResampleDesc* d_descs, int descA, int descB, int nplanes, const float* src, float* dstA,
    float* dstB, const int32_t* d_it, const float* d_ft, int nF)
{
    const bool pair = descB >= 0;
    StripArgs sa{};
    sa.src = src;
    sa.dstA = dstA;
    sa.dstB = dstB;
    sa.descs = d_descs;
    sa.it = d_it;
    sa.ft = d_ft;
    sa.descA = descA;
    sa.descB = descB;
    sa.yt = sp.yt;
    sa.nty = sp.nty;
    sa.nSteps = sp.nSteps;
    sa.tileY = sp.tileY;
    sa.tileX = sp.tileX;
    sa.rowsP = sp.rowsP;
    sa.maxCols = sp.maxCols;
    sa.ntyB = sp.ntyB;
    sa.nStepsB = sp.nStepsB;
    sa.cpsMagic = sp.magic;
    sa.slowRows = sp.slowRows;
    sa.fillRounds = sp.fillRounds;
    sa.tileFloats = sp.tileFloats;
    sa.dump = c->d_dump;
    // column segments (a resample has no history along x: segments are free), each at least 8 steps long: the count that
    // minimises (rounds of workgroups over what the device holds at once) x (steps per workgroup)
    const size_t ldsS = sp.lds + size_t(8) * (sp.nSteps + 2);
    const int64_t wgs = int64_t(nplanes) * sp.nty * nF;
    const int64_t resident = int64_t(c->numCus) * std::max<int64_t>(1, std::min<int64_t>(4, int64_t(c->ldsPerCu) / int64_t((ldsS + 1279) / 1280 * 1280)));
    int64_t bestCost = -1;
    sa.nSplit = 1;
    for (int n = 1; n <= std::max(1, sp.nSteps / 8); n++)
    {
        const int64_t cost = ((wgs * n + resident - 1) / resident) * (cdiv(sp.nSteps, n) + 2);
        if (bestCost < 0 || cost < bestCost)
        {
            bestCost = cost;
            sa.nSplit = n;
        }
    }
    const dim3 sgrid(nplanes * sp.nty * sa.nSplit, 1, nF);
    const bool slow = sp.slowRows > 0;
    if (pair && slow)
    {
        hipLaunchKernelGGL((k_resample_strip<true, true>), sgrid, dim3(RS_NT), ldsS, c->stream, sa);
    }
    else if (pair)
    {
        hipLaunchKernelGGL((k_resample_strip<true, false>), sgrid, dim3(RS_NT), ldsS, c->stream, sa);
    }
    else if (slow)
    {
        hipLaunchKernelGGL((k_resample_strip<false, true>), sgrid, dim3(RS_NT), ldsS, c->stream, sa);
    }
    else
    {
        hipLaunchKernelGGL((k_resample_strip<false, false>), sgrid, dim3(RS_NT), ldsS, c->stream, sa);
    }
}

// Tiling of a down-sampling descriptor for the passes on LDS tiles (k_ldcf_tile): output columns per tile (the largest of 32/16/8 whose
// source tile + x-pass buffer fit 64 KB of LDS), the largest source tile, and the per-tile source ranges appended to the
// int arena ({rowLo,rowHi} per row tile at tile_y, {colLo,colHi} per column tile at tile_x).  rows == 0: not eligible.
// (yo: output rows per tile — RT_YO for the resample kernels, k_ldcf_tile chooses its own; forceXo may be any column count)
static ResampleTiling resampleTilePlan(const ResampleDesc& dd, TableArena& arena, int forceXo = 0, int64_t ldsBudget = int64_t(64) * 1024, int yo = RT_YO,
    int maxRows = 1 << 30, int maxCols = 1 << 30)
{
    ResampleTiling tl;
    if (!((dd.xmode == RS_DOWN || dd.xmode == RS_EXACT) && (dd.ymode == RS_DOWN || dd.ymode == RS_EXACT)))
    {
        return tl;
    }
    std::vector<int32_t> ty;
    int maxR = 0;
    {
        const int32_t* it = arena.ints.data();
        for (int yb0 = 0; yb0 < dd.hb; yb0 += yo)
        {
            const int yb1 = std::min(yb0 + yo, dd.hb);
            int lo, hi;
            if (dd.ymode == RS_EXACT)
            {
                lo = dd.yk * yb0;
                hi = dd.yk * (yb1 - 1) + dd.yk - 1;
            }
            else
            {
                lo = it[dd.y_src + it[dd.y_start + yb0]];
                hi = std::max(it[dd.y_src + it[dd.y_start + yb1 - 1]] + dd.ybd0 - 1, it[dd.y_src + it[dd.y_start + yb1] - 1]);
            }
            ty.push_back(lo);
            ty.push_back(hi);
            maxR = std::max(maxR, hi - lo + 1);
        }
    }
    for (int xo : { forceXo ? forceXo : 32, 16, 8 })
    {
        if (forceXo && xo != forceXo)
        {
            continue;
        }
        std::vector<int32_t> tx;
        int maxC = 0;
        const int32_t* it = arena.ints.data();
        for (int xb0 = 0; xb0 < dd.wb; xb0 += xo)
        {
            const int xb1 = std::min(xb0 + xo, dd.wb);
            const int lo = it[dd.x_col + 8 * xb0], hi = it[dd.x_col + 8 * (xb1 - 1)] + it[dd.x_col + 8 * (xb1 - 1) + 1] - 1;
            tx.push_back(lo);
            tx.push_back(hi);
            maxC = std::max(maxC, hi - lo + 1);
        }
        if (maxR > 0 && maxC > 0 && (int64_t(maxC) + xo) * maxR * 4 <= ldsBudget && maxR <= maxRows && maxC <= maxCols)
        {
            tl.rows = maxR;
            tl.cols = maxC;
            tl.xo = xo;
            tl.tile_y = int(arena.ints.size());
            arena.ints.insert(arena.ints.end(), ty.begin(), ty.end());
            tl.tile_x = int(arena.ints.size());
            arena.ints.insert(arena.ints.end(), tx.begin(), tx.end());
            return tl;
        }
    }
    return tl;
}
