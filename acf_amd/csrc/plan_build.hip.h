// plan_build.hip.h — part of acf_hip.hip (included there and nowhere else): acf_hip_plan's work, one function per stage of the path the
// plan serves.  Everything a run needs — buffers for max_batch frames, resampling tables, job lists, cascade tables — is built here
// once; nothing is allocated on the hot path.  PlanBuild holds what the stages share (the table arena, the job lists that are
// uploaded together at the end).
//
//   subBatchContexts()   option "streams" > 1: the plan's geometry stays here, every device buffer lives in a child context
//   colourBuffer()       rgbConvert's output (chnsPyramid.cpp:230-263)
//   realScales()         the real scales' images, smoothed planes, M / O / U / S (chnsPyramid.cpp:297-338)
//   imageStrips()        k_resample_strip's plans for the down-sampling image resamples
//   approxLevels()       resampling descriptors of the approximated levels (chnsPyramid.cpp:385-397)
//   finalJobs()          final smoothing + padding jobs, the rank pyramid's layout (chnsPyramid.cpp:399-435)
//   levelJobs()          the fused level kernels' job lists
//   ldcf()               the LDCF post-stage's levels, tiles and buffers (acf_hip_params::ldcfK)
//   uploadAndScratch()   tables to the device; raw channels, pyramid, the segment kernels' hand-over states
//   cascade()            cascade tables, box mapping, hit / queue buffers
#pragma once

namespace
{
struct PlanBuild
{
    acf_hip_ctx* const c;
    const acf_hip_params& p;
    Plan& pl;
    const int H, W, d_in, B, max_hits, shrink, d;
    const int64_t np0;
    TableArena arena;
    std::vector<SmoothJob> realJobs, finalJobList;
    std::vector<int64_t> rankOffs;
    std::vector<PadJob> padJobs, padJobsR;
    int py = 0, px = 0; // padding in cells

    PlanBuild(acf_hip_ctx* ctx, int H_, int W_, int d_in_, int max_batch, int max_hits_)
        : c(ctx), p(ctx->p), pl(ctx->plan), H(H_), W(W_), d_in(d_in_), B(max_batch), max_hits(max_hits_), shrink(ctx->p.shrink), d(ctx->plan.d), np0(int64_t(H_) * W_)
    {
    }
    int colourBuffer();
    int realScales();
    int imageStrips();
    int approxLevels();
    int finalJobs();
    int levelJobs();
    int ldcf();
    int uploadAndScratch();
    int cascade();
};

int PlanBuild::colourBuffer()
{
    int rc = ACF_HIP_OK;
    (void)rc;
    // colour conversion buffer (chnsPyramid.cpp:230-263)
    const bool passthrough = (d_in >= 3) && (p.colorSpace == ACF_HIP_CS_ORIG || p.colorSpace == ACF_HIP_CS_RGB || (p.isLuv && p.colorSpace == ACF_HIP_CS_LUV));
    if (!passthrough)
    {
        if ((rc = devAlloc(c, &c->d_color, size_t(B) * d * np0)))
        {
            return rc;
        }
    }

    return ACF_HIP_OK;
}

int PlanBuild::realScales()
{
    int rc = ACF_HIP_OK;
    (void)rc;
    // real scales: mirror the reference's shallow-copy bookkeeping (chnsPyramid.cpp:297-338)
    c->h_descs.clear();
    c->real.clear();
    int curH = H, curW = W;
    for (size_t k = 0; k < pl.real.size(); k++)
    {
        RealScale rs;
        rs.level = pl.real[k];
        rs.h = pl.real_h[k];
        rs.w = pl.real_w[k];
        const double s = pl.levels[rs.level].scale;
        const bool same = (H == rs.h && W == rs.w); // sz == sz1 (:303), compared against the ORIGINAL size
        if (same && (curH != H || curW != W))
        {
            return fail(c, ACF_HIP_E_UNSUPPORTED, "plan: scale order");
        }
        rs.resampled = !same;
        rs.src_h = curH;
        rs.src_w = curW;
        const int64_t np = int64_t(rs.h) * rs.w;
        if (rs.resampled)
        {
            ResampleDesc dd;
            if ((rc = buildResample(curH, curW, rs.h, rs.w, dd, arena)))
            {
                return fail(c, rc, "plan: degenerate resample geometry");
            }
            const double one[3] = { 1.0, 1.0, 1.0 };
            setResampleGain(dd, one, d, d);
            dd.nplanes = d;
            dd.src_off = 0;
            dd.dst_off = 0;
            dd.src_frame_stride = int64_t(d) * curH * curW;
            dd.dst_frame_stride = int64_t(d) * np;
            rs.descIndex = int(c->h_descs.size());
            c->h_descs.push_back(dd);
            if ((rc = devAlloc(c, &rs.img, size_t(B) * d * np)))
            {
                return rc;
            }
        }
        const bool halfCond = (s == 0.5) && ((p.nApprox > 0) || (p.nPerOct == 1)); // :313-316
        rs.adoptAsI = same || halfCond;
        if (rs.adoptAsI)
        {
            curH = rs.h;
            curW = rs.w;
        }
        if ((rc = devAlloc(c, &rs.sm, size_t(B) * d * np)))
        {
            return rc;
        }
        if (p.gradMagEnabled || p.gradHistEnabled)
        {
            rs.moFloats = std::max<int64_t>(np, moBlockedFloats(rs.h, rs.w));
            if ((rc = devAlloc(c, &rs.M, size_t(B) * size_t(rs.moFloats))) || (rc = devAlloc(c, &rs.O, size_t(B) * size_t(rs.moFloats))))
            {
                return rc;
            }
            if (p.normRad)
            {
                rs.uFloats = std::max<int64_t>(np, uBlockedFloats(rs.h, rs.w));
                if ((rc = devAlloc(c, &rs.U, size_t(B) * size_t(rs.uFloats))) || (rc = devAlloc(c, &rs.S, size_t(B) * np)))
                {
                    return rc;
                }
            }
            if (c->taps)
            {
                if ((rc = devAlloc(c, &rs.Mn, size_t(B) * np)))
                {
                    return rc;
                }
            }
        }
        SmoothJob j{};
        j.h = rs.h;
        j.w = rs.w;
        j.nplanes = d;
        j.out_cs = rs.h;
        j.in_off = 0;
        j.out_off = 0;
        j.in_ps = np;
        j.out_ps = np;
        realJobs.push_back(j);
        c->real.push_back(rs);
    }
    return ACF_HIP_OK;
}

int PlanBuild::imageStrips()
{
    int rc = ACF_HIP_OK;
    (void)rc;
    c->nImgDescs = int(c->h_descs.size());
    // k_resample_strip (the march over strips of output columns) for every down-sampling image resample, and for two consecutive
    // real scales that share their source — the two small scales of a 1080p pyramid — in one pass (A/B: ACF_HIP_RESAMPLE_NO_STRIP)
    if (!fallbackForced(FB_RESAMPLE_NO_STRIP))
    {
        for (size_t k = 0; k < c->real.size(); k++)
        {
            RealScale& ra = c->real[k];
            if (!ra.resampled)
            {
                continue;
            }
            ra.strip = stripPlan(c->h_descs[ra.descIndex], nullptr, arena);
            if (k + 1 < c->real.size() && !fallbackForced(FB_RESAMPLE_NO_PAIR))
            {
                const RealScale& rb = c->real[k + 1];
                if (rb.resampled && !ra.adoptAsI && ra.src_h == rb.src_h && ra.src_w == rb.src_w && !(k > 0 && c->real[k - 1].stripPair.ok) && rb.w <= ra.w && rb.h <= ra.h)
                {
                    ra.stripPair = stripPlan(c->h_descs[ra.descIndex], &c->h_descs[rb.descIndex], arena);
                }
            }
        }
    }

    return ACF_HIP_OK;
}

int PlanBuild::approxLevels()
{
    int rc = ACF_HIP_OK;
    (void)rc;
    // approximated levels (chnsPyramid.cpp:385-397)
    const int nColor = p.colorEnabled ? d : 0;
    const int nMag = p.gradMagEnabled ? 1 : 0;
    c->approxMaxBlocks = 0;
    for (size_t i = 0; i < pl.levels.size(); i++)
    {
        const acf_hip_level& l = pl.levels[i];
        if (l.isReal)
        {
            continue;
        }
        const acf_hip_level& lr = pl.levels[l.realIndex];
        ResampleDesc dd;
        if ((rc = buildResample(lr.hC, lr.wC, l.hC, l.wC, dd, arena)))
        {
            return fail(c, rc, "plan: degenerate resample geometry (approximated level)");
        }
        double ratio[3];
        for (int j = 0; j < 3; j++)
        {
            ratio[j] = std::pow(l.scale / lr.scale, -(p.nLambdas == 3 ? p.lambdas[j] : 0.0)); // :393 (image-specific lambdas: rewritten per frame)
        }
        setResampleGain(dd, ratio, nColor, nColor + nMag);
        dd.nplanes = pl.nChns;
        dd.src_off = pl.raw_off[l.realIndex];
        dd.dst_off = pl.raw_off[i];
        dd.src_frame_stride = pl.raw_floats;
        dd.dst_frame_stride = pl.raw_floats;
        c->h_descs.push_back(dd);
        c->approxMaxBlocks = std::max(c->approxMaxBlocks, resampleBlocks(dd));
    }
    c->nApproxDescs = int(c->h_descs.size()) - c->nImgDescs;
    c->autoLambdas = p.nApprox > 0 && p.nLambdas != 3;
    c->h_lambdas.assign(size_t(B) * 3, 0.0);
    if (c->autoLambdas && (rc = devAlloc(c, &c->d_planeSums, size_t(B) * 2 * pl.nChns)))
    {
        return rc;
    }

    return ACF_HIP_OK;
}

int PlanBuild::finalJobs()
{
    int rc = ACF_HIP_OK;
    (void)rc;
    // final smoothing + padding jobs (chnsPyramid.cpp:399-435)
    int64_t rankOff = 0; // the rank pyramid's layout (rankPitch): levels in order, nChns planes [wP][pitch] each
    c->finalMaxH = 0;
    c->padMaxElems = 0;
    py = p.pad_h / shrink;
    px = p.pad_w / shrink;
    for (size_t i = 0; i < pl.levels.size(); i++)
    {
        const acf_hip_level& l = pl.levels[i];
        SmoothJob j{};
        j.h = l.hC;
        j.w = l.wC;
        j.nplanes = pl.nChns;
        j.out_cs = l.hP;
        j.in_off = pl.raw_off[i];
        j.out_off = l.offset + int64_t(px) * l.hP + py;
        j.in_ps = int64_t(l.hC) * l.wC;
        j.out_ps = int64_t(l.hP) * l.wP;
        finalJobList.push_back(j);
        c->finalMaxH = std::max(c->finalMaxH, l.hC);
        PadJob q{};
        q.hC = l.hC;
        q.wC = l.wC;
        q.hP = l.hP;
        q.wP = l.wP;
        q.py = py;
        q.px = px;
        q.nplanes = pl.nChns;
        q.pitch = l.hP;
        q.off = l.offset;
        padJobs.push_back(q);
        q.pitch = rankPitch(l.hP);
        q.off = rankOff;
        padJobsR.push_back(q);
        c->padMaxElems = std::max<int64_t>(c->padMaxElems, int64_t(pl.nChns) * (int64_t(l.wP) * (l.hP - l.hC) + int64_t(l.wP - l.wC) * l.hC)); // border cells
        rankOffs.push_back(rankOff);
        rankOff += int64_t(pl.nChns) * rankPitch(l.hP) * l.wP;
    }
    return ACF_HIP_OK;
}

int PlanBuild::levelJobs()
{
    int rc = ACF_HIP_OK;
    (void)rc;
    // level jobs (k_level): every level, real ones read their raw channels, approximated ones resample on the
    // fly; sorted into runs of equal (rows-per-lane R, mode) because both are template parameters of the kernel
    {
        struct Keyed
        {
            int key;
            LevelJob j;
        };
        std::vector<Keyed> fusedJobs, rawJobs;
        c->fusedOk = p.smooth > 0 && c->finalMaxH <= 64 * LEVEL_MAX_R_REAL;
        int ai = 0;
        for (size_t i = 0; i < pl.levels.size(); i++)
        {
            const acf_hip_level& l = pl.levels[i];
            LevelJob j{};
            j.hC = l.hC;
            j.wC = l.wC;
            j.out_cs = l.hP;
            j.in_off = pl.raw_off[i];
            j.raw_off = pl.raw_off[i];
            j.out_off = l.offset + int64_t(px) * l.hP + py;
            j.in_ps = int64_t(l.hC) * l.wC;
            j.out_ps = int64_t(l.hP) * l.wP;
            j.rank_cs = rankPitch(l.hP);
            j.rank_ps = int64_t(j.rank_cs) * l.wP;
            j.rank_off = rankOffs[i] + int64_t(px) * j.rank_cs + py;
            j.desc = -1;
            const int R = (l.hC + 63) / 64;
            int mode = LM_REAL;
            rawJobs.push_back({ R * 8 + LM_REAL, j });
            if (!l.isReal)
            {
                j.desc = ai;
                const ResampleDesc& dd = c->h_descs[size_t(c->nImgDescs + ai)];
                if ((dd.ymode == RS_DOWN && dd.ybd0 > 3) || (dd.xmode == RS_DOWN && dd.xbd0 > 3) || dd.ymode == RS_EXACT || dd.xmode == RS_EXACT)
                {
                    c->fusedOk = false; // more than three taps on an axis or an exact 1/k ratio: separate launches
                }
                mode = dd.xmode == RS_DOWN ? (dd.ymode == RS_DOWN ? LM_DD : LM_DU) : (dd.ymode == RS_DOWN ? LM_UD : LM_UU);
                if (R > 8)
                {
                    c->fusedOk = false; // (only copies of real levels are instantiated beyond eight rows per lane)
                }
                if (dd.ha >= 64 * (dd.ymode == RS_DOWN ? (3 * R + 1) / 2 : R))
                {
                    // more source rows than LevelWindow's registers hold (ratio beyond 2^(1/2)), or no row left in the
                    // column buffer for the zeros that taps beyond the source's last row read (level_column)
                    c->fusedOk = false;
                }
                if (mode == LM_DU || mode == LM_UD)
                {
                    // one axis up, the other down: cannot happen with getScales' isotropic scales (an exhaustive scan of
                    // 64..330 x 64..330 frames finds none), so the fused kernel is not instantiated for it
                    c->fusedOk = false;
                }
                ai++;
            }
            fusedJobs.push_back({ R * 8 + mode, j });
        }
        auto pack = [&](std::vector<Keyed>& v, std::vector<acf_hip_ctx::LevelGroup>& groups, LevelJob** dst, int* nAll) {
            auto jobCost = [](const Keyed& k) {
                // one wave walks wC column steps; a step costs ~R row registers x (1 for a copy, 3 for a resampled column) + the recursion
                const int R = k.key / 8, mode = k.key % 8;
                return double(k.j.wC) * (R * (mode == LM_REAL ? 1.0 : 3.0) + 2.0);
            };
            for (auto& k : v)
            {
                k.j.kind = k.key;
            }
            // k_level_all takes every job whose specialisation fits 128 VGPRs, longest chain first; the rest go out as
            // one launch per (R, mode) run on the side streams
            std::vector<Keyed> all, rest;
            for (const auto& k : v)
            {
                const int R = k.key / 8, mode = k.key % 8;
                ((R <= 4 || (mode == LM_REAL && R <= 8)) && !fallbackForced(FB_LEVEL_GROUPS) ? all : rest).push_back(k);
            }
            std::stable_sort(all.begin(), all.end(), [&](const Keyed& a, const Keyed& b) { return jobCost(a) > jobCost(b); });
            std::stable_sort(rest.begin(), rest.end(), [](const Keyed& a, const Keyed& b) { return a.key < b.key; });
            std::vector<LevelJob> flat;
            for (const auto& k : all)
            {
                flat.push_back(k.j);
            }
            *nAll = int(all.size());
            groups.clear();
            for (const auto& k : rest)
            {
                if (groups.empty() || groups.back().R * 8 + groups.back().mode != k.key)
                {
                    groups.push_back({ k.key / 8, k.key % 8, int(flat.size()), 0, 0.0, 0 });
                }
                groups.back().count++;
                groups.back().cost += jobCost(k);
                flat.push_back(k.j);
            }
            return devUpload(c, dst, flat);
        };
        if ((rc = pack(fusedJobs, c->levelGroups, &c->d_levelJobs, &c->nAllJobs)) || (rc = pack(rawJobs, c->levelGroupsRaw, &c->d_levelJobsRaw, &c->nAllJobsRaw)))
        {
            return rc;
        }
        c->levelsEmitRank = c->fusedOk && c->levelGroups.empty() && c->nAllJobs == int(pl.levels.size());
        for (const auto& l : pl.levels)
        {
            if ((py & 1) && l.hC % 64 == 0)
            {
                c->levelsEmitRank = false; // (level_body's paired rank stores: no lane holds row hC)
            }
        }
    }
    return ACF_HIP_OK;
}

int PlanBuild::ldcf()
{
    int rc = ACF_HIP_OK;
    (void)rc;
    // ---- LDCF post-stage (acf_hip_params::ldcfK): level table of the filtered, halved pyramid + one resample per level
    c->ldcfLevels.clear();
    c->ldcfFloats = c->ldcfTmpFloats = 0;
    if (p.ldcfK > 0)
    {
        const int shrink2 = 2 * p.shrink, nCk = pl.nChns * p.ldcfK;
        c->ldcfDescBase = int(c->h_descs.size());
        c->ldcfMaxCells = c->ldcfMaxBlocks = 0;
        std::vector<LdcfJob> ldcfJobs;
        int64_t off = 0;
        for (size_t i = 0; i < pl.levels.size(); i++)
        {
            acf_hip_level l = pl.levels[i];
            const acf_hip_level& s0 = pl.levels[i];
            l.hP = l.hC = int(std::floor(0.5 * s0.hP + 0.5)); // imResample(C, .5): round(.5 * size)
            l.wP = l.wC = int(std::floor(0.5 * s0.wP + 0.5));
            if (l.hP < 1 || l.wP < 1)
            {
                return fail(c, ACF_HIP_E_UNSUPPORTED, "plan: LDCF level smaller than one cell");
            }
            l.nWinR = std::max(0, int(std::ceil(float(l.hP * shrink2 - p.modelDsPad_h + 1) / p.stride)));
            l.nWinC = std::max(0, int(std::ceil(float(l.wP * shrink2 - p.modelDsPad_w + 1) / p.stride)));
            l.offset = off;
            off += int64_t(nCk) * l.hP * l.wP;
            c->ldcfLevels.push_back(l);
            ResampleDesc dd;
            if ((rc = buildResample(s0.hP, s0.wP, l.hP, l.wP, dd, arena)))
            {
                return fail(c, rc, "plan: degenerate LDCF resample geometry");
            }
            const double one[3] = { 1.0, 1.0, 1.0 };
            setResampleGain(dd, one, nCk, nCk);
            dd.nplanes = nCk;
            dd.src_off = int64_t(p.ldcfK) * s0.offset; // the filtered scratch holds every level: k planes per channel plane
            dd.dst_off = l.offset;
            c->h_descs.push_back(dd);
            LdcfJob j{};
            j.h = s0.hP;
            j.w = s0.wP;
            j.inOff = s0.offset;
            j.outOff = int64_t(p.ldcfK) * s0.offset;
            ldcfJobs.push_back(j);
            c->ldcfMaxCells = std::max(c->ldcfMaxCells, s0.hP * s0.wP);
            c->ldcfMaxBlocks = std::max(c->ldcfMaxBlocks, resampleBlocks(dd));
        }
        c->ldcfTmpFloats = int64_t(p.ldcfK) * pl.pyr_floats;
        c->ldcfFloats = off;
        {
            // fused path: every level tiled for k_resample_tile's passes with 16 output columns per tile
            std::vector<LdcfTileJob> tj;
            int maxR = 0, maxC = 0;
            bool ok = !fallbackForced(FB_LDCF_UNFUSED);
            const int xoMax = 16; // output columns per tile at most (source tile: twice as many columns)
            for (size_t i = 0; i < c->ldcfLevels.size() && ok; i++)
            {
                const ResampleDesc& dd = c->h_descs[size_t(c->ldcfDescBase) + i];
                // the largest tile of at most 64 x 16 outputs whose source tile is at most 128 rows x 32 columns: the filter
                // stage then has exactly two tile rows per lane and eight column quads (k_ldcf_tile)
                ResampleTiling tl;
                int yo = 0, xo = 0;
                {
                    std::vector<std::pair<int, int>> cand;
                    for (int y = RT_YO; y >= 32; y--)
                    {
                        for (int x = xoMax; x >= xoMax / 2; x--)
                        {
                            cand.push_back({ y, x });
                        }
                    }
                    std::stable_sort(cand.begin(), cand.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first * a.second > b.first * b.second; });
                    for (const auto& yx : cand)
                    {
                        tl = resampleTilePlan(dd, arena, yx.second, int64_t(40) * 1024, yx.first, 128, 2 * xoMax);
                        if (tl.rows > 0)
                        {
                            yo = yx.first;
                            xo = yx.second;
                            break;
                        }
                    }
                }
                if (tl.rows <= 0)
                {
                    ok = false;
                    break;
                }
                maxR = std::max(maxR, tl.rows);
                maxC = std::max(maxC, tl.cols);
                const int ntY = cdiv(dd.hb, yo), ntX = cdiv(dd.wb, xo);
                for (int x = 0; x < ntX; x++)
                {
                    for (int y = 0; y < ntY; y++)
                    {
                        LdcfTileJob j{};
                        j.level = int(i);
                        j.ytile = y;
                        j.xtile = x;
                        j.tile_y = tl.tile_y;
                        j.tile_x = tl.tile_x;
                        j.yo = yo;
                        j.xo = xo;
                        tj.push_back(j);
                    }
                }
            }
            c->ldcfTiles = 0;
            if (ok && !tj.empty())
            {
                c->ldcfTiles = int(tj.size());
                c->ldcfTileRows = maxR;
                c->ldcfTileCols = maxC;
                if ((rc = devUpload(c, &c->d_ldcfTileJobs, tj)))
                {
                    return rc;
                }
            }
        }
        for (size_t i = 0; i < c->ldcfLevels.size(); i++)
        {
            ResampleDesc& dd = c->h_descs[size_t(c->ldcfDescBase) + i];
            dd.src_frame_stride = c->ldcfTmpFloats;
            dd.dst_frame_stride = c->ldcfFloats;
        }
        if ((rc = devUpload(c, &c->d_ldcfJobs, ldcfJobs)))
        {
            return rc;
        }
        if ((rc = devUpload(c, &c->d_ldcfFilt, c->ldcfFilters)) || (c->ldcfTiles == 0 && (rc = devAlloc(c, &c->d_ldcfTmp, size_t(B) * c->ldcfTmpFloats + 64))) ||
            (rc = devAlloc(c, &c->d_ldcfPyr, size_t(B) * c->ldcfFloats + 64)))
        {
            return rc;
        }
    }
    return ACF_HIP_OK;
}

int PlanBuild::uploadAndScratch()
{
    int rc = ACF_HIP_OK;
    (void)rc;
    if ((rc = devUpload(c, &c->d_descs, c->h_descs)) || (rc = devUpload(c, &c->d_it, arena.ints)) || (rc = devUpload(c, &c->d_ft, arena.floats)) ||
        (rc = devUpload(c, &c->d_realJobs, realJobs)) || (rc = devUpload(c, &c->d_finalJobs, finalJobList)) || (rc = devUpload(c, &c->d_padJobs, padJobs)) || (rc = devUpload(c, &c->d_padJobsR, padJobsR)))
    {
        return rc;
    }
    if ((rc = devAlloc(c, &c->d_chns, size_t(B) * pl.raw_floats)) || (rc = devAlloc(c, &c->d_pyr, size_t(B) * pl.pyr_floats + 64)) /* + slack: the cascade's 16-byte tile fill may read a few floats past the last plane */)
    {
        return rc;
    }
    {
        // k_smooth_vec's column segments: hand-over states of up to 32 segments per plane, one repair flag per plane
        c->segCap = 32;
        const size_t nState = size_t(B) * d * c->segCap * size_t(std::max(H, 4));
        if ((rc = devAlloc(c, &c->d_specState, nState)) || (rc = devAlloc(c, &c->d_trueState, nState)) || (rc = devAlloc(c, &c->d_redo, size_t(B) * d)))
        {
            return rc;
        }
        // (zero between calls: the repair launch takes its flags down; cleared on the context's own stream, which does not
        // synchronise with the null stream, and again on the error returns between a verify and its repair launch: clearRepairFlags)
        HIPCHK(c, hipMemsetAsync(c->d_redo, 0, sizeof(int32_t) * size_t(B) * d, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->redoInts = size_t(B) * d;
    }
    {
        // k_level_all's column segments: only small batches are bound by a level's chain length
        c->levelSegFrames = std::min(B, 8);
        c->levelSegCap = 8;
        c->levelHMax = std::max(c->finalMaxH, 4);
        const size_t nState = size_t(c->levelSegFrames) * pl.levels.size() * pl.nChns * c->levelSegCap * size_t(c->levelHMax);
        if ((rc = devAlloc(c, &c->d_lvSpec, nState)) || (rc = devAlloc(c, &c->d_lvTrue, nState)) ||
            (rc = devAlloc(c, &c->d_lvRedo, size_t(c->levelSegFrames) * pl.levels.size() * pl.nChns)))
        {
            return rc;
        }
        HIPCHK(c, hipMemsetAsync(c->d_lvRedo, 0, sizeof(int32_t) * size_t(c->levelSegFrames) * pl.levels.size() * pl.nChns, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->lvRedoInts = size_t(c->levelSegFrames) * pl.levels.size() * pl.nChns;
    }
    return ACF_HIP_OK;
}

int PlanBuild::cascade()
{
    int rc = ACF_HIP_OK;
    (void)rc;
    // cascade
    // with LDCF the cascade runs on the filtered pyramid: its tables are built for those levels, nChns*k channels, shrink*2
    const std::vector<acf_hip_level>& cascLevels = p.ldcfK > 0 ? c->ldcfLevels : pl.levels;
    {
        ShrinkScope ss(c, p.ldcfK > 0 ? 2 : 1);
        if ((rc = buildCascadeTables(c, cascLevels, pl.nChns * std::max(p.ldcfK, 1), c->cs, p.ldcfK <= 0)))
        {
            return rc;
        }
    }
    if ((rc = devUpload(c, &c->cs.d_thrs, c->thrs)) || (rc = devUpload(c, &c->cs.d_hs, c->hs)) || (rc = devUpload(c, &c->cs.d_child, c->child)) ||
        (rc = devUpload(c, &c->cs.d_fids, c->fids)))
    {
        return rc;
    }
    std::vector<BoxLevel> box(pl.levels.size());
    for (size_t i = 0; i < pl.levels.size(); i++)
    {
        box[i].shw_h = pl.levels[i].scalehw_h;
        box[i].shw_w = pl.levels[i].scalehw_w;
        // cv::Size(cv::Size2d(modelDs) / scale): saturate_cast<int>(double) == cvRound (ACF.cpp:304)
        box[i].bh = int(std::lrint(double(p.modelDs_h) / pl.levels[i].scale));
        box[i].bw = int(std::lrint(double(p.modelDs_w) / pl.levels[i].scale));
    }
    if ((rc = devUpload(c, &c->d_boxLevels, box)))
    {
        return rc;
    }
    if (c->cs.dedupQ > 1 && (rc = devAlloc(c, &c->cs.d_hitsX, size_t(B) * max_hits)))
    {
        return rc;
    }
    if ((rc = devAlloc(c, &c->cs.d_hits, size_t(B) * max_hits)) || (rc = devAlloc(c, &c->cs.d_sorted, size_t(B) * max_hits)) ||
        (rc = devAlloc(c, &c->cs.d_dets, size_t(B) * max_hits)) || (rc = devAlloc(c, &c->cs.d_counts, size_t(B))))
    {
        return rc;
    }
    {
        int64_t nWinTotal = 0;
        for (const auto& l : cascLevels)
        {
            nWinTotal += int64_t(l.nWinR) * l.nWinC;
        }
        c->cs.qcap = int(std::max<int64_t>(nWinTotal, 1));
        if ((rc = devAlloc(c, &c->cs.d_queue[0], size_t(B) * c->cs.qcap)) || (rc = devAlloc(c, &c->cs.d_queue[1], size_t(B) * c->cs.qcap)) ||
            (rc = devAlloc(c, &c->cs.d_qcounts, size_t(8) * B + 8))) // (+ 8: k_cascade_tile3's tile counters, one per XCD)
        {
            return rc;
        }
    }
    return ACF_HIP_OK;
}
} // namespace
