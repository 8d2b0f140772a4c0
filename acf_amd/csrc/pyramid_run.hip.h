// pyramid_run.hip.h — part of acf_hip.hip (included there and nowhere else): acf_hip_pyramid's launch logic, one function per stage of
// Detector::chnsPyramid (chnsPyramid.cpp:160-456) in the order the reference runs them.  PyramidRun holds what the stages of ONE call
// share; every stage enqueues on the context's stream(s) and returns an ACF_HIP_* status.
//
//   ingest()            packed 8-bit frames -> planar f32 (ACF.cpp:114-119,137), the apps' Resizer in front of it
//   colour()            rgbConvert once at full resolution (chnsPyramid.cpp:230-263)
//   prepareScales()     which image every real scale is resampled from (the I = I1 adoption, :313-316), the scales' streams
//   realScale(k)        imageResample(k) -> smoothImage(k) -> gradientChannels(k): chnsCompute at a real scale (:297-338, chnsCompute.cpp:146-338)
//   joinScales()
//   levels(f0, n)       the approximated levels + final smoothing + padding of frames [f0, f0 + n) (:385-435)
//   levelsWithImageLambdas()   the same behind the per-image lambda estimate (:341-374)
#pragma once

namespace
{
// what smoothImage leaves for gradientChannels
struct SmoothOut
{
    bool colorDone = false;                       // the level's colour channels were written from the smoothing chain's registers
    bool gradFused = false, gradBlocked = false;  // M and O written by k_smooth_grad (in 64 x 16 blocks)
    bool triXFused = false;                       // ... and U by k_smooth_grad_tri
};

struct PyramidRun
{
    acf_hip_ctx* const c;
    const acf_hip_params& p;
    const Plan& pl;
    const int nF;
    const int H, W, d, d_in, shrink;
    const int64_t np0;
    const float* frames;                          // planar f32 input planes (null once the 8-bit ingest has converted them itself)
    const float* cur = nullptr;                   // "I" of chnsPyramid: the image the next real scale is resampled from
    int64_t cur_fs = 0;
    int curH = 0, curW = 0;
    float pColor = 0.f;
    std::vector<int> srcIdx;                      // per real scale: the real scale whose smoothed image it is resampled from (-1: the frame)
    std::vector<char> halfDone, pairDone;         // the scale's image was produced already (by the previous scale's smoothing / its strip pair)
    bool scalePar = false;                        // the real scales on their own streams
    hipStream_t mainStream;
    bool wroteRank = false, wroteF32 = true;      // what the level kernels left: 16-bit rank cells, floats
    PackedSrc reduced{};

    PyramidRun(acf_hip_ctx* ctx, const float* frames_, int nF_)
        : c(ctx), p(ctx->p), pl(ctx->plan), nF(nF_), H(pl.H), W(pl.W), d(pl.d), d_in(pl.d_in), shrink(ctx->p.shrink), np0(int64_t(pl.H) * pl.W), frames(frames_),
          mainStream(ctx->stream)
    {
    }
    ~PyramidRun() { c->stream = mainStream; } // (the launch helpers all use c->stream: realScale points it at the scale's stream; back on every way out)

    int ingest(const PackedSrc* u8, bool& ingestConverted);
    int colour(bool ingestConverted);
    int prepareScales();
    int realScale(size_t k);
    int imageResample(size_t k, const float*& img, int64_t& img_fs);
    int smoothImage(size_t k, const float* img, int64_t img_fs, SmoothOut& so);
    int gradientChannels(size_t k, const SmoothOut& so);
    int joinScales();
    int levels(int f0, int nLF);
    int levelsWithImageLambdas();
};

int PyramidRun::ingest(const PackedSrc* u8, bool& ingestConverted)
{
    int rc;
    if (d_in == 5)
    {
        return fail(c, ACF_HIP_E_INVALID, "pyramid_u8: a plan for five input planes (image + M, O) takes float frames");
    }
    if (u8 && c->rz.on)
    {
        // the apps' Resizer (acf.cpp:117-148) in front of the ingest: the caller's frames are rz.rows x rz.cols
        const int cpp = pixCpp(u8->pix);
        const int stride = u8->rowStride > 0 ? u8->rowStride : c->rz.cols * cpp;
        if (stride < c->rz.cols * cpp)
        {
            return fail(c, ACF_HIP_E_INVALID, "pyramid_u8: row stride smaller than a row of the unreduced frame");
        }
        if ((rc = launchResizeU8(c, c->rz, u8->frames, cpp, stride, nF, c->rz.d_out)))
        {
            return rc;
        }
        reduced = PackedSrc{ c->rz.d_out, u8->pix, 0 };
        u8 = &reduced;
    }
    if (u8)
    {
        // 8-bit ingest (ACF.cpp:114-119,137; MatP.cpp:51-73), fused with the colour conversion below when there is one
        prof(c, "k_ingest_u8");
        ingestConverted = c->d_color && p.colorSpace != ACF_HIP_CS_HSV; // (hsv: planar ingest, then k_rgb2hsv like a float frame)
        if (ingestConverted)
        {
            if ((rc = launchIngest(c, *u8, nF, c->d_color, int64_t(d) * np0, true)))
            {
                return rc;
            }
            frames = nullptr;
        }
        else
        {
            if (!c->d_stage && (rc = devAlloc(c, &c->d_stage, size_t(c->maxBatch) * d_in * np0)))
            {
                return rc;
            }
            if ((rc = launchIngest(c, *u8, nF, c->d_stage, int64_t(d_in) * np0, false)))
            {
                return rc;
            }
            frames = c->d_stage;
        }
    }
    return ACF_HIP_OK;
}

// ---- colour conversion, once at full resolution (chnsPyramid.cpp:230-263)
int PyramidRun::colour(bool ingestConverted)
{
    cur = frames;
    cur_fs = int64_t(d_in) * np0;
    curH = H;
    curW = W;
    if (ingestConverted)
    {
        cur = c->d_color;
        cur_fs = int64_t(d) * np0;
    }
    else if (c->d_color)
    {
        prof(c, "k_colour");
        dim3 grid(cdiv(np0, 256), 1, nF), block(256);
        const int64_t out_fs = int64_t(d) * np0;
        if (p.colorSpace == ACF_HIP_CS_LUV)
        {
            // d_in == 3, RGB -> LUV; the reference takes the SSE body iff n % 4 == 0 (rgbConvertMex.cpp:92,343)
            if (np0 % 4 == 0)
            {
                hipLaunchKernelGGL(k_rgb2luv<true>, grid, block, 0, c->stream, frames, c->d_color, (const float*)c->d_lTable, makeLuvConsts(), int(np0), cur_fs, out_fs, x86T(c));
            }
            else
            {
                hipLaunchKernelGGL(k_rgb2luv<false>, grid, block, 0, c->stream, frames, c->d_color, (const float*)c->d_lTable, makeLuvConsts(), int(np0), cur_fs, out_fs, x86T(c));
            }
        }
        else if (p.colorSpace == ACF_HIP_CS_GRAY)
        {
            const float mr = (float).2989360213 * 1.0f, mg = (float).5870430745 * 1.0f, mb = (float).1140209043 * 1.0f;
            if (d_in == 1)
            {
                hipLaunchKernelGGL(k_rgb2gray<true>, grid, block, 0, c->stream, frames, c->d_color, int(np0), cur_fs, out_fs, mr, mg, mb);
            }
            else
            {
                hipLaunchKernelGGL(k_rgb2gray<false>, grid, block, 0, c->stream, frames, c->d_color, int(np0), cur_fs, out_fs, mr, mg, mb);
            }
        }
        else if (p.colorSpace == ACF_HIP_CS_HSV)
        {
            hipLaunchKernelGGL(k_rgb2hsv, grid, block, 0, c->stream, frames, c->d_color, int(np0), cur_fs, out_fs); // (d_in == 3: the plan checked)
        }
        else // ORIG / RGB with a 1-plane input: replicate
        {
            hipLaunchKernelGGL(k_replicate3, grid, block, 0, c->stream, frames, c->d_color, int(np0), cur_fs, out_fs);
        }
        LAUNCHCHK(c, "colour conversion");
        cur = c->d_color;
        cur_fs = out_fs;
    }
    return ACF_HIP_OK;
}

// ---- real scales, in order (chnsPyramid.cpp:297-338 + chnsCompute.cpp:146-338)
int PyramidRun::prepareScales()
{
    pColor = p.colorSmooth > 0 ? float(12.0 / p.colorSmooth / (p.colorSmooth + 2.0) - 2.0) : 0.f;
    // which real scale's smoothed image each real scale is resampled from (-1: the input frame), following the
    // reference's I = I1 adoption (chnsPyramid.cpp:313-316)
    srcIdx.assign(c->real.size(), -1);
    halfDone.assign(c->real.size(), 0);
    pairDone.assign(c->real.size() + 1, 0);
    {
        int curIdx = -1;
        for (size_t k = 0; k < c->real.size(); k++)
        {
            srcIdx[k] = curIdx;
            if (c->real[k].adoptAsI)
            {
                curIdx = int(k);
            }
        }
    }
    // Real scale k + 1 needs scale k's SMOOTHED image only (its exact half, or the adopted image it is resampled from); what
    // follows the smoothing of scale k — gradMag, convTri, the cells: column-sequential chains that leave most of the machine
    // idle — runs beside the smaller scales' own chains: every scale gets a stream, ordered by "scale k has been smoothed"
    // events (option scale_streams).
    if (c->scaleStreams && c->real.size() > 1 && !c->taps)
    {
        ensureSide(c);
    }
    const size_t nSideS = c->side.size();
    scalePar = c->scaleStreams && c->real.size() > 1 && nSideS >= c->real.size() - 1 && c->evJoin.size() >= nSideS && !c->taps;
    while (scalePar && c->evScale.size() < c->real.size())
    {
        hipEvent_t e;
        HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->evScale.push_back(e);
    }
    return ACF_HIP_OK;
}

int PyramidRun::realScale(size_t k)
{
    int rc;
    RealScale& rs = c->real[k];
    const int64_t np = int64_t(rs.h) * rs.w;
    if (scalePar)
    {
        if (k > 0)
        {
            prof(c, "(end)"); // closes the previous scale's last kernel on its stream
        }
        c->stream = k == 0 ? mainStream : c->side[k - 1];
        if (k > 0)
        {
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->evScale[k - 1], 0));
        }
    }
    const float* img = cur;
    int64_t img_fs = cur_fs;
    if (rs.resampled && (rc = imageResample(k, img, img_fs)))
    {
        return rc;
    }
    SmoothOut so;
    if ((rc = smoothImage(k, img, img_fs, so)))
    {
        return rc;
    }
    if (scalePar)
    {
        HIPCHK(c, hipEventRecord(c->evScale[k], c->stream));
    }
    if (rs.adoptAsI)
    {
        cur = rs.sm;
        cur_fs = int64_t(d) * np;
        curH = rs.h;
        curW = rs.w;
    }
    return gradientChannels(k, so);
}

// the real scale's image from the adopted image (chnsPyramid.cpp:305-311)
int PyramidRun::imageResample(size_t k, const float*& img, int64_t& img_fs)
{
    RealScale& rs = c->real[k];
    const int64_t np = int64_t(rs.h) * rs.w;
    if (rs.src_h != curH || rs.src_w != curW)
    {
        return fail(c, ACF_HIP_E_INVALID, "pyramid: internal plan mismatch");
    }
    if (c->h_descs[rs.descIndex].src_frame_stride != cur_fs)
    {
        return fail(c, ACF_HIP_E_INVALID, "pyramid: internal frame-stride mismatch");
    }
    prof(c, "k_resample(image)");
    const ResampleDesc& hd = c->h_descs[rs.descIndex];
    const bool exactHalf = hd.xmode == RS_EXACT && hd.ymode == RS_EXACT && hd.xk == 2 && hd.yk == 2 && hd.ha % 4 == 0 && hd.hb % 2 == 0 &&
        hd.src_frame_stride % 4 == 0 && hd.dst_frame_stride % 2 == 0 && hd.src_off % 4 == 0 && hd.dst_off % 2 == 0 &&
        (uintptr_t(cur) & 15) == 0 && (uintptr_t(rs.img) & 7) == 0; // (k_resample_half's case: one thread per output pair)
    if (halfDone[k] || pairDone[k])
    {
        // already produced by the previous scale's k_smooth_vec, or together with the previous scale's image (k_resample_strip)
    }
    else if (rs.strip.ok && !exactHalf && !resampleGenericOnly() && (uintptr_t(cur) & 15) == 0)
    {
        // (with the scales on their own streams the next scale's chain would need one more event: the pair is for the one-stream order)
        const bool pair = rs.stripPair.ok && k + 1 < c->real.size() && srcIdx[k + 1] == srcIdx[k] && !c->taps && !scalePar;
        if (pair)
        {
            pairDone[k + 1] = 1;
        }
        launchStrip(c, pair ? rs.stripPair : rs.strip, c->d_descs, rs.descIndex, pair ? c->real[k + 1].descIndex : -1, hd.nplanes, cur, rs.img,
            pair ? c->real[k + 1].img : nullptr, c->d_it, c->d_ft, nF);
    }
    else if (exactHalf)
    {
        const int64_t items = int64_t(hd.hb / 2) * hd.wb * hd.nplanes;
        hipLaunchKernelGGL(k_resample_half, dim3(cdiv(items, 256), 1, nF), dim3(256), 0, c->stream, cur, rs.img,
            (const ResampleDesc*)(c->d_descs + rs.descIndex));
    }
    else if (resampleUpOk(hd) && !resampleGenericOnly() && (uintptr_t(rs.img) & 15) == 0)
    {
        // nOctUp > 0: the frame up-sampled (cfg 4)
        const int nw = hd.nplanes * cdiv(hd.hb, 256) * cdiv(hd.wb, RSU_XC);
        hipLaunchKernelGGL(k_resample_up, dim3(cdiv(nw, 4), 1, nF), dim3(256), 0, c->stream, cur, rs.img,
            (const ResampleDesc*)(c->d_descs + rs.descIndex), (const int32_t*)c->d_it, (const float*)c->d_ft);
    }
    else
    {
        // small planes: fewer columns per wave, more waves
        const int xt = int64_t(resampleBlocks(hd)) * nF < 4096 ? 2 : RS_XT;
        hipLaunchKernelGGL(k_resample, dim3(resampleBlocks(hd, xt), 1, nF), dim3(64, 4), 0, c->stream, cur, rs.img,
            (const ResampleDesc*)(c->d_descs + rs.descIndex), (const int32_t*)c->d_it, (const float*)c->d_ft, xt);
    }
    LAUNCHCHK(c, "k_resample(image)");
    img = rs.img;
    img_fs = int64_t(d) * np;
    return ACF_HIP_OK;
}

int PyramidRun::smoothImage(size_t k, const float* img, int64_t img_fs, SmoothOut& so)
{
    int rc;
    RealScale& rs = c->real[k];
    const int64_t np = int64_t(rs.h) * rs.w;
    // convTri(I, I, pColor.smooth, 1) in place (chnsCompute.cpp:239)
    bool& colorDone = so.colorDone;
    bool &gradFused = so.gradFused, &gradBlocked = so.gradBlocked, &triXFused = so.triXFused;
    const bool fuseSm = p.colorSmooth > 0 && p.colorEnabled && !c->taps && !c->noFusedSmooth && shrink == 4 && rs.h % 4 == 0 && rs.w % 4 == 0 && rs.w >= 16 &&
        rs.h / 4 <= SV_MAXW * SV_OWN && img_fs % 4 == 0 && np % 4 == 0 && pl.raw_floats % 1 == 0 && (uintptr_t(img) & 15) == 0;
    if (fuseSm)
    {
        // k_smooth_vec: smoothing + the level's colour channels (+ the next real scale's image when it is an exact half
        // of this one) from registers; full-resolution smoothed planes are written only where something still reads them
        const bool halfNext = k + 1 < c->real.size() && srcIdx[k + 1] == int(k) && c->real[k + 1].resampled &&
            [&] { const ResampleDesc& nd = c->h_descs[c->real[k + 1].descIndex];
                  return nd.xmode == RS_EXACT && nd.ymode == RS_EXACT && nd.xk == 2 && nd.yk == 2 && nd.ha == rs.h && nd.wa == rs.w; }();
        bool needFullAll = false;
        for (size_t m = k + 1; m < c->real.size(); m++)
        {
            if (srcIdx[m] == int(k) && !(m == k + 1 && halfNext))
            {
                needFullAll = true;
            }
        }
        SmoothVecArgs sa{};
        sa.in = img;
        sa.in_fs = img_fs;
        sa.in_ps = np;
        sa.sm = rs.sm;
        sa.sm_fs = int64_t(d) * np;
        sa.sm_ps = np;
        sa.chns = c->d_chns + pl.raw_off[rs.level];
        sa.chns_fs = pl.raw_floats;
        sa.cells = int64_t(rs.h / 4) * (rs.w / 4);
        sa.h = rs.h;
        sa.w = rs.w;
        sa.p = pColor;
        sa.rq_y = shrinkGainY(shrink);
        sa.dump = c->d_dump;
        if (halfNext)
        {
            const RealScale& nx = c->real[k + 1];
            const ResampleDesc& nd = c->h_descs[nx.descIndex];
            sa.half = nx.img;
            sa.half_ps = int64_t(nx.h) * nx.w;
            sa.half_fs = int64_t(d) * sa.half_ps;
            sa.rkHalf = nd.rk[0];
            halfDone[k + 1] = true;
        }
        const int nq = rs.h / 4, nt = cdiv(nq, SV_OWN) * 64; // a wave owns SV_OWN row quads and shadows SV_K of each neighbour
        const size_t ldsB = size_t(2) * SV_MAXW * 2 * SV_K * 4 * sizeof(float);
        {
            uint32_t fullMask = 0;
            for (int z = 0; z < d; z++)
            {
                if (needFullAll || ((p.gradMagEnabled || p.gradHistEnabled) && z == p.colorChn))
                {
                    fullMask |= 1u << z;
                }
            }
            sa.plane0 = 0;
            sa.nPlanes = d;
            // column segments: as many as give a launch ~6 waves per SIMD (a plane is a chain of column steps with
            // nt / 64 waves), each at least 4 warm-ups long; one segment = the plain recursion.  `planesOf`: planes in the launch
            const int warm = std::max(16, c->smoothWarm);
            auto segmentsFor = [&](int planesOf, int& segW_) {
                int n = c->smoothSegments;
                if (n == 0 && c->sharedDevice && nF >= 64)
                {
                    n = 1; // (beside other contexts: the plain chain — no warm-up columns, nothing to verify or repair)
                }
                if (n == 0)
                {
                    const int64_t waves = int64_t(std::max(planesOf, 1)) * nF * (nt / 64);
                    n = int(std::min<int64_t>((6 * 1024 + waves - 1) / waves, rs.w / (4 * warm)));
                }
                n = std::max(1, std::min(n, std::min(c->segCap, rs.w / 16)));
                segW_ = cdiv(cdiv(rs.w, n), 16) * 16;
                return cdiv(rs.w, segW_);
            };
            int segW = 0;
            int nSeg = segmentsFor(d, segW);
            sa.segW = segW;
            sa.warm = warm;
            sa.nSeg = nSeg;
            sa.segStride = nSeg;
            sa.specState = c->d_specState;
            sa.trueState = c->d_trueState;
            sa.redo = nullptr;
            sa.skipZ = -1;
            // the gradient plane's chain also emits gradMag (k_smooth_grad): its smoothed plane is then written only
            // where a later scale is resampled from it, and k_grad_mag_vec does not run for this scale
            const bool gradVecOk = rs.h % 4 == 0 && np % 4 == 0;
            // Where it pays (measured at 1080p, 3 x 96 frames: +4 % frames/s with scale 0 fused, +2 % with every scale; one
            // frame alone 1.21 -> 1.61 / 2.40 ms): the gradient work rides on a chain of column steps, so it needs many
            // chains (frames x segments) and a plane big enough for the saved round trip to matter.  A/B: the variables.
            const int64_t gradMinPx = int64_t(1) << 20;
            const int gradMinF = 16;
            const bool wantGrad = (p.gradMagEnabled || p.gradHistEnabled) && gradVecOk && !(d_in == 5 && k == 0) && // (five planes: M, O came with the image)
                (c->fusedGrad >= 2 || (c->fusedGrad == 1 && np >= gradMinPx && nF >= gradMinF));
            if (wantGrad)
            {
                const bool wantTri0 = (p.gradMagEnabled || p.gradHistEnabled) && p.normRad;
                // (the layout of M and O: the decision the y pass takes below, from the same inputs)
                ChnsArgs a0{};
                a0.M = rs.M;
                a0.O = rs.O;
                a0.Mn = nullptr; // (no taps on this path)
                a0.doNorm = p.normRad != 0;
                a0.colorDone = 1;
                a0.colorEnabled = p.colorEnabled;
                a0.magEnabled = p.gradMagEnabled;
                a0.histEnabled = p.gradHistEnabled;
                a0.nOrients = p.nOrients;
                const bool blocked0 = wantTri0 && triPlan(rs.M, rs.U, rs.h, rs.w, p.normRad, np, &a0, rs.uFloats, rs.moFloats, true).blocked;
                gradBlocked = blocked0;
                sa.skipZ = p.colorChn;
                sa.gM = rs.M;
                sa.gO = rs.O;
                sa.acos = c->d_acos;
                sa.mo_fs = blocked0 ? moBlockedFloats(rs.h, rs.w) : np;
                sa.nybM = blocked0 ? (rs.h + 15) / 16 : 0;
                sa.full = p.full;
                gradFused = true;
                if (!needFullAll)
                {
                    fullMask &= ~(1u << p.colorChn);
                }
                // convTri's x pass on the same chain: running sums have no warm-up, so the gradient plane is then one segment
                // (96 workgroups for 96 frames: slower alone, faster beside other contexts' kernels — DESIGN.md 3.0)
                const int triMinF = 64;
                if (blocked0 && p.normRad == 5 && rs.w >= 48 && nt <= 512 && (c->fusedTri >= 2 || (c->fusedTri == 1 && c->sharedDevice && nF >= triMinF)))
                {
                    triXFused = true;
                    sa.tU = rs.U;
                    sa.u_fs = uBlockedFloats(rs.h, rs.w);
                    sa.nybU = (rs.h + 8 + 15) / 16;
                }
            }
            // the two launches cut their planes on their own: the gradient plane's launch has a third of the chains (more
            // segments), the other planes' launch two thirds
            int nSegG = nSeg, segWG = segW;
            if (wantGrad && d > 1)
            {
                nSeg = segmentsFor(d - 1, segW);
                nSegG = segmentsFor(1, segWG);
                if (triXFused)
                {
                    nSegG = 1;
                    segWG = cdiv(rs.w, 16) * 16;
                }
                sa.segW = segW;
                sa.nSeg = nSeg;
                sa.segStride = std::max(nSeg, nSegG);
            }
            const size_t ldsG = ldsB + size_t(GM_ACOS_N) * sizeof(float) + (c->arith ? size_t(X86_GM_N) * sizeof(uint2) : 0); // (+ gradMag's table pairs, option "arith")
            if (wantGrad && d == 1 && triXFused)
            {
                nSeg = nSegG = 1; // (one launch: the gradient plane's)
                segW = segWG = cdiv(rs.w, 16) * 16;
                sa.segW = segW;
                sa.nSeg = 1;
                sa.segStride = 1;
            }
            if (wantGrad)
            {
                int rcl = 0;
                if ((rcl = allowLds(c, reinterpret_cast<const void*>(&k_smooth_grad<true, false>), ldsG)) ||
                    (rcl = allowLds(c, reinterpret_cast<const void*>(&k_smooth_grad<false, false>), ldsG)) ||
                    (rcl = allowLds(c, reinterpret_cast<const void*>(&k_smooth_grad_tri<true, false>), ldsG)) ||
                    (rcl = allowLds(c, reinterpret_cast<const void*>(&k_smooth_grad_tri<false, false>), ldsG)) ||
                    (rcl = allowLds(c, reinterpret_cast<const void*>(&k_smooth_grad<true, true>), ldsG)) ||
                    (rcl = allowLds(c, reinterpret_cast<const void*>(&k_smooth_grad<false, true>), ldsG)) ||
                    (rcl = allowLds(c, reinterpret_cast<const void*>(&k_smooth_grad_tri<true, true>), ldsG)) ||
                    (rcl = allowLds(c, reinterpret_cast<const void*>(&k_smooth_grad_tri<false, true>), ldsG)))
                {
                    return rcl;
                }
            }
            auto launchSv = [&](dim3 grid) {
                // (profile names: the gradient plane's launch by its form, the other planes' launch "k_smooth_vec")
                prof(c, !wantGrad ? "k_smooth_vec" : triXFused ? "k_smooth_grad_tri" : "k_smooth_grad");
                if (wantGrad)
                {
                    // (first: its chains are the long ones)
                    SmoothVecArgs sg = sa;
                    sg.plane0 = p.colorChn;
                    sg.skipZ = -1;
                    if (sa.nSeg > 1 || nSegG > 1) // (not the repair launch: that one is one chain per plane)
                    {
                        if (!sa.redo)
                        {
                            sg.segW = segWG;
                            sg.nSeg = nSegG;
                            grid.y = unsigned(nSegG);
                        }
                    }
                    if (sa.redo && nSegG == 1)
                    {
                        // (nothing to repair: the plane was one chain)
                    }
                    else
                    {
                        // (the gradient plane's launch: with convTri's x pass on the chain or without, the next scale's half image or not,
                        // exact reciprocals or — option "arith" — one x86 CPU's)
                        sg.x86 = x86T(c);
                        const dim3 gg = triXFused ? dim3(1, 1, grid.z) : dim3(1, grid.y, grid.z);
#define SG_LAUNCH(KERNEL)                                                                                          \
    if (halfNext)                                                                                                  \
    {                                                                                                              \
        if (sg.x86)                                                                                                \
        {                                                                                                          \
            hipLaunchKernelGGL((KERNEL<true, true>), gg, dim3(nt), ldsG, c->stream, sg, fullMask);                 \
        }                                                                                                          \
        else                                                                                                       \
        {                                                                                                          \
            hipLaunchKernelGGL((KERNEL<true, false>), gg, dim3(nt), ldsG, c->stream, sg, fullMask);                \
        }                                                                                                          \
    }                                                                                                              \
    else if (sg.x86)                                                                                               \
    {                                                                                                              \
        hipLaunchKernelGGL((KERNEL<false, true>), gg, dim3(nt), ldsG, c->stream, sg, fullMask);                    \
    }                                                                                                              \
    else                                                                                                           \
    {                                                                                                              \
        hipLaunchKernelGGL((KERNEL<false, false>), gg, dim3(nt), ldsG, c->stream, sg, fullMask);                   \
    }
                        if (triXFused)
                        {
                            SG_LAUNCH(k_smooth_grad_tri)
                        }
                        else
                        {
                            SG_LAUNCH(k_smooth_grad)
                        }
#undef SG_LAUNCH
                    }
                    grid.x -= 1;
                    grid.y = unsigned(sa.nSeg);
                    if (grid.x == 0)
                    {
                        return;
                    }
                    prof(c, "k_smooth_vec");
                }
                if (halfNext)
                {
                    hipLaunchKernelGGL((k_smooth_vec<true>), grid, dim3(nt), ldsB, c->stream, sa, fullMask);
                }
                else
                {
                    hipLaunchKernelGGL((k_smooth_vec<false>), grid, dim3(nt), ldsB, c->stream, sa, fullMask);
                }
            };
            const int nSegMax = std::max(nSeg, wantGrad ? nSegG : nSeg);
            // (the repair flags are zero here: set by k_smooth_verify, taken down by the repair launch that reads them)
            launchSv(dim3(d, nSeg, nF));
            LAUNCHCHK(c, "k_smooth_vec");
            if (nSegMax > 1)
            {
                // the segments' hand-overs, bit for bit; planes with a difference are recomputed as one chain
                hipLaunchKernelGGL(k_smooth_verify, dim3(nSegMax - 1, d, nF), dim3(256), 0, c->stream, (const float*)c->d_specState, (const float*)c->d_trueState,
                    rs.h, sa.segStride, d, c->d_redo, c->smoothForceRedo, nSeg, wantGrad ? p.colorChn : -1, nSegG);
                LAUNCHCHK(c, "k_smooth_verify");
                if (c->countRepairs)
                {
                    std::vector<int32_t> fl(size_t(nF) * d);
                    HIPCHK(c, hipMemcpyAsync(fl.data(), c->d_redo, fl.size() * 4, hipMemcpyDeviceToHost, c->stream));
                    HIPCHK(c, hipStreamSynchronize(c->stream));
                    c->repairs[0] += int64_t(fl.size());
                    for (int32_t v : fl)
                    {
                        c->repairs[1] += v != 0;
                    }
                }
                sa.segW = rs.w;
                sa.warm = 0;
                sa.nSeg = 1;
                sa.redo = c->d_redo;
                launchSv(dim3(d, 1, nF));
                LAUNCHCHK(c, "k_smooth_vec(repair)");
            }
        }
        colorDone = true;
    }
    else if (p.colorSmooth > 0)
    {
        if ((rc = launchSmooth(c, img, rs.sm, c->d_realJobs + k, 1, d, rs.h, img_fs, int64_t(d) * np, nF, pColor, true)))
        {
            return rc;
        }
    }
    else
    {
        hipLaunchKernelGGL(k_copy_planes, dim3(cdiv(np * d, 256), 1, nF), dim3(256), 0, c->stream, img, rs.sm,
            (const SmoothJob*)(c->d_realJobs + k), img_fs, int64_t(d) * np);
        LAUNCHCHK(c, "k_copy_planes");
    }
    return ACF_HIP_OK;
}

// gradMag, convTri(M), gradMagNorm, gradHist and the shrunk channels of a real scale (chnsCompute.cpp:263-331, addChn :340-370)
int PyramidRun::gradientChannels(size_t k, const SmoothOut& so)
{
    int rc;
    RealScale& rs = c->real[k];
    const int64_t np = int64_t(rs.h) * rs.w;
    const bool gradFused = so.gradFused, gradBlocked = so.gradBlocked, triXFused = so.triXFused;
    ChnsArgs a{};
    a.sm = rs.sm;
    a.M = rs.M;
    a.S = rs.S;
    a.O = rs.O;
    a.Mn = c->taps ? rs.Mn : nullptr;
    a.chns = c->d_chns + pl.raw_off[rs.level];
    a.sm_fs = int64_t(d) * np;
    a.m_fs = np;
    a.chns_fs = pl.raw_floats;
    a.h = rs.h;
    a.w = rs.w;
    a.d = d;
    a.colorEnabled = p.colorEnabled;
    a.colorDone = so.colorDone ? 1 : 0;
    a.magEnabled = p.gradMagEnabled;
    a.histEnabled = p.gradHistEnabled;
    a.nOrients = p.nOrients;
    a.doNorm = p.normRad != 0;
    a.full = p.full;
    a.hardBin = p.softBin < 0;
    a.normConst = float(p.normConst);
    a.rq_y = shrinkGainY(shrink);
    // M, O and U of this scale in 64 x 16 blocks when every kernel that touches them is the vector form (triPlan)
    if (d_in == 5 && k == 0)
    {
        // chnsCompute.cpp:219-226,265-269 (chnsPyramid.cpp:318-322): the first real scale takes the M, O planes that came with the image —
        // no gradientMag, no normalisation; magnitude channel and histogram from them (k_chns)
        a.M = frames + 3 * np0;
        a.O = frames + 4 * np0;
        a.S = nullptr;
        a.Mn = nullptr;
        a.m_fs = int64_t(d_in) * np0;
        a.doNorm = 0;
        return launchChns(c, a, shrink, nF);
    }
    const bool gradVec = rs.h % 4 == 0 && np % 4 == 0;
    const bool fuseCells = shrink == 4 && !c->taps; // k_triy_chns; else S is written and k_chns normalises
    a.x86 = x86T(c);
    const bool wantTri = (p.gradMagEnabled || p.gradHistEnabled) && p.normRad;
    const bool blockedMO = wantTri &&
        triPlan(rs.M, rs.U, rs.h, rs.w, p.normRad, np, fuseCells ? &a : nullptr, rs.uFloats, rs.moFloats, gradVec).blocked;
    if (gradFused)
    {
        // (M and O are there already)
        if (gradBlocked != blockedMO)
        {
            return fail(c, ACF_HIP_E_INVALID, "k_smooth_grad: M / O layout differs from the y pass's");
        }
    }
    else if (p.gradMagEnabled || p.gradHistEnabled)
    {
        prof(c, "k_grad_mag");
        if (gradVec)
        {
            // 16 bytes per lane, persistent grid (one 16-wave workgroup per CU around the 80 KB LDS table), grid-stride over (frame, strip, row quad)
            const int64_t items = int64_t(cdiv(rs.w, GMV_XT)) * (rs.h / 4) * nF;
            const int gmvMax = 256;
            const int blocks = int(std::min<int64_t>(gmvMax, (items + GMV_BLOCK - 1) / GMV_BLOCK));
            const float* gsrc = rs.sm + int64_t(p.colorChn) * np;
            const uint32_t* xt = x86T(c);
#define GMV_LAUNCH(BL_, AR_, FS_, NYB_)                                                                                                     \
    hipLaunchKernelGGL((k_grad_mag_vec<BL_, AR_>), dim3(blocks), dim3(GMV_BLOCK), 0, c->stream, gsrc, rs.M, rs.O, (const float*)c->d_acos, rs.h, rs.w, p.full, \
        int64_t(d) * np, FS_, nF, NYB_, xt);
            if (blockedMO)
            {
                if (xt)
                {
                    GMV_LAUNCH(true, true, moBlockedFloats(rs.h, rs.w), (rs.h + 15) / 16)
                }
                else
                {
                    GMV_LAUNCH(true, false, moBlockedFloats(rs.h, rs.w), (rs.h + 15) / 16)
                }
            }
            else if (xt)
            {
                GMV_LAUNCH(false, true, np, 0)
            }
            else
            {
                GMV_LAUNCH(false, false, np, 0)
            }
#undef GMV_LAUNCH
        }
        else
        {
            // enough workgroups to fill the chip twice over, each long enough to amortise its 80 KB table copy
            const int rowBlocks = cdiv(rs.h, GM_ROWS), nStrips = cdiv(rs.w, GM_XT);
            const int want = std::max(1, cdiv(1024, rowBlocks * nF));
            const int spb = std::max(8, cdiv(nStrips, want));
            hipLaunchKernelGGL(k_grad_mag_strip, dim3(rowBlocks, cdiv(nStrips, spb), nF), dim3(GM_ROWS), 0, c->stream,
                (const float*)(rs.sm + int64_t(p.colorChn) * np), rs.M, rs.O, (const float*)c->d_acos, rs.h, rs.w, p.full, int64_t(d) * np, np, spb, x86T(c));
        }
        LAUNCHCHK(c, "k_grad_mag");
    }
    bool cellsDone = false;
    if ((p.gradMagEnabled || p.gradHistEnabled) && p.normRad)
    {
        // convTri(M, normRad): x running sums, then the y pass — fused with the channel cells when the level allows it
        // (S then never reaches HBM), else S is written for k_chns
        if ((rc = launchTri(c, rs.M, rs.U, rs.S, rs.h, rs.w, p.normRad, np, nF, fuseCells ? &a : nullptr, &cellsDone, rs.uFloats, rs.moFloats,
                 blockedMO, triXFused)))
        {
            return rc;
        }
    }
    if (cellsDone)
    {
        return ACF_HIP_OK;
    }
    if ((rc = launchChns(c, a, shrink, nF)))
    {
        return rc;
    }
    return ACF_HIP_OK;
}

int PyramidRun::joinScales()
{
    if (scalePar)
    {
        prof(c, "(end)");
        c->stream = mainStream;
        for (size_t k = 1; k < c->real.size(); k++)
        {
            HIPCHK(c, hipEventRecord(c->evJoin[k - 1], c->side[k - 1]));
            HIPCHK(c, hipStreamWaitEvent(mainStream, c->evJoin[k - 1], 0));
        }
    }
    return ACF_HIP_OK;
}

// ---- approximated levels, smoothing and padding of frames [f0, f0 + nLF) of the batch (chnsPyramid.cpp:385-435)
int PyramidRun::levels(int f0, int nLF)
{
    int rc;
    float* const chnsF = c->d_chns + int64_t(f0) * pl.raw_floats;
    float* const pyrF = c->d_pyr + int64_t(f0) * pl.pyr_floats;
    const int nL = int(pl.levels.size());
    const bool waveSmooth = p.smooth > 0 && c->finalMaxH <= 64 * LEVEL_MAX_R_REAL && c->levelMode != 0;
    const bool fused = waveSmooth && c->fusedOk && c->levelMode == 1;
    if (!fused && c->nApproxDescs > 0)
    {
        // ---- approximated levels: one launch, blockIdx.y = level (chnsPyramid.cpp:385-397)
        prof(c, "k_resample(approx)");
        hipLaunchKernelGGL(k_resample, dim3(c->approxMaxBlocks, c->nApproxDescs, nLF), dim3(64, 4), 0, c->stream,
            (const float*)chnsF, chnsF, (const ResampleDesc*)(c->d_descs + c->nImgDescs), (const int32_t*)c->d_it, (const float*)c->d_ft, RS_XT);
        LAUNCHCHK(c, "k_resample(approx)");
    }
    if (waveSmooth)
    {
        // ---- (approximated-scale resample +) smoothing + placement in the padded pyramid: one wave per plane (k_level)
        const float pS = float(12.0 / p.smooth / (p.smooth + 2.0) - 2.0);
        float* rawOut = (fused && c->taps) ? chnsF : nullptr;
        const LevelJob* ljobs = fused ? c->d_levelJobs : c->d_levelJobsRaw;
        const auto& groups = fused ? c->levelGroups : c->levelGroupsRaw;
        const ResampleDesc* dd = c->d_descs + c->nImgDescs;
        const int lvChain = (c->cascTurns & 4) ? 0 : 1;
        const bool lvTurns = (c->cascTurns & 6) && fused && c->nAllJobs > 0;
        if (lvTurns && (rc = turnBegin(c, 1, lvChain)))
        {
            return rc;
        }
        prof(c, fused ? "k_level(fused)" : "k_level(smooth)");
        // fork: every group is an independent launch (disjoint outputs); biggest planes first
        if (!groups.empty())
        {
            ensureSide(c);
        }
        const size_t nSide = groups.empty() ? 0 : c->side.size();
        if (nSide && c->evFork)
        {
            HIPCHK(c, hipEventRecord(c->evFork, c->stream));
            for (size_t k = 0; k < nSide; k++)
            {
                HIPCHK(c, hipStreamWaitEvent(c->side[k], c->evFork, 0));
            }
        }
        const int nAll = fused ? c->nAllJobs : c->nAllJobsRaw;
        // what the levels leave as: floats (always, unless the caller has declared the float pyramid unneeded), and the
        // cascade's 16-bit rank cells when every level goes through this one launch
        // (the rank cells have one reader, the tile kernel: a plan or option that routes the cascade elsewhere keeps floats)
        const bool emitRank = fused && c->levelsEmitRank && c->cs.useRank && !c->noRank && !c->taps && (c->cs.useTiles || c->cs.useRankD) && !c->noTiles;
        // (depths other than 2 keep the floats: their queue's overflow path reads them)
        const bool emitF32 = !(emitRank && !c->keepPyramid && !c->autoLambdas) || c->cs.useRankD;
        wroteRank = emitRank;
        wroteF32 = emitF32;
        if (nAll > 0)
        {
            LevelRankArgs ra{};
            size_t ldsL = size_t(LEVEL_WAVES) * (emitRank ? LEVEL_ALL_WF_RANK : LEVEL_ALL_WF) * sizeof(float);
            if (emitRank)
            {
                ra.out = c->cs.d_pyrR + int64_t(f0) * c->cs.pyrRCells;
                ra.fs = c->cs.pyrRCells;
                ra.chan = c->cs.d_rankChan;
                ra.rec = c->cs.d_rankRec;
                ldsL += size_t(c->cs.rankMaxRec) * sizeof(RankRec);
            }
            // column segments for small batches (a level's chain of up to wC column steps is then the launch's duration):
            // as many as give the launch ~4 waves per SIMD, at most levelSegCap; every hand-over verified on the device
            int nSegL = 1;
            const int warmL = std::max(4, c->levelWarm / 4 * 4);
            if (fused && !c->taps && nLF <= c->levelSegFrames && c->levelSegments != 1 && !c->autoLambdas)
            {
                const int64_t wavesL = int64_t(nAll) * pl.nChns * nLF;
                nSegL = c->levelSegments > 1 ? c->levelSegments : int((4 * 1024 + wavesL - 1) / wavesL); // 0: ~4 waves per SIMD
                nSegL = std::max(1, std::min(nSegL, c->levelSegCap));
            }
            LevelSegArgs lsa{};
            lsa.nSeg = nSegL;
            lsa.warm = warmL;
            lsa.hMax = c->levelHMax;
            lsa.nJobs = nAll;
            lsa.spec = c->d_lvSpec;
            lsa.tru = c->d_lvTrue;
            lsa.redo = nullptr;
            dim3 lgrid(pl.nChns, cdiv(nLF, LEVEL_WAVES), nAll * nSegL), lblock(64 * LEVEL_WAVES);
#define LVALL_LAUNCH(OUT)                                                                                                   \
if (nSegL > 1)                                                                                                          \
{                                                                                                                       \
    if ((rc = allowLds(c, reinterpret_cast<const void*>(&k_level_all<OUT, 1>), ldsL)))                                  \
        return rc;                                                                                                      \
    hipLaunchKernelGGL((k_level_all<OUT, 1>), lgrid, lblock, ldsL, c->stream, (const float*)chnsF, pyrF, rawOut, ljobs, dd, \
        (const int32_t*)c->d_it, (const float*)c->d_ft, pl.nChns, pl.raw_floats, pl.pyr_floats, pS, c->d_dump, nLF, ra, lsa); \
    hipLaunchKernelGGL(k_level_verify, dim3(nAll * nSegL, pl.nChns, nLF), dim3(64), 0, c->stream, (const float*)c->d_lvSpec,  \
        (const float*)c->d_lvTrue, ljobs, lsa, pl.nChns, c->d_lvRedo, c->smoothForceRedo);                               \
    if (c->countRepairs)                                                                                                \
    {                                                                                                                   \
        std::vector<int32_t> fl(size_t(nLF) * nAll * pl.nChns);                                                         \
        HIPCHK(c, hipMemcpyAsync(fl.data(), c->d_lvRedo, fl.size() * 4, hipMemcpyDeviceToHost, c->stream));             \
        HIPCHK(c, hipStreamSynchronize(c->stream));                                                                     \
        c->repairs[2] += int64_t(fl.size());                                                                            \
        for (int32_t v : fl)                                                                                            \
        {                                                                                                               \
            c->repairs[3] += v != 0;                                                                                    \
        }                                                                                                               \
    }                                                                                                                   \
    lsa.nSeg = 1;                                                                                                       \
    lsa.redo = c->d_lvRedo;                                                                                             \
    hipLaunchKernelGGL((k_level_all<OUT, 1>), dim3(pl.nChns, cdiv(nLF, LEVEL_WAVES), nAll), lblock, ldsL, c->stream,    \
        (const float*)chnsF, pyrF, rawOut, ljobs, dd, (const int32_t*)c->d_it, (const float*)c->d_ft, pl.nChns,         \
        pl.raw_floats, pl.pyr_floats, pS, c->d_dump, nLF, ra, lsa);                                                     \
}                                                                                                                       \
else                                                                                                                    \
{                                                                                                                       \
    if ((rc = allowLds(c, reinterpret_cast<const void*>(&k_level_all<OUT, 0>), ldsL)))                                  \
        return rc;                                                                                                      \
    hipLaunchKernelGGL((k_level_all<OUT, 0>), lgrid, lblock, ldsL, c->stream, (const float*)chnsF, pyrF, rawOut, ljobs, dd, \
        (const int32_t*)c->d_it, (const float*)c->d_ft, pl.nChns, pl.raw_floats, pl.pyr_floats, pS, c->d_dump, nLF, ra, lsa); \
}
            if (emitRank && emitF32)
            {
                LVALL_LAUNCH(LO_F32 | LO_RANK);
            }
            else if (emitRank)
            {
                LVALL_LAUNCH(LO_RANK);
            }
            else
            {
                LVALL_LAUNCH(LO_F32);
            }
#undef LVALL_LAUNCH
            LAUNCHCHK(c, "k_level_all");
            if (lvTurns && (rc = turnEnd(c, 1, lvChain)))
            {
                return rc;
            }
        }
        size_t gi = 0;
        for (auto git = groups.rbegin(); git != groups.rend(); ++git, ++gi)
        {
            const auto& g = *git;
            hipStream_t lst = (nSide && c->evFork) ? c->side[gi % nSide] : c->stream;
            dim3 grid(pl.nChns, g.count, cdiv(nLF, LEVEL_WAVES)), block(64 * LEVEL_WAVES);
#define LV_LAUNCH(RR, MM)                                                                                                         \
    hipLaunchKernelGGL((k_level<RR, MM>), grid, block, 0, lst, (const float*)chnsF, pyrF, rawOut, ljobs + g.first, dd, \
        (const int32_t*)c->d_it, (const float*)c->d_ft, pl.nChns, pl.raw_floats, pl.pyr_floats, pS, c->d_dump, nLF);
#define LV_MODES(RR)                                  \
    switch (g.mode)                                   \
    {                                                 \
        case LM_REAL: LV_LAUNCH(RR, LM_REAL); break;  \
        case LM_DD: LV_LAUNCH(RR, LM_DD); break;      \
        default: LV_LAUNCH(RR, LM_UU); break;         \
    }
            switch (g.R)
            {
                case 1: LV_MODES(1); break;
                case 2: LV_MODES(2); break;
                case 3: LV_MODES(3); break;
                case 4: LV_MODES(4); break;
                case 5: LV_MODES(5); break;
                case 6: LV_MODES(6); break;
                case 7: LV_MODES(7); break;
                case 8: LV_MODES(8); break;
                default:
                    // nine rows per lane: the full-resolution level of a 4K frame (540 cells); a real level's copy + smoothing only
                    if (g.R != LEVEL_MAX_R_REAL || g.mode != LM_REAL)
                    {
                        return fail(c, ACF_HIP_E_UNSUPPORTED, "pyramid: level taller than the level kernels' rows per lane");
                    }
                    LV_LAUNCH(9, LM_REAL);
                    break;
            }
#undef LV_MODES
#undef LV_LAUNCH
            LAUNCHCHK(c, "k_level");
        }
        if (nSide && c->evFork)
        {
            for (size_t k = 0; k < std::min(nSide, groups.size()); k++)
            {
                HIPCHK(c, hipEventRecord(c->evJoin[k], c->side[k]));
                HIPCHK(c, hipStreamWaitEvent(c->stream, c->evJoin[k], 0));
            }
        }
    }
    // ---- smooth every plane of every level into the fused, padded pyramid (chnsPyramid.cpp:399-435)
    if (waveSmooth)
    {
    }
    else if (p.smooth > 0)
    {
        const float pS = float(12.0 / p.smooth / (p.smooth + 2.0) - 2.0);
        if ((rc = launchSmooth(c, chnsF, pyrF, c->d_finalJobs, nL, pl.nChns, c->finalMaxH, pl.raw_floats, pl.pyr_floats, nLF, pS, true)))
        {
            return rc;
        }
    }
    else
    {
        int64_t maxE = 0;
        for (const auto& l : pl.levels)
        {
            maxE = std::max<int64_t>(maxE, int64_t(pl.nChns) * l.hC * l.wC);
        }
        hipLaunchKernelGGL(k_copy_planes, dim3(cdiv(maxE, 256), nL, nLF), dim3(256), 0, c->stream, (const float*)chnsF, pyrF,
            (const SmoothJob*)c->d_finalJobs, pl.raw_floats, pl.pyr_floats);
        LAUNCHCHK(c, "k_copy_planes(final)");
    }
    if (p.pad_h / shrink > 0 || p.pad_w / shrink > 0)
    {
        prof(c, "k_pad_reflect");
        if (wroteF32)
        {
            hipLaunchKernelGGL(k_pad_reflect<float>, dim3(cdiv(c->padMaxElems, 256), nL, nLF), dim3(256), 0, c->stream, pyrF, (const PadJob*)c->d_padJobs, pl.pyr_floats);
        }
        if (wroteRank)
        {
            hipLaunchKernelGGL(k_pad_reflect<uint16_t>, dim3(cdiv(c->padMaxElems, 256), nL, nLF), dim3(256), 0, c->stream,
                c->cs.d_pyrR + int64_t(f0) * c->cs.pyrRCells, (const PadJob*)c->d_padJobsR, c->cs.pyrRCells);
        }
        LAUNCHCHK(c, "k_pad_reflect");
    }
    return ACF_HIP_OK;
}

// Image-specific lambdas (chnsPyramid.cpp:341-374), then the levels frame by frame with that frame's gains
int PyramidRun::levelsWithImageLambdas()
{
    int rc;
    // Image-specific lambdas (chnsPyramid.cpp:341-374): per-type means of the raw channels at two real levels (f64 plane
    // sums on the device, k_plane_sums), lambda = -log2(f0/f1) / log2(s0/s1) on the host with the C library the
    // oracle uses, then the approximated levels frame by frame with that frame's gains written into the descriptors
    // (stream-ordered copies).  A fallback path for models that ship without lambdas: correctness first — it
    // synchronises once per batch and launches per frame.
    prof(c, "k_plane_sums");
    const int lv0 = pl.lambdaLevel[0], lv1 = pl.lambdaLevel[1];
    SumJob j0{ pl.raw_off[size_t(lv0)], pl.levels[size_t(lv0)].hC * pl.levels[size_t(lv0)].wC, 0 };
    SumJob j1{ pl.raw_off[size_t(lv1)], pl.levels[size_t(lv1)].hC * pl.levels[size_t(lv1)].wC, 0 };
    hipLaunchKernelGGL(k_plane_sums, dim3(pl.nChns, 2, nF), dim3(256), 0, c->stream, (const float*)c->d_chns, pl.raw_floats, j0, j1, pl.nChns, c->d_planeSums);
    LAUNCHCHK(c, "k_plane_sums");
    std::vector<double> sums(size_t(nF) * 2 * pl.nChns);
    HIPCHK(c, hipMemcpyAsync(sums.data(), c->d_planeSums, sums.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int nColorL = p.colorEnabled ? d : 0, nMagL = p.gradMagEnabled ? 1 : 0, nHistL = p.gradHistEnabled ? p.nOrients : 0;
    const int nTypeCh[3] = { nColorL, nMagL, nHistL };
    // one copy of the approximated levels' descriptors per frame: each stays untouched until its stream-ordered upload is done
    std::vector<std::vector<ResampleDesc>> descsAll(size_t(nF),
        std::vector<ResampleDesc>(c->h_descs.begin() + c->nImgDescs, c->h_descs.begin() + c->nImgDescs + c->nApproxDescs));
    for (int f = 0; f < nF; f++)
    {
        std::vector<ResampleDesc>& descs = descsAll[size_t(f)];
        double lam[3] = { 0, 0, 0 };
        int z0 = 0;
        for (int j = 0; j < 3; j++)
        {
            if (!nTypeCh[j])
            {
                continue;
            }
            double s0 = 0, s1 = 0; // sum(MatP): the per-plane sums added in plane order (MatP.cpp:97-106)
            for (int k = 0; k < nTypeCh[j]; k++)
            {
                s0 += sums[(size_t(f) * 2 + 0) * pl.nChns + z0 + k];
                s1 += sums[(size_t(f) * 2 + 1) * pl.nChns + z0 + k];
            }
            const double f0 = s0 / (double(nTypeCh[j]) * j0.cells), f1 = s1 / (double(nTypeCh[j]) * j1.cells);
            lam[j] = -(std::log(f0 / f1) / std::log(2.0)) / (std::log(pl.levels[size_t(lv0)].scale / pl.levels[size_t(lv1)].scale) / std::log(2.0));
            z0 += nTypeCh[j];
        }
        for (int j = 0; j < 3; j++)
        {
            c->h_lambdas[size_t(f) * 3 + j] = lam[j];
        }
        size_t ai = 0;
        for (size_t i = 0; i < pl.levels.size(); i++)
        {
            const acf_hip_level& l = pl.levels[i];
            if (l.isReal)
            {
                continue;
            }
            const acf_hip_level& lr = pl.levels[size_t(l.realIndex)];
            double ratio[3];
            for (int j = 0; j < 3; j++)
            {
                ratio[j] = std::pow(l.scale / lr.scale, -lam[j]); // :393
            }
            setResampleGain(descs[ai], ratio, nColorL, nColorL + nMagL);
            ai++;
        }
        if (!descs.empty())
        {
            HIPCHK(c, hipMemcpyAsync(c->d_descs + c->nImgDescs, descs.data(), descs.size() * sizeof(ResampleDesc), hipMemcpyHostToDevice, c->stream));
        }
        if ((rc = levels(f, 1)))
        {
            return rc;
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream)); // descsAll is read by the uploads above
    return ACF_HIP_OK;
}

int pyramidBody(acf_hip_ctx* c, const float* frames, const PackedSrc* u8, int nF)
{
    if (c && !c->kids.empty())
    {
        if (u8)
        {
            return fail(c, ACF_HIP_E_UNSUPPORTED, "pyramid_u8: not available with option \"streams\" > 1");
        }
        if (!frames || nF <= 0 || nF > c->maxBatch)
        {
            return fail(c, ACF_HIP_E_INVALID, "pyramid: n_frames out of range");
        }
        const size_t per = size_t(c->plan.d_in) * c->plan.H * c->plan.W;
        int rc = kidsFork(c);
        for (size_t i = 0; i < c->kids.size() && !rc; i++)
        {
            const int n = kidCount(c, i, nF);
            if (n > 0 && (rc = acf_hip_pyramid(c->kids[i], frames + i * size_t(c->kidChunk) * per, n)))
            {
                return kidFail(c, c->kids[i], rc);
            }
        }
        c->lastBatch = nF;
        c->pyramidValid = true;
        c->detectValid = false;
        return rc ? rc : kidsJoin(c);
    }
    if (!c || !c->hasPlan)
    {
        return c ? fail(c, ACF_HIP_E_NOPLAN, "pyramid: plan first") : ACF_HIP_E_INVALID;
    }
    if ((!frames && !u8) || (u8 && !u8->frames) || nF <= 0 || nF > c->maxBatch)
    {
        return fail(c, ACF_HIP_E_INVALID, "pyramid: n_frames out of range");
    }
    HIPCHK(c, hipSetDevice(c->device));
    c->pyramidValid = c->detectValid = false;
    int rc;
    PyramidRun run(c, frames, nF);
    bool ingestConverted = false; // the 8-bit ingest wrote the converted colour planes itself
    if (u8 && (rc = run.ingest(u8, ingestConverted)))
    {
        return rc;
    }
    c->lastFrames = run.frames;
    if ((rc = run.colour(ingestConverted)) || (rc = run.prepareScales()))
    {
        return rc;
    }
    for (size_t k = 0; k < c->real.size(); k++)
    {
        if ((rc = run.realScale(k)))
        {
            return rc;
        }
    }
    if ((rc = run.joinScales()))
    {
        return rc;
    }
    if (!c->autoLambdas)
    {
        for (int f = 0; f < nF; f++)
        {
            c->h_lambdas[size_t(f) * 3 + 0] = c->p.lambdas[0];
            c->h_lambdas[size_t(f) * 3 + 1] = c->p.lambdas[1];
            c->h_lambdas[size_t(f) * 3 + 2] = c->p.lambdas[2];
        }
        if ((rc = run.levels(0, nF)))
        {
            return rc;
        }
    }
    else if ((rc = run.levelsWithImageLambdas()))
    {
        return rc;
    }
    prof(c, "(end)");
    c->lastBatch = nF;
    c->pyramidValid = true;
    c->ranksValid = run.wroteRank;   // else the cascade converts the float pyramid first (k_rank)
    c->floatPyramid = run.wroteF32;
    return ACF_HIP_OK;
}
} // namespace
