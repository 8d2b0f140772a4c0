// acf_hip_detect — C++ host CLI over acf::HipDetector (the role of the
// reference's acf-detect app, src/app/acf/acf.cpp, for this path only).
//
//   acf_hip_detect --model m.acfm --frames f.raw --rows W --cols H --channels d --count N
//                  [--luv] [--nms] [--batch] [--via-pyramid] [--max-count K] [--prune-ratio R]
//   acf_hip_detect --model m.acfm --frames f.u8 --u8 rgb|bgr|rgba|bgra|gray --rows H --cols W --count N [--stream B] [--min-width M] ...
//                  packed 8-bit upright frames; --stream B: batches of B frames through streamSubmit/streamCollect
//   acf_hip_detect --model m.acfm --frames f.raw --rows W --cols H --channels d --count N --chns out.raw [--luv] [--log-taps] [--defaults]
//                  Detector::chnsCompute per frame (static; --defaults: computeChannels' toolbox defaults instead of the model's Chns):
//                  the channels [nChns][W/shrink][H/shrink] of every frame appended to out.raw, one "chns ..." line per frame
//   ... --ref-arith host|tables.bin   the reference's rcpps / rsqrtps bits (probed from this CPU, or 4096 + 8192 uint32 from a file)
//   ... --log-levels                  Detector::setLogger: a "level <tag> <hash>" line per pyramid level
//   acf_hip_detect --convert in.acfm|in.cpb --out out.cpb                                  (model file conversion, no GPU)
//   acf_hip_detect --dump-defaults                                                         (default Options tree, no GPU)
//   acf_hip_detect --nms-only boxes.txt [--type maxg] [--overlap .65] [--ovrdnm min]   (host logic only, no GPU)
//
// Model file ("ACFHIPM1", written by acf_amd/modelio.py): text header of
// "key value" lines terminated by "END", then raw little-endian arrays
// fids u32, thrs f32, hs f32, child u32, each nTrees*nTreeNodes.
// Frames: raw f32, transposed planar [count][channels][rows=W][cols=H].
// Output: "frame i n" then n lines "x y w h score scorebits".
#include "HipDetector.h"
#include "ModelIO.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>

using acf::HipDetector;

static bool loadModel(const std::string& path, HipDetector::Options& o, HipDetector::Classifier& c)
{
    return acf::loadModelAny(path, o, c) == 0; // "*.cpb": the reference's cereal files; otherwise the ACFHIPM1 container
}

static uint64_t fnv(const acf::MatP& m)
{
    uint64_t hsh = 1469598103934665603ull;
    const unsigned char* b = reinterpret_cast<const unsigned char*>(m.data());
    for (size_t i = 0; i < m.numel() * 4; i++)
    {
        hsh = (hsh ^ b[i]) * 1099511628211ull;
    }
    return hsh;
}

static void printFrame(int f, const HipDetector::RectVec& objs, const HipDetector::RealVec& scores)
{
    std::printf("frame %d %zu\n", f, objs.size());
    for (size_t i = 0; i < objs.size(); i++)
    {
        const float s = float(scores[i]);
        uint32_t b;
        std::memcpy(&b, &s, 4);
        std::printf("%d %d %d %d %.9g %08x\n", objs[i].x, objs[i].y, objs[i].width, objs[i].height, scores[i], b);
    }
}

int main(int argc, char** argv)
{
    std::map<std::string, std::string> a;
    for (int i = 1; i < argc; i++)
    {
        std::string k = argv[i];
        if (k.rfind("--", 0) == 0)
        {
            if (i + 1 < argc && std::string(argv[i + 1]).rfind("--", 0) != 0)
            {
                a[k.substr(2)] = argv[++i];
            }
            else
            {
                a[k.substr(2)] = "1";
            }
        }
    }
    try
    {
        if (a.count("nms-only"))
        {
            HipDetector::Options::Nms p;
            if (a.count("type")) p.type = a["type"];
            if (a.count("overlap")) p.overlap = std::stod(a["overlap"]);
            if (a.count("ovrdnm")) p.ovrDnm = a["ovrdnm"];
            if (a.count("thr")) p.thr = std::stod(a["thr"]);
            std::ifstream is(a["nms-only"]);
            HipDetector::DetectionVec in, out;
            HipDetector::Detection d;
            while (is >> d.roi.x >> d.roi.y >> d.roi.width >> d.roi.height >> d.score)
            {
                in.push_back(d);
            }
            HipDetector::bbNms(in, p, out);
            HipDetector det; // prune() needs only the knobs
            if (a.count("max-count")) det.setMaxDetectionCount(size_t(std::stoul(a["max-count"])));
            if (a.count("prune-ratio")) det.setDetectionScorePruneRatio(std::stod(a["prune-ratio"]));
            HipDetector::RectVec objs;
            HipDetector::RealVec scores;
            for (auto& b : out)
            {
                objs.push_back(b.roi);
                scores.push_back(b.score);
            }
            if (a.count("prune"))
            {
                det.prune(objs, scores);
            }
            printFrame(0, objs, scores);
            return 0;
        }
        HipDetector::Options o;
        HipDetector::Classifier c;
        if (a.count("dump-defaults"))
        {
            // the default Options tree (what Detector::initializeOpts / chnsCompute / chnsPyramid fill in, ACF.cpp:48-113,
            // chnsCompute.cpp:161-203, chnsPyramid.cpp:183-215): host only
            const auto& py = o.pPyramid;
            const auto& ch = py.pChns;
            std::printf("shrink %d\ncolor.enabled %d\ncolor.smooth %g\ncolor.colorSpace %s\n", ch.shrink, ch.pColor.enabled, ch.pColor.smooth, ch.pColor.colorSpace.c_str());
            std::printf("gradMag.enabled %d\ngradMag.colorChn %d\ngradMag.normRad %d\ngradMag.normConst %g\ngradMag.full %d\n", ch.pGradMag.enabled,
                ch.pGradMag.colorChn, ch.pGradMag.normRad, ch.pGradMag.normConst, ch.pGradMag.full);
            std::printf("gradHist.enabled %d\ngradHist.binSize %d\ngradHist.nOrients %d\ngradHist.softBin %d\n", ch.pGradHist.enabled, ch.pGradHist.binSize,
                ch.pGradHist.nOrients, ch.pGradHist.softBin);
            std::printf("nPerOct %d\nnOctUp %d\nnApprox %d\npad %d %d\nminDs %d %d\nsmooth %g\n", py.nPerOct, py.nOctUp, py.nApprox, py.pad.width, py.pad.height,
                py.minDs.width, py.minDs.height, py.smooth);
            std::printf("stride %d\ncascThr %g\nnms.type %s\nnms.overlap %g\nnms.ovrDnm %s\n", o.stride, o.cascThr, o.pNms.type.c_str(), o.pNms.overlap, o.pNms.ovrDnm.c_str());
            HipDetector det0;
            std::printf("maxDetectionCount 10\ngood %d\n", det0.good() ? 1 : 0);
            return 0;
        }
        if (a.count("convert"))
        {
            // acf_hip_detect --convert in.{acfm,cpb} --out out.cpb : host only
            if (!loadModel(a["convert"], o, c))
            {
                std::fprintf(stderr, "cannot read model %s\n", a["convert"].c_str());
                return 2;
            }
            std::ofstream os(a.at("out"), std::ios::binary);
            acf::saveCpb(os, o, c);
            return os ? 0 : 2;
        }
        HipDetector det(a.at("model")); // Detector(filename): "*.cpb" or the ACFHIPM1 container
        if (!det.good())
        {
            std::fprintf(stderr, "detector not good (no device / bad model)\n");
            return 3;
        }
        det.setIsLuv(a.count("luv") != 0);
        det.setDoNonMaximaSuppression(a.count("nms") != 0);
        if (a.count("max-count")) det.setMaxDetectionCount(size_t(std::stoul(a["max-count"])));
        if (a.count("prune-ratio")) det.setDetectionScorePruneRatio(std::stod(a["prune-ratio"]));
        if (a.count("min-width")) det.setMinObjectWidth(std::stoi(a["min-width"])); // the apps' Resizer (acf.cpp:117-148), 8-bit entries
        if (a.count("casc-cal"))
        {
            HipDetector::Modify m;
            m.has_cascCal = true;
            m.cascCal = std::stod(a["casc-cal"]);
            det.acfModify(m);
        }
        std::vector<uint32_t> arithTables;
        if (a.count("ref-arith"))
        {
            if (a["ref-arith"] == "host")
            {
                det.setReferenceArithmetic(true);
                std::vector<uint32_t> r, q;
                HipDetector::probeHostArithmetic(r, q);
                arithTables = r;
                arithTables.insert(arithTables.end(), q.begin(), q.end());
            }
            else
            {
                arithTables.resize(12288);
                std::ifstream is(a["ref-arith"], std::ios::binary);
                is.read(reinterpret_cast<char*>(arithTables.data()), 12288 * 4);
                if (!is)
                {
                    std::fprintf(stderr, "short table file\n");
                    return 2;
                }
                det.setReferenceArithmetic(arithTables.data(), arithTables.data() + 4096);
            }
        }
        if (a.count("log-levels"))
        {
            det.setLogger([](const acf::MatP& m, const std::string& t) {
                std::printf("level %s %dx%d %016llx\n", t.c_str(), m.cols(), m.rows(), static_cast<unsigned long long>(fnv(m)));
                return 0;
            });
        }
        if (a.count("u8"))
        {
            static const std::map<std::string, int> kPix = { { "rgb", ACF_HIP_PIX_RGB }, { "bgr", ACF_HIP_PIX_BGR }, { "rgba", ACF_HIP_PIX_RGBA },
                { "bgra", ACF_HIP_PIX_BGRA }, { "gray", ACF_HIP_PIX_GRAY } };
            const int pix = kPix.at(a["u8"]);
            const int cpp = pix == ACF_HIP_PIX_GRAY ? 1 : (pix == ACF_HIP_PIX_RGBA || pix == ACF_HIP_PIX_BGRA) ? 4 : 3;
            const int H = std::stoi(a.at("rows")), W = std::stoi(a.at("cols")), cnt = std::stoi(a.at("count"));
            const size_t per = size_t(H) * W * cpp;
            std::vector<uint8_t> frames(per * size_t(cnt));
            std::ifstream is(a.at("frames"), std::ios::binary);
            is.read(reinterpret_cast<char*>(frames.data()), std::streamsize(frames.size()));
            if (!is)
            {
                std::fprintf(stderr, "short frames file\n");
                return 2;
            }
            if (a.count("stream"))
            {
                const int B = std::stoi(a["stream"]), depth = 2;
                det.streamOpen(H, W, pix, 0, B, depth, 8192);
                uint8_t* pin[2] = { static_cast<uint8_t*>(HipDetector::pinnedAlloc(per * size_t(B))), static_cast<uint8_t*>(HipDetector::pinnedAlloc(per * size_t(B))) };
                std::vector<std::pair<int, int>> inflight; // ticket, first frame
                auto drain = [&]() {
                    std::vector<HipDetector::RectVec> objs;
                    std::vector<HipDetector::RealVec> scores;
                    det.streamCollect(inflight.front().first, objs, &scores);
                    for (size_t f = 0; f < objs.size(); f++)
                    {
                        printFrame(inflight.front().second + int(f), objs[f], scores[f]);
                    }
                    inflight.erase(inflight.begin());
                };
                int k = 0;
                for (int f0 = 0; f0 < cnt; f0 += B, k++)
                {
                    if (int(inflight.size()) == depth)
                    {
                        drain();
                    }
                    const int n = std::min(B, cnt - f0);
                    std::memcpy(pin[k % 2], frames.data() + per * size_t(f0), per * size_t(n));
                    inflight.emplace_back(det.streamSubmit(pin[k % 2], n), f0);
                }
                while (!inflight.empty())
                {
                    drain();
                }
                det.streamClose();
                HipDetector::pinnedFree(pin[0]);
                HipDetector::pinnedFree(pin[1]);
            }
            else
            {
                for (int f = 0; f < cnt; f++)
                {
                    HipDetector::RectVec objs;
                    HipDetector::RealVec scores;
                    det(frames.data() + per * size_t(f), H, W, pix, 0, objs, &scores);
                    printFrame(f, objs, scores);
                }
            }
            return 0;
        }
        const int rows = std::stoi(a.at("rows")), cols = std::stoi(a.at("cols")), ch = std::stoi(a.at("channels")), cnt = std::stoi(a.at("count"));
        std::vector<float> frames(size_t(rows) * cols * ch * cnt);
        {
            std::ifstream is(a.at("frames"), std::ios::binary);
            is.read(reinterpret_cast<char*>(frames.data()), std::streamsize(frames.size() * 4));
            if (!is)
            {
                std::fprintf(stderr, "short frames file\n");
                return 2;
            }
        }
        const size_t per = size_t(rows) * cols * ch;
        if (a.count("chns"))
        {
            // Detector::chnsCompute / computeChannels (static, ACF.h:342-349,419-420) on every frame
            if (!arithTables.empty())
            {
                HipDetector::setChnsComputeReferenceArithmetic(arithTables.data(), arithTables.data() + 4096);
            }
            std::ofstream os(a["chns"], std::ios::binary);
            for (int f = 0; f < cnt; f++)
            {
                acf::MatP Ip(rows, cols, ch, frames.data() + per * size_t(f));
                HipDetector::MatLoggerType logger;
                if (a.count("log-taps"))
                {
                    logger = [](const acf::MatP& m, const std::string& t) {
                        std::printf("tap %s %016llx\n", t.c_str(), static_cast<unsigned long long>(fnv(m)));
                        return 0;
                    };
                }
                if (a.count("defaults"))
                {
                    acf::MatP fused;
                    HipDetector::computeChannels(Ip, fused, logger);
                    std::printf("chns %d fused %dx%d\n", f, fused.cols(), fused.rows());
                    os.write(reinterpret_cast<const char*>(fused.data()), std::streamsize(fused.numel() * 4));
                    continue;
                }
                HipDetector::Options::Pyramid::Chns pc = det.opts.pPyramid.pChns;
                pc.isLuv = a.count("luv") != 0;
                HipDetector::Channels chns;
                HipDetector::chnsCompute(Ip, pc, chns, false, logger);
                std::printf("chns %d types %d", f, chns.nTypes);
                for (size_t t = 0; t < chns.data.size(); t++)
                {
                    std::printf(" [%s|%d|%s|%dx%d]", chns.info[t].name.c_str(), chns.info[t].nChns, chns.info[t].padWith.c_str(), chns.data[t].cols(), chns.data[t].rows());
                    os.write(reinterpret_cast<const char*>(chns.data[t].data()), std::streamsize(chns.data[t].numel() * 4));
                }
                std::printf("\n");
            }
            return os ? 0 : 2;
        }
        if (a.count("pool"))
        {
            // acf::HipDetectorPool: one detector per visible device ("--pool N": the first N devices, several times the same
            // device allowed through --pool-devices 0,0: contexts are independent), frames in contiguous blocks
            std::vector<int> devs;
            if (a.count("pool-devices"))
            {
                std::stringstream ss(a["pool-devices"]);
                std::string tok;
                while (std::getline(ss, tok, ','))
                {
                    devs.push_back(std::stoi(tok));
                }
            }
            acf::HipDetectorPool pool(a["model"], devs);
            pool.setIsLuv(a.count("luv") != 0);
            pool.setDoNonMaximaSuppression(a.count("nms") != 0);
            if (a.count("max-count")) pool.setMaxDetectionCount(size_t(std::stoul(a["max-count"])));
            std::vector<HipDetector::RectVec> objs;
            std::vector<HipDetector::RealVec> scores;
            pool.detectBatch(frames.data(), cnt, rows, cols, ch, objs, &scores);
            std::fprintf(stderr, "pool: %zu detector(s)\n", pool.size());
            for (int f = 0; f < cnt; f++)
            {
                printFrame(f, objs[size_t(f)], scores[size_t(f)]);
            }
        }
        else if (a.count("batch"))
        {
            std::vector<HipDetector::RectVec> objs;
            std::vector<HipDetector::RealVec> scores;
            det.detectBatch(frames.data(), cnt, rows, cols, ch, objs, &scores);
            for (int f = 0; f < cnt; f++)
            {
                printFrame(f, objs[size_t(f)], scores[size_t(f)]);
            }
        }
        else
        {
            for (int f = 0; f < cnt; f++)
            {
                acf::MatP Ip(rows, cols, ch, frames.data() + per * size_t(f));
                HipDetector::RectVec objs;
                HipDetector::RealVec scores;
                if (a.count("log-taps"))
                {
                    // chnsPyramid with a MatLoggerType: print every tag with an FNV-1a hash of the plane's bytes
                    HipDetector::Pyramid P;
                    det.chnsPyramid(Ip, nullptr, P, true, [](const acf::MatP& m, const std::string& t) {
                        uint64_t hsh = 1469598103934665603ull;
                        const unsigned char* b = reinterpret_cast<const unsigned char*>(m.data());
                        for (size_t i = 0; i < m.numel() * 4; i++)
                        {
                            hsh = (hsh ^ b[i]) * 1099511628211ull;
                        }
                        std::printf("tap %s %016llx\n", t.c_str(), static_cast<unsigned long long>(hsh));
                        return 0;
                    });
                    continue;
                }
                if (a.count("via-atlas"))
                {
                    // the Pyramid a GL-style backend hands over: every level as ONE atlas plane with a roi per channel
                    // (ACF.h:377-378, acfDetect1.cpp:346-366): channels side by side with a gap, row stride > channel rows
                    HipDetector::Pyramid P, A;
                    det.computePyramid(Ip, P);
                    A = P;
                    A.rois.assign(size_t(P.nScales), {});
                    for (int i = 0; i < P.nScales; i++)
                    {
                        const acf::MatP& L = P.data[size_t(i)][0];
                        const int hP = L.cols(), wP = L.rows() / P.nChns;
                        // channels side by side along the atlas rows (GPUACF's texture layout): element (c, r) of channel z at
                        // c * rowStride + z * chnStride + r
                        const int chnStride = hP + 3, rowStride = P.nChns * chnStride;
                        acf::MatP view(wP, rowStride, 1);
                        std::fill(view.data(), view.data() + view.numel(), -1.f);
                        for (int z = 0; z < P.nChns; z++)
                        {
                            for (int c = 0; c < wP; c++)
                            {
                                std::memcpy(view.data() + size_t(c) * rowStride + size_t(z) * chnStride, L.data() + (size_t(z) * wP + c) * hP, sizeof(float) * hP);
                            }
                            A.rois[size_t(i)].push_back(acf::Rect(z * chnStride, 0, hP, wP));
                        }
                        A.data[size_t(i)][0] = view;
                    }
                    det(A, objs, &scores);
                }
                else if (a.count("via-pyramid"))
                {
                    // computePyramid -> host Pyramid -> another image through the detector -> operator()(Pyramid): the
                    // detections must be P's (the reference re-runs acfDetect1 on the pyramid it is handed, ACF.cpp:268-367)
                    HipDetector::Pyramid P;
                    det.computePyramid(Ip, P);
                    if (cnt > 1)
                    {
                        acf::MatP other(rows, cols, ch, frames.data() + per * size_t((f + 1) % cnt));
                        HipDetector::RectVec o2;
                        det(other, o2, nullptr);
                    }
                    det(P, objs, &scores);
                }
                else
                {
                    det(Ip, objs, &scores);
                }
                printFrame(f, objs, scores);
            }
        }
    }
    catch (const std::exception& e)
    {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
