// acf_hip.hip — implementation of the C ABI in include/acf_hip.h: context,
// planning, buffer ownership and kernel launches.  All device work is
// enqueued on the context's stream; nothing here falls back to the CPU.
#include "kernels.hip.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

using namespace acfhip;

namespace
{

struct DevBuf
{
    void* p = nullptr;
    size_t bytes = 0;
};

// LDS-tile plan of a down-sampling descriptor (resampleTilePlan; k_ldcf_tile)
struct ResampleTiling
{
    int rows = 0, cols = 0, xo = 0, tile_y = 0, tile_x = 0;
};

// k_resample_strip plan of a down-sampling image resample, alone or together with the next real scale's (same source)
struct StripPlan
{
    bool ok = false;
    int yt = 0, nty = 0, nSteps = 0, ntyB = 0, nStepsB = 0, tileY = 0, tileX = 0, rowsP = 0, maxCols = 0, slowRows = 0, fillRounds = 0, tileFloats = 0;
    uint32_t magic = 0;
    size_t lds = 0;
};

struct RealScale
{
    int level = 0, h = 0, w = 0;
    bool resampled = false; // image produced by imResample (else it is the current I itself)
    bool adoptAsI = false;  // after this scale, I := smoothed image of this scale
    int src_h = 0, src_w = 0;
    int descIndex = -1;
    StripPlan strip, stripPair;     // k_resample_strip: this scale's image alone / together with the next real scale's (stripPair.ok)
    float *img = nullptr, *sm = nullptr, *M = nullptr, *O = nullptr, *U = nullptr, *S = nullptr, *Mn = nullptr;
    int64_t uFloats = 0, moFloats = 0; // floats per frame of U and of M, O (room for the blocked layouts' padding)
};

} // namespace

// Everything the cascade launch needs on the device.
struct CascState
{
    CascLevel* d_cascLevels = nullptr;
    int32_t* d_blockLevel = nullptr;
    int blocksPerFrame = 0;
    uint32_t* d_cidAll = nullptr;
    float *d_thrs = nullptr, *d_hs = nullptr;
    uint32_t* d_child = nullptr;
    uint32_t* d_fids = nullptr;
    CascNode2* d_nodes2 = nullptr;
    acf_hip_hit *d_hits = nullptr, *d_sorted = nullptr;
    // stride < shrink: the cascade runs on the grid of distinct window offsets (kernels.hip.h, k_expand_hits)
    int dedupQ = 1;                  // shrink / stride when that is an integer > 1, else 1
    int2* d_realWin = nullptr;       // [level] {nWinR, nWinC} of the real window grid
    acf_hip_hit* d_hitsX = nullptr;  // the expanded hits (k_sort_map's input)
    acf_hip_detection* d_dets = nullptr;
    int32_t* d_counts = nullptr;
    uint2* d_queue[2] = { nullptr, nullptr }; // survivor queues between cascade stages
    int32_t* d_qcounts = nullptr;             // [stage][frame]
    int qcap = 0;
    // LDS-tiled path (depth-2 models, stride a multiple of shrink)
    bool useTiles = false;
    CascTile* d_tiles = nullptr;
    int nTiles = 0;
    TreeNode* d_tileNodes = nullptr; // offsets in the tile's LDS layout, every tree
    uint32_t* d_tileNodesS = nullptr; // stage A of k_cascade_tile3: 40 dwords per batch of four trees
    int aTB = 4;                      // trees per stage-A batch of the tile kernels
    TreeNode* d_tailNodes = nullptr; // offsets = feature ids (window-local layout), all trees
    TileGeom geom{};
    int tailWaves = 0;
    float* d_tailScratch = nullptr; // k_cascade_tail3 leaf matrices: [blocks][tailWaves][TAIL_G][tailPad]
    int tailPad = 0, tailSlab = 0, tailBlocks = 0, tailNodesLds = 0;
    // stage E of the tile kernels + k_tail_scan: leaf codes of the first codeCap queue entries per frame
    uint8_t* d_tailCodes = nullptr;
    int codeCap = 0, codePitch = 0;
    // fixed depths other than 2 (k_cascade_tileD): stage 0 of the staged path on float tiles
    bool useTileD = false;
    CascTile* d_tilesD = nullptr;
    uint32_t* d_tileOffD = nullptr;
    // the same tiles over rank cells (k_cascade_tile3D<.., CellRank>)
    bool useRankD = false;
    TileGeom geomDR{};
    CascTile* d_tilesDR = nullptr;
    int nTilesDR = 0;
    uint32_t *d_nodesDR = nullptr, *d_tileOffDR = nullptr, *d_thrsRankD = nullptr;
    int nTilesD = 0, tbD = 0, t1D = 0;
    TileGeom geomD{};
    uint32_t* d_nodesD = nullptr;
    // fixed depths 1..4, last stage [128, nTrees) of the staged path as leaf codes + ordered scan (k_tail_codesD / k_tail_scanD)
    uint8_t* d_codesD = nullptr;
    int codeCapD = 0, codePitchD = 0;
    // threshold-rank cells (host_plan.h): the tile kernel's second form, reading a 16-bit pyramid
    bool useRank = false;
    TileGeom geomR{};
    CascTile* d_tilesR = nullptr;
    int nTilesR = 0;
    TreeNode* d_tileNodesR = nullptr;  // rank-tile offsets, thresholds as rank indices
    TreeNode* d_tailNodesR = nullptr;  // k_cascade_tail_rank: packed (z, c, r) feature positions, thresholds as rank indices
    uint32_t* d_tileNodesSR = nullptr;
    RankChan* d_rankChan = nullptr;
    RankRec* d_rankRec = nullptr;
    int rankMaxRec = 0;
    RankJob* d_rankJobs = nullptr;
    uint16_t* d_pyrR = nullptr;
    int64_t pyrRCells = 0; // cells per frame
    int rankMaxWP = 0;
};

struct acf_hip_ctx
{
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    // side streams: independent launches of one stage (the level groups) run concurrently, forked from and
    // joined back into `stream` with events, so the stage costs its longest launch instead of their sum
    std::vector<hipStream_t> side;
    // Sub-batch contexts (option "streams" = K > 1): the batch is cut into K contiguous chunks, each run by a child
    // context with its own buffers and stream, forked from / joined into `stream` by events.  The path is a chain of
    // kernels several of which are latency-bound inside a frame (column recursions, per-tile dependency chains):
    // chunks at different stages of the chain fill each other's idle issue slots and memory time.
    std::vector<acf_hip_ctx*> kids;
    int nStreams = 1, kidChunk = 0;
    hipEvent_t evFork = nullptr;
    std::vector<hipEvent_t> evJoin;
    mutable std::string err;
    bool hasModel = false, hasPlan = false;
    int taps = 0;
    int arith = 0;            // option "arith": 1 = the reference's rcpps / rsqrtps bits from d_x86 (acf_hip_set_x86_tables)
    bool x86Owned = false;
    uint32_t* d_x86 = nullptr; // [X86_BUF_N]: rcp over [1, 2) by m >> 11 (4096), rsqrt over [1, 4) by parity and m >> 11 (2 x 4096), then gradMag's composed pairs
    int profile = 0;
    int noFusedSmooth = 0; // option "fused_smooth" = 0: separate smoothing / half resample / colour-channel kernels
    // option "fused_grad": 0 = gradMag as its own kernel (k_grad_mag_vec), 1 = inside the gradient plane's smoothing chain
    // (k_smooth_grad) where that pays (big planes, many frames), 2 = wherever k_smooth_grad applies
    int fusedGrad = 1;
    // option "fused_tri": convTri's x pass over M inside that chain as well (k_smooth_grad_tri; the gradient plane is then ONE segment):
    // 0 = never (k_tri_x5v), 1 = where k_smooth_grad runs in batches of >= 64 frames of a context that shares its device, 2 = wherever
    // k_smooth_grad runs
    int fusedTri = 1;
    // option "shared_device": this context runs beside other contexts on the same device (the pools set it).  Kernel forms are then
    // chosen for the least WORK instead of the shortest time alone: a lone context cuts the smoothing chains into segments to fill the
    // machine (20 % more columns, a verify and a repair launch per scale) and keeps the x pass a kernel of its own; beside other
    // contexts a thin chain's latency is covered by their kernels and only its work counts (3 x 96 frames at 1080p: +4.3 % frames/s;
    // alone, 96 frames: -7 %)
    int sharedDevice = 0;
    // option "tile_persist": the pooled tile kernel runs as persistent workgroups that draw tiles from a counter (best alone on the
    // device: -8 % on that kernel) or one short-lived workgroup per tile (best beside other contexts' kernels, which then find free LDS)
    int tilePersist = 1;
    // the device as hipGetDeviceProperties describes it (acf_hip_create): the persistent grids are sized from these.  The
    // counters the persistent workgroups draw tiles from are eight (blockIdx.x & 7: one per XCD of an MI355X, whose
    // dispatcher deals consecutive workgroups to consecutive XCDs); on a part with another XCD count the ranges still
    // cover every tile, only their locality is lost.
    int numCus = 256;
    size_t ldsPerCu = size_t(160) * 1024;
    int noTiles = 0; // option "cascade_tiles" = 0: force the global-memory staged cascade (A/B and parity of both paths)
    int noRank = 0;  // option "rank_cells" = 0: the tile kernel reads the float pyramid (A/B and parity of both forms)
    bool ranksValid = false; // the rank pyramid of the last batch has been written (by the level kernels or by k_rank)
    bool floatPyramid = true; // the float pyramid of the last batch exists (option keep_pyramid = 0: it may not)
    std::vector<hipEvent_t> evPool;
    std::vector<const char*> evName;
    size_t evUsed = 0;
    std::vector<hipStream_t> evStream; // the stream each profile event was recorded on (a kernel ends at the next event of ITS stream)
    std::vector<hipEvent_t> evScale;   // pyramid: "real scale k has been smoothed" (the next real scale starts from it on its own stream)
    bool scaleStreams = true;          // option scale_streams
    // option cascade_turns: the tile kernels of the contexts of one device take turns (each waits for the tile kernel submitted
    // before it on that device, whichever context's): contexts that run in phase otherwise run their cascades — bound by
    // LDS and VALU, not by memory — beside each other instead of beside the other contexts' memory-bound pyramid kernels
    int cascTurns = 0;
    hipEvent_t evTurn[2] = { nullptr, nullptr }; // [0] tile kernel, [1] level kernel (bit 1 of the option: own turns; bit 2: the tile kernels' turns)

    acf_hip_params p{};
    std::vector<uint32_t> fids, child;
    std::vector<float> thrs, hs;
    // LDCF post-stage (acf_hip_params::ldcfK > 0): filters, level table of the LDCF pyramid, per-level resample descriptors
    std::vector<float> ldcfFilters;
    std::vector<acf_hip_level> ldcfLevels;
    int ldcfDescBase = 0;
    int64_t ldcfFloats = 0, ldcfTmpFloats = 0;
    float* d_ldcfFilt = nullptr;
    LdcfJob* d_ldcfJobs = nullptr;
    // k_ldcf_tile (filters + half resample fused): flat list of output tiles over all levels, LDS geometry
    LdcfTileJob* d_ldcfTileJobs = nullptr;
    int ldcfTiles = 0, ldcfTileRows = 0, ldcfTileCols = 0;
    int ldcfMaxCells = 0, ldcfMaxBlocks = 0;
    float* d_ldcfTmp = nullptr;
    float* d_ldcfPyr = nullptr;

    Plan plan;
    // image-specific lambdas (model without lambdas, chnsPyramid.cpp:341-374): per-frame plane sums of two real levels on
    // the device, the lambdas of every frame of the last batch on the host
    bool autoLambdas = false;
    double* d_planeSums = nullptr;     // [maxBatch][2][nChns]
    std::vector<double> h_lambdas;     // [maxBatch][3]
    int maxBatch = 0, maxHits = 0, lastBatch = 0;
    bool pyramidValid = false, detectValid = false;

    std::vector<void*> allocs;
    // constant tables
    float* d_lTable = nullptr;
    float* d_acos = nullptr;
    // planning tables
    std::vector<RealScale> real;
    std::vector<ResampleDesc> h_descs; // [real-image resamples..., approx levels...]
    int nImgDescs = 0, nApproxDescs = 0;
    ResampleDesc* d_descs = nullptr;
    int32_t* d_it = nullptr;
    float* d_ft = nullptr;
    SmoothJob* d_realJobs = nullptr; // one per real scale
    SmoothJob* d_finalJobs = nullptr;
    LevelJob* d_levelJobs = nullptr;
    float* d_dump = nullptr; // 64 floats nobody reads: target of stores from lanes past the end of a plane (keeps kernels branch-free)
    LevelJob* d_levelJobsRaw = nullptr; // same levels, every one read from its raw channels (after a separate resample launch)
    struct LevelGroup
    {
        int R, mode, first, count;
        double cost; // ~ column steps x work per step of the group's longest plane chain and of all its planes (see pack)
        int lane;    // side stream the launch goes to (longest-processing-time-first over the hardware queues)
    };
    std::vector<LevelGroup> levelGroups, levelGroupsRaw; // jobs sorted into runs of equal (R, mode)
    int nAllJobs = 0, nAllJobsRaw = 0; // leading jobs of d_levelJobs(/Raw) that go to the single k_level_all launch
    int levelMode = 1;                  // option "fused_levels": 1 fused resample+smooth, 2 separate resample + wave-per-plane smooth, 0 separate launches
    bool fusedOk = false;      // the fused resample+smooth level kernel covers this plan
    int noFused = 0;           // option "fused_levels" = 0: separate resample and smoothing launches
    PadJob* d_padJobs = nullptr;
    PadJob* d_padJobsR = nullptr; // the same borders in the rank pyramid's layout
    bool levelsEmitRank = false;  // every level goes through k_level_all: the level kernels can write the rank cells themselves
    // k_smooth_vec's speculative column segments (kernels.hip.h): options smooth_segments (0 auto, 1 off, n), smooth_warm, smooth_force_redo
    int smoothSegments = 0;
    // 96 warm-up columns: a value decays fourfold per column, so 12-25 columns settle the last bit of ordinary values — but where
    // the image turns exactly 0 (a black bar) the true chain carries a tail that only reaches 0 by underflow, after ~75 columns,
    // while a warm-up started inside the bar is 0 at once (profiles/r03_repair_rates.json: 3 % of planes repaired at 48, none
    // at 64+ on frames with black and flat bands); at 96 frames per launch 48 / 64 / 96 columns cost the same (0.36-0.38 ms)
    int smoothWarm = 64, smoothForceRedo = 0; // (profiles/r03_repair_rates.json: no repair from 64 columns on)
    float *d_specState = nullptr, *d_trueState = nullptr;
    bool countersZeroed = false; // acf_hip_run has cleared the tiled cascade's counters in front of the pyramid's launches
    int32_t* d_redo = nullptr;
    size_t redoInts = 0, lvRedoInts = 0; // their sizes (clearRepairFlags)
    // the same for the level chains (k_level_all<OUT, 1>), for batches of at most levelSegFrames frames
    float *d_lvSpec = nullptr, *d_lvTrue = nullptr;
    int32_t* d_lvRedo = nullptr;
    int levelSegFrames = 0, levelSegCap = 0, levelHMax = 0;
    // The level chains' segments: option level_segments 0 (default) = AUTO — batches of at most levelSegFrames (8) frames run the
    // speculative segmented k_level_all + k_level_verify + the repair launch, bigger batches one chain per plane; 1 = off; n > 1 =
    // n segments whatever the batch.  A level is at most 480 columns, so segments only pay with warm-ups of ~32 columns (option
    // level_warm; one frame: 372 -> 124 us), and channel planes have exactly-zero regions wherever the image is flat (no
    // gradient): a plane whose hand-over differs costs its whole chain again in the repair launch (tests/test_gpu_segments.py
    // counts the repairs of a frame with flat bands so that this cost stays visible).
    int levelSegments = 0; // option level_segments: 0 = auto (small batches only), 1 = off
    int levelWarm = 32; // option level_warm
    // option count_repairs: planes the repair launches had to recompute (synchronises after every verify: measurements only)
    int countRepairs = 0;
    // option "graph": acf_hip_run replays a HIP graph captured from its own launches (same frames pointer, same batch size):
    // one host call instead of ~45 launches per frame — what a single frame's latency is made of when the kernels take 20 us
    int useGraph = 0;
    // option "graph": captured runs keyed by (input pointer, batch size); a caller that alternates between a few input buffers
    // (double / triple buffering) replays one graph per buffer instead of re-capturing on every call
    struct GraphSlot
    {
        hipGraphExec_t exec = nullptr;
        const float* frames = nullptr;
        int n = 0;
        bool ranksValid = false, floatPyramid = true; // what the captured host code left behind
    };
    static constexpr int GRAPH_SLOTS = 4;
    GraphSlot graphs[GRAPH_SLOTS];
    int graphNext = 0; // slot the next capture replaces (round robin)
    int plainRuns = 0, graphBroken = 0;
    int64_t repairs[4] = { 0, 0, 0, 0 }; // {smoothing planes checked, redone, level planes checked, redone}
    int segCap = 0;               // segments the state buffers hold per plane
    int keepPyramid = 1;          // option "keep_pyramid": 0 = a run()/detect-only caller does not need the float pyramid (levels leave as rank cells only)
    int finalMaxH = 0;
    int approxMaxBlocks = 0;
    int64_t padMaxElems = 0;
    // frame buffers
    float* d_color = nullptr; // colour-converted full-resolution image (if a conversion is needed)
    float* d_chns = nullptr;
    float* d_pyr = nullptr;
    const float* lastFrames = nullptr;
    float* d_stage = nullptr; // H2D staging for run_host; planar f32 target of the 8-bit ingest when no colour conversion follows
    // streaming front end (acf_hip_stream_*)
    struct StreamSlot
    {
        uint8_t* d_in = nullptr;   // device copy of the packed 8-bit batch
        int32_t* d_rec = nullptr;  // device records (acf_hip_export_detections layout)
        int32_t* h_rec = nullptr;  // pinned host records
        hipEvent_t evH2D = nullptr, evDone = nullptr;
        int ticket = -1, nFrames = 0;
    };
    std::vector<StreamSlot> slots;
    hipStream_t copyStream = nullptr;
    int stPix = 0, stStride = 0, stCap = 0, nextTicket = 0, nextCollect = 0;
    // cascade (tables, model arrays, work queues, outputs): one value so that
    // acf_hip_op_acf_detect1 can swap in a temporary set and restore the plan's
    CascState cs;
    BoxLevel* d_boxLevels = nullptr;
    // device bbNms + prune (acf_hip_set_nms): survivors of every frame, what get_detections / export then return
    bool nmsOn = false;
    acf_hip_nms_params nms{};
    int32_t *d_nmsKeep = nullptr, *d_nmsN = nullptr, *d_nmsCounts = nullptr;
    acf_hip_detection* d_nmsDets = nullptr;
    std::vector<int32_t> h_counts;
    bool countsFetched = false;
    // acf_hip_set_input_resize: the apps' resize to a minimum object width in front of the 8-bit entries (k_resize_u8)
    struct InputResize
    {
        bool on = false;
        int rows = 0, cols = 0;
        double scale = 1;
        ResizeTables t;
        int32_t *d_xlin = nullptr, *d_ylin = nullptr, *d_xrun = nullptr, *d_yrun = nullptr, *d_xtap = nullptr, *d_ytap = nullptr;
        uint8_t* d_out = nullptr; // [maxBatch][plan.H][plan.W][4]
    } rz;
};

#define HIPCHK(ctx, call)                                                                                  \
    do                                                                                                     \
    {                                                                                                      \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess)                                                                              \
        {                                                                                                  \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                               \
            return ACF_HIP_E_HIP;                                                                          \
        }                                                                                                  \
    } while (0)

#define LAUNCHCHK(ctx, what)                                                                               \
    do                                                                                                     \
    {                                                                                                      \
        hipError_t e_ = hipGetLastError();                                                                 \
        if (e_ != hipSuccess)                                                                              \
        {                                                                                                  \
            (ctx)->err = std::string("launch ") + what + ": " + hipGetErrorString(e_);                    \
            return ACF_HIP_E_HIP;                                                                          \
        }                                                                                                  \
    } while (0)

namespace
{

// ACF_HIP_FORCE_FALLBACK: the library's ONE form-selecting environment switch (read once per process).  Every stage keeps one
// fallback form for the shapes its default form does not cover; a comma-separated list of the names below makes those fallbacks
// run where the default would, so that tests/test_gpu_variants.py can hold each of them against the oracle on ordinary frames.
// All forms give identical results.  (The other two variables, ACF_HIP_CASC_BOUNDS and ACF_HIP_TILE_TR / _NW, are the tile
// kernels' tuning knobs: stage boundaries and tile geometry.)
enum : uint32_t
{
    FB_TRIY_UNFUSED = 1u << 0,      // convTri's y pass writes S, k_chns forms the cells (instead of k_triy_chns)
    FB_MOU_PLAIN = 1u << 1,         // M, O, U as plain planes (instead of 64 x 16 blocks)
    FB_RESAMPLE_NO_STRIP = 1u << 2, // image resamples without k_resample_strip
    FB_RESAMPLE_NO_PAIR = 1u << 3,  // ... without the two small scales in one pass
    FB_RESAMPLE_NO_UP = 1u << 4,    // up-sampling (nOctUp > 0) through the generic k_resample
    FB_RESAMPLE_GENERIC = 1u << 5,  // every resample through the generic k_resample
    FB_LEVEL_GROUPS = 1u << 6,      // the levels as one k_level launch per (rows per lane, mode) run instead of k_level_all
    FB_LDCF_UNFUSED = 1u << 7,      // LDCF as k_ldcf_conv + k_resample instead of k_ldcf_tile
    FB_NO_DEDUP = 1u << 8,          // stride < shrink: one cascade evaluation per window instead of one per distinct offset
    FB_NO_TAIL_CODES = 1u << 9,     // the tail without leaf codes (k_cascade_tail3 / _tail_rank on every tail window)
    FB_TAIL3 = 1u << 10,            // every tail window through k_cascade_tail3
    FB_TILED_STAGED = 1u << 11,     // depths 1, 3, 4: the staged queue instead of the pooled tile kernel
    FB_TILED_POOLED1 = 1u << 12,    // depth 1 on float cells: the pooled tile kernel instead of the staged queue
};
uint32_t fallbackMask()
{
    static const uint32_t mask = [] {
        static const struct { const char* name; uint32_t bit; } kNames[] = {
            { "triy_unfused", FB_TRIY_UNFUSED }, { "mou_plain", FB_MOU_PLAIN }, { "resample_no_strip", FB_RESAMPLE_NO_STRIP },
            { "resample_no_pair", FB_RESAMPLE_NO_PAIR }, { "resample_no_up", FB_RESAMPLE_NO_UP }, { "resample_generic", FB_RESAMPLE_GENERIC },
            { "level_groups", FB_LEVEL_GROUPS }, { "ldcf_unfused", FB_LDCF_UNFUSED }, { "no_dedup", FB_NO_DEDUP }, { "no_tail_codes", FB_NO_TAIL_CODES },
            { "tail3", FB_TAIL3 }, { "tiled_staged", FB_TILED_STAGED }, { "tiled_pooled1", FB_TILED_POOLED1 } };
        uint32_t m = 0;
        const char* e = getenv("ACF_HIP_FORCE_FALLBACK");
        std::string s = e ? e : "";
        size_t i = 0;
        while (i < s.size())
        {
            size_t j = s.find(',', i);
            j = j == std::string::npos ? s.size() : j;
            const std::string tok = s.substr(i, j - i);
            bool known = tok.empty();
            for (const auto& n : kNames)
            {
                if (tok == n.name)
                {
                    m |= n.bit;
                    known = true;
                }
            }
            if (!known)
            {
                fprintf(stderr, "acf_hip: ACF_HIP_FORCE_FALLBACK: unknown name '%s'\n", tok.c_str());
                abort(); // (a test that names a form that does not exist must not pass on the default form)
            }
            i = j + 1;
        }
        return m;
    }();
    return mask;
}
inline bool fallbackForced(uint32_t which)
{
    return (fallbackMask() & which) != 0;
}

int fail(const acf_hip_ctx* c, int code, const std::string& msg)
{
    c->err = msg;
    return code;
}

// Profile marker: an event recorded on the stream before the launch named
// `name`; a kernel's time is the span to the next marker.
// the CPU tables the three approximate sites evaluate (option "arith" = 1), or null: exact arithmetic
inline const uint32_t* x86T(const acf_hip_ctx* c)
{
    return c->arith ? c->d_x86 : nullptr;
}

void prof(acf_hip_ctx* c, const char* name)
{
    if (!c->profile)
    {
        return;
    }
    if (c->evUsed == c->evPool.size())
    {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess)
        {
            return;
        }
        c->evPool.push_back(e);
        c->evName.push_back(name);
        c->evStream.push_back(c->stream);
    }
    c->evName[c->evUsed] = name;
    c->evStream[c->evUsed] = c->stream;
    (void)hipEventRecord(c->evPool[c->evUsed], c->stream);
    c->evUsed++;
}

template <class T>
int devAlloc(acf_hip_ctx* c, T** out, size_t count)
{
    void* p = nullptr;
    HIPCHK(c, hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    c->allocs.push_back(p);
    *out = reinterpret_cast<T*>(p);
    return ACF_HIP_OK;
}

template <class T>
int devUpload(acf_hip_ctx* c, T** out, const std::vector<T>& v)
{
    int rc = devAlloc(c, out, v.size());
    if (rc)
    {
        return rc;
    }
    if (!v.empty())
    {
        HIPCHK(c, hipMemcpy(*out, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    }
    return ACF_HIP_OK;
}

// one buffer of devAlloc / devUpload given back before the context's teardown (the caller has made sure nothing in flight reads it)
template <class T>
void devRelease(acf_hip_ctx* c, T*& p)
{
    if (!p)
    {
        return;
    }
    auto it = std::find(c->allocs.begin(), c->allocs.end(), static_cast<void*>(p));
    if (it != c->allocs.end())
    {
        c->allocs.erase(it);
        (void)hipFree(p);
    }
    p = nullptr;
}

void freeAll(acf_hip_ctx* c)
{
    for (void* p : c->allocs)
    {
        (void)hipFree(p);
    }
    c->allocs.clear();
    c->hasPlan = false;
    c->d_lTable = c->d_acos = nullptr;
    c->d_dump = nullptr;
    c->d_specState = c->d_trueState = nullptr;
    c->d_redo = nullptr;
    c->segCap = 0;
    c->d_lvSpec = c->d_lvTrue = nullptr;
    c->d_lvRedo = nullptr;
    c->levelSegFrames = 0;
    c->d_color = c->d_stage = c->d_chns = c->d_pyr = nullptr; // a re-plan must not see the previous plan's buffers
    c->rz = acf_hip_ctx::InputResize();
    c->d_ldcfFilt = c->d_ldcfTmp = c->d_ldcfPyr = nullptr;
    c->d_ldcfJobs = nullptr;
    c->d_ldcfTileJobs = nullptr;
    c->d_nmsKeep = c->d_nmsN = c->d_nmsCounts = nullptr;
    c->d_nmsDets = nullptr;
    c->ldcfTiles = 0;
    c->lastFrames = nullptr;
    c->pyramidValid = c->detectValid = false;
}

// rgb2luv_setup's table (toolbox/rgbConvertMex.cpp:39-58)
std::vector<float> makeLTable()
{
    std::vector<float> t(1064);
    const float y0 = (float)((6.0 / 29) * (6.0 / 29) * (6.0 / 29));
    const float a = (float)((29.0 / 3) * (29.0 / 3) * (29.0 / 3));
    const float maxi = (float)1.0 / 270;
    for (int i = 0; i < 1025; i++)
    {
        float y = (float)(i / 1024.0);
        float l = y > y0 ? 116 * (float)pow((double)y, 1.0 / 3.0) - 16 : y * a;
        t[i] = l * maxi;
    }
    for (int i = 1025; i < 1064; i++)
    {
        t[i] = t[i - 1];
    }
    return t;
}

LuvConsts makeLuvConsts()
{
    LuvConsts k;
    const float z = 1.0f;
    k.un = (float)0.197833;
    k.vn = (float)0.468331;
    k.mr[0] = (float)0.430574 * z;
    k.mr[1] = (float)0.222015 * z;
    k.mr[2] = (float)0.020183 * z;
    k.mg[0] = (float)0.341550 * z;
    k.mg[1] = (float)0.706655 * z;
    k.mg[2] = (float)0.129553 * z;
    k.mb[0] = (float)0.178325 * z;
    k.mb[1] = (float)0.071330 * z;
    k.mb[2] = (float)0.939180 * z;
    const float maxi = (float)1.0 / 270;
    k.minu = -88 * maxi;
    k.minv = -134 * maxi;
    k.cun = 13 * k.un;
    k.cvn = 13 * k.vn;
    return k;
}

// ACosTable (toolbox/gradientMex.cpp:103-165): 2*(n+b) entries, centre at n+b.
std::vector<float> makeAcosTable()
{
    const int n = 10000, b = 10;
    const float PI = 3.14159265f;
    std::vector<float> a(2 * n + 2 * b);
    float* a1 = a.data() + n + b;
    for (int i = -n - b; i < -n; i++)
    {
        a1[i] = PI;
    }
    for (int i = -n; i < n; i++)
    {
        a1[i] = float(std::acos(i / float(n)));
    }
    for (int i = n; i < n + b; i++)
    {
        a1[i] = 0;
    }
    for (int i = -n - b; i < n / 10; i++)
    {
        if (a1[i] > PI - 1e-6f)
        {
            a1[i] = PI - 1e-6f;
        }
    }
    return a;
}

int ensureConstTables(acf_hip_ctx* c)
{
    if (c->d_lTable)
    {
        return ACF_HIP_OK;
    }
    int rc = devUpload(c, &c->d_lTable, makeLTable());
    if (rc)
    {
        return rc;
    }
    if ((rc = devUpload(c, &c->d_acos, makeAcosTable())))
    {
        return rc;
    }
    return devAlloc(c, &c->d_dump, 1024);
}

// rank pyramid: cells between the columns of a level (every column starts on 16 bytes)
inline int rankPitch(int hP)
{
    return (hP + 7) / 8 * 8;
}

inline int cdiv(int64_t a, int64_t b)
{
    return int((a + b - 1) / b);
}

// ---- kernel launch helpers (shared by the pyramid and the op_* entry points) ----

int launchSmooth(acf_hip_ctx* c, const float* in, float* out, const SmoothJob* d_jobs, int nJobs, int maxPlanes, int maxH,
    int64_t in_fs, int64_t out_fs, int nFrames, float p, bool aliased)
{
    // threads along y, R interleaved rows per thread: the (nt, R) with the fewest idle row slots (every slot
    // is computed: the kernel clamps instead of branching), fewer rows per thread on a tie
    int nt = 64, R = 0;
    int64_t best = -1;
    for (int r = 1; r <= 4; r++)
    {
        const int t = ((maxH + 64 * r - 1) / (64 * r)) * 64;
        if (t <= 1024 && (best < 0 || int64_t(t) * r < best))
        {
            best = int64_t(t) * r;
            nt = t;
            R = r;
        }
    }
    if (!R)
    {
        return fail(c, ACF_HIP_E_UNSUPPORTED, "smooth: plane taller than 4096 rows");
    }
    const int ldsStride = maxH + 1;
    prof(c, nJobs > 1 ? "k_smooth_tri1(levels)" : "k_smooth_tri1(image)");
    const size_t lds = 2 * size_t(ldsStride) * sizeof(float);
    dim3 grid(maxPlanes, nJobs, nFrames), block(nt);
#define SM_LAUNCH(RR)                                                                                                    \
    if (aliased)                                                                                                         \
        hipLaunchKernelGGL((k_smooth_tri1<RR, true>), grid, block, lds, c->stream, in, out, d_jobs, in_fs, out_fs, p, ldsStride, c->d_dump); \
    else                                                                                                                 \
        hipLaunchKernelGGL((k_smooth_tri1<RR, false>), grid, block, lds, c->stream, in, out, d_jobs, in_fs, out_fs, p, ldsStride, c->d_dump);
    switch (R)
    {
        case 1:
            SM_LAUNCH(1);
            break;
        case 2:
            SM_LAUNCH(2);
            break;
        case 3:
            SM_LAUNCH(3);
            break;
        case 4:
            SM_LAUNCH(4);
            break;
        default:
            return fail(c, ACF_HIP_E_UNSUPPORTED, "smooth: plane taller than 4096 rows");
    }
#undef SM_LAUNCH
    LAUNCHCHK(c, "k_smooth_tri1");
    return ACF_HIP_OK;
}

// fuse: the ChnsArgs of the level when convTriY may be followed at once by the channel cells (k_triy_chns); *fused reports it
// floats per frame of U in the blocked layout: [ceil(w / 64)][ceil((h + 8) / 16)] blocks of 64 columns x 16 rows
static int64_t uBlockedFloats(int h, int w)
{
    return int64_t(cdiv(w, 64)) * ((h + 8 + 15) / 16) * 1024;
}

// ... and of M / O: [ceil(w / 64)][ceil(h / 16)] blocks
static int64_t moBlockedFloats(int h, int w)
{
    return int64_t(cdiv(w, 64)) * ((h + 15) / 16) * 1024;
}

struct TriPlan
{
    bool vecX, doFuse, blocked;
};

// Which forms convTri(M) + the channel cells of a level take: the vector x pass, the fused y pass + cells, and the blocked
// layout of M, O and U (kernels.hip.h, k_tri_x5v) — possible when gradMag, the x pass and the fused y pass are all the vector
// forms and the buffers have room for the blocks' padding (A/B: ACF_HIP_MOU_PLAIN).  gradVec: k_grad_mag_vec writes M and O.
static TriPlan triPlan(const float* in, const float* U, int h, int w, int rad, int64_t fs, const ChnsArgs* fuse, int64_t uCapacity, int64_t moCapacity,
    bool gradVec)
{
    const bool noFuse = fallbackForced(FB_TRIY_UNFUSED), plain = fallbackForced(FB_MOU_PLAIN);
    TriPlan t;
    t.vecX = rad == 5 && h % 4 == 0 && w >= 48 && fs % 4 == 0 && ((uintptr_t(in) | uintptr_t(U)) & 15) == 0;
    t.doFuse = fuse && !noFuse && rad == 5 && h % 4 == 0 && h >= 48 && w % 4 == 0 && fs % 4 == 0 && (uintptr_t(U) & 15) == 0 && fuse->doNorm &&
        !fuse->Mn && (fuse->colorDone || !fuse->colorEnabled) && (fuse->magEnabled || fuse->histEnabled) && fuse->nOrients <= 12 &&
        ((uintptr_t(fuse->M) | uintptr_t(fuse->O)) & 15) == 0;
    t.blocked = t.vecX && t.doFuse && gradVec && !plain && uCapacity >= uBlockedFloats(h, w) && moCapacity >= moBlockedFloats(h, w);
    return t;
}

// uCapacity / moCapacity: floats per frame the U and the M, O buffers hold (>= fs); blocked: M and O ARE in the blocked layout
// (the caller ran k_grad_mag_vec<true> after asking triPlan)
// xDone: U is there already (k_smooth_grad_tri wrote it, blocked)
int launchTri(acf_hip_ctx* c, const float* in, float* U, float* S, int h, int w, int rad, int64_t fs, int nFrames, ChnsArgs* fuse = nullptr,
    bool* fused = nullptr, int64_t uCapacity = 0, int64_t moCapacity = 0, bool blocked = false, bool xDone = false)
{
    if (rad > 15)
    {
        return fail(c, ACF_HIP_E_UNSUPPORTED, "convTri: radius > 15");
    }
    const TriPlan tp = triPlan(in, U, h, w, rad, fs, fuse, uCapacity, moCapacity, blocked);
    if (blocked && !tp.blocked)
    {
        return fail(c, ACF_HIP_E_INVALID, "convTri: blocked M/O without the kernels that read them");
    }
    const bool vecX = tp.vecX, doFuse = tp.doFuse, ut = blocked;
    const int nyb = (h + 8 + 15) / 16, nybM = (h + 15) / 16;
    const int64_t ufs = uBlockedFloats(h, w), mfs = moBlockedFloats(h, w);
    if (xDone)
    {
        if (!ut || !doFuse)
        {
            return fail(c, ACF_HIP_E_INVALID, "convTri: the x pass rode on the smoothing chain, but the y pass does not read its blocks");
        }
    }
    else
    {
        prof(c, "k_tri_x");
        if (vecX)
        {
            if (ut)
            {
                hipLaunchKernelGGL((k_tri_x5v<true>), dim3(cdiv(h / 4, 64), 1, nFrames), dim3(64), 0, c->stream, in, U, h, w, mfs, ufs, nyb, nybM);
            }
            else
            {
                hipLaunchKernelGGL((k_tri_x5v<false>), dim3(cdiv(h / 4, 64), 1, nFrames), dim3(64), 0, c->stream, in, U, h, w, fs, fs, 0, 0);
            }
        }
        else
        {
            hipLaunchKernelGGL(k_tri_x, dim3(cdiv(h, 256), 1, nFrames), dim3(256), 0, c->stream, in, U, h, w, rad, fs);
        }
        LAUNCHCHK(c, "k_tri_x");
    }
    if (fused)
    {
        *fused = false;
    }
    if (doFuse)
    {
        prof(c, "k_triy_chns");
        if (ut)
        {
            fuse->m_fs = mfs;
            fuse->nybM = nybM;
        }
        const dim3 grid(cdiv(w, 256), 1, nFrames), block(256);
#define TRIY_LAUNCH(MAXO_, UT_)                                                                                                         \
    if (fuse->x86)                                                                                                                      \
    {                                                                                                                                   \
        hipLaunchKernelGGL((k_triy_chns<MAXO_, UT_, true>), grid, block, 0, c->stream, (const float*)U, *fuse, UT_ ? ufs : fs, UT_ ? nyb : 0); \
    }                                                                                                                                   \
    else                                                                                                                                \
    {                                                                                                                                   \
        hipLaunchKernelGGL((k_triy_chns<MAXO_, UT_, false>), grid, block, 0, c->stream, (const float*)U, *fuse, UT_ ? ufs : fs, UT_ ? nyb : 0); \
    }
        if (fuse->nOrients <= 6)
        {
            if (ut)
            {
                TRIY_LAUNCH(6, true)
            }
            else
            {
                TRIY_LAUNCH(6, false)
            }
        }
        else if (ut)
        {
            TRIY_LAUNCH(12, true)
        }
        else
        {
            TRIY_LAUNCH(12, false)
        }
#undef TRIY_LAUNCH
        LAUNCHCHK(c, "k_triy_chns");
        *fused = true;
        return ACF_HIP_OK;
    }
    prof(c, "k_tri_y");
    if (rad == 5 && h % 4 == 0 && h >= 48 && fs % 4 == 0 && ((uintptr_t(U) | uintptr_t(S)) & 15) == 0)
    {
        hipLaunchKernelGGL(k_tri_y5s, dim3(cdiv(w, 256), 1, nFrames), dim3(256), 0, c->stream, (const float*)U, S, h, w, fs);
    }
    else
    {
        hipLaunchKernelGGL(k_tri_y, dim3(cdiv(w, 64), 1, nFrames), dim3(64), 0, c->stream, (const float*)U, S, h, w, rad, fs);
    }
    LAUNCHCHK(c, "k_tri_y");
    return ACF_HIP_OK;
}

float shrinkGainY(int S)
{
    // imResampleMex.cpp:145-157 with r = 1 and wa == S*wb, then :286/:316 r/S
    float r = 1.0f;
    r /= (float)S;
    r /= float(1 + 1e-6);
    return r / (float)S;
}

int launchChns(acf_hip_ctx* c, const ChnsArgs& a, int shrink, int nFrames)
{
    const int hc = a.h / shrink, wc = a.w / shrink;
    dim3 grid(cdiv(int64_t(hc) * wc, 256), 1, nFrames), block(256);
    prof(c, "k_chns");
    if (shrink == 4)
    {
        if (a.nOrients <= 6)
        {
            hipLaunchKernelGGL((k_chns<4, 6>), grid, block, 0, c->stream, a);
        }
        else
        {
            hipLaunchKernelGGL((k_chns<4, 12>), grid, block, 0, c->stream, a);
        }
    }
    else
    {
        if (a.nOrients <= 6)
        {
            hipLaunchKernelGGL((k_chns<2, 6>), grid, block, 0, c->stream, a);
        }
        else
        {
            hipLaunchKernelGGL((k_chns<2, 12>), grid, block, 0, c->stream, a);
        }
    }
    LAUNCHCHK(c, "k_chns");
    return ACF_HIP_OK;
}

} // namespace

namespace
{
struct Scratch
{
    std::vector<void*> ptrs;
    ~Scratch()
    {
        for (void* p : ptrs)
        {
            (void)hipFree(p);
        }
    }
    template <class T>
    T* alloc(size_t n)
    {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess)
        {
            return nullptr;
        }
        ptrs.push_back(p);
        return reinterpret_cast<T*>(p);
    }
    template <class T>
    T* upload(const T* h, size_t n)
    {
        T* d = alloc<T>(n);
        if (d && n && hipMemcpy(d, h, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess)
        {
            return nullptr;
        }
        return d;
    }
};
} // namespace

// gradMagNorm as a stand-alone kernel (only the op_gradient_mag entry point
// needs it; the pyramid fuses it into k_chns).  toolbox/gradientMex.cpp:254-275.
__global__ void __launch_bounds__(256) k_norm(float* __restrict__ M, const float* __restrict__ S, int n, float norm, const uint32_t* __restrict__ x86)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
    {
        return;
    }
    const int n4 = (n / 4) * 4;
    if (i < n4)
    {
        M[i] = x86 ? M[i] * x86_rcp(S[i] + norm, x86) : M[i] * (1.0f / (S[i] + norm));
    }
    else
    {
        M[i] = M[i] / (S[i] + norm);
    }
}

// Strided plane copy for smooth == 0 (convTri.cpp:206-210: J = I).
__global__ void __launch_bounds__(256) k_copy_planes(const float* __restrict__ in, float* __restrict__ out, const SmoothJob* __restrict__ jobs,
    int64_t in_fs, int64_t out_fs)
{
    const SmoothJob j = jobs[blockIdx.y];
    const int64_t per = int64_t(j.h) * j.w;
    const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= per * j.nplanes)
    {
        return;
    }
    const int z = int(e / per);
    const int rem = int(e - int64_t(z) * per);
    const int x = rem / j.h, y = rem - x * j.h;
    out[int64_t(blockIdx.z) * out_fs + j.out_off + int64_t(z) * j.out_ps + int64_t(x) * j.out_cs + y] =
        in[int64_t(blockIdx.z) * in_fs + j.in_off + int64_t(z) * j.in_ps + rem];
}


namespace
{
int kidsFork(acf_hip_ctx* c)
{
    HIPCHK(c, hipEventRecord(c->evFork, c->stream));
    for (acf_hip_ctx* k : c->kids)
    {
        HIPCHK(c, hipStreamWaitEvent(k->stream, c->evFork, 0));
    }
    return ACF_HIP_OK;
}
int kidsJoin(acf_hip_ctx* c)
{
    for (size_t i = 0; i < c->kids.size(); i++)
    {
        HIPCHK(c, hipEventRecord(c->evJoin[i], c->kids[i]->stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->evJoin[i], 0));
    }
    return ACF_HIP_OK;
}
int kidFail(acf_hip_ctx* c, acf_hip_ctx* k, int rc)
{
    c->err = k->err;
    return rc;
}
// frames of the last batch owned by child i: [i * chunk, min(n, (i+1) * chunk))
int kidCount(const acf_hip_ctx* c, size_t i, int n)
{
    return std::max(0, std::min(n, int(i + 1) * c->kidChunk) - int(i) * c->kidChunk);
}
} // namespace

// option cascade_turns: the last submission of every device per kernel class (an event owned by the submitting context)
namespace
{
std::mutex g_turnMu;
hipEvent_t g_turnLast[2][64] = {};
// which: 0 = the tile kernel, 1 = the level kernel; chain: whose turns it takes (0 / 1)
int turnBegin(acf_hip_ctx* c, int which, int chain)
{
    if (c->device < 0 || c->device >= 64)
    {
        return ACF_HIP_OK;
    }
    if (!c->evTurn[which])
    {
        HIPCHK(c, hipEventCreateWithFlags(&c->evTurn[which], hipEventDisableTiming));
    }
    std::lock_guard<std::mutex> lk(g_turnMu);
    hipEvent_t last = g_turnLast[chain][c->device];
    if (last && last != c->evTurn[0] && last != c->evTurn[1])
    {
        HIPCHK(c, hipStreamWaitEvent(c->stream, last, 0));
    }
    return ACF_HIP_OK;
}
int turnEnd(acf_hip_ctx* c, int which, int chain)
{
    if (!c->evTurn[which] || c->device < 0 || c->device >= 64)
    {
        return ACF_HIP_OK;
    }
    std::lock_guard<std::mutex> lk(g_turnMu);
    HIPCHK(c, hipEventRecord(c->evTurn[which], c->stream));
    g_turnLast[chain][c->device] = c->evTurn[which];
    return ACF_HIP_OK;
}
void cascTurnForget(acf_hip_ctx* c)
{
    std::lock_guard<std::mutex> lk(g_turnMu);
    for (int w = 0; w < 2; w++)
    {
        if (c->evTurn[w])
        {
            for (int ch = 0; ch < 2; ch++)
            {
                if (c->device >= 0 && c->device < 64 && g_turnLast[ch][c->device] == c->evTurn[w])
                {
                    g_turnLast[ch][c->device] = nullptr;
                }
            }
            (void)hipEventDestroy(c->evTurn[w]);
            c->evTurn[w] = nullptr;
        }
    }
}
} // namespace

extern "C" {


int acf_hip_abi_version(void)
{
    return ACF_HIP_ABI_VERSION;
}

// The side streams of a context (real scales beside each other, level groups): created when something first forks.
// The HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES (default 4) hardware queues, and kernels of
// two streams that share a queue never overlap: with seven streams per context the main streams of three contexts
// shared queues (rocprofv3 timeline: never more than two kernels at once; profiles/timeline.py).
constexpr int ACF_SIDE_STREAMS = 6;
constexpr int LEVEL_MAX_R_REAL = 9; // k_level<R, LM_REAL> is instantiated up to nine rows per lane (a 4K frame's 540-cell level), resampling forms up to eight
static void ensureSide(acf_hip_ctx* c)
{
    while (int(c->side.size()) < std::min<int>(ACF_SIDE_STREAMS, int(c->evJoin.size())))
    {
        hipStream_t st;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess)
        {
            (void)hipGetLastError();
            break;
        }
        c->side.push_back(st);
    }
}

int acf_hip_create(int device, void* stream, acf_hip_ctx** out)
{
    if (!out)
    {
        return ACF_HIP_E_INVALID;
    }
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n)
    {
        return ACF_HIP_E_NODEVICE;
    }
    if (hipSetDevice(device) != hipSuccess)
    {
        return ACF_HIP_E_NODEVICE;
    }
    acf_hip_ctx* c = new (std::nothrow) acf_hip_ctx();
    if (!c)
    {
        return ACF_HIP_E_HIP;
    }
    c->device = device;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess)
        {
            c->numCus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : c->numCus;
            c->ldsPerCu = prop.maxSharedMemoryPerMultiProcessor > 0 ? size_t(prop.maxSharedMemoryPerMultiProcessor) : c->ldsPerCu;
        }
    }
    if (stream)
    {
        c->stream = reinterpret_cast<hipStream_t>(stream);
    }
    else
    {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess)
        {
            delete c;
            return ACF_HIP_E_HIP;
        }
        c->ownStream = true;
    }
    // (the side streams themselves are created at their first use, ensureSide: every HIP stream takes a share of the
    // runtime's few hardware queues, and contexts that never fork must not push each other's main streams onto one queue)
    for (int i = 0; i < ACF_SIDE_STREAMS; i++)
    {
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess)
        {
            c->evJoin.push_back(ev);
        }
    }
    (void)hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming);
    *out = c;
    return ACF_HIP_OK;
}

int acf_hip_device_count(int* count)
{
    if (!count)
    {
        return ACF_HIP_E_INVALID;
    }
    *count = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    {
        (void)hipGetLastError();
        return ACF_HIP_E_NODEVICE;
    }
    for (int i = 0; i < n; i++)
    {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, i) == hipSuccess && std::string(pr.gcnArchName).rfind("gfx950", 0) == 0)
        {
            (*count)++;
        }
    }
    return *count > 0 ? ACF_HIP_OK : ACF_HIP_E_NODEVICE;
}

static void dropGraph(acf_hip_ctx* c)
{
    if (!c)
    {
        return;
    }
    bool synced = false;
    for (auto& g : c->graphs)
    {
        if (g.exec)
        {
            if (!synced)
            {
                // (the last replay may still be running: an executable graph is not released under it)
                (void)hipSetDevice(c->device);
                (void)hipStreamSynchronize(c->stream);
                synced = true;
            }
            (void)hipGraphExecDestroy(g.exec);
        }
        g = acf_hip_ctx::GraphSlot();
    }
    c->graphNext = 0;
    c->plainRuns = 0; // the next run is a plain one again (it may allocate: NMS buffers, side streams)
}

int acf_hip_destroy(acf_hip_ctx* c)
{
    dropGraph(c);
    if (c)
    {
        for (acf_hip_ctx* k : c->kids)
        {
            (void)acf_hip_destroy(k);
        }
        c->kids.clear();
    }
    if (!c)
    {
        return ACF_HIP_E_INVALID;
    }
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)acf_hip_stream_close(c);
    cascTurnForget(c);
    freeAll(c);
    if (c->d_x86 && c->x86Owned)
    {
        (void)hipFree(c->d_x86);
    }
    for (hipEvent_t e : c->evPool)
    {
        (void)hipEventDestroy(e);
    }
    for (hipStream_t st : c->side)
    {
        (void)hipStreamDestroy(st);
    }
    for (hipEvent_t e : c->evJoin)
    {
        (void)hipEventDestroy(e);
    }
    if (c->evFork)
    {
        (void)hipEventDestroy(c->evFork);
    }
    for (hipEvent_t e : c->evScale)
    {
        (void)hipEventDestroy(e);
    }
    if (c->ownStream)
    {
        (void)hipStreamDestroy(c->stream);
    }
    delete c;
    return ACF_HIP_OK;
}

const char* acf_hip_last_error(const acf_hip_ctx* c)
{
    return c ? c->err.c_str() : "null context";
}

int acf_hip_set_option(acf_hip_ctx* c, const char* key, int value)
{
    if (!c || !key)
    {
        return ACF_HIP_E_INVALID;
    }
    dropGraph(c); // (whatever changes, a captured graph no longer describes the next run)
    if (!strcmp(key, "graph"))
    {
        c->useGraph = value;
        c->graphBroken = 0;
        return ACF_HIP_OK;
    }
    if (!strcmp(key, "streams"))
    {
        if (value < 1 || value > 6)
        {
            return fail(c, ACF_HIP_E_INVALID, "streams: 1..6");
        }
        c->nStreams = value; // takes effect at the next acf_hip_plan
        return ACF_HIP_OK;
    }
    for (acf_hip_ctx* k : c->kids)
    {
        (void)acf_hip_set_option(k, key, value);
    }
    if (!strcmp(key, "taps"))
    {
        c->taps = value != 0;
        return ACF_HIP_OK;
    }
    if (!strcmp(key, "arith"))
    {
        if (value != 0 && !c->d_x86)
        {
            return fail(c, ACF_HIP_E_INVALID, "arith: install a CPU's tables first (acf_hip_set_x86_tables)");
        }
        c->arith = value != 0;
        return ACF_HIP_OK;
    }
    if (!strcmp(key, "profile"))
    {
        c->profile = value != 0;
        c->evUsed = 0;
        return ACF_HIP_OK;
    }
    if (!strcmp(key, "fused_levels"))
    {
        c->noFused = value == 0;
        c->levelMode = value;
        return ACF_HIP_OK;
    }
    if (!strcmp(key, "fused_smooth"))
    {
        c->noFusedSmooth = value == 0;
        return ACF_HIP_OK;
    }
    if (!strcmp(key, "fused_grad"))
    {
        c->fusedGrad = value;
        return ACF_HIP_OK;
    }
    if (!strcmp(key, "fused_tri"))
    {
        c->fusedTri = value;
        return ACF_HIP_OK;
    }
    if (!strcmp(key, "shared_device"))
    {
        c->sharedDevice = value != 0;
        return ACF_HIP_OK;
    }
    if (!strcmp(key, "cascade_tiles"))
    {
        c->noTiles = value == 0;
        return ACF_HIP_OK;
    }
    if (!strcmp(key, "count_repairs") || !strcmp(key, "level_warm") || !strcmp(key, "level_segments"))
    {
        if ((!strcmp(key, "level_warm") && (value < 4 || value % 4 != 0)) || value < 0)
        {
            return fail(c, ACF_HIP_E_INVALID, "option: level_warm a positive multiple of 4, level_segments >= 0");
        }
        (!strcmp(key, "count_repairs") ? c->countRepairs : (!strcmp(key, "level_warm") ? c->levelWarm : c->levelSegments)) = value;
        for (acf_hip_ctx* k : c->kids)
        {
            (void)acf_hip_set_option(k, key, value);
        }
        return ACF_HIP_OK;
    }
    if (!strcmp(key, "smooth_segments") || !strcmp(key, "smooth_warm") || !strcmp(key, "smooth_force_redo"))
    {
        // the image smoothing's recursion along image-x cut into column segments that start `smooth_warm` columns early and are
        // checked against each other (kernels.hip.h, "speculative segments"): 0 = as many as fill the GPU for the batch, 1 = one
        // chain per plane, n = n segments; smooth_force_redo = 1 marks every plane for the repair launch (tests)
        int& dst = !strcmp(key, "smooth_segments") ? c->smoothSegments : (!strcmp(key, "smooth_warm") ? c->smoothWarm : c->smoothForceRedo);
        if (value < 0 || (!strcmp(key, "smooth_warm") && value % 16 != 0))
        {
            return fail(c, ACF_HIP_E_INVALID, "option: smooth_segments >= 0, smooth_warm a multiple of 16");
        }
        dst = value;
        for (acf_hip_ctx* k : c->kids)
        {
            (void)acf_hip_set_option(k, key, value);
        }
        return ACF_HIP_OK;
    }
    if (!strcmp(key, "keep_pyramid"))
    {
        // 0: the caller only wants detections (acf_hip_run / acf_hip_detect): when the levels leave as rank cells the float
        // pyramid is not written at all (acf_hip_read_level and the Pyramid-returning entries then fail with
        // ACF_HIP_E_INVALID); 1 (default): both
        c->keepPyramid = value != 0;
        for (acf_hip_ctx* k : c->kids)
        {
            k->keepPyramid = c->keepPyramid;
        }
        return ACF_HIP_OK;
    }
    if (!strcmp(key, "rank_cells"))
    {
        // 1 (default): the tile kernel of depth-2 models reads 16-bit threshold-rank cells (host_plan.h) — identical decisions,
        // half the tile; 0: it reads the float pyramid
        c->noRank = value == 0;
        c->detectValid = false;
        for (acf_hip_ctx* k : c->kids)
        {
            k->noRank = c->noRank;
            k->detectValid = false;
        }
        return ACF_HIP_OK;
    }
    if (!strcmp(key, "tile_persist"))
    {
        c->tilePersist = value;
        for (acf_hip_ctx* k : c->kids)
        {
            k->tilePersist = value;
        }
        return ACF_HIP_OK;
    }
    if (!strcmp(key, "cascade_turns"))
    {
        c->cascTurns = value;
        for (acf_hip_ctx* k : c->kids)
        {
            k->cascTurns = value;
        }
        return ACF_HIP_OK;
    }
    if (!strcmp(key, "scale_streams"))
    {
        // 1 (default): the real scales of a batch run on their own streams (best for ONE context: +8 % frames/s, batch-1 latency
        // 2.1 -> 1.5 ms); 0: on the context's stream in order — what an application that runs several contexts side by side
        // wants (three contexts: 12.1k instead of 11.6k frames/s; they already fill each other's gaps, more streams only
        // multiplex the hardware queues)
        c->scaleStreams = value != 0;
        return ACF_HIP_OK;
    }
    return fail(c, ACF_HIP_E_INVALID, std::string("unknown option ") + key);
}

int acf_hip_get_scales(int nPerOct, int nOctUp, int minDs_h, int minDs_w, int shrink, int h, int w,
    double* scales, double* shw_h, double* shw_w, int cap, int* n)
{
    if (!n || nPerOct <= 0 || minDs_h <= 0 || minDs_w <= 0 || shrink <= 0)
    {
        return ACF_HIP_E_INVALID;
    }
    ScaleList sl = getScales(nPerOct, nOctUp, minDs_h, minDs_w, shrink, h, w);
    *n = int(sl.scales.size());
    for (int i = 0; i < *n && i < cap; i++)
    {
        if (scales)
        {
            scales[i] = sl.scales[i];
        }
        if (shw_h)
        {
            shw_h[i] = sl.shw_h[i];
        }
        if (shw_w)
        {
            shw_w[i] = sl.shw_w[i];
        }
    }
    return ACF_HIP_OK;
}

int acf_hip_plan_levels(const acf_hip_params* p, int h, int w, int d, acf_hip_level* out, int cap, int* nScales, int* nChns)
{
    if (!p || !nScales)
    {
        return ACF_HIP_E_INVALID;
    }
    Plan plan;
    std::string err;
    int rc = buildPlan(*p, h, w, d, plan, err);
    if (rc)
    {
        return rc;
    }
    *nScales = int(plan.levels.size());
    if (nChns)
    {
        *nChns = plan.nChns;
    }
    for (int i = 0; i < *nScales && i < cap && out; i++)
    {
        out[i] = plan.levels[i];
    }
    return ACF_HIP_OK;
}

int acf_hip_set_model(acf_hip_ctx* c, const acf_hip_params* p)
{
    dropGraph(c);
    if (c)
    {
        for (acf_hip_ctx* k : c->kids)
        {
            (void)acf_hip_destroy(k);
        }
        c->kids.clear();
    }
    if (!c || !p)
    {
        return ACF_HIP_E_INVALID;
    }
    if (p->nTrees <= 0 || p->nTreeNodes <= 0 || !p->fids || !p->thrs || !p->hs)
    {
        return fail(c, ACF_HIP_E_INVALID, "set_model: empty classifier");
    }
    if (p->treeDepth < 0 || p->treeDepth > 8)
    {
        return fail(c, ACF_HIP_E_INVALID, "set_model: treeDepth must be 0..8 (acfDetect1.cpp:204-226)");
    }
    if (p->treeDepth == 0 && !p->child)
    {
        return fail(c, ACF_HIP_E_INVALID, "set_model: treeDepth 0 needs child[]");
    }
    if (p->treeDepth > 0 && p->nTreeNodes < (1 << (p->treeDepth + 1)) - 1)
    {
        return fail(c, ACF_HIP_E_INVALID, "set_model: nTreeNodes too small for treeDepth");
    }
    if (p->stride <= 0 || p->shrink <= 0 || p->modelDsPad_h % p->shrink || p->modelDsPad_w % p->shrink)
    {
        return fail(c, ACF_HIP_E_INVALID, "set_model: stride/shrink/modelDsPad");
    }
    if (p->nOrients < 1 || p->nOrients > 12)
    {
        return fail(c, ACF_HIP_E_UNSUPPORTED, "set_model: nOrients must be 1..12");
    }
    const bool ldcf = p->ldcfK > 0 && p->ldcfFilters;
    if (ldcf && (p->ldcfK > 16 || p->modelDsPad_h % (2 * p->shrink) || p->modelDsPad_w % (2 * p->shrink)))
    {
        return fail(c, ACF_HIP_E_INVALID, "set_model: LDCF needs 1..16 filters per channel and modelDsPad divisible by 2*shrink");
    }
    const size_t n = size_t(p->nTrees) * p->nTreeNodes;
    c->p = *p;
    c->ldcfFilters.clear();
    c->p.ldcfK = ldcf ? p->ldcfK : 0;
    c->p.ldcfFilters = nullptr;
    if (ldcf)
    {
        const int dcol = p->colorSpace == ACF_HIP_CS_GRAY ? 1 : 3;
        const int nC = (p->colorEnabled ? dcol : 0) + (p->gradMagEnabled ? 1 : 0) + (p->gradHistEnabled ? p->nOrients : 0);
        c->ldcfFilters.assign(p->ldcfFilters, p->ldcfFilters + size_t(p->ldcfK) * nC * 25);
        c->p.ldcfFilters = c->ldcfFilters.data();
    }
    c->fids.assign(p->fids, p->fids + n);
    c->thrs.assign(p->thrs, p->thrs + n);
    c->hs.assign(p->hs, p->hs + n);
    if (p->child)
    {
        c->child.assign(p->child, p->child + n);
    }
    else
    {
        c->child.assign(n, 0u);
    }
    c->p.fids = c->fids.data();
    c->p.thrs = c->thrs.data();
    c->p.hs = c->hs.data();
    c->p.child = c->child.data();
    c->hasModel = true;
    if (c->hasPlan)
    {
        (void)hipStreamSynchronize(c->stream);
        freeAll(c);
    }
    return ACF_HIP_OK;
}

#include "cascade_plan.hip.h"  // ShrinkScope, TileSet, buildTileSet, buildCascadeTables

#include "resample_plan.hip.h" // stripPlan, launchStrip, resampleTilePlan

#include "plan_build.hip.h" // acf_hip_plan's stages (PlanBuild), after the helpers they call

int acf_hip_plan(acf_hip_ctx* c, int H, int W, int d_in, int max_batch, int max_hits)
{
    if (!c)
    {
        return ACF_HIP_E_INVALID;
    }
    dropGraph(c);
    if (!c->hasModel)
    {
        return fail(c, ACF_HIP_E_NOMODEL, "plan: set_model first");
    }
    if (max_batch <= 0 || max_hits <= 0)
    {
        return fail(c, ACF_HIP_E_INVALID, "plan: max_batch/max_hits");
    }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)acf_hip_stream_close(c); // its slots are sized by the plan
    freeAll(c);
    const acf_hip_params& p = c->p;
    int rc = buildPlan(p, H, W, d_in, c->plan, c->err);
    if (rc)
    {
        return rc;
    }
    for (acf_hip_ctx* k : c->kids)
    {
        (void)acf_hip_destroy(k);
    }
    c->kids.clear();
    if (c->nStreams > 1 && max_batch >= 2 * c->nStreams && c->evFork && int(c->evJoin.size()) >= c->nStreams)
    {
        // sub-batch contexts: this context keeps only the plan's geometry; every device buffer lives in a child
        c->kidChunk = (max_batch + c->nStreams - 1) / c->nStreams;
        for (int i = 0; i < c->nStreams; i++)
        {
            acf_hip_ctx* k = nullptr;
            if ((rc = acf_hip_create(c->device, nullptr, &k)))
            {
                return fail(c, rc, "plan: cannot create a sub-batch context");
            }
            c->kids.push_back(k);
            k->taps = c->taps;
            k->d_x86 = c->d_x86; // (the parent's tables; the parent frees them)
            k->arith = c->arith;
            k->profile = c->profile;
            k->noTiles = c->noTiles;
            k->noRank = c->noRank;
            k->keepPyramid = c->keepPyramid;
            k->smoothSegments = c->smoothSegments;
            k->smoothWarm = c->smoothWarm;
            k->smoothForceRedo = c->smoothForceRedo;
            k->levelWarm = c->levelWarm;
            k->levelSegments = c->levelSegments;
            k->countRepairs = c->countRepairs;
            k->noFusedSmooth = c->noFusedSmooth;
            k->fusedGrad = c->fusedGrad;
            k->fusedTri = c->fusedTri;
            k->sharedDevice = c->sharedDevice;
            k->noFused = c->noFused;
            k->levelMode = c->levelMode;
            k->scaleStreams = c->scaleStreams;
            k->cascTurns = c->cascTurns;
            k->tilePersist = c->tilePersist;
            k->nmsOn = c->nmsOn; // (acf_hip_set_nms before the plan, or a re-plan: the new children run what the parent reports)
            k->nms = c->nms;
            if ((rc = acf_hip_set_model(k, &c->p)) || (rc = acf_hip_plan(k, H, W, d_in, c->kidChunk, max_hits)))
            {
                return kidFail(c, k, rc);
            }
        }
        c->maxBatch = max_batch;
        c->maxHits = max_hits;
        c->hasPlan = true;
        c->pyramidValid = c->detectValid = false;
        c->lastBatch = 0;
        return ACF_HIP_OK;
    }
    // smoothing radii the recursion kernel implements (convTri.cpp:215-218)
    for (double r : { p.colorSmooth, p.smooth })
    {
        if (!(r == 0.0 || (r > 0 && r <= 1.0)))
        {
            return fail(c, ACF_HIP_E_UNSUPPORTED, "plan: smoothing radius must be 0 or in (0,1]");
        }
    }
    if (p.normRad < 0 || p.normRad == 1 || p.normRad > 15)
    {
        return fail(c, ACF_HIP_E_UNSUPPORTED, "plan: normRad must be 0 or 2..15");
    }
    if ((rc = ensureConstTables(c)))
    {
        return rc;
    }
    c->maxBatch = max_batch;
    c->maxHits = max_hits;
    PlanBuild b(c, H, W, d_in, max_batch, max_hits);
    if ((rc = b.colourBuffer()) || (rc = b.realScales()) || (rc = b.imageStrips()) || (rc = b.approxLevels()) || (rc = b.finalJobs()) || (rc = b.levelJobs()) ||
        (rc = b.ldcf()) || (rc = b.uploadAndScratch()) || (rc = b.cascade()))
    {
        return rc;
    }
    c->h_counts.assign(size_t(max_batch), 0);
    c->hasPlan = true;
    c->pyramidValid = c->detectValid = false;
    c->lastBatch = 0;
    return ACF_HIP_OK;
}

int acf_hip_num_levels(const acf_hip_ctx* c, int* nScales, int* nChns)
{
    if (!c || !c->hasPlan)
    {
        return ACF_HIP_E_NOPLAN;
    }
    if (nScales)
    {
        *nScales = int(c->plan.levels.size());
    }
    if (nChns)
    {
        *nChns = c->plan.nChns;
    }
    return ACF_HIP_OK;
}

int acf_hip_get_levels(const acf_hip_ctx* c, acf_hip_level* out, int cap)
{
    if (!c || !c->hasPlan)
    {
        return ACF_HIP_E_NOPLAN;
    }
    if (!out || cap < int(c->plan.levels.size()))
    {
        return fail(c, ACF_HIP_E_INVALID, "get_levels: capacity");
    }
    std::copy(c->plan.levels.begin(), c->plan.levels.end(), out);
    return ACF_HIP_OK;
}

int acf_hip_get_ldcf_levels(const acf_hip_ctx* c, acf_hip_level* out, int cap)
{
    if (!c || !c->hasPlan || !out)
    {
        return ACF_HIP_E_INVALID;
    }
    const acf_hip_ctx* src = c->kids.empty() ? c : c->kids[0];
    if (src->p.ldcfK <= 0 || src->ldcfLevels.empty())
    {
        return ACF_HIP_E_INVALID;
    }
    const int n = std::min(cap, int(src->ldcfLevels.size()));
    std::copy(src->ldcfLevels.begin(), src->ldcfLevels.begin() + n, out);
    return ACF_HIP_OK;
}

int acf_hip_pyramid_floats(const acf_hip_ctx* c, int64_t* n)
{
    if (!c || !c->hasPlan)
    {
        return ACF_HIP_E_NOPLAN;
    }
    *n = c->plan.pyr_floats;
    return ACF_HIP_OK;
}

int acf_hip_get_repairs(acf_hip_ctx* c, int64_t out[4])
{
    if (!c || !out)
    {
        return ACF_HIP_E_INVALID;
    }
    for (int i = 0; i < 4; i++)
    {
        out[i] = c->repairs[i];
        for (const acf_hip_ctx* k : c->kids)
        {
            out[i] += k->repairs[i];
        }
    }
    return ACF_HIP_OK;
}

int acf_hip_get_lambdas(acf_hip_ctx* c, int frame, double out[3])
{
    if (!c || !out)
    {
        return ACF_HIP_E_INVALID;
    }
    if (!c->kids.empty())
    {
        if (frame < 0 || frame >= c->lastBatch)
        {
            return fail(c, ACF_HIP_E_INVALID, "get_lambdas: frame out of range");
        }
        const int k = frame / std::max(c->kidChunk, 1);
        return acf_hip_get_lambdas(c->kids[size_t(k)], frame - k * c->kidChunk, out);
    }
    if (!c->hasPlan || !c->pyramidValid)
    {
        return fail(c, ACF_HIP_E_NOPLAN, "get_lambdas: no pyramid");
    }
    if (frame < 0 || frame >= c->lastBatch)
    {
        return fail(c, ACF_HIP_E_INVALID, "get_lambdas: frame out of range");
    }
    for (int j = 0; j < 3; j++)
    {
        out[j] = c->h_lambdas[size_t(frame) * 3 + j];
    }
    return ACF_HIP_OK;
}

static int allowLds(acf_hip_ctx* c, const void* kernel, size_t bytes);

// A/B knob: ACF_HIP_RESAMPLE_GENERIC forces the gather kernel for the image resamples (read once)
static bool resampleGenericOnly()
{
    const bool v = fallbackForced(FB_RESAMPLE_GENERIC);
    return v;
}

// k_resample_up applies: both axes up-sampled, whole quads of output rows, 16-byte aligned planes
static bool resampleUpOk(const ResampleDesc& d)
{
    const bool off = fallbackForced(FB_RESAMPLE_NO_UP);
    return !off && d.xmode == RS_UP && d.ymode == RS_UP && d.hb % 4 == 0 && d.ha >= 4 && d.dst_off % 4 == 0 && d.dst_frame_stride % 4 == 0;
}

namespace
{
// packed 8-bit source of a batch (acf_hip_pyramid_u8)
struct PackedSrc
{
    const uint8_t* frames;
    int pix, rowStride;
};
int pyramidImpl(acf_hip_ctx* c, const float* frames, const PackedSrc* u8, int nF);
} // namespace

int acf_hip_pyramid(acf_hip_ctx* c, const float* frames, int nF)
{
    return pyramidImpl(c, frames, nullptr, nF);
}

namespace
{
int launchIngest(acf_hip_ctx* c, const PackedSrc& u, int nF, float* dst, int64_t out_fs, bool convert)
{
    const acf_hip_params& p = c->p;
    const Plan& pl = c->plan;
    IngestArgs a{};
    a.in = u.frames;
    a.out = dst;
    a.lTable = c->d_lTable;
    a.k = makeLuvConsts();
    a.x86 = x86T(c);
    a.mr = (float).2989360213 * 1.0f;
    a.mg = (float).5870430745 * 1.0f;
    a.mb = (float).1140209043 * 1.0f;
    a.H = pl.H;
    a.W = pl.W;
    switch (u.pix)
    {
        case ACF_HIP_PIX_RGB: a.cpp = 3, a.ro = 0, a.go = 1, a.bo = 2; break;
        case ACF_HIP_PIX_BGR: a.cpp = 3, a.ro = 2, a.go = 1, a.bo = 0; break;
        case ACF_HIP_PIX_RGBA: a.cpp = 4, a.ro = 0, a.go = 1, a.bo = 2; break;
        case ACF_HIP_PIX_BGRA: a.cpp = 4, a.ro = 2, a.go = 1, a.bo = 0; break;
        case ACF_HIP_PIX_GRAY: a.cpp = 1, a.ro = 0, a.go = 0, a.bo = 0; break;
        default: return fail(c, ACF_HIP_E_INVALID, "pyramid_u8: pixel layout");
    }
    if ((a.cpp == 1) != (pl.d_in == 1))
    {
        return fail(c, ACF_HIP_E_INVALID, "pyramid_u8: the plan's input planes (d) must be 3 for colour layouts and 1 for GRAY");
    }
    a.rowStride = u.rowStride > 0 ? u.rowStride : pl.W * a.cpp;
    if (a.rowStride < pl.W * a.cpp)
    {
        return fail(c, ACF_HIP_E_INVALID, "pyramid_u8: row stride smaller than a row");
    }
    a.in_fs = int64_t(a.rowStride) * pl.H;
    a.out_fs = out_fs;
    a.vecStore = (pl.H % 4 == 0) && (out_fs % 4 == 0) && (reinterpret_cast<uintptr_t>(dst) % 16 == 0);
    const bool aligned = (reinterpret_cast<uintptr_t>(u.frames) % 4 == 0) && (a.rowStride % 4 == 0);
    const int64_t np0 = int64_t(pl.H) * pl.W;
    int mode = IG_PLANAR;
    a.nOut = pl.d_in;
    if (convert)
    {
        if (p.colorSpace == ACF_HIP_CS_LUV)
        {
            mode = (np0 % 4 == 0) ? IG_LUV_VEC : IG_LUV; // rgbConvertMex.cpp:92,343
        }
        else if (p.colorSpace == ACF_HIP_CS_GRAY)
        {
            mode = IG_GRAY; // a 1-plane input is replicated first (chnsPyramid.cpp:234-244): r == g == b
        }
        else
        {
            a.nOut = 3; // ORIG / RGB with a 1-plane input: replicate (chnsPyramid.cpp:242-243)
        }
    }
    dim3 grid(cdiv(pl.W, IG_T), cdiv(pl.H, IG_T), nF), block(256);
#define IG_LAUNCH(M)                                                                  \
    if (aligned)                                                                      \
    {                                                                                 \
        hipLaunchKernelGGL((k_ingest_u8<M, true>), grid, block, 0, c->stream, a);     \
    }                                                                                 \
    else                                                                              \
    {                                                                                 \
        hipLaunchKernelGGL((k_ingest_u8<M, false>), grid, block, 0, c->stream, a);    \
    }
    switch (mode)
    {
        case IG_LUV_VEC: IG_LAUNCH(IG_LUV_VEC) break;
        case IG_LUV: IG_LAUNCH(IG_LUV) break;
        case IG_GRAY: IG_LAUNCH(IG_GRAY) break;
        default: IG_LAUNCH(IG_PLANAR) break;
    }
#undef IG_LAUNCH
    LAUNCHCHK(c, "k_ingest_u8");
    return ACF_HIP_OK;
}

static int pixCpp(int pix)
{
    return pix == ACF_HIP_PIX_GRAY ? 1 : ((pix == ACF_HIP_PIX_RGBA || pix == ACF_HIP_PIX_BGRA) ? 4 : 3);
}

// k_resize_u8 over a batch: `src` frames of t.rows x t.cols (row stride `stride` bytes) -> tight t.drows x t.dcols frames at dst
static int launchResizeU8(acf_hip_ctx* c, const acf_hip_ctx::InputResize& rz, const uint8_t* src, int cpp, int stride, int nF, uint8_t* dst)
{
    const ResizeTables& t = rz.t;
    ResizeArgs a{};
    a.src = src;
    a.dst = dst;
    a.rows = t.rows;
    a.cols = t.cols;
    a.cn = cpp;
    a.stride = stride;
    a.drows = t.drows;
    a.dcols = t.dcols;
    a.src_fs = int64_t(stride) * t.rows;
    a.dst_fs = int64_t(t.drows) * t.dcols * cpp;
    a.mode = t.mode;
    a.isx = t.isx;
    a.isy = t.isy;
    a.xlin = reinterpret_cast<const int4*>(rz.d_xlin);
    a.ylin = reinterpret_cast<const int4*>(rz.d_ylin);
    a.xrun = reinterpret_cast<const int2*>(rz.d_xrun);
    a.yrun = reinterpret_cast<const int2*>(rz.d_yrun);
    a.xtap = reinterpret_cast<const int2*>(rz.d_xtap);
    a.ytap = reinterpret_cast<const int2*>(rz.d_ytap);
    prof(c, "k_resize_u8");
    hipLaunchKernelGGL(k_resize_u8, dim3(cdiv(t.dcols, 64), cdiv(t.drows, 4), nF), dim3(256), 0, c->stream, a);
    LAUNCHCHK(c, "k_resize_u8");
    return ACF_HIP_OK;
}

static int uploadResizeTables(acf_hip_ctx* c, acf_hip_ctx::InputResize& rz)
{
    int rc;
    if ((rc = devUpload(c, &rz.d_xlin, rz.t.xlin)) || (rc = devUpload(c, &rz.d_ylin, rz.t.ylin)) || (rc = devUpload(c, &rz.d_xrun, rz.t.xrun)) ||
        (rc = devUpload(c, &rz.d_yrun, rz.t.yrun)) || (rc = devUpload(c, &rz.d_xtap, rz.t.xtap)) || (rc = devUpload(c, &rz.d_ytap, rz.t.ytap)))
    {
        return rc;
    }
    return ACF_HIP_OK;
}

// A call that fails between a verify launch and its repair launch (a launch error, the count_repairs read-back) would leave
// repair flags set for the next call — harmless for results (a flagged plane is recomputed as one chain), but counted by
// count_repairs and paid for.  Every failing pyramid call therefore takes the flags down again.
static void clearRepairFlags(acf_hip_ctx* c)
{
    if (c->d_redo && c->redoInts)
    {
        (void)hipMemsetAsync(c->d_redo, 0, sizeof(int32_t) * c->redoInts, c->stream);
    }
    if (c->d_lvRedo && c->lvRedoInts)
    {
        (void)hipMemsetAsync(c->d_lvRedo, 0, sizeof(int32_t) * c->lvRedoInts, c->stream);
    }
}

int pyramidBody(acf_hip_ctx* c, const float* frames, const PackedSrc* u8, int nF);

int pyramidImpl(acf_hip_ctx* c, const float* frames, const PackedSrc* u8, int nF)
{
    const int rc = pyramidBody(c, frames, u8, nF);
    if (rc && c && c->hasPlan && c->kids.empty())
    {
        clearRepairFlags(c);
    }
    return rc;
}

} // namespace

#include "pyramid_run.hip.h"

int acf_hip_pyramid_u8(acf_hip_ctx* c, const uint8_t* frames, int nF, int pix, int rowStride)
{
    if (!c)
    {
        return ACF_HIP_E_INVALID;
    }
    PackedSrc u{ frames, pix, rowStride };
    return pyramidImpl(c, nullptr, &u, nF);
}

int acf_hip_resize_dims(int rows, int cols, double scale, int* out_rows, int* out_cols)
{
    if (!out_rows || !out_cols || rows < 1 || cols < 1 || !(scale > 0))
    {
        return ACF_HIP_E_INVALID;
    }
    resizeDims(rows, cols, scale, *out_rows, *out_cols);
    return (*out_rows >= 1 && *out_cols >= 1) ? ACF_HIP_OK : ACF_HIP_E_INVALID;
}

int acf_hip_set_input_resize(acf_hip_ctx* c, int srcRows, int srcCols, double scale)
{
    if (!c || !c->hasPlan)
    {
        return c ? fail(c, ACF_HIP_E_NOPLAN, "set_input_resize: plan first (for the reduced size, acf_hip_resize_dims)") : ACF_HIP_E_INVALID;
    }
    if (!c->kids.empty() || !c->slots.empty())
    {
        return fail(c, ACF_HIP_E_UNSUPPORTED, "set_input_resize: not with option \"streams\" > 1; before acf_hip_stream_open");
    }
    dropGraph(c);
    if (srcRows <= 0 || srcCols <= 0)
    {
        c->rz.on = false;
        return ACF_HIP_OK;
    }
    if (c->rz.d_out && c->rz.rows == srcRows && c->rz.cols == srcCols && c->rz.scale == scale)
    {
        c->rz.on = true; // same geometry as the tables already on the device (a re-opened stream): nothing to build
        return ACF_HIP_OK;
    }
    acf_hip_ctx::InputResize rz;
    if (buildResizeTables(srcRows, srcCols, scale, rz.t) || rz.t.drows != c->plan.H || rz.t.dcols != c->plan.W)
    {
        return fail(c, ACF_HIP_E_INVALID, "set_input_resize: the plan must be for the reduced size (acf_hip_resize_dims of the frame size and scale)");
    }
    rz.rows = srcRows;
    rz.cols = srcCols;
    rz.scale = scale;
    HIPCHK(c, hipSetDevice(c->device));
    auto release = [&](acf_hip_ctx::InputResize& r) {
        devRelease(c, r.d_xlin);
        devRelease(c, r.d_ylin);
        devRelease(c, r.d_xrun);
        devRelease(c, r.d_yrun);
        devRelease(c, r.d_xtap);
        devRelease(c, r.d_ytap);
        devRelease(c, r.d_out);
    };
    int rc;
    if ((rc = uploadResizeTables(c, rz)) || (rc = devAlloc(c, &rz.d_out, size_t(c->maxBatch) * c->plan.H * c->plan.W * 4)))
    {
        release(rz); // (what was uploaded before the failure)
        return rc;
    }
    // the previous geometry's tables: a batch in flight may still read them
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->rz.on = false;
    release(c->rz);
    rz.on = true;
    c->rz = rz;
    return ACF_HIP_OK;
}

int acf_hip_op_resize_u8(acf_hip_ctx* c, const uint8_t* src, int rows, int cols, int cpp, int rowStride, double scale, uint8_t* dst, int dstRows, int dstCols)
{
    if (!c || !src || !dst || cpp < 1 || cpp > 4)
    {
        return c ? fail(c, ACF_HIP_E_INVALID, "op_resize_u8: arguments") : ACF_HIP_E_INVALID;
    }
    if (cpp == 2)
    {
        // OpenCV's 2 x 2 area path rounds two-channel pixels with saturate_cast(sum * 0.25f) (half to even), not (sum + 2) >> 2 as for
        // 1, 3 and 4 channels; no pixel format of the path has two channels (ACF_HIP_PIX_*), so the form is not restated
        return fail(c, ACF_HIP_E_UNSUPPORTED, "op_resize_u8: two-channel images are not supported (1, 3 or 4 bytes per pixel)");
    }
    acf_hip_ctx::InputResize rz;
    if (buildResizeTables(rows, cols, scale, rz.t) || rz.t.drows != dstRows || rz.t.dcols != dstCols)
    {
        return fail(c, ACF_HIP_E_INVALID, "op_resize_u8: dst size must be acf_hip_resize_dims(rows, cols, scale)");
    }
    const int stride = rowStride > 0 ? rowStride : cols * cpp;
    if (stride < cols * cpp)
    {
        return fail(c, ACF_HIP_E_INVALID, "op_resize_u8: row stride smaller than a row");
    }
    HIPCHK(c, hipSetDevice(c->device));
    // one-off buffers (an operator entry, not the hot path)
    uint8_t *dS = nullptr, *dD = nullptr;
    const size_t nS = size_t(stride) * rows, nD = size_t(dstRows) * dstCols * cpp;
    std::vector<void*> tmp;
    auto up = [&](int32_t** d, const std::vector<int32_t>& v) {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(v.size(), 1) * 4) != hipSuccess)
        {
            return false;
        }
        tmp.push_back(p);
        *d = static_cast<int32_t*>(p);
        return v.empty() || hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    };
    bool ok = hipMalloc(reinterpret_cast<void**>(&dS), nS) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&dD), nD) == hipSuccess;
    ok = ok && up(&rz.d_xlin, rz.t.xlin) && up(&rz.d_ylin, rz.t.ylin) && up(&rz.d_xrun, rz.t.xrun) && up(&rz.d_yrun, rz.t.yrun) && up(&rz.d_xtap, rz.t.xtap) &&
        up(&rz.d_ytap, rz.t.ytap);
    int rc = ACF_HIP_OK;
    if (ok)
    {
        ok = hipMemcpy(dS, src, nS, hipMemcpyHostToDevice) == hipSuccess;
        if (ok)
        {
            rc = launchResizeU8(c, rz, dS, cpp, stride, 1, dD);
        }
        ok = ok && !rc && hipStreamSynchronize(c->stream) == hipSuccess && hipMemcpy(dst, dD, nD, hipMemcpyDeviceToHost) == hipSuccess;
    }
    for (void* p : tmp)
    {
        (void)hipFree(p);
    }
    (void)hipFree(dS);
    (void)hipFree(dD);
    if (!ok && !rc)
    {
        (void)hipGetLastError();
        return fail(c, ACF_HIP_E_HIP, "op_resize_u8: device allocation or copy failed");
    }
    return rc;
}

#include "cascade_run.hip.h"    // allowLds, runCascadeTiled, launchNms, runCascade

int acf_hip_detect(acf_hip_ctx* c)
{
    if (c && !c->kids.empty())
    {
        if (!c->pyramidValid)
        {
            return fail(c, ACF_HIP_E_INVALID, "detect: no pyramid (call acf_hip_pyramid)");
        }
        int rc = kidsFork(c);
        for (size_t i = 0; i < c->kids.size() && !rc; i++)
        {
            if (kidCount(c, i, c->lastBatch) > 0 && (rc = acf_hip_detect(c->kids[i])))
            {
                return kidFail(c, c->kids[i], rc);
            }
        }
        c->detectValid = true;
        return rc ? rc : kidsJoin(c);
    }
    if (!c || !c->hasPlan)
    {
        return c ? fail(c, ACF_HIP_E_NOPLAN, "detect: plan first") : ACF_HIP_E_INVALID;
    }
    if (!c->pyramidValid)
    {
        return fail(c, ACF_HIP_E_INVALID, "detect: no pyramid (call acf_hip_pyramid)");
    }
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if (c->p.ldcfK > 0)
    {
        // LDCF: every level is filtered (k 5x5 filters per channel) into a scratch buffer, halved into the LDCF pyramid, and the
        // cascade runs there with cells of 2*shrink pixels (include/acf_hip.h, acf_hip_params::ldcfK)
        const Plan& pl = c->plan;
        const int nF = c->lastBatch, nCk = pl.nChns * c->p.ldcfK;
        const int nL = int(pl.levels.size());
        if (c->ldcfTiles > 0)
        {
            // filters + half resample in one kernel: the filtered full-resolution planes stay in LDS
            prof(c, "k_ldcf_tile");
            const int xoT = 16;
            const size_t ldsB = ldcfTileLdsFloats(c->ldcfTileRows, c->ldcfTileCols, xoT) * sizeof(float);
            auto kern = &k_ldcf_tile<8, 4>;
            if ((rc = allowLds(c, reinterpret_cast<const void*>(kern), ldsB)))
            {
                return rc;
            }
            hipLaunchKernelGGL(kern, dim3(c->ldcfTiles, pl.nChns, nF), dim3(256), ldsB, c->stream, (const float*)c->d_pyr, c->d_ldcfPyr,
                (const float*)c->d_ldcfFilt, (const LdcfTileJob*)c->d_ldcfTileJobs, (const LdcfJob*)c->d_ldcfJobs,
                (const ResampleDesc*)(c->d_descs + c->ldcfDescBase), (const int32_t*)c->d_it, (const float*)c->d_ft, c->ldcfTileRows, c->ldcfTileCols, xoT,
                c->p.ldcfK, pl.nChns, pl.pyr_floats);
            LAUNCHCHK(c, "k_ldcf_tile");
        }
        else
        {
            prof(c, "k_ldcf_conv");
            hipLaunchKernelGGL(k_ldcf_conv, dim3(cdiv(c->ldcfMaxCells, 256), nCk, nF * nL), dim3(256), 0, c->stream, (const float*)c->d_pyr, c->d_ldcfTmp,
                (const float*)c->d_ldcfFilt, (const LdcfJob*)c->d_ldcfJobs, nL, pl.nChns, pl.pyr_floats, c->ldcfTmpFloats);
            LAUNCHCHK(c, "k_ldcf_conv");
            prof(c, "k_resample(ldcf)");
            hipLaunchKernelGGL(k_resample, dim3(c->ldcfMaxBlocks, nL, nF), dim3(64, 4), 0, c->stream, (const float*)c->d_ldcfTmp, c->d_ldcfPyr,
                (const ResampleDesc*)(c->d_descs + c->ldcfDescBase), (const int32_t*)c->d_it, (const float*)c->d_ft, RS_XT);
            LAUNCHCHK(c, "k_resample(ldcf)");
        }
        ShrinkScope ss(c, 2);
        rc = runCascade(c, c->d_ldcfPyr, c->ldcfFloats, c->d_boxLevels, nF, nCk);
    }
    else
    {
        rc = runCascade(c, c->d_pyr, c->plan.pyr_floats, c->d_boxLevels, c->lastBatch, c->plan.nChns);
    }
    if (rc)
    {
        return rc;
    }
    c->detectValid = true;
    return ACF_HIP_OK;
}

// Host-side state after a replayed (or just captured) run: what runPyramid / runCascade's host code would have left.
static void graphRunState(acf_hip_ctx* c, const acf_hip_ctx::GraphSlot& gs)
{
    c->lastBatch = gs.n;
    c->pyramidValid = c->detectValid = true;
    c->ranksValid = gs.ranksValid;
    c->floatPyramid = gs.floatPyramid;
    c->countsFetched = false; // the counts on the host are the previous run's
}

// acf_hip_run knows that the cascade follows the pyramid: the tiled cascade's counters (hit counts; its queue's count and head and the
// tile counters) are cleared in front of the pyramid's launches instead of between the level kernel and the cascade, where a single
// frame waits for every node of the chain (runCascade / runCascadeTiled then skip their own memsets)
static int preZeroCounters(acf_hip_ctx* c, int nF)
{
    if (!c || !c->hasPlan || !c->kids.empty() || !tiledCascadeSelected(c) || c->p.ldcfK > 0 || !c->cs.d_counts || !c->cs.d_qcounts || nF <= 0 ||
        nF > c->maxBatch)
    {
        return ACF_HIP_OK;
    }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(c->cs.d_counts, 0, sizeof(int32_t) * nF, c->stream));
    HIPCHK(c, hipMemsetAsync(c->cs.d_qcounts, 0, sizeof(int32_t) * (2 * size_t(c->maxBatch) + 8), c->stream));
    c->countersZeroed = true;
    return ACF_HIP_OK;
}

int acf_hip_run(acf_hip_ctx* c, const float* frames, int nF)
{
    if (c && !c->kids.empty())
    {
        if (!frames || nF <= 0 || nF > c->maxBatch)
        {
            return fail(c, ACF_HIP_E_INVALID, "run: n_frames out of range");
        }
        const size_t per = size_t(c->plan.d_in) * c->plan.H * c->plan.W;
        int rc = kidsFork(c);
        for (size_t i = 0; i < c->kids.size() && !rc; i++)
        {
            const int n = kidCount(c, i, nF);
            if (n > 0 && (rc = acf_hip_run(c->kids[i], frames + i * size_t(c->kidChunk) * per, n)))
            {
                return kidFail(c, c->kids[i], rc);
            }
        }
        c->lastBatch = nF;
        c->pyramidValid = c->detectValid = true;
        return rc ? rc : kidsJoin(c);
    }
    // option "graph": the launches of this call — same frames pointer, same batch size, nothing in the path that reads back or
    // records events of its own — as one captured graph.  The first calls run plainly (they may allocate); a capture that the
    // runtime refuses switches the option off for this context and the call runs plainly.
    const bool graphable = c && c->useGraph && !c->graphBroken && c->hasPlan && !c->profile && !c->taps && !c->countRepairs && !c->autoLambdas &&
        c->cascTurns == 0 && frames && nF > 0 && nF <= c->maxBatch;
    if (graphable)
    {
        for (auto& gs : c->graphs)
        {
            if (gs.exec && gs.frames == frames && gs.n == nF)
            {
                HIPCHK(c, hipSetDevice(c->device));
                HIPCHK(c, hipGraphLaunch(gs.exec, c->stream));
                graphRunState(c, gs);
                return ACF_HIP_OK;
            }
        }
    }
    if (graphable && c->plainRuns >= 1)
    {
        HIPCHK(c, hipSetDevice(c->device));
        acf_hip_ctx::GraphSlot& gs = c->graphs[c->graphNext];
        if (gs.exec)
        {
            (void)hipStreamSynchronize(c->stream); // (a replay of the slot being replaced may still be running)
            (void)hipGraphExecDestroy(gs.exec);
            gs = acf_hip_ctx::GraphSlot();
        }
        hipGraph_t g = nullptr;
        bool ok = hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed) == hipSuccess;
        int rcc = ACF_HIP_OK;
        if (ok)
        {
            rcc = preZeroCounters(c, nF);
            if (!rcc)
            {
                rcc = acf_hip_pyramid(c, frames, nF);
            }
            if (!rcc)
            {
                rcc = acf_hip_detect(c);
            }
            c->countersZeroed = false;
            ok = hipStreamEndCapture(c->stream, &g) == hipSuccess && g != nullptr && !rcc;
        }
        if (ok)
        {
            ok = hipGraphInstantiate(&gs.exec, g, nullptr, nullptr, 0) == hipSuccess && gs.exec != nullptr;
        }
        if (g)
        {
            (void)hipGraphDestroy(g);
        }
        if (ok)
        {
            gs.frames = frames;
            gs.n = nF;
            gs.ranksValid = c->ranksValid;
            gs.floatPyramid = c->floatPyramid;
            c->graphNext = (c->graphNext + 1) % acf_hip_ctx::GRAPH_SLOTS;
            HIPCHK(c, hipGraphLaunch(gs.exec, c->stream));
            graphRunState(c, gs);
            return ACF_HIP_OK;
        }
        (void)hipGetLastError();
        gs = acf_hip_ctx::GraphSlot();
        c->graphBroken = 1; // (fall through: the plain path below does the work)
    }
    int rc = preZeroCounters(c, nF);
    if (!rc)
    {
        rc = acf_hip_pyramid(c, frames, nF);
    }
    if (rc)
    {
        if (c)
        {
            c->countersZeroed = false;
        }
        return rc;
    }
    rc = acf_hip_detect(c);
    c->countersZeroed = false;
    if (!rc && c)
    {
        c->plainRuns++;
    }
    return rc;
}

int acf_hip_run_host(acf_hip_ctx* c, const float* frames_host, int nF)
{
    if (c && !c->kids.empty())
    {
        if (!frames_host || nF <= 0 || nF > c->maxBatch)
        {
            return fail(c, ACF_HIP_E_INVALID, "run_host: n_frames out of range");
        }
        const size_t per = size_t(c->plan.d_in) * c->plan.H * c->plan.W;
        int rc = kidsFork(c);
        for (size_t i = 0; i < c->kids.size() && !rc; i++)
        {
            const int n = kidCount(c, i, nF);
            if (n > 0 && (rc = acf_hip_run_host(c->kids[i], frames_host + i * size_t(c->kidChunk) * per, n)))
            {
                return kidFail(c, c->kids[i], rc);
            }
        }
        c->lastBatch = nF;
        c->pyramidValid = c->detectValid = true;
        return rc ? rc : kidsJoin(c);
    }
    if (!c || !c->hasPlan)
    {
        return c ? fail(c, ACF_HIP_E_NOPLAN, "run_host: plan first") : ACF_HIP_E_INVALID;
    }
    if (!frames_host || nF <= 0 || nF > c->maxBatch)
    {
        return fail(c, ACF_HIP_E_INVALID, "run_host: n_frames out of range");
    }
    HIPCHK(c, hipSetDevice(c->device));
    const size_t per = size_t(c->plan.d_in) * c->plan.H * c->plan.W;
    if (!c->d_stage)
    {
        int rc = devAlloc(c, &c->d_stage, size_t(c->maxBatch) * per);
        if (rc)
        {
            return rc;
        }
    }
    HIPCHK(c, hipMemcpyAsync(c->d_stage, frames_host, per * nF * sizeof(float), hipMemcpyHostToDevice, c->stream));
    return acf_hip_run(c, c->d_stage, nF);
}

int acf_hip_run_u8(acf_hip_ctx* c, const uint8_t* frames, int nF, int pix, int rowStride)
{
    int rc = acf_hip_pyramid_u8(c, frames, nF, pix, rowStride);
    if (rc)
    {
        return rc;
    }
    return acf_hip_detect(c);
}

// ---------------------------------------------------------------------------
// streaming front end: copy stream + per-slot events; see include/acf_hip.h
// ---------------------------------------------------------------------------
int acf_hip_stream_close(acf_hip_ctx* c)
{
    if (!c)
    {
        return ACF_HIP_E_INVALID;
    }
    (void)hipSetDevice(c->device);
    if (c->copyStream)
    {
        (void)hipStreamSynchronize(c->copyStream);
    }
    if (c->stream)
    {
        (void)hipStreamSynchronize(c->stream);
    }
    for (auto& s : c->slots)
    {
        if (s.d_in)
        {
            (void)hipFree(s.d_in);
        }
        if (s.d_rec)
        {
            (void)hipFree(s.d_rec);
        }
        if (s.h_rec)
        {
            (void)hipHostFree(s.h_rec);
        }
        if (s.evH2D)
        {
            (void)hipEventDestroy(s.evH2D);
        }
        if (s.evDone)
        {
            (void)hipEventDestroy(s.evDone);
        }
    }
    c->slots.clear();
    if (c->copyStream)
    {
        (void)hipStreamDestroy(c->copyStream);
        c->copyStream = nullptr;
    }
    return ACF_HIP_OK;
}

int acf_hip_stream_open(acf_hip_ctx* c, int pix, int rowStride, int cap, int depth)
{
    if (!c || !c->hasPlan)
    {
        return c ? fail(c, ACF_HIP_E_NOPLAN, "stream_open: plan first") : ACF_HIP_E_INVALID;
    }
    if (!c->kids.empty())
    {
        return fail(c, ACF_HIP_E_UNSUPPORTED, "stream_open: not available with option \"streams\" > 1");
    }
    if (depth < 2 || depth > 8 || cap <= 0 || pix < ACF_HIP_PIX_RGB || pix > ACF_HIP_PIX_GRAY)
    {
        return fail(c, ACF_HIP_E_INVALID, "stream_open: depth must be 2..8, cap > 0, pix one of ACF_HIP_PIX_*");
    }
    const int cpp = (pix == ACF_HIP_PIX_GRAY) ? 1 : (pix == ACF_HIP_PIX_RGBA || pix == ACF_HIP_PIX_BGRA) ? 4 : 3;
    if ((cpp == 1) != (c->plan.d_in == 1))
    {
        return fail(c, ACF_HIP_E_INVALID, "stream_open: the plan's input planes (d) must be 3 for colour layouts and 1 for GRAY");
    }
    // (with acf_hip_set_input_resize the submitted frames are the unreduced ones)
    const int inRows = c->rz.on ? c->rz.rows : c->plan.H, inCols = c->rz.on ? c->rz.cols : c->plan.W;
    const int stride = rowStride > 0 ? rowStride : inCols * cpp;
    if (stride < inCols * cpp)
    {
        return fail(c, ACF_HIP_E_INVALID, "stream_open: row stride smaller than a row");
    }
    acf_hip_stream_close(c);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamCreateWithFlags(&c->copyStream, hipStreamNonBlocking));
    c->stPix = pix;
    c->stStride = stride;
    c->stCap = cap;
    c->nextTicket = c->nextCollect = 0;
    c->slots.resize(size_t(depth));
    const size_t inBytes = size_t(c->maxBatch) * stride * inRows;
    const size_t recBytes = size_t(c->maxBatch) * (1 + 6 * size_t(cap)) * sizeof(int32_t);
    for (auto& s : c->slots)
    {
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&s.d_in), inBytes));
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&s.d_rec), recBytes));
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&s.h_rec), recBytes, hipHostMallocDefault));
        HIPCHK(c, hipEventCreateWithFlags(&s.evH2D, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&s.evDone, hipEventDisableTiming));
        s.ticket = -1;
    }
    return ACF_HIP_OK;
}

int acf_hip_stream_submit(acf_hip_ctx* c, const uint8_t* frames_host, int nF, int* ticket)
{
    if (!c || c->slots.empty())
    {
        return c ? fail(c, ACF_HIP_E_INVALID, "stream_submit: stream_open first") : ACF_HIP_E_INVALID;
    }
    if (!frames_host || nF <= 0 || nF > c->maxBatch)
    {
        return fail(c, ACF_HIP_E_INVALID, "stream_submit: n_frames out of range");
    }
    const int depth = int(c->slots.size());
    if (c->nextTicket - c->nextCollect >= depth)
    {
        return fail(c, ACF_HIP_E_CAPACITY, "stream_submit: every slot is in flight; collect the oldest ticket first");
    }
    HIPCHK(c, hipSetDevice(c->device));
    acf_hip_ctx::StreamSlot& s = c->slots[size_t(c->nextTicket % depth)];
    // The slot's previous batch was collected (checked above), so its device input and host records are free.
    const size_t bytes = size_t(nF) * c->stStride * (c->rz.on ? c->rz.rows : c->plan.H);
    HIPCHK(c, hipMemcpyAsync(s.d_in, frames_host, bytes, hipMemcpyHostToDevice, c->copyStream));
    HIPCHK(c, hipEventRecord(s.evH2D, c->copyStream));
    HIPCHK(c, hipStreamWaitEvent(c->stream, s.evH2D, 0));
    int rc = acf_hip_run_u8(c, s.d_in, nF, c->stPix, c->stStride);
    if (rc)
    {
        return rc;
    }
    if ((rc = acf_hip_export_detections(c, s.d_rec, c->stCap)))
    {
        return rc;
    }
    HIPCHK(c, hipMemcpyAsync(s.h_rec, s.d_rec, size_t(nF) * (1 + 6 * size_t(c->stCap)) * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipEventRecord(s.evDone, c->stream));
    s.ticket = c->nextTicket;
    s.nFrames = nF;
    if (ticket)
    {
        *ticket = s.ticket;
    }
    c->nextTicket++;
    return ACF_HIP_OK;
}

int acf_hip_stream_collect(acf_hip_ctx* c, int ticket, const int32_t** records, int* nFrames)
{
    if (!c || c->slots.empty())
    {
        return c ? fail(c, ACF_HIP_E_INVALID, "stream_collect: stream_open first") : ACF_HIP_E_INVALID;
    }
    if (ticket != c->nextCollect || ticket >= c->nextTicket)
    {
        return fail(c, ACF_HIP_E_INVALID, "stream_collect: tickets are collected in submission order");
    }
    acf_hip_ctx::StreamSlot& s = c->slots[size_t(ticket % int(c->slots.size()))];
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventSynchronize(s.evDone));
    if (records)
    {
        *records = s.h_rec;
    }
    if (nFrames)
    {
        *nFrames = s.nFrames;
    }
    c->nextCollect++;
    return ACF_HIP_OK;
}

int acf_hip_host_alloc(size_t bytes, void** out)
{
    if (!out || bytes == 0)
    {
        return ACF_HIP_E_INVALID;
    }
    return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? ACF_HIP_OK : ACF_HIP_E_HIP;
}

int acf_hip_host_free(void* p)
{
    return (!p || hipHostFree(p) == hipSuccess) ? ACF_HIP_OK : ACF_HIP_E_HIP;
}

int acf_hip_synchronize(acf_hip_ctx* c)
{
    if (!c)
    {
        return ACF_HIP_E_INVALID;
    }
    for (acf_hip_ctx* k : c->kids)
    {
        HIPCHK(c, hipStreamSynchronize(k->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ACF_HIP_OK;
}

int acf_hip_profile_get(acf_hip_ctx* c, int* n, const char** names, float* ms, int* launches, int cap)
{
    if (!c || !n)
    {
        return ACF_HIP_E_INVALID;
    }
    if (!c->kids.empty())
    {
        // sum over the sub-batch contexts (their kernels overlap in time: the sum exceeds the wall clock)
        std::vector<const char*> nm;
        std::vector<float> tot;
        std::vector<int> cnt;
        for (acf_hip_ctx* k : c->kids)
        {
            int kn = 0;
            const char* knames[64];
            float kms[64];
            int kl[64];
            int rc = acf_hip_profile_get(k, &kn, knames, kms, kl, 64);
            if (rc)
            {
                return kidFail(c, k, rc);
            }
            for (int i = 0; i < std::min(kn, 64); i++)
            {
                size_t j = 0;
                for (; j < nm.size() && strcmp(nm[j], knames[i]); j++)
                {
                }
                if (j == nm.size())
                {
                    nm.push_back(knames[i]);
                    tot.push_back(0.f);
                    cnt.push_back(0);
                }
                tot[j] += kms[i];
                cnt[j] += kl[i];
            }
        }
        *n = int(nm.size());
        for (int i = 0; i < std::min(*n, cap); i++)
        {
            if (names) names[i] = nm[size_t(i)];
            if (ms) ms[i] = tot[size_t(i)];
            if (launches) launches[i] = cnt[size_t(i)];
        }
        return ACF_HIP_OK;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream)); // (the side streams were joined into it)
    std::vector<const char*> nm;
    std::vector<float> tot;
    std::vector<int> cnt;
    for (size_t i = 0; i + 1 < c->evUsed; i++)
    {
        const char* name = c->evName[i];
        if (!strcmp(name, "(end)"))
        {
            continue;
        }
        // the kernel ends where the next event of the same stream was recorded (real scales run on their own streams)
        size_t j = i + 1;
        while (j < c->evUsed && c->evStream[j] != c->evStream[i])
        {
            j++;
        }
        float dt = 0;
        if (j >= c->evUsed || hipEventElapsedTime(&dt, c->evPool[i], c->evPool[j]) != hipSuccess)
        {
            continue;
        }
        size_t k = 0;
        for (; k < nm.size(); k++)
        {
            if (!strcmp(nm[k], name))
            {
                break;
            }
        }
        if (k == nm.size())
        {
            nm.push_back(name);
            tot.push_back(0.f);
            cnt.push_back(0);
        }
        tot[k] += dt;
        cnt[k]++;
    }
    c->evUsed = 0;
    *n = int(nm.size());
    for (int k = 0; k < *n && k < cap; k++)
    {
        if (names)
        {
            names[k] = nm[k];
        }
        if (ms)
        {
            ms[k] = tot[k];
        }
        if (launches)
        {
            launches[k] = cnt[k];
        }
    }
    return ACF_HIP_OK;
}

static int fetchCounts(acf_hip_ctx* c)
{
    if (!c->detectValid)
    {
        return fail(c, ACF_HIP_E_INVALID, "no detections (call acf_hip_detect)");
    }
    if (!c->countsFetched)
    {
        HIPCHK(c, hipMemcpyAsync(c->h_counts.data(), nmsActive(c) ? c->d_nmsCounts : c->cs.d_counts, sizeof(int32_t) * c->lastBatch, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->countsFetched = true;
    }
    return ACF_HIP_OK;
}

static int checkNmsParams(acf_hip_ctx* c, const acf_hip_nms_params* q)
{
    if (q->type < 0 || q->type > 2 || !(q->overlap == q->overlap) || (q->prune && q->maxCount < 0))
    {
        return fail(c, ACF_HIP_E_INVALID, "nms: type 0..2, overlap a number, maxCount >= 0");
    }
    return ACF_HIP_OK;
}

int acf_hip_set_nms(acf_hip_ctx* c, const acf_hip_nms_params* q)
{
    if (!c)
    {
        return ACF_HIP_E_INVALID;
    }
    dropGraph(c);
    if (q)
    {
        const int rc = checkNmsParams(c, q);
        if (rc)
        {
            return rc;
        }
    }
    for (acf_hip_ctx* k : c->kids)
    {
        (void)acf_hip_set_nms(k, q);
    }
    c->nmsOn = q != nullptr;
    if (q)
    {
        c->nms = *q;
    }
    c->detectValid = false; // the resident detections were produced under the previous setting
    return ACF_HIP_OK;
}

int acf_hip_op_nms(acf_hip_ctx* c, const int32_t* boxes, const double* scores, int n, const acf_hip_nms_params* q, int32_t* keep_idx, int* count)
{
    if (!c || !q || !count || n < 0 || (n > 0 && (!boxes || !scores || !keep_idx)))
    {
        return c ? fail(c, ACF_HIP_E_INVALID, "op_nms: arguments") : ACF_HIP_E_INVALID;
    }
    int rc = checkNmsParams(c, q);
    if (rc)
    {
        return rc;
    }
    if (n == 0 || q->type == 0)
    {
        // bbNms returns its input for an empty list and for type "none" (bbNms.cpp:262-273); prune still applies to the caller
        for (int i = 0; i < n; i++)
        {
            keep_idx[i] = i;
        }
        *count = n;
        return ACF_HIP_OK;
    }
    if (n > NMS_CAP)
    {
        return fail(c, ACF_HIP_E_CAPACITY, "op_nms: more than ACF_HIP_NMS_CAP boxes");
    }
    HIPCHK(c, hipSetDevice(c->device));
    // one allocation, freed on every exit: boxes [n][4] int32 | scores [n] f64 | keep [NMS_CAP] int32 | count
    const size_t offSc = (size_t(n) * 16 + 7) / 8 * 8, offKeep = offSc + size_t(n) * 8, offN = offKeep + size_t(NMS_CAP) * 4;
    char* d_all = nullptr;
    HIPCHK(c, hipMalloc(&d_all, offN + 4));
    int32_t *d_box = reinterpret_cast<int32_t*>(d_all), *d_keep = reinterpret_cast<int32_t*>(d_all + offKeep), *d_n = reinterpret_cast<int32_t*>(d_all + offN);
    double* d_sc = reinterpret_cast<double*>(d_all + offSc);
    auto cleanup = [&]() { (void)hipFree(d_all); };
    NmsArgs a{};
    a.boxes = d_box;
    a.scores = d_sc;
    a.nOp = n;
    fillNmsArgs(a, *q);
    a.keep = d_keep;
    a.nKeep = d_n;
    hipError_t e = hipMemcpyAsync(d_box, boxes, size_t(n) * 16, hipMemcpyHostToDevice, c->stream);
    e = e ? e : hipMemcpyAsync(d_sc, scores, size_t(n) * 8, hipMemcpyHostToDevice, c->stream);
    if (!e && (rc = allowLds(c, reinterpret_cast<const void*>(&k_nms), kNmsLds)))
    {
        cleanup();
        return rc;
    }
    int32_t m = 0;
    if (!e)
    {
        hipLaunchKernelGGL(k_nms, dim3(1), dim3(1024), kNmsLds, c->stream, a);
        e = hipGetLastError();
    }
    e = e ? e : hipMemcpyAsync(&m, d_n, 4, hipMemcpyDeviceToHost, c->stream);
    e = e ? e : hipStreamSynchronize(c->stream);
    if (!e && m > 0)
    {
        e = hipMemcpy(keep_idx, d_keep, size_t(m) * 4, hipMemcpyDeviceToHost);
    }
    cleanup();
    if (e)
    {
        c->err = std::string("op_nms: ") + hipGetErrorString(e);
        return ACF_HIP_E_HIP;
    }
    *count = m;
    return ACF_HIP_OK;
}

int acf_hip_get_detections(acf_hip_ctx* c, int frame, acf_hip_detection* out, int cap, int* count)
{
    if (c && !c->kids.empty())
    {
        if (frame < 0 || frame >= c->lastBatch)
        {
            return fail(c, ACF_HIP_E_INVALID, "frame index");
        }
        acf_hip_ctx* k = c->kids[size_t(frame / c->kidChunk)];
        const int rc = acf_hip_get_detections(k, frame % c->kidChunk, out, cap, count);
        return rc ? kidFail(c, k, rc) : rc;
    }
    if (!c || !c->hasPlan)
    {
        return ACF_HIP_E_NOPLAN;
    }
    int rc = fetchCounts(c);
    if (rc)
    {
        return rc;
    }
    if (frame < 0 || frame >= c->lastBatch)
    {
        return fail(c, ACF_HIP_E_INVALID, "get_detections: frame index");
    }
    const int n = c->h_counts[frame];
    if (n < 0)
    {
        return fail(c, ACF_HIP_E_CAPACITY, "more than ACF_HIP_NMS_CAP detections into the device NMS: take the raw list (acf_hip_get_raw_detections) and suppress it on the host");
    }
    if (count)
    {
        *count = n;
    }
    const int m = std::min(std::min(n, c->maxHits), cap);
    if (m > 0 && out)
    {
        HIPCHK(c, hipMemcpy(out, (nmsActive(c) ? c->d_nmsDets : c->cs.d_dets) + size_t(frame) * c->maxHits, sizeof(acf_hip_detection) * m, hipMemcpyDeviceToHost));
    }
    if (n > c->maxHits)
    {
        return fail(c, ACF_HIP_E_CAPACITY, "more hits than max_hits; re-plan with a larger capacity");
    }
    return ACF_HIP_OK;
}

// raw (pre-NMS) count of one frame of the last batch
static int rawCount(acf_hip_ctx* c, int frame, int* n)
{
    if (!nmsActive(c))
    {
        *n = c->h_counts[frame];
        return ACF_HIP_OK;
    }
    int32_t v = 0;
    HIPCHK(c, hipMemcpyAsync(&v, c->cs.d_counts + frame, sizeof(v), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *n = v;
    return ACF_HIP_OK;
}

int acf_hip_get_raw_detections(acf_hip_ctx* c, int frame, acf_hip_detection* out, int cap, int* count)
{
    if (c && !c->kids.empty())
    {
        if (frame < 0 || frame >= c->lastBatch)
        {
            return fail(c, ACF_HIP_E_INVALID, "frame index");
        }
        acf_hip_ctx* k = c->kids[size_t(frame / c->kidChunk)];
        const int rc = acf_hip_get_raw_detections(k, frame % c->kidChunk, out, cap, count);
        return rc ? kidFail(c, k, rc) : rc;
    }
    if (!c || !c->hasPlan)
    {
        return ACF_HIP_E_NOPLAN;
    }
    int rc = fetchCounts(c);
    if (rc)
    {
        return rc;
    }
    if (frame < 0 || frame >= c->lastBatch)
    {
        return fail(c, ACF_HIP_E_INVALID, "get_raw_detections: frame index");
    }
    int n = 0;
    if ((rc = rawCount(c, frame, &n)))
    {
        return rc;
    }
    if (count)
    {
        *count = n;
    }
    const int m = std::min(std::min(n, c->maxHits), cap);
    if (m > 0 && out)
    {
        HIPCHK(c, hipMemcpy(out, c->cs.d_dets + size_t(frame) * c->maxHits, sizeof(acf_hip_detection) * m, hipMemcpyDeviceToHost));
    }
    if (n > c->maxHits)
    {
        return fail(c, ACF_HIP_E_CAPACITY, "more hits than max_hits; re-plan with a larger capacity");
    }
    return ACF_HIP_OK;
}

int acf_hip_get_hits(acf_hip_ctx* c, int frame, acf_hip_hit* out, int cap, int* count)
{
    if (c && !c->kids.empty())
    {
        if (frame < 0 || frame >= c->lastBatch)
        {
            return fail(c, ACF_HIP_E_INVALID, "frame index");
        }
        acf_hip_ctx* k = c->kids[size_t(frame / c->kidChunk)];
        const int rc = acf_hip_get_hits(k, frame % c->kidChunk, out, cap, count);
        return rc ? kidFail(c, k, rc) : rc;
    }
    if (!c || !c->hasPlan)
    {
        return ACF_HIP_E_NOPLAN;
    }
    int rc = fetchCounts(c);
    if (rc)
    {
        return rc;
    }
    if (frame < 0 || frame >= c->lastBatch)
    {
        return fail(c, ACF_HIP_E_INVALID, "get_hits: frame index");
    }
    const int n = c->h_counts[frame];
    if (nmsActive(c))
    {
        // hit i belongs to detection i of acf_hip_get_detections: the survivors' entries of the raw list, through k_nms's keep indices
        if (n < 0)
        {
            return fail(c, ACF_HIP_E_CAPACITY, "more than ACF_HIP_NMS_CAP detections into the device NMS: take the raw list (acf_hip_get_raw_detections)");
        }
        if (n > c->maxHits)
        {
            return fail(c, ACF_HIP_E_CAPACITY, "more hits than max_hits; re-plan with a larger capacity");
        }
        if (count)
        {
            *count = n;
        }
        const int m = std::min(n, cap);
        if (m > 0 && out)
        {
            int nRaw = 0;
            if ((rc = rawCount(c, frame, &nRaw)))
            {
                return rc;
            }
            nRaw = std::min(nRaw, c->maxHits);
            std::vector<int32_t> keep(static_cast<size_t>(m));
            std::vector<acf_hip_hit> raw(static_cast<size_t>(std::max(nRaw, 1)));
            HIPCHK(c, hipMemcpy(keep.data(), c->d_nmsKeep + size_t(frame) * NMS_CAP, sizeof(int32_t) * m, hipMemcpyDeviceToHost));
            HIPCHK(c, hipMemcpy(raw.data(), c->cs.d_sorted + size_t(frame) * c->maxHits, sizeof(acf_hip_hit) * nRaw, hipMemcpyDeviceToHost));
            for (int i = 0; i < m; i++)
            {
                if (keep[size_t(i)] < 0 || keep[size_t(i)] >= nRaw)
                {
                    return fail(c, ACF_HIP_E_INVALID, "get_hits: survivor index outside the raw list");
                }
                out[i] = raw[size_t(keep[size_t(i)])];
            }
        }
        return ACF_HIP_OK;
    }
    if (count)
    {
        *count = n;
    }
    const int m = std::min(std::min(n, c->maxHits), cap);
    if (m > 0 && out)
    {
        HIPCHK(c, hipMemcpy(out, c->cs.d_sorted + size_t(frame) * c->maxHits, sizeof(acf_hip_hit) * m, hipMemcpyDeviceToHost));
    }
    if (n > c->maxHits)
    {
        return fail(c, ACF_HIP_E_CAPACITY, "more hits than max_hits; re-plan with a larger capacity");
    }
    return ACF_HIP_OK;
}

int acf_hip_export_detections(acf_hip_ctx* c, int32_t* dst_dev, int cap)
{
    if (c && !c->kids.empty())
    {
        if (!c->detectValid || !dst_dev || cap <= 0)
        {
            return fail(c, ACF_HIP_E_INVALID, "export_detections: nothing to export");
        }
        int rc = kidsFork(c);
        for (size_t i = 0; i < c->kids.size() && !rc; i++)
        {
            if (kidCount(c, i, c->lastBatch) > 0 &&
                (rc = acf_hip_export_detections(c->kids[i], dst_dev + i * size_t(c->kidChunk) * (1 + 6 * size_t(cap)), cap)))
            {
                return kidFail(c, c->kids[i], rc);
            }
        }
        return rc ? rc : kidsJoin(c);
    }
    if (!c || !c->hasPlan)
    {
        return ACF_HIP_E_NOPLAN;
    }
    if (!c->detectValid || !dst_dev || cap <= 0)
    {
        return fail(c, ACF_HIP_E_INVALID, "export_detections: nothing to export");
    }
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_export, dim3(cdiv(cap, 256), c->lastBatch), dim3(256), 0, c->stream,
        (const acf_hip_detection*)(nmsActive(c) ? c->d_nmsDets : c->cs.d_dets), (const int32_t*)(nmsActive(c) ? c->d_nmsCounts : c->cs.d_counts), c->maxHits, cap, dst_dev);
    LAUNCHCHK(c, "k_export");
    return ACF_HIP_OK;
}

int acf_hip_read_level(acf_hip_ctx* c, int frame, int level, float* host_out)
{
    if (c && !c->kids.empty())
    {
        if (frame < 0 || frame >= c->lastBatch)
        {
            return fail(c, ACF_HIP_E_INVALID, "frame index");
        }
        acf_hip_ctx* k = c->kids[size_t(frame / c->kidChunk)];
        const int rc = acf_hip_read_level(k, frame % c->kidChunk, level, host_out);
        return rc ? kidFail(c, k, rc) : rc;
    }
    if (!c || !c->hasPlan)
    {
        return ACF_HIP_E_NOPLAN;
    }
    if (!c->pyramidValid || frame < 0 || frame >= c->lastBatch || level < 0 || level >= int(c->plan.levels.size()) || !host_out)
    {
        return fail(c, ACF_HIP_E_INVALID, "read_level: arguments");
    }
    if (!c->floatPyramid)
    {
        return fail(c, ACF_HIP_E_INVALID, "read_level: the float pyramid was not kept (option keep_pyramid = 0: the levels left as rank cells only)");
    }
    const acf_hip_level& l = c->plan.levels[level];
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(host_out, c->d_pyr + size_t(frame) * c->plan.pyr_floats + l.offset, sizeof(float) * c->plan.nChns * l.hP * l.wP, hipMemcpyDeviceToHost));
    return ACF_HIP_OK;
}

int acf_hip_read_rank_level(acf_hip_ctx* c, int frame, int level, uint16_t* host_out)
{
    if (c && !c->kids.empty())
    {
        if (frame < 0 || frame >= c->lastBatch)
        {
            return fail(c, ACF_HIP_E_INVALID, "frame index");
        }
        acf_hip_ctx* k = c->kids[size_t(frame / c->kidChunk)];
        const int rc = acf_hip_read_rank_level(k, frame % c->kidChunk, level, host_out);
        return rc ? kidFail(c, k, rc) : rc;
    }
    if (!c || !c->hasPlan)
    {
        return ACF_HIP_E_NOPLAN;
    }
    if (!c->cs.useRank || c->noRank)
    {
        return fail(c, ACF_HIP_E_UNSUPPORTED, "read_rank_level: the plan has no rank cells (model not depth 2 / thresholds too dense / option rank_cells off / LDCF)");
    }
    if (!c->pyramidValid || !c->ranksValid || frame < 0 || frame >= c->lastBatch || level < 0 || level >= int(c->plan.levels.size()) || !host_out)
    {
        return fail(c, ACF_HIP_E_INVALID, "read_rank_level: arguments, or no detect since the last pyramid");
    }
    const acf_hip_level& l = c->plan.levels[level];
    const int pitch = rankPitch(l.hP);
    int64_t off = 0;
    for (int i = 0; i < level; i++)
    {
        off += int64_t(c->plan.nChns) * rankPitch(c->plan.levels[i].hP) * c->plan.levels[i].wP;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy2D(host_out, size_t(l.hP) * 2, c->cs.d_pyrR + size_t(frame) * c->cs.pyrRCells + off, size_t(pitch) * 2, size_t(l.hP) * 2,
        size_t(c->plan.nChns) * l.wP, hipMemcpyDeviceToHost));
    return ACF_HIP_OK;
}

int acf_hip_rank_cells_host(const acf_hip_params* p, int nChns, int chn, const float* v, int n, uint16_t* cells_out, uint32_t* thr_index_out, int32_t* info)
{
    if (!p || !p->fids || !p->thrs || nChns <= 0 || chn < 0 || chn >= nChns || n < 0 || (n > 0 && !v) || p->treeDepth <= 0 || p->shrink <= 0)
    {
        return ACF_HIP_E_INVALID;
    }
    const int mH = p->modelDsPad_h / p->shrink, mW = p->modelDsPad_w / p->shrink;
    if (mH <= 0 || mW <= 0)
    {
        return ACF_HIP_E_INVALID;
    }
    const size_t nNodes = size_t(p->nTrees) * p->nTreeNodes;
    std::vector<int32_t> chnOfNode(nNodes, -1);
    for (int t = 0; t < p->nTrees; t++)
    {
        for (int k = 0; k < (1 << p->treeDepth) - 1 && k < p->nTreeNodes; k++)
        {
            const size_t q = size_t(t) * p->nTreeNodes + k;
            chnOfNode[q] = int32_t(p->fids[q] / uint32_t(mW * mH));
        }
    }
    RankTables rt;
    buildRankTables(p->thrs, chnOfNode.data(), nNodes, nChns, rt);
    if (info)
    {
        info[0] = rt.ok ? 1 : 0;
        info[1] = rt.ok ? rt.chan[size_t(chn)].shift : 0;
        info[2] = rt.ok ? rt.chan[size_t(chn)].nb : 0;
        info[3] = rt.ok ? rt.chan[size_t(chn)].nThr : 0;
    }
    if (!rt.ok)
    {
        return ACF_HIP_E_UNSUPPORTED;
    }
    for (int i = 0; i < n && cells_out; i++)
    {
        cells_out[i] = uint16_t(rt.rankOfCell(chn, v[i]));
    }
    for (size_t q = 0; q < nNodes && thr_index_out; q++)
    {
        thr_index_out[q] = chnOfNode[q] >= 0 ? rt.rankOfThreshold(chnOfNode[q], p->thrs[q]) : 0u;
    }
    return ACF_HIP_OK;
}

int acf_hip_read_tap(acf_hip_ctx* c, int frame, int tap, int index, float* host_out, int64_t cap)
{
    if (c && !c->kids.empty())
    {
        if (frame < 0 || frame >= c->lastBatch)
        {
            return fail(c, ACF_HIP_E_INVALID, "frame index");
        }
        acf_hip_ctx* k = c->kids[size_t(frame / c->kidChunk)];
        const int rc = acf_hip_read_tap(k, frame % c->kidChunk, tap, index, host_out, cap);
        return rc ? kidFail(c, k, rc) : rc;
    }
    if (!c || !c->hasPlan)
    {
        return ACF_HIP_E_NOPLAN;
    }
    if (!c->pyramidValid || frame < 0 || frame >= c->lastBatch || !host_out)
    {
        return fail(c, ACF_HIP_E_INVALID, "read_tap: arguments");
    }
    const Plan& pl = c->plan;
    const float* src = nullptr;
    int64_t n = 0;
    if (tap == ACF_HIP_TAP_LDCF)
    {
        if (c->p.ldcfK <= 0 || !c->detectValid || index < 0 || index >= int(c->ldcfLevels.size()))
        {
            return fail(c, ACF_HIP_E_INVALID, "read_tap: LDCF levels exist after acf_hip_detect on a model with ldcfK > 0");
        }
        const acf_hip_level& l = c->ldcfLevels[size_t(index)];
        n = int64_t(pl.nChns) * c->p.ldcfK * l.hP * l.wP;
        src = c->d_ldcfPyr + size_t(frame) * c->ldcfFloats + l.offset;
    }
    else if (tap == ACF_HIP_TAP_CHNS)
    {
        if (index < 0 || index >= int(pl.levels.size()))
        {
            return fail(c, ACF_HIP_E_INVALID, "read_tap: level");
        }
        const acf_hip_level& l = pl.levels[index];
        n = int64_t(pl.nChns) * l.hC * l.wC;
        src = c->d_chns + size_t(frame) * pl.raw_floats + pl.raw_off[index];
    }
    else
    {
        if (index < 0 || index >= int(c->real.size()))
        {
            return fail(c, ACF_HIP_E_INVALID, "read_tap: real-scale ordinal");
        }
        const RealScale& rs = c->real[index];
        const int64_t np = int64_t(rs.h) * rs.w;
        switch (tap)
        {
            case ACF_HIP_TAP_IMAGE:
                n = np * pl.d;
                if (rs.resampled)
                {
                    src = rs.img + size_t(frame) * n;
                }
                else if (c->d_color)
                {
                    src = c->d_color + size_t(frame) * n;
                }
                else
                {
                    src = c->lastFrames + size_t(frame) * n;
                }
                break;
            case ACF_HIP_TAP_SMOOTHED:
                n = np * pl.d;
                src = rs.sm + size_t(frame) * n;
                break;
            case ACF_HIP_TAP_M:
                n = np;
                src = rs.M ? rs.M + size_t(frame) * n : nullptr;
                break;
            case ACF_HIP_TAP_O:
                n = np;
                src = rs.O ? rs.O + size_t(frame) * n : nullptr;
                break;
            case ACF_HIP_TAP_S:
                n = np;
                src = rs.S ? rs.S + size_t(frame) * n : nullptr;
                break;
            case ACF_HIP_TAP_MNORM:
                n = np;
                src = rs.Mn ? rs.Mn + size_t(frame) * n : nullptr;
                break;
            default:
                return fail(c, ACF_HIP_E_INVALID, "read_tap: unknown tap");
        }
        if (!src)
        {
            return fail(c, ACF_HIP_E_INVALID, "read_tap: tap not kept (set option \"taps\" before acf_hip_plan)");
        }
    }
    if (cap < n)
    {
        return fail(c, ACF_HIP_E_INVALID, "read_tap: capacity");
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(host_out, src, sizeof(float) * n, hipMemcpyDeviceToHost));
    return ACF_HIP_OK;
}

/* ---- single operators on host planes ---------------------------------- */


#define OP_PROLOGUE(c)                      \
    if (!(c))                               \
    {                                       \
        return ACF_HIP_E_INVALID;           \
    }                                       \
    HIPCHK(c, hipSetDevice((c)->device));   \
    {                                       \
        int rc0_ = ensureConstTables(c);    \
        if (rc0_)                           \
        {                                   \
            return rc0_;                    \
        }                                   \
    }

int acf_hip_op_rgb_convert(acf_hip_ctx* c, const float* in, float* out, int h, int w, int flag)
{
    OP_PROLOGUE(c);
    if (!in || !out || h <= 0 || w <= 0)
    {
        return fail(c, ACF_HIP_E_INVALID, "op_rgb_convert: arguments");
    }
    const int n = h * w;
    Scratch s;
    float* di = s.upload(in, size_t(3) * n);
    const int dOut = flag == ACF_HIP_CS_GRAY ? 1 : 3;
    float* dout = s.alloc<float>(size_t(dOut) * n);
    if (!di || !dout)
    {
        return fail(c, ACF_HIP_E_HIP, "op_rgb_convert: allocation");
    }
    dim3 grid(cdiv(n, 256), 1, 1), block(256);
    if (flag == ACF_HIP_CS_LUV)
    {
        if (n % 4 == 0)
        {
            hipLaunchKernelGGL(k_rgb2luv<true>, grid, block, 0, c->stream, (const float*)di, dout, (const float*)c->d_lTable, makeLuvConsts(), n, int64_t(0), int64_t(0), x86T(c));
        }
        else
        {
            hipLaunchKernelGGL(k_rgb2luv<false>, grid, block, 0, c->stream, (const float*)di, dout, (const float*)c->d_lTable, makeLuvConsts(), n, int64_t(0), int64_t(0), x86T(c));
        }
    }
    else if (flag == ACF_HIP_CS_GRAY)
    {
        const float mr = (float).2989360213 * 1.0f, mg = (float).5870430745 * 1.0f, mb = (float).1140209043 * 1.0f;
        hipLaunchKernelGGL(k_rgb2gray<false>, grid, block, 0, c->stream, (const float*)di, dout, n, int64_t(0), int64_t(0), mr, mg, mb);
    }
    else if (flag == ACF_HIP_CS_HSV)
    {
        hipLaunchKernelGGL(k_rgb2hsv, grid, block, 0, c->stream, (const float*)di, dout, n, int64_t(0), int64_t(0));
    }
    else
    {
        return fail(c, ACF_HIP_E_UNSUPPORTED, "op_rgb_convert: flag");
    }
    LAUNCHCHK(c, "op_rgb_convert");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, dout, sizeof(float) * dOut * n, hipMemcpyDeviceToHost));
    return ACF_HIP_OK;
}

int acf_hip_op_conv_tri(acf_hip_ctx* c, const float* in, float* out, int h, int w, int d, double r, int aliased)
{
    OP_PROLOGUE(c);
    if (!in || !out || h <= 0 || w <= 0 || d <= 0)
    {
        return fail(c, ACF_HIP_E_INVALID, "op_conv_tri: arguments");
    }
    const int m = std::min(h, w);
    if (m < 4 || (2 * r + 1) >= m)
    {
        return fail(c, ACF_HIP_E_UNSUPPORTED, "op_conv_tri: plane too small (sepFilter2D fallback, convTri.cpp:224-251)");
    }
    const int64_t np = int64_t(h) * w;
    Scratch s;
    float* di = s.upload(in, size_t(np) * d);
    float* dout = s.alloc<float>(size_t(np) * d);
    if (!di || !dout)
    {
        return fail(c, ACF_HIP_E_HIP, "op_conv_tri: allocation");
    }
    int rc;
    if (r > 0 && r <= 1.0)
    {
        SmoothJob j{};
        j.h = h;
        j.w = w;
        j.nplanes = d;
        j.out_cs = h;
        j.in_ps = np;
        j.out_ps = np;
        SmoothJob* dj = s.upload(&j, 1);
        if (!dj)
        {
            return fail(c, ACF_HIP_E_HIP, "op_conv_tri: allocation");
        }
        const float p = float(12.0 / r / (r + 2.0) - 2.0);
        if ((rc = launchSmooth(c, di, dout, dj, 1, d, h, 0, 0, 1, p, aliased != 0)))
        {
            return rc;
        }
    }
    else if (r > 1)
    {
        const int ri = int(std::round(float(r)));
        if (ri >= m / 2)
        {
            return fail(c, ACF_HIP_E_INVALID, "op_conv_tri: mask larger than image (convTri.cpp:166-169)");
        }
        float* dU = s.alloc<float>(size_t(np) * d);
        if (!dU)
        {
            return fail(c, ACF_HIP_E_HIP, "op_conv_tri: allocation");
        }
        if ((rc = launchTri(c, di, dU, dout, h, w, ri, np, d)))
        {
            return rc;
        }
    }
    else
    {
        return fail(c, ACF_HIP_E_UNSUPPORTED, "op_conv_tri: radius");
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, dout, sizeof(float) * np * d, hipMemcpyDeviceToHost));
    return ACF_HIP_OK;
}

// the device's table functions on the host (kernels.hip.h: x86_rcp_bits), for the composed table of gm_inv_x86g
static uint32_t hostX86RcpBits(uint32_t u, const uint32_t* rcp4096)
{
    const uint32_t s = u & 0x80000000u, e = (u >> 23) & 0xffu, m = u & 0x7fffffu;
    if (e == 0xffu)
    {
        return m ? (u | 0x400000u) : s;
    }
    if (e == 0)
    {
        return s | 0x7f800000u;
    }
    const uint32_t t = rcp4096[m >> 11];
    const int re = int((t >> 23) & 0xffu) + 127 - int(e);
    return re <= 0 ? s : (s | (uint32_t(re) << 23) | (t & 0x7fffffu));
}

int acf_hip_set_x86_tables(acf_hip_ctx* c, const uint32_t* rcp4096, const uint32_t* rsqrt8192)
{
    OP_PROLOGUE(c);
    if (!rcp4096 || !rsqrt8192)
    {
        return fail(c, ACF_HIP_E_INVALID, "set_x86_tables: null table");
    }
    dropGraph(c);
    if (!c->d_x86)
    {
        // (not one of the plan's buffers: it outlives a re-plan; freed by acf_hip_destroy)
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_x86), X86_BUF_N * sizeof(uint32_t)));
        c->x86Owned = true;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream)); // (a run in flight may be reading the previous tables)
    // [rcp 4096][rsqrt 8192][gradMag's pairs {RSQ[i], rcp(RSQ[i])} x 8192][rcp(1e10)] (kernels_channels.hip.h: gm_inv_x86g)
    std::vector<uint32_t> buf(X86_BUF_N, 0u);
    memcpy(buf.data(), rcp4096, 4096 * sizeof(uint32_t));
    memcpy(buf.data() + 4096, rsqrt8192, 8192 * sizeof(uint32_t));
    for (int i = 0; i < X86_GM_N; i++)
    {
        buf[size_t(X86_TABLE_N) + 2 * size_t(i)] = rsqrt8192[i];
        buf[size_t(X86_TABLE_N) + 2 * size_t(i) + 1] = hostX86RcpBits(rsqrt8192[i], rcp4096);
    }
    {
        const float big = 1e10f;
        uint32_t bb;
        memcpy(&bb, &big, 4);
        buf[size_t(X86_TABLE_N) + 2 * size_t(X86_GM_N)] = hostX86RcpBits(bb, rcp4096);
    }
    HIPCHK(c, hipMemcpy(c->d_x86, buf.data(), buf.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    for (acf_hip_ctx* k : c->kids)
    {
        k->d_x86 = c->d_x86;
    }
    return ACF_HIP_OK;
}

int acf_hip_selftest_x86(acf_hip_ctx* c, uint32_t first_bits, uint64_t count, uint32_t stride, uint64_t digest[3])
{
    OP_PROLOGUE(c);
    if (!digest || !c->d_x86)
    {
        return fail(c, ACF_HIP_E_INVALID, "selftest_x86: install tables first (acf_hip_set_x86_tables)");
    }
    Scratch s;
    unsigned long long* d = s.alloc<unsigned long long>(3);
    if (!d)
    {
        return fail(c, ACF_HIP_E_HIP, "selftest_x86: allocation");
    }
    HIPCHK(c, hipMemset(d, 0, 3 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_x86_digest, dim3(4096), dim3(256), 0, c->stream, (const uint32_t*)c->d_x86, first_bits, (unsigned long long)count, stride, d);
    LAUNCHCHK(c, "k_x86_digest");
    // ... and gradMag's one-read form against the two table functions over the same patterns (negative non-NaN ones skipped)
    hipLaunchKernelGGL(k_gm_x86_selftest, dim3(4096), dim3(256), 0, c->stream, (const uint32_t*)c->d_x86, first_bits, (unsigned long long)count, stride, d + 2);
    LAUNCHCHK(c, "k_gm_x86_selftest");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    unsigned long long out[3];
    HIPCHK(c, hipMemcpy(out, d, sizeof(out), hipMemcpyDeviceToHost));
    digest[0] = out[0];
    digest[1] = out[1];
    digest[2] = out[2];
    return ACF_HIP_OK;
}

int acf_hip_selftest_gradmag(acf_hip_ctx* c, uint32_t first_bits, uint32_t last_bits, uint64_t* mismatches, uint32_t* first_bad_bits)
{
    OP_PROLOGUE(c);
    if (!mismatches || last_bits < first_bits)
    {
        return fail(c, ACF_HIP_E_INVALID, "selftest_gradmag: arguments");
    }
    Scratch s;
    unsigned long long* d = s.alloc<unsigned long long>(2);
    if (!d)
    {
        return fail(c, ACF_HIP_E_HIP, "selftest_gradmag: allocation");
    }
    const unsigned long long init[2] = { 0ull, ~0ull };
    HIPCHK(c, hipMemcpy(d, init, sizeof(init), hipMemcpyHostToDevice));
    const unsigned long long count = (unsigned long long)(last_bits - first_bits) + 1ull;
    hipLaunchKernelGGL(k_gm_inv_selftest, dim3(4096), dim3(256), 0, c->stream, first_bits, count, d);
    LAUNCHCHK(c, "k_gm_inv_selftest");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    unsigned long long out[2];
    HIPCHK(c, hipMemcpy(out, d, sizeof(out), hipMemcpyDeviceToHost));
    *mismatches = out[0];
    if (first_bad_bits)
    {
        *first_bad_bits = uint32_t(out[1]);
    }
    return ACF_HIP_OK;
}

int acf_hip_op_gradient_mag(acf_hip_ctx* c, const float* in, float* M, float* O, float* S_out, int h, int w, int normRad, double normConst, int full)
{
    OP_PROLOGUE(c);
    if (!in || !M || !O || h < 2 || w < 2)
    {
        return fail(c, ACF_HIP_E_INVALID, "op_gradient_mag: arguments");
    }
    const int64_t np = int64_t(h) * w;
    Scratch s;
    float* di = s.upload(in, size_t(np));
    float *dM = s.alloc<float>(np), *dO = s.alloc<float>(np), *dU = s.alloc<float>(np), *dS = s.alloc<float>(np);
    if (!di || !dM || !dO || !dU || !dS)
    {
        return fail(c, ACF_HIP_E_HIP, "op_gradient_mag: allocation");
    }
    hipLaunchKernelGGL(k_grad_mag_strip, dim3(cdiv(h, GM_ROWS), 1, 1), dim3(GM_ROWS), 0, c->stream, (const float*)di, dM, dO, (const float*)c->d_acos, h, w, full, int64_t(0), int64_t(0), cdiv(w, GM_XT), x86T(c));
    LAUNCHCHK(c, "k_grad_mag");
    if (normRad)
    {
        if (std::min(h, w) < 4 || 2 * normRad + 1 >= std::min(h, w) || normRad < 2)
        {
            return fail(c, ACF_HIP_E_UNSUPPORTED, "op_gradient_mag: normRad");
        }
        int rc = launchTri(c, dM, dU, dS, h, w, normRad, np, 1);
        if (rc)
        {
            return rc;
        }
        hipLaunchKernelGGL(k_norm, dim3(cdiv(np, 256)), dim3(256), 0, c->stream, dM, (const float*)dS, int(np), float(normConst), x86T(c));
        LAUNCHCHK(c, "k_norm");
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(M, dM, sizeof(float) * np, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(O, dO, sizeof(float) * np, hipMemcpyDeviceToHost));
    if (S_out && normRad)
    {
        HIPCHK(c, hipMemcpy(S_out, dS, sizeof(float) * np, hipMemcpyDeviceToHost));
    }
    return ACF_HIP_OK;
}

// Detector::chnsCompute (ACF.h:342-349, chnsCompute.cpp:146-338): the channels of ONE image at its own scale, no pyramid plan.
int acf_hip_chns_compute(acf_hip_ctx* c, const acf_hip_params* pIn, const float* in, int h, int w, int d, float* out, int64_t cap, int* nChnsOut, int* hCOut,
    int* wCOut)
{
    OP_PROLOGUE(c);
    const acf_hip_params* pp = pIn ? pIn : (c->hasModel ? &c->p : nullptr);
    if (!pp)
    {
        return fail(c, ACF_HIP_E_INVALID, "chns_compute: no parameters (pass them, or set a model first)");
    }
    acf_hip_params p = *pp;
    if (!in || h <= 0 || w <= 0 || (d != 1 && d != 3 && d != 5)) // (5: the image's own M, O planes behind three image planes, chnsCompute.cpp:219-226)
    {
        return fail(c, ACF_HIP_E_INVALID, "chns_compute: arguments");
    }
    {
        std::string err;
        const int rcc = checkChnsParams(p, d, err); // (the checks acf_hip_plan makes on the same fields)
        if (rcc)
        {
            return fail(c, rcc, "chns_compute: " + err);
        }
    }
    if (p.nOrients < 1 || p.nOrients > 12)
    {
        return fail(c, ACF_HIP_E_UNSUPPORTED, "chns_compute: nOrients must be 1..12");
    }
    const int shrink = p.shrink;
    const int hc = h - h % shrink, wc = w - w % shrink; // crop so divisible by shrink (chnsCompute.cpp:203-217)
    const int dcol = colorPlanes(p), nC = numChannels(p);
    const int hs = hc / shrink, ws = wc / shrink;
    if (nChnsOut)
    {
        *nChnsOut = nC;
    }
    if (hCOut)
    {
        *hCOut = hs;
    }
    if (wCOut)
    {
        *wCOut = ws;
    }
    if (!out)
    {
        return ACF_HIP_OK; // (a size query)
    }
    if (hc < 2 || wc < 2 || hs < 1 || ws < 1)
    {
        return fail(c, ACF_HIP_E_INVALID, "chns_compute: image smaller than a cell");
    }
    if (cap < int64_t(nC) * hs * ws)
    {
        return fail(c, ACF_HIP_E_CAPACITY, "chns_compute: output buffer too small");
    }
    const int64_t np = int64_t(hc) * wc;
    Scratch s;
    // the cropped planes, tight: [d][wc][hc]
    std::vector<float> crop;
    const float* src = in;
    if (hc != h || wc != w)
    {
        crop.resize(size_t(d) * np);
        for (int z = 0; z < d; z++)
        {
            for (int x = 0; x < wc; x++)
            {
                memcpy(&crop[(size_t(z) * wc + x) * hc], in + (size_t(z) * w + x) * h, sizeof(float) * hc);
            }
        }
        src = crop.data();
    }
    float* dI = s.upload(src, size_t(d) * np);
    float* dCol = s.alloc<float>(size_t(dcol) * np);
    float *dM = s.alloc<float>(np), *dO = s.alloc<float>(np), *dU = s.alloc<float>(np), *dS = s.alloc<float>(np);
    float* dOut = s.alloc<float>(size_t(nC) * hs * ws);
    if (!dI || !dCol || !dM || !dO || !dU || !dS || !dOut)
    {
        return fail(c, ACF_HIP_E_HIP, "chns_compute: allocation");
    }
    // rgbConvert(I, I, colorSpace, true, isLuv) (chnsCompute.cpp:235; rgbConvert.cpp:101-170)
    const bool passthrough = (d >= 3) && (p.colorSpace == ACF_HIP_CS_ORIG || p.colorSpace == ACF_HIP_CS_RGB || (p.isLuv && p.colorSpace == ACF_HIP_CS_LUV));
    {
        dim3 grid(cdiv(np, 256), 1, 1), block(256);
        if (passthrough)
        {
            HIPCHK(c, hipMemcpyAsync(dCol, dI, sizeof(float) * 3 * np, hipMemcpyDeviceToDevice, c->stream));
        }
        else if (p.colorSpace == ACF_HIP_CS_LUV)
        {
            if (np % 4 == 0)
            {
                hipLaunchKernelGGL(k_rgb2luv<true>, grid, block, 0, c->stream, (const float*)dI, dCol, (const float*)c->d_lTable, makeLuvConsts(), int(np), int64_t(0), int64_t(0), x86T(c));
            }
            else
            {
                hipLaunchKernelGGL(k_rgb2luv<false>, grid, block, 0, c->stream, (const float*)dI, dCol, (const float*)c->d_lTable, makeLuvConsts(), int(np), int64_t(0), int64_t(0), x86T(c));
            }
        }
        else if (p.colorSpace == ACF_HIP_CS_GRAY)
        {
            const float mr = (float).2989360213 * 1.0f, mg = (float).5870430745 * 1.0f, mb = (float).1140209043 * 1.0f;
            if (d == 1)
            {
                hipLaunchKernelGGL(k_rgb2gray<true>, grid, block, 0, c->stream, (const float*)dI, dCol, int(np), int64_t(0), int64_t(0), mr, mg, mb);
            }
            else
            {
                hipLaunchKernelGGL(k_rgb2gray<false>, grid, block, 0, c->stream, (const float*)dI, dCol, int(np), int64_t(0), int64_t(0), mr, mg, mb);
            }
        }
        else if (p.colorSpace == ACF_HIP_CS_HSV)
        {
            hipLaunchKernelGGL(k_rgb2hsv, grid, block, 0, c->stream, (const float*)dI, dCol, int(np), int64_t(0), int64_t(0));
        }
        else
        {
            hipLaunchKernelGGL(k_replicate3, grid, block, 0, c->stream, (const float*)dI, dCol, int(np), int64_t(0), int64_t(0));
        }
        LAUNCHCHK(c, "chns_compute: colour conversion");
    }
    // convTri(I, I, pColor.smooth, 1) in place (chnsCompute.cpp:239)
    if (p.colorSmooth > 0)
    {
        if (p.colorSmooth > 1)
        {
            return fail(c, ACF_HIP_E_UNSUPPORTED, "chns_compute: pColor.smooth > 1 (convTri with a radius) is not built for the colour planes");
        }
        SmoothJob j{};
        j.h = hc;
        j.w = wc;
        j.nplanes = dcol;
        j.out_cs = hc;
        j.in_ps = np;
        j.out_ps = np;
        SmoothJob* dJ = s.upload(&j, 1);
        if (!dJ)
        {
            return fail(c, ACF_HIP_E_HIP, "chns_compute: allocation");
        }
        const float pColor = float(12.0 / p.colorSmooth / (p.colorSmooth + 2.0) - 2.0);
        int rc = launchSmooth(c, dCol, dCol, dJ, 1, dcol, hc, int64_t(dcol) * np, int64_t(dcol) * np, 1, pColor, true);
        if (rc)
        {
            return rc;
        }
    }
    ChnsArgs a{};
    a.sm = dCol;
    a.M = dM;
    a.S = dS;
    a.O = dO;
    a.chns = dOut;
    a.sm_fs = int64_t(dcol) * np;
    a.m_fs = np;
    a.chns_fs = int64_t(nC) * hs * ws;
    a.h = hc;
    a.w = wc;
    a.d = dcol;
    a.colorEnabled = p.colorEnabled;
    a.magEnabled = p.gradMagEnabled;
    a.histEnabled = p.gradHistEnabled;
    a.nOrients = p.nOrients;
    a.doNorm = p.normRad != 0;
    a.full = p.full;
    a.hardBin = p.softBin < 0;
    a.normConst = float(p.normConst);
    a.rq_y = shrinkGainY(shrink);
    a.x86 = x86T(c);
    if (d == 5)
    {
        // M = MO[0], O = MO[1] (chnsCompute.cpp:265-269): no gradientMag, no normalisation
        a.M = dI + 3 * np;
        a.O = dI + 4 * np;
        a.doNorm = 0;
    }
    else if (p.gradMagEnabled || p.gradHistEnabled)
    {
        hipLaunchKernelGGL(k_grad_mag_strip, dim3(cdiv(hc, GM_ROWS), 1, 1), dim3(GM_ROWS), 0, c->stream, (const float*)(dCol + int64_t(p.colorChn) * np), dM, dO,
            (const float*)c->d_acos, hc, wc, p.full, int64_t(0), int64_t(0), cdiv(wc, GM_XT), x86T(c));
        LAUNCHCHK(c, "chns_compute: k_grad_mag");
        if (p.normRad)
        {
            if (std::min(hc, wc) < 4 || 2 * p.normRad + 1 >= std::min(hc, wc) || p.normRad < 2)
            {
                return fail(c, ACF_HIP_E_UNSUPPORTED, "chns_compute: normRad against the image size");
            }
            int rc = launchTri(c, dM, dU, dS, hc, wc, p.normRad, np, 1);
            if (rc)
            {
                return rc;
            }
        }
    }
    int rc = launchChns(c, a, shrink, 1);
    if (rc)
    {
        return rc;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, dOut, sizeof(float) * size_t(nC) * hs * ws, hipMemcpyDeviceToHost));
    return ACF_HIP_OK;
}

int acf_hip_op_gradient_hist(acf_hip_ctx* c, const float* M, const float* O, float* H, int h, int w, int bin, int nOrients, int softBin, int full)
{
    OP_PROLOGUE(c);
    if (!M || !O || !H || h <= 0 || w <= 0 || nOrients < 1 || nOrients > 12)
    {
        return fail(c, ACF_HIP_E_INVALID, "op_gradient_hist: arguments");
    }
    if (softBin % 2 != 0)
    {
        return fail(c, ACF_HIP_E_UNSUPPORTED, "op_gradient_hist: odd softBin (trilinear spatial binning) is not built");
    }
    if ((bin != 2 && bin != 4) || h % bin || w % bin)
    {
        return fail(c, ACF_HIP_E_UNSUPPORTED, "op_gradient_hist: bin must be 2 or 4 and divide h, w");
    }
    const int64_t np = int64_t(h) * w, nc = np / (bin * bin);
    Scratch s;
    float *dM = s.upload(M, np), *dO = s.upload(O, np), *dH = s.alloc<float>(nc * nOrients);
    if (!dM || !dO || !dH)
    {
        return fail(c, ACF_HIP_E_HIP, "op_gradient_hist: allocation");
    }
    ChnsArgs a{};
    a.M = dM;
    a.S = dM;
    a.O = dO;
    a.chns = dH;
    a.h = h;
    a.w = w;
    a.d = 0;
    a.histEnabled = 1;
    a.nOrients = nOrients;
    a.full = full;
    a.hardBin = softBin < 0;
    a.rq_y = shrinkGainY(bin);
    int rc = launchChns(c, a, bin, 1);
    if (rc)
    {
        return rc;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(H, dH, sizeof(float) * nc * nOrients, hipMemcpyDeviceToHost));
    return ACF_HIP_OK;
}

int acf_hip_op_im_resample(acf_hip_ctx* c, const float* in, float* out, int ha, int wa, int hb, int wb, int d, double nrm)
{
    OP_PROLOGUE(c);
    if (!in || !out || ha <= 0 || wa <= 0 || hb <= 0 || wb <= 0 || d <= 0)
    {
        return fail(c, ACF_HIP_E_INVALID, "op_im_resample: arguments");
    }
    TableArena arena;
    ResampleDesc dd;
    int rc = buildResample(ha, wa, hb, wb, dd, arena);
    if (rc)
    {
        return fail(c, rc, "op_im_resample: degenerate geometry");
    }
    const double ratio[3] = { nrm, nrm, nrm };
    setResampleGain(dd, ratio, d, d);
    dd.nplanes = d;
    Scratch s;
    float* di = s.upload(in, size_t(d) * ha * wa);
    float* dout = s.alloc<float>(size_t(d) * hb * wb);
    dd.src_frame_stride = int64_t(d) * ha * wa;
    dd.dst_frame_stride = int64_t(d) * hb * wb;
    const StripPlan sp = stripPlan(dd, nullptr, arena); // appends its tile ranges to the int arena
    ResampleDesc* ddesc = s.upload(&dd, 1);
    int32_t* dit = s.upload(arena.ints.data(), arena.ints.size());
    float* dft = s.upload(arena.floats.data(), arena.floats.size());
    if (!di || !dout || !ddesc || !dit || !dft)
    {
        return fail(c, ACF_HIP_E_HIP, "op_im_resample: allocation");
    }
    if (sp.ok && !resampleGenericOnly() && (rc = ensureConstTables(c)) == 0)
    {
        // the strip march the pyramid uses for its down-sampled real scales
        launchStrip(c, sp, ddesc, 0, -1, d, di, dout, nullptr, dit, dft, 1);
    }
    else if (resampleUpOk(dd) && !resampleGenericOnly())
    {
        hipLaunchKernelGGL(k_resample_up, dim3(cdiv(d * cdiv(hb, 256) * cdiv(wb, RSU_XC), 4), 1, 1), dim3(256), 0, c->stream, (const float*)di, dout,
            (const ResampleDesc*)ddesc, (const int32_t*)dit, (const float*)dft);
    }
    else
    {
        hipLaunchKernelGGL(k_resample, dim3(resampleBlocks(dd), 1, 1), dim3(64, 4), 0, c->stream, (const float*)di, dout,
            (const ResampleDesc*)ddesc, (const int32_t*)dit, (const float*)dft, RS_XT);
    }
    LAUNCHCHK(c, "k_resample");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, dout, sizeof(float) * d * hb * wb, hipMemcpyDeviceToHost));
    return ACF_HIP_OK;
}

// thrs.convertTo(thrsU8, CV_8UC1, 255.0f) (ACFIOArchive.h:96-99): OpenCV's 32f->8u cvtScale works in f32 and rounds
// with cvRound (to nearest even), then saturates.
int acf_hip_thrs_u8(const float* thrs, int n, uint8_t* out)
{
    if (!thrs || !out || n < 0)
    {
        return ACF_HIP_E_INVALID;
    }
    for (int i = 0; i < n; i++)
    {
        const float v = thrs[i] * 255.0f;
        // cvRound is cvtss2si: half to even; NaN and |v| >= 2^31 convert to INT_MIN, which saturate_cast<uchar> maps to 0
        const bool inRange = v > -2147483648.0f && v < 2147483648.0f;
        const float r = inRange ? std::nearbyint(v) : -1.0f;
        out[i] = uint8_t(r < 0.f ? 0 : (r > 255.f ? 255 : int(r)));
    }
    return ACF_HIP_OK;
}

static int opAcfDetect1(acf_hip_ctx* c, const void* chns, bool u8, int hP, int wP, int nChns, acf_hip_hit* out, int cap, int* count);

int acf_hip_op_acf_detect1(acf_hip_ctx* c, const float* chns, int hP, int wP, int nChns, acf_hip_hit* out, int cap, int* count)
{
    return opAcfDetect1(c, chns, false, hP, wP, nChns, out, cap, count);
}

// The uint8_t body of acfDetect1 (acfDetect1.cpp:157-166,187-192): features and thresholds are bytes, compared after
// promotion to float (`float ftr = chns1[...]; ftr < thrs[k]`).  Bytes widen to f32 exactly, so the planes are widened
// on the device and the f32 cascade runs on them with the widened thresholds: same comparisons, same sums.
int acf_hip_op_acf_detect1_u8(acf_hip_ctx* c, const uint8_t* chns, int hP, int wP, int nChns, const uint8_t* thrsU8, acf_hip_hit* out, int cap, int* count)
{
    if (!c || !c->hasModel)
    {
        return c ? fail(c, ACF_HIP_E_NOMODEL, "op_acf_detect1_u8: set_model first") : ACF_HIP_E_INVALID;
    }
    std::vector<uint8_t> derived;
    if (!thrsU8)
    {
        derived.resize(c->thrs.size());
        acf_hip_thrs_u8(c->thrs.data(), int(c->thrs.size()), derived.data());
        thrsU8 = derived.data();
    }
    std::vector<float> wide(c->thrs.size());
    for (size_t i = 0; i < wide.size(); i++)
    {
        wide[i] = float(thrsU8[i]);
    }
    c->thrs.swap(wide);
    const int rc = opAcfDetect1(c, chns, true, hP, wP, nChns, out, cap, count);
    c->thrs.swap(wide);
    return rc;
}

static int opAcfDetect1(acf_hip_ctx* c, const void* chns, bool u8, int hP, int wP, int nChns, acf_hip_hit* out, int cap, int* count)
{
    OP_PROLOGUE(c);
    if (!c->hasModel)
    {
        return fail(c, ACF_HIP_E_NOMODEL, "op_acf_detect1: set_model first");
    }
    if (!chns || hP <= 0 || wP <= 0 || nChns <= 0 || cap <= 0 || !count)
    {
        return fail(c, ACF_HIP_E_INVALID, "op_acf_detect1: arguments");
    }
    const acf_hip_params& p = c->p;
    std::vector<acf_hip_level> lv(1);
    lv[0] = acf_hip_level{};
    lv[0].scale = 1.0;
    lv[0].scalehw_h = lv[0].scalehw_w = 1.0;
    lv[0].hP = lv[0].hC = hP;
    lv[0].wP = lv[0].wC = wP;
    lv[0].nWinR = std::max(0, int(std::ceil(float(hP * p.shrink - p.modelDsPad_h + 1) / p.stride)));
    lv[0].nWinC = std::max(0, int(std::ceil(float(wP * p.shrink - p.modelDsPad_w + 1) / p.stride)));
    lv[0].offset = 0;
    // temporary tables are registered in the context's allocation list only for the duration of this call;
    // the plan's cascade state is swapped out and back
    const size_t mark = c->allocs.size();
    const int savedMaxHits = c->maxHits, savedMaxBatch = c->maxBatch;
    const CascState saved = c->cs;
    c->cs = CascState{};
    auto restore = [&]() {
        for (size_t i = mark; i < c->allocs.size(); i++)
        {
            (void)hipFree(c->allocs[i]);
        }
        c->allocs.resize(mark);
        c->maxHits = savedMaxHits;
        c->maxBatch = savedMaxBatch;
        c->cs = saved;
    };
    BoxLevel* dBox = nullptr;
    float* dChn = nullptr;
    int rc = buildCascadeTables(c, lv, nChns, c->cs);
    std::vector<BoxLevel> box(1);
    box[0].shw_h = box[0].shw_w = 1.0;
    box[0].bw = p.modelDs_w;
    box[0].bh = p.modelDs_h;
    c->maxHits = cap;
    c->maxBatch = 1;
    if (!rc)
    {
        rc = devUpload(c, &dBox, box);
    }
    if (!rc)
    {
        rc = devUpload(c, &c->cs.d_thrs, c->thrs);
    }
    if (!rc)
    {
        rc = devUpload(c, &c->cs.d_hs, c->hs);
    }
    if (!rc)
    {
        rc = devUpload(c, &c->cs.d_child, c->child);
    }
    if (!rc)
    {
        rc = devUpload(c, &c->cs.d_fids, c->fids);
    }
    if (!rc)
    {
        rc = devAlloc(c, &c->cs.d_hits, size_t(cap));
    }
    if (!rc && c->cs.dedupQ > 1)
    {
        rc = devAlloc(c, &c->cs.d_hitsX, size_t(cap));
    }
    if (!rc)
    {
        rc = devAlloc(c, &c->cs.d_sorted, size_t(cap));
    }
    if (!rc)
    {
        rc = devAlloc(c, &c->cs.d_dets, size_t(cap));
    }
    if (!rc)
    {
        rc = devAlloc(c, &c->cs.d_counts, 1);
    }
    if (!rc)
    {
        c->cs.qcap = std::max(1, lv[0].nWinR * lv[0].nWinC);
        rc = devAlloc(c, &c->cs.d_queue[0], size_t(c->cs.qcap));
    }
    if (!rc)
    {
        rc = devAlloc(c, &c->cs.d_queue[1], size_t(c->cs.qcap));
    }
    if (!rc)
    {
        rc = devAlloc(c, &c->cs.d_qcounts, 16);
    }
    if (!rc)
    {
        rc = devAlloc(c, &dChn, size_t(nChns) * hP * wP + 64);
    }
    if (!rc && !u8 && hipMemcpy(dChn, chns, sizeof(float) * nChns * hP * wP, hipMemcpyHostToDevice) != hipSuccess)
    {
        rc = fail(c, ACF_HIP_E_HIP, "op_acf_detect1: upload");
    }
    if (!rc && u8)
    {
        uint8_t* dBytes = nullptr;
        const size_t n = size_t(nChns) * hP * wP;
        rc = devAlloc(c, &dBytes, n);
        if (!rc && hipMemcpy(dBytes, chns, n, hipMemcpyHostToDevice) != hipSuccess)
        {
            rc = fail(c, ACF_HIP_E_HIP, "op_acf_detect1_u8: upload");
        }
        if (!rc)
        {
            hipLaunchKernelGGL(k_widen_u8, dim3(cdiv(int64_t(n), 256)), dim3(256), 0, c->stream, (const uint8_t*)dBytes, dChn, int64_t(n));
            if (hipGetLastError() != hipSuccess)
            {
                rc = fail(c, ACF_HIP_E_HIP, "launch k_widen_u8");
            }
        }
    }
    if (!rc)
    {
        rc = runCascade(c, dChn, 0, dBox, 1, nChns);
    }
    int n = 0;
    if (!rc)
    {
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(&n, c->cs.d_counts, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
        {
            rc = fail(c, ACF_HIP_E_HIP, "op_acf_detect1: sync");
        }
    }
    if (!rc)
    {
        *count = n;
        const int m = std::min(n, cap);
        if (m > 0 && out && hipMemcpy(out, c->cs.d_sorted, sizeof(acf_hip_hit) * m, hipMemcpyDeviceToHost) != hipSuccess)
        {
            rc = fail(c, ACF_HIP_E_HIP, "op_acf_detect1: download");
        }
        if (!rc && n > cap)
        {
            rc = fail(c, ACF_HIP_E_CAPACITY, "op_acf_detect1: more hits than cap");
        }
    }
    restore();
    return rc;
}

int acf_hip_op_evaluate(acf_hip_ctx* c, const float* chns, int hP, int wP, int nChns, double cascThr, float* score)
{
    OP_PROLOGUE(c);
    if (!c->hasModel)
    {
        return fail(c, ACF_HIP_E_NOMODEL, "op_evaluate: set_model first");
    }
    const acf_hip_params& p = c->p;
    const int mH = p.modelDsPad_h / p.shrink, mW = p.modelDsPad_w / p.shrink;
    if (!chns || !score || nChns <= 0 || hP < mH || wP < mW)
    {
        return fail(c, ACF_HIP_E_INVALID, "op_evaluate: the buffer must hold at least one window");
    }
    Scratch s;
    const size_t n = size_t(nChns) * hP * wP;
    float* dC = s.upload(chns, n);
    uint32_t* dF = s.upload(c->fids.data(), c->fids.size());
    float* dT = s.upload(c->thrs.data(), c->thrs.size());
    float* dH = s.upload(c->hs.data(), c->hs.size());
    uint32_t* dCh = c->child.empty() ? nullptr : s.upload(c->child.data(), c->child.size());
    float* dS = s.alloc<float>(1);
    if (!dC || !dF || !dT || !dH || !dS || (p.treeDepth == 0 && !dCh))
    {
        return fail(c, ACF_HIP_E_HIP, "op_evaluate: allocation");
    }
    hipLaunchKernelGGL(k_evaluate_window, dim3(1), dim3(64), 0, c->stream, (const float*)dC, hP, wP, mH, mW, (const uint32_t*)dF, (const float*)dT,
        (const float*)dH, (const uint32_t*)dCh, p.nTrees, p.nTreeNodes, p.treeDepth, float(cascThr), dS);
    LAUNCHCHK(c, "k_evaluate_window");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(score, dS, sizeof(float), hipMemcpyDeviceToHost));
    return ACF_HIP_OK;
}

} // extern "C"
