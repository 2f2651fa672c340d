// C-ABI layer of libcudaParticleAdvection.so: context, device memory, call-order checks, errors.
// Every entry point declared in include/cpf.h is defined here; kernels live in cpf_kernels.hip
// and cpf_handoff.hip, the host-side mesh ingest in cpf_mesh.cpp.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "cpf.h"
#include "cpf_device.h"
#include "cpf_internal.h"

struct cpf_context {
    int device = 0;
    hipStream_t ownStream = nullptr, stream = nullptr;
    mutable std::string err;
    // mesh
    bool haveMesh = false, haveU = false;
    cpf::HostTables host;
    int32_t* d_cellOff = nullptr;
    double4* d_planes = nullptr;
    int32_t* d_nbr = nullptr;
    int32_t *d_groupOff = nullptr, *d_groupNbr = nullptr;
    bool zFold = true;               // "z_fold": mirror the kicked end point about the planes of a one-cell-thick mesh before the walk
    double4* d_U = nullptr;
    double* d_U3 = nullptr;     // staging for host uploads
    double* d_boxRec = nullptr;     // 128-byte box records (meshes of axis-aligned boxes with records; cpf_walk.h "box records")
    bool boxRecords = true;         // "box_records": 0 = never use them (diagnostics; bit-identical either way)
    double4* d_cellRec = nullptr;   // packed per-cell records (all-hex meshes; mixed meshes: cpf_walk.h "cell records")
    unsigned long long* d_occupied = nullptr;   // [0] occupied cells, [1] live particles of the last sort (device) ...
    hipEvent_t evFieldFlag = nullptr;           // recorded behind the read-back of "the field has a z component" (cpf_set_velocity_dev)
    bool fieldFlagPending = false;              // ... and not yet seen complete: until then the field counts as having one
    unsigned long long* h_occupied = nullptr;   // ... and their pinned host copy (StreamState::occupiedHost)
    int64_t nSecondRecords = 0;     // second records (cells with 7..12 slots), behind the nCells first ones
    float* d_cellBox = nullptr;     // per-cell boxes for the sort key
    int32_t* d_curveRank = nullptr; // per-cell rank along the Morton curve: the sort's major key for sparse clouds ("sort_curve")
    int sortMethod = 2;             // "sort_method": the (key, index) sort: 2 = this library's radix sort (cpf_kernels.hip, rt_sort_pairs), 0 = hipcub's;
                                    // the same order either way
    int sortCurve = -1;             // "sort_curve": -1 = Morton rank when the cloud has fewer than 8 particles per cell, 0 = cell id, 1 = Morton rank
    int32_t* d_binOff = nullptr;
    int32_t* d_binCells = nullptr;
    size_t meshBytes = 0;
    // owned cloud
    int64_t cap = 0, n = 0;
    double *x = nullptr, *y = nullptr, *z = nullptr, *vel = nullptr;
    int32_t* cell = nullptr;
    int64_t* gid = nullptr;
    // second set of x, y, z, cell, gid: the sort writes the reordered cloud there and the sets swap roles (no copy back);
    // allocated by the first sort
    double *x2 = nullptr, *y2 = nullptr, *z2 = nullptr;
    int32_t* cell2 = nullptr;
    int64_t* gid2 = nullptr;
    bool located = false;
    // scratch
    void* scratch = nullptr;
    size_t scratchBytes = 0;
    // counters / rng
    // [kCounterSlots][4] = steps, hops, reflections, lost, sharded by block id (one hot word would
    // serialise every block of every launch on a single L2 atomic unit), then 4 scratch words
    unsigned long long* d_counters = nullptr;
    uint32_t seed = 1591593751u;                // cuda/particles.cu:544
    uint32_t stepCounter = 0;
    int sortInterval = 50;                      // "sort_interval": cpf_step re-sorts the owned cloud by cell every N cycles
    uint32_t lastSortStep = 0;
    bool stats = false;                         // "stats": per-launch counters (steps, cells visited, reflections, lost)
    int stepVariant = -1;                       // cpf_set_option("step_variant"), see include/cpf.h: -1 = choose per launch
    bool vtuBinary = false;                     // cpf_set_option("vtu_binary"): frames with raw appended arrays instead of the reference's ASCII
    bool mixedRecords = true;                   // cpf_set_option("mixed_records"): build cell records for hex-dominant meshes too (before cpf_set_mesh)
    cpf::StreamState streamState;               // chunk counter + tuning of the streaming step kernel
    int64_t lastStepN = -1;                     // particle count of the most recent step launch (cpf_step_kernel_name)
    int lastStepCycles = 1;                     // ... and its cycles per launch
    // "VertexVelocity" advect only: the tet decomposition and one velocity per tet-mesh vertex
    double* d_tetPos = nullptr; int32_t* d_tets = nullptr; double* d_vertVel = nullptr;
    int64_t nTetVerts = 0, nTets = 0; int tetsPerCell = 0; bool haveVertVel = false;
    double* d_vertCone = nullptr;               // cone-locate tet records (cpf_walk.h, VertexField), if the decomposition admits them
    double* d_vertApex = nullptr;               // ... and the cells' apexes
    bool vertexFast = true;                     // cpf_set_option("vertex_fast")
    std::string vertConeWhy;                    // why the tables were not built (cpf_step_kernel_name says so)
    // asynchronous output (cpf_write_vtu_async): one frame in flight
    std::thread writer;
    bool writerLive = false;
    int writerStatus = CPF_OK;
    // the frame's snapshot: packed in particle-id order on the compute stream into `snapDev`, copied to pinned host memory on
    // `ioStream` behind an event, read by the worker thread only -- the step loop's stream never waits for PCIe (round 6)
    void* snapDev = nullptr; void* snapHost = nullptr;
    size_t snapBytes = 0;
    hipStream_t ioStream = nullptr;
    hipEvent_t evSnap = nullptr, evCopied = nullptr;
    double frameKE = 0.0;                       // of the frame the worker wrote last (valid after its join)
    std::mutex keMutex; std::condition_variable keCv; bool keReady = false;
    // timing
    int timingStride = 1;                       // "timing_stride": bracket every k-th step launch only
    uint64_t timingLaunch = 0;
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<hipEvent_t> eventPool;
};

namespace {

std::string g_createError = "";
std::mutex g_mutex;

// Contexts with an output frame still being written.  A host that never calls cpf_destroy (the reference's
// solvers simply return from main) must not lose its last frame: the registry's destructor runs at exit and
// joins the workers.
struct WriterRegistry {
    std::vector<cpf_context*> live;
    ~WriterRegistry();
} g_writers;

int fail(const cpf_context* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    else { std::lock_guard<std::mutex> lk(g_mutex); g_createError = msg; }
    return code;
}

#define CPF_HIP(ctx, call)                                                                              \
    do {                                                                                                \
        hipError_t e__ = (call);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return fail(ctx, CPF_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__));          \
    } while (0)

#define CPF_REQUIRE(ctx, cond, code, msg) \
    do { if (!(cond)) return fail(ctx, code, msg); } while (0)

template <typename T>
void freeDev(T*& p) { if (p) { (void)hipFree(p); p = nullptr; } }

cpf::MeshView meshView(const cpf_context* c) {
    cpf::MeshView m;
    m.cellOff = c->d_cellOff; m.planes = c->d_planes; m.nbr = c->d_nbr; m.U = c->d_U;
    m.groupOff = c->d_groupOff; m.groupNbr = c->d_groupNbr;
    m.cellRec = c->d_cellRec;
    m.boxRec = c->boxRecords ? reinterpret_cast<const double4*>(c->d_boxRec) : nullptr;
    m.nCells = (int32_t)c->host.nCells;
    m.allHex = (c->host.minCellFaces == 6 && c->host.maxCellFaces == 6 && c->host.nGroups() == 0) ? 1 : 0;
    m.zPairLast = c->host.zPairLast ? 1 : 0;
    m.zThin = (c->host.zThin && c->zFold) ? 1 : 0;
    m.zSide0 = c->host.zSide0 ? 1 : 0;
    m.mixed = (c->d_cellRec && !m.allHex) ? (c->host.nBigCells > 0 ? 2 : 1) : 0;      // 2: two-record and / or header cells
    return m;
}
cpf::GridView gridView(const cpf_context* c) {
    cpf::GridView g;
    for (int k = 0; k < 3; ++k) {
        g.origin[k] = c->host.origin[k]; g.invBin[k] = c->host.invBin[k];
        g.lo[k] = c->host.lo[k]; g.hi[k] = c->host.hi[k]; g.dims[k] = c->host.dims[k];
    }
    g.binOff = c->d_binOff; g.binCells = c->d_binCells;
    return g;
}

int ensureScratch(cpf_context* ctx, size_t bytes) {
    if (bytes <= ctx->scratchBytes) return CPF_OK;
    CPF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    freeDev(ctx->scratch);
    ctx->scratchBytes = 0;
    CPF_HIP(ctx, hipMalloc(&ctx->scratch, bytes));
    ctx->scratchBytes = bytes;
    return CPF_OK;
}

int sortEndBit(const cpf_context* ctx) {
    int bits = 1;
    while (((int64_t)1 << bits) < ctx->host.nCells + 2) ++bits;   // the all-ones key of lost/frozen sorts last
    return std::min(bits + ctx->host.subBits[0] + ctx->host.subBits[1] + ctx->host.subBits[2], 32);  // + sub-cell bits
}

void freeMesh(cpf_context* c) {
    freeDev(c->d_cellOff); freeDev(c->d_planes); freeDev(c->d_nbr); freeDev(c->d_groupOff); freeDev(c->d_groupNbr); freeDev(c->d_U); freeDev(c->d_U3); freeDev(c->d_cellRec); freeDev(c->d_boxRec); freeDev(c->d_cellBox); freeDev(c->d_curveRank);
    freeDev(c->d_binOff); freeDev(c->d_binCells);
    c->haveMesh = c->haveU = false; c->meshBytes = 0;
    c->nSecondRecords = 0;
    // the tet decomposition of the "VertexVelocity" mode belongs to the mesh it was made for (cpf_set_tets checks it against
    // that mesh's cell count): a new mesh starts without one
    freeDev(c->d_tetPos); freeDev(c->d_tets); freeDev(c->d_vertVel);
    c->haveVertVel = false; c->nTets = c->nTetVerts = 0; c->tetsPerCell = 0;
}
void freeCloud(cpf_context* c) {
    freeDev(c->x); freeDev(c->y); freeDev(c->z); freeDev(c->vel); freeDev(c->cell); freeDev(c->gid);
    freeDev(c->x2); freeDev(c->y2); freeDev(c->z2); freeDev(c->cell2); freeDev(c->gid2);
    c->cap = c->n = 0; c->located = false;
}

template <typename Label>
int setMeshImpl(cpf_context* ctx, const double* points, int64_t nPoints, const Label* faceOffsets,
                const Label* faceVerts, int64_t nFaces, const Label* owner, const Label* neighbour,
                int64_t nInternal, int64_t nCells) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, points && faceOffsets && faceVerts && owner && (neighbour || nInternal == 0), CPF_ERR_ARG,
                "cpf_set_mesh: null array");
    cpf::HostTables t;
    std::string why = cpf::build_tables<Label>(points, nPoints, faceOffsets, faceVerts, nFaces, owner, neighbour,
                                               nInternal, nCells, t);
    if (!why.empty()) return fail(ctx, CPF_ERR_MESH, "cpf_set_mesh: " + why);
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    CPF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    freeMesh(ctx);
    ctx->streamState.flatField = false;
    ctx->host = std::move(t);
    const cpf::HostTables& h = ctx->host;
    auto up = [&](auto*& dptr, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc((void**)&dptr, std::max<size_t>(bytes, 16));
        if (e != hipSuccess) return e;
        ctx->meshBytes += bytes;
        return hipMemcpy(dptr, src, bytes, hipMemcpyHostToDevice);
    };
    CPF_HIP(ctx, up(ctx->d_cellOff, h.cellOff.data(), h.cellOff.size() * 4));
    CPF_HIP(ctx, up(ctx->d_planes, h.planes.data(), h.planes.size() * 8));
    CPF_HIP(ctx, up(ctx->d_nbr, h.nbr.data(), h.nbr.size() * 4));
    CPF_HIP(ctx, up(ctx->d_groupOff, h.groupOff.data(), h.groupOff.size() * 4));
    CPF_HIP(ctx, up(ctx->d_groupNbr, h.groupNbr.data(), h.groupNbr.size() * 4));
    CPF_HIP(ctx, up(ctx->d_binOff, h.binOff.data(), h.binOff.size() * 4));
    CPF_HIP(ctx, up(ctx->d_binCells, h.binCells.data(), h.binCells.size() * 4));
    CPF_HIP(ctx, up(ctx->d_cellBox, h.cellBox.data(), h.cellBox.size() * 4));
    CPF_HIP(ctx, up(ctx->d_curveRank, h.curveRank.data(), h.curveRank.size() * 4));
    CPF_HIP(ctx, hipMalloc((void**)&ctx->d_U, (size_t)nCells * sizeof(double4)));
    CPF_HIP(ctx, hipMalloc((void**)&ctx->d_U3, (size_t)nCells * 3 * sizeof(double)));
    CPF_HIP(ctx, hipMemset(ctx->d_U, 0, (size_t)nCells * sizeof(double4)));
    if (ctx->host.minCellFaces == 6 && ctx->host.maxCellFaces == 6 && ctx->host.nGroups() == 0) {
        CPF_HIP(ctx, hipMalloc((void**)&ctx->d_cellRec, (size_t)nCells * 8 * sizeof(double4)));
        CPF_HIP(ctx, cpf::launch_build_cell_records(ctx->stream, ctx->d_planes, ctx->d_nbr, ctx->d_U, ctx->d_cellRec, nCells));
        CPF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->meshBytes += (size_t)nCells * 256;
        if (!h.boxRec.empty()) CPF_HIP(ctx, up(ctx->d_boxRec, h.boxRec.data(), h.boxRec.size() * 8));
    } else if (ctx->host.nHugeCells * 4 <= nCells && ctx->mixedRecords) {
        // not all-hex, but at most a quarter of the cells have more than TWELVE slots: records for the streaming kernel --
        // padded where a cell has fewer than six slots, a second record for slots 6..11 of a cell with 7..12 (true polyhedra
        // keep the LDS face test: two rounds per visit), a header record + CSR walk only beyond that (cpf_walk.h "cell
        // records").  Refinement interfaces need none of it: their split faces are face groups, one slot each.
        std::vector<int32_t> recB((size_t)nCells, -1);
        int64_t nSecond = 0;
        for (int64_t c = 0; c < nCells; ++c) {
            const int nf = ctx->host.cellOff[(size_t)c + 1] - ctx->host.cellOff[(size_t)c];
            if (nf > 6 && nf <= 12) recB[(size_t)c] = (int32_t)(nCells + nSecond++);
        }
        CPF_REQUIRE(ctx, nCells + nSecond < ((int64_t)1 << 31), CPF_ERR_MESH, "cpf_set_mesh: too many cell records");
        int32_t* d_recB = nullptr;
        hipError_t e = hipSuccess;
        if (nSecond > 0) e = up(d_recB, recB.data(), recB.size() * 4);          // (freed below on every path)
        if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_cellRec, (size_t)(nCells + nSecond) * 8 * sizeof(double4));
        if (e == hipSuccess) e = cpf::launch_build_cell_records_mixed(ctx->stream, ctx->d_cellOff, ctx->d_planes, ctx->d_nbr, ctx->d_U, d_recB,
                                                                      ctx->d_cellRec, nCells);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        freeDev(d_recB);
        CPF_HIP(ctx, e);
        ctx->nSecondRecords = nSecond;
        ctx->meshBytes += (size_t)(nCells + nSecond) * 256;
        // every cell an axis-aligned box although the mesh has face groups (2:1-refined boxes): box records too
        if (!h.boxRec.empty() && nSecond == 0) CPF_HIP(ctx, up(ctx->d_boxRec, h.boxRec.data(), h.boxRec.size() * 8));
    }
    ctx->meshBytes += (size_t)nCells * (sizeof(double4) + 24);
    ctx->haveMesh = true;
    ctx->located = false;
    if (ctx->h_occupied) ctx->h_occupied[0] = ctx->h_occupied[1] = 0;      // (what the last sort counted belonged to the old mesh)
    return CPF_OK;
}

}  // namespace

namespace {
WriterRegistry::~WriterRegistry() {
    for (cpf_context* c : live)
        if (c->writerLive && c->writer.joinable()) { c->writer.join(); c->writerLive = false; }
}
}  // namespace

namespace cpf {
bool vtu_binary(const cpf_context* ctx) { return ctx && ctx->vtuBinary; }
void* context_stream(const cpf_context* ctx) { return (void*)ctx->stream; }
int context_device(const cpf_context* ctx) { return ctx->device; }
bool context_timing(const cpf_context* ctx) { return ctx->timing; }
int64_t context_cells(const cpf_context* ctx) { return ctx->haveMesh ? ctx->host.nCells : 0; }
}  // namespace cpf

namespace {
// Is the tet decomposition of every cell a FAN about one apex that covers every direction exactly once?  (What the reference's
// fragment builds: apex = the cell centre, one tet per face triangle, src/initCuda.H:99-105.)  Then no two tets of a cell
// overlap, which is what makes the cone locate of the "VertexVelocity" advect exact (cpf_kernels.hip).  Checked per cell: every
// tet starts at the same vertex; all determinants have one sign and none is flat beyond a condition of 1e5 (weights then carry
// rounding errors below 1e-10, two orders inside the locate's margin); the base triangles form a closed oriented surface (every
// directed edge once, its reverse once); the solid angles the tets subtend at the apex (Van Oosterom-Strackee) add up to
// 4 pi -- a closed surface seen from its inner side everywhere that winds round the apex once projects one-to-one onto the
// sphere of directions, i.e. the cones do not overlap.
// Returns "" or the first defect.
std::string tetFanDefect(const double* pos, const int32_t* tets, int64_t nCells, int tetsPerCell) {
    auto P = [&](int32_t v, int k) { return pos[3 * (int64_t)v + k]; };
    const double fourPi = 12.566370614359172;
    std::vector<uint64_t> edges;
    for (int64_t c = 0; c < nCells; ++c) {
        const int32_t* t0 = tets + 4 * c * tetsPerCell;
        const int32_t apex = t0[0];
        double omega = 0.0;
        int sign = 0;
        for (int k = 0; k < tetsPerCell; ++k) {
            const int32_t* ix = t0 + 4 * k;
            if (ix[0] != apex) return "cell " + std::to_string(c) + ": its tets do not share their first vertex";
            double e[3][3], len[3];
            for (int j = 0; j < 3; ++j) {
                for (int q = 0; q < 3; ++q) e[j][q] = P(ix[j + 1], q) - P(apex, q);
                len[j] = std::sqrt(e[j][0] * e[j][0] + e[j][1] * e[j][1] + e[j][2] * e[j][2]);
            }
            const double det = e[0][0] * (e[1][1] * e[2][2] - e[1][2] * e[2][1]) - e[0][1] * (e[1][0] * e[2][2] - e[1][2] * e[2][0]) +
                               e[0][2] * (e[1][0] * e[2][1] - e[1][1] * e[2][0]);
            const double scale = len[0] * len[1] * len[2];
            if (!(std::fabs(det) * 1e5 > scale)) return "cell " + std::to_string(c) + ": a flat (or badly conditioned) tet";
            const int sg = det > 0 ? 1 : -1;
            if (sign == 0) sign = sg;
            else if (sg != sign) return "cell " + std::to_string(c) + ": tets of both orientations";
            auto dot = [&](int i, int j) { return e[i][0] * e[j][0] + e[i][1] * e[j][1] + e[i][2] * e[j][2]; };
            omega += 2.0 * std::atan2(std::fabs(det), scale + dot(0, 1) * len[2] + dot(0, 2) * len[1] + dot(1, 2) * len[0]);
        }
        if (std::fabs(omega - fourPi) > 1e-6) return "cell " + std::to_string(c) + ": the tets' solid angles at the apex do not add up to 4 pi";
        // the base triangles form a closed oriented surface: every directed edge once, its reverse once (a tet listed twice in
        // place of its mirror image keeps the angles' sum and is caught here)
        edges.clear();
        for (int k = 0; k < tetsPerCell; ++k) {
            const int32_t* ix = t0 + 4 * k;
            for (int j = 0; j < 3; ++j) edges.push_back(((uint64_t)(uint32_t)ix[1 + j] << 32) | (uint32_t)ix[1 + (j + 1) % 3]);
        }
        std::sort(edges.begin(), edges.end());
        for (size_t i = 0; i < edges.size(); ++i) {
            const uint64_t rev = (edges[i] << 32) | (edges[i] >> 32);
            if ((i > 0 && edges[i] == edges[i - 1]) || !std::binary_search(edges.begin(), edges.end(), rev))
                return "cell " + std::to_string(c) + ": the tets' base triangles do not form a closed surface";
        }
    }
    return "";
}

// U[nCells][3] (device) -> the padded field and the cell records; on a mesh that qualifies for the flat walk (cpf_walk.h) the
// kernel also notes whether any cell has a z component, and the note is read back behind it (8 bytes, asynchronous)
hipError_t layOutField(cpf_context* ctx, const double* dU3, int64_t nCells) {
    const bool ask = ctx->host.zSide0 && ctx->d_occupied && ctx->h_occupied;
    ctx->streamState.flatField = false;
    ctx->fieldFlagPending = false;
    hipError_t e = hipSuccess;
    if (ask) e = hipMemsetAsync(ctx->d_occupied + 2, 0, 8, ctx->stream);
    if (e == hipSuccess) e = cpf::launch_u3_to_u4(ctx->stream, dU3, ctx->d_U, nCells, ask ? ctx->d_occupied + 2 : nullptr);
    if (e == hipSuccess && ctx->d_cellRec) e = cpf::launch_update_record_velocity(ctx->stream, ctx->d_U, ctx->d_cellRec, ctx->d_boxRec, nCells);
    if (e == hipSuccess && ask) {
        e = hipMemcpyAsync(ctx->h_occupied + 2, ctx->d_occupied + 2, 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(ctx->evFieldFlag, ctx->stream);
        ctx->fieldFlagPending = e == hipSuccess;
    }
    return e;
}
// the read-back is known to be complete (after a synchronise, or its event has been seen): take the note
void fieldFlagArrived(cpf_context* ctx) {
    if (!ctx->fieldFlagPending) return;
    ctx->streamState.flatField = ctx->h_occupied[2] == 0;
    ctx->fieldFlagPending = false;
}
void pollFieldFlag(cpf_context* ctx) {
    if (ctx->fieldFlagPending && hipEventQuery(ctx->evFieldFlag) == hipSuccess) fieldFlagArrived(ctx);
}
}  // namespace

extern "C" {

int cpf_abi_version(void) { return CPF_ABI_VERSION; }

int cpf_create(int device, cpf_context** out) {
    if (!out) return fail(nullptr, CPF_ERR_ARG, "cpf_create: out is null");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, CPF_ERR_HIP,
                    std::string("cpf_create: no HIP device available (") + hipGetErrorString(e) + ")");
    if (device < 0 || device >= count) return fail(nullptr, CPF_ERR_ARG, "cpf_create: device index out of range");
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(nullptr, CPF_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    cpf_context* ctx = new (std::nothrow) cpf_context();
    if (!ctx) return fail(nullptr, CPF_ERR_NOMEM, "cpf_create: out of host memory");
    ctx->device = device;
    e = hipStreamCreateWithFlags(&ctx->ownStream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_counters, (cpf::kCounterSlots * 4 + 4) * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(ctx->d_counters, 0, (cpf::kCounterSlots * 4 + 4) * sizeof(unsigned long long));
    // ([2]: "the velocity field has a z component", written by the kernel that lays the field out -- the flat walk)
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_occupied, 32);
    if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->h_occupied, 32, hipHostMallocDefault);
    if (e == hipSuccess) { ctx->h_occupied[0] = ctx->h_occupied[1] = 0; ctx->h_occupied[2] = 1; ctx->streamState.occupiedHost = ctx->h_occupied; }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->evFieldFlag, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->streamState.d_grab, cpf::kStreamGrabBytes);
    if (e == hipSuccess) e = hipMemset(ctx->streamState.d_grab, 0, cpf::kStreamGrabBytes);
    if (e == hipSuccess) {
        int cus = 0;
        e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
        if (e == hipSuccess && cus > 0) ctx->streamState.numCU = cus;
    }
    if (e == hipSuccess) {
        // overflow area of the streaming kernel's per-wave pool of wall hit points: 1.5 KB per wave slot the chip can hold
        ctx->streamState.hitSpillWaves = 32 * ctx->streamState.numCU;
        e = hipMalloc((void**)&ctx->streamState.d_hitSpill,
                      (size_t)ctx->streamState.hitSpillWaves * cpf::kStreamHitSpillDoubles * sizeof(double));
    }
    if (e != hipSuccess) {
        std::string m = std::string("cpf_create: ") + hipGetErrorString(e);
        if (ctx->ownStream) (void)hipStreamDestroy(ctx->ownStream);
        delete ctx;
        return fail(nullptr, CPF_ERR_HIP, m);
    }
    ctx->stream = ctx->ownStream;
    *out = ctx;
    return CPF_OK;
}

int cpf_destroy(cpf_context* ctx) {
    if (!ctx) return CPF_OK;
    (void)cpf_write_vtu_wait(ctx);                                             // never lose a frame
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    freeMesh(ctx); freeCloud(ctx);
    freeDev(ctx->scratch); freeDev(ctx->d_counters); freeDev(ctx->streamState.d_grab); freeDev(ctx->streamState.d_hitSpill);
    freeDev(ctx->d_occupied);
    if (ctx->evFieldFlag) { (void)hipEventDestroy(ctx->evFieldFlag); ctx->evFieldFlag = nullptr; }
    if (ctx->h_occupied) { (void)hipHostFree(ctx->h_occupied); ctx->h_occupied = nullptr; ctx->streamState.occupiedHost = nullptr; }
    freeDev(ctx->d_tetPos); freeDev(ctx->d_tets); freeDev(ctx->d_vertVel); freeDev(ctx->d_vertCone); freeDev(ctx->d_vertApex);
    freeDev(ctx->snapDev);
    if (ctx->snapHost) { (void)hipHostFree(ctx->snapHost); ctx->snapHost = nullptr; }
    if (ctx->evSnap) (void)hipEventDestroy(ctx->evSnap);
    if (ctx->evCopied) (void)hipEventDestroy(ctx->evCopied);
    if (ctx->ioStream) (void)hipStreamDestroy(ctx->ioStream);
    for (auto& p : ctx->events) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto& ev : ctx->eventPool) (void)hipEventDestroy(ev);
    if (ctx->ownStream) (void)hipStreamDestroy(ctx->ownStream);
    delete ctx;
    return CPF_OK;
}

const char* cpf_last_error(const cpf_context* ctx) {
    if (ctx) return ctx->err.c_str();
    std::lock_guard<std::mutex> lk(g_mutex);
    return g_createError.c_str();
}

int cpf_set_stream(cpf_context* ctx, void* hip_stream) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    fieldFlagArrived(ctx);                      // (a field note still on its way was recorded on the old stream: complete now)
    ctx->stream = (hipStream_t)hip_stream;      // NULL == HIP's default stream, a valid choice
    return CPF_OK;
}

int cpf_use_own_stream(cpf_context* ctx) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    fieldFlagArrived(ctx);
    ctx->stream = ctx->ownStream;
    return CPF_OK;
}

int cpf_synchronize(cpf_context* ctx) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CPF_OK;
}

int cpf_set_mesh(cpf_context* ctx, const double* points, int64_t nPoints, const int32_t* faceOffsets,
                 const int32_t* faceVerts, int64_t nFaces, const int32_t* owner, const int32_t* neighbour,
                 int64_t nInternal, int64_t nCells) {
    return setMeshImpl<int32_t>(ctx, points, nPoints, faceOffsets, faceVerts, nFaces, owner, neighbour, nInternal,
                                nCells);
}
int cpf_set_mesh_l64(cpf_context* ctx, const double* points, int64_t nPoints, const int64_t* faceOffsets,
                     const int64_t* faceVerts, int64_t nFaces, const int64_t* owner, const int64_t* neighbour,
                     int64_t nInternal, int64_t nCells) {
    return setMeshImpl<int64_t>(ctx, points, nPoints, faceOffsets, faceVerts, nFaces, owner, neighbour, nInternal,
                                nCells);
}

int cpf_build_mesh_tables_host(const double* points, int64_t nPoints, const int32_t* faceOffsets, const int32_t* faceVerts,
                               int64_t nFaces, const int32_t* owner, const int32_t* neighbour, int64_t nInternal, int64_t nCells,
                               int64_t* nSlots, int64_t* nGroups, int64_t* nMembers, int32_t* cellOff, double* planes,
                               int32_t* nbr, int32_t* groupOff, int32_t* groupNbr) {
    if (!points || !faceOffsets || !faceVerts || !owner || (!neighbour && nInternal != 0)) return CPF_ERR_ARG;
    cpf::HostTables t;
    try {
        const std::string why = cpf::build_tables<int32_t>(points, nPoints, faceOffsets, faceVerts, nFaces, owner, neighbour, nInternal, nCells, t);
        if (!why.empty()) return CPF_ERR_MESH;
    } catch (const std::bad_alloc&) {
        return CPF_ERR_NOMEM;
    }
    const int64_t g = t.nGroups(), mem = t.groupOff[(size_t)g];
    if (nSlots) *nSlots = t.nSlots;
    if (nGroups) *nGroups = g;
    if (nMembers) *nMembers = mem;
    if (cellOff) std::memcpy(cellOff, t.cellOff.data(), t.cellOff.size() * 4);
    if (planes) std::memcpy(planes, t.planes.data(), t.planes.size() * 8);
    if (nbr) std::memcpy(nbr, t.nbr.data(), t.nbr.size() * 4);
    if (groupOff) std::memcpy(groupOff, t.groupOff.data(), (size_t)(g + 1) * 4);
    if (groupNbr) std::memcpy(groupNbr, t.groupNbr.data(), (size_t)mem * 4);
    return CPF_OK;
}

int cpf_mesh_flags_host(const double* points, int64_t nPoints, const int32_t* faceOffsets, const int32_t* faceVerts,
                        int64_t nFaces, const int32_t* owner, const int32_t* neighbour, int64_t nInternal, int64_t nCells,
                        int32_t* allHex, int32_t* zLayered, int32_t* zThin, int32_t* mixed) {
    if (!points || !faceOffsets || !faceVerts || !owner || (!neighbour && nInternal != 0)) return CPF_ERR_ARG;
    cpf::HostTables t;
    try {
        const std::string why = cpf::build_tables<int32_t>(points, nPoints, faceOffsets, faceVerts, nFaces, owner, neighbour, nInternal, nCells, t);
        if (!why.empty()) return CPF_ERR_MESH;
    } catch (const std::bad_alloc&) {
        return CPF_ERR_NOMEM;
    }
    // (as meshView() derives them from a context's tables, with the default options)
    const bool hex = t.minCellFaces == 6 && t.maxCellFaces == 6 && t.nGroups() == 0;
    if (allHex) *allHex = hex ? 1 : 0;
    if (zLayered) *zLayered = t.zPairLast ? 1 : 0;
    if (zThin) *zThin = t.zThin ? 1 : 0;
    if (mixed) *mixed = hex ? 0 : (t.nHugeCells * 4 <= nCells ? (t.nBigCells > 0 ? 2 : 1) : 0);
    return CPF_OK;
}

int cpf_mesh_box_records_host(const double* points, int64_t nPoints, const int32_t* faceOffsets, const int32_t* faceVerts,
                              int64_t nFaces, const int32_t* owner, const int32_t* neighbour, int64_t nInternal, int64_t nCells,
                              int32_t* isBox, double* boxRec) {
    if (!points || !faceOffsets || !faceVerts || !owner || (!neighbour && nInternal != 0)) return CPF_ERR_ARG;
    cpf::HostTables t;
    try {
        const std::string why = cpf::build_tables<int32_t>(points, nPoints, faceOffsets, faceVerts, nFaces, owner, neighbour, nInternal, nCells, t);
        if (!why.empty()) return CPF_ERR_MESH;
    } catch (const std::bad_alloc&) {
        return CPF_ERR_NOMEM;
    }
    if (isBox) *isBox = t.boxRec.empty() ? 0 : 1;
    if (boxRec && !t.boxRec.empty()) std::memcpy(boxRec, t.boxRec.data(), t.boxRec.size() * 8);
    return CPF_OK;
}

int cpf_mesh_info(const cpf_context* ctx, int64_t* nCells, int64_t* nSlots, int64_t* deviceBytes) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, ctx->haveMesh, CPF_ERR_STATE, "cpf_mesh_info: no mesh set");
    if (nCells) *nCells = ctx->host.nCells;
    if (nSlots) *nSlots = ctx->host.nSlots;
    if (deviceBytes) *deviceBytes = (int64_t)ctx->meshBytes;
    return CPF_OK;
}

int cpf_get_mesh_tables(const cpf_context* ctx, int32_t* cellOff, double* planes, int32_t* nbr) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, ctx->haveMesh, CPF_ERR_STATE, "cpf_get_mesh_tables: no mesh set");
    const cpf::HostTables& h = ctx->host;
    if (cellOff) std::memcpy(cellOff, h.cellOff.data(), h.cellOff.size() * 4);
    if (planes) std::memcpy(planes, h.planes.data(), h.planes.size() * 8);
    if (nbr) std::memcpy(nbr, h.nbr.data(), h.nbr.size() * 4);
    return CPF_OK;
}

int cpf_get_mesh_groups(const cpf_context* ctx, int64_t* nGroups, int64_t* nMembers, int32_t* groupOff, int32_t* groupNbr) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, ctx->haveMesh, CPF_ERR_STATE, "cpf_get_mesh_groups: no mesh set");
    const cpf::HostTables& h = ctx->host;
    const int64_t g = h.nGroups();
    if (nGroups) *nGroups = g;
    if (nMembers) *nMembers = h.groupOff[(size_t)g];
    if (groupOff) std::memcpy(groupOff, h.groupOff.data(), (size_t)(g + 1) * 4);
    if (groupNbr) std::memcpy(groupNbr, h.groupNbr.data(), (size_t)h.groupOff[(size_t)g] * 4);
    return CPF_OK;
}

int cpf_get_mesh_flags(const cpf_context* ctx, int32_t* allHex, int32_t* zLayered, int32_t* zThin, int32_t* mixed) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, ctx->haveMesh, CPF_ERR_STATE, "cpf_get_mesh_flags: no mesh set");
    const cpf::MeshView m = meshView(ctx);
    if (allHex) *allHex = m.allHex;
    if (zLayered) *zLayered = m.zPairLast;
    if (zThin) *zThin = m.zThin;
    if (mixed) *mixed = m.mixed;
    return CPF_OK;
}

int cpf_set_velocity(cpf_context* ctx, const double* U, int64_t nCells) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, ctx->haveMesh, CPF_ERR_STATE, "cpf_set_velocity: call cpf_set_mesh first");
    CPF_REQUIRE(ctx, U && nCells == ctx->host.nCells, CPF_ERR_ARG, "cpf_set_velocity: U is null or nCells differs from the mesh");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    CPF_HIP(ctx, hipMemcpyAsync(ctx->d_U3, U, (size_t)nCells * 24, hipMemcpyHostToDevice, ctx->stream));
    CPF_HIP(ctx, layOutField(ctx, ctx->d_U3, nCells));
    CPF_HIP(ctx, hipStreamSynchronize(ctx->stream));   // U may be pageable host memory owned by the caller
    fieldFlagArrived(ctx);
    ctx->haveU = true;
    return CPF_OK;
}

int cpf_set_velocity_dev(cpf_context* ctx, const double* dU, int64_t nCells) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, ctx->haveMesh, CPF_ERR_STATE, "cpf_set_velocity_dev: call cpf_set_mesh first");
    CPF_REQUIRE(ctx, dU && nCells == ctx->host.nCells, CPF_ERR_ARG, "cpf_set_velocity_dev: bad arguments");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    CPF_HIP(ctx, layOutField(ctx, dU, nCells));        // (asynchronous: the flat walk waits until the flag has been seen to arrive)
    ctx->haveU = true;
    return CPF_OK;
}

int cpf_alloc_particles(cpf_context* ctx, int64_t capacity) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, capacity > 0 && capacity < ((int64_t)1 << 31), CPF_ERR_ARG, "cpf_alloc_particles: capacity must be in (0, 2^31)");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    CPF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    freeCloud(ctx);
    const size_t c = (size_t)capacity;
    CPF_HIP(ctx, hipMalloc((void**)&ctx->x, c * 8));
    CPF_HIP(ctx, hipMalloc((void**)&ctx->y, c * 8));
    CPF_HIP(ctx, hipMalloc((void**)&ctx->z, c * 8));
    CPF_HIP(ctx, hipMalloc((void**)&ctx->vel, c * 24));
    CPF_HIP(ctx, hipMalloc((void**)&ctx->cell, c * 4));
    CPF_HIP(ctx, hipMalloc((void**)&ctx->gid, c * 8));
    CPF_HIP(ctx, hipMemsetAsync(ctx->vel, 0, c * 24, ctx->stream));          // src/initCuda.H:148-149
    CPF_HIP(ctx, hipMemsetAsync(ctx->cell, 0xFF, c * 4, ctx->stream));       // -1, src/initCuda.H:145
    ctx->cap = capacity;
    return CPF_OK;
}

int cpf_seed_box(cpf_context* ctx, int64_t n, const double lower[3], const double upper[3], int order) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, lower && upper && n > 0, CPF_ERR_ARG, "cpf_seed_box: bad arguments");
    if (ctx->cap < n) { int r = cpf_alloc_particles(ctx, n); if (r) return r; }
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    CPF_HIP(ctx, cpf::launch_seed_box(ctx->stream, ctx->x, ctx->y, ctx->z, 0, n, lower, upper, order));
    CPF_HIP(ctx, cpf::launch_iota64(ctx->stream, ctx->gid, n, 0));
    CPF_HIP(ctx, hipMemsetAsync(ctx->cell, 0xFF, (size_t)n * 4, ctx->stream));
    ctx->n = n; ctx->located = false;
    return CPF_OK;
}

int cpf_set_particles(cpf_context* ctx, int64_t n, const double* xyz, const int32_t* cell) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, xyz && n > 0, CPF_ERR_ARG, "cpf_set_particles: bad arguments");
    if (ctx->cap < n) { int r = cpf_alloc_particles(ctx, n); if (r) return r; }
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    int r = ensureScratch(ctx, (size_t)n * 24);
    if (r) return r;
    CPF_HIP(ctx, hipMemcpyAsync(ctx->scratch, xyz, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
    CPF_HIP(ctx, cpf::launch_unpack_xyz(ctx->stream, (const double*)ctx->scratch, ctx->x, ctx->y, ctx->z, n));
    CPF_HIP(ctx, cpf::launch_iota64(ctx->stream, ctx->gid, n, 0));
    if (cell) CPF_HIP(ctx, hipMemcpyAsync(ctx->cell, cell, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    else CPF_HIP(ctx, hipMemsetAsync(ctx->cell, 0xFF, (size_t)n * 4, ctx->stream));
    CPF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n = n; ctx->located = cell != nullptr;
    return CPF_OK;
}

int cpf_locate_initial_dev(cpf_context* ctx, const double* x, const double* y, const double* z, int32_t* cell,
                           int64_t n) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, ctx->haveMesh, CPF_ERR_STATE, "cpf_locate_initial: call cpf_set_mesh first");
    CPF_REQUIRE(ctx, n >= 0 && (n == 0 || (x && y && z && cell)), CPF_ERR_ARG, "cpf_locate_initial: null array");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    CPF_HIP(ctx, cpf::launch_locate_initial(ctx->stream, x, y, z, cell, n, meshView(ctx), gridView(ctx)));
    return CPF_OK;
}

int cpf_locate_initial(cpf_context* ctx, int64_t* nOutside) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, ctx->n > 0, CPF_ERR_STATE, "cpf_locate_initial: no particles (seed or set them first)");
    int r = cpf_locate_initial_dev(ctx, ctx->x, ctx->y, ctx->z, ctx->cell, ctx->n);
    if (r) return r;
    ctx->located = true;
    if (nOutside) {
        unsigned long long* cnt = ctx->d_counters + cpf::kCounterSlots * 4;
        CPF_HIP(ctx, hipMemsetAsync(cnt, 0, 8, ctx->stream));
        CPF_HIP(ctx, cpf::launch_count_negative(ctx->stream, ctx->cell, ctx->n, cnt));
        unsigned long long h = 0;
        CPF_HIP(ctx, hipMemcpyAsync(&h, cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
        CPF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        *nOutside = (int64_t)h;
    }
    return CPF_OK;
}

int cpf_step_dev(cpf_context* ctx, double* x, double* y, double* z, int32_t* cell, const int64_t* gid, double* vel,
                 int64_t n, double dt, double D, uint32_t step0, int nCycles, unsigned flags) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, ctx->haveMesh, CPF_ERR_STATE, "cpf_step: call cpf_set_mesh first");
    CPF_REQUIRE(ctx, ctx->haveU, CPF_ERR_STATE, "cpf_step: call cpf_set_velocity first");
    CPF_REQUIRE(ctx, n >= 0 && nCycles >= 0, CPF_ERR_ARG, "cpf_step: negative count");
    CPF_REQUIRE(ctx, n == 0 || (x && y && z && cell), CPF_ERR_ARG, "cpf_step: null particle array");
    CPF_REQUIRE(ctx, std::isfinite(dt) && std::isfinite(D) && D >= 0.0, CPF_ERR_ARG, "cpf_step: dt/D not finite or D < 0");
    const bool storeVel = (flags & CPF_STEP_STORE_VEL) != 0;
    CPF_REQUIRE(ctx, !storeVel || vel, CPF_ERR_ARG, "cpf_step: CPF_STEP_STORE_VEL needs a vel array");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    const bool reflect = (flags & CPF_STEP_NO_REFLECT) == 0;
    const bool vertexU = (flags & CPF_STEP_VERTEX_VELOCITY) != 0;
    CPF_REQUIRE(ctx, !vertexU || (ctx->haveVertVel && ctx->nTets == (int64_t)ctx->tetsPerCell * ctx->host.nCells), CPF_ERR_STATE,
                "cpf_step: CPF_STEP_VERTEX_VELOCITY needs cpf_set_tets and cpf_set_vertex_velocity for the current mesh");
    const cpf::MeshView m = meshView(ctx);
    const bool fuse = (flags & CPF_STEP_FUSE_CYCLES) != 0;
    pollFieldFlag(ctx);
    const int nLaunch = fuse ? 1 : nCycles;   // fused with nCycles == 0: load+store only (bandwidth calibration)
    const int cycPerLaunch = fuse ? nCycles : 1;
    for (int c = 0; c < nLaunch; ++c) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        bool stamped = false;
        const bool timed = ctx->timing && (ctx->timingLaunch++ % (uint64_t)ctx->timingStride) == 0;
        if (timed) {
            auto take = [&](hipEvent_t& ev) -> hipError_t {
                if (!ctx->eventPool.empty()) { ev = ctx->eventPool.back(); ctx->eventPool.pop_back(); return hipSuccess; }
                // (device-scope release is all a time stamp needs; measured against the default flags: no difference)
#ifndef CPF_TIMING_EVENT_FLAGS
#define CPF_TIMING_EVENT_FLAGS hipEventReleaseToDevice
#endif
                return hipEventCreateWithFlags(&ev, CPF_TIMING_EVENT_FLAGS);
            };
            CPF_HIP(ctx, take(e0)); CPF_HIP(ctx, take(e1));
            // the streaming launcher stamps the events with the dispatch's own begin / end (cpf_device.h, StreamState);
            // any other kernel is bracketed by two event records
            stamped = vertexU ? cpf::step_vertex_streams(m, ctx->vertexFast ? ctx->d_vertCone : nullptr, ctx->tetsPerCell, ctx->stepVariant, &ctx->streamState, cycPerLaunch)
                              : cpf::effective_step_variant(ctx->stepVariant, m, true, cycPerLaunch, ctx->streamState.coopMaxCells) == 4;
            if (stamped) { ctx->streamState.evStart = e0; ctx->streamState.evStop = e1; }
            else CPF_HIP(ctx, hipEventRecord(e0, ctx->stream));
        }
        ctx->lastStepN = n; ctx->lastStepCycles = cycPerLaunch;
        const hipError_t le = vertexU
            ? cpf::launch_step_vertex(ctx->stream, x, y, z, cell, gid, vel, n, dt, D, step0 + (uint32_t)c, cycPerLaunch, ctx->seed, reflect,
                                      storeVel, m, ctx->stats ? ctx->d_counters : nullptr, ctx->d_tetPos, ctx->d_tets, ctx->tetsPerCell,
                                      ctx->d_vertVel, ctx->vertexFast ? ctx->d_vertCone : nullptr, ctx->d_vertApex, ctx->stepVariant, &ctx->streamState)
            : cpf::launch_step(ctx->stream, x, y, z, cell, gid, vel, n, dt, D, step0 + (uint32_t)c, cycPerLaunch, ctx->seed,
                               reflect, storeVel, m, ctx->stats ? ctx->d_counters : nullptr, ctx->stepVariant,
                               &ctx->streamState);
        if (le != hipSuccess) {
            // a launch that did not go out (occupancy query, tile count, missing spill area): its time stamps must not be
            // left for the next, unrelated launch to take, and the two events go back to the pool instead of leaking
            ctx->streamState.evStart = ctx->streamState.evStop = nullptr;
            if (e0) ctx->eventPool.push_back(e0);
            if (e1) ctx->eventPool.push_back(e1);
        }
        CPF_HIP(ctx, le);
        if (timed) {
            if (!stamped) CPF_HIP(ctx, hipEventRecord(e1, ctx->stream));
            else if (ctx->streamState.evStart != nullptr) {  // (cannot happen: the streaming launcher always takes them)
                ctx->streamState.evStart = ctx->streamState.evStop = nullptr;
                CPF_HIP(ctx, hipEventRecord(e0, ctx->stream)); CPF_HIP(ctx, hipEventRecord(e1, ctx->stream));
            }
            ctx->events.emplace_back(e0, e1);
        }
    }
    return CPF_OK;
}

int cpf_step(cpf_context* ctx, double dt, double D, int nCycles, unsigned flags) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, ctx->n > 0, CPF_ERR_STATE, "cpf_step: no particles");
    CPF_REQUIRE(ctx, ctx->located, CPF_ERR_STATE, "cpf_step: particles have no cells yet (call cpf_locate_initial)");
    // a frame holds the velocities of the particles this call steps; one that is not stepped -- lost, frozen -- has none (the
    // reference's array keeps the velocity of such a particle's last advect, cuda/particles.cu:316-373; with the velocities stored
    // on frame cycles only that would be the one of its last FRAME, and a sharded cloud does not carry it along at all)
    if ((flags & CPF_STEP_STORE_VEL) && ctx->vel) {
        CPF_HIP(ctx, hipSetDevice(ctx->device));
        CPF_HIP(ctx, hipMemsetAsync(ctx->vel, 0, (size_t)ctx->n * 24, ctx->stream));
    }
    int r = cpf_step_dev(ctx, ctx->x, ctx->y, ctx->z, ctx->cell, ctx->gid, ctx->vel, ctx->n, dt, D, ctx->stepCounter,
                         nCycles, flags);
    if (r != CPF_OK) return r;
    // a cycle of zero length without a kick moves nothing: it is the frame-0 idiom (velocities of one advect in the
    // first output file, out-of-domain particles frozen; src/initCuda.H:184-201) and not a step of the run, so the
    // counter-based Brownian stream and the sort cadence do not see it
    if (!(dt == 0.0 && D == 0.0)) ctx->stepCounter += (uint32_t)nCycles;
    // keep waves cell-coherent: particle ids (and stored velocities) travel with the particles, so callers
    // never see the reordering
    if (ctx->sortInterval > 0 && ctx->stepCounter - ctx->lastSortStep >= (uint32_t)ctx->sortInterval) {
        r = cpf_sort_by_cell(ctx);
        ctx->lastSortStep = ctx->stepCounter;
    }
    return r;
}

int cpf_seed_box_dev(cpf_context* ctx, double* x, double* y, double* z, int64_t first, int64_t n,
                     const double lower[3], const double upper[3], int order) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, n >= 0 && first >= 0 && lower && upper && (n == 0 || (x && y && z)), CPF_ERR_ARG, "cpf_seed_box_dev: bad arguments");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    CPF_HIP(ctx, cpf::launch_seed_box(ctx->stream, x, y, z, first, n, lower, upper, order));
    return CPF_OK;
}

namespace {
// in place (ox == nullptr) or into the out arrays
int sortImpl(cpf_context* ctx, double* x, double* y, double* z, int32_t* cell, int64_t* gid, int64_t n, double* ox, double* oy,
             double* oz, int32_t* ocell, int64_t* ogid, double* vel3) {
    CPF_REQUIRE(ctx, ctx->haveMesh, CPF_ERR_STATE, "cpf_sort_by_cell: call cpf_set_mesh first");
    CPF_REQUIRE(ctx, n >= 0 && n < ((int64_t)1 << 31) && (n == 0 || (x && y && z && cell)), CPF_ERR_ARG, "cpf_sort_by_cell: bad arguments");
    CPF_REQUIRE(ctx, ctx->host.nCells < ((int64_t)1 << 26) - 2, CPF_ERR_STATE, "cpf_sort_by_cell: more than 2^26 cells");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    const int endBit = sortEndBit(ctx);
    const bool census = ctx->streamState.densityLookup != 0 && ctx->d_occupied && ctx->h_occupied;
    // sparse clouds (the regime of the streaming kernel's LOOKUP 4: fewer than 8 particles per cell) are ordered along the mesh
    // layer's Morton curve instead of by cell id: measured 0.164 -> 0.153 ms per step at 0.6 particles per cell on the 2.1e6-cell
    // box; dense clouds on a 3-D mesh LOSE 7 % with it (blockMesh's numbering runs along the flow), hence by regime
    const bool curve = ctx->d_curveRank != nullptr && (ctx->sortCurve == 1 || (ctx->sortCurve < 0 && n < 8 * ctx->host.nCells));
    int r = ensureScratch(ctx, cpf::sort_scratch_bytes(n, endBit));
    if (r) return r;
    CPF_HIP(ctx, cpf::sort_by_cell(ctx->stream, x, y, z, cell, gid, vel3, n, endBit, ctx->d_cellBox, ctx->host.subBits,
                                   ctx->host.subOrder, ctx->scratch, ctx->scratchBytes, ox, oy, oz, ocell, ogid, census ? ctx->d_occupied : nullptr,
                                   curve ? ctx->d_curveRank : nullptr, ctx->sortMethod));
    // how many cells hold particles: what the streaming kernel's lookup method goes by with "stream_lookup_by_density"
    // (StreamState::occupiedHost).  Only then: the 16-byte device-to-host copy behind the sort costs 0.9 ms on this stack
    // (measured: 1.50 against 0.60 ms per sort of 1e7 particles) -- more than the sort itself.
    if (census)
        CPF_HIP(ctx, hipMemcpyAsync(ctx->h_occupied, ctx->d_occupied, 16, hipMemcpyDeviceToHost, ctx->stream));
    return CPF_OK;
}
}  // namespace

int cpf_sort_by_cell_dev(cpf_context* ctx, double* x, double* y, double* z, int32_t* cell, int64_t* gid, int64_t n) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    if (n <= 1) return CPF_OK;
    return sortImpl(ctx, x, y, z, cell, gid, n, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}

int cpf_sort_by_cell_dev_to(cpf_context* ctx, const double* x, const double* y, const double* z, const int32_t* cell,
                            const int64_t* gid, double* ox, double* oy, double* oz, int32_t* ocell, int64_t* ogid, int64_t n) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, n == 0 || (ox && oy && oz && ocell && (ogid || !gid)), CPF_ERR_ARG, "cpf_sort_by_cell_dev_to: null output array");
    CPF_REQUIRE(ctx, n == 0 || (ox != x && oy != y && oz != z && ocell != cell), CPF_ERR_ARG, "cpf_sort_by_cell_dev_to: outputs alias inputs");
    if (n <= 0) return CPF_OK;
    if (n == 1) {           // nothing to sort, but the contract is "the cloud is in the out arrays"
        CPF_HIP(ctx, hipSetDevice(ctx->device));
        CPF_HIP(ctx, hipMemcpyAsync(ox, x, 8, hipMemcpyDeviceToDevice, ctx->stream));
        CPF_HIP(ctx, hipMemcpyAsync(oy, y, 8, hipMemcpyDeviceToDevice, ctx->stream));
        CPF_HIP(ctx, hipMemcpyAsync(oz, z, 8, hipMemcpyDeviceToDevice, ctx->stream));
        CPF_HIP(ctx, hipMemcpyAsync(ocell, cell, 4, hipMemcpyDeviceToDevice, ctx->stream));
        if (gid) CPF_HIP(ctx, hipMemcpyAsync(ogid, gid, 8, hipMemcpyDeviceToDevice, ctx->stream));
        return CPF_OK;
    }
    return sortImpl(ctx, const_cast<double*>(x), const_cast<double*>(y), const_cast<double*>(z), const_cast<int32_t*>(cell),
                    const_cast<int64_t*>(gid), n, ox, oy, oz, ocell, ogid, nullptr);
}

int cpf_sort_by_cell(cpf_context* ctx) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, ctx->n > 0 && ctx->located, CPF_ERR_STATE, "cpf_sort_by_cell: no located particles");
    if (ctx->n <= 1) return CPF_OK;
    // the context's own cloud: sorted into its second set of arrays, then the sets swap roles
    if (ctx->x2 == nullptr) {
        CPF_HIP(ctx, hipSetDevice(ctx->device));
        const size_t c = (size_t)ctx->cap;
        // all five or none: a failure part-way must not leave x2 set and the later pointers null (the next call would
        // skip this block and scatter into null pointers)
        void* b[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        const size_t bytes[5] = {c * 8, c * 8, c * 8, c * 4, c * 8};
        hipError_t e = hipSuccess;
        for (int k = 0; k < 5 && e == hipSuccess; ++k) e = hipMalloc(&b[k], bytes[k]);
        if (e == hipSuccess) e = hipMemsetAsync(b[3], 0xFF, bytes[3], ctx->stream);
        if (e != hipSuccess) {
            for (int k = 0; k < 5; ++k) if (b[k]) (void)hipFree(b[k]);
            CPF_HIP(ctx, e);
        }
        ctx->x2 = (double*)b[0]; ctx->y2 = (double*)b[1]; ctx->z2 = (double*)b[2];
        ctx->cell2 = (int32_t*)b[3]; ctx->gid2 = (int64_t*)b[4];
    }
    int r = sortImpl(ctx, ctx->x, ctx->y, ctx->z, ctx->cell, ctx->gid, ctx->n, ctx->x2, ctx->y2, ctx->z2, ctx->cell2, ctx->gid2,
                     ctx->vel);
    if (r) return r;
    std::swap(ctx->x, ctx->x2); std::swap(ctx->y, ctx->y2); std::swap(ctx->z, ctx->z2);
    std::swap(ctx->cell, ctx->cell2); std::swap(ctx->gid, ctx->gid2);
    return CPF_OK;
}

int cpf_num_particles(const cpf_context* ctx, int64_t* n) {
    CPF_REQUIRE(ctx, ctx && n, CPF_ERR_ARG, "null argument");
    *n = ctx->n;
    return CPF_OK;
}

int cpf_get_particles(cpf_context* ctx, double* xyzw, int32_t* cell, double* vel) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, ctx->n > 0, CPF_ERR_STATE, "cpf_get_particles: no particles");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)ctx->n;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    int r = ensureScratch(ctx, al(n * 32) + al(n * 4) + al(n * 32));
    if (r) return r;
    char* p = (char*)ctx->scratch;
    double* dX = (double*)p; p += al(n * 32);
    int32_t* dC = (int32_t*)p; p += al(n * 4);
    double* dV = (double*)p;
    CPF_HIP(ctx, cpf::launch_pack_by_gid(ctx->stream, ctx->x, ctx->y, ctx->z, ctx->cell, ctx->gid, ctx->vel,
                                         xyzw ? dX : nullptr, cell ? dC : nullptr, vel ? dV : nullptr, ctx->n));
    if (xyzw) CPF_HIP(ctx, hipMemcpyAsync(xyzw, dX, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    if (cell) CPF_HIP(ctx, hipMemcpyAsync(cell, dC, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (vel) CPF_HIP(ctx, hipMemcpyAsync(vel, dV, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    CPF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CPF_OK;
}

int cpf_get_counters(cpf_context* ctx, int64_t out[4]) {
    CPF_REQUIRE(ctx, ctx && out, CPF_ERR_ARG, "null argument");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<unsigned long long> h((size_t)cpf::kCounterSlots * 4);
    CPF_HIP(ctx, hipMemcpyAsync(h.data(), ctx->d_counters, h.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    CPF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 4; ++k) {
        unsigned long long sum = 0;
        for (int s = 0; s < cpf::kCounterSlots; ++s) sum += h[(size_t)s * 4 + k];
        out[k] = (int64_t)sum;
    }
    return CPF_OK;
}

int cpf_set_option(cpf_context* ctx, const char* key, double value) {
    CPF_REQUIRE(ctx, ctx && key, CPF_ERR_ARG, "null argument");
    const std::string k(key);
    if (k == "step_variant") {
        CPF_REQUIRE(ctx, value >= -1 && value <= 5 && value == (int)value, CPF_ERR_ARG, "step_variant must be -1..5");
#ifndef CPF_EXPERIMENTS
        CPF_REQUIRE(ctx, value != 1 && value != 2 && value != 5, CPF_ERR_ARG,
                    "step_variant 1, 2 and 5 are experiments (measured slower on every mesh) and not in this build: make EXPERIMENTS=1");
#endif
        ctx->stepVariant = (int)value;
        return CPF_OK;
    }
    if (k == "vertex_fast") {
        CPF_REQUIRE(ctx, value == 0 || value == 1, CPF_ERR_ARG, "vertex_fast must be 0 or 1");
        ctx->vertexFast = value != 0;
        return CPF_OK;
    }
    if (k == "z_fold") {
        CPF_REQUIRE(ctx, value == 0 || value == 1, CPF_ERR_ARG, "z_fold must be 0 or 1");
        ctx->zFold = value != 0;
        return CPF_OK;
    }
    if (k == "mixed_records") {
        CPF_REQUIRE(ctx, value == 0 || value == 1, CPF_ERR_ARG, "mixed_records must be 0 or 1");
        ctx->mixedRecords = value != 0;
        return CPF_OK;
    }
    if (k == "flat_walk") {
        CPF_REQUIRE(ctx, value == 0 || value == 1, CPF_ERR_ARG, "flat_walk must be 0 or 1");
        ctx->streamState.flat = (int)value;
        return CPF_OK;
    }
    if (k == "box_records") {
        CPF_REQUIRE(ctx, value == 0 || value == 1, CPF_ERR_ARG, "box_records must be 0 or 1");
        ctx->boxRecords = value != 0;
        return CPF_OK;
    }
    if (k == "stream_tiles_per_chunk") {
        CPF_REQUIRE(ctx, value >= 1 && value <= 1024 && value == (int)value, CPF_ERR_ARG, "stream_tiles_per_chunk must be 1..1024");
        ctx->streamState.tilesPerChunk = (int)value;
        return CPF_OK;
    }
    if (k == "sort_key_bits") {
        // sub-cell sort key layout: 100*bx + 10*by + bz bits for the position inside the cell's box along x, y, z
        // (most significant axis first as chosen at mesh ingest); default chosen by cpf_set_mesh
        CPF_REQUIRE(ctx, ctx->haveMesh, CPF_ERR_STATE, "sort_key_bits: call cpf_set_mesh first");
        const int v = (int)value, nb[3] = {v / 100, (v / 10) % 10, v % 10};
        CPF_REQUIRE(ctx, value == v && v >= 0 && nb[0] <= 9 && nb[0] + nb[1] + nb[2] <= 12, CPF_ERR_ARG, "sort_key_bits: at most 12 bits");
        for (int a = 0; a < 3; ++a) {
            const float f = std::ldexp(1.0f, nb[a] - ctx->host.subBits[a]);
            for (size_t c = 0; c < ctx->host.cellBox.size() / 6; ++c) ctx->host.cellBox[6 * c + 3 + a] *= f;
            ctx->host.subBits[a] = nb[a];
        }
        CPF_HIP(ctx, hipSetDevice(ctx->device));
        CPF_HIP(ctx, hipMemcpyAsync(ctx->d_cellBox, ctx->host.cellBox.data(), ctx->host.cellBox.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        CPF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return CPF_OK;
    }
    if (k == "stream_tail_fraction") {
        CPF_REQUIRE(ctx, value >= 0 && value <= 1, CPF_ERR_ARG, "stream_tail_fraction must be in [0, 1]");
        ctx->streamState.tailFraction = value;
        return CPF_OK;
    }
    if (k == "coop_max_cells") {
        CPF_REQUIRE(ctx, value >= 0 && value <= (1 << 24), CPF_ERR_ARG, "coop_max_cells must be in [0, 2^24]");
        ctx->streamState.coopMaxCells = (int)value;
        return CPF_OK;
    }
    if (k == "stream_lookup") {
        // (2, 3, 5 on an all-hex mesh: diagnostics -- what the mixed-mesh instantiations cost by themselves; same results)
        CPF_REQUIRE(ctx, value == -1 || (value >= 0 && value <= 6 && value == (int)value), CPF_ERR_ARG, "stream_lookup must be -1 (auto) or 0 ... 6");  // (8: chosen by the library, "flat_walk")
        ctx->streamState.lookup = (int)value;
        return CPF_OK;
    }
    if (k == "stream_lookup_by_density") {
        CPF_REQUIRE(ctx, value == 0 || value == 1, CPF_ERR_ARG, "stream_lookup_by_density must be 0 or 1");
        ctx->streamState.densityLookup = (int)value;
        return CPF_OK;
    }
    if (k == "sort_method") {
        CPF_REQUIRE(ctx, value == 0 || value == 2, CPF_ERR_ARG, "sort_method must be 2 (this library's radix sort) or 0 (hipcub's)");
        ctx->sortMethod = (int)value;
        return CPF_OK;
    }
    if (k == "sort_curve") {
        CPF_REQUIRE(ctx, value == -1 || value == 0 || value == 1, CPF_ERR_ARG, "sort_curve must be -1 (by regime), 0 (cell id) or 1 (Morton rank)");
        ctx->sortCurve = (int)value;
        return CPF_OK;
    }
    if (k == "vtu_binary") {
        CPF_REQUIRE(ctx, value == 0 || value == 1, CPF_ERR_ARG, "vtu_binary must be 0 or 1");
        ctx->vtuBinary = value != 0;
        return CPF_OK;
    }
    if (k == "stream_debug") {
        ctx->streamState.debug = (int)value;
        return CPF_OK;
    }
    if (k == "stream_waves_per_cu") {
        CPF_REQUIRE(ctx, value >= 0 && value <= 32 && value == (int)value, CPF_ERR_ARG, "stream_waves_per_cu must be 0..32 (0 = auto)");
        ctx->streamState.wavesPerCU = (int)value;
        return CPF_OK;
    }
    if (k == "sort_interval") {
        CPF_REQUIRE(ctx, value >= 0 && value <= 1e9, CPF_ERR_ARG, "sort_interval must be >= 0 (0 = never)");
        ctx->sortInterval = (int)value;
        return CPF_OK;
    }
    if (k == "timing_stride") {
        CPF_REQUIRE(ctx, value >= 1 && value <= 1e6 && value == (int)value, CPF_ERR_ARG, "timing_stride must be an integer >= 1");
        ctx->timingStride = (int)value;
        return CPF_OK;
    }
    if (k == "stats") {
        ctx->stats = value != 0;
        return CPF_OK;
    }
    return fail(ctx, CPF_ERR_ARG, "cpf_set_option: unknown key '" + k + "'");
}

}  // extern "C"   (helper for the other translation units, C++ linkage)
namespace cpf {
void set_context_error(cpf_context* ctx, const char* message) {
    if (ctx) ctx->err = message ? message : "";
}
}  // namespace cpf
extern "C" {

int cpf_step_kernel_name(cpf_context* ctx, double D, unsigned flags, char* buf, size_t bufBytes) {
    CPF_REQUIRE(ctx, ctx && buf && bufBytes > 0, CPF_ERR_ARG, "null argument");
    CPF_REQUIRE(ctx, ctx->haveMesh, CPF_ERR_STATE, "cpf_step_kernel_name: call cpf_set_mesh first");
    const cpf::MeshView m = meshView(ctx);
    pollFieldFlag(ctx);
    // (a fused launch's kernel depends on how many cycles it fuses: the most recent launch's count stands in)
    const int v = cpf::effective_step_variant(ctx->stepVariant, m, true, (flags & CPF_STEP_FUSE_CYCLES) ? ctx->lastStepCycles : 1,
                                              ctx->streamState.coopMaxCells);
    const char* b[2] = {"false", "true"};
    const bool brown = D > 0.0, reflect = (flags & CPF_STEP_NO_REFLECT) == 0, sv = (flags & CPF_STEP_STORE_VEL) != 0;
    char tmp[192];
    if ((flags & CPF_STEP_VERTEX_VELOCITY) &&
        cpf::step_vertex_streams(m, ctx->vertexFast ? ctx->d_vertCone : nullptr, ctx->tetsPerCell, ctx->stepVariant, &ctx->streamState,
                                 (flags & CPF_STEP_FUSE_CYCLES) ? ctx->lastStepCycles : 1))
        snprintf(tmp, sizeof tmp, "cpf::step_kernel_stream_vertex<%s, %s, %s, %s, %d> (cone locate)", b[brown], b[reflect], b[sv],
                 b[ctx->stats ? 1 : 0], cpf::stream_vertex_lookup_mode(ctx->lastStepN > 0 ? ctx->lastStepN : ctx->n, m, ctx->streamState));
    else if (flags & CPF_STEP_VERTEX_VELOCITY)
        snprintf(tmp, sizeof tmp, "cpf::step_kernel_vertex<%s, %s, %s> (%s)", b[brown], b[reflect], b[sv],
                 (ctx->vertexFast && ctx->d_vertCone) ? "cone locate" : (ctx->d_vertCone || ctx->vertConeWhy.empty() ? "all tets" : ("all tets: " + ctx->vertConeWhy).c_str()));
    else if (v == 5 && !brown && !sv && !(flags & CPF_STEP_FUSE_CYCLES))
        snprintf(tmp, sizeof tmp, "cpf::step_kernel_ahead<%s, %s>", b[reflect], b[ctx->stats]);
    else if (v == 4 || v == 5) {
        // (the record lookup is picked per launch from the particle count: the most recent launch's, else the owned cloud's)
        const int lf = cpf::stream_lookup_mode(ctx->lastStepN >= 0 ? ctx->lastStepN : ctx->n, m, ctx->streamState, brown);
        snprintf(tmp, sizeof tmp, "cpf::step_kernel_stream<%s, %s, %s, %s, %d>", b[brown], b[reflect], b[sv], b[ctx->stats], lf);
    }
    else if (v == 3) snprintf(tmp, sizeof tmp, "cpf::step_kernel_coop<%s, %s, %s, %s>", b[brown], b[reflect], b[sv], b[ctx->stats]);
    else snprintf(tmp, sizeof tmp, "cpf::step_kernel<%d, %s, %s, %s>", v, b[brown], b[reflect], b[sv]);
    snprintf(buf, bufBytes, "%s", tmp);
    return CPF_OK;
}

int cpf_set_seed(cpf_context* ctx, uint32_t seed) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    ctx->seed = seed;
    return CPF_OK;
}

int cpf_pack_leavers_dev(cpf_context* ctx, double* x, double* y, double* z, int32_t* cell, int64_t* gid, int64_t n,
                         const int32_t* cellLo_dev, int nRanks, int myRank, double* sendbuf, int64_t sendCapacity,
                         int64_t* counts_dev, int64_t* nStay_dev) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, n >= 0 && n < ((int64_t)1 << 31) && nRanks >= 1 && nRanks <= CPF_MAX_RANKS && myRank >= 0 && myRank < nRanks,
                CPF_ERR_ARG, "cpf_pack_leavers_dev: bad sizes (1 .. CPF_MAX_RANKS ranks)");
    CPF_REQUIRE(ctx, cellLo_dev && counts_dev && nStay_dev && (sendbuf || sendCapacity == 0) && (n == 0 || (x && y && z && cell)),
                CPF_ERR_ARG, "cpf_pack_leavers_dev: null array");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    int r = ensureScratch(ctx, cpf::handoff_scratch_bytes(n, nRanks));
    if (r) return r;
    CPF_HIP(ctx, cpf::pack_leavers(ctx->stream, x, y, z, cell, gid, n, cellLo_dev, nRanks, myRank, sendbuf,
                                   sendCapacity, counts_dev, nStay_dev, ctx->scratch, ctx->scratchBytes));
    return CPF_OK;
}

int cpf_cell_histogram_dev(cpf_context* ctx, const int32_t* cell, int64_t n, double scale, double* weights_dev) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, ctx->haveMesh, CPF_ERR_STATE, "cpf_cell_histogram_dev: call cpf_set_mesh first");
    CPF_REQUIRE(ctx, n >= 0 && weights_dev && (cell || n == 0), CPF_ERR_ARG, "cpf_cell_histogram_dev: bad arguments");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    int r = ensureScratch(ctx, cpf::histogram_scratch_bytes(ctx->host.nCells));
    if (r != CPF_OK) return r;
    CPF_HIP(ctx, cpf::cell_histogram(ctx->stream, cell, n, ctx->host.nCells, scale, weights_dev, ctx->scratch,
                                     ctx->scratchBytes));
    return CPF_OK;
}

int cpf_cell_ranges_dev(cpf_context* ctx, const double* weights_dev, int nRanks, int32_t* cellLo_dev) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, ctx->haveMesh, CPF_ERR_STATE, "cpf_cell_ranges_dev: call cpf_set_mesh first");
    CPF_REQUIRE(ctx, weights_dev && cellLo_dev && nRanks >= 1 && nRanks <= CPF_MAX_RANKS, CPF_ERR_ARG,
                "cpf_cell_ranges_dev: bad arguments (1 <= nRanks <= CPF_MAX_RANKS)");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    CPF_HIP(ctx, cpf::cell_ranges(ctx->stream, weights_dev, ctx->host.nCells, nRanks, cellLo_dev));
    return CPF_OK;
}

int cpf_unpack_arrivals_dev(cpf_context* ctx, double* x, double* y, double* z, int32_t* cell, int64_t* gid,
                            int64_t nStay, const double* recvbuf, int64_t nRecv) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, nStay >= 0 && nRecv >= 0 && (nRecv == 0 || (x && y && z && cell && recvbuf)), CPF_ERR_ARG,
                "cpf_unpack_arrivals_dev: bad arguments");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    CPF_HIP(ctx, cpf::unpack_arrivals(ctx->stream, x, y, z, cell, gid, nStay, recvbuf, nRecv));
    return CPF_OK;
}

// ---- device memory helpers ------------------------------------------------------------------
int cpf_dev_alloc(cpf_context* ctx, size_t bytes, void** out) {
    CPF_REQUIRE(ctx, ctx && out, CPF_ERR_ARG, "null argument");
    *out = nullptr;
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    CPF_HIP(ctx, hipMalloc(out, std::max<size_t>(bytes, 16)));
    return CPF_OK;
}
int cpf_dev_free(cpf_context* ctx, void* ptr) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    if (!ptr) return CPF_OK;
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    CPF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    CPF_HIP(ctx, hipFree(ptr));
    return CPF_OK;
}
int cpf_dev_memset(cpf_context* ctx, void* ptr, int value, size_t bytes) {
    CPF_REQUIRE(ctx, ctx && (ptr || bytes == 0), CPF_ERR_ARG, "null argument");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    if (bytes) CPF_HIP(ctx, hipMemsetAsync(ptr, value, bytes, ctx->stream));
    return CPF_OK;
}
int cpf_copy_to_device(cpf_context* ctx, void* dst, const void* src, size_t bytes) {
    CPF_REQUIRE(ctx, ctx && ((dst && src) || bytes == 0), CPF_ERR_ARG, "null argument");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    if (bytes) {
        CPF_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        CPF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return CPF_OK;
}
int cpf_copy_to_host(cpf_context* ctx, void* dst, const void* src, size_t bytes) {
    CPF_REQUIRE(ctx, ctx && ((dst && src) || bytes == 0), CPF_ERR_ARG, "null argument");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    if (bytes) {
        CPF_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        CPF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return CPF_OK;
}

int cpf_copy_dev(cpf_context* ctx, void* dst, const void* src, size_t bytes) {
    CPF_REQUIRE(ctx, ctx && ((dst && src) || bytes == 0), CPF_ERR_ARG, "null argument");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    if (bytes) CPF_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return CPF_OK;
}

// ---- stage-by-stage entry points (reference layouts) --------------------------------------------
#define CPF_STAGE_PRE(name, needU)                                                                      \
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");                                                 \
    CPF_REQUIRE(ctx, ctx->haveMesh, CPF_ERR_STATE, name ": call cpf_set_mesh first");                   \
    CPF_REQUIRE(ctx, !(needU) || ctx->haveU, CPF_ERR_STATE, name ": call cpf_set_velocity first");      \
    CPF_REQUIRE(ctx, n >= 0, CPF_ERR_ARG, name ": negative particle count");                            \
    CPF_HIP(ctx, hipSetDevice(ctx->device))

int cpf_stage_seed_box(cpf_context* ctx, double* particles, int64_t n, const double lower[3], const double upper[3],
                       int order) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, n >= 0 && lower && upper && (particles || n == 0), CPF_ERR_ARG, "cpf_stage_seed_box: bad arguments");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    int r = ensureScratch(ctx, (size_t)std::max<int64_t>(n, 1) * 24);
    if (r) return r;
    double* x = (double*)ctx->scratch; double* y = x + n; double* z = y + n;
    CPF_HIP(ctx, cpf::launch_seed_box(ctx->stream, x, y, z, 0, n, lower, upper, order));
    CPF_HIP(ctx, cpf::launch_soa_to_aos(ctx->stream, x, y, z, particles, n));
    return CPF_OK;
}
int cpf_stage_locate_initial(cpf_context* ctx, const double* particles, int32_t* ids, int64_t n) {
    CPF_STAGE_PRE("cpf_stage_locate_initial", false);
    CPF_REQUIRE(ctx, n == 0 || (particles && ids), CPF_ERR_ARG, "cpf_stage_locate_initial: null array");
    int r = ensureScratch(ctx, (size_t)std::max<int64_t>(n, 1) * 24);
    if (r) return r;
    double* x = (double*)ctx->scratch; double* y = x + n; double* z = y + n;
    CPF_HIP(ctx, cpf::launch_aos_to_soa(ctx->stream, particles, x, y, z, n));
    CPF_HIP(ctx, cpf::launch_locate_initial(ctx->stream, x, y, z, ids, n, meshView(ctx), gridView(ctx)));
    return CPF_OK;
}
int cpf_stage_count_outside(cpf_context* ctx, const int32_t* ids, int64_t n, int64_t* nNegative) {
    CPF_REQUIRE(ctx, ctx && nNegative && (ids || n == 0) && n >= 0, CPF_ERR_ARG, "cpf_stage_count_outside: bad arguments");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    unsigned long long* cnt = ctx->d_counters + cpf::kCounterSlots * 4;
    CPF_HIP(ctx, hipMemsetAsync(cnt, 0, 8, ctx->stream));
    CPF_HIP(ctx, cpf::launch_count_negative(ctx->stream, ids, n, cnt));
    unsigned long long h = 0;
    CPF_HIP(ctx, hipMemcpyAsync(&h, cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
    CPF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *nNegative = (int64_t)h;
    return CPF_OK;
}
int cpf_stage_advect(cpf_context* ctx, double* particles, const int32_t* ids, double* vels, double* disps, double dt,
                     int64_t n) {
    CPF_STAGE_PRE("cpf_stage_advect", true);
    CPF_REQUIRE(ctx, n == 0 || (particles && ids && vels && disps), CPF_ERR_ARG, "cpf_stage_advect: null array");
    CPF_HIP(ctx, cpf::launch_stage_advect(ctx->stream, particles, ids, vels, disps, dt, n, meshView(ctx)));
    return CPF_OK;
}
int cpf_stage_advect_const(cpf_context* ctx, double* particles, const int32_t* ids, const double* vels, double* disps, double dt,
                           int64_t n) {
    CPF_STAGE_PRE("cpf_stage_advect_const", false);
    CPF_REQUIRE(ctx, n == 0 || (particles && ids && vels && disps), CPF_ERR_ARG, "cpf_stage_advect_const: null array");
    CPF_HIP(ctx, cpf::launch_stage_advect_const(ctx->stream, particles, ids, vels, disps, dt, n));
    return CPF_OK;
}
int cpf_set_tets(cpf_context* ctx, const double* positions, int64_t nVerts, const int32_t* tets, int64_t nTets,
                 int tetsPerCell) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, ctx->haveMesh, CPF_ERR_STATE, "cpf_set_tets: call cpf_set_mesh first");
    CPF_REQUIRE(ctx, positions && tets && nVerts > 0 && tetsPerCell > 0, CPF_ERR_ARG, "cpf_set_tets: bad arguments");
    CPF_REQUIRE(ctx, nTets == (int64_t)tetsPerCell * ctx->host.nCells, CPF_ERR_MESH,
                "cpf_set_tets: nTets must be tetsPerCell x nCells (tets in cell order, src/initCuda.H:99-105)");
    for (int64_t k = 0; k < 4 * nTets; ++k)
        CPF_REQUIRE(ctx, tets[k] >= 0 && tets[k] < nVerts, CPF_ERR_MESH, "cpf_set_tets: tet vertex out of range");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    CPF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    freeDev(ctx->d_tetPos); freeDev(ctx->d_tets); freeDev(ctx->d_vertVel); freeDev(ctx->d_vertCone); freeDev(ctx->d_vertApex);
    ctx->haveVertVel = false;
    CPF_HIP(ctx, hipMalloc((void**)&ctx->d_tetPos, (size_t)nVerts * 24));
    CPF_HIP(ctx, hipMalloc((void**)&ctx->d_tets, (size_t)nTets * 16));
    CPF_HIP(ctx, hipMalloc((void**)&ctx->d_vertVel, (size_t)nVerts * 24));
    CPF_HIP(ctx, hipMemcpyAsync(ctx->d_tetPos, positions, (size_t)nVerts * 24, hipMemcpyHostToDevice, ctx->stream));
    CPF_HIP(ctx, hipMemcpyAsync(ctx->d_tets, tets, (size_t)nTets * 16, hipMemcpyHostToDevice, ctx->stream));
    CPF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->nTetVerts = nVerts; ctx->nTets = nTets; ctx->tetsPerCell = tetsPerCell;
    // the cone locate (cpf_kernels.hip, VertexField) is exact only on a decomposition whose tets cannot overlap: decided here
    ctx->vertConeWhy = tetFanDefect(positions, tets, ctx->host.nCells, tetsPerCell);
    if (ctx->vertConeWhy.empty()) {
        CPF_HIP(ctx, hipMalloc((void**)&ctx->d_vertCone, (size_t)nTets * 256));
        CPF_HIP(ctx, hipMalloc((void**)&ctx->d_vertApex, (size_t)ctx->host.nCells * 32));
        CPF_HIP(ctx, cpf::launch_vertex_cone_tables(ctx->stream, ctx->d_tetPos, ctx->d_tets, nTets, tetsPerCell, ctx->d_vertCone, ctx->d_vertApex));
        CPF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return CPF_OK;
}
int cpf_set_vertex_velocity(cpf_context* ctx, const double* vertexU, int64_t nVerts) {
    CPF_REQUIRE(ctx, ctx && vertexU, CPF_ERR_ARG, "null argument");
    CPF_REQUIRE(ctx, ctx->d_tets, CPF_ERR_STATE, "cpf_set_vertex_velocity: call cpf_set_tets first");
    CPF_REQUIRE(ctx, nVerts == ctx->nTetVerts, CPF_ERR_ARG, "cpf_set_vertex_velocity: one velocity per tet-mesh vertex");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    CPF_HIP(ctx, hipMemcpyAsync(ctx->d_vertVel, vertexU, (size_t)nVerts * 24, hipMemcpyHostToDevice, ctx->stream));
    if (ctx->d_vertCone) CPF_HIP(ctx, cpf::launch_vertex_record_velocity(ctx->stream, ctx->d_tets, ctx->d_vertVel, ctx->nTets, ctx->d_vertCone));
    CPF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->haveVertVel = true;
    return CPF_OK;
}
int cpf_stage_advect_vertex(cpf_context* ctx, double* particles, const int32_t* ids, double* vels, double* disps,
                            double dt, int64_t n) {
    CPF_STAGE_PRE("cpf_stage_advect_vertex", false);
    CPF_REQUIRE(ctx, ctx->haveVertVel && ctx->nTets == (int64_t)ctx->tetsPerCell * ctx->host.nCells, CPF_ERR_STATE,
                "cpf_stage_advect_vertex: call cpf_set_tets and cpf_set_vertex_velocity (for the current mesh) first");
    CPF_REQUIRE(ctx, n == 0 || (particles && ids && vels && disps), CPF_ERR_ARG, "cpf_stage_advect_vertex: null array");
    CPF_HIP(ctx, cpf::launch_stage_advect_vertex(ctx->stream, particles, ids, vels, disps, dt, n, ctx->d_tetPos, ctx->d_tets,
                                                 ctx->tetsPerCell, ctx->d_vertVel, ctx->vertexFast ? ctx->d_vertCone : nullptr, ctx->d_vertApex));
    return CPF_OK;
}
int cpf_stage_brownian(cpf_context* ctx, const double* particles, double* disps, double dt, int64_t n, double D,
                       uint32_t step) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, n >= 0 && (n == 0 || (particles && disps)) && D >= 0.0, CPF_ERR_ARG, "cpf_stage_brownian: bad arguments");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    CPF_HIP(ctx, cpf::launch_stage_brownian(ctx->stream, particles, disps, dt, n, D, step, ctx->seed));
    return CPF_OK;
}
int cpf_stage_locate(cpf_context* ctx, const double* particles, const double* disps, int32_t* ids, int64_t n) {
    CPF_STAGE_PRE("cpf_stage_locate", false);
    CPF_REQUIRE(ctx, n == 0 || (particles && disps && ids), CPF_ERR_ARG, "cpf_stage_locate: null array");
    CPF_HIP(ctx, cpf::launch_stage_locate(ctx->stream, particles, disps, ids, n, meshView(ctx)));
    return CPF_OK;
}
int cpf_stage_reflect(cpf_context* ctx, int32_t* ids, double* particles, double* vels, double* disps, int64_t n) {
    CPF_STAGE_PRE("cpf_stage_reflect", false);
    CPF_REQUIRE(ctx, n == 0 || (particles && disps && ids && vels), CPF_ERR_ARG, "cpf_stage_reflect: null array");
    CPF_HIP(ctx, cpf::launch_stage_reflect(ctx->stream, ids, particles, vels, disps, n, meshView(ctx)));
    return CPF_OK;
}
int cpf_stage_move(cpf_context* ctx, double* particles, double* disps, int64_t n) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    CPF_REQUIRE(ctx, n >= 0 && (n == 0 || (particles && disps)), CPF_ERR_ARG, "cpf_stage_move: bad arguments");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    CPF_HIP(ctx, cpf::launch_stage_move(ctx->stream, particles, disps, n));
    return CPF_OK;
}

int cpf_write_vtu_wait(cpf_context* ctx) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    if (!ctx->writerLive) return CPF_OK;
    ctx->writer.join();
    ctx->writerLive = false;
    {
        std::lock_guard<std::mutex> lk(g_mutex);
        auto& v = g_writers.live;
        v.erase(std::remove(v.begin(), v.end(), ctx), v.end());
    }
    const int r = ctx->writerStatus;
    ctx->writerStatus = CPF_OK;
    if (r != CPF_OK && r != CPF_WARN_NAN) return fail(ctx, r, "cpf_write_vtu_async: the frame could not be written");
    return r;
}

int cpf_write_vtu_async(cpf_context* ctx, const char* path, double* totalKE) {
    CPF_REQUIRE(ctx, ctx && path, CPF_ERR_ARG, "null argument");
    int r = cpf_write_vtu_wait(ctx);                       // one frame in flight; reports the previous frame's failure
    if (r != CPF_OK && r != CPF_WARN_NAN) return r;
    CPF_REQUIRE(ctx, ctx->n > 0, CPF_ERR_STATE, "cpf_write_vtu_async: no particles");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)ctx->n;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t offC = al(n * 32), offV = offC + al(n * 4), need = offV + al(n * 32);
    if (!ctx->ioStream) {
        CPF_HIP(ctx, hipStreamCreateWithFlags(&ctx->ioStream, hipStreamNonBlocking));
        CPF_HIP(ctx, hipEventCreateWithFlags(&ctx->evSnap, hipEventDisableTiming));
        CPF_HIP(ctx, hipEventCreateWithFlags(&ctx->evCopied, hipEventDisableTiming));
    }
    if (need > ctx->snapBytes) {                           // (the previous frame's worker has been joined: nobody reads these)
        if (ctx->snapDev) { (void)hipFree(ctx->snapDev); ctx->snapDev = nullptr; }
        if (ctx->snapHost) { (void)hipHostFree(ctx->snapHost); ctx->snapHost = nullptr; }
        ctx->snapBytes = 0;
        const size_t want = need + need / 8;
        hipError_t e = hipMalloc(&ctx->snapDev, want);
        if (e == hipSuccess) e = hipHostMalloc(&ctx->snapHost, want, hipHostMallocDefault);
        if (e != hipSuccess) {
            if (ctx->snapDev) { (void)hipFree(ctx->snapDev); ctx->snapDev = nullptr; }
            return fail(ctx, e == hipErrorOutOfMemory ? CPF_ERR_NOMEM : CPF_ERR_HIP, std::string("cpf_write_vtu_async: snapshot buffers: ") + hipGetErrorString(e));
        }
        ctx->snapBytes = want;
    }
    // ---- the snapshot: ONE kernel on the compute stream (particle-id order, the layouts the writer reads); everything else --
    // PCIe, the energy sum, formatting, the file -- happens behind the caller's back
    char* d = (char*)ctx->snapDev;
    CPF_HIP(ctx, cpf::launch_pack_by_gid(ctx->stream, ctx->x, ctx->y, ctx->z, ctx->cell, ctx->gid, ctx->vel, (double*)d, (int32_t*)(d + offC),
                                         (double*)(d + offV), ctx->n));
    CPF_HIP(ctx, hipEventRecord(ctx->evSnap, ctx->stream));
    CPF_HIP(ctx, hipStreamWaitEvent(ctx->ioStream, ctx->evSnap, 0));
    CPF_HIP(ctx, hipMemcpyAsync(ctx->snapHost, ctx->snapDev, need, hipMemcpyDeviceToHost, ctx->ioStream));
    CPF_HIP(ctx, hipEventRecord(ctx->evCopied, ctx->ioStream));
    const std::string file(path);
    { std::lock_guard<std::mutex> lk(g_mutex); g_writers.live.push_back(ctx); }
    ctx->writerLive = true;
    ctx->keReady = false;
    const bool binary = ctx->vtuBinary;
    ctx->writer = std::thread([ctx, file, n, binary, offC, offV] {
        (void)hipSetDevice(ctx->device);
        const hipError_t e = hipEventSynchronize(ctx->evCopied);
        const char* h = (const char*)ctx->snapHost;
        const double* xyzw = (const double*)h; const int32_t* cell = (const int32_t*)(h + offC); const double* vel = (const double*)(h + offV);
        double total = 0.0;                                // in index order, like the reference's running sum
        if (e == hipSuccess)
            for (size_t i = 0; i < n; ++i) total += 0.5 * (vel[4 * i] * vel[4 * i] + vel[4 * i + 1] * vel[4 * i + 1] + vel[4 * i + 2] * vel[4 * i + 2]);
        { std::lock_guard<std::mutex> lk(ctx->keMutex); ctx->frameKE = total; ctx->keReady = true; }
        ctx->keCv.notify_all();
        ctx->writerStatus = e != hipSuccess ? CPF_ERR_HIP
                                            : (binary ? cpf_write_vtu_arrays_binary : cpf_write_vtu_arrays)(file.c_str(), (int64_t)n, xyzw, cell, vel, nullptr);
    });
    if (!totalKE) return CPF_OK;                           // the caller is back in its step loop after the one kernel launch
    // the energy at once (a host that prints it where the reference does): that host waits for the copy and one pass over it
    std::unique_lock<std::mutex> lk(ctx->keMutex);
    ctx->keCv.wait(lk, [ctx] { return ctx->keReady; });
    *totalKE = ctx->frameKE;
    return std::isnan(ctx->frameKE) ? CPF_WARN_NAN : CPF_OK;
}

int cpf_timing_enable(cpf_context* ctx, int on) {
    CPF_REQUIRE(ctx, ctx, CPF_ERR_ARG, "null context");
    ctx->timing = on != 0;
    return CPF_OK;
}

int cpf_timing_read(cpf_context* ctx, int64_t* launches, double* total_ms) {
    CPF_REQUIRE(ctx, ctx && launches && total_ms, CPF_ERR_ARG, "null argument");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    CPF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double tot = 0.0;
    for (auto& p : ctx->events) {
        float ms = 0.f;
        CPF_HIP(ctx, hipEventElapsedTime(&ms, p.first, p.second));
        tot += (double)ms;
        ctx->eventPool.push_back(p.first); ctx->eventPool.push_back(p.second);
    }
    *launches = (int64_t)ctx->events.size();
    *total_ms = tot;
    ctx->events.clear();
    return CPF_OK;
}

int cpf_timing_poll(cpf_context* ctx, int64_t* launches, double* total_ms) {
    CPF_REQUIRE(ctx, ctx && launches && total_ms, CPF_ERR_ARG, "null argument");
    CPF_HIP(ctx, hipSetDevice(ctx->device));
    double tot = 0.0;
    size_t done = 0;
    for (; done < ctx->events.size(); ++done) {          // launches complete in stream order
        auto& p = ctx->events[done];
        const hipError_t q = hipEventQuery(p.second);
        if (q == hipErrorNotReady) break;
        CPF_HIP(ctx, q);
        float ms = 0.f;
        CPF_HIP(ctx, hipEventElapsedTime(&ms, p.first, p.second));
        tot += (double)ms;
        ctx->eventPool.push_back(p.first); ctx->eventPool.push_back(p.second);
    }
    ctx->events.erase(ctx->events.begin(), ctx->events.begin() + (std::ptrdiff_t)done);
    *launches = (int64_t)done;
    *total_ms = tot;
    return CPF_OK;
}

}  // extern "C"
