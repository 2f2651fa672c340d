// The sharded cloud's host logic (include/cpf.h "cpf_shard"): ownership ranges, re-cut, split, counts all-gather,
// all-to-all-v on a side stream while the step loop runs on, append, catch-up replay, growth and overflow retry.
//
// Written against a small device interface `Dev` (memory, two streams, events, and the hand-off / step kernels) so that
// the SAME logic is compiled twice: in the product with the HIP device (cpf_shard.cpp -- there is no other device in the
// product), and in tests/host_shard with a host-memory stand-in whose "kernels" are the CPU checker, which lets the
// world-size 2 / 3 / 8 gloo tests drive this very code on a machine without a GPU.
//
// No reference counterpart: the reference drives ONE GPU from the MPI master rank (src/advect.H:59-89).
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "cpf.h"

namespace cpf {

constexpr double kCostUnitMs = 2.0e-8;   // ms per particle-step that counts as cost 1 (~ the measured single-GPU rate)

inline double nowMs() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <class Dev>
class ShardCore {
  public:
    typedef typename Dev::Stream Stream;
    typedef typename Dev::Event Event;

    Dev dev;
    cpf_comm comm{};            // a copy of the caller's table (self == nullptr and nRanks == 1: no communicator)
    bool haveComm = false;
    int rank = 0, W = 1;
    std::string err;

    // the shard: SoA arrays with slack, a second set the sort writes into, velocities only while a frame is due
    double *x = nullptr, *y = nullptr, *z = nullptr, *vel = nullptr;
    int32_t* cell = nullptr;
    int64_t* gid = nullptr;
    double *ax = nullptr, *ay = nullptr, *az = nullptr;
    int32_t* acell = nullptr;
    int64_t* agid = nullptr;
    int64_t cap = 0, n = 0;

    // ownership and hand-off buffers
    std::vector<int32_t> cellLo;          // [W + 1] host copy (refreshed by every hand-off)
    int32_t* d_cellLo = nullptr;
    int32_t* h_cellLo = nullptr;          // pinned
    double *sendbuf = nullptr, *recvbuf = nullptr;
    int64_t sendCap = 0, recvCap = 0;
    // one row per rank: counts[W] | nStay | wanted overlap depth.  The split kernels write counts and nStay; the rows of all
    // ranks are all-gathered into d_table and copied to the pinned h_table ONCE per hand-off
    int64_t *d_meta = nullptr, *d_table = nullptr, *h_table = nullptr, *h_want = nullptr;
    int RW = 3;
    double* d_weights = nullptr;
    int64_t nCells = 0;
    double* d_scalar = nullptr;           // [W + 2] doubles: small collectives (counts as doubles are exact below 2^53)
    double* h_scalar = nullptr;           // pinned
    double* d_Ufull = nullptr;            // [nCells][3] staging of cpf_shard_set_velocity_slice
    std::vector<int64_t> pieceCells;      // cells of every rank's mesh piece (set_velocity_slice)

    // options
    int exchangeInterval = 0, rebalanceInterval = 0, overlapSteps = 0, sortInterval = 0;
    bool balanceByTime = false, forceCollectives = false, profileComm = false;
    double sendFraction = 0.25;

    // state of the step loop
    uint32_t stepIndex = 0;
    bool velValid = false;       // `vel` holds the last cycle's velocities of the particles AS THEY ARE ORDERED NOW (round-5 advisory)
    struct Pending { bool on = false; uint32_t step = 0; } pending;
    bool sortDue = false, deferSort = false, deferRecut = false, deferExchange = false;
    bool haveArgs = false;
    double argDt = 0, argD = 0;
    unsigned argFlags = 0;
    int depthNext = 2;                    // overlap depth of the NEXT hand-off when derived ("overlap_steps" -1): the maximum of
                                          // what the ranks asked for in the last table, so that every rank completes a hand-off at
                                          // the same step (an overflow re-split "at the current step" is then current everywhere)
    bool haveCost = false, costWaited = false;
    int64_t launchesIssued = 0, cyclesIssued = 0;   // timed step launches and their cycles since the last cost reading
    double costPerParticle = 1.0;
    double hostWorkMs = 0.5;              // running mean of the host's own work per hand-off, without the wait

    // events (re-used)
    Event evPack = Event(), evGot = Event(), evDone = Event(), evRepack = Event();
    std::vector<std::pair<Event, Event>> commEvents;   // "profile_comm"
    size_t commEventsRead = 0;
    double commMsRead = 0.0;

    // asynchronous frames (cpf_shard_write_vtu on the root): one in flight, formatted and written on a worker thread
    std::thread writer;
    bool writerLive = false;
    int writerStatus = CPF_OK;
    // the frame in flight (round 6: nothing of a frame is copied or summed on the caller's thread, and the step loop's stream
    // does not wait for PCIe): the gathered cloud stays in device buffers that live as long as the shard, goes to pinned host
    // memory on the `io` stream behind an event, and the worker thread takes it from there
    double *fOut = nullptr, *fIn = nullptr, *fXyzw = nullptr, *fVel = nullptr;
    int32_t* fCell = nullptr; int64_t* fBad = nullptr;
    int64_t fOutCap = 0, fInCap = 0;                       // records
    char* fHost = nullptr; size_t fHostBytes = 0;          // pinned: xyzw | vel | cell | bad
    Event evFrame = Event(), evFrameCopied = Event();
    double frameKE = 0.0;
    std::mutex keMutex; std::condition_variable keCv; bool keReady = false;

    // statistics
    int64_t particleSteps = 0, handedOff = 0, exchanges = 0, rebalances = 0, grown = 0, sendGrown = 0, kernelLaunches = 0;
    double kernelMs = 0.0, handoffHostMs = 0.0, handoffWaitMs = 0.0;

#define CPF_SH(call)                        \
    do {                                    \
        const int r__ = (call);             \
        if (r__ != CPF_OK) return note(r__); \
    } while (0)
#define CPF_SH_COMM(call)                                   \
    do {                                                    \
        const int r__ = (call);                             \
        if (r__ != CPF_OK) return commError(r__, #call);    \
    } while (0)

    int note(int code) {
        if (err.empty()) err = dev.lastError();
        return code;
    }
    int fail(int code, const std::string& m) { err = m; return code; }
    int commError(int code, const char* what) {
        const char* m = (comm.last_error && comm.self) ? comm.last_error(comm.self) : nullptr;
        err = std::string("communicator: ") + what + (m && *m ? std::string(": ") + m : std::string(" failed"));
        return code;
    }
    bool distOn() const { return W > 1 || forceCollectives; }

    // ---------------------------------------------------------------------------------------------- set-up
    int init(const cpf_comm* c, int64_t capacity, const int32_t* lo) {
        if (capacity < 1 || capacity >= ((int64_t)1 << 31)) return fail(CPF_ERR_ARG, "cpf_shard_create: capacity must be in (0, 2^31)");
        nCells = dev.nCells();
        if (nCells <= 0) return fail(CPF_ERR_STATE, "cpf_shard_create: set the mesh on the context first");
        if (c) {
            if (!c->all_gather || !c->all_reduce_sum_f64 || !c->all_to_all_v || c->nRanks < 1 || c->nRanks > CPF_MAX_RANKS ||
                c->rank < 0 || c->rank >= c->nRanks)
                return fail(CPF_ERR_ARG, "cpf_shard_create: incomplete communicator (three collectives, 1 <= nRanks <= CPF_MAX_RANKS)");
            comm = *c; haveComm = true; rank = c->rank; W = c->nRanks;
        }
        RW = W + 2;
        cellLo.assign((size_t)W + 1, 0);
        for (int r = 0; r <= W; ++r) cellLo[(size_t)r] = lo ? lo[r] : (int32_t)(nCells * r / W);
        if (cellLo[0] != 0 || cellLo[(size_t)W] != (int32_t)nCells) return fail(CPF_ERR_ARG, "cpf_shard_create: cellLo must run from 0 to nCells");
        for (int r = 0; r < W; ++r)
            if (cellLo[(size_t)r] > cellLo[(size_t)r + 1]) return fail(CPF_ERR_ARG, "cpf_shard_create: cellLo must be non-decreasing");
        CPF_SH(allocArrays(capacity));
        CPF_SH(dev.alloc((void**)&d_cellLo, (size_t)(W + 1) * 4));
        CPF_SH(dev.hostAlloc((void**)&h_cellLo, (size_t)(W + 1) * 4));
        CPF_SH(dev.alloc((void**)&d_meta, (size_t)RW * 8));
        CPF_SH(dev.alloc((void**)&d_table, (size_t)W * RW * 8));
        CPF_SH(dev.hostAlloc((void**)&h_table, (size_t)W * RW * 8));
        CPF_SH(dev.hostAlloc((void**)&h_want, 8));
        CPF_SH(dev.alloc((void**)&d_scalar, (size_t)(W + 2) * 8));
        CPF_SH(dev.hostAlloc((void**)&h_scalar, (size_t)(W + 2) * 8));
        std::memcpy(h_cellLo, cellLo.data(), (size_t)(W + 1) * 4);
        CPF_SH(dev.copy(d_cellLo, h_cellLo, (size_t)(W + 1) * 4, dev.compute()));
        CPF_SH(dev.fill(d_meta, 0, (size_t)RW * 8, dev.compute()));
        CPF_SH(dev.streamSync(dev.compute()));
        CPF_SH(dev.eventCreate(&evPack, false)); CPF_SH(dev.eventCreate(&evGot, false));
        CPF_SH(dev.eventCreate(&evDone, false)); CPF_SH(dev.eventCreate(&evRepack, false));
        return resizeSend(std::max<int64_t>(1024, (int64_t)((double)cap * sendFraction)));
    }

    // the frame still being written (if any) is complete when this returns; its status is reported once
    int writerWait() {
        if (!writerLive) return CPF_OK;
        writer.join();
        writerLive = false;
        const int r = writerStatus;
        writerStatus = CPF_OK;
        return r;
    }

    void destroy() {
        (void)writerWait();                                    // never lose a frame
        (void)dev.streamSync(dev.compute()); (void)dev.streamSync(dev.side());
        for (void* p : {(void*)x, (void*)y, (void*)z, (void*)vel, (void*)cell, (void*)gid, (void*)ax, (void*)ay, (void*)az, (void*)acell,
                        (void*)agid, (void*)d_cellLo, (void*)sendbuf, (void*)recvbuf, (void*)d_meta, (void*)d_table, (void*)d_weights,
                        (void*)d_scalar, (void*)d_Ufull})
            if (p) dev.release(p);
        for (void* p : {(void*)fOut, (void*)fIn, (void*)fXyzw, (void*)fVel, (void*)fCell, (void*)fBad})
            if (p) dev.release(p);
        for (void* p : {(void*)h_cellLo, (void*)h_table, (void*)h_want, (void*)h_scalar, (void*)fHost})
            if (p) dev.hostRelease(p);
        for (Event e : {evFrame, evFrameCopied})
            if (e != Event()) dev.eventDestroy(e);
        for (Event e : {evPack, evGot, evDone, evRepack})
            if (e != Event()) dev.eventDestroy(e);
        for (auto& p : commEvents) { dev.eventDestroy(p.first); dev.eventDestroy(p.second); }
        x = y = z = vel = nullptr; cell = nullptr; gid = nullptr;
    }

    int allocArrays(int64_t c) {
        CPF_SH(dev.alloc((void**)&x, (size_t)c * 8)); CPF_SH(dev.alloc((void**)&y, (size_t)c * 8));
        CPF_SH(dev.alloc((void**)&z, (size_t)c * 8)); CPF_SH(dev.alloc((void**)&cell, (size_t)c * 4));
        CPF_SH(dev.alloc((void**)&gid, (size_t)c * 8));
        CPF_SH(dev.fill(cell, 0xFF, (size_t)c * 4, dev.compute()));
        cap = c;
        return CPF_OK;
    }

    int resizeSend(int64_t records) {
        if (sendbuf) { CPF_SH(dev.streamSync(dev.side())); CPF_SH(dev.streamSync(dev.compute())); dev.release(sendbuf); sendbuf = nullptr; }
        sendCap = records;
        CPF_SH(dev.alloc((void**)&sendbuf, (size_t)std::max<int64_t>(records, 1) * CPF_HANDOFF_DOUBLES * 8));
        return CPF_OK;
    }
    int ensureRecv(int64_t records) {
        if (records <= recvCap) return CPF_OK;
        if (recvbuf) { CPF_SH(dev.streamSync(dev.side())); CPF_SH(dev.streamSync(dev.compute())); dev.release(recvbuf); recvbuf = nullptr; }
        recvCap = records + records / 8 + 1024;
        CPF_SH(dev.alloc((void**)&recvbuf, (size_t)recvCap * CPF_HANDOFF_DOUBLES * 8));
        return CPF_OK;
    }

    // More arrivals than slack: move the shard into larger arrays (HBM is plentiful; on the compute stream, so it is ordered
    // after the steps in flight).
    int grow(int64_t needed, int64_t nKeep) {
        const int64_t c = std::max<int64_t>(needed, (int64_t)((double)cap * 1.5)) + 4096;
        if (c >= ((int64_t)1 << 31)) return fail(CPF_ERR_NOMEM, "sharded cloud: a rank would hold 2^31 particles or more");
        double *nx = nullptr, *ny = nullptr, *nz = nullptr; int32_t* nc = nullptr; int64_t* ng = nullptr;
        CPF_SH(dev.alloc((void**)&nx, (size_t)c * 8)); CPF_SH(dev.alloc((void**)&ny, (size_t)c * 8));
        CPF_SH(dev.alloc((void**)&nz, (size_t)c * 8)); CPF_SH(dev.alloc((void**)&nc, (size_t)c * 4));
        CPF_SH(dev.alloc((void**)&ng, (size_t)c * 8));
        // the velocities of a frame that is due (CPF_STEP_STORE_VEL: the arrivals' catch-up replay fills in theirs) move along
        double* nv = nullptr;
        if (vel) CPF_SH(dev.alloc((void**)&nv, (size_t)c * 24));
        Stream s = dev.compute();
        CPF_SH(dev.fill(nc, 0xFF, (size_t)c * 4, s));
        if (nKeep > 0) {
            CPF_SH(dev.copy(nx, x, (size_t)nKeep * 8, s)); CPF_SH(dev.copy(ny, y, (size_t)nKeep * 8, s));
            CPF_SH(dev.copy(nz, z, (size_t)nKeep * 8, s)); CPF_SH(dev.copy(nc, cell, (size_t)nKeep * 4, s));
            CPF_SH(dev.copy(ng, gid, (size_t)nKeep * 8, s));
            if (nv) CPF_SH(dev.copy(nv, vel, (size_t)nKeep * 24, s));
        }
        CPF_SH(dev.streamSync(s));
        for (void* p : {(void*)x, (void*)y, (void*)z, (void*)cell, (void*)gid, (void*)vel, (void*)ax, (void*)ay, (void*)az, (void*)acell, (void*)agid})
            if (p) dev.release(p);
        x = nx; y = ny; z = nz; cell = nc; gid = ng;
        vel = nv; ax = ay = az = nullptr; acell = nullptr; agid = nullptr;          // (the sort's second set: re-made on demand at the new size)
        const bool wholeShard = sendCap >= cap;                                      // "a send buffer as large as the shard" stays so
        cap = c;
        ++grown;
        if (wholeShard) CPF_SH(resizeSend(c));
        return CPF_OK;
    }

    int setOption(const std::string& k, double v) {
        const bool whole = v == std::floor(v);
        if (k == "exchange_interval") { if (v < 0 || !whole) return fail(CPF_ERR_ARG, "exchange_interval must be an integer >= 0"); exchangeInterval = (int)v; }
        else if (k == "rebalance_interval") { if (v < 0 || !whole) return fail(CPF_ERR_ARG, "rebalance_interval must be an integer >= 0"); rebalanceInterval = (int)v; }
        else if (k == "overlap_steps") {
            if (v < -1 || !whole) return fail(CPF_ERR_ARG, "overlap_steps must be an integer >= -1");
            overlapSteps = (int)v;
        }
        else if (k == "sort_interval") { if (v < 0 || !whole) return fail(CPF_ERR_ARG, "sort_interval must be an integer >= 0"); sortInterval = (int)v; }
        else if (k == "balance_by_time") { balanceByTime = v != 0; if (balanceByTime) CPF_SH(dev.timingEnable(true)); }
        else if (k == "force_collectives") forceCollectives = v != 0;
        else if (k == "profile_comm") profileComm = v != 0;
        else if (k == "send_fraction") {
            if (!(v >= 0 && v <= 1)) return fail(CPF_ERR_ARG, "send_fraction must be in [0, 1]");
            sendFraction = v;
            CPF_SH(finishExchange());
            CPF_SH(resizeSend(std::max<int64_t>(1024, (int64_t)((double)cap * sendFraction))));
        }
        else if (k == "step_index") { if (v < 0 || !whole) return fail(CPF_ERR_ARG, "step_index must be an integer >= 0"); CPF_SH(finishExchange()); stepIndex = (uint32_t)v; }
        else return fail(CPF_ERR_ARG, "cpf_shard_set_option: unknown key '" + k + "'");
        return CPF_OK;
    }

    // ---------------------------------------------------------------------------------------------- filling
    int setParticlesDev(const double* sx, const double* sy, const double* sz, const int32_t* sc, const int64_t* sg, int64_t count,
                        int64_t firstGid) {
        if (count < 0 || (count > 0 && (!sx || !sy || !sz))) return fail(CPF_ERR_ARG, "cpf_shard_set_particles_dev: bad arguments");
        CPF_SH(finishExchange());
        velValid = false;
        if (count > cap) CPF_SH(grow(count, 0));
        Stream s = dev.compute();
        if (count > 0) {
            CPF_SH(dev.copy(x, sx, (size_t)count * 8, s)); CPF_SH(dev.copy(y, sy, (size_t)count * 8, s));
            CPF_SH(dev.copy(z, sz, (size_t)count * 8, s));
            if (sg) CPF_SH(dev.copy(gid, sg, (size_t)count * 8, s)); else CPF_SH(dev.iota(gid, count, firstGid));
            if (sc) CPF_SH(dev.copy(cell, sc, (size_t)count * 4, s)); else CPF_SH(dev.locate(x, y, z, cell, count));
        }
        n = count;
        return CPF_OK;
    }

    int seedBox(int64_t nTotal, const double lower[3], const double upper[3], int order, int64_t* nOutside) {
        if (nTotal < 1 || !lower || !upper) return fail(CPF_ERR_ARG, "cpf_shard_seed_box: bad arguments");
        CPF_SH(finishExchange());
        const int64_t first = nTotal * rank / W, last = nTotal * (rank + 1) / W;      // (nTotal < 2^37, W <= 64: no overflow)
        const int64_t count = last - first;
        velValid = false;
        if (count > cap) CPF_SH(grow(count, 0));
        if (count > 0) {
            CPF_SH(dev.seed(x, y, z, first, count, lower, upper, order));
            CPF_SH(dev.iota(gid, count, first));
            CPF_SH(dev.locate(x, y, z, cell, count));
        }
        n = count;
        {   // (the count is a collective: every rank takes part whether or not it asked for the answer)
            int64_t mine = 0;
            CPF_SH(dev.countNegative(cell, n, &mine));
            double tot = (double)mine;
            CPF_SH(allReduceScalar(&tot));
            if (nOutside) *nOutside = (int64_t)tot;
        }
        if (distOn()) {
            CPF_SH(recut(false));
            CPF_SH(beginExchange());
            CPF_SH(finishExchange());
        }
        return sort();
    }

    // sum of one double over the ranks (W == 1: unchanged)
    int allReduceScalar(double* v) {
        if (!haveComm || W == 1) return CPF_OK;
        Stream s = dev.compute();
        h_scalar[0] = *v;
        CPF_SH(dev.copy(d_scalar, h_scalar, 8, s));
        CPF_SH_COMM(comm.all_reduce_sum_f64(comm.self, d_scalar, 1, s));
        CPF_SH(dev.copy(h_scalar, d_scalar, 8, s));
        CPF_SH(dev.streamSync(s));
        *v = h_scalar[0];
        return CPF_OK;
    }

    int globalCount(int64_t* out) {
        CPF_SH(finishExchange());
        double v = (double)n;
        CPF_SH(allReduceScalar(&v));
        *out = (int64_t)v;
        return CPF_OK;
    }

    // ---------------------------------------------------------------------------------------------- the hot loop
    int step(double dt, double D, int nCycles, unsigned flags) {
        if (nCycles < 0) return fail(CPF_ERR_ARG, "cpf_shard_step: negative cycle count");
        if (pending.on && haveArgs && !(argDt == dt && argD == D && argFlags == flags))
            CPF_SH(finishExchange());                     // the catch-up replays the window with ONE set of arguments
        argDt = dt; argD = D; argFlags = flags; haveArgs = true;
        velValid = false;
        const bool dist = distOn();
        const bool storeVel = (flags & CPF_STEP_STORE_VEL) != 0, fuse = (flags & CPF_STEP_FUSE_CYCLES) != 0;
        if (storeVel && !vel) CPF_SH(dev.alloc((void**)&vel, (size_t)cap * 24));
        // what fell due on the last cycle of a call that stored velocities runs now (the frame has been written)
        if (deferSort || deferRecut || deferExchange) {
            const bool ds = deferSort, dr = deferRecut, de = deferExchange;
            deferSort = deferRecut = deferExchange = false;
            if (ds) { if (!pending.on) CPF_SH(sort()); else sortDue = true; }
            if (dr) { CPF_SH(finishExchange()); CPF_SH(recut(balanceByTime)); CPF_SH(beginExchange()); }
            else if (de) { CPF_SH(finishExchange()); CPF_SH(beginExchange()); }
        }
        // a cycle of zero length without a kick moves nothing: the frame-0 idiom (velocities of one advect in the first output
        // file, out-of-domain particles frozen; src/initCuda.H:184-201) and not a step of the run -- like cpf_step, the
        // counter-based Brownian stream and the cadences do not see it
        const bool frameZero = dt == 0.0 && D == 0.0;
        for (int c = 0; c < nCycles;) {
            if (pending.on && (int64_t)stepIndex - (int64_t)pending.step >= overlapDepth()) CPF_SH(finishExchange());
            // the cycle whose velocities are kept runs on a settled shard: a hand-off in flight completes BEFORE it (its arrivals,
            // an overflow's second split and a sort that was waiting for it would otherwise reorder the particles under the
            // stored velocities, or leave arrivals without any)
            if (storeVel && c == nCycles - 1 && pending.on) CPF_SH(finishExchange());
            // (a particle that is not stepped -- lost, frozen -- has no velocity in the frame, as with cpf_step; the slots are not
            //  even the ones its last velocity was written to: the shard has been compacted, sorted or grown since)
            if (storeVel && c == nCycles - 1 && n > 0) CPF_SH(dev.fill(vel, 0, (size_t)n * 24, dev.compute()));
            // CPF_STEP_FUSE_CYCLES: the cycles up to the next thing that falls due -- the end of the call, a sort, a hand-off or
            // re-cut, the completion of the hand-off in flight -- run inside ONE launch (U is frozen during the call;
            // bit-identical to single launches)
            int64_t run = 1;
            if (fuse && !frameZero && !storeVel) {
                run = nCycles - c;
                if (pending.on) run = std::min<int64_t>(run, (int64_t)overlapDepth() - ((int64_t)stepIndex - (int64_t)pending.step));
                if (sortInterval) run = std::min<int64_t>(run, sortInterval - (int64_t)(stepIndex % (uint32_t)sortInterval));
                if (dist && rebalanceInterval) run = std::min<int64_t>(run, rebalanceInterval - (int64_t)(stepIndex % (uint32_t)rebalanceInterval));
                if (dist && exchangeInterval) run = std::min<int64_t>(run, exchangeInterval - (int64_t)(stepIndex % (uint32_t)exchangeInterval));
                run = std::max<int64_t>(run, 1);
            }
            CPF_SH(dev.step(x, y, z, cell, gid, storeVel ? vel : nullptr, n, dt, D, stepIndex, (int)run, flags));
            c += (int)run;
            if (frameZero) continue;
            stepIndex += (uint32_t)run;
            particleSteps += n * run;
            ++launchesIssued; cyclesIssued += run;
            const bool hold = storeVel && c == nCycles;               // velocities must stay aligned with the particles
            if (sortInterval && stepIndex % (uint32_t)sortInterval == 0) {
                if (hold) deferSort = true;
                else if (!pending.on) CPF_SH(sort());
                else sortDue = true;                                   // never reorder while stale tail slots are in the range
            }
            if (dist) {
                if (rebalanceInterval && stepIndex % (uint32_t)rebalanceInterval == 0) {
                    if (hold) deferRecut = true;
                    else { CPF_SH(finishExchange()); CPF_SH(recut(balanceByTime)); CPF_SH(beginExchange()); }
                } else if (exchangeInterval && stepIndex % (uint32_t)exchangeInterval == 0) {
                    if (hold) deferExchange = true;
                    else { CPF_SH(finishExchange()); CPF_SH(beginExchange()); }
                }
            }
        }
        if (overlapSteps == 0 || storeVel) CPF_SH(finishExchange());
        // from here the stored velocities line up with the particles -- until something reorders them: an explicit sort,
        // exchange, re-cut or refill clears the flag and a frame written after that carries zero velocities, not misplaced ones
        velValid = storeVel && nCycles > 0;
        return CPF_OK;
    }

    // Cycles the loop runs on between a split and its exchange.  "overlap_steps" >= 0: that many.  -1: what the ranks agreed on
    // in the last hand-off's table (the maximum of their wishes), 2 before the first.
    int overlapDepth() const { return overlapSteps >= 0 ? overlapSteps : depthNext; }

    // This rank's wish for the NEXT hand-off: enough queued cycles to cover the host's own work per hand-off -- everything it
    // does after the one wait: reading the table, enqueueing the all-to-all, the unpack, the catch-up launch -- measured as a
    // running mean, against this rank's step time: ceil(host / step) + 1, at least 2, at most half the hand-off interval.
    int wantedDepth() const {
        if (overlapSteps >= 0) return overlapSteps;
        const double stepMs = (haveCost ? costPerParticle : 1.0) * kCostUnitMs * (double)std::max<int64_t>(n, 1);
        const int interval = rebalanceInterval ? rebalanceInterval : (exchangeInterval ? exchangeInterval : 16);
        const int want = (int)std::ceil(hostWorkMs / std::max(stepMs, 1e-3)) + 1;
        return std::min(std::max(want, 2), std::max(1, interval / 2));
    }

    int exchange() {
        if (!distOn()) return CPF_OK;
        CPF_SH(finishExchange()); CPF_SH(beginExchange());
        return finishExchange();
    }
    int rebalance() {
        if (!distOn()) return CPF_OK;
        CPF_SH(finishExchange()); CPF_SH(recut(balanceByTime)); CPF_SH(beginExchange());
        return finishExchange();
    }

    // Split the shard on the compute stream: leavers into the send buffer, stayers compacted into [0, nStay), the stale tail
    // marked inactive.  Counts and nStay stay in device memory for now.
    int beginExchange() {
        const double t0 = nowMs();
        velValid = false;                                      // the split compacts the stayers: `vel` does not follow
        CPF_SH(dev.pack(x, y, z, cell, gid, n, d_cellLo, W, rank, sendbuf, sendCap, d_meta, d_meta + W));
        CPF_SH(dev.eventRecord(evPack, dev.compute()));
        pending.on = true; pending.step = stepIndex;
        handoffHostMs += nowMs() - t0;
        return CPF_OK;
    }

    // Counts all-gather + payload all-to-all-v on the side stream (the compute stream keeps running the cycles queued since
    // the split), then append the arrivals and let them catch up on those cycles.
    //
    // One host synchronisation, on the side stream only: the rows of every rank are all-gathered on the device and copied to
    // the host once, which gives this rank both its send sizes (its row) and its receive sizes (its column).
    //
    // Send-buffer overflow cannot lose particles or hang the job: a split whose leavers do not fit reports nStay < 0 and
    // moves NOTHING (cpf_pack_leavers_dev).  Every rank reads that in the same table, so all of them take the same branch: the
    // overflowing ranks enlarge their send buffers and split again (on the compute stream, i.e. at the current step -- the same
    // step on every rank, see depthNext -- so their leavers need no catch-up), everybody repeats the all-gather, and only then
    // does the all-to-all run.
    int finishExchange() {
        if (!pending.on) return CPF_OK;
        const double t0 = nowMs(), wait0 = handoffWaitMs;
        pending.on = false;
        const uint32_t splitStep = pending.step;
        Stream side = dev.side(), compute = dev.compute();
        CPF_SH(dev.streamWait(side, evPack));
        Event ev0 = Event(), ev1 = Event();
        if (profileComm) {
            CPF_SH(dev.eventCreate(&ev0, true)); CPF_SH(dev.eventCreate(&ev1, true));
            CPF_SH(dev.eventRecord(ev0, side));
        }
        std::vector<char> repacked((size_t)W, 0);
        int attempts = 0;
        for (;;) {
            *h_want = (int64_t)wantedDepth();
            CPF_SH(dev.copy(d_meta + W + 1, h_want, 8, side));
            if (haveComm) CPF_SH_COMM(comm.all_gather(comm.self, d_meta, d_table, (size_t)RW * 8, side));
            else CPF_SH(dev.copy(d_table, d_meta, (size_t)RW * 8, side));
            CPF_SH(dev.copy(h_table, d_table, (size_t)W * RW * 8, side));
            CPF_SH(dev.copy(h_cellLo, d_cellLo, (size_t)(W + 1) * 4, side));
            CPF_SH(dev.eventRecord(evGot, side));
            const double tw = nowMs();
            CPF_SH(dev.eventSync(evGot));                              // the hand-off's one host wait (side stream only)
            handoffWaitMs += nowMs() - tw;
            bool anyOver = false, meOver = false;
            for (int r = 0; r < W; ++r)
                if (h_table[(size_t)r * RW + W] < 0) { anyOver = true; repacked[(size_t)r] = 1; if (r == rank) meOver = true; }
            if (!anyOver) break;
            if (++attempts > 8) return fail(CPF_ERR_STATE, "hand-off: send buffers still overflow after 8 enlargements");   // same table, same answer on every rank
            if (meOver) {
                // The repeated split runs at the CURRENT step: the leavers have kept accumulating since the aborted one (this
                // rank moved nothing in between), so size for that -- and if it still does not fit, the loop grows again.
                int64_t need = 0;
                for (int r = 0; r < W; ++r) need += h_table[(size_t)rank * RW + r];
                const int64_t missed = std::max<int64_t>(0, (int64_t)stepIndex - (int64_t)splitStep);
                const int64_t g = attempts == 1 ? need * (1 + missed) : std::max(need, sendCap) * 2;
                const int64_t want = std::min(std::max(g + g / 8 + 1024, sendCap), cap);
                CPF_SH(resizeSend(want));
                ++sendGrown;
                CPF_SH(dev.pack(x, y, z, cell, gid, n, d_cellLo, W, rank, sendbuf, sendCap, d_meta, d_meta + W));   // compute stream: after the cycles queued so far
                CPF_SH(dev.eventRecord(evRepack, compute));
                CPF_SH(dev.streamWait(side, evRepack));
            }
        }
        std::memcpy(cellLo.data(), h_cellLo, (size_t)(W + 1) * 4);
        std::vector<int64_t> sOff((size_t)W), sBytes((size_t)W), rOff((size_t)W), rBytes((size_t)W);
        int64_t nSend = 0, nRecv = 0, depth = 0;
        constexpr int64_t kRec = CPF_HANDOFF_DOUBLES * 8;
        for (int r = 0; r < W; ++r) {
            const int64_t s = h_table[(size_t)rank * RW + r], q = h_table[(size_t)r * RW + rank];
            sOff[(size_t)r] = nSend * kRec; sBytes[(size_t)r] = s * kRec; nSend += s;
            rOff[(size_t)r] = nRecv * kRec; rBytes[(size_t)r] = q * kRec; nRecv += q;
            depth = std::max(depth, h_table[(size_t)r * RW + W + 1]);
        }
        if (overlapSteps < 0) depthNext = (int)std::max<int64_t>(1, depth);
        const int64_t nStay = h_table[(size_t)rank * RW + W];
        if (nSend > sendCap) return fail(CPF_ERR_STATE, "hand-off: more leavers than the send buffer holds");
        CPF_SH(ensureRecv(nRecv));
        if (haveComm) CPF_SH_COMM(comm.all_to_all_v(comm.self, sendbuf, sOff.data(), sBytes.data(), recvbuf, rOff.data(), rBytes.data(), side));
        else if (nSend > 0) CPF_SH(dev.copy(recvbuf, sendbuf, (size_t)(nSend * kRec), side));      // one rank, "force_collectives"
        if (profileComm) { CPF_SH(dev.eventRecord(ev1, side)); commEvents.emplace_back(ev0, ev1); }
        CPF_SH(dev.eventRecord(evDone, side));
        CPF_SH(dev.streamWait(compute, evDone));                       // also orders the next split after this all-to-all
        const int64_t missed = (int64_t)stepIndex - (int64_t)splitStep;
        if (!repacked[(size_t)rank]) particleSteps -= (n - nStay) * missed;      // the inactive tail was not real work
        if (nStay + nRecv > cap) CPF_SH(grow(nStay + nRecv, nStay));
        CPF_SH(dev.unpack(x, y, z, cell, gid, nStay, recvbuf, nRecv));
        if (missed > 0 && nRecv > 0) {
            // arrivals sit in source-rank order; those from a rank that split again at the current step are current, the
            // others replay the cycles they missed -- one launch per run of consecutive sources
            int64_t first = nStay, runFirst = nStay, runCount = 0;
            for (int src = 0; src < W; ++src) {
                const int64_t k = h_table[(size_t)src * RW + rank];
                if (repacked[(size_t)src]) {
                    if (runCount) CPF_SH(stepSlice(runFirst, runCount, splitStep, (int)missed));
                    runFirst = first + k; runCount = 0;
                } else runCount += k;
                first += k;
            }
            if (runCount) CPF_SH(stepSlice(runFirst, runCount, splitStep, (int)missed));
        }
        n = nStay + nRecv;
        handedOff += nSend;
        ++exchanges;
        if (sortDue) { sortDue = false; CPF_SH(sort()); }
        const double total = nowMs() - t0;
        handoffHostMs += total;
        hostWorkMs = 0.5 * hostWorkMs + 0.5 * std::max(0.0, total - (handoffWaitMs - wait0));
        return CPF_OK;
    }

    // Steps only particles [first, first + count) for nCycles in one fused launch: arrivals catching up on the cycles they
    // missed while in flight.  Not a timed launch (keeps the balancer's per-launch times clean).
    int stepSlice(int64_t first, int64_t count, uint32_t step0, int nCycles) {
        const bool timing = dev.timingEnabled();
        if (timing) CPF_SH(dev.timingEnable(false));
        const bool storeVel = (argFlags & CPF_STEP_STORE_VEL) != 0 && vel != nullptr;
        const unsigned fl = (storeVel ? argFlags : (argFlags & ~CPF_STEP_STORE_VEL)) | CPF_STEP_FUSE_CYCLES;
        const int r = dev.step(x + first, y + first, z + first, cell + first, gid + first, storeVel ? vel + 3 * first : nullptr, count, argDt,
                               argD, step0, nCycles, fl);
        if (timing) CPF_SH(dev.timingEnable(true));
        if (r != CPF_OK) return note(r);
        particleSteps += count * nCycles;
        return CPF_OK;
    }

    // Re-cut the cell ranges so that every rank owns the same number of particles (or the same measured cost): per-cell
    // histogram (HIP kernel) -> all-reduce -> prefix sums and cut search in one kernel (no host round trip).  Legal at any time
    // because the mesh is replicated.  For a cloud that drifts with the flow the equal-count cuts drift with it, so re-cutting
    // hands over far fewer particles than keeping the ranges fixed would (and nothing piles up on the outlet rank).
    int recut(bool byTime) {
        const double t0 = nowMs();
        if (!d_weights) CPF_SH(dev.alloc((void**)&d_weights, (size_t)nCells * 8));
        // equal-COST cuts when balancing by time: every rank scales its counts by its measured ms per particle-step (hops per
        // step differ across the mesh: fine cells cost more), so the all-reduced histogram is a cost density
        double scale = 1.0;
        if (byTime) CPF_SH(measuredCost(&scale));
        CPF_SH(dev.histogram(cell, n, scale, d_weights));
        if (haveComm) CPF_SH_COMM(comm.all_reduce_sum_f64(comm.self, d_weights, (size_t)nCells, dev.compute()));
        CPF_SH(dev.ranges(d_weights, W, d_cellLo));
        ++rebalances;
        handoffHostMs += nowMs() - t0;
        return CPF_OK;
    }

    // This rank's cost per particle-step in units of kCostUnitMs, from the HIP-event times of its own step launches.  Never
    // stalls the launch queue except for the very first measurement; smoothed 50/50 with the previous value; clamped: one
    // rank's bad measurement must not pull most of the cloud onto another rank.
    int measuredCost(double* out) {
        const bool first = !haveCost;
        int64_t launches = 0; double ms = 0.0;
        // (the one wait: a re-cut before any step has run finds nothing to wait for and must not make the next one wait again)
        const bool wait = first && !costWaited && stepIndex > 0;
        if (wait) costWaited = true;
        CPF_SH(dev.timingRead(wait, &launches, &ms));
        kernelMs += ms; kernelLaunches += launches;
        // (a launch may hold several cycles: the mean number per launch since the last reading)
        const double cyclesPerLaunch = launchesIssued > 0 ? (double)cyclesIssued / (double)launchesIssued : 1.0;
        launchesIssued = 0; cyclesIssued = 0;
        if (launches > 0 && n > 0) {
            const double c = std::min(4.0, std::max(0.25, ms / ((double)launches * cyclesPerLaunch) / (double)n / kCostUnitMs));
            costPerParticle = first ? c : 0.5 * (costPerParticle + c);
            haveCost = true;
        }
        *out = haveCost ? costPerParticle : 1.0;
        return CPF_OK;
    }

    // Into the shard's second set of arrays, which then swap roles with the first (no staging, no copy back).
    int sort() {
        if (n <= 1) return CPF_OK;
        velValid = false;                                      // x, y, z, cell, gid are permuted, `vel` is not
        if (!ax) {
            CPF_SH(dev.alloc((void**)&ax, (size_t)cap * 8)); CPF_SH(dev.alloc((void**)&ay, (size_t)cap * 8));
            CPF_SH(dev.alloc((void**)&az, (size_t)cap * 8)); CPF_SH(dev.alloc((void**)&acell, (size_t)cap * 4));
            CPF_SH(dev.alloc((void**)&agid, (size_t)cap * 8));
            CPF_SH(dev.fill(acell, 0xFF, (size_t)cap * 4, dev.compute()));
        }
        CPF_SH(dev.sortTo(x, y, z, cell, gid, ax, ay, az, acell, agid, n));
        std::swap(x, ax); std::swap(y, ay); std::swap(z, az); std::swap(cell, acell); std::swap(gid, agid);
        return CPF_OK;
    }

    // ---------------------------------------------------------------------------------------------- velocity
    int setVelocity(const double* U, int64_t cells) {
        CPF_SH(finishExchange());      // its arrivals replay the cycles they missed with the field those were stepped with
        return note(dev.setVelocityHost(U, cells));
    }

    int setVelocitySlice(const double* Uslice, int64_t nLocal) {
        if (nLocal < 0 || (nLocal > 0 && !Uslice)) return fail(CPF_ERR_ARG, "cpf_shard_set_velocity_slice: bad arguments");
        CPF_SH(finishExchange());
        Stream s = dev.compute();
        if (pieceCells.empty() || pieceCells[(size_t)rank] != nLocal) {
            // once: how many cells every rank's piece has (global cell id = cells of the lower ranks + local id)
            h_scalar[0] = (double)nLocal;
            CPF_SH(dev.copy(d_scalar, h_scalar, 8, s));
            if (haveComm) CPF_SH_COMM(comm.all_gather(comm.self, d_scalar, d_scalar + 1, 8, s));
            else CPF_SH(dev.copy(d_scalar + 1, d_scalar, 8, s));
            CPF_SH(dev.copy(h_scalar + 1, d_scalar + 1, (size_t)W * 8, s));
            CPF_SH(dev.streamSync(s));
            pieceCells.assign((size_t)W, 0);
            int64_t total = 0;
            for (int r = 0; r < W; ++r) { pieceCells[(size_t)r] = (int64_t)h_scalar[1 + r]; total += pieceCells[(size_t)r]; }
            if (total != nCells) { pieceCells.clear(); return fail(CPF_ERR_ARG, "cpf_shard_set_velocity_slice: the ranks' slices do not add up to the mesh's cells"); }
        }
        if (!d_Ufull) CPF_SH(dev.alloc((void**)&d_Ufull, (size_t)nCells * 24));
        std::vector<int64_t> sOff((size_t)W), sBytes((size_t)W), rOff((size_t)W), rBytes((size_t)W);
        int64_t off = 0, mine = 0;
        for (int r = 0; r < W; ++r) { rOff[(size_t)r] = off * 24; rBytes[(size_t)r] = pieceCells[(size_t)r] * 24; if (r == rank) mine = off; off += pieceCells[(size_t)r]; }
        for (int r = 0; r < W; ++r) { sOff[(size_t)r] = mine * 24; sBytes[(size_t)r] = nLocal * 24; }
        sBytes[(size_t)rank] = rBytes[(size_t)rank] = 0;                           // this rank's slice is already in place
        if (nLocal > 0) CPF_SH(dev.copy(d_Ufull + 3 * mine, Uslice, (size_t)nLocal * 24, s));
        CPF_SH(dev.streamSync(s));                                                  // (Uslice may be pageable host memory of the caller)
        if (haveComm && W > 1) CPF_SH_COMM(comm.all_to_all_v(comm.self, d_Ufull, sOff.data(), sBytes.data(), d_Ufull, rOff.data(), rBytes.data(), s));
        return note(dev.setVelocityDev(d_Ufull, nCells));
    }

    // ---------------------------------------------------------------------------------------------- inspection / output
    int getLocal(int64_t* hg, double* hx, double* hy, double* hz, int32_t* hc) {
        CPF_SH(finishExchange());
        Stream s = dev.compute();
        if (n > 0) {
            if (hg) CPF_SH(dev.copy(hg, gid, (size_t)n * 8, s));
            if (hx) CPF_SH(dev.copy(hx, x, (size_t)n * 8, s));
            if (hy) CPF_SH(dev.copy(hy, y, (size_t)n * 8, s));
            if (hz) CPF_SH(dev.copy(hz, z, (size_t)n * 8, s));
            if (hc) CPF_SH(dev.copy(hc, cell, (size_t)n * 4, s));
        }
        return note(dev.streamSync(s));
    }

    // the whole cloud in particle-id order on `root`; nGlobalOut (nullable) = its size, on every rank
    int gather(int root, double* xyzw, int32_t* cellOut, double* velOut, int64_t* nGlobalOut) {
        if (root < 0 || root >= W) return fail(CPF_ERR_ARG, "cpf_shard_gather: root out of range");
        CPF_SH(finishExchange());
        Stream s = dev.compute();
        h_scalar[0] = (double)n;
        CPF_SH(dev.copy(d_scalar, h_scalar, 8, s));
        if (haveComm) CPF_SH_COMM(comm.all_gather(comm.self, d_scalar, d_scalar + 1, 8, s));
        else CPF_SH(dev.copy(d_scalar + 1, d_scalar, 8, s));
        CPF_SH(dev.copy(h_scalar + 1, d_scalar + 1, (size_t)W * 8, s));
        CPF_SH(dev.streamSync(s));
        std::vector<int64_t> counts((size_t)W);
        int64_t total = 0;
        for (int r = 0; r < W; ++r) { counts[(size_t)r] = (int64_t)h_scalar[1 + r]; total += counts[(size_t)r]; }
        if (nGlobalOut) *nGlobalOut = total;
        constexpr int64_t kRec = 8 * 8;                                             // kOutputDoubles doubles
        double *out = nullptr, *in = nullptr, *dXyzw = nullptr, *dVel = nullptr; int32_t* dCell = nullptr; int64_t* dBad = nullptr;
        int rc = CPF_OK;
        auto cleanup = [&] { for (void* p : {(void*)out, (void*)in, (void*)dXyzw, (void*)dVel, (void*)dCell, (void*)dBad}) if (p) dev.release(p); };
#define CPF_SHG(call) do { rc = (call); if (rc != CPF_OK) { (void)note(rc); cleanup(); return rc; } } while (0)
        CPF_SHG(dev.alloc((void**)&out, (size_t)std::max<int64_t>(n, 1) * kRec));
        CPF_SHG(dev.packOutput(x, y, z, cell, gid, velValid ? vel : nullptr, out, n));
        std::vector<int64_t> sOff((size_t)W, 0), sBytes((size_t)W, 0), rOff((size_t)W, 0), rBytes((size_t)W, 0);
        sBytes[(size_t)root] = n * kRec;
        if (rank == root) {
            CPF_SHG(dev.alloc((void**)&in, (size_t)std::max<int64_t>(total, 1) * kRec));
            int64_t off = 0;
            for (int r = 0; r < W; ++r) { rOff[(size_t)r] = off * kRec; rBytes[(size_t)r] = counts[(size_t)r] * kRec; off += counts[(size_t)r]; }
        }
        if (haveComm) {
            rc = comm.all_to_all_v(comm.self, out, sOff.data(), sBytes.data(), in, rOff.data(), rBytes.data(), s);
            if (rc != CPF_OK) { (void)commError(rc, "all_to_all_v (gather)"); cleanup(); return rc; }
        } else if (n > 0) CPF_SHG(dev.copy(in, out, (size_t)(n * kRec), s));
        if (rank == root && total > 0) {
            if (xyzw) CPF_SHG(dev.alloc((void**)&dXyzw, (size_t)total * 32));
            if (velOut) CPF_SHG(dev.alloc((void**)&dVel, (size_t)total * 32));
            if (cellOut) CPF_SHG(dev.alloc((void**)&dCell, (size_t)total * 4));
            CPF_SHG(dev.alloc((void**)&dBad, 8));
            CPF_SHG(dev.fill(dBad, 0, 8, s));
            CPF_SHG(dev.scatterOutput(in, total, total, dXyzw, dCell, dVel, dBad));
            int64_t bad = 0;
            if (xyzw) CPF_SHG(dev.copy(xyzw, dXyzw, (size_t)total * 32, s));
            if (velOut) CPF_SHG(dev.copy(velOut, dVel, (size_t)total * 32, s));
            if (cellOut) CPF_SHG(dev.copy(cellOut, dCell, (size_t)total * 4, s));
            CPF_SHG(dev.copy(&bad, dBad, 8, s));
            CPF_SHG(dev.streamSync(s));
            if (bad != 0) { cleanup(); return fail(CPF_ERR_STATE, "cpf_shard_gather: particle ids are not 0 .. nGlobal-1"); }
        } else CPF_SHG(dev.streamSync(s));
#undef CPF_SHG
        cleanup();
        return CPF_OK;
    }

    // COLLECTIVE: the cloud is gathered to the root's DEVICE (pack, counts all-gather, all-to-all-v, scatter into particle-id
    // order -- all on the compute stream, in buffers that live as long as the shard), an event is recorded, and the call returns:
    // the copy to pinned host memory runs on the io stream behind that event, and the root's worker thread sums the energy,
    // formats and writes (1e5 particles: 0.1 s of formatting against milliseconds of GPU time between two frames).  One frame is
    // in flight; the next call (or cpf_shard_write_vtu_wait / cpf_shard_destroy) waits for it and reports its failure.  totalKE
    // non-null: the root waits for the copy and the sum (a host that prints the energy where the reference does).
    int writeVtu(int root, const char* path, double* totalKE) {
        if (!path) return fail(CPF_ERR_ARG, "cpf_shard_write_vtu: null path");
        if (root < 0 || root >= W) return fail(CPF_ERR_ARG, "cpf_shard_write_vtu: root out of range");
        const int prev = writerWait();                    // (its buffers are free again)
        if (totalKE) *totalKE = 0.0;
        CPF_SH(finishExchange());
        Stream s = dev.compute();
        // how many particles every rank holds: one small all-gather and the call's only wait for the compute stream
        h_scalar[0] = (double)n;
        CPF_SH(dev.copy(d_scalar, h_scalar, 8, s));
        if (haveComm) CPF_SH_COMM(comm.all_gather(comm.self, d_scalar, d_scalar + 1, 8, s));
        else CPF_SH(dev.copy(d_scalar + 1, d_scalar, 8, s));
        CPF_SH(dev.copy(h_scalar + 1, d_scalar + 1, (size_t)W * 8, s));
        CPF_SH(dev.streamSync(s));
        std::vector<int64_t> counts((size_t)W);
        int64_t total = 0;
        for (int r = 0; r < W; ++r) { counts[(size_t)r] = (int64_t)h_scalar[1 + r]; total += counts[(size_t)r]; }
        constexpr int64_t kRec = 8 * 8;                   // kOutputDoubles doubles
        auto room = [&](void** p, int64_t& cap, int64_t want, size_t unit) -> int {
            if (want <= cap && *p) return CPF_OK;
            if (*p) { dev.release(*p); *p = nullptr; }
            cap = 0;
            const int64_t c = std::max<int64_t>(want + want / 8, 1);
            CPF_SH(dev.alloc(p, (size_t)c * unit));
            cap = c;
            return CPF_OK;
        };
        CPF_SH(room((void**)&fOut, fOutCap, n, (size_t)kRec));
        CPF_SH(dev.packOutput(x, y, z, cell, gid, velValid ? vel : nullptr, fOut, n));
        std::vector<int64_t> sOff((size_t)W, 0), sBytes((size_t)W, 0), rOff((size_t)W, 0), rBytes((size_t)W, 0);
        sBytes[(size_t)root] = n * kRec;
        if (rank == root) {
            if (total > fInCap || !fIn) {
                for (void** q : {(void**)&fIn, (void**)&fXyzw, (void**)&fVel, (void**)&fCell})
                    if (*q) { dev.release(*q); *q = nullptr; }
                fInCap = 0;
                const int64_t c = std::max<int64_t>(total + total / 8, 1);
                CPF_SH(dev.alloc((void**)&fIn, (size_t)c * kRec)); CPF_SH(dev.alloc((void**)&fXyzw, (size_t)c * 32));
                CPF_SH(dev.alloc((void**)&fVel, (size_t)c * 32)); CPF_SH(dev.alloc((void**)&fCell, (size_t)c * 4));
                fInCap = c;
            }
            if (!fBad) CPF_SH(dev.alloc((void**)&fBad, 8));
            int64_t off = 0;
            for (int r = 0; r < W; ++r) { rOff[(size_t)r] = off * kRec; rBytes[(size_t)r] = counts[(size_t)r] * kRec; off += counts[(size_t)r]; }
        }
        if (haveComm) CPF_SH_COMM(comm.all_to_all_v(comm.self, fOut, sOff.data(), sBytes.data(), fIn, rOff.data(), rBytes.data(), s));
        else if (n > 0) CPF_SH(dev.copy(fIn, fOut, (size_t)(n * kRec), s));
        if (rank != root) {
            if (prev != CPF_OK && prev != CPF_WARN_NAN) return fail(prev, "cpf_shard_write_vtu: the previous frame could not be written");
            return CPF_OK;
        }
        // ---- root: into particle-id order on the device, then off to the host behind the caller's back
        const size_t offV = ((size_t)total * 32 + 255) & ~(size_t)255, offC = offV + (((size_t)total * 32 + 255) & ~(size_t)255);
        const size_t offB = offC + (((size_t)total * 4 + 255) & ~(size_t)255), needHost = offB + 8;
        if (needHost > fHostBytes) {
            if (fHost) { dev.hostRelease(fHost); fHost = nullptr; }
            fHostBytes = 0;
            CPF_SH(dev.hostAlloc((void**)&fHost, needHost + needHost / 8));
            fHostBytes = needHost + needHost / 8;
        }
        if (evFrame == Event()) { CPF_SH(dev.eventCreate(&evFrame, false)); CPF_SH(dev.eventCreate(&evFrameCopied, false)); }
        CPF_SH(dev.fill(fBad, 0, 8, s));
        if (total > 0) CPF_SH(dev.scatterOutput(fIn, total, total, fXyzw, fCell, fVel, fBad));
        CPF_SH(dev.eventRecord(evFrame, s));
        Stream io = dev.io();
        CPF_SH(dev.streamWait(io, evFrame));
        CPF_SH(dev.copy(fHost, fXyzw, (size_t)total * 32, io)); CPF_SH(dev.copy(fHost + offV, fVel, (size_t)total * 32, io));
        CPF_SH(dev.copy(fHost + offC, fCell, (size_t)total * 4, io)); CPF_SH(dev.copy(fHost + offB, fBad, 8, io));
        CPF_SH(dev.eventRecord(evFrameCopied, io));
        const std::string file(path);
        writerLive = true;
        keReady = false;
        writer = std::thread([this, file, total, offV, offC, offB] {
            int st = dev.bindThread();
            if (st == CPF_OK) st = dev.eventSyncQuiet(evFrameCopied);
            const double* xyzw = (const double*)fHost; const double* v = (const double*)(fHost + offV);
            double ke = 0.0;                              // in index order, like the reference's running sum
            if (st == CPF_OK)
                for (int64_t i = 0; i < total; ++i) ke += 0.5 * (v[4 * i] * v[4 * i] + v[4 * i + 1] * v[4 * i + 1] + v[4 * i + 2] * v[4 * i + 2]);
            { std::lock_guard<std::mutex> lk(keMutex); frameKE = ke; keReady = true; }
            keCv.notify_all();
            int64_t bad = 0;
            if (st == CPF_OK) std::memcpy(&bad, fHost + offB, 8);
            if (st == CPF_OK && bad != 0) st = CPF_ERR_STATE;        // particle ids are not 0 .. nGlobal-1: no frame
            writerStatus = st != CPF_OK ? st : dev.writeVtuArrays(file.c_str(), total, xyzw, (const int32_t*)(fHost + offC), v, nullptr);
        });
        // (the current frame is on its way whatever happened to the previous one; a failure of that one is reported now)
        if (prev != CPF_OK && prev != CPF_WARN_NAN) return fail(prev, "cpf_shard_write_vtu: the previous frame could not be written");
        if (!totalKE) return CPF_OK;
        std::unique_lock<std::mutex> lk(keMutex);
        keCv.wait(lk, [this] { return keReady; });
        *totalKE = frameKE;
        return std::isnan(frameKE) ? CPF_WARN_NAN : CPF_OK;
    }

    void stats(cpf_shard_stats* o) {
        // device time of the hand-off collectives so far ("profile_comm"): completed pairs are read once; never waits
        size_t done = 0;
        for (; done < commEvents.size(); ++done) {
            float ms = 0.f;
            if (!dev.eventDone(commEvents[done].second)) break;
            if (dev.eventElapsed(commEvents[done].first, commEvents[done].second, &ms) != CPF_OK) break;
            commMsRead += (double)ms;
            dev.eventDestroy(commEvents[done].first); dev.eventDestroy(commEvents[done].second);   // (a long run does not pile them up)
        }
        commEvents.erase(commEvents.begin(), commEvents.begin() + (std::ptrdiff_t)done);
        commEventsRead += done;
        o->n = n; o->capacity = cap; o->stepIndex = (int64_t)stepIndex;
        o->particleSteps = particleSteps; o->handedOff = handedOff; o->exchanges = exchanges; o->rebalances = rebalances;
        o->grown = grown; o->sendGrown = sendGrown; o->kernelLaunches = kernelLaunches; o->kernelMs = kernelMs;
        o->handoffHostMs = handoffHostMs; o->handoffWaitMs = handoffWaitMs; o->hostWorkMsPerHandoff = hostWorkMs;
        o->commDeviceMs = commMsRead; o->commEvents = (int64_t)commEventsRead;
        o->overlapDepth = overlapDepth(); o->nRanks = W; o->rank = rank;
    }
#undef CPF_SH
#undef CPF_SH_COMM
};

}  // namespace cpf
