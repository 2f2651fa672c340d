// TEST INFRASTRUCTURE ONLY.  The sharded cloud's host logic (cudaparticlesfoam_amd/csrc/cpf_shard_core.h -- the very
// source the product compiles for HIP) instantiated on a HOST-MEMORY stand-in for the device, so that the gloo tests
// (world size 2 / 3 / 8, no GPU) drive the product's own hand-off code: "device" memory is malloc, streams and events
// are no-ops, the step and the initial locate are the CPU checker (oracle/liboracle_cellwalk.so), the split / histogram
// / cut / sort kernels are the loops below.  Exports the same cpf_shard_* names from tests/host_shard/libcpf_shard_host.so;
// the "context" handed to cpf_shard_create is a cpf_host_case (below).  Nothing in the product links or loads this.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <string>
#include <vector>

#include "cpf.h"
#include "cpf_shard_core.h"

extern "C" {
// oracle/cellwalk.c
void cw_step(double* px, double* py, double* pz, int* cell, double* vel_out, int n, double dt, int cycles, const int* cellOff,
             const double* planes, const int* nbr, const int* groupOff, const int* groupNbr, const double* U, int nthreads,
             long long* stats, double D, const int64_t* gid, uint32_t step0, uint32_t seed);
void cw_locate_initial(const double* px, const double* py, const double* pz, int* cell, int n, int nCells, const int* cellOff,
                       const double* planes, int nthreads);
}

struct cpf_host_case {
    std::vector<int32_t> cellOff, nbr, groupOff, groupNbr;
    std::vector<double> planes, U;
    int64_t nCells = 0;
    double fakeMsPerParticle = 2.0e-8;     // pretend step cost of THIS rank (load-balancer tests)
    uint32_t seed = 0;
    bool timing = false;
    int64_t launches = 0;
    int64_t stepLaunches = 0;      // every step launch ever (the fused-cycles test counts them)
    double ms = 0.0;
};

namespace cpf_host_test {

struct HostDev {
    typedef void* Stream;
    typedef int Event;
    cpf_host_case* hc = nullptr;
    std::string err;

    const char* lastError() const { return err.c_str(); }
    int bind() { err.clear(); return CPF_OK; }
    Stream compute() const { return nullptr; }
    Stream side() const { return nullptr; }
    Stream io() { return nullptr; }
    int bindThread() const { return CPF_OK; }
    int eventSyncQuiet(Event) const { return CPF_OK; }
    int64_t nCells() const { return hc->nCells; }

    int alloc(void** p, size_t bytes) { *p = std::calloc(std::max<size_t>(bytes, 16), 1); return *p ? CPF_OK : CPF_ERR_NOMEM; }
    void release(void* p) { std::free(p); }
    int hostAlloc(void** p, size_t bytes) { return alloc(p, bytes); }
    void hostRelease(void* p) { std::free(p); }
    int copy(void* dst, const void* src, size_t bytes, Stream) { if (bytes) std::memmove(dst, src, bytes); return CPF_OK; }
    int fill(void* p, int byte, size_t bytes, Stream) { if (bytes) std::memset(p, byte, bytes); return CPF_OK; }
    int eventCreate(Event* e, bool) { *e = 1; return CPF_OK; }
    void eventDestroy(Event) {}
    int eventRecord(Event, Stream) { return CPF_OK; }
    int streamWait(Stream, Event) { return CPF_OK; }
    int eventSync(Event) { return CPF_OK; }
    bool eventDone(Event) { return true; }
    int eventElapsed(Event, Event, float* ms) { *ms = 0.f; return CPF_OK; }
    int streamSync(Stream) { return CPF_OK; }

    int step(double* x, double* y, double* z, int32_t* cell, const int64_t* gid, double* vel, int64_t n, double dt, double D,
             uint32_t step0, int nCycles, unsigned) {
        hc->stepLaunches += 1;
        if (hc->timing) { hc->launches += 1; hc->ms += (double)n * (double)nCycles * hc->fakeMsPerParticle; }   // (a real launch takes as long as its cycles)
        // (the checker writes the velocities of the last cycle as [n][3], the layout the shard keeps)
        cw_step(x, y, z, cell, vel, (int)n, dt, nCycles, hc->cellOff.data(), hc->planes.data(), hc->nbr.data(),
                hc->groupOff.data(), hc->groupNbr.data(), hc->U.data(), 1, nullptr, D, gid, step0, hc->seed);
        return CPF_OK;
    }
    static int ownerOf(int c, const int32_t* lo, int W) { int r = 0; for (int q = 1; q < W; ++q) r += c >= lo[q]; return r; }
    // like the HIP split: leavers grouped by destination in index order; more than the buffer holds: aborted, nothing moved
    int pack(double* x, double* y, double* z, int32_t* cell, int64_t* gid, int64_t n, const int32_t* lo, int W, int rank,
             double* sendbuf, int64_t sendCap, int64_t* counts, int64_t* nStay) {
        std::vector<int> dest((size_t)n);
        std::vector<int64_t> cnt((size_t)W, 0);
        for (int64_t i = 0; i < n; ++i) {
            int d = -1;
            if (cell[i] >= 0) { const int o = ownerOf(cell[i], lo, W); if (o != rank) d = o; }
            dest[(size_t)i] = d;
            if (d >= 0) ++cnt[(size_t)d];
        }
        int64_t total = 0;
        for (int r = 0; r < W; ++r) { counts[r] = cnt[(size_t)r]; total += cnt[(size_t)r]; }
        if (total > sendCap) { *nStay = -1; return CPF_OK; }
        std::vector<int64_t> base((size_t)W, 0);
        for (int r = 1; r < W; ++r) base[(size_t)r] = base[(size_t)r - 1] + cnt[(size_t)r - 1];
        int64_t k = 0;
        for (int64_t i = 0; i < n; ++i) {
            const int d = dest[(size_t)i];
            if (d >= 0) {
                double* rec = sendbuf + base[(size_t)d]++ * CPF_HANDOFF_DOUBLES;
                rec[0] = x[i]; rec[1] = y[i]; rec[2] = z[i]; rec[3] = (double)cell[i]; rec[4] = (double)gid[i];
            } else { x[k] = x[i]; y[k] = y[i]; z[k] = z[i]; cell[k] = cell[i]; gid[k] = gid[i]; ++k; }
        }
        for (int64_t i = k; i < n; ++i) { cell[i] = CPF_CELL_LOST; x[i] = y[i] = z[i] = std::nan(""); }   // stale tail: inactive, never read again
        *nStay = k;
        return CPF_OK;
    }
    int unpack(double* x, double* y, double* z, int32_t* cell, int64_t* gid, int64_t nStay, const double* recvbuf, int64_t nRecv) {
        for (int64_t k = 0; k < nRecv; ++k) {
            const double* rec = recvbuf + k * CPF_HANDOFF_DOUBLES;
            x[nStay + k] = rec[0]; y[nStay + k] = rec[1]; z[nStay + k] = rec[2];
            cell[nStay + k] = (int32_t)rec[3]; gid[nStay + k] = (int64_t)rec[4];
        }
        return CPF_OK;
    }
    int histogram(const int32_t* cell, int64_t n, double scale, double* w) {
        std::vector<int64_t> h((size_t)hc->nCells, 0);
        for (int64_t i = 0; i < n; ++i) if (cell[i] >= 0) ++h[(size_t)cell[i]];
        for (int64_t c = 0; c < hc->nCells; ++c) w[c] = (double)h[(size_t)c] * scale;
        return CPF_OK;
    }
    // the rule of cell_ranges_kernel (cpf_handoff.hip): cut_q = #{ i in 0..nCells : cum0[i] < total*q/W }
    int ranges(const double* w, int W, int32_t* lo) {
        const int64_t nC = hc->nCells;
        double total = 0.0;
        for (int64_t c = 0; c < nC; ++c) total += w[c];
        lo[0] = 0;
        int prev = 0;
        for (int q = 1; q < W; ++q) {
            double run = 0.0; int cnt = 0;
            for (int64_t i = 0; i <= nC; ++i) { if (run < total * (double)q / (double)W) ++cnt; if (i < nC) run += w[i]; }
            prev = std::max(prev, cnt); lo[q] = prev;
        }
        lo[W] = (int32_t)nC;
        return CPF_OK;
    }
    int sortTo(const double* x, const double* y, const double* z, const int32_t* cell, const int64_t* gid, double* ox, double* oy,
               double* oz, int32_t* oc, int64_t* og, int64_t n) {
        std::vector<int64_t> idx((size_t)n);
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) { return (uint32_t)cell[a] < (uint32_t)cell[b]; });
        for (int64_t k = 0; k < n; ++k) { const int64_t i = idx[(size_t)k]; ox[k] = x[i]; oy[k] = y[i]; oz[k] = z[i]; oc[k] = cell[i]; og[k] = gid[i]; }
        return CPF_OK;
    }
    int locate(const double* x, const double* y, const double* z, int32_t* cell, int64_t n) {
        cw_locate_initial(x, y, z, cell, (int)n, (int)hc->nCells, hc->cellOff.data(), hc->planes.data(), 1);
        return CPF_OK;
    }
    int seed(double*, double*, double*, int64_t, int64_t, const double*, const double*, int) {
        err = "the host test device has no seeding kernel";
        return CPF_ERR_STATE;
    }
    int iota(int64_t* gid, int64_t n, int64_t first) { for (int64_t i = 0; i < n; ++i) gid[i] = first + i; return CPF_OK; }
    int countNegative(const int32_t* cell, int64_t n, int64_t* out) { int64_t c = 0; for (int64_t i = 0; i < n; ++i) c += cell[i] < 0; *out = c; return CPF_OK; }
    int packOutput(const double* x, const double* y, const double* z, const int32_t* cell, const int64_t* gid, const double* vel3,
                   double* rec, int64_t n) {
        for (int64_t i = 0; i < n; ++i) {
            double* r = rec + 8 * i;
            r[0] = x[i]; r[1] = y[i]; r[2] = z[i]; r[3] = (double)cell[i]; r[4] = (double)gid[i];
            for (int k = 0; k < 3; ++k) r[5 + k] = vel3 ? vel3[3 * i + k] : 0.0;
        }
        return CPF_OK;
    }
    int scatterOutput(const double* rec, int64_t nRec, int64_t nGlobal, double* xyzw, int32_t* cellOut, double* velOut, int64_t* bad) {
        for (int64_t k = 0; k < nRec; ++k) {
            const double* r = rec + 8 * k;
            const int64_t g = (int64_t)r[4];
            if (g < 0 || g >= nGlobal) { ++*bad; continue; }
            const int32_t c = (int32_t)r[3];
            if (xyzw) { xyzw[4 * g] = r[0]; xyzw[4 * g + 1] = r[1]; xyzw[4 * g + 2] = r[2]; xyzw[4 * g + 3] = c == CPF_CELL_FROZEN ? 0.0 : 1.0; }
            if (cellOut) cellOut[g] = c;
            if (velOut) { velOut[4 * g] = r[5]; velOut[4 * g + 1] = r[6]; velOut[4 * g + 2] = r[7]; velOut[4 * g + 3] = -1.0; }
        }
        return CPF_OK;
    }
    bool timingEnabled() const { return hc->timing; }
    int timingEnable(bool on) { hc->timing = on; return CPF_OK; }
    int timingRead(bool, int64_t* launches, double* ms) { *launches = hc->launches; *ms = hc->ms; hc->launches = 0; hc->ms = 0.0; return CPF_OK; }
    int setVelocityHost(const double* U, int64_t cells) {
        if (cells != hc->nCells) { err = "velocity field of the wrong size"; return CPF_ERR_ARG; }
        hc->U.assign(U, U + 3 * cells);
        return CPF_OK;
    }
    int setVelocityDev(const double* U, int64_t cells) { return setVelocityHost(U, cells); }
    // (the product's own writer, csrc/cpf_io.cpp, compiled into this library: frames of the sharded cloud are compared byte for byte)
    int writeVtuArrays(const char* path, int64_t n, const double* xyzw, const int32_t* cell, const double* vel, double* ke) {
        return cpf_write_vtu_arrays(path, n, xyzw, cell, vel, ke);
    }
};

typedef HostDev CpfShardDev;

bool cpfMakeDev(cpf_context* ctx, HostDev& d, std::string& why) {
    if (!ctx) { why = "null host case"; return false; }
    d.hc = reinterpret_cast<cpf_host_case*>(ctx);
    return true;
}

}  // namespace cpf_host_test
using cpf_host_test::CpfShardDev;
using cpf_host_test::cpfMakeDev;

#include "cpf_shard_abi.inc"

extern "C" {

// the "context" of the host test device: the checker's tables and the velocity field
cpf_host_case* cpf_host_case_create(const int32_t* cellOff, const double* planes, const int32_t* nbr, int64_t nSlots,
                                    const int32_t* groupOff, int64_t nGroups, const int32_t* groupNbr, int64_t nMembers, const double* U,
                                    int64_t nCells, double fakeMsPerParticle) {
    cpf_host_case* h = new cpf_host_case();
    h->cellOff.assign(cellOff, cellOff + nCells + 1);
    h->planes.assign(planes, planes + 4 * nSlots);
    h->nbr.assign(nbr, nbr + nSlots);
    h->groupOff.assign(groupOff, groupOff + nGroups + 1);
    h->groupNbr.assign(groupNbr, groupNbr + std::max<int64_t>(nMembers, 1));
    h->U.assign(U, U + 3 * nCells);
    h->nCells = nCells;
    h->fakeMsPerParticle = fakeMsPerParticle;
    return h;
}
void cpf_host_case_destroy(cpf_host_case* h) { delete h; }
void cpf_host_case_timing(cpf_host_case* h, int on) { h->timing = on != 0; }
int64_t cpf_host_case_step_launches(const cpf_host_case* h) { return h->stepLaunches; }
void cpf_host_case_timing_read(cpf_host_case* h, int64_t* launches, double* ms) { *launches = h->launches; *ms = h->ms; h->launches = 0; h->ms = 0.0; }

}  // extern "C"

// csrc/cpf_io.cpp also holds the context-level writers; what they call of the product is not in this library and never runs here
namespace cpf { bool vtu_binary(const cpf_context*) { return false; } }
extern "C" int cpf_get_particles(cpf_context*, double*, int32_t*, double*) { return CPF_ERR_STATE; }
extern "C" int cpf_num_particles(const cpf_context*, int64_t* n) { if (n) *n = 0; return CPF_ERR_STATE; }
