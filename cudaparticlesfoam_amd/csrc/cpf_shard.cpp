// cpf_shard_* (include/cpf.h): the sharded cloud's logic (cpf_shard_core.h) on the HIP device -- the only device the
// product has.  The device interface below is a thin skin over this library's own C-ABI (cpf_step_dev, cpf_pack_leavers_dev,
// ...): the shard layer is a client of the single-GPU layer, on the context's stream plus one side stream for the hand-off's
// collectives.
#include <hip/hip_runtime.h>

#include <new>
#include <string>

#include "cpf.h"
#include "cpf_device.h"
#include "cpf_internal.h"
#include "cpf_shard_core.h"

namespace {

struct HipDev {
    typedef hipStream_t Stream;
    typedef hipEvent_t Event;

    cpf_context* ctx = nullptr;
    int device = 0;
    hipStream_t sideStream = nullptr;
    hipStream_t ioStream = nullptr;                  // device-to-host copies of output frames (made on first use)
    std::string err;

    ~HipDev() {
        if (sideStream) { (void)hipSetDevice(device); (void)hipStreamDestroy(sideStream); }
        if (ioStream) { (void)hipSetDevice(device); (void)hipStreamDestroy(ioStream); }
    }

    int hip(hipError_t e, const char* what) {
        if (e == hipSuccess) return CPF_OK;
        err = std::string(what) + ": " + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? CPF_ERR_NOMEM : CPF_ERR_HIP;
    }
    int api(int r) {
        if (r != CPF_OK) err = cpf_last_error(ctx);
        return r;
    }
    const char* lastError() const { return err.c_str(); }
#define CPF_DH(call) hip((call), #call)

    int bind() { err.clear(); return CPF_DH(hipSetDevice(device)); }
    Stream compute() const { return (hipStream_t)cpf::context_stream(ctx); }     // (read per call: cpf_set_stream may change it)
    Stream side() const { return sideStream; }
    Stream io() {                                    // (on failure: the side stream -- correct, merely shared with the hand-offs)
        if (!ioStream && hipStreamCreateWithFlags(&ioStream, hipStreamNonBlocking) != hipSuccess) ioStream = nullptr;
        return ioStream ? ioStream : sideStream;
    }
    // for the frame's worker thread: no access to `err` (the caller's thread owns it)
    int bindThread() const { return hipSetDevice(device) == hipSuccess ? CPF_OK : CPF_ERR_HIP; }
    int eventSyncQuiet(Event e) const { return hipEventSynchronize(e) == hipSuccess ? CPF_OK : CPF_ERR_HIP; }
    int64_t nCells() const { return cpf::context_cells(ctx); }

    // memory
    int alloc(void** p, size_t bytes) { return CPF_DH(hipMalloc(p, bytes < 16 ? 16 : bytes)); }
    void release(void* p) { (void)hipFree(p); }
    int hostAlloc(void** p, size_t bytes) { return CPF_DH(hipHostMalloc(p, bytes < 16 ? 16 : bytes, hipHostMallocDefault)); }
    void hostRelease(void* p) { (void)hipHostFree(p); }
    int copy(void* dst, const void* src, size_t bytes, Stream s) { return bytes ? CPF_DH(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, s)) : CPF_OK; }
    int fill(void* p, int byte, size_t bytes, Stream s) { return bytes ? CPF_DH(hipMemsetAsync(p, byte, bytes, s)) : CPF_OK; }

    // streams and events
    int eventCreate(Event* e, bool timing) { return CPF_DH(hipEventCreateWithFlags(e, timing ? hipEventDefault : hipEventDisableTiming)); }
    void eventDestroy(Event e) { (void)hipEventDestroy(e); }
    int eventRecord(Event e, Stream s) { return CPF_DH(hipEventRecord(e, s)); }
    int streamWait(Stream s, Event e) { return CPF_DH(hipStreamWaitEvent(s, e, 0)); }
    int eventSync(Event e) { return CPF_DH(hipEventSynchronize(e)); }
    bool eventDone(Event e) { return hipEventQuery(e) == hipSuccess; }
    int eventElapsed(Event a, Event b, float* ms) { return CPF_DH(hipEventElapsedTime(ms, a, b)); }
    int streamSync(Stream s) { return CPF_DH(hipStreamSynchronize(s)); }

    // kernels, on the context's stream
    int step(double* x, double* y, double* z, int32_t* cell, const int64_t* gid, double* vel, int64_t n, double dt, double D,
             uint32_t step0, int nCycles, unsigned flags) {
        return api(cpf_step_dev(ctx, x, y, z, cell, gid, vel, n, dt, D, step0, nCycles, flags));
    }
    int pack(double* x, double* y, double* z, int32_t* cell, int64_t* gid, int64_t n, const int32_t* cellLo, int W, int rank,
             double* sendbuf, int64_t sendCap, int64_t* counts, int64_t* nStay) {
        return api(cpf_pack_leavers_dev(ctx, x, y, z, cell, gid, n, cellLo, W, rank, sendbuf, sendCap, counts, nStay));
    }
    int unpack(double* x, double* y, double* z, int32_t* cell, int64_t* gid, int64_t nStay, const double* recvbuf, int64_t nRecv) {
        return api(cpf_unpack_arrivals_dev(ctx, x, y, z, cell, gid, nStay, recvbuf, nRecv));
    }
    int histogram(const int32_t* cell, int64_t n, double scale, double* weights) { return api(cpf_cell_histogram_dev(ctx, cell, n, scale, weights)); }
    int ranges(const double* weights, int W, int32_t* cellLo) { return api(cpf_cell_ranges_dev(ctx, weights, W, cellLo)); }
    int sortTo(const double* x, const double* y, const double* z, const int32_t* cell, const int64_t* gid, double* ox, double* oy,
               double* oz, int32_t* oc, int64_t* og, int64_t n) {
        return api(cpf_sort_by_cell_dev_to(ctx, x, y, z, cell, gid, ox, oy, oz, oc, og, n));
    }
    int locate(const double* x, const double* y, const double* z, int32_t* cell, int64_t n) { return api(cpf_locate_initial_dev(ctx, x, y, z, cell, n)); }
    int seed(double* x, double* y, double* z, int64_t first, int64_t n, const double lower[3], const double upper[3], int order) {
        return api(cpf_seed_box_dev(ctx, x, y, z, first, n, lower, upper, order));
    }
    int iota(int64_t* gid, int64_t n, int64_t first) { return CPF_DH(cpf::launch_iota64(compute(), gid, n, first)); }
    int countNegative(const int32_t* cell, int64_t n, int64_t* out) { return api(cpf_stage_count_outside(ctx, cell, n, out)); }
    int packOutput(const double* x, const double* y, const double* z, const int32_t* cell, const int64_t* gid, const double* vel3,
                   double* rec, int64_t n) {
        return CPF_DH(cpf::pack_output(compute(), x, y, z, cell, gid, vel3, rec, n));
    }
    int scatterOutput(const double* rec, int64_t nRec, int64_t nGlobal, double* xyzw, int32_t* cellOut, double* velOut, int64_t* bad) {
        return CPF_DH(cpf::scatter_output(compute(), rec, nRec, nGlobal, xyzw, cellOut, velOut, (unsigned long long*)bad));
    }

    // timing of the step launches (the balancer's clock)
    bool timingEnabled() const { return cpf::context_timing(ctx); }
    int timingEnable(bool on) { return api(cpf_timing_enable(ctx, on ? 1 : 0)); }
    int timingRead(bool wait, int64_t* launches, double* ms) {
        return api(wait ? cpf_timing_read(ctx, launches, ms) : cpf_timing_poll(ctx, launches, ms));
    }

    int setVelocityHost(const double* U, int64_t cells) { return api(cpf_set_velocity(ctx, U, cells)); }
    int setVelocityDev(const double* dU, int64_t cells) { return api(cpf_set_velocity_dev(ctx, dU, cells)); }
    int writeVtuArrays(const char* path, int64_t n, const double* xyzw, const int32_t* cell, const double* vel, double* ke) {
        return (cpf::vtu_binary(ctx) ? cpf_write_vtu_arrays_binary : cpf_write_vtu_arrays)(path, n, xyzw, cell, vel, ke);
    }
#undef CPF_DH
};

typedef HipDev CpfShardDev;

bool cpfMakeDev(cpf_context* ctx, HipDev& d, std::string& why) {
    if (!ctx) { why = "null context"; return false; }
    d.ctx = ctx;
    d.device = cpf::context_device(ctx);
    hipError_t e = hipSetDevice(d.device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&d.sideStream, hipStreamNonBlocking);
    if (e != hipSuccess) { why = std::string("side stream: ") + hipGetErrorString(e); return false; }
    return true;
}

}  // namespace

#include "cpf_shard_abi.inc"
