// Communicators of the sharded cloud (include/cpf.h "cpf_comm"): the three collectives of a hand-off on device memory.
//
//   RCCL        one rank per GPU -- processes or threads -- over xGMI: all-gather, all-reduce, and the all-to-all-v as ONE
//               group of ncclSend / ncclRecv pairs (xGMI is a point-to-point mesh: every pair's traffic takes its own link).
//               librccl.so.1 is opened at run time, so a single-GPU host needs no RCCL at all.
//   in-process  the ranks are threads of this process, possibly on the SAME device (which RCCL refuses): device-to-device
//               copies between the ranks' buffers, ordered by a barrier.  For single-process hosts and for the tests that play
//               several ranks on the one GPU of a development box.
//
// No reference counterpart: the reference's parallel runs gather everything to the MPI master, which drives one GPU
// (src/initCuda.H:207-484, src/advect.H:59-89).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and prototypes only: the library itself is dlopen()ed

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "cpf.h"
#include "cpf_device.h"

namespace {

thread_local std::string t_commError;
int commFail(int code, const std::string& msg) { t_commError = msg; return code; }

// ------------------------------------------------------------------------------------------------------------------
// RCCL, loaded on first use
// ------------------------------------------------------------------------------------------------------------------
struct RcclApi {
    void* lib = nullptr;
    std::string why;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // the soname first: a host that already carries RCCL (a framework's bundled copy) shares that one
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (!api.lib) {
            const char* e = dlerror();                         // (once: reading the error clears it)
            api.why = std::string("librccl.so.1 not found (") + (e ? e : "?") + ")";
            return;
        }
        auto sym = [&](const char* s) -> void* {
            void* p = dlsym(api.lib, s);
            if (!p && api.why.empty()) api.why = std::string("librccl: missing symbol ") + s;
            return p;
        };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.Send = (decltype(api.Send))sym("ncclSend");
        api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    });
    return api;
}

struct RcclComm {
    ncclComm_t comm = nullptr;
    int rank = 0, n = 1, device = 0;
    std::string err;
};

#define CPF_NCCL(c, call)                                                                                   \
    do {                                                                                                    \
        ncclResult_t r__ = (call);                                                                          \
        if (r__ != ncclSuccess) {                                                                           \
            (c)->err = std::string(#call) + ": " + rccl().GetErrorString(r__);                              \
            return CPF_ERR_HIP;                                                                             \
        }                                                                                                   \
    } while (0)

int rcclAllGather(void* self, const void* send, void* recv, size_t bytes, void* stream) {
    RcclComm* c = (RcclComm*)self;
    if (bytes == 0) return CPF_OK;
    CPF_NCCL(c, rccl().AllGather(send, recv, bytes, ncclInt8, c->comm, (hipStream_t)stream));
    return CPF_OK;
}
int rcclAllReduce(void* self, double* buf, size_t count, void* stream) {
    RcclComm* c = (RcclComm*)self;
    if (count == 0) return CPF_OK;
    CPF_NCCL(c, rccl().AllReduce(buf, buf, count, ncclDouble, ncclSum, c->comm, (hipStream_t)stream));
    return CPF_OK;
}
int rcclAllToAllV(void* self, const void* send, const int64_t* sendOff, const int64_t* sendBytes, void* recv,
                  const int64_t* recvOff, const int64_t* recvBytes, void* stream) {
    RcclComm* c = (RcclComm*)self;
    CPF_NCCL(c, rccl().GroupStart());
    ncclResult_t bad = ncclSuccess;
    const char* what = "";
    for (int r = 0; r < c->n && bad == ncclSuccess; ++r) {
        if (sendBytes[r] > 0) {
            bad = rccl().Send((const char*)send + sendOff[r], (size_t)sendBytes[r], ncclInt8, r, c->comm, (hipStream_t)stream);
            what = "ncclSend";
        }
        if (bad == ncclSuccess && recvBytes[r] > 0) {
            bad = rccl().Recv((char*)recv + recvOff[r], (size_t)recvBytes[r], ncclInt8, r, c->comm, (hipStream_t)stream);
            what = "ncclRecv";
        }
    }
    const ncclResult_t end = rccl().GroupEnd();               // (always: a group left open would swallow every later call)
    if (bad != ncclSuccess) { c->err = std::string(what) + ": " + rccl().GetErrorString(bad); return CPF_ERR_HIP; }
    if (end != ncclSuccess) { c->err = std::string("ncclGroupEnd: ") + rccl().GetErrorString(end); return CPF_ERR_HIP; }
    return CPF_OK;
}
void rcclDestroy(void* self) {
    RcclComm* c = (RcclComm*)self;
    if (c->comm) { (void)hipSetDevice(c->device); (void)rccl().CommDestroy(c->comm); }
    delete c;
}
const char* rcclLastError(void* self) { return ((RcclComm*)self)->err.c_str(); }

// ------------------------------------------------------------------------------------------------------------------
// in-process: ranks are threads
// ------------------------------------------------------------------------------------------------------------------
constexpr char kLocalMagic[] = "CPF-INPROCESS";

struct LocalGroup {
    int n = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0, joined = 0, left = 0;
    uint64_t gen = 0;
    bool broken = false;
    std::vector<const void*> sendPtr;
    std::vector<const int64_t*> sendOff, sendBytes;
    double timeoutS = 300.0;

    // false: another rank never arrived (or the group was broken by one that failed) -- every waiter gets the same answer
    bool barrier() {
        std::unique_lock<std::mutex> lk(m);
        if (broken) return false;
        const uint64_t g = gen;
        if (++arrived == n) { arrived = 0; ++gen; cv.notify_all(); return true; }
        const bool ok = cv.wait_for(lk, std::chrono::duration<double>(timeoutS), [&] { return gen != g || broken; });
        if (!ok || broken) { broken = true; cv.notify_all(); return false; }
        return true;
    }
    void breakGroup() { std::lock_guard<std::mutex> lk(m); broken = true; cv.notify_all(); }
};

std::mutex g_groupsMutex;
std::map<std::string, std::shared_ptr<LocalGroup>> g_groups;
std::atomic<uint64_t> g_groupSerial{1};

struct LocalComm {
    std::shared_ptr<LocalGroup> g;
    std::string key;
    int rank = 0, device = 0;
    double* scratch = nullptr;
    size_t scratchBytes = 0;
    std::string err;
};

#define CPF_LHIP(c, call)                                                                                   \
    do {                                                                                                    \
        hipError_t e__ = (call);                                                                            \
        if (e__ != hipSuccess) {                                                                            \
            (c)->err = std::string(#call) + ": " + hipGetErrorString(e__);                                  \
            (c)->g->breakGroup();                                                                           \
            return CPF_ERR_HIP;                                                                             \
        }                                                                                                   \
    } while (0)
#define CPF_LBARRIER(c)                                                                                     \
    do {                                                                                                    \
        if (!(c)->g->barrier()) {                                                                           \
            if ((c)->err.empty()) (c)->err = "in-process communicator: a rank did not arrive (it failed, or timed out)"; \
            return CPF_ERR_STATE;                                                                           \
        }                                                                                                   \
    } while (0)

int localAllGather(void* self, const void* send, void* recv, size_t bytes, void* stream) {
    LocalComm* c = (LocalComm*)self;
    hipStream_t st = (hipStream_t)stream;
    CPF_LHIP(c, hipSetDevice(c->device));
    CPF_LHIP(c, hipStreamSynchronize(st));                      // my contribution is complete
    c->g->sendPtr[(size_t)c->rank] = send;
    CPF_LBARRIER(c);
    for (int r = 0; r < c->g->n && bytes; ++r)
        CPF_LHIP(c, hipMemcpyAsync((char*)recv + (size_t)r * bytes, c->g->sendPtr[(size_t)r], bytes, hipMemcpyDefault, st));
    CPF_LHIP(c, hipStreamSynchronize(st));
    CPF_LBARRIER(c);                                            // nobody touches a send buffer before everybody has read it
    return CPF_OK;
}
int localAllReduce(void* self, double* buf, size_t count, void* stream) {
    LocalComm* c = (LocalComm*)self;
    hipStream_t st = (hipStream_t)stream;
    const int n = c->g->n;
    CPF_LHIP(c, hipSetDevice(c->device));
    if (c->scratchBytes < (size_t)n * count * 8) {
        CPF_LHIP(c, hipStreamSynchronize(st));
        if (c->scratch) CPF_LHIP(c, hipFree(c->scratch));
        c->scratch = nullptr; c->scratchBytes = 0;
        CPF_LHIP(c, hipMalloc((void**)&c->scratch, std::max<size_t>((size_t)n * count * 8, 16)));
        c->scratchBytes = (size_t)n * count * 8;
    }
    CPF_LHIP(c, hipStreamSynchronize(st));
    c->g->sendPtr[(size_t)c->rank] = buf;
    CPF_LBARRIER(c);
    for (int r = 0; r < n && count; ++r)
        CPF_LHIP(c, hipMemcpyAsync(c->scratch + (size_t)r * count, c->g->sendPtr[(size_t)r], count * 8, hipMemcpyDefault, st));
    CPF_LHIP(c, hipStreamSynchronize(st));
    CPF_LBARRIER(c);                                            // every rank holds every row: the buffers may be overwritten
    CPF_LHIP(c, cpf::sum_rows(st, c->scratch, n, count, buf));  // rows summed in rank order: the same bits on every rank
    return CPF_OK;
}
int localAllToAllV(void* self, const void* send, const int64_t* sendOff, const int64_t* sendBytes, void* recv,
                   const int64_t* recvOff, const int64_t* recvBytes, void* stream) {
    LocalComm* c = (LocalComm*)self;
    hipStream_t st = (hipStream_t)stream;
    const int n = c->g->n, me = c->rank;
    CPF_LHIP(c, hipSetDevice(c->device));
    CPF_LHIP(c, hipStreamSynchronize(st));
    c->g->sendPtr[(size_t)me] = send; c->g->sendOff[(size_t)me] = sendOff; c->g->sendBytes[(size_t)me] = sendBytes;
    CPF_LBARRIER(c);
    for (int p = 0; p < n; ++p) {
        const int64_t bytes = c->g->sendBytes[(size_t)p][me];
        if (bytes != recvBytes[p]) {
            c->err = "in-process all-to-all-v: rank " + std::to_string(p) + " sends " + std::to_string(bytes) + " B to rank " +
                     std::to_string(me) + ", which expects " + std::to_string(recvBytes[p]);
            c->g->breakGroup();
            return CPF_ERR_ARG;
        }
        if (bytes > 0)
            CPF_LHIP(c, hipMemcpyAsync((char*)recv + recvOff[p], (const char*)c->g->sendPtr[(size_t)p] + c->g->sendOff[(size_t)p][me],
                                       (size_t)bytes, hipMemcpyDefault, st));
    }
    CPF_LHIP(c, hipStreamSynchronize(st));
    CPF_LBARRIER(c);
    return CPF_OK;
}
void localDestroy(void* self) {
    LocalComm* c = (LocalComm*)self;
    if (c->scratch) { (void)hipSetDevice(c->device); (void)hipFree(c->scratch); }
    {
        std::lock_guard<std::mutex> lk(g_groupsMutex);
        std::lock_guard<std::mutex> lk2(c->g->m);
        if (++c->g->left == c->g->n) g_groups.erase(c->key);
    }
    delete c;
}
const char* localLastError(void* self) { return ((LocalComm*)self)->err.c_str(); }

int wantedKind(int kind) {
    if (kind == CPF_COMM_RCCL || kind == CPF_COMM_INPROCESS) return kind;
    const char* e = std::getenv("CPF_COMM");
    if (e && (std::strcmp(e, "inprocess") == 0 || std::strcmp(e, "local") == 0)) return CPF_COMM_INPROCESS;
    return CPF_COMM_RCCL;
}

}  // namespace

extern "C" {

int cpf_device_count(int* count) {
    if (!count) return commFail(CPF_ERR_ARG, "cpf_device_count: null argument");
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return commFail(CPF_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *count = n;
    return CPF_OK;
}

int cpf_comm_default_kind(void) { return wantedKind(0); }

int cpf_comm_unique_id(void* id, int kind) {
    if (!id) return commFail(CPF_ERR_ARG, "cpf_comm_unique_id: null id");
    if (kind != 0 && kind != CPF_COMM_RCCL && kind != CPF_COMM_INPROCESS) return commFail(CPF_ERR_ARG, "cpf_comm_unique_id: unknown kind");
    std::memset(id, 0, CPF_COMM_ID_BYTES);
    if (wantedKind(kind) == CPF_COMM_INPROCESS) {
        std::memcpy(id, kLocalMagic, sizeof kLocalMagic);
        const uint64_t serial = g_groupSerial.fetch_add(1);
        std::memcpy((char*)id + 16, &serial, 8);
        return CPF_OK;
    }
    RcclApi& api = rccl();
    if (!api.lib || !api.why.empty()) return commFail(CPF_ERR_STATE, "cpf_comm_unique_id: " + api.why);
    static_assert(sizeof(ncclUniqueId) == CPF_COMM_ID_BYTES, "CPF_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");
    ncclUniqueId u;
    const ncclResult_t r = api.GetUniqueId(&u);
    if (r != ncclSuccess) return commFail(CPF_ERR_HIP, std::string("ncclGetUniqueId: ") + api.GetErrorString(r));
    std::memcpy(id, &u, CPF_COMM_ID_BYTES);
    return CPF_OK;
}

int cpf_comm_create(const void* id, int rank, int nRanks, int device, cpf_comm** out) {
    if (!out) return commFail(CPF_ERR_ARG, "cpf_comm_create: out is null");
    *out = nullptr;
    if (!id || nRanks < 1 || nRanks > CPF_MAX_RANKS || rank < 0 || rank >= nRanks)
        return commFail(CPF_ERR_ARG, "cpf_comm_create: bad arguments (1 <= nRanks <= CPF_MAX_RANKS, 0 <= rank < nRanks)");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return commFail(CPF_ERR_HIP, std::string("cpf_comm_create: no HIP device (") + hipGetErrorString(e) + ")");
    if (device < 0 || device >= count) return commFail(CPF_ERR_ARG, "cpf_comm_create: device index out of range");
    e = hipSetDevice(device);
    if (e != hipSuccess) return commFail(CPF_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    cpf_comm* c = new (std::nothrow) cpf_comm();
    if (!c) return commFail(CPF_ERR_NOMEM, "cpf_comm_create: out of host memory");
    c->rank = rank; c->nRanks = nRanks;
    if (std::memcmp(id, kLocalMagic, sizeof kLocalMagic) == 0) {
        const std::string key((const char*)id, CPF_COMM_ID_BYTES);
        std::shared_ptr<LocalGroup> g;
        {
            std::lock_guard<std::mutex> lk(g_groupsMutex);
            auto& slot = g_groups[key];
            if (!slot) {
                slot = std::make_shared<LocalGroup>();
                slot->n = nRanks;
                slot->sendPtr.assign((size_t)nRanks, nullptr); slot->sendOff.assign((size_t)nRanks, nullptr); slot->sendBytes.assign((size_t)nRanks, nullptr);
                if (const char* t = std::getenv("CPF_COMM_TIMEOUT")) slot->timeoutS = std::max(1.0, std::atof(t));
            }
            g = slot;
        }
        if (g->n != nRanks) { delete c; return commFail(CPF_ERR_ARG, "cpf_comm_create: the ranks of one communicator disagree on nRanks"); }
        LocalComm* lc = new LocalComm();
        lc->g = g; lc->key = key; lc->rank = rank; lc->device = device;
        c->self = lc;
        c->all_gather = localAllGather; c->all_reduce_sum_f64 = localAllReduce; c->all_to_all_v = localAllToAllV;
        c->destroy = localDestroy; c->last_error = localLastError;
        if (!g->barrier()) {                                   // collective: everybody has joined
            localDestroy(lc); delete c;
            return commFail(CPF_ERR_STATE, "cpf_comm_create: not all ranks joined the in-process communicator");
        }
        *out = c;
        return CPF_OK;
    }
    RcclApi& api = rccl();
    if (!api.lib || !api.why.empty()) { delete c; return commFail(CPF_ERR_STATE, "cpf_comm_create: " + api.why); }
    RcclComm* rc = new RcclComm();
    rc->rank = rank; rc->n = nRanks; rc->device = device;
    ncclUniqueId u;
    std::memcpy(&u, id, CPF_COMM_ID_BYTES);
    const ncclResult_t r = api.CommInitRank(&rc->comm, nRanks, u, rank);
    if (r != ncclSuccess) {
        const std::string m = std::string("ncclCommInitRank: ") + api.GetErrorString(r);
        delete rc; delete c;
        return commFail(CPF_ERR_HIP, m);
    }
    c->self = rc;
    c->all_gather = rcclAllGather; c->all_reduce_sum_f64 = rcclAllReduce; c->all_to_all_v = rcclAllToAllV;
    c->destroy = rcclDestroy; c->last_error = rcclLastError;
    *out = c;
    return CPF_OK;
}

void cpf_comm_destroy(cpf_comm* comm) {
    if (!comm) return;
    if (comm->destroy) comm->destroy(comm->self);
    delete comm;
}

const char* cpf_comm_last_error(void) { return t_commError.c_str(); }

}  // extern "C"
