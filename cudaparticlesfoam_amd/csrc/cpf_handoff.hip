// Multi-GPU hand-off kernels: split a rank's cloud into stayers and leavers after a step.
//
// No reference counterpart (the reference drives ONE GPU from the MPI master rank,
// src/advect.H:59-89).  Rank r owns cells [cellLo[r], cellLo[r+1]); the mesh
// is replicated.  Leavers (2-3 % of the cloud per step for x-slab ownership) are written,
// grouped by destination and in stable index order, into one send buffer that goes straight
// into an RCCL all-to-all; the holes they leave below nStay are filled with the stayers that
// sit above nStay, so only O(leavers) particles move.  Compaction uses wave64 ballots and
// popcount prefixes; cross-wave offsets go through LDS.
#include "cpf_device.h"

namespace cpf {

constexpr int kItems = 8;                    // particles per thread per tile
constexpr int kTile = kBlock * kItems;       // 2048 particles per block
constexpr int kMaxRanks = 16;

__device__ __forceinline__ int owner_rank(int c, const int32_t* __restrict__ cellLo, int nRanks) {
    int r = 0;
#pragma unroll 1
    for (int q = 1; q < nRanks; ++q) r += (c >= cellLo[q]) ? 1 : 0;
    return r;
}

// dest = -1 stays (own cell range, or lost/frozen), else destination rank
__device__ __forceinline__ int classify(int c, const int32_t* __restrict__ cellLo, int nRanks, int myRank) {
    if (c < 0) return -1;
    const int r = owner_rank(c, cellLo, nRanks);
    return r == myRank ? -1 : r;
}

__global__ __launch_bounds__(kBlock) void count_leavers_kernel(const int32_t* __restrict__ cell, int64_t n,
                                                               const int32_t* __restrict__ cellLo, int nRanks,
                                                               int myRank, int32_t* __restrict__ blockCnt) {
    __shared__ int sCnt[kMaxRanks];
    if (threadIdx.x < kMaxRanks) sCnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kTile;
    int mine[kMaxRanks];
#pragma unroll
    for (int r = 0; r < kMaxRanks; ++r) mine[r] = 0;
    for (int it = 0; it < kItems; ++it) {
        const int64_t i = base + (int64_t)it * kBlock + threadIdx.x;
        const int d = (i < n) ? classify(cell[i], cellLo, nRanks, myRank) : -1;
        for (int r = 0; r < nRanks; ++r) {
            const unsigned long long b = __ballot(d == r);
            if ((threadIdx.x & 63) == 0) mine[r] += __popcll(b);
        }
    }
    if ((threadIdx.x & 63) == 0)
        for (int r = 0; r < nRanks; ++r)
            if (mine[r]) atomicAdd(&sCnt[r], mine[r]);
    __syncthreads();
    if (threadIdx.x < nRanks) blockCnt[(int64_t)blockIdx.x * nRanks + threadIdx.x] = sCnt[threadIdx.x];
}

// single block: exclusive scan of blockCnt over blocks per destination; totals, bases, nStay
__global__ __launch_bounds__(kBlock) void scan_leavers_kernel(int32_t* __restrict__ blockCnt, int nBlocks, int nRanks,
                                                              int64_t n, int64_t* __restrict__ counts,
                                                              int64_t* __restrict__ destBase,
                                                              int64_t* __restrict__ nStay,
                                                              unsigned long long* __restrict__ holeFill) {
    __shared__ long long sPart[kBlock];
    __shared__ long long sTot[kMaxRanks];
    for (int r = 0; r < nRanks; ++r) {
        // each thread owns a contiguous chunk of blocks
        const int per = (nBlocks + kBlock - 1) / kBlock;
        const int b0 = threadIdx.x * per, b1 = min(nBlocks, b0 + per);
        long long s = 0;
        for (int b = b0; b < b1; ++b) s += blockCnt[(int64_t)b * nRanks + r];
        sPart[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            long long run = 0;
            for (int t = 0; t < kBlock; ++t) { const long long v = sPart[t]; sPart[t] = run; run += v; }
            sTot[r] = run;
        }
        __syncthreads();
        long long run = sPart[threadIdx.x];
        for (int b = b0; b < b1; ++b) {
            const int v = blockCnt[(int64_t)b * nRanks + r];
            blockCnt[(int64_t)b * nRanks + r] = (int32_t)run;
            run += v;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        long long run = 0;
        for (int r = 0; r < nRanks; ++r) { counts[r] = sTot[r]; destBase[r] = run; run += sTot[r]; }
        *nStay = n - run;
        holeFill[0] = 0; holeFill[1] = 0;
    }
}

__global__ __launch_bounds__(kBlock) void write_leavers_kernel(
    const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z,
    const int32_t* __restrict__ cell, const int64_t* __restrict__ gid, int64_t n, const int32_t* __restrict__ cellLo,
    int nRanks, int myRank, const int32_t* __restrict__ blockOff, const int64_t* __restrict__ destBase,
    const int64_t* __restrict__ nStayPtr, double* __restrict__ sendbuf, int64_t sendCapacity,
    int32_t* __restrict__ holes, int32_t* __restrict__ fillers, unsigned long long* __restrict__ holeFill) {
    __shared__ int sWave[kBlock / 64][kMaxRanks];
    __shared__ int sRun[kMaxRanks];
    if (threadIdx.x < kMaxRanks) sRun[threadIdx.x] = 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long ltMask = lane ? (~0ull >> (64 - lane)) : 0ull;
    const int64_t nStay = *nStayPtr;
    const int64_t base = (int64_t)blockIdx.x * kTile;
    for (int it = 0; it < kItems; ++it) {
        __syncthreads();
        const int64_t i = base + (int64_t)it * kBlock + threadIdx.x;
        const int c = (i < n) ? cell[i] : -1;
        const int d = (i < n) ? classify(c, cellLo, nRanks, myRank) : -1;
        int myPrefix = 0;
        for (int r = 0; r < nRanks; ++r) {
            const unsigned long long b = __ballot(d == r);
            if (lane == 0) sWave[wave][r] = __popcll(b);
            if (d == r) myPrefix = __popcll(b & ltMask);
        }
        __syncthreads();
        if (d >= 0) {
            int off = sRun[d] + myPrefix;
            for (int w = 0; w < wave; ++w) off += sWave[w][d];
            const int64_t slot = destBase[d] + blockOff[(int64_t)blockIdx.x * nRanks + d] + off;
            if (slot < sendCapacity) {
                double* rec = sendbuf + slot * CPF_HANDOFF_DOUBLES;
                rec[0] = x[i]; rec[1] = y[i]; rec[2] = z[i];
                rec[3] = (double)c;                              // exact: |c| < 2^31
                rec[4] = (double)(gid ? gid[i] : i);             // exact below 2^53
            }
            if (i < nStay) holes[atomicAdd(&holeFill[0], 1ull)] = (int32_t)i;
        } else if (i < n && i >= nStay) {
            fillers[atomicAdd(&holeFill[1], 1ull)] = (int32_t)i;
        }
        __syncthreads();
        if (threadIdx.x < nRanks) {
            int s = 0;
            for (int w = 0; w < kBlock / 64; ++w) s += sWave[w][threadIdx.x];
            sRun[threadIdx.x] += s;
        }
    }
}

__global__ __launch_bounds__(kBlock) void fill_holes_kernel(double* __restrict__ x, double* __restrict__ y,
                                                            double* __restrict__ z, int32_t* __restrict__ cell,
                                                            int64_t* __restrict__ gid,
                                                            const int32_t* __restrict__ holes,
                                                            const int32_t* __restrict__ fillers,
                                                            const unsigned long long* __restrict__ holeFill) {
    const unsigned long long nH = holeFill[0];
    for (unsigned long long k = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; k < nH;
         k += (unsigned long long)gridDim.x * kBlock) {
        const int h = holes[k], f = fillers[k];
        x[h] = x[f]; y[h] = y[f]; z[h] = z[f]; cell[h] = cell[f];
        if (gid) gid[h] = gid[f];
    }
}

__global__ __launch_bounds__(kBlock) void unpack_arrivals_kernel(double* __restrict__ x, double* __restrict__ y,
                                                                 double* __restrict__ z, int32_t* __restrict__ cell,
                                                                 int64_t* __restrict__ gid, int64_t nStay,
                                                                 const double* __restrict__ recvbuf, int64_t nRecv) {
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= nRecv) return;
    const double* rec = recvbuf + k * CPF_HANDOFF_DOUBLES;
    const int64_t i = nStay + k;
    x[i] = rec[0]; y[i] = rec[1]; z[i] = rec[2];
    cell[i] = (int32_t)rec[3];
    if (gid) gid[i] = (int64_t)rec[4];
}

static inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

size_t handoff_scratch_bytes(int64_t n, int nRanks) {
    const int64_t nBlocks = (n + kTile - 1) / kTile;
    return al256((size_t)nBlocks * nRanks * 4) + al256((size_t)kMaxRanks * 8) + al256(16) + 2 * al256((size_t)n * 4);
}

hipError_t pack_leavers(hipStream_t st, double* x, double* y, double* z, int32_t* cell, int64_t* gid, int64_t n,
                        const int32_t* cellLo, int nRanks, int myRank, double* sendbuf, int64_t sendCapacity,
                        int64_t* counts, int64_t* nStay, void* scratch, size_t scratchBytes) {
    if (nRanks < 1 || nRanks > kMaxRanks) return hipErrorInvalidValue;
    if (scratchBytes < handoff_scratch_bytes(n, nRanks)) return hipErrorInvalidValue;
    const int nBlocks = (int)((n + kTile - 1) / kTile);
    char* p = (char*)scratch;
    int32_t* blockCnt = (int32_t*)p; p += al256((size_t)std::max(nBlocks, 1) * nRanks * 4);
    int64_t* destBase = (int64_t*)p; p += al256((size_t)kMaxRanks * 8);
    unsigned long long* holeFill = (unsigned long long*)p; p += al256(16);
    int32_t* holes = (int32_t*)p; p += al256((size_t)n * 4);
    int32_t* fillers = (int32_t*)p;
    if (nBlocks > 0)
        hipLaunchKernelGGL(count_leavers_kernel, dim3(nBlocks), dim3(kBlock), 0, st, cell, n, cellLo, nRanks, myRank,
                           blockCnt);
    hipLaunchKernelGGL(scan_leavers_kernel, dim3(1), dim3(kBlock), 0, st, blockCnt, nBlocks, nRanks, n, counts,
                       destBase, nStay, holeFill);
    if (nBlocks > 0) {
        hipLaunchKernelGGL(write_leavers_kernel, dim3(nBlocks), dim3(kBlock), 0, st, x, y, z, cell, gid, n, cellLo,
                           nRanks, myRank, blockCnt, destBase, nStay, sendbuf, sendCapacity, holes, fillers, holeFill);
        const int fillBlocks = (int)std::min<int64_t>(1024, (n + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(fill_holes_kernel, dim3(fillBlocks), dim3(kBlock), 0, st, x, y, z, cell, gid, holes, fillers,
                           holeFill);
    }
    return hipGetLastError();
}

hipError_t unpack_arrivals(hipStream_t st, double* x, double* y, double* z, int32_t* cell, int64_t* gid,
                           int64_t nStay, const double* recvbuf, int64_t nRecv) {
    if (nRecv > 0)
        hipLaunchKernelGGL(unpack_arrivals_kernel, dim3((unsigned)((nRecv + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                           x, y, z, cell, gid, nStay, recvbuf, nRecv);
    return hipGetLastError();
}

}  // namespace cpf
