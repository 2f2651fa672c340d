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
constexpr int kMaxRanks = CPF_MAX_RANKS;   // 64: per-destination counters live in LDS, nothing is unrolled over ranks

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
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < kItems; ++it) {
        const int64_t i = base + (int64_t)it * kBlock + threadIdx.x;
        const int d = (i < n) ? classify(cell[i], cellLo, nRanks, myRank) : -1;
        // one pass per DISTINCT destination among the wave's leavers (usually none or one)
        unsigned long long any = __ballot(d >= 0);
        while (any) {
            const int r = __builtin_amdgcn_readlane(d, __ffsll((long long)any) - 1);
            const unsigned long long b = __ballot(d == r);
            if (lane == 0) atomicAdd(&sCnt[r], (int)__popcll(b));
            any &= ~b;
        }
    }
    __syncthreads();
    if (threadIdx.x < nRanks) blockCnt[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = sCnt[threadIdx.x];     // [rank][block]
}

// Exclusive scan of blockCnt[rank][block] over the blocks, one workgroup per destination rank (the rows are contiguous: every
// thread sums a run of blocks, the runs' sums are scanned through LDS, every thread rewrites its run); rowTotal[rank] = the
// destination's leavers.  (Until round 5 ONE workgroup did all ranks in turn over a block-major table, thread 0 scanning 256
// partial sums per rank: 114 us of the split's 190 at 1.25e7 particles and 8 ranks.)
__global__ __launch_bounds__(kBlock) void scan_leavers_kernel(int32_t* __restrict__ blockCnt, int nBlocks, long long* __restrict__ rowTotal) {
    __shared__ long long sPart[kBlock];
    __shared__ long long sWaveTot[kBlock / 64];
    int32_t* const row = blockCnt + (int64_t)blockIdx.x * nBlocks;
    const int per = (nBlocks + kBlock - 1) / kBlock;
    const int b0 = min(nBlocks, (int)threadIdx.x * per), b1 = min(nBlocks, b0 + per);
    long long s = 0;
    for (int b = b0; b < b1; ++b) s += row[b];
    // inclusive scan of the 256 run sums: inside each wave by shuffles, across the four waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const long long up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) sWaveTot[wave] = incl;
    __syncthreads();
    long long base = 0;
    for (int w = 0; w < wave; ++w) base += sWaveTot[w];
    long long run = base + incl - s;
    for (int b = b0; b < b1; ++b) { const int v = row[b]; row[b] = (int32_t)run; run += v; }
    if (threadIdx.x == kBlock - 1) rowTotal[blockIdx.x] = base + incl;
    (void)sPart;
}

// totals -> counts, first record of every destination, nStay.  Leavers that do not fit into the send buffer abort the split HERE,
// on the device: *nStay = -1, holeFill[2] = 1, and the three kernels after this one touch nothing -- the shard stays exactly as it
// was (the caller keeps stepping it), counts[] say how large the buffer has to be, and the host can repeat the split
// (csrc/cpf_shard_core.h, finishExchange).
__global__ void finish_scan_kernel(const long long* __restrict__ rowTotal, int nRanks, int64_t n, int64_t sendCapacity,
                                   int64_t* __restrict__ counts, int64_t* __restrict__ destBase, int64_t* __restrict__ nStay,
                                   unsigned long long* __restrict__ holeFill) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    long long run = 0;
    for (int r = 0; r < nRanks; ++r) { counts[r] = rowTotal[r]; destBase[r] = run; run += rowTotal[r]; }
    const bool fits = run <= sendCapacity;
    *nStay = fits ? n - run : -1;
    holeFill[0] = 0; holeFill[1] = 0; holeFill[2] = fits ? 0ull : 1ull;
}

__global__ __launch_bounds__(kBlock) void write_leavers_kernel(
    const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z,
    const int32_t* __restrict__ cell, const int64_t* __restrict__ gid, int64_t n, const int32_t* __restrict__ cellLo,
    int nRanks, int myRank, const int32_t* __restrict__ blockOff, const int64_t* __restrict__ destBase,
    const int64_t* __restrict__ nStayPtr, double* __restrict__ sendbuf, int64_t sendCapacity,
    int32_t* __restrict__ holes, int32_t* __restrict__ fillers, unsigned long long* __restrict__ holeFill) {
    __shared__ int sWave[kBlock / 64][kMaxRanks];
    __shared__ int sRun[kMaxRanks];
    static_assert((kBlock / 64) * kMaxRanks == kBlock, "sWave is cleared one entry per thread");
    if (holeFill[2] != 0ull) return;                          // split aborted (send buffer too small): move nothing
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
        ((int*)sWave)[threadIdx.x] = 0;                          // kBlock == (kBlock / 64) * kMaxRanks entries
        __syncthreads();
        unsigned long long any = __ballot(d >= 0);
        while (any) {                                            // per distinct destination among the wave's leavers
            const int r = __builtin_amdgcn_readlane(d, __ffsll((long long)any) - 1);
            const unsigned long long b = __ballot(d == r);
            if (lane == 0) sWave[wave][r] = __popcll(b);
            if (d == r) myPrefix = __popcll(b & ltMask);
            any &= ~b;
        }
        __syncthreads();
        if (d >= 0) {
            int off = sRun[d] + myPrefix;
            for (int w = 0; w < wave; ++w) off += sWave[w][d];
            const int64_t slot = destBase[d] + blockOff[(int64_t)d * gridDim.x + blockIdx.x] + off;
            double* rec = sendbuf + slot * CPF_HANDOFF_DOUBLES;   // slot < sendCapacity: the scan kernel checked the total
            rec[0] = x[i]; rec[1] = y[i]; rec[2] = z[i];
            rec[3] = (double)c;                                  // exact: |c| < 2^31
            rec[4] = (double)(gid ? gid[i] : i);                 // exact below 2^53
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

// After the split the slots [nStay, n) hold stale copies (leavers and moved fillers).  Marking them CPF_CELL_LOST
// lets the caller keep stepping the old range [0, n) while the all-to-all is still in flight (the host learns nStay
// only with the counts): inactive lanes cost one 4-byte load.
__global__ __launch_bounds__(kBlock) void mark_tail_kernel(int32_t* __restrict__ cell, const int64_t* __restrict__ nStayPtr,
                                                           int64_t n) {
    const int64_t nStay = *nStayPtr;
    if (nStay < 0) return;                                    // split aborted: nothing is stale
    for (int64_t i = nStay + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        cell[i] = CPF_CELL_LOST;
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

// ---------------------------------------------------------------------------------------------------------
// Ownership re-cut: per-cell particle counts -> (all-reduce over ranks, by the host layer) -> range cuts.
// ---------------------------------------------------------------------------------------------------------
constexpr int kHistBlock = 1024;             // 16 waves per block: one block per CU keeps a private LDS histogram
constexpr int kHistBlocks = 512;
constexpr int kHistMaxLdsCells = 32768;      // 128 KB of the CU's 160 KB LDS as u32 bins

// Path A (meshes up to 32768 cells): every block counts a contiguous chunk of the shard into an LDS histogram
// (ds_add_u32; runs of equal cells in a wave -- the cloud is mostly cell-sorted -- are pre-counted with a ballot
// so they cost one LDS atomic), then stores it densely as one row of partial[block][cell].  No global atomics,
// independent of how sorted the input is, and deterministic.
__global__ __launch_bounds__(kHistBlock) void cell_histogram_lds_kernel(const int32_t* __restrict__ cell, int64_t n,
                                                                        int nCells, int64_t perBlock,
                                                                        unsigned int* __restrict__ partial) {
    extern __shared__ unsigned int sHist[];
    for (int b = threadIdx.x; b < nCells; b += kHistBlock) sHist[b] = 0u;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * perBlock;
    const int64_t hi = (lo + perBlock < n) ? lo + perBlock : n;
    const int lane = threadIdx.x & 63;
    constexpr int kUnroll = 4;                                        // 4 independent loads in flight per lane
    for (int64_t base = lo; base < hi; base += kUnroll * kHistBlock) {   // wave-uniform trip count
        int cs[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t i = base + (int64_t)u * kHistBlock + threadIdx.x;
            cs[u] = (i < hi) ? cell[i] : -1;
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int c = cs[u];
            bool todo = c >= 0;
#pragma unroll 1
            for (int round = 0; round < 2; ++round) {
                const unsigned long long m = __ballot(todo);
                if (m == 0ull) break;
                const int leader = __ffsll((long long)m) - 1;
                const int cl = __builtin_amdgcn_readlane(c, leader);
                const unsigned long long same = __ballot(todo && c == cl);
                if (lane == leader) atomicAdd(&sHist[cl], (unsigned int)__popcll(same));
                if (c == cl) todo = false;
            }
            if (todo) atomicAdd(&sHist[c], 1u);
        }
    }
    __syncthreads();
    unsigned int* row = partial + (int64_t)blockIdx.x * nCells;
    for (int b = threadIdx.x; b < nCells; b += kHistBlock) row[b] = sHist[b];
}

// weights[c] = scale * sum over blocks of partial[block][c].  A block takes 64 cells; its 4 waves split the rows
// (lane = cell, so every row read is one coalesced 256-byte segment) and combine through LDS.
__global__ __launch_bounds__(kBlock) void cell_histogram_reduce_kernel(const unsigned int* __restrict__ partial,
                                                                       int nBlocks, int nCells, double scale,
                                                                       double* __restrict__ weights) {
    __shared__ unsigned long long sPart[kBlock / 64][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int b = blockIdx.x * 64 + lane;
    unsigned long long s = 0;
    if (b < nCells) {
#pragma unroll 8
        for (int k = grp; k < nBlocks; k += kBlock / 64) s += partial[(int64_t)k * nCells + b];
    }
    sPart[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && b < nCells) {
        unsigned long long tot = 0;
#pragma unroll
        for (int g = 0; g < kBlock / 64; ++g) tot += sPart[g][lane];
        weights[b] = (double)tot * scale;
    }
}

// Path B (larger meshes): global 64-bit atomics, one per run of equal cells in a wave.
__global__ __launch_bounds__(kBlock) void cell_histogram_global_kernel(const int32_t* __restrict__ cell, int64_t n,
                                                                       unsigned long long* __restrict__ hist) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int c = (i < n) ? cell[i] : -1;
    bool todo = c >= 0;
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (int round = 0; round < 4; ++round) {
        const unsigned long long m = __ballot(todo);
        if (m == 0ull) break;
        const int leader = __ffsll((long long)m) - 1;
        const int cl = __builtin_amdgcn_readlane(c, leader);
        const unsigned long long same = __ballot(todo && c == cl);
        if (lane == leader) atomicAdd(&hist[cl], (unsigned long long)__popcll(same));
        if (c == cl) todo = false;
    }
    if (todo) atomicAdd(&hist[c], 1ull);
}

__global__ __launch_bounds__(kBlock) void counts_to_weights_kernel(const unsigned long long* __restrict__ hist,
                                                                   int nCells, double scale, double* __restrict__ weights) {
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b < nCells) weights[b] = (double)hist[b] * scale;
}

size_t histogram_scratch_bytes(int64_t nCells) {
    return nCells <= kHistMaxLdsCells ? (size_t)kHistBlocks * (size_t)nCells * 4 : (size_t)nCells * 8;
}

hipError_t cell_histogram(hipStream_t st, const int32_t* cell, int64_t n, int64_t nCells, double scale, double* weights,
                          void* scratch, size_t scratchBytes) {
    if (scratchBytes < histogram_scratch_bytes(nCells)) return hipErrorInvalidValue;
    const int cellBlocks = (int)((nCells + kBlock - 1) / kBlock);
    if (nCells <= kHistMaxLdsCells) {
        static bool attrSet = false;
        if (!attrSet) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cell_histogram_lds_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, kHistMaxLdsCells * 4);
            if (e != hipSuccess) return e;
            attrSet = true;
        }
        int64_t perBlock = (n + kHistBlocks - 1) / kHistBlocks;
        perBlock = (perBlock + 4 * kHistBlock - 1) / (4 * kHistBlock) * (4 * kHistBlock);   // whole unrolled trips
        if (perBlock == 0) perBlock = 4 * kHistBlock;
        hipLaunchKernelGGL(cell_histogram_lds_kernel, dim3(kHistBlocks), dim3(kHistBlock), (size_t)nCells * 4, st, cell, n,
                           (int)nCells, perBlock, (unsigned int*)scratch);
        hipLaunchKernelGGL(cell_histogram_reduce_kernel, dim3((unsigned)((nCells + 63) / 64)), dim3(kBlock), 0, st,
                           (const unsigned int*)scratch, kHistBlocks, (int)nCells, scale, weights);
    } else {
        hipError_t e = hipMemsetAsync(scratch, 0, (size_t)nCells * 8, st);
        if (e != hipSuccess) return e;
        if (n > 0)
            hipLaunchKernelGGL(cell_histogram_global_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                               st, cell, n, (unsigned long long*)scratch);
        hipLaunchKernelGGL(counts_to_weights_kernel, dim3(cellBlocks), dim3(kBlock), 0, st,
                           (const unsigned long long*)scratch, (int)nCells, scale, weights);
    }
    return hipGetLastError();
}

// Equal-weight cuts of cells 0..nCells-1 into nRanks contiguous ranges, the rule of parallel.slab_cell_ranges:
// cum0[i] = w[0] + ... + w[i-1] (i = 0..nCells), cut_q = #{ i : cum0[i] < total*q/nRanks }  (== searchsorted-left,
// cum0 being non-decreasing).  One block: per-thread chunk sums, a serial scan of the 1024 chunk sums in LDS, then
// every thread walks its chunk again and counts.  A few tens of microseconds even for 1e6 cells; runs once per
// re-cut.
__global__ __launch_bounds__(kHistBlock) void cell_ranges_kernel(const double* __restrict__ w, int nCells, int nRanks,
                                                                 int32_t* __restrict__ cellLo) {
    __shared__ double sBase[kHistBlock];
    __shared__ double sTotal;
    __shared__ int sCut[kMaxRanks];
    const int nIdx = nCells + 1;                                   // cum0 has nCells + 1 entries
    const int per = (nIdx + kHistBlock - 1) / kHistBlock;
    const int i0 = min(nIdx, (int)threadIdx.x * per), i1 = min(nIdx, i0 + per);
    double s = 0.0;
    for (int i = i0; i < i1; ++i) s += (i < nCells) ? w[i] : 0.0;
    sBase[threadIdx.x] = s;
    if (threadIdx.x < kMaxRanks) sCut[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        double run = 0.0;
        for (int t = 0; t < kHistBlock; ++t) { const double v = sBase[t]; sBase[t] = run; run += v; }
        sTotal = run;
    }
    __syncthreads();
    const double total = sTotal;
    // element i counts for every cut q with run_i < total*q/nRanks; the thresholds grow with q, so that is every q >= qmin(i),
    // and run only grows along the chunk, so qmin does too: one LDS count per element at qmin, prefix-summed over q below
    double run = sBase[threadIdx.x];
    int q = 1;
    for (int i = i0; i < i1; ++i) {
        while (q < nRanks && !(run < total * (double)q / (double)nRanks)) ++q;
        if (q < nRanks) atomicAdd(&sCut[q], 1);
        if (i < nCells) run += w[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int prev = 0, acc = 0;
        cellLo[0] = 0;
        for (int q = 1; q < nRanks; ++q) { acc += sCut[q]; prev = max(prev, acc); cellLo[q] = prev; }
        cellLo[nRanks] = nCells;
    }
}

hipError_t cell_ranges(hipStream_t st, const double* weights, int64_t nCells, int nRanks, int32_t* cellLo) {
    if (nRanks < 1 || nRanks > kMaxRanks) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cell_ranges_kernel, dim3(1), dim3(kHistBlock), 0, st, weights, (int)nCells, nRanks, cellLo);
    return hipGetLastError();
}

static inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

size_t handoff_scratch_bytes(int64_t n, int nRanks) {
    const int64_t nBlocks = std::max<int64_t>(1, (n + kTile - 1) / kTile);
    return al256((size_t)nBlocks * nRanks * 4) + 2 * al256((size_t)kMaxRanks * 8) + al256(24) + 2 * al256((size_t)n * 4);
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
    long long* rowTotal = (long long*)p; p += al256((size_t)kMaxRanks * 8);
    unsigned long long* holeFill = (unsigned long long*)p; p += al256(24);
    int32_t* holes = (int32_t*)p; p += al256((size_t)n * 4);
    int32_t* fillers = (int32_t*)p;
    if (nBlocks > 0)
        hipLaunchKernelGGL(count_leavers_kernel, dim3(nBlocks), dim3(kBlock), 0, st, cell, n, cellLo, nRanks, myRank,
                           blockCnt);
    hipLaunchKernelGGL(scan_leavers_kernel, dim3(nRanks), dim3(kBlock), 0, st, blockCnt, nBlocks, rowTotal);
    hipLaunchKernelGGL(finish_scan_kernel, dim3(1), dim3(64), 0, st, rowTotal, nRanks, n, sendCapacity, counts, destBase, nStay, holeFill);
    if (nBlocks > 0) {
        hipLaunchKernelGGL(write_leavers_kernel, dim3(nBlocks), dim3(kBlock), 0, st, x, y, z, cell, gid, n, cellLo,
                           nRanks, myRank, blockCnt, destBase, nStay, sendbuf, sendCapacity, holes, fillers, holeFill);
        const int fillBlocks = (int)std::min<int64_t>(1024, (n + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(fill_holes_kernel, dim3(fillBlocks), dim3(kBlock), 0, st, x, y, z, cell, gid, holes, fillers,
                           holeFill);
        hipLaunchKernelGGL(mark_tail_kernel, dim3(fillBlocks), dim3(kBlock), 0, st, cell, nStay, n);
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

// ---------------------------------------------------------------------------------------------------------
// Output of a sharded cloud (cpf_shard_gather): every rank packs its particles as records of CPF_OUTPUT_DOUBLES
// doubles (x, y, z, cell, gid, vx, vy, vz), the root receives all of them and scatters them into particle-id order.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void pack_output_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                                             const double* __restrict__ z, const int32_t* __restrict__ cell,
                                                             const int64_t* __restrict__ gid, const double* __restrict__ vel3,
                                                             double* __restrict__ rec, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    double* r = rec + i * kOutputDoubles;
    r[0] = x[i]; r[1] = y[i]; r[2] = z[i];
    r[3] = (double)cell[i];
    r[4] = (double)gid[i];
    r[5] = vel3 ? vel3[3 * i] : 0.0; r[6] = vel3 ? vel3[3 * i + 1] : 0.0; r[7] = vel3 ? vel3[3 * i + 2] : 0.0;
}

// bad[0] counts records whose id is outside [0, nGlobal): they are dropped
__global__ __launch_bounds__(kBlock) void scatter_output_kernel(const double* __restrict__ rec, int64_t nRec, int64_t nGlobal,
                                                                double* __restrict__ xyzw, int32_t* __restrict__ cellOut,
                                                                double* __restrict__ velOut, unsigned long long* __restrict__ bad) {
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= nRec) return;
    const double* r = rec + k * kOutputDoubles;
    const int64_t g = (int64_t)r[4];
    if (g < 0 || g >= nGlobal) { atomicAdd(bad, 1ull); return; }
    const int32_t c = (int32_t)r[3];
    if (xyzw) {
        xyzw[4 * g] = r[0]; xyzw[4 * g + 1] = r[1]; xyzw[4 * g + 2] = r[2];
        xyzw[4 * g + 3] = (c == CPF_CELL_FROZEN) ? 0.0 : 1.0;
    }
    if (cellOut) cellOut[g] = c;
    if (velOut) { velOut[4 * g] = r[5]; velOut[4 * g + 1] = r[6]; velOut[4 * g + 2] = r[7]; velOut[4 * g + 3] = -1.0; }
}

hipError_t pack_output(hipStream_t st, const double* x, const double* y, const double* z, const int32_t* cell,
                       const int64_t* gid, const double* vel3, double* rec, int64_t n) {
    if (n > 0)
        hipLaunchKernelGGL(pack_output_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, x, y, z, cell,
                           gid, vel3, rec, n);
    return hipGetLastError();
}

hipError_t scatter_output(hipStream_t st, const double* rec, int64_t nRec, int64_t nGlobal, double* xyzw, int32_t* cellOut,
                          double* velOut, unsigned long long* bad) {
    if (nRec > 0)
        hipLaunchKernelGGL(scatter_output_kernel, dim3((unsigned)((nRec + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, rec, nRec,
                           nGlobal, xyzw, cellOut, velOut, bad);
    return hipGetLastError();
}

// In-process communicator (cpf_comm.cpp): buf[i] = rows[0][i] + rows[1][i] + ... in rank order -- the same bits on every rank
__global__ __launch_bounds__(kBlock) void sum_rows_kernel(const double* __restrict__ rows, int nRows, size_t count,
                                                          double* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    double s = rows[i];
    for (int r = 1; r < nRows; ++r) s += rows[(size_t)r * count + i];
    out[i] = s;
}

hipError_t sum_rows(hipStream_t st, const double* rows, int nRows, size_t count, double* out) {
    if (count > 0)
        hipLaunchKernelGGL(sum_rows_kernel, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, rows, nRows,
                           count, out);
    return hipGetLastError();
}

}  // namespace cpf
