// Shared by the streaming step kernels (cpf_stream.hip, cpf_ahead.hip): the inline-assembly memory operations the
// compiler must not keep books on, and the work distribution arguments.
#pragma once
#include "cpf_walk.h"

namespace cpf {

#ifndef CPF_STREAM_GROUPS
#define CPF_STREAM_GROUPS 256
#endif
constexpr int kStreamGroups = CPF_STREAM_GROUPS;        // wave groups sharing a chunk counter (power of two, <= 256)
constexpr int kStreamCounterStride = 16;                 // unsigned words between two counters (64 B)

// ------------------------------------------------------------------------------------------------
// memory operations the COMPILER must not keep books on (see the head of this file)
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) const char* lds_cptr;
__device__ __forceinline__ unsigned lds_addr(const void* p) {           // byte address inside the workgroup's LDS
    return (unsigned)(uintptr_t)(lds_cptr)p;
}
// a value the program knows to be wave-uniform, pinned to scalar registers (the "s" operands below need it)
__device__ __forceinline__ unsigned uniform32(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int64_t uniform64(int64_t v) {
    const unsigned lo = uniform32((unsigned)v), hi = uniform32((unsigned)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
// 16 bytes per active lane, global -> LDS[ldsDst + 16 * lane id], no register in between.  Pending LDS reads are
// retired first (the DMA must not overtake a read of the bytes it replaces); M0 is restored (compiler-reserved).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned ldsDst) {
    unsigned keep;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(ldsDst) : "memory");
}
// (s_nop 4: a scalar base the compiler has just restored from a spill lane with v_readlane must not be read by a
// vector-memory instruction within five wait states; hipcc pads its own instructions, not the inside of an asm)
__device__ __forceinline__ void async_store(double* base, unsigned byteOff, double v) {
    asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2" : : "v"(byteOff), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ void async_store(int32_t* base, unsigned byteOff, int v) {
    asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2" : : "v"(byteOff), "v"(v), "s"(base) : "memory");
}
// returning atomic increment, complete when the statement ends (the caller masks it to one lane)
__device__ __forceinline__ unsigned grab_sync(unsigned* base) {
    unsigned r;
    asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(0u), "v"(1u), "s"(base) : "memory");
    return r;
}
// three doubles at base + byteOff, + 512, + 1024 bytes, straight from L2, complete when the statement ends -- and everything
// this wave stored before it has been written back first (the values were stored by this very lane a few rounds ago)
__device__ __forceinline__ void load3_sync(const double* base, unsigned byteOff, double& a, double& b, double& c) {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 4\n\tglobal_load_dwordx2 %0, %3, %4 sc0 sc1\n\tglobal_load_dwordx2 %1, %3, %4 offset:512 sc0 sc1\n\t"
                 "global_load_dwordx2 %2, %3, %4 offset:1024 sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(byteOff), "s"(base) : "memory");
}
// Kernel arguments that are needed once per tile (or less) are read from the kernarg segment WHERE they are needed
// instead of sitting in scalar registers for the whole kernel: the streaming kernel's occupancy is bounded by its
// scalar registers (800 per SIMD: <= 96 per wave for 7 waves, <= 80 for 8; MI355X_MICROARCH.md "Residency"), and four
// array pointers are 8 of them.  A scalar load that hits the constant cache costs a wave ~100 cycles once per tile.
// (lgkmcnt(0) also retires the wave's pending LDS reads; the callers need those next anyway.)
struct CloudPtrs { double* x; double* y; double* z; int32_t* cell; };
__device__ __forceinline__ CloudPtrs kernarg_cloud_ptrs() {          // the kernel's first four parameters: bytes 0..31
    typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
    u32x8 r;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(__builtin_amdgcn_kernarg_segment_ptr()) : "memory");
    auto ptr = [](unsigned lo, unsigned hi) { return (uintptr_t)(((unsigned long long)hi << 32) | lo); };
    return {reinterpret_cast<double*>(ptr(r[0], r[1])), reinterpret_cast<double*>(ptr(r[2], r[3])),
            reinterpret_cast<double*>(ptr(r[4], r[5])), reinterpret_cast<int32_t*>(ptr(r[6], r[7]))};
}
template <int BYTE_OFFSET>
__device__ __forceinline__ void* kernarg_pointer() {                  // one 8-byte pointer argument at a fixed offset
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 r;
    asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(BYTE_OFFSET) : "memory");
    return reinterpret_cast<void*>((uintptr_t)(((unsigned long long)r[1] << 32) | r[0]));
}
// two 4-byte arguments / one double at fixed offsets (the Brownian kick's step0 + seed and sigma: needed once per cycle)
template <int OFF_A, int OFF_B>
__device__ __forceinline__ void kernarg_u32_pair(uint32_t& a, uint32_t& b) {
    asm volatile("s_load_dword %0, %2, %3\n\ts_load_dword %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(OFF_A), "n"(OFF_B) : "memory");
}
template <int BYTE_OFFSET>
__device__ __forceinline__ double kernarg_f64() {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 r;
    asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(BYTE_OFFSET) : "memory");
    return __longlong_as_double((long long)(((unsigned long long)r[1] << 32) | r[0]));
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// Work distribution.  ONE global chunk counter does not work: returning atomics on one address are served at about
// one per 12 ns chip-wide (measured: 78 125 grabs = 0.97 ms for a zero-cycle launch), and every wave's next grab
// queues behind every other wave's.  So the waves form kStreamGroups groups (block id mod kStreamGroups; where a
// wave runs does not matter), group g owns the chunks g, g + G, g + 2G, ... -- evenly spaced samples of the sorted
// cloud, so every group sees the same mix of slow (inlet, fine cells) and fast regions -- and has its own counter,
// 64 bytes from the next; within a group the chunks are dealt first come, first served -- the first one included: a
// wave that only starts when others have finished (a grid larger than what is resident at once: the occupancy
// query was one wave per CU too optimistic in the measurements) then finds its group's counter exhausted and exits,
// instead of sitting on a statically assigned chunk until the end of the launch (measured: a 20 us tail).
// (A group is in effect a CU: blocks go round-robin over 8 XCDs x 32 CUs, so block id mod 256 names the CU while the
// grid is resident at once.  Measured and dropped, tools/stream_timeline.py --groups: the groups' last waves end
// between 122 and 134 us, but dealing the last 10 % of the tiles from counters shared by 4 or 8 neighbouring groups
// moves neither the median nor the last end -- the late waves are single waves on a slow last tile, 3 ... 9.7 us per
// tile -- and 8 counters for the whole chip cost 20 % to contention; giving each XCD runs of 32 consecutive
// chunks so that neighbouring cells meet behind one L2 changes nothing either, on any mesh.)
// The counter set is double-buffered: a launch zeroes the set the NEXT launch uses.
struct StreamArgs {
    unsigned* grab;        // this launch's kStreamGroups counters, kStreamCounterStride apart
    unsigned* grabNext;    // the other set: zeroed here for the next launch on the stream
    int wavesPerGroup;     // grid = kStreamGroups * wavesPerGroup single-wave workgroups
    int tilesPerChunk;
    unsigned bigChunks;    // chunks 0 .. bigChunks-1 have tilesPerChunk tiles; every chunk after them is ONE tile: the tail
                           // of the launch is dealt in small pieces, so that all waves finish within a tile's time of
                           // each other (measured with 4-tile chunks throughout: the last wave 20 us after the median)
    int debug;             // diagnostics only (results are wrong): 1 = no stores, 2 = no loads after a wave's first tile
    double* hitSpill;      // kStreamHitSpillDoubles per workgroup: wall hit points beyond the wave's LDS pool
};

// After a streaming launch: the launch zeroed the OTHER counter set for its successor, so the sets swap roles -- but only
// if the launch really went out.  A failed launch has zeroed nothing: both sets are cleared on the stream instead, so the
// next launch never starts on counters an older launch left exhausted (every wave would exit at once and the particles
// would go unstepped without an error).
inline hipError_t stream_launch_done(hipStream_t st, StreamState& ss) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) { ss.parity ^= 1; return e; }
    (void)hipMemsetAsync(ss.d_grab, 0, kStreamGrabBytes, st);
    return e;
}

}  // namespace cpf
