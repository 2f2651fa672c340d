// Streaming step kernel (gfx950 / CDNA4, wave64): all-hex meshes, and meshes with a minority of other cells (mixed cell
// records: padded, face groups, header records -- the LOOKUP 2 / 3 / 5 instantiations; docs/design_r04.md 5.1 has the table; LOOKUP 6:
// box records, 8 / 9: the flat walk of 2-D cases).
//
// Same fused cycle and the same per-particle arithmetic as step_kernel_coop (cpf_kernels.hip) -- advect -> Brownian
// kick -> plane-exit walk -> wall reflect -> move, src/advect.H:96-161 -- organised around what the measurements of
// that kernel say bounds it (docs/design_r04.md section 5): a wave that walks has no HBM request in flight, so the particle
// stream and the walk ADD instead of overlapping, and every round of the walk pays an L2 round trip for its records.
//
//   * The grid is persistent: single-wave workgroups, as many as the chip holds.  A wave works through CHUNKS of
//     64-particle tiles; chunks are dealt per wave GROUP from per-group counters (see StreamArgs).
//   * While a wave walks tile t, tile t+1 is on its way from HBM straight into the wave's LDS (global_load_lds: no
//     destination register, so nothing the register allocator may copy, spill or reuse before the data lands) and
//     the stores of tile t-1 have just been issued.  All of these are inline assembly: hipcc's s_waitcnt placement
//     is a conservative data-flow over the memory operations it can SEE, and with them visible every round of the walk
//     began with "s_waitcnt vmcnt(0)".  The kernel waits for them itself, once, where a tile ends; the only wait
//     inside the walk is the counted one for records that were missing on chip (vmcnt(number of younger operations)).
//   * The cell records a wave needs live in a per-wave LDS cache whose tags survive from round to round and from
//     tile to tile.  The cloud is kept sorted by (cell, sub-box), a chunk sits in one or two cells and their
//     downstream neighbours, so after a chunk's first tile almost every round finds all its records on chip and
//     issues no memory request at all.  Tags live in ONE vector register (lane k = the cell in slot k): the lookup
//     is a compare against the scalar cell id per distinct cell of the wave -- or, on meshes with few particles per
//     cell, six compares of every lane's cell against the broadcast tags (template parameter LOOKUP_FIXED) -- the
//     update one predicated move.
//
// Crossing particles are not compacted into dense waves: what compaction is meant to buy -- finished lanes' slots
// going to particles whose loads are already in flight -- is what the prefetch does at tile granularity, and the
// sort key (cell, position in the cell's box) keeps the lanes of a tile in step.  (The per-lane version of that idea,
// lanes running ahead into the next tile, is step_kernel_ahead in cpf_ahead.hip: fewer rounds, dearer rounds,
// slower on every mesh -- docs/design_r04.md 5.4.)
#include "cpf_stream_ops.h"

#include <hip/hip_ext.h>

#include <type_traits>

namespace cpf {

#ifndef CPF_STREAM_SLOTS
#define CPF_STREAM_SLOTS 6
#endif
#ifndef CPF_STREAM_SLOTS_FIXED
#define CPF_STREAM_SLOTS_FIXED 9
#endif
#ifndef CPF_STREAM_WAVES
#define CPF_STREAM_WAVES 7
#endif
// Record slots per wave (4..32) and records requested per round, by lookup method.  Many particles per cell (the loop lookup:
// 1-3 distinct cells per round): 6 slots, up to 4 requests.  Few particles per cell (the fixed compare: 5-13 distinct cells per
// tile): 9 slots -- what fits under the LDS a 7th wave leaves, 5632 of 5851 bytes -- and up to 8 requests per round, so that
// fewer lanes sit a round out for want of a slot.  Measured on one box, 6 / 4 -> 9 / 8: 3-D bench box 0.2759 -> 0.2663 ms,
// TJunction 0.1595 / 0.1757 -> 0.1557 / 0.1683, 2.1e6-cell box with 5 particles per cell 0.738 -> 0.682 (12 slots: a wave
// per SIMD less, 0.267 / 0.692); pitzDaily with 9 / 8 LOSES 5 % (0.1190 -> 0.1250), hence per lookup method.
constexpr int kStreamSlots = CPF_STREAM_SLOTS;
constexpr int kStreamSlotsFixed = CPF_STREAM_SLOTS_FIXED;
constexpr int kSlotStride = 8;                          // double4 per slot = one 256-byte record, see `slots`
static_assert(kStreamSlots >= 4 && kStreamSlots <= 32 && kStreamSlotsFixed >= 4 && kStreamSlotsFixed <= 32, "slots");
#ifndef CPF_STREAM_GATHER_LANES
#define CPF_STREAM_GATHER_LANES 32
#endif
constexpr int kStreamGatherLanes = CPF_STREAM_GATHER_LANES;   // lanes without a record slot from which a round gathers (0: always)
// two faces per wave-uniform decision on 3-D meshes (trace_lds6_paired): 4 more registers, i.e. a wave per SIMD less
#ifndef CPF_STREAM_PAIRED
#define CPF_STREAM_PAIRED 0
#endif
// Reflect inside the round (see the round's `again` loop): 0 never, 1 with the Brownian kick only, 2 always
#ifndef CPF_STREAM_INROUND
#define CPF_STREAM_INROUND 0
#endif
#ifndef CPF_STREAM_HIT_POOL
#define CPF_STREAM_HIT_POOL 40
#endif
constexpr int kVertexTets = 12;        // the decomposition the staged locate knows (src/initCuda.H:64: 12 tets per hex); others: per lane
#ifndef CPF_VERTEX_BLOCKS
#define CPF_VERTEX_BLOCKS 4
#endif
#ifndef CPF_VERTEX_PICK_UNROLL
#define CPF_VERTEX_PICK_UNROLL 12
#endif
constexpr int kVertexBlocks = CPF_VERTEX_BLOCKS;       // distinct cells staged per pass
constexpr int kStreamSparsePerCell = 8;                      // fewer particles per cell than this: the sparse instantiation (LOOKUP 4)
constexpr int kSitOut = INT32_MIN + 7;                        // "next cell" of a lane that did not trace this round

// The kernel's parameter list as the kernarg segment lays it out (each parameter at its natural alignment, in order):
// kernarg_cloud_ptrs() / kernarg_pointer<>() read single arguments from there.  Checked against the launch in
// launch_stream_inst (the arguments are passed in this order, nothing else).
struct StreamKernArgs {
    double *x, *y, *z; int32_t* cell; const int64_t* gid; double* vel; int64_t n; double dt, sigma; uint32_t step0; int nCyc;
    uint32_t seed; MeshView m; unsigned long long* counters; StreamArgs sa;
};
static_assert(offsetof(StreamKernArgs, x) == 0 && offsetof(StreamKernArgs, cell) == 24, "kernarg_cloud_ptrs reads bytes 0..31");
constexpr int kKernArgHitSpill = (int)(offsetof(StreamKernArgs, sa) + offsetof(StreamArgs, hitSpill));
// the Brownian kick's own arguments are read from the kernarg segment where they are needed -- the particle ids once per tile,
// step0 / seed / sigma once per cycle -- instead of holding seven scalar registers for the whole kernel (CPF_STREAM_BROWN_KERNARG)
constexpr int kKernArgGid = (int)offsetof(StreamKernArgs, gid), kKernArgSigma = (int)offsetof(StreamKernArgs, sigma);
constexpr int kKernArgStep0 = (int)offsetof(StreamKernArgs, step0), kKernArgSeed = (int)offsetof(StreamKernArgs, seed);
#ifndef CPF_STREAM_BROWN_KERNARG
#define CPF_STREAM_BROWN_KERNARG 1
#endif

// the zero-denominator skip of a face (cpf_walk.h) in the fixed-lookup instantiation: few particles per cell = a 3-D mesh, where no
// face is parallel to everybody's displacement (the z pair of a layered mesh is dropped by its own test either way)
#ifndef CPF_STREAM_L1_ZERO_SKIP
#define CPF_STREAM_L1_ZERO_SKIP 1
#endif
#ifndef CPF_STREAM_LAZY_USED
#define CPF_STREAM_LAZY_USED 1
#endif
#ifndef CPF_STREAM_BOX_SPARSE
#define CPF_STREAM_BOX_SPARSE 1
#endif
// the zero-denominator vote per face in the flat walk: off -- a 2-D mesh whose side faces are exactly parallel to everybody's
// displacement is a mesh of boxes in axis-aligned flow; pitzDaily's trapezoids never are, and the vote costs the headline 2.3 %
// (0.1048-0.1063 -> 0.1032-0.1033 ms, analytic field 0.0986-0.0999 -> 0.0985-0.0986).  Same results either way.
// the flat walk under the Brownian kick (cpf_walk.h, trace_lds4_flat_z): 0 = off (A/B builds)
#ifndef CPF_STREAM_FLAT_KICK
#define CPF_STREAM_FLAT_KICK 0
#endif
#ifndef CPF_STREAM_FLAT_ZERO_SKIP
#define CPF_STREAM_FLAT_ZERO_SKIP 0
#endif
#ifndef CPF_STREAM_SLOTS_BOX
#define CPF_STREAM_SLOTS_BOX 9
#endif
#ifndef CPF_STREAM_SLOTS_BOX_B
#define CPF_STREAM_SLOTS_BOX_B 12
#endif
#ifndef CPF_STREAM_WAVES_BOX_B
#define CPF_STREAM_WAVES_BOX_B 6
#endif
#ifndef CPF_STREAM_WAVES_BOX
#define CPF_STREAM_WAVES_BOX 6
#endif
#ifndef CPF_STREAM_WAVES_B1
#define CPF_STREAM_WAVES_B1 5
#endif
#ifndef CPF_STREAM_WAVES_L2
#define CPF_STREAM_WAVES_L2 6
#endif
// the Brownian loop-lookup instantiation (pitzDaily with the tutorials' diffusion coefficient): 7 waves per SIMD since round 5 --
// 70 VGPRs once the kick's own arguments are read from the kernarg segment (no scalar spills into vector lanes), and a hit
// pool of 18 instead of 40 entries keeps the wave's LDS at 5816 of 5851 bytes.  Measured with Philox-7, one box: 0.1865 -> 0.181 ms
#ifndef CPF_STREAM_WAVES_B0
#define CPF_STREAM_WAVES_B0 7
#endif
#ifndef CPF_STREAM_HIT_POOL_B0
#define CPF_STREAM_HIT_POOL_B0 18
#endif
// the flat walk with the loop lookup (the headline): 54 VGPRs / 84 SGPRs / 4864 B would allow an eighth wave (A/B: CPF_STREAM_WAVES_FLAT)
#ifndef CPF_STREAM_WAVES_FLAT
#define CPF_STREAM_WAVES_FLAT CPF_STREAM_WAVES
#endif
template <bool BROWNIAN, bool STORE_VEL, bool STATS, int LOOKUP>
// LOOKUP 2 / 3 (mixed records): one wave less; LOOKUP 4 (sparse clouds: pipelined per-lane gathers, 24 more registers): 5
struct StreamOccupancy {
    static constexpr bool kMixed = LOOKUP == 2 || LOOKUP == 3 || LOOKUP == 5 || LOOKUP == 11;
    // (LOOKUP 2 carries the state of a half-done visit of a two-record cell: 80 registers, exactly what 6 waves allow)
    static constexpr int waves = (STORE_VEL || STATS) ? 1 : (LOOKUP == 4 ? (BROWNIAN ? 4 : 5) : (BROWNIAN ? (kMixed ? 5 : ((LOOKUP == 6 || LOOKUP == 11) ? CPF_STREAM_WAVES_BOX_B : (LOOKUP == 1 ? CPF_STREAM_WAVES_B1 : CPF_STREAM_WAVES_B0))) : (LOOKUP == 2 ? CPF_STREAM_WAVES_L2 : ((kMixed && LOOKUP != 11) ? 6 : ((LOOKUP == 6 || LOOKUP == 11) ? CPF_STREAM_WAVES_BOX : (LOOKUP == 8 ? CPF_STREAM_WAVES_FLAT : CPF_STREAM_WAVES))))));
};

// The kernel's body.  VERTEX: the advect takes the velocity INTERPOLATED at the particle from tet-vertex values (the reference's
// "VertexVelocity" mode, cuda/particles.cu:244-313; vertex_velocity() in cpf_walk.h, per-lane reads of the tet tables from L2)
// instead of the record's cell-constant one; everything else -- the record cache, the walk, the prefetch -- is the same code.
// Two kernels wrap it (below): step_kernel_stream, whose name and parameter list every profile and test of rounds 2-5 knows,
// and step_kernel_stream_vertex, which appends the VertexField to the same parameter list (the kernarg offsets stay valid).
template <bool BROWNIAN, bool REFLECT, bool STORE_VEL, bool STATS, int LOOKUP, bool VERTEX>
__device__ __forceinline__ void stream_body(
    const int64_t* __restrict__ gid, double* __restrict__ vel, int64_t n, double dt, double sigma, uint32_t step0,
    int nCyc, uint32_t seed, const MeshView& m, unsigned long long* __restrict__ counters, const StreamArgs& sa, const VertexField& vf) {
    // (with the kick the landing zone and the hit pool are larger: 7 slots keep the sixth wave, 6600 of 6826 bytes)
#ifndef CPF_STREAM_SLOTS_BROWN
#define CPF_STREAM_SLOTS_BROWN 10
#endif
    constexpr int NS = (LOOKUP == 0 || LOOKUP == 5 || LOOKUP == 8) ? kStreamSlots : ((LOOKUP == 6 || LOOKUP == 11) ? (BROWNIAN ? CPF_STREAM_SLOTS_BOX_B : CPF_STREAM_SLOTS_BOX) : (BROWNIAN && LOOKUP == 1 ? CPF_STREAM_SLOTS_BROWN : kStreamSlotsFixed));
    // LOOKUP 6: the mesh's 128-byte BOX records instead of the 256-byte ones (cpf_walk.h "box records")
    constexpr bool BOX = LOOKUP == 6 || LOOKUP == 11;             // (11: ... on a mesh with face groups: 2:1-refined boxes)
    constexpr int kStride = BOX ? 4 : kSlotStride;               // double4 per slot
    constexpr unsigned kRecBytes = 32u * kStride;
    constexpr unsigned ALL = NS == 32 ? 0xFFFFFFFFu : ((1u << NS) - 1u);
    // the wave's record cache.  (256 bytes per slot is one full turn of the 64 LDS banks, so lanes reading the same plane
    // of different slots conflict: 50 conflict cycles per tile on pitzDaily, 351 on the 3-D bench mesh.  Padding the
    // slots to 288 bytes removes the conflicts and was SLOWER on every mesh, 3-4 %: the 192 bytes cost the 24th wave per
    // CU, and with 5 padded slots instead of 6 -- same LDS as now -- the extra misses cost more than the conflicts.)
    __shared__ double4 slots[NS][kStride];
    __shared__ unsigned sCnt[4];
    // Per-lane end point E, parked between rounds (see step_kernel_coop) -- the slot also carries the three Brownian
    // deviates from the first round of a cycle to the lane's advect, and the previous tile's position from the tile's end
    // to the next tile's hook.  The LAST WALL HIT POINT of a reflected lane (read once, by the move at the cycle's end)
    // used to have 64 x 24 bytes of LDS as well; those 1.5 KB are what stood between this kernel and a 7th wave per SIMD
    // (160 KB / 28 waves = 5851 B).  Two ways out, chosen per instantiation by what it has room for:
    //   * HIT_IN_REGS (no Brownian kick, all-hex mesh): the hit point stays in six VGPRs -- the scalar-register diet and
    //     the single round instance left exactly that room under the 72 registers 7 waves allow;
    //   * otherwise a small per-wave POOL in LDS: a lane takes an entry at its first reflection of the cycle (LDS counter)
    //     and entries beyond the pool go to the wave's own 1.5 KB of global memory (sa.hitSpill).  (A pool of 10 for
    //     every instantiation, tried first, cost bench.py's window 3 %: there the cloud drifts into the outlet wall
    //     and whole tiles reflect, 54 of 64 lanes through the spill path.)
#ifndef CPF_STREAM_HIT_REGS_L2
#define CPF_STREAM_HIT_REGS_L2 0
#endif
    constexpr bool HIT_IN_REGS = !BROWNIAN && (LOOKUP != 2 || CPF_STREAM_HIT_REGS_L2);
    constexpr bool kInRound = CPF_STREAM_INROUND == 2 || (CPF_STREAM_INROUND == 1 && BROWNIAN);
    static_assert(!(kInRound && (LOOKUP == 2 || LOOKUP == 3 || LOOKUP == 5 || LOOKUP == 11)), "in-round reflection knows neither face groups nor two-record cells");
    constexpr int kPool = HIT_IN_REGS ? 1 : ((LOOKUP == 2 || LOOKUP == 3 || LOOKUP == 5 || LOOKUP == 11) ? 16 : ((BROWNIAN && LOOKUP == 0) ? CPF_STREAM_HIT_POOL_B0 : CPF_STREAM_HIT_POOL));
    __shared__ double sLane[3][64];
    __shared__ double sPool[3][kPool];
    __shared__ unsigned sPoolUsed;
    // landing zone of the next tile: x[64] | y[64] | z[64] | cell[64] (int32) | gid[64] (Brownian only)
    __shared__ double sPre[BROWNIAN ? 288 : 224];
    // (VERTEX) the cone rows + 1/det of the 12 tets of up to kVertexBlocks cells: 80 bytes a tet, staged once per cycle and distinct cell of the wave
    __shared__ double sCone[VERTEX ? kVertexBlocks : 1][VERTEX ? kVertexTets * 10 : 1];
    double(*sE)[64] = sLane;
    const int lane = threadIdx.x;
    const unsigned ul = threadIdx.x;                 // unsigned lane index: scalar base + 32-bit lane offset addressing
    const unsigned preBase = uniform32(lds_addr(sPre));
    const unsigned slotBase = uniform32(lds_addr(slots));
    const int tpc = sa.tilesPerChunk;
    // LOOKUP: how a wave finds its cells in its record cache, and which records can turn up --
    //   0 loop over the distinct cells of the wave, 1 fixed tag compare, 2 / 3 fixed tag compare on a mesh that is not
    //   all-hex (MeshView::mixed): 3 = every cell has at most six slots -- face groups and padded records only, the usual
    //   2:1-refined hex mesh --, 2 = header records of cells with more than six slots may turn up as well
    //   4 = fixed tag compare for SPARSE clouds on all-hex meshes (fewer than kStreamSparsePerCell particles per cell: nearly
    //   every lane of a tile sits in a cell of its own and walks by per-lane gathers from L2 / HBM): the gather walk keeps three
    //   planes in flight instead of one -- two dependent round trips per visit instead of six -- for 24 more registers, i.e.
    //   five waves per SIMD.  2.1e6-cell box: 1.25e6 particles (one rank's share of BASELINE configs[4]) 0.187 -> 0.141 ms,
    //   1e7 particles 0.673 / 0.631 -> 0.641 / 0.607; in the dense regime the same change costs 6 % (0.259 -> 0.277)
    //   5 = as 3 (mixed records without header records) with the LOOP lookup and six slots: a refined mesh that still holds
    //   hundreds of particles per cell (pitzDaily with a 2:1 patch: 0.157 -> see docs/design_r04.md 5.6)
    //   6 = as 1 on the mesh's 128-byte BOX records (every cell an axis-aligned box: cpf_walk.h "box records") -- dense and sparse
    //   clouds alike: three candidate faces per visit, one LDS round trip per record, one cache line per gathered record
    //   8 = as 0 with the FLAT walk (cpf_walk.h): a 2-D mesh extruded straight in z, a field without a z component, no kick
    constexpr bool LOOKUP_FIXED = LOOKUP != 0 && LOOKUP != 5 && LOOKUP != 8;
    //   9 = as 1 with the flat walk (a 2-D mesh with fewer than 128 particles per cell: refined 2-D cases)
    constexpr bool FLAT = LOOKUP == 8 || LOOKUP == 9;
    constexpr bool mixed = LOOKUP == 2 || LOOKUP == 3 || LOOKUP == 5 || LOOKUP == 11;
    constexpr bool bigCells = LOOKUP == 2;
    constexpr int kGatherAhead = LOOKUP == 4 ? 3 : 0;
    const bool zFold = !BOX && BROWNIAN && REFLECT && m.zThin != 0;      // (stream_lookup_mode: no box records on a mesh one cell thick)
    const bool zLast = !FLAT && !BROWNIAN && m.zPairLast != 0;   // (with the kick every particle moves in z: the test would be wasted)
    // the kick on a one-cell-thick mesh whose side faces have nz == 0 exactly: the four side faces with two-term dot products
    // whenever every lane's mirrored end point is clear of the z planes (cpf_walk.h, trace_lds4_flat_z)
    const bool flatKick = zFold && m.zSide0 != 0;
    // tile and chunk numbers are 32-bit (the launcher refuses clouds of 2^31 tiles = 1.4e11 particles): half the scalar
    // registers and none of the 64-bit multiply sequences of the first version.  Chunk numbers past the end of the cloud
    // (a group's counter keeps counting) saturate instead of wrapping.
    const unsigned nTiles = (unsigned)((n + 63) >> 6);
    const unsigned nFull = (unsigned)(n >> 6), nTail = (unsigned)n & 63u;     // full tiles, particles in the last, partial one
    const unsigned big = sa.bigChunks;
    const unsigned nChunks = big + (nTiles - big * (unsigned)tpc);
    // first tile of chunk c, and how many of its tiles exist (0: the chunk lies past the end of the cloud)
    // (Measured and dropped in round 3: dealing the chunks from the END of the cloud towards its start, so that the launch
    // finishes on the inlet's tiles instead of the outlet wall's -- 0.1245 ms either way.)
    auto chunk_first = [&](unsigned c) -> unsigned {
        return c < big ? c * (unsigned)tpc : big * (unsigned)tpc + (c - big);
    };
    auto chunk_tiles = [&](unsigned c, unsigned firstTile) -> int {
        if (c >= nChunks) return 0;
        const unsigned k = nTiles - firstTile;
        const unsigned want = c < big ? (unsigned)tpc : 1u;
        return (int)(k < want ? k : want);                     // (k < want never happens: chunks tile the cloud exactly)
    };
    // chunk j of the group (j = what the group's counter returns) is chunk grp + G*j of the cloud
    // (Measured and dropped, rounds 2 and 4: XCD x taking RUNS of 32 consecutive chunks, one per CU, so that the records of
    // neighbouring cells meet behind one L2 -- nothing on the dense 3-D mesh, nothing on the sparse one: 0.1647 / 0.1641 against
    // 0.1635 / 0.1636 ms; an XCD's 640 waves work on 50 MB of records at a time, an L2 holds 4.)
    auto chunk_of = [&](unsigned j) -> unsigned {
        const unsigned long long c = (unsigned long long)(blockIdx.x & (kStreamGroups - 1)) + (unsigned long long)kStreamGroups * j;
        return c < (unsigned long long)nChunks ? (unsigned)c : nChunks;
    };
    StepStats st = {0, 0, 0, 0};
#ifdef CPF_STREAM_TIMELINE
    // diagnostic build only (tools/stream_timeline.py): per wave start / end time (100 MHz), tiles and rounds done,
    // written to the `vel` array of a launch that does not store velocities
    const uint64_t tl0 = __builtin_amdgcn_s_memrealtime();
    unsigned tlTiles = 0, tlRounds = 0;
    // time (100 MHz ticks) spent in the wait for missing records / in the wait that ends a tile, and rounds with a miss
    unsigned tlWaitRec = 0, tlWaitEnd = 0, tlMissRounds = 0;
    // census of the rounds (tools/stream_timeline.py --census), behind the per-wave rows in the same array: busy lanes per
    // round index, lanes that sat a round out, rounds per tile against the tile's largest visit count
    unsigned long long* const tlH = (!STORE_VEL && vel != nullptr && (sa.debug & 4)) ? reinterpret_cast<unsigned long long*>(vel) + 8 * 16384 : nullptr;
    unsigned tlVis = 0, tlRoundIdx = 0;
#endif

    for (unsigned k = blockIdx.x; k < (unsigned)kStreamGroups; k += gridDim.x)
        if (lane == 0) sa.grabNext[k * kStreamCounterStride] = 0u;
    const unsigned grp = blockIdx.x & (kStreamGroups - 1);
    unsigned* const myGrab = sa.grab + grp * kStreamCounterStride;

    // Requests tile t: full tiles by LDS-DMA (returns the number of memory operations issued), the cloud's last,
    // partial tile by ordinary masked loads (complete when the function returns: 0).
    auto prefetch = [&](unsigned t, const CloudPtrs& cp) __attribute__((always_inline)) -> int {
        double* const x = cp.x; double* const y = cp.y; double* const z = cp.z; int32_t* const cell = cp.cell;
        t = uniform32(t);
        const int64_t b = (int64_t)t * 64;
        // (the lane number through an opaque copy: the per-lane source offsets below are then recomputed per tile -- three
        // vector instructions -- instead of being hoisted out of the tile loop into registers that live for the whole kernel)
        unsigned ul = threadIdx.x;
        asm volatile("" : "+v"(ul));
        if (t < nFull) {
            const char* s1 = (ul < 32u ? reinterpret_cast<const char*>(x + b) : reinterpret_cast<const char*>(y + b)) + (ul & 31u) * 16u;
            glds16(s1, preBase);                                               // x -> [0, 512), y -> [512, 1024)
            if (ul < 48u) {
                const char* s2 = ul < 32u ? reinterpret_cast<const char*>(z + b) + ul * 16u
                                          : reinterpret_cast<const char*>(cell + b) + (ul - 32u) * 16u;
                glds16(s2, preBase + 1024u);                                   // z -> [1024, 1536), cell -> [1536, 1792)
            }
            if (BROWNIAN) {
                const int64_t* const g = CPF_STREAM_BROWN_KERNARG ? static_cast<const int64_t*>(kernarg_pointer<kKernArgGid>()) : gid;
                if (g != nullptr) {
                    if (ul < 32u) glds16(reinterpret_cast<const char*>(g + b) + ul * 16u, preBase + 1792u);
                    return 3;
                }
            }
            return 2;
        }
        const unsigned lim = nTail - 1u;
        const unsigned l = ul < lim ? ul : lim;
        const double vx = (x + b)[l], vy = (y + b)[l], vz = (z + b)[l];
        const int vc = (cell + b)[l];
        sPre[ul] = vx; sPre[64 + ul] = vy; sPre[128 + ul] = vz;
        reinterpret_cast<int*>(sPre + 192)[ul] = vc;
        if (BROWNIAN) {
            const int64_t* const g = CPF_STREAM_BROWN_KERNARG ? static_cast<const int64_t*>(kernarg_pointer<kKernArgGid>()) : gid;
            if (g != nullptr) reinterpret_cast<int64_t*>(sPre + 224)[ul] = (g + b)[l];
        }
        return 0;
    };

    unsigned got0 = 0;
    if (lane == 0) got0 = grab_sync(myGrab);
    const unsigned chunk0 = chunk_of(uniform32(got0));
    if (chunk_tiles(chunk0, chunk_first(chunk0)) > 0) {
        // tags of the record cache: lane k holds the cell id whose record sits in slot k (-1: none)
        int tagv = -1;
        unsigned fifo = 0;

        unsigned tile = chunk_first(chunk0);
        int tilesLeft = chunk_tiles(chunk0, tile);
        (void)prefetch(tile, kernarg_cloud_ptrs());
        wait_vmcnt<0>();

        // results of the previous tile, stored one tile late.  Its positions wait in the lanes' parked-E slots (sE): a
        // slot is free from the end of a tile's last cycle to the next tile's advect, and every round begins by reading
        // it anyway (Epre) -- six registers that are not live across the walk.
        bool havePrev = false;
        unsigned rtile = 0;
        double rvx = 0, rvy = 0, rvz = 0;
        bool ralive = false;                         // (stored velocity) the lane carried a live particle when the tile was LOADED
        int rc = 0;
        unsigned rlim = 63;

        for (;;) {
            tile = uniform32(tile);
            // ---- take the tile out of the landing zone
            const unsigned plim = tile < nFull ? 63u : nTail - 1u;      // last lane with a particle slot
            double px = sPre[ul], py = sPre[64 + ul], pz = sPre[128 + ul];
            int pc = reinterpret_cast<const int*>(sPre + 192)[ul];
            uint64_t pid = 0;
            if (BROWNIAN) {
                const bool haveIds = (CPF_STREAM_BROWN_KERNARG ? kernarg_pointer<kKernArgGid>() : (void*)gid) != nullptr;
                pid = haveIds ? (uint64_t) reinterpret_cast<const int64_t*>(sPre + 224)[ul] : (uint64_t)tile * 64 + ul;
            }
            // no particle in this lane (beyond the cloud's end, frozen, or lost in an earlier step: w = 0 from now on,
            // cuda/particles.cu:333-338): CPF_CELL_FROZEN, and the lane stores back what it loaded.  Which lanes carry a
            // particle is read off `cur` wherever it is needed (cur >= 0) instead of being kept in wave masks: scalar
            // registers are what this kernel is shortest of.
            if (ul > plim || pc < 0) pc = CPF_CELL_FROZEN;

            int cur = pc;
            // (meshes with two-record cells, LOOKUP 2) a lane between the two halves of a visit -- key2 != 0: the first record has
            // been tested, the second one (record index key2 >= nCells > 0) is what the lane looks up next -- carries
            // (carDT, carNext, carBest) over
            int key2 = 0, carNext = 0, carBest = -1;
            double carDT = 2.0;
            // the particle's position IS the walk's running start point S_: between two cycles (and for a lane without an
            // active particle, always) S_ holds the position; a cycle's walk advances it from crossing to crossing and the
            // move at the cycle's end overwrites it (one register triple, not two)
            D3 S_ = {px, py, pz}, v = {0, 0, 0};
            const uint64_t id = pid;

            // where the wave goes next
            unsigned ntile = 0;
            int ntilesLeft = 0;                      // 0: this was the wave's last tile

            // Once per tile, right after round 1's missing records have been REQUESTED: the next chunk if this is the
            // chunk's last tile, the previous tile's stores, the next tile's loads.  Returns how many of these
            // operations are certainly younger than the record requests: s_waitcnt vmcnt(that many) then lets
            // exactly the records complete.
            auto hook = [&](const D3& prev) __attribute__((always_inline)) -> int {
                int younger = 0;
                __builtin_amdgcn_sched_barrier(0);
                if (tilesLeft > 1) {
                    ntile = tile + 1;
                    ntilesLeft = tilesLeft - 1;
                } else {
                    unsigned got = 0;
                    if (lane == 0) got = grab_sync(myGrab);
                    const unsigned nxt = chunk_of(uniform32(got));
                    ntile = chunk_first(nxt);
                    ntilesLeft = chunk_tiles(nxt, ntile);
                }
                const CloudPtrs cp = kernarg_cloud_ptrs();
                double* const x = cp.x; double* const y = cp.y; double* const z = cp.z; int32_t* const cell = cp.cell;
                if (havePrev && !(sa.debug & 1)) {
                    const int64_t b = (int64_t)uniform32(rtile) * 64;
                    // (every particle that was alive when the launch began gets its velocity stored -- of its last live cycle if
                    // it was lost on the way, like the reference's vels[] after an advect that skips it, cuda/particles.cu:333-338,
                    // and like the other step kernels; round 4: this kernel used to skip particles lost DURING a fused launch)
                    if (STORE_VEL && ralive) {
                        double* vv = vel + 3 * b;
                        async_store(vv, ul * 24u, rvx); async_store(vv + 1, ul * 24u, rvy); async_store(vv + 2, ul * 24u, rvz);
                    }
                    if (ul <= rlim) {
                        async_store(x + b, ul * 8u, prev.x); async_store(y + b, ul * 8u, prev.y); async_store(z + b, ul * 8u, prev.z);
                        async_store(cell + b, ul * 4u, rc);
                    }
                    younger += 4;
                }
                if (ntilesLeft > 0 && !(sa.debug & 2)) younger += prefetch(ntile, cp);
                __builtin_amdgcn_sched_barrier(0);
                return younger;
            };

            // ---- per-cycle walk state (step_kernel_coop's state machine).  Only `busy` is a wave mask; the others are
            // read off the lane's own registers: a lane has yet to advect while its token is still INT32_MIN (no face
            // crossed or hit in this cycle), it has been reflected iff j != 0, and it is lost in this cycle iff
            // j >= kMaxReflect (still on a wall after 5 bounces; a wall without reflection sets j = kMaxReflect too).
            bool busy = false;
            bool zUnclear = false;                   // (kick on a one-cell-thick mesh) some lane's mirrored end point is not clear of the z planes
            int token = INT32_MIN, h = 0, j = 0;
            int hitAt = -1;                          // pool: where this lane's hit point is parked (< kPool: pool entry)
            // parks the wall hit point of a lane that is being reflected (it is read back once, by the move at cycle end)
            D3 hitReg = {0, 0, 0};                   // HIT_IN_REGS: the last wall hit point
            auto park_hit = [&](const D3& Hp) __attribute__((always_inline)) {
                if (HIT_IN_REGS) { hitReg = Hp; return; }
                if (hitAt < 0) hitAt = (int)atomicAdd(&sPoolUsed, 1u);          // first reflection of the cycle
                if (hitAt < kPool) { sPool[0][hitAt] = Hp.x; sPool[1][hitAt] = Hp.y; sPool[2][hitAt] = Hp.z; }
                else {
                    double* sp = static_cast<double*>(kernarg_pointer<kKernArgHitSpill>()) + (size_t)blockIdx.x * kStreamHitSpillDoubles;
                    async_store(sp, ul * 8u, Hp.x); async_store(sp + 64, ul * 8u, Hp.y); async_store(sp + 128, ul * 8u, Hp.z);
                }
            };

            auto cycle_begin = [&](int c) __attribute__((always_inline)) {
                if (cur < 0) cur = CPF_CELL_FROZEN;                                  // lost in the previous cycle: w = 0
                busy = cur >= 0;
                token = INT32_MIN; h = 0; j = 0;
                if (bigCells) key2 = 0;
                if (!HIT_IN_REGS && REFLECT) { hitAt = -1; if (lane == 0) sPoolUsed = 0u; }
                zUnclear = false;
                if (STATS && busy) ++st.steps;
                if (VERTEX) {
                    // ---- the cycle's velocities, HERE where the wave is whole: the interpolated velocity depends on the position
                    // and the cell the particle begins the cycle with, nothing else.  The distinct cells of the wave (a sorted
                    // cloud: one to three) get an LDS block each, kVertexBlocks per pass; the lanes fetch the blocks' cone rows together
                    // -- lane l the l-th 16 bytes of each, ONE L2 round trip for all tets of all those cells instead of one per
                    // tet and lane --, every lane picks its tet from its cell's block (broadcast reads) and evaluates that ONE
                    // tet's record.  vertex_velocity() (cpf_walk.h) states the rule and takes the lanes the shortcut does not
                    // cover, and every lane of a decomposition that is not twelve tets a cell.  (The first streaming version
                    // called it per lane in the advect: twelve dependent round trips per particle, 0.50 ms against the
                    // cell-constant cycle's 0.11; one cell per pass with the pick inside: 0.40.)
                    bool slow = busy;
#if defined(CPF_VERTEX_AB) && CPF_VERTEX_AB == 1        // A/B builds only (WRONG results): nothing of the interpolation, the cell's own U
                    if (busy) { const double4 u = m.U[cur]; v = {u.x, u.y, u.z}; slow = false; }
#endif
                    if (vf.tetsPerCell == kVertexTets) {
                        unsigned long long todo = ballot64(slow);
                        while (todo != 0ull) {
                            int myBlock = -1;
                            __syncthreads();                                   // (one wave: orders the LDS traffic of two passes)
#pragma unroll
                            for (int q = 0; q < kVertexBlocks; ++q) {
                                if (todo != 0ull) {
                                    const int ck = __builtin_amdgcn_readlane(cur, __ffsll((long long)todo) - 1);
                                    const bool mine = slow && cur == ck;
                                    if (mine) myBlock = q;
                                    if (lane < 5 * kVertexTets) {
                                        const int tk = lane / 5, piece = lane - 5 * tk;
                                        const double2 row = reinterpret_cast<const double2*>(vf.cone + kConeDoubles * ((int64_t)ck * kVertexTets + tk))[piece];
                                        reinterpret_cast<double2*>(&sCone[q][0] + 10 * tk)[piece] = row;
                                    }
                                    todo &= ~ballot64(mine);
                                }
                            }
                            __syncthreads();
                            if (myBlock >= 0) {
                                const double4 ap = vf.apex[cur];
                                const D3 A = {ap.x, ap.y, ap.z};
                                const D3 r = {S_.x - A.x, S_.y - A.y, S_.z - A.z};
                                int cand = 0;
                                double candMin = -1e300;
                                const double* blk = &sCone[myBlock][0];
#pragma unroll CPF_VERTEX_PICK_UNROLL
                                for (int k = 0; k < kVertexTets; ++k) {
                                    const double2* g = reinterpret_cast<const double2*>(blk + 10 * k);
                                    const double2 g01 = g[0], g23 = g[1], g45 = g[2], g67 = g[3], g89 = g[4];
                                    const double cb = fma(g23.x, r.z, fma(g01.y, r.y, g01.x * r.x)), cc = fma(g45.y, r.z, fma(g45.x, r.y, g23.y * r.x));
                                    const double cd = fma(g89.x, r.z, fma(g67.y, r.y, g67.x * r.x));
                                    const double mm = fmin(cb, fmin(cc, cd));
                                    if (mm > candMin) { candMin = mm; cand = k; }
                                }
#if defined(CPF_VERTEX_AB) && CPF_VERTEX_AB == 2        // A/B builds only (WRONG results): staging + pick, no evaluation; the cell's own U moves the particle
                                { const double4 u = m.U[cur]; v = {u.x + 1e-300 * cand, u.y, u.z}; slow = false; }
#else
                                D3 vv;
                                if (vertex_velocity_in_tet(vf, S_, A, (int64_t)cur * kVertexTets + cand, vv)) { v = vv; slow = false; }
#endif
                            }
                        }
                    }
                    if (slow) (void)vertex_velocity(vf, S_, cur, v);           // within the margin of a tet's boundary: all tets, as ever
                }
            };

            auto round = [&](bool hookDue, bool cycleStart, int c) __attribute__((always_inline)) {
#ifdef CPF_STREAM_TIMELINE
                ++tlRounds;
                bool tlSat = false;
#endif
                // ---- record cache lookup, two ways (LOOKUP_FIXED, chosen per launch by stream_lookup_mode()).  Few particles per cell
                // (3-D meshes: 5-10 distinct cells per round): every lane compares its cell with the NS tags (tag k
                // broadcast from lane k of tagv), a fixed, branch-free sequence of 3 vector instructions per tag.  Many
                // particles per cell (pitzDaily: 1-3 distinct cells per round): one scalar iteration per DISTINCT cell
                // of the busy lanes, ~7 vector + 14 scalar instructions each -- fewer VECTOR instructions there, and those
                // are what that case is short of (measured: the fixed sequence costs pitzDaily 1.5-3 %, and saves the
                // large meshes 4-7 %; both in ONE kernel behind a run-time flag: the worse of the two everywhere, hence
                // the template parameter).  A round that finds every cell on chip (the common case) issues no memory request.
                const unsigned long long busyMask = ballot64(busy);
                // the lane's parked end point, requested before the lookup: its LDS round trip hides behind it (1 %)
                const D3 Epre = {sE[0][lane], sE[1][lane], sE[2][lane]};
                // what the lane looks up: its cell -- or, between the two halves of a visit of a two-record cell, the second record
                const int lk = (bigCells && key2 != 0) ? key2 : cur;
                int myslot = -1;
                unsigned used = 0;
                unsigned long long missLanes = busyMask;
                if (LOOKUP_FIXED) {
                    // (one cell for the whole wave -- nine first rounds of ten on a mesh with hundreds of particles per cell
                    // -- needs one compare against the tag vector, not six against the cells)
                    const int cell0 = __builtin_amdgcn_readlane(lk, __ffsll((long long)busyMask) - 1);
                    const unsigned long long in0 = __builtin_amdgcn_uicmp((unsigned)lk, (unsigned)cell0, 32 /* eq */) & busyMask;
                    if (in0 == busyMask) {
                        const unsigned long long hitTag = __builtin_amdgcn_uicmp((unsigned)tagv, (unsigned)cell0, 32 /* eq */) & ALL;
                        if (hitTag != 0ull) {
                            myslot = __ffsll((long long)hitTag) - 1;
                            used = (unsigned)hitTag;
                            missLanes = 0ull;
                        }
                    } else {
#if CPF_STREAM_LAZY_USED
                        // the sweep itself is vector work only; which slots are in use (`used`) is scalar work -- four
                        // operations per tag -- that only a round with a miss needs (the victims of its jobs)
                        unsigned long long eqs[NS];
#pragma unroll
                        for (int k = 0; k < NS; ++k) {
                            const int tk = __builtin_amdgcn_readlane(tagv, k);
                            eqs[k] = __builtin_amdgcn_uicmp((unsigned)lk, (unsigned)tk, 32 /* eq */);
                            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(myslot) : "v"(myslot), "n"(k), "s"(eqs[k]));
                        }
                        missLanes = ballot64(myslot < 0) & busyMask;
                        if (missLanes != 0ull) {
#pragma unroll
                            for (int k = 0; k < NS; ++k) used |= ((eqs[k] & busyMask) != 0ull ? 1u : 0u) << k;
                        }
#else
#pragma unroll
                        for (int k = 0; k < NS; ++k) {
                            const int tk = __builtin_amdgcn_readlane(tagv, k);
                            const unsigned long long eqK = __builtin_amdgcn_uicmp((unsigned)lk, (unsigned)tk, 32 /* eq */);
                            const unsigned long long inK = eqK & busyMask;
                            // (one select per tag on the compare's own mask; left alone, hipcc issues a second, inverted
                            // compare per tag to get the constant into the operand slot it prefers)
                            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(myslot) : "v"(myslot), "n"(k), "s"(eqK));
                            used |= (inK != 0ull ? 1u : 0u) << k;
                            missLanes &= ~inK;
                        }
#endif
                    }
                } else {
                    unsigned long long todo = busyMask;
                    missLanes = 0ull;
                    while (todo != 0ull) {
                        const int leader = __ffsll((long long)todo) - 1;
                        const int ck = __builtin_amdgcn_readlane(lk, leader);
                        const bool mine = lk == ck;
                        const unsigned long long same = ballot64(mine) & busyMask;
                        const unsigned long long hitTag = __builtin_amdgcn_uicmp((unsigned)tagv, (unsigned)ck, 32 /* eq */);
                        if (hitTag != 0ull) {
                            const int sl = __ffsll((long long)hitTag) - 1;
                            used |= 1u << sl;
                            if (mine) myslot = sl;
                        } else {
                            missLanes |= same;
                        }
                        todo &= ~same;
                    }
                }
                // ---- misses: up to four (fixed lookup: eight) records per round, each a 256-byte LDS-DMA by lanes 0..15 straight into its
                // slot; victims are taken oldest-first among the slots nobody reads this round
                int nJobs = 0;
                if (missLanes != 0ull) {
#define CPF_JOB(J)                                                                                          \
                    if (missLanes != 0ull && used != ALL) {                                                 \
                        const int leader = __ffsll((long long)missLanes) - 1;                               \
                        const int ck = __builtin_amdgcn_readlane(lk, leader);                               \
                        const unsigned cand = ~used & ALL;                                                  \
                        const unsigned hi = cand & (ALL << fifo) & ALL;                                     \
                        const int victim = __ffs((int)(hi ? hi : cand)) - 1;                                \
                        fifo = (unsigned)(victim + 1) == (unsigned)NS ? 0u : (unsigned)(victim + 1);          \
                        used |= 1u << victim;                                                               \
                        if (lane == victim) tagv = ck;                                                      \
                        const bool mine = lk == ck;                                                         \
                        if (mine) myslot = victim;                                                          \
                        missLanes &= ~ballot64(mine);                                                       \
                        if (ul < kRecBytes / 16u)                                                            \
                            glds16(reinterpret_cast<const char*>(BOX ? m.boxRec : m.cellRec) + (int64_t)ck * kRecBytes + ul * 16u,  \
                                   uniform32(slotBase + (unsigned)victim * kRecBytes));                     \
                        nJobs = J + 1;                                                                      \
                    }
                    CPF_JOB(0) CPF_JOB(1) CPF_JOB(2) CPF_JOB(3)
                    if (LOOKUP_FIXED) { CPF_JOB(4) CPF_JOB(5) CPF_JOB(6) CPF_JOB(7) }
#undef CPF_JOB
                }
                int younger = 0;
                if (hookDue) younger = hook(Epre);                            // (the tile's first round: Epre is the PREVIOUS tile's result)
                if (BROWNIAN && cycleStart && busy) {
                    // Philox + Box-Muller for the cycle, HERE: behind the hook (the parked result it stores has left the
                    // E slot) and behind the record requests (the deviates overlap their round trip), with almost nothing
                    // of the walk live.  They wait in the lane's E slot -- not needed before the advect, which consumes them.
                    uint32_t kStep0 = step0, kSeed = seed;
                    if (CPF_STREAM_BROWN_KERNARG) kernarg_u32_pair<kKernArgStep0, kKernArgSeed>(kStep0, kSeed);
                    const D3 xi = normal3(id, kStep0 + (uint32_t)c, kSeed);
                    sE[0][lane] = xi.x; sE[1][lane] = xi.y; sE[2][lane] = xi.z;
                }
#ifdef CPF_STREAM_TIMELINE
                const uint64_t tlw0 = __builtin_amdgcn_s_memrealtime();
#endif
                if (nJobs != 0) {
                    // the requested records are older than everything the hook issued: wait for exactly them
                    if (younger >= 7) wait_vmcnt<7>();
                    else if (younger == 6) wait_vmcnt<6>();
                    else if (younger >= 4) wait_vmcnt<4>();
                    else if (younger >= 2) wait_vmcnt<2>();
                    else wait_vmcnt<0>();
#ifdef CPF_STREAM_TIMELINE
                    tlWaitRec += (unsigned)(__builtin_amdgcn_s_memrealtime() - tlw0); ++tlMissRounds;
#endif
                }
                // ---- every busy lane does one cell visit -- unless it was left without a record slot (more new cells
                // than the round can place) in a round where few lanes were: the second and third round of a tile on a
                // 3-D mesh, 5 ... 10 distinct cells.  Such a lane sits the round out (next = kSitOut) and its cell is
                // placed in the next round; running the per-lane gather walk for a handful of lanes makes the whole
                // wave pay a second trace (measured on the 3-D bench mesh: two of every three rounds did).  When many
                // lanes are without a slot (a cloud that is not kept sorted) the gather walk keeps the wave moving.
                const bool gatherRound = __popcll(missLanes) >= kStreamGatherLanes;
                // advect (particles.cu:355-362) with the record's velocity: end point E, parked in the lane's E slot
                auto advect = [&](const double4* rec) __attribute__((always_inline)) -> D3 {
                    D3 Pn;
                    if (VERTEX) {
                        // (a decomposition admitted to the cone locate always yields a tet: cpf_set_tets; multiply, round, add --
                        // what the staged advect and step_kernel_vertex do, cpf_kernels.hip particle_cycles)
                        // (v: interpolated at the cycle's start, cycle_begin)
                        Pn = {S_.x + dt * v.x, S_.y + dt * v.y, S_.z + dt * v.z};
                    } else {
                        double4 u;
                        if (BOX) { const double2 uxy = reinterpret_cast<const double2*>(rec)[5]; u = {uxy.x, uxy.y, reinterpret_cast<const double*>(rec)[12], 0.0}; }
                        else u = rec[6];
                        v = {u.x, u.y, u.z};
                        Pn = axpy(dt, v, S_);
                    }
                    D3 disp = {Pn.x - S_.x, Pn.y - S_.y, Pn.z - S_.z};           // not yet walked in this cycle: S_ is the position
                    if (BROWNIAN) {                                               // the deviates drawn in the cycle's first round
                        const D3 xi = {sE[0][lane], sE[1][lane], sE[2][lane]};
                        disp = axpy(CPF_STREAM_BROWN_KERNARG ? kernarg_f64<kKernArgSigma>() : sigma, xi, disp);
                    }
                    D3 E = {S_.x + disp.x, S_.y + disp.y, S_.z + disp.z};
                    if (BROWNIAN && REFLECT && zFold) {                           // one cell thick in z: cpf_walk.h, fold_z
                        bool clear;
                        const int nb = fold_z(E.z, rec[BOX ? 0 : 4], rec[BOX ? 0 : 5], clear);      // (never with box records: zFold)
                        zUnclear |= ballot64(!clear) != 0ull;                      // (wave-uniform; the advecting lanes vote)
                        if (STATS) st.refl += nb;
                        if (STORE_VEL && (nb & 1)) v.z = -v.z;
                    }
                    sE[0][lane] = E.x; sE[1][lane] = E.y; sE[2][lane] = E.z;
                    return E;
                };
                if (busy) {
                    int next, outSlot = 0;
                    double4 wallPlane;                             // (assigned on every path that reads it: wherever a boundary face is met)
                    const bool needAdvect = token == INT32_MIN && !(bigCells && key2 != 0);      // (second half of a first visit: advected in the first)
                    D3 E = S_;
                    if (!needAdvect) E = Epre;
                    // (mixed meshes with big cells) the slot may hold the HEADER record of a cell with more than twelve slots: that
                    // lane walks the cell's CSR slots per lane from global memory, in the gather branch below
                    bool hdrCell = false;
                    if (bigCells && myslot >= 0) hdrCell = *reinterpret_cast<const int*>(&slots[0][0] + myslot * kSlotStride + 7) == kBigCellMark;
                    if (myslot >= 0 && !(bigCells && hdrCell)) {
                        const double4* rec = &slots[0][0] + myslot * kStride;
                        if (needAdvect) E = advect(rec);
                        // ---- the visit, and -- in the same round -- the visits after a wall.  A wall hit never changes the
                        // cell and its record is in the slot already, so a lane that hits a wall mirrors its end point and
                        // walks on at once (the j < 5 loop of ConvexQuery.cu:353-409, which re-walks in the same `cur`)
                        // instead of sending the whole wave through another round -- lookup, ballots, hook logic -- for
                        // it.  ONE instance of the face tests inside a per-lane loop: from its second trip on only the
                        // reflecting lanes are active and most faces drop out wave-uniformly.  Same arithmetic in the same
                        // order as a round per reflection: bit-identical.
                        bool twoRec = false;
                        int recB = 0;
                        const bool second0 = key2 != 0;                         // which record of its cell this round tests
                        if (bigCells) {
                            const int2 cont = reinterpret_cast<const int2*>(rec + 7)[3];
                            twoRec = !second0 && cont.x == kTwoRecMark;      // (a second record never carries the mark)
                            recB = cont.y;
                        }
                        bool again;
                        do {
                            again = false;
                            if (bigCells) {
                                // ONE record of the cell per round, from LDS: an ordinary cell's only one, or one of the two of a
                                // cell with 7 ... 12 slots (cpf_walk.h)
                                const D3 P0 = S_;
                                const D3 Pd = {E.x - P0.x, E.y - P0.y, E.z - P0.z};
                                double dTm = second0 ? carDT : 2.0;
                                int nx = second0 ? carNext : cur, bs = second0 ? carBest : -1;
                                trace_lds6_record<!BROWNIAN, mixed>(P0, Pd, rec, token, second0 ? 6 : 0, dTm, nx, bs);
                                if (twoRec) {                                  // first of two: nothing happens to the particle yet
                                    carDT = dTm; carNext = nx; carBest = bs;
                                    key2 = recB;
                                    next = kSitOut;
                                } else {                                       // the cell's last record: the visit's outcome
                                    key2 = 0;
                                    if (bs >= 0) { S_ = axpy(dTm, Pd, P0); outSlot = bs; }
                                    next = nx;
                                    if (STATS) ++st.hops;
                                }
                            } else {
                            // few particles per cell = a 3-D mesh: every face is live, two faces per decision (cpf_walk.h;
                            // measured 1-2 % there, nothing with the Brownian kick, and a loss where faces drop out for
                            // zero denominators)
                            if (BOX) next = trace_box<!BROWNIAN, mixed>(S_, E, cur, rec, token, outSlot);
                            else if (FLAT) next = trace_lds4_flat<(CPF_STREAM_FLAT_ZERO_SKIP != 0)>(S_, E, cur, rec, token, outSlot);
                            else if (CPF_STREAM_FLAT_KICK && BROWNIAN && !mixed && flatKick && !zUnclear)
                            next = trace_lds4_flat_z<false>(S_, E, cur, rec, token, outSlot);     // (cpf_walk.h: flat walk under the kick)
                            else
                            next = (CPF_STREAM_PAIRED && LOOKUP_FIXED && !BROWNIAN && !mixed) ? trace_lds6_paired(S_, E, cur, rec, token, outSlot, zLast)
                                                                : trace_lds6<(!BROWNIAN && (CPF_STREAM_L1_ZERO_SKIP || LOOKUP != 1)), mixed>(S_, E, cur, rec, token, outSlot, zLast, zFold && !zUnclear);
                            if (STATS) ++st.hops;
                            }
                            if (REFLECT && next < 0 && !(mixed && is_group(next))) {       // (is_group: face-group codes and kSitOut)
                                // The wall's plane is read HERE, where the record's address space is known (one expression
                                // choosing between the LDS slot and the global record becomes a flat load: vmcnt + lgkmcnt 0).
                                if (bigCells && lk != cur) {
                                    // concluded on a second record: slots 6..11 are in this slot, slots 0..5 only in global memory now
                                    if (outSlot >= 6) wallPlane = rec[outSlot - 6];
                                    else {
                                        wallPlane = m.planes[m.cellOff[cur] + outSlot];
                                        asm volatile("" : "+v"(wallPlane.x), "+v"(wallPlane.y), "+v"(wallPlane.z), "+v"(wallPlane.w));
                                    }
                                } else if (BOX) wallPlane = box_wall_plane(rec, outSlot);
                                else wallPlane = rec[outSlot];
                                // E comes from its parking slot again -- it is there, from this round's advect or an earlier
                                // round -- so that it need not stay in registers across the face tests
                                asm volatile("" ::: "memory");
                                E = {sE[0][lane], sE[1][lane], sE[2][lane]};
                                if (kInRound) {
                                    // reflect INSIDE the round (CPF_STREAM_INROUND; measured slower, docs/design_r04.md 5.5): mirror here and
                                    // walk on at once instead of in the next round
                                    park_hit(S_);
                                    if (STATS) ++st.refl;
                                    const D3 nn = {wallPlane.x, wallPlane.y, wallPlane.z};
                                    const double sd = dot3(wallPlane, E) - wallPlane.w;
                                    E = axpy(-2.0 * sd, nn, E);
                                    sE[0][lane] = E.x; sE[1][lane] = E.y; sE[2][lane] = E.z;
                                    v = axpy(-2.0 * dot3(wallPlane, v), nn, v);
                                    token = next;
                                    h = 0;
                                    if (++j == kMaxReflect) { busy = false; next = cur; }   // still on a wall after 5 bounces: lost
                                    else again = true;
                                }
                            }
                        } while (again);
                    } else {
                        // no slot: more distinct new cells in the wave than the round can place (a cloud that is not
                        // kept sorted).  Per-lane gathers keep such a wave moving.
                        next = kSitOut;
                        if (gatherRound || (bigCells && hdrCell)) {
                            const double4* rec = BOX ? m.boxRec + 4 * (int64_t)cur : m.cellRec + 8 * (int64_t)cur;
                            if (needAdvect) E = advect(rec);
                            int gS0 = 0;
                            bool gBig = false;
                            if (LOOKUP_FIXED && bigCells) {
                                // header cells and two-record cells alike walk their CSR slots here (a lane between the two halves
                                // of a visit starts the visit over: same tests, same result)
                                const int4 hdr = *reinterpret_cast<const int4*>(rec + 7);
                                const int2 cont = reinterpret_cast<const int2*>(rec + 7)[3];
                                gBig = hdr.x == kBigCellMark || cont.x == kTwoRecMark;
                                if (gBig) {
                                    gS0 = m.cellOff[cur];
                                    next = trace_csr(S_, E, cur, m.planes, m.nbr, gS0, m.cellOff[cur + 1] - gS0, token, outSlot);
                                    rec = m.planes + gS0;
                                    key2 = 0;
                                }
                            }
                            if (BOX) next = trace_box<false, mixed>(S_, E, cur, rec, token, outSlot);
                            else if (!gBig)
                            next = trace_fixed<6, false, mixed, kGatherAhead>(S_, E, cur, rec, reinterpret_cast<const int32_t*>(rec + 7), token, outSlot, 0);
                            if (STATS) ++st.hops;
                            // (the empty asm makes the compiler wait for this load HERE: a load of its own left pending
                            // at the loop's back edge costs every round an s_waitcnt vmcnt(0), i.e. a wait for the prefetch)
                            if (REFLECT && next < 0) {
                                wallPlane = BOX ? box_wall_plane(rec, outSlot) : rec[outSlot];
                                asm volatile("" : "+v"(wallPlane.x), "+v"(wallPlane.y), "+v"(wallPlane.z), "+v"(wallPlane.w));
                            }
                        }
                    }
                    // (mixed meshes) left through a face group -- the coplanar pieces of a split face: the piece is chosen
                    // at the exit point S_ (cpf_walk.h; rare: per-lane reads of the CSR tables)
                    if (mixed && next != kSitOut && is_group(next)) next = resolve_group(next, S_, m.cellOff, m.planes, m.groupOff, m.groupNbr);
                    // ---- what the visit led to.  The common outcomes -- the segment ends here, or it crosses into a neighbour --
                    // are applied with selects, not branches: nested branches made the compiler shuttle cell, token and the two
                    // counters between registers at every join (a dozen v_mov per round).  Only the wall is a branch (rare).
#ifdef CPF_STREAM_TIMELINE
                    if (next == kSitOut) tlSat = true; else ++tlVis;
#endif
                    const bool ends = next == cur;                                 // segment ends in this cell
                    const bool wall = next < 0 && next != kSitOut;                 // a boundary face (kSitOut: no visit this round)
                    const bool cross = next >= 0 && !ends;
                    if (wall) {
                        if (!REFLECT) j = kMaxReflect;                             // lost
                        else {
                            // mirror end point and velocity about the wall (ConvexQuery.cu:286-309); walks on next round
                            park_hit(S_);
                            if (STATS) ++st.refl;
                            const D3 nn = {wallPlane.x, wallPlane.y, wallPlane.z};
                            const double sd = dot3(wallPlane, E) - wallPlane.w;
                            E = axpy(-2.0 * sd, nn, E);
                            sE[0][lane] = E.x; sE[1][lane] = E.y; sE[2][lane] = E.z;
                            v = axpy(-2.0 * dot3(wallPlane, v), nn, v);
                            h = 0;
                            ++j;                                                   // j == 5: still on a wall after 5 bounces, lost
                        }
                    }
                    token = cross ? cur : (wall ? next : token);                   // (a wall's code as token: the advect is done)
                    cur = cross ? next : cur;
                    h += cross ? 1 : 0;
                    busy = !(ends || (cross && h == kMaxHops) || (wall && j >= kMaxReflect));   // hop cap: keep the last cell
                }
#ifdef CPF_STREAM_TIMELINE
                {
                    const unsigned nb = (unsigned)__popcll(busyMask), ns = (unsigned)__popcll(ballot64(tlSat));
                    const unsigned r = tlRoundIdx < 15u ? tlRoundIdx : 15u;
                    if (tlH != nullptr && lane == 0) {
                        atomicAdd(&tlH[r * 65u + nb], 1ull); atomicAdd(&tlH[16 * 65 + r], (unsigned long long)ns);
                        atomicAdd(&tlH[16 * 65 + 16 + 256 + r], (unsigned long long)nJobs);
                    }
                    ++tlRoundIdx;
                }
#endif
            };

            auto cycle_end = [&]() __attribute__((always_inline)) {
                // ---- move (particles.cu:693-701); reflected: p = P_hit, disp = P_end - P_hit; else P + disp == E
                if (cur >= 0) {                              // every lane that began the cycle with a particle
                    const D3 E = {sE[0][lane], sE[1][lane], sE[2][lane]};
                    if (REFLECT && j != 0) {                 // reflected at least once
                        D3 hit;
                        if (HIT_IN_REGS) hit = hitReg;
                        else if (hitAt < kPool) hit = {sPool[0][hitAt], sPool[1][hitAt], sPool[2][hitAt]};
                        else load3_sync(static_cast<const double*>(kernarg_pointer<kKernArgHitSpill>()) + (size_t)blockIdx.x * kStreamHitSpillDoubles,
                                        ul * 8u, hit.x, hit.y, hit.z);
                        S_ = {hit.x + (E.x - hit.x), hit.y + (E.y - hit.y), hit.z + (E.z - hit.z)};
                    } else S_ = E;
                    if (j >= kMaxReflect) { cur = CPF_CELL_LOST; if (STATS) ++st.lost; }
                }
            };

            if (nCyc > 0) {
                // ONE instance of the round in the kernel's code: the tile's first round carries the hook behind a
                // wave-uniform flag (three inlined copies -- hook round, other rounds of the first cycle, rounds of the later
                // cycles of a fused launch -- were 3 x 1400 instructions and a dozen register shuffles at every loop edge).
                // A cycle's first round always runs, even with no busy lane (cycle 0's carries the hook).
                for (int c = 0; c < nCyc; ++c) {
                    cycle_begin(c);
#ifdef CPF_STREAM_TIMELINE
                    tlVis = 0; tlRoundIdx = 0;
#endif
                    bool hookDue = c == 0, cycleStart = true;
#ifdef CPF_STREAM_CAP_TEST
                    // upper-bound experiment (results are WRONG): a cycle stops after CPF_STREAM_CAP_TEST rounds; lanes still busy
                    // then do not move in this step (position and cell re-read from global memory)
                    int capRounds = 0, capLimit = CPF_STREAM_CAP_TEST;
                    asm volatile("" : "+s"(capLimit));                      // (opaque: a constant bound makes hipcc unroll the round loop)
                    do { round(hookDue, cycleStart, c); hookDue = false; cycleStart = false; ++capRounds; } while (ballot64(busy) != 0ull && capRounds < capLimit);
                    const bool capDropped = busy;
                    const D3 capKeep = S_;
                    const int capCur = cur;
#else
                    do { round(hookDue, cycleStart, c); hookDue = false; cycleStart = false; } while (ballot64(busy) != 0ull);
#endif
#ifdef CPF_STREAM_TIMELINE
                    {
                        unsigned mx = tlVis;
                        for (int off = 32; off > 0; off >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)mx, off, 64); mx = o > mx ? o : mx; }
                        const unsigned r = tlRoundIdx < 15u ? tlRoundIdx : 15u, iv = mx < 15u ? mx : 15u;
                        if (tlH != nullptr && lane == 0) atomicAdd(&tlH[16 * 65 + 16 + r * 16u + iv], 1ull);
                    }
#endif
                    cycle_end();
#ifdef CPF_STREAM_CAP_TEST
                    if (capDropped) { S_ = capKeep; cur = capCur; }       // stops where its last crossing left it (a consistent state)
#endif
                }
            } else {
                const D3 prev = {sE[0][lane], sE[1][lane], sE[2][lane]};
                (void)hook(prev);                            // zero cycles: loads + stores only (bandwidth calibration)
            }

            // everything this tile's hook issued has landed (the next tile in LDS) or been accepted (the stores)
#ifdef CPF_STREAM_TIMELINE
            const uint64_t tle0 = __builtin_amdgcn_s_memrealtime();
#endif
            wait_vmcnt<0>();
#ifdef CPF_STREAM_TIMELINE
            tlWaitEnd += (unsigned)(__builtin_amdgcn_s_memrealtime() - tle0);
            ++tlTiles;
#endif
            // ---- this tile's results wait in registers until the next tile's hook.  Every lane of the tile stores
            // x, y, z and the cell: a frozen or lost particle gets the bytes it was loaded with (and CPF_CELL_FROZEN),
            // which keeps the NUMBER of stores per tile fixed -- the counted wait above needs it
            rtile = tile; havePrev = true;
            sE[0][lane] = S_.x; sE[1][lane] = S_.y; sE[2][lane] = S_.z;
            if (STORE_VEL) { rvx = v.x; rvy = v.y; rvz = v.z; ralive = pc >= 0; }
            rc = cur;
            rlim = plim;
            if (ntilesLeft <= 0) break;
            tile = ntile; tilesLeft = ntilesLeft;
        }
        // ---- the last tile's results
        {
            const int64_t b = (int64_t)rtile * 64;
            const CloudPtrs cp = kernarg_cloud_ptrs();
            double* const x = cp.x; double* const y = cp.y; double* const z = cp.z; int32_t* const cell = cp.cell;
            const double rx = sE[0][lane], ry = sE[1][lane], rz = sE[2][lane];
            if (ul <= rlim) { (x + b)[ul] = rx; (y + b)[ul] = ry; (z + b)[ul] = rz; (cell + b)[ul] = rc; }
            if (STORE_VEL && ralive) {
                double* vv = vel + 3 * b;
                vv[3 * ul] = rvx; vv[3 * ul + 1] = rvy; vv[3 * ul + 2] = rvz;
            }
        }
    }
#ifdef CPF_STREAM_TIMELINE
    if (!STORE_VEL && vel != nullptr && lane == 0) {
        uint64_t* o = reinterpret_cast<uint64_t*>(vel) + 8 * (uint64_t)blockIdx.x;
        o[0] = tl0; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = tlTiles; o[3] = tlRounds;
        o[4] = tlWaitRec; o[5] = tlWaitEnd; o[6] = tlMissRounds; o[7] = 0;
    }
#endif
    if (STATS) flush_stats(st, counters, sCnt);
}

template <bool BROWNIAN, bool REFLECT, bool STORE_VEL, bool STATS, int LOOKUP>
__global__ __launch_bounds__(64, (StreamOccupancy<BROWNIAN, STORE_VEL, STATS, LOOKUP>::waves)) void step_kernel_stream(
    double* __restrict__ /* x */, double* __restrict__ /* y */, double* __restrict__ /* z */, int32_t* __restrict__ /* cell */,   // read through kernarg_cloud_ptrs()
    const int64_t* __restrict__ gid, double* __restrict__ vel, int64_t n, double dt, double sigma, uint32_t step0,
    int nCyc, uint32_t seed, MeshView m, unsigned long long* __restrict__ counters, StreamArgs sa) {
    stream_body<BROWNIAN, REFLECT, STORE_VEL, STATS, LOOKUP, false>(gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters, sa, VertexField{});
}

// the "VertexVelocity" cycle (CPF_STEP_VERTEX_VELOCITY) on the streaming kernel: all-hex meshes, loop / fixed lookup; the tet
// tables add ~40 vector registers to the advect: four waves per SIMD
#ifndef CPF_STREAM_WAVES_VERTEX
#define CPF_STREAM_WAVES_VERTEX 4
#endif
template <bool BROWNIAN, bool REFLECT, bool STORE_VEL, bool STATS, int LOOKUP>
__global__ __launch_bounds__(64, ((STORE_VEL || STATS) ? 1 : CPF_STREAM_WAVES_VERTEX)) void step_kernel_stream_vertex(
    double* __restrict__ /* x */, double* __restrict__ /* y */, double* __restrict__ /* z */, int32_t* __restrict__ /* cell */,
    const int64_t* __restrict__ gid, double* __restrict__ vel, int64_t n, double dt, double sigma, uint32_t step0,
    int nCyc, uint32_t seed, MeshView m, unsigned long long* __restrict__ counters, StreamArgs sa, VertexField vf) {
    stream_body<BROWNIAN, REFLECT, STORE_VEL, STATS, LOOKUP, true>(gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters, sa, vf);
}

// ------------------------------------------------------------------------------------------------
// launcher: persistent grid sized by the occupancy of the instantiation
// ------------------------------------------------------------------------------------------------
template <bool B, bool R_, bool SV, bool ST, int LF, bool VX = false>
static hipError_t launch_stream_inst(hipStream_t st, double* x, double* y, double* z, int32_t* cell, const int64_t* gid,
                                     double* vel, int64_t n, double dt, double sigma, uint32_t step0, int nCyc,
                                     uint32_t seed, const MeshView& m, unsigned long long* counters, StreamState& ss,
                                     const VertexField* vf = nullptr) {
    static int wavesPerCU = 0;                       // per instantiation; benign race (same value)
    if (wavesPerCU == 0) {
        int nb = 0;
        hipError_t e;
        if constexpr (VX) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, step_kernel_stream_vertex<B, R_, SV, ST, LF>, 64, 0);
        else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, step_kernel_stream<B, R_, SV, ST, LF>, 64, 0);
        if (e != hipSuccess) return e;
        wavesPerCU = nb < 1 ? 1 : (nb > 32 ? 32 : nb);
    }
    const int64_t nTiles = (n + 63) >> 6;
    if (nTiles >= ((int64_t)1 << 31)) return hipErrorInvalidValue;      // the kernel numbers tiles and chunks in 32 bits
    const int64_t slotsOnChip = (int64_t)((ss.wavesPerCU > 0 && !VX) ? ss.wavesPerCU : wavesPerCU) * ss.numCU;
    // small clouds: shorter chunks, so that every wave slot still gets several
    // dealing knobs, by lookup method unless set: few particles per cell = slower tiles (more rounds, record misses), so
    // finer chunks and a longer tile-by-tile tail (tools/sweep3d_opts.sh: 3 / 0.2 against 4 / 0.1 on the 3-D box 0.2632 /
    // 0.2701 -> 0.2585 / 0.2661 ms, TJunction 0.1581 / 0.1708 -> 0.1569 / 0.1687, 5 particles per cell 0.682 / 0.670 -> 0.677 /
    // 0.636; pitzDaily is at its optimum with 4 / 0.1, tools/tail_sweep.sh)
    int tpc = ss.tilesPerChunk > 0 ? ss.tilesPerChunk : ((LF == 0 || LF == 5 || LF == 8) ? 4 : 3);
    const double tailFraction = ss.tailFraction >= 0.0 ? ss.tailFraction : ((LF == 0 || LF == 5 || LF == 8) ? 0.1 : 0.2);
    while (tpc > 1 && nTiles / tpc < 4 * slotsOnChip) tpc = tpc > 2 ? tpc - 1 : 1;
    // the first (1 - tailFraction) of the cloud in chunks of tpc tiles, the rest tile by tile
    int64_t bigChunks = (int64_t)((double)(nTiles / tpc) * (1.0 - tailFraction));
    if (tpc == 1 || bigChunks < 0) bigChunks = 0;
    const int64_t nChunks = bigChunks + (nTiles - bigChunks * tpc);
    int64_t R = slotsOnChip / kStreamGroups;                 // waves per group
    const int64_t need = (nChunks + kStreamGroups - 1) / kStreamGroups;
    if (R > need) R = need;
    if (R < 1) R = 1;
    unsigned* cur = ss.d_grab + (size_t)(ss.parity & 1) * kStreamGroups * kStreamCounterStride;
    unsigned* nxt = ss.d_grab + (size_t)((ss.parity & 1) ^ 1) * kStreamGroups * kStreamCounterStride;
    if (R * kStreamGroups > ss.hitSpillWaves) R = ss.hitSpillWaves / kStreamGroups;      // (never: the area is sized for the chip)
    if (R < 1 || ss.d_hitSpill == nullptr) return hipErrorInvalidValue;
    StreamArgs sa = {cur, nxt, (int)R, tpc, (unsigned)bigChunks, ss.debug, ss.d_hitSpill};
    if constexpr (VX) {
        if (vf == nullptr) return hipErrorInvalidValue;
        if (ss.evStart != nullptr && ss.evStop != nullptr) {
            hipExtLaunchKernelGGL((step_kernel_stream_vertex<B, R_, SV, ST, LF>), dim3((unsigned)(R * kStreamGroups)), dim3(64), 0, st, ss.evStart,
                                  ss.evStop, 0, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters, sa, *vf);
            ss.evStart = ss.evStop = nullptr;
        } else
        hipLaunchKernelGGL((step_kernel_stream_vertex<B, R_, SV, ST, LF>), dim3((unsigned)(R * kStreamGroups)), dim3(64), 0, st, x, y, z, cell,
                           gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters, sa, *vf);
        return stream_launch_done(st, ss);
    } else {
    if (ss.evStart != nullptr && ss.evStop != nullptr) {
        hipExtLaunchKernelGGL((step_kernel_stream<B, R_, SV, ST, LF>), dim3((unsigned)(R * kStreamGroups)), dim3(64), 0, st, ss.evStart,
                              ss.evStop, 0, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters, sa);
        ss.evStart = ss.evStop = nullptr;                    // taken
    } else
    hipLaunchKernelGGL((step_kernel_stream<B, R_, SV, ST, LF>), dim3((unsigned)(R * kStreamGroups)), dim3(64), 0, st, x, y, z, cell,
                       gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters, sa);
    return stream_launch_done(st, ss);
    }
}

// few particles per cell => many distinct cells per 64-particle tile => the fixed tag compare
int stream_lookup_mode(int64_t n, const MeshView& m, const StreamState& ss, bool brown) {
    // particles per cell that HOLDS particles, if the last sort counted them for a cloud of about this size (StreamState)
    int64_t cells = m.nCells;
    if (ss.densityLookup && ss.occupiedHost != nullptr) {
        const int64_t occ = (int64_t)ss.occupiedHost[0], live = (int64_t)ss.occupiedHost[1];
        if (occ > 0 && occ <= cells && live > 0 && live <= 2 * n && n <= 2 * live) cells = occ;
    }
    // not all-hex: with / without big cells (more than six slots); without them and with many particles per cell, the loop lookup
    // (every cell a box although the mesh has face groups -- 2:1-refined boxes: box records with group slots, 11)
    if (m.mixed == 1 && m.boxRec != nullptr && m.zThin == 0 && ss.lookup < 0) return 11;
    if (m.mixed) return m.mixed == 2 ? 2 : ((ss.lookup >= 0 ? ss.lookup == 0 : n >= 128 * cells) ? 5 : 3);
    const bool box = m.boxRec != nullptr && m.zThin == 0;
    if (ss.lookup >= 0) return (ss.lookup == 6 && !box) ? 1 : ss.lookup;      // "stream_lookup": 0, 1, 4 or 6 (2, 3, 5: diagnostics)
    // box records whatever the density: also above 128 particles per cell they beat the loop lookup on 256-byte records (round 4,
    // 3-D boxes of 2 048 / 20 480 / 61 440 cells with 1e7 particles: 0.1216 / 0.1648 / 0.2036 -> 0.1203 / 0.1617 / 0.1892 ms, with the
    // kick 0.1584 / 0.2178 / 0.2729 -> 0.1568 / 0.2052 / 0.2375)
    if (box && CPF_STREAM_BOX_SPARSE) return 6;
    if (n < kStreamSparsePerCell * cells) return 4;
    // a 2-D mesh, a field without a z component, no kick: the flat walk (8 = 0, 9 = 1 with it)
    const bool flat = !brown && ss.flat && ss.flatField && m.zSide0 != 0;
    if (n < 128 * cells) return flat ? 9 : 1;
    return flat ? 8 : 0;
}

// the "VertexVelocity" cycle streams on all-hex meshes with the loop or the fixed lookup on the 256-byte records (no flat walk:
// the interpolated velocity may have a z component whatever the cell field says; mixed meshes keep step_kernel_vertex)
bool stream_vertex_capable(const MeshView& m) { return m.mixed == 0 && m.cellRec != nullptr; }
int stream_vertex_lookup_mode(int64_t n, const MeshView& m, const StreamState& ss) {
    int64_t cells = m.nCells;
    if (ss.densityLookup && ss.occupiedHost != nullptr) {
        const int64_t occ = (int64_t)ss.occupiedHost[0], live = (int64_t)ss.occupiedHost[1];
        if (occ > 0 && occ <= cells && live > 0 && live <= 2 * n && n <= 2 * live) cells = occ;
    }
    return n < 128 * cells ? 1 : 0;
}

hipError_t launch_step_stream_vertex(hipStream_t st, double* x, double* y, double* z, int32_t* cell, const int64_t* gid,
                                     double* vel, int64_t n, double dt, double sigma, uint32_t step0, int nCyc, uint32_t seed,
                                     bool brown, bool reflect, bool storeVel, const MeshView& m, unsigned long long* counters,
                                     StreamState& ss, const VertexField& vf) {
    const int lf = stream_vertex_lookup_mode(n, m, ss);
#define CPF_STREAM_VGO(B, R, SV, ST)                                                                                         \
    do {                                                                                                                     \
        if (lf == 1) return launch_stream_inst<B, R, SV, ST, 1, true>(st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters, ss, &vf);  \
        return launch_stream_inst<B, R, SV, ST, 0, true>(st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters, ss, &vf);               \
    } while (0)
#define CPF_STREAM_VSV(B, R)                                                                     \
    do {                                                                                        \
        if (storeVel) { if (counters) CPF_STREAM_VGO(B, R, true, true); else CPF_STREAM_VGO(B, R, true, false); } \
        else { if (counters) CPF_STREAM_VGO(B, R, false, true); else CPF_STREAM_VGO(B, R, false, false); }        \
    } while (0)
    if (brown) { if (reflect) CPF_STREAM_VSV(true, true); else CPF_STREAM_VSV(true, false); }
    else { if (reflect) CPF_STREAM_VSV(false, true); else CPF_STREAM_VSV(false, false); }
#undef CPF_STREAM_VSV
#undef CPF_STREAM_VGO
}

hipError_t launch_step_stream(hipStream_t st, double* x, double* y, double* z, int32_t* cell, const int64_t* gid,
                              double* vel, int64_t n, double dt, double sigma, uint32_t step0, int nCyc, uint32_t seed,
                              bool brown, bool reflect, bool storeVel, const MeshView& m, unsigned long long* counters,
                              StreamState& ss) {
    const int lf = stream_lookup_mode(n, m, ss, brown);
#define CPF_STREAM_GO(B, R, SV, ST)                                                                                          \
    do {                                                                                                                     \
        if (lf == 8) { if (!B) return launch_stream_inst<false, R, SV, ST, 8>(st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters, ss); return hipErrorInvalidValue; } \
        if (lf == 11) return launch_stream_inst<B, R, SV, ST, 11>(st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters, ss);  \
        if (lf == 9) { if (!B) return launch_stream_inst<false, R, SV, ST, 9>(st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters, ss); return hipErrorInvalidValue; } \
        if (lf == 6) return launch_stream_inst<B, R, SV, ST, 6>(st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters, ss);  \
        if (lf == 5) return launch_stream_inst<B, R, SV, ST, 5>(st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters, ss);  \
        if (lf == 4) return launch_stream_inst<B, R, SV, ST, 4>(st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters, ss);  \
        if (lf == 3) return launch_stream_inst<B, R, SV, ST, 3>(st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters, ss);  \
        if (lf == 2) return launch_stream_inst<B, R, SV, ST, 2>(st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters, ss);  \
        if (lf == 1) return launch_stream_inst<B, R, SV, ST, 1>(st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters, ss);  \
        return launch_stream_inst<B, R, SV, ST, 0>(st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters, ss);         \
    } while (0)
#define CPF_STREAM_SV(B, R)                                                                     \
    do {                                                                                        \
        if (storeVel) { if (counters) CPF_STREAM_GO(B, R, true, true); else CPF_STREAM_GO(B, R, true, false); } \
        else { if (counters) CPF_STREAM_GO(B, R, false, true); else CPF_STREAM_GO(B, R, false, false); }        \
    } while (0)
    if (brown) { if (reflect) CPF_STREAM_SV(true, true); else CPF_STREAM_SV(true, false); }
    else { if (reflect) CPF_STREAM_SV(false, true); else CPF_STREAM_SV(false, false); }
#undef CPF_STREAM_SV
#undef CPF_STREAM_GO
}

}  // namespace cpf
