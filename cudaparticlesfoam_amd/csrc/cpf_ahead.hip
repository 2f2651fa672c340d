// Streaming step kernel with lanes that RUN AHEAD (variant 5): step_kernel_stream (cpf_stream.hip) whose lanes do
// not wait for the slowest particle of their tile.
//
// In step_kernel_stream a tile costs as many rounds as its slowest particle has cell visits: 2.23 rounds for 1.77 visits
// per particle on pitzDaily, 3.6 rounds for 2.0 visits on the 245 760-cell 3-D mesh (lanes leave a 3-D cell through
// different faces).  Here a lane that has finished its particle of the current tile takes the particle in ITS slot of
// the next tile (already in LDS: two landing zones used alternately, loads TWO tiles ahead -- a zone is free again as
// soon as every lane has taken its particle out of it, which is at the boundary INTO that tile) and walks on; its finished result waits in
// registers for the wave's tile boundary -- the moment no lane is on the current tile any more -- where the tile is
// stored with four coalesced stores as before.  A lane can be at most one tile ahead: one that also finishes the next
// tile's particle parks until the boundary.  Same arithmetic per particle, same record cache, same chunk dealing;
// one cycle per launch, no Brownian kick, no stored velocity (launch_step uses step_kernel_stream for those).
//
// Lane states: A walking the current tile's particle | B done with it (result in r*), no next particle yet |
//              C done, walking the NEXT tile's particle ("ahead") | D done, next tile's particle finished too (parked).
#include "cpf_stream_ops.h"

namespace cpf {

#ifndef CPF_AHEAD_SLOTS
#define CPF_AHEAD_SLOTS 5
#endif
#ifndef CPF_AHEAD_WAVES
#define CPF_AHEAD_WAVES 5
#endif
constexpr int kAheadSlots = CPF_AHEAD_SLOTS;

template <bool REFLECT, bool STATS>
__global__ __launch_bounds__(64, (STATS ? 1 : CPF_AHEAD_WAVES)) void step_kernel_ahead(
    double* __restrict__ x, double* __restrict__ y, double* __restrict__ z, int32_t* __restrict__ cell, int64_t n,
    double dt, MeshView m, unsigned long long* __restrict__ counters, StreamArgs sa, double* __restrict__ dbg) {
    constexpr int NS = kAheadSlots;
    constexpr unsigned ALL = (1u << NS) - 1u;
    __shared__ double4 slots[NS][8];
    __shared__ unsigned sCnt[4];
    __shared__ double sLane[6][64];
    __shared__ double sPre[2][224];                  // two landing zones: x[64] | y[64] | z[64] | cell[64] (int32)
    double(*sE)[64] = sLane;
    double(*sHit)[64] = sLane + 3;
    const int lane = threadIdx.x;
    const unsigned ul = threadIdx.x;
    const unsigned preBase = uniform32(lds_addr(sPre));
    const unsigned slotBase = uniform32(lds_addr(slots));
    const int tpc = sa.tilesPerChunk;
    const int64_t nTiles = (n + 63) >> 6;
    const unsigned long long big = sa.bigChunks;
    auto chunk_first = [&](unsigned long long c) -> int64_t {
        return c < big ? (int64_t)c * tpc : (int64_t)big * tpc + (int64_t)(c - big);
    };
    auto chunk_tiles = [&](unsigned long long c, int64_t firstTile) -> int {
        const int64_t k = nTiles - firstTile;
        const int64_t want = c < big ? (int64_t)tpc : (int64_t)1;
        return (int)(k < 0 ? 0 : (k < want ? k : want));
    };
    StepStats st = {0, 0, 0, 0};
#ifdef CPF_STREAM_TIMELINE
    const uint64_t tl0 = __builtin_amdgcn_s_memrealtime();
    unsigned tlTiles = 0, tlRounds = 0;
#endif
    for (unsigned k = blockIdx.x; k < (unsigned)kStreamGroups; k += gridDim.x)
        if (lane == 0) sa.grabNext[k * kStreamCounterStride] = 0u;
    const unsigned grp = blockIdx.x & (kStreamGroups - 1);
    unsigned* const myGrab = sa.grab + grp * kStreamCounterStride;

    // ---- the wave's tile sequence: chunks from the group's counter, tiles of a chunk in order
    int64_t seqTile = -1;
    int seqLeft = 0;
    auto next_tile = [&]() __attribute__((always_inline)) -> int64_t {
        if (seqLeft > 1) { --seqLeft; return ++seqTile; }
        unsigned got = 0;
        if (lane == 0) got = grab_sync(myGrab);
        const unsigned long long c = (unsigned long long)grp + (unsigned long long)kStreamGroups * uniform32(got);
        const int64_t ft = chunk_first(c);
        seqLeft = chunk_tiles(c, ft);
        seqTile = seqLeft > 0 ? ft : -1;
        if (seqLeft <= 0) seqLeft = 0;
        return seqTile;
    };
    // requests tile t into landing zone `zone` (see step_kernel_stream::prefetch); returns the operations issued
    auto prefetch = [&](int64_t t, unsigned zone) __attribute__((always_inline)) -> int {
        const int64_t b = uniform64(t * 64);
        const int64_t left = n - b;
        const unsigned zb = uniform32(preBase + zone * 1792u);
        if (left >= 64) {
            const char* s1 = (ul < 32u ? reinterpret_cast<const char*>(x + b) : reinterpret_cast<const char*>(y + b)) + (ul & 31u) * 16u;
            glds16(s1, zb);
            if (ul < 48u) {
                const char* s2 = ul < 32u ? reinterpret_cast<const char*>(z + b) + ul * 16u
                                          : reinterpret_cast<const char*>(cell + b) + (ul - 32u) * 16u;
                glds16(s2, zb + 1024u);
            }
            return 2;
        }
        const unsigned lim = (unsigned)(left - 1);
        const unsigned l = ul < lim ? ul : lim;
        const double vx = (x + b)[l], vy = (y + b)[l], vz = (z + b)[l];
        const int vc = (cell + b)[l];
        double* zp = sPre[zone];
        zp[ul] = vx; zp[64 + ul] = vy; zp[128 + ul] = vz;
        reinterpret_cast<int*>(zp + 192)[ul] = vc;
        return 0;
    };
    auto tile_lim = [&](int64_t t) -> unsigned {
        const int64_t left = n - t * 64;
        return (unsigned)(left < 64 ? left : (int64_t)64) - 1u;
    };

    int64_t curTile = next_tile();
    if (curTile >= 0) {
        int64_t nextTile = next_tile();
        int64_t next2Tile = -1;                      // the tile after the next: its loads are in flight during this tile
        unsigned par = 0;                            // landing zone of the current tile; the next tile's is par ^ 1
        (void)prefetch(curTile, 0);
        if (nextTile >= 0) (void)prefetch(nextTile, 1);
        wait_vmcnt<0>();
        unsigned curLim = tile_lim(curTile);
        bool firstIssue = nextTile >= 0;             // zone 0 is free once every lane has taken tile 0: request tile 2

        int tagv = -1;
        unsigned fifo = 0;

        // ---- per-lane state
        D3 P = {0, 0, 0}, S_ = {0, 0, 0};
        int cur = CPF_CELL_FROZEN;
        bool busy = false, needAdvect = false, reflected = false, lostNow = false, hadParticle = false;
        int token = INT32_MIN, h = 0, j = 0;
        bool doneCur = false, ahead = false;          // states A..D, see the head of the file
        double rx = 0, ry = 0, rz = 0;
        int rc = CPF_CELL_FROZEN;

        // the lane picks up the particle in its slot of landing zone `zone` (lanes past the cloud's end: none)
        auto take = [&](unsigned zone, unsigned lim) __attribute__((always_inline)) {
            const double* zp = sPre[zone];
            P = {zp[ul], zp[64 + ul], zp[128 + ul]};
            cur = reinterpret_cast<const int*>(zp + 192)[ul];
            if (ul > lim) cur = CPF_CELL_FROZEN;
            hadParticle = cur >= 0;
            busy = hadParticle; needAdvect = busy; reflected = false; lostNow = false;
            token = INT32_MIN; h = 0; j = 0;
            S_ = P;
            if (STATS && busy) ++st.steps;
        };
        // the live particle is finished: where it ended up (the move of particles.cu:693-701 happened in finish())
        auto result_to_r = [&]() __attribute__((always_inline)) {
            rx = P.x; ry = P.y; rz = P.z;
            rc = hadParticle ? cur : CPF_CELL_FROZEN;
        };
        auto finish = [&]() __attribute__((always_inline)) {        // step_kernel_stream::cycle_end for this lane
            if (hadParticle) {
                const D3 E = {sE[0][lane], sE[1][lane], sE[2][lane]};
                if (reflected) {
                    const D3 hit = {sHit[0][lane], sHit[1][lane], sHit[2][lane]};
                    P = {hit.x + (E.x - hit.x), hit.y + (E.y - hit.y), hit.z + (E.z - hit.z)};
                } else P = E;
                if (lostNow) { cur = CPF_CELL_LOST; if (STATS) ++st.lost; }
            }
        };

        take(0, curLim);
        if (!busy) { result_to_r(); doneCur = true; }             // no particle in this slot: done at once
        bool boundaryPending = false;

        for (;;) {
#ifdef CPF_STREAM_TIMELINE
            ++tlRounds;
#endif
            int64_t storeTile = -1;
            unsigned storeLim = 63;
            if (boundaryPending) {
                // ---- (a) the next tile becomes the current one; lanes without a live particle take theirs
                storeTile = curTile; storeLim = curLim;
                // everything in flight is a tile old: the stores of the last boundary and the loads of the tile after the
                // next, so this wait is short -- and after it the new current tile AND the one after it are in LDS
                wait_vmcnt<0>();
                curTile = nextTile;
                nextTile = next2Tile;
                if (curTile >= 0) {
                    par ^= 1u;
                    curLim = tile_lim(curTile);
                    if (!ahead) {                                     // state B: r* still holds the OLD tile's result
                        take(par, curLim);
                        // (a slot without a particle is handled after the stores, with the parked lanes)
                    }
                }
            }
            // ---- record cache lookup (see step_kernel_stream::round)
            const unsigned long long busyMask = ballot64(busy);
            int myslot = -1;
            unsigned used = 0;
            unsigned long long todo = busyMask, missLanes = 0ull;
            while (todo != 0ull) {
                const int leader = __ffsll((long long)todo) - 1;
                const int ck = __builtin_amdgcn_readlane(cur, leader);
                const bool mine = cur == ck;
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
            int nJobs = 0;
            if (missLanes != 0ull) {
#define CPF_JOB(J)                                                                                          \
                if (missLanes != 0ull && used != ALL) {                                                     \
                    const int leader = __ffsll((long long)missLanes) - 1;                                   \
                    const int ck = __builtin_amdgcn_readlane(cur, leader);                                  \
                    const unsigned cand = ~used & ALL;                                                      \
                    const unsigned hi = cand & (ALL << fifo) & ALL;                                         \
                    const int victim = __ffs((int)(hi ? hi : cand)) - 1;                                    \
                    fifo = (unsigned)(victim + 1) == (unsigned)NS ? 0u : (unsigned)(victim + 1);            \
                    used |= 1u << victim;                                                                   \
                    if (lane == victim) tagv = ck;                                                          \
                    const bool mine = cur == ck;                                                            \
                    if (mine) myslot = victim;                                                              \
                    missLanes &= ~ballot64(mine);                                                           \
                    if (ul < 16u)                                                                           \
                        glds16(reinterpret_cast<const char*>(m.cellRec) + (int64_t)ck * 256 + ul * 16u,      \
                               uniform32(slotBase + (unsigned)victim * 256u));                              \
                    nJobs = J + 1;                                                                          \
                }
                CPF_JOB(0) CPF_JOB(1) CPF_JOB(2) CPF_JOB(3)
#undef CPF_JOB
            }
            int younger = 0;
            if (boundaryPending) {
                // ---- (c) the finished tile's stores, the loads of the tile after the next, the parked lanes
                __builtin_amdgcn_sched_barrier(0);
                boundaryPending = false;
                if (!(sa.debug & 1)) {
                    const int64_t b = uniform64(storeTile * 64);
                    if (ul <= storeLim) {
                        async_store(x + b, ul * 8u, rx); async_store(y + b, ul * 8u, ry); async_store(z + b, ul * 8u, rz);
                        async_store(cell + b, ul * 4u, rc);
                    }
                    younger += 4;
                }
#ifdef CPF_STREAM_TIMELINE
                ++tlTiles;
#endif
                if (curTile < 0) break;                               // that was the wave's last tile
                // the new current tile's zone has just been emptied (a): it takes the tile after the next
                next2Tile = nextTile >= 0 ? next_tile() : -1;
                if (next2Tile >= 0) younger += prefetch(next2Tile, par);
                // lanes whose live particle already belongs to the (new) current tile
                if (ahead) {
                    ahead = false;
                    if (busy) doneCur = false;                        // C -> A
                    else { result_to_r(); doneCur = true; }           // D -> B: r* is free now
                } else {
                    // took the tile's particle in (a): A, or B at once when its slot holds no particle
                    if (busy) doneCur = false; else { result_to_r(); doneCur = true; }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (firstIssue) {                                         // first round of the wave: zone 0 is empty
                firstIssue = false;
                next2Tile = next_tile();
                if (next2Tile >= 0) younger += prefetch(next2Tile, 0u);
            }
            if (nJobs != 0) {
                if (younger >= 6) wait_vmcnt<6>();
                else if (younger >= 4) wait_vmcnt<4>();
                else if (younger >= 2) wait_vmcnt<2>();
                else wait_vmcnt<0>();
            }
            // ---- every busy lane does one cell visit (step_kernel_stream::round)
            if (busy) {
                int next, outSlot = 0;
                double4 wallPlane = {0, 0, 0, 0};
                D3 E = S_;
                if (!needAdvect) E = {sE[0][lane], sE[1][lane], sE[2][lane]};
                if (myslot >= 0) {
                    const double4* rec = &slots[0][0] + myslot * 8;
                    if (needAdvect) {
                        const double4 u = rec[6];
                        const D3 v = {u.x, u.y, u.z};
                        const D3 Pn = axpy(dt, v, P);                              // particles.cu:355-362
                        const D3 disp = {Pn.x - P.x, Pn.y - P.y, Pn.z - P.z};
                        E = {P.x + disp.x, P.y + disp.y, P.z + disp.z};
                        sE[0][lane] = E.x; sE[1][lane] = E.y; sE[2][lane] = E.z;
                        needAdvect = false;
                    }
                    next = trace_lds6(S_, E, cur, rec, token, outSlot);
                    if (REFLECT && next < 0) wallPlane = rec[outSlot];
                } else {
                    const double4* rec = m.cellRec + 8 * (int64_t)cur;
                    if (needAdvect) {
                        const double4 u = rec[6];
                        const D3 v = {u.x, u.y, u.z};
                        const D3 Pn = axpy(dt, v, P);
                        const D3 disp = {Pn.x - P.x, Pn.y - P.y, Pn.z - P.z};
                        E = {P.x + disp.x, P.y + disp.y, P.z + disp.z};
                        sE[0][lane] = E.x; sE[1][lane] = E.y; sE[2][lane] = E.z;
                        needAdvect = false;
                    }
                    next = trace_fixed<6, false>(S_, E, cur, rec, reinterpret_cast<const int32_t*>(rec + 7), token, outSlot, 0);
                    if (REFLECT && next < 0) {
                        wallPlane = rec[outSlot];
                        asm volatile("" : "+v"(wallPlane.x), "+v"(wallPlane.y), "+v"(wallPlane.z), "+v"(wallPlane.w));
                    }
                }
                if (STATS) ++st.hops;
                if (next == cur) {
                    busy = false;
                } else if (next < 0) {
                    if (!REFLECT) { busy = false; lostNow = true; }
                    else {
                        sHit[0][lane] = S_.x; sHit[1][lane] = S_.y; sHit[2][lane] = S_.z;
                        reflected = true; if (STATS) ++st.refl;
                        const D3 nn = {wallPlane.x, wallPlane.y, wallPlane.z};
                        const double sd = dot3(wallPlane, E) - wallPlane.w;
                        E = axpy(-2.0 * sd, nn, E);
                        sE[0][lane] = E.x; sE[1][lane] = E.y; sE[2][lane] = E.z;
                        token = next;
                        h = 0;
                        if (++j == kMaxReflect) { busy = false; lostNow = true; }
                    }
                } else {
                    token = cur;
                    cur = next;
                    if (++h == kMaxHops) busy = false;
                }
                if (!busy) {                                          // the live particle has just finished
                    finish();
                    if (!ahead) { result_to_r(); doneCur = true; }    // A -> B;  (C -> D needs nothing: parked)
                }
            }
            // ---- lanes in state B take their particle of the next tile and run ahead (B -> C, or D at once)
            if (nextTile >= 0) {
                const bool want = doneCur && !ahead;
                const unsigned long long wantMask = ballot64(want);
                if (wantMask != 0ull) {
                    if (want) {
                        take(par ^ 1u, tile_lim(nextTile));
                        ahead = true;
                    }
                }
            }
            // ---- nobody is on the current tile any more: tile boundary at the head of the next round
            if (ballot64(!doneCur) == 0ull) boundaryPending = true;
        }
    }
#ifdef CPF_STREAM_TIMELINE
    if (dbg != nullptr && lane == 0) {
        uint64_t* o = reinterpret_cast<uint64_t*>(dbg) + 4 * (uint64_t)blockIdx.x;
        o[0] = tl0; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = tlTiles; o[3] = tlRounds;
    }
#endif
    if (STATS) flush_stats(st, counters, sCnt);
}

template <bool R_, bool ST>
static hipError_t launch_ahead_inst(hipStream_t st, double* x, double* y, double* z, int32_t* cell, int64_t n, double dt,
                                    const MeshView& m, unsigned long long* counters, StreamState& ss, double* dbg) {
    static int wavesPerCU = 0;
    if (wavesPerCU == 0) {
        int nb = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, step_kernel_ahead<R_, ST>, 64, 0);
        if (e != hipSuccess) return e;
        wavesPerCU = nb < 1 ? 1 : (nb > 32 ? 32 : nb);
    }
    const int64_t nTiles = (n + 63) >> 6;
    const int64_t slotsOnChip = (int64_t)(ss.wavesPerCU > 0 ? ss.wavesPerCU : wavesPerCU) * ss.numCU;
    int tpc = ss.tilesPerChunk > 0 ? ss.tilesPerChunk : 4;
    while (tpc > 1 && nTiles / tpc < 4 * slotsOnChip) tpc >>= 1;
    int64_t bigChunks = (int64_t)((double)(nTiles / tpc) * (1.0 - (ss.tailFraction >= 0.0 ? ss.tailFraction : 0.1)));
    if (tpc == 1 || bigChunks < 0) bigChunks = 0;
    const int64_t nChunks = bigChunks + (nTiles - bigChunks * tpc);
    int64_t R = slotsOnChip / kStreamGroups;
    const int64_t need = (nChunks + kStreamGroups - 1) / kStreamGroups;
    if (R > need) R = need;
    if (R < 1) R = 1;
    unsigned* cur = ss.d_grab + (size_t)(ss.parity & 1) * kStreamGroups * kStreamCounterStride;
    unsigned* nxt = ss.d_grab + (size_t)((ss.parity & 1) ^ 1) * kStreamGroups * kStreamCounterStride;
    StreamArgs sa = {cur, nxt, (int)R, tpc, (unsigned)bigChunks, ss.debug};
    hipLaunchKernelGGL((step_kernel_ahead<R_, ST>), dim3((unsigned)(R * kStreamGroups)), dim3(64), 0, st, x, y, z, cell, n, dt, m,
                       counters, sa, dbg);
    ss.parity ^= 1;
    return hipGetLastError();
}

// dbg: diagnostic builds (-DCPF_STREAM_TIMELINE) write per-wave times there (the caller's unused `vel` array)
hipError_t launch_step_ahead(hipStream_t st, double* x, double* y, double* z, int32_t* cell, int64_t n, double dt, bool reflect,
                             const MeshView& m, unsigned long long* counters, StreamState& ss, double* dbg) {
    if (reflect) return counters ? launch_ahead_inst<true, true>(st, x, y, z, cell, n, dt, m, counters, ss, dbg)
                                 : launch_ahead_inst<true, false>(st, x, y, z, cell, n, dt, m, counters, ss, dbg);
    return counters ? launch_ahead_inst<false, true>(st, x, y, z, cell, n, dt, m, counters, ss, dbg)
                    : launch_ahead_inst<false, false>(st, x, y, z, cell, n, dt, m, counters, ss, dbg);
}

}  // namespace cpf
