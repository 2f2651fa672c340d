// HIP kernels (gfx950 / CDNA4, wave64) for the per-timestep particle loop, and their launchers.
//
// One fused kernel per Lagrangian cycle replaces the reference's five launches + five device
// syncs (src/advect.H:96-161): advect -> Brownian kick -> locate (plane-exit walk) -> wall
// reflect -> move.  The walk runs on the polyMesh cells themselves (CSR face slots with
// precomputed inward planes), not on a 12-tets-per-cell decomposition, and needs no BVH.
//
// Arithmetic contract ("Numerics, the oracle, and what is pinned"): fp64 like the reference (cuda/common.h:26);
// dot products and axpy use explicit fma(), nothing else may be contracted (-ffp-contract=off),
// divisions are IEEE.  tests/ compare bit-for-bit with an independent CPU statement.
#include "cpf_device.h"

#include "cpf_walk.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>

namespace cpf {

// ------------------------------------------------------------------------------------------------
// fused step: one thread per particle, nCyc cycles per launch (1 = the reference's per-cycle
// structure; >1 keeps the particle in registers between cycles)
// ------------------------------------------------------------------------------------------------

template <class TRACER, bool BROWNIAN, bool REFLECT, bool STORE_VEL, bool VERTEX = false>
__device__ __forceinline__ void particle_cycles(const TRACER& tr, const MeshView& m, D3& P, int& cur, D3& v,
                                                uint64_t id, double dt, double sigma, uint32_t step0, int nCyc,
                                                uint32_t seed, StepStats& st, const VertexField* vf = nullptr) {
    for (int c = 0; c < nCyc; ++c) {
        if (cur < 0) { cur = CPF_CELL_FROZEN; break; }           // lost in the previous cycle: w = 0
        // ---- advect (cuda/particles.cu:355-362): disp = (P + dt*U[cell]) - P
        D3 Pn;
        if (VERTEX) {
            // (no tet of the cell holds P -- degenerate decomposition: the reference switches the particle off, :262-266)
            if (!vertex_velocity(*vf, P, cur, v)) { cur = CPF_CELL_FROZEN; break; }
            Pn = {P.x + dt * v.x, P.y + dt * v.y, P.z + dt * v.z};       // multiply, round, add: what the staged advect does
        } else {
            const double4 u = tr.velocity(cur);
            v = {u.x, u.y, u.z};
            Pn = axpy(dt, v, P);
        }
        ++st.steps;
        D3 disp = {Pn.x - P.x, Pn.y - P.y, Pn.z - P.z};
        if (BROWNIAN) {                                          // particles.cu:560-569
            const D3 xi = normal3(id, step0 + (uint32_t)c, seed);
            disp = axpy(sigma, xi, disp);
        }
        // ---- locate + reflect (ConvexQuery.cu:135-216, :320-436)
        D3 E = {P.x + disp.x, P.y + disp.y, P.z + disp.z};
        bool folded = false;
        if (BROWNIAN && REFLECT && m.zThin) {                    // one cell thick in z: cpf_walk.h, fold_z
            const int s0 = m.cellOff[cur];
            bool clear;
            const int nb = fold_z(E.z, m.planes[s0 + 4], m.planes[s0 + 5], clear);
            st.refl += nb;
            if (nb & 1) v.z = -v.z;
            folded = nb != 0;
        }
        D3 S = P, hit = P;
        int token = INT32_MIN, next = cur, outSlot = 0;
        bool reflected = false;
        for (int j = 0; j < kMaxReflect; ++j) {
            for (int h = 0; h < kMaxHops; ++h) {
                next = tr.trace(S, E, cur, token, outSlot);
                ++st.hops;
                if (next == cur || next < 0) break;
                token = cur;
                cur = next;
            }
            if (next >= 0 || !REFLECT) break;
            // wall: mirror end point and velocity about the boundary face (ConvexQuery.cu:286-309)
            hit = S; reflected = true; ++st.refl;
            const double4 pl = m.planes[outSlot];
            const D3 nn = {pl.x, pl.y, pl.z};
            const double sd = dot3(pl, E) - pl.w;
            E = axpy(-2.0 * sd, nn, E);
            v = axpy(-2.0 * dot3(pl, v), nn, v);
            token = next;
        }
        // ---- move (particles.cu:693-701); reflected: p = P_hit, disp = P_end - P_hit
        if (reflected) P = {hit.x + (E.x - hit.x), hit.y + (E.y - hit.y), hit.z + (E.z - hit.z)};
        else P = {P.x + disp.x, P.y + disp.y, folded ? E.z : P.z + disp.z};
        if (next < 0) { next = CPF_CELL_LOST; ++st.lost; }
        cur = next;
    }
}

template <int VARIANT, bool BROWNIAN, bool REFLECT, bool STORE_VEL>
__global__ __launch_bounds__(kBlock) void step_kernel(double* __restrict__ x, double* __restrict__ y,
                                                      double* __restrict__ z, int32_t* __restrict__ cell,
                                                      const int64_t* __restrict__ gid, double* __restrict__ vel,
                                                      int64_t n, double dt, double sigma, uint32_t step0, int nCyc,
                                                      uint32_t seed, MeshView m,
                                                      unsigned long long* __restrict__ counters) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    StepStats st = {0, 0, 0, 0};
    if (i < n) {
        int cur = cell[i];
        if (cur >= 0) {
            D3 P = {x[i], y[i], z[i]};
            D3 v = {0, 0, 0};
            const uint64_t id = gid ? (uint64_t)gid[i] : (uint64_t)i;
            const GlobalTracer<VARIANT> tr{m};
            particle_cycles<GlobalTracer<VARIANT>, BROWNIAN, REFLECT, STORE_VEL>(tr, m, P, cur, v, id, dt, sigma, step0,
                                                                                 nCyc, seed, st);
            x[i] = P.x; y[i] = P.y; z[i] = P.z;
            cell[i] = cur;
            if (STORE_VEL) { vel[3 * i] = v.x; vel[3 * i + 1] = v.y; vel[3 * i + 2] = v.z; }
        } else if (cur == CPF_CELL_LOST) {
            cell[i] = CPF_CELL_FROZEN;
        }
    }
    __shared__ unsigned sCnt[4];
    flush_stats(st, counters, sCnt);
}

// The fused cycle with the "VertexVelocity" advect (cuda/particles.cu:428-437; CPF_STEP_VERTEX_VELOCITY): the generic walk
// with the velocity interpolated at the particle's position.  Same stages in the same order as the reference's five calls with
// that mode, one launch; the advect shares its arithmetic with cpf_stage_advect_vertex (tests assert equality).
template <bool BROWNIAN, bool REFLECT, bool STORE_VEL>
__global__ __launch_bounds__(kBlock) void step_kernel_vertex(double* __restrict__ x, double* __restrict__ y,
                                                             double* __restrict__ z, int32_t* __restrict__ cell,
                                                             const int64_t* __restrict__ gid, double* __restrict__ vel,
                                                             int64_t n, double dt, double sigma, uint32_t step0, int nCyc,
                                                             uint32_t seed, MeshView m, VertexField vf,
                                                             unsigned long long* __restrict__ counters) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    StepStats st = {0, 0, 0, 0};
    if (i < n) {
        int cur = cell[i];
        if (cur >= 0) {
            D3 P = {x[i], y[i], z[i]};
            D3 v = {0, 0, 0};
            const uint64_t id = gid ? (uint64_t)gid[i] : (uint64_t)i;
            const GlobalTracer<kVariantGeneric> tr{m};
            particle_cycles<GlobalTracer<kVariantGeneric>, BROWNIAN, REFLECT, STORE_VEL, true>(tr, m, P, cur, v, id, dt, sigma, step0,
                                                                                               nCyc, seed, st, &vf);
            x[i] = P.x; y[i] = P.y; z[i] = P.z;
            cell[i] = cur;
            if (STORE_VEL) { vel[3 * i] = v.x; vel[3 * i + 1] = v.y; vel[3 * i + 2] = v.z; }
        } else if (cur == CPF_CELL_LOST) {
            cell[i] = CPF_CELL_FROZEN;
        }
    }
    __shared__ unsigned sCnt[4];
    flush_stats(st, counters, sCnt);
}

// ------------------------------------------------------------------------------------------------
// Wave-cooperative variant (all-hex meshes).  Mesh data lives in packed 256-byte cell records
// (6 planes | U | 6 neighbour ids).  Each round, a wave finds the distinct cells its busy lanes sit in
// (ballot + readlane), fetches up to four whole records with ONE coalesced 16-byte-per-lane load
// (16 lanes per record) into its private LDS slots, and every lane then reads its cell's planes with
// ds_read_b128 (same-cell lanes broadcast).  That replaces 18 scattered per-lane gathers per visit
// and their serialised round trips by a single L2 round trip per visit.  Lanes whose cell did not
// get a slot (more than four distinct cells in the wave: unsorted clouds) read the record directly
// from global memory in the same round, so every busy lane advances every round.
// The walk is written as a wave-synchronous state machine; per lane it performs exactly the loop
// nest of particle_cycles (<= 50 hops per segment, <= 5 reflections), in the same arithmetic order.
// ------------------------------------------------------------------------------------------------
constexpr int kCoopSlots = 4;
constexpr int kCoopMaxCells = 1 << 24;       // 32-bit record byte offsets in step_kernel_coop (launch_step falls back above that)
#ifndef CPF_COOP_BLOCK
#define CPF_COOP_BLOCK 128
#endif
constexpr int kCoopBlock = CPF_COOP_BLOCK;   // waves of a block share nothing but the dispatch slot

// Waves per SIMD the register allocator must make room for.  The plain cycle (no Brownian kick, no stored velocity,
// no statistics) fits 72 VGPRs = 7 waves; measured time follows T = 0.10 ms + 0.59 ms / waves on the bench cloud,
// i.e. the kernel is bound by the length of each wave's dependent chain, and resident waves are what hides it.
template <bool BROWNIAN, bool STORE_VEL, bool STATS>
struct CoopOccupancy { static constexpr int waves = (!STORE_VEL && !STATS) ? 7 : 1; };

template <bool BROWNIAN, bool REFLECT, bool STORE_VEL, bool STATS>
__global__ __launch_bounds__(kCoopBlock, (CoopOccupancy<BROWNIAN, STORE_VEL, STATS>::waves)) void step_kernel_coop(double* __restrict__ x, double* __restrict__ y,
                                                           double* __restrict__ z, int32_t* __restrict__ cell,
                                                           const int64_t* __restrict__ gid, double* __restrict__ vel,
                                                           int64_t n, double dt, double sigma, uint32_t step0,
                                                           int nCyc, uint32_t seed, MeshView m,
                                                           unsigned long long* __restrict__ counters) {
    __shared__ double4 sRec[kCoopBlock / 64][kCoopSlots][8];
    __shared__ unsigned sCnt[4];
    // Per-lane end point E and last wall hit point, parked in LDS between rounds (SoA: conflict-free).  Neither is
    // needed while the six planes are tested, and the kernel's speed is set by waves per SIMD: 12 VGPRs less
    // is one more resident wave.  A lane only ever reads back what it wrote itself, so no barrier is involved.
    __shared__ double sLane[6][kCoopBlock];            // one array: one address register + immediate offsets
    double(*sE)[kCoopBlock] = sLane;
    double(*sHit)[kCoopBlock] = sLane + 3;
    const int tid = threadIdx.x;
    const int lane = threadIdx.x & 63;
    double4(*slots)[8] = sRec[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];   // wave-uniform: lives in SGPRs
    const int64_t i = (int64_t)blockIdx.x * kCoopBlock + threadIdx.x;
    int cur = (i < n) ? cell[i] : CPF_CELL_FROZEN;
    const bool wasLost = cur == CPF_CELL_LOST;
    const bool hadParticle = cur >= 0;
    bool valid = hadParticle;
    D3 P = {0, 0, 0}, v = {0, 0, 0};
    if (valid) P = {x[i], y[i], z[i]};
    const uint64_t id = (valid && gid) ? (uint64_t)gid[i] : (uint64_t)i;
    StepStats st = {0, 0, 0, 0};

    for (int c = 0; c < nCyc; ++c) {
        if (valid && cur < 0) { cur = CPF_CELL_FROZEN; valid = false; }      // lost in the previous cycle: w = 0
        bool busy = valid;
        bool needAdvect = busy, reflected = false, lostNow = false;
        int token = INT32_MIN, h = 0, j = 0;
        D3 S = P;
        if (STATS && busy) ++st.steps;
        if (BROWNIAN && busy) {
            // Philox + Box-Muller (fp64 log / sqrt / sincos: a few hundred instructions and many registers) run HERE,
            // with almost nothing else live, and leave the three deviates in the lane's wall-hit LDS slot, which is
            // not needed before the first reflection and the kick is consumed in the first round.  Inside the round
            // loop they cost the whole walk its occupancy (97 VGPRs = 4 waves).
            const D3 xi = normal3(id, step0 + (uint32_t)c, seed);
            sHit[0][tid] = xi.x; sHit[1][tid] = xi.y; sHit[2][tid] = xi.z;
        }
        while (ballot64(busy) != 0ull) {
            // ---- distinct cells of the busy lanes -> slots (wave-uniform scalars)
            const unsigned long long busyMask = ballot64(busy);
            const D3 Epre = {sE[0][tid], sE[1][tid], sE[2][tid]};                  // requested before the cell discovery: its LDS round trip hides behind it (2 %)
            unsigned long long todo = busyMask;
            int myslot = -1, c0 = -1, c1 = -1, c2 = -1, c3 = -1;
#pragma unroll
            for (int k = 0; k < kCoopSlots; ++k) {
                if (todo != 0ull) {
                    const int leader = __ffsll((long long)todo) - 1;
                    const int ck = __builtin_amdgcn_readlane(cur, leader);
                    if (busy && cur == ck) myslot = k;
                    // the compare mask itself, combined in the scalar unit (see ballot64)
                    todo &= ~(__builtin_amdgcn_uicmp((unsigned)cur, (unsigned)ck, 32 /* eq */) & busyMask);
                    if (k == 0) c0 = ck; else if (k == 1) c1 = ck; else if (k == 2) c2 = ck; else c3 = ck;
                }
            }
            // ---- one coalesced load: lanes 16g..16g+15 copy record g (256 B) into LDS slot g
            const int g = lane >> 4;
            const int cg = g == 0 ? c0 : (g == 1 ? c1 : (g == 2 ? c2 : c3));
            if (cg >= 0) {
                // scalar base + 32-bit lane offset (records of up to 2^24 cells): no 64-bit address kept per lane
                const unsigned off = (unsigned)cg * 256u + (unsigned)(lane & 15) * 16u;
                const double2 val = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(m.cellRec) + off);
                reinterpret_cast<double2*>(slots[g])[lane & 15] = val;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- every busy lane does one cell visit
            if (busy) {
                int next, outSlot = 0;
                D3 E = S;
                if (!needAdvect) E = Epre;
                if (myslot >= 0) {
                    const double4* rec = slots[myslot];
                    if (needAdvect) {
                        const double4 u = rec[6];
                        v = {u.x, u.y, u.z};
                    }
                    if (needAdvect) {
                        const D3 Pn = axpy(dt, v, P);                              // particles.cu:355-362
                        D3 disp = {Pn.x - P.x, Pn.y - P.y, Pn.z - P.z};
                        if (BROWNIAN) {                                            // the deviates drawn before the round loop
                            const D3 xi = {sHit[0][tid], sHit[1][tid], sHit[2][tid]};
                            disp = axpy(sigma, xi, disp);
                        }
                        E = {P.x + disp.x, P.y + disp.y, P.z + disp.z};
                        if (BROWNIAN && REFLECT && m.zThin) {                      // one cell thick in z: cpf_walk.h, fold_z
                            bool clear;
                            const int nb = fold_z(E.z, rec[4], rec[5], clear);
                            if (STATS) st.refl += nb;
                            if (STORE_VEL && (nb & 1)) v.z = -v.z;
                        }
                        sE[0][tid] = E.x; sE[1][tid] = E.y; sE[2][tid] = E.z;
                        needAdvect = false;
                    }
                    next = trace_lds6(S, E, cur, rec, token, outSlot, m.zPairLast != 0);
                } else {
                    // no slot: more than kCoopSlots distinct cells in the wave (a cloud that is not kept sorted).
                    // Per-lane gathers keep such a wave moving; letting these lanes wait for a free slot instead
                    // measured 1.7x slower on an unsorted cloud and no faster on a sorted one.
                    const double4* rec = m.cellRec + 8 * (int64_t)cur;
                    if (needAdvect) {
                        const double4 u = rec[6];
                        v = {u.x, u.y, u.z};
                        const D3 Pn = axpy(dt, v, P);
                        D3 disp = {Pn.x - P.x, Pn.y - P.y, Pn.z - P.z};
                        if (BROWNIAN) {                                            // the deviates drawn before the round loop
                            const D3 xi = {sHit[0][tid], sHit[1][tid], sHit[2][tid]};
                            disp = axpy(sigma, xi, disp);
                        }
                        E = {P.x + disp.x, P.y + disp.y, P.z + disp.z};
                        if (BROWNIAN && REFLECT && m.zThin) {
                            bool clear;
                            const int nb = fold_z(E.z, rec[4], rec[5], clear);
                            if (STATS) st.refl += nb;
                            if (STORE_VEL && (nb & 1)) v.z = -v.z;
                        }
                        sE[0][tid] = E.x; sE[1][tid] = E.y; sE[2][tid] = E.z;
                        needAdvect = false;
                    }
                    next = trace_fixed<6, false>(S, E, cur, rec, reinterpret_cast<const int32_t*>(rec + 7), token, outSlot, 0);
                }
                if (STATS) ++st.hops;
                if (next == cur) {
                    busy = false;                                                  // segment ends in this cell
                } else if (next < 0) {                                             // boundary face
                    if (!REFLECT) { busy = false; lostNow = true; }
                    else {
                        // mirror end point and velocity about the wall (ConvexQuery.cu:286-309)
                        sHit[0][tid] = S.x; sHit[1][tid] = S.y; sHit[2][tid] = S.z;
                        reflected = true; if (STATS) ++st.refl;
                        // the wall's plane is fetched here, by the few lanes that reflect: reading it after every
                        // visit cost 2 KB of LDS return bandwidth per wave-round, and LDS bandwidth is what the
                        // throughput-bound part of this kernel's time consists of
                        const double4 wallPlane = myslot >= 0 ? slots[myslot][outSlot] : m.cellRec[8 * (int64_t)cur + outSlot];
                        const D3 nn = {wallPlane.x, wallPlane.y, wallPlane.z};
                        const double sd = dot3(wallPlane, E) - wallPlane.w;
                        E = axpy(-2.0 * sd, nn, E);
                        sE[0][tid] = E.x; sE[1][tid] = E.y; sE[2][tid] = E.z;
                        v = axpy(-2.0 * dot3(wallPlane, v), nn, v);
                        token = next;
                        h = 0;
                        if (++j == kMaxReflect) { busy = false; lostNow = true; }  // still on a wall after 5 bounces
                    }
                } else {
                    token = cur;
                    cur = next;
                    if (++h == kMaxHops) busy = false;                             // hop cap: keep the last cell
                }
            }
            __builtin_amdgcn_wave_barrier();                                       // slots are rewritten next round
        }
        // ---- move (particles.cu:693-701); reflected: p = P_hit, disp = P_end - P_hit; else P + disp == E
        if (valid) {
            const D3 E = {sE[0][tid], sE[1][tid], sE[2][tid]};
            if (reflected) {
                const D3 hit = {sHit[0][tid], sHit[1][tid], sHit[2][tid]};
                P = {hit.x + (E.x - hit.x), hit.y + (E.y - hit.y), hit.z + (E.z - hit.z)};
            } else P = E;
            if (lostNow) { cur = CPF_CELL_LOST; if (STATS) ++st.lost; }
        }
    }
    // the particle index again, from the block id (kept out of the long-lived registers on purpose)
    unsigned bid = blockIdx.x;
    asm volatile("" : "+s"(bid));
    const int64_t io = (int64_t)bid * kCoopBlock + threadIdx.x;
    if (hadParticle) {
        x[io] = P.x; y[io] = P.y; z[io] = P.z;
        cell[io] = cur;
        if (STORE_VEL) { vel[3 * io] = v.x; vel[3 * io + 1] = v.y; vel[3 * io + 2] = v.z; }
    } else if (wasLost) {
        cell[io] = CPF_CELL_FROZEN;
    }
    if (STATS) flush_stats(st, counters, sCnt);
}

template <int V, bool B, bool R>
static void launch_step_sv(bool storeVel, dim3 grid, hipStream_t st, double* x, double* y, double* z, int32_t* cell,
                           const int64_t* gid, double* vel, int64_t n, double dt, double sigma, uint32_t step0,
                           int nCyc, uint32_t seed, const MeshView& m, unsigned long long* counters) {
    if (storeVel)
        hipLaunchKernelGGL((step_kernel<V, B, R, true>), grid, dim3(kBlock), 0, st, x, y, z, cell, gid, vel, n, dt,
                           sigma, step0, nCyc, seed, m, counters);
    else
        hipLaunchKernelGGL((step_kernel<V, B, R, false>), grid, dim3(kBlock), 0, st, x, y, z, cell, gid, vel, n, dt,
                           sigma, step0, nCyc, seed, m, counters);
}

template <bool B, bool R>
static void launch_step_coop_sv(bool storeVel, dim3 grid, hipStream_t st, double* x, double* y, double* z,
                                int32_t* cell, const int64_t* gid, double* vel, int64_t n, double dt, double sigma,
                                uint32_t step0, int nCyc, uint32_t seed, const MeshView& m,
                                unsigned long long* counters) {
    const dim3 cgrid((unsigned)((n + kCoopBlock - 1) / kCoopBlock));
    (void)grid;
    // statistics are a template flag here: the four per-lane counters cost registers, and registers are waves
#define CPF_LAUNCH_COOP(SV, ST)                                                                                       \
    hipLaunchKernelGGL((step_kernel_coop<B, R, SV, ST>), cgrid, dim3(kCoopBlock), 0, st, x, y, z, cell, gid, vel, n, dt, \
                       sigma, step0, nCyc, seed, m, counters)
    if (storeVel) { if (counters) CPF_LAUNCH_COOP(true, true); else CPF_LAUNCH_COOP(true, false); }
    else { if (counters) CPF_LAUNCH_COOP(false, true); else CPF_LAUNCH_COOP(false, false); }
#undef CPF_LAUNCH_COOP
}

template <int V>
static void launch_step_v(bool brown, bool reflect, bool storeVel, dim3 grid, hipStream_t st, double* x, double* y,
                          double* z, int32_t* cell, const int64_t* gid, double* vel, int64_t n, double dt,
                          double sigma, uint32_t step0, int nCyc, uint32_t seed, const MeshView& m,
                          unsigned long long* counters) {
    if (brown) {
        if (reflect) launch_step_sv<V, true, true>(storeVel, grid, st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters);
        else launch_step_sv<V, true, false>(storeVel, grid, st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters);
    } else {
        if (reflect) launch_step_sv<V, false, true>(storeVel, grid, st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters);
        else launch_step_sv<V, false, false>(storeVel, grid, st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters);
    }
}

int effective_step_variant(int variant, const MeshView& m, bool haveStream, int cyclesPerLaunch, int coopMaxCells) {
    if (coopMaxCells <= 0 || coopMaxCells > kCoopMaxCells) coopMaxCells = kCoopMaxCells;
    if (!m.allHex) {
        // not all-hex: the streaming kernel where the mesh layer built mixed records for it (few cells with more than six
        // faces: they take the CSR walk inside the kernel), else the generic CSR walk; the other variants need 6 faces per cell
        const bool wantsStream = variant == kVariantAuto || variant == kVariantStream || variant == kVariantCoop;
        return (m.mixed && m.cellRec && haveStream && wantsStream) ? kVariantStream : kVariantGeneric;
    }
    // Several cycles fused into one launch (CPF_STEP_FUSE_CYCLES: what advect.H does between two output points): the
    // particle stream is loaded and stored once per launch, so hiding it behind the walk buys nothing, and the
    // wave-cooperative kernel's 8 waves per SIMD (the streaming kernel: 7) win -- by less since round 3, measured per
    // cycle on pitzDaily: 3 cycles per launch the streaming kernel is 4 % FASTER (0.0982 vs 0.1026 ms; with the Brownian
    // kick 0.181 vs 0.190), 8 cycles 2 % slower (0.0930 vs 0.0911; kick: equal).  Round 2: 5 % slower at 3, 10 % at 8.
    // (Final build of round 3, exact face normals: 8 cycles 0.0865 vs 0.0846 ms per cycle, with the kick 0.167 vs 0.172.)
    // Round 4: the streaming kernel's flat walk and box records turned that around -- 8 cycles per launch, pitzDaily 0.0799 vs
    // 0.0863 ms per cycle (16 cycles: 0.0783 vs 0.0822), with the kick 0.159 vs 0.169, TJunction 0.0855 vs 0.1216 -- and
    // kFusedCoopCycles went up to "never" (the switch stays for builds without the streaming kernel).
    if (variant == kVariantAuto)
        variant = (haveStream && !(cyclesPerLaunch >= kFusedCoopCycles && m.nCells <= coopMaxCells)) ? kVariantStream : kVariantCoop;
    if ((variant == kVariantStream || variant == kVariantAhead) && !haveStream) variant = kVariantCoop;
    // the wave-cooperative kernel addresses records with a 32-bit byte offset (256 B x 2^24 cells)
    if (variant == kVariantCoop && m.nCells > coopMaxCells) variant = haveStream ? kVariantStream : kVariantGeneric;
#ifndef CPF_EXPERIMENTS
    // (variants 1, 2 and 5 are not in this build: cpf_set_option refuses them; belt and braces)
    if (variant == kVariantFixed || variant == kVariantFixedScalar) variant = kVariantCoop;
    if (variant == kVariantAhead) variant = haveStream ? kVariantStream : kVariantCoop;
#endif
    return variant;
}

hipError_t launch_step(hipStream_t st, double* x, double* y, double* z, int32_t* cell, const int64_t* gid,
                       double* vel, int64_t n, double dt, double D, uint32_t step0, int nCyc, uint32_t seed,
                       bool reflect, bool storeVel, const MeshView& m, unsigned long long* counters, int variant,
                       StreamState* ss) {
    if (n <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    const bool brown = D > 0.0;
    const double sigma = brown ? sqrt(2.00 * D * dt) : 0.0;   // particles.cu:564
    variant = effective_step_variant(variant, m, ss != nullptr, nCyc, ss ? ss->coopMaxCells : 0);
    switch (variant) {
#ifdef CPF_EXPERIMENTS
        case kVariantAhead:
            // lanes that run ahead into the next tile: one plain cycle per launch only; everything else streams
            if (!brown && !storeVel && nCyc == 1) return launch_step_ahead(st, x, y, z, cell, n, dt, reflect, m, counters, *ss, vel);
            [[fallthrough]];
#endif
        case kVariantStream:
            return launch_step_stream(st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, brown, reflect, storeVel, m,
                                      counters, *ss);
#ifdef CPF_EXPERIMENTS
        case kVariantFixed:
            launch_step_v<kVariantFixed>(brown, reflect, storeVel, grid, st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters);
            break;
        case kVariantFixedScalar:
            launch_step_v<kVariantFixedScalar>(brown, reflect, storeVel, grid, st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters);
            break;
#endif
        case kVariantCoop:
            if (brown) {
                if (reflect) launch_step_coop_sv<true, true>(storeVel, grid, st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters);
                else launch_step_coop_sv<true, false>(storeVel, grid, st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters);
            } else {
                if (reflect) launch_step_coop_sv<false, true>(storeVel, grid, st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters);
                else launch_step_coop_sv<false, false>(storeVel, grid, st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters);
            }
            break;
        default:
            launch_step_v<kVariantGeneric>(brown, reflect, storeVel, grid, st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, counters);
    }
    return hipGetLastError();
}

// the streaming kernel takes the cycle where it would take the cell-constant one (variant 4: -1 on a mesh with cell records),
// the decomposition is admitted to the cone locate (its advect never fails there) and the mesh is all-hex
// (twelve tets a cell -- the reference's only decomposition, src/initCuda.H:64 -- is what the kernel's staged locate is built for)
bool step_vertex_streams(const MeshView& m, const double* cone, int tetsPerCell, int variant, const StreamState* ss, int nCyc) {
    return cone != nullptr && tetsPerCell == 12 && ss != nullptr && stream_vertex_capable(m) &&
           effective_step_variant(variant, m, true, nCyc, ss->coopMaxCells) == kVariantStream;
}

hipError_t launch_step_vertex(hipStream_t st, double* x, double* y, double* z, int32_t* cell, const int64_t* gid,
                              double* vel, int64_t n, double dt, double D, uint32_t step0, int nCyc, uint32_t seed,
                              bool reflect, bool storeVel, const MeshView& m, unsigned long long* counters,
                              const double* pos, const int32_t* tets, int tetsPerCell, const double* vertVel, const double* cone,
                              const double* apex, int variant, StreamState* ss) {
    if (n <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    const bool brown = D > 0.0;
    const double sigma = brown ? sqrt(2.00 * D * dt) : 0.0;   // particles.cu:564
    const VertexField vf{pos, tets, vertVel, tetsPerCell, cone, reinterpret_cast<const double4*>(apex)};
    if (step_vertex_streams(m, cone, tetsPerCell, variant, ss, nCyc))
        return launch_step_stream_vertex(st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, brown, reflect, storeVel, m, counters, *ss, vf);
#define CPF_VTX(B, R, SV) hipLaunchKernelGGL((step_kernel_vertex<B, R, SV>), grid, dim3(kBlock), 0, st, x, y, z, cell, gid, vel, n, dt, sigma, step0, nCyc, seed, m, vf, counters)
    if (brown) {
        if (reflect) { if (storeVel) CPF_VTX(true, true, true); else CPF_VTX(true, true, false); }
        else { if (storeVel) CPF_VTX(true, false, true); else CPF_VTX(true, false, false); }
    } else {
        if (reflect) { if (storeVel) CPF_VTX(false, true, true); else CPF_VTX(false, true, false); }
        else { if (storeVel) CPF_VTX(false, false, true); else CPF_VTX(false, false, false); }
    }
#undef CPF_VTX
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Stage-by-stage kernels on the REFERENCE's own array layouts (Particle = double4 AoS, vec4d
// disps / vels, int ids), one per wrapper of third_party/RTXAdvect/cuda/common.h and
// query/ConvexQuery.h, for hosts that keep the reference's five-call cycle (compat shims).
// ids are cell ids here (reference: tet ids, cell = tet / 12); the wall encoding between
// locate and reflect is the reference's: -(cell at the START of the step + 1).
// The fused step_kernel above is the same arithmetic in the same order (tests assert equality).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void stage_advect_kernel(double4* __restrict__ P, const int32_t* __restrict__ ids,
                                                              double4* __restrict__ vels, double4* __restrict__ disps,
                                                              double dt, int64_t n, MeshView m) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    double4 p = P[i];
    if (!p.w) return;
    const int id = ids[i];
    if (id < 0) { p.w = 0.0; P[i] = p; return; }                 // particles.cu:333-338
    const double4 u = m.U[id];
    const D3 v = {u.x, u.y, u.z}, Pp = {p.x, p.y, p.z};
    const D3 Pn = axpy(dt, v, Pp);
    vels[i] = make_double4(v.x, v.y, v.z, -1.0);
    disps[i] = make_double4(Pn.x - Pp.x, Pn.y - Pp.y, Pn.z - Pp.z, -1.0);
}

// "VertexVelocity" advect (cuda/particles.cu:244-313) on cell ids; see cpf_stage_advect_vertex in include/cpf.h.
// Plain products and sums in the reference's order (no fma: the file is built with -ffp-contract=off), so that
// oracle/cellwalk.c's cw_advect_vertex is reproduced bit for bit.
__global__ __launch_bounds__(kBlock) void stage_advect_vertex_kernel(double4* __restrict__ P, const int32_t* __restrict__ ids,
                                                                     double4* __restrict__ vels, double4* __restrict__ disps,
                                                                     double dt, int64_t n, VertexField f) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    double4 p = P[i];
    if (!p.w) return;
    const int c = ids[i];
    if (c < 0) { p.w = 0.0; P[i] = p; return; }                  // particles.cu:262-266
    D3 v;
    if (!vertex_velocity(f, D3{p.x, p.y, p.z}, c, v)) { p.w = 0.0; P[i] = p; return; }
    const D3 Pn = {p.x + dt * v.x, p.y + dt * v.y, p.z + dt * v.z};
    vels[i] = make_double4(v.x, v.y, v.z, -1.0);
    disps[i] = make_double4(Pn.x - p.x, Pn.y - p.y, Pn.z - p.z, -1.0);
}

__global__ __launch_bounds__(kBlock) void stage_brownian_kernel(const double4* __restrict__ P,
                                                                double4* __restrict__ disps, double sigma, int64_t n,
                                                                uint32_t step, uint32_t seed) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    if (!P[i].w) return;
    double4 d = disps[i];
    const D3 r = axpy(sigma, normal3((uint64_t)i, step, seed), D3{d.x, d.y, d.z});
    disps[i] = make_double4(r.x, r.y, r.z, d.w);
}

__global__ __launch_bounds__(kBlock) void stage_locate_kernel(const double4* __restrict__ P,
                                                              const double4* __restrict__ disps,
                                                              int32_t* __restrict__ ids, int64_t n, MeshView m) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double4 p = P[i];
    if (!p.w) return;
    const int start = ids[i];
    if (start < 0) return;
    const double4 d = disps[i];
    const D3 E = {p.x + d.x, p.y + d.y, p.z + d.z};
    D3 S = {p.x, p.y, p.z};
    int cur = start, next = start, token = INT32_MIN, outSlot = 0;
    for (int h = 0; h < kMaxHops; ++h) {
        next = trace_in_cell(S, E, cur, m, token, outSlot);
        if (next == cur || next < 0) break;
        token = cur;
        cur = next;
    }
    ids[i] = next < 0 ? -(start + 1) : next;                      // ConvexQuery.cu:204-215
}

__global__ __launch_bounds__(kBlock) void stage_reflect_kernel(int32_t* __restrict__ ids, double4* __restrict__ P,
                                                               double4* __restrict__ vels, double4* __restrict__ disps,
                                                               int64_t n, MeshView m) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    double4 p = P[i];
    if (!p.w) return;
    const int id = ids[i];
    if (id >= 0) return;                                          // only particles that hit a wall
    double4 d = disps[i], vv = vels[i];
    D3 S = {p.x, p.y, p.z};
    D3 E = {p.x + d.x, p.y + d.y, p.z + d.z};
    D3 v = {vv.x, vv.y, vv.z};
    D3 hit = {-1.0, -1.0, -1.0};                                  // ConvexQuery.cu:352
    int cur = -id - 1, next = cur, token = INT32_MIN, outSlot = 0;
    for (int j = 0; j < kMaxReflect; ++j) {
        for (int h = 0; h < kMaxHops; ++h) {
            next = trace_in_cell(S, E, cur, m, token, outSlot);
            if (next == cur || next < 0) break;
            token = cur;
            cur = next;
        }
        if (next >= 0) break;
        hit = S;
        const double4 pl = m.planes[outSlot];
        const D3 nn = {pl.x, pl.y, pl.z};
        const double sd = dot3(pl, E) - pl.w;
        E = axpy(-2.0 * sd, nn, E);
        v = axpy(-2.0 * dot3(pl, v), nn, v);
        token = next;
    }
    P[i] = make_double4(hit.x, hit.y, hit.z, p.w);
    disps[i] = make_double4(E.x - hit.x, E.y - hit.y, E.z - hit.z, d.w);
    vels[i] = make_double4(v.x, v.y, v.z, vv.w);
    ids[i] = next < 0 ? CPF_CELL_LOST : next;
}

__global__ __launch_bounds__(kBlock) void stage_move_kernel(double4* __restrict__ P, double4* __restrict__ disps,
                                                            int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    double4 p = P[i];
    if (!p.w) return;
    double4 d = disps[i];
    p.x += d.x; p.y += d.y; p.z += d.z;
    d.x = 0.0; d.y = 0.0; d.z = 0.0;
    P[i] = p; disps[i] = d;
}

__global__ __launch_bounds__(kBlock) void aos_to_soa_kernel(const double4* __restrict__ P, double* x, double* y,
                                                            double* z, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) { const double4 p = P[i]; x[i] = p.x; y[i] = p.y; z[i] = p.z; }
}
__global__ __launch_bounds__(kBlock) void soa_to_aos_kernel(const double* x, const double* y, const double* z,
                                                            double4* __restrict__ P, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) P[i] = make_double4(x[i], y[i], z[i], 1.0);
}

static inline dim3 grid_of(int64_t n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

hipError_t launch_stage_advect(hipStream_t st, double* P, const int32_t* ids, double* vels, double* disps, double dt,
                               int64_t n, const MeshView& m) {
    if (n > 0) hipLaunchKernelGGL(stage_advect_kernel, grid_of(n), dim3(kBlock), 0, st, (double4*)P, ids, (double4*)vels, (double4*)disps, dt, n, m);
    return hipGetLastError();
}
// "ConstantVelocity" advect (cuda/particles.cu:376-399): the particle's own stored velocity, first-order Euler
__global__ __launch_bounds__(kBlock) void stage_advect_const_kernel(double4* __restrict__ P, const int32_t* __restrict__ ids,
                                                                    const double4* __restrict__ vels, double4* __restrict__ disps,
                                                                    double dt, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    double4 p = P[i];
    if (!p.w) return;
    if (ids[i] < 0) { p.w = 0.0; P[i] = p; return; }             // particles.cu:390-394
    const double4 v = vels[i];
    disps[i] = make_double4(v.x * dt, v.y * dt, v.z * dt, -1.0);  // particles.cu:398
}
hipError_t launch_stage_advect_const(hipStream_t st, double* P, const int32_t* ids, const double* vels, double* disps, double dt,
                                     int64_t n) {
    if (n > 0) hipLaunchKernelGGL(stage_advect_const_kernel, grid_of(n), dim3(kBlock), 0, st, (double4*)P, ids, (const double4*)vels, (double4*)disps, dt, n);
    return hipGetLastError();
}
hipError_t launch_stage_advect_vertex(hipStream_t st, double* P, const int32_t* ids, double* vels, double* disps, double dt,
                                      int64_t n, const double* pos, const int32_t* tets, int tetsPerCell,
                                      const double* vertVel, const double* cone, const double* apex) {
    if (n > 0)
        hipLaunchKernelGGL(stage_advect_vertex_kernel, grid_of(n), dim3(kBlock), 0, st, (double4*)P, ids, (double4*)vels,
                           (double4*)disps, dt, n, VertexField{pos, tets, vertVel, tetsPerCell, cone, reinterpret_cast<const double4*>(apex)});
    return hipGetLastError();
}

// the cone-locate tables of the "VertexVelocity" advect (see VertexField): per tet the rows of the inverse of [B-A C-A D-A] --
// (B-A, C-A, D-A) coordinates of P - A, a guess only -- and 1 / det4(A, B, C, D) by the expressions vertex_velocity() itself uses
__global__ __launch_bounds__(kBlock) void vertex_cone_tables_kernel(const double* __restrict__ pos, const int32_t* __restrict__ tets,
                                                                    int64_t nTets, int tetsPerCell, double* __restrict__ cone,
                                                                    double4* __restrict__ apex) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= nTets) return;
    auto ld = [](const double* a, int k) { return D3{a[3 * (int64_t)k], a[3 * (int64_t)k + 1], a[3 * (int64_t)k + 2]}; };
    const int32_t* ix = tets + 4 * t;
    const D3 A = ld(pos, ix[0]), B = ld(pos, ix[1]), C = ld(pos, ix[2]), D = ld(pos, ix[3]);
    const double den = det4(A, B, C, D);
    const D3 b = {B.x - A.x, B.y - A.y, B.z - A.z}, c = {C.x - A.x, C.y - A.y, C.z - A.z}, d = {D.x - A.x, D.y - A.y, D.z - A.z};
    const D3 cd = {c.y * d.z - c.z * d.y, c.z * d.x - c.x * d.z, c.x * d.y - c.y * d.x};
    const D3 db = {d.y * b.z - d.z * b.y, d.z * b.x - d.x * b.z, d.x * b.y - d.y * b.x};
    const D3 bc = {b.y * c.z - b.z * c.y, b.z * c.x - b.x * c.z, b.x * c.y - b.y * c.x};
    const double vol = b.x * cd.x + b.y * cd.y + b.z * cd.z;
    const double s = vol != 0.0 ? 1.0 / vol : 0.0;
    double* g = cone + kConeDoubles * t;
    g[0] = cd.x * s; g[1] = cd.y * s; g[2] = cd.z * s;
    g[3] = db.x * s; g[4] = db.y * s; g[5] = db.z * s;
    g[6] = bc.x * s; g[7] = bc.y * s; g[8] = bc.z * s;
    g[9] = den == 0.0 ? 0.0 : 1. / den;        // (a mesh with a flat tet is not admitted to the cone locate: cpf_set_tets)
    g[10] = B.x; g[11] = B.y; g[12] = B.z; g[13] = C.x; g[14] = C.y; g[15] = C.z; g[16] = D.x; g[17] = D.y; g[18] = D.z;
    g[31] = 0.0;
    if (t % tetsPerCell == 0) apex[t / tetsPerCell] = make_double4(A.x, A.y, A.z, 0.0);
}
// the velocity part of the tet records: [19..30] = the velocities of the tet's four vertices (every cpf_set_vertex_velocity)
__global__ __launch_bounds__(kBlock) void vertex_record_velocity_kernel(const int32_t* __restrict__ tets, const double* __restrict__ vel,
                                                                        int64_t nTets, double* __restrict__ cone) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= 4 * nTets) return;
    const int64_t t = i >> 2;
    const int k = (int)(i & 3);
    const int64_t vtx = tets[i];
    double* g = cone + kConeDoubles * t + 19 + 3 * k;
    g[0] = vel[3 * vtx]; g[1] = vel[3 * vtx + 1]; g[2] = vel[3 * vtx + 2];
}
hipError_t launch_vertex_cone_tables(hipStream_t st, const double* pos, const int32_t* tets, int64_t nTets, int tetsPerCell, double* cone,
                                     double* apex) {
    if (nTets > 0) hipLaunchKernelGGL(vertex_cone_tables_kernel, grid_of(nTets), dim3(kBlock), 0, st, pos, tets, nTets, tetsPerCell, cone, (double4*)apex);
    return hipGetLastError();
}
hipError_t launch_vertex_record_velocity(hipStream_t st, const int32_t* tets, const double* vel, int64_t nTets, double* cone) {
    if (nTets > 0) hipLaunchKernelGGL(vertex_record_velocity_kernel, grid_of(4 * nTets), dim3(kBlock), 0, st, tets, vel, nTets, cone);
    return hipGetLastError();
}
hipError_t launch_stage_brownian(hipStream_t st, const double* P, double* disps, double dt, int64_t n, double D,
                                 uint32_t step, uint32_t seed) {
    if (n > 0 && D > 0.0)
        hipLaunchKernelGGL(stage_brownian_kernel, grid_of(n), dim3(kBlock), 0, st, (const double4*)P, (double4*)disps, sqrt(2.00 * D * dt), n, step, seed);
    return hipGetLastError();
}
hipError_t launch_stage_locate(hipStream_t st, const double* P, const double* disps, int32_t* ids, int64_t n,
                               const MeshView& m) {
    if (n > 0) hipLaunchKernelGGL(stage_locate_kernel, grid_of(n), dim3(kBlock), 0, st, (const double4*)P, (const double4*)disps, ids, n, m);
    return hipGetLastError();
}
hipError_t launch_stage_reflect(hipStream_t st, int32_t* ids, double* P, double* vels, double* disps, int64_t n,
                                const MeshView& m) {
    if (n > 0) hipLaunchKernelGGL(stage_reflect_kernel, grid_of(n), dim3(kBlock), 0, st, ids, (double4*)P, (double4*)vels, (double4*)disps, n, m);
    return hipGetLastError();
}
hipError_t launch_stage_move(hipStream_t st, double* P, double* disps, int64_t n) {
    if (n > 0) hipLaunchKernelGGL(stage_move_kernel, grid_of(n), dim3(kBlock), 0, st, (double4*)P, (double4*)disps, n);
    return hipGetLastError();
}
hipError_t launch_aos_to_soa(hipStream_t st, const double* P, double* x, double* y, double* z, int64_t n) {
    if (n > 0) hipLaunchKernelGGL(aos_to_soa_kernel, grid_of(n), dim3(kBlock), 0, st, (const double4*)P, x, y, z, n);
    return hipGetLastError();
}
hipError_t launch_soa_to_aos(hipStream_t st, const double* x, const double* y, const double* z, double* P, int64_t n) {
    if (n > 0) hipLaunchKernelGGL(soa_to_aos_kernel, grid_of(n), dim3(kBlock), 0, st, x, y, z, (double4*)P, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// initial locate: bin grid + point-in-convex-cell plane test (replaces OptiX query + baryQuery,
// query/RTQuery.cu:295-310).  Lowest-numbered containing cell wins (bins list cells ascending).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void locate_initial_kernel(const double* __restrict__ x,
                                                                const double* __restrict__ y,
                                                                const double* __restrict__ z,
                                                                int32_t* __restrict__ cell, int64_t n, MeshView m,
                                                                GridView g) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const D3 P = {x[i], y[i], z[i]};
    int found = CPF_CELL_LOST;
    const bool inBox = P.x >= g.lo[0] && P.x <= g.hi[0] && P.y >= g.lo[1] && P.y <= g.hi[1] && P.z >= g.lo[2] &&
                       P.z <= g.hi[2];
    if (inBox) {
        int b[3];
        const double pv[3] = {P.x, P.y, P.z};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            long long q = (long long)floor((pv[k] - g.origin[k]) * g.invBin[k]);
            q = q < 0 ? 0 : (q > g.dims[k] - 1 ? g.dims[k] - 1 : q);
            b[k] = (int)q;
        }
        const int64_t bin = ((int64_t)b[2] * g.dims[1] + b[1]) * g.dims[0] + b[0];
        for (int k = g.binOff[bin]; k < g.binOff[bin + 1] && found < 0; ++k) {
            const int c = g.binCells[k];
            bool inside = true;
            for (int s = m.cellOff[c]; s < m.cellOff[c + 1]; ++s) {
                const double4 pl = m.planes[s];
                if (!(plane_dist(pl, P) <= 0.0)) { inside = false; break; }
            }
            if (inside) found = c;
        }
    }
    cell[i] = found;
}

hipError_t launch_locate_initial(hipStream_t st, const double* x, const double* y, const double* z, int32_t* cell,
                                 int64_t n, const MeshView& m, const GridView& g) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(locate_initial_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, x, y,
                       z, cell, n, m, g);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// seeding: cudaInitParticles (cuda/particles.cu:78-108) + LCG<16> (owl/common/math/random.h:56-91)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void seed_box_kernel(double* __restrict__ x, double* __restrict__ y,
                                                          double* __restrict__ z, int64_t first, int64_t n,
                                                          D3 lower, D3 size, int order) {
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= n) return;
    const int64_t id = first + k;
    uint32_t v0 = (uint32_t)(id % 128), v1 = (uint32_t)(id / 128), s0 = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    uint32_t state = v0;
    float r[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        state = 1664525u * state + 1013904223u;
        r[q] = ldexpf((float)state, -32);
    }
    if (order) { const float t = r[0]; r[0] = r[2]; r[2] = t; }
    x[k] = lower.x + (double)r[0] * size.x;
    y[k] = lower.y + (double)r[1] * size.y;
    z[k] = lower.z + (double)r[2] * size.z;
}

hipError_t launch_seed_box(hipStream_t st, double* x, double* y, double* z, int64_t first, int64_t n,
                           const double lower[3], const double upper[3], int order) {
    if (n <= 0) return hipSuccess;
    const D3 lo = {lower[0], lower[1], lower[2]};
    const D3 sz = {upper[0] - lower[0], upper[1] - lower[1], upper[2] - lower[2]};
    hipLaunchKernelGGL(seed_box_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, x, y, z,
                       first, n, lo, sz, order);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// small utility kernels
// ------------------------------------------------------------------------------------------------
__global__ void iota_kernel(int32_t* p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) p[i] = (int32_t)i;
}
__global__ void iota64_kernel(int64_t* p, int64_t n, int64_t first) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) p[i] = first + i;
}
template <typename T>
__global__ void gather_kernel(const T* __restrict__ src, T* __restrict__ dst, const int32_t* __restrict__ perm,
                              int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}
__global__ void gather3_kernel(const double* __restrict__ src, double* __restrict__ dst,
                               const int32_t* __restrict__ perm, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) {
        const int64_t j = perm[i];
        dst[3 * i] = src[3 * j]; dst[3 * i + 1] = src[3 * j + 1]; dst[3 * i + 2] = src[3 * j + 2];
    }
}
// The sort's permutation applied to several arrays per launch (perm is read once, the scattered reads of one
// particle's fields are in flight together), then one launch copies the staged arrays back.
__global__ void gather_xyz_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                  const double* __restrict__ z, double* __restrict__ sx, double* __restrict__ sy,
                                  double* __restrict__ sz, const int32_t* __restrict__ perm, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) {
        const int64_t j = perm[i];
        const double a = x[j], b = y[j], c = z[j];
        sx[i] = a; sy[i] = b; sz[i] = c;
    }
}
__global__ void copy_xyz_kernel(double* __restrict__ x, double* __restrict__ y, double* __restrict__ z,
                                const double* __restrict__ sx, const double* __restrict__ sy,
                                const double* __restrict__ sz, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) { x[i] = sx[i]; y[i] = sy[i]; z[i] = sz[i]; }
}
__global__ void gather_ids_kernel(const int32_t* __restrict__ cell, const int64_t* __restrict__ gid,
                                  int32_t* __restrict__ sc, int64_t* __restrict__ sg,
                                  const int32_t* __restrict__ perm, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) {
        const int64_t j = perm[i];
        sc[i] = cell[j];
        if (gid) sg[i] = gid[j];
    }
}
// the sort's gathers in one pass (into a second set of arrays): one read of the permutation, five independent gathers in
// flight per thread; a live particle's cell comes out of its sorted key (no gather), a lost / frozen one's is fetched
__global__ __launch_bounds__(kBlock) void gather_all_kernel(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z,
                                                            const int32_t* __restrict__ cell, const int64_t* __restrict__ gid,
                                                            double* __restrict__ ox, double* __restrict__ oy, double* __restrict__ oz,
                                                            int32_t* __restrict__ ocell, int64_t* __restrict__ ogid,
                                                            const int32_t* __restrict__ perm, const uint32_t* __restrict__ keys, int nSub, int64_t n,
                                                            bool cellFromKey) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int64_t j = perm[i];
    const uint32_t k = cellFromKey ? keys[i] : 0xFFFFFFFFu;
    const double a = x[j], b = y[j], c = z[j];
    const int64_t g = gid ? gid[j] : 0;
    const int32_t cc = (cellFromKey && k != 0xFFFFFFFFu) ? (int32_t)(k >> nSub) : cell[j];
    ox[i] = a; oy[i] = b; oz[i] = c;
    ocell[i] = cc;
    if (gid) ogid[i] = g;
}
__global__ void copy_ids_kernel(int32_t* __restrict__ cell, int64_t* __restrict__ gid, const int32_t* __restrict__ sc,
                                const int64_t* __restrict__ sg, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) { cell[i] = sc[i]; if (gid) gid[i] = sg[i]; }
}
// AoS <-> SoA conversion for the reference-shaped accessors (Particle = double4)
__global__ void unpack_xyz_kernel(const double* __restrict__ xyz, double* x, double* y, double* z, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) { x[i] = xyz[3 * i]; y[i] = xyz[3 * i + 1]; z[i] = xyz[3 * i + 2]; }
}
// scatter back to original-id order: out[gid] = (x,y,z,w)
__global__ void pack_by_gid_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                   const double* __restrict__ z, const int32_t* __restrict__ cell,
                                   const int64_t* __restrict__ gid, const double* __restrict__ vel,
                                   double* __restrict__ xyzw, int32_t* __restrict__ cellOut,
                                   double* __restrict__ velOut, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int64_t g = gid[i];
    const int32_t c = cell[i];
    if (xyzw) {
        xyzw[4 * g] = x[i]; xyzw[4 * g + 1] = y[i]; xyzw[4 * g + 2] = z[i];
        xyzw[4 * g + 3] = (c == CPF_CELL_FROZEN) ? 0.0 : 1.0;
    }
    if (cellOut) cellOut[g] = c;
    if (velOut) {
        velOut[4 * g] = vel ? vel[3 * i] : 0.0; velOut[4 * g + 1] = vel ? vel[3 * i + 1] : 0.0;
        velOut[4 * g + 2] = vel ? vel[3 * i + 2] : 0.0; velOut[4 * g + 3] = -1.0;   // particles.cu:361
    }
}
// (zFlag: may be null; set to 1 if any cell's velocity has a z component -- what decides the flat walk, cpf_walk.h)
__global__ void u3_to_u4_kernel(const double* __restrict__ u3, double4* __restrict__ u4, int64_t nCells, unsigned long long* zFlag) {
    const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (c >= nCells) return;
    const double uz = u3[3 * c + 2];
    u4[c] = make_double4(u3[3 * c], u3[3 * c + 1], uz, 0.0);
    if (zFlag != nullptr && !(uz == 0.0)) *zFlag = 1ull;          // (NaN counts as a component; every writer writes the same value)
}
// packed 256-byte cell records for all-hex meshes: [0..5] planes, [6] U, [7] six neighbour ids + pad
__global__ void build_cell_records_kernel(const double4* __restrict__ planes, const int32_t* __restrict__ nbr,
                                          const double4* __restrict__ U, double4* __restrict__ rec, int64_t nCells) {
    const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (c >= nCells) return;
    double4* r = rec + 8 * c;
#pragma unroll
    for (int s = 0; s < 6; ++s) r[s] = planes[6 * c + s];
    r[6] = U[c];
    int32_t* nb = reinterpret_cast<int32_t*>(r + 7);
#pragma unroll
    for (int s = 0; s < 6; ++s) nb[s] = nbr[6 * c + s];
    nb[6] = 0; nb[7] = 0;
}
// records of a mesh that is not all-hex (layout: cpf_walk.h "cell records").  recB[c] (may be null): index of the cell's
// SECOND record (slots 6..11 of a cell with 7..12 slots), behind the nCells first ones; -1 = none
__global__ void build_cell_records_mixed_kernel(const int32_t* __restrict__ cellOff, const double4* __restrict__ planes,
                                                const int32_t* __restrict__ nbr, const double4* __restrict__ U,
                                                const int32_t* __restrict__ recB, double4* __restrict__ rec, int64_t nCells) {
    const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (c >= nCells) return;
    const int s0 = cellOff[c], nf = cellOff[c + 1] - s0;
    const int second = recB ? recB[c] : -1;
    double4* r = rec + 8 * c;
    int32_t* nb = reinterpret_cast<int32_t*>(r + 7);
    const double4 nullPlane = make_double4(0.0, 0.0, 0.0, -1.0);
    const bool header = nf > 6 && second < 0;
    for (int s = 0; s < 6; ++s) {
        const bool real = !header && s < nf;
        r[s] = real ? planes[s0 + s] : nullPlane;
        nb[s] = real ? nbr[s0 + s] : kNullNbr;
    }
    r[6] = U[c];
    nb[6] = 0; nb[7] = 0;
    if (header) { nb[0] = kBigCellMark; nb[1] = s0; nb[2] = nf; }
    if (second >= 0) {
        nb[6] = kTwoRecMark; nb[7] = second;
        double4* r2 = rec + 8 * (int64_t)second;
        int32_t* nb2 = reinterpret_cast<int32_t*>(r2 + 7);
        for (int s = 0; s < 6; ++s) {
            const bool real = 6 + s < nf;
            r2[s] = real ? planes[s0 + 6 + s] : nullPlane;
            nb2[s] = real ? nbr[s0 + 6 + s] : kNullNbr;
        }
        r2[6] = U[c];
        nb2[6] = 0; nb2[7] = 0;
    }
}
// (box: the mesh's box records, or null -- U sits in doubles 10..12 of each, cpf_walk.h "box records")
__global__ void update_record_velocity_kernel(const double4* __restrict__ U, double4* __restrict__ rec, double* __restrict__ box,
                                              int64_t nCells) {
    const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (c >= nCells) return;
    const double4 u = U[c];
    rec[8 * c + 6] = u;
    if (box != nullptr) { box[16 * c + 10] = u.x; box[16 * c + 11] = u.y; box[16 * c + 12] = u.z; }
}
__global__ void count_negative_kernel(const int32_t* __restrict__ cell, int64_t n, unsigned long long* out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    unsigned v = (i < n && cell[i] < 0) ? 1u : 0u;
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, (unsigned long long)v);
}

static inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

hipError_t launch_iota64(hipStream_t st, int64_t* p, int64_t n, int64_t first) {
    if (n > 0) hipLaunchKernelGGL(iota64_kernel, grid_for(n), dim3(kBlock), 0, st, p, n, first);
    return hipGetLastError();
}
hipError_t launch_unpack_xyz(hipStream_t st, const double* xyz, double* x, double* y, double* z, int64_t n) {
    if (n > 0) hipLaunchKernelGGL(unpack_xyz_kernel, grid_for(n), dim3(kBlock), 0, st, xyz, x, y, z, n);
    return hipGetLastError();
}
hipError_t launch_pack_by_gid(hipStream_t st, const double* x, const double* y, const double* z, const int32_t* cell,
                              const int64_t* gid, const double* vel, double* xyzw, int32_t* cellOut, double* velOut,
                              int64_t n) {
    if (n > 0)
        hipLaunchKernelGGL(pack_by_gid_kernel, grid_for(n), dim3(kBlock), 0, st, x, y, z, cell, gid, vel, xyzw, cellOut,
                           velOut, n);
    return hipGetLastError();
}
hipError_t launch_u3_to_u4(hipStream_t st, const double* u3, double4* u4, int64_t nCells, unsigned long long* zFlag) {
    if (nCells > 0) hipLaunchKernelGGL(u3_to_u4_kernel, grid_for(nCells), dim3(kBlock), 0, st, u3, u4, nCells, zFlag);
    return hipGetLastError();
}
hipError_t launch_build_cell_records(hipStream_t st, const double4* planes, const int32_t* nbr, const double4* U,
                                     double4* rec, int64_t nCells) {
    if (nCells > 0) hipLaunchKernelGGL(build_cell_records_kernel, grid_for(nCells), dim3(kBlock), 0, st, planes, nbr, U, rec, nCells);
    return hipGetLastError();
}
hipError_t launch_build_cell_records_mixed(hipStream_t st, const int32_t* cellOff, const double4* planes, const int32_t* nbr,
                                           const double4* U, const int32_t* recB, double4* rec, int64_t nCells) {
    if (nCells > 0) hipLaunchKernelGGL(build_cell_records_mixed_kernel, grid_for(nCells), dim3(kBlock), 0, st, cellOff, planes, nbr, U, recB, rec, nCells);
    return hipGetLastError();
}
hipError_t launch_update_record_velocity(hipStream_t st, const double4* U, double4* rec, double* box, int64_t nCells) {
    if (nCells > 0) hipLaunchKernelGGL(update_record_velocity_kernel, grid_for(nCells), dim3(kBlock), 0, st, U, rec, box, nCells);
    return hipGetLastError();
}
hipError_t launch_count_negative(hipStream_t st, const int32_t* cell, int64_t n, unsigned long long* out) {
    if (n > 0) hipLaunchKernelGGL(count_negative_kernel, grid_for(n), dim3(kBlock), 0, st, cell, n, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// sort by cell: radix sort (cell, index) pairs on the low bits only, then gather every array.
// Stable and deterministic, so the order of particles is reproducible run to run.
// ------------------------------------------------------------------------------------------------
// Sort key = (cell, position inside the cell's bounding box quantised per axis): particles that share a wave then
// also share a neighbourhood INSIDE the cell, so they cross the same faces in the same round.  The bit layout
// (bits per axis, axis significance) is chosen at mesh ingest (cpf_mesh.cpp): 4 x 4 x 4 for 3-D meshes, 4 x 32 for
// a 2-D case like pitzDaily.  Measured on the bench cloud: rounds per wave 2.68 (cell only) -> 2.31 -> 2.23.
struct SubKey { int bits[3]; int order[3]; };
// (aos != nullptr: the particle's (x, y, z, id) also goes out as ONE 32-byte record, which the sort's gather then fetches with one
// L2 transaction instead of four -- see sort_by_cell)
#ifndef CPF_SORT_AOS
#define CPF_SORT_AOS 1
#endif
#ifndef CPF_SORT_XCD
#define CPF_SORT_XCD 1
#endif
struct __attribute__((aligned(16))) Pair64 { double a, b; };
__device__ __forceinline__ void put_aos(double* __restrict__ aos, int64_t i, double px, double py, double pz, const int64_t* __restrict__ gid) {
    Pair64* rec = reinterpret_cast<Pair64*>(aos) + 2 * i;
    rec[0] = Pair64{px, py};
    rec[1] = Pair64{pz, __longlong_as_double(gid ? gid[i] : 0ll)};
}
__device__ __forceinline__ uint32_t sort_key_of(double px, double py, double pz, int32_t c, const float* __restrict__ cellBox,
                                                const int32_t* __restrict__ rank, const SubKey& sk, int subBits) {
    if (c < 0) return 0xFFFFFFFFu;                          // lost / frozen particles go to the tail
    const float* b = cellBox + 6 * (int64_t)c;              // lo.xyz, 2^bits / extent.xyz
    const float r[3] = {((float)px - b[0]) * b[3], ((float)py - b[1]) * b[4], ((float)pz - b[2]) * b[5]};
    uint32_t sub = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int a = sk.order[k];
        const int q = min((1 << sk.bits[a]) - 1, max(0, (int)(a == 0 ? r[0] : (a == 1 ? r[1] : r[2]))));
        sub = (sub << sk.bits[a]) | (uint32_t)q;
    }
    // (rank: the cell's place along the mesh layer's Morton curve instead of its id -- sparse clouds, see sort_by_cell)
    return ((uint32_t)(rank ? rank[c] : c) << subBits) | sub;
}
__global__ void sort_keys_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                 const double* __restrict__ z, const int32_t* __restrict__ cell,
                                 const float* __restrict__ cellBox, const int32_t* __restrict__ rank, SubKey sk, int subBits,
                                 uint32_t* __restrict__ keys, int64_t n, const int64_t* __restrict__ gid, double* __restrict__ aos) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double px = x[i], py = y[i], pz = z[i];
    if (aos) put_aos(aos, i, px, py, pz, gid);
    keys[i] = sort_key_of(px, py, pz, cell[i], cellBox, rank, sk, subBits);
}
// the gather out of the 32-byte records: positions and id with one transaction; the cell out of the sorted key, or fetched.
// kGatherItems destinations per thread, all their record loads in flight at once: the chain permutation -> record -> store is
// two memory latencies long.  MEASURED: 1 / 2 / 4 / 8 destinations per thread: 0.58 / 0.54 / 0.53-0.54 / 0.53-0.55 ms per sort of 1e7
// (pitzDaily), 0.26-0.27 / 0.244-0.248 / 0.255 / 0.258 of 4e6 (TJunction).
// The permutation, the keys and the output arrays are streamed past the caches (non-temporal loads / stores: the L2 is for the
// records' lines): -1.4 % (pitzDaily), -2.9 % (TJunction) per sort, three alternating runs each.
#ifndef CPF_GATHER_ITEMS
#define CPF_GATHER_ITEMS 2
#endif
#ifndef CPF_GATHER_NT
#define CPF_GATHER_NT 1
#endif
constexpr int kGatherItems = CPF_GATHER_ITEMS;
__global__ __launch_bounds__(kBlock) void gather_aos_kernel(const double* __restrict__ aos, const int32_t* __restrict__ cell,
                                                            double* __restrict__ ox, double* __restrict__ oy, double* __restrict__ oz,
                                                            int32_t* __restrict__ ocell, int64_t* __restrict__ ogid,
                                                            const int32_t* __restrict__ perm, const uint32_t* __restrict__ keys, int nSub, int64_t n,
                                                            bool cellFromKey) {
    // (workgroups go to the 8 XCDs in turn: give every XCD one contiguous eighth of the destinations, so that the records of
    //  a 128-byte source line -- whose destinations are neighbours -- are fetched into ONE L2; the grid is 8 * ceil(blocks / 8))
#if CPF_SORT_XCD
    const int64_t per = gridDim.x >> 3;
    const int64_t i0 = (((int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3)) * kBlock * kGatherItems) + threadIdx.x;
#else
    const int64_t i0 = (int64_t)blockIdx.x * kBlock * kGatherItems + threadIdx.x;
#endif
    int64_t j[kGatherItems]; uint32_t k[kGatherItems];
#pragma unroll
    for (int u = 0; u < kGatherItems; ++u) {
        const int64_t i = i0 + (int64_t)u * kBlock;
#if CPF_GATHER_NT
        j[u] = i < n ? __builtin_nontemporal_load(perm + i) : 0;
        k[u] = (cellFromKey && i < n) ? __builtin_nontemporal_load(keys + i) : 0xFFFFFFFFu;
#else
        j[u] = i < n ? perm[i] : 0;
        k[u] = (cellFromKey && i < n) ? keys[i] : 0xFFFFFFFFu;
#endif
    }
    Pair64 lo[kGatherItems], hi[kGatherItems]; int32_t cc[kGatherItems];
#pragma unroll
    for (int u = 0; u < kGatherItems; ++u) {
        const Pair64* rec = reinterpret_cast<const Pair64*>(aos) + 2 * j[u];
        lo[u] = rec[0]; hi[u] = rec[1];
        cc[u] = (cellFromKey && k[u] != 0xFFFFFFFFu) ? (int32_t)(k[u] >> nSub) : cell[j[u]];
    }
#pragma unroll
    for (int u = 0; u < kGatherItems; ++u) {
        const int64_t i = i0 + (int64_t)u * kBlock;
        if (i < n) {
#if CPF_GATHER_NT
            __builtin_nontemporal_store(lo[u].a, ox + i); __builtin_nontemporal_store(lo[u].b, oy + i); __builtin_nontemporal_store(hi[u].a, oz + i);
            __builtin_nontemporal_store(cc[u], ocell + i);
            if (ogid) __builtin_nontemporal_store((int64_t)__double_as_longlong(hi[u].b), ogid + i);
#else
            ox[i] = lo[u].a; oy[i] = lo[u].b; oz[i] = hi[u].a;
            ocell[i] = cc[u];
            if (ogid) ogid[i] = __double_as_longlong(hi[u].b);
#endif
        }
    }
}
static inline dim3 grid8_for(int64_t n) { const int64_t b = (n + kBlock * kGatherItems - 1) / (kBlock * kGatherItems); return dim3((unsigned)((b + 7) / 8 * 8)); }
__global__ void copy_cells_kernel(int32_t* __restrict__ cell, const int32_t* __restrict__ sc, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) cell[i] = sc[i];
}

// ------------------------------------------------------------------------------------------------
// The key sort, hand-written (round 5; option "sort_method" 2, the default; 0 = hipcub::DeviceRadixSort): a stable LSD radix sort
// of (key, index) pairs with digits of <= 8 bits (as many passes as the library's) as chunked counting sorts.
//   * The cloud is cut into up to 1024 contiguous chunks of whole 4096-element tiles, one workgroup (4 waves) each.  Inside a tile
//     wave w owns elements [1024 w, 1024 (w + 1)) and ranks them in order against private per-wave counters in LDS: stability then
//     needs no ordering BETWEEN waves -- wave w's elements rank behind those of the waves before it.  Lanes with equal digits find
//     each other with one ballot per digit bit; rank = the counter + the peers before me; the group's first lane advances the counter.
//   * One pass over [wave][bin] turns the counters into first positions; the tile is written to LDS in digit order and copied out
//     with the bins' running global positions, so that every bin receives ONE contiguous run per tile (a first version with
//     11-bit digits and no reorder wrote 8 bytes per 64-byte line on a random digit: 0.65 ms per 1e7; docs/experiments.md).
//   * Per pass: rt_hist_kernel (per-chunk digit counts, bin-major: 16-byte loads, a tile's worth in flight per thread),
//     rt_scan_kernel (one wave per bin over the chunks), rt_scatter_kernel.  The first pass reads no index array (the index is the
//     position), the last writes no keys unless the gather wants the cell from them.
// MEASURED (profiles/r05_sort_breakdown.json; 1e7 particles on pitzDaily sorted 25 cycles of D = 1.5e-5 ago): counts 10-13 us, scan
// 5, scatter 56 per pass -- 0.54 ms per sort against 0.64 with the library's onesweep passes (88 us each); TJunction, 4e6 particles,
// 24 key bits: 0.24 against 0.27.  The scatter is bound by its own ranking arithmetic (8 ballots and ~65 vector instructions per
// element: coalesced stores instead of the scatter, or no loads at all, change its time by < 7 %); 256 x 16, 256 x 8, 512 x 8 and
// 512 x 16 threads x items tie.
// ------------------------------------------------------------------------------------------------
struct SortPlan {
    int passes, bits[4], shift[4];
    int nChunks;
    int64_t chunk;       // elements per chunk: whole tiles
};
static inline size_t sort_al(size_t b) { return (b + 255) & ~(size_t)255; }
#ifndef CPF_RT_STHREADS
#define CPF_RT_STHREADS 256
#endif
#ifndef CPF_RT_SITEMS
#define CPF_RT_SITEMS 16
#endif
constexpr int kRtThreads = 256, kRtWaves = 4, kRtMaxBits = 8, kRtMaxChunks = 1024;        // the counting and scanning kernels
constexpr int kRtSThreads = CPF_RT_STHREADS, kRtSWaves = kRtSThreads / 64, kRtItems = CPF_RT_SITEMS, kRtTile = kRtSThreads * kRtItems;   // the scatter
static_assert(kRtTile % (kRtThreads * 4) == 0, "a tile is whole batches of the counting kernel's 16-byte loads");
static SortPlan rt_plan(int64_t n, int endBit) {
    SortPlan p{};
    p.passes = (endBit + kRtMaxBits - 1) / kRtMaxBits;
    const int d = (endBit + p.passes - 1) / p.passes;
    for (int k = 0; k < p.passes; ++k) { p.shift[k] = k * d; p.bits[k] = std::min(d, endBit - k * d); }
    int64_t chunk = (n + kRtMaxChunks - 1) / kRtMaxChunks;
    chunk = std::max<int64_t>(kRtTile, (chunk + kRtTile - 1) / kRtTile * kRtTile);
    p.chunk = chunk;
    p.nChunks = (int)((n + chunk - 1) / chunk);
    return p;
}
static size_t rt_scratch_bytes(int64_t n, int endBit) {
    const SortPlan p = rt_plan(n, endBit);
    (void)p;
    return sort_al(4 * (size_t)n) * 4 + sort_al((size_t)kRtMaxChunks * (1u << kRtMaxBits) * 4) + sort_al((1u << kRtMaxBits) * 4) + sort_al(32 * (size_t)n);
}

// counts of one digit per chunk, BIN-major: counts[bin * stride + chunk].  A chunk is whole tiles of 4096 keys, the key buffers
// are 256-byte aligned: every thread has its 16-byte loads of a tile in flight at once (one memory latency per tile, not one per
// key), counts in per-wave private LDS histograms; a wave whose 64 x 4 keys share a digit (the high digits of an almost sorted
// cloud) adds them with one atomic.
constexpr int kRtHistBatch = 4;       // uint4 loads in flight per thread: 4 x 4 x 256 = 4096 keys
__global__ __launch_bounds__(kRtThreads) void rt_hist_kernel(const uint32_t* __restrict__ keys, int64_t n, int64_t chunk, int shift, int bits,
                                                             uint32_t* __restrict__ counts, int stride) {
    extern __shared__ unsigned sH[];                    // [kRtWaves][bins]
    const int bins = 1 << bits;
    const int wave = threadIdx.x >> 6;
    for (int b = threadIdx.x; b < kRtWaves * bins; b += kRtThreads) sH[b] = 0u;
    __syncthreads();
    unsigned* mine = sH + wave * bins;
    const uint32_t mask = (uint32_t)bins - 1u;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = min(n, lo + chunk);
    for (int64_t t0 = lo; t0 < hi; t0 += (int64_t)kRtHistBatch * 4 * kRtThreads) {
        uint4 v[kRtHistBatch];
#pragma unroll
        for (int u = 0; u < kRtHistBatch; ++u) {
            const int64_t i = t0 + ((int64_t)u * kRtThreads + threadIdx.x) * 4;
            if (i + 3 < hi) v[u] = *reinterpret_cast<const uint4*>(keys + i);
            else {
                v[u].x = i < hi ? keys[i] : 0u; v[u].y = i + 1 < hi ? keys[i + 1] : 0u; v[u].z = i + 2 < hi ? keys[i + 2] : 0u; v[u].w = 0u;
            }
        }
#pragma unroll
        for (int u = 0; u < kRtHistBatch; ++u) {
            const int64_t i = t0 + ((int64_t)u * kRtThreads + threadIdx.x) * 4;
            const int live = (int)max<int64_t>(0, min<int64_t>(4, hi - i));
            const uint32_t d0 = (v[u].x >> shift) & mask, d1 = (v[u].y >> shift) & mask, d2 = (v[u].z >> shift) & mask, d3 = (v[u].w >> shift) & mask;
            const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)d0);
            const bool same = live == 4 && d0 == first && d1 == first && d2 == first && d3 == first;
            if (__ballot(same) == ~0ull) {
                if ((threadIdx.x & 63) == 0) atomicAdd(&mine[first], 256u);
            } else {
                if (live > 0) atomicAdd(&mine[d0], 1u);
                if (live > 1) atomicAdd(&mine[d1], 1u);
                if (live > 2) atomicAdd(&mine[d2], 1u);
                if (live > 3) atomicAdd(&mine[d3], 1u);
            }
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < bins; b += kRtThreads) {
        unsigned c = 0;
#pragma unroll
        for (int w = 0; w < kRtWaves; ++w) c += sH[w * bins + b];
        counts[(int64_t)b * stride + blockIdx.x] = c;
    }
}
// counts[bin][chunk] -> the bin's elements in the chunks BEFORE this one (in place); totals[bin] = the bin's size.  One wave per
// bin: a lane sums its run of consecutive chunks, one shuffle scan across the lanes, the run is written back.
__global__ __launch_bounds__(kRtThreads) void rt_scan_kernel(uint32_t* __restrict__ counts, int nChunks, int stride, int bins,
                                                             uint32_t* __restrict__ totals) {
    const int bin = blockIdx.x * kRtWaves + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (bin >= bins) return;
    constexpr int PER = kRtMaxChunks / 64;               // 16
    uint32_t* row = counts + (int64_t)bin * stride;
    uint32_t v[PER]; uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) { const int c = lane * PER + k; v[k] = c < nChunks ? row[c] : 0u; sum += v[k]; }
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t up = __shfl_up(incl, off, 64); if (lane >= off) incl += up; }
    uint32_t run = incl - sum;
#pragma unroll
    for (int k = 0; k < PER; ++k) { const int c = lane * PER + k; if (c < nChunks) row[c] = run; run += v[k]; }
    if (lane == 63) totals[bin] = incl;
}

template <int BITS>
__global__ __launch_bounds__(kRtSThreads) void rt_scatter_kernel(const uint32_t* __restrict__ keysIn, const int32_t* __restrict__ idxIn,
                                                                uint32_t* __restrict__ keysOut, int32_t* __restrict__ idxOut, int64_t n,
                                                                int64_t chunk, int shift, const uint32_t* __restrict__ before, int stride,
                                                                const uint32_t* __restrict__ totals) {
    constexpr int BINS = 1 << BITS;
    __shared__ uint32_t sKey[kRtTile];
    __shared__ int32_t sIdx[kRtTile];
    __shared__ unsigned sCnt[kRtSWaves][BINS];          // per tile: the waves' digit counts, then their first positions in the tile
    __shared__ unsigned sBinStart[BINS + 1];           // per tile: first position of a bin inside the tile
    __shared__ unsigned sGlobal[BINS];                 // running: where the bin's next element of this chunk goes
    __shared__ unsigned sScan[kRtSThreads];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t mask = (uint32_t)BINS - 1u;
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    // bin starts of the whole array = exclusive scan of the totals (BINS <= 256 == kRtThreads), + what the chunks before this one hold
    {
        const unsigned tot = threadIdx.x < BINS ? totals[threadIdx.x] : 0u;
        unsigned incl = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const unsigned up = __shfl_up(incl, off, 64); if (lane >= off) incl += up; }
        if (lane == 63) sScan[wave] = incl;
        __syncthreads();
        unsigned base = 0;
        for (int w = 0; w < wave; ++w) base += sScan[w];
        if (threadIdx.x < BINS) sGlobal[threadIdx.x] = base + incl - tot + before[(int64_t)threadIdx.x * stride + blockIdx.x];
        __syncthreads();                                   // (sScan is used again by the first tile)
    }
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = min(n, lo + chunk);
    for (int64_t t0 = lo; t0 < hi; t0 += kRtTile) {
        const int tileN = (int)min<int64_t>(kRtTile, hi - t0);
        for (int b = lane; b < BINS; b += 64) sCnt[wave][b] = 0u;
        __syncthreads();                                   // (also: the previous tile's copy-out has left sKey / sIdx / sBinStart)
        // ---- load the wave's 1024 elements, rank them in order against the wave's counters
        uint32_t k[kRtItems]; int32_t v[kRtItems]; unsigned rnk[kRtItems];
        const int sub = wave * (kRtTile / kRtSWaves);
#pragma unroll
        for (int s = 0; s < kRtItems; ++s) {
            const int j = sub + s * 64 + lane;
            const int64_t i = t0 + j;
            k[s] = j < tileN ? keysIn[i] : 0xFFFFFFFFu;
            v[s] = j < tileN ? (idxIn ? idxIn[i] : (int32_t)i) : 0;
        }
#pragma unroll
        for (int s = 0; s < kRtItems; ++s) {
            const int j = sub + s * 64 + lane;
            const bool live = j < tileN;
            const uint32_t d = (k[s] >> shift) & mask;
            unsigned long long same = __ballot(live);
#pragma unroll
            for (int b = 0; b < BITS; ++b) {
                const unsigned long long one = __ballot(((d >> b) & 1u) != 0u);
                same &= ((d >> b) & 1u) ? one : ~one;
            }
            if (live) {
                const unsigned first = sCnt[wave][d];           // (every lane of the group reads before its first lane writes)
                rnk[s] = first + (unsigned)__popcll(same & lt);
                if ((same & lt) == 0ull) sCnt[wave][d] = first + (unsigned)__popcll(same);
            } else rnk[s] = 0;
        }
        __syncthreads();
        // ---- counts -> the waves' first positions inside each bin; bins' first positions inside the tile
        unsigned binCount = 0;
        if (threadIdx.x < BINS) {
            unsigned run = 0;
#pragma unroll
            for (int w = 0; w < kRtSWaves; ++w) { const unsigned c = sCnt[w][threadIdx.x]; sCnt[w][threadIdx.x] = run; run += c; }
            binCount = run;
        }
        {   // exclusive scan of binCount over the bins (BINS <= 256 == kRtThreads)
            unsigned incl = binCount;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const unsigned up = __shfl_up(incl, off, 64); if (lane >= off) incl += up; }
            if (lane == 63) sScan[wave] = incl;
            __syncthreads();
            unsigned base = 0;
            for (int w = 0; w < wave; ++w) base += sScan[w];
            if (threadIdx.x < BINS) sBinStart[threadIdx.x] = base + incl - binCount;
        }
        __syncthreads();
        // ---- the tile in digit order in LDS
#pragma unroll
        for (int s = 0; s < kRtItems; ++s) {
            const int j = sub + s * 64 + lane;
            if (j < tileN) {
                const uint32_t d = (k[s] >> shift) & mask;
                const unsigned pos = sBinStart[d] + sCnt[wave][d] + rnk[s];
                sKey[pos] = k[s]; sIdx[pos] = v[s];
            }
        }
        __syncthreads();
        // ---- out: element j of the ordered tile is the (j - start of its bin)-th of its bin in this tile
        for (int j = threadIdx.x; j < tileN; j += kRtSThreads) {
            const uint32_t kk = sKey[j];
            const uint32_t d = (kk >> shift) & mask;
            const unsigned g = sGlobal[d] + ((unsigned)j - sBinStart[d]);
            if (keysOut) keysOut[g] = kk;
            idxOut[g] = sIdx[j];
        }
        __syncthreads();
        if (threadIdx.x < BINS) sGlobal[threadIdx.x] += binCount;
    }
}

static hipError_t rt_sort_pairs(hipStream_t st, const double* x, const double* y, const double* z, const int32_t* cell, const int64_t* gid, double* aos,
                                const float* cellBox, const int32_t* rank, const SubKey& sk, int nSub, int64_t n, int endBit,
                                char* scratch, bool keepKeys, const uint32_t** keysSorted, const int32_t** perm) {
    const SortPlan p = rt_plan(n, endBit);
    uint32_t* kA = (uint32_t*)scratch; scratch += sort_al(4 * (size_t)n);
    uint32_t* kB = (uint32_t*)scratch; scratch += sort_al(4 * (size_t)n);
    int32_t* iA = (int32_t*)scratch; scratch += sort_al(4 * (size_t)n);
    int32_t* iB = (int32_t*)scratch; scratch += sort_al(4 * (size_t)n);
    uint32_t* counts = (uint32_t*)scratch; scratch += sort_al((size_t)kRtMaxChunks * (1u << kRtMaxBits) * 4);
    uint32_t* totals = (uint32_t*)scratch;
    const int stride = kRtMaxChunks;
    hipLaunchKernelGGL(sort_keys_kernel, grid_for(n), dim3(kBlock), 0, st, x, y, z, cell, cellBox, rank, sk, nSub, kA, n, gid, aos);
    const uint32_t* kin = kA; const int32_t* iin = nullptr;
    for (int k = 0; k < p.passes; ++k) {
        const int bins = 1 << p.bits[k];
        hipLaunchKernelGGL(rt_hist_kernel, dim3(p.nChunks), dim3(kRtThreads), (size_t)kRtWaves * bins * 4, st, kin, n, p.chunk, p.shift[k], p.bits[k],
                           counts, stride);
        hipLaunchKernelGGL(rt_scan_kernel, dim3((bins + kRtWaves - 1) / kRtWaves), dim3(kRtThreads), 0, st, counts, p.nChunks, stride, bins, totals);
        const bool last = k == p.passes - 1;
        uint32_t* kout = (kin == kA) ? kB : kA;
        int32_t* iout = (iin == iA) ? iB : iA;
        uint32_t* ko = (last && !keepKeys) ? nullptr : kout;
#define CPF_RT_LAUNCH(B) hipLaunchKernelGGL(rt_scatter_kernel<B>, dim3(p.nChunks), dim3(kRtSThreads), 0, st, kin, iin, ko, iout, n, p.chunk, p.shift[k], counts, stride, totals)
        switch (p.bits[k]) {
            case 1: CPF_RT_LAUNCH(1); break; case 2: CPF_RT_LAUNCH(2); break; case 3: CPF_RT_LAUNCH(3); break; case 4: CPF_RT_LAUNCH(4); break;
            case 5: CPF_RT_LAUNCH(5); break; case 6: CPF_RT_LAUNCH(6); break; case 7: CPF_RT_LAUNCH(7); break; default: CPF_RT_LAUNCH(8); break;
        }
#undef CPF_RT_LAUNCH
        kin = kout; iin = iout;
    }
    *keysSorted = keepKeys ? kin : nullptr;
    *perm = iin;
    return hipGetLastError();
}

size_t sort_scratch_bytes(int64_t n, int endBit) {
    size_t tmp = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                       (const int32_t*)nullptr, (int32_t*)nullptr, (int)n, 0, endBit);
    // keys in + keys out + iota + perm + one staging array (32 bytes per particle: the (x, y, z, id) records; also serves the
    // velocity triples)
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return std::max(al(tmp) + al(4 * (size_t)n) * 4 + al(32 * (size_t)n), rt_scratch_bytes(n, endBit));
}

// Out arrays (ox ... ogid) given: the sorted cloud is written there and the input arrays are left alone -- no staging,
// no copy back (a fifth of the sort's time); the caller swaps its buffers.  All null: in place.
// how many cells hold particles: the runs of equal cell ids in the sorted keys (lost / frozen particles -- the all-ones key --
// do not count).  out[0] += runs, out[1] += live particles
__global__ __launch_bounds__(kBlock) void count_cell_runs_kernel(const uint32_t* __restrict__ keys, int64_t n, int nSub, uint32_t lostKey,
                                                                 unsigned long long* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    bool live = false, start = false;
    if (i < n) {
        const uint32_t k = keys[i];
        live = k != lostKey;
        start = live && (i == 0 || (keys[i - 1] >> nSub) != (k >> nSub));
    }
    const unsigned runs = (unsigned)__popcll(__builtin_amdgcn_ballot_w64(start)), alive = (unsigned)__popcll(__builtin_amdgcn_ballot_w64(live));
    __shared__ unsigned sR, sA;
    if (threadIdx.x == 0) { sR = 0; sA = 0; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { atomicAdd(&sR, runs); atomicAdd(&sA, alive); }
    __syncthreads();
    if (threadIdx.x == 0 && (sR | sA)) { atomicAdd(&out[0], (unsigned long long)sR); atomicAdd(&out[1], (unsigned long long)sA); }
}

hipError_t sort_by_cell(hipStream_t st, double* x, double* y, double* z, int32_t* cell, int64_t* gid, double* vel3,
                        int64_t n, int endBit, const float* cellBox, const int* subBits, const int* subOrder,
                        void* scratch, size_t scratchBytes, double* ox, double* oy, double* oz, int32_t* ocell,
                        int64_t* ogid, unsigned long long* occupied, const int32_t* rank, int method) {
    if (n <= 1) return hipSuccess;
    SubKey sk;
    for (int k = 0; k < 3; ++k) { sk.bits[k] = subBits[k]; sk.order[k] = subOrder[k]; }
    const int nSub = subBits[0] + subBits[1] + subBits[2];
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    if (method != 0) {
        // ---- the hand-written key sort (rt_sort_pairs); the same gathers as the library path below
        const size_t need = rt_scratch_bytes(n, endBit);
        if (need > scratchBytes) return hipErrorInvalidValue;
        double* stage = (double*)((char*)scratch + need - sort_al(32 * (size_t)n));
        double* aos = CPF_SORT_AOS ? stage : nullptr;
        const bool cellFromKey = rank == nullptr && endBit < 32;
        const bool keepKeys = occupied != nullptr || (aos && cellFromKey);
        const uint32_t* keysSorted = nullptr; const int32_t* perm = nullptr;
        hipError_t e = rt_sort_pairs(st, x, y, z, cell, gid, aos, cellBox, rank, sk, nSub, n, endBit, (char*)scratch, keepKeys, &keysSorted, &perm);
        if (e != hipSuccess) return e;
        if (occupied != nullptr) {
            e = hipMemsetAsync(occupied, 0, 16, st);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(count_cell_runs_kernel, grid_for(n), dim3(kBlock), 0, st, keysSorted, n, nSub, 0xFFFFFFFFu, occupied);
        }
        if (ox != nullptr) {
            if (aos)
                hipLaunchKernelGGL(gather_aos_kernel, grid8_for(n), dim3(kBlock), 0, st, aos, cell, ox, oy, oz, ocell, gid ? ogid : nullptr, perm,
                                   keysSorted, nSub, n, cellFromKey);
            else
                hipLaunchKernelGGL(gather_all_kernel, grid_for(n), dim3(kBlock), 0, st, x, y, z, cell, gid, ox, oy, oz, ocell, ogid, perm,
                                   (const uint32_t*)nullptr, nSub, n, false);
        } else if (aos) {
            // (the index buffer the last pass did not write is free)
            int32_t* iA = (int32_t*)((char*)scratch + 2 * sort_al(4 * (size_t)n));
            int32_t* iB = (int32_t*)((char*)iA + sort_al(4 * (size_t)n));
            int32_t* spare = perm == iA ? iB : iA;
            hipLaunchKernelGGL(gather_aos_kernel, grid8_for(n), dim3(kBlock), 0, st, aos, cell, x, y, z, spare, gid, perm, keysSorted, nSub, n, cellFromKey);
            hipLaunchKernelGGL(copy_cells_kernel, grid_for(n), dim3(kBlock), 0, st, cell, spare, n);
        } else {
            double* sx = stage; double* sy = stage + n; double* sz = stage + 2 * n;
            hipLaunchKernelGGL(gather_xyz_kernel, grid_for(n), dim3(kBlock), 0, st, x, y, z, sx, sy, sz, perm, n);
            hipLaunchKernelGGL(copy_xyz_kernel, grid_for(n), dim3(kBlock), 0, st, x, y, z, sx, sy, sz, n);
            int64_t* sg = reinterpret_cast<int64_t*>(stage);
            int32_t* sc = reinterpret_cast<int32_t*>(stage + n);
            hipLaunchKernelGGL(gather_ids_kernel, grid_for(n), dim3(kBlock), 0, st, cell, gid, sc, sg, perm, n);
            hipLaunchKernelGGL(copy_ids_kernel, grid_for(n), dim3(kBlock), 0, st, cell, gid, sc, sg, n);
        }
        if (vel3) {
            hipLaunchKernelGGL(gather3_kernel, grid_for(n), dim3(kBlock), 0, st, vel3, stage, perm, n);
            e = hipMemcpyAsync(vel3, stage, 24 * (size_t)n, hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) return e;
        }
        return hipGetLastError();
    }
    size_t tmpBytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tmpBytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                       (const int32_t*)nullptr, (int32_t*)nullptr, (int)n, 0, endBit);
    char* p = (char*)scratch;
    void* tmp = p; p += al(tmpBytes);
    uint32_t* keysIn = (uint32_t*)p; p += al(4 * (size_t)n);
    uint32_t* keysOut = (uint32_t*)p; p += al(4 * (size_t)n);
    int32_t* idx = (int32_t*)p; p += al(4 * (size_t)n);
    int32_t* perm = (int32_t*)p; p += al(4 * (size_t)n);
    double* stage = (double*)p; p += al(32 * (size_t)n);
    if ((size_t)(p - (char*)scratch) > scratchBytes) return hipErrorInvalidValue;
    double* aos = CPF_SORT_AOS ? stage : nullptr;
    hipLaunchKernelGGL(iota_kernel, grid_for(n), dim3(kBlock), 0, st, idx, n);
    hipLaunchKernelGGL(sort_keys_kernel, grid_for(n), dim3(kBlock), 0, st, x, y, z, cell, cellBox, rank, sk, nSub, keysIn, n, gid, aos);
    // endBit covers the cell bits + sub-cell bits; the all-ones key of lost/frozen particles sorts last
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(tmp, tmpBytes, keysIn, keysOut, idx, perm, (int)n, 0, endBit, st);
    if (e != hipSuccess) return e;
    if (occupied != nullptr) {
        e = hipMemsetAsync(occupied, 0, 16, st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(count_cell_runs_kernel, grid_for(n), dim3(kBlock), 0, st, keysOut, n, nSub, 0xFFFFFFFFu, occupied);    // (sort_keys_kernel's lost key)
    }
    // (a live particle's cell out of its sorted key: only where no cell bit was shifted out of the 32-bit key)
    const bool cellFromKey = rank == nullptr && endBit < 32;
    if (ox != nullptr) {
        if (aos)
            hipLaunchKernelGGL(gather_aos_kernel, grid8_for(n), dim3(kBlock), 0, st, aos, cell, ox, oy, oz, ocell, gid ? ogid : nullptr, perm, keysOut,
                               nSub, n, cellFromKey);
        else
            hipLaunchKernelGGL(gather_all_kernel, grid_for(n), dim3(kBlock), 0, st, x, y, z, cell, gid, ox, oy, oz, ocell, ogid, perm, keysOut, nSub, n,
                               cellFromKey);
    } else if (aos) {
        // in place: the records ARE the copy -- positions and ids go straight back; the cells pass through the spent index array
        hipLaunchKernelGGL(gather_aos_kernel, grid8_for(n), dim3(kBlock), 0, st, aos, cell, x, y, z, idx, gid, perm, keysOut, nSub, n, cellFromKey);
        hipLaunchKernelGGL(copy_cells_kernel, grid_for(n), dim3(kBlock), 0, st, cell, idx, n);
    } else {
        double* sx = stage; double* sy = stage + n; double* sz = stage + 2 * n;          // the 24n-byte staging area
        hipLaunchKernelGGL(gather_xyz_kernel, grid_for(n), dim3(kBlock), 0, st, x, y, z, sx, sy, sz, perm, n);
        hipLaunchKernelGGL(copy_xyz_kernel, grid_for(n), dim3(kBlock), 0, st, x, y, z, sx, sy, sz, n);
        int64_t* sg = reinterpret_cast<int64_t*>(stage);                                  // 8n, then 4n for the cells
        int32_t* sc = reinterpret_cast<int32_t*>(stage + n);
        hipLaunchKernelGGL(gather_ids_kernel, grid_for(n), dim3(kBlock), 0, st, cell, gid, sc, sg, perm, n);
        hipLaunchKernelGGL(copy_ids_kernel, grid_for(n), dim3(kBlock), 0, st, cell, gid, sc, sg, n);
    }
    if (vel3) {
        hipLaunchKernelGGL(gather3_kernel, grid_for(n), dim3(kBlock), 0, st, vel3, stage, perm, n);
        e = hipMemcpyAsync(vel3, stage, 24 * (size_t)n, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

}  // namespace cpf
