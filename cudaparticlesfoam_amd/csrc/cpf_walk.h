// Device-side helpers shared by the step kernels (cpf_kernels.hip, cpf_stream.hip): the plane-exit test of one
// cell visit in its three data-source flavours, the Philox/Box-Muller deviates and the statistics reduction.
// Arithmetic contract: see the head of cpf_kernels.hip (explicit fma() only, -ffp-contract=off, IEEE division).
#pragma once
#include "cpf_device.h"

namespace cpf {

// Wave-wide vote on a bool.  HIP's ballot64(int) first turns the predicate into 0/1 in a VGPR and compares it
// again (two VALU instructions per vote on a kernel that is VALU-issue bound); the builtin takes the i1 as is.
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
struct D3 { double x, y, z; };

__device__ __forceinline__ double dot3(const double4& n, const D3& v) {
    return fma(n.z, v.z, fma(n.y, v.y, n.x * v.x));
}
__device__ __forceinline__ double dot3(const D3& n, const D3& v) {
    return fma(n.z, v.z, fma(n.y, v.y, n.x * v.x));
}
// signed plane distance (Cf - P).n = d - n.P as one fma chain (<= 0 on the inner side of the face)
__device__ __forceinline__ double plane_dist(const double4& pl, const D3& P) {
    return fma(-pl.z, P.z, fma(-pl.y, P.y, fma(-pl.x, P.x, pl.w)));
}
__device__ __forceinline__ D3 axpy(double s, const D3& a, const D3& b) {
    return {fma(s, a.x, b.x), fma(s, a.y, b.y), fma(s, a.z, b.z)};
}

constexpr double kTol = 1e-13;     // query/ConvexQuery.cu:42
constexpr int kMaxHops = 50;       // query/ConvexQuery.cu:169
constexpr int kMaxReflect = 5;     // query/ConvexQuery.cu:353

// FACE GROUPS (cpf_mesh.cpp): the coplanar faces of a cell -- the pieces of a face split by a 2:1 refinement next door --
// share ONE slot, whose neighbour code names the group.  Two additions to the reference's rule, only where a slot is a
// group (no reference semantics exist for such cells, src/initCuda.H:64: hexes only; stated independently in
// oracle/cellwalk.c):
//   1. OUTWARD CROSSINGS ONLY (den < 0).  A particle that came in through one piece sits on the group's plane, a rounding
//      error outside it (fd = +4e-16), moving inward: the reference's acceptance test takes that for an exit at
//      dT ~ 2e-13 > tol, and the token cannot skip the slot (it names the piece's cell, not the group).  A convex cell
//      is left against the face's inward normal, so den < 0 loses no real exit.
//   2. the cell entered is chosen at the exit point X: the piece whose CELL holds X best -- the smallest maximum, over
//      that cell's slots, of X's signed plane distance; the first piece on equal scores (resolve_group; a rare path:
//      per-lane reads of the CSR tables).
__device__ __forceinline__ bool is_group(int nb) { return nb < -(1 << 30); }
__device__ __forceinline__ int resolve_group(int code, const D3& X, const int32_t* __restrict__ cellOff, const double4* __restrict__ planes,
                                             const int32_t* __restrict__ groupOff, const int32_t* __restrict__ groupNbr) {
    const int g = code - kGroupBase;
    const int k0 = groupOff[g], k1 = groupOff[g + 1];
    double bestScore = 1e301;
    int pick = groupNbr[k0];
    for (int k = k0; k < k1; ++k) {
        const int nb = groupNbr[k];
        double score = -1e300;
        const int q0 = cellOff[nb], q1 = cellOff[nb + 1];
        for (int q = q0; q < q1; ++q) {
            const double d = plane_dist(planes[q], X);
            if (d > score) score = d;
        }
        if (score < bestScore) { bestScore = score; pick = nb; }
    }
    return pick;
}

// THIN z-LAYERED MESHES AND THE BROWNIAN KICK (MeshView::zThin: every cell is a hex whose slots 4, 5 are boundary faces with
// normals exactly along z -- a 2-D case extruded one cell thick, both tutorials' pitzDaily).  With the kick 5-10 % of the
// particles cross the front or back plane per cycle; the reference walks each of them to the plane, mirrors the end point
// and walks on (ConvexQuery.cu:286-309).  The side faces of such a mesh have nz == 0 exactly, so the cells a segment
// crosses depend on its x-y projection alone, and mirroring about a z plane leaves that projection alone: the end point
// can be mirrored about the planes BEFORE the walk -- same cells, same final point (E' = E - 2 (n.E - d) n, the
// reference's own formula, applied to the end point instead of at the hit) -- and the walk never meets the planes.  In exact
// arithmetic the two are the same trajectory; in floating point the final z differs by the rounding of the hit point the
// reference passes through (parity with the kick is statistical by contract, SURVEY.md 8c: the generator differs).  Only
// with the kick and reflecting walls; D = 0 runs keep the reference's order of operations bit for bit.
// Returns the number of mirrorings (statistics; an odd number flips the stored velocity's z).
// `clear`: the mirrored end point lies between the planes with a margin of 1e-9 of the slab's thickness.  Then NO visit of
// this particle's walk can accept a z face, in the reference's own arithmetic: from inside, fd = z_wall - S.z and den =
// fl(E.z - S.z) have equal signs only if the segment heads for that plane, and then |den| <= (|S.z - z_wall| - margin)(1 +
// 2^-53) < |fd|, so fl(fd / den) > 1 and the face is rejected (cpf_walk.h, trace_fixed: the <= 1 test is exact); the
// planes of all cells of such a mesh agree to 1e-12 of the thickness (cpf_mesh.cpp), far inside the margin, and a
// reflection off a side wall (nz == 0 exactly) leaves E.z alone.  A wave whose lanes are all clear may therefore leave both
// z faces out of its rounds -- bit-identical to testing them.
__device__ __forceinline__ int fold_z(double& ez, const double4& pa, const double4& pb, bool& clear) {
    const double za = pa.w * pa.z, zb = pb.w * pb.z;          // plane (0, 0, nz, d), nz = +-1: z = d / nz = d * nz, exactly
    const double lo = fmin(za, zb), hi = fmax(za, zb);
    int nb = 0;
#pragma unroll 1
    for (int k = 0; k < kMaxReflect && ballot64(ez < lo || ez > hi) != 0ull; ++k) {
        if (ez < lo) { ez = fma(-2.0, ez - lo, ez); ++nb; }
        else if (ez > hi) { ez = fma(-2.0, ez - hi, ez); ++nb; }
    }
    const double margin = 1e-9 * (hi - lo);
    clear = ez > lo + margin && ez < hi - margin;
    return nb;
}

// One cell of the walk: traceIntet (query/ConvexQuery.cu:32-131) on a polyhedral cell.
// Exit through the face slot with the smallest admissible dT in (tol, 1]; the slot we came in
// through (nbr == token) is skipped.  Returns the next cell (== cur: segment ends here; < 0:
// boundary code) and advances S to the exit point.
__device__ __forceinline__ int trace_in_cell(D3& S, const D3& E, int cur, const MeshView& m, int token,
                                             int& outSlot) {
    const D3 P0 = S;
    const D3 Pd = {E.x - P0.x, E.y - P0.y, E.z - P0.z};
    int next = cur, best = -1;
    double dTmin = 1.1;
    const int s0 = m.cellOff[cur], s1 = m.cellOff[cur + 1];
    for (int s = s0; s < s1; ++s) {
        const double4 pl = m.planes[s];
        const double fd = plane_dist(pl, P0);           // (Cf - P0).n  (<= 0 inside)
        const double den = dot3(pl, Pd);
        double dT = fd / den;
        if (__builtin_isinf(dT)) dT = -1.0;             // segment parallel to the face
        const int nb = m.nbr[s];
        if (nb == token) continue;
        if (is_group(nb) && !(den < 0.0)) continue;     // face groups: outward crossings only (see above)
        if (fd < kTol && dT > kTol && dT <= 1.0 && dT < dTmin) {
            dTmin = dT;
            next = nb;
            S = axpy(dT, Pd, P0);
            best = s;
        }
    }
    if (best >= 0) {
        if (is_group(next)) next = resolve_group(next, S, m.cellOff, m.planes, m.groupOff, m.groupNbr);
        outSlot = best;
    }
    return next;
}

// Same test for meshes whose cells all have NF faces (every mesh the reference can run is all-hex,
// src/initCuda.H:64): slot s of cell c is 6c+s, loads are issued back to back, and the IEEE
// division is only executed for faces that can still be accepted.  The pre-filter is EXACT, not
// approximate: with fd and den of equal sign, fl(fd/den) <= 1  <=>  |fd| <= |den| (1 is
// representable and rounding is monotone; |fd| > |den| gives a quotient >= 1 + 2^-52), a zero or
// opposite-sign pair can never give dT > tol, and den == 0 / NaN fall out of every comparison
// exactly like the isinf -> -1 substitution of ConvexQuery.cu:89.
// `pl`/`nb` may be wave-uniform pointers (scalar loads, one fetch per wave) or per-lane ones.
// GROUPS: the mesh may hold face groups -- such a slot is only left with den < 0 (see above)
// AHEAD: request that many planes (and all neighbour ids) before the first test and keep that many in flight -- per-lane
// gathers from L2 / HBM, where six dependent round trips are the cost of the visit
template <int NF, bool SKIP_ZERO_DEN, bool GROUPS = false, int AHEAD = 0>
__device__ __forceinline__ int trace_fixed(D3& S, const D3& E, int cur, const double4* __restrict__ pl,
                                           const int32_t* __restrict__ nb, int token, int& outSlot, int slotBase) {
    const D3 P0 = S;
    const D3 Pd = {E.x - P0.x, E.y - P0.y, E.z - P0.z};
    int next = cur, best = -1;
    double dTmin = 1.1;
    double4 ahead[AHEAD > 0 ? AHEAD : 1];
    int nbAll[AHEAD > 0 ? NF : 1];
    if (AHEAD > 0) {
#pragma unroll
        for (int s = 0; s < AHEAD; ++s) ahead[s] = pl[s];
#pragma unroll
        for (int s = 0; s < NF; ++s) nbAll[s] = nb[s];
#pragma unroll
        for (int s = 0; s < AHEAD; ++s) asm volatile("" : "+v"(ahead[s].x), "+v"(ahead[s].y), "+v"(ahead[s].z), "+v"(ahead[s].w));
    }
#pragma unroll
    for (int s = 0; s < NF; ++s) {
        double4 p;
        if (AHEAD > 0) {
            constexpr int A = AHEAD > 0 ? AHEAD : 1;
            p = ahead[s % A];
            if (s + AHEAD < NF) ahead[s % A] = pl[s + AHEAD];
        } else p = pl[s];
        const double den = dot3(p, Pd);
        // no lane of the wave moves across this face's plane (den == +-0 exactly, e.g. the front/back
        // faces of a one-cell-thick mesh, or wall-parallel faces in aligned flow): dT would be +-inf
        // (-> -1) or NaN, never accepted (ConvexQuery.cu:86-95), so the face costs nothing more
        // (only worth a branch when the planes are already on chip: it would serialise global gathers)
        if (SKIP_ZERO_DEN && ballot64(den != 0.0) == 0ull) continue;
        const int bs = AHEAD > 0 ? nbAll[s] : nb[s];
        const double fd = plane_dist(p, P0);
        // |fd| <= |den| (one compare with abs modifiers) and equal sign bits (integer test); zeros and
        // NaNs that slip through give dT = 0 / NaN and fail dT > tol below, as in the reference
        const bool c1 = fabs(fd) <= fabs(den), c2 = (__double2hiint(fd) ^ __double2hiint(den)) >= 0;
        const bool c3 = fd < kTol, c4 = bs != token && !(GROUPS && is_group(bs) && !(den < 0.0));
        const bool cand = c1 && c2 && c3 && c4;
        // wave-uniform skip: hipcc would otherwise if-convert and run the ~12-instruction IEEE division for
        // every face of every lane; most faces have no candidate lane at all.  The vote is the AND of the four
        // compare masks (scalar ALU); voting on `cand` itself makes the compiler rebuild it as 0/1 in a VGPR.
        if ((ballot64(c1) & ballot64(c2) & ballot64(c3) & __builtin_amdgcn_uicmp((unsigned)bs, (unsigned)token, 33 /* ne */)) != 0ull) {
            if (cand) {
                const double dT = fd / den;
                if (dT > kTol && dT < dTmin) { dTmin = dT; next = bs; best = s; }
            }
        }
    }
    if (best >= 0) {                   // exit point of the LAST accepted face == the smallest dT
        S = axpy(dTmin, Pd, P0);
        outSlot = slotBase + best;
    }
    return next;
}

// The same six-face test for a hex record sitting in LDS (wave-cooperative kernel).  Identical arithmetic and
// acceptance order; what differs is the SCHEDULE: the planes are fetched two at a time in straight-line code
// before the wave-uniform skips, so a round pays three LDS round trips instead of six (the per-face branches keep
// the compiler from hoisting the reads itself, and this kernel is bound by the length of each wave's dependent
// chain, not by instruction issue).
// ZERO_SKIP: try the zero-denominator skip at all -- pointless (a compare and a branch per face) once every particle
// has a displacement along every axis, i.e. with the Brownian kick; results are the same with or without it.
template <bool ZERO_SKIP, bool GROUPS = false>
__device__ __forceinline__ void face_test(const double4& p, int bs, const D3& P0, const D3& Pd, int token, int s,
                                          double& dTmin, int& next, int& best) {
    const double den = dot3(p, Pd);
    if (ZERO_SKIP && ballot64(den != 0.0) == 0ull) return;          // see trace_fixed: nobody crosses this plane
    const double fd = plane_dist(p, P0);
    // Cheap wave-uniform skip, exact test only where it can matter.  A lane can be accepted only if |fd| <= |den| with
    // equal signs and fd < tol: from inside (fd < 0) that means den <= fd; every other acceptable lane has fd >= 0
    // (on or outside the plane: the face it came in through, or an entry point rounded across a neighbouring face).
    // (den <= fd) | (fd >= 0) is therefore a superset of the candidates -- two compares per face instead of five, two
    // scalar operations instead of five -- and the exact predicate below decides inside the branch.
    if (((ballot64(den <= fd) | ballot64(fd >= 0.0)) & __builtin_amdgcn_uicmp((unsigned)bs, (unsigned)token, 33 /* ne */)) != 0ull) {
        // c2 only prunes divisions (a face the lane moves away from): "den < 0 or fd >= 0" holds whenever the exact
        // condition "equal sign bits" can still lead to an accepted face, and whatever else slips through has a
        // quotient <= 0 and fails dT > tol below, exactly as in the reference
        // (GROUPS: a face-group slot is only left with den < 0 -- see "face groups" above)
        const bool c1 = fabs(fd) <= fabs(den), c2 = den < 0.0 || (fd >= 0.0 && !(GROUPS && is_group(bs)));
        const bool c3 = fd < kTol, c4 = bs != token;
        if (c1 && c2 && c3 && c4) {
            const double dT = fd / den;
            if (dT > kTol && dT < dTmin) { dTmin = dT; next = bs; best = s; }
        }
    }
}

// The plane offsets are requested together with the normals (the empty asm pins them): left alone, the compiler sinks
// their LDS reads behind the first wave-uniform skip of each pair -- one more LDS round trip on the wave's dependent
// chain per pair.  Measured at steady clocks: 1-2 % on the 3-D meshes (no face is skipped there before its offset is
// needed), nothing either way on pitzDaily.
#define CPF_PIN_W(a, b) asm volatile("" : "+v"(a.w), "+v"(b.w));
// zLast (wave-uniform; MeshView::zPairLast): slots 4 and 5 are the cell's two faces with an exactly z-parallel normal.
// When no active lane moves in z (Pd.z == +-0 exactly: 2-D flow on a z-extruded mesh, no diffusion) both denominators
// are exactly +-0 for every lane -- nx == ny == 0 leaves den = nz * 0 -- so neither face can be accepted
// (ConvexQuery.cu:86-95) and ONE test replaces two plane fetches, two denominators and two votes.  Exact, not
// approximate: it is the zero-denominator skip of face_test, decided for the pair up front.
// zNever (wave-uniform): no active lane can leave through a z face at all (the Brownian kick on a one-cell-thick mesh with
// every end point mirrored clear of the planes: fold_z) -- the pair is left out whatever Pd.z is.
template <bool ZERO_SKIP = true, bool GROUPS = false>
__device__ __forceinline__ int trace_lds6(D3& S, const D3& E, int cur, const double4* rec, int token, int& outSlot, bool zLast = false,
                                          bool zNever = false) {
    const D3 P0 = S;
    const D3 Pd = {E.x - P0.x, E.y - P0.y, E.z - P0.z};
    int next = cur, best = -1;
    double dTmin = 2.0;                 // any start value > 1 is equivalent (candidates have dT <= 1); 2.0 is an inline constant
    const int2* nb = reinterpret_cast<const int2*>(rec + 7);
    {
        double4 p0 = rec[0], p1 = rec[1];
        const int2 b = nb[0];
        CPF_PIN_W(p0, p1)
        face_test<ZERO_SKIP, GROUPS>(p0, b.x, P0, Pd, token, 0, dTmin, next, best);
        face_test<ZERO_SKIP, GROUPS>(p1, b.y, P0, Pd, token, 1, dTmin, next, best);
    }
    {
        double4 p2 = rec[2], p3 = rec[3];
        const int2 b = nb[1];
        CPF_PIN_W(p2, p3)
        face_test<ZERO_SKIP, GROUPS>(p2, b.x, P0, Pd, token, 2, dTmin, next, best);
        face_test<ZERO_SKIP, GROUPS>(p3, b.y, P0, Pd, token, 3, dTmin, next, best);
    }
    if (!(zNever || (zLast && ballot64(Pd.z != 0.0) == 0ull))) {
        double4 p4 = rec[4], p5 = rec[5];
        const int2 b = nb[2];
        CPF_PIN_W(p4, p5)
        face_test<ZERO_SKIP, GROUPS>(p4, b.x, P0, Pd, token, 4, dTmin, next, best);
        face_test<ZERO_SKIP, GROUPS>(p5, b.y, P0, Pd, token, 5, dTmin, next, best);
    }
    if (best >= 0) {
        S = axpy(dTmin, Pd, P0);
        outSlot = best;
    }
    return next;
}

// ONE RECORD of a visit on a mesh with two-record cells (7 ... 12 slots, see "cell records" below): the six face tests of the
// record -- slots base .. base + 5 of the cell, base = 0 or 6 -- continuing from (dTmin, next, best) and leaving the exit
// point to the caller, which concludes after the cell's last record.  Same tests, same order, same strict "<" as one pass
// over the cell's slots.  (The LOOKUP 2 instantiation runs EVERY visit through this one instance -- an ordinary cell is a
// first record that is also the last --: three inlined copies of the six tests cost the instantiation 25 % on a mesh without
// a single big cell.)
template <bool ZERO_SKIP, bool GROUPS>
__device__ __forceinline__ void trace_lds6_record(const D3& P0, const D3& Pd, const double4* rec, int token, int base, double& dTmin, int& next, int& best) {
    const int2* nb = reinterpret_cast<const int2*>(rec + 7);
    {
        double4 p0 = rec[0], p1 = rec[1];
        const int2 b = nb[0];
        CPF_PIN_W(p0, p1)
        face_test<ZERO_SKIP, GROUPS>(p0, b.x, P0, Pd, token, base + 0, dTmin, next, best);
        face_test<ZERO_SKIP, GROUPS>(p1, b.y, P0, Pd, token, base + 1, dTmin, next, best);
    }
    {
        double4 p2 = rec[2], p3 = rec[3];
        const int2 b = nb[1];
        CPF_PIN_W(p2, p3)
        face_test<ZERO_SKIP, GROUPS>(p2, b.x, P0, Pd, token, base + 2, dTmin, next, best);
        face_test<ZERO_SKIP, GROUPS>(p3, b.y, P0, Pd, token, base + 3, dTmin, next, best);
    }
    {
        double4 p4 = rec[4], p5 = rec[5];
        const int2 b = nb[2];
        CPF_PIN_W(p4, p5)
        face_test<ZERO_SKIP, GROUPS>(p4, b.x, P0, Pd, token, base + 4, dTmin, next, best);
        face_test<ZERO_SKIP, GROUPS>(p5, b.y, P0, Pd, token, base + 5, dTmin, next, best);
    }
}

// FLAT WALK (MeshView::zSide0 + a velocity field without a z component + no Brownian kick: a 2-D case as both tutorials'
// pitzDaily run it with D = 0): the mesh is z-layered with its z faces in slots 4, 5 (zPairLast) and the four side faces of
// every cell have nz == 0 EXACTLY; no particle has a displacement in z (Pd.z == +-0 exactly: dt * 0 added to and subtracted
// from P.z, and a mirror about a wall with nz == 0 leaves E.z alone).  Then, bit for bit: the z faces have den == +-0 for
// every lane and are never accepted (the zLast shortcut of trace_lds6, here decided at launch instead of per round), and a
// side face's den = fma(nz, Pd.z, t) = t and fd = fma(-nz, P0.z, u) = u up to the sign of a zero result, which no comparison
// sees -- two FMAs per face that need not be issued, a vote and a branch per round that need not be taken, and no z in the walk.
template <bool ZERO_SKIP>
__device__ __forceinline__ void face_test_flat(const double4& p, int bs, const D3& P0, const D3& Pd, int token, int s,
                                               double& dTmin, int& next, int& best) {
    const double den = fma(p.y, Pd.y, p.x * Pd.x);
    if (ZERO_SKIP && ballot64(den != 0.0) == 0ull) return;
    const double fd = fma(-p.y, P0.y, fma(-p.x, P0.x, p.w));
    if (((ballot64(den <= fd) | ballot64(fd >= 0.0)) & __builtin_amdgcn_uicmp((unsigned)bs, (unsigned)token, 33 /* ne */)) != 0ull) {
        const bool c1 = fabs(fd) <= fabs(den), c2 = den < 0.0 || fd >= 0.0;
        const bool c3 = fd < kTol, c4 = bs != token;
        if (c1 && c2 && c3 && c4) {
            const double dT = fd / den;
            if (dT > kTol && dT < dTmin) { dTmin = dT; next = bs; best = s; }
        }
    }
}
// (S.z is left alone: fma(dT, +-0, P0.z) is P0.z)
template <bool ZERO_SKIP>
__device__ __forceinline__ int trace_lds4_flat(D3& S, const D3& E, int cur, const double4* rec, int token, int& outSlot) {
    const D3 P0 = S;
    const D3 Pd = {E.x - P0.x, E.y - P0.y, 0.0};
    int next = cur, best = -1;
    double dTmin = 2.0;
    const int2* nb = reinterpret_cast<const int2*>(rec + 7);
    {
        double4 p0 = rec[0], p1 = rec[1];
        const int2 b = nb[0];
        CPF_PIN_W(p0, p1)
        face_test_flat<ZERO_SKIP>(p0, b.x, P0, Pd, token, 0, dTmin, next, best);
        face_test_flat<ZERO_SKIP>(p1, b.y, P0, Pd, token, 1, dTmin, next, best);
    }
    {
        double4 p2 = rec[2], p3 = rec[3];
        const int2 b = nb[1];
        CPF_PIN_W(p2, p3)
        face_test_flat<ZERO_SKIP>(p2, b.x, P0, Pd, token, 2, dTmin, next, best);
        face_test_flat<ZERO_SKIP>(p3, b.y, P0, Pd, token, 3, dTmin, next, best);
    }
    if (best >= 0) {
        S.x = fma(dTmin, Pd.x, P0.x); S.y = fma(dTmin, Pd.y, P0.y);
        outSlot = best;
    }
    return next;
}

// FLAT WALK UNDER THE KICK (round 6).  With the Brownian kick every particle moves in z, so the flat walk above does not apply --
// but on a one-cell-thick mesh whose side faces have nz == 0 exactly (zThin && zSide0), once fold_z has mirrored every lane's end
// point clear of the z planes (`zFold && !zUnclear`: the condition that already drops the z pair from trace_lds6), the four
// side faces are all that is tested, and their dropped terms are 0 * finite: den = fma(0, Pd.z, t) = t and fd = fma(-0, P0.z, u)
// = u up to the sign of a zero, which no comparison sees (E.z and P0.z are finite: a lane with a NaN or infinite end point is
// never `clear`).  The exit point keeps its z: S = P0 + dT * Pd with the real Pd.z.  Same bits as trace_lds6(..., zNever = true).
template <bool ZERO_SKIP>
__device__ __forceinline__ int trace_lds4_flat_z(D3& S, const D3& E, int cur, const double4* rec, int token, int& outSlot) {
    const D3 P0 = S;
    const D3 Pd = {E.x - P0.x, E.y - P0.y, E.z - P0.z};
    int next = cur, best = -1;
    double dTmin = 2.0;
    const int2* nb = reinterpret_cast<const int2*>(rec + 7);
    {
        double4 p0 = rec[0], p1 = rec[1];
        const int2 b = nb[0];
        CPF_PIN_W(p0, p1)
        face_test_flat<ZERO_SKIP>(p0, b.x, P0, Pd, token, 0, dTmin, next, best);
        face_test_flat<ZERO_SKIP>(p1, b.y, P0, Pd, token, 1, dTmin, next, best);
    }
    {
        double4 p2 = rec[2], p3 = rec[3];
        const int2 b = nb[1];
        CPF_PIN_W(p2, p3)
        face_test_flat<ZERO_SKIP>(p2, b.x, P0, Pd, token, 2, dTmin, next, best);
        face_test_flat<ZERO_SKIP>(p3, b.y, P0, Pd, token, 3, dTmin, next, best);
    }
    if (best >= 0) {
        S = axpy(dTmin, Pd, P0);
        outSlot = best;
    }
    return next;
}

// The same six face tests, two faces per wave-uniform decision: both denominators and both plane distances are computed
// up front (four independent FMA chains instead of two short ones between branches), then ONE test decides whether either
// face has a candidate lane.  No zero-denominator skip: a lane with den == 0 is a candidate only with fd >= 0, and is then
// rejected by the exact predicate (|fd| <= |den| leaves fd == 0, whose quotient is NaN) exactly as in face_test.  More
// vector instructions where faces used to be skipped for zero denominators (2-D cases), fewer branches and shorter
// dependent chains everywhere: used where every face is live anyway (3-D meshes, the Brownian kick).
__device__ __forceinline__ void face_accept(double den, double fd, int bs, int token, int s, double& dTmin, int& next, int& best) {
    const bool c1 = fabs(fd) <= fabs(den), c2 = den < 0.0 || fd >= 0.0;
    const bool c3 = fd < kTol, c4 = bs != token;
    if (c1 && c2 && c3 && c4) {
        const double dT = fd / den;
        if (dT > kTol && dT < dTmin) { dTmin = dT; next = bs; best = s; }
    }
}
__device__ __forceinline__ void face_pair_test(const double4& pa, const double4& pb, int2 b, const D3& P0, const D3& Pd, int token,
                                               int s, double& dTmin, int& next, int& best) {
    const double denA = dot3(pa, Pd), denB = dot3(pb, Pd);
    const double fdA = plane_dist(pa, P0), fdB = plane_dist(pb, P0);
    const unsigned long long mA = (ballot64(denA <= fdA) | ballot64(fdA >= 0.0)) & __builtin_amdgcn_uicmp((unsigned)b.x, (unsigned)token, 33 /* ne */);
    const unsigned long long mB = (ballot64(denB <= fdB) | ballot64(fdB >= 0.0)) & __builtin_amdgcn_uicmp((unsigned)b.y, (unsigned)token, 33 /* ne */);
    if ((mA | mB) != 0ull) {
        if (mA != 0ull) face_accept(denA, fdA, b.x, token, s, dTmin, next, best);
        if (mB != 0ull) face_accept(denB, fdB, b.y, token, s + 1, dTmin, next, best);
    }
}
__device__ __forceinline__ int trace_lds6_paired(D3& S, const D3& E, int cur, const double4* rec, int token, int& outSlot, bool zLast = false) {
    const D3 P0 = S;
    const D3 Pd = {E.x - P0.x, E.y - P0.y, E.z - P0.z};
    int next = cur, best = -1;
    double dTmin = 2.0;
    const int2* nb = reinterpret_cast<const int2*>(rec + 7);
    { const double4 p0 = rec[0], p1 = rec[1]; face_pair_test(p0, p1, nb[0], P0, Pd, token, 0, dTmin, next, best); }
    { const double4 p2 = rec[2], p3 = rec[3]; face_pair_test(p2, p3, nb[1], P0, Pd, token, 2, dTmin, next, best); }
    if (!(zLast && ballot64(Pd.z != 0.0) == 0ull)) {          // see trace_lds6
        const double4 p4 = rec[4], p5 = rec[5]; face_pair_test(p4, p5, nb[2], P0, Pd, token, 4, dTmin, next, best);
    }
    if (best >= 0) {
        S = axpy(dTmin, Pd, P0);
        outSlot = best;
    }
    return next;
}

// Cell records (MeshView::cellRec, 256 bytes per cell = 8 double4): [0..5] the planes of face slots 0..5, [6] U,
// [7] six neighbour ids (int32) + two spare words.  On meshes that are not all-hex (MeshView::mixed):
//   * a cell with FEWER than 6 faces (prism, tet, pyramid) is padded with null faces -- plane (0, 0, 0, -1), neighbour
//     kNullNbr: den == 0 and fd == -1 for every particle, so the face can never be accepted (ConvexQuery.cu:86-95) and
//     the six-slot tests give what the walk over its real faces gives, in the same slot order;
//   * a cell with MORE than 6 faces (next to a 2:1 refinement: 7 ... 24) has a HEADER record: neighbour word 0 =
//     kBigCellMark, word 1 = its first CSR slot, word 2 = its face count, null planes, U as usual.  A lane in such a
//     cell walks the CSR tables (planes / nbr in global memory) with trace_csr -- same arithmetic, same slot order.
// The reference cannot run such meshes at all (src/initCuda.H:64: tetsPerCell = 12).
//   * a cell with 7 ... 12 slots (true polyhedra: a pentagonal prism has seven planes, a polyDualMesh cell a dozen) has TWO
//     records: the first holds slots 0..5 and U, spare word 6 = kTwoRecMark, spare word 7 = the index of the second one,
//     which lies behind the nCells first records and holds slots 6..11 (null-padded).  A lane in such a cell tests the
//     first record in one round and the second in the next, carrying (dTmin, next, best) across -- the same twelve tests
//     in the same order with the same strict "<", so bit-identical to the walk over the CSR slots -- and both stay LDS
//     tests; a cell with more than 12 slots keeps the header record.
constexpr int kNullNbr = INT32_MIN + 5;
constexpr int kBigCellMark = INT32_MIN + 6;
constexpr int kTwoRecMark = INT32_MIN + 9;

// BOX RECORDS (MeshView::boxRec, 128 bytes = 16 doubles per cell; built by cpf_mesh.cpp when EVERY cell of the mesh is an
// axis-aligned box: six planes whose normals are exactly +-e_x, +-e_y, +-e_z, one of each -- blockMesh cases such as the
// TJunction tutorial).  Canonical slot k = 2 * axis + (normal component == -1):
//   doubles 0..5   the plane offsets d_k
//   bytes  48..71  the six neighbour codes (int32), bytes 72..75 the ORDER CODE: bits 3k..3k+2 = the slot's place in the
//                  walk's own slot order (HostTables), bits 18+2k, 19+2k = sign bits of the normal's two zero components
//   doubles 10..12 U
// What the six-face test becomes, bit for bit (the dropped terms of the reference's dot products are exact zeros, which
// change at most the sign of a zero result -- and no comparison below sees the sign of a zero):
//   den_k = n_a * Pd_a = +-Pd_a exactly,   fd_k = fl(d_k - n_a * P0_a)  (one rounding, as the innermost live fma of plane_dist).
// A face can be accepted (ConvexQuery.cu:86-95: fd < tol, tol < dT <= 1, smallest dT, ties to the lower slot) only if
// fd and den have equal signs; with den > 0 -- the face of the pair the particle moves AWAY from -- that needs fd > 0: the
// particle outside its cell by a rounding (never beyond 1e-13: fd < tol).  Hence per axis only the face the particle moves
// TOWARDS is tested (den = -|Pd_a|): three candidates per visit instead of six, no normal ever fetched, one LDS round trip
// for the whole record instead of three.  The rest -- a lane with fd > 0 on any face that is not the one it came in through,
// two axes with exactly equal dT (the ORIGINAL slot order decides, not the canonical one), a displacement that is not
// finite (0 * inf = NaN poisons every denominator of the reference form) -- sends the WAVE through trace_box_slow, which
// evaluates the reference's predicate on all six faces and breaks ties by original slot.  Parity: tests/test_gpu_box.py,
// tools/fuzz_parity.py box (structured clouds that sit on faces, edges and diagonals).
__device__ __forceinline__ double4 box_wall_plane(const double4* rec4, int k) {
    const unsigned code = reinterpret_cast<const unsigned*>(rec4)[18];
    const int axis = k >> 1;
    const double one = (k & 1) ? -1.0 : 1.0;
    const double z1 = ((code >> (18 + 2 * k)) & 1u) ? -0.0 : 0.0, z2 = ((code >> (19 + 2 * k)) & 1u) ? -0.0 : 0.0;
    double4 p;
    p.x = axis == 0 ? one : z1;
    p.y = axis == 1 ? one : (axis == 0 ? z1 : z2);
    p.z = axis == 2 ? one : z2;
    p.w = reinterpret_cast<const double*>(rec4)[k];
    return p;
}
// the reference's predicate on all six faces of a box record, ties to the lower ORIGINAL slot (rare path of trace_box)
template <bool GROUPS>
__device__ __forceinline__ void trace_box_slow(const D3& P0, const D3& Pd, const double4* rec4, int token, double& dTmin, int& next, int& best) {
    const double* rd = reinterpret_cast<const double*>(rec4);
    const int* ri = reinterpret_cast<const int*>(rec4);
    const unsigned code = (unsigned)ri[18];
    // (what the zero components of the normals contribute in the reference form: +-0, or NaN next to an infinity or a NaN)
    const double nanD = Pd.x * 0.0 + Pd.y * 0.0 + Pd.z * 0.0, nanP = P0.x * 0.0 + P0.y * 0.0 + P0.z * 0.0;
    int ordBest = 8;
#pragma unroll 1
    for (int k = 0; k < 6; ++k) {
        const int a = k >> 1;
        const double pa = a == 0 ? P0.x : (a == 1 ? P0.y : P0.z), da = a == 0 ? Pd.x : (a == 1 ? Pd.y : Pd.z);
        const double w = rd[k];
        const int nb = ri[12 + k];
        const double fd = ((k & 1) ? w + pa : w - pa) + nanP, den = ((k & 1) ? -da : da) + nanD;
        const int ord = (int)((code >> (3 * k)) & 7u);
        double dT = fd / den;
        if (__builtin_isinf(dT)) dT = -1.0;
        if (nb == token) continue;
        if (GROUPS && is_group(nb) && !(den < 0.0)) continue;       // face groups: outward crossings only (trace_in_cell)
        if (fd < kTol && dT > kTol && dT <= 1.0 && (dT < dTmin || (dT == dTmin && ord < ordBest))) { dTmin = dT; next = nb; best = k; ordBest = ord; }
    }
}
// GROUPS (a mesh of boxes with face groups -- 2:1-refined boxes, the castellated kind: a box with a split face keeps one slot for
// it, neighbour code = the group's): a group slot is only ever left with den < 0, i.e. through the face the particle moves
// TOWARDS -- the three candidates cover it as they are; away from it (den > 0) it is never accepted, so it does not count
// among the "outside a face" lanes either (a lane that came INTO the big cell through one of the group's pieces sits on that
// face with a token that is not the group's code)
template <bool ZERO_SKIP, bool GROUPS = false>
__device__ __forceinline__ int trace_box(D3& S, const D3& E, int cur, const double4* rec4, int token, int& outSlot) {
    const D3 P0 = S;
    const D3 Pd = {E.x - P0.x, E.y - P0.y, E.z - P0.z};
    const double2* rd = reinterpret_cast<const double2*>(rec4);
    const double2 w01 = rd[0], w23 = rd[1], w45 = rd[2];
    const int4 nA = reinterpret_cast<const int4*>(rec4)[3];
    const int2 nB = reinterpret_cast<const int2*>(rec4)[8];
    const double f0 = w01.x - P0.x, f1 = w01.y + P0.x, f2 = w23.x - P0.y, f3 = w23.y + P0.y, f4 = w45.x - P0.z, f5 = w45.y + P0.z;
    int next = cur, best = -1;
    double dTmin = 2.0;
    // lanes the three-candidate form does not cover (see above); voted on once, after the candidates
#define CPF_BOX_ODD(F, NB) (ballot64(F > 0.0) & __builtin_amdgcn_uicmp((unsigned)NB, (unsigned)token, 33 /* ne */) & (GROUPS ? ballot64(!is_group(NB)) : ~0ull))
    const unsigned long long odd = CPF_BOX_ODD(f0, nA.x) | CPF_BOX_ODD(f1, nA.y) | CPF_BOX_ODD(f2, nA.z) | CPF_BOX_ODD(f3, nA.w) |
                                   CPF_BOX_ODD(f4, nB.x) | CPF_BOX_ODD(f5, nB.y) |
                                   ballot64(!(fabs(Pd.x) + fabs(Pd.y) + fabs(Pd.z) < __builtin_inf()));
#undef CPF_BOX_ODD
    bool tie = false;
#define CPF_BOX_AXIS(A, D, FD0, FD1, NB0, NB1)                                                                     \
    if (!ZERO_SKIP || ballot64(D != 0.0) != 0ull) {                                                                \
        const bool pos = D > 0.0;                                                                                  \
        const double fdt = pos ? FD1 : FD0;                                                                        \
        const int nbt = pos ? NB1 : NB0;                                                                           \
        const bool c1 = fabs(fdt) <= fabs(D), c3 = fdt < kTol, c4 = nbt != token;                                  \
        if ((ballot64(c1) & ballot64(c3) & __builtin_amdgcn_uicmp((unsigned)nbt, (unsigned)token, 33)) != 0ull) {  \
            if (c1 && c3 && c4) {                                                                                  \
                const double dT = fdt / -fabs(D);                                                                  \
                const bool ok = dT > kTol;                                                                         \
                tie |= ok && dT == dTmin;                                                                          \
                if (ok && dT < dTmin) { dTmin = dT; next = nbt; best = 2 * A + (pos ? 1 : 0); }                    \
            }                                                                                                      \
        }                                                                                                          \
    }
    CPF_BOX_AXIS(0, Pd.x, f0, f1, nA.x, nA.y)
    CPF_BOX_AXIS(1, Pd.y, f2, f3, nA.z, nA.w)
    CPF_BOX_AXIS(2, Pd.z, f4, f5, nB.x, nB.y)
#undef CPF_BOX_AXIS
    if ((odd | ballot64(tie)) != 0ull) {
        next = cur; best = -1; dTmin = 2.0;
        trace_box_slow<GROUPS>(P0, Pd, rec4, token, dTmin, next, best);
    }
    if (best >= 0) {
        S = axpy(dTmin, Pd, P0);
        outSlot = best;
    }
    return next;
}

// trace_in_cell on the CSR slots [s0, s0 + nf) of one cell; outSlot is returned relative to s0.  A group code may come
// back as `next`: the caller resolves it (resolve_group), as after the six-slot tests.
__device__ __forceinline__ int trace_csr(D3& S, const D3& E, int cur, const double4* __restrict__ planes, const int32_t* __restrict__ nbr,
                                         int s0, int nf, int token, int& outSlot) {
    const D3 P0 = S;
    const D3 Pd = {E.x - P0.x, E.y - P0.y, E.z - P0.z};
    int next = cur, best = -1;
    double dTmin = 1.1;
    for (int s = 0; s < nf; ++s) {
        const double4 pl = planes[s0 + s];
        const int nb = nbr[s0 + s];
        const double fd = plane_dist(pl, P0), den = dot3(pl, Pd);
        double dT = fd / den;
        if (__builtin_isinf(dT)) dT = -1.0;
        if (nb == token) continue;
        if (is_group(nb) && !(den < 0.0)) continue;
        if (fd < kTol && dT > kTol && dT <= 1.0 && dT < dTmin) { dTmin = dT; next = nb; best = s; }
    }
    if (best >= 0) {
        S = axpy(dTmin, Pd, P0);
        outSlot = best;
    }
    return next;
}

// ------------------------------------------------------------------------------------------------
// "VertexVelocity" advect mode (cuda/particles.cu:244-313, 428-437): the velocity at P is the barycentric interpolation of
// VERTEX velocities in a tet of the particle's cell -- the tet whose smallest barycentric weight of P is largest (the product
// tracks cells, and the interpolant is continuous across the tets of a cell), then weighed exactly like the reference
// (w_X = det(tet with X := P) * (1 / det(tet))).  Shared by the staged advect and the fused cycle: same bits.
//
// CONE LOCATE (round 6; VertexField::cone != nullptr).  Evaluating all tetsPerCell tets costs 12 x 5 determinants per particle
// and cycle: 7.8 x the cell-constant cycle on pitzDaily (BENCH: config.vertex_velocity).  When the cell's tets are a FAN about one
// apex -- the cell centre, src/initCuda.H:99-105 -- that covers every direction exactly once (checked on the host at
// cpf_set_tets: shared apex, determinants of one sign, solid angles adding up to 4 pi, bounded condition), the tets' interiors
// are disjoint.  So if ONE tet holds P with every weight above a margin far beyond the determinants' rounding errors, every
// other tet of the cell has a negative weight in exact arithmetic -- a computed minimum of at most a rounding error -- and the
// arg-max of the full evaluation is that tet: its weights, computed with the very same expressions, ARE the full evaluation's
// result, bit for bit.  The candidate comes from a cheap approximate test (P - apex in the tet's cone: three dot products with
// precomputed rows, 9 FMAs per tet); how good that guess is affects only the speed.  A particle within the margin of a tet's
// face, edge or the apex, or outside its cell by a rounding, takes the full evaluation as before.
// cone: one 256-byte RECORD per tet, everything the cone locate needs of it behind one address -- [0..8] the rows of the
// inverse of [B-A C-A D-A] (approximate: the guess), [9] 1 / det4(A, B, C, D) (exact, computed by the same expressions on the
// device), [10..18] B, C, D, [19..30] the velocities of A, B, C, D (rewritten by cpf_set_vertex_velocity), [31] spare; apex: the
// cells' shared vertex A, one double4 per cell.  A particle's advect is two dependent fetches -- the rows of its cell's tets,
// then the rest of ONE record -- where the index-chasing form (tet -> vertex ids -> positions -> velocities) was five.
struct VertexField { const double* pos; const int32_t* tets; const double* vel; int tetsPerCell; const double* cone; const double4* apex; };
constexpr double kVertexMargin = 1e-8;          // on barycentric weights (O(1)); the host admits meshes whose weights carry errors < 1e-10
constexpr int kConeDoubles = 32;                // doubles per tet record (VertexField::cone)
__device__ __forceinline__ double det4(const D3& A, const D3& B, const D3& C, const D3& D) {
    const D3 a = {B.x - A.x, B.y - A.y, B.z - A.z}, b = {C.x - A.x, C.y - A.y, C.z - A.z}, d = {D.x - A.x, D.y - A.y, D.z - A.z};
    const D3 c = {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
    return d.x * c.x + d.y * c.y + d.z * c.z;
}
// ONE tet of the cone locate: the reference's weights of P in tet `t` (record f.cone[t], apex A) and, if every weight clears the
// margin, the interpolated velocity -- exactly what the evaluation of all tets returns then (see above)
__device__ __forceinline__ bool vertex_velocity_in_tet(const VertexField& f, const D3& Pp, const D3& A, int64_t t, D3& v) {
    const double2* R = reinterpret_cast<const double2*>(f.cone + kConeDoubles * t);
    const double2 q4 = R[4], q5 = R[5], q6 = R[6], q7 = R[7], q8 = R[8], q9 = R[9];
    const double2 q10 = R[10], q11 = R[11], q12 = R[12], q13 = R[13], q14 = R[14], q15 = R[15];
    const double inv = q4.y;                                                     // 1. / det4(A, B, C, D), computed on the device
    const D3 B = {q5.x, q5.y, q6.x}, C = {q6.y, q7.x, q7.y}, D = {q8.x, q8.y, q9.x};
    const double a = det4(Pp, B, C, D) * inv, b = det4(A, Pp, C, D) * inv;
    const double cc = det4(A, B, Pp, D) * inv, d = det4(A, B, C, Pp) * inv;
    if (!(fmin(fmin(a, b), fmin(cc, d)) > kVertexMargin)) return false;
    // P is well inside this tet: no other tet of the fan can win; its vertex velocities ride in the record
    const D3 vA = {q9.y, q10.x, q10.y}, vB = {q11.x, q11.y, q12.x}, vC = {q12.y, q13.x, q13.y}, vD = {q14.x, q14.y, q15.x};
    v = {((a * vA.x + b * vB.x) + cc * vC.x) + d * vD.x, ((a * vA.y + b * vB.y) + cc * vC.y) + d * vD.y,
         ((a * vA.z + b * vB.z) + cc * vC.z) + d * vD.z};
    return true;
}
__device__ __forceinline__ bool vertex_velocity(const VertexField& f, const D3& Pp, int c, D3& v) {
    auto ld = [](const double* a, int k) { return D3{a[3 * (int64_t)k], a[3 * (int64_t)k + 1], a[3 * (int64_t)k + 2]}; };
    int best = -1;
    double bestMin = 0.0, w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    if (f.cone != nullptr) {
        const int64_t t0 = (int64_t)c * f.tetsPerCell;
        int cand = 0;
        // the candidate: the tet in whose cone about the apex P lies best (per-lane reads of the records' first 72 bytes; the
        // streaming kernel stages them in LDS once per cell instead: cpf_stream.hip, cycle_begin)
        const double4 ap = f.apex[c];
        const D3 A = {ap.x, ap.y, ap.z};
        const D3 r = {Pp.x - A.x, Pp.y - A.y, Pp.z - A.z};
        double candMin = -1e300;
#pragma unroll 4
        for (int k = 0; k < f.tetsPerCell; ++k) {
            const double2* g = reinterpret_cast<const double2*>(f.cone + kConeDoubles * (t0 + k));
            const double2 g01 = g[0], g23 = g[1], g45 = g[2], g67 = g[3], g89 = g[4];
            const double cb = fma(g23.x, r.z, fma(g01.y, r.y, g01.x * r.x)), cc = fma(g45.y, r.z, fma(g45.x, r.y, g23.y * r.x));
            const double cd = fma(g89.x, r.z, fma(g67.y, r.y, g67.x * r.x));
            const double m = fmin(cb, fmin(cc, cd));
            if (m > candMin) { candMin = m; cand = k; }
        }
        if (vertex_velocity_in_tet(f, Pp, A, t0 + cand, v)) return true;
    }
    const bool full = best < 0;                                   // no cone tables, or the candidate is not clear of its tet's boundary
    for (int k = 0; full && k < f.tetsPerCell; ++k) {
        const int32_t* ix = f.tets + 4 * ((int64_t)c * f.tetsPerCell + k);
        const D3 A = ld(f.pos, ix[0]), B = ld(f.pos, ix[1]), C = ld(f.pos, ix[2]), D = ld(f.pos, ix[3]);
        const double den = det4(A, B, C, D);
        if (den == 0.0) continue;                                 // a bad tet (particles.cu:275-278) cannot hold P
        const double a = det4(Pp, B, C, D) * (1. / den), b = det4(A, Pp, C, D) * (1. / den);
        const double cc = det4(A, B, Pp, D) * (1. / den), d = det4(A, B, C, Pp) * (1. / den);
        const double m = fmin(fmin(a, b), fmin(cc, d));
        if (best < 0 || m > bestMin) { best = k; bestMin = m; w0 = a; w1 = b; w2 = cc; w3 = d; }
    }
    if (best < 0) return false;
    const int32_t* ix = f.tets + 4 * ((int64_t)c * f.tetsPerCell + best);
    const D3 vA = ld(f.vel, ix[0]), vB = ld(f.vel, ix[1]), vC = ld(f.vel, ix[2]), vD = ld(f.vel, ix[3]);
    v = {((w0 * vA.x + w1 * vB.x) + w2 * vC.x) + w3 * vD.x, ((w0 * vA.y + w1 * vB.y) + w2 * vC.y) + w3 * vD.y,
         ((w0 * vA.z + w1 * vB.z) + w2 * vC.z) + w3 * vD.z};
    return true;
}

// step-kernel variants (cpf_set_option "step_variant"); all give bit-identical results
enum { kVariantAuto = -1, kVariantGeneric = 0, kVariantFixed = 1, kVariantFixedScalar = 2, kVariantCoop = 3, kVariantStream = 4, kVariantAhead = 5 };

// Where a walk gets its mesh data from.  Every tracer runs the same arithmetic in the same order.
template <int VARIANT>
struct GlobalTracer {
    const MeshView& m;
    __device__ __forceinline__ int trace(D3& S, const D3& E, int cur, int token, int& outSlot) const {
        if (VARIANT == kVariantGeneric) return trace_in_cell(S, E, cur, m, token, outSlot);
        if (VARIANT == kVariantFixedScalar) {
            // particles are kept sorted by cell, so most waves sit in ONE cell for their first visit:
            // fetch that cell's planes once per wave through the scalar cache instead of 64 times
            const int ucur = __builtin_amdgcn_readfirstlane(cur);
            if (ballot64(cur != ucur) == 0ull)
                return trace_fixed<6, false>(S, E, cur, m.planes + 6 * (int64_t)ucur, m.nbr + 6 * (int64_t)ucur, token,
                                      outSlot, 6 * ucur);
        }
        return trace_fixed<6, false>(S, E, cur, m.planes + 6 * (int64_t)cur, m.nbr + 6 * (int64_t)cur, token, outSlot,
                              6 * cur);
    }
    __device__ __forceinline__ double4 velocity(int cur) const { return m.U[cur]; }
};

// Philox4x32-10 (Salmon et al. SC'11) keyed by (seed, "CPF1"), counter (gid, step): replaces the
// 48-byte-per-particle cuRAND XORWOW state of cuda/particles.cu:524-575 with nothing at all.
// Philox4x32-R (Salmon et al., SC'11).  R = 7: the variant with the fewest rounds that the paper reports as passing the
// whole of BigCrush ("Crush-resistant"; R = 10 is Random123's default, with a safety margin).  Each round is two
// quarter-rate 32 x 32 -> 64 multiplies (v_mad_u64_u32) and four XORs per lane and the streaming kernel is bound by its
// vector ALU: measured on one box (pitzDaily, D = 1.5e-5, 1e7 particles) 10 -> 7 rounds = 0.1865 -> 0.1825 ms per launch;
// the deviates as a whole (this + Box-Muller below) are 0.052 of the 0.188 ms (A/B build that returns constants instead).
// oracle/cellwalk.c states the same function; its known-answer test runs at R = 10, where Random123 publishes vectors.
#ifndef CPF_PHILOX_ROUNDS
#define CPF_PHILOX_ROUNDS 7
#endif
__device__ __forceinline__ void philox4x32(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < CPF_PHILOX_ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
// Three N(0,1) deviates for (particle gid, step) from ONE Philox4x32-7 block (parity with the reference's cuRAND stream is
// statistical by contract, SURVEY.md 8c, so the arithmetic is free): Box-Muller evaluated with the fp32 hardware
// transcendentals (v_log_f32, v_sqrt_f32, v_sin_f32 / v_cos_f32, whose argument is in revolutions: no 2*pi, no range
// reduction) -- words 0,1 give two deviates (one log, one sqrt), words 2,3 the third.  The radius uniform uses ALL 32
// bits of its word, u = (w + 0.5) * 2^-32: log2 u = log2(mantissa) + (exponent - 32) of x = float(w) + 0.5, so the tail
// reaches sqrt(2 * 33 * ln 2) = 6.76 sigma (round 2's 23-bit uniforms stopped at 5.77) and the result keeps its relative
// precision near u = 1; the angle uniform is 23-bit, ((w >> 9) + 0.5) * 2^-23.  The fp64 version this replaces (two
// log, two sqrt, sincospi + cospi in double) was a third of the Brownian kernel's time and cost it its occupancy; the
// kick itself, disp += sigma * xi, stays an fp64 fma.  oracle/cellwalk.c (cw_normal3_words) states the same transform
// with libm; what still differs from curand_normal_double is listed there.
__device__ __forceinline__ float log2_radius_uniform(uint32_t w) {
    const float x = (float)w + 0.5f;
    return __builtin_amdgcn_logf(__builtin_amdgcn_frexp_mantf(x)) + (float)(__builtin_amdgcn_frexp_expf(x) - 32);
}
__device__ __forceinline__ D3 normal3(uint64_t gid, uint32_t step, uint32_t seed) {
#ifdef CPF_AB_NO_RNG       // A/B builds only (tools/build_ab.sh): what the kick costs besides its deviates
    return {0.3 + (double)(gid & 1), -0.2, 0.1 * (double)(step & 1)};
#endif
    uint32_t c[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), step, 0u};
    philox4x32(c, seed, 0x43504631u);
    const float s = 1.0f / 8388608.0f;                                    // 2^-23
    const float u1 = ((float)(c[1] >> 9) + 0.5f) * s, u3 = ((float)(c[3] >> 9) + 0.5f) * s;
    const float k = -1.3862943611198906f;                                 // -2 ln 2: -2 ln u = k * log2 u
    const float r0 = __builtin_amdgcn_sqrtf(k * log2_radius_uniform(c[0]));
    const float r1 = __builtin_amdgcn_sqrtf(k * log2_radius_uniform(c[2]));
    return {(double)(r0 * __builtin_amdgcn_cosf(u1)), (double)(r0 * __builtin_amdgcn_sinf(u1)),
            (double)(r1 * __builtin_amdgcn_cosf(u3))};
}

__device__ __forceinline__ unsigned wave_sum(unsigned v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

struct StepStats { unsigned steps, hops, refl, lost; };

// block-level counter reduction: 4 global atomics per block, sharded over kCounterSlots slots
__device__ __forceinline__ void flush_stats(StepStats st, unsigned long long* __restrict__ counters, unsigned* sCnt) {
    if (counters == nullptr) return;             // statistics switched off (cpf_set_option "stats" 0)
    if (threadIdx.x < 4) sCnt[threadIdx.x] = 0;
    __syncthreads();
    st.steps = wave_sum(st.steps); st.hops = wave_sum(st.hops); st.refl = wave_sum(st.refl); st.lost = wave_sum(st.lost);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&sCnt[0], st.steps); atomicAdd(&sCnt[1], st.hops);
        atomicAdd(&sCnt[2], st.refl); atomicAdd(&sCnt[3], st.lost);
    }
    __syncthreads();
    if (threadIdx.x < 4 && sCnt[threadIdx.x])
        atomicAdd(&counters[(blockIdx.x & (kCounterSlots - 1)) * 4 + threadIdx.x], (unsigned long long)sCnt[threadIdx.x]);
}

}  // namespace cpf
