/* TEST INFRASTRUCTURE ONLY (oracle).  CPU statement of the polyhedral-cell variant of the
 * reference's walk: the SAME plane-exit test, tolerances, inlet-face skip, hop/reflection
 * caps and advect/reflect/move order as oracle/tetwalk.c (reference:
 * third_party/RTXAdvect/query/ConvexQuery.cu:32-216, :239-436; cuda/particles.cu:316-373,
 * :659-704; src/advect.H:96-161), applied to the polyMesh cells themselves instead of to the
 * 12-tets-per-cell decomposition.  Because the reference's velocity is cell-constant
 * (src/initCuda.H:106-108) both walks give the same positions whenever they agree on the
 * containing cell; tests/test_oracle_cellwalk.py checks that against tetwalk.c / oracle/_ref.
 *
 * It exists so that the HIP kernels (which implement this formulation) can be compared
 * BIT-EXACTLY at sizes where the tet walk is too slow, and as the "port" CPU baseline.
 * Arithmetic contract shared with the kernels (written independently on both sides):
 * fp64, explicit fma() in dot products / axpy, no other contraction (-ffp-contract=off),
 * IEEE division, planes (unit inward normal n, offset d = n.Cf) per (cell, face-slot).
 * Nothing under cudaparticlesfoam_amd/ may call, link or import this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define CW_TOL 1e-13
#define CW_MAX_HOPS 50     /* ConvexQuery.cu:169 (tets there, cells here) */
#define CW_MAX_REFLECT 5   /* ConvexQuery.cu:353 */
#define CW_LOST (-1)       /* still at a wall after 5 reflections / left the domain: tetID -1 */
#define CW_FROZEN (-2)     /* w == 0 in the reference (cuda/particles.cu:333-338) */

typedef struct { double x, y, z; } v3;
static inline v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline double dotf(v3 a, v3 b) { return fma(a.z, b.z, fma(a.y, b.y, a.x * b.x)); }
/* signed plane distance (Cf - P).n = d - n.P as one fma chain: <= 0 on the inner side */
static inline double plane_dist(v3 n, double d, v3 P) { return fma(-n.z, P.z, fma(-n.y, P.y, fma(-n.x, P.x, d))); }
static inline v3 axpy(double s, v3 a, v3 b) { return V(fma(s, a.x, b.x), fma(s, a.y, b.y), fma(s, a.z, b.z)); }
/* plain (un-fused) forms for the one-off host geometry */
static inline double dotp(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 crossp(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

/* ------------------------------------------------------------------------------------------
 * polyMesh -> CSR cell/face-slot tables.  Slot order = mesh.cells()[c] order
 * (primitiveMesh::calcCells: owned faces ascending, then neighbour faces ascending) -- except on z-layered meshes
 * (EVERY cell has exactly six faces, exactly two of them with nx == 0 and ny == 0 exactly): there the two z faces of
 * each cell come last, in their original relative order, after the other four in theirs.  The order matters only for
 * exact ties in dT (the lower slot wins, trace_in_cell); the product's mesh layer (csrc/cpf_mesh.cpp) states the same
 * rule so that its kernels can drop both z faces with one test when no lane moves in z.
 *   planes[4*s..] = (nx, ny, nz, d): unit normal pointing INTO the cell, d = n . Cf
 *   nbr[s]        = neighbour cell, or -(face+1) for a boundary face, or CW_GROUP_BASE + g for a face group
 *
 * COPLANAR FACES OF ONE CELL ARE ONE SLOT.  Next to a 2:1 refinement a cell has the four (two) pieces of a split face,
 * all in one plane: the plane-exit test cannot tell which piece a segment leaves through (the exit parameters tie, exactly
 * or within rounding), so the slots of a cell are its distinct PLANES, not its faces.  In face order, a face joins the
 * FIRST earlier slot of its cell that is of the same kind (internal / boundary) and whose plane agrees with its own --
 * every normal component within CW_COPLANAR, offsets within CW_COPLANAR * (1 + |d|) -- and otherwise opens a new slot with
 * its own plane.  A boundary slot keeps the code of its first face (every boundary face reflects alike).  An internal
 * slot that collected several faces becomes face group g (numbered in cell, then slot order): nbr = CW_GROUP_BASE + g,
 * groupNbr[groupOff[g] .. groupOff[g+1]) = the pieces' neighbour cells in face order; which of them a segment enters
 * is decided at the exit point (resolve_group).  No reference semantics exist for such cells (src/initCuda.H:64: hexes
 * only; OpenFOAM's own tracking walks the tet decomposition, where the pieces belong to different tets); the product's
 * mesh layer (csrc/cpf_mesh.cpp) states the same rule.  A hex mesh has no coplanar faces and is untouched.
 * Face centre/area vector: triangle fan about the vertex average (OpenFOAM's
 * primitiveMeshFaceCentresAndAreas scheme, restated; exact for planar faces).
 * ------------------------------------------------------------------------------------------ */
#define CW_GROUP_BASE (INT_MIN + 16)          /* nbr code of face group g: CW_GROUP_BASE + g */
#define CW_IS_GROUP(nb) ((nb) < -(1 << 30))   /* boundary codes -(face+1) stay above: nFaces < 2^30 */
#define CW_COPLANAR 1e-9

int cw_build(const double* points, int nPoints, const int* faceOff, const int* faceVerts, int nFaces,
             const int* owner, const int* neighbour, int nInternal, int nCells,
             int* cellOff /*nCells+1*/, double* planes /*4*ncf*/, int* nbr /*ncf*/,
             int* groupOff /*ncf+1*/, int* groupNbr /*ncf*/, int* nGroupsOut) {
    (void)nPoints;
    if (nFaces >= (1 << 30) - 1) return -1;
    int* cnt = (int*)calloc((size_t)nCells + 1, sizeof(int));
    if (!cnt) return -1;
    for (int f = 0; f < nFaces; ++f) cnt[owner[f]]++;
    for (int f = 0; f < nInternal; ++f) cnt[neighbour[f]]++;
    cellOff[0] = 0;
    for (int c = 0; c < nCells; ++c) cellOff[c + 1] = cellOff[c] + cnt[c];
    memset(cnt, 0, ((size_t)nCells + 1) * sizeof(int));
    for (int pass = 0; pass < 2; ++pass) {
        int nf = pass == 0 ? nFaces : nInternal;
        for (int f = 0; f < nf; ++f) {
            int c = pass == 0 ? owner[f] : neighbour[f];
            int s = cellOff[c] + cnt[c]++;
            int b = faceOff[f], nv = faceOff[f + 1] - faceOff[f];
            v3 est = V(0, 0, 0);
            for (int i = 0; i < nv; ++i) {
                const double* p = points + 3 * faceVerts[b + i];
                est = add(est, V(p[0], p[1], p[2]));
            }
            est = V(est.x / nv, est.y / nv, est.z / nv);
            v3 sumN = V(0, 0, 0), sumAc = V(0, 0, 0);
            double sumA = 0.0;
            for (int i = 0; i < nv; ++i) {
                const double* pp = points + 3 * faceVerts[b + i];
                const double* qq = points + 3 * faceVerts[b + (i + 1 == nv ? 0 : i + 1)];
                v3 p = V(pp[0], pp[1], pp[2]), q = V(qq[0], qq[1], qq[2]);
                v3 c3 = add(add(p, q), est);
                v3 nrm = crossp(sub(q, p), sub(est, p));
                double a = sqrt(dotp(nrm, nrm));
                sumN = add(sumN, nrm);
                sumA += a;
                sumAc = add(sumAc, V(a * c3.x, a * c3.y, a * c3.z));
            }
            v3 Cf = sumA > 0.0 ? V(sumAc.x / (3.0 * sumA), sumAc.y / (3.0 * sumA), sumAc.z / (3.0 * sumA)) : est;
            double len = sqrt(dotp(sumN, sumN));
            v3 n = V(sumN.x / len, sumN.y / len, sumN.z / len);   /* owner -> neighbour (outward for owner) */
            /* rounding noise in a unit normal (|n_k| <= 1e-12) is zero: axis-aligned faces get exactly axis-aligned normals
             * (the product's mesh layer states the same rule; what it is for: csrc/cpf_mesh.cpp) */
            if (fabs(n.x) <= 1e-12 || fabs(n.y) <= 1e-12 || fabs(n.z) <= 1e-12) {
                if (fabs(n.x) <= 1e-12) n.x = 0.0;
                if (fabs(n.y) <= 1e-12) n.y = 0.0;
                if (fabs(n.z) <= 1e-12) n.z = 0.0;
                double l2 = sqrt(dotp(n, n));
                n = V(n.x / l2, n.y / l2, n.z / l2);
            }
            if (pass == 0) n = V(-n.x, -n.y, -n.z);               /* inward for the owner */
            planes[4 * s] = n.x; planes[4 * s + 1] = n.y; planes[4 * s + 2] = n.z;
            planes[4 * s + 3] = dotp(n, Cf);
            if (pass == 0) nbr[s] = f < nInternal ? neighbour[f] : -(f + 1);
            else nbr[s] = owner[f];
        }
    }
    free(cnt);
    /* coplanar faces of a cell -> one slot (see above); compacts the tables in place */
    {
        int w = 0, nGroups = 0, nMembers = 0;
        /* members are collected per cell in a scratch list: (logical slot, neighbour) in face order */
        int* memSlot = NULL; int* memNbr = NULL; int memCap = 0;
        groupOff[0] = 0;
        for (int c = 0; c < nCells; ++c) {
            const int s0 = cellOff[c], s1 = cellOff[c + 1];
            const int w0 = w;
            int nMem = 0;
            if (s1 - s0 > memCap) { memCap = s1 - s0; memSlot = (int*)realloc(memSlot, sizeof(int) * memCap); memNbr = (int*)realloc(memNbr, sizeof(int) * memCap); }
            for (int s = s0; s < s1; ++s) {
                const double* q = planes + 4 * s;
                const int nb = nbr[s];
                int into = -1;
                for (int r = w0; r < w && into < 0; ++r) {
                    const double* pr = planes + 4 * r;
                    const int nr = nbr[r];
                    if ((nr >= 0) != (nb >= 0)) continue;
                    if (fabs(q[0] - pr[0]) <= CW_COPLANAR && fabs(q[1] - pr[1]) <= CW_COPLANAR && fabs(q[2] - pr[2]) <= CW_COPLANAR &&
                        fabs(q[3] - pr[3]) <= CW_COPLANAR * (1.0 + fabs(pr[3]))) into = r;
                }
                if (into < 0) {                        /* a new slot (w <= s: compaction never overtakes the read position) */
                    memmove(planes + 4 * w, q, 4 * sizeof(double));
                    nbr[w] = nb;
                    into = w++;
                }
                if (nb >= 0) { memSlot[nMem] = into; memNbr[nMem] = nb; ++nMem; }
            }
            /* internal slots with more than one member become groups, in slot order */
            for (int r = w0; r < w; ++r) {
                if (nbr[r] < 0) continue;
                int k = 0;
                for (int i = 0; i < nMem; ++i) k += memSlot[i] == r;
                if (k < 2) continue;
                for (int i = 0; i < nMem; ++i) if (memSlot[i] == r) groupNbr[nMembers++] = memNbr[i];
                nbr[r] = CW_GROUP_BASE + nGroups;
                groupOff[++nGroups] = nMembers;
            }
            cellOff[c] = w0;
        }
        cellOff[nCells] = w;
        free(memSlot); free(memNbr);
        if (nGroups > (1 << 30) - 32) return -1;
        *nGroupsOut = nGroups;
    }
    /* z-layered meshes: z faces last (see above) */
    int layered = 1;
    for (int c = 0; c < nCells && layered; ++c) {
        if (cellOff[c + 1] - cellOff[c] != 6) { layered = 0; break; }
        int nz = 0;
        for (int s = cellOff[c]; s < cellOff[c + 1]; ++s) nz += (planes[4 * s] == 0.0 && planes[4 * s + 1] == 0.0);
        if (nz != 2) layered = 0;
    }
    for (int c = 0; c < nCells && layered; ++c) {
        double pl[24]; int nb[6], k = 0;
        const int s0 = cellOff[c];
        for (int wantZ = 0; wantZ < 2; ++wantZ)
            for (int s = s0; s < s0 + 6; ++s) {
                const int isZ = planes[4 * s] == 0.0 && planes[4 * s + 1] == 0.0;
                if (isZ != wantZ) continue;
                memcpy(pl + 4 * k, planes + 4 * s, 4 * sizeof(double)); nb[k] = nbr[s]; ++k;
            }
        memcpy(planes + 4 * s0, pl, sizeof(pl));
        memcpy(nbr + s0, nb, sizeof(nb));
    }
    return cellOff[nCells];
}

typedef struct {            /* the tables of cw_build */
    const int* cellOff; const double* planes; const int* nbr; const int* groupOff; const int* groupNbr;
} cw_tables;

/* Which piece of face group g does a segment enter that leaves the cell at X?  The piece whose CELL holds X best: the
 * smallest maximum, over that cell's slots, of X's signed plane distance (<= 0 inside); the first piece on equal scores. */
static int resolve_group(int g, v3 X, const cw_tables* t) {
    double bestScore = 1e301;
    int pick = t->groupNbr[t->groupOff[g]];
    for (int k = t->groupOff[g]; k < t->groupOff[g + 1]; ++k) {
        const int nb = t->groupNbr[k];
        double score = -1e300;
        for (int q = t->cellOff[nb]; q < t->cellOff[nb + 1]; ++q) {
            const double d = plane_dist(V(t->planes[4 * q], t->planes[4 * q + 1], t->planes[4 * q + 2]), t->planes[4 * q + 3], X);
            if (d > score) score = d;
        }
        if (score < bestScore) { bestScore = score; pick = nb; }
    }
    return pick;
}

/* One cell of the walk (traceIntet on a polyhedral cell).  `token` identifies the face we came
 * in through: the previous cell id after a hop, the boundary code after a reflection.
 *
 * Face groups (cw_build) add two things to the reference's rule, both only where a slot IS a group:
 *   1. outward crossings only (n.Pd < 0 for the inward-signed plane).  A particle that came in through one piece of a
 *      group lies on the group's plane, a rounding error outside it (fd = +4e-16) and moving inward, which the
 *      reference's acceptance test takes for an exit at dT ~ 2e-13 > tol; the token cannot skip the slot (it names the
 *      piece's cell, not the group).  A convex cell is only ever left with n.Pd < 0, so no real exit is lost.
 *   2. the cell entered is chosen at the exit point (resolve_group). */
static int trace_in_cell(v3* Ps, v3 Pe, int cur, const cw_tables* t, int token, int* outSlot) {
    const int* cellOff = t->cellOff; const double* planes = t->planes; const int* nbr = t->nbr;
    const v3 P0 = *Ps;
    const v3 Pd = sub(Pe, P0);
    int next = cur, best = -1;
    double dTmin = 1.1;
    for (int s = cellOff[cur]; s < cellOff[cur + 1]; ++s) {
        v3 n = V(planes[4 * s], planes[4 * s + 1], planes[4 * s + 2]);
        double fd = plane_dist(n, planes[4 * s + 3], P0);  /* (Cf - P0).n, <= 0 inside */
        double den = dotf(n, Pd);
        double dT = fd / den;
        if (isinf(dT)) dT = -1.0;
        if (nbr[s] == token) continue;
        if (CW_IS_GROUP(nbr[s]) && !(den < 0.0)) continue;
        if (fd < CW_TOL && dT > CW_TOL && dT <= 1.0 && dT < dTmin) {
            dTmin = dT;
            next = nbr[s];
            *Ps = axpy(dT, Pd, P0);
            *outSlot = s;
            best = s;
        }
    }
    if (best >= 0 && CW_IS_GROUP(next)) next = resolve_group(next - CW_GROUP_BASE, *Ps, t);
    return next;
}

/* ------------------------------------------------------------------------------------------
 * Philox4x32-R (Salmon et al., SC'11 "Parallel random numbers: as easy as 1, 2, 3"; constants
 * and known-answer vectors from the Random123 distribution, checked in tests/test_rng.py).
 * Replaces the reference's per-particle cuRAND XORWOW state (cuda/particles.cu:524-575) with
 * a stateless counter (global particle id, step) -- Brownian parity is statistical only.
 * ------------------------------------------------------------------------------------------ */
static inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1], n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
void cw_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]}, k[2] = {key[0], key[1]};
    for (int r = 0; r < rounds; ++r) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
    memcpy(out, c, sizeof(c));
}
/* R = 10: Random123's default, the round count its known-answer vectors are published for (tests/test_rng.py) */
void cw_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { cw_philox4x32(ctr, key, 10, out); }
/* What the kernels draw from (csrc/cpf_walk.h, CPF_PHILOX_ROUNDS): R = 7, the fewest rounds the paper reports as passing
 * BigCrush -- the same round function and key schedule as the vector-checked R = 10, three rounds earlier. */
#define CW_PHILOX_ROUNDS 7
/* Box-Muller on the four words of one Philox block, in single precision (the kernels' normal3, csrc/cpf_walk.h, which
 * evaluates log2 / sqrt / sin / cos with the fp32 hardware instructions: the two agree to a few 1e-6, asserted; the
 * statistics are identical).  Words 0,1 -> two deviates, words 2,3 -> the third.
 *   radius (words 0, 2): ALL 32 bits, u = (w + 0.5) * 2^-32 in (0,1), so that the tail reaches sqrt(2*33*ln 2) = 6.76
 *     sigma (23-bit uniforms stopped at 5.77: at 1e7 particles x 3 deviates x 1000 cycles a true Gaussian draws ~250
 *     values beyond that).  log2 u = log2(mantissa) + (exponent - 32) with x = (float)w + 0.5f = mantissa * 2^exponent,
 *     mantissa in [0.5,1): the integer part is exact, the fractional part keeps its full relative precision near u = 1.
 *   angle (words 1, 3): 23-bit uniforms ((w >> 9) + 0.5) * 2^-23 revolutions.
 * What remains different from the reference's curand_normal_double (cuda/particles.cu:551-575; parity statistical by
 * contract): the generator (Philox vs XORWOW), single- instead of double-precision transcendentals
 * (deviates carry ~1e-7 relative error) and the hard cap at 6.76 sigma (cuRAND's double Box-Muller: ~8.6 sigma). */
void cw_normal3_words(const uint32_t w[4], double out[3]) {
    const float s = 1.0f / 8388608.0f, twopi = 6.283185307179586f, k = -1.3862943611198906f;
    int e0, e2;
    const float m0 = frexpf((float)w[0] + 0.5f, &e0), m2 = frexpf((float)w[2] + 0.5f, &e2);
    const float l0 = log2f(m0) + (float)(e0 - 32), l2 = log2f(m2) + (float)(e2 - 32);
    const float u1 = ((float)(w[1] >> 9) + 0.5f) * s, u3 = ((float)(w[3] >> 9) + 0.5f) * s;
    const float r0 = sqrtf(k * l0), r1 = sqrtf(k * l2);
    out[0] = (double)(r0 * cosf(twopi * u1));
    out[1] = (double)(r0 * sinf(twopi * u1));
    out[2] = (double)(r1 * cosf(twopi * u3));
}
/* three N(0,1) deviates for (particle gid, step) */
void cw_normal3(uint64_t gid, uint32_t step, uint32_t seed, double out[3]) {
    uint32_t key[2] = {seed, 0x43504631u /* "CPF1" */};
    uint32_t ctr[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), step, 0}, w[4];
    cw_philox4x32(ctr, key, CW_PHILOX_ROUNDS, w);
    cw_normal3_words(w, out);
}
/* test helper: the (gid, step) in [0, n) x [step0, step0 + nSteps) whose FIRST radius word is smallest, i.e. whose
 * first two deviates lie farthest out */
void cw_scan_min_radius_word(uint32_t seed, uint32_t step0, int nSteps, int64_t n, int64_t* bestGid, uint32_t* bestStep,
                             uint32_t* bestWord) {
    uint32_t key[2] = {seed, 0x43504631u}, best = 0xFFFFFFFFu, bs = step0;
    int64_t bg = 0;
    for (int t = 0; t < nSteps; ++t)
        for (int64_t g = 0; g < n; ++g) {
            uint32_t ctr[4] = {(uint32_t)g, (uint32_t)((uint64_t)g >> 32), step0 + (uint32_t)t, 0}, w[4];
            cw_philox4x32(ctr, key, CW_PHILOX_ROUNDS, w);
            if (w[0] < best) { best = w[0]; bg = g; bs = step0 + (uint32_t)t; }
        }
    *bestGid = bg; *bestStep = bs; *bestWord = best;
}

typedef struct { long long hops, reflections, lost; } cw_stats;

/* advect + locate + reflect + move for one particle (src/advect.H:96-161, D = 0) */
static void step_one(int i, double* px, double* py, double* pz, int* cell, double* vel_out, double dt,
                     const cw_tables* t, const double* U, cw_stats* st,
                     double D, const int64_t* gid, uint32_t step, uint32_t seed) {
    const double* planes = t->planes;
    int cur = cell[i];
    if (cur < 0) { if (cur == CW_LOST) cell[i] = CW_FROZEN; return; }
    const v3 P = V(px[i], py[i], pz[i]);
    v3 vel = V(U[3 * cur], U[3 * cur + 1], U[3 * cur + 2]);
    const v3 Pn = axpy(dt, vel, P);
    v3 disp = sub(Pn, P);                      /* disp = (P + dt*vel) - P, particles.cu:358-359 */
    if (D > 0.0) {                             /* disp += xi*sqrt(2 D dt), particles.cu:560-569 */
        double xi[3];
        cw_normal3((uint64_t)(gid ? gid[i] : i), step, seed, xi);
        disp = axpy(sqrt(2.00 * D * dt), V(xi[0], xi[1], xi[2]), disp);
    }
    v3 Pe = add(P, disp);
    v3 Ps = P, Phit = P;
    int token = INT_MIN, next = cur, outSlot = -1, reflected = 0;
    for (int j = 0; j < CW_MAX_REFLECT; ++j) {
        for (int h = 0; h < CW_MAX_HOPS; ++h) {
            next = trace_in_cell(&Ps, Pe, cur, t, token, &outSlot);
            st->hops++;
            if (next == cur) break;
            if (next < 0) break;
            token = cur;
            cur = next;
        }
        if (next >= 0) break;                  /* segment ends inside `next` (or hop cap reached) */
        /* wall: mirror end point and velocity about the boundary face just hit */
        Phit = Ps; reflected = 1; st->reflections++;
        v3 n = V(planes[4 * outSlot], planes[4 * outSlot + 1], planes[4 * outSlot + 2]);
        double sd = dotf(n, Pe) - planes[4 * outSlot + 3];
        Pe = axpy(-2.0 * sd, n, Pe);
        vel = axpy(-2.0 * dotf(n, vel), n, vel);
        token = next;                          /* skip this boundary face when the walk resumes */
    }
    v3 Pnew;
    if (reflected) Pnew = add(Phit, sub(Pe, Phit));   /* p = P_hit; disp = P_end - P_hit; p += disp */
    else Pnew = add(P, disp);
    px[i] = Pnew.x; py[i] = Pnew.y; pz[i] = Pnew.z;
    if (next < 0) { next = CW_LOST; st->lost++; }
    cell[i] = next;
    if (vel_out) { vel_out[3 * i] = vel.x; vel_out[3 * i + 1] = vel.y; vel_out[3 * i + 2] = vel.z; }
}

void cw_step(double* px, double* py, double* pz, int* cell, double* vel_out, int n, double dt, int cycles,
             const int* cellOff, const double* planes, const int* nbr, const int* groupOff, const int* groupNbr,
             const double* U, int nthreads, long long* stats /* [hops, reflections, lost] or NULL */,
             double D, const int64_t* gid, uint32_t step0, uint32_t seed) {
    long long H = 0, R = 0, L = 0;
    const cw_tables tab = {cellOff, planes, nbr, groupOff, groupNbr};
    /* particles are independent: all cycles of one particle back to back, one parallel region */
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1) reduction(+ : H, R, L)
    for (int i = 0; i < n; ++i) {
        cw_stats st = {0, 0, 0};
        for (int c = 0; c < cycles; ++c)
            step_one(i, px, py, pz, cell, vel_out, dt, &tab, U, &st, D, gid, step0 + (uint32_t)c, seed);
        H += st.hops; R += st.reflections; L += st.lost;
    }
    if (stats) { stats[0] = H; stats[1] = R; stats[2] = L; }
}

/* Initial locate contract (query/RTQuery.cu:295-310: containing element id, < 0 if outside):
 * the LOWEST cell id whose every face has (Cf - P).n <= 0; -1 if none.  Brute force. */
void cw_locate_initial(const double* px, const double* py, const double* pz, int* cell, int n, int nCells,
                       const int* cellOff, const double* planes, int nthreads) {
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads > 0 ? nthreads : 1)
    for (int i = 0; i < n; ++i) {
        v3 P = V(px[i], py[i], pz[i]);
        int found = -1;
        for (int c = 0; c < nCells && found < 0; ++c) {
            int inside = 1;
            for (int s = cellOff[c]; s < cellOff[c + 1]; ++s) {
                v3 nn = V(planes[4 * s], planes[4 * s + 1], planes[4 * s + 2]);
                if (!(plane_dist(nn, planes[4 * s + 3], P) <= 0.0)) { inside = 0; break; }
            }
            if (inside) found = c;
        }
        cell[i] = found;
    }
}

/* "VertexVelocity" advect on CELL ids (what cpf_stage_advect_vertex implements): the product tracks cells, not tets,
 * so the tet of the cell that holds P is found first -- the one whose smallest barycentric weight is largest, lowest
 * index on ties -- and then weighted exactly like cuda/particles.cu:270-291.  The interpolant is continuous across
 * the tets of a cell, so a particle on a shared tet face gets the same velocity (to rounding) from either side. */
static double cw_det4(v3 A, v3 B, v3 Cc, v3 D) {
    const v3 a = sub(B, A), b = sub(Cc, A), d = sub(D, A);
    const v3 c = V(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
    return d.x * c.x + d.y * c.y + d.z * c.z;
}
void cw_advect_vertex(double* P, const int* cells, double* vels, double* disps, double dt, int n, const int* tets,
                      int tetsPerCell, const double* pos, const double* vertvel, int nthreads) {
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (int i = 0; i < n; ++i) {
        double* p = P + 4 * i;
        if (!p[3]) continue;
        const int c = cells[i];
        if (c < 0) { p[3] = 0.0; continue; }
        const v3 Pp = V(p[0], p[1], p[2]);
        int best = -1; double bestMin = 0.0, bw[4] = {0, 0, 0, 0};
        for (int k = 0; k < tetsPerCell; ++k) {
            const int* ix = tets + 4 * ((long long)c * tetsPerCell + k);
            const v3 A = V(pos[3 * ix[0]], pos[3 * ix[0] + 1], pos[3 * ix[0] + 2]), B = V(pos[3 * ix[1]], pos[3 * ix[1] + 1], pos[3 * ix[1] + 2]);
            const v3 Cc = V(pos[3 * ix[2]], pos[3 * ix[2] + 1], pos[3 * ix[2] + 2]), D = V(pos[3 * ix[3]], pos[3 * ix[3] + 1], pos[3 * ix[3] + 2]);
            const double den = cw_det4(A, B, Cc, D);
            if (den == 0.0) continue;
            const double w[4] = {cw_det4(Pp, B, Cc, D) * (1. / den), cw_det4(A, Pp, Cc, D) * (1. / den),
                                 cw_det4(A, B, Pp, D) * (1. / den), cw_det4(A, B, Cc, Pp) * (1. / den)};
            double m = w[0]; for (int q = 1; q < 4; ++q) if (w[q] < m) m = w[q];
            if (best < 0 || m > bestMin) { best = k; bestMin = m; for (int q = 0; q < 4; ++q) bw[q] = w[q]; }
        }
        if (best < 0) { p[3] = 0.0; continue; }                /* every tet of the cell degenerate */
        const int* ix = tets + 4 * ((long long)c * tetsPerCell + best);
        double vel[3];
        for (int q = 0; q < 3; ++q)
            vel[q] = ((bw[0] * vertvel[3 * ix[0] + q] + bw[1] * vertvel[3 * ix[1] + q]) + bw[2] * vertvel[3 * ix[2] + q]) +
                     bw[3] * vertvel[3 * ix[3] + q];
        for (int q = 0; q < 3; ++q) {
            const double pn = p[q] + dt * vel[q];
            vels[4 * i + q] = vel[q]; disps[4 * i + q] = pn - p[q];
        }
        vels[4 * i + 3] = -1.0; disps[4 * i + 3] = -1.0;
    }
}

int cw_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* diagnostics: one cycle, per-particle number of cell visits and reflections (divergence studies) */
void cw_step_count(double* px, double* py, double* pz, int* cell, int n, double dt, const int* cellOff,
                   const double* planes, const int* nbr, const int* groupOff, const int* groupNbr, const double* U,
                   int nthreads, int* visits, int* reflections) {
    const cw_tables tab = {cellOff, planes, nbr, groupOff, groupNbr};
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (int i = 0; i < n; ++i) {
        cw_stats st = {0, 0, 0};
        step_one(i, px, py, pz, cell, NULL, dt, &tab, U, &st, 0.0, NULL, 0, 0);
        visits[i] = (int)st.hops; reflections[i] = (int)st.reflections;
    }
}
