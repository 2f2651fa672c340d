// Host-side mesh ingest: OpenFOAM polyMesh arrays -> CSR cell/face-slot planes + locate grid.
//
// Replaces, for this path, what the reference's init fragment does on the host before any
// kernel runs: the polyMeshTetDecomposition loop (src/initCuda.H:86-110), the std::map face
// table HostTetMesh::getBoundaryMesh (third_party/RTXAdvect/cuda/HostTetMesh.h:307-430) and the
// OptiX BVH build (src/initCuda.H:132-139).  No tets, no BVH: cells keep their own faces.
//
// Compiled with -ffp-contract=off: the plane coefficients must not depend on how a host
// compiler chooses to fuse multiply-adds (tests compare them with an independent build).
#include "cpf_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <utility>

namespace cpf {
namespace {

struct Vec { double x, y, z; };
inline Vec operator+(Vec a, Vec b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec operator-(Vec a, Vec b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline double dot(Vec a, Vec b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec cross(Vec a, Vec b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline Vec at(const double* p, int64_t i) { return {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }

// Unit normal (owner -> neighbour) and centre of one face: triangle fan about the vertex
// average, area-weighted centre (the scheme of OpenFOAM's face centres/areas; exact for planar
// faces, best-fit plane through the centre otherwise).
template <typename Label>
bool face_plane(const double* pts, const Label* verts, int nv, Vec& n, Vec& centre) {
    Vec est{0, 0, 0};
    for (int i = 0; i < nv; ++i) est = est + at(pts, verts[i]);
    est = {est.x / nv, est.y / nv, est.z / nv};
    Vec sumN{0, 0, 0}, sumAc{0, 0, 0};
    double sumA = 0.0;
    for (int i = 0; i < nv; ++i) {
        Vec p = at(pts, verts[i]), q = at(pts, verts[i + 1 == nv ? 0 : i + 1]);
        Vec c3 = (p + q) + est;
        Vec nrm = cross(q - p, est - p);
        double a = std::sqrt(dot(nrm, nrm));
        sumN = sumN + nrm;
        sumA += a;
        sumAc = sumAc + Vec{a * c3.x, a * c3.y, a * c3.z};
    }
    centre = sumA > 0.0 ? Vec{sumAc.x / (3.0 * sumA), sumAc.y / (3.0 * sumA), sumAc.z / (3.0 * sumA)} : est;
    double len = std::sqrt(dot(sumN, sumN));
    if (!(len > 0.0)) return false;
    n = {sumN.x / len, sumN.y / len, sumN.z / len};
    // Components that are rounding noise (|n_k| <= 1e-12 of a unit normal: the cross products of a face that lies in a
    // coordinate plane leave 1e-16 ... 1e-13 there) are ZERO: an axis-aligned face then has an exactly axis-aligned
    // normal, its denominator n . Pd is exactly 0 for every particle that does not move along that axis, and the walk's
    // zero-denominator skips (cpf_walk.h) -- in particular the one test that drops both z faces of a 2-D case -- can
    // fire.  pitzDaily: 30 % of the front/back faces came out with |nx|, |ny| ~ 1e-16, which switched the z-pair skip
    // off for the whole mesh.  Stated independently in oracle/cellwalk.c (the tables are compared bit for bit).
    const bool sx = std::fabs(n.x) <= 1e-12, sy = std::fabs(n.y) <= 1e-12, sz = std::fabs(n.z) <= 1e-12;
    if (sx || sy || sz) {
        if (sx) n.x = 0.0;
        if (sy) n.y = 0.0;
        if (sz) n.z = 0.0;
        const double l2 = std::sqrt(dot(n, n));
        n = {n.x / l2, n.y / l2, n.z / l2};
    }
    return true;
}

}  // namespace

template <typename Label>
std::string build_tables(const double* points, int64_t nPoints, const Label* faceOff, const Label* faceVerts,
                         int64_t nFaces, const Label* owner, const Label* neighbour, int64_t nInternal,
                         int64_t nCells, HostTables& out) {
    if (nCells <= 0 || nFaces <= 0 || nPoints <= 0) return "empty mesh";
    if (nInternal < 0 || nInternal > nFaces) return "nInternal out of range";
    // (boundary faces are coded -(face + 1) in the neighbour table; the few codes next to INT32_MIN are reserved:
    // the walk's "no cell yet" token and the streaming kernel's "sat this round out")
    if (nCells > INT32_MAX - 16 || nFaces > INT32_MAX - 16) return "mesh too large for 32-bit cell/face ids";
    if (faceOff[0] != 0) return "faceOffsets[0] != 0";
    for (int64_t f = 0; f < nFaces; ++f) {
        int64_t nv = (int64_t)faceOff[f + 1] - (int64_t)faceOff[f];
        if (nv < 3) return "face " + std::to_string(f) + " has fewer than 3 vertices";
        if (owner[f] < 0 || owner[f] >= nCells) return "owner[" + std::to_string(f) + "] out of range";
        if (f < nInternal && (neighbour[f] < 0 || neighbour[f] >= nCells))
            return "neighbour[" + std::to_string(f) + "] out of range";
        if (f < nInternal && neighbour[f] == owner[f]) return "face " + std::to_string(f) + " has owner == neighbour";
    }
    const int64_t nVertsTotal = faceOff[nFaces];
    for (int64_t i = 0; i < nVertsTotal; ++i)
        if (faceVerts[i] < 0 || faceVerts[i] >= nPoints) return "face vertex id out of range";

    out = HostTables();
    out.nCells = nCells;
    out.nSlots = nFaces + nInternal;
    if (out.nSlots > INT32_MAX - 2) return "too many cell-face slots";
    out.cellOff.assign((size_t)nCells + 1, 0);
    std::vector<int32_t> fill((size_t)nCells, 0);
    for (int64_t f = 0; f < nFaces; ++f) out.cellOff[(size_t)owner[f] + 1]++;
    for (int64_t f = 0; f < nInternal; ++f) out.cellOff[(size_t)neighbour[f] + 1]++;
    for (int64_t c = 0; c < nCells; ++c) {
        int32_t k = out.cellOff[(size_t)c + 1];
        if (k < 4) return "cell " + std::to_string(c) + " has fewer than 4 faces";
        out.cellOff[(size_t)c + 1] += out.cellOff[(size_t)c];
    }
    out.planes.resize((size_t)out.nSlots * 4);
    out.nbr.resize((size_t)out.nSlots);
    // cell AABBs for the locate grid, gathered while we touch every face anyway
    std::vector<double> bmin((size_t)nCells * 3, 1e300), bmax((size_t)nCells * 3, -1e300);
    auto grow = [&](int64_t c, const Label* v, int nv) {
        for (int i = 0; i < nv; ++i)
            for (int k = 0; k < 3; ++k) {
                double val = points[3 * (int64_t)v[i] + k];
                bmin[3 * c + k] = std::min(bmin[3 * c + k], val);
                bmax[3 * c + k] = std::max(bmax[3 * c + k], val);
            }
    };
    // slot order = mesh.cells()[c]: owned faces ascending, then neighbour faces ascending
    for (int pass = 0; pass < 2; ++pass) {
        const int64_t nf = pass == 0 ? nFaces : nInternal;
        for (int64_t f = 0; f < nf; ++f) {
            const int64_t c = pass == 0 ? owner[f] : neighbour[f];
            const int64_t s = out.cellOff[(size_t)c] + fill[(size_t)c]++;
            const Label* v = faceVerts + faceOff[f];
            const int nv = (int)(faceOff[f + 1] - faceOff[f]);
            Vec n, cf;
            if (!face_plane(points, v, nv, n, cf)) return "face " + std::to_string(f) + " has zero area";
            if (pass == 0) n = {-n.x, -n.y, -n.z};   // into the owner
            out.planes[4 * s + 0] = n.x;
            out.planes[4 * s + 1] = n.y;
            out.planes[4 * s + 2] = n.z;
            out.planes[4 * s + 3] = dot(n, cf);
            out.nbr[(size_t)s] = pass == 0 ? (f < nInternal ? (int32_t)neighbour[f] : (int32_t)(-(f + 1)))
                                           : (int32_t)owner[f];
            grow(c, v, nv);
        }
    }

    // ---- one slot per distinct PLANE of a cell.  Next to a 2:1 refinement a cell holds the four (two) pieces of a split
    // face, all in one plane; the plane-exit test cannot tell which piece a segment leaves through (their exit parameters
    // tie, exactly or within rounding), and a particle that came in through one piece sits on its siblings' plane.  So
    // the faces of a cell that share a plane share a slot: going through the cell's faces in order, a face joins the
    // first earlier slot of the same kind (internal / boundary) whose plane matches -- each normal component within
    // kCoplanar, the offsets within kCoplanar * (1 + |d|) -- or else opens the next slot with its own plane.  A boundary
    // slot keeps its first face's code (all boundary faces reflect alike).  An internal slot with several faces becomes a
    // FACE GROUP: neighbour code kGroupBase + g, the pieces' cells listed in groupNbr[groupOff[g] ..) in face order;
    // the walk picks the piece at the exit point (cpf_walk.h, resolve_group).  The reference has no such cells
    // (src/initCuda.H:64, hexes only); the rule is this repo's, stated here and in oracle/cellwalk.c independently.
    // A mesh without coplanar faces -- every hex mesh -- passes through unchanged.
    {
        constexpr double kCoplanar = 1e-9;
        if (nFaces >= (int64_t(1) << 30) - 1) return "mesh too large: face ids must stay below 2^30 (face-group codes)";
        out.groupOff.assign(1, 0);
        int64_t w = 0;                                       // compacted write position (never ahead of the read position)
        std::vector<std::pair<int32_t, int32_t>> members;    // (slot, neighbour cell) of the cell's internal faces, face order
        for (int64_t c = 0; c < nCells; ++c) {
            const int64_t s0 = out.cellOff[(size_t)c], s1 = out.cellOff[(size_t)c + 1], w0 = w;
            members.clear();
            for (int64_t s = s0; s < s1; ++s) {
                const double* q = &out.planes[4 * (size_t)s];
                const int32_t nb = out.nbr[(size_t)s];
                int64_t slot = -1;
                for (int64_t r = w0; r < w && slot < 0; ++r) {
                    const double* p = &out.planes[4 * (size_t)r];
                    if ((out.nbr[(size_t)r] >= 0) != (nb >= 0)) continue;
                    const bool same = std::fabs(q[0] - p[0]) <= kCoplanar && std::fabs(q[1] - p[1]) <= kCoplanar &&
                                      std::fabs(q[2] - p[2]) <= kCoplanar && std::fabs(q[3] - p[3]) <= kCoplanar * (1.0 + std::fabs(p[3]));
                    if (same) slot = r;
                }
                if (slot < 0) {
                    std::memmove(&out.planes[4 * (size_t)w], q, 4 * sizeof(double));
                    out.nbr[(size_t)w] = nb;
                    slot = w++;
                }
                if (nb >= 0) members.emplace_back((int32_t)slot, nb);
            }
            for (int64_t r = w0; r < w; ++r) {
                if (out.nbr[(size_t)r] < 0) continue;
                int k = 0;
                for (const auto& mb : members) k += mb.first == (int32_t)r;
                if (k < 2) continue;
                for (const auto& mb : members) if (mb.first == (int32_t)r) out.groupNbr.push_back(mb.second);
                out.nbr[(size_t)r] = kGroupBase + (int32_t)(out.groupOff.size() - 1);
                out.groupOff.push_back((int32_t)out.groupNbr.size());
            }
            out.cellOff[(size_t)c] = (int32_t)w0;
            const int32_t k = (int32_t)(w - w0);
            out.maxCellFaces = std::max(out.maxCellFaces, k);
            if (k > 6) ++out.nBigCells;
            if (k > 12) ++out.nHugeCells;
            out.minCellFaces = c == 0 ? k : std::min(out.minCellFaces, k);
        }
        out.cellOff[(size_t)nCells] = (int32_t)w;
        if ((int64_t)out.groupOff.size() > (int64_t(1) << 30) - 32) return "too many face groups for 32-bit neighbour codes";
        out.nSlots = w;
        out.planes.resize((size_t)w * 4);
        out.nbr.resize((size_t)w);
        if (out.groupNbr.empty()) out.groupNbr.push_back(0);   // (uploads want a non-empty table)
    }

    // ---- z-layered meshes: the two z faces of every cell go LAST.  If every cell has exactly six faces of which exactly
    // two have a normal along z EXACTLY (nx == 0 and ny == 0: the front/back faces of a 2-D case extruded in z -- both
    // tutorials' pitzDaily -- and the layers of any mesh extruded along z), those two take slots 4 and 5, everything
    // else keeps OpenFOAM's order.  A particle whose displacement has dz == 0 exactly (2-D flow without diffusion)
    // has a denominator of exactly +-0 against such a face and can never leave through it (ConvexQuery.cu:86-95:
    // dT = +-inf -> -1, or NaN); with the pair in a fixed place a wave of such particles drops both faces with ONE test
    // instead of fetching both planes and evaluating two denominators (cpf_walk.h, trace_lds6).
    // (Measured and dropped in round 3: the same idea for ALL three axes of an axis-aligned mesh -- slots ordered x, y, z
    // pairs and a two-operation form of the face test for faces exactly along an axis, den = n_a * Pd_a, fd = d - n_a * P_a,
    // bit-identical because the dropped terms are exact zeros.  5-7 % SLOWER on every mesh, pitzDaily included, where
    // only the code around the unchanged tests differed: the kernel is bound by how its instructions are scheduled, and
    // three more wave-uniform branches and a second form of every pair cost more than 24 operations per round save.)  The slot order is part
    // of the walk's definition (ties in dT go to the lower slot), so oracle/cellwalk.c states the same rule.
    out.zPairLast = out.minCellFaces == 6 && out.maxCellFaces == 6;
    for (int64_t c = 0; c < nCells && out.zPairLast; ++c) {
        int nz = 0;
        for (int s = 0; s < 6; ++s) {
            const double* pl = &out.planes[4 * ((size_t)out.cellOff[(size_t)c] + s)];
            nz += (pl[0] == 0.0 && pl[1] == 0.0) ? 1 : 0;
        }
        if (nz != 2) out.zPairLast = false;
    }
    if (out.zPairLast) {
        for (int64_t c = 0; c < nCells; ++c) {
            const size_t s0 = (size_t)out.cellOff[(size_t)c];
            double pl[6][4]; int32_t nb[6];
            int k = 0;
            for (int pass = 0; pass < 2; ++pass)               // stable: first the four others, then the two z faces
                for (int s = 0; s < 6; ++s) {
                    const double* q = &out.planes[4 * (s0 + s)];
                    const bool isZ = q[0] == 0.0 && q[1] == 0.0;
                    if (isZ != (pass == 1)) continue;
                    std::memcpy(pl[k], q, sizeof(pl[k])); nb[k] = out.nbr[s0 + s]; ++k;
                }
            std::memcpy(&out.planes[4 * s0], pl, sizeof(pl));
            std::memcpy(&out.nbr[s0], nb, sizeof(nb));
        }
    }

    // ---- ... and extruded STRAIGHT: the four other faces of every cell have nz == 0 exactly (the flat walk, cpf_walk.h)
    out.zSide0 = out.zPairLast;
    for (int64_t c = 0; c < nCells && out.zSide0; ++c) {
        const size_t s0 = (size_t)out.cellOff[(size_t)c];
        for (int s = 0; s < 4; ++s)
            if (out.planes[4 * (s0 + s) + 2] != 0.0) out.zSide0 = false;
    }

    // ---- one cell thick in z: z-layered, the two z faces of every cell are boundary faces, and all cells share the two
    // planes (to 1e-12 of the thickness: the offsets come out of per-face centroids).  What it is for: cpf_walk.h, fold_z.
    out.zThin = out.zPairLast;
    {
        double loMin = 1e300, loMax = -1e300, hiMin = 1e300, hiMax = -1e300;
        for (int64_t c = 0; c < nCells && out.zThin; ++c) {
            const size_t s0 = (size_t)out.cellOff[(size_t)c];
            if (out.nbr[s0 + 4] >= 0 || out.nbr[s0 + 5] >= 0) { out.zThin = false; break; }
            // ... and the four side faces have nz == 0 EXACTLY: fold_z's argument (a segment's cells depend on its x-y
            // projection alone; a reflection off a side wall leaves E.z alone) needs it.  A one-cell-thick mesh with
            // sheared or tapered side faces keeps its z faces in the walk.
            for (int s = 0; s < 4; ++s)
                if (out.planes[4 * (s0 + s) + 2] != 0.0) out.zThin = false;
            if (!out.zThin) break;
            const double za = out.planes[4 * (s0 + 4) + 3] * out.planes[4 * (s0 + 4) + 2];
            const double zb = out.planes[4 * (s0 + 5) + 3] * out.planes[4 * (s0 + 5) + 2];
            const double lo = std::min(za, zb), hi = std::max(za, zb);
            loMin = std::min(loMin, lo); loMax = std::max(loMax, lo); hiMin = std::min(hiMin, hi); hiMax = std::max(hiMax, hi);
        }
        if (out.zThin) {
            const double thick = hiMin - loMax;
            if (!(thick > 0.0) || loMax - loMin > 1e-12 * thick || hiMax - hiMin > 1e-12 * thick) out.zThin = false;
        }
    }

    // ---- box records (cpf_walk.h "box records"): every cell an axis-aligned box.  Canonical slot k = 2 * axis + (normal
    // component -1 ? 1 : 0); the record keeps each plane's offset d, the neighbour, the slot's place in the walk's own order
    // (ties in dT go to the lower ORIGINAL slot) and the signs of the normal's two zero components (a mirrored -0.0
    // coordinate keeps its sign exactly as with the full planes).  U (doubles 10..12) is filled in on the device.
    out.boxRec.clear();
    // (face groups are welcome -- a box with a split face keeps ONE slot for it, its neighbour code the group's: the castellated
    // kind of refined mesh, 2:1-refined boxes without snapping)
    if (out.minCellFaces == 6 && out.maxCellFaces == 6 && nCells > 0) {
        std::vector<double> box((size_t)nCells * 16, 0.0);
        bool ok = true;
        for (int64_t c = 0; c < nCells && ok; ++c) {
            const size_t s0 = (size_t)out.cellOff[(size_t)c];
            double* r = &box[(size_t)c * 16];
            int32_t* nb = reinterpret_cast<int32_t*>(r + 6);
            uint32_t code = 0, seen = 0;
            for (int s = 0; s < 6 && ok; ++s) {
                const double* pl = &out.planes[4 * (s0 + s)];
                int axis = -1;
                for (int a = 0; a < 3; ++a)
                    if (pl[a] == 1.0 || pl[a] == -1.0) {
                        if (axis >= 0) ok = false;
                        axis = a;
                    } else if (pl[a] != 0.0) ok = false;
                if (axis < 0 || !std::isfinite(pl[3])) ok = false;
                if (!ok) break;
                const int k = 2 * axis + (pl[axis] < 0.0 ? 1 : 0);
                if (seen & (1u << k)) { ok = false; break; }
                seen |= 1u << k;
                r[k] = pl[3];
                nb[k] = out.nbr[s0 + s];
                code |= (uint32_t)s << (3 * k);
                const int o1 = axis == 0 ? 1 : 0, o2 = axis == 2 ? 1 : 2;     // the two other components, in axis order
                code |= (std::signbit(pl[o1]) ? 1u : 0u) << (18 + 2 * k);
                code |= (std::signbit(pl[o2]) ? 1u : 0u) << (19 + 2 * k);
            }
            if (ok && seen != 63u) ok = false;
            nb[6] = (int32_t)code; nb[7] = 0;
        }
        if (ok) out.boxRec.swap(box);
    }

    // ---- uniform bin grid (initial locate; replaces the OptiX BVH, src/initCuda.H:134-139)
    for (int k = 0; k < 3; ++k) { out.lo[k] = 1e300; out.hi[k] = -1e300; }
    for (int64_t p = 0; p < nPoints; ++p)
        for (int k = 0; k < 3; ++k) {
            out.lo[k] = std::min(out.lo[k], points[3 * p + k]);
            out.hi[k] = std::max(out.hi[k], points[3 * p + k]);
        }
    // ---- per-cell boxes + bit layout of the sub-cell sort key.  The key groups particles that sit close together
    // INSIDE a cell so that the lanes of a wave cross the same faces in the same round.  An axis in which the mesh
    // is one cell thick (2-D cases extruded in z) gets no bits; with two axes left the longest domain axis (the
    // main flow direction of a channel) gets 2 bits and leads, the transverse one gets 5: measured rounds per wave
    // on the pitzDaily bench cloud 2.31 (4x4x4) -> 2.23 (4x32x1); three axes keep 4x4x4.
    {
        bool thin[3];
        int order[3] = {0, 1, 2};
        double dext[3];
        for (int k = 0; k < 3; ++k) {
            dext[k] = out.hi[k] - out.lo[k];
            thin[k] = true;
            for (int64_t c = 0; c < nCells && thin[k]; ++c)
                if (bmax[3 * c + k] - bmin[3 * c + k] < 0.999 * dext[k]) thin[k] = false;
        }
        std::sort(order, order + 3, [&](int a, int b) {
            if (thin[a] != thin[b]) return !thin[a];
            return dext[a] > dext[b] || (dext[a] == dext[b] && a < b);
        });
        const int nThick = (thin[0] ? 0 : 1) + (thin[1] ? 0 : 1) + (thin[2] ? 0 : 1);
        for (int r = 0; r < 3; ++r) {
            const int k = order[r];
            out.subOrder[r] = k;
            out.subBits[k] = thin[k] ? 0 : (nThick == 3 ? 2 : (nThick == 2 ? (r == 0 ? 2 : 5) : 7));
        }
        // Rank of every cell along a Morton curve through the centres of the cell boxes (10 bits per axis): the sort key of SPARSE
        // clouds (fewer than 8 particles per cell), where the step kernel is bound by record traffic and a cloud ordered along the
        // curve has a cell's neighbours in all three directions close by -- not just the one the mesh numbers along.
        {
            std::vector<std::pair<uint32_t, int32_t>> code((size_t)nCells);
            auto spread = [](uint32_t v) { v = (v | (v << 16)) & 0x030000FFu; v = (v | (v << 8)) & 0x0300F00Fu; v = (v | (v << 4)) & 0x030C30C3u; return (v | (v << 2)) & 0x09249249u; };
            for (int64_t c = 0; c < nCells; ++c) {
                uint32_t q[3];
                for (int k = 0; k < 3; ++k) {
                    const double mid = 0.5 * (bmin[3 * c + k] + bmax[3 * c + k]);
                    const double t = dext[k] > 0.0 ? (mid - out.lo[k]) / dext[k] : 0.0;
                    q[k] = (uint32_t)std::min(1023.0, std::max(0.0, t * 1024.0));
                }
                code[(size_t)c] = {spread(q[0]) | (spread(q[1]) << 1) | (spread(q[2]) << 2), (int32_t)c};
            }
            std::sort(code.begin(), code.end());
            out.curveRank.resize((size_t)nCells);
            for (int64_t r = 0; r < nCells; ++r) out.curveRank[(size_t)code[(size_t)r].second] = (int32_t)r;
        }
        out.cellBox.resize((size_t)nCells * 6);
        for (int64_t c = 0; c < nCells; ++c)
            for (int k = 0; k < 3; ++k) {
                const double e = bmax[3 * c + k] - bmin[3 * c + k];
                out.cellBox[6 * c + k] = (float)bmin[3 * c + k];
                out.cellBox[6 * c + 3 + k] = e > 0.0 ? (float)((double)(1 << out.subBits[k]) / e) : 0.0f;
            }
    }

    double ext[3], vol = 1.0, diag = 0.0;
    for (int k = 0; k < 3; ++k) {
        ext[k] = out.hi[k] - out.lo[k];
        diag += ext[k] * ext[k];
    }
    diag = std::sqrt(diag);
    const double pad = 1e-9 * diag;
    for (int k = 0; k < 3; ++k) {
        ext[k] = std::max(ext[k], 1e-12 * diag) + 2 * pad;
        out.origin[k] = out.lo[k] - pad;
        vol *= ext[k];
    }
    const double h = std::cbrt(vol / (double)nCells);
    int64_t nBins = 1;
    for (int k = 0; k < 3; ++k) {
        int64_t d = (int64_t)std::floor(ext[k] / h);
        d = std::max<int64_t>(1, std::min<int64_t>(d, 2048));
        out.dims[k] = (int32_t)d;
        nBins *= d;
    }
    while (nBins > 8 * nCells + 64) {   // keep the table small on very anisotropic boxes
        int kmax = 0;
        for (int k = 1; k < 3; ++k) if (out.dims[k] > out.dims[kmax]) kmax = k;
        nBins /= out.dims[kmax];
        out.dims[kmax] = std::max(1, out.dims[kmax] / 2);
        nBins *= out.dims[kmax];
    }
    for (int k = 0; k < 3; ++k) out.invBin[k] = (double)out.dims[k] / ext[k];
    auto binOf = [&](double v, int k) {
        int64_t b = (int64_t)std::floor((v - out.origin[k]) * out.invBin[k]);
        return (int32_t)std::max<int64_t>(0, std::min<int64_t>(b, out.dims[k] - 1));
    };
    out.binOff.assign((size_t)nBins + 1, 0);
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<int32_t> cursor;
        if (pass == 1) {
            for (int64_t b = 0; b < nBins; ++b) out.binOff[(size_t)b + 1] += out.binOff[(size_t)b];
            out.binCells.resize((size_t)out.binOff[(size_t)nBins]);
            cursor.assign(out.binOff.begin(), out.binOff.end() - 1);
        }
        for (int64_t c = 0; c < nCells; ++c) {
            int32_t b0[3], b1[3];
            for (int k = 0; k < 3; ++k) {
                b0[k] = binOf(bmin[3 * c + k] - pad, k);
                b1[k] = binOf(bmax[3 * c + k] + pad, k);
            }
            for (int32_t kz = b0[2]; kz <= b1[2]; ++kz)
                for (int32_t ky = b0[1]; ky <= b1[1]; ++ky)
                    for (int32_t kx = b0[0]; kx <= b1[0]; ++kx) {
                        size_t b = ((size_t)kz * out.dims[1] + ky) * out.dims[0] + kx;
                        if (pass == 0) out.binOff[b + 1]++;
                        else out.binCells[(size_t)cursor[b]++] = (int32_t)c;
                    }
        }
    }
    return std::string();
}

template std::string build_tables<int32_t>(const double*, int64_t, const int32_t*, const int32_t*, int64_t,
                                           const int32_t*, const int32_t*, int64_t, int64_t, HostTables&);
template std::string build_tables<int64_t>(const double*, int64_t, const int64_t*, const int64_t*, int64_t,
                                           const int64_t*, const int64_t*, int64_t, int64_t, HostTables&);

}  // namespace cpf
