// Rank-direct mesh ingest: stitch the per-rank pieces of a decomposed polyMesh into the global mesh (host code).
//
// Replaces the reference's gather-to-master (src/initCuda.H:207-371): there every rank sends its points, cell
// centres and 12-tets-per-cell list, and the master merges coincident points with a linear search per point
// (DynamicList::find inside a loop over all points: O(n^2), :337-352).  Here every rank hands over its piece as
// it is -- points, faces, owner/neighbour; processor patches are ordinary boundary faces of the piece -- and
//   * points that coincide EXACTLY (the reference's criterion, :341 `pointsSeen.find(pos)`) become one point,
//     found through a hash of the three coordinates' bit patterns: O(n);
//   * a boundary face of piece A and a boundary face of piece B on the same (merged) points are the two sides of
//     one interior face: owner = the lower global cell id, orientation = that side's copy (its normal points out
//     of the owner, OpenFOAM's convention); all other boundary faces stay boundary;
//   * global cell id = (cells of the pieces before it) + local id -- `globalIndex` in the reference (:229-230),
//     so the concatenation of the ranks' U slices is the global U (:310-313);
//   * interior faces are put in OpenFOAM's upper-triangular order (by owner, then neighbour), boundary faces keep
//     piece order: a mesh that is split and stitched again gets its interior faces back in their original order.
#include <algorithm>
#include <cstring>
#include <numeric>
#include <string>
#include <unordered_map>
#include <vector>

#include "cpf.h"
#include "cpf_internal.h"

struct cpf_merged_mesh {
    std::vector<double> points;
    std::vector<int64_t> faceOffsets, faceVerts, owner, neighbour;
    int64_t nInternal = 0, nCells = 0;
};

namespace {

struct PointKey {
    uint64_t a, b, c;
    bool operator==(const PointKey& o) const { return a == o.a && b == o.b && c == o.c; }
};
struct PointHash {
    size_t operator()(const PointKey& k) const {
        uint64_t h = k.a * 0x9E3779B97F4A7C15ull;
        h ^= (k.b + 0xBF58476D1CE4E5B9ull + (h << 6) + (h >> 2));
        h ^= (k.c + 0x94D049BB133111EBull + (h << 6) + (h >> 2));
        return (size_t)h;
    }
};
inline uint64_t bits_of(double v) {
    if (v == 0.0) v = 0.0;                      // -0.0 == 0.0 for the reference's operator==: one key
    uint64_t u; std::memcpy(&u, &v, 8);
    return u;
}

struct FaceKey {                                 // the face's merged point ids, sorted
    std::vector<int64_t> v;
    bool operator==(const FaceKey& o) const { return v == o.v; }
};
struct FaceHash {
    size_t operator()(const FaceKey& k) const {
        uint64_t h = 0xCBF29CE484222325ull;
        for (int64_t x : k.v) { h ^= (uint64_t)x; h *= 0x100000001B3ull; }
        return (size_t)h;
    }
};

struct Face { std::vector<int64_t> verts; int64_t owner, neighbour; };

std::string merge_impl(const cpf_mesh_part* parts, int nParts, cpf_merged_mesh& out) {
    if (!parts || nParts < 1) return "cpf_merge_mesh_parts: no pieces";
    std::unordered_map<PointKey, int64_t, PointHash> pointId;
    std::vector<Face> interior, boundary;
    std::unordered_map<FaceKey, size_t, FaceHash> open;         // unmatched boundary faces -> index into `boundary`
    std::vector<char> boundaryDead;
    std::vector<int> boundaryPiece;                             // which piece a boundary face came from
    int64_t cellBase = 0;
    for (int p = 0; p < nParts; ++p) {
        const cpf_mesh_part& m = parts[p];
        const std::string where = "piece " + std::to_string(p) + ": ";
        if (m.nPoints < 0 || m.nFaces < 0 || m.nCells < 0 || m.nInternalFaces < 0 || m.nInternalFaces > m.nFaces)
            return where + "negative or inconsistent sizes";
        if (m.labelBytes != 4 && m.labelBytes != 8) return where + "labelBytes must be 4 or 8";
        auto label = [&](const void* base, int64_t i) -> int64_t {
            return m.labelBytes == 8 ? static_cast<const int64_t*>(base)[i] : (int64_t) static_cast<const int32_t*>(base)[i];
        };
        if ((m.nPoints && !m.points) || (m.nFaces && (!m.faceOffsets || !m.faceVerts || !m.owner)) ||
            (m.nInternalFaces && !m.neighbour))
            return where + "null array";
        std::vector<int64_t> local((size_t)m.nPoints);
        for (int64_t i = 0; i < m.nPoints; ++i) {
            const PointKey k{bits_of(m.points[3 * i]), bits_of(m.points[3 * i + 1]), bits_of(m.points[3 * i + 2])};
            auto it = pointId.find(k);
            if (it == pointId.end()) {
                it = pointId.emplace(k, (int64_t)(out.points.size() / 3)).first;
                out.points.insert(out.points.end(), {m.points[3 * i], m.points[3 * i + 1], m.points[3 * i + 2]});
            }
            local[(size_t)i] = it->second;
        }
        for (int64_t f = 0; f < m.nFaces; ++f) {
            const int64_t a = label(m.faceOffsets, f), b = label(m.faceOffsets, f + 1);
            if (a < 0 || b < a) return where + "face offsets not ascending";
            Face face;
            face.verts.reserve((size_t)(b - a));
            for (int64_t k = a; k < b; ++k) {
                const int64_t v = label(m.faceVerts, k);
                if (v < 0 || v >= m.nPoints) return where + "face vertex out of range";
                face.verts.push_back(local[(size_t)v]);
            }
            const int64_t own = label(m.owner, f);
            if (own < 0 || own >= m.nCells) return where + "owner out of range";
            face.owner = cellBase + own;
            if (f < m.nInternalFaces) {
                const int64_t nei = label(m.neighbour, f);
                if (nei < 0 || nei >= m.nCells) return where + "neighbour out of range";
                face.neighbour = cellBase + nei;
                if (face.neighbour < face.owner) {                     // keep owner < neighbour: flip the face
                    std::swap(face.owner, face.neighbour);
                    std::reverse(face.verts.begin(), face.verts.end());
                }
                interior.push_back(std::move(face));
                continue;
            }
            face.neighbour = -1;
            FaceKey key{face.verts};
            std::sort(key.v.begin(), key.v.end());
            auto it = open.find(key);
            if (it != open.end() && boundaryPiece[it->second] == p) {
                // two coincident boundary faces INSIDE one piece are a baffle (a zero-thickness wall, legal in
                // OpenFOAM), not a processor patch: both stay boundary faces, particles reflect off either side
                if (boundary[it->second].owner == face.owner) return where + "a cell has the same boundary face twice";
                boundary.push_back(std::move(face));
                boundaryDead.push_back(0);
                boundaryPiece.push_back(p);
            } else if (it == open.end()) {
                open.emplace(std::move(key), boundary.size());
                boundary.push_back(std::move(face));
                boundaryDead.push_back(0);
                boundaryPiece.push_back(p);
            } else {
                // the other side of a processor patch: one interior face, oriented out of the lower cell
                Face& first = boundary[it->second];
                if (first.owner == face.owner) return where + "a cell has the same boundary face twice";
                Face joined = first.owner < face.owner ? first : face;
                joined.neighbour = std::max(first.owner, face.owner);
                joined.owner = std::min(first.owner, face.owner);
                boundaryDead[it->second] = 1;
                open.erase(it);
                interior.push_back(std::move(joined));
            }
        }
        cellBase += m.nCells;
    }
    // upper-triangular order of the interior faces; stable, so equal (owner, neighbour) pairs keep piece order
    std::vector<size_t> order(interior.size());
    std::iota(order.begin(), order.end(), (size_t)0);
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) {
        return interior[x].owner != interior[y].owner ? interior[x].owner < interior[y].owner
                                                      : interior[x].neighbour < interior[y].neighbour;
    });
    out.nInternal = (int64_t)interior.size();
    out.nCells = cellBase;
    out.faceOffsets.push_back(0);
    auto emit = [&](const Face& f) {
        out.faceVerts.insert(out.faceVerts.end(), f.verts.begin(), f.verts.end());
        out.faceOffsets.push_back((int64_t)out.faceVerts.size());
        out.owner.push_back(f.owner);
    };
    for (size_t k : order) { emit(interior[k]); out.neighbour.push_back(interior[k].neighbour); }
    for (size_t k = 0; k < boundary.size(); ++k)
        if (!boundaryDead[k]) emit(boundary[k]);
    return "";
}

thread_local std::string g_mergeError;

}  // namespace

extern "C" {

int cpf_merge_mesh_parts(const cpf_mesh_part* parts, int nParts, cpf_merged_mesh** out) {
    if (!out) return CPF_ERR_ARG;
    *out = nullptr;
    cpf_merged_mesh* m = new (std::nothrow) cpf_merged_mesh;
    if (!m) return CPF_ERR_NOMEM;
    try {
        g_mergeError = merge_impl(parts, nParts, *m);
    } catch (const std::bad_alloc&) {
        delete m;
        return CPF_ERR_NOMEM;
    }
    if (!g_mergeError.empty()) { delete m; return CPF_ERR_MESH; }
    *out = m;
    return CPF_OK;
}

const char* cpf_merge_last_error(void) { return g_mergeError.c_str(); }

int cpf_merged_mesh_sizes(const cpf_merged_mesh* m, int64_t* nPoints, int64_t* nFaces, int64_t* nFaceVerts,
                          int64_t* nInternalFaces, int64_t* nCells) {
    if (!m) return CPF_ERR_ARG;
    if (nPoints) *nPoints = (int64_t)(m->points.size() / 3);
    if (nFaces) *nFaces = (int64_t)m->owner.size();
    if (nFaceVerts) *nFaceVerts = (int64_t)m->faceVerts.size();
    if (nInternalFaces) *nInternalFaces = m->nInternal;
    if (nCells) *nCells = m->nCells;
    return CPF_OK;
}

int cpf_merged_mesh_copy(const cpf_merged_mesh* m, double* points, int64_t* faceOffsets, int64_t* faceVerts,
                         int64_t* owner, int64_t* neighbour) {
    if (!m) return CPF_ERR_ARG;
    if (points) std::memcpy(points, m->points.data(), m->points.size() * 8);
    if (faceOffsets) std::memcpy(faceOffsets, m->faceOffsets.data(), m->faceOffsets.size() * 8);
    if (faceVerts) std::memcpy(faceVerts, m->faceVerts.data(), m->faceVerts.size() * 8);
    if (owner) std::memcpy(owner, m->owner.data(), m->owner.size() * 8);
    if (neighbour) std::memcpy(neighbour, m->neighbour.data(), m->neighbour.size() * 8);
    return CPF_OK;
}

void cpf_merged_mesh_free(cpf_merged_mesh* m) { delete m; }

int cpf_set_mesh_parts(cpf_context* ctx, const cpf_mesh_part* parts, int nParts) {
    if (!ctx) return CPF_ERR_ARG;
    cpf_merged_mesh* m = nullptr;
    const int r = cpf_merge_mesh_parts(parts, nParts, &m);
    if (r != CPF_OK) {                                         // the reason also becomes the context's last error
        cpf::set_context_error(ctx, cpf_merge_last_error());
        return r;
    }
    const int s = cpf_set_mesh_l64(ctx, m->points.data(), (int64_t)(m->points.size() / 3), m->faceOffsets.data(),
                                   m->faceVerts.data(), (int64_t)m->owner.size(), m->owner.data(), m->neighbour.data(),
                                   m->nInternal, m->nCells);
    cpf_merged_mesh_free(m);
    return s;
}

}  // extern "C"
