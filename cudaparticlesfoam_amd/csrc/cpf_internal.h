// Internal declarations shared by the host-side mesh ingest and the HIP translation unit.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "cpf.h"

namespace cpf {

// neighbour code of face group g; boundary codes -(face + 1) stay above -(2^30), group codes and the walk's few special
// values (INT32_MIN ... INT32_MIN + 15) below
constexpr int32_t kGroupBase = INT32_MIN + 16;

// Host-built connectivity, laid out exactly as it is uploaded (DESIGN.md "Data layout in HBM").
struct HostTables {
    int64_t nCells = 0, nSlots = 0;
    // one slot per distinct PLANE of a cell (cpf_mesh.cpp: coplanar faces of a cell share a slot)
    std::vector<int32_t> cellOff;   // [nCells+1]   CSR offsets, slot order = mesh.cells()[c] (z-layered meshes: z faces last, see zPairLast)
    std::vector<double> planes;     // [nSlots][4]  unit normal INTO the cell, d = n . faceCentre
    std::vector<int32_t> nbr;       // [nSlots]     neighbour cell, or -(face+1) on the boundary, or kGroupBase + g: face group g
    std::vector<int32_t> groupOff;  // [nGroups+1]  face groups: the cells behind the coplanar pieces of one slot ...
    std::vector<int32_t> groupNbr;  // ... in face order
    // uniform bin grid over the mesh bounding box for the initial locate
    double origin[3] = {0, 0, 0}, invBin[3] = {0, 0, 0}, lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    int32_t dims[3] = {1, 1, 1};
    std::vector<int32_t> binOff;    // [nBins+1]
    std::vector<int32_t> binCells;  // candidate cells per bin, ascending cell id
    int32_t maxCellFaces = 0, minCellFaces = 0;   // slots per cell
    int64_t nBigCells = 0;          // cells with more than 6 slots
    int64_t nHugeCells = 0;         // ... of which: more than 12 (header record + CSR walk; 7..12: two records, cpf_walk.h)
    int64_t nGroups() const { return (int64_t)groupOff.size() - 1; }
    bool zThin = false;             // zPairLast and the two z faces of every cell are boundary faces (one cell thick in z)
    bool zSide0 = false;            // zPairLast and the other four faces of every cell have nz == 0 exactly: a 2-D mesh extruded in z (the flat walk, cpf_walk.h)
    bool zPairLast = false;         // all-hex mesh whose cells each have exactly two faces with an exactly z-parallel normal: they sit in slots 4, 5
    // [nCells][16] BOX RECORDS, or empty: every cell is an axis-aligned box (six planes with normals exactly +-e_x, +-e_y,
    // +-e_z, one of each: blockMesh cases such as the TJunction tutorial) -- 128 bytes per cell instead of the 256-byte
    // record, and a face test that never touches a normal.  Layout and the exactness argument: cpf_walk.h "box records"
    std::vector<double> boxRec;
    std::vector<float> cellBox;     // [nCells][6]  AABB lower corner and 2^subBits/extent per axis (sub-cell sort key)
    std::vector<int32_t> curveRank; // [nCells]     rank of the cell along a Morton curve through the cell boxes' centres (sort key of sparse clouds)
    // sub-cell sort key layout: bits per axis (0 for an axis in which the mesh is one cell thick) and the axes
    // from most to least significant (the longest domain axis first)
    int32_t subBits[3] = {2, 2, 2}, subOrder[3] = {0, 1, 2};
};

// polyMesh -> HostTables.  Returns empty string on success, else the reason (CPF_ERR_MESH).
template <typename Label>
std::string build_tables(const double* points, int64_t nPoints, const Label* faceOff, const Label* faceVerts,
                         int64_t nFaces, const Label* owner, const Label* neighbour, int64_t nInternal,
                         int64_t nCells, HostTables& out);

extern template std::string build_tables<int32_t>(const double*, int64_t, const int32_t*, const int32_t*, int64_t,
                                                  const int32_t*, const int32_t*, int64_t, int64_t, HostTables&);
extern template std::string build_tables<int64_t>(const double*, int64_t, const int64_t*, const int64_t*, int64_t,
                                                  const int64_t*, const int64_t*, int64_t, int64_t, HostTables&);

// stores `message` as the context's last error (cpf_last_error); for entry points implemented outside cpf_api.cpp
void set_context_error(cpf_context* ctx, const char* message);
// what the sharded-cloud layer (cpf_shard.cpp) needs to know about a context it borrows
void* context_stream(const cpf_context* ctx);       // the hipStream_t its kernels are launched on
int context_device(const cpf_context* ctx);
bool context_timing(const cpf_context* ctx);        // cpf_timing_enable state
int64_t context_cells(const cpf_context* ctx);      // 0: no mesh yet
bool vtu_binary(const cpf_context* ctx);            // option "vtu_binary"

}  // namespace cpf
