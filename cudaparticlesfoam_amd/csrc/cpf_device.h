// Device-side views and launcher declarations shared by the kernels and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "cpf.h"
#include "cpf_internal.h"   // kGroupBase

namespace cpf {

constexpr int kBlock = 256;   // 4 waves of 64
constexpr int kCounterSlots = 1024;   // statistics counters are sharded over this many 32-byte slots

// Mesh as the kernels see it (all arrays resident in HBM, L2-resident for tutorial-size meshes)
struct MeshView {
    const int32_t* cellOff;   // [nCells+1]
    const double4* planes;    // [nSlots]   (nx, ny, nz, d), unit normal into the cell
    const int32_t* nbr;       // [nSlots]   neighbour cell | -(face + 1) on the boundary | kGroupBase + g: face group g (cpf_mesh.cpp)
    const int32_t* groupOff;  // [nGroups+1] face groups: the cells behind the coplanar pieces of one slot, in face order
    const int32_t* groupNbr;
    const double4* U;         // [nCells]   cell-constant velocity, w unused (32-B aligned gathers)
    const double4* cellRec;   // [nCells][8] packed 256-B records (null: generic walk only); layout: cpf_walk.h "cell records"
    const double4* boxRec;    // [nCells][4] 128-B box records (null: some cell is not an axis-aligned box); cpf_walk.h "box records"
    int32_t nCells;
    int32_t allHex;           // every cell has exactly 6 face slots (slot = 6*cell + s) and there are no face groups
    int32_t zPairLast;        // ... and slots 4, 5 of every cell are its two faces with an exactly z-parallel normal (cpf_mesh.cpp)
    int32_t zSide0;           // ... and the four other faces of every cell have nz == 0 exactly (2-D mesh extruded straight in z)
    int32_t zThin;            // ... and both are boundary faces in every cell: one cell thick in z (cpf_walk.h, fold_z)
    int32_t mixed;            // records exist although the mesh is not all-hex: 1 = padded records (< 6 slots) and face groups only, 2 = header-only records (> 6 slots) as well
};

struct GridView {
    double origin[3], invBin[3], lo[3], hi[3];
    int32_t dims[3];
    const int32_t* binOff;
    const int32_t* binCells;
};

// Host-side state of the streaming step kernel (cpf_stream.hip): two sets of per-group chunk counters (a launch
// uses one and zeroes the other for the launch after it on the same stream) and the tuning knobs.
constexpr size_t kStreamGrabBytes = 2 * 256 * 16 * sizeof(unsigned);
constexpr int kStreamHitSpillDoubles = 3 * 64;          // x[64] | y[64] | z[64] per wave
struct StreamState {
    unsigned* d_grab = nullptr;
    // timing of ONE launch (cpf_timing_enable): set by the caller before launch_step, taken (and cleared) by the streaming
    // launcher, which hands them to hipExtLaunchKernelGGL -- start and stop are then the dispatch's own begin / end time
    // stamps, what rocprofv3's kernel trace reports, instead of events recorded around the launch (which also time the gap
    // an event record puts between two otherwise back-to-back kernels: 5 % on a 0.12 ms kernel)
    hipEvent_t evStart = nullptr, evStop = nullptr;
    double* d_hitSpill = nullptr;   // wall hit points that do not fit a wave's LDS pool: kStreamHitSpillDoubles per wave slot (cpf_stream.hip)
    int hitSpillWaves = 0;          // wave slots d_hitSpill has room for (the launcher never starts more single-wave workgroups)
    int parity = 0;
    int numCU = 256;
    int tilesPerChunk = 0;    // "stream_tiles_per_chunk"; 0 = by lookup method: 4 (loop lookup), 3 (fixed lookup: slower tiles, finer dealing)
    int wavesPerCU = 0;       // "stream_waves_per_cu": 0 = what the occupancy query says
    int coopMaxCells = 0;     // "coop_max_cells": test hook, lowers the wave-cooperative kernel's 2^24-cell limit (0 = the limit)
    int lookup = -1;          // "stream_lookup": 0 loop over distinct cells, 1 fixed tag compare, -1 = by particles per cell
    double tailFraction = -1.0; // "stream_tail_fraction": share of the cloud dealt tile by tile at the end of a launch; < 0 = by lookup method: 0.1 / 0.2
    int debug = 0;            // "stream_debug": diagnostics only (1 = no stores, 2 = no loads; results are wrong)
    // What the last sort found: [0] cells that hold particles, [1] live particles (pinned host memory, written by an async copy
    // behind every sort; null / zeros = not known).  With "stream_lookup_by_density" 1 the lookup method goes by particles per
    // OCCUPIED cell instead of per cell of the whole mesh: the tutorials seed their clouds in a small box (TJunction: 4e6
    // particles in 20 000 of 248 000 cells -- 200 per cell, not 16).  Read without synchronisation: an old value only costs
    // a launch or two on the other -- bit-identical -- instantiation.  OFF by default, measured (round 4, one box): TJunction
    // as its dictionary runs it, D = 1.5e-5, takes 0.150 ms per step with the fixed compare the whole-mesh figure picks and
    // 0.177 ms with the loop lookup the density picks -- on a 3-D mesh with the kick a tile's lanes spread over more cells
    // per round than the 128-per-cell threshold, tuned on pitzDaily, assumes.
    const volatile unsigned long long* occupiedHost = nullptr;
    int densityLookup = 0;    // "stream_lookup_by_density"
    // the velocity field last set has no z component anywhere (U.z == +-0 in every cell; found by the kernel that lays the field
    // out, read back behind it): with MeshView::zSide0 and without the kick the FLAT instantiation runs (cpf_walk.h "flat walk")
    bool flatField = false;
    int flat = 1;             // "flat_walk": 0 = never (diagnostics; bit-identical either way)
};

hipError_t launch_step_ahead(hipStream_t st, double* x, double* y, double* z, int32_t* cell, int64_t n, double dt, bool reflect,
                             const MeshView& m, unsigned long long* counters, StreamState& ss, double* dbg);
int stream_lookup_mode(int64_t n, const MeshView& m, const StreamState& ss, bool brown = true);    // 0 loop, 1 fixed compare, 4 fixed compare for sparse clouds, 2 / 3 fixed compare + mixed records with / without header records, 5 loop + mixed records, 6 fixed compare + box records
// the variant launch_step really runs for a requested one (non-hex meshes: generic; record-offset limits)
int effective_step_variant(int variant, const MeshView& m, bool haveStream, int cyclesPerLaunch, int coopMaxCells);
constexpr int kFusedCoopCycles = 1 << 30;  // fused launches of this many cycles or more run the wave-cooperative kernel: never since round 4
                                          // (round 3, 8: the streaming kernel was 4 % faster per cycle at 3 fused cycles, 2 % slower at 8; with
                                          // the flat walk and box records it is 5-30 % faster at 8 and 16 -- cpf_kernels.hip, effective_step_variant)
// ss == nullptr: the streaming variant is not available (falls back to the wave-cooperative kernel)
hipError_t launch_step(hipStream_t st, double* x, double* y, double* z, int32_t* cell, const int64_t* gid,
                       double* vel, int64_t n, double dt, double D, uint32_t step0, int nCyc, uint32_t seed,
                       bool reflect, bool storeVel, const MeshView& m, unsigned long long* counters, int variant,
                       StreamState* ss);
// the fused cycle with the "VertexVelocity" advect mode (generic walk; CPF_STEP_VERTEX_VELOCITY)
hipError_t launch_step_vertex(hipStream_t st, double* x, double* y, double* z, int32_t* cell, const int64_t* gid,
                              double* vel, int64_t n, double dt, double D, uint32_t step0, int nCyc, uint32_t seed,
                              bool reflect, bool storeVel, const MeshView& m, unsigned long long* counters,
                              const double* pos, const int32_t* tets, int tetsPerCell, const double* vertVel, const double* cone,
                              const double* apex, int variant, StreamState* ss);
// does launch_step_vertex stream (step_kernel_stream_vertex) or run step_kernel_vertex, for these arguments?
bool step_vertex_streams(const MeshView& m, const double* cone, int tetsPerCell, int variant, const StreamState* ss, int nCyc);
struct VertexField;                  // cpf_walk.h
// the "VertexVelocity" cycle on the streaming kernel (all-hex meshes with cone-locate tables; else launch_step_vertex's own kernel)
bool stream_vertex_capable(const MeshView& m);
int stream_vertex_lookup_mode(int64_t n, const MeshView& m, const StreamState& ss);
hipError_t launch_step_stream_vertex(hipStream_t st, double* x, double* y, double* z, int32_t* cell, const int64_t* gid,
                                     double* vel, int64_t n, double dt, double sigma, uint32_t step0, int nCyc, uint32_t seed,
                                     bool brown, bool reflect, bool storeVel, const MeshView& m, unsigned long long* counters,
                                     StreamState& ss, const VertexField& vf);
hipError_t launch_step_stream(hipStream_t st, double* x, double* y, double* z, int32_t* cell, const int64_t* gid,
                              double* vel, int64_t n, double dt, double sigma, uint32_t step0, int nCyc, uint32_t seed,
                              bool brown, bool reflect, bool storeVel, const MeshView& m, unsigned long long* counters,
                              StreamState& ss);
hipError_t launch_locate_initial(hipStream_t st, const double* x, const double* y, const double* z, int32_t* cell,
                                 int64_t n, const MeshView& m, const GridView& g);
hipError_t launch_seed_box(hipStream_t st, double* x, double* y, double* z, int64_t first, int64_t n,
                           const double lower[3], const double upper[3], int order);
hipError_t launch_iota64(hipStream_t st, int64_t* p, int64_t n, int64_t first);
hipError_t launch_unpack_xyz(hipStream_t st, const double* xyz, double* x, double* y, double* z, int64_t n);
hipError_t launch_pack_by_gid(hipStream_t st, const double* x, const double* y, const double* z, const int32_t* cell,
                              const int64_t* gid, const double* vel, double* xyzw, int32_t* cellOut, double* velOut,
                              int64_t n);
hipError_t launch_count_negative(hipStream_t st, const int32_t* cell, int64_t n, unsigned long long* out);
hipError_t launch_build_cell_records(hipStream_t st, const double4* planes, const int32_t* nbr, const double4* U,
                                     double4* rec, int64_t nCells);
hipError_t launch_build_cell_records_mixed(hipStream_t st, const int32_t* cellOff, const double4* planes, const int32_t* nbr,
                                           const double4* U, const int32_t* recB, double4* rec, int64_t nCells);
hipError_t launch_update_record_velocity(hipStream_t st, const double4* U, double4* rec, double* box, int64_t nCells);
hipError_t launch_u3_to_u4(hipStream_t st, const double* u3, double4* u4, int64_t nCells, unsigned long long* zFlag = nullptr);

// stage-by-stage kernels on the reference's AoS layouts
hipError_t launch_stage_advect(hipStream_t st, double* P, const int32_t* ids, double* vels, double* disps, double dt,
                               int64_t n, const MeshView& m);
hipError_t launch_stage_advect_const(hipStream_t st, double* P, const int32_t* ids, const double* vels, double* disps, double dt,
                                     int64_t n);
hipError_t launch_stage_advect_vertex(hipStream_t st, double* P, const int32_t* ids, double* vels, double* disps, double dt,
                                      int64_t n, const double* pos, const int32_t* tets, int tetsPerCell,
                                      const double* vertVel, const double* cone, const double* apex);
// tet records of the "VertexVelocity" advect's cone locate: cone[nTets][32], apex[nCells][4] (cpf_walk.h, VertexField); the
// velocity part of the records follows every cpf_set_vertex_velocity
hipError_t launch_vertex_cone_tables(hipStream_t st, const double* pos, const int32_t* tets, int64_t nTets, int tetsPerCell, double* cone,
                                     double* apex);
hipError_t launch_vertex_record_velocity(hipStream_t st, const int32_t* tets, const double* vel, int64_t nTets, double* cone);
hipError_t launch_stage_brownian(hipStream_t st, const double* P, double* disps, double dt, int64_t n, double D,
                                 uint32_t step, uint32_t seed);
hipError_t launch_stage_locate(hipStream_t st, const double* P, const double* disps, int32_t* ids, int64_t n,
                               const MeshView& m);
hipError_t launch_stage_reflect(hipStream_t st, int32_t* ids, double* P, double* vels, double* disps, int64_t n,
                                const MeshView& m);
hipError_t launch_stage_move(hipStream_t st, double* P, double* disps, int64_t n);
hipError_t launch_aos_to_soa(hipStream_t st, const double* P, double* x, double* y, double* z, int64_t n);
hipError_t launch_soa_to_aos(hipStream_t st, const double* x, const double* y, const double* z, double* P, int64_t n);

size_t sort_scratch_bytes(int64_t n, int endBit);
hipError_t sort_by_cell(hipStream_t st, double* x, double* y, double* z, int32_t* cell, int64_t* gid, double* vel3,
                        int64_t n, int endBit, const float* cellBox, const int* subBits, const int* subOrder,
                        void* scratch, size_t scratchBytes, double* ox = nullptr, double* oy = nullptr, double* oz = nullptr,
                        int32_t* ocell = nullptr, int64_t* ogid = nullptr, unsigned long long* occupied = nullptr,
                        const int32_t* rank = nullptr, int method = 1);      // method 0: the library radix sort (the order tests compare with)

// multi-GPU hand-off (cpf_handoff.hip)
size_t handoff_scratch_bytes(int64_t n, int nRanks);
hipError_t pack_leavers(hipStream_t st, double* x, double* y, double* z, int32_t* cell, int64_t* gid, int64_t n,
                        const int32_t* cellLo, int nRanks, int myRank, double* sendbuf, int64_t sendCapacity,
                        int64_t* counts, int64_t* nStay, void* scratch, size_t scratchBytes);
size_t histogram_scratch_bytes(int64_t nCells);
hipError_t cell_histogram(hipStream_t st, const int32_t* cell, int64_t n, int64_t nCells, double scale, double* weights,
                          void* scratch, size_t scratchBytes);
hipError_t cell_ranges(hipStream_t st, const double* weights, int64_t nCells, int nRanks, int32_t* cellLo);
hipError_t unpack_arrivals(hipStream_t st, double* x, double* y, double* z, int32_t* cell, int64_t* gid,
                           int64_t nStay, const double* recvbuf, int64_t nRecv);
// output of a sharded cloud (cpf_shard_gather): records of kOutputDoubles doubles = x, y, z, cell, gid, vx, vy, vz
constexpr int kOutputDoubles = 8;
hipError_t pack_output(hipStream_t st, const double* x, const double* y, const double* z, const int32_t* cell,
                       const int64_t* gid, const double* vel3, double* rec, int64_t n);
hipError_t scatter_output(hipStream_t st, const double* rec, int64_t nRec, int64_t nGlobal, double* xyzw, int32_t* cellOut,
                          double* velOut, unsigned long long* bad);
hipError_t sum_rows(hipStream_t st, const double* rows, int nRows, size_t count, double* out);

}  // namespace cpf
