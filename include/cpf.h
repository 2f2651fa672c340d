/* cpf.h -- C-ABI of libcudaParticleAdvection.so (MI355X / gfx950 build).
 *
 * Drop-in boundary for the per-timestep particle advect+locate loop of
 * simzero/cudaParticlesFoam.  The reference has no C interface: its OpenFOAM solvers
 * #include two header fragments (src/initCuda.H, src/advect.H) that call C++ functions of
 * `namespace advect` in libcudaParticleAdvection (declared in third_party/RTXAdvect/
 * cuda/common.h, query/ConvexQuery.h, query/RTQuery.h) with raw CUDA device pointers.
 * Each entry point below names the reference interface it replaces (paths relative to the
 * reference root).  Header-only C++ shims with the reference's own names sit on top of this
 * ABI in cudaparticlesfoam_amd/compat/ (see INTEGRATION.md).
 *
 * Conventions
 *   - plain pointers and sizes only; all functions return a cpf_status (0 = ok) and never
 *     exit() the process (the reference's cudaCheck does, cuda/cudaHelpers.cuh:32-40);
 *     cpf_last_error() gives the message.
 *   - "host" pointers are ordinary memory; "dev" pointers are HIP device memory.
 *   - labels are 32-bit (OpenFOAM WM_LABEL_SIZE=32); *_l64 variants take 64-bit labels.
 *   - particle state is fp64 (reference: Particle = double4, cuda/common.h:26).
 *   - cell ids: >= 0 containing cell; CPF_CELL_LOST (-1) left the domain / still on a wall
 *     after 5 reflections (reference tetID -1); CPF_CELL_FROZEN (-2) == reference w = 0
 *     (cuda/particles.cu:333-338).  The reference's tetID maps to cell = tetID / 12
 *     (src/initCuda.H:64,99-105).
 */
#ifndef CPF_H
#define CPF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPF_ABI_VERSION 1
#define CPF_CELL_LOST (-1)
#define CPF_CELL_FROZEN (-2)

typedef struct cpf_context cpf_context;

typedef enum cpf_status {
    CPF_OK = 0,
    CPF_ERR_ARG = 1,    /* null / out-of-range argument */
    CPF_ERR_STATE = 2,  /* call order: mesh / velocity / particles not set yet */
    CPF_ERR_MESH = 3,   /* inconsistent polyMesh (owner/neighbour/face lists) */
    CPF_ERR_HIP = 4,    /* HIP runtime error (no device, allocation, launch) */
    CPF_ERR_NOMEM = 5,
    CPF_WARN_NAN = 6    /* NOT a failure: the frame writers' "total kinetic energy is NaN" (the reference stops in
                           system("pause") there, cuda/utils.cpp:255); the frame has been written / queued like any other.
                           Nothing else returns it, so a caller may ignore exactly this code */
} cpf_status;

/* flags for cpf_step / cpf_step_device */
#define CPF_STEP_DEFAULT 0u
#define CPF_STEP_NO_REFLECT 1u    /* reflectWall=false (src/initCuda.H:67): wall hit => CPF_CELL_LOST */
#define CPF_STEP_STORE_VEL 2u     /* also write per-particle velocity (d_particle_vels, for the VTU writer); a particle the call
                                     does not step (lost, frozen) has velocity 0 in that frame -- the reference's array keeps
                                     the velocity of its last advect */
#define CPF_STEP_FUSE_CYCLES 4u   /* run all nCycles inside ONE launch, particle state kept in registers
                                     (legal because U is frozen during the loop, src/advect.H:86) */
#define CPF_STEP_VERTEX_VELOCITY 8u /* advect with the velocity INTERPOLATED at the particle's position from vertex velocities
                                     (cudaAdvect(..., "VertexVelocity"), cuda/particles.cu:244-313, 428-437) instead of the
                                     cell-constant one: needs cpf_set_tets + cpf_set_vertex_velocity; the same five stages in
                                     one launch of the generic walk, equal to the staged calls bit for bit */

/* ---------------------------------------------------------------------------------------------
 * context
 * ------------------------------------------------------------------------------------------- */
int cpf_abi_version(void);
/* hipSetDevice(device) + stream/event creation.  Replaces the implicit CUDA context of the
 * reference (single GPU, default stream: SURVEY.md fact 4). */
int cpf_create(int device, cpf_context** out);
int cpf_destroy(cpf_context* ctx);
/* message of the last failing call on ctx (ctx == NULL: last cpf_create failure). Never NULL. */
const char* cpf_last_error(const cpf_context* ctx);
/* Run all subsequent work on a caller-owned hipStream_t (e.g. the framework's current stream).
 * NULL is a valid handle: HIP's default (null) stream.  cpf_use_own_stream() goes back to the
 * context's private non-blocking stream (the default after cpf_create). */
int cpf_set_stream(cpf_context* ctx, void* hip_stream);
int cpf_use_own_stream(cpf_context* ctx);
int cpf_synchronize(cpf_context* ctx);   /* replaces cudaDeviceSynchronize() after every wrapper */

/* ---------------------------------------------------------------------------------------------
 * mesh + velocity
 * ------------------------------------------------------------------------------------------- */
/* ---------------------------------------------------------------------------------------------
 * rank-direct ingest of a decomposed mesh (replaces src/initCuda.H:207-371: gather-to-master of points, cell
 * centres and 12 tets per cell, coincident points merged by a linear search per point)
 * ------------------------------------------------------------------------------------------- */
/* One rank's piece of the polyMesh, exactly as that rank's fvMesh holds it: processor patches are ordinary
 * boundary faces of the piece.  labelBytes = sizeof(Foam::label) (4 or 8). */
typedef struct cpf_mesh_part {
    const double* points;      int64_t nPoints;          /* [nPoints][3] */
    const void* faceOffsets;   const void* faceVerts;    /* [nFaces+1], flattened vertex lists */
    int64_t nFaces;
    const void* owner;         const void* neighbour;    /* [nFaces], [nInternalFaces] */
    int64_t nInternalFaces;    int64_t nCells;
    int labelBytes;
} cpf_mesh_part;
typedef struct cpf_merged_mesh cpf_merged_mesh;
/* Stitches the pieces (in rank order) into the global mesh on the host; no GPU involved.  Points that coincide
 * exactly become one (the reference's criterion, :341), two boundary faces of different pieces on the same points
 * become one interior face, global cell id = cells of the preceding pieces + local id (globalIndex, :229-230; so
 * the concatenated U slices are the global U), interior faces come out in upper-triangular order.
 * CPF_ERR_MESH with the reason in cpf_merge_last_error(). */
int cpf_merge_mesh_parts(const cpf_mesh_part* parts, int nParts, cpf_merged_mesh** out);
const char* cpf_merge_last_error(void);
int cpf_merged_mesh_sizes(const cpf_merged_mesh* m, int64_t* nPoints, int64_t* nFaces, int64_t* nFaceVerts,
                          int64_t* nInternalFaces, int64_t* nCells);
/* 64-bit labels out; any pointer may be NULL */
int cpf_merged_mesh_copy(const cpf_merged_mesh* m, double* points, int64_t* faceOffsets, int64_t* faceVerts,
                         int64_t* owner, int64_t* neighbour);
void cpf_merged_mesh_free(cpf_merged_mesh* m);
/* cpf_merge_mesh_parts + cpf_set_mesh_l64 */
int cpf_set_mesh_parts(cpf_context* ctx, const cpf_mesh_part* parts, int nParts);

/* Hand over the polyMesh.  Replaces src/initCuda.H:76-130 (polyMeshTetDecomposition loop,
 * HostTetMesh::getBoundaryMesh -- cuda/HostTetMesh.h:307-430 --, DeviceTetMesh::upload --
 * cuda/DeviceTetMesh.cuh:59-72) and the OptiX BVH build (src/initCuda.H:132-139).
 *   points      [nPoints][3]   mesh.points()            (zero-copy from pointField)
 *   faceOffsets [nFaces+1], faceVerts [faceOffsets[nFaces]]   mesh.faces() flattened
 *   owner [nFaces], neighbour [nInternal]               mesh.faceOwner()/faceNeighbour()
 * Builds on the host: CSR cell -> face-slot table in mesh.cells() order, per-slot inward unit
 * plane (n, d) and neighbour id, a uniform bin grid for the initial locate; uploads them.
 * One slot per distinct PLANE of a cell: coplanar faces of a cell (the pieces of a face split by a 2:1
 * refinement next door) share a slot; where they lead to different cells the slot's neighbour id names
 * a face group (cpf_get_mesh_groups) and the walk picks the piece at the exit point.  Hex meshes -- all
 * the reference can run (src/initCuda.H:64) -- have no coplanar faces: slots == faces there.
 * NO REFERENCE COUNTERPART: trajectories on meshes with face groups (2:1 refinement interfaces) or with cells that are
 * not hexes follow this library's own rule for such cells -- leave a group slot only outward, enter the piece whose cell
 * holds the exit point best (csrc/cpf_walk.h) -- stated twice, here and in oracle/cellwalk.c, and checked by the property
 * "every particle lies inside the cell it claims"; the reference cannot run such meshes, so no reference fixture pins them.
 * The same holds for two conventions that are exact on the reference's meshes: components of a unit face normal <= 1e-12
 * are stored as zero, and with the Brownian kick on a one-cell-thick mesh the end point is mirrored about the front / back
 * plane before the walk (option "z_fold"). */
int cpf_set_mesh(cpf_context* ctx, const double* points, int64_t nPoints, const int32_t* faceOffsets,
                 const int32_t* faceVerts, int64_t nFaces, const int32_t* owner, const int32_t* neighbour,
                 int64_t nInternal, int64_t nCells);
int cpf_set_mesh_l64(cpf_context* ctx, const double* points, int64_t nPoints, const int64_t* faceOffsets,
                     const int64_t* faceVerts, int64_t nFaces, const int64_t* owner, const int64_t* neighbour,
                     int64_t nInternal, int64_t nCells);
int cpf_mesh_info(const cpf_context* ctx, int64_t* nCells, int64_t* nCellFaceSlots, int64_t* deviceBytes);
/* Copy the host-built tables back out (tests: compared against the oracle's own build). Any
 * pointer may be NULL.  cellOff[nCells+1], planes[slots][4], nbr[slots]. */
int cpf_get_mesh_tables(const cpf_context* ctx, int32_t* cellOff, double* planes, int32_t* nbr);
/* The face groups: nbr == INT32_MIN + 16 + g marks group g, whose pieces lead to the cells
 * groupNbr[groupOff[g] .. groupOff[g+1]).  Any pointer may be NULL; groupOff[nGroups+1], groupNbr[nMembers]. */
int cpf_get_mesh_groups(const cpf_context* ctx, int64_t* nGroups, int64_t* nMembers, int32_t* groupOff, int32_t* groupNbr);

/* What the mesh layer recognised, i.e. which shortcuts of the walk are live (any pointer may be NULL): allHex = six slots per
 * cell and no face groups; zLayered = ... and the two faces of every cell with an exactly z-parallel normal sit in slots 4, 5
 * (one test drops both when nothing moves in z); zThin = ... and they are boundary faces on two common planes (one cell thick:
 * with the Brownian kick the end point is mirrored before the walk, option "z_fold"); mixed = 1 / 2: cell records for a mesh
 * that is not all-hex, without / with header records. */
int cpf_get_mesh_flags(const cpf_context* ctx, int32_t* allHex, int32_t* zLayered, int32_t* zThin, int32_t* mixed);
/* The tables cpf_set_mesh would build and upload, on the host alone: no context, no GPU (a check of the mesh layer on a
 * machine without a device; what an exotic mesh is ingested as).  Call with the array pointers NULL for the sizes, then
 * again with cellOff[nCells+1], planes[nSlots][4], nbr[nSlots], groupOff[nGroups+1], groupNbr[nMembers].  CPF_ERR_MESH
 * for a mesh cpf_set_mesh would refuse. */
int cpf_build_mesh_tables_host(const double* points, int64_t nPoints, const int32_t* faceOffsets, const int32_t* faceVerts,
                               int64_t nFaces, const int32_t* owner, const int32_t* neighbour, int64_t nInternal, int64_t nCells,
                               int64_t* nSlots, int64_t* nGroups, int64_t* nMembers, int32_t* cellOff, double* planes,
                               int32_t* nbr, int32_t* groupOff, int32_t* groupNbr);
/* ... and the flags cpf_get_mesh_flags would report for it (same meaning; any pointer may be NULL). */
int cpf_mesh_flags_host(const double* points, int64_t nPoints, const int32_t* faceOffsets, const int32_t* faceVerts,
                        int64_t nFaces, const int32_t* owner, const int32_t* neighbour, int64_t nInternal, int64_t nCells,
                        int32_t* allHex, int32_t* zLayered, int32_t* zThin, int32_t* mixed);
/* ... and its BOX RECORDS (NO REFERENCE COUNTERPART: the reference walks 12 tets per cell whatever the cell's shape,
 * src/initCuda.H:64).  *isBox = 1 if every cell is an axis-aligned box -- six planes with normals exactly +-e_x, +-e_y, +-e_z,
 * blockMesh cases such as the TJunction tutorial -- and boxRec (may be NULL) then receives [nCells][16] doubles: the 128-byte
 * records the streaming kernel walks on such a mesh instead of the 256-byte ones (layout and the bit-exactness argument:
 * csrc/cpf_walk.h "box records"; option "box_records" 0 turns them off -- same results, bit for bit).  U (doubles 10..12) is
 * zero here: the device copy gets it from cpf_set_velocity. */
int cpf_mesh_box_records_host(const double* points, int64_t nPoints, const int32_t* faceOffsets, const int32_t* faceVerts,
                              int64_t nFaces, const int32_t* owner, const int32_t* neighbour, int64_t nInternal, int64_t nCells,
                              int32_t* isBox, double* boxRec);

/* Cell-constant velocity U[nCells][3] (host, zero-copy from U.primitiveField()).  Replaces the
 * 12x replication loop + cudaUpdateVelocity of src/advect.H:44-57 (cuda/particles.cu:718-749):
 * nCells*24 B cross PCIe instead of 12*nCells*24 B. */
int cpf_set_velocity(cpf_context* ctx, const double* U, int64_t nCells);
/* same from device memory (async on the context stream) */
int cpf_set_velocity_dev(cpf_context* ctx, const double* dU, int64_t nCells);

/* ---------------------------------------------------------------------------------------------
 * context-owned particle cloud (what the fragments use)
 * ------------------------------------------------------------------------------------------- */
/* cudaMalloc block of src/initCuda.H:141-150 */
int cpf_alloc_particles(cpf_context* ctx, int64_t capacity);
/* cudaInitParticles (cuda/particles.cu:78-108): LCG<16> seeded (i%128, i/128), three draws,
 * pos = lower + r * (upper - lower), w = 1.  The reference leaves the draw->axis assignment to
 * unspecified argument evaluation order; order=0: x,y,z = draws 1,2,3; order=1: draws 3,2,1. */
int cpf_seed_box(cpf_context* ctx, int64_t n, const double lower[3], const double upper[3], int order);
/* inject positions (host [n][3]) and optionally cells (host [n], NULL = unknown, run locate). */
int cpf_set_particles(cpf_context* ctx, int64_t n, const double* xyz, const int32_t* cell);
/* Initial point location.  Replaces RTQuery(OptixQuery&, DeviceTetMesh, double4*, int*, int)
 * (query/RTQuery.cu:295-310: OptiX ray query + baryQuery fix-up) and cudaReportParticles
 * (cuda/particles.cu:763-775).  Contract: cell = lowest-numbered cell whose every face plane has
 * the point on its inner side, CPF_CELL_LOST if none.  nOutside (nullable) = #particles outside. */
int cpf_locate_initial(cpf_context* ctx, int64_t* nOutside);
/* nCycles Lagrangian sub-steps of length dt: the loop body of src/advect.H:86-184 in the
 * ConvexPoly build = cudaAdvect (particles.cu:403-448) -> cudaBrownianMotion (:577-599) ->
 * convexTetQuery (query/ConvexQuery.cu:218-234) -> convexWallReflect (:438-458) ->
 * cudaMoveParticles (particles.cu:706-716), fused into one kernel launch per cycle.
 * D = diffusionCoeff (0 adds exactly nothing, as in the reference).
 * Validity domain of one cycle: a segment is followed through at most 50 CELLS (the reference: 50 TETS of its
 * 12-per-cell decomposition, query/ConvexQuery.cu:169, i.e. about a dozen cells) and at most 5 wall reflections
 * (:353); where both give up, both keep the particle in the last cell reached, but for steps that long (more than
 * ~12 cells per cycle) the two stop in different cells.  The tutorials move a particle 1-3 cells per cycle.
 * dt == 0 with D == 0 is the frame-0 idiom (src/initCuda.H:184-201): nothing moves, CPF_STEP_STORE_VEL stores
 * U[cell], out-of-domain particles become inactive, and the call does not count as a step of the run (the
 * counter-based Brownian stream and the sort cadence do not advance). */
int cpf_step(cpf_context* ctx, double dt, double D, int nCycles, unsigned flags);
/* Reorder the cloud by containing cell (coalesced mesh reads); ids travel with the particles. */
int cpf_sort_by_cell(cpf_context* ctx);
int cpf_num_particles(const cpf_context* ctx, int64_t* n);
/* D2H for output, in ORIGINAL particle-id order.  Replaces the cudaMemcpy block of
 * writeParticles2VTU (cuda/utils.cpp:144-283).  xyzw [n][4] (w = 1 active, 0 frozen),
 * cell [n], vel [n][4] (w = -1 like cuda/particles.cu:361); any may be NULL. */
int cpf_get_particles(cpf_context* ctx, double* xyzw, int32_t* cell, double* vel);
/* cumulative counters since creation (accumulated only while option "stats" is 1): particle-steps done, cells visited,
 * wall reflections, lost */
int cpf_get_counters(cpf_context* ctx, int64_t out[4]);
/* Seed of the counter-based Brownian stream (Philox keyed by seed, particle id, step); default 1591593751, the
 * constant the reference seeds cuRAND with (cuda/particles.cu:544). */
int cpf_set_seed(cpf_context* ctx, uint32_t seed);
/* tuning knobs, never semantics (every step variant is bit-identical):
 *   "step_variant"  -1 (default): 4 wherever the mesh has cell records, fused launches (CPF_STEP_FUSE_CYCLES) of any length
 *                     included (until round 4 launches fusing 8 or more cycles went to 3, which was a few per cent faster
 *                     per cycle there then)
 *                   0 generic CSR walk (any polyhedral mesh; runs by itself when more than a quarter of the cells have
 *                     more than six faces, or with option "mixed_records" 0)
 *                   3 wave-cooperative LDS cell cache on packed 256-byte cell records, one block per 128 particles
 *                     (all-hex meshes)
 *                   4 streaming kernel: persistent waves, next tile prefetched into LDS while the current
 *                     one is walked, per-wave record cache kept across tiles (cudaparticlesfoam_amd/csrc/cpf_stream.hip);
 *                     all-hex meshes, and meshes with a minority of other cells ("mixed_records")
 *                   1, 2, 5 experiments, measured SLOWER than 3 / 4 on every mesh (docs/design_r04.md 5.4): per-lane gathers,
 *                     + scalar plane fetches, run-ahead lanes (cpf_ahead.hip).  Only in libraries built with
 *                     `make EXPERIMENTS=1`; the default build answers CPF_ERR_ARG
 *   "z_fold" (1)    with the Brownian kick on a mesh that is one cell thick in z, mirror the kicked end point about the front /
 *                   back plane before the walk instead of at the hit (same trajectory, fewer cell visits; 0 = the reference's
 *                   order of operations, what the staged entry points always use)
 *   "vertex_fast" (1) the "VertexVelocity" advect finds the particle's tet by a cone test about the cell's apex and evaluates
 *                   that ONE tet, instead of all tetsPerCell -- the same result, bit for bit, whenever the tet holds the particle
 *                   with a margin (else all tets, as with 0).  Only on decompositions that are fans about one apex per cell
 *                   covering every direction once (checked at cpf_set_tets; the reference's 12 tets per hex are);
 *                   cpf_step_kernel_name says which of the two runs
 *   "mixed_records" (1; set BEFORE cpf_set_mesh) on a mesh that is not all-hex, build cell records anyway if at most a
 *                   quarter of the cells have more than TWELVE distinct planes: cells with fewer than six get padded records,
 *                   cells with seven to twelve a second record (a visit there takes two rounds of kernel 4, both tests from
 *                   LDS), cells beyond that a header record and are walked through the CSR tables inside kernel 4.
 *                   0 = such meshes run the generic walk (kernel 0)
 *   "flat_walk"     (1) a 2-D case -- mesh extruded straight in z (z faces exactly along z, every other face with nz == 0 exactly:
 *                   both tutorials' pitzDaily), velocity field without a z component (noted on the device while the field is
 *                   laid out; after cpf_set_velocity_dev the note arrives asynchronously and the field counts as having one
 *                   until it has), D == 0, at least 8 particles per cell -- runs kernel 4's FLAT instantiations: four side faces with two-term dot products, no z faces, no z in the
 *                   walk.  0 = never.  Bit-identical either way (csrc/cpf_walk.h "flat walk", tests/test_gpu_parity.py)
 *   "box_records"   (1) on a mesh whose cells are ALL axis-aligned boxes (cpf_mesh_box_records_host; 2:1-refined boxes with their
 *                   face groups included) kernel 4 walks 128-byte box
 *                   records -- three candidate faces per visit instead of six -- at every cloud density (measured faster than
 *                   the loop lookup on 256-byte records up to 4 900 particles per cell); 0 = the 256-byte records everywhere.
 *                   Bit-identical either way (tests/test_gpu_box.py)
 *   "stream_tiles_per_chunk" (4; 3 on meshes with few particles per cell), "stream_tail_fraction" (0.1; 0.2), "stream_waves_per_cu" (0 = occupancy query):
 *                   work distribution of variant 4; "stream_lookup" (-1 = by particles per cell, 0 loop over the
 *                   distinct cells of a wave, 1 fixed tag compare, 4 the same for sparse clouds -- fewer than 8 particles
 *                   per cell --: pipelined per-lane record gathers, 6 fixed tag compare + box records where the mesh has
 *                   them (else 1); on meshes that are not all-hex only 0 / 1 apply): how a wave finds its cells
 *                   in its record cache; with "stream_lookup_by_density" 1 (default 0) "particles per cell" means per cell that
 *                   HOLDS particles, as counted by the last sort (the tutorials seed 4e6 particles into 20 000 of TJunction's
 *                   248 000 cells; measured slower there, hence off);
 *                   "stream_debug" is a diagnostic (results are WRONG when non-zero)
 *   "coop_max_cells" test hook: kernel 3 addresses cell records with 32-bit byte offsets and is not used for meshes of
 *                   more than 2^24 cells (kernel 4 runs instead); a smaller limit exercises that switch on small meshes
 *   "sort_method"   (2) the (key, index) sort behind cpf_sort_by_cell*: 2 = this library's stable radix sort with 8-bit digits and an
 *                   LDS tile reorder (csrc/cpf_kernels.hip, rt_sort_pairs), 0 = hipcub::DeviceRadixSort.  The same order either way,
 *                   particle for particle (tests).  Measured, round 5, 1e7 particles sorted 25 cycles of diffusion ago: 0.54 / 0.64
 *                   ms per sort -- 0.30 ms of it building the keys and gathering the particle records, which both share; on
 *                   TJunction's 24 key bits (4e6 particles) 0.24 / 0.27 ms
 *   "sort_curve"    (-1) the sort's major key: 0 = the cell id, 1 = the cell's rank along a Morton curve through the cell centres
 *                   (built at cpf_set_mesh), -1 = the rank for sparse clouds (fewer than 8 particles per cell: the step kernel is
 *                   bound by record traffic there and a cell's neighbours in all three directions should be close by in the
 *                   sorted cloud), the id otherwise.  Results do not depend on the order
 *   "vtu_binary"    (0) 1: cpf_write_vtu / cpf_write_vtu_async write frames with raw appended arrays (cpf_write_vtu_arrays_binary)
 *                   instead of the reference's ASCII layout; the replacement fragments set it from the dictionary key binaryFrames
 *   "stats"         1 accumulate the cpf_get_counters statistics, 0 (default) skip that work: the reference
 *                   has no such diagnostics, and they cost the step kernel a resident wave (0.16 -> 0.22 ms
 *                   per 1e7-particle launch)
 *   "timing_stride" cpf_timing_enable brackets every k-th step launch only (default 1)
 *   "sort_interval" cpf_step re-sorts the context-owned cloud by cell every N cycles (default 50, 0 = never);
 *                   invisible to callers: cpf_get_particles always answers in particle-id order.  The sort writes
 *                   into a second set of particle arrays that then swap roles with the first: 36 B per particle
 *                   slot more device memory from the first sort on.  With a diffusion coefficient the order decays
 *                   ~5x faster: the replacement fragments (compat/src/initCuda.H) and api.CudaParticles set 25 then,
 *                   bench.py uses 100 (D = 0) */
int cpf_set_option(cpf_context* ctx, const char* key, double value);
/* Name of the kernel instantiation cpf_step / cpf_step_dev launches for this diffusion coefficient and these flags with
 * the current mesh and options (and the particle count of the most recent step launch, which picks the record lookup of
 * the streaming kernel), as a profiler prints it (e.g. "cpf::step_kernel_stream<false, true, false, false, 0>"):
 * lets a benchmark label its roofline with what actually ran. */
int cpf_step_kernel_name(cpf_context* ctx, double D, unsigned flags, char* buf, size_t bufBytes);

/* ---------------------------------------------------------------------------------------------
 * device-array level (framework hosts that own the particle arrays, multi-GPU sharding)
 * SoA fp64 positions x,y,z [n]; cell [n] int32; gid [n] int64 global particle id (nullable:
 * gid = index); vel [n][3] (nullable unless CPF_STEP_STORE_VEL).
 * ------------------------------------------------------------------------------------------- */
int cpf_step_dev(cpf_context* ctx, double* x, double* y, double* z, int32_t* cell, const int64_t* gid,
                 double* vel, int64_t n, double dt, double D, uint32_t step0, int nCycles, unsigned flags);
int cpf_locate_initial_dev(cpf_context* ctx, const double* x, const double* y, const double* z,
                           int32_t* cell, int64_t n);
/* cudaInitParticles on caller arrays; particle ids first..first+n-1 (sharded seeding) */
int cpf_seed_box_dev(cpf_context* ctx, double* x, double* y, double* z, int64_t first, int64_t n,
                     const double lower[3], const double upper[3], int order);
/* perm[n] (int32 scratch, out): stable order by cell; arrays are permuted in place via tmp
 * buffers owned by the context. */
int cpf_sort_by_cell_dev(cpf_context* ctx, double* x, double* y, double* z, int32_t* cell, int64_t* gid,
                         int64_t n);
/* The same order written into a second set of arrays (none may alias its input; gid / ogid nullable together); the
 * inputs are left as they are.  No staging and no copy back: a fifth less time than the in-place form -- for hosts
 * that keep two sets of arrays and swap them (cpf_sort_by_cell does so for the context's own cloud). */
int cpf_sort_by_cell_dev_to(cpf_context* ctx, const double* x, const double* y, const double* z, const int32_t* cell,
                            const int64_t* gid, double* ox, double* oy, double* oz, int32_t* ocell, int64_t* ogid,
                            int64_t n);
/* Multi-GPU hand-off (no reference counterpart: SURVEY.md 8e).  Rank r owns cells
 * [cellLo[r], cellLo[r+1]).  Partitions the n particles into stay / leave: stayers are
 * left in / moved into [0, nStay) (holes left by leavers are filled from the tail, so only
 * O(leavers) particles move; order is not preserved), leavers are written, grouped by destination
 * rank and in index order, into sendbuf as records of CPF_HANDOFF_DOUBLES doubles (x, y, z, cell-as-double,
 * gid-as-double).  counts[nRanks] (device int64) = records per destination; nStay (device).
 * Lost/frozen particles stay where they are.  The slots [nStay, n) are then marked CPF_CELL_LOST, so a
 * caller that learns nStay late (it sits in device memory) may keep stepping [0, n) while the all-to-all is
 * in flight and let the arrivals catch up afterwards (cpf_step_dev on the appended slice, same step0).
 * More leavers than sendCapacity records: the split is ABORTED on the device -- *nStay = -1, counts[] still hold the
 * records per destination (their sum is the capacity needed), and no particle is moved, marked or written: the shard
 * is exactly as before the call, so the caller can enlarge sendbuf and call again (at any later step). */
#define CPF_HANDOFF_DOUBLES 5
int cpf_pack_leavers_dev(cpf_context* ctx, double* x, double* y, double* z, int32_t* cell, int64_t* gid,
                         int64_t n, const int32_t* cellLo_dev, int nRanks, int myRank, double* sendbuf,
                         int64_t sendCapacity, int64_t* counts_dev, int64_t* nStay_dev);
/* Ownership re-cut, step 1: weights_dev[c] = scale * (number of this shard's particles in cell c), c < nCells
 * (device doubles, overwritten; lost/frozen particles are not counted).  scale = this rank's measured cost per
 * particle-step (or 1 for equal-count cuts); the host layer all-reduces the weights over the ranks. */
int cpf_cell_histogram_dev(cpf_context* ctx, const int32_t* cell, int64_t n, double scale, double* weights_dev);
/* step 2: cellLo_dev[0..nRanks] (device int32) = contiguous cell ranges of equal total weight:
 * cellLo[q] = #{ i in 0..nCells : w[0]+...+w[i-1] < total*q/nRanks }, cellLo[0] = 0, cellLo[nRanks] = nCells. */
int cpf_cell_ranges_dev(cpf_context* ctx, const double* weights_dev, int nRanks, int32_t* cellLo_dev);
/* append nRecv received records at index nStay.. of the arrays */
int cpf_unpack_arrivals_dev(cpf_context* ctx, double* x, double* y, double* z, int32_t* cell, int64_t* gid,
                            int64_t nStay, const double* recvbuf, int64_t nRecv);

/* ---------------------------------------------------------------------------------------------
 * the sharded cloud: ONE RANK PER GPU behind this ABI.  Replaces the reference's parallel
 * branch, where every MPI rank gathers to the master and the master alone drives one GPU
 * (src/initCuda.H:207-484, src/advect.H:59-89): here every rank of `mpirun -np N <solver> -parallel` creates its
 * own context on its own device, joins one communicator and owns the particles of a contiguous cell range; the
 * mesh and U are replicated.  The layer below runs the whole hand-off -- re-cut (per-cell histogram -> all-reduce
 * -> equal-cost cuts), split, ONE all-gather of the counts, ONE all-to-all-v of the leavers on a side stream while
 * the step loop runs on, append, catch-up replay of the cycles the arrivals missed, growth / overflow retry --
 * so that a host only calls cpf_shard_step.  cudaparticlesfoam_amd/parallel.py is a binding of these calls.
 * ------------------------------------------------------------------------------------------- */
#define CPF_MAX_RANKS 64             /* ranks of one communicator (the split kernels' per-destination counters) */
#define CPF_COMM_ID_BYTES 128        /* == NCCL_UNIQUE_ID_BYTES */
#define CPF_COMM_RCCL 1              /* kinds for cpf_comm_unique_id */
#define CPF_COMM_INPROCESS 2
/* A communicator is a table of three collectives on DEVICE memory, asynchronous on the hipStream_t they are given
 * (every rank calls the same collective in the same order, like MPI).  The library ships two implementations
 * (cpf_comm_create) and accepts any other: a host that would rather route the hand-off through its own transport
 * fills the table itself (the CPU tests do, over gloo).  Offsets and sizes are BYTES. */
typedef struct cpf_comm {
    void* self;
    int rank, nRanks;
    /* recv[r * bytesPerRank ...] = rank r's send[0 .. bytesPerRank) */
    int (*all_gather)(void* self, const void* send, void* recv, size_t bytesPerRank, void* stream);
    /* buf[i] = sum over ranks of buf[i], in place, every rank gets the same bits */
    int (*all_reduce_sum_f64)(void* self, double* buf, size_t count, void* stream);
    /* send[sendOff[r] .. + sendBytes[r]) goes to rank r; recv[recvOff[r] .. + recvBytes[r]) comes from rank r;
     * arrays of nRanks entries in host memory, valid only during the call; a pair with zero bytes is skipped */
    int (*all_to_all_v)(void* self, const void* send, const int64_t* sendOff, const int64_t* sendBytes, void* recv,
                        const int64_t* recvOff, const int64_t* recvBytes, void* stream);
    void (*destroy)(void* self);     /* nullable */
    const char* (*last_error)(void* self);   /* nullable */
} cpf_comm;
/* Number of HIP devices visible to this process (a rank picks device = local rank % count). */
int cpf_device_count(int* count);
/* The rendezvous token of a new communicator: ONE rank calls this and the host broadcasts the bytes to the others
 * over whatever it has (Pstream::scatter, MPI_Bcast, a torch store).  kind CPF_COMM_RCCL: ncclGetUniqueId -- one
 * process per GPU, or one thread per GPU, over xGMI.  kind CPF_COMM_INPROCESS: the ranks are threads of THIS
 * process (any devices, also the same one -- which RCCL refuses): device-to-device copies and a barrier, for
 * single-process hosts and for tests.  kind 0: the environment variable CPF_COMM ("rccl" / "inprocess"), else RCCL. */
int cpf_comm_unique_id(void* id /* [CPF_COMM_ID_BYTES] */, int kind);
/* What kind 0 means in this process: CPF_COMM_INPROCESS if the environment says CPF_COMM=inprocess, else CPF_COMM_RCCL.
 * (A host deciding whether its ranks may share a device asks this: RCCL refuses two ranks on one GPU.) */
int cpf_comm_default_kind(void);
/* Joins the communicator `id` names as rank `rank` of `nRanks` on HIP device `device` (collective: returns when
 * every rank has joined).  RCCL is loaded at run time (librccl.so.1): a single-GPU host needs none. */
int cpf_comm_create(const void* id, int rank, int nRanks, int device, cpf_comm** out);
void cpf_comm_destroy(cpf_comm* comm);
const char* cpf_comm_last_error(void);      /* of the calling thread's last failing cpf_comm_* call */

typedef struct cpf_shard cpf_shard;
/* This rank's shard on ctx (mesh and velocity set on ctx as for a single GPU -- every rank holds the WHOLE mesh).
 * comm == NULL: one rank, no collectives.  capacity: particle slots to start with (the arrays grow when arrivals
 * need more).  cellLo[nRanks + 1] (nullable: equal cell counts) = the initial ownership ranges.
 * The shard borrows ctx and comm: destroy the shard first. */
int cpf_shard_create(cpf_context* ctx, const cpf_comm* comm, int64_t capacity, const int32_t* cellLo, cpf_shard** out);
int cpf_shard_destroy(cpf_shard* s);
const char* cpf_shard_last_error(const cpf_shard* s);
/* options (set the same on every rank):
 *   "exchange_interval"  (0) hand particles to their owners every N cycles with FIXED ranges (0 = only inside re-cuts)
 *   "rebalance_interval" (0) re-cut the ranges to equal cost + hand-off every N cycles (0 = never)
 *   "overlap_steps"      (0) cycles the step loop runs on between a split and its exchange; -1 = derived from the
 *                        measured host time of a hand-off, agreed between the ranks through the counts table
 *   "sort_interval"      (0) re-sort the shard by cell every N cycles
 *   "balance_by_time"    (0) cuts of equal MEASURED step time instead of equal particle counts
 *   "send_fraction"      (0.25) initial send-buffer size as a fraction of the capacity (it grows on overflow)
 *   "force_collectives"  (0) run the hand-off path even with one rank (smoke of the N > 1 code on one GPU)
 *   "profile_comm"       (0) keep device time stamps around every hand-off's collectives (cpf_shard_get_stats) */
int cpf_shard_set_option(cpf_shard* s, const char* key, double value);
/* Fill the shard from device arrays (copied; cell == NULL: located here; gid == NULL: first, first+1, ...). */
int cpf_shard_set_particles_dev(cpf_shard* s, const double* x, const double* y, const double* z, const int32_t* cell,
                                const int64_t* gid, int64_t n, int64_t firstGid);
/* cudaInitParticles + RTQuery + cudaReportParticles (src/initCuda.H:152-183) for the sharded cloud: rank r seeds the
 * particle ids [r*nTotal/N, (r+1)*nTotal/N) of the SAME LCG stream a single GPU would draw (cpf_seed_box), locates
 * them, the ranges are cut to equal particle counts and every particle goes to its owner.  nOutside (nullable):
 * out-of-domain particles of the whole cloud. */
int cpf_shard_seed_box(cpf_shard* s, int64_t nTotal, const double lower[3], const double upper[3], int order,
                       int64_t* nOutside);
/* nCycles Lagrangian cycles of this rank's particles (cpf_step's contract), with the hand-offs, re-cuts and sorts
 * that fall due in between.  COLLECTIVE: every rank calls it with the same arguments.  With CPF_STEP_STORE_VEL the
 * velocities of the last cycle stay aligned with the particles until the next call (whatever falls due on that
 * cycle runs at the start of the next call instead; a particle that is not stepped -- lost or frozen -- has velocity 0 in
 * such a frame, as with cpf_step).  With CPF_STEP_FUSE_CYCLES the cycles up to the next thing that
 * falls due (a sort, a hand-off, a re-cut, the completion of the hand-off in flight, the end of the call) run inside
 * one launch, as in cpf_step; without it every cycle is a launch.  Same results either way, bit for bit. */
int cpf_shard_step(cpf_shard* s, double dt, double D, int nCycles, unsigned flags);
int cpf_shard_flush(cpf_shard* s);       /* completes a hand-off still in flight (arrivals appended and caught up) */
int cpf_shard_exchange(cpf_shard* s);    /* one synchronous hand-off with the current ranges */
int cpf_shard_rebalance(cpf_shard* s);   /* re-cut + hand-off, synchronously */
int cpf_shard_sort(cpf_shard* s);
/* New cell velocities (transient solvers, src/advect.H:44-84).  A hand-off in flight is completed first: its
 * arrivals replay the cycles they missed with the field those cycles were stepped with.
 * cpf_shard_set_velocity: the whole field from host memory (every rank holds it).
 * cpf_shard_set_velocity_slice: THIS rank's slice -- the cells of its piece of the decomposed mesh, in the order
 * of cpf_set_mesh_parts (global cell id = cells of the lower ranks + local id) -- 24 B per local cell cross PCIe,
 * the slices are all-gathered between the GPUs over the communicator (the reference gathers them to the master
 * over MPI and pushes 288 B per cell, src/advect.H:59-84). */
int cpf_shard_set_velocity(cpf_shard* s, const double* U, int64_t nCells);
int cpf_shard_set_velocity_slice(cpf_shard* s, const double* Uslice, int64_t nLocalCells);
/* device pointers of the shard's arrays and its particle count (any pointer may be NULL); valid until the next call
 * that may grow, sort or exchange.  A hand-off in flight is completed first. */
int cpf_shard_arrays(cpf_shard* s, double** x, double** y, double** z, int32_t** cell, int64_t** gid, int64_t* n,
                     int64_t* capacity);
/* this rank's particles on the host (arrays of cpf_shard_arrays' n entries; any may be NULL) */
int cpf_shard_get_local(cpf_shard* s, int64_t* gid, double* x, double* y, double* z, int32_t* cell);
int cpf_shard_global_count(cpf_shard* s, int64_t* nGlobal);       /* collective */
int cpf_shard_cell_ranges(cpf_shard* s, int32_t* cellLo /* [nRanks + 1] */);
/* COLLECTIVE: the whole cloud in particle-id order on rank `root` (ids must be 0 .. nGlobal-1, as seeded):
 * xyzw [nGlobal][4], cell [nGlobal], vel [nGlobal][4] as cpf_get_particles; other ranks pass NULLs. */
int cpf_shard_gather(cpf_shard* s, int root, double* xyzw, int32_t* cell, double* vel);
/* COLLECTIVE: the frame writer of cpf_write_vtu_async for the sharded cloud (option "vtu_binary" of the context): the cloud is
 * gathered to `root`'s GPU in particle-id order (pack, counts all-gather, all-to-all-v, scatter: on the compute stream) and the
 * call returns; the copy to pinned host memory (a stream of its own), the energy sum, formatting and file I/O run on a worker
 * thread of the root.  totalKE NULL: no rank waits for PCIe; non-NULL: the root waits for the copy and the sum and returns the
 * energy (other ranks: 0).  One frame is in flight: the next call, cpf_shard_write_vtu_wait or cpf_shard_destroy waits for it
 * and reports its failure (ids that are not 0 .. nGlobal-1: CPF_ERR_STATE, no file). */
int cpf_shard_write_vtu(cpf_shard* s, int root, const char* path, double* totalKE);
int cpf_shard_write_vtu_wait(cpf_shard* s);
typedef struct cpf_shard_stats {
    int64_t n, capacity, stepIndex;
    int64_t particleSteps, handedOff, exchanges, rebalances, grown, sendGrown;
    int64_t kernelLaunches;          /* step launches whose device time the balancer drained ... */
    double kernelMs;                 /* ... and their summed time */
    double handoffHostMs, handoffWaitMs, hostWorkMsPerHandoff;
    double commDeviceMs;             /* "profile_comm": device time of the hand-offs' collectives so far */
    int64_t commEvents;
    int32_t overlapDepth;            /* cycles the next hand-off will overlap */
    int32_t nRanks, rank;
} cpf_shard_stats;
int cpf_shard_get_stats(cpf_shard* s, cpf_shard_stats* out);

/* ---------------------------------------------------------------------------------------------
 * stage-by-stage entry points on the REFERENCE's array layouts, for hosts that keep the
 * reference's five-call cycle (the compat shims advect::cudaAdvect(...) etc. forward here).
 * particles/disps/vels: device [n][4] doubles (Particle / vec4d), ids: device int32 [n] holding
 * CELL ids (reference: tet ids).  Between locate and reflect a wall hit is encoded exactly like
 * the reference: ids[i] = -(cell at the start of the step + 1).
 * ------------------------------------------------------------------------------------------- */
/* device memory helpers so a C++ host needs no HIP headers (the reference's fragments call
 * cudaMalloc/cudaMemset/cudaMemcpy directly, src/initCuda.H:141-150) */
int cpf_dev_alloc(cpf_context* ctx, size_t bytes, void** out);
int cpf_dev_free(cpf_context* ctx, void* ptr);
int cpf_dev_memset(cpf_context* ctx, void* ptr, int value, size_t bytes);
int cpf_copy_to_device(cpf_context* ctx, void* dst_dev, const void* src_host, size_t bytes);
int cpf_copy_to_host(cpf_context* ctx, void* dst_host, const void* src_dev, size_t bytes);
int cpf_copy_dev(cpf_context* ctx, void* dst_dev, const void* src_dev, size_t bytes);    /* device to device, asynchronous on the context's stream */
/* cudaInitParticles (cuda/particles.cu:100-108) */
int cpf_stage_seed_box(cpf_context* ctx, double* particles, int64_t n, const double lower[3], const double upper[3],
                       int order);
/* RTQuery(OptixQuery&, DeviceTetMesh, double4*, int*, int) (query/RTQuery.cu:295-310) */
int cpf_stage_locate_initial(cpf_context* ctx, const double* particles, int32_t* ids, int64_t n);
/* cudaReportParticles (cuda/particles.cu:763-775): number of negative ids */
int cpf_stage_count_outside(cpf_context* ctx, const int32_t* ids, int64_t n, int64_t* nNegative);
/* cudaAdvect, "TetVelocity" mode (cuda/particles.cu:403-448 -> :316-373) */
int cpf_stage_advect(cpf_context* ctx, double* particles, const int32_t* ids, double* vels, double* disps, double dt,
                     int64_t n);
/* cudaAdvect, "VertexVelocity" mode (cuda/particles.cu:428-437 -> particleAdvectKernel, :244-313): velocity at P =
 * barycentric interpolation of VERTEX velocities in the tet that holds P.  The walk needs no tets, this mode does:
 * cpf_set_tets hands over the decomposition the reference's fragment builds (src/initCuda.H:86-124: tetsPerCell tets
 * per cell in cell order, vertex ids into positions = mesh.points() ++ mesh.C()), replacing DeviceTetMesh::upload of
 * d_positions / d_indices (cuda/DeviceTetMesh.cuh:59-72); cpf_set_vertex_velocity replaces d_velocities for this mode
 * (one vector per tet-mesh vertex).  ids are CELL ids: the kernel takes the tet of that cell whose smallest barycentric
 * weight of P is largest (the interpolant is continuous across the tets of a cell), then weighs exactly like the
 * reference (w_X = det(tet with X := P) * (1 / det(tet)); vel = wA*velA + wB*velB + wC*velC + wD*velD).  Option
 * "vertex_fast": that tet is found without evaluating the others where that is provably the same thing. */
int cpf_set_tets(cpf_context* ctx, const double* positions, int64_t nVerts, const int32_t* tets, int64_t nTets,
                 int tetsPerCell);
int cpf_set_vertex_velocity(cpf_context* ctx, const double* vertexU, int64_t nVerts);
int cpf_stage_advect_vertex(cpf_context* ctx, double* particles, const int32_t* ids, double* vels, double* disps,
                            double dt, int64_t n);
/* cudaAdvect, "ConstantVelocity" mode (cuda/particles.cu:439-445 -> particleAdvectConstVel, :376-399): every particle
 * keeps the velocity ALREADY in vels -- disps = (vel.x*dt, vel.y*dt, vel.z*dt, -1); a particle whose id is negative is
 * switched off (w = 0) like in the other modes.  No caller in the reference's src/ uses this mode. */
int cpf_stage_advect_const(cpf_context* ctx, double* particles, const int32_t* ids, const double* vels, double* disps,
                           double dt, int64_t n);
/* cudaBrownianMotion (cuda/particles.cu:577-599); step selects the counter-based stream */
int cpf_stage_brownian(cpf_context* ctx, const double* particles, double* disps, double dt, int64_t n, double D,
                       uint32_t step);
/* convexTetQuery (query/ConvexQuery.cu:218-234) */
int cpf_stage_locate(cpf_context* ctx, const double* particles, const double* disps, int32_t* ids, int64_t n);
/* convexWallReflect (query/ConvexQuery.cu:438-458) */
int cpf_stage_reflect(cpf_context* ctx, int32_t* ids, double* particles, double* vels, double* disps, int64_t n);
/* cudaMoveParticles (cuda/particles.cu:706-716) */
int cpf_stage_move(cpf_context* ctx, double* particles, double* disps, int64_t n);

/* ---------------------------------------------------------------------------------------------
 * output (the next component either side of the path: SURVEY.md 8f #1)
 * ------------------------------------------------------------------------------------------- */
/* writeParticles2VTU (cuda/utils.cpp:144-283): D2H of the context-owned cloud in particle-id
 * order + ASCII particle_%04d.vtu layout.  totalKE (nullable) = "System Kinetic Energy". */
int cpf_write_vtu(cpf_context* ctx, const char* path, double* totalKE);
/* The same frame, written behind the caller's back.  The call snapshots the cloud ON THE DEVICE -- one kernel on the context's
 * stream that packs it in particle-id order into a second buffer -- and returns; the device-to-host copy (pinned memory, a stream
 * of its own, behind an event), the energy sum, formatting and file I/O -- 1e5 particles: 0.1 s, three orders of magnitude more
 * than the GPU needs for the cycles between two frames -- belong to a worker thread, and the step loop's stream never waits for
 * PCIe (round 6: a 1e7-particle frame used to hold the loop for 20 ms, now for the launch of one kernel).
 * totalKE NULL: returns at once.  totalKE non-NULL (a host that prints the energy where the reference does, cuda/utils.cpp:250):
 * the call waits until the copy has arrived and been summed -- the frame is still written behind the caller's back.
 * One frame is in flight per context: the next call (or cpf_write_vtu_wait, or cpf_destroy) waits for it and reports its status
 * (CPF_WARN_NAN included when the energy was not asked for here). */
int cpf_write_vtu_async(cpf_context* ctx, const char* path, double* totalKE);
int cpf_write_vtu_wait(cpf_context* ctx);
/* same formatter on host arrays: xyzw [n][4], cell [n], vel [n][4] */
int cpf_write_vtu_arrays(const char* path, int64_t n, const double* xyzw, const int32_t* cell, const double* vel,
                         double* totalKE);
/* The same frame with its arrays RAW behind the XML ("binary-appended as an option"): the same DataArrays --
 * names, types, components, order -- with format='appended' and offsets into one <AppendedData encoding='raw'> section (per
 * array a UInt64 byte count, then little-endian bytes).  Not the reference's bytes (its writer is ASCII only, cuda/utils.cpp:
 * 144-283), the same data: positions exact instead of 15 decimals.  cpf_set_option(ctx, "vtu_binary", 1) makes cpf_write_vtu /
 * cpf_write_vtu_async write this form. */
int cpf_write_vtu_arrays_binary(const char* path, int64_t n, const double* xyzw, const int32_t* cell, const double* vel,
                                double* totalKE);

/* Trajectory collection and its two writers: addToTrajectories / saveTrajectories / writeStreamline2VTK
 * (cuda/utils.cpp:7-94, cuda/common.h:87-92; called from src/advect.H:163-175 behind saveStreamlinetoFile).  A trajectory is
 * the list of fp32 positions a particle had at the sampling instants at which it was active (w != 0); trajectories with fewer
 * than two points are left out of both files.  Both files byte for byte as the reference writes them. */
typedef struct cpf_traj cpf_traj;
int cpf_traj_create(cpf_traj** out);
void cpf_traj_destroy(cpf_traj* t);
/* one sample of every active particle: of the context-owned cloud (particle-id order) / of the reference's AoS particle
 * array in device memory (Particle = double4, [n][4]) / of a host array [n][4].  The first sample fixes the particle count. */
int cpf_traj_add(cpf_context* ctx, cpf_traj* t);
int cpf_traj_add_stage(cpf_context* ctx, cpf_traj* t, const double* particles_dev, int64_t n);
int cpf_traj_add_host(cpf_traj* t, const double* xyzw, int64_t n);
int cpf_traj_sizes(const cpf_traj* t, int64_t* nTrajectories, int64_t* nPoints);
int cpf_traj_save_obj(const cpf_traj* t, const char* path);       /* saveTrajectories: Wavefront OBJ, "v" + "l" lines */
int cpf_traj_write_vtk(const cpf_traj* t, const char* path);      /* writeStreamline2VTK: legacy VTK POLYDATA poly-lines */
/* the same writers on a host's own storage: trajectory k = points offsets[k] .. offsets[k+1] of xyz[][3] */
int cpf_traj_save_obj_arrays(const char* path, int64_t nTrajectories, const int64_t* offsets, const float* xyz);
int cpf_traj_write_vtk_arrays(const char* path, int64_t nTrajectories, const int64_t* offsets, const float* xyz);

/* ---------------------------------------------------------------------------------------------
 * measurement
 * ------------------------------------------------------------------------------------------- */
/* When enabled, every step-kernel launch (every k-th with cpf_set_option "timing_stride" k: an event pair costs
 * ~5 us of idle GPU between two back-to-back launches) is bracketed by a hipEvent pair on the launch stream. */
int cpf_timing_enable(cpf_context* ctx, int on);
/* Drains the recorded pairs: number of launches and their summed device time in ms. */
int cpf_timing_read(cpf_context* ctx, int64_t* launches, double* total_ms);
/* Same, but never waits: drains only the launches that have already finished (possibly none).  Feeds the
 * multi-GPU load balancer with this rank's measured step time without stalling the launch queue. */
int cpf_timing_poll(cpf_context* ctx, int64_t* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* CPF_H */
