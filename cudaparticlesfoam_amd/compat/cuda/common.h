// Compat shim: the names of third_party/RTXAdvect/cuda/common.h (reference @ v1) on top of the
// C-ABI of libcudaParticleAdvection.so (include/cpf.h).  Header-only, plain C++14, no HIP/CUDA
// headers needed by the including solver.  Written from scratch: nothing here is the reference's
// code, only its call surface (function names, argument order and meaning):
//
//   reference declaration (cuda/common.h)            forwards to
//   ------------------------------------------------ ---------------------------
//   cudaInitParticles(Particle*, int, box3d)   :32   cpf_stage_seed_box
//   cudaAdvect(..., "TetVelocity")             :45   cpf_stage_advect
//   initRandomGenerator / cudaBrownianMotion   :64-71 cpf_set_seed / cpf_stage_brownian
//   cudaMoveParticles(Particle*, vec4d*, int, int*) :76 cpf_stage_move
//   cudaReportParticles(int, int*)             :79   cpf_stage_count_outside
//   cudaUpdateVelocity(std::vector<vec3d>, ...) :81  cpf_set_velocity  (one vec3d per CELL here)
//   writeParticles2VTU(...)                    :93   cpf_copy_to_host + cpf_write_vtu_arrays
//
// "ids" are CELL ids (the reference's tet ids / 12).  Device pointers come from deviceAlloc().
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "cpf.h"

namespace advect {

struct vec3d { double x, y, z; vec3d() : x(0), y(0), z(0) {} vec3d(double a, double b, double c) : x(a), y(b), z(c) {} };
struct vec4d { double x, y, z, w; };
struct vec4i { int x, y, z, w; };
struct vec3f { float x, y, z; };
struct box3d {
    vec3d lower, upper;
    box3d() {}
    box3d(const vec3d& lo, const vec3d& hi) : lower(lo), upper(hi) {}
    vec3d size() const { return vec3d(upper.x - lower.x, upper.y - lower.y, upper.z - lower.z); }
};
struct double4 { double x, y, z, w; };
typedef double4 Particle;                      // cuda/common.h:26

// The reference exit()s on any CUDA error (cuda/cudaHelpers.cuh:32-40); the shims throw instead,
// carrying cpf_last_error(); a solver that wants the old behaviour catches and exits.
struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string& m) : std::runtime_error(m), status(s) {}
};
inline void check(cpf_context* ctx, int status) {
    if (status != CPF_OK) throw Error(status, std::string("libcudaParticleAdvection: ") + cpf_last_error(ctx));
}

// Replaces curandState_t* rand_states (48 B/particle in the reference): the stream is counter based.
struct RandomState { std::uint32_t step = 0; };
typedef RandomState curandState_t;

// Stand-in for the struct of raw device pointers the reference passes by value
// (cuda/DeviceTetMesh.cuh:26-37): same member names so call sites read the same, but the mesh lives
// inside the context.  d_indices/d_positions/d_velocities are opaque tokens, never dereferenced.
struct DeviceTetMesh {
    cpf_context* ctx = nullptr;
    vec4i* d_indices = nullptr;
    vec3d* d_positions = nullptr;
    vec3d* d_velocities = nullptr;
    std::int64_t nCells = 0;
    // polyMesh hand-over (replaces HostTetMesh fill + upload, src/initCuda.H:76-130)
    template <typename Label>
    void upload(const double* points, std::int64_t nPoints, const Label* faceOffsets, const Label* faceVerts,
                std::int64_t nFaces, const Label* owner, const Label* neighbour, std::int64_t nInternal,
                std::int64_t nCellsIn) {
        static_assert(sizeof(Label) == 4 || sizeof(Label) == 8, "label must be 32 or 64 bit");
        if (!ctx) check(nullptr, cpf_create(0, &ctx));
        if (sizeof(Label) == 4)
            check(ctx, cpf_set_mesh(ctx, points, nPoints, (const std::int32_t*)faceOffsets, (const std::int32_t*)faceVerts,
                                    nFaces, (const std::int32_t*)owner, (const std::int32_t*)neighbour, nInternal, nCellsIn));
        else
            check(ctx, cpf_set_mesh_l64(ctx, points, nPoints, (const std::int64_t*)faceOffsets, (const std::int64_t*)faceVerts,
                                        nFaces, (const std::int64_t*)owner, (const std::int64_t*)neighbour, nInternal, nCellsIn));
        nCells = nCellsIn;
    }
};

template <typename T>
inline T* deviceAlloc(const DeviceTetMesh& m, std::size_t count, int fillByte = 0) {
    void* p = nullptr;
    check(m.ctx, cpf_dev_alloc(m.ctx, count * sizeof(T), &p));
    check(m.ctx, cpf_dev_memset(m.ctx, p, fillByte, count * sizeof(T)));
    return static_cast<T*>(p);
}
inline void deviceFree(const DeviceTetMesh& m, void* p) { check(m.ctx, cpf_dev_free(m.ctx, p)); }
inline void deviceSynchronize(const DeviceTetMesh& m) { check(m.ctx, cpf_synchronize(m.ctx)); }

// The current mesh is remembered so that the reference's context-free signatures keep working.
inline DeviceTetMesh& currentMesh() { static DeviceTetMesh m; return m; }
inline void bindMesh(const DeviceTetMesh& m) { currentMesh() = m; }

inline void cudaInitParticles(Particle* d_particles, int N, const box3d& worldBounds, int order = 1) {
    const double lo[3] = {worldBounds.lower.x, worldBounds.lower.y, worldBounds.lower.z};
    const double hi[3] = {worldBounds.upper.x, worldBounds.upper.y, worldBounds.upper.z};
    cpf_context* c = currentMesh().ctx;
    check(c, cpf_stage_seed_box(c, &d_particles->x, N, lo, hi, order));
    check(c, cpf_synchronize(c));
}

inline void cudaAdvect(Particle* d_particles, int* d_tetIDs, vec4d* d_vels, vec4d* d_disp, double dt,
                       int numParticles, vec4i* /*d_tetIndices*/, vec3d* /*d_vertexPositions*/,
                       vec3d* /*d_velocities*/, std::string mode = "TetVelocity") {
    cpf_context* c = currentMesh().ctx;
    if (mode == "TetVelocity") {
        check(c, cpf_stage_advect(c, &d_particles->x, d_tetIDs, &d_vels->x, &d_disp->x, dt, numParticles));
    } else if (mode == "VertexVelocity") {
        // needs the decomposition and the vertex velocities in the library: cpf_set_tets + cpf_set_vertex_velocity
        check(c, cpf_stage_advect_vertex(c, &d_particles->x, d_tetIDs, &d_vels->x, &d_disp->x, dt, numParticles));
    } else if (mode == "ConstantVelocity") {
        check(c, cpf_stage_advect_const(c, &d_particles->x, d_tetIDs, &d_vels->x, &d_disp->x, dt, numParticles));
    } else {
        // (the reference silently does nothing for any other string, cuda/particles.cu:417-445: a typo there is a particle
        // cloud that never moves; here it is an error)
        throw Error(CPF_ERR_ARG, "cudaAdvect: mode must be \"TetVelocity\", \"VertexVelocity\" or \"ConstantVelocity\" (cuda/particles.cu:417-445)");
    }
    check(c, cpf_synchronize(c));
}

inline void initRandomGenerator(int /*numParticles*/, curandState_t* rand_states) {
    if (rand_states) rand_states->step = 0;
    check(currentMesh().ctx, cpf_set_seed(currentMesh().ctx, 1591593751u));   // cuda/particles.cu:544
}

inline void cudaBrownianMotion(Particle* d_particles, vec4d* d_disp, curandState_t* states, double dt,
                               int numParticles, double diffusionCoeff) {
    cpf_context* c = currentMesh().ctx;
    const std::uint32_t step = states ? states->step++ : 0u;
    check(c, cpf_stage_brownian(c, &d_particles->x, &d_disp->x, dt, numParticles, diffusionCoeff, step));
    check(c, cpf_synchronize(c));
}

inline void cudaMoveParticles(Particle* d_particles, vec4d* d_disps, int numParticles, int* /*d_tetIDs*/) {
    cpf_context* c = currentMesh().ctx;
    check(c, cpf_stage_move(c, &d_particles->x, &d_disps->x, numParticles));
    check(c, cpf_synchronize(c));
}

inline void cudaReportParticles(int numParticles, int* d_tetIDs) {
    cpf_context* c = currentMesh().ctx;
    std::int64_t bad = 0;
    check(c, cpf_stage_count_outside(c, d_tetIDs, numParticles, &bad));
    std::printf("#adv: Out-of-domain particles(-tetID) = %lld\n", (long long)bad);
}

// One velocity per CELL (the reference wants 12 copies per cell: src/advect.H:44-54).
inline void cudaUpdateVelocity(const std::vector<vec3d>& cellVelocities, int numCells, vec4i* /*d_tetIndices*/,
                               vec3d* /*d_Velocities*/) {
    cpf_context* c = currentMesh().ctx;
    check(c, cpf_set_velocity(c, &cellVelocities[0].x, numCells));
}

inline void writeParticles2VTU(unsigned int ti, Particle* d_particles, vec4d* d_vels, int* /*d_tetIDs*/,
                               int numParticles, int* d_tetIDs_Convex = nullptr) {
    cpf_context* c = currentMesh().ctx;
    std::vector<double> P((std::size_t)numParticles * 4), V((std::size_t)numParticles * 4);
    std::vector<std::int32_t> ids((std::size_t)numParticles, -1);
    check(c, cpf_copy_to_host(c, P.data(), d_particles, P.size() * 8));
    check(c, cpf_copy_to_host(c, V.data(), d_vels, V.size() * 8));
    if (d_tetIDs_Convex) check(c, cpf_copy_to_host(c, ids.data(), d_tetIDs_Convex, ids.size() * 4));
    char name[64];
    std::snprintf(name, sizeof(name), "particle_%04u.vtu", ti);
    double ke = 0.0;
    const int r = cpf_write_vtu_arrays(name, numParticles, P.data(), ids.data(), V.data(), &ke);
    if (r != CPF_OK && r != CPF_WARN_NAN) throw Error(r, std::string("writeParticles2VTU: cannot write ") + name);
    std::printf("#adv: System Kinetic Energy=%lf\n", ke);
}

// Trajectory collection and its two writers (cuda/common.h:87-92, cuda/utils.cpp:7-94): in the reference's fragments they sit
// behind `saveStreamlinetoFile`, which src/initCuda.H:68 hard-codes to false.  Same signatures: the samples live in the
// caller's vectors; the device-to-host copy and the two file formats are the library's (cpf_traj_*, csrc/cpf_traj.cpp).
inline void addToTrajectories(Particle* d_particles, int numParticles, std::vector<std::vector<vec3f>>& trajectories) {
    cpf_context* c = currentMesh().ctx;
    if (trajectories.empty()) trajectories.resize((std::size_t)numParticles);
    std::vector<double> P((std::size_t)numParticles * 4);
    check(c, cpf_copy_to_host(c, P.data(), d_particles, P.size() * 8));
    for (int i = 0; i < numParticles; ++i) {
        if (!P[4 * (std::size_t)i + 3]) continue;                    // inactive particle: no sample (cuda/utils.cpp:21)
        trajectories[(std::size_t)i].push_back(vec3f{(float)P[4 * (std::size_t)i], (float)P[4 * (std::size_t)i + 1], (float)P[4 * (std::size_t)i + 2]});
    }
}
namespace detail {
inline void flattenTrajectories(const std::vector<std::vector<vec3f>>& trajectories, std::vector<std::int64_t>& off, std::vector<float>& xyz) {
    off.assign(trajectories.size() + 1, 0);
    std::int64_t total = 0;
    for (std::size_t t = 0; t < trajectories.size(); ++t) { off[t] = total; total += (std::int64_t)trajectories[t].size(); }
    off[trajectories.size()] = total;
    xyz.reserve((std::size_t)total * 3);
    for (const auto& traj : trajectories)
        for (const auto& p : traj) { xyz.push_back(p.x); xyz.push_back(p.y); xyz.push_back(p.z); }
}
}  // namespace detail
inline void saveTrajectories(const std::string& fileName, std::vector<std::vector<vec3f>>& trajectories) {
    std::vector<std::int64_t> off; std::vector<float> xyz;
    detail::flattenTrajectories(trajectories, off, xyz);
    const int r = cpf_traj_save_obj_arrays(fileName.c_str(), (std::int64_t)trajectories.size(), off.data(), xyz.data());
    if (r != CPF_OK) throw Error(r, "saveTrajectories: cannot write " + fileName);
}
inline void writeStreamline2VTK(const std::string& fileName, std::vector<std::vector<vec3f>>& trajectories) {
    std::vector<std::int64_t> off; std::vector<float> xyz;
    detail::flattenTrajectories(trajectories, off, xyz);
    const int r = cpf_traj_write_vtk_arrays(fileName.c_str(), (std::int64_t)trajectories.size(), off.data(), xyz.data());
    if (r != CPF_OK) throw Error(r, "writeStreamline2VTK: cannot write " + fileName);
}

inline std::string prettyNumber(std::size_t s) {
    char buf[64];
    const double v = (double)s;
    if (v >= 1e9) std::snprintf(buf, sizeof buf, "%.2fG", v / 1e9);
    else if (v >= 1e6) std::snprintf(buf, sizeof buf, "%.2fM", v / 1e6);
    else if (v >= 1e3) std::snprintf(buf, sizeof buf, "%.2fK", v / 1e3);
    else std::snprintf(buf, sizeof buf, "%zu", s);
    return buf;
}

}  // namespace advect
