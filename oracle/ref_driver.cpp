// TEST INFRASTRUCTURE ONLY (oracle/_ref driver).  Never linked into, imported by,
// or called from the product (cudaparticlesfoam_amd/): only tests/, smoke() and
// bench.py's cpu_baseline leg may load the .so this file is built into.
//
// This translation unit wraps the REFERENCE's own device functions (spliced in
// at build time by oracle/build_ref.sh from /root/reference, never stored in
// this repo) behind a flat extern "C" surface so Python/ctypes can drive them
// as if each CUDA kernel were launched with blockDim=128 (the reference launch
// shape, e.g. third_party/RTXAdvect/query/ConvexQuery.cu:220-231).
//
// What is supplied here is ONLY the nvcc language built-ins that g++ lacks.
#include <cmath>
#include <math.h>
#include <stdlib.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <fstream>
#include <sstream>
#include <map>
#include <vector>
#include <algorithm>
#include <string>
#include <cassert>
#ifdef _OPENMP
#include <omp.h>
#endif

#define __device__
#define __global__
#define __host__
struct double4 { double x, y, z, w; };
static inline double4 make_double4(double x, double y, double z, double w) { return double4{x, y, z, w}; }
struct RefDim3 { unsigned x, y, z; };
static thread_local RefDim3 threadIdx, blockIdx, blockDim, gridDim;

// reference headers, included where they lie (see -I flags in build_ref.sh)
#include "cuda/HostTetMesh.h"              // HostTetMesh, FaceInfo, getBoundaryMesh, createBoxMesh
#include "owl/common/math/random.h"        // LCG<16>

using std::isinf;
using std::isnan;

namespace advect {
typedef double4 Particle;
#include "ref_extract.inc"                 // generated into a temp dir by build_ref.sh
#include "ref_traj_writers.inc"            // saveTrajectories, writeStreamline2VTK (cuda/utils.cpp:30-47, 49-94), whole
}  // namespace advect

using namespace advect;

static_assert(sizeof(vec3d) == 24 && sizeof(vec4d) == 32 && sizeof(vec4i) == 16 && sizeof(FaceInfo) == 8,
              "layout assumptions of the flat driver (SURVEY.md Appendix C)");

namespace {
struct CoutSilencer {
    std::streambuf* old;
    std::ostringstream sink;
    CoutSilencer() : old(std::cout.rdbuf(sink.rdbuf())) {}
    ~CoutSilencer() { std::cout.rdbuf(old); }
};

template <typename F>
void launch(int n, int nthreads, F&& body) {
    // emulate <<<ceil(n/128),128>>>: threadIdx.x = i%128, blockIdx.x = i/128
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (int i = 0; i < n; ++i) {
        blockDim.x = 128; blockDim.y = blockDim.z = 1;
        gridDim.x = (unsigned)((n + 127) / 128); gridDim.y = gridDim.z = 1;
        threadIdx.x = (unsigned)(i % 128); threadIdx.y = threadIdx.z = 0;
        blockIdx.x = (unsigned)(i / 128); blockIdx.y = blockIdx.z = 0;
        body();
    }
}
}  // namespace

extern "C" {

// HostTetMesh::getBoundaryMesh (cuda/HostTetMesh.h:307-430): face table of a tet mesh.
// facets_out[4*cap], tetfacets_out[4*nT], faceinfos_out[2*cap]; returns #facets or -1 if cap too small.
int ref_face_table(const double* positions, int nV, const int* tets, int nT,
                   int* facets_out, int* tetfacets_out, int* faceinfos_out, int cap) {
    HostTetMesh m;
    m.positions.resize(nV);
    std::memcpy(m.positions.data(), positions, sizeof(vec3d) * (size_t)nV);
    m.indices.resize(nT);
    std::memcpy(m.indices.data(), tets, sizeof(vec4i) * (size_t)nT);
    {
        CoutSilencer quiet;  // the reference prints every boundary vertex (HostTetMesh.h:389)
        (void)m.getBoundaryMesh();
    }
    int nF = (int)m.facets.size();
    if (nF > cap || (int)m.tetfacets.size() != nT) return -1;
    std::memcpy(facets_out, m.facets.data(), sizeof(vec4i) * (size_t)nF);
    std::memcpy(tetfacets_out, m.tetfacets.data(), sizeof(vec4i) * (size_t)nT);
    std::memcpy(faceinfos_out, m.faceInfos.data(), sizeof(FaceInfo) * (size_t)nF);
    return nF;
}

// HostTetMesh::createBoxMesh (cuda/HostTetMesh.h:62-144); returns sizes through nV/nT, copies if buffers given.
int ref_box_mesh(int nx, int ny, int nz, double* positions, int* tets, int capV, int capT) {
    HostTetMesh m;
    {
        CoutSilencer quiet;
        m = HostTetMesh::createBoxMesh(nx, ny, nz);
    }
    if ((int)m.positions.size() > capV || (int)m.indices.size() > capT) return -1;
    std::memcpy(positions, m.positions.data(), sizeof(vec3d) * m.positions.size());
    std::memcpy(tets, m.indices.data(), sizeof(vec4i) * m.indices.size());
    return (int)m.indices.size();
}

// initParticlesKernel (cuda/particles.cu:78-97)
void ref_init_particles(double* particles, int n, const double* lower, const double* upper, int nthreads) {
    box3d box(vec3d(lower[0], lower[1], lower[2]), vec3d(upper[0], upper[1], upper[2]));
    launch(n, nthreads, [&] { initParticlesKernel((Particle*)particles, n, box); });
}

// particleAdvectKernelTetVel (cuda/particles.cu:316-373)
void ref_advect(double* particles, int* tetIDs, double* vels, double* disps, double dt, int n,
                const int* indices, const double* positions, const double* tetVel, int nthreads) {
    launch(n, nthreads, [&] {
        particleAdvectKernelTetVel((Particle*)particles, tetIDs, (vec4d*)vels, (vec4d*)disps, dt, n,
                                   (vec4i*)indices, (vec3d*)positions, (vec3d*)tetVel);
    });
}

// particleAdvectKernel, the "VertexVelocity" mode (cuda/particles.cu:244-313)
void ref_advect_vertex(double* particles, int* tetIDs, double* vels, double* disps, double dt, int n,
                       const int* indices, const double* positions, const double* vertVel, int nthreads) {
    launch(n, nthreads, [&] {
        particleAdvectKernel((Particle*)particles, tetIDs, (vec4d*)vels, (vec4d*)disps, dt, n,
                             (vec4i*)indices, (vec3d*)positions, (vec3d*)vertVel);
    });
}

// particleLocator (query/ConvexQuery.cu:135-216)
void ref_locate(double* particles, int* tetIDs, double* disps, int n, const int* indices,
                const double* positions, const int* tetfacets, const int* facets, const int* faceinfos,
                int nthreads) {
    launch(n, nthreads, [&] {
        particleLocator((double4*)particles, tetIDs, (vec4d*)disps, n, (vec4i*)indices, (vec3d*)positions,
                        (vec4i*)tetfacets, (vec4i*)facets, (FaceInfo*)faceinfos);
    });
}

// convexReflector (query/ConvexQuery.cu:320-436)
void ref_reflect(double* particles, int* tetIDs, double* disps, double* vels, int n, const int* indices,
                 const double* positions, const int* tetfacets, const int* facets, const int* faceinfos,
                 int nthreads) {
    launch(n, nthreads, [&] {
        convexReflector((double4*)particles, tetIDs, (vec4d*)disps, (vec4d*)vels, n, (vec4i*)indices,
                        (vec3d*)positions, (vec4i*)tetfacets, (vec4i*)facets, (FaceInfo*)faceinfos);
    });
}

// particleMoveKernel, disp variant (cuda/particles.cu:659-704)
void ref_move(double* particles, double* disps, int* tetIDs, int n, int nthreads) {
    launch(n, nthreads, [&] { particleMoveKernel((Particle*)particles, (vec4d*)disps, tetIDs, n); });
}

// baryQuery (query/RTQuery.cu:189-218): the fp64 fix-up half of the initial locate.
void ref_bary_query(double* particles, int* tetIDs, int n, const double* positions, const int* indices,
                    const int* facets, const int* tetfacets, const int* faceinfos, int nthreads) {
    launch(n, nthreads, [&] {
        baryQuery((Particle*)particles, tetIDs, n, (vec3d*)positions, (vec4i*)indices, (vec4i*)facets,
                  (vec4i*)tetfacets, (FaceInfo*)faceinfos);
    });
}

// baryQueryDisp + RTreflection (query/RTQuery.cu:221-248, 109-186): the RTX=true per-step variant,
// kept as a second, independent oracle for the containing-tet decision.
void ref_bary_query_disp(double* particles, double* disps, int* tetIDs, int n, const double* positions,
                         const int* indices, const int* facets, const int* tetfacets, const int* faceinfos,
                         int nthreads) {
    launch(n, nthreads, [&] {
        baryQueryDisp((Particle*)particles, (vec4d*)disps, tetIDs, n, (vec3d*)positions, (vec4i*)indices,
                      (vec4i*)facets, (vec4i*)tetfacets, (FaceInfo*)faceinfos);
    });
}

// Full Lagrangian cycles in the reference's order (src/advect.H:96-161, ConvexPoly build,
// Brownian term omitted == diffusionCoeff 0 which adds exactly 0, cuda/particles.cu:564-569).
// Each "thread" (particle) runs its four kernels `cycles` times back to back: particles are
// independent, so this equals the kernel-by-kernel order and needs a single parallel region.
// Used for golden vectors and as the "reference" CPU baseline.
void ref_cycles(double* particles, int* tetIDs, double* vels, double* disps, double dt, int n, int cycles,
                const int* indices, const double* positions, const double* tetVel, const int* tetfacets,
                const int* facets, const int* faceinfos, int nthreads) {
    launch(n, nthreads, [&] {
        for (int c = 0; c < cycles; ++c) {
            particleAdvectKernelTetVel((Particle*)particles, tetIDs, (vec4d*)vels, (vec4d*)disps, dt, n,
                                       (vec4i*)indices, (vec3d*)positions, (vec3d*)tetVel);
            particleLocator((double4*)particles, tetIDs, (vec4d*)disps, n, (vec4i*)indices, (vec3d*)positions,
                            (vec4i*)tetfacets, (vec4i*)facets, (FaceInfo*)faceinfos);
            convexReflector((double4*)particles, tetIDs, (vec4d*)disps, (vec4d*)vels, n, (vec4i*)indices,
                            (vec3d*)positions, (vec4i*)tetfacets, (vec4i*)facets, (FaceInfo*)faceinfos);
            particleMoveKernel((Particle*)particles, (vec4d*)disps, tetIDs, n);
        }
    });
}

// The reference's VTU writer (cuda/utils.cpp:144-283) on host arrays: writes ./particle_%04d.vtu (ti) in the current
// directory, exactly as the solver does.  Only the device-to-host copies are replaced by the arguments.
int ref_write_vtu(unsigned int ti, const double* particles /* [n][4] */, const double* vels /* [n][4] */,
                  const int* tetIDs, int numParticles, const int* d_tetIDs_Convex) {
    std::vector<Particle> hostParticles(numParticles);
    std::vector<int> h_tetIDs(tetIDs, tetIDs + numParticles);
    std::vector<vec4d> h_vels(numParticles);
    std::memcpy(hostParticles.data(), particles, sizeof(Particle) * (size_t)numParticles);
    std::memcpy(h_vels.data(), vels, sizeof(vec4d) * (size_t)numParticles);
#include "ref_vtu_head.inc"            // utils.cpp:172-217, up to the declaration of h_ctetIDs
            std::memcpy(h_ctetIDs.data(), d_tetIDs_Convex, sizeof(int) * (size_t)numParticles);   // for :218-221
#include "ref_vtu_tail.inc"            // utils.cpp:222-282, through fclose(fp)
    return 0;
}

// The reference's trajectory collection (cuda/utils.cpp:7-28) on a host array -- only its device-to-host copy is replaced
// by the argument -- and its two writers (:30-94) as they stand.
static std::vector<std::vector<vec3f>> g_trajectories;
void ref_traj_reset(void) { g_trajectories.clear(); }
void ref_traj_add(const double* particles /* [n][4] */, int numParticles) {
    std::vector<std::vector<vec3f>>& trajectories = g_trajectories;
    if (trajectories.empty()) trajectories.resize(numParticles);           // :10-11
    std::vector<Particle> hostParticles(numParticles);
    std::memcpy(hostParticles.data(), particles, sizeof(Particle) * (size_t)numParticles);
#include "ref_traj_add.inc"                // utils.cpp:20-27, the sampling loop
}
void ref_traj_save_obj(const char* path) { saveTrajectories(path, g_trajectories); }
void ref_traj_write_vtk(const char* path) { writeStreamline2VTK(path, g_trajectories); }

int ref_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
