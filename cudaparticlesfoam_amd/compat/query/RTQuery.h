// Compat: third_party/RTXAdvect/query/RTQuery.h:34-63.  Both of the reference's per-step variants
// (RTX=true barycentric walk, RTX=false plane walk) resolve to the plane walk here; the initial
// locate keeps its contract (containing element id, negative if outside).
#pragma once
#include "cuda/common.h"
#include "optix/OptixQuery.h"
#include "query/ConvexQuery.h"
namespace advect {
// initial location (query/RTQuery.cu:295-310)
inline void RTQuery(OptixQuery& /*cellLocator*/, DeviceTetMesh devMesh, double4* d_particles, int* out_tetIDs,
                    int numParticles) {
    check(devMesh.ctx, cpf_stage_locate_initial(devMesh.ctx, &d_particles->x, out_tetIDs, numParticles));
    check(devMesh.ctx, cpf_synchronize(devMesh.ctx));
}
// displacement-mode query (query/RTQuery.cu:335-346)
inline void RTQuery(DeviceTetMesh devMesh, double4* d_particles, vec4d* d_disps, int* out_tetIDs, int numParticles) {
    convexTetQuery(devMesh, d_particles, d_disps, out_tetIDs, numParticles);
}
// note the swapped disps/vels order relative to convexWallReflect (query/RTQuery.h:57-63)
inline void RTWallReflect(DeviceTetMesh devMesh, int* d_tetIDs, Particle* d_particles, vec4d* d_disps, vec4d* d_vels,
                          int numParticles) {
    convexWallReflect(devMesh, d_tetIDs, d_particles, d_vels, d_disps, numParticles);
}
}  // namespace advect
