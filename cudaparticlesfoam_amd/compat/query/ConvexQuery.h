// Compat: third_party/RTXAdvect/query/ConvexQuery.h:33-46 -- same names and argument order.
#pragma once
#include "cuda/common.h"
namespace advect {
// per-step locate: plane-exit walk from the current cell along disp (query/ConvexQuery.cu:218-234)
inline void convexTetQuery(DeviceTetMesh d_mesh, double4* d_particles, vec4d* d_disps, int* inout_tetIDs,
                           int numParticles) {
    check(d_mesh.ctx, cpf_stage_locate(d_mesh.ctx, &d_particles->x, &d_disps->x, inout_tetIDs, numParticles));
    check(d_mesh.ctx, cpf_synchronize(d_mesh.ctx));
}
// specular wall reflection, up to 5 bounces (query/ConvexQuery.cu:438-458)
inline void convexWallReflect(DeviceTetMesh d_mesh, int* d_tetIDs, Particle* d_particles, vec4d* d_vels,
                              vec4d* d_disps, int numParticles) {
    check(d_mesh.ctx, cpf_stage_reflect(d_mesh.ctx, d_tetIDs, &d_particles->x, &d_vels->x, &d_disps->x, numParticles));
    check(d_mesh.ctx, cpf_synchronize(d_mesh.ctx));
}
}  // namespace advect
