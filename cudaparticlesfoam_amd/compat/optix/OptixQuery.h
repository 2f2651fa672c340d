// Compat: third_party/RTXAdvect/optix/OptixQuery.h:66-78.  No OptiX, no BVH: the initial locate runs on
// a bin grid built by cpf_set_mesh.  The class only keeps `OptixQuery tetQueryAccelerator(...)` compiling.
#pragma once
#include "cuda/common.h"
namespace advect {
struct double3 { double x, y, z; };
struct int4 { int x, y, z, w; };
class OptixQuery {
public:
    OptixQuery() {}
    OptixQuery(const double3*, int, const int4*, int, bool = false) {}
};
struct cudaTimer {                         // cuda/cudaHelpers.cuh:44-87 (only used around the BVH build)
    void start() {}
    double stop() { return 0.0; }
};
}  // namespace advect
