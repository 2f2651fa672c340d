// Compat: third_party/RTXAdvect/optix/OptixQuery.h:66-78.  No OptiX, no BVH: the initial locate runs on
// a bin grid built by cpf_set_mesh.  The class only keeps `OptixQuery tetQueryAccelerator(...)` compiling.
#pragma once
#include <chrono>

#include "cuda/common.h"
namespace advect {
struct double3 { double x, y, z; };
struct int4 { int x, y, z, w; };
class OptixQuery {
public:
    OptixQuery() {}
    OptixQuery(const double3*, int, const int4*, int, bool = false) {}
};
// cuda/cudaHelpers.cuh:44-87 (the fragments only use it around the BVH build, src/initCuda.H:132-138): host wall
// clock in milliseconds between start() and stop() -- what the reference's event pair measures around host-blocking
// calls, and never a silent 0
struct cudaTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void start() { t0 = std::chrono::steady_clock::now(); }
    double stop() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace advect
