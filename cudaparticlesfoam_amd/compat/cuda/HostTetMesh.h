// Compat: third_party/RTXAdvect/cuda/HostTetMesh.h.  There is no host tet mesh in this build: the
// polyMesh is handed over as it is (DeviceTetMesh::upload in cuda/common.h).  The type exists so that
// `HostTetMesh* hostTetMesh = new HostTetMesh; ... delete hostTetMesh;` (src/initCuda.H:33,205) compiles.
#pragma once
#include "cuda/common.h"
namespace advect {
struct HostTetMesh {
    box3d worldBounds;
};
}  // namespace advect
