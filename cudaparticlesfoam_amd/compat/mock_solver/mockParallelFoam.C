// Test harness: the reference's parallel run shape (mpirun -np N ... -parallel, e.g.
// tutorials/incompressible/cudaParticlesPimpleFoam/TJunction/Allrun-parallel:9-12) against the MOCK OpenFOAM
// types.  ONE process plays the N ranks one after the other -- the non-masters first, the master last -- each with
// the piece of the decomposed case decomposePar would have given it (<caseDir>/processor<r>/), running the same two
// fragment includes as every solver.  mock Pstream::gatherList is the mailbox between them (mock_openfoam/fvCFD.H).
// Usage: mockParallelFoam <caseDir> <nProcs>; output files like mockUncoupledFoam.
#include "cuda/common.h"
#include "cuda/DeviceTetMesh.cuh"
#include "cuda/HostTetMesh.h"
#include "query/ConvexQuery.h"
#include "query/RTQuery.h"
#include "optix/OptixQuery.h"

#include <cstring>

#include "fvCFD.H"
#include "case_io.H"

namespace advect {

static int rankMain(const std::string& caseDir, int rank, int nProcs)
{
    Pstream::simulate(rank, nProcs);
    fvMesh mesh;
    volVectorField U;
    Time runTime;
    IOdictionary cudaParticleAdvectionDict;
    loadCase(caseDir + "/processor" + std::to_string(rank), mesh, U, runTime, cudaParticleAdvectionDict);

    #include "initCuda.H"

    #include "advect.H"

    if (Pstream::master())
    {
        std::vector<double> xyzw((size_t)numParticles * 4);
        std::vector<int32_t> cells((size_t)numParticles);
        advect::check(cpfCtx, cpf_get_particles(cpfCtx, xyzw.data(), cells.data(), nullptr));
        FILE* fp = std::fopen("particles_out.f64", "wb"); std::fwrite(xyzw.data(), 8, xyzw.size(), fp); std::fclose(fp);
        fp = std::fopen("cells_out.i32", "wb"); std::fwrite(cells.data(), 4, cells.size(), fp); std::fclose(fp);
        cpf_destroy(cpfCtx);
    }
    return 0;
}

extern "C" int main(int argc, char* argv[])
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s <caseDir> <nProcs>\n", argv[0]); return 2; }
    const int nProcs = std::atoi(argv[2]);
    try {
        for (int rank = nProcs - 1; rank >= 0; --rank) rankMain(argv[1], rank, nProcs);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "FATAL: %s\n", e.what());
        return 1;
    }
    Info<< "End\n" << endl;
    return 0;
}

}
