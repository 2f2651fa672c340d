// Test harness: a stand-in for applications/cudaParticlesUncoupledFoam/cudaParticlesUncoupledFoam.C
// (reference :40-89) built against the MOCK OpenFOAM types: same include order, same
// `namespace advect { extern "C" int main` shape, same two fragment includes -- but the fragments
// are this repo's replacements.  Usage: mockUncoupledFoam <caseDir> [<eulerianSteps>]; writes particle_*.vtu and
// particles_out.f64 / cells_out.i32 into the current directory.
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

extern "C" int main(int argc, char* argv[])
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s <caseDir>\n", argv[0]); return 2; }
    fvMesh mesh;
    volVectorField U;
    Time runTime;
    IOdictionary cudaParticleAdvectionDict;
    try {
        loadCase(argv[1], mesh, U, runTime, cudaParticleAdvectionDict);

        #include "initCuda.H"

        const int eulerianSteps = argc > 2 ? std::atoi(argv[2]) : 1;
        for (int cpfE = 0; cpfE < eulerianSteps; ++cpfE)
        {
            if (cpfE > 0)
            {
                // a transient solver's next step: time moves on and the field changes (as in mockParallelFoam)
                runTime.t += runTime.dT;
                for (auto& v : U.f.d) v = vector(0.9*v.x(), 0.9*v.y() + 0.01, 0.9*v.z());
            }
            #include "advect.H"
        }

        // harness output: final state in particle-id order
        std::vector<double> xyzw((size_t)numParticles * 4);
        std::vector<int32_t> cells((size_t)numParticles);
        advect::check(cpfCtx, cpf_get_particles(cpfCtx, xyzw.data(), cells.data(), nullptr));
        FILE* fp = std::fopen("particles_out.f64", "wb"); std::fwrite(xyzw.data(), 8, xyzw.size(), fp); std::fclose(fp);
        fp = std::fopen("cells_out.i32", "wb"); std::fwrite(cells.data(), 4, cells.size(), fp); std::fclose(fp);
        cpf_destroy(cpfCtx);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "FATAL: %s\n", e.what());
        return 1;
    }
    Info<< "End\n" << endl;
    return 0;
}

}
