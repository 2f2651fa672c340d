// Test harness: the reference's parallel run shape (mpirun -np N ... -parallel, e.g.
// tutorials/incompressible/cudaParticlesPimpleFoam/TJunction/Allrun-parallel:9-12) against the MOCK OpenFOAM
// types.  ONE process plays the N ranks as N THREADS, each with the piece of the decomposed case decomposePar would have
// given it (<caseDir>/processor<r>/), running the same two fragment includes as every solver: every rank creates its own
// context and shard, the ranks join one communicator -- in-process here (CPF_COMM=inprocess: N ranks on the ONE GPU of a
// development box, which RCCL refuses), RCCL under a real mpirun -- and hand particles over between them.  mock Pstream
// moves the lists between the threads (mock_openfoam/fvCFD.H).
// Usage: mockParallelFoam <caseDir> <nProcs> [<eulerianSteps>]; output files like mockUncoupledFoam.
#include "cuda/common.h"
#include "cuda/DeviceTetMesh.cuh"
#include "cuda/HostTetMesh.h"
#include "query/ConvexQuery.h"
#include "query/RTQuery.h"
#include "optix/OptixQuery.h"

#include <cstring>
#include <thread>

#include "fvCFD.H"
#include "case_io.H"

namespace advect {

static int rankMain(const std::string& caseDir, int rank, int nProcs, int eulerianSteps)
{
    Pstream::simulate(rank, nProcs);
    fvMesh mesh;
    volVectorField U;
    Time runTime;
    IOdictionary cudaParticleAdvectionDict;
    loadCase(caseDir + "/processor" + std::to_string(rank), mesh, U, runTime, cudaParticleAdvectionDict);

    #include "initCuda.H"

    for (int cpfE = 0; cpfE < eulerianSteps; ++cpfE)
    {
        if (cpfE > 0)
        {
            // a transient solver's next step: time moves on and the field changes (every rank scales its own slice)
            runTime.t += runTime.dT;
            for (auto& v : U.f.d) v = vector(0.9*v.x(), 0.9*v.y() + 0.01, 0.9*v.z());
        }
        #include "advect.H"
    }

    // the whole cloud in particle-id order on the master (collective when the cloud is sharded)
    std::vector<double> xyzw; std::vector<int32_t> cells;
    if (Pstream::master()) { xyzw.resize((size_t)numParticles * 4); cells.resize((size_t)numParticles); }
    if (cpfShard)
        cpfCheck(cpf_shard_gather(cpfShard, 0, Pstream::master() ? xyzw.data() : nullptr, Pstream::master() ? cells.data() : nullptr, nullptr));
    else if (cpfCtx)                       // the reference's topology: the master's one context holds everything
        cpfCheck(cpf_get_particles(cpfCtx, xyzw.data(), cells.data(), nullptr));
    if (Pstream::master())
    {
        FILE* fp = std::fopen("particles_out.f64", "wb"); std::fwrite(xyzw.data(), 8, xyzw.size(), fp); std::fclose(fp);
        fp = std::fopen("cells_out.i32", "wb"); std::fwrite(cells.data(), 4, cells.size(), fp); std::fclose(fp);
        if (cpfShard)
        {
            cpf_shard_stats st;
            cpfCheck(cpf_shard_get_stats(cpfShard, &st));
            std::printf("#mock: rank 0 of %d: %lld particles, %lld hand-offs, %lld re-cuts, %lld handed off\n", nProcs, (long long)st.n,
                        (long long)st.exchanges, (long long)st.rebalances, (long long)st.handedOff);
        }
        else std::printf("#mock: rank 0 of %d drives the one GPU\n", nProcs);
    }
    if (cpfShard) cpf_shard_destroy(cpfShard);
    if (cpfComm) cpf_comm_destroy(cpfComm);
    if (cpfCtx) cpf_destroy(cpfCtx);
    return 0;
}

extern "C" int main(int argc, char* argv[])
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s <caseDir> <nProcs> [<eulerianSteps>]\n", argv[0]); return 2; }
    const int nProcs = std::atoi(argv[2]);
    const int eulerianSteps = argc > 3 ? std::atoi(argv[3]) : 1;
    if (!std::getenv("CPF_COMM")) setenv("CPF_COMM", "inprocess", 1);      // ranks are threads of this process
    Pstream::init(nProcs);
    std::vector<std::thread> ranks;
    std::vector<std::string> failures((size_t)nProcs);
    for (int rank = 0; rank < nProcs; ++rank)
        ranks.emplace_back([&, rank] {
            try { rankMain(argv[1], rank, nProcs, eulerianSteps); }
            catch (const std::exception& e) { failures[(size_t)rank] = e.what(); std::fprintf(stderr, "FATAL (rank %d): %s\n", rank, e.what()); std::exit(1); }
        });
    for (auto& t : ranks) t.join();
    Info<< "End\n" << endl;
    return 0;
}

}
