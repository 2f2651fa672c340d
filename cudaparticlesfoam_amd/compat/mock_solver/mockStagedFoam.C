// Test harness: drives the compat shims (cuda/common.h, query/*.h) in the REFERENCE's five-call
// order -- cudaAdvect, cudaBrownianMotion, convexTetQuery, convexWallReflect, cudaMoveParticles --
// on caller-owned device arrays with the reference's layouts, the way a host that kept the
// reference's own loop structure (src/advect.H:96-161) would.  Usage: mockStagedFoam <caseDir>.
#include "cuda/common.h"
#include "cuda/DeviceTetMesh.cuh"
#include "query/ConvexQuery.h"
#include "query/RTQuery.h"
#include <string>
#include "optix/OptixQuery.h"

#include "fvCFD.H"
#include "case_io.H"

namespace advect {

extern "C" int main(int argc, char* argv[])
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s <caseDir>\n", argv[0]); return 2; }
    fvMesh mesh;
    volVectorField U;
    Time runTime;
    IOdictionary dict;
    try {
        loadCase(argv[1], mesh, U, runTime, dict);
        const int numParticles = dict.getOrDefault("numParticles", 1000);
        const double dt = dict.getOrDefault("dt", 1e-4);
        const double diffusionCoeff = dict.getOrDefault("diffusionCoeff", 5.7e-6);
        const boundBox sb = dict.getOrDefault("seedingBox", boundBox(point(0, 0, 0), point(30, 30, 30)));

        std::vector<label> off(mesh.faces().size() + 1, 0), verts;
        forAll(mesh.faces(), f) {
            off[f + 1] = off[f] + mesh.faces()[f].size();
            forAll(mesh.faces()[f], k) verts.push_back(mesh.faces()[f][k]);
        }
        DeviceTetMesh devMesh;
        devMesh.upload(reinterpret_cast<const double*>(mesh.points().cdata()), mesh.points().size(), off.data(),
                       verts.data(), mesh.faces().size(), mesh.faceOwner().cdata(), mesh.faceNeighbour().cdata(),
                       mesh.nInternalFaces(), mesh.nCells());
        bindMesh(devMesh);
        std::vector<vec3d> cellVel;
        forAll(U.primitiveField(), c) cellVel.push_back(vec3d(U[c].x(), U[c].y(), U[c].z()));
        cudaUpdateVelocity(cellVel, mesh.nCells(), devMesh.d_indices, devMesh.d_velocities);

        Particle* d_particles = deviceAlloc<Particle>(devMesh, numParticles);
        int* d_particles_ConvextetIDs = deviceAlloc<int>(devMesh, numParticles, 0xFF);
        vec4d* d_particle_disps = deviceAlloc<vec4d>(devMesh, numParticles);
        vec4d* d_particle_vels = deviceAlloc<vec4d>(devMesh, numParticles);
        curandState_t rand_states;
        initRandomGenerator(numParticles, &rand_states);

        box3d initBox(vec3d(sb.min().x(), sb.min().y(), sb.min().z()), vec3d(sb.max().x(), sb.max().y(), sb.max().z()));
        cudaInitParticles(d_particles, numParticles, initBox);
        OptixQuery tetQueryAccelerator;
        RTQuery(tetQueryAccelerator, devMesh, d_particles, d_particles_ConvextetIDs, numParticles);
        cudaReportParticles(numParticles, d_particles_ConvextetIDs);

        int nCycles = max(ceil(runTime.deltaT().value() / dt), 1);
        double cycleDt = runTime.deltaT().value() / nCycles;
        // argv[2] == "rtx": the reference's usingRTX branch of the cycle (src/advect.H:126-135): RTQuery in displacement mode and
        // RTWallReflect, whose disps / vels arguments come in the OTHER order than convexWallReflect's
        const bool usingRTX = argc > 2 && std::string(argv[2]) == "rtx";
        cudaTimer timer;
        timer.start();
        for (int i = 0; i < nCycles; i++) {
            cudaAdvect(d_particles, d_particles_ConvextetIDs, d_particle_vels, d_particle_disps, cycleDt, numParticles,
                       devMesh.d_indices, devMesh.d_positions, devMesh.d_velocities, "TetVelocity");
            cudaBrownianMotion(d_particles, d_particle_disps, &rand_states, cycleDt, numParticles, diffusionCoeff);
            if (usingRTX) {
                RTQuery(devMesh, d_particles, d_particle_disps, d_particles_ConvextetIDs, numParticles);
                RTWallReflect(devMesh, d_particles_ConvextetIDs, d_particles, d_particle_disps, d_particle_vels, numParticles);
            } else {
                convexTetQuery(devMesh, d_particles, d_particle_disps, d_particles_ConvextetIDs, numParticles);
                convexWallReflect(devMesh, d_particles_ConvextetIDs, d_particles, d_particle_vels, d_particle_disps,
                                  numParticles);
            }
            cudaMoveParticles(d_particles, d_particle_disps, numParticles, d_particles_ConvextetIDs);
        }
        const double cycleMs = timer.stop();
        std::printf("#mock: %s particles, %d cycles (%s) in %.3f ms\n", prettyNumber((std::size_t)numParticles).c_str(), nCycles,
                    usingRTX ? "RTQuery + RTWallReflect" : "convexTetQuery + convexWallReflect", cycleMs);
        writeParticles2VTU(nCycles, d_particles, d_particle_vels, d_particles_ConvextetIDs, numParticles,
                           d_particles_ConvextetIDs);
        std::vector<double> xyzw((size_t)numParticles * 4);
        std::vector<int32_t> cells((size_t)numParticles);
        check(devMesh.ctx, cpf_copy_to_host(devMesh.ctx, xyzw.data(), d_particles, xyzw.size() * 8));
        check(devMesh.ctx, cpf_copy_to_host(devMesh.ctx, cells.data(), d_particles_ConvextetIDs, cells.size() * 4));
        FILE* fp = std::fopen("particles_out.f64", "wb"); std::fwrite(xyzw.data(), 8, xyzw.size(), fp); std::fclose(fp);
        fp = std::fopen("cells_out.i32", "wb"); std::fwrite(cells.data(), 4, cells.size(), fp); std::fclose(fp);
        deviceFree(devMesh, d_particles); deviceFree(devMesh, d_particles_ConvextetIDs);
        deviceFree(devMesh, d_particle_disps); deviceFree(devMesh, d_particle_vels);
        cpf_destroy(devMesh.ctx);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "FATAL: %s\n", e.what());
        return 1;
    }
    return 0;
}

}
