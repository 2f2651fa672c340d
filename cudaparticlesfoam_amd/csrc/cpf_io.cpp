// Output path: D2H + ASCII .vtu writer with the reference's on-disk layout.
//
// Restates the format of writeParticles2VTU (third_party/RTXAdvect/cuda/utils.cpp:144-283):
// one VTK_VERTEX cell per particle, arrays Position (Float64, "%.15lf"), ParticleType (w),
// ParticleID, ParticleTetID, ConvexTetID, vels ("%lf", NaN -> 0), KEs, then connectivity /
// offsets / types.  Differences, on purpose:
//   * ParticleTetID / ConvexTetID both hold the containing CELL id (there are no tets here; the
//     reference's ParticleTetID is uninitialised memory in its default ConvexPoly build);
//   * KEs holds the kinetic energy (the reference prints 0 for every non-zero KE);
//   * no system("pause") on NaN: the function returns CPF_ERR_STATE instead.
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "cpf.h"

extern "C" int cpf_write_vtu_arrays(const char* path, int64_t n, const double* xyzw, const int32_t* cell,
                                    const double* vel, double* totalKE) {
    if (!path || n < 0 || (n > 0 && (!xyzw || !cell || !vel))) return CPF_ERR_ARG;
    FILE* fp = std::fopen(path, "w");
    if (!fp) return CPF_ERR_ARG;
    const long long N = (long long)n;
    std::fprintf(fp, "<VTKFile type='UnstructuredGrid' version='1.0' byte_order='LittleEndian' header_type='UInt64'>\n");
    std::fprintf(fp, "<UnstructuredGrid>\n");
    std::fprintf(fp, "<Piece NumberOfCells='%lld' NumberOfPoints='%lld'>\n", N, N);
    std::fprintf(fp, "<Points>\n");
    std::fprintf(fp, "<DataArray NumberOfComponents='3' type='Float64' Name='Position' format='ascii'>\n");
    for (long long i = 0; i < N; ++i)
        std::fprintf(fp, "%.15lf %.15lf %.15lf\n", xyzw[4 * i], xyzw[4 * i + 1], xyzw[4 * i + 2]);
    std::fprintf(fp, "</DataArray>\n</Points>\n<PointData>\n");
    std::fprintf(fp, "<DataArray NumberOfComponents='1' type='Int32' Name='ParticleType' format='ascii'>\n");
    for (long long i = 0; i < N; ++i) std::fprintf(fp, "%d\n", (int)xyzw[4 * i + 3]);
    std::fprintf(fp, "</DataArray>\n");
    std::fprintf(fp, "<DataArray NumberOfComponents='1' type='Int32' Name='ParticleID' format='ascii'>\n");
    for (long long i = 0; i < N; ++i) std::fprintf(fp, "%lld\n", i);
    std::fprintf(fp, "</DataArray>\n");
    for (const char* name : {"ParticleTetID", "ConvexTetID"}) {
        std::fprintf(fp, "<DataArray NumberOfComponents='1' type='Int32' Name='%s' format='ascii'>\n", name);
        for (long long i = 0; i < N; ++i) std::fprintf(fp, "%d\n", cell[i]);
        std::fprintf(fp, "</DataArray>\n");
    }
    std::fprintf(fp, "<DataArray NumberOfComponents='3' type='Float32' Name='vels' format='ascii'>\n");
    for (long long i = 0; i < N; ++i) {
        if (std::isnan(vel[4 * i])) std::fprintf(fp, "%lf %lf %lf\n", 0.0, 0.0, 0.0);
        else std::fprintf(fp, "%lf %lf %lf\n", vel[4 * i], vel[4 * i + 1], vel[4 * i + 2]);
    }
    std::fprintf(fp, "</DataArray>\n");
    std::fprintf(fp, "<DataArray NumberOfComponents='1' type='Float32' Name='KEs' format='ascii'>\n");
    double total = 0.0;
    for (long long i = 0; i < N; ++i) {
        const double ke = 0.5 * (vel[4 * i] * vel[4 * i] + vel[4 * i + 1] * vel[4 * i + 1] + vel[4 * i + 2] * vel[4 * i + 2]);
        std::fprintf(fp, "%lf\n", ke);
        total += ke;
    }
    std::fprintf(fp, "</DataArray>\n</PointData>\n<Cells>\n");
    std::fprintf(fp, "<DataArray type='Int32' Name='connectivity' format='ascii'>\n");
    for (long long i = 0; i < N; ++i) std::fprintf(fp, "%lld\n", i);
    std::fprintf(fp, "</DataArray>\n<DataArray type='Int32' Name='offsets' format='ascii'>\n");
    for (long long i = 0; i < N; ++i) std::fprintf(fp, "%lld\n", i + 1);
    std::fprintf(fp, "</DataArray>\n<DataArray type='UInt8' Name='types' format='ascii'>\n");
    for (long long i = 0; i < N; ++i) std::fprintf(fp, "1\n");
    std::fprintf(fp, "</DataArray>\n</Cells>\n</Piece>\n</UnstructuredGrid>\n</VTKFile>\n");
    const bool bad = std::ferror(fp) != 0;
    std::fclose(fp);
    if (totalKE) *totalKE = total;
    if (bad) return CPF_ERR_ARG;
    return std::isnan(total) ? CPF_ERR_STATE : CPF_OK;
}

extern "C" int cpf_write_vtu(cpf_context* ctx, const char* path, double* totalKE) {
    if (!ctx || !path) return CPF_ERR_ARG;
    int64_t n = 0;
    int r = cpf_num_particles(ctx, &n);
    if (r) return r;
    std::vector<double> xyzw((size_t)n * 4), vel((size_t)n * 4);
    std::vector<int32_t> cell((size_t)n);
    r = cpf_get_particles(ctx, xyzw.data(), cell.data(), vel.data());
    if (r) return r;
    return cpf_write_vtu_arrays(path, n, xyzw.data(), cell.data(), vel.data(), totalKE);
}
