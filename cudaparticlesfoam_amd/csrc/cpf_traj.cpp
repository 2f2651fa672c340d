// Trajectory collection and its two writers: what the reference's addToTrajectories / saveTrajectories /
// writeStreamline2VTK do (third_party/RTXAdvect/cuda/utils.cpp:7-94, declared cuda/common.h:87-92, called from
// src/advect.H:163-175 behind `saveStreamlinetoFile`), behind the C-ABI.
//
// A trajectory is the list of single-precision positions one particle had at the sampling instants at which it was
// ACTIVE (w != 0, utils.cpp:21); particles with fewer than two samples are left out of both files (:36, :55).  The
// OBJ file is "v x y z" lines followed by "l a b" segments with 1-based vertex numbers per trajectory (:33-43); the VTK
// file is legacy POLYDATA with one poly-line per trajectory and a StreamlineID cell field (:62-91).  Numbers are printed
// as the reference prints them: operator<< of a float on a default std::ofstream (six significant digits, %g style).
#include <fstream>
#include <new>
#include <string>
#include <vector>

#include "cpf.h"

struct cpf_traj {
    std::vector<std::vector<float>> pts;      // per particle: x0 y0 z0 x1 y1 z1 ...
};

namespace {

// offsets[nTraj + 1] into xyz[][3]; trajectories with fewer than two points are skipped by both writers
struct Flat {
    int64_t nTraj;
    const int64_t* off;
    const float* xyz;
    int64_t size(int64_t t) const { return off[t + 1] - off[t]; }
    const float* point(int64_t t, int64_t k) const { return xyz + 3 * (off[t] + k); }
};

int write_obj(const char* path, const Flat& f) {
    std::ofstream out(path);
    if (!out) return CPF_ERR_ARG;
    long long written = 0;
    for (int64_t t = 0; t < f.nTraj; ++t) {
        const int64_t k = f.size(t);
        if (k <= 1) continue;
        const long long first = written + 1;                  // OBJ numbers vertices from 1
        for (int64_t i = 0; i < k; ++i) {
            const float* p = f.point(t, i);
            out << "v " << p[0] << " " << p[1] << " " << p[2] << std::endl;
        }
        written += k;
        for (int64_t i = 0; i + 1 < k; ++i) out << "l " << (first + i) << " " << (first + i + 1) << std::endl;
    }
    out.close();
    return out.fail() ? CPF_ERR_ARG : CPF_OK;
}

int write_vtk(const char* path, const Flat& f) {
    std::ofstream out(path);
    if (!out) return CPF_ERR_ARG;
    long long lines = 0, verts = 0;
    for (int64_t t = 0; t < f.nTraj; ++t)
        if (f.size(t) > 1) { ++lines; verts += f.size(t); }
    out << "# vtk DataFile Version 4.1\n" << "vtk output\n" << "ASCII\n" << "DATASET POLYDATA\n";
    out << "POINTS " << verts << " float\n";
    for (int64_t t = 0; t < f.nTraj; ++t) {
        if (f.size(t) <= 1) continue;
        for (int64_t i = 0; i < f.size(t); ++i) {
            const float* p = f.point(t, i);
            out << p[0] << " " << p[1] << " " << p[2] << std::endl;
        }
    }
    out << "\n";
    out << "LINES " << lines << " " << verts + lines << "\n";
    long long id = 0;
    for (int64_t t = 0; t < f.nTraj; ++t) {
        const int64_t k = f.size(t);
        if (k <= 1) continue;
        out << (unsigned long)k;
        for (int64_t i = 0; i < k; ++i) out << " " << id++;
        out << "\n";
    }
    out << "\n\n";
    out << "CELL_DATA " << lines << "\n" << "FIELD FieldData 1\n";
    out << "StreamlineID 1 " << lines << " int\n";
    for (long long i = 0; i < lines; ++i) out << i << " " << "\n";
    out.close();
    return out.fail() ? CPF_ERR_ARG : CPF_OK;
}

void flatten(const cpf_traj* t, std::vector<int64_t>& off, std::vector<float>& xyz) {
    off.assign(t->pts.size() + 1, 0);
    size_t total = 0;
    for (size_t i = 0; i < t->pts.size(); ++i) { off[i] = (int64_t)(total / 3); total += t->pts[i].size(); }
    off[t->pts.size()] = (int64_t)(total / 3);
    xyz.clear(); xyz.reserve(total);
    for (const auto& p : t->pts) xyz.insert(xyz.end(), p.begin(), p.end());
}

}  // namespace

extern "C" {

int cpf_traj_create(cpf_traj** out) {
    if (!out) return CPF_ERR_ARG;
    *out = new (std::nothrow) cpf_traj();
    return *out ? CPF_OK : CPF_ERR_NOMEM;
}
void cpf_traj_destroy(cpf_traj* t) { delete t; }

int cpf_traj_add_host(cpf_traj* t, const double* xyzw, int64_t n) {
    if (!t || n < 0 || (n > 0 && !xyzw)) return CPF_ERR_ARG;
    if (t->pts.empty()) t->pts.resize((size_t)n);                  // sized by the first sample (utils.cpp:10-11)
    if ((int64_t)t->pts.size() != n) return CPF_ERR_ARG;
    try {
        for (int64_t i = 0; i < n; ++i) {
            const double* p = xyzw + 4 * i;
            if (!p[3]) continue;                                   // inactive: no sample
            auto& v = t->pts[(size_t)i];
            v.push_back((float)p[0]); v.push_back((float)p[1]); v.push_back((float)p[2]);
        }
    } catch (const std::bad_alloc&) { return CPF_ERR_NOMEM; }
    return CPF_OK;
}

int cpf_traj_add(cpf_context* ctx, cpf_traj* t) {
    if (!ctx || !t) return CPF_ERR_ARG;
    int64_t n = 0;
    int r = cpf_num_particles(ctx, &n);
    if (r) return r;
    std::vector<double> xyzw((size_t)n * 4);
    r = cpf_get_particles(ctx, xyzw.data(), nullptr, nullptr);     // particle-id order: trajectory i is particle i's
    if (r) return r;
    return cpf_traj_add_host(t, xyzw.data(), n);
}

int cpf_traj_add_stage(cpf_context* ctx, cpf_traj* t, const double* particles_dev, int64_t n) {
    if (!ctx || !t || n < 0 || (n > 0 && !particles_dev)) return CPF_ERR_ARG;
    std::vector<double> xyzw((size_t)n * 4);
    const int r = cpf_copy_to_host(ctx, xyzw.data(), particles_dev, xyzw.size() * 8);
    if (r) return r;
    return cpf_traj_add_host(t, xyzw.data(), n);
}

int cpf_traj_sizes(const cpf_traj* t, int64_t* nTrajectories, int64_t* nPoints) {
    if (!t) return CPF_ERR_ARG;
    if (nTrajectories) *nTrajectories = (int64_t)t->pts.size();
    if (nPoints) { int64_t s = 0; for (const auto& p : t->pts) s += (int64_t)p.size() / 3; *nPoints = s; }
    return CPF_OK;
}

int cpf_traj_save_obj_arrays(const char* path, int64_t nTrajectories, const int64_t* offsets, const float* xyz) {
    if (!path || nTrajectories < 0 || !offsets || (offsets[nTrajectories] > 0 && !xyz)) return CPF_ERR_ARG;
    return write_obj(path, Flat{nTrajectories, offsets, xyz});
}
int cpf_traj_write_vtk_arrays(const char* path, int64_t nTrajectories, const int64_t* offsets, const float* xyz) {
    if (!path || nTrajectories < 0 || !offsets || (offsets[nTrajectories] > 0 && !xyz)) return CPF_ERR_ARG;
    return write_vtk(path, Flat{nTrajectories, offsets, xyz});
}
int cpf_traj_save_obj(const cpf_traj* t, const char* path) {
    if (!t || !path) return CPF_ERR_ARG;
    std::vector<int64_t> off; std::vector<float> xyz;
    flatten(t, off, xyz);
    return write_obj(path, Flat{(int64_t)t->pts.size(), off.data(), xyz.data()});
}
int cpf_traj_write_vtk(const cpf_traj* t, const char* path) {
    if (!t || !path) return CPF_ERR_ARG;
    std::vector<int64_t> off; std::vector<float> xyz;
    flatten(t, off, xyz);
    return write_vtk(path, Flat{(int64_t)t->pts.size(), off.data(), xyz.data()});
}

}  // extern "C"
