"""Writes a case directory in the tiny raw format the mock solvers read (compat/mock_solver/case_io.H)."""
import os

import numpy as np


def dump_case(path, mesh, U, dict_entries, run_time_value, delta_t):
    os.makedirs(path, exist_ok=True)
    np.ascontiguousarray(mesh.points, np.float64).tofile(os.path.join(path, "points.f64"))
    np.ascontiguousarray(mesh.face_offsets, np.int32).tofile(os.path.join(path, "faceoff.i32"))
    np.ascontiguousarray(mesh.face_verts, np.int32).tofile(os.path.join(path, "faceverts.i32"))
    np.ascontiguousarray(mesh.owner, np.int32).tofile(os.path.join(path, "owner.i32"))
    np.ascontiguousarray(mesh.neighbour, np.int32).tofile(os.path.join(path, "neighbour.i32"))
    np.ascontiguousarray(U, np.float64).tofile(os.path.join(path, "U.f64"))
    with open(os.path.join(path, "dict.txt"), "w") as f:
        for k, v in dict_entries.items():
            if k == "seedingBox":
                f.write("seedingBox %s\n" % " ".join(repr(float(x)) for x in (*v[0], *v[1])))
            else:
                f.write("%s %r\n" % (k, float(v)))
        f.write("runTimeValue %r\ndeltaT %r\n" % (float(run_time_value), float(delta_t)))
