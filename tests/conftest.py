import os
import sys

import numpy as np
import pytest

try:
    # torch bundles its own HIP runtime: it must be the FIRST HIP runtime loaded into the process,
    # otherwise (libcudaParticleAdvection.so loaded first -> /opt/rocm's runtime) torch finds no GPU
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_libs():
    """The CPU checkers (built on demand; oracle/_ref only where it was prebuilt from the reference)."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def pitz():
    from cudaparticlesfoam_amd.cases import pitzdaily as pz
    mesh = pz.pitzdaily_mesh()
    centres, vols = mesh.cell_centres_volumes()
    return dict(mesh=mesh, centres=centres, vols=vols, U_uniform=pz.uniform_u(mesh),
                U_analytic=pz.analytic_step_u(mesh, centres), pz=pz)


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    """Creates contexts on cuda:0 through the C-ABI; fails loudly if the library or GPU is missing."""
    from cudaparticlesfoam_amd.api import Context
    made = []

    def make():
        c = Context(0)
        c.set_option("stats", 1)          # the parity tests also compare the visit / reflection counters
        made.append(c)
        return c
    yield make
    for c in made:
        c.close()


def domain_diag(mesh):
    lo, hi = mesh.bounds()
    return float(np.linalg.norm(hi - lo))
