"""The reference's TJunction tutorial case (the one ``cudaParticlesPimpleFoam`` ships with) restated as
synthetic input.

Geometry and mesh density restate
``/root/reference/tutorials/incompressible/cudaParticlesPimpleFoam/TJunction/system/blockMeshDict:36-128``
(20 vertices, 4 ungraded hex blocks of 1 mm cells: inlet duct 200 x 20 x 20, junction 20^3, two 20 x 200 x 20
branches; patches inlet / outlet1 / outlet2, everything else ``defaultFaces`` wall).  The particle-run parameters
restate ``.../system/cudaParticlesDict:17-30`` and ``.../system/controlDict:27`` (``deltaT 0.001``: ten Lagrangian
cycles of ``dt 1e-4`` per Eulerian step, ``src/advect.H:34-38``).  The transient velocity the tutorial gets from
``pimpleFoam`` cannot be produced here (no OpenFOAM), so a closed-form, time-modulated stand-in is provided.
"""
from __future__ import annotations

import numpy as np

from .blockmesh import block_mesh
from .polymesh import PolyMesh

# blockMeshDict:38-68 (metres, scale 1)
_XY = [(0.0, -0.01), (0.2, -0.01), (0.2, 0.01), (0.0, 0.01), (0.22, -0.01), (0.22, 0.01), (0.2, -0.21), (0.22, -0.21),
       (0.2, 0.21), (0.22, 0.21)]
VERTICES = np.array([(x, y, 0.0) for x, y in _XY] + [(x, y, 0.02) for x, y in _XY], dtype=np.float64)

# blockMeshDict:70-83
BLOCKS = [
    dict(hex=(0, 1, 2, 3, 10, 11, 12, 13), n=(200, 20, 20), simple=(1, 1, 1)),    # inlet duct
    dict(hex=(1, 4, 5, 2, 11, 14, 15, 12), n=(20, 20, 20), simple=(1, 1, 1)),     # junction
    dict(hex=(6, 7, 4, 1, 16, 17, 14, 11), n=(20, 200, 20), simple=(1, 1, 1)),    # branch to outlet1 (-y)
    dict(hex=(2, 5, 9, 8, 12, 15, 19, 18), n=(20, 200, 20), simple=(1, 1, 1)),    # branch to outlet2 (+y)
]

# blockMeshDict:89-122
PATCHES = [("inlet", "patch"), ("outlet1", "patch"), ("outlet2", "patch"), ("defaultFaces", "wall")]
# (block, local face slot) -> patch; slots 0 x-, 1 x+, 2 y-, 3 y+, 4 z-, 5 z+ in block-local axes
_SIDE_PATCH = {(0, 0): 0, (2, 2): 1, (3, 3): 2}

# topology that follows from the dict
N_CELLS, N_POINTS, N_FACES, N_INTERNAL = 248000, 273861, 769200, 718800
PATCH_SIZES = dict(inlet=400, outlet1=400, outlet2=400, defaultFaces=49200)

# cudaParticlesDict:17-30, controlDict:27
PARTICLE_DICT = dict(startTime=0.5, diffusionCoeff=1.5e-05, numParticles=4000000,
                     seedingBox=((0.0, -0.01, 0.0), (0.05, 0.01, 0.02)), dt=1e-04, saveInterval=2)
EULERIAN_DT = 1e-3
DOMAIN_BOX = ((0.0, -0.21, 0.0), (0.22, 0.21, 0.02))


def _classify(fcen, own, slot, block):
    pid = np.full(own.shape, 3, dtype=np.int64)
    for (b, s), p in _SIDE_PATCH.items():
        pid[(block == b) & (slot == s)] = p
    return pid


def tjunction_mesh(refine: int = 1) -> PolyMesh:
    """The 248 000-cell tutorial mesh (``refine=1``); ``refine=k`` multiplies every cell count by k."""
    blocks = [dict(b, n=tuple(v * refine for v in b["n"])) for b in BLOCKS]
    return block_mesh(VERTICES, blocks, scale=1.0, classify=_classify, patch_names=PATCHES)


def split_flow_u(mesh: PolyMesh, centres: np.ndarray | None = None, t: float = 0.0, u0: float = 3.0) -> np.ndarray:
    """Closed-form stand-in for the pimpleFoam solution, sampled at cell centres: a parabolic-ish duct flow of bulk
    speed ``u0`` that turns into the two branches at the junction, half each, pulsating with t (the tutorial's inlet is
    a time-varying total pressure, ``0/p``).  |U| dt stays below one 1 mm cell per Lagrangian cycle for u0 <= 5."""
    if centres is None:
        centres, _ = mesh.cell_centres_volumes()
    x, y, z = centres[:, 0], centres[:, 1], centres[:, 2]
    amp = u0 * (1.0 + 0.3 * np.sin(2.0 * np.pi * 4.0 * t))
    # duct profile across y (|y| <= 0.01) and z (0..0.02) in the inlet duct, across x (0.2..0.22) and z in the branches
    pz = 1.5 * (1.0 - ((z - 0.01) / 0.01) ** 2)
    s = np.clip((x - 0.19) / 0.02, 0.0, 1.0)
    s = s * s * (3.0 - 2.0 * s)                                   # 0 in the duct, 1 in the junction and branches
    py = 1.5 * (1.0 - np.clip(np.abs(y) / 0.01, 0.0, 1.0) ** 2)
    px = 1.5 * (1.0 - np.clip((x - 0.21) / 0.01, -1.0, 1.0) ** 2)
    side = np.tanh(y / 0.004)
    ux = amp * (1.0 - s) * py * pz / 1.5
    uy = amp * 0.5 * s * side * px * pz / 1.5
    # a weak secondary swirl so that the field is genuinely three-dimensional
    uz = 0.05 * amp * np.sin(np.pi * z / 0.02) * np.cos(40.0 * (x + y))
    return np.stack([ux, uy, uz], axis=1)
