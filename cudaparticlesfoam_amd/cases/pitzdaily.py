"""The reference's pitzDaily tutorial case restated as synthetic input.

Geometry and mesh density restate
``/root/reference/tutorials/incompressible/cudaParticlesUncoupledFoam/pitzDaily/system/blockMeshDict:17-150``
(22 vertices, 5 graded hex blocks one cell thick, ``scale 0.001``, 5 patches).  The
particle-run parameters restate ``.../system/cudaParticlesDict:17-29`` and the inlet
velocity ``.../0/U:26``.  The frozen velocity field the tutorial gets from ``simpleFoam``
cannot be produced here (no OpenFOAM), so two closed-form stand-ins are provided
(SURVEY.md 8d config 2): uniform (10,0,0) and an analytic backward-facing-step field.
"""
from __future__ import annotations

import numpy as np

from .blockmesh import block_mesh
from .polymesh import PolyMesh

# blockMeshDict:19-44 (mm; scale 0.001 applied by block_mesh)
_XY = [(-20.6, 0), (-20.6, 25.4), (0, -25.4), (0, 0), (0, 25.4), (206, -25.4), (206, 0), (206, 25.4),
       (290, -16.6), (290, 0), (290, 16.6)]
VERTICES = np.array([(x, y, -0.5) for x, y in _XY] + [(x, y, 0.5) for x, y in _XY], dtype=np.float64)

# blockMeshDict:46-63
NEG_Y = [(2, 4, 1), (1, 3, 0.3)]
POS_Y = [(1, 4, 2), (2, 3, 4), (2, 4, 0.25)]
POS_YR = [(2, 1, 1), (1, 1, 0.25)]

# blockMeshDict:66-87
BLOCKS = [
    dict(hex=(0, 3, 4, 1, 11, 14, 15, 12), n=(18, 30, 1), simple=(0.5, POS_Y, 1)),
    dict(hex=(2, 5, 6, 3, 13, 16, 17, 14), n=(180, 27, 1), edge=[4, 4, 4, 4, NEG_Y, 1, 1, NEG_Y, 1, 1, 1, 1]),
    dict(hex=(3, 6, 7, 4, 14, 17, 18, 15), n=(180, 30, 1),
         edge=[4, 4, 4, 4, POS_Y, POS_YR, POS_YR, POS_Y, 1, 1, 1, 1]),
    dict(hex=(5, 8, 9, 6, 16, 19, 20, 17), n=(25, 27, 1), simple=(2.5, 1, 1)),
    dict(hex=(6, 9, 10, 7, 17, 20, 21, 18), n=(25, 30, 1), simple=(2.5, POS_YR, 1)),
]

# blockMeshDict:93-150
PATCHES = [("inlet", "patch"), ("outlet", "patch"), ("upperWall", "wall"), ("lowerWall", "wall"),
           ("frontAndBack", "empty")]
# (block, local face slot) -> patch index; slots 0 x-, 1 x+, 2 y-, 3 y+ ; z faces are frontAndBack
_SIDE_PATCH = {(0, 0): 0, (0, 2): 3, (0, 3): 2, (1, 0): 3, (1, 2): 3, (2, 3): 2, (3, 2): 3, (3, 1): 1,
               (4, 3): 2, (4, 1): 1}

# topology that follows from the dict
N_CELLS, N_POINTS, N_FACES, N_INTERNAL = 12225, 25012, 49180, 24170
PATCH_SIZES = dict(inlet=30, outlet=57, upperWall=223, lowerWall=250, frontAndBack=24450)

# cudaParticlesDict:17-29 and controlDict deltaT (Allrun:11 runs ONE advect.H pass)
PARTICLE_DICT = dict(startTime=282.0, endTime=382.0, diffusionCoeff=1.5e-05, numParticles=100000,
                     seedingBox=((-0.02, 0.025, 0.0001), (0.0, 0.0, -0.0001)), dt=1e-04, saveInterval=10)
U_INLET = (10.0, 0.0, 0.0)   # 0/U:26


def _classify(fcen, own, slot, block):
    pid = np.full(own.shape, 4, dtype=np.int64)
    for (b, s), p in _SIDE_PATCH.items():
        pid[(block == b) & (slot == s)] = p
    bad = (slot < 4) & (pid == 4)
    if bad.any():
        raise ValueError("unexpected exposed block side in pitzDaily")
    return pid


def pitzdaily_mesh(refine: int = 1) -> PolyMesh:
    """The 12 225-cell tutorial mesh (``refine=1``); ``refine=k`` multiplies the in-plane
    cell counts by k (synthetic larger meshes for the transient-U config)."""
    blocks = [dict(b, n=(b["n"][0] * refine, b["n"][1] * refine, 1)) for b in BLOCKS]
    return block_mesh(VERTICES, blocks, scale=0.001, classify=_classify, patch_names=PATCHES)


# ----------------------------------------------------------------------------- velocity fields
def uniform_u(mesh: PolyMesh, u=U_INLET) -> np.ndarray:
    return np.tile(np.asarray(u, dtype=np.float64), (mesh.n_cells, 1))


def _walls(x):
    """Channel floor/ceiling y(x) in metres (piecewise linear through the dict's vertices)."""
    top = np.interp(x, [-0.0206, 0.206, 0.290], [0.0254, 0.0254, 0.0166])
    bot = np.where(x < 0.0, 0.0, np.interp(x, [0.0, 0.206, 0.290], [-0.0254, -0.0254, -0.0166]))
    return bot, top


def analytic_step_u(mesh: PolyMesh, centres: np.ndarray | None = None) -> np.ndarray:
    """Closed-form, seedless stand-in for the frozen simpleFoam solution, sampled at cell
    centres: mass-conserving parabolic through-flow between the local floor and ceiling
    plus a Gaussian recirculation vortex behind the step.  |U| <= ~16 m/s so that
    |U|*dt (dt=1e-4) stays within ~1-3 cells per Lagrangian sub-step."""
    if centres is None:
        centres, _ = mesh.cell_centres_volumes()
    x, y = centres[:, 0], centres[:, 1]
    bot, top = _walls(x)
    h = top - bot
    eta = np.clip((y - bot) / h, 0.0, 1.0)
    ubulk = 10.0 * 0.0254 / h
    ux = ubulk * 6.0 * eta * (1.0 - eta)
    # slope-following vertical component (keeps the through-flow tangential to both walls)
    dx = 1e-6
    b2, t2 = _walls(x + dx)
    dbot = (b2 - bot) / dx; dtop = (t2 - top) / dx
    uy = ux * ((1.0 - eta) * dbot + eta * dtop)
    # recirculation bubble: solid-body-like Gaussian vortex, clockwise, centred behind the step
    xc, yc, rc, om = 0.06, -0.0127, 0.011, -450.0
    rx, ry = x - xc, y - yc
    g = om * np.exp(-(rx * rx / (4 * rc) ** 2 + ry * ry / rc ** 2))
    ux = ux - g * ry
    uy = uy + g * rx * (rc / (4 * rc)) ** 2
    return np.stack([ux, uy, np.zeros_like(ux)], axis=1)


# ----------------------------------------------------------------------------- seeding helpers
def splitmix64(seed: int, n: int) -> np.ndarray:
    """n uint64 words of the SplitMix64 stream (seedable, portable, no global state)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, n + 1, dtype=np.uint64))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform_points(seed: int, n: int, lower, upper) -> np.ndarray:
    """n points uniform in a box from SplitMix64 (53-bit mantissas), shape (n,3) float64."""
    w = splitmix64(seed, 3 * n).reshape(n, 3)
    u = (w >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    lo = np.asarray(lower, dtype=np.float64); hi = np.asarray(upper, dtype=np.float64)
    return lo + u * (hi - lo)


INLET_BOX = ((-0.02, 0.0, -1e-4), (0.0, 0.025, 1e-4))          # config 2
DOMAIN_BOX = ((-0.0206, -0.0254, -4e-4), (0.29, 0.0254, 4e-4))  # config 3: whole fluid bbox, rejection-resampled
