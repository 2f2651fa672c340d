"""Synthetic inputs shaped like the reference's tutorial cases (test/bench data, not product logic)."""
from .polymesh import PolyMesh, build_polymesh_from_cells, split_into_parts  # noqa: F401
from .blockmesh import block_mesh, box_mesh, hexes_to_polymesh, line_divide  # noqa: F401
from . import pitzdaily  # noqa: F401
from .refine import refine_hexes, refined_box, refined_pitzdaily  # noqa: F401
from . import foamfile  # noqa: F401  (a case directory as OpenFOAM stores it: polyMesh files, volVectorFields)
