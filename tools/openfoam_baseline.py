#!/usr/bin/env python3
"""OpenFOAM's own CPU Lagrangian tracker as a baseline, IF an OpenFOAM installation is present.

north_star names "OpenFOAM's CPU kinematicCloud on the host cores of the same box".  OpenFOAM is in neither container of
this project (no $WM_PROJECT_DIR, no network), so this path has NEVER RUN: `probe()` is what bench.py reports --
{"available": false, ...} on every box so far.  Should a box ever carry OpenFOAM, `run()` writes the bench's pitzDaily mesh
and field as an OpenFOAM case (cases/foamfile.py), seeds a kinematicCloud by manualInjection with the bench's own points,
runs `icoUncoupledKinematicParcelFoam` for a bounded number of steps under a timeout and reports Mparticle-steps/s from the
solver's ExecutionTime -- best effort, every failure is returned as text instead of raised.
"""
import os
import re
import shutil
import subprocess
import tempfile
import time

SOLVERS = ("icoUncoupledKinematicParcelFoam", "kinematicParcelFoam")


def probe():
    root = os.environ.get("WM_PROJECT_DIR")
    solver = next((s for s in SOLVERS if shutil.which(s)), None)
    return {"available": bool(root and solver), "WM_PROJECT_DIR": root, "solver": solver,
            "version": os.environ.get("WM_PROJECT_VERSION")}


_HEADER = "FoamFile\n{\n    version 2.0;\n    format ascii;\n    class %s;\n    object %s;\n}\n"


def _dict(path, cls, obj, body):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(_HEADER % (cls, obj) + body)


def run(mesh, U, xyz, dt=1e-4, steps=50, timeout_s=120.0, threads=1):
    """One serial run (OpenFOAM parallelises by domain decomposition, not threads: `threads` is reported, not used)."""
    p = probe()
    if not p["available"]:
        return dict(p, error="OpenFOAM not installed")
    from cudaparticlesfoam_amd.cases import foamfile as ff
    case = tempfile.mkdtemp(prefix="cpf_of_")
    try:
        ff.write_polymesh(mesh, case)
        ff.write_vector_field(U, os.path.join(case, "0", "U"))
        n = int(xyz.shape[0])
        _dict(os.path.join(case, "system", "controlDict"), "dictionary", "controlDict",
              "application %s;\nstartFrom startTime;\nstartTime 0;\nstopAt endTime;\nendTime %g;\ndeltaT %g;\n"
              "writeControl timeStep;\nwriteInterval %d;\nwriteFormat binary;\nwritePrecision 10;\nrunTimeModifiable false;\n"
              % (p["solver"], dt * steps, dt, steps * 10))
        _dict(os.path.join(case, "system", "fvSchemes"), "dictionary", "fvSchemes",
              "ddtSchemes { default none; }\ngradSchemes { default none; }\ndivSchemes { default none; }\n"
              "laplacianSchemes { default none; }\ninterpolationSchemes { default linear; }\nsnGradSchemes { default none; }\n")
        _dict(os.path.join(case, "system", "fvSolution"), "dictionary", "fvSolution", "solvers {}\n")
        _dict(os.path.join(case, "constant", "transportProperties"), "dictionary", "transportProperties",
              "rhoInf [1 -3 0 0 0 0 0] 1.2;\ntransportModel Newtonian;\nnu [0 2 -1 0 0 0 0] 1e-05;\n")
        _dict(os.path.join(case, "constant", "turbulenceProperties"), "dictionary", "turbulenceProperties", "simulationType laminar;\n")
        _dict(os.path.join(case, "constant", "g"), "uniformDimensionedVectorField", "g", "dimensions [0 1 -2 0 0 0 0];\nvalue (0 0 0);\n")
        pos = "\n".join("(%.12g %.12g %.12g)" % tuple(r) for r in xyz)
        _dict(os.path.join(case, "constant", "kinematicCloudPositions"), "vectorField", "kinematicCloudPositions", "%d\n(\n%s\n)\n" % (n, pos))
        _dict(os.path.join(case, "constant", "kinematicCloudProperties"), "dictionary", "kinematicCloudProperties",
              "solution\n{\n active true;\n coupled false;\n transient yes;\n cellValueSourceCorrection off;\n maxCo 1e9;\n"
              " interpolationSchemes { rho cell; U cell; mu cell; }\n integrationSchemes { U Euler; }\n}\n"
              "constantProperties\n{\n rho0 1.2;\n}\n"
              "subModels\n{\n particleForces { }\n injectionModels\n {\n  model1\n  {\n   type manualInjection;\n   massTotal 0;\n"
              "   parcelBasisType fixed;\n   nParticle 1;\n   SOI 0;\n   positionsFile \"kinematicCloudPositions\";\n   U0 (10 0 0);\n"
              "   sizeDistribution { type fixedValue; fixedValueDistribution { value 1e-6; } }\n  }\n }\n"
              " dispersionModel none;\n patchInteractionModel standardWallInteraction;\n"
              " standardWallInteractionCoeffs { type rebound; e 1; mu 0; }\n"
              " surfaceFilmModel none;\n stochasticCollisionModel none;\n collisionModel none;\n heatTransferModel none;\n}\n"
              "cloudFunctions { }\n")
        t0 = time.perf_counter()
        r = subprocess.run([p["solver"], "-case", case], capture_output=True, text=True, timeout=timeout_s)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return dict(p, error="solver exited %d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:]))
        ex = re.findall(r"ExecutionTime = ([0-9.eE+-]+) s", r.stdout)
        secs = float(ex[-1]) - (float(ex[0]) if len(ex) > 1 else 0.0) if ex else wall
        done = max(1, len(ex) - 1) if len(ex) > 1 else steps
        return dict(p, value=round(n * done / max(secs, 1e-9) / 1e6, 3), unit="Mparticle-steps/s", cores=1, particles=n, steps=done,
                    seconds=round(secs, 2), note="OpenFOAM %s, one core, never validated (no OpenFOAM on any box of this project so far)" % p["solver"])
    except Exception as e:                                        # noqa: BLE001 -- a baseline that fails must not take the bench line with it
        return dict(p, error="%s: %s" % (type(e).__name__, str(e)[:300]))
    finally:
        shutil.rmtree(case, ignore_errors=True)


if __name__ == "__main__":
    import json
    print(json.dumps(probe()))
