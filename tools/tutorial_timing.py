#!/usr/bin/env python3
"""Developer tool (GPU box): the reference tutorial's own size (pitzDaily, 1e5 particles, D = 1.5e-5, 1000 Lagrangian
cycles in one advect.H pass) through the host mirror of the fragments: per-cycle launches vs cycles fused between
output points.  python tools/tutorial_timing.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cudaparticlesfoam_amd.api import CudaParticles
from cudaparticlesfoam_amd.cases import pitzdaily as pz

mesh = pz.pitzdaily_mesh(); U = pz.analytic_step_u(mesh)
for label, writer in (("no output (1 fused launch)", None), ("output every 10 cycles (discarded)", lambda *a: None)):
    p = CudaParticles(mesh, U, dict(pz.PARTICLE_DICT), writer=writer)
    p.ctx.set_option("stats", 0)
    p.ctx.synchronize(); t0 = time.perf_counter()
    n = p.advect(300.0, 0.1)
    p.ctx.synchronize(); el = time.perf_counter() - t0
    print("%-38s %d cycles x %d particles in %.1f ms = %.1f Mparticle-steps/s" % (label, n, p.numParticles, el * 1e3,
                                                                                 n * p.numParticles / el / 1e6))
    p.close()
