"""GPU box: what one frame costs the step loop (the blocking part of cpf_write_vtu_async) against a cycle."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from cudaparticlesfoam_amd.api import Context
from cudaparticlesfoam_amd.cases import pitzdaily as pz
mesh = pz.pitzdaily_mesh(); c, _ = mesh.cell_centres_volumes()
for n in [int(float(v)) for v in os.environ.get("FRAME_N", "1e6,4e6,1e7").split(",")]:
    ctx = Context(0); ctx.set_mesh(mesh); ctx.set_velocity(pz.analytic_step_u(mesh, c)); ctx.set_option("vtu_binary", int(os.environ.get("FRAME_BINARY", "1")))
    ctx.seed_box(n, *pz.DOMAIN_BOX, 1); ctx.locate_initial()
    ctx.step(1e-4, 1.5e-5, 10, 4); ctx.synchronize()
    t0 = time.perf_counter(); ctx.step(1e-4, 1.5e-5, 10, 4); ctx.synchronize(); cyc = (time.perf_counter() - t0) / 10 * 1e3
    row = {"particles": n, "ms_per_cycle_fused": round(cyc, 4), "frames": []}
    for k in range(3):
        ctx.step(1e-4, 1.5e-5, 1, 2); ctx.synchronize()
        ke = C.c_double(); path = ("/tmp/frame_%d.vtu" % k).encode()
        t0 = time.perf_counter(); st = ctx.lib.cpf_write_vtu_async(ctx.h, path, C.byref(ke)); t1 = time.perf_counter()
        ctx.step(1e-4, 1.5e-5, 10, 4); ctx.synchronize(); t2 = time.perf_counter()
        ctx.lib.cpf_write_vtu_wait(ctx.h); t3 = time.perf_counter()
        row["frames"].append({"status": st, "call_ms": round((t1 - t0) * 1e3, 2), "next_10_cycles_ms": round((t2 - t1) * 1e3, 2),
                              "writer_done_after_ms": round((t3 - t0) * 1e3, 1), "ke": ke.value})
        os.remove(path)
    print(json.dumps(row), flush=True)
    ctx.close()
