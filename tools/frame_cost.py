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
        t0 = time.perf_counter(); st = ctx.lib.cpf_write_vtu_async(ctx.h, path, C.byref(ke) if os.environ.get("FRAME_KE", "0") == "1" else None); t1 = time.perf_counter()
        ctx.step(1e-4, 1.5e-5, 10, 4); ctx.synchronize(); t2 = time.perf_counter()
        ctx.lib.cpf_write_vtu_wait(ctx.h); t3 = time.perf_counter()
        row["frames"].append({"status": st, "call_ms": round((t1 - t0) * 1e3, 2), "next_10_cycles_ms": round((t2 - t1) * 1e3, 2),
                              "writer_done_after_ms": round((t3 - t0) * 1e3, 1), "ke": ke.value})
        os.remove(path)
    # the tutorial's cadence: a frame every 10 cycles (saveInterval 10), 50 cycles, as the fragments call it -- against the same
    # cycles without frames
    def run(frames):
        ctx.synchronize(); t0 = time.perf_counter()
        for k in range(5):
            ctx.step(1e-4, 1.5e-5, 9, 4); ctx.step(1e-4, 1.5e-5, 1, 2)
            if frames:
                ctx.lib.cpf_write_vtu_async(ctx.h, b"/tmp/frame_cadence.vtu", None)
        ctx.synchronize(); t1 = time.perf_counter()
        ctx.lib.cpf_write_vtu_wait(ctx.h)
        return round((t1 - t0) * 1e3, 2), round((time.perf_counter() - t0) * 1e3, 1)
    row["cadence_50_cycles_ms"] = {"no_frames": run(False)[0], "frame_every_10": dict(zip(("step_loop_ms", "until_last_frame_on_disk_ms"), run(True)))}
    print(json.dumps(row), flush=True)
    ctx.close()
