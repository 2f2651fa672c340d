#!/usr/bin/env python3
"""Developer tool (GPU box): the sustained rate with the tutorial's diffusion (what bench.py reports as config.brownian_steady)
for a list of option settings, e.g. the sub-cell sort key against the cell id alone, sort intervals, the key sort's method.
  python tools/brownian_steady.py --case pitz --particles 1e7 --set sort_key_bits=0 --set sort_key_bits=25 --intervals 25,50"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="pitz"); ap.add_argument("--field", default=None); ap.add_argument("--particles", type=float, default=1e7)
    ap.add_argument("--D", type=float, default=1.5e-5); ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--set", action="append", default=[], help="k=v[,k=v...] one variant per --set ('' = defaults)")
    ap.add_argument("--intervals", default="25")
    ap.add_argument("--chunk", type=int, default=0, help="also: the same run through cpf_shard_step in calls of this many cycles "
                    "(what the parallel fragments do between two frames; the shard fuses the cycles up to the next sort)")
    a = ap.parse_args()
    import torch
    from _cases import make_case
    from _spinup import device_spinup
    from cudaparticlesfoam_amd.api import Context
    dev = torch.device("cuda", 0)
    n = int(a.particles)
    for variant in (a.set or [""]):
        for interval in [int(v) for v in a.intervals.split(",")]:
            ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            mesh, x, y, z, c, fields = make_case(a.case, ctx, torch, n, dev, a.field)
            for kv in [s for s in variant.split(",") if s]:
                k, v = kv.split("="); ctx.set_option(k, float(v))
            g = torch.arange(n, dtype=torch.int64, device=dev)
            alt = [torch.empty_like(t) for t in (x, y, z, c, g)]
            cur = [x, y, z, c, g]
            p = lambda t: t.data_ptr()   # noqa: E731

            def sort():
                nonlocal cur, alt
                ctx.sort_by_cell_dev_to(*[p(t) for t in cur], *[p(t) for t in alt], n)
                cur, alt = alt, cur
            sort()
            ctx.step_dev(*[p(t) for t in cur[:4]], p(cur[4]), None, n, 1e-4, a.D, 0, 10, 0)
            sort()
            device_spinup(ctx, torch, *cur[:4], n, 1e-4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in range(a.steps):
                ctx.step_dev(*[p(t) for t in cur[:4]], p(cur[4]), None, n, 1e-4, a.D, 10 + s, 1, 0)
                if (s + 1) % interval == 0:
                    sort()
            torch.cuda.synchronize()
            per = (time.perf_counter() - t0) / a.steps * 1e3
            print(json.dumps({"case": a.case, "particles": n, "D": a.D, "options": variant, "sort_interval": interval, "steps": a.steps,
                              "ms_per_step": round(per, 4), "frac": round(64 * n / (per * 1e-3) / 8e12, 4),
                              "kernel": ctx.step_kernel_name(a.D, 0)}), flush=True)
            if a.chunk > 0:
                from cudaparticlesfoam_amd.parallel import ShardedCloud
                ctx.use_own_stream()
                cloud = ShardedCloud(ctx, None, n + 4096)
                cloud.sort_interval = interval
                cloud.set_particles(*cur)
                cloud.step(1e-4, a.chunk, D=a.D, flags=4); cloud.arrays(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for s in range(0, a.steps, a.chunk):
                    cloud.step(1e-4, min(a.chunk, a.steps - s), D=a.D, flags=4)        # CPF_STEP_FUSE_CYCLES
                cloud.arrays(); ctx.synchronize()
                per = (time.perf_counter() - t0) / a.steps * 1e3
                print(json.dumps({"case": a.case, "particles": n, "D": a.D, "options": variant, "sort_interval": interval, "steps": a.steps,
                                  "through": "cpf_shard_step, %d cycles per call" % a.chunk, "ms_per_step": round(per, 4),
                                  "frac": round(64 * n / (per * 1e-3) / 8e12, 4)}), flush=True)
                cloud.close()
            ctx.close()


if __name__ == "__main__":
    main()
