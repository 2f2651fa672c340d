import sys, os, json
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from _cases import make_case
from cudaparticlesfoam_amd.api import Context
dev = torch.device("cuda", 0)
for case, n in (("pitz", 10_000_000), ("tjunction", 4_000_000)):
    for age in (25, 50):
        ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        mesh, x, y, z, c, fields = make_case(case, ctx, torch, n, dev, None)
        ctx.set_velocity(list(fields.values())[-1])
        g = torch.arange(n, dtype=torch.int64, device=dev)
        p = lambda t: t.data_ptr()
        ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)
        ctx.step_dev(p(x), p(y), p(z), p(c), p(g), None, n, 1e-4, 1.5e-5, 0, age, 0)
        g = torch.arange(n, dtype=torch.int64, device=dev)
        ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)
        torch.cuda.synchronize()
        perm = g
        row = {"case": case, "age": age}
        for T in (1024, 2048, 4096):
            m = n // T * T
            t = perm[:m].view(-1, T)
            w = (t.max(1).values - t.min(1).values + 1).double()
            med = t.median(1).values
            row["T%d" % T] = {"win_med": float(w.median()), "win_p90": float(w.quantile(0.9)), "win_p99": float(w.quantile(0.99)),
                              **{"frac_tiles_le_%dT" % k: float((w <= k * T).double().mean()) for k in (2, 3, 4)},
                              **{"frac_elems_in_%dT_about_median" % k: float(((t - med[:, None]).abs() <= k * T // 2).double().mean()) for k in (2, 4)}}
        print(json.dumps(row), flush=True)
        ctx.close()
