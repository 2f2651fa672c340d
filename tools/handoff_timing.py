#!/usr/bin/env python3
"""Developer tool (GPU box, 1 GPU): wall time of the hand-off path pieces with a 1-rank RCCL group."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import bench
from cudaparticlesfoam_amd import _lib as L
from cudaparticlesfoam_amd.api import Context
from cudaparticlesfoam_amd.cases import pitzdaily as pz
from cudaparticlesfoam_amd.parallel import HipOps, ShardedCloud, slab_cell_ranges, x_slab_renumbering

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
m0 = pz.pitzdaily_mesh(); c0, _ = m0.cell_centres_volumes(); mesh = m0.renumber_cells(x_slab_renumbering(c0))
cen, vol = mesh.cell_centres_volumes()
ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream); ctx.set_mesh(mesh); ctx.set_velocity(pz.uniform_u(mesh))
n = 10_000_000
x, y, z, c = bench.seed_in_fluid(ctx, torch, n, pz.DOMAIN_BOX, 1, dev)
cloud = ShardedCloud(HipOps(ctx), slab_cell_ranges(vol, 1), n + 4096, dev, 0, 1, send_fraction=1.0)
cloud.force_collectives = True
cloud.set_particles(x, y, z, c, torch.arange(n, dtype=torch.int64, device=dev))

def t(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

print("pack kernels only      %.3f ms" % t(lambda: cloud.ops.pack(cloud)))
print("pack + counts .cpu()   %.3f ms" % t(lambda: (cloud.ops.pack(cloud), torch.cat([cloud.counts_dev[:1], cloud.nstay_dev]).cpu())))
cloud.weights_dev = torch.zeros(mesh.n_cells, dtype=torch.float64, device=dev)
print("histogram (unsorted)   %.3f ms" % t(lambda: cloud.ops.histogram(cloud, 1.0)))
cloud.sort()
print("histogram (sorted)     %.3f ms" % t(lambda: cloud.ops.histogram(cloud, 1.0)))
print("all_reduce weights     %.3f ms" % t(lambda: dist.all_reduce(cloud.weights_dev)))
print("cell_ranges kernel     %.3f ms" % t(lambda: cloud.ops.cell_ranges(cloud)))
meta = torch.cat([cloud.counts_dev[:1], cloud.nstay_dev]); rows = [torch.empty_like(meta)]
print("all_gather counts      %.3f ms" % t(lambda: dist.all_gather(rows, meta)))
print("unpack (0 arrivals)    %.3f ms" % t(lambda: cloud.ops.unpack(cloud, cloud.n, cloud.recvbuf, 0)))
sc = torch.tensor([0], dtype=torch.int64, device=dev); rc = torch.empty_like(sc)
print("a2a counts             %.3f ms" % t(lambda: dist.all_to_all_single(rc, sc)))
print("a2a counts + .cpu()    %.3f ms" % t(lambda: (dist.all_to_all_single(rc, sc), rc.cpu())))
print("full exchange()        %.3f ms" % t(cloud.exchange))
print("full rebalance()       %.3f ms" % t(lambda: cloud.rebalance(mesh.n_cells)))
cloud.enable_time_balancing()
print("rebalance(), by time   %.3f ms" % t(lambda: cloud.rebalance(mesh.n_cells)))
print("sort()                 %.3f ms" % t(cloud.sort))
print("step                   %.3f ms" % t(lambda: cloud.ops.step(cloud, 1e-4, 0.0, 0, 1, 0)))
dist.destroy_process_group()
