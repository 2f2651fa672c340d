#!/usr/bin/env python3
"""Developer tool (GPU box, 1 GPU): wall time of the hand-off path's pieces with a one-rank RCCL communicator made by the
library (cpf_comm_create) -- the *_dev kernels one by one, then the shard layer's own exchange() / rebalance()."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from cudaparticlesfoam_amd import _lib as L
from cudaparticlesfoam_amd.api import Context
from cudaparticlesfoam_amd.cases import pitzdaily as pz
from cudaparticlesfoam_amd.parallel import Communicator, ShardedCloud, unique_id, x_slab_renumbering

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
m0 = pz.pitzdaily_mesh(); c0, _ = m0.cell_centres_volumes(); mesh = m0.renumber_cells(x_slab_renumbering(c0))
ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream); ctx.set_mesh(mesh); ctx.set_velocity(pz.uniform_u(mesh))
n = 10_000_000
x, y, z, c = bench.seed_in_fluid(ctx, torch, n, pz.DOMAIN_BOX, 1, dev)
g = torch.arange(n, dtype=torch.int64, device=dev)
p = lambda a: a.data_ptr()   # noqa: E731


def t(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


lo = torch.tensor([0, mesh.n_cells], dtype=torch.int32, device=dev)
sendbuf = torch.empty(n * L.HANDOFF_DOUBLES, dtype=torch.float64, device=dev)
counts = torch.zeros(1, dtype=torch.int64, device=dev); nstay = torch.zeros(1, dtype=torch.int64, device=dev)
w = torch.zeros(mesh.n_cells, dtype=torch.float64, device=dev)
print("pack kernels only      %.3f ms" % t(lambda: ctx.pack_leavers_dev(p(x), p(y), p(z), p(c), p(g), n, p(lo), 1, 0, p(sendbuf), n, p(counts), p(nstay))))
print("histogram (unsorted)   %.3f ms" % t(lambda: ctx.cell_histogram_dev(p(c), n, 1.0, p(w))))
print("cell_ranges kernel     %.3f ms" % t(lambda: ctx.cell_ranges_dev(p(w), 1, p(lo))))
print("unpack (0 arrivals)    %.3f ms" % t(lambda: ctx.unpack_arrivals_dev(p(x), p(y), p(z), p(c), p(g), n, p(sendbuf), 0)))
comm = Communicator(unique_id(L.COMM_RCCL), 0, 1, 0)
cloud = ShardedCloud(ctx, None, n + 4096, comm, send_fraction=1.0, exchange_interval=0)
cloud.force_collectives = True
cloud.set_particles(x, y, z, c, g)
print("full exchange()        %.3f ms" % t(cloud.exchange))
print("full rebalance()       %.3f ms" % t(cloud.rebalance))
cloud.sort()
print("rebalance(), sorted    %.3f ms" % t(cloud.rebalance))
cloud.enable_time_balancing()
print("rebalance(), by time   %.3f ms" % t(cloud.rebalance))
st = cloud.stats()
print("host ms per hand-off %.3f (of which waiting %.3f)" % (st.handoffHostMs / max(1, st.exchanges), st.handoffWaitMs / max(1, st.exchanges)))
cloud.close(); comm.close(); ctx.close()
