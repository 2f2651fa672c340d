"""Worker for tests/test_parallel_gloo.py: one rank of a world_size-N gloo job on CPU."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

from cudaparticlesfoam_amd.cases import box_mesh                           # noqa: E402
from cudaparticlesfoam_amd.parallel import slab_cell_ranges, x_slab_renumbering  # noqa: E402
import hostshard as H                                                         # noqa: E402
from oracle import oracle as O                                                # noqa: E402


def main():
    out_path, interval = sys.argv[1], int(sys.argv[2])
    rebalance = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    overlap = int(sys.argv[5]) if len(sys.argv) > 5 else 0               # steps the loop runs on while a hand-off is in flight
    capacity = int(sys.argv[6]) if len(sys.argv) > 6 else 0              # 0: room for the whole cloud on every rank
    slow_rank0 = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0     # > 0: balance by "measured" time, rank 0 that much slower
    u_step = int(sys.argv[7]) if len(sys.argv) > 7 else 0                # > 0: the velocity field changes after that many steps
    send_fraction = float(sys.argv[8]) if len(sys.argv) > 8 else 1.0    # small: the send buffer overflows and must grow
    slices = int(sys.argv[9]) if len(sys.argv) > 9 else 0               # 1: the new field arrives as per-rank slices; + the collective gather
    flags = int(os.environ.get("CPF_TEST_STEP_FLAGS", "0"))             # 4 = CPF_STEP_FUSE_CYCLES: one launch up to the next trigger
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    m0 = box_mesh(12, 5, 4)
    c0, _ = m0.cell_centres_volumes()
    mesh = m0.renumber_cells(x_slab_renumbering(c0))
    centres, vols = mesh.cell_centres_volumes()
    rng = np.random.default_rng(5)
    U = rng.normal(size=(mesh.n_cells, 3)) + np.array([1.5, 0, 0])
    cw = O.CellWalk(); t = cw.build(mesh)
    n_total = 6000
    xyz = np.random.default_rng(8).uniform([0, 0, 0], [12, 5, 4], size=(n_total, 3))
    cell_lo = slab_cell_ranges(vols, world)
    # every rank starts with an arbitrary slice of the cloud (NOT its own slab): first exchange fixes that
    mine = np.arange(rank, n_total, world)
    # the product's hand-off logic (csrc/cpf_shard_core.h) over the host stand-in device, collectives over gloo
    case = H.HostCase(t, U, 2.0e-8 * (slow_rank0 if (slow_rank0 and rank == 0) else 1.0))
    cloud = H.cloud(case, cell_lo, capacity or (n_total + 16), H.GlooComm(dist), send_fraction=send_fraction,
                    exchange_interval=interval)
    cloud.set_particles(xyz[mine, 0].copy(), xyz[mine, 1].copy(), xyz[mine, 2].copy(), None, mine.astype(np.int64))
    cloud.exchange()
    g, x, y, z, c = cloud.gather_to_numpy()
    owned_ok = bool(((c >= cell_lo[rank]) & (c < cell_lo[rank + 1]) | (c < 0)).all())
    total0 = cloud.global_count()
    cloud.rebalance_interval = rebalance
    cloud.overlap_steps = overlap
    cloud.sort_interval = 7 if overlap else 0
    if slow_rank0:
        cloud.enable_time_balancing()
    if u_step:
        cloud.step(0.2, u_step, flags=flags)
        U2 = U[::-1].copy() * 0.5                          # a transient solver's new field, mid hand-off window
        if slices:
            # every rank hands over ITS slice only (the cells of its piece of a decomposed mesh, here an uneven cut); the
            # slices are all-gathered through the communicator's all-to-all-v (cpf_shard_set_velocity_slice)
            cut = [0] + [int(mesh.n_cells * (r + 1) ** 2 / world ** 2) for r in range(world)]
            cloud.set_velocity_slice(U2[cut[rank]:cut[rank + 1]])
        else:
            cloud.set_velocity(U2)
        cloud.step(0.2, 30 - u_step, flags=flags)
    else:
        cloud.step(0.2, 30, flags=flags)
    if interval > 1:
        cloud.exchange()
    total1 = cloud.global_count()
    g, x, y, z, c = cloud.gather_to_numpy()
    cell_lo = cloud.cell_lo
    owned_ok2 = bool(((c >= cell_lo[rank]) & (c < cell_lo[rank + 1]) | (c < 0)).all())
    whole = cloud.gather(0, want_vel=False) if slices else (None, None, None)     # collective: the cloud in particle-id order on rank 0
    if slices and rank == 0:
        np.savez(out_path + ".whole.npz", xyzw=whole[0], cell=whole[1])
    np.savez(out_path + ".rank%d.npz" % rank, gid=g, x=x, y=y, z=z, cell=c, owned_ok=owned_ok, owned_ok2=owned_ok2,
             total0=total0, total1=total1, handed=cloud.handed_off, exchanges=cloud.exchanges, rebalances=cloud.rebalances,
             n_local=cloud.n, launches=case.step_launches() if hasattr(case, 'step_launches') else -1, grown=cloud.grown, send_grown=cloud.send_grown, cell_lo=np.asarray(cell_lo))
    cloud.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
