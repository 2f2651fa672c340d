"""Worker for tests/test_bench_contract.py: one rank of bench.run() on CPU over gloo.

bench.py has no CPU path; this stand-in for its `GpuMachine` (the product's hand-off logic compiled over host memory with
the cell-walk oracle as its step -- tests/host_shard, tests/hostshard.py -- collectives over gloo) lets the N > 1 ORCHESTRATION of bench.run() -- seeding per slab, re-cut, overlapped hand-offs,
max-over-ranks timing, the one JSON line -- run without a GPU.  The numbers it prints are meaningless as performance.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench                                                                  # noqa: E402
import hostshard as H                                                         # noqa: E402
from oracle import oracle as O                                                # noqa: E402


class CpuMachine:
    collectives = "gloo all-to-all (CPU test double)"

    def __init__(self, rank, world):
        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        self.device = torch.device("cpu")
        self.cw = O.CellWalk()
        self.ctx = None
        self.comm = H.GlooComm(dist)
        self.comm_init_s = 0.0

    def set_case(self, mesh, U):
        self.t = self.cw.build(mesh)
        self.ctx = H.HostCase(self.t, U)            # the host stand-in's "context" (tests/hostshard.py)

    def make_cloud(self, cell_lo, capacity, **kw):
        # parallel.ShardedCloud -- the binding the GPU path uses -- on the product's hand-off logic compiled over host memory
        return H.cloud(self.ctx, cell_lo, capacity, self.comm, **kw)

    def sync(self):
        pass

    def seed_in_fluid(self, n, box, seed, cell_range=None):
        rng = np.random.default_rng(seed)
        xs, cs, have = [], [], 0
        while have < n:
            m = int((n - have) * 1.5) + 256
            p = rng.uniform(box[0], box[1], size=(m, 3))
            c = self.cw.locate_initial(p[:, 0].copy(), p[:, 1].copy(), p[:, 2].copy(), self.t)
            keep = c >= 0
            if cell_range is not None:
                keep &= (c >= cell_range[0]) & (c < cell_range[1])
            xs.append(p[keep]); cs.append(c[keep]); have += int(keep.sum())
        p = np.concatenate(xs)[:n]; c = np.concatenate(cs)[:n]
        return (torch.from_numpy(p[:, 0].copy()), torch.from_numpy(p[:, 1].copy()), torch.from_numpy(p[:, 2].copy()),
                torch.from_numpy(c.astype(np.int32)))

    def spinup(self, cloud, dt, ms):
        return None

    def extras(self, cloud, dt, args, box):
        return None, None, None, None, {}

    def finish(self):
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = bench.parse(sys.argv[1:])
    # (like bench.main: ranks started by somebody else's launcher carry their own watchdog)
    dog = bench.rank_watchdog(args, int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))
    bench.stage("rccl_init")
    if os.environ.get("BENCH_TEST_HANG_RANK") == os.environ.get("RANK"):
        import time
        time.sleep(3600)                                 # a rank that never reaches its first collective (test_bench_contract.py)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == args.gpus
    out = bench.run(args, CpuMachine(rank, world))
    if dog is not None:
        dog.cancel()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
