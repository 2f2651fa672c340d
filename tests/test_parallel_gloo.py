"""CPU, world_size 2 (and 3), gloo: the sharded cloud's host logic -- ownership, counts exchange,
variable-size all-to-all, append -- gives the same per-particle results as one process."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _single_process_answer(oracle_libs):
    from cudaparticlesfoam_amd.cases import box_mesh
    from cudaparticlesfoam_amd.parallel import x_slab_renumbering
    m0 = box_mesh(12, 5, 4)
    c0, _ = m0.cell_centres_volumes()
    mesh = m0.renumber_cells(x_slab_renumbering(c0))
    rng = np.random.default_rng(5)
    U = rng.normal(size=(mesh.n_cells, 3)) + np.array([1.5, 0, 0])
    cw = oracle_libs.CellWalk(); t = cw.build(mesh)
    xyz = np.random.default_rng(8).uniform([0, 0, 0], [12, 5, 4], size=(6000, 3))
    x, y, z = (xyz[:, k].copy() for k in range(3))
    c = cw.locate_initial(x, y, z, t)
    cw.step(x, y, z, c, 0.2, 30, t, U)
    return x, y, z, c


@pytest.mark.parametrize("world,interval,rebalance", [(2, 1, 0), (2, 4, 0), (3, 1, 0), (2, 4, 10)])
def test_sharded_equals_single_process(world, interval, rebalance, tmp_path, oracle_libs):
    out = str(tmp_path / "shard")
    env = dict(os.environ, OMP_NUM_THREADS="1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(HERE, "_gloo_worker.py"), out, str(interval), str(rebalance)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    x, y, z, c = _single_process_answer(oracle_libs)
    seen = np.zeros(6000, bool)
    handed = 0
    for rank in range(world):
        d = np.load(out + ".rank%d.npz" % rank)
        assert bool(d["owned_ok"]) and bool(d["owned_ok2"])      # after an exchange every particle is with its owner
        assert int(d["total0"]) == 6000 and int(d["total1"]) == 6000   # nothing lost or duplicated in flight
        g = d["gid"]
        assert not seen[g].any()
        seen[g] = True
        assert np.array_equal(d["x"], x[g]) and np.array_equal(d["y"], y[g]) and np.array_equal(d["z"], z[g])
        assert np.array_equal(d["cell"], c[g])
        handed += int(d["handed"])
    assert seen.all() and handed > 0
    if rebalance:
        ds = [np.load(out + ".rank%d.npz" % r) for r in range(world)]
        assert all(int(d["rebalances"]) == 3 for d in ds)
        counts = [int(d["n_local"]) for d in ds]
        assert max(counts) - min(counts) <= 0.1 * 6000        # equal-count cuts (cell granularity)
