"""CPU, world_size 2, 3 and 8, gloo: the sharded cloud's host logic -- ownership, counts exchange,
variable-size all-to-all, append -- gives the same per-particle results as one process."""
import os
import socket
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _single_process_answer(oracle_libs, u_step=0):
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
    if u_step:
        cw.step(x, y, z, c, 0.2, u_step, t, U)
        cw.step(x, y, z, c, 0.2, 30 - u_step, t, U[::-1].copy() * 0.5, step0=u_step)
    else:
        cw.step(x, y, z, c, 0.2, 30, t, U)
    return x, y, z, c


def _run_workers(world, out, *args, step_flags=0):
    env = dict(os.environ, OMP_NUM_THREADS="1", PYTHONPATH=ROOT, CPF_TEST_STEP_FLAGS=str(step_flags))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(HERE, "_gloo_worker.py"), out] + [str(a) for a in args]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def test_shard_grows_when_arrivals_exceed_its_slack(tmp_path, oracle_libs):
    """Capacity just above the initial share: the drifting cloud piles up on the downstream rank, whose arrays
    must be enlarged in flight (overlapped hand-off on) without losing or corrupting a particle."""
    out = str(tmp_path / "grow")
    _run_workers(2, out, 2, 0, 0, 1, 3050)                  # fixed ranges, hand-off every 2 steps, 1 overlapped
    x, y, z, c = _single_process_answer(oracle_libs)
    ds = [np.load(out + ".rank%d.npz" % r) for r in range(2)]
    assert sum(int(d["grown"]) for d in ds) >= 1
    assert sum(int(d["n_local"]) for d in ds) == 6000
    for d in ds:
        g = d["gid"]
        assert np.array_equal(d["x"], x[g]) and np.array_equal(d["y"], y[g]) and np.array_equal(d["z"], z[g])
        assert np.array_equal(d["cell"], c[g])


@pytest.mark.parametrize("overlap", [0, 3])
def test_send_buffer_overflow_grows_instead_of_dropping_or_hanging(overlap, tmp_path, oracle_libs):
    """send_fraction 0.01: the first hand-offs find far more leavers than the send buffer holds (every rank starts with
    an arbitrary slice of the cloud).  The split aborts without moving anything, EVERY rank sees that in the all-gathered
    table, the overflowing ranks enlarge their buffers and split again, and the run gives the particles of one process
    -- with the step loop running on in between (overlap 3) as well as synchronously."""
    out = str(tmp_path / "ovf")
    _run_workers(2, out, 2, 0, 0, overlap, 0, 0, 0.01)       # fixed ranges, hand-off every 2 steps
    x, y, z, c = _single_process_answer(oracle_libs)
    ds = [np.load(out + ".rank%d.npz" % r) for r in range(2)]
    assert sum(int(d["send_grown"]) for d in ds) >= 2 and sum(int(d["n_local"]) for d in ds) == 6000
    seen = np.zeros(6000, bool)
    for d in ds:
        g = d["gid"]; seen[g] = True
        assert int(d["total0"]) == 6000 and int(d["total1"]) == 6000
        assert np.array_equal(d["x"], x[g]) and np.array_equal(d["y"], y[g]) and np.array_equal(d["z"], z[g])
        assert np.array_equal(d["cell"], c[g])
    assert seen.all()


def test_velocity_update_inside_a_handoff_window(tmp_path, oracle_libs):
    """A transient solver sets a new U between two step() calls while a hand-off is still in flight (4 overlapped
    steps): ShardedCloud.set_velocity completes the hand-off first, so the arrivals replay their missed cycles with
    the field those cycles were stepped with -- same particles as one process."""
    out = str(tmp_path / "uchg")
    _run_workers(2, out, 0, 5, 0, 4, 0, 13)                 # re-cut + hand-off every 5 steps, 4 overlapped, U changes after 13
    x, y, z, c = _single_process_answer(oracle_libs, u_step=13)
    seen = np.zeros(6000, bool)
    for r in range(2):
        d = np.load(out + ".rank%d.npz" % r)
        g = d["gid"]; seen[g] = True
        assert np.array_equal(d["x"], x[g]) and np.array_equal(d["y"], y[g]) and np.array_equal(d["z"], z[g])
        assert np.array_equal(d["cell"], c[g])
    assert seen.all()


@pytest.mark.parametrize("world", [2, 3])
def test_velocity_slices_and_the_collective_gather(world, tmp_path, oracle_libs):
    """cpf_shard_set_velocity_slice: every rank hands over only the cells of its own (uneven) piece and the ranks all-gather the
    slices among themselves (an all-to-all-v with one shared source range) -- the same particles as with the whole field on every
    rank; and cpf_shard_gather brings the whole cloud to rank 0 in particle-id order."""
    out = str(tmp_path / "slices")
    _run_workers(world, out, 0, 5, 0, 4, 0, 13, 1.0, 1)
    x, y, z, c = _single_process_answer(oracle_libs, u_step=13)
    w = np.load(out + ".whole.npz")
    assert np.array_equal(w["xyzw"][:, 0], x) and np.array_equal(w["xyzw"][:, 1], y) and np.array_equal(w["xyzw"][:, 2], z)
    assert np.array_equal(w["cell"], c) and (w["xyzw"][:, 3] == 1).all()
    for r in range(world):
        d = np.load(out + ".rank%d.npz" % r)
        g = d["gid"]
        assert np.array_equal(d["x"], x[g]) and np.array_equal(d["cell"], c[g])


@pytest.mark.parametrize("world,interval,rebalance,overlap,u_step", [(2, 4, 0, 0, 0), (2, 4, 10, 3, 0), (3, 0, 5, 2, 13), (2, 0, 6, -1, 0)])
def test_fused_cycles_between_the_triggers(world, interval, rebalance, overlap, u_step, tmp_path, oracle_libs):
    """CPF_STEP_FUSE_CYCLES on cpf_shard_step: the cycles up to the next sort / hand-off / re-cut / completion of the hand-off in
    flight run inside one launch -- fewer launches than cycles, the same particles bit for bit (fixed and derived overlap depth,
    a field that changes between two calls)."""
    out = str(tmp_path / "fused")
    _run_workers(world, out, interval, rebalance, 0, overlap, 0, u_step, step_flags=4)
    x, y, z, c = _single_process_answer(oracle_libs, u_step=u_step) if u_step else _single_process_answer(oracle_libs)
    ds = [np.load(out + ".rank%d.npz" % r) for r in range(world)]
    assert sum(int(d["n_local"]) for d in ds) == 6000
    for d in ds:
        g = d["gid"]
        assert np.array_equal(d["x"], x[g]) and np.array_equal(d["y"], y[g]) and np.array_equal(d["z"], z[g])
        assert np.array_equal(d["cell"], c[g]) and bool(d["owned_ok2"])
        assert 0 < int(d["launches"]) <= 24        # (30 cycles; unfused: 30 launches + the catch-up replays)


def test_time_balancing_gives_the_slow_rank_fewer_particles(tmp_path, oracle_libs):
    """Re-cut by measured cost: rank 0 pretends to be 3x slower per particle, so the cuts must converge towards
    3*n0 == n1 (n0 -> N/4) instead of n0 == n1; the particle results stay those of one process."""
    out = str(tmp_path / "tb")
    _run_workers(2, out, 0, 5, 3.0)
    x, y, z, c = _single_process_answer(oracle_libs)
    ds = [np.load(out + ".rank%d.npz" % r) for r in range(2)]
    for d in ds:
        g = d["gid"]
        assert np.array_equal(d["x"], x[g]) and np.array_equal(d["y"], y[g]) and np.array_equal(d["z"], z[g])
        assert np.array_equal(d["cell"], c[g]) and bool(d["owned_ok2"])
    n0, n1 = int(ds[0]["n_local"]), int(ds[1]["n_local"])
    assert n0 + n1 == 6000 and 0.15 * 6000 < n0 < 0.36 * 6000, (n0, n1)


@pytest.mark.parametrize("world,interval,rebalance,overlap", [(2, 1, 0, 0), (2, 4, 0, 0), (3, 1, 0, 0), (2, 4, 10, 0),
                                                             (3, 0, 5, 0), (2, 0, 5, 3), (3, 4, 0, 4), (2, 0, 6, 10),
                                                             (2, 0, 6, -1),      # overlap depth derived from measured host time, agreed between the ranks (cpf_shard_core.h)
                                                             (8, 0, 15, 4)])     # eight ranks, overlapped hand-offs like the bench (the run is 30 steps: 15 divides it)
def test_sharded_equals_single_process(world, interval, rebalance, overlap, tmp_path, oracle_libs):
    """overlap > 0: the step loop keeps running for that many cycles after the split while counts and payload
    travel; the arrivals then replay the cycles they missed (same per-particle Philox stream, same result)."""
    out = str(tmp_path / "shard")
    _run_workers(world, out, interval, rebalance, 0, overlap)
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
    if rebalance and 30 % rebalance == 0:
        ds = [np.load(out + ".rank%d.npz" % r) for r in range(world)]
        assert all(int(d["rebalances"]) == 30 // rebalance for d in ds)
        # the last re-cut happened after the last step: the ranges must be THE equal-count cuts of the final
        # global per-cell histogram (cell granularity: a cell is never split), identical on every rank
        from cudaparticlesfoam_amd.parallel import slab_cell_ranges
        hist = np.bincount(c[c >= 0], minlength=240).astype(np.float64)
        want = slab_cell_ranges(hist, world)
        assert all(np.array_equal(d["cell_lo"], want) for d in ds)
        assert [int(d["n_local"]) for d in ds] == [int(((c >= want[r]) & (c < want[r + 1])).sum()) for r in range(world)]

def test_overflow_with_derived_overlap_depth(tmp_path, oracle_libs):
    """`overlap_steps = -1` AND a send buffer that overflows (round-4 advisory): the depth every rank derives from its own
    measured host time is agreed between the ranks through the counts table, so every rank completes a hand-off at the same
    step and an overflowing rank's repeated split "at the current step" is current on every receiver -- same particles as one
    process."""
    out = str(tmp_path / "ovfauto")
    _run_workers(2, out, 2, 0, 0, -1, 0, 0, 0.01)            # fixed ranges, hand-off every 2 steps, derived depth, tiny send buffer
    x, y, z, c = _single_process_answer(oracle_libs)
    ds = [np.load(out + ".rank%d.npz" % r) for r in range(2)]
    assert sum(int(d["send_grown"]) for d in ds) >= 2 and sum(int(d["n_local"]) for d in ds) == 6000
    for d in ds:
        g = d["gid"]
        assert int(d["total0"]) == 6000 and int(d["total1"]) == 6000
        assert np.array_equal(d["x"], x[g]) and np.array_equal(d["y"], y[g]) and np.array_equal(d["z"], z[g])
        assert np.array_equal(d["cell"], c[g])
