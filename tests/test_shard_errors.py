"""CPU: the sharded cloud's argument and state checks (include/cpf.h "cpf_shard"): every misuse comes back as an error code with
a message -- through the product's own entry points (csrc/cpf_shard_abi.inc, compiled over the host stand-in) -- and leaves the
shard usable."""
import ctypes as C
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import pytest


@pytest.fixture()
def case(oracle_libs):
    import hostshard as H
    from cudaparticlesfoam_amd.cases import box_mesh
    mesh = box_mesh(6, 3, 2)
    cw = oracle_libs.CellWalk(); t = cw.build(mesh)
    U = np.tile([1.0, 0.2, 0.0], (mesh.n_cells, 1))
    hc = H.HostCase(t, U)
    yield H, mesh, hc
    hc.close()


def test_create_rejects_bad_ranges_and_sizes(case):
    from cudaparticlesfoam_amd import _lib as L
    H, mesh, hc = case
    for lo, cap in (([1, mesh.n_cells], 100), ([0, mesh.n_cells - 1], 100), ([0, mesh.n_cells], 0), ([0, mesh.n_cells], 1 << 31)):
        with pytest.raises(L.CpfError) as e:
            H.cloud(hc, lo, cap)
        assert e.value.status == L.CPF_ERR_ARG and ("cellLo" in str(e.value) or "capacity" in str(e.value))
    H.cloud(hc, None, 100).close()                            # (default ranges: fine)


def test_options_steps_and_slices_are_checked_and_the_shard_survives(case):
    from cudaparticlesfoam_amd import _lib as L
    H, mesh, hc = case
    cl = H.cloud(hc, None, 500)
    n = 300
    xyz = np.random.default_rng(1).uniform([0, 0, 0], [6, 3, 2], size=(n, 3))
    cl.set_particles(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), None, None)
    for key, value in (("no_such_option", 1), ("exchange_interval", -1), ("exchange_interval", 2.5), ("overlap_steps", -2),
                       ("send_fraction", 1.5), ("sort_interval", -3), ("step_index", -1)):
        with pytest.raises(L.CpfError) as e:
            cl.set_option(key, value)
        assert e.value.status == L.CPF_ERR_ARG and (key in str(e.value) or "unknown key" in str(e.value))
    with pytest.raises(L.CpfError) as e:
        cl.step(0.1, -1)
    assert "negative cycle count" in str(e.value)
    with pytest.raises(L.CpfError) as e:                      # one rank: its slice must be the whole field
        cl.set_velocity_slice(np.zeros((mesh.n_cells - 1, 3)))
    assert "do not add up" in str(e.value)
    with pytest.raises(L.CpfError) as e:
        cl.gather(3)
    assert "root out of range" in str(e.value)
    assert cl.lib.cpf_shard_set_option(cl.h, None, 1.0) == L.CPF_ERR_ARG
    assert cl.lib.cpf_shard_step(None, 0.1, 0.0, 1, 0) == L.CPF_ERR_ARG       # a null shard: an error, not a crash
    # ... and it still steps and answers
    cl.step(0.1, 3)
    g, x, y, z, c = cl.gather_to_numpy()
    assert g.size == n and cl.global_count() == n and cl.step_index == 3
    xyzw, cell, vel = cl.gather(0)
    assert xyzw.shape == (n, 4) and np.array_equal(np.sort(g), np.arange(n))
    cl.close()


def test_a_rank_that_fails_breaks_the_collective_for_everybody(case, oracle_libs):
    """Two rank threads; rank 1's communicator fails inside the counts all-gather: both ranks come back with an error (nobody
    hangs), the message names the collective."""
    from cudaparticlesfoam_amd import _lib as L
    H, mesh, hc = case
    cw = oracle_libs.CellWalk(); t = cw.build(mesh)
    U = np.tile([1.0, 0.2, 0.0], (mesh.n_cells, 1))
    group = H.ThreadGroup(2)
    errors = [None, None]

    def main(rank):
        c = H.HostCase(t, U)
        comm = H.ThreadComm(group, rank)
        if rank == 1:                                         # its all-gather raises
            def bad(_self, send, recv, nbytes, _stream):
                group.abort()
                return L.CPF_ERR_STATE
            comm._keep = (L.ALL_GATHER_FN(bad),) + comm._keep[1:]
            comm.struct.all_gather = comm._keep[0]
        cl = H.cloud(c, None, 400, comm, exchange_interval=1)
        xyz = np.random.default_rng(rank).uniform([0, 0, 0], [6, 3, 2], size=(100, 3))
        try:
            cl.set_particles(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), None, np.arange(100, dtype=np.int64) + 100 * rank)
            cl.exchange()
        except L.CpfError as e:
            errors[rank] = str(e)
        cl.close(); c.close()

    th = [threading.Thread(target=main, args=(r,)) for r in range(2)]
    [t_.start() for t_ in th]; [t_.join(timeout=120) for t_ in th]
    assert all(not t_.is_alive() for t_ in th)
    assert errors[0] and errors[1] and all("all_gather" in e for e in errors), errors


def test_stored_velocities_do_not_outlive_a_reordering(case):
    """Round-5 advisory: after a cycle with CPF_STEP_STORE_VEL the frame's velocities line up with the particles -- until an
    explicit cpf_shard_sort / exchange / refill reorders x, y, z, cell, gid without `vel`.  A frame gathered after that must
    not pair velocities with the wrong particles: it carries none (zeros), like a frame of a cycle that stored none."""
    from cudaparticlesfoam_amd import _lib as L
    H, mesh, hc = case
    U = np.zeros((mesh.n_cells, 3)); U[:, 0] = 1.0 + np.arange(mesh.n_cells)          # a velocity that names the cell
    cl = H.cloud(hc, None, 500)
    cl.set_velocity(U)
    n = 300
    xyz = np.random.default_rng(7).uniform([0.05, 0.05, 0.05], [5.95, 2.95, 1.95], size=(n, 3))
    cl.set_particles(xyz[::-1, 0].copy(), xyz[::-1, 1].copy(), xyz[::-1, 2].copy(), None, None)   # (unsorted on purpose)
    cl.step(0.0, 1, flags=L.STEP_STORE_VEL)                                             # the frame-0 idiom: nothing moves
    xyzw, cell, vel = cl.gather(0, want_vel=True)
    assert np.array_equal(vel[:, 0], U[cell, 0]) and vel[:, 0].min() >= 1.0              # aligned: every particle has ITS cell's U
    cl.sort()
    xyzw2, cell2, vel2 = cl.gather(0, want_vel=True)
    assert np.array_equal(xyzw2, xyzw) and np.array_equal(cell2, cell)                  # (gather is in particle-id order)
    assert not vel2[:, :3].any()                                                        # no velocities rather than misplaced ones
    cl.step(0.0, 1, flags=L.STEP_STORE_VEL)
    assert np.array_equal(cl.gather(0, want_vel=True)[2][:, 0], U[cell, 0])             # the next storing cycle restores them
    cl.close()
