"""CPU: randomised differential test of the sharded cloud's host logic (csrc/cpf_shard_core.h, compiled over the host stand-in of
tests/host_shard) against ONE process of the CPU checker.  The ranks are threads, the collectives a barrier and slots
(tests/hostshard.py ThreadComm): dozens of worlds, cadences, buffer sizes and call patterns per second, every one required to
end with every particle bit-identical to the single-process run -- positions, cells, and nobody lost or doubled."""
import ctypes as C
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import pytest

N_TOTAL, CYCLES, DT = 8000, 36, 0.2
FUSE, STORE_VEL = 4, 2
CASES_PER_BLOCK = 25


def _draw(rng):
    world = int(rng.integers(2, 6))
    ex = int(rng.choice([0, 1, 2, 3, 4, 7]))
    rb = int(rng.choice([0, 0, 3, 5, 6, 10]))
    if ex == 0 and rb == 0:
        ex = 2
    cfg = dict(world=world, exchange=ex, rebalance=rb, overlap=int(rng.choice([-1, 0, 1, 2, 3, 5])),
               sort=int(rng.choice([0, 0, 3, 7])), send_fraction=float(rng.choice([1.0, 1.0, 0.02])),
               small_capacity=bool(rng.integers(0, 2)), by_time=bool(rng.integers(0, 2)), D=float(rng.choice([0.0, 0.0, 0.02])),
               frame0=bool(rng.integers(0, 3) == 0))
    calls, left = [], CYCLES
    while left > 0:
        k = int(min(left, rng.integers(1, 10)))
        kind = rng.integers(0, 14)
        calls.append(dict(cycles=k, flags=(FUSE if rng.integers(0, 2) else 0) | (STORE_VEL if kind == 0 else 0),
                          new_u=(kind in (1, 2)), slices=(kind == 2), gather=(kind == 3),
                          # explicit collectives and knobs between two calls (the same on every rank): they change who holds what
                          # and in which order, never a particle
                          extra={4: "sort", 5: "exchange", 6: "rebalance", 7: "send_fraction", 8: "flush"}.get(int(kind))))
        left -= k
    cfg["calls"] = calls
    return cfg


def _fields(n_cells, count, rng):
    return [rng.normal(size=(n_cells, 3)) * 0.6 + np.array([1.2, 0, 0]) for _ in range(count)]


@pytest.mark.parametrize("block", range(int(os.environ.get("CPF_FUZZ_BLOCKS", "8"))))      # (a longer campaign: CPF_FUZZ_BLOCKS=200)
def test_random_worlds_cadences_and_call_patterns(block, oracle_libs, tmp_path):
    import hostshard as H
    from cudaparticlesfoam_amd.cases import box_mesh
    from cudaparticlesfoam_amd.parallel import slab_cell_ranges, x_slab_renumbering
    m0 = box_mesh(10, 4, 3)
    c0, _ = m0.cell_centres_volumes()
    mesh = m0.renumber_cells(x_slab_renumbering(c0))
    _, vols = mesh.cell_centres_volumes()
    cw = oracle_libs.CellWalk(); t = cw.build(mesh)
    tmp = str(tmp_path)
    tally = dict(handed=0, exchanges=0, grown=0, send_grown=0, launches=0)
    for case_no in range(CASES_PER_BLOCK):
        seed = 1000 * block + case_no
        rng = np.random.default_rng(seed)
        cfg = _draw(rng)
        W = cfg["world"]
        fields = _fields(mesh.n_cells, 1 + sum(c["new_u"] for c in cfg["calls"]), rng)
        xyz = rng.uniform([0, 0, 0], [10, 4, 3], size=(N_TOTAL, 3))
        gid_all = np.arange(N_TOTAL, dtype=np.int64)
        # ---- one process
        x, y, z = (xyz[:, k].copy() for k in range(3))
        c = cw.locate_initial(x, y, z, t)
        step0, fi, frames = 0, 0, []
        for call in cfg["calls"]:
            if call["new_u"]:
                fi += 1
            k = call["cycles"]
            if call["flags"] & STORE_VEL:                     # the frame: the state and the velocities of the call's last cycle
                if k > 1:
                    cw.step(x, y, z, c, DT, k - 1, t, fields[fi], D=cfg["D"], gid=gid_all, step0=step0, seed=0)
                v = np.zeros((N_TOTAL, 3))
                cw.step(x, y, z, c, DT, 1, t, fields[fi], vel_out=v, D=cfg["D"], gid=gid_all, step0=step0 + k - 1, seed=0)
                frames.append((x.copy(), y.copy(), z.copy(), c.copy(), v))
            else:
                cw.step(x, y, z, c, DT, k, t, fields[fi], D=cfg["D"], gid=gid_all, step0=step0, seed=0)
            step0 += k
        # ---- the same cloud over W rank threads
        group = H.ThreadGroup(W)
        cell_lo = slab_cell_ranges(vols, W)
        out, errors = [None] * W, []
        cut = [0] + [int(mesh.n_cells * (r + 1) ** 2 / W ** 2) for r in range(W)]         # uneven mesh pieces (velocity slices)

        def rank_main(rank):
            try:
                case = H.HostCase(t, fields[0], 2.0e-8 * (1.0 + rank if cfg["by_time"] else 1.0))
                comm = H.ThreadComm(group, rank)
                cap = (N_TOTAL // W + 150) if cfg["small_capacity"] else N_TOTAL + 16
                cloud = H.cloud(case, cell_lo, cap, comm, send_fraction=cfg["send_fraction"], exchange_interval=cfg["exchange"])
                mine = np.arange(rank, N_TOTAL, W)
                cloud.set_particles(xyz[mine, 0].copy(), xyz[mine, 1].copy(), xyz[mine, 2].copy(), None, mine.astype(np.int64))
                cloud.exchange()
                cloud.rebalance_interval = cfg["rebalance"]
                cloud.overlap_steps = cfg["overlap"]
                cloud.sort_interval = cfg["sort"]
                if cfg["by_time"]:
                    cloud.enable_time_balancing()
                if cfg["frame0"]:
                    cloud.step(0.0, 1, D=0.0, flags=STORE_VEL)                         # the frame-0 idiom: not a cycle of the run
                fi, nframe = 0, 0
                for call in cfg["calls"]:
                    if call["new_u"]:
                        fi += 1
                        if call["slices"]:
                            cloud.set_velocity_slice(fields[fi][cut[rank]:cut[rank + 1]])
                        else:
                            cloud.set_velocity(fields[fi])
                    cloud.step(DT, call["cycles"], D=cfg["D"], flags=call["flags"])
                    if call["flags"] & STORE_VEL:
                        fx, fy, fz, fc, fv = frames[nframe]; nframe += 1
                        xyzw, wc, wv = cloud.gather(0, want_vel=True)          # what cpf_shard_write_vtu formats
                        # the frame as the fragments write it (collective; formatted on the root's worker thread) == the product's
                        # writer on the one-process arrays, byte for byte
                        path = os.path.join(tmp, "f%d_%d.vtu" % (seed, nframe))
                        cloud.write_vtu(path, want_ke=bool(len(path) & 1))
                        assert cloud.lib.cpf_shard_write_vtu_wait(cloud.h) == 0
                        if rank == 0:
                            ref = path + ".ref"
                            fxyzw = np.column_stack([fx, fy, fz, np.where(fc == -2, 0.0, 1.0)])       # (w = 0: frozen, CPF_CELL_FROZEN)
                            fvel = np.column_stack([fv, -np.ones(N_TOTAL)])
                            ke = C.c_double()
                            assert cloud.lib.cpf_write_vtu_arrays(ref.encode(), N_TOTAL, fxyzw.ctypes.data_as(C.c_void_p),
                                                                  fc.ctypes.data_as(C.c_void_p), fvel.ctypes.data_as(C.c_void_p), C.byref(ke)) == 0
                            got = open(path, "rb").read()
                            assert len(got) > 100 * N_TOTAL and got == open(ref, "rb").read(), "frame file"
                            os.remove(path); os.remove(ref)
                            assert np.array_equal(xyzw[:, 0], fx) and np.array_equal(xyzw[:, 1], fy) and np.array_equal(xyzw[:, 2], fz)
                            bad = np.flatnonzero((wv[:, :3] != fv).any(axis=1))
                            assert np.array_equal(wc, fc) and bad.size == 0, ("frame velocities", bad.size, bad[:5], wv[bad[:3]], fv[bad[:3]], wc[bad[:5]])
                    if call["gather"]:
                        whole = cloud.gather(0)
                        assert (whole[0] is not None) == (rank == 0)
                    extra = call.get("extra")
                    if extra == "sort":
                        cloud.sort()
                    elif extra == "exchange":
                        cloud.exchange()
                    elif extra == "rebalance":
                        cloud.rebalance()
                    elif extra == "send_fraction":
                        cloud.send_fraction = 0.02 if cloud.send_fraction > 0.5 else 1.0
                    elif extra == "flush":
                        cloud.flush()
                cloud.flush()
                total = cloud.global_count()
                g, gx, gy, gz, gc = cloud.gather_to_numpy()
                lo = cloud.cell_lo
                out[rank] = dict(g=g, x=gx, y=gy, z=gz, c=gc, total=total, lo=lo.copy(), step=cloud.step_index, handed=cloud.handed_off,
                                 exchanges=cloud.exchanges, grown=cloud.grown, send_grown=cloud.send_grown, launches=case.step_launches())
                cloud.close(); case.close()
            except BaseException as e:                       # noqa: BLE001 -- reported by the main thread
                import traceback
                errors.append((rank, repr(e), traceback.format_exc()))
                group.abort()

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(W)]
        for th in threads:
            th.start()
        for th in threads:
            th.join(timeout=300)
        assert not errors, (seed, cfg, errors[:1])
        assert all(o is not None for o in out), (seed, cfg)
        seen = np.zeros(N_TOTAL, bool)
        for o in out:
            g = o["g"]
            assert not seen[g].any(), (seed, cfg)
            seen[g] = True
            assert o["total"] == N_TOTAL and o["step"] == CYCLES, (seed, cfg)
            ok = np.array_equal(o["x"], x[g]) and np.array_equal(o["y"], y[g]) and np.array_equal(o["z"], z[g]) and np.array_equal(o["c"], c[g])
            assert ok, (seed, cfg)
            assert np.array_equal(o["lo"], out[0]["lo"]), (seed, cfg)
        assert seen.all(), (seed, cfg)
        for k in tally:
            tally[k] += sum(int(o[k]) for o in out)
    # the draws did exercise the machinery: particles changed hands, shards and send buffers had to grow
    assert tally["handed"] > 1000 and tally["exchanges"] > 50 and tally["grown"] > 0 and tally["send_grown"] > 0, tally
    if os.environ.get("CPF_FUZZ_VERBOSE"):
        print(block, tally)
