"""GPU: the randomised differential test of tests/test_shard_fuzz.py through the PRODUCT -- HIP split / unpack / histogram / cut
kernels, the streaming step kernel, the in-process communicator -- with the ranks as threads on the one GPU: every case must end
with every particle bit-identical to one process of the CPU checker."""
import ctypes as C
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import pytest

from test_shard_fuzz import CYCLES, DT, STORE_VEL, _draw, _fields

pytestmark = pytest.mark.gpu
N_TOTAL = 40_000


@pytest.mark.parametrize("block", range(int(os.environ.get("CPF_FUZZ_BLOCKS", "6"))))       # (a longer campaign: CPF_FUZZ_BLOCKS=60)
def test_random_worlds_on_the_gpu(block, oracle_libs, tmp_path):
    import torch
    from cudaparticlesfoam_amd import _lib as L
    from cudaparticlesfoam_amd.api import Context
    from cudaparticlesfoam_amd.cases import box_mesh
    from cudaparticlesfoam_amd.parallel import Communicator, ShardedCloud, slab_cell_ranges, unique_id, x_slab_renumbering
    m0 = box_mesh(10, 4, 3)
    c0, _ = m0.cell_centres_volumes()
    mesh = m0.renumber_cells(x_slab_renumbering(c0))
    _, vols = mesh.cell_centres_volumes()
    cw = oracle_libs.CellWalk(); t = cw.build(mesh)
    dev = torch.device("cuda", 0)
    tmp = str(tmp_path)
    tally = dict(handed=0, exchanges=0, grown=0, send_grown=0)
    for case_no in range(12):
        seed = 77_000 + 1000 * block + case_no
        rng = np.random.default_rng(seed)
        cfg = _draw(rng)
        W = cfg["world"]
        fields = _fields(mesh.n_cells, 1 + sum(c["new_u"] for c in cfg["calls"]), rng)
        xyz = rng.uniform([0, 0, 0], [10, 4, 3], size=(N_TOTAL, 3))
        # a third of the cases: the cloud is SEEDED by the shards (cpf_shard_seed_box: every rank draws its share of the one LCG
        # stream, the ranges are cut, everybody goes to its owner) from a box that sticks out of the domain
        seeded = bool(rng.integers(0, 3) == 0)
        box_lo, box_hi = np.array([-0.7, 0.2, 0.1]), np.array([9.0, 3.9, 2.8])
        # ---- ONE context on the GPU: the same kernels on the whole cloud (the counter-based kicks make the answer independent of
        # who holds a particle; with D = 0 it is also the CPU checker's, bit for bit)
        p = lambda t_: t_.data_ptr()   # noqa: E731
        one = Context(0); one.set_mesh(mesh); one.set_velocity(fields[0]); one.set_seed(0)
        one.set_stream(torch.cuda.current_stream().cuda_stream)
        tx, ty, tz = (torch.from_numpy(xyz[:, k].copy()).to(dev) for k in range(3))
        if seeded:
            one.seed_box_dev(p(tx), p(ty), p(tz), 0, N_TOTAL, box_lo, box_hi, 1)
        tc = torch.empty(N_TOTAL, dtype=torch.int32, device=dev)
        tg = torch.arange(N_TOTAL, dtype=torch.int64, device=dev)
        tv = torch.zeros(N_TOTAL, 3, dtype=torch.float64, device=dev)
        one.locate_initial_dev(p(tx), p(ty), p(tz), p(tc), N_TOTAL)
        torch.cuda.synchronize()
        lost0 = int((tc < 0).sum())
        assert (lost0 > 0) == seeded
        step0, fi, frames = 0, 0, []
        for call in cfg["calls"]:
            if call["new_u"]:
                fi += 1
                one.set_velocity(fields[fi])
            k = call["cycles"]
            if call["flags"] & STORE_VEL:                     # the frame: the state and the velocities of the call's last cycle
                if k > 1:
                    one.step_dev(p(tx), p(ty), p(tz), p(tc), p(tg), None, N_TOTAL, DT, cfg["D"], step0, k - 1, 0)
                tv.zero_()                                    # (a sharded frame: no velocity for particles that are not stepped)
                one.step_dev(p(tx), p(ty), p(tz), p(tc), p(tg), p(tv), N_TOTAL, DT, cfg["D"], step0 + k - 1, 1, STORE_VEL)
                torch.cuda.synchronize()
                frames.append(tuple(a_.cpu().numpy().copy() for a_ in (tx, ty, tz, tc, tv)))
            else:
                one.step_dev(p(tx), p(ty), p(tz), p(tc), p(tg), None, N_TOTAL, DT, cfg["D"], step0, k, 0)
            step0 += k
        torch.cuda.synchronize()
        x, y, z, c = (a_.cpu().numpy() for a_ in (tx, ty, tz, tc))
        one.use_own_stream(); one.close()
        if cfg["D"] == 0.0 and not seeded:
            hx, hy, hz = (xyz[:, k].copy() for k in range(3))
            hc = cw.locate_initial(hx, hy, hz, t, nthreads=cw.max_threads)
            fi = 0
            for call in cfg["calls"]:
                if call["new_u"]:
                    fi += 1
                cw.step(hx, hy, hz, hc, DT, call["cycles"], t, fields[fi], nthreads=cw.max_threads)
            assert np.array_equal(hx, x) and np.array_equal(hy, y) and np.array_equal(hz, z) and np.array_equal(hc, c), (seed, cfg)
        token = unique_id(L.COMM_INPROCESS)
        cell_lo = slab_cell_ranges(vols, W)
        out, errors = [None] * W, []
        cut = [0] + [int(mesh.n_cells * (r + 1) ** 2 / W ** 2) for r in range(W)]

        def rank_main(rank):
            try:
                ctx = Context(0)
                ctx.set_mesh(mesh); ctx.set_velocity(fields[0]); ctx.set_seed(0)
                comm = Communicator(token, rank, W, 0)
                cap = (N_TOTAL // W + 2000) if cfg["small_capacity"] else N_TOTAL + 16
                cloud = ShardedCloud(ctx, cell_lo, cap, comm, send_fraction=cfg["send_fraction"], exchange_interval=cfg["exchange"])
                if seeded:
                    n_out = cloud.seed_box(N_TOTAL, box_lo, box_hi)
                    assert n_out == lost0, (n_out, lost0)
                else:
                    mine = np.arange(rank, N_TOTAL, W)
                    tx, ty, tz = (torch.from_numpy(xyz[mine, k].copy()).to(dev) for k in range(3))
                    tg = torch.from_numpy(mine.astype(np.int64)).to(dev)
                    torch.cuda.synchronize()
                    cloud.set_particles(tx, ty, tz, None, tg)
                    cloud.exchange()
                cloud.rebalance_interval = cfg["rebalance"]
                cloud.overlap_steps = cfg["overlap"]
                cloud.sort_interval = cfg["sort"]
                if cfg["by_time"]:
                    cloud.enable_time_balancing()
                if cfg["frame0"]:
                    cloud.step(0.0, 1, D=0.0, flags=STORE_VEL)
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
                        if rank == 0:
                            badc = np.flatnonzero(wc != fc)
                            assert badc.size == 0, ("frame cells", badc.size, badc[:5], wc[badc[:5]], fc[badc[:5]])
                            badv = np.flatnonzero((wv[:, :3] != fv).any(axis=1))
                            assert badv.size == 0, ("frame velocities", badv.size, badv[:5], wv[badv[:3]], fv[badv[:3]], fc[badv[:5]])
                            badw = np.flatnonzero(xyzw[:, 3] != np.where(fc == -2, 0.0, 1.0))
                            assert badw.size == 0, ("frame w", badw.size, xyzw[badw[:5], 3], fc[badw[:5]])
                            assert np.array_equal(xyzw[:, 0], fx) and np.array_equal(xyzw[:, 1], fy) and np.array_equal(xyzw[:, 2], fz)
                        path = os.path.join(tmp, "f%d_%d.vtu" % (seed, nframe))
                        cloud.write_vtu(path, want_ke=bool(len(path) & 1))           # collective; formatted on the root's worker thread (with / without the energy)
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
                out[rank] = dict(g=g, x=gx, y=gy, z=gz, c=gc, total=total, lo=cloud.cell_lo.copy(), step=cloud.step_index,
                                 handed=cloud.handed_off, exchanges=cloud.exchanges, grown=cloud.grown, send_grown=cloud.send_grown)
                cloud.close(); comm.close(); ctx.close()
            except BaseException as e:                       # noqa: BLE001 -- reported by the main thread
                import traceback
                errors.append((rank, repr(e), traceback.format_exc()))

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(W)]
        for th in threads:
            th.start()
        for th in threads:
            th.join(timeout=600)
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
    assert tally["handed"] > 1000 and tally["exchanges"] > 10, tally
