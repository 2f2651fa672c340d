#!/usr/bin/env python3
"""Headline benchmark: Mparticle-steps/s of the fused advect+locate+reflect+move cycle on the
pitzDaily mesh (12 225 cells), uniform U = (10,0,0) m/s, dt = 1e-4 s, D = 0, fp64 particles seeded over the
whole fluid domain.  --gpus 1: 1e7 particles (BASELINE.json configs[2]).  --gpus N > 1:
the north star's scaling experiment, 1e8 particles IN TOTAL sharded over the N ranks (configs[3]; "strong").

  python bench.py --gpus N --steps K --warmup W            (N > 1: starts its own N ranks as a child process)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one Lagrangian cycle of the whole cloud (one launch of the fused kernel per rank,
plus, when N > 1, the ownership re-cut + hand-off every --rebalance-interval steps).  Inputs are resident in HBM when the
timed region starts.  Rank 0 prints ONE JSON line.

Layout: `run(args, M)` is the measurement; everything it needs from the machine -- device, context, collectives,
seeding, the single-GPU extras -- comes from `M` (`GpuMachine`: one MI355X per rank, RCCL).  tests/ drive the same
`run()` with a CPU stand-in for `M` over gloo to check the N > 1 orchestration and the JSON contract without a GPU;
this script itself has no CPU path.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_PARTICLE_STEP = 56      # fp64 SoA: read x,y,z (24) + cell (4), write x,y,z (24) + cell (4)
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--particles", type=float, default=None,
                    help="particles per GPU (weak) / in total (strong); default 1e7 at --gpus 1, 1e8 in total at --gpus N > 1")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="default: weak at --gpus 1 (one GPU, 1e7), strong at --gpus N > 1 (1e8 in total)")
    ap.add_argument("--field", choices=["uniform", "analytic"], default="uniform")
    ap.add_argument("--exchange-interval", type=int, default=0,
                    help="N>1: hand-off with FIXED cell ranges every that many steps (0 = only inside the re-cuts)")
    ap.add_argument("--rebalance-interval", type=int, default=-1,
                    help="N>1: re-cut the cell ranges to equal cost + hand-off every that many steps, COUNTED FROM THE FIRST TIMED "
                         "STEP; -1 (default) = max(2, min(32, steps // 2)): the fragments' production cadence (32) where the timed "
                         "region is long enough for two of those, else as close to it as keeps TWO re-cuts + all-to-all-v hand-offs "
                         "inside the clock whatever --steps is -- the first overlapped with the step loop and completed inside it, "
                         "the second completed by the flush before the clock stops (with --steps 20: after timed steps 10 and 20)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--balance", choices=["time", "count"], default="time",
                    help="N>1: re-cut the ranges to equal MEASURED step time per rank (default) or equal particle counts")
    ap.add_argument("--overlap-steps", type=int, default=-1,
                    help="N>1: cycles the step loop runs on while a hand-off's counts and payload are in flight; -1 = derived "
                         "per rank from the measured host time of a hand-off: ceil(host_ms / step_ms) + 1 (parallel.py, _overlap)")
    ap.add_argument("--fused-extra", type=int, default=10,
                    help="after the timed region, also time this many launches of 8 fused cycles -- what the replacement "
                         "advect.H does between two output points (extra field config.extra_fused_cycles, never `value`; "
                         "0 = skip)")
    ap.add_argument("--spinup-ms", type=float, default=100.0,
                    help="before the warm-up steps, keep the device busy this long with step launches on a SCRATCH copy "
                         "of the cloud (discarded): an MI355X needs ~30 ms of load after an idle phase to reach its "
                         "steady clocks (tools/drift_check.py: 0.148 -> 0.119 ms per launch of the same work), and the "
                         "set-up before the timed region leaves it idle for seconds; 0 = off")
    ap.add_argument("--timing-stride", type=int, default=10,
                    help="HIP-event pair around every k-th step launch of the timed region (roofline.kernel_avg_ms).  The stamps are "
                         "not free, by an amount that depends on the box: 100 steps, every 4th / 10th / 25th launch: 85.4-86.2 / "
                         "87.1-88.5 / 88.9-90.4 G on one box, 87.4 against 88.0 G (4th / 20th, six alternating runs each) on another")
    ap.add_argument("--dry-collectives", action="store_true",
                    help="N>1 (or --force-dist): run ONLY communicator init -> first re-cut -> one all-to-all-v with per-stage "
                         "timings, print them as the one JSON line and exit: a failing scaling run then costs seconds and "
                         "names the collective")
    ap.add_argument("--force-dist", action="store_true",
                    help="single rank, but still create the RCCL group and run hand-off + rebalance (smoke of the N>1 path)")
    ap.add_argument("--self-launch", action="store_true",
                    help="start the ranks through torch.distributed.run even for --gpus 1 (what --gpus N > 1 does by "
                         "itself when no launcher set WORLD_SIZE)")
    ap.add_argument("--launch-timeout", type=float, default=900.0,
                    help="self-launched ranks (--gpus N without a launcher): wall-clock limit of the child process tree; on expiry "
                         "it is killed and ONE JSON line {\"error\": ..., \"stage\": ...} is printed, exit code 124")
    ap.add_argument("--collective-timeout", type=float, default=120.0,
                    help="N>1: torch.distributed process-group timeout in seconds -- a collective that does not complete raises "
                         "in every rank instead of hanging the job")
    ap.add_argument("--ipc-legacy", choices=["0", "1", "unset", "inherit"], default="inherit",
                    help="HSA_ENABLE_IPC_MODE_LEGACY for the ranks: inherit = keep the environment's value and export 0 if it has "
                         "none (this pool's driver only supports dmabuf IPC); unset = remove it; 0 / 1 = force")
    ap.add_argument("--child-script", default=None, help=argparse.SUPPRESS)      # tests: the rank program self_launch starts
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-sort", action="store_true")
    ap.add_argument("--sort-interval", type=int, default=100, help="re-sort the cloud by cell every that many steps")
    ap.add_argument("--steady-steps", type=int, default=-1,
                    help="after the timed region: that many more steps (default: one full sort interval, N = 1 only) "
                         "timed the same way, reported as ms_per_step_steady (the periodic re-sort included); 0 = skip")
    ap.add_argument("--brownian-extra", type=int, default=20,
                    help="after the timed region (N = 1): that many launches with the tutorial's D = 1.5e-5, reported "
                         "as config.brownian; 0 = skip")
    ap.add_argument("--no-fused-tutorial", action="store_true",
                    help="leave the fused launches (fused_8_cycles_per_launch, fragment_calls) out of brownian_steady / tjunction_as_run: "
                         "a profile of the run then holds that kernel's single-cycle launches only")
    ap.add_argument("--brownian-steady-steps", type=int, default=100,
                    help="after the timed region (N = 1): a freshly seeded cloud of the same size stepped that many times with "
                         "the tutorial's D = 1.5e-5 and the fragments' own sort interval for diffusing clouds (25), sorts "
                         "included, reported as config.brownian_steady; 0 = skip")
    ap.add_argument("--analytic-extra", type=int, default=20,
                    help="after the timed region (N = 1): that many steps of the frozen analytic step-flow field (SURVEY.md 8d "
                         "config 3(ii)) on a freshly seeded, sorted cloud, reported as config.analytic_field; 0 = skip")
    ap.add_argument("--tjunction-steps", type=int, default=100,
                    help="after the timed region (N = 1): the reference's second tutorial as its dictionary runs it -- 4e6 "
                         "particles from the seeding box, D = 1.5e-5, on the 248 000-cell TJunction mesh -- that many steps, "
                         "reported as config.tjunction_as_run; 0 = skip")
    ap.add_argument("--tjunction-particles", type=float, default=4e6)
    ap.add_argument("--vertex-steps", type=int, default=20,
                    help="after the timed region (N = 1): that many cycles with the reference's \"VertexVelocity\" advect (velocity "
                         "interpolated at the particle from tet-vertex values, cuda/particles.cu:244-313; CPF_STEP_VERTEX_VELOCITY) on "
                         "pitzDaily with the analytic field sampled at the vertices, reported as config.vertex_velocity; 0 = skip")
    ap.add_argument("--polyhedral-steps", type=int, default=50,
                    help="after the timed region (N = 1): the non-hex share of BASELINE configs[4] -- a 114 540-cell 3-D mesh with a "
                         "2:1-refined block (9-faced cells, face groups), D = 1.5e-5, a new U uploaded every 10 cycles -- that many "
                         "cycles, reported as config.polyhedral_as_run; 0 = skip")
    ap.add_argument("--polyhedral-particles", type=float, default=1e7)
    ap.add_argument("--anchor-particles", type=float, default=1e8,
                    help="after the timed region (N = 1): the strong-scaling experiment's total cloud (what --gpus N > 1 "
                         "shards) stepped on this ONE GPU, reported as config.strong_anchor_1e8 -- the N = 1 point of "
                         "the strong-scaling curve, never `value`; 0 = skip")
    ap.add_argument("--anchor-steps", type=int, default=10)
    a = ap.parse_args(argv)
    if a.scaling is None:
        a.scaling = "weak" if a.gpus == 1 else "strong"
    if a.particles is None:
        a.particles = 1e7 if (a.gpus == 1 or a.scaling == "weak") else 1e8
    a.rebalance_interval_auto = a.rebalance_interval < 0
    if a.rebalance_interval_auto:
        a.rebalance_interval = max(2, min(32, a.steps // 2))
    return a


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_baseline(mesh, centres, U, seconds, batch_cycles=50):
    """The reference's own functions (oracle/_ref, kind "reference") -- or the C restatement of the same algorithm (kind
    "port") -- timed on this box's host cores on a bounded sample of the same workload, IN THE TIMED REGION'S REGIME: 2e5
    particles seeded over the fluid domain take `batch_cycles` cycles (5 cm of travel at most, like the bench's warm-up + timed
    steps) and are then put back where they started -- not thousands of cycles into the outlet wall, as until round 4.  All
    threads the container may use, then one thread (a sixth of the time budget).  Plus the probe for an OpenFOAM installation
    (north_star's kinematicCloud baseline): tools/openfoam_baseline.py."""
    from oracle import oracle as O
    from oracle.tetmesh import poly_to_tets
    from cudaparticlesfoam_amd.cases import pitzdaily as pz
    O.build()
    cw = O.CellWalk()
    if O.have_ref():
        lib, kind = O.RefLib(), "reference"
    else:
        lib, kind = O.TetWalk(), "port"
    pos, tets, tcell, tu = poly_to_tets(mesh, centres, U)
    m = lib.tables(pos, tets, tu)
    t = cw.build(mesh)
    n = 200000
    xyz = pz.uniform_points(4242, int(n * 1.4), *pz.DOMAIN_BOX)
    cell = cw.locate_initial(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), t, nthreads=cw.max_threads)
    xyz = xyz[cell >= 0][:n]; cell = cell[cell >= 0][:n]
    n = xyz.shape[0]
    P0 = np.zeros((n, 4)); P0[:, :3] = xyz; P0[:, 3] = 1
    ids0 = (cell * 12).astype(np.int32)
    lib.bary_query(P0, ids0, m, lib.max_threads)
    vels = np.zeros((n, 4)); disps = np.zeros((n, 4))

    def timed(threads, budget):
        """batches of `batch_cycles` cycles from the seeded state until the budget is used; the resets are outside the clock"""
        done, el = 0, 0.0
        while el < budget and done < 200000:
            P, ids = P0.copy(), ids0.copy()
            vels[:] = 0; disps[:] = 0
            t0 = time.perf_counter(); lib.cycles(P, ids, vels, disps, 1e-4, batch_cycles, m, threads); el += time.perf_counter() - t0
            done += batch_cycles
        return n * done / el / 1e6, done, el

    # the container may own only a slice of the box's hardware threads: pick the team size that is fastest
    best = None
    th = lib.hw_threads
    while th >= 1:
        P, ids = P0.copy(), ids0.copy()
        t0 = time.perf_counter(); lib.cycles(P, ids, vels, disps, 1e-4, 2, m, th); cal = (time.perf_counter() - t0) / 2
        if best is None or cal < best[1]:
            best = (th, cal)
        th //= 2
    th = best[0]
    rate, cycles, el = timed(th, seconds)
    rate1, cycles1, el1 = timed(1, max(1.0, seconds / 6.0))
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import openfoam_baseline as ofb
        openfoam = ofb.probe()
        if openfoam["available"]:
            openfoam = ofb.run(mesh, U, xyz[:20000], steps=batch_cycles)
    except Exception as e:                                      # noqa: BLE001
        openfoam = {"available": False, "error": repr(e)[:200]}
    return dict(value=round(rate, 3), unit="Mparticle-steps/s", cores=int(th), kind=kind, cpu_model=cpu_model(),
                hw_threads=int(lib.hw_threads), one_thread=round(rate1, 3),
                sample="%d particles seeded over the fluid domain x %d cycles in batches of %d from the seeded state (the timed "
                       "region's regime), pitzDaily 146700-tet decomposition, uniform U, OpenMP over particles, %.1f s; one thread: "
                       "%d cycles, %.1f s" % (n, cycles, batch_cycles, el, cycles1, el1),
                openfoam_kinematicCloud=openfoam)


def seed_in_fluid(ctx, torch, n, box, seed, device, cell_range=None, chunk=20_000_000):
    """n points uniform in `box`, rejection-resampled until located in a cell;
    with cell_range=(lo, hi) only points whose cell lies in [lo, hi) are kept (a rank seeding its own slab)."""
    g = torch.Generator(device=device); g.manual_seed(seed)
    lo = torch.tensor(box[0], dtype=torch.float64, device=device)
    ext = torch.tensor(box[1], dtype=torch.float64, device=device) - lo
    xs, ys, zs, cs = [], [], [], []
    have = 0
    while have < n:
        m = min(int((n - have) * 1.35) + 1024, chunk)
        u = torch.rand((3, m), generator=g, dtype=torch.float64, device=device)
        x = (lo[0] + u[0] * ext[0]).contiguous(); y = (lo[1] + u[1] * ext[1]).contiguous()
        z = (lo[2] + u[2] * ext[2]).contiguous()
        del u
        c = torch.empty(m, dtype=torch.int32, device=device)
        ctx.locate_initial_dev(x.data_ptr(), y.data_ptr(), z.data_ptr(), c.data_ptr(), m)
        torch.cuda.synchronize()
        keep = c >= 0
        if cell_range is not None:
            keep &= (c >= cell_range[0]) & (c < cell_range[1])
        xs.append(x[keep]); ys.append(y[keep]); zs.append(z[keep]); cs.append(c[keep])
        have += int(keep.sum())
    cat = lambda l: torch.cat(l)[:n].contiguous()   # noqa: E731
    return cat(xs), cat(ys), cat(zs), cat(cs)


_STAGE = "launch"
STAGES = ("launch", "rccl_init", "first_exchange", "warmup", "timed_region", "extras", "done")


def stage(name):
    """Rank 0 leaves a trail of where the run is (the parent's watchdog reports the last entry when it has to kill the
    ranks): one line on stderr and, if BENCH_STAGE_FILE is set (self_launch sets it), the name in that file."""
    global _STAGE
    _STAGE = name
    if int(os.environ.get("RANK", "0")) != 0:
        return
    print("[bench stage] %s" % name, file=sys.stderr, flush=True)
    path = os.environ.get("BENCH_STAGE_FILE")
    if path:
        try:
            with open(path, "w") as f:
                f.write(name)
        except OSError:
            pass


def rank_watchdog(args, rank, world):
    """Ranks started by somebody else's launcher (the driver's torch.distributed.run) have no parent of ours watching them:
    every rank of an N > 1 run arms a timer of its own.  On expiry rank 0 prints the error line (the run's one JSON line
    then) and every rank leaves with os._exit(124) -- from a thread, so a main thread stuck inside a collective does not
    matter."""
    import threading
    if world <= 1 or args.launch_timeout <= 0:
        return None

    def expire():
        if rank == 0:
            print(json.dumps({"error": "rank watchdog: no result after %.0f s" % args.launch_timeout, "stage": _STAGE,
                              "n_gpus": world}), flush=True)
        print("[bench rank %d] watchdog expired in stage %s" % (rank, _STAGE), file=sys.stderr, flush=True)
        os._exit(124)
    # (rank 0 first: it owns the error line, and a launcher that sees another rank die may take rank 0 down before it has spoken)
    t = threading.Timer(args.launch_timeout + (0.0 if rank == 0 else 5.0), expire)
    t.daemon = True
    t.start()
    return t


def rank_env(args, base=None):
    """Environment of the rank processes (see --ipc-legacy)."""
    env = dict(os.environ if base is None else base)
    if args.ipc_legacy == "inherit":
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    elif args.ipc_legacy == "unset":
        env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    else:
        env["HSA_ENABLE_IPC_MODE_LEGACY"] = args.ipc_legacy
    return env


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks here as a CHILD process tree -- this process has
    not touched the GPU and never does -- the way the driver starts them for N > 1, forward rank 0's one JSON line as
    this process's only stdout line (anything else the ranks wrote to stdout goes to stderr) and return the child's
    exit code.  The child runs under a wall-clock limit (--launch-timeout): when it expires the whole process group is
    killed and the one stdout line is {"error": ..., "stage": ..., "n_gpus": N} with exit code 124 -- a hang becomes a
    diagnosis.  Never a re-exec: a fresh child or a non-zero exit.
    (The reference has no counterpart: one GPU, driven by the MPI master only, src/advect.H:59-89.)"""
    import signal
    import socket
    import subprocess
    import tempfile
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    argv, skip = [], False
    for a in sys.argv[1:]:
        if skip:
            skip = False
        elif a == "--child-script":
            skip = True
        elif a != "--self-launch" and not a.startswith("--child-script="):
            argv.append(a)
    script = os.path.abspath(args.child_script or __file__)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + argv
    fd, stage_file = tempfile.mkstemp(prefix="bench_stage_"); os.close(fd)
    env = rank_env(args)
    env["BENCH_STAGE_FILE"] = stage_file
    with open(stage_file, "w") as f:
        f.write("launch")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, start_new_session=True)
    timed_out = False
    try:
        out, _ = proc.communicate(timeout=args.launch_timeout if args.launch_timeout > 0 else None)
    except subprocess.TimeoutExpired:
        timed_out = True
        for sig in (signal.SIGTERM, signal.SIGKILL):          # the launcher, its ranks and whatever they started: one group
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        try:
            out, _ = proc.communicate(timeout=10)
        except Exception:
            out = ""
    try:
        last = open(stage_file).read().strip() or "launch"
    except OSError:
        last = "launch"
    finally:
        try:
            os.unlink(stage_file)
        except OSError:
            pass
    json_lines = []
    for ln in (out or "").splitlines():
        if ln.startswith("{"):
            json_lines.append(ln)
        else:
            print(ln, file=sys.stderr)
    if timed_out:
        print(json.dumps({"error": "ranks killed after --launch-timeout %.0f s" % args.launch_timeout, "stage": last,
                          "n_gpus": args.gpus}), flush=True)
        return 124
    if proc.returncode != 0:
        # (rank 0's own error line, if it got one out, says more than this process can)
        print(json_lines[-1] if json_lines else
              json.dumps({"error": "ranks exited with code %d" % proc.returncode, "stage": last, "n_gpus": args.gpus}), flush=True)
        return proc.returncode
    if len(json_lines) != 1:
        print("bench.py: expected ONE JSON line from rank 0, got %d" % len(json_lines), file=sys.stderr)
        print(json.dumps({"error": "expected one JSON line from rank 0, got %d" % len(json_lines), "stage": last,
                          "n_gpus": args.gpus}), flush=True)
        return 3
    print(json_lines[-1], flush=True)
    return 0


class GpuMachine:
    """What `run()` needs from the machine: one MI355X per rank through the C-ABI, collectives over RCCL.
    No CPU fallback: without a GPU or without the HIP library the constructor exits."""

    collectives = "RCCL all-gather + all-reduce + grouped send/recv all-to-all-v"

    def __init__(self, args, rank, world, local):
        import torch
        import torch.distributed as dist
        from cudaparticlesfoam_amd.api import Context
        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        if not torch.cuda.is_available():
            sys.exit("bench.py needs a GPU (there is no CPU fallback for the product path)")
        torch.cuda.set_device(local)
        self.device = torch.device("cuda", local)
        self.dist_on = world > 1 or args.force_dist
        self.comm = None
        self.control = False                                # a torch.distributed (gloo) group for the control plane exists
        if self.dist_on:
            # stdout belongs to the ONE JSON line.  This image exports NCCL_DEBUG=VERSION, which makes every rank print a
            # five-line banner on stdout (NCCL_DEBUG_FILE does not move it): ask for warnings only instead, send whatever
            # else RCCL logs to stderr, and print the JSON after everything else has been flushed (main)
            if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
                os.environ["NCCL_DEBUG"] = "WARN"
            os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
            import datetime
            from cudaparticlesfoam_amd import _lib as L
            from cudaparticlesfoam_amd.parallel import Communicator, unique_id
            tmo = datetime.timedelta(seconds=max(1.0, args.collective_timeout))
            # who is where, before the first collective (stderr; one line per rank)
            print("[bench rank %d/%d] device cuda:%d of %d visible, pid %d, HSA_ENABLE_IPC_MODE_LEGACY=%s"
                  % (rank, world, local, torch.cuda.device_count(), os.getpid(), os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")),
                  file=sys.stderr, flush=True)
            stage("rccl_init")
            t_init = time.perf_counter()
            # CONTROL plane (rendezvous token, barriers, the max-over-ranks clock): torch.distributed over gloo on host memory.
            # DATA plane: the library's own RCCL communicator (cpf_comm_create: ncclCommInitRank from the token; the hand-off's
            # all-gather / all-reduce / grouped send-recv all-to-all-v are issued by csrc/cpf_shard_core.h, not from Python)
            token = [None]
            if world > 1:
                dist.init_process_group("gloo", timeout=tmo)
                self.control = True
                if rank == 0:
                    token[0] = unique_id(L.COMM_RCCL)
                dist.broadcast_object_list(token, src=0)
            else:
                token[0] = unique_id(L.COMM_RCCL)
            self.comm = Communicator(token[0], rank, world, local)
            self.comm_init_s = time.perf_counter() - t_init
            if rank == 0:
                print("[bench] rccl_ranks %d (cpf_comm_create %.2f s)" % (self.comm.world, self.comm_init_s), file=sys.stderr, flush=True)
        self.ctx = Context(local)
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    def set_case(self, mesh, U):
        self._case = (mesh, U)
        self.ctx.set_mesh(mesh)
        self.ctx.set_velocity(U)

    def make_cloud(self, cell_lo, capacity, **kw):
        from cudaparticlesfoam_amd.parallel import ShardedCloud
        return ShardedCloud(self.ctx, cell_lo, capacity, self.comm, **kw)

    def sync(self):
        self.torch.cuda.synchronize()

    def seed_in_fluid(self, n, box, seed, cell_range=None):
        return seed_in_fluid(self.ctx, self.torch, n, box, seed, self.device, cell_range)

    def spinup(self, cloud, dt, ms):
        """Device spin-up (see --spinup-ms) on a SCRATCH copy of the rank's shard as it stands (sorted): the cloud itself is
        untouched, the W warm-up steps and the K timed steps are the first steps it ever takes."""
        torch, ctx = self.torch, self.ctx
        a = cloud.arrays()                               # device addresses of the shard's arrays (cpf_shard_arrays)
        ns = int(a["n"])
        if ms <= 0 or ns <= 0:
            return None
        sp = lambda t: t.data_ptr()   # noqa: E731
        sx, sy, sz = (torch.empty(ns, dtype=torch.float64, device=self.device) for _ in range(3))
        sc = torch.empty(ns, dtype=torch.int32, device=self.device)
        for dst, key, width in ((sx, "x", 8), (sy, "y", 8), (sz, "z", 8), (sc, "cell", 4)):
            ctx._ck(ctx.lib.cpf_copy_dev(ctx.h, dst.data_ptr(), a[key], ns * width))
        torch.cuda.synchronize()
        # (the statistics-on instantiation, like the warm-up steps: a profiler's per-kernel average of the headline
        # instantiation then covers the timed launches and nothing else)
        ctx.set_option("stats", 1)
        torch.cuda.synchronize()
        ts, launches = time.perf_counter(), 0
        while (time.perf_counter() - ts) * 1e3 < ms:
            for _ in range(40):
                ctx.step_dev(sp(sx), sp(sy), sp(sz), sp(sc), None, None, ns, dt, 0.0, 0, 1, 0)
            launches += 40
            torch.cuda.synchronize()
        return {"ms": round((time.perf_counter() - ts) * 1e3, 1), "launches": launches,
                "on": "a scratch copy of the rank's cloud, discarded; the cloud's own first steps are the warm-up steps"}

    def extras(self, cloud, dt, args, box):
        """Outside the timed region, single GPU only, never `value`: (brownian, fused, steady, anchor)."""
        from cudaparticlesfoam_amd import _lib as L
        torch, ctx = self.torch, self.ctx
        # right after the timed region (the cloud is still where the timed steps left it): the tutorial's diffusion
        # coefficient (pitzDaily/system/cudaParticlesDict: diffusionCoeff 1.5e-5) -- another instantiation of the same kernel
        brown = None
        if args.brownian_extra > 0:
            Db = 1.5e-5
            cloud.step(dt, 3, D=Db)
            torch.cuda.synchronize()
            ctx.timing_enable(True); ctx.timing_read()
            tb = time.perf_counter()
            cloud.step(dt, args.brownian_extra, D=Db)
            torch.cuda.synchronize()
            tb = time.perf_counter() - tb
            lb, msb = ctx.timing_read()
            ctx.timing_enable(False)
            kb = msb / max(lb, 1)
            bytes_b = ALGO_BYTES_PER_PARTICLE_STEP + 8          # + the 8-byte particle id the Philox counter needs
            brown = {"D": Db, "ms_per_step": round(tb / args.brownian_extra * 1e3, 4), "kernel_avg_ms": round(kb, 4),
                     "kernel": ctx.step_kernel_name(Db, 0), "algorithmic_bytes_per_particle_step": bytes_b,
                     "achieved_GBs": round(bytes_b * cloud.n / (kb * 1e-3) / 1e9, 1) if kb > 0 else None,
                     "frac": round(bytes_b * cloud.n / (kb * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if kb > 0 else None}
            if not args.no_sort:
                cloud.sort()        # diffusion scrambles the order within a few dozen steps: the extras below start sorted again

        # the same cloud stepped with 8 cycles fused into one launch (CPF_STEP_FUSE_CYCLES: what the replacement
        # advect.H does between two output points; results identical, tests/test_gpu_parity.py)
        fused = None
        if args.fused_extra > 0:
            K = 8
            a = cloud.arrays()                   # device addresses of the shard's arrays (cpf_shard_arrays)
            s0 = cloud.step_index
            ctx.step_dev(a["x"], a["y"], a["z"], a["cell"], a["gid"], None, a["n"], dt, 0.0, s0, K, L.STEP_FUSE_CYCLES)
            torch.cuda.synchronize()
            tf = time.perf_counter()
            for r in range(args.fused_extra):
                ctx.step_dev(a["x"], a["y"], a["z"], a["cell"], a["gid"], None, a["n"], dt, 0.0, s0 + K * (r + 1), K,
                             L.STEP_FUSE_CYCLES)
            torch.cuda.synchronize()
            tf = time.perf_counter() - tf
            fused = {"cycles_per_launch": K, "launches": args.fused_extra,
                     "Mparticle_steps_per_s": round(a["n"] * K * args.fused_extra / tf / 1e6, 1),
                     "ms_per_cycle": round(tf / (K * args.fused_extra) * 1e3, 4)}

        # steady state -- one full sort interval, so the periodic re-sort the short window may miss is in
        steady = None
        k = args.steady_steps if args.steady_steps >= 0 else (0 if args.no_sort else args.sort_interval)
        if k > 0:
            torch.cuda.synchronize()
            ts = time.perf_counter()
            cloud.step(dt, k)
            torch.cuda.synchronize()
            # (not the same workload as the timed region any more: with every boundary reflecting and a uniform
            # (10,0,0) field the cloud drifts towards the outlet wall, 1 mm per step, and piles up there)
            steady = {"steps": k, "ms_per_step": round((time.perf_counter() - ts) / k * 1e3, 4),
                      "first_step": cloud.step_index - k,
                      "sorts_inside": (cloud.step_index // max(1, cloud.sort_interval)) -
                                      ((cloud.step_index - k) // max(1, cloud.sort_interval)) if cloud.sort_interval else 0}

        # the N = 1 point of the strong-scaling curve: the cloud `--gpus N > 1` shards (1e8 in total), on this one GPU,
        # same mesh, field, seeding rule, sort and kernel; its own scratch arrays, freed afterwards
        anchor = None
        na = int(args.anchor_particles)
        if na > 0 and args.anchor_steps > 0:
            p = lambda a: a.data_ptr()   # noqa: E731
            ax, ay, az, ac = self.seed_in_fluid(na, box, 4711)
            ag = torch.arange(na, dtype=torch.int64, device=self.device)
            if not args.no_sort:
                o = [torch.empty_like(t_) for t_ in (ax, ay, az, ac, ag)]
                ctx.sort_by_cell_dev_to(p(ax), p(ay), p(az), p(ac), p(ag), *[p(t_) for t_ in o], na)
                torch.cuda.synchronize()
                ax, ay, az, ac, ag = o
            for s_ in range(3):
                ctx.step_dev(p(ax), p(ay), p(az), p(ac), p(ag), None, na, dt, 0.0, s_, 1, 0)
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for s_ in range(args.anchor_steps):
                ctx.step_dev(p(ax), p(ay), p(az), p(ac), p(ag), None, na, dt, 0.0, 3 + s_, 1, 0)
            torch.cuda.synchronize()
            ta = time.perf_counter() - ta
            alive = int((ac >= 0).sum())
            anchor = {"particles": na, "particles_after": alive, "steps": args.anchor_steps,
                      "ms_per_step": round(ta / args.anchor_steps * 1e3, 4),
                      "Mparticle_steps_per_s": round(na * args.anchor_steps / ta / 1e6, 1),
                      "note": "one GPU, no sharding, no hand-off: divide the --gpus N value by this for the strong-scaling ratio"}
            del ax, ay, az, ac, ag
        # (an extra that fails must not take the headline line with it: its key then holds the error)
        more = {}
        for key, fn in (("brownian_steady", lambda: self._brownian_steady(cloud, dt, args, box)),
                        ("analytic_field", lambda: self._analytic_field(cloud, dt, args, box)),
                        ("tjunction_as_run", lambda: self._tjunction_as_run(dt, args)),
                        ("vertex_velocity", lambda: self._vertex_velocity(cloud, dt, args, box)),
                        ("polyhedral_as_run", lambda: self._polyhedral_as_run(dt, args))):
            try:
                more[key] = fn()
            except Exception as e:                                   # noqa: BLE001
                more[key] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
                if key == "analytic_field":                          # the field must not stay switched
                    try:
                        ctx.set_velocity(self._case[1])
                    except Exception:                                # noqa: BLE001
                        pass
        return brown, fused, steady, anchor, more

    # ---- what the tutorials actually run (never `value`)
    def _fresh_cloud(self, n, box, seed, ctx=None, sort_interval=0):
        """A new single-rank cloud of n particles seeded over the fluid domain, located and sorted, on `ctx`."""
        from cudaparticlesfoam_amd.parallel import ShardedCloud
        torch = self.torch
        ctx = ctx or self.ctx
        x, y, z, c = seed_in_fluid(ctx, torch, n, box, seed, self.device)
        cl = ShardedCloud(ctx, None, n + 4096, None, send_fraction=0.0, exchange_interval=0)
        cl.set_particles(x, y, z, c, None)
        del x, y, z, c
        cl.sort_interval = sort_interval
        cl.sort()
        torch.cuda.synchronize()
        return cl

    @staticmethod
    def _timed_steps(torch, ctx, cl, dt, steps, D, bytes_per):
        ctx.set_option("timing_stride", 10)
        ctx.timing_enable(True); ctx.timing_read()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cl.step(dt, steps, D=D)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        launches, ms = ctx.timing_read()
        ctx.timing_enable(False)
        k = ms / max(launches, 1)
        per = el / steps * 1e3
        return {"steps": steps, "ms_per_step": round(per, 4), "kernel_avg_ms": round(k, 4),
                "Mparticle_steps_per_s": round(cl.n / per / 1e3, 1),
                "algorithmic_bytes_per_particle_step": bytes_per,
                "frac": round(bytes_per * cl.n / (per * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "kernel_frac": round(bytes_per * cl.n / (k * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k > 0 else None}

    @staticmethod
    def _fragment_calls(torch, cl, dt, steps, D, bytes_per, chunk=10):
        """The same cycles the way the replacement advect.H issues them between two frames of either tutorial (saveInterval 10):
        cpf_shard_step(chunk cycles, CPF_STEP_FUSE_CYCLES) -- one launch up to the next sort -- sorts inside the clock."""
        from cudaparticlesfoam_amd import _lib as L
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s_ in range(0, steps, chunk):
            cl.step(dt, min(chunk, steps - s_), D=D, flags=L.STEP_FUSE_CYCLES)
        cl.arrays()
        torch.cuda.synchronize()
        per = (time.perf_counter() - t0) / steps * 1e3
        return {"cycles_per_call": chunk, "ms_per_cycle": round(per, 4), "Mparticle_steps_per_s": round(cl.n / per / 1e3, 1),
                "frac": round(bytes_per * cl.n / (per * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "note": "cpf_shard_step with CPF_STEP_FUSE_CYCLES, sorts included; the fraction is taken on the per-cycle bytes of "
                        "UNFUSED launches: a fused launch loads and stores once per run of cycles"}

    def _brownian_steady(self, cloud, dt, args, box):
        """Sustained rate with the tutorial's diffusion: a fresh cloud, the fragments' sort interval for diffusing clouds
        (compat/src/initCuda.H: 25), the sorts inside the clock (pitzDaily/system/cudaParticlesDict:17-29)."""
        if args.brownian_steady_steps <= 0:
            return None
        torch, ctx = self.torch, self.ctx
        Db, interval = 1.5e-5, 25
        cl = self._fresh_cloud(cloud.n, box, 2025, sort_interval=interval)
        cl.step(dt, 10, D=Db)
        cl.sort(); cl.step_index = 0         # steady state: every interval of the clock starts on a fresh sort and pays for the next one
        r = self._timed_steps(torch, ctx, cl, dt, args.brownian_steady_steps, Db, ALGO_BYTES_PER_PARTICLE_STEP + 8)
        r.update({"D": Db, "sort_interval": interval, "sorts_inside": args.brownian_steady_steps // interval,
                  "kernel": ctx.step_kernel_name(Db, 0), "particles": cl.n,
                  "note": "frac = 64 B x particles / ms_per_step (sorts included); kernel_frac = the step kernel alone"})
        # one re-sort of that cloud by itself, three times: 25 cycles of diffusion without a sort, then the sort between two syncs
        cl.sort_interval = 0
        sorts = []
        for _ in range(3):
            cl.step(dt, interval, D=Db)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cl.sort()
            torch.cuda.synchronize()
            sorts.append((time.perf_counter() - t0) * 1e3)
        r["resort"] = {"ms": round(sorted(sorts)[1], 4), "after_cycles": interval, "particles": cl.n,
                       "key_sort": "this library's radix sort (option sort_method 2, the default)",
                       "note": "keys + 32-byte records, (key, index) radix sort, one gather; median of 3, host-timed between two syncs"}
        # the launch the replacement advect.H issues between two frames of either tutorial (saveInterval 10: the cycles between
        # two output points fused into one launch), on a second fresh cloud in the same state as the one above started from
        from cudaparticlesfoam_amd import _lib as L
        del cl
        if args.no_fused_tutorial:
            return r
        cl = self._fresh_cloud(cloud.n, box, 2025, sort_interval=0)
        cl.step(dt, 10, D=Db)
        cl.sort()
        a = cl.arrays()
        K, reps = 8, 6
        ctx.step_dev(a["x"], a["y"], a["z"], a["cell"], a["gid"], None, a["n"], dt, Db, 10, K, L.STEP_FUSE_CYCLES)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for q in range(reps):
            ctx.step_dev(a["x"], a["y"], a["z"], a["cell"], a["gid"], None, a["n"], dt, Db, 10 + K * (q + 1), K, L.STEP_FUSE_CYCLES)
        torch.cuda.synchronize()
        tb = (time.perf_counter() - tb) / (K * reps) * 1e3
        r["fused_8_cycles_per_launch"] = {"ms_per_cycle": round(tb, 4), "Mparticle_steps_per_s": round(a["n"] / tb / 1e3, 1),
                                          "frac": round((ALGO_BYTES_PER_PARTICLE_STEP + 8) * a["n"] / (tb * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                          "note": "what the fragments launch between two frames; the fraction is taken on the per-cycle "
                                                  "bytes of UNFUSED launches (64 B): a fused launch loads and stores once per 8 cycles"}
        del cl
        cl = self._fresh_cloud(cloud.n, box, 2025, sort_interval=interval)
        cl.step(dt, 10, D=Db)
        cl.sort(); cl.step_index = 0
        r["fragment_calls"] = self._fragment_calls(torch, cl, dt, args.brownian_steady_steps, Db, ALGO_BYTES_PER_PARTICLE_STEP + 8)
        del cl
        return r

    def _analytic_field(self, cloud, dt, args, box):
        """the frozen (analytic step-flow) field instead of the uniform one."""
        if args.analytic_extra <= 0:
            return None
        from cudaparticlesfoam_amd.cases import pitzdaily as pz
        torch, ctx = self.torch, self.ctx
        mesh, U0 = self._case
        centres, _ = mesh.cell_centres_volumes()
        ctx.set_velocity(pz.analytic_step_u(mesh, centres))
        cl = self._fresh_cloud(cloud.n, box, 2026)
        ctx.set_option("stats", 1); c0 = ctx.counters()
        cl.step(dt, 5)
        torch.cuda.synchronize(); c1 = ctx.counters(); ctx.set_option("stats", 0)
        r = self._timed_steps(torch, ctx, cl, dt, args.analytic_extra, 0.0, ALGO_BYTES_PER_PARTICLE_STEP)
        r.update({"field": "analytic step-flow (cases/pitzdaily.py: analytic_step_u)", "kernel": ctx.step_kernel_name(0.0, 0),
                  "particles": cl.n, "first_step": 5,
                  "cells_visited_per_particle_step": round((c1["cells_visited"] - c0["cells_visited"]) /
                                                           max(1, c1["particle_steps"] - c0["particle_steps"]), 3)})
        del cl
        ctx.set_velocity(U0)
        return r

    def _tjunction_as_run(self, dt, args):
        """The reference's second tutorial as its dictionary runs it (TJunction/system/cudaParticlesDict:17-28: 4e6 particles
        from the seeding box, diffusionCoeff 1.5e-05, dt 1e-4) on the 248 000-cell mesh of system/blockMeshDict:68-81 --
        Brownian AND 3-D.  pimpleFoam's field is replaced by the closed-form split flow of cases/tjunction.py."""
        if args.tjunction_steps <= 0:
            return None
        from cudaparticlesfoam_amd.api import Context
        from cudaparticlesfoam_amd.cases import tjunction as tj
        torch = self.torch
        mesh = tj.tjunction_mesh()
        centres, _ = mesh.cell_centres_volumes()
        d = tj.PARTICLE_DICT
        ctx2 = Context(self.device.index or 0)
        ctx2.set_stream(torch.cuda.current_stream().cuda_stream)
        try:
            ctx2.set_mesh(mesh)
            ctx2.set_velocity(tj.split_flow_u(mesh, centres, d["startTime"]))
            n, interval, Db = int(args.tjunction_particles), 25, d["diffusionCoeff"]
            cl = self._fresh_cloud(n, d["seedingBox"], 2027, ctx=ctx2, sort_interval=interval)
            ctx2.set_option("stats", 1); c0 = ctx2.counters()
            cl.step(dt, 10, D=Db)
            torch.cuda.synchronize(); c1 = ctx2.counters(); ctx2.set_option("stats", 0)
            cl.sort(); cl.step_index = 0     # (as in _brownian_steady)
            r = self._timed_steps(torch, ctx2, cl, dt, args.tjunction_steps, Db, ALGO_BYTES_PER_PARTICLE_STEP + 8)
            rec_once = (128 if ctx2.step_kernel_name(Db, 0).endswith(", 6>") else 256) * mesh.n_cells     # (box records: 128 B)
            r.update({"D": Db, "particles": n, "cells": mesh.n_cells, "sort_interval": interval,
                      "sorts_inside": args.tjunction_steps // interval, "kernel": ctx2.step_kernel_name(Db, 0),
                      "mesh_flags": ctx2.mesh_flags(), "records_bytes_once": rec_once,
                      "cells_visited_per_particle_step": round((c1["cells_visited"] - c0["cells_visited"]) /
                                                               max(1, c1["particle_steps"] - c0["particle_steps"]), 3),
                      "seeding_box": [list(d["seedingBox"][0]), list(d["seedingBox"][1])],
                      "field": "closed-form split flow, u0 = 3 m/s at t = 0.5 s (stand-in for pimpleFoam's U)"})
            if not args.no_fused_tutorial:
                cl.sort(); cl.step_index = 0
                r["fragment_calls"] = self._fragment_calls(torch, cl, dt, args.tjunction_steps, Db, ALGO_BYTES_PER_PARTICLE_STEP + 8)
                r["fragment_calls"]["note"] += ("; 10 cycles = one Eulerian step of the tutorial (deltaT 1e-3, dt 1e-4) with the dictionary's "
                                                "alternative `saveInterval 1e16` (TJunction/system/cudaParticlesDict:29); with its `saveInterval 2` "
                                                "every launch is ONE cycle and every second one writes a frame of 4e6 particles: the unfused "
                                                "figure above, in a run that is bound by the frame writer")
            del cl
        finally:
            ctx2.close()
        return r

    def _vertex_velocity(self, cloud, dt, args, box):
        """the cycle with the reference's "VertexVelocity" advect (cuda/particles.cu:244-313, dispatch :428-437) --
        the velocity interpolated at the particle's position from values on the vertices of the cell's tets (src/initCuda.H:86-124:
        12 per hex) -- as ONE launch per cycle of step_kernel_vertex (the generic CSR walk; no streaming instantiation).  No solver
        sets the mode (src/initCuda.H:72); pitzDaily, the analytic step-flow field sampled at the tet-mesh vertices."""
        if args.vertex_steps <= 0:
            return None
        from cudaparticlesfoam_amd import _lib as L
        from cudaparticlesfoam_amd.cases import pitzdaily as pz
        torch, ctx = self.torch, self.ctx
        mesh, _ = self._case
        centres, _ = mesh.cell_centres_volumes()
        pos, tets = mesh.tet_decomposition(centres)            # what the fragment builds: points ++ centres, 12 tets per hex
        ctx.set_tets(pos, tets, 12)
        ctx.set_vertex_velocity(pz.analytic_step_u(mesh, pos))
        n = cloud.n
        x, y, z, c = seed_in_fluid(ctx, torch, n, box, 2028, self.device)
        g = torch.arange(n, dtype=torch.int64, device=self.device)
        p = lambda a: a.data_ptr()   # noqa: E731
        ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)
        fl = L.STEP_VERTEX_VELOCITY
        ctx.set_option("stats", 1); c0 = ctx.counters()
        ctx.step_dev(p(x), p(y), p(z), p(c), p(g), None, n, dt, 0.0, 0, 3, fl)
        torch.cuda.synchronize(); c1 = ctx.counters(); ctx.set_option("stats", 0)
        ctx.set_option("timing_stride", 1)
        ctx.timing_enable(True); ctx.timing_read()
        t0 = time.perf_counter()
        ctx.step_dev(p(x), p(y), p(z), p(c), p(g), None, n, dt, 0.0, 3, args.vertex_steps, fl)
        torch.cuda.synchronize()
        per = (time.perf_counter() - t0) / args.vertex_steps * 1e3
        launches, ms = ctx.timing_read(); ctx.timing_enable(False)
        k = ms / max(launches, 1)
        alive = int((c >= 0).sum())
        return {"steps": args.vertex_steps, "particles": n, "particles_after": alive, "ms_per_step": round(per, 4),
                "kernel_avg_ms": round(k, 4), "kernel": ctx.step_kernel_name(0.0, fl),
                "Mparticle_steps_per_s": round(n / per / 1e3, 1), "algorithmic_bytes_per_particle_step": ALGO_BYTES_PER_PARTICLE_STEP,
                "frac": round(ALGO_BYTES_PER_PARTICLE_STEP * n / (per * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "kernel_frac": round(ALGO_BYTES_PER_PARTICLE_STEP * n / (k * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k > 0 else None,
                "tets": int(tets.shape[0]), "tet_vertices": int(pos.shape[0]),
                "cells_visited_per_particle_step": round((c1["cells_visited"] - c0["cells_visited"]) /
                                                         max(1, c1["particle_steps"] - c0["particle_steps"]), 3),
                "field": "analytic step-flow sampled at the tet-mesh vertices (points ++ cell centres)",
                "note": "56 B x particles / time, like the cell-constant cycle (config.analytic_field is that cycle on the same "
                        "field at the cell centres); the tet and vertex tables (%.1f MB) stay in L2"
                        % ((tets.nbytes + pos.nbytes * 2) / 1e6)}

    def _polyhedral_as_run(self, dt, args):
        """The non-hex ("polyMesh") share of BASELINE configs[4] on one GPU: a graded 40 x 40 x 40 box whose central block is refined
        2 x 2 x 2 -- 114 540 cells, 2 242 of them with 9 faces, split faces as face groups: a mesh the reference cannot run
        (src/initCuda.H:64) -- with the tutorials' diffusion and a transient solver's field: a new U uploaded every 10 cycles
        (src/advect.H:44-57), the upload timed by itself; the fragments' sort cadence for diffusing clouds inside the clock."""
        if args.polyhedral_steps <= 0:
            return None
        from cudaparticlesfoam_amd.api import Context
        from cudaparticlesfoam_amd.cases import refined_box
        torch = self.torch
        lo3, hi3 = (0.0, 0.0, 0.0), (0.3, 0.05, 0.05)
        mesh, _ = refined_box(40, 40, 40, lo3, hi3, ((0.075, 0.0125, 0.0125), (0.225, 0.0375, 0.0375)), grading=(2.0, 1.0, 0.5))
        cc, _ = mesh.cell_centres_volumes()
        field = lambda a, b: np.ascontiguousarray(np.stack([10.0 + 0 * cc[:, 0], a * np.sin(40 * cc[:, 2]), b * np.cos(40 * cc[:, 1])], 1))   # noqa: E731
        ctx2 = Context(self.device.index or 0)
        ctx2.set_stream(torch.cuda.current_stream().cuda_stream)
        try:
            ctx2.set_mesh(mesh)
            ctx2.set_velocity(field(4.0, 4.0))
            n, interval, Db, per_u = int(args.polyhedral_particles), 25, 1.5e-5, 10
            cl = self._fresh_cloud(n, (lo3, hi3), 2029, ctx=ctx2, sort_interval=interval)
            ctx2.set_option("stats", 1); c0 = ctx2.counters()
            cl.step(dt, 10, D=Db)
            torch.cuda.synchronize(); c1 = ctx2.counters(); ctx2.set_option("stats", 0)
            cl.sort(); cl.step_index = 0
            fields = [field(4.0 - 0.1 * e, 3.0 + 0.1 * e) for e in range(4)]
            ctx2.set_option("timing_stride", 5)
            ctx2.timing_enable(True); ctx2.timing_read()
            torch.cuda.synchronize()
            upload_s, done, e = 0.0, 0, 0
            t0 = time.perf_counter()
            while done < args.polyhedral_steps:
                torch.cuda.synchronize()                               # (the cycles queued so far are not the upload's time)
                tu = time.perf_counter()
                cl.set_velocity(fields[e % len(fields)])           # (cpf_shard_set_velocity: 24 B per cell, host -> device)
                torch.cuda.synchronize()
                upload_s += time.perf_counter() - tu
                k = min(per_u, args.polyhedral_steps - done)
                cl.step(dt, k, D=Db)
                done += k; e += 1
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            launches, ms = ctx2.timing_read(); ctx2.timing_enable(False)
            kk = ms / max(launches, 1)
            per = (el - upload_s) / args.polyhedral_steps * 1e3
            bytes_per = ALGO_BYTES_PER_PARTICLE_STEP + 8
            nf = np.diff(mesh.cell_faces()[0])
            return {"steps": args.polyhedral_steps, "particles": n, "particles_after": cl.global_count(), "cells": mesh.n_cells,
                    "cells_with_9_faces": int((nf == 9).sum()), "D": Db, "sort_interval": interval,
                    "sorts_inside": args.polyhedral_steps // interval, "kernel": ctx2.step_kernel_name(Db, 0),
                    "mesh_flags": ctx2.mesh_flags(), "ms_per_step": round(per, 4), "kernel_avg_ms": round(kk, 4),
                    "Mparticle_steps_per_s": round(n / per / 1e3, 1), "algorithmic_bytes_per_particle_step": bytes_per,
                    "frac": round(bytes_per * n / (per * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "kernel_frac": round(bytes_per * n / (kk * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if kk > 0 else None,
                    "velocity_uploads": e, "cycles_per_upload": per_u, "upload_ms_each": round(upload_s / max(e, 1) * 1e3, 4),
                    "upload_bytes_each": int(24 * mesh.n_cells),
                    "ms_per_step_with_uploads": round(el / args.polyhedral_steps * 1e3, 4),
                    "cells_visited_per_particle_step": round((c1["cells_visited"] - c0["cells_visited"]) /
                                                             max(1, c1["particle_steps"] - c0["particle_steps"]), 3),
                    "note": "ms_per_step: the cycles with their sorts, uploads taken out; upload_ms_each: cpf_shard_set_velocity "
                            "between two syncs (24 B per cell + the record rebuild on the device)"}
        finally:
            ctx2.close()

    def finish(self):
        if self.control:
            self.dist.barrier()
            self.dist.destroy_process_group()
        if self.comm is not None:
            self.comm.close()
        self.ctx.close()


def run(args, M):
    """The measurement.  Returns the JSON record on rank 0, None elsewhere."""
    from cudaparticlesfoam_amd.cases import pitzdaily as pz
    from cudaparticlesfoam_amd.parallel import slab_bounding_box, slab_cell_ranges, x_slab_renumbering
    torch, dist, rank, world, device = M.torch, M.dist, M.rank, M.world, M.device
    dist_on = world > 1 or args.force_dist

    # ---- synthetic case: pitzDaily mesh renumbered into x-slabs (same mesh for every N)
    mesh0 = pz.pitzdaily_mesh()
    c0, _ = mesh0.cell_centres_volumes()
    mesh = mesh0.renumber_cells(x_slab_renumbering(c0))
    centres, vols = mesh.cell_centres_volumes()
    U = pz.uniform_u(mesh) if args.field == "uniform" else pz.analytic_step_u(mesh, centres)
    cell_lo = slab_cell_ranges(vols, world)
    M.set_case(mesh, U)
    ctx = M.ctx

    n_total = int(args.particles) * (world if args.scaling == "weak" else 1)
    n_local = n_total // world
    # every rank seeds its own x-slab: the sampling box is clipped to the slab's points, then points are kept
    # only if their cell belongs to the rank
    box = [list(pz.DOMAIN_BOX[0]), list(pz.DOMAIN_BOX[1])]
    full_box = [list(box[0]), list(box[1])]
    if world > 1:
        lo_pt, hi_pt = slab_bounding_box(mesh, int(cell_lo[rank]), int(cell_lo[rank + 1]))
        box[0][0] = max(box[0][0], float(lo_pt[0]) - 1e-9)
        box[1][0] = min(box[1][0], float(hi_pt[0]) + 1e-9)
    x, y, z, c = M.seed_in_fluid(n_local, box, 1000 + rank,
                                 (int(cell_lo[rank]), int(cell_lo[rank + 1])) if world > 1 else None)
    # 288 GB of HBM: slack is free.  3x capacity and a send buffer as large as the shard make an overflow
    # impossible even if a whole neighbouring slab drains into this rank between two rebalances.
    cap = (int(n_local * 3.0) if world > 1 else n_local) + 4096
    # the shard, its hand-off buffers and every collective live behind the C-ABI (cpf_shard_*, csrc/cpf_shard_core.h)
    cloud = M.make_cloud(cell_lo, cap, send_fraction=1.0 if dist_on else 0.01, exchange_interval=args.exchange_interval)
    cloud.force_collectives = args.force_dist
    cloud.rebalance_interval = args.rebalance_interval
    cloud.overlap_steps = args.overlap_steps
    if dist_on and args.balance == "time":
        cloud.enable_time_balancing()
    cloud.sort_interval = 0 if args.no_sort else args.sort_interval
    cloud.set_particles(x, y, z, c, None, first_gid=rank * n_local)
    del x, y, z, c
    dry = None
    if dist_on:
        stage("first_exchange")
        M.sync(); t_fx = time.perf_counter()
        cloud.rebalance(mesh.n_cells)         # also pays RCCL's one-time all-reduce / all-to-all set-up before timing
        M.sync(); t_rb = time.perf_counter()
        cloud.exchange()
        M.sync(); t_ex = time.perf_counter()
        dry = {"comm_init_s": round(getattr(M, "comm_init_s", 0.0), 3), "first_recut_and_handoff_s": round(t_rb - t_fx, 3),
               "second_handoff_s": round(t_ex - t_rb, 4), "handed_off": cloud.handed_off, "particles_on_rank0": cloud.n}
        if args.dry_collectives:
            # only init -> first re-cut -> one all-to-all-v, with per-stage timings: a failing scaling run costs seconds and
            # names the collective (the stage trail on stderr says how far it got)
            total = cloud.global_count()
            cloud.close()
            M.finish()
            stage("done")
            return {"dry_collectives": dry, "n_gpus": world, "particles_total": total, "rccl_ranks": world} if rank == 0 else None
    if not args.no_sort:
        cloud.sort()
    M.sync()

    def barrier():
        if world > 1:
            dist.barrier()                          # (control plane: gloo)

    dt = 1e-4
    spinup = M.spinup(cloud, dt, args.spinup_ms)
    ctx.set_option("stats", 1)
    counters0 = ctx.counters()
    stage("warmup")
    if dist_on and args.balance == "time":
        ctx.set_option("timing_stride", 1)
        ctx.timing_enable(True)                    # the balancer's first cost reading comes from the warm-up steps
    cloud.step(dt, args.warmup)                    # statistics counters on: feeds the config fields below
    if dist_on:
        # the balancer reads the measured step times for the first time here -- its one wait for the launch queue per run
        # (cpf_shard_core.h, measuredCost) -- and re-cuts by them: warm-up, like RCCL's first collectives above
        cloud.rebalance(mesh.n_cells)
    M.sync(); barrier()
    counters = {k: v - counters0[k] for k, v in ctx.counters().items()}
    ctx.set_option("stats", 0)                     # diagnostics off in the timed region (the reference has none)
    n_before = cloud.global_count()
    # live kernel timing for the roofline: HIP events around every k-th launch of the timed region (a pair around
    # EVERY launch costs 3.5 % of the throughput it is there to measure, around every 4th 0.6-4 %: see --timing-stride);
    # at least five samples, and every 8th launch at most where the balancer cuts by these times (N > 1)
    args.timing_stride = max(1, min(args.timing_stride, args.steps // 5, 8 if (args.gpus > 1 or args.force_dist) else 1 << 30))
    ctx.set_option("timing_stride", args.timing_stride)
    ctx.timing_enable(True)
    ctx.timing_read()                              # drop the warm-up launches' events
    handed0, ms0, launches0, psteps0 = cloud.handed_off, cloud.kernel_ms, cloud.kernel_launches, cloud.particle_steps
    hhost0, ex0 = cloud.handoff_host_ms, cloud.exchanges
    hwait0 = cloud.handoff_wait_ms
    cloud.profile_comm = True                      # keep the (start, end) events of the hand-offs' collectives
    comm0 = cloud.comm_ms()
    if dist_on:
        # the hand-off cadence counts from the first timed step (a hand-off the warm-up left in flight completes here, outside
        # the clock): re-cuts fall on timed steps r, 2r, ... whatever the warm-up was.  (D = 0: no random stream sees the index.)
        cloud.step_index = 0
    M.sync(); barrier()
    stage("timed_region")
    t0 = time.perf_counter()
    cloud.step(dt, args.steps)
    cloud.flush()                                  # a hand-off still in flight belongs to the timed region
    M.sync()
    el = time.perf_counter() - t0                  # this rank's clock stops at its OWN sync; MAX over ranks below
    barrier()
    launches, kernel_ms = ctx.timing_read()        # + what the load balancer drained during the timed region
    launches += cloud.kernel_launches - launches0; kernel_ms += cloud.kernel_ms - ms0
    psteps = cloud.particle_steps - psteps0
    ctx.timing_enable(False)
    handoff_host_ms = cloud.handoff_host_ms - hhost0
    handoff_comm_ms = cloud.comm_ms() - comm0
    handoffs = cloud.exchanges - ex0
    cloud.profile_comm = False
    rccl_ranks = world if dist_on else 1
    per_rank = [cloud.n]
    if world > 1:                                  # control plane (gloo, host tensors)
        tn = torch.tensor([cloud.n], dtype=torch.int64)
        rows = [torch.empty_like(tn) for _ in range(world)]
        dist.all_gather(rows, tn)
        per_rank = [int(r.item()) for r in rows]
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    n_after = cloud.global_count()

    brown = fused = steady = anchor = None
    more = {}
    if world == 1 and not args.force_dist:
        stage("extras")
        brown, fused, steady, anchor, more = M.extras(cloud, dt, args, full_box)

    out = None
    if rank == 0 and world > 1 and handoffs == 0:
        # work skipped inside the timed region is no measurement: N > 1 without a single re-cut / all-to-all-v inside the clock
        # would be N independent replicas
        out = {"error": "no hand-off inside the timed region (steps %d, rebalance interval %d, exchange interval %d): "
                        "not a measurement of the sharded path" % (args.steps, args.rebalance_interval, args.exchange_interval),
               "stage": "timed_region", "n_gpus": world}
    elif rank == 0:
        value = n_before * args.steps / el / 1e6
        avg_kernel_s = kernel_ms / max(launches, 1) / 1e3
        per_launch = psteps / max(args.steps, 1)         # rank 0's particles per launch (varies when N > 1)
        achieved = ALGO_BYTES_PER_PARTICLE_STEP * per_launch / avg_kernel_s / 1e9 if (launches and avg_kernel_s > 0) else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))              # PMC passes are separate rocprofv3 runs (tools/pmc.sh)
                if rec.get("particles_per_launch") == n_local and world == 1:
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mparticle-steps/s", "value": round(value, 2), "unit": "Mparticle-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "pitzDaily 12225-cell polyMesh (blockMeshDict restated), %s U, dt 1e-4, D 0, "
                                   "%s, all boundaries reflecting"
                                   % ("uniform (10,0,0)" if args.field == "uniform" else "analytic step-flow",
                                      ("%d fp64 particles on one GPU seeded over the fluid domain (BASELINE configs[2])" % n_total)
                                      if world == 1 else
                                      ("%d fp64 particles in total sharded over %d GPUs by x-slab, mesh replicated, %s "
                                       "hand-off (BASELINE configs[3], %s scaling; --gpus 1 runs the 1e7 single-GPU "
                                       "config and reports this cloud on one GPU as config.strong_anchor_1e8)"
                                       % (n_total, world, M.collectives, args.scaling))),
                       "particles_total": n_before, "particles_after": n_after, "cells": mesh.n_cells,
                       # which shortcuts of the walk the mesh layer found live on this mesh (include/cpf.h, cpf_get_mesh_flags)
                       "mesh_flags": ctx.mesh_flags() if hasattr(ctx, "mesh_flags") else None,
                       "exchange_interval": args.exchange_interval if dist_on else None,
                       "rebalance_interval": args.rebalance_interval if dist_on else None,
                       "rebalance_interval_counted_from": "the first timed step" if dist_on else None,
                       "overlap_steps": (cloud._overlap() if args.overlap_steps < 0 else args.overlap_steps) if dist_on else None,
                       "overlap_steps_auto": (args.overlap_steps < 0) if dist_on else None,
                       "balance": (("measured step time" if args.balance == "time" else "particle count")
                                   if dist_on else None),
                       "particles_per_rank_at_end": per_rank if world > 1 else None,
                       "handoff_fraction_per_step": (round((cloud.handed_off - handed0) / max(1, cloud.n) / args.steps, 6)
                                                     if dist_on else None),
                       "rccl_ranks": rccl_ranks,
                       "collectives": (M.collectives + ", issued by the library (cpf_shard_*, csrc/cpf_shard_core.h)") if dist_on else None,
                       "first_collectives": dry,
                       "ms_in_handoff": ({"host_ms_total": round(handoff_host_ms, 3), "collectives_device_ms_total": round(handoff_comm_ms, 3),
                                          "handoffs": handoffs,
                                          "host_wait_ms_total": round(cloud.handoff_wait_ms - hwait0, 3),
                                          "host_work_ms_per_handoff": round((handoff_host_ms - (cloud.handoff_wait_ms - hwait0)) / max(1, handoffs), 4),
                                          "host_ms_per_step": round(handoff_host_ms / max(1, args.steps), 4)}
                                         if dist_on else None),
                       "ms_per_step_steady": steady, "brownian": brown, "device_spinup": spinup,
                       "extra_fused_cycles": fused, "strong_anchor_1e8": anchor,
                       # what the tutorials actually run, after the timed region (never `value`)
                       "brownian_steady": more.get("brownian_steady"), "analytic_field": more.get("analytic_field"),
                       "tjunction_as_run": more.get("tjunction_as_run"),
                       # what is built but had no number in the driver-run line until round 6
                       "vertex_velocity": more.get("vertex_velocity"), "polyhedral_as_run": more.get("polyhedral_as_run"),
                       "cells_visited_per_particle_step": round(counters["cells_visited"] / max(1, counters["particle_steps"]), 3),
                       "reflections_per_particle_step": round(counters["reflections"] / max(1, counters["particle_steps"]), 4),
                       "sorted_by_cell": not args.no_sort, "sort_interval": 0 if args.no_sort else args.sort_interval, "visit_stats_from": "the %d warm-up steps" % args.warmup},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": ("profiles/pmc_latest.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                            "this command, tools/pmc.sh; not collected in this run)" if traffic is not None
                                            else None),
                         "kernel": ctx.step_kernel_name(0.0, 0), "kernel_avg_ms": round(avg_kernel_s * 1e3, 4),
                         "launches": launches, "launches_sampled_every": args.timing_stride, "algorithmic_bytes_per_launch": int(ALGO_BYTES_PER_PARTICLE_STEP * per_launch)},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(mesh, centres, U, args.cpu_seconds)
            except Exception as e:          # the checker being absent must not hide the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "Mparticle-steps/s", "cores": 0, "kind": "port",
                                       "sample": "unavailable: %r" % (e,)}
    cloud.close()                       # (the shard borrows the context and the communicator M.finish() destroys)
    M.finish()
    stage("done")
    return out


def main():
    args = parse()
    launched = "WORLD_SIZE" in os.environ
    if not launched and (args.gpus > 1 or args.self_launch):
        sys.exit(self_launch(args))                 # before any import that could touch the GPU
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d was started with WORLD_SIZE=%d: one rank per GPU" % (args.gpus, world))
    dog = rank_watchdog(args, rank, world)
    try:
        out = run(args, GpuMachine(args, rank, world, local))
    except Exception as e:                          # a failed collective, a HIP error ...: say where, leave non-zero
        import traceback
        traceback.print_exc()
        if rank == 0 and world > 1:
            print(json.dumps({"error": "%s: %s" % (type(e).__name__, str(e)[:400]), "stage": _STAGE, "n_gpus": world}), flush=True)
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(1)                                 # (not sys.exit: a process group left half-initialised may hang in its destructor)
    if dog is not None:
        dog.cancel()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: anything a library left in C stdio's buffer goes out before it
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
        if "error" in out:
            sys.exit(5)


if __name__ == "__main__":
    main()
