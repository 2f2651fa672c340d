"""bench.py's one-line JSON contract: the committed round profile (CPU) and a small live run (GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}
ROOFLINE = {"bound", "achieved", "peak", "unit", "frac", "traffic"}


def _check(line, need_cpu_baseline, scaling="weak"):
    d = json.loads(line)
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert d["metric"] == "Mparticle-steps/s" and d["unit"] == "Mparticle-steps/s" and d["higher_is_better"] is True
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None and d["scaling"] == scaling
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert ROOFLINE <= set(r) and r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 <= r["frac"] < 1
    # value = particles * steps / time, in units of 1e6
    # (both figures are rounded for printing: two decimals of `value`, four of `ms_per_step`)
    assert abs(d["value"] - d["config"]["particles_total"] * 1e-6 / (d["ms_per_step"] * 1e-3)) < 2e-3 * d["value"] + 0.006
    if need_cpu_baseline:
        c = d["cpu_baseline"]
        assert {"value", "unit", "cores", "kind", "sample"} <= set(c) and c["kind"] in ("reference", "port")
        assert c["unit"] == d["unit"] and c["cores"] >= 1 and c["value"] > 0
    return d


def test_defaults_follow_the_north_star(monkeypatch):
    """--gpus 1 is the 1e7 single-GPU config (BASELINE configs[2]); --gpus N > 1 is configs[3]: 1e8 particles in
    total over the N ranks, i.e. strong scaling -- unless the caller says otherwise."""
    sys.path.insert(0, ROOT)
    import bench
    for argv, want in ((["bench.py"], ("weak", 1e7)), (["bench.py", "--gpus", "1"], ("weak", 1e7)),
                       (["bench.py", "--gpus", "8"], ("strong", 1e8)), (["bench.py", "--gpus", "2"], ("strong", 1e8)),
                       (["bench.py", "--gpus", "4", "--scaling", "weak"], ("weak", 1e7)),
                       (["bench.py", "--gpus", "4", "--particles", "2e7"], ("strong", 2e7))):
        monkeypatch.setattr(sys, "argv", argv)
        a = bench.parse()
        assert (a.scaling, a.particles) == want, (argv, a.scaling, a.particles)


def test_committed_round_profile_follows_the_contract():
    line = open(os.path.join(ROOT, "profiles", "r01_bench_1gpu.json")).read().strip()
    d = _check(line, need_cpu_baseline=True)
    assert d["n_gpus"] == 1 and d["config"]["particles_total"] == 10_000_000 and d["config"]["cells"] == 12225
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 56 * 10_000_000
    assert 0.95 < d["roofline"]["traffic"] / d["roofline"]["algorithmic_bytes_per_launch"] < 1.15   # PMC: no wasted re-reads


def test_world_2_orchestration_over_gloo_prints_one_strong_scaling_line():
    """The N > 1 path of bench.run() -- per-slab seeding, re-cut by (pretended) step time, overlapped hand-offs,
    max-over-ranks clock, rank 0's single JSON line -- with a CPU stand-in for the machine (tests/_bench_worker.py:
    oracle + gloo).  The figures are not performance; the contract and the bookkeeping are what is checked."""
    import socket
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_bench_worker.py"), "--gpus", "2",
           "--particles", "4000", "--steps", "8", "--warmup", "2", "--rebalance-interval", "4", "--overlap-steps", "2",
           "--sort-interval", "5"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=dict(os.environ, OMP_NUM_THREADS="1", PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = _check(lines[0], need_cpu_baseline=False, scaling="strong")
    c = d["config"]
    assert d["n_gpus"] == 2 and c["rccl_ranks"] == 2 and "cpu_baseline" not in d
    assert c["particles_total"] == 4000 and c["particles_after"] == 4000       # every boundary reflects
    assert sum(c["particles_per_rank_at_end"]) == 4000 and min(c["particles_per_rank_at_end"]) > 1000
    assert c["ms_in_handoff"]["handoffs"] == 2 and c["handoff_fraction_per_step"] > 0 and c["balance"] == "measured step time"
    assert c["strong_anchor_1e8"] is None and c["brownian"] is None            # single-GPU extras do not run at N > 1
    assert "configs[3]" in c["workload"] and "strong" in c["workload"]


def _gloo_world_2(extra, timeout=900):
    import socket
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_bench_worker.py"), "--gpus", "2"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT,
                       env=dict(os.environ, OMP_NUM_THREADS="1", PYTHONPATH=ROOT))
    return r, [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_the_drivers_own_arguments_put_handoffs_inside_the_timed_region():
    """The driver runs N > 1 as `bench.py --gpus N --steps 20 --warmup 5` and NOTHING else.  With those arguments the timed region
    must hold re-cuts + all-to-all-v hand-offs (round-5 verdict: the old default, every 32 steps counted from step 0, put none
    into 25 steps).  The default cadence is max(2, min(32, steps // 2)) = 10 counted from the first timed step: hand-offs after
    timed steps 10 and 20 (production cadence: 32, what a longer region gets)."""
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse(["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert a.rebalance_interval == 10 and a.rebalance_interval_auto and a.exchange_interval == 0
    assert bench.parse(["--gpus", "8"]).rebalance_interval == 32                    # bench.py's own default --steps 100: the production cadence
    assert bench.parse(["--gpus", "2", "--steps", "3"]).rebalance_interval == 2     # never rarer than the region is long
    r, lines = _gloo_world_2(["--particles", "4000", "--steps", "20", "--warmup", "5"])       # (the cloud is the only thing scaled down)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert len(lines) == 1
    d = _check(lines[0], need_cpu_baseline=False, scaling="strong")
    c = d["config"]
    assert d["steps"] == 20 and d["warmup"] == 5 and c["rebalance_interval"] == 10 and c["rccl_ranks"] == 2
    assert c["ms_in_handoff"]["handoffs"] >= 2 and c["handoff_fraction_per_step"] > 0
    assert c["particles_total"] == 4000 and c["particles_after"] == 4000


def test_a_timed_region_without_a_handoff_is_an_error_line_not_a_value():
    """N > 1 with no re-cut and no hand-off inside the clock = N independent replicas: rank 0 prints {"error": ...} and no value."""
    r, lines = _gloo_world_2(["--particles", "4000", "--steps", "4", "--warmup", "1", "--rebalance-interval", "0",
                              "--exchange-interval", "0"])
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(lines[0])
    assert "value" not in d and "metric" not in d and "no hand-off inside the timed region" in d["error"] and d["n_gpus"] == 2


def test_dry_collectives_over_gloo_prints_the_stage_timings_and_exits():
    """`--dry-collectives`: communicator init -> first re-cut -> one all-to-all-v, the three timed, ONE JSON line, exit 0 -- what a
    first multi-GPU run is started with so that a failing collective costs seconds and is named (here: world 2 on CPU)."""
    import socket
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_bench_worker.py"), "--gpus", "2",
           "--particles", "4000", "--dry-collectives"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env=dict(os.environ, OMP_NUM_THREADS="1", PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    dry = d["dry_collectives"]
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["particles_total"] == 4000
    assert dry["first_recut_and_handoff_s"] >= 0 and dry["second_handoff_s"] >= 0 and dry["handed_off"] >= 0
    assert 0 < dry["particles_on_rank0"] < 4000


def test_a_rank_that_hangs_becomes_an_error_line_within_the_limit():
    """`python bench.py --gpus 2` starts its ranks as a child process tree under a wall-clock limit.  Here rank 1 never
    reaches its first collective (the CPU stand-in over gloo sleeps instead): the parent must kill the whole tree when
    --launch-timeout expires, print ONE JSON line {"error", "stage", "n_gpus"} and exit non-zero -- a hang in the first
    multi-GPU run turns into a diagnosis instead of burning the caller's time limit."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(BENCH_TEST_HANG_RANK="1", OMP_NUM_THREADS="1", PYTHONPATH=ROOT)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--particles", "2000", "--steps", "2",
                        "--warmup", "1", "--launch-timeout", "25", "--child-script", os.path.join(ROOT, "tests", "_bench_worker.py")],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    took = time.time() - t0
    assert r.returncode == 124, (r.returncode, r.stdout[-1000:], r.stderr[-2000:])
    assert 25 <= took < 90
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["stage"] == "rccl_init" and "launch-timeout" in d["error"]
    # nothing of the tree is left behind
    ps = subprocess.run(["ps", "-eo", "pid,args"], capture_output=True, text=True).stdout
    assert "_bench_worker.py" not in ps, ps


def test_ranks_under_a_foreign_launcher_watch_themselves():
    """The driver starts N > 1 ranks with its own `torch.distributed.run`: no parent of ours watches them.  Every rank arms a timer
    (bench.rank_watchdog); here rank 1 never reaches the rendezvous, rank 0 waits in it -- when the timers fire rank 0 prints the
    error line (stage: rccl_init) and every rank leaves with 124, so the launcher returns non-zero instead of hanging."""
    import socket
    import time
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_bench_worker.py"), "--gpus", "2", "--particles", "2000",
           "--steps", "2", "--warmup", "1", "--launch-timeout", "20"]
    env = dict(os.environ, BENCH_TEST_HANG_RANK="1", OMP_NUM_THREADS="1", PYTHONPATH=ROOT)
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and 20 <= time.time() - t0 < 120
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["stage"] == "rccl_init" and d["n_gpus"] == 2 and "watchdog" in d["error"]


def test_self_launched_cpu_ranks_forward_one_json_line():
    """The same launcher on the happy path (CPU stand-in, gloo world 2): rank 0's line is the parent's only stdout line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(OMP_NUM_THREADS="1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--particles", "4000", "--steps", "8",
                        "--warmup", "2", "--rebalance-interval", "4", "--overlap-steps", "2", "--launch-timeout", "600",
                        "--child-script", os.path.join(ROOT, "tests", "_bench_worker.py")],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = _check(lines[0], need_cpu_baseline=False, scaling="strong")
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2
    assert "[bench stage] timed_region" in r.stderr


def test_plain_gpus_n_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher in the environment must get as far as the ranks themselves (here:
    their refusal to run without a GPU), i.e. fail inside the child ranks and not in the argument handling -- and hand
    the child's exit code on."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--particles", "1e4", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs here: covered by the driver's scaling run")
    assert r.returncode != 0
    assert "must be launched with" not in r.stderr
    # both ranks came up under torch.distributed.run and stopped where the product path needs its GPU
    assert "bench.py needs a GPU" in r.stderr or "invalid device ordinal" in r.stderr or "NCCL" in r.stderr or "HIP" in r.stderr, r.stderr[-3000:]
    # the one stdout line is an error record (no metric), naming the stage the ranks died in
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert "error" in d and "metric" not in d and d["n_gpus"] == 2 and d["stage"] in ("launch", "rccl_init")


@pytest.mark.gpu
def test_dry_collectives_on_a_one_rank_rccl_group():
    """The same on the GPU: a real RCCL communicator of one rank (--force-dist), the library's all-gather / all-reduce / grouped
    send-recv issued once each, timings printed, exit 0."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--particles", "2e5", "--force-dist", "--dry-collectives"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29579")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["rccl_ranks"] == 1 and d["particles_total"] == 200_000
    assert d["dry_collectives"]["comm_init_s"] > 0 and d["dry_collectives"]["particles_on_rank0"] == 200_000


@pytest.mark.gpu
def test_the_drivers_arguments_on_a_one_rank_rccl_group_hold_handoffs():
    """`--steps 20 --warmup 5` and no cadence flag, as the driver calls N > 1, on a real RCCL communicator of one rank: the timed
    region holds >= 2 re-cuts + all-to-all-v's (default cadence 10, counted from the first timed step)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--particles", "2e5", "--steps", "20", "--warmup", "5", "--force-dist",
           "--no-cpu-baseline"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29581")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = _check(lines[0], need_cpu_baseline=False)
    c = d["config"]
    assert c["rebalance_interval"] == 10 and c["rccl_ranks"] == 1 and c["ms_in_handoff"]["handoffs"] >= 2
    assert c["ms_in_handoff"]["collectives_device_ms_total"] > 0 and c["particles_after"] == 200_000


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--force-dist"], ["--self-launch", "--force-dist"]])
def test_bench_runs_and_prints_one_json_line(extra):
    """A small live run (2e5 particles, 6 steps): exit code 0, exactly one JSON line on stdout, contract fields;
    --force-dist drives the N>1 host path (RCCL group of one rank) through the same script."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--particles", "2e5", "--steps", "6", "--warmup", "2",
           "--no-cpu-baseline", "--rebalance-interval", "3", "--overlap-steps", "1", "--anchor-particles", "4e5",
           "--anchor-steps", "3", "--brownian-steady-steps", "30", "--analytic-extra", "4", "--tjunction-steps", "30",
           "--tjunction-particles", "3e5", "--vertex-steps", "3", "--polyhedral-steps", "20", "--polyhedral-particles", "3e5"] + extra
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and r.stdout.strip() == lines[0]
    d = _check(lines[0], need_cpu_baseline=False)
    assert d["steps"] == 6 and d["warmup"] == 2 and d["config"]["particles_total"] == 200_000
    assert d["config"]["particles_after"] == 200_000                      # every boundary reflects: nobody is lost
    # what really ran: 2e5 particles on 12 225 cells = 16 per cell, so the fixed-compare record lookup -- with the flat walk
    # of a 2-D case without a z velocity (last parameter: 9 = 1 + flat)
    assert d["roofline"]["kernel"] == "cpf::step_kernel_stream<false, true, false, false, 9>"
    assert d["roofline"]["traffic"] is None and d["roofline"]["traffic_source"] is None      # other launch size than the PMC run
    if extra:
        h = d["config"]["ms_in_handoff"]
        assert d["config"]["rccl_ranks"] == 1 and h["handoffs"] >= 1 and h["host_ms_total"] > 0
        assert h["collectives_device_ms_total"] > 0 and d["config"]["handoff_fraction_per_step"] is not None
    else:
        b, st = d["config"]["brownian"], d["config"]["ms_per_step_steady"]
        assert b["D"] == 1.5e-5 and b["kernel"] == "cpf::step_kernel_stream<true, true, false, false, 1>" and 0 < b["frac"] < 1
        assert st["steps"] == 100 and st["sorts_inside"] == 1 and st["ms_per_step"] > 0
        f = d["config"]["extra_fused_cycles"]
        assert f["cycles_per_launch"] == 8 and f["launches"] == 10 and f["Mparticle_steps_per_s"] > 0
        # what the tutorials actually run: sustained diffusion (sorts inside), the frozen analytic field, TJunction as its dictionary runs it
        bs, af, tj = d["config"]["brownian_steady"], d["config"]["analytic_field"], d["config"]["tjunction_as_run"]
        assert bs["D"] == 1.5e-5 and bs["sort_interval"] == 25 and bs["sorts_inside"] == 1 and bs["steps"] == 30 and 0 < bs["frac"] < 1
        assert bs["kernel"] == b["kernel"] and bs["frac"] <= bs["kernel_frac"] * 1.02
        assert bs["fused_8_cycles_per_launch"]["ms_per_cycle"] > 0 and 0 < bs["fused_8_cycles_per_launch"]["frac"] < 1
        assert bs["resort"]["ms"] > 0 and bs["resort"]["after_cycles"] == 25
        for blk in (bs, tj):            # the cycles as the fragments issue them: cpf_shard_step(10, CPF_STEP_FUSE_CYCLES), sorts inside
            assert blk["fragment_calls"]["cycles_per_call"] == 10 and blk["fragment_calls"]["ms_per_cycle"] > 0
        assert af["steps"] == 4 and af["kernel"] == d["roofline"]["kernel"] and 0 < af["frac"] < 1 and af["cells_visited_per_particle_step"] > 1
        assert tj["particles"] == 300_000 and tj["cells"] == 248_000 and tj["D"] == 1.5e-5 and tj["records_bytes_once"] == 128 * 248_000      # (box records)
        assert tj["kernel"].startswith("cpf::step_kernel_stream<true, true, false, false,") and 0 < tj["frac"] < 1
        assert tj["mesh_flags"]["all_hex"] == 1 and tj["mesh_flags"]["z_thin"] == 0
        # what is built and had no number in the driver-run line before round 6: the VertexVelocity cycle, the non-hex mesh of configs[4]
        vv, ph = d["config"]["vertex_velocity"], d["config"]["polyhedral_as_run"]
        assert vv["kernel"] == "cpf::step_kernel_stream_vertex<false, true, false, false, 1> (cone locate)" and vv["steps"] == 3 and vv["particles_after"] == 200_000
        assert vv["tets"] == 12 * 12225 and 0 < vv["frac"] <= vv["kernel_frac"] * 1.02 < 1
        assert ph["cells"] == 114540 and ph["cells_with_9_faces"] == 2242 and ph["particles_after"] == 300_000 and ph["D"] == 1.5e-5
        assert ph["velocity_uploads"] == 2 and ph["upload_ms_each"] > 0 and ph["ms_per_step_with_uploads"] >= ph["ms_per_step"]
        assert ph["kernel"].startswith("cpf::step_kernel_stream<true, true, false, false,") and 0 < ph["frac"] < 1
        a = d["config"]["strong_anchor_1e8"]               # the N = 1 point of the strong-scaling curve (here: 4e5)
        assert a["particles"] == 400_000 and a["particles_after"] == 400_000 and a["steps"] == 3 and a["Mparticle_steps_per_s"] > 0
