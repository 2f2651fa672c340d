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


def _check(line, need_cpu_baseline):
    d = json.loads(line)
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert d["metric"] == "Mparticle-steps/s" and d["unit"] == "Mparticle-steps/s" and d["higher_is_better"] is True
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None and d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert ROOFLINE <= set(r) and r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    # value = particles * steps / time, in units of 1e6
    assert abs(d["value"] - d["config"]["particles_total"] * 1e-6 / (d["ms_per_step"] * 1e-3)) / d["value"] < 2e-3
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


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--force-dist"]])
def test_bench_runs_and_prints_one_json_line(extra):
    """A small live run (2e5 particles, 6 steps): exit code 0, exactly one JSON line on stdout, contract fields;
    --force-dist drives the N>1 host path (RCCL group of one rank) through the same script."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--particles", "2e5", "--steps", "6", "--warmup", "2",
           "--no-cpu-baseline", "--rebalance-interval", "3", "--overlap-steps", "1"] + extra
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = _check(lines[0], need_cpu_baseline=False)
    assert d["steps"] == 6 and d["warmup"] == 2 and d["config"]["particles_total"] == 200_000
    assert d["config"]["particles_after"] == 200_000                      # every boundary reflects: nobody is lost
    # what really ran: 2e5 particles on 12 225 cells = 16 per cell, so the fixed-compare record lookup (last parameter)
    assert d["roofline"]["kernel"] == "cpf::step_kernel_stream<false, true, false, false, true>"
    assert d["roofline"]["traffic"] is None and d["roofline"]["traffic_source"] is None      # other launch size than the PMC run
    if extra:
        h = d["config"]["ms_in_handoff"]
        assert d["config"]["rccl_ranks"] == 1 and h["handoffs"] >= 1 and h["host_ms_total"] > 0
        assert h["collectives_device_ms_total"] > 0 and d["config"]["handoff_fraction_per_step"] is not None
    else:
        b, st = d["config"]["brownian"], d["config"]["ms_per_step_steady"]
        assert b["D"] == 1.5e-5 and b["kernel"] == "cpf::step_kernel_stream<true, true, false, false, true>" and 0 < b["frac"] < 1
        assert st["steps"] == 100 and st["sorts_inside"] == 1 and st["ms_per_step"] > 0
        f = d["config"]["extra_fused_cycles"]
        assert f["cycles_per_launch"] == 8 and f["launches"] == 10 and f["Mparticle_steps_per_s"] > 0
