#!/usr/bin/env python3
"""Turns gpurun_out/<label>_summary.json (tools/profile_run.sh) into the small files committed under profiles/."""
import glob, json, os, shutil, sys

label = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, "gpurun_out", label + "_summary.json")))
# the timed launches run the statistics-off instantiation, the warm-up launches the statistics-on one: take the
# instantiation with the most launches
main = "<false, true, false, false, 0>"         # no Brownian kick, reflecting walls, no stored velocity, statistics off, loop lookup
ks = max((k for k in d["kernels"] if "step_kernel" in k["kernel"] and main in k["kernel"]), key=lambda k: k["calls"])
pmc = max((v for k, v in d["pmc"].items() if main in k), key=lambda v: v["dispatches"])
cal = max(d["calibration_zero_cycle_step"].values(), key=lambda v: v["dispatches"])
n = 10_000_000
ff = (28 * n) / (cal["FETCH_SIZE_KB"] * 1024); wf = (28 * n) / (cal["WRITE_SIZE_KB"] * 1024)
spin = [k for k in d["kernels"] if "step_kernel" in k["kernel"] and "<false, true, false, true, 0>" in k["kernel"]]
hbm = pmc["FETCH_SIZE_KB"] * 1024 * ff + pmc["WRITE_SIZE_KB"] * 1024 * wf
out = dict(label=label, kernel=ks["kernel"].split("(")[0], particles_per_launch=n,
           rocprofv3_kernel_trace=dict(calls=ks["calls"], avg_us=round(ks["avg_us"], 2), min_us=round(ks["min_us"], 2),
                                       max_us=round(ks["max_us"], 2), pct_of_gpu_time=round(ks["pct"], 1),
                                       note="`bench.py --no-cpu-baseline --steady-steps 0 --brownian-extra 0 --fused-extra 0 --anchor-particles 0` under rocprofv3 --kernel-trace --stats: the launches of the headline (statistics-off) instantiation are exactly the 100 timed steps; the device spin-up on a scratch copy of the cloud and the 10 warm-up steps run the statistics-on instantiation <false, true, false, true>, listed separately in the kernel stats (its first launches hit a device that has been idle)"),
           spinup_and_warmup_instantiation=(dict(calls=spin[0]["calls"], avg_us=round(spin[0]["avg_us"], 2),
                                                 first100_avg_us=round(spin[0].get("first100_avg_us", 0.0), 2),
                                                 last100_avg_us=round(spin[0].get("last100_avg_us", 0.0), 2)) if spin else None),
           pmc_raw=dict(FETCH_SIZE_KB=pmc["FETCH_SIZE_KB"], WRITE_SIZE_KB=pmc["WRITE_SIZE_KB"], dispatches=pmc["dispatches"]),
           calibration=dict(what="same kernel, zero cycles: 280 MB read + 280 MB written (known)",
                            FETCH_SIZE_KB=cal["FETCH_SIZE_KB"], WRITE_SIZE_KB=cal["WRITE_SIZE_KB"],
                            fetch_correction=round(ff, 4), write_correction=round(wf, 4)),
           hbm_bytes_per_launch=int(hbm), algorithmic_bytes_per_launch=56 * n,
           traffic_over_algorithmic=round(hbm / (56 * n), 4))
P = os.path.join(ROOT, "profiles")
json.dump(out, open(os.path.join(P, "pmc_latest.json"), "w"), indent=1)
json.dump(out, open(os.path.join(P, label + "_pmc_hbm.json"), "w"), indent=1)
json.dump(dict(label=label, kernels=d["kernels"]), open(os.path.join(P, label + "_rocprofv3_kernel_stats.json"), "w"), indent=1)
for f in glob.glob(os.path.join(ROOT, "gpurun_out", label, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(P, label + "_kernel_stats.csv"))
for f in glob.glob(os.path.join(ROOT, "gpurun_out", label, "stats", "**", "*domain_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(P, label + "_domain_stats.csv"))
print(json.dumps(out, indent=1))
