#!/usr/bin/env python3
"""Turns gpurun_out/<label>_summary.json (tools/pmc.sh) into the small files committed under profiles/:
  profiles/<label>_kernel_stats.json     every kernel of the run (rocprofv3 --kernel-trace --stats)
  profiles/<label>_kernel_stats.csv      rocprofv3's own stats table
  profiles/<label>_pmc_hbm.json          the step kernels: calls, average duration, HBM bytes per launch (PMC, corrected)
and, with --bench (the run was `bench.py`'s timed region), profiles/pmc_latest.json, which bench.py replays as
roofline.traffic.  python tools/condense_profile.py LABEL [--bench] [--algo-bytes N [--match SUBSTRING]]
(--algo-bytes annotates the step kernels whose name contains SUBSTRING -- all of them without --match -- with the algorithmic bytes
of one launch: particles x 56, or x 64 with the kick)"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
label = sys.argv[1]
algo = int(sys.argv[sys.argv.index("--algo-bytes") + 1]) if "--algo-bytes" in sys.argv else None
match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
d = json.load(open(os.path.join(ROOT, "gpurun_out", label + "_summary.json")))
P = os.path.join(ROOT, "profiles")
steps = d.get("step_kernels", [])
for k in steps:
    if algo and match in k["kernel"]:
        k["algorithmic_bytes_per_launch"] = algo
        k["traffic_over_algorithmic"] = round(k["hbm_bytes_per_launch"] / algo, 4)
        k["achieved_GBs"] = round(algo / (k["avg_us"] * 1e-6) / 1e9, 1)
        k["frac_of_8TBs"] = round(algo / (k["avg_us"] * 1e-6) / 8e12, 4)
json.dump(dict(label=label, step_kernels=steps), open(os.path.join(P, label + "_pmc_hbm.json"), "w"), indent=1)
json.dump(dict(label=label, kernels=d["kernels"]), open(os.path.join(P, label + "_kernel_stats.json"), "w"), indent=1)
for f in glob.glob(os.path.join(ROOT, "gpurun_out", label, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(P, label + "_kernel_stats.csv"))
if "--bench" in sys.argv:
    # headline: no kick, reflecting walls, no stored velocity, statistics off, loop lookup -- with the flat walk (8) since round 4
    mains = ("<false, true, false, false, 8>", "<false, true, false, false, 0>")
    k = max((k for k in steps if any(m in k["kernel"] for m in mains)), key=lambda k: k["calls"])
    n = 10_000_000
    out = dict(label=label, kernel=k["kernel"], particles_per_launch=n,
               rocprofv3_kernel_trace=dict(calls=k["calls"], avg_us=round(k["avg_us"], 2), min_us=round(k["min_us"], 2), max_us=round(k["max_us"], 2),
                                           note="bench.py's timed region under rocprofv3 --kernel-trace --stats (tools/pmc.sh): the launches of the "
                                                "statistics-off instantiation are exactly the timed steps; spin-up and warm-up run the statistics-on one"),
               pmc_raw=dict(FETCH_SIZE_KB=k["FETCH_SIZE_KB"], WRITE_SIZE_KB=k["WRITE_SIZE_KB"]), corrections=k["corrections"],
               hbm_bytes_per_launch=k["hbm_bytes_per_launch"], algorithmic_bytes_per_launch=56 * n,
               traffic_over_algorithmic=round(k["hbm_bytes_per_launch"] / (56 * n), 4))
    json.dump(out, open(os.path.join(P, "pmc_latest.json"), "w"), indent=1)
print(json.dumps(steps, indent=1)[:3000])
