#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output directories into one small JSON/markdown summary (kept under profiles/).
usage: prof_summary.py <label> <stats_dir> [<pmc_fetch_dir> <pmc_write_dir> [<calib_fetch_dir> <calib_write_dir>]]"""
import csv
import glob
import json
import os
import sys


def rows(d, suffix):
    out = []
    for f in glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True):
        with open(f, newline="") as fh:
            out.extend(csv.DictReader(fh))
    return out


def kernel_stats(d):
    tr = rows(d, "kernel_trace.csv")
    agg = {}
    for r in tr:
        name = r.get("Kernel_Name", "?")
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3   # us
        a = agg.setdefault(name, dict(calls=0, total_us=0.0, min_us=1e30, max_us=0.0, vgpr=r.get("VGPR_Count") or r.get("Arch_VGPR_Count"),
                                      sgpr=r.get("SGPR_Count"), lds=r.get("LDS_Block_Size"), wg=r.get("Workgroup_Size")))
        a["calls"] += 1; a["total_us"] += dur; a["min_us"] = min(a["min_us"], dur); a["max_us"] = max(a["max_us"], dur)
        a.setdefault("_each", []).append((int(r["Start_Timestamp"]), dur))
    tot = sum(a["total_us"] for a in agg.values()) or 1.0
    lst = []
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["total_us"]):
        each = [d_ for _, d_ in sorted(a.pop("_each"))]
        a = dict(a, kernel=name[:140], avg_us=a["total_us"] / a["calls"], pct=100 * a["total_us"] / tot)
        # the LAST 100 launches of a kernel: for the step kernel of `bench.py --steady-steps 0 --brownian-extra 0` that
        # is the timed region (what comes before is the device spin-up on a scratch copy and nothing comes after)
        if len(each) > 100:
            a["last100_avg_us"] = sum(each[-100:]) / 100.0
            a["first100_avg_us"] = sum(each[:100]) / 100.0
        lst.append(a)
    return lst


def counter_per_kernel(d, counter):
    tr = rows(d, "counter_collection.csv")
    agg = {}
    for r in tr:
        if r.get("Counter_Name") != counter:
            continue
        name = r.get("Kernel_Name", "?")
        a = agg.setdefault(name, [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
    return {k: dict(dispatches=v[0], mean=v[1] / v[0]) for k, v in agg.items()}


def main():
    label, stats_dir = sys.argv[1], sys.argv[2]
    out = dict(label=label, kernels=kernel_stats(stats_dir)[:12])
    if len(sys.argv) >= 5:
        f = counter_per_kernel(sys.argv[3], "FETCH_SIZE"); w = counter_per_kernel(sys.argv[4], "WRITE_SIZE")
        out["pmc"] = {k[:140]: dict(FETCH_SIZE_KB=f[k]["mean"], WRITE_SIZE_KB=w.get(k, {}).get("mean"), dispatches=f[k]["dispatches"])
                      for k in f if "step_kernel" in k}
    if len(sys.argv) >= 7:
        f = counter_per_kernel(sys.argv[5], "FETCH_SIZE"); w = counter_per_kernel(sys.argv[6], "WRITE_SIZE")
        out["calibration_zero_cycle_step"] = {k[:140]: dict(FETCH_SIZE_KB=f[k]["mean"], WRITE_SIZE_KB=w.get(k, {}).get("mean"),
                                                            dispatches=f[k]["dispatches"]) for k in f if "step_kernel" in k}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
