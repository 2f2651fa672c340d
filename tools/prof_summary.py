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


FETCH_CORR, WRITE_CORR = 1.9975, 0.9934      # gfx950: calibrated on the zero-cycle launch (profiles/r03_pmc_hbm.json, tools/calib.py)


def sq_table(out_dir):
    """Per step-kernel instantiation: every SQ / GRBM counter of the sq* passes, mean per launch and per 64-particle tile
    (tiles from the kernel's grid are not in the CSV: the caller divides by its own tile count; 1e7 particles = 156 250)."""
    import collections
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(out_dir, "sq*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "step_kernel" not in kn:
                continue
            a = agg[kn.split("(")[0][-48:]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for key, d in agg.items():
        print("==", key)
        for c, (n, tot) in sorted(d.items()):
            print("  %-34s launches %4d  mean/launch %16.1f   per tile of a 1e7-particle launch %10.2f" % (c, n, tot / n, tot / n / 156250.0))


def main():
    if sys.argv[1] == "--sq":
        return sq_table(sys.argv[2])
    label, stats_dir = sys.argv[1], sys.argv[2]
    out = dict(label=label, kernels=kernel_stats(stats_dir)[:12])
    if len(sys.argv) >= 5:
        f = counter_per_kernel(sys.argv[3], "FETCH_SIZE"); w = counter_per_kernel(sys.argv[4], "WRITE_SIZE")
        steps = []
        for k in out["kernels"]:
            full = [n for n in f if n[:140] == k["kernel"]]
            if "step_kernel" not in k["kernel"] or not full:
                continue
            fk, wk = f[full[0]]["mean"], (w.get(full[0]) or {}).get("mean") or 0.0
            steps.append(dict(kernel=k["kernel"].split("(")[0], calls=k["calls"], avg_us=k["avg_us"], min_us=k["min_us"], max_us=k["max_us"],
                              FETCH_SIZE_KB=fk, WRITE_SIZE_KB=wk,
                              hbm_bytes_per_launch=int(fk * 1024 * FETCH_CORR + wk * 1024 * WRITE_CORR),
                              corrections="FETCH x %.4f, WRITE x %.4f" % (FETCH_CORR, WRITE_CORR)))
        out["step_kernels"] = steps
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
