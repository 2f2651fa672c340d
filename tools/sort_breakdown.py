#!/usr/bin/env python3
"""Condenses a rocprofv3 kernel trace of tools/sort_timing.py into profiles/rNN_sort_breakdown.json: the kernels of one re-sort
in launch order, averaged over every sort of each kind (library / hand-written) and case.
  rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python3 tools/sort_timing.py
  python tools/sort_breakdown.py /tmp/prof/.../*_kernel_trace.csv profiles/r05_sort_breakdown.json"""
import collections, csv, json, re, sys


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    if "onesweep" in n:
        return "rocprim onesweep kernel"
    if "rocprim" in n:
        return "rocprim " + n.split("::")[-1][:30]
    return n


def main():
    rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
    starts = ("cpf::sort_keys_kernel", "cpf::rs_keys_hist_kernel", "cpf::iota_kernel")
    case, cur, acc = None, [], collections.OrderedDict()
    for r in rows:
        n, d = short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if "step_kernel_stream" in n:                      # the ageing launch: what follows are this case's timed sorts
            case, cur = ("pitzDaily 1e7" if ", 0>" in n else "TJunction 4e6"), []
            continue
        if case is None:
            continue
        cur.append((n, d, int(r["Grid_Size_X"])))
        if n.startswith("cpf::gather_a"):
            first = [k for k, c in enumerate(cur) if c[0] in starts]
            seq, cur = (cur[min(first):] if first else []), []
            if not seq or (seq[-1][2] > 3e6) != case.startswith("pitz"):     # (the gather takes two destinations per thread)
                continue
            kind = "0 (hipcub)" if any("rocprim" in c[0] for c in seq) else ("2 (tile reorder)" if any("rt_scatter" in c[0] for c in seq)
                                                                             else "1 (wide digits, removed)")
            acc.setdefault("%s, sort_method %s" % (case, kind), []).append([(a, b) for a, b, _ in seq])
    cases = {}
    for k, seqs in acc.items():
        L = len(seqs[0]); seqs = [s for s in seqs if len(s) == L]
        ks = [(seqs[0][i][0], round(sum(s[i][1] for s in seqs) / len(seqs), 2)) for i in range(L)]
        cases[k] = {"sorts_averaged": len(seqs), "kernels": ks, "sum_us": round(sum(d for _, d in ks), 1)}
    out = {"source": "rocprofv3 --kernel-trace --output-format csv -- python3 tools/sort_timing.py (one MI355X); every sort of each kind per "
                     "case averaged, kernels in launch order, microseconds (the profiler adds to every launch: wall times per sort are in "
                     "the *_sort_timing.jsonl beside this file)", "cases": cases}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    for k, v in cases.items():
        print(k, v["sorts_averaged"], v["sum_us"], [(a.split("::")[-1][:16], b) for a, b in v["kernels"]])


if __name__ == "__main__":
    main()
