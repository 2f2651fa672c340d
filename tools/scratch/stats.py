import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if re.search(sys.argv[2], n):
        print(re.sub(r"\(.*", "", n)[:60], r["Calls"], "avg %.1f us" % (float(r["AverageNs"]) / 1e3), "min %.1f max %.1f" % (int(r["MinNs"]) / 1e3, int(r["MaxNs"]) / 1e3))
