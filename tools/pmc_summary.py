"""Folds the counter_collection CSVs of separate `rocprofv3 --pmc` passes into one summary:
mean per dispatch of every (kernel, counter).  usage: pmc_summary.py OUT.csv PASSNAME=DIR ..."""
import csv, glob, re, sys
from collections import defaultdict


def short(name):
    m = re.search(r"([a-z][a-z0-9_]*_kernel)", name)
    kern = m.group(1) if m else name[:40]
    if "ILb0E" in name or "<false>" in name:
        kern += "<false>"
    if "ILb1E" in name or "<true>" in name:
        kern += "<true>"
    return kern


rows = []
for arg in sys.argv[2:]:
    pname, d = arg.split("=", 1)
    acc = defaultdict(lambda: [0.0, set()])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (short(r["Kernel_Name"]), r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1].add(r["Dispatch_Id"])
    for (kern, ctr), (tot, disp) in sorted(acc.items()):
        if "kernel" in kern and "synth" not in kern:
            rows.append((pname, kern, ctr, len(disp), round(tot / len(disp), 3)))
with open(sys.argv[1], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["pass", "kernel", "counter", "dispatches", "mean_per_dispatch"])
    w.writerows(rows)
print(f"{len(rows)} rows -> {sys.argv[1]}")
