"""Aggregates rocprofv3 --pmc CSVs (one directory per pass) per kernel: dispatch count and the per-dispatch mean of
every counter.  Usage: python tools/pmc_report.py <pass dir> [<pass dir> ...] [--match substring]"""
import csv, glob, os, re, sys
from collections import defaultdict

args = [a for a in sys.argv[1:] if not a.startswith("--")]
match = None
if "--match" in sys.argv:
    match = sys.argv[sys.argv.index("--match") + 1]
    args = [a for a in args if a != match]
tab = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> [value per dispatch]
for d in args:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(float)                   # (dispatch, kernel, counter) -> sum over rows (dimensions)
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"])
            name = re.sub(r"^void ", "", name)
            per[(r["Dispatch_Id"], name, r["Counter_Name"])] += float(r["Counter_Value"])
        for (disp, name, ctr), v in per.items():
            tab[name][ctr].append(v)
for name in sorted(tab):
    if match and match not in name:
        continue
    for ctr in sorted(tab[name]):
        v = tab[name][ctr]
        print(f"{name} | {ctr} | dispatches {len(v)} | mean {sum(v) / len(v):.6g} | min {min(v):.6g} | max {max(v):.6g}")
