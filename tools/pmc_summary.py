"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter per launch."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void mbar::", "").replace("mbar::", "")
    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, ctrs in agg.items():
    if not any(k in name for k in ("k_lse", "k_gram")):
        continue
    print(name)
    for c, v in sorted(ctrs.items()):
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
