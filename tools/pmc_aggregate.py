"""Average a rocprofv3 --pmc counter per kernel name from *_counter_collection.csv files.
usage: python tools/pmc_aggregate.py DIR [skip_first_n_launches_per_kernel]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
acc = defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v = v[skip:] if len(v) > skip else v
    print(f"{c}\t{sum(v) / len(v):14.1f}\tn={len(v)}\t{k[:110]}")
