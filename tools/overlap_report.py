"""From a rocprofv3 --kernel-trace CSV of bench.py: for every conv-stack kernel launch, how much of it ran while a k_logmel
launch (the side-stream input stage) was executing, and the mean duration of the launches that overlapped it fully, partly or
not at all.  usage: python tools/overlap_report.py KERNEL_TRACE.csv"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
name_key = "Kernel_Name"
lm = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_logmel" in r[name_key])
lm = lm[3:]                       # skip the warm-up / "alone" launches


def overlap(a, b):
    t = 0
    for s, e in lm:
        if e <= a:
            continue
        if s >= b:
            break
        t += min(b, e) - max(a, s)
    return t


t_first = lm[0][0] if lm else 0
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    n = r[name_key]
    short = n.split("(")[0].split("<")[0].split("IDF")[0].replace("_ZN12_GLOBAL__N_1", "").lstrip("0123456789")
    if "k_logmel" in n or not short.startswith("k_"):
        continue
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if a < t_first:
        continue
    f = overlap(a, b) / max(b - a, 1)
    cls = "full" if f > 0.95 else ("none" if f < 0.05 else "part")
    acc[short][cls].append((b - a) / 1e3)
print(f"{'kernel':22s} {'none: n  mean us':>20s} {'part: n  mean us':>20s} {'full: n  mean us':>20s}")
for k, d in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
    cells = []
    for cls in ("none", "part", "full"):
        v = d.get(cls, [])
        cells.append(f"{len(v):5d} {sum(v) / len(v):9.1f}" if v else f"{0:5d} {'-':>9s}")
    print(f"{k:22s} {cells[0]:>20s} {cells[1]:>20s} {cells[2]:>20s}")
