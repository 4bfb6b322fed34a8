"""Per-launch view of a rocprofv3 *_kernel_trace.csv: average duration of every (kernel, grid size) pair -- for models whose
layers reuse one kernel at many shapes (MobileNetV3) the per-kernel average of --stats hides which layer is slow.
usage: python tools/kernel_trace_by_grid.py FILE [name-substring ...]"""
import csv
import re
import sys
from collections import defaultdict

rows = csv.DictReader(open(sys.argv[1]))
want = sys.argv[2:]
acc = defaultdict(list)
for r in rows:
    m = re.search(r"k_[a-z0-9_]+", r["Kernel_Name"])
    name = m.group(0) if m else re.sub(r"\(.*", "", r["Kernel_Name"])[-40:]
    if want and not any(w in name for w in want):
        continue
    grid = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    acc[(name, grid, int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid, wg), d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{name:28s} grid {grid:6d} x {wg:4d}  calls {len(d):5d}  avg {sum(d) / len(d):8.2f} us  min {min(d):8.2f}  total {sum(d) / 1e3:8.3f} ms")
