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
    gx = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    gy = int(r.get("Grid_Size_Y", 1)) // max(1, int(r.get("Workgroup_Size_Y", 1)))
    gz = int(r.get("Grid_Size_Z", 1)) // max(1, int(r.get("Workgroup_Size_Z", 1)))
    tmpl = re.search(r"k_[a-z0-9_]+<([^>]*)>", r["Kernel_Name"])          # template arguments tell a GEMM's operand layout apart
    name = name + ("<" + tmpl.group(1).replace(" ", "") + ">" if tmpl and len(tmpl.group(1)) < 40 else "")
    acc[(name, (gx, gy, gz), int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid, wg), d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    g = "x".join(str(v) for v in grid)
    print(f"{name:44s} grid {g:>14s} x {wg:4d}  calls {len(d):5d}  avg {sum(d) / len(d):8.2f} us  min {min(d):8.2f}  total {sum(d) / 1e3:8.3f} ms")
