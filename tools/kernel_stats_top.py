"""Print the top rows of a rocprofv3 *_kernel_stats.csv with short kernel names.  usage: python tools/kernel_stats_top.py FILE [N]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)} launches of {len(rows)} kernels")
for r in rows[:n]:
    m = re.search(r"k_[a-z0-9_]+", r["Name"])
    name = m.group(0) if m else re.sub(r"\(.*", "", r["Name"])[-48:]
    print(f"{name:40s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs']) / 1e3:9.2f} us  total {float(r['TotalDurationNs']) / 1e6:8.3f} ms  {float(r['Percentage']):6.2f} %")
