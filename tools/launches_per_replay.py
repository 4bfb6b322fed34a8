"""Launches per replayed training step from a rocprofv3 *_kernel_trace.csv: kernels between two consecutive k_step_ctl_advance
launches (the first node of the Trainer's captured graph), plus the busy time of one replay.
usage: python tools/launches_per_replay.py FILE"""
import csv
import statistics
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_step_ctl_advance" in r["Kernel_Name"]]
if len(marks) < 3:
    raise SystemExit("fewer than three replays in the trace")
gaps = [b - a for a, b in zip(marks[:-1], marks[1:])]
busy = [sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a:b]) / 1e3 for a, b in zip(marks[:-1], marks[1:])]
span = [(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3 for a, b in zip(marks[:-1], marks[1:])]
print(f"replays {len(gaps)}: launches per replay median {statistics.median(gaps)} (min {min(gaps)}, max {max(gaps)}), "
      f"kernel time per replay {statistics.median(busy):.1f} us, replay-to-replay {statistics.median(span):.1f} us")
