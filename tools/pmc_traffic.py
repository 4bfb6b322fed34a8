"""Aggregate two rocprofv3 --pmc passes of bench.py (FETCH_SIZE, WRITE_SIZE) into profiles/pmc_traffic.json: HBM-side bytes per
launch of every conv-stack kernel class, tied to the batch / dtype / kernel-source digest they were measured on (bench.py reports
roofline.traffic from this file only when all three match the running build).

    rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline
    rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d gpurun_out/pmc_write -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write LABEL [batch] [dtype]

FETCH_SIZE is reported in KB at 1/2 of the streamed bytes on gfx950 (MI355X_MICROARCH.md: x2), WRITE_SIZE in KB, exact."""
import csv
import glob
import json
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bench import kernel_source_digest  # noqa: E402

CLASS_OF = {"k_dw_bwd": "dwconv3x3_bwd", "k_pw_bwd": "pwconv1x1_bwd", "k_dw_fwd": "dwconv3x3_fwd", "k_pw_fwd": "pwconv1x1_fwd",
            "k_stem_bwd": "conv_stem_bwd", "k_stem_fwd": "conv_stem_fwd", "k_gap_fwd": "gap_fwd", "k_logmel": "logmel_specaug"}


def per_kernel(d, counter, skip=2):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, v in acc.items():
        v = v[skip:] if len(v) > skip else v
        for key, cls in CLASS_OF.items():
            if key in k:
                out.setdefault(cls, []).extend(v)
    return {c: sum(v) / len(v) for c, v in out.items()}


fetch_dir, write_dir, label = sys.argv[1], sys.argv[2], sys.argv[3]
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 512
dtype = sys.argv[5] if len(sys.argv) > 5 else "bf16"
fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
rows = {c: {"fetch_kb_raw": fetch[c], "read_bytes": 2 * 1024 * fetch[c], "write_bytes": 1024 * write.get(c, 0.0)} for c in fetch}
out = {"label": label, "batch": batch, "dtype": dtype, "csrc_digest": kernel_source_digest(),
       "method": "rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes over bench.py; FETCH_SIZE x2 (gfx950), KB -> bytes; "
                 "mean over launches after the first two of each kernel",
       "bytes_per_launch": {c: round(r["read_bytes"] + r["write_bytes"]) for c, r in rows.items()}, "detail": rows}
(ROOT / "profiles" / "pmc_traffic.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out["bytes_per_launch"]))
