#!/bin/bash
# SQ counters of k_logmel alone (B = 512, full-device grid): instruction mix and what the waves wait for.
# usage (GPU box): bash tools/logmel_sq_counters.sh LABEL   -> gpurun_out/LABEL/logmel_sq.txt
L=${1:-sq}
O=gpurun_out/$L
mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INST_CYCLES_SALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d $O/$tag -- python3 tools/time_logmel.py 512 > /dev/null 2>&1 || echo "pass failed: $set"
done
python3 - $O <<'PY' | tee $O/logmel_sq.txt
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_logmel" in r["Kernel_Name"]:
            t = tot[r["Counter_Name"]]; t[0] += float(r["Counter_Value"]); t[1] += 1
for k, (v, n) in sorted(tot.items()):
    print(f"{k:28s} {v / max(n, 1):16.0f} per launch ({n} launches)")
PY
