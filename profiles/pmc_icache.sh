#!/bin/bash
# Instruction-cache counter pass for the fused SQP kernel.  Usage, on the GPU box:
#   bash profiles/pmc_icache.sh <outdir> <batch>
set -u
OUT=${1:-gpurun_out/pmc_ic}
BATCH=${2:-1000000}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 60 rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1
grep -o -E "SQC_ICACHE[A-Z_]*|SQ_IFETCH[A-Z_]*|SQC_INST[A-Z_]*" "$OUT/avail.txt" | sort -u > "$OUT/avail_ic.txt"
timeout -s KILL 240 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES \
  --output-format csv -d "$OUT/ic" -- python3 bench.py --batch "$BATCH" --steps 1 --warmup 0 --cpu-sample 0 --no-host-to-host --no-extra > "$OUT/ic.log" 2>&1 || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float)
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "mpcx_sqp_kernel" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
with open(out + "/summary.txt", "w") as fh:
    for k in sorted(tot):
        fh.write(f"{k:28s} {tot[k]:.6g}\n")
PY
