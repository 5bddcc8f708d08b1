#!/bin/bash
# Instruction-fetch latency of the fused SQP kernel: SQ_IFETCH_LEVEL / SQ_IFETCH = average cycles a
# fetch is in flight.  Usage, on the GPU box:  bash profiles/pmc_ifetch.sh <outdir> <batch> [<batch> ...]
set -u
OUT=${1:-gpurun_out/pmc_if}; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for B in "$@"; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES \
    --output-format csv -d "$OUT/b$B" -- python3 bench.py --batch "$B" --steps 1 --warmup 0 --cpu-sample 0 --no-host-to-host --no-extra > "$OUT/b$B.log" 2>&1 || exit 1
done
python3 - "$OUT" "$@" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for B in sys.argv[2:]:
    tot = collections.defaultdict(float)
    for f in glob.glob(f"{out}/b{B}/*/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if "mpcx_sqp_kernel" in row["Kernel_Name"]:
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
    print(f"B={B}: " + "  ".join(f"{k}={v:.4g}" for k, v in sorted(tot.items()))
          + (f"  level/fetch={tot['SQ_IFETCH_LEVEL'] / tot['SQ_IFETCH']:.1f}" if tot.get("SQ_IFETCH") else ""))
PY
