#!/bin/bash
# Counter passes for the fused SQP kernel; one rocprofv3 run per pass (the
# hardware holds 8 SQ / 4 TCC counters at a time).  Usage, on the GPU box:
#   bash profiles/pmc_passes.sh <outdir> <batch>
set -u
OUT=${1:-gpurun_out/pmc}
BATCH=${2:-131072}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- \
    python3 bench.py --batch "$BATCH" --steps 1 --warmup 0 --cpu-sample 0 --no-host-to-host --no-extra > "$OUT/$name.log" 2>&1 || return 1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU &&
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU &&
run tcc1 TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float)
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "mpcx_sqp_kernel" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
for k in sorted(tot):
    print(f"{k:28s} {tot[k]:.6g}")
PY
