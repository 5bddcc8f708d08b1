#!/bin/bash
# Bytes on the memory side of the L2 for two service thresholds (FETCH_SIZE and WRITE_SIZE in passes of their own).
set -u
OUT=gpurun_out/pmc_svc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for s in 4 16; do
  export MPCX_SVC_MIN=$s
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d "$OUT/s${s}_$c" -- \
      python3 bench.py --batch 1000000 --steps 1 --warmup 0 --cpu-sample 0 --no-host-to-host --no-extra > "$OUT/s${s}_$c.log" 2>&1 || exit 1
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/s*_*_SIZE")):
    tot = 0.0
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if "mpcx_sqp_kernel" in row["Kernel_Name"]:
                tot += float(row["Counter_Value"])
    print(d.split('/')[-1], f"{tot:.6g}")
PY
