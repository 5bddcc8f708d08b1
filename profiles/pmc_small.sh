#!/bin/bash
# Memory-side traffic of small batches (every problem in single-problem mode): FETCH_SIZE and
# WRITE_SIZE per launch next to the batch's iteration count.  Usage, on the GPU box:
#   bash profiles/pmc_small.sh <outdir> <batch> [<batch> ...]
set -u
OUT=${1:-gpurun_out/pmc_small}; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for B in "$@"; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d "$OUT/b${B}_$C" -- \
      python3 bench.py --batch "$B" --steps 1 --warmup 0 --cpu-sample 0 --no-host-to-host --no-extra > "$OUT/b${B}_$C.log" 2>&1 || exit 1
  done
done
python3 - "$OUT" "$@" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
for B in sys.argv[2:]:
    vals = {}
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"{out}/b{B}_{C}/*/*counter_collection.csv"):
            for row in csv.DictReader(open(f)):
                if "mpcx_sqp_kernel" in row["Kernel_Name"]:
                    vals[C] = vals.get(C, 0.0) + float(row["Counter_Value"])
    line = [json.loads(l) for l in open(f"{out}/b{B}_FETCH_SIZE.log") if l.startswith(chr(123) + chr(34) + "metric")][-1]
    fl = line["roofline"]["algorithmic_flops_per_launch"]
    it = fl / (50 * 3325.0)
    print(f"B={B:>7s} kernel {line['roofline']['kernel_ms']:.1f} ms  ~{it:.0f} IP iterations  "
          f"FETCH {vals.get('FETCH_SIZE', 0):.4g} KiB  WRITE {vals.get('WRITE_SIZE', 0):.4g} KiB  "
          f"per iteration: fetch {vals.get('FETCH_SIZE', 0) / it:.1f} KiB (x2 = {2 * vals.get('FETCH_SIZE', 0) / it:.1f}), write {vals.get('WRITE_SIZE', 0) / it:.1f} KiB")
PY
