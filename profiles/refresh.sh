#!/bin/bash
# Regenerates the measured artefacts of this directory on the GPU box (one gpurun call):
#   bash profiles/refresh.sh          -> files under gpurun_out/refresh/
# Each GPU step has its own time limit; the script stops at the first failure.
set -u
OUT=gpurun_out/refresh
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
timeout -k 10 300 python3 bench.py --reachable --steps 3 --warmup 1 > "$OUT/bench_reachable.json" 2> "$OUT/bench_reachable.err" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- \
    python3 bench.py --steps 3 --warmup 0 --cpu-sample 0 --no-host-to-host --no-extra > "$OUT/prof.log" 2>&1 || exit 1
cp "$OUT"/prof/*/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null
timeout -k 10 200 python3 profiles/latency.py > "$OUT/latency.txt" 2>&1 || exit 1
timeout -k 10 300 python3 profiles/grid_full.py >> "$OUT/latency.txt" 2>&1 || exit 1
timeout -k 10 300 python3 profiles/rti_latency.py > "$OUT/rti_latency.txt" 2>&1 || exit 1
timeout -k 10 120 python3 profiles/stamps.py 1000000 > "$OUT/stamps_1M.txt" 2>&1 || exit 1
timeout -k 10 60 python3 profiles/stamps.py 8 > "$OUT/stamps_B8.txt" 2>&1 || exit 1
bash profiles/pmc_passes.sh "$OUT/pmc" 1000000 > "$OUT/pmc_summary.txt" 2>&1 || exit 1
bash profiles/pmc_extra.sh "$OUT/pmc_extra" 1000000 > "$OUT/pmc_extra.txt" 2>&1 || exit 1
