#!/bin/bash
# Round-2 refresh of the measured artefacts that bench.py does not produce (one gpurun call).
set -u
OUT=gpurun_out/refresh2
mkdir -p "$OUT"
timeout -k 10 120 python3 profiles/stamps.py 1000000 > "$OUT/stamps_1M.txt" 2>&1 || exit 1
timeout -k 10 60 python3 profiles/stamps.py 8 > "$OUT/stamps_B8.txt" 2>&1 || exit 1
timeout -k 10 200 python3 profiles/latency.py > "$OUT/latency.txt" 2>&1 || exit 1
timeout -k 10 300 python3 profiles/grid_full.py >> "$OUT/latency.txt" 2>&1 || exit 1
timeout -k 10 300 python3 profiles/rti_latency.py > "$OUT/rti_latency.txt" 2>&1 || exit 1
timeout -k 10 200 python3 bench.py --reachable --steps 3 --warmup 1 > "$OUT/bench_reachable.json" 2> "$OUT/bench_reachable.err" || exit 1
