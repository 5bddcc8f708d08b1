#!/bin/bash
# A/B of the dedication threshold (MPCX_DEDICATE: iterations after which a problem keeps its wavefront
# from taking new problems; 0 = never) on the headline batch, two other seeds and the reachable workload.
set -u
out=gpurun_out/r03_dedicate_sweep.txt
: > $out
run() {   # label, env value, bench flags...
    local label=$1 ded=$2; shift 2
    local line
    line=$(MPCX_DEDICATE=$ded timeout -k 10 150 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra "$@") || { echo "$label FAILED" >> $out; exit 1; }
    echo "$label dedicate=$ded $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.1f value %.0f" % (d["ms_per_step"], d["value"]))')" >> $out
}
for d in 0 512 256 1024 2048; do run "seed 20250310" $d; done
for s in 20250311 20250315; do for d in 0 512; do run "seed $s" $d --seed $s; done; done
for d in 0 512 1024; do run "reachable 20250310" $d --reachable; done
cat $out
