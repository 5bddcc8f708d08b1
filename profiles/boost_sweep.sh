#!/bin/bash
# the ageing threshold (iterations in the one-lane-per-problem sweeps before a problem goes to single-problem
# mode) and the hold policy, re-swept after the burst iteration got cheaper
set -u
out=gpurun_out/r03_boost_sweep.txt
: > $out
run() {
    local label=$1; shift
    local line
    line=$(env "$@" timeout -k 10 150 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra) || { echo "$label FAILED" >> $out; exit 1; }
    echo "$label $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.1f value %.0f" % (d["ms_per_step"], d["value"]))')" >> $out
}
run "default (16,1,8192,4)" X=1
for m in 8 10 12 14 20; do run "boost min $m" MPCX_BOOST=$m,1,8192,4; done
run "boost min 12 lanes 8" MPCX_BOOST=12,1,8192,8
run "boost min 12 hold 0,0" MPCX_BOOST=12,1,8192,4 MPCX_HOLD=0,0
run "boost min 16 hold 0,0" MPCX_HOLD=0,0
run "boost min 12 mul 2" MPCX_BOOST=12,1,8192,4,2
cat $out
