#!/bin/bash
# Same-box sweep of the service threshold (lanes that must wait before write-out / fetch / sweep L run).
set -u
mkdir -p gpurun_out/svc
run() {
  tag=$1; wl=$2; shift; shift
  env "$@" timeout -k 10 150 python3 bench.py $wl --steps 3 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra > gpurun_out/svc/$tag.json 2> gpurun_out/svc/$tag.err || return 1
  python3 -c "import json; d=json.loads(open('gpurun_out/svc/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],1))"
}
for wl in "" "--obstacle" "--reachable"; do
  t=h; [ "$wl" = "--obstacle" ] && t=o; [ "$wl" = "--reachable" ] && t=r
  run ${t}_s4 "$wl" A=0 && run ${t}_s6 "$wl" MPCX_SVC_MIN=6 && run ${t}_s8 "$wl" MPCX_SVC_MIN=8 && run ${t}_s12 "$wl" MPCX_SVC_MIN=12 && run ${t}_s16 "$wl" MPCX_SVC_MIN=16 && run ${t}_s4b "$wl" A=0 || exit 1
done
