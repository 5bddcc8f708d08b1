#!/bin/bash
# Same-box sweep of the hold policy (MPCX_HOLD=alpha,extra) on the headline, the obstacle and the reachable workload.
set -u
mkdir -p gpurun_out/hold
run() {
  tag=$1; wl=$2; shift; shift
  env "$@" timeout -k 10 150 python3 bench.py $wl --steps 2 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra > gpurun_out/hold/$tag.json 2> gpurun_out/hold/$tag.err || return 1
  python3 -c "import json; d=json.loads(open('gpurun_out/hold/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],1))"
}
for wl in "" "--obstacle" "--reachable"; do
  t=h; [ "$wl" = "--obstacle" ] && t=o; [ "$wl" = "--reachable" ] && t=r
  run ${t}_off "$wl" A=0 && run ${t}_h01_24 "$wl" MPCX_HOLD=0.01,24 && run ${t}_h01_12 "$wl" MPCX_HOLD=0.01,12 && run ${t}_h001_24 "$wl" MPCX_HOLD=0.001,24 && run ${t}_off2 "$wl" A=0 || exit 1
done
