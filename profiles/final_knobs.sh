#!/bin/bash
# Same-box re-check of the scheduling knobs on the final kernel (one-obstacle and reachable workloads).
set -u
mkdir -p gpurun_out/knob
run() {
  tag=$1; wl=$2; shift; shift
  env "$@" timeout -k 10 150 python3 bench.py $wl --steps 3 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra > gpurun_out/knob/$tag.json 2> gpurun_out/knob/$tag.err || return 1
  python3 -c "import json; d=json.loads(open('gpurun_out/knob/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],1))"
}
for wl in "--obstacle" "--reachable"; do
  t=o; [ "$wl" = "--reachable" ] && t=r
  run ${t}_base "$wl" A=0 && run ${t}_svc6 "$wl" MPCX_SVC_MIN=6 && run ${t}_svc12 "$wl" MPCX_SVC_MIN=12 && run ${t}_hold5e-3 "$wl" MPCX_HOLD=0.005,48 && run ${t}_hold2e-4 "$wl" MPCX_HOLD=0.0002,48 && run ${t}_b14 "$wl" MPCX_BOOST=14,1,8192,4 && run ${t}_b18 "$wl" MPCX_BOOST=18,1,8192,4 && run ${t}_l6 "$wl" MPCX_BOOST=16,1,8192,6 && run ${t}_base2 "$wl" A=0 || exit 1
done
