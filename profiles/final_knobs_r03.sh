#!/bin/bash
# Same-box re-check of the scheduling knobs on the round's final kernel, three workloads, kernel ms per launch.
set -u
mkdir -p gpurun_out/knob3
run() {
  tag=$1; wl=$2; shift; shift
  env "$@" timeout -k 10 150 python3 bench.py $wl --steps 3 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra > gpurun_out/knob3/$tag.json 2> gpurun_out/knob3/$tag.err || return 1
  python3 -c "import json; d=json.loads(open('gpurun_out/knob3/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],1))"
}
for wl in "" "--obstacle" "--reachable"; do
  t=h; [ "$wl" = "--obstacle" ] && t=o; [ "$wl" = "--reachable" ] && t=r
  run ${t}_base "$wl" A=0 && run ${t}_svc6 "$wl" MPCX_SVC_MIN=6 && run ${t}_svc12 "$wl" MPCX_SVC_MIN=12 && run ${t}_lanes2 "$wl" MPCX_BOOST=16,1,8192,2 && run ${t}_lanes8 "$wl" MPCX_BOOST=16,1,8192,8 && run ${t}_limit16 "$wl" MPCX_HANDOVER=320,16,128 && run ${t}_limit256 "$wl" MPCX_HANDOVER=320,256,128 && run ${t}_base2 "$wl" A=0 || exit 1
done
