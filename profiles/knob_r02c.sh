#!/bin/bash
# Same-box sweep of the ageing threshold with the hold policy on (headline, reachable, obstacle).
set -u
mkdir -p gpurun_out/knob
run() {
  tag=$1; wl=$2; shift; shift
  env "$@" timeout -k 10 150 python3 bench.py $wl --steps 3 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra > gpurun_out/knob/$tag.json 2> gpurun_out/knob/$tag.err || return 1
  python3 -c "import json; d=json.loads(open('gpurun_out/knob/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],1))"
}
for wl in "" "--reachable" "--obstacle"; do
  t=h; [ "$wl" = "--obstacle" ] && t=o; [ "$wl" = "--reachable" ] && t=r
  run ${t}_base "$wl" A=0 && run ${t}_b8 "$wl" MPCX_BOOST=8,1,8192,4 && run ${t}_b12 "$wl" MPCX_BOOST=12,1,8192,4 && run ${t}_b20 "$wl" MPCX_BOOST=20,1,8192,4 && run ${t}_b16l8 "$wl" MPCX_BOOST=16,1,8192,8 && run ${t}_base2 "$wl" A=0 || exit 1
done
