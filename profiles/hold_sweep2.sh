#!/bin/bash
set -u
mkdir -p gpurun_out/hold
run() {
  tag=$1; wl=$2; shift; shift
  env "$@" timeout -k 10 150 python3 bench.py $wl --steps 3 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra > gpurun_out/hold/$tag.json 2> gpurun_out/hold/$tag.err || return 1
  python3 -c "import json; d=json.loads(open('gpurun_out/hold/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],1))"
}
for wl in "" "--obstacle"; do
  t=h; [ "$wl" = "--obstacle" ] && t=o
  run ${t}_off "$wl" A=0 && run ${t}_1e-4_24 "$wl" MPCX_HOLD=0.0001,24 && run ${t}_1e-3_24 "$wl" MPCX_HOLD=0.001,24 && run ${t}_1e-3_48 "$wl" MPCX_HOLD=0.001,48 && run ${t}_1e-2_48 "$wl" MPCX_HOLD=0.01,48 && run ${t}_1e-1_24 "$wl" MPCX_HOLD=0.1,24 && run ${t}_1e-3_24b "$wl" MPCX_HOLD=0.001,24 && run ${t}_off2 "$wl" A=0 || exit 1
done
