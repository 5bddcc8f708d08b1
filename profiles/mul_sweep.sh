#!/bin/bash
# Same-box sweep of the growth of the ageing quota (MPCX_BOOST=min,div,cap,lanes,mul) on the headline batch.
set -u
mkdir -p gpurun_out/knob
run() {
  tag=$1; shift
  env "$@" timeout -k 10 150 python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra > gpurun_out/knob/$tag.json 2> gpurun_out/knob/$tag.err || return 1
  python3 -c "import json; d=json.loads(open('gpurun_out/knob/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],1))"
}
for round in 1 2 3; do
  run m1_$round MPCX_BOOST=16,1,8192,4,1 && run m2_$round MPCX_BOOST=16,1,8192,4,2 && run m3_$round MPCX_BOOST=16,1,8192,4,3 || exit 1
done
