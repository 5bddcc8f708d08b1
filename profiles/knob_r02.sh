#!/bin/bash
# Same-box sweep of the ageing knobs on the headline batch.
set -u
mkdir -p gpurun_out/knob
run() {
  tag=$1; shift
  env "$@" timeout -k 10 150 python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra > gpurun_out/knob/$tag.json 2> gpurun_out/knob/$tag.err || return 1
  python3 -c "import json; d=json.loads(open('gpurun_out/knob/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],1))"
}
run base A=0 && run b12 MPCX_BOOST=12,1,8192,4 && run b20 MPCX_BOOST=20,1,8192,4 && run b16l8 MPCX_BOOST=16,1,8192,8 && run b16l2 MPCX_BOOST=16,1,8192,2 && run b16d2 MPCX_BOOST=16,2,8192,4 && run slots12 MPCX_SLOTS=12 && run svc8 MPCX_SVC_MIN=8 && run svc2 MPCX_SVC_MIN=2 && run base2 A=0
