#!/bin/bash
# Same-box sensitivity of the ageing threshold on the reachable workload (single-problem mode dominates there).
set -u
mkdir -p gpurun_out/reach
run() {
  tag=$1; shift
  env "$@" timeout -k 10 150 python3 bench.py --reachable --steps 2 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra > gpurun_out/reach/$tag.json 2> gpurun_out/reach/$tag.err || return 1
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/reach/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['value'])"
}
run base A=0 && run b32 MPCX_BOOST=32,1,8192,4 && run b64 MPCX_BOOST=64,1,8192,4 && run b128 MPCX_BOOST=128,1,8192,4 && run b24l8 MPCX_BOOST=24,1,8192,8 && run b16l8 MPCX_BOOST=16,1,8192,8 && run off MPCX_BOOST=0,0,0,0
