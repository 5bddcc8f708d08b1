#!/bin/bash
# Headline batch under different diagnostic knobs.  Usage on the GPU box:
#   [BENCH_ARGS='--seed 7'] bash profiles/knob_sweep.sh "MPCX_GROUP=0" "MPCX_GROUP=4 MPCX_BOOST=12,1,8192,4" ...
# One line per setting in gpurun_out/sweep.txt: ms per step, converged OCP/s.
for cfg in "$@"; do
  echo "$cfg" >> gpurun_out/sweep.txt
  env $cfg timeout -k 10 120 python bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra $BENCH_ARGS 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['value'])" >> gpurun_out/sweep.txt || exit 1
done
