#!/bin/bash
# Same-box comparison of several builds of the library on one workload:
#   bash profiles/libs_ab.sh "<bench args>" lib1.so lib2.so ...   (two rounds, alternating)
set -u
ARGS=$1; shift
mkdir -p gpurun_out/libs
for round in 1 2; do
  for lib in "$@"; do
    tag=$(basename $lib .so)_$round
    timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra --lib $lib $ARGS > gpurun_out/libs/$tag.json 2> gpurun_out/libs/$tag.err || exit 1
    python3 -c "import json; d=json.loads(open('gpurun_out/libs/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],1))"
  done
done
