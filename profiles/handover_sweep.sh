#!/bin/bash
# hand-over of old problems to worker wavefronts: how many wavefronts are workers from the start
set -u
for k in 0 128 256 320 384 448 512; do
  line=$(MPCX_HANDOVER=$k timeout -k 10 150 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra "$@" 2>/dev/null) || { echo "workers $k FAILED"; exit 1; }
  echo "workers $k $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.1f value %.0f" % (d["ms_per_step"], d["value"]))')"
done
