#!/bin/bash
# Fetch-order experiment (one gpurun call): run time of the 1 M batch in the caller's order, sorted
# longest-first by the iteration counts of a first run, and by a geometric guess of difficulty.
set -u
O=gpurun_out/order4.txt; : > $O
for sd in 20250310 20250311; do
  MPCX_BENCH_SAVE_COUNTS=/tmp/c_$sd.npy timeout -k 10 120 python bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra --seed $sd > /tmp/o.json 2>/dev/null || exit 1
  echo "seed $sd plain $(python -c "import json;d=json.load(open('/tmp/o.json'));print(d['ms_per_step'])")" >> $O
  MPCX_DEAL=0 timeout -k 10 120 python bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra --seed $sd > /tmp/o.json 2>/dev/null || exit 1
  echo "seed $sd plain, MPCX_DEAL=0 $(python -c "import json;d=json.load(open('/tmp/o.json'));print(d['ms_per_step'])")" >> $O
  for ord in counts:/tmp/c_$sd.npy reverse; do
    timeout -k 10 120 python bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra --seed $sd --order $ord > /tmp/o.json 2>/dev/null || exit 1
    echo "seed $sd $ord $(python -c "import json;d=json.load(open('/tmp/o.json'));print(d['ms_per_step'])")" >> $O
  done
done
cat $O
