set -u
run() { local label=$1; shift; line=$(env "$@" 2>/dev/null) || { echo "$label FAILED"; exit 1; }
  echo "$label $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.1f value %.0f" % (d["ms_per_step"], d["value"]))')"; }
B="timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra"
for k in 0 384; do
  run "reachable workers=$k" MPCX_HANDOVER=$k $B --reachable
  run "obstacle workers=$k" MPCX_HANDOVER=$k $B --obstacle
  run "seed 311 workers=$k" MPCX_HANDOVER=$k $B --seed 20250311
  run "seed 315 workers=$k" MPCX_HANDOVER=$k $B --seed 20250315
  run "seed 312 workers=$k" MPCX_HANDOVER=$k $B --seed 20250312
  run "B=262144 workers=$k" MPCX_HANDOVER=$k $B --batch 262144
  run "B=131072 workers=$k" MPCX_HANDOVER=$k $B --batch 131072
done
