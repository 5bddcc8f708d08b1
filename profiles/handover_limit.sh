set -u
run() { local label=$1; shift; line=$(env "$@" 2>/dev/null) || { echo "$label FAILED"; exit 1; }
  echo "$label $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.1f value %.0f" % (d["ms_per_step"], d["value"]))')"; }
B="timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra"
for cfg in 384,16 384,64 384,256 384,2048 320,64 256,64; do
  run "headline $cfg" MPCX_HANDOVER=$cfg $B
  run "reachable $cfg" MPCX_HANDOVER=$cfg $B --reachable
  run "obstacle $cfg" MPCX_HANDOVER=$cfg $B --obstacle
done
