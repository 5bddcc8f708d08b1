set -u
run() { local label=$1; shift; line=$(env "$@" 2>/dev/null) || { echo "$label FAILED"; exit 1; }
  echo "$label $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.1f" % d["ms_per_step"])')"; }
B="timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra"
for cfg in 0 128,64 192,64 256,64 256,32 256,128; do
  run "headline $cfg" MPCX_HANDOVER=$cfg $B
  run "seed312 $cfg" MPCX_HANDOVER=$cfg $B --seed 20250312
  run "B=262144 $cfg" MPCX_HANDOVER=$cfg $B --batch 262144
  run "B=131072 $cfg" MPCX_HANDOVER=$cfg $B --batch 131072
done
