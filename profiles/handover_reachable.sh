set -u
run() { local label=$1; shift; line=$(env "$@" 2>/dev/null) || { echo "$label FAILED"; exit 1; }
  echo "$label $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.1f" % d["ms_per_step"])')"; }
B="timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra"
for cfg in 0 320,64 512,64 640,64 768,64 896,64 640,1024; do
  run "reachable $cfg" MPCX_HANDOVER=$cfg $B --reachable
  run "reachable seed 311 $cfg" MPCX_HANDOVER=$cfg $B --reachable --seed 20250311
done
