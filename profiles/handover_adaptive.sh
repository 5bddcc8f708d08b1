set -u
run() { local label=$1; shift; line=$(env "$@" 2>/dev/null) || { echo "$label FAILED"; exit 1; }
  echo "$label $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.1f" % d["ms_per_step"])')"; }
B="timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra"
for cfg in default 0; do
  if [ $cfg = default ]; then E="X=1"; else E="MPCX_HANDOVER=0"; fi
  run "headline $cfg" $E $B
  run "reachable $cfg" $E $B --reachable
  run "reachable seed 311 $cfg" $E $B --reachable --seed 20250311
  run "obstacle $cfg" $E $B --obstacle
  run "B=262144 $cfg" $E $B --batch 262144
done
