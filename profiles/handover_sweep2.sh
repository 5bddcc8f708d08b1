set -u
run() { line=$(env "$@" timeout -k 10 150 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra 2>/dev/null) || { echo "$* FAILED"; exit 1; }
  echo "$* $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.1f" % d["ms_per_step"])')"; }
run MPCX_HANDOVER=384
run MPCX_HANDOVER=352
run MPCX_HANDOVER=416
run MPCX_HANDOVER=448 MPCX_BOOST=12,1,8192,4
run MPCX_HANDOVER=512 MPCX_BOOST=12,1,8192,4
run MPCX_HANDOVER=416 MPCX_BOOST=14,1,8192,4
run MPCX_HANDOVER=352 MPCX_BOOST=20,1,8192,4
run MPCX_HANDOVER=384 MPCX_BOOST=16,1,8192,8
run MPCX_HANDOVER=384 MPCX_HOLD=0,0
run MPCX_HANDOVER=576 MPCX_BOOST=10,1,8192,4
