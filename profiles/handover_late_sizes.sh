for b in 131072 262144 524288; do
  for late in 0 128; do
    MPCX_HANDOVER=320,64,$late timeout -k 10 200 python bench.py --batch $b --steps 4 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('batch $b late $late  %7.1f ms' % d['ms_per_step'])" || exit 1
  done
done
