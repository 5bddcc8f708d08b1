# Second offer of kept problems (MPCX_HANDOVER=workers,limit,late): late 0 = off, otherwise the age from which a
# kept problem is offered again; three workloads, kernel milliseconds per launch (3 steps each).
for late in 0 64 32 128 256; do
  for w in "" "--reachable" "--obstacle"; do
    MPCX_HANDOVER=320,64,$late timeout -k 10 200 python bench.py $w --steps 3 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('late $late  workload [$w]  %7.1f ms' % d['ms_per_step'])" >> gpurun_out/handover_late.txt || exit 1
  done
done
