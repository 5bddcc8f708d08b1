# Number of worker wavefronts when the caller's priority list is given (every wavefront then has the same share
# of problems that grow old): own order / sorted by a previous solve, milliseconds per launch.
for w in 192 256 320 384 448; do
  MPCX_HANDOVER=$w,64,128 timeout -k 10 200 python bench.py --steps 2 --warmup 1 --extra-steps 3 --cpu-sample 0 --no-host-to-host 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); o=d['callers_order']
        print('workers $w  own order %7.1f ms   sorted %7.1f ms   reachable %7.1f ms' % (d['ms_per_step'], o['ms_per_step'], d['reachable']['ms_per_step']))" >> gpurun_out/handover_order_sweep.txt || exit 1
done
