for a in 12 14 16 20; do
  MPCX_BOOST=$a,1,8192,4 timeout -k 10 200 python bench.py --steps 2 --warmup 1 --extra-steps 3 --cpu-sample 0 --no-host-to-host 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); o=d['callers_order']
        print('ageing threshold $a  own order %7.1f ms   sorted %7.1f ms' % (d['ms_per_step'], o['ms_per_step']))" >> gpurun_out/boost_order.txt || exit 1
done
