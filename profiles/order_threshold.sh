for t in 0 30 100 300 1000; do
  timeout -k 10 200 python bench.py --steps 1 --warmup 1 --extra-steps 2 --cpu-sample 0 --no-host-to-host --order-threshold $t 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); o=d['callers_order']
        print('threshold $t  own order %7.1f ms   longest first %7.1f ms   (%d problems moved, same results: %s)' % (d['ms_per_step'], o['ms_per_step'], o['moved_to_the_front'], o['same_results']))" >> gpurun_out/order_threshold.txt || exit 1
done
