# The seed sweep once more, each batch solved a second time with mpcx_batch.order = longest_first(iteration
# counts of the first solve): library's own order against the caller's, milliseconds per launch.
for sd in 20250311 20250312 20250313 20250314 20250315 20250316 20250317; do
  timeout -k 10 200 python bench.py --steps 1 --warmup 1 --extra-steps 2 --cpu-sample 0 --no-host-to-host --seed $sd 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); o=d['callers_order']
        print('seed $sd  own order %7.1f ms   longest first %7.1f ms   (%d problems moved, same results: %s)' % (d['ms_per_step'], o['ms_per_step'], o['moved_to_the_front'], o['same_results']))" >> gpurun_out/seeds_order.txt || exit 1
done
