for sd in 20250311 20250312 20250313 20250314 20250315 20250316 20250317; do
  echo "seed $sd" >> gpurun_out/seeds.txt
  timeout -k 10 120 python bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra --seed $sd 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['value'], d['config']['converged_fraction'])" >> gpurun_out/seeds.txt || exit 1
done
