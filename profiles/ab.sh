#!/bin/bash
# Same-box A/B of two builds of the library: bash profiles/ab.sh <other.so> [bench args...]
# alternates base / new / base / new so that drift of the box shows.
set -u
OTHER=$1; shift
mkdir -p gpurun_out/ab
run() {
  tag=$1; shift
  timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-host-to-host --no-extra "$@" > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err || return 1
  python3 -c "import json; d=json.loads(open('gpurun_out/ab/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],1), round(d['value']), d['config']['ip_iterations_total'])"
}
run base1 --lib "$OTHER" "$@" && run new1 "$@" && run base2 --lib "$OTHER" "$@" && run new2 "$@"
