#!/bin/bash
# Further counter passes, one rocprofv3 run each.  Usage, on the GPU box:
#   bash profiles/pmc_extra.sh <outdir> <batch>
#  * FP64 instruction mix of the fused SQP kernel (measured flops against the algorithmic count)
#  * LDS activity and bank conflicts of the same kernel
#  * FETCH_SIZE / WRITE_SIZE of the memory-pattern micro-benchmark (profiles/micro/stream_rw: the
#    solver's access shape, 8 bytes per lane in 512-byte segments, with a known byte count per
#    launch), which gives the factor between these counters and real bytes for this pattern
set -u
OUT=${1:-gpurun_out/pmc_extra}
BATCH=${2:-1000000}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bench() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- \
    python3 bench.py --batch "$BATCH" --steps 1 --warmup 0 --cpu-sample 0 --no-host-to-host --no-extra > "$OUT/$name.log" 2>&1 || return 1
}
micro() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- \
    profiles/micro/stream_rw > "$OUT/$name.log" 2>&1 || return 1
}
bench f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT &&
bench lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_WAVES SQ_WAVE_CYCLES &&
micro cal_fetch FETCH_SIZE &&
micro cal_write WRITE_SIZE
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float)
cal = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        kn = row["Kernel_Name"]
        if "mpcx_sqp_kernel" in kn:
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
        elif "k_rw" in kn:
            cal[kn][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    fh.write("fused SQP kernel, one launch:\n")
    for k in sorted(tot):
        fh.write(f"  {k:28s} {tot[k]:.6g}\n")
    if all(k in tot for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64")):
        # wavefront-level instruction counts; every FP64 instruction of the kernel is 64 lanes wide
        fl = 64.0 * (tot["SQ_INSTS_VALU_ADD_F64"] + tot["SQ_INSTS_VALU_MUL_F64"] + 2.0 * tot["SQ_INSTS_VALU_FMA_F64"]
                     + tot.get("SQ_INSTS_VALU_TRANS_F64", 0.0))
        fh.write(f"  executed FP64 flops if every lane were active (64 x (add + mul + 2 fma + trans)): {fl:.6g}\n")
    fh.write("memory-pattern micro-benchmark (launches of 2 and 20 passes; 1024 wavefronts x 51 stages x 512 B per field):\n")
    for kn in sorted(cal):
        for cn, vals in cal[kn].items():
            fh.write(f"  {kn[:60]:60s} {cn:12s} " + " ".join(f"{v:.6g}" for v in vals) + "\n")
print(open(out + "/summary.txt").read())
PY
