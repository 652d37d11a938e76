#!/bin/bash
# PMC passes (each its own run, counters only + kernel trace), as the MI355X guide prescribes.  usage: pmc2.sh TAG [passes]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$1
PASSES=${2:-"p1 p2 p3 p4"}
mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-extras ${BENCH_ARGS} > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
for p in $PASSES; do
case $p in
p1) run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM;;
p2) run p2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC;;
p3) run p3 FETCH_SIZE GRBM_GUI_ACTIVE;;
p4) run p4 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum;;
p5) run p5 SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_WAVES SQ_CYCLES;;
esac
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
for p in "$PASSES".split():
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % p):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:48]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in sorted(acc.items()):
            if "cx_" in k or "cxp" in k or "per_read" in k or "mhl" in k or "tile_pass" in k:
                for c, v in sorted(d.items()):
                    print("%-50s %-24s %14.0f  (n=%d)" % (k, c, sum(v)/len(v), len(v)))
PY
cat $OUT/summary.txt
