#!/bin/bash
# Memory-pipeline counters of the dominant kernel (each pass its own run, counters + kernel trace only).  usage: pmc_mem.sh TAG [passes]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcm_$1
PASSES=${2:-"q1 q2 q3 q4 q5"}
mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-extras --no-selfcheck ${BENCH_ARGS} > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
for p in $PASSES; do
case $p in
q1) run q1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE;;
q2) run q2 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum;;
q3) run q3 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_LFIFO_FULL_sum;;
q4) run q4 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_32B_sum;;
q5) run q5 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES;;
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
            if "cx_tiles" in k or "cxp" in k or "mhl_fused" in k or "row_loads" in k or "stream" in k:
                for c, v in sorted(d.items()):
                    print("%-50s %-44s %16.0f  (n=%d)" % (k, c, sum(v)/len(v), len(v)))
PY
cat $OUT/summary.txt
