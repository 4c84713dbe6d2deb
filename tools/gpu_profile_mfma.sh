#!/bin/bash
# Long-filter evaluation under rocprofv3: kernel-trace stats of tools/eval_mfma.py (all four kernels,
# interleaved) and the HBM traffic counters of the int8 matrix kernel (separate --pmc passes).
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_mfma
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/eval_mfma.py 28 > $OUT/eval.log 2>&1 || { echo "stats pass failed"; tail -5 $OUT/eval.log; exit 1; }
grep "^{" $OUT/eval.log > $OUT/eval.jsonl
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/run_one.py 28 511 10 1 0 0 5 3 > $OUT/fetch.log 2>&1 || { echo "fetch pass failed"; tail -5 $OUT/fetch.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/run_one.py 28 511 10 1 0 0 5 3 > $OUT/write.log 2>&1 || { echo "write pass failed"; tail -5 $OUT/write.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 tools/run_one.py 28 511 10 1 0 0 5 3 > $OUT/sq.log 2>&1 || { echo "sq pass failed"; tail -5 $OUT/sq.log; }
cat $OUT/eval.jsonl
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$OUT/pmc_sq/*/*_counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "fir_mfma_i8" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
