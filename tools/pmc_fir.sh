#!/bin/bash
# SQ counter passes for one FIR configuration: tools/pmc_fir.sh <tag> log2n T D mode R NT pipe
set -u
export TMPDIR=/tmp
tag=$1; shift
OUT=$PWD/gpurun_out/pmc_$tag
mkdir -p $OUT
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 ${RUNNER:-tools/run_one.py} $ARGS > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; return 1; }
}
ARGS="$*"
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY || exit 1
pass b SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA || exit 1
pass c SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT || true
pass d GRBM_GUI_ACTIVE GRBM_COUNT || true
pass e SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES || true
pass f SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA || true
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$OUT/*/*/*_counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if ("fir_" in r["Kernel_Name"] or "fft" in r["Kernel_Name"]) and "synth" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
