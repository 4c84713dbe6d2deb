#!/bin/bash
# One runner for every GPU-box job of a round (replaces round 2's one-off gpu_r02_*.sh scripts).
#   gpurun -- 'bash tools/gpu.sh <out-tag> <step> [<step> ...]'
# Every step runs under its own time limit; the script stops at the first step that fails (after a GPU
# fault or a timeout nothing more is started).  Output goes to gpurun_out/<out-tag>/.
# Steps:
#   tests            the GPU test suite (pytest -m gpu)
#   tests:<expr>     pytest -m gpu -k <expr>
#   wip:<file>       one not-yet-collected test file (tests/wip_*.py)
#   bench            bench.py with the driver's flags (--steps 20 --warmup 5)
#   bench:<args>     bench.py with other arguments ("," separates arguments)
#   ab:<libs>[:rounds[:shapes]]  this build against other .so files (comma separated paths), interleaved (tools/ab_libs.py)
#   py:<script,args> any script under tools/ ("," separates arguments)
#   trace:<args>     rocprofv3 --kernel-trace --stats around bench.py <args>  -> <out>/trace_*
#   pmc:<ctrs>:<args> one rocprofv3 --pmc pass (ctrs "+"-separated) around bench.py <args>
#   pmcpy:<ctrs>:<script,args>  the same around a tools/ script
set -u
export TMPDIR=/tmp
tag=$1; shift
O=gpurun_out/$tag
mkdir -p "$O"
run() {  # name limit cmd...
    local name=$1 limit=$2; shift 2
    echo "== $name: $*" | cut -c1-300
    timeout -k 10 "$limit" "$@" > "$O/$name.log" 2>&1
    local rc=$?
    tail -n 12 "$O/$name.log" | cut -c1-600
    if [ $rc -ne 0 ]; then echo "== $name FAILED rc=$rc (log: $O/$name.log)"; exit $rc; fi
}
i=0
for step in "$@"; do
    i=$((i + 1))
    kind=${step%%:*}; arg=""; [ "$step" != "$kind" ] && arg=${step#*:}
    case $kind in
    tests)
        if [ -n "$arg" ]; then run "tests_$i" 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "$arg"
        else run "tests_$i" 1100 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread; fi ;;
    wip)   run "wip_$i" 600 python3 -u -m pytest "$arg" -x -q --timeout 300 --timeout-method thread ;;
    bench)
        if [ -n "$arg" ]; then IFS=, read -ra a <<< "$arg"; run "bench_$i" 600 python3 -u bench.py "${a[@]}"
        else run "bench_$i" 600 python3 -u bench.py --steps 20 --warmup 5; fi ;;
    ab)    IFS=: read -ra a <<< "$arg"; run "ab_$i" 900 python3 -u tools/ab_libs.py "${a[@]}" ;;
    py)    IFS=, read -ra a <<< "$arg"; run "py_$i" 900 python3 -u "tools/${a[0]}" "${a[@]:1}" ;;
    trace) IFS=, read -ra a <<< "$arg"
           run "trace_$i" 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_$i" -- python3 bench.py "${a[@]}" ;;
    pmc)   ctrs=${arg%%:*}; rest=${arg#*:}; IFS=, read -ra a <<< "$rest"
           run "pmc_$i" 300 timeout -s KILL 280 rocprofv3 --pmc ${ctrs//+/ } --output-format csv -d "$O/pmc_$i" -- python3 bench.py "${a[@]}" ;;
    pmcpy) ctrs=${arg%%:*}; rest=${arg#*:}; IFS=, read -ra a <<< "$rest"
           run "pmcpy_$i" 300 timeout -s KILL 280 rocprofv3 --pmc ${ctrs//+/ } --output-format csv -d "$O/pmcpy_$i" -- python3 "tools/${a[0]}" "${a[@]:1}" ;;
    *) echo "unknown step $step"; exit 2 ;;
    esac
done
echo "== all steps ok"
