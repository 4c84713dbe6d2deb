#!/usr/bin/env python3
"""Split the dispatches of the headline kernel in a rocprofv3 --kernel-trace of bench.py into the bench's
phases (pre-warm batches, --warmup launches, the K timed launches, the library timer's launches after
the parity check) and write the tracked summary profiles/<tag>_trace.json.

  trace_phases.py <stats_dir> <tag> <bench_log> [steps warmup]"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stats, tag, log = sys.argv[1:4]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
warmup = int(sys.argv[5]) if len(sys.argv) > 5 else 5
f = sorted(glob.glob(os.path.join(stats, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if "fir_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
line = [json.loads(l) for l in open(log) if l.startswith("{")][-1]
pre = line["prewarm"]["launches"]
timed = dur[pre + warmup: pre + warmup + steps]
after = dur[pre + warmup + steps:]
bytes_per_launch = line["roofline"]["algorithmic_bytes_per_sample"] * line["config"]["samples_per_gpu"]
out = {
    "command": f"rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps {steps} --warmup {warmup}",
    "kernel": rows[0]["Kernel_Name"].split("(")[1].split("::")[-1] if "::" in rows[0]["Kernel_Name"] else rows[0]["Kernel_Name"],
    "dispatches": len(dur),
    "phases_avg_ns": {
        f"prewarm_{pre}": sum(dur[:pre]) / max(pre, 1),
        f"warmup_{warmup}": sum(dur[pre: pre + warmup]) / max(warmup, 1),
        f"timed_{steps}": sum(timed) / len(timed),
        f"library_timer_{len(after)}_after_idle": sum(after) / max(len(after), 1),
    },
    "first_12_ns": dur[:12],
    f"timed_{steps}_ns": timed,
    f"frac_of_8TBps_timed_{steps}": bytes_per_launch / (sum(timed) / len(timed) * 1e-9) / 8e12,
    "bench_line_kernel_ms_same_run": line["roofline"]["kernel_ms"],
}
json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_trace.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("kernel", "dispatches", "phases_avg_ns", f"frac_of_8TBps_timed_{steps}", "bench_line_kernel_ms_same_run")}, indent=1))
