#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small files kept under profiles/.

  pmc_summarize.py <tag> <stats_dir> <fetch_dir> <write_dir> [kernel-substring [json-name]]

Writes profiles/<tag>_kernel_stats.csv (copy of *_kernel_stats.csv), profiles/<tag>_pmc_hbm.csv
(mean FETCH_SIZE / WRITE_SIZE per kernel, in the counters' KB unit) and, for the kernel whose name
contains the substring (default fir_), profiles/hbm_traffic.json with the per-launch HBM bytes:
FETCH_SIZE doubled (gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE, both x1024.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(d, pat):
    hits = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True), key=os.path.getmtime)
    if not hits:
        raise SystemExit("no %s under %s" % (pat, d))
    return hits[-1]                      # the newest run (gpurun_out/ accumulates)


def counter_means(d):
    acc = defaultdict(lambda: [0.0, 0])
    with open(find(d, "*counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            k = (row["Counter_Name"], row["Kernel_Name"])
            acc[k][0] += float(row["Counter_Value"])
            acc[k][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    tag, stats, fetch, write = sys.argv[1:5]
    sub = sys.argv[5] if len(sys.argv) > 5 else "fir_"
    jname = sys.argv[6] if len(sys.argv) > 6 else "hbm_traffic.json"
    prof = os.path.join(ROOT, "profiles")
    shutil.copy(find(stats, "*kernel_stats.csv"), os.path.join(prof, tag + "_kernel_stats.csv"))
    rows = {}
    rows.update(counter_means(fetch))
    rows.update(counter_means(write))
    with open(os.path.join(prof, tag + "_pmc_hbm.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["counter", "dispatches", "mean_value_KB", "kernel"])
        for (c, k), (m, n) in sorted(rows.items()):
            w.writerow([c, n, "%.3f" % m, k])
    fk = [k for (c, k) in rows if c == "FETCH_SIZE" and sub in k]
    if fk:
        # the kernel with the most traffic among the matches (the bench's hot kernel)
        k = max(fk, key=lambda kk: rows[("FETCH_SIZE", kk)][0])
        fb = rows[("FETCH_SIZE", k)][0] * 1024 * 2
        wb = rows[("WRITE_SIZE", k)][0] * 1024
        avg_ns = None
        with open(os.path.join(prof, tag + "_kernel_stats.csv")) as f:
            for row in csv.DictReader(f):
                if row["Name"] == k:
                    avg_ns = float(row["AverageNs"])
        import subprocess
        try:
            head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], text=True).strip()
            dirty = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "sdr_amd", "bench.py"], text=True).strip())
        except Exception:
            head, dirty = None, None
        json.dump({"fir_bytes_per_launch": int(fb + wb), "commit": head, "sources_modified_since_commit": dirty, "fetch_bytes_corrected_x2": int(fb), "write_bytes": int(wb),
                   "kernel": k, "rocprof_avg_kernel_ns": avg_ns, "profile_tag": tag,
                   "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on python3 bench.py, mean over "
                             "the kernel's dispatches; FETCH_SIZE doubled per the guide's gfx950 correction; KB -> bytes x1024"},
                  open(os.path.join(prof, jname), "w"), indent=1)
        print("traffic", int(fb + wb), "avg_ns", avg_ns)


if __name__ == "__main__":
    main()
