#!/usr/bin/env python3
"""Condense the rocprofv3 --pmc passes of tools/pmc_fir.sh (gpurun_out/pmc_<tag>/<pass>/**/*counter_collection.csv)
into one tracked CSV: profiles/<name>.csv with the mean counter value per dispatch of the kernels whose
name contains the given substring, plus derived per-launch figures (SQ_* cycle counters are in quad-cycles).

  sq_to_profile.py <pmc_dir> <profile-name> <kernel-substring> [note]
"""
import csv
import glob
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    d, name, sub = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    acc = defaultdict(lambda: [0.0, 0])
    kernels = set()
    newest = {}                                     # one file per pass directory: the latest run (gpurun_out/ accumulates)
    for f in glob.glob(os.path.join(d, "*", "**", "*counter_collection.csv"), recursive=True):
        key = os.path.relpath(f, d).split(os.sep)[0]
        if key not in newest or os.path.getmtime(f) > os.path.getmtime(newest[key]):
            newest[key] = f
    for f in sorted(newest.values()):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                kernels.add(r["Kernel_Name"])
                a = acc[r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    if not acc:
        raise SystemExit("no dispatch of a kernel matching %r under %s" % (sub, d))
    m = {k: v[0] / v[1] for k, v in acc.items()}
    out = os.path.join(ROOT, "profiles", name + ".csv")
    with open(out, "w") as f:
        w = csv.writer(f)
        w.writerow(["# kernels", " | ".join(sorted(kernels))])
        if note:
            w.writerow(["# run", note])
        w.writerow(["counter", "mean_per_dispatch", "dispatches"])
        for k in sorted(m):
            w.writerow([k, "%.0f" % m[k], acc[k][1]])
        wc = m.get("SQ_WAVE_CYCLES")
        w.writerow(["# derived (shares of SQ_WAVE_CYCLES, all in quad-cycles)"])
        if wc:
            for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_FLAT", "SQ_ACTIVE_INST_SCA",
                      "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
                if k in m:
                    w.writerow([k + "/SQ_WAVE_CYCLES", "%.4f" % (m[k] / wc), ""])
        if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
            w.writerow(["SQ_LDS_BANK_CONFLICT/SQ_LDS_IDX_ACTIVE", "%.4f" % (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]), ""])
        if "SQ_INSTS_VALU" in m and m.get("SQ_WAVES"):
            w.writerow(["VALU instructions per wave", "%.0f" % (m["SQ_INSTS_VALU"] / m["SQ_WAVES"]), ""])
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("SQ_BUSY_CU_CYCLES"):
            w.writerow(["SQ_VALU_MFMA_BUSY_CYCLES/SQ_BUSY_CU_CYCLES (matrix pipe busy; cycles / cycles)",
                        "%.4f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_BUSY_CU_CYCLES"]), ""])
        if "GRBM_GUI_ACTIVE" in m:
            w.writerow(["GRBM_GUI_ACTIVE/8 (shader cycles per dispatch, mean over the 8 XCDs)", "%.0f" % (m["GRBM_GUI_ACTIVE"] / 8), ""])
    print(out)


if __name__ == "__main__":
    main()
