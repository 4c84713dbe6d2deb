#!/usr/bin/env python3
"""Register / scratch use of every kernel of one translation unit as hipcc reports it
(-Rpass-analysis=kernel-resource-usage), with the per-file flags of sdr_amd/capi.py.
usage: tools/kernel_resources.py sdr_amd/csrc/fir_mfma_i8.hip [filter-substring] [-- extra hipcc flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdr_amd import capi

def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--"); extra = args[i + 1:]; args = args[:i]
    src = args[0]; filt = args[1] if len(args) > 1 else ""
    name = os.path.basename(src)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + capi.INCLUDE, "-I" + os.path.join(ROOT, "sdr_amd", "csrc")]
    cmd += capi.EXTRA_FLAGS.get(name, []) + extra + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}; rows.append(cur); continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
        if " error" in line:
            print(line)
    for r in rows:
        dem = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("(anonymous namespace)::", "").replace("void ", "")
        dem = re.sub(r"\(.*", "", dem)
        if filt and filt not in dem:
            continue
        print("%-52s vgpr %4s agpr %3s sgpr %3s scratch %5s spillV %3s spillS %3s occ %s" % (
            dem[:52], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"),
            r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("Occupancy")))

if __name__ == "__main__":
    main()
