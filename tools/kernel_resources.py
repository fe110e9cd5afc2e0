#!/usr/bin/env python3
"""Print VGPR/AGPR/spill/LDS/occupancy per kernel of libtemx (hipcc -Rpass-analysis)."""
import re, subprocess, sys
src = "pytemdiags_amd/csrc/temx.hip"
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                      "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", src],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"^void ", "", re.sub(r"\(.*", "", cur))
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
filt = sys.argv[1] if len(sys.argv) > 1 else ""
spill = sum(1 for v in rows.values() if v.get("VGPRs Spill") or v.get("SGPRs Spill"))
print("# python tools/kernel_resources.py (hipcc -Rpass-analysis=kernel-resource-usage on temx.hip): every kernel of libtemx.so;")
print("# %d kernels, %d of them spill" % (len(rows), spill))
print("%-105s %5s %5s %6s %11s %4s" % ("kernel", "VGPR", "AGPR", "spill", "LDS(static)", "occ"))
for k in sorted(rows):
    v = rows[k]
    if filt in k:
        print("%-105s %5s %5s %6s %11s %4s" % (k[:105], v.get("VGPRs"), v.get("AGPRs"), v.get("VGPRs Spill"),
              v.get("LDS Size"), v.get("Occupancy")))
