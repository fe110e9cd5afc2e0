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
    m = re.search(r"remark: (?:Function )?Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur)
        rows[cur] = {}
        continue
    m = re.search(r"remark: \s*([A-Za-z ]+?)(?: \[bytes/\w+\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
filt = sys.argv[1] if len(sys.argv) > 1 else ""
print("%-70s %5s %5s %6s %6s %5s" % ("kernel", "VGPR", "AGPR", "spill", "scratch", "occ"))
for k, v in rows.items():
    if filt in k:
        print("%-70s %5s %5s %6s %6s %5s" % (k[:70], v.get("VGPRs"), v.get("AGPRs"), v.get("VGPRs Spill"),
              v.get("ScratchSize"), v.get("Occupancy")))
