#!/bin/bash
# resource usage of the sweep kernels of tools/sweep_lab.hip (or any .hip given): name, VGPR, AGPR, spill, LDS, occupancy
SRC=${1:-tools/sweep_lab.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage $LAB_FLAGS -c -o /dev/null "$SRC" 2>&1 |
 awk '/Function Name:/ {n=$0; sub(/.*Function Name: /,"",n); sub(/ \[.*/,"",n)}
      /VGPRs:/ && !/Spill/ {v=$0; sub(/.*VGPRs: /,"",v); sub(/ .*/,"",v)}
      /AGPRs:/ {a=$0; sub(/.*AGPRs: /,"",a); sub(/ .*/,"",a)}
      /Occupancy/ {o=$0; sub(/.*: /,"",o); sub(/ .*/,"",o)}
      /VGPRs Spill:/ {s=$0; sub(/.*Spill: /,"",s); sub(/ .*/,"",s)}
      /LDS Size/ {l=$0; sub(/.*: /,"",l); sub(/ .*/,"",l); print n, v, a, s, l, o}' |
 while read n v a s l o; do echo "$(echo $n | c++filt | sed 's/(.*//; s/void //') VGPR=$v AGPR=$a spill=$s LDS=$l occ=$o"; done | grep -E "${2:-sweep}"
