#!/bin/bash
# lab: the single sweep (row map, tools/sweep_lab.hip "osr") with parts of its work left out
# TEMX_OS_SKIP bits: 1 projection chunks, 2 reference MFMAs, 4 accumulation, 8 the two barriers of a group (results wrong)
#   tools/lab_skip_run.sh build     in the container: one lab binary per mask under tools/ab/ (git-ignored, travels with gpurun)
#   tools/lab_skip_run.sh           on the GPU box: time them (fp32 and fp64 inputs, ne120 x 72 x 30 with the real classes:
#                                   python tools/dump_classes.py 120 tools/ab/cls120_split.bin 1 size)
cd "$(dirname "$0")/.."
MASKS="1 2 4 5 8 15"
if [ "$1" = build ]; then
  mkdir -p tools/ab
  for k in $MASKS; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DLAB_OS -DLAB_OS_FEW -DTEMX_OS_SKIP=$k -o tools/ab/lab_os_skip$k tools/sweep_lab.hip &
  done
  wait
  exit 0
fi
for dt in f32 f64; do
  for k in $MASKS; do
    echo "== $dt skip=$k"
    LAB_CLASSES=tools/ab/cls120_split.bin timeout -k 10 100 tools/ab/lab_os_skip$k 777602 2160 6 $dt "osr" 2>&1 | grep "^osr.*PD=2" | cut -c80-200
  done
done
