#!/bin/bash
# lab: the single sweep (row map) with parts of its work left out
# TEMX_OS_SKIP bits: 1 projection chunks, 2 reference MFMAs, 4 accumulation, 8 the two barriers of a group (results wrong)
cd "$(dirname "$0")/.."
for dt in f32 f64; do
  for k in 1 2 4 5 8 15; do
    echo "== $dt skip=$k"
    LAB_CLASSES=tools/ab/cls120_split.bin timeout -k 10 100 tools/ab/lab_os_skip$k 777602 2160 6 $dt "osr" 2>&1 | grep "^osr.*PD=2" | cut -c80-200
  done
done
