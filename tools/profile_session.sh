#!/bin/bash
# One measurement session on the GPU box (run from the repo root through gpurun):
#   tools/profile_session.sh r04
# writes bench JSON lines, rocprofv3 kernel traces and PMC passes under gpurun_out/;
# `python tools/collect_profiles.py r04` then copies the summaries into profiles/ (in the container).
# Every rocprofv3 command has the program itself after `--` (python3 script ...), counters in passes of their own.
set -eo pipefail
R=${1:-r04}
REPO=$PWD
O=$REPO/gpurun_out/session_$R
rm -rf "$O"
mkdir -p "$O"
python bench.py 2> "$O/bench.err" | tail -n 1 > "$O/bench.json"
echo "bench default done"
python bench.py --no-cpu-baseline --class-sums --also= 2>> "$O/bench.err" | tail -n 1 > "$O/bench_classsums.json"
python bench.py --no-cpu-baseline --two-pass --also= 2>> "$O/bench.err" | tail -n 1 > "$O/bench_twopass.json"
python bench.py --no-cpu-baseline --no-classes --also= 2>> "$O/bench.err" | tail -n 1 > "$O/bench_paired.json"
python bench.py --no-cpu-baseline --exact-mirror --also= 2>> "$O/bench.err" | tail -n 1 > "$O/bench_exact_mirror.json"
echo "bench variants done"
python tools/tracers_bench.py 120x72x30 f64 2 > "$O/tracers.log" 2>&1
python tools/tracers_bench.py 120x72x30 f32 2 >> "$O/tracers.log" 2>&1
python tools/tracers_bench.py 30x72x91 f64 3 >> "$O/tracers.log" 2>&1
python tools/graph_probe.py ne30x72x1 > "$O/graph_probe.log" 2>&1
echo "tracer / graph probes done"
cd /tmp
export TMPDIR=/tmp
A="--steps 20 --warmup 3 --no-cpu-baseline --also="
rocprofv3 --kernel-trace --output-format csv -d "$O/kt_bench" -- python3 "$REPO/bench.py" $A > "$O/prof.log" 2>&1
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 "$REPO/bench.py" $A >> "$O/prof.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 "$REPO/bench.py" $A >> "$O/prof.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d "$O/pmc_sq" -- python3 "$REPO/bench.py" $A >> "$O/prof.log" 2>&1
echo "pmc passes (headline) done"
# the fp32 shapes and one rank's block of configs[2]: kernel trace + FETCH / WRITE passes of tools/run_shape.py
for S in "ne240x128x1 f32" "ne120x72x30 f32" "ne30x72x91 f64"; do
  T=$(echo $S | tr ' ' '_')
  rocprofv3 --kernel-trace --output-format csv -d "$O/kt_$T" -- python3 "$REPO/tools/run_shape.py" $S auto 10 >> "$O/prof.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmcf_$T" -- python3 "$REPO/tools/run_shape.py" $S auto 5 >> "$O/prof.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmcw_$T" -- python3 "$REPO/tools/run_shape.py" $S auto 5 >> "$O/prof.log" 2>&1
done
echo "other shapes done"
cd "$REPO"
# the lab run whose round-3 log ended in a fault of the HARNESS (output buffers sized for 16 splits): same shape, once
timeout -k 10 120 tools/sweep_lab_os 3110402 128 3 f32 > "$O/lab_d128_f32.log" 2>&1 || echo "lab exit $?" >> "$O/lab_d128_f32.log"
echo "lab done"
