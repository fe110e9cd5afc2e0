#!/bin/bash
# One measurement session on the GPU box (run from the repo root through gpurun):
#   tools/profile_session.sh r01
# writes bench JSON lines, the rocprofv3 kernel-trace summary and three PMC passes under gpurun_out/;
# tools/collect_profiles.py stats_<round> pmc_<round> <round>  then copies the summaries into profiles/.
set -eo pipefail
R=${1:-r02}
REPO=$PWD
O=$REPO/gpurun_out
mkdir -p "$O"
python bench.py 2> "$O/bench_$R.err" | tail -n 1 > "$O/bench_$R.json"
echo "bench default done"
python bench.py --no-cpu-baseline --class-sums --also= 2>> "$O/bench_$R.err" | tail -n 1 > "$O/bench_${R}_classsums.json"
python bench.py --no-cpu-baseline --two-pass 2>> "$O/bench_$R.err" | tail -n 1 > "$O/bench_${R}_twopass.json"
python bench.py --no-cpu-baseline --no-classes 2>> "$O/bench_$R.err" | tail -n 1 > "$O/bench_${R}_paired.json"
python bench.py --no-cpu-baseline --no-symmetry 2>> "$O/bench_$R.err" | tail -n 1 > "$O/bench_${R}_generic.json"
echo "bench variants done"
cd /tmp
export TMPDIR=/tmp
A="--steps 5 --warmup 1 --no-cpu-baseline --also="
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$R" -- python3 "$REPO/bench.py" $A > "$O/prof_$R.log" 2>&1
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_${R}_fetch" -- python3 "$REPO/bench.py" $A >> "$O/prof_$R.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_${R}_write" -- python3 "$REPO/bench.py" $A >> "$O/prof_$R.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d "$O/pmc_${R}_sq" -- python3 "$REPO/bench.py" $A >> "$O/prof_$R.log" 2>&1
echo "pmc passes done"
