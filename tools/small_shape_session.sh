#!/bin/bash
# Development aid: configs[1] (ne30 x 72 x 1) timed in each form, with kernel traces of the default
# (two class passes) and of the single sweep forced onto it.   gpurun -- 'bash tools/small_shape_session.sh'
set -eo pipefail
REPO=$PWD
O=$REPO/gpurun_out/small
rm -rf $O; mkdir -p $O
for F in auto class-sums single-sweep; do python tools/run_shape.py ne30x72x1 f64 $F 400 >> $O/times.log 2>&1; done
cd /tmp; export TMPDIR=/tmp
for F in auto single-sweep; do
rocprofv3 --kernel-trace --output-format csv -d $O/kt_$F -- python3 $REPO/tools/run_shape.py ne30x72x1 f64 $F 50 >> $O/prof.log 2>&1
done
cd $REPO
for F in auto single-sweep; do python tools/kernel_table.py $(find $O/kt_$F -name '*kernel_trace.csv') > $O/table_$F.txt 2>&1 || true; done
