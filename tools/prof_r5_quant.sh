#!/bin/bash
# Dev (round 5): kernel statistics of the quantize / dequantize legs (bench.quant_f3)
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r5_quant
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PS_DEV_ENV=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/dev_r5_quant.py > $OUT/run.log 2> $OUT/run.err
cd $GRAFT_REPO_ROOT
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
grep "quant" $OUT/kernel_stats.csv | cut -c1-150
rm -rf $OUT/trace
