#!/bin/bash
# Dev (round 6): kernel statistics of the eigh fast path on rank-deficient vs Wishart statistics
# usage: prof_r6_td.sh lowrank|wishart
KIND=${1:-lowrank}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r6_td_$KIND
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PS_DEV_ENV=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/dev_r6_td_prof.py $KIND > $OUT/run.log 2> $OUT/run.err
cd $GRAFT_REPO_ROOT
grep -v amdgpu.ids $OUT/run.log | tail -3
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
head -16 $OUT/kernel_stats.csv | cut -c1-160
rm -rf $OUT/trace
