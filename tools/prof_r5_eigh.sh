#!/bin/bash
# Dev (round 5): kernel statistics of cfg3 eigh root calls (64 x 2048^2) on the tridiagonalisation path
# usage: prof_r5_eigh.sh [tag] [ENV=VALUE ...]
TAG=${1:-default}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r5_eigh_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PS_DEV_ENV=1
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/dev_eigh_one.py > $OUT/run.log 2> $OUT/run.err
cd $GRAFT_REPO_ROOT
tail -2 $OUT/run.log
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
head -${LINES_SHOWN:-14} $OUT/kernel_stats.csv | cut -c1-150
rm -rf $OUT/trace
