#!/bin/bash
# Dev (round 6): kernel statistics of the ViT-B-shaped eigh recomputes (tools/dev_vitb_eigh.py, solver auto)
# usage: prof_r6_vitb_eigh.sh random|lowrank
KIND=${1:-lowrank}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r6_vitb_eigh_$KIND
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PS_DEV_ENV=1 && export VITB_ONLY=auto
ARG=""; [ "$KIND" = "lowrank" ] && ARG=lowrank
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/dev_vitb_eigh.py $ARG > $OUT/run.log 2> $OUT/run.err
cd $GRAFT_REPO_ROOT
grep -v amdgpu.ids $OUT/run.log | tail -2
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
head -14 $OUT/kernel_stats.csv | cut -c1-170
rm -rf $OUT/trace
