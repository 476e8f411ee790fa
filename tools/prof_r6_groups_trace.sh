#!/bin/bash
# Dev (round 6): do the stage launches of two stream groups overlap?  kernel trace of one cfg2 call with 2 groups
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r6_groups
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PS_DEV_ENV=1 && export PS_NEWTON_GROUPS=${1:-2}
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/dev_stage_launch.py cfg2_256x512_p4 > $OUT/run.log 2> $OUT/run.err
cd $GRAFT_REPO_ROOT
python3 tools/summarize_trace.py $OUT/trace 0.6 | head -8
rm -rf $OUT/trace
