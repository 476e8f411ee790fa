#!/bin/bash
# Dev: kernel timeline of the cfg5 factor updates (8 factors batched; FD_FACTORS=1: one factor, the 8-GPU case)
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_fd
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PS_DEV_ENV=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/dev_fd_profile.py ${FD_FACTORS:-8} > $OUT/run.log 2> $OUT/run.err
cd $GRAFT_REPO_ROOT
tail -2 $OUT/run.log
python3 tools/dump_timeline.py $OUT/trace ${1:-0.9} 2000 > $OUT/timeline.txt
tail -1 $OUT/timeline.txt
find $OUT -name "*.csv" -size +5M -delete
