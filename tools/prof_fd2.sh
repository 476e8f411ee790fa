#!/bin/bash
# Dev: kernel trace of the cfg5 factor updates, summarised over the LAST update only.
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_fd
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PS_DEV_ENV=1
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/dev_fd_profile.py > $OUT/run.log 2> $OUT/run.err
cd $GRAFT_REPO_ROOT
tail -2 $OUT/run.log
python3 tools/summarize_trace.py $OUT/trace 0.85
