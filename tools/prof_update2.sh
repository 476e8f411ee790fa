#!/bin/bash
# Dev: kernel timeline of update() steps on the ViT-B tree
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_update
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PS_DEV_ENV=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/dev_update_step.py > $OUT/run.log 2> $OUT/run.err
cd $GRAFT_REPO_ROOT
grep -E "update ms|unsynchronised" $OUT/run.log
python3 tools/dump_timeline.py $OUT/trace 0.0 100000 > $OUT/timeline.txt
tail -1 $OUT/timeline.txt
find $OUT -name "*.csv" -size +8M -delete
