#!/bin/bash
# Dev: rocprofv3 kernel stats of update() steps on the ViT-B tree (run on the GPU box).
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_update
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PS_DEV_ENV=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/dev_update_step.py > $OUT/run.log 2> $OUT/run.err
cd $GRAFT_REPO_ROOT
grep -E "update ms|unsynchronised" $OUT/run.log
python - <<PY
import csv, glob
for f in glob.glob("gpurun_out/prof_update/trace/**/*kernel_stats.csv", recursive=True):
  for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
