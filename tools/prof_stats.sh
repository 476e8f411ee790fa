#!/bin/bash
# Dev: rocprofv3 kernel stats of the ViT-B statistics launch (run on the GPU box).
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_stats
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PS_DEV_ENV=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/dev_stats_vitb.py > $OUT/run.log 2> $OUT/run.err
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
for f in glob.glob("gpurun_out/prof_stats/trace/**/*kernel_stats.csv", recursive=True):
  for r in list(csv.DictReader(open(f)))[:6]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
