#!/bin/bash
# Dev: per-dispatch durations of the statistics launches of the ViT-B tree (matrix part, vector part)
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_stats2
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/dev_stats_vitb.py > $OUT/run.log 2> $OUT/run.err
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
for f in glob.glob("gpurun_out/prof_stats2/trace/**/*kernel_trace.csv", recursive=True):
  rows = [r for r in csv.DictReader(open(f)) if "stats_grouped" in r["Kernel_Name"]]
  by = collections.defaultdict(list)
  for r in rows:
    by[int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
  for g, v in sorted(by.items()):
    print("grid", g, "dispatches", len(v), "avg us %.1f min %.1f" % (sum(v) / len(v), min(v)))
PY
