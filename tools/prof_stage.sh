#!/bin/bash
# usage: prof_stage.sh <tag> <workload>; env inherited.  Prints the durations of the full-size stage launches.
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_stage_$1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/dev_stage_launch.py $2 > $OUT/run.log 2> $OUT/run.err
cd $GRAFT_REPO_ROOT
tail -1 $OUT/run.log
python3 - <<PY
import csv, glob, statistics
f = sorted(glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True))[-1]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "newton_stage_kernel" in r["Kernel_Name"]]
d.sort(reverse=True)
top = d[:max(6, len(d) // 3)]
print("$1: stage launches %d, longest third: median %.1f us  min %.1f  max %.1f; all: sum %.2f ms" % (len(d), statistics.median(top), min(top), max(top), sum(d) / 1e3))
PY
