#!/bin/bash
# Dev (round 5): kernel statistics of the FD branch with ONE factor (the 8-GPU shape of cfg5)
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r5_fd_one
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PS_DEV_ENV=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/dev_fd_profile.py 1 > $OUT/run.log 2> $OUT/run.err
cd $GRAFT_REPO_ROOT
tail -2 $OUT/run.log
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
python3 - $OUT/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print(f"kernel time {tot / 1e6:.1f} ms in {calls} launches")
for r in rows[:28]:
  print(f"{r['Name'][:84]:84s} {int(r['Calls']):6d} {int(r['TotalDurationNs']) / 1e6:8.2f} ms {float(r['AverageNs']) / 1e3:8.1f} us")
PY
rm -rf $OUT/trace
