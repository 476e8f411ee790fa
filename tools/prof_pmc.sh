#!/bin/bash
# usage: tools/prof_pmc.sh <outdir-under-gpurun_out> <python script + args ...>
# Kernel trace + separate rocprofv3 --pmc passes (one counter group each, never combined with
# other trace domains) of an arbitrary dev script; summarize with tools/summarize_pmc2.py.
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PS_DEV_ENV=1
B="python3 $GRAFT_REPO_ROOT/$@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/run.log 2> $OUT/trace.err
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
  name=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$name -- $B > /dev/null 2> $OUT/pmc_$name.err
done
find $OUT -name "*_agent_info.csv" -delete
find $OUT -name "*.csv" -size +20M -delete
cd $GRAFT_REPO_ROOT
python3 tools/summarize_pmc2.py $OUT $OUT/summary.json > /dev/null
tail -3 $OUT/run.log
