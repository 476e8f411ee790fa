#!/bin/bash
# usage: tools/prof_cfg2.sh <outdir-under-gpurun_out> [extra bench args]; env is inherited (PS_NEWTON_PERSISTENT ...)
# Separate rocprofv3 passes (kernel trace + one PMC group each) of the cfg2 bench.
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PS_DEV_ENV=1
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-headline --no-extras --steps 3 --warmup 2 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/bench_trace.json 2> $OUT/trace.err
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$name -- $B > /dev/null 2> $OUT/pmc_$name.err
done
# keep only the small csv files
find $OUT -name "*_agent_info.csv" -delete
find $OUT -name "*.csv" -size +20M -delete
ls -la $OUT/*/*/ 2>/dev/null | head -40
