#!/bin/bash
# Dev: rocprofv3 kernel trace of the cfg3 eigh step (Cholesky-Jacobi path); prints per-kernel
# totals and how much kernel time overlaps between the two streams.
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_eigh_cj
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/dev_eigh_cj_sweep.py "$@" > $OUT/run.log 2> $OUT/run.err
cd $GRAFT_REPO_ROOT
cat $OUT/run.log | tail -3
python3 tools/summarize_trace.py $OUT/trace
