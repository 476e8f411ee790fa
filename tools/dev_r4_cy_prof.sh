#!/bin/bash
# dev (GPU box): kernel-only durations (rocprofv3 --kernel-trace --stats) of the FD product kernels per build
# variant of gemm_bf16.hip (tools/ab_build.sh; the script run under the profiler: CY_SCRIPT, default dev_r4_cystep.py).
# usage: tools/dev_r4_cy_prof.sh name:flags ...   e.g.  hv0:-DPS_HVAR=0 hs4:-DPS_HSETS=4,-DPS_HVAR=6
cd "$(dirname "$0")/.."
ROOT=$(pwd)
for a in "$@"; do v=${a%%:*}; fl=$(echo "${a#*:}" | tr ',' ' '); tools/ab_build.sh $v gemm_bf16.hip $fl > /dev/null 2>&1 & done; wait
export TMPDIR=/tmp
for a in "$@"; do
  v=${a%%:*}
  out=/tmp/cyprof_$v; rm -rf $out
  cd /tmp
  PS_AB_LIB=$ROOT/.ab/$v/libprecondition_amd.so rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $ROOT/tools/${CY_SCRIPT:-dev_r4_cystep.py} > /tmp/cyprof_$v.log 2>&1
  cd $ROOT
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  if [ -z "$f" ]; then echo "$v: no stats"; tail -3 /tmp/cyprof_$v.log; continue; fi
  python3 -c "
import csv, sys
for r in csv.DictReader(open('$f')):
  if 'gemm_bf16' in r['Name'] or 'fd_cy' in r['Name']:
    print('%-14s %-46s calls %4s avg %7.1f us  min %7.1f' % ('$v', r['Name'][:46], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
"
done
