#!/bin/bash
# usage: tools/prof_sq.sh <outdir-under-gpurun_out> <python script + args ...>
# SQ-level counters of the kernels of a dev script (separate rocprofv3 --pmc passes; never
# combined with other trace domains): where a wavefront's cycles go.
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PS_DEV_ENV=1
B="python3 $GRAFT_REPO_ROOT/$@"
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_LEVEL_LDS" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_DEP_WAIT"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-60)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$name -- $B > /dev/null 2> $OUT/pmc_$name.err
done
find $OUT -name "*_agent_info.csv" -delete
find $OUT -name "*.csv" -size +20M -delete
cd $GRAFT_REPO_ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json, os
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
  for r in csv.DictReader(open(f)):
    k = r.get("Kernel_Name", "")
    if "stage" not in k: continue
    agg[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
json.dump(res, open(out + "/sq_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
