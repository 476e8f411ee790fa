#!/bin/bash
# Round-4 evidence run (on the GPU box): the full bench line, then rocprofv3 kernel statistics + the
# separate --pmc passes for cfg2, the headline set, the ViT-B recompute, the eigh path, the FD branch
# and the every-step update(); small summaries are collected under gpurun_out/r04/.
cd $GRAFT_REPO_ROOT
R=gpurun_out/r04; mkdir -p $R
timeout 1200 python3 bench.py --steps 20 --warmup 5 > $R/r04_bench_full.json 2> $R/r04_bench_full.err
bash tools/prof_cfg2.sh r04_cfg2 > $R/prof_cfg2.log 2>&1
python3 tools/summarize_pmc2.py gpurun_out/r04_cfg2 $R/r04_cfg2_pmc_by_kernel.json > /dev/null 2>&1
cp gpurun_out/r04_cfg2/trace/*/*kernel_stats.csv $R/r04_cfg2_kernel_stats.csv 2>/dev/null
cp gpurun_out/r04_cfg2/bench_trace.json $R/r04_cfg2_bench_under_rocprof.json 2>/dev/null
bash tools/prof_cfg2.sh r04_headline --workload headline_64x1024_p4 > $R/prof_headline.log 2>&1
python3 tools/summarize_pmc2.py gpurun_out/r04_headline $R/r04_headline_pmc_by_kernel.json > /dev/null 2>&1
cp gpurun_out/r04_headline/trace/*/*kernel_stats.csv $R/r04_headline_kernel_stats.csv 2>/dev/null
cp gpurun_out/r04_headline/bench_trace.json $R/r04_headline_bench_under_rocprof.json 2>/dev/null
for pair in "vitb:tools/dev_vitb_step.py" "eigh:tools/dev_eigh_cj_sweep.py default" "fd_cfg5:tools/dev_fd_profile.py" "update_step_vitb:tools/dev_update_step.py"; do
  name=${pair%%:*}; cmd=${pair#*:}
  bash tools/prof_pmc.sh r04_$name $cmd > $R/prof_$name.log 2>&1
  cp gpurun_out/r04_$name/summary.json $R/r04_${name}_pmc_by_kernel.json 2>/dev/null
  cp gpurun_out/r04_$name/trace/*/*kernel_stats.csv $R/r04_${name}_kernel_stats.csv 2>/dev/null
  cp gpurun_out/r04_$name/run.log $R/r04_${name}_run.log 2>/dev/null
done
ls -la $R
