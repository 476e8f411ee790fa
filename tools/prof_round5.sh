#!/bin/bash
# Round-5 evidence run (on the GPU box): the full bench line, then rocprofv3 kernel statistics + the
# separate --pmc passes for cfg2, the headline set, the eigh path (cfg3, tridiagonalisation), the
# every-step update() in both state modes, the quantized-state kernels and the ViT-B recompute;
# small summaries are collected under gpurun_out/r05/.
cd $GRAFT_REPO_ROOT
R=gpurun_out/r05; mkdir -p $R
timeout 1500 python3 bench.py --steps 20 --warmup 5 > $R/r05_bench_full.json 2> $R/r05_bench_full.err
# the N > 1 code path (RCCL init, async all-gathers, barriers, sharded ViT-B / FD legs) with a one-rank group
PS_BENCH_FORCE_DIST=1 timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $R/r05_bench_rccl_world1_forced_dist.json 2> $R/r05_bench_rccl_world1.err
bash tools/prof_cfg2.sh r05_cfg2 > $R/prof_cfg2.log 2>&1
python3 tools/summarize_pmc2.py gpurun_out/r05_cfg2 $R/r05_cfg2_pmc_by_kernel.json > /dev/null 2>&1
cp gpurun_out/r05_cfg2/trace/*/*kernel_stats.csv $R/r05_cfg2_kernel_stats.csv 2>/dev/null
cp gpurun_out/r05_cfg2/bench_trace.json $R/r05_cfg2_bench_under_rocprof.json 2>/dev/null
bash tools/prof_cfg2.sh r05_headline --workload headline_64x1024_p4 > $R/prof_headline.log 2>&1
python3 tools/summarize_pmc2.py gpurun_out/r05_headline $R/r05_headline_pmc_by_kernel.json > /dev/null 2>&1
cp gpurun_out/r05_headline/trace/*/*kernel_stats.csv $R/r05_headline_kernel_stats.csv 2>/dev/null
cp gpurun_out/r05_headline/bench_trace.json $R/r05_headline_bench_under_rocprof.json 2>/dev/null
for pair in "eigh:tools/dev_eigh_one.py" "update_step_vitb_donated:tools/dev_update_step.py --donate" "update_step_vitb:tools/dev_update_step.py" "quant:tools/dev_r5_quant.py 1" "vitb:tools/dev_vitb_step.py" "fd_cfg5:tools/dev_fd_profile.py"; do
  name=${pair%%:*}; cmd=${pair#*:}
  bash tools/prof_pmc.sh r05_$name $cmd > $R/prof_$name.log 2>&1
  cp gpurun_out/r05_$name/summary.json $R/r05_${name}_pmc_by_kernel.json 2>/dev/null
  cp gpurun_out/r05_$name/trace/*/*kernel_stats.csv $R/r05_${name}_kernel_stats.csv 2>/dev/null
  cp gpurun_out/r05_$name/run.log $R/r05_${name}_run.log 2>/dev/null
done
# the eigh path on ONE stream group: kernel durations that do not overlap (the default run's sums do), and the
# mat-vec's duration per trailing-matrix size against the bytes it reads
LINES_SHOWN=40 bash tools/prof_r5_eigh.sh one PS_EIGH_TD_STREAMS=1 > $R/prof_eigh_one_stream_group.log 2>&1
cp gpurun_out/prof_r5_eigh_one/kernel_stats.csv $R/r05_eigh_one_stream_group_kernel_stats.csv 2>/dev/null
bash tools/prof_r5_symv_by_column.sh 1 > $R/r05_eigh_matvec_by_column.txt 2>&1
rm -rf gpurun_out/prof_r5_eigh_one gpurun_out/prof_r5_symv_cols
# the raw traces stay on the box: only the summaries travel back (gpurun merges <= 64 MiB)
rm -rf gpurun_out/r05_*
ls -la $R
