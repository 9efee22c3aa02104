#!/bin/bash
# Runs on the GPU box (gpurun): the four bench lines kept under profiles/ plus the in-step kernel summaries of the two
# profiled workloads, written to gpurun_out/refresh/ (copy to profiles/r02_* afterwards).
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/refresh
mkdir -p $O
cd $GRAFT_REPO_ROOT
IGCN_BENCH_PROFILE_DIR=$O/full timeout -k 5 400 python bench.py > $O/r02_bench_full.json 2> $O/full.err
cp $O/full/full_kernel_stats.csv $O/r02_bench_full_kernel_stats.csv
IGCN_BENCH_PROFILE_DIR=$O/stress timeout -k 5 400 python bench.py --workload stress > $O/r02_bench_stress.json 2> $O/stress.err
cp $O/stress/stress_kernel_stats.csv $O/r02_bench_stress_kernel_stats.csv
timeout -k 5 300 python bench.py --workload sgcn > $O/r02_bench_sgcn.json 2> $O/sgcn.err
timeout -k 5 300 python bench.py --rotate 8 --no-roofline > $O/r02_bench_rotate.json 2> $O/rotate.err
rm -rf $O/full $O/stress
for f in full stress sgcn rotate; do python -c "
import json,sys
d=json.load(open('$O/r02_bench_$f.json')); print('$f', d['value'], d['unit'], d['ms_per_step'], 'ms', d.get('rotating_batches',''))"; done
