#!/bin/bash
# Runs on the GPU box (gpurun): the bench lines kept under profiles/ plus the in-step kernel summaries and the replay-only
# traces of the two profiled workloads, written to gpurun_out/refresh/ (copy to profiles/r03_* afterwards).
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/refresh
rm -rf $O
mkdir -p $O
cd $GRAFT_REPO_ROOT
IGCN_BENCH_PROFILE_DIR=$O/full timeout -k 5 600 python bench.py --steps 20 --warmup 5 > $O/r03_bench_full.json 2> $O/full.err
cp $O/full/full_kernel_stats.csv $O/r03_bench_full_kernel_stats.csv
IGCN_BENCH_PROFILE_DIR=$O/stress timeout -k 5 400 python bench.py --workload stress > $O/r03_bench_stress.json 2> $O/stress.err
cp $O/stress/stress_kernel_stats.csv $O/r03_bench_stress_kernel_stats.csv
timeout -k 5 300 python bench.py --workload sgcn > $O/r03_bench_sgcn.json 2> $O/sgcn.err
timeout -k 5 400 python bench.py --pipeline --no-roofline --no-cpu-baseline --no-stress > $O/r03_bench_pipeline.json 2> $O/pipeline.err
rm -rf $O/full $O/stress
tools/replay_trace.sh refresh/replay_full full 10 > /dev/null
tools/replay_trace.sh refresh/replay_stress stress 6 > /dev/null
for w in full stress; do
  cp $O/replay_$w/replay_summary.csv $O/r03_replay_${w}_summary.csv
  cp $O/replay_$w/replay_summary_one_replay.csv $O/r03_replay_${w}_one_replay.csv
done
rm -rf $O/replay_full $O/replay_stress
for f in full stress sgcn pipeline; do python -c "
import json,sys
d=json.load(open('$O/r03_bench_$f.json')); print('$f', d['value'], d['unit'], d['ms_per_step'], 'ms', (d.get('stress') or {}).get('ms_per_step',''), json.dumps(d.get('pipeline',''))[:300])"; done
tail -4 $O/r03_replay_full_summary.csv; tail -4 $O/r03_replay_stress_summary.csv
