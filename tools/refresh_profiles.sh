#!/bin/bash
# Runs on the GPU box (gpurun): the bench lines kept under profiles/ plus the in-step kernel summaries and the replay-only
# traces of the two profiled workloads, written to gpurun_out/refresh/ (copy to profiles/<tag>_* afterwards).
#   tools/refresh_profiles.sh [tag]      (tag = r05 by default: the round the files are named after)
set -e
TAG=${1:-r05}
O=$GRAFT_REPO_ROOT/gpurun_out/refresh
rm -rf $O
mkdir -p $O
cd $GRAFT_REPO_ROOT
IGCN_BENCH_PROFILE_DIR=$O/full timeout -k 5 600 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_full.json 2> $O/full.err
cp $O/full/full_kernel_stats.csv $O/${TAG}_bench_full_kernel_stats.csv
cp $O/full/full_replay_kernel_stats.csv $O/${TAG}_bench_full_replay_kernel_stats.csv      # the timed replays only
cp $O/full/stress_child/stress_replay_kernel_stats.csv $O/${TAG}_bench_full_stress_child_replay_kernel_stats.csv || true
IGCN_BENCH_PROFILE_DIR=$O/stress timeout -k 5 400 python bench.py --workload stress > $O/${TAG}_bench_stress.json 2> $O/stress.err
cp $O/stress/stress_kernel_stats.csv $O/${TAG}_bench_stress_kernel_stats.csv
cp $O/stress/stress_replay_kernel_stats.csv $O/${TAG}_bench_stress_replay_kernel_stats.csv
timeout -k 5 300 python bench.py --workload sgcn > $O/${TAG}_bench_sgcn.json 2> $O/sgcn.err
IGCN_BENCH_GDC_SERIAL=1 timeout -k 5 400 python bench.py --no-roofline --no-cpu-baseline --no-stress > $O/${TAG}_bench_pipeline.json 2> $O/pipeline.err
rm -rf $O/full $O/stress
tools/replay_trace.sh refresh/replay_full full 10 > /dev/null
tools/replay_trace.sh refresh/replay_stress stress 6 > /dev/null
for w in full stress; do
  cp $O/replay_$w/replay_summary.csv $O/${TAG}_replay_${w}_summary.csv
  cp $O/replay_$w/replay_summary_one_replay.csv $O/${TAG}_replay_${w}_one_replay.csv
done
rm -rf $O/replay_full $O/replay_stress
python tools/gdc_bench.py > $O/${TAG}_gdc_bench.txt 2>/dev/null
for f in full stress sgcn pipeline; do python -c "
import json,sys
d=json.load(open('$O/${TAG}_bench_$f.json')); print('$f', d['value'], d['unit'], d['ms_per_step'], 'ms', (d.get('stress') or {}).get('ms_per_step',''), json.dumps(d.get('pipeline',''))[:300])"; done
tail -4 $O/${TAG}_replay_full_summary.csv; tail -4 $O/${TAG}_replay_stress_summary.csv
# configs[4]: k_ds_agg with parts compiled out (kernel durations inside the captured step) and the in-step phase stamps
tools/dense_ablate.sh > $O/dense_ablate.log 2>&1 && cp $GRAFT_REPO_ROOT/gpurun_out/dense_ablate.txt $O/${TAG}_dense_ablate.txt
python tools/dense_ablate_report.py $O/${TAG}_dense_ablate.txt | grep -E "==|k_ds_agg<" || true
