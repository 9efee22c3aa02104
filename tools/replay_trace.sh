#!/bin/bash
# GPU box: replay-only kernel trace of a workload -> gpurun_out/<tag>/{replay_summary.csv,txt}
#   tools/replay_trace.sh <tag> [workload] [replays]
set -e
TAG=$1; WL=${2:-full}; K=${3:-10}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp IGCN_REPLAY_META=$O/replay_meta.txt
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $GRAFT_REPO_ROOT/tools/replay_trace.py $WL $K > $O/run.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/replay_trace.py --summarise $O $O/replay_summary.csv > $O/replay_summary.txt
rm -rf $O/trace
head -70 $O/replay_summary.txt
