#!/bin/bash
# GPU box: per-kernel HBM traffic of the graphed step's replays -> gpurun_out/<tag>/step_traffic_<workload>.csv
#   bash tools/step_traffic.sh <tag> [workload] [replays]
set -e
TAG=$1; WL=${2:-full}; K=${3:-6}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $GRAFT_REPO_ROOT/tools/replay_trace.py $WL $K > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $GRAFT_REPO_ROOT/tools/replay_trace.py $WL $K > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $GRAFT_REPO_ROOT/tools/replay_trace.py $WL $K > $O/write.log 2>&1
cd $GRAFT_REPO_ROOT
CAL=$(ls profiles/r0[0-9]_pmc/traffic.json | tail -1)
python3 tools/step_traffic.py $O $K $CAL > $O/step_traffic_$WL.csv
rm -rf $O/trace $O/fetch $O/write
cat $O/step_traffic_$WL.csv
