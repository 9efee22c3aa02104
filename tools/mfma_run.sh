#!/bin/bash
# GPU box: matrix-pipe counters of the MFMA kernels -> gpurun_out/<tag>_pmc/mfma.json   (usage: tools/mfma_run.sh [tag])
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05}
O=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/mtrace -- python3 $GRAFT_REPO_ROOT/tools/mfma_kernels.py > $O/mfma_trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/mpmc -- python3 $GRAFT_REPO_ROOT/tools/mfma_kernels.py > $O/mfma_pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/mfma_report.py $O > $O/mfma${IGCN_ATTN_EXACT_FP32:+_exact_fp32}.json
rm -rf $O/mtrace $O/mpmc
cat $O/mfma${IGCN_ATTN_EXACT_FP32:+_exact_fp32}.json | head -150
