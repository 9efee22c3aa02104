set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r04}
O=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $GRAFT_REPO_ROOT/tools/roofline_kernel.py > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $GRAFT_REPO_ROOT/tools/roofline_kernel.py > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $GRAFT_REPO_ROOT/tools/roofline_kernel.py > $O/write.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_report.py $O > $O/traffic.json
cat $O/traffic.json | head -60
