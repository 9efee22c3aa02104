#!/bin/bash
# Runs on the GPU box: what k_ds_agg (configs[4], both passes per launch) costs INSIDE the captured stress step with parts of
# its arithmetic compiled out (DS_ABL in csrc/sgcn_dense.hip) — the results are wrong in those builds, only the durations
# mean anything.  Writes gpurun_out/dense_ablate.txt; the shipped library is put back at the end.
#   0 = as shipped   1 = stream only (loads, one VALU fma per product slot, same epilogue and stores)
#   2 = masks kept, products on the VALU   3 = matrix products kept, masks dropped
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/dense_ablate
LIB=ig-gcn_amd/lib/libigcn.so
rm -rf $O gpurun_out/dense_ablate.txt; mkdir -p $O/obj
cp $LIB $O/shipped.so
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-result -Werror=return-type"
SRCS=$(python -c "import sys; sys.path.insert(0, 'ig-gcn_amd'); import build; print(' '.join(build.SOURCES))")
echo $SRCS | tr ' ' '\n' | grep -v sgcn_dense | xargs -P 12 -I{} /opt/rocm/bin/hipcc $FLAGS -c ig-gcn_amd/csrc/{} -o $O/obj/{}.o || exit 1
# DS_ABL_LIST: variants separated by blanks; a number n means -DDS_ABL=n, anything else is passed to hipcc as it stands
for abl in ${DS_ABL_LIST:-1 2 3 0}; do
  case $abl in [0-9]) def="-DDS_ABL=$abl";; *) def="$abl";; esac
  /opt/rocm/bin/hipcc $FLAGS $def -c ig-gcn_amd/csrc/sgcn_dense.hip -o $O/obj/sgcn_dense.hip.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/obj/*.o -o $LIB || exit 1
  IGCN_BENCH_PROFILE_DIR=$O/p$abl timeout -k 5 300 python bench.py --workload stress --no-cpu-baseline > $O/bench_$abl.json 2> $O/bench_$abl.err || { cp $O/shipped.so $LIB; exit 1; }
  echo "== DS_ABL=$abl  $(python -c "import json; d=json.load(open('$O/bench_$abl.json')); print(d['ms_per_step'], 'ms/step')")" >> gpurun_out/dense_ablate.txt
  grep -E "k_ds_" $O/p$abl/stress_replay_kernel_stats.csv >> gpurun_out/dense_ablate.txt
  rm -rf $O/p$abl
done
cp $O/shipped.so $LIB
rm -rf $O/obj $O/shipped.so
cat gpurun_out/dense_ablate.txt
