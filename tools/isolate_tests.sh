#!/bin/bash
# GPU box: every test of the given files in a process of its own (order dependence shows as a test that passes in the suite
# and fails alone).  Writes gpurun_out/isolate.log: one line per FAILED test, then the counts.
#   tools/isolate_tests.sh tests/test_gpu_model.py [more files]
cd $GRAFT_REPO_ROOT
L=gpurun_out/isolate.log
: > $L
n=0; bad=0
for id in $(python -m pytest "$@" -m gpu --collect-only -q -p no:cacheprovider 2>/dev/null | grep "::"); do
  n=$((n+1))
  if ! timeout -k 10 300 python -m pytest "$id" -q -x -p no:cacheprovider > gpurun_out/isolate_one.log 2>&1; then
    bad=$((bad+1)); echo "FAILED $id" >> $L; tail -n 5 gpurun_out/isolate_one.log >> $L
  fi
  if [ $((n % 20)) = 0 ]; then echo "[isolate] $n run, $bad failed"; fi
done
echo "ran $n failed $bad" >> $L
tail -n 30 $L
