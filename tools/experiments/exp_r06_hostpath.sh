#!/bin/bash
# Bisect of the asynchronous host path: the SAME box runs tools/time_host_path.py of four source states
# alternately (library + Python plumbing of each state, built in the build container under tools/_variants/hp_<sha>/).
# usage (GPU box): bash tools/experiments/exp_r06_hostpath.sh > gpurun_out/r06_hostpath_bisect.txt
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
python -c "import torch" 2>/dev/null
for round in 1 2; do
  for v in hp_4996e04 hp_a0e01d5 hp_0f78847 HEAD; do
    if [ $v = HEAD ]; then d=.; else d=tools/_variants/$v; fi
    echo "=== $v (pass $round)"
    (cd $d && timeout 300 python tools/time_host_path.py 2>&1 | grep -A4 "^asynchronous")
    (cd $d && timeout 300 python tools/time_host_path.py s16 2>&1 | grep -A4 "^asynchronous")
  done
done
