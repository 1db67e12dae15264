#!/bin/bash
# before / after of the stream queries in dabgpu_chain_collect: one-frame batches, 6000 of them, twice each
cd "$(dirname "$0")/../.."
for pass in 1 2; do
  for v in tools/_variants/hp_0f78847 .; do
    echo "=== $v (pass $pass)"
    (cd $v && python tools/experiments/exp_r06_async_series.py complexf 1 6144 512 2>&1 | grep -v amdgpu.ids)
  done
done
echo "=== . s16 B=8 / complexf B=32"
python tools/experiments/exp_r06_async_series.py s16 8 512 64 2>&1 | grep -v amdgpu.ids
python tools/experiments/exp_r06_async_series.py complexf 32 256 32 2>&1 | grep -v amdgpu.ids
