#!/bin/bash
# Build tuning variants of libdabgpu.so into gpurun_out-independent tools/_variants/ (git-ignored).
# usage: tools/variants.sh name "-DDABGPU_TF_WAVES=2 ..." [name flags]...
set -e
cd "$(dirname "$0")/../odr-dabmod_amd/csrc"
mkdir -p ../../tools/_variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden ${NOLSO--Xclang -target-feature -Xclang -load-store-opt} ${NOSLP--fno-slp-vectorize} $flags \
      -shared -o ../../tools/_variants/libdabgpu_$name.so dabgpu_kernels.hip dabgpu_api.hip \
      -Rpass-analysis=kernel-resource-usage 2> ../../tools/_variants/$name.log &
done
wait
for f in ../../tools/_variants/*.log; do
  echo "== $f"; grep -A6 "tf_kernelILi11ELb1ELb1ELb1ELb1ELi48" $f | grep -E "VGPRs:|Scratch|Occupancy" | sed 's/.*remark: //; s/\[-R.*//'
done
