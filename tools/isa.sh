#!/bin/bash
# Device ISA + resource usage of one tf_kernel instantiation (tuning aid).
# usage: tools/isa.sh [extra hipcc flags]   -> /tmp/isa/kernels.s, /tmp/isa/cfg3.s
mkdir -p /tmp/isa && cd /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fvisibility=hidden ${NOLSO--Xclang -target-feature -Xclang -load-store-opt} ${NOSLP--fno-slp-vectorize} "$@" -S --cuda-device-only \
    -Rpass-analysis=kernel-resource-usage -o kernels.s /root/repo/odr-dabmod_amd/csrc/dabgpu_kernels.hip 2> remarks.log
# default: the cfg 3 kernel, tf_kernel<11, FROM_BITS, GAIN, GUARD, FIR, 45 taps, no CFR, no GVAR, ZONLY>
S=${KERNEL:-_ZN6dabgpu12_GLOBAL__N_19tf_kernelILi11ELb1ELb1ELb1ELb1ELi45ELb0ELb0ELb1EEEvNS_6TfArgsE}
awk -v s="$S:" '$1==s{p=1} p{print} p&&/s_endpgm/{exit}' kernels.s > cfg3.s
[ -s cfg3.s ] || { echo "symbol $S not found in kernels.s" >&2; exit 1; }
grep -A10 "Function Name: $S" remarks.log | grep -E "VGPRs:|AGPRs|Scratch|Occupancy|LDS" | sed 's/.*remark: //; s/\[-R.*//'
# (pointers that lose their LDS address space turn into FLAT accesses; spills into scratch_: neither belongs in these kernels)
echo "flat_ instructions in the file: $(grep -c '^\s*flat_' kernels.s), scratch_: $(grep -c '^\s*scratch_' kernels.s)"
