#!/bin/bash
# Device ISA + resource usage of one frame-kernel instantiation (tuning aid).
# usage: [TF_LOGN=11 TF_NT=45 KERNEL=<mangled name>] tools/isa.sh [extra hipcc flags]   -> /tmp/isa/kernels.s, /tmp/isa/kernel.s
mkdir -p /tmp/isa && cd /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fvisibility=hidden ${NOLSO--Xclang -target-feature -Xclang -load-store-opt} ${NOSLP--fno-slp-vectorize} "$@" -S --cuda-device-only \
    -DTF_LOGN=${TF_LOGN:-11} -DTF_NT=${TF_NT:-45} -I/root/repo/include \
    -Rpass-analysis=kernel-resource-usage -o kernels.s /root/repo/odr-dabmod_amd/csrc/${SRC:-tf_inst.hip} 2> remarks.log
# default: the cfg 3 kernel, tf_kernel<11, FROM_BITS, GAIN, GUARD, FIR, 45 taps, ..., EQ>
S=${KERNEL:-_ZN6dabgpu12_GLOBAL__N_19tf_kernelILi11ELb1ELb1ELb1ELb1ELi45ELb0ELb0ELb0ELi0ELb0ELb1EEEvNS_6TfArgsE}
awk -v s="$S:" '$1==s{p=1} p{print} p&&/s_endpgm/{exit}' kernels.s > kernel.s
[ -s kernel.s ] || { echo "symbol $S not found in kernels.s" >&2; exit 1; }
grep -A10 "Function Name: $S" remarks.log | grep -E "VGPRs:|AGPRs|Scratch|Occupancy|LDS" | sed 's/.*remark: //; s/\[-R.*//'
# (pointers that lose their LDS address space turn into FLAT accesses; spills into scratch_: neither belongs in these kernels)
echo "flat_ instructions in the file: $(grep -c '^\s*flat_' kernels.s), scratch_: $(grep -c '^\s*scratch_' kernels.s)"
