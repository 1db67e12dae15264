#!/bin/bash
# Round 6: what ANY faster boundary filter could buy in transmission modes II / III -- the packed-dual-transform kernels with the
# boundary filter removed altogether (a scratch copy of the sources with one `return;`, WRONG samples: an upper bound).
#   bash tools/experiments/exp_r06_no_boundary_bound.sh          (here: builds tools/_variants/libdabgpu_exp_nobnd.so)
#   gpurun -- 'for l in "" tools/_variants/libdabgpu_exp_nobnd.so; do DABGPU_LIB=${l:+$PWD/$l} python tools/time_modes.py 23 16384; done'
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
d=$ROOT/tools/_variants/src_exp_nobnd
rm -rf $d; mkdir -p $d/odr-dabmod_amd $d/include
cp -r $ROOT/odr-dabmod_amd/csrc $d/odr-dabmod_amd/csrc; cp $ROOT/include/*.h $d/include/
rm -f $d/odr-dabmod_amd/csrc/libdabgpu.so $d/odr-dabmod_amd/csrc/tf_inst_8_45.o $d/odr-dabmod_amd/csrc/tf_inst_9_45.o $d/odr-dabmod_amd/csrc/tf_inst_10_45.o
python3 - "$d/odr-dabmod_amd/csrc/tf_kernel.h" <<'PY'
import sys
p = sys.argv[1]; s = open(p).read()
old = '            static_assert(!BWIN || NT == 45, "eleven quads of four outputs, four groups of twelve taps");\n'
assert s.count(old) == 1
open(p, "w").write(s.replace(old, old + "            return;   // EXPERIMENT (upper bound, wrong samples): no boundary filter at all\n"))
PY
make -s -C $d/odr-dabmod_amd/csrc -j8
cp $d/odr-dabmod_amd/csrc/libdabgpu.so $ROOT/tools/_variants/libdabgpu_exp_nobnd.so
echo "built tools/_variants/libdabgpu_exp_nobnd.so"
