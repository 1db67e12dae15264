#!/bin/bash
# TIMING EXPERIMENT (wrong results, never part of the library): an UPPER BOUND for a 16-points-per-lane 2048-point transform in
# the frame kernel -- 16 . 16 . 8 on 128 lanes, two LDS exchanges per transform where 8 . 8 . 8 . 4 on 256 lanes has three
# (DESIGN.md section 9.1).  Per symbol such a kernel issues the same LDS bytes per exchange and one exchange fewer, i.e. two
# thirds of today's exchange instructions and barriers.  The scratch copy simply leaves out the THIRD exchange of every
# single-symbol 2048-point inverse transform (scatter, barriers, gather): the instruction stream of the two-exchange kernel
# without any of its costs (twice the state per lane at half the waves per CU, radix-16 butterflies, 28 resident twiddles).
# Builds tools/_variants/libdabgpu_{base,twoex}.so; time with
#   DABGPU_LIB=tools/_variants/libdabgpu_x.so python tools/experiments/exp_r05.py cfg3power 32768
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
"$ROOT/tools/variants.sh" base ""
d="$ROOT/tools/_variants/src_twoex"
rm -rf "$d"; mkdir -p "$d/odr-dabmod_amd" "$d/include"
cp -r "$ROOT/odr-dabmod_amd/csrc" "$d/odr-dabmod_amd/csrc"; cp "$ROOT/include/"*.h "$d/include/"
rm -f "$d/odr-dabmod_amd/csrc/"*.o "$d/odr-dabmod_amd/csrc/libdabgpu.so"
python3 - "$d/odr-dabmod_amd/csrc/device_common.h" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
old = '''            if (NR8 > 3 || RF > 1) exchange<64, DBUF, V>(v, DABGPU_NEXT_BUF, t);'''
new = '''            if ((NR8 > 3 || RF > 1) && !(LOGN == 11 && S > 0 && std::is_same<V, cf>::value))   // EXPERIMENT: no third exchange
                exchange<64, DBUF, V>(v, DABGPU_NEXT_BUF, t);'''
assert s.count(old) == 1
s = s.replace(old, new, 1)
open(p, "w").write(s)
PY
make -s -C "$d/odr-dabmod_amd/csrc" -j8 > "$ROOT/tools/_variants/twoex.log" 2>&1
cp "$d/odr-dabmod_amd/csrc/libdabgpu.so" "$ROOT/tools/_variants/libdabgpu_twoex.so"
cmp -s "$ROOT/tools/_variants/libdabgpu_twoex.so" "$ROOT/tools/_variants/libdabgpu_base.so" && { echo "the patch did not change the library" >&2; exit 1; }
echo "built tools/_variants/libdabgpu_twoex.so"
