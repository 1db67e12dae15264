#!/bin/bash
# TIMING EXPERIMENT (wrong boundary outputs at the ends of runs, never part of the library): an UPPER BOUND for handing the
# boundary between two runs of symbols from run to run through memory instead of transforming one symbol twice.  Today a run
# of the cfg 3 kernel transforms the symbol AFTER its last one as well (the look-ahead that its last 44 boundary outputs need):
# 3 transforms for 2 symbols at 16 frames per call on one lane, 4 for 3 on three lanes, 2 for 1 when a single frame is cut
# into 77 runs.  The scratch copy simply drops the look-ahead iteration of the equalised-boundary variant.
# Builds tools/_variants/libdabgpu_{base,nolook}.so; time with tools/experiments/exp_r05.py lanes (DABGPU_LIB=...).
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
"$ROOT/tools/variants.sh" base ""
d="$ROOT/tools/_variants/src_nolook"
rm -rf "$d"; mkdir -p "$d/odr-dabmod_amd" "$d/include"
cp -r "$ROOT/odr-dabmod_amd/csrc" "$d/odr-dabmod_amd/csrc"; cp "$ROOT/include/"*.h "$d/include/"
rm -f "$d/odr-dabmod_amd/csrc/"*.o "$d/odr-dabmod_amd/csrc/libdabgpu.so"
python3 - "$d/odr-dabmod_amd/csrc/tf_kernel.h" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
old = "    const int s_stop = ((FIR || WIN) && s_end < nsym) ? s_end + 1 : s_end;"
new = "    const int s_stop = ((FIR || WIN) && !EQ && s_end < nsym) ? s_end + 1 : s_end;   // EXPERIMENT: no look-ahead transform"
assert s.count(old) == 1
open(p, "w").write(s.replace(old, new, 1))
PY
make -s -C "$d/odr-dabmod_amd/csrc" -j8 > "$ROOT/tools/_variants/nolook.log" 2>&1
cp "$d/odr-dabmod_amd/csrc/libdabgpu.so" "$ROOT/tools/_variants/libdabgpu_nolook.so"
cmp -s "$ROOT/tools/_variants/libdabgpu_nolook.so" "$ROOT/tools/_variants/libdabgpu_base.so" && { echo "the patch did not change the library" >&2; exit 1; }
echo "built tools/_variants/libdabgpu_nolook.so"
