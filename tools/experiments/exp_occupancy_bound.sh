#!/bin/bash
# TIMING EXPERIMENT (same samples, same instruction stream): what the cfg 3 frame kernel loses when fewer of its workgroups fit
# a CU.  A 16-points-per-lane variant (16 . 16 . 8 on 128 lanes: the two-exchange transform whose upper bound is
# tools/experiments/exp_two_exchanges_bound.sh) keeps one 16.6 KB exchange buffer per TWO waves instead of per four, i.e. ~27.5 KB of LDS per
# 128-lane workgroup: five per CU = 10 waves where today's kernel has 16.  Here the product kernel simply asks for more dynamic
# LDS than it uses (41 KB -> three workgroups = 12 waves per CU; 54 KB -> two = 8 waves).
# Builds tools/_variants/libdabgpu_{base,occ3,occ2}.so; time with
#   DABGPU_LIB=tools/_variants/libdabgpu_x.so python tools/experiments/exp_r05.py cfg3power 32768
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
"$ROOT/tools/variants.sh" base ""
for v in "occ3 41984" "occ2 55296"; do
  set -- $v
  d="$ROOT/tools/_variants/src_$1"
  rm -rf "$d"; mkdir -p "$d/odr-dabmod_amd" "$d/include"
  cp -r "$ROOT/odr-dabmod_amd/csrc" "$d/odr-dabmod_amd/csrc"; cp "$ROOT/include/"*.h "$d/include/"
  rm -f "$d/odr-dabmod_amd/csrc/"*.o "$d/odr-dabmod_amd/csrc/libdabgpu.so"
  python3 - "$d/odr-dabmod_amd/csrc/tf_launch.hip" "$2" <<'PY'
import sys
p, want = sys.argv[1], int(sys.argv[2])
s = open(p).read()
old = "    return b;\n}\n\n// the frame-kernel variants that window"
new = "    if (eq && b < %d) b = %d;      // EXPERIMENT: fewer workgroups per CU\n    return b;\n}\n\n// the frame-kernel variants that window" % (want, want)
assert s.count(old) == 1
open(p, "w").write(s.replace(old, new, 1))
PY
  make -s -C "$d/odr-dabmod_amd/csrc" -j8 > "$ROOT/tools/_variants/$1.log" 2>&1
  cp "$d/odr-dabmod_amd/csrc/libdabgpu.so" "$ROOT/tools/_variants/libdabgpu_$1.so"
  echo "built tools/_variants/libdabgpu_$1.so"
done
