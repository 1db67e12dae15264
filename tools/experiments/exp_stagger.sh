#!/bin/bash
# TIMING EXPERIMENT: do the workgroups of a small launch (one wave of 1024 or fewer, all started together and therefore in the
# same phase of their symbol loops) lose throughput to lockstep -- their LDS and VALU phases colliding instead of interleaving?
# Scratch builds with a start stagger of (blockIdx >> 8) & 3 quarters of a symbol period.  Time with tools/time_small.py.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
"$ROOT/tools/variants.sh" base ""
for q in 20 40 80; do
  d="$ROOT/tools/_variants/src_stag$q"
  rm -rf "$d"; mkdir -p "$d/odr-dabmod_amd" "$d/include"
  cp -r "$ROOT/odr-dabmod_amd/csrc" "$d/odr-dabmod_amd/csrc"; cp "$ROOT/include/"*.h "$d/include/"
  rm -f "$d/odr-dabmod_amd/csrc/"*.o "$d/odr-dabmod_amd/csrc/libdabgpu.so"
  python3 - "$d/odr-dabmod_amd/csrc/tf_kernel.h" $q <<'PY'
import sys
p, q = sys.argv[1], sys.argv[2]
s = open(p).read()
old = '''    lds_barrier();

    constexpr int K = G::K, nsym = G::nb_symbols + 1;'''
new = '''    lds_barrier();
    if (EQ) {
        const int k = (int)((blockIdx.x >> 8) & 3u);
        for (int i = 0; i < k; ++i) __builtin_amdgcn_s_sleep(%s);
    }

    constexpr int K = G::K, nsym = G::nb_symbols + 1;''' % q
assert old in s
s = s.replace(old, new, 1)
open(p, "w").write(s)
PY
  make -s -C "$d/odr-dabmod_amd/csrc" -j8 > "$ROOT/tools/_variants/stag$q.log" 2>&1
  cp "$d/odr-dabmod_amd/csrc/libdabgpu.so" "$ROOT/tools/_variants/libdabgpu_stag$q.so"
  echo "built stag$q"
done
