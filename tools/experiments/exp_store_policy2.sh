#!/bin/bash
# A/B over every bench workload: non-temporal output stores in the frame kernel (all variants) and in the x4 resampler.
# Run on the GPU box:  for l in tools/_variants/*.so; do DABGPU_LIB=$l python bench.py --no-cpu-baseline --counters off; done
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
mkdir -p "$ROOT/tools/_variants"
"$ROOT/tools/variants.sh" base ""
d="$ROOT/tools/_variants/src_ntall"
rm -rf "$d"; mkdir -p "$d/odr-dabmod_amd" "$d/include"
cp -r "$ROOT/odr-dabmod_amd/csrc" "$d/odr-dabmod_amd/csrc"; cp "$ROOT/include/"*.h "$d/include/"
rm -f "$d/odr-dabmod_amd/csrc/"*.o "$d/odr-dabmod_amd/csrc/libdabgpu.so"
# (the frame kernel: every variant's stores non-temporal through the tool-only flag, tf_kernel.h: kStoreAux)
grep -q "DABGPU_STORE_AUX" "$d/odr-dabmod_amd/csrc/tf_kernel.h" || { echo "tf_kernel.h no longer reads DABGPU_STORE_AUX" >&2; exit 1; }
python3 - "$d/odr-dabmod_amd/csrc/resampler.hip" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
old = '''                d4[0] = make_float4(a0.x, a0.y, a1.x, a1.y);
                if (Q == 4) d4[1] = make_float4(a2.x, a2.y, a3.x, a3.y);'''
new = '''                typedef float v4f_ __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store((v4f_){a0.x, a0.y, a1.x, a1.y}, reinterpret_cast<v4f_ *>(d4));
                if (Q == 4) __builtin_nontemporal_store((v4f_){a2.x, a2.y, a3.x, a3.y}, reinterpret_cast<v4f_ *>(d4 + 1));'''
assert old in s
s = s.replace(old, new, 1)
open(p, "w").write(s)
PY
make -s -C "$d/odr-dabmod_amd/csrc" -j8 NOSLP="-fno-slp-vectorize -DDABGPU_STORE_AUX=2" > "$ROOT/tools/_variants/ntall.log" 2>&1
cp "$d/odr-dabmod_amd/csrc/libdabgpu.so" "$ROOT/tools/_variants/libdabgpu_ntall.so"
echo built ntall
