#!/bin/bash
# TIMING EXPERIMENT (wrong results): what would the cfg 3 kernel gain if its final FFT stage left every lane with PAIRS of
# consecutive samples, so that a symbol went out in 16-byte stores, 1 KB contiguous per wave (four for the body, one or two for the
# cyclic prefix) instead of ten 512-byte ones?  The scratch build stores the registers it has as if they were such pairs.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
"$ROOT/tools/variants.sh" base ""
d="$ROOT/tools/_variants/src_st16"
rm -rf "$d"; mkdir -p "$d/odr-dabmod_amd" "$d/include"
cp -r "$ROOT/odr-dabmod_amd/csrc" "$d/odr-dabmod_amd/csrc"; cp "$ROOT/include/"*.h "$d/include/"
rm -f "$d/odr-dabmod_amd/csrc/"*.o "$d/odr-dabmod_amd/csrc/libdabgpu.so"
python3 - "$d/odr-dabmod_amd/csrc/tf_kernel.h" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
old = '''        if (lane_on && !(EQ && lookahead)) {
            const int m_cp = (N - cpl) / T;   // first register slot that is also copied into the prefix'''
new = '''        if constexpr (EQ) {
            if (!lookahead) {
                typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
                // body: samples [0, N - C) as pairs: lane t, slot k -> pair index t + 256 k (k < 4): 1 KB contiguous per wave
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const cf y0 = scaled(v[2 * k]), y1 = scaled(v[2 * k + 1]);
                    const v4u_ d = {__builtin_bit_cast(unsigned, y0.x), __builtin_bit_cast(unsigned, y0.y),
                                    __builtin_bit_cast(unsigned, y1.x), __builtin_bit_cast(unsigned, y1.y)};
                    if (2 * (t + 256 * k) < N - C)
                        __builtin_amdgcn_raw_buffer_store_b128(d, orsrc, t * 16, (pos + cpl + 512 * k) * 8, kStoreAux);
                    // prefix: the last cp samples = pairs 772 ... 1023 -> slot 3 (pairs 768 + t), t >= 4
                    if (k == 3 && t >= 4)
                        __builtin_amdgcn_raw_buffer_store_b128(d, orsrc, (t - 4) * 16, pos * 8, kStoreAux);
                }
            }
        } else
        if (lane_on && !(EQ && lookahead)) {
            const int m_cp = (N - cpl) / T;   // first register slot that is also copied into the prefix'''
assert old in s
s = s.replace(old, new, 1)
open(p, "w").write(s)
PY
make -s -C "$d/odr-dabmod_amd/csrc" -j8 > "$ROOT/tools/_variants/st16.log" 2>&1
cp "$d/odr-dabmod_amd/csrc/libdabgpu.so" "$ROOT/tools/_variants/libdabgpu_st16.so"
echo "built st16"
