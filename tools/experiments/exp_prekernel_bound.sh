#!/bin/bash
# TIMING EXPERIMENT (wrong results, never part of the library): what would the cfg 3 frame kernel cost if QpskSymbolMapper ..
# DifferentialModulator ran in a pre-kernel that hands every lane its six carriers' accumulated phases as ONE dword per
# symbol?  The scratch copy replaces the symbol loop's bit gather (barrier, 12 LDS byte reads, ~45 VALU, one LDS store) by a
# per-lane dword load requested a symbol ahead -- the instruction stream such a kernel would have; the phases it reads are
# garbage.  Builds tools/_variants/libdabgpu_{base,prek}.so; time with tools/time_cfg3_variants.py.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
"$ROOT/tools/variants.sh" base ""
d="$ROOT/tools/_variants/src_prek"
rm -rf "$d"; mkdir -p "$d/odr-dabmod_amd" "$d/include"
cp -r "$ROOT/odr-dabmod_amd/csrc" "$d/odr-dabmod_amd/csrc"; cp "$ROOT/include/"*.h "$d/include/"
rm -f "$d/odr-dabmod_amd/csrc/"*.o "$d/odr-dabmod_amd/csrc/libdabgpu.so"
python3 - "$d/odr-dabmod_amd/csrc/tf_kernel.h" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
old = '''            lds_barrier();                    // bitbuf[bb] written (prologue / previous iteration)
            if (s >= 2) advance(bitbuf + bb * kBitStride);
            pf = fetch_block(s - 1);            // block of symbol s+1 (clamped; unused past the end)'''
new = '''            if (EQ) {
                P = pf_prev;                      // EXPERIMENT: the phases arrive as one dword per lane and symbol
                pf = reinterpret_cast<const uint32_t *>(fbits)[(size_t)(min(s, 74)) * (K / 16) + (t % (K / 16))];
            } else {
            lds_barrier();                    // bitbuf[bb] written (prologue / previous iteration)
            if (s >= 2) advance(bitbuf + bb * kBitStride);
            pf = fetch_block(s - 1);            // block of symbol s+1 (clamped; unused past the end)
            }'''
assert old in s
s = s.replace(old, new, 1)
old = '''            bitbuf[(bb ^ 1) * kBitStride + bit_slot] = pf;'''
new = '''            if (!EQ) bitbuf[(bb ^ 1) * kBitStride + bit_slot] = pf;
            else { pf_prev = pf & 0x333333u; asm volatile("" : "+v"(pf_prev)); }   // (the wait for the load lands HERE, before the stores)'''
assert old in s
s = s.replace(old, new, 1)
old = '''    int bb = 0;                 // which bitbuf half holds the block of the current symbol'''
new = '''    int bb = 0;                 // which bitbuf half holds the block of the current symbol
    uint32_t pf_prev = 0x123123u;'''
assert old in s
s = s.replace(old, new, 1)
open(p, "w").write(s)
PY
make -s -C "$d/odr-dabmod_amd/csrc" -j8 > "$ROOT/tools/_variants/prek.log" 2>&1
cp "$d/odr-dabmod_amd/csrc/libdabgpu.so" "$ROOT/tools/_variants/libdabgpu_prek.so"
echo "built tools/_variants/libdabgpu_prek.so"
