#!/bin/bash
# TIMING EXPERIMENT (wrong results, never part of the library): what can replacing ONE of the three LDS exchanges of the
# 2048-point transform by an in-wave register <-> lane transposition buy at most?
#
# The decomposition that keeps 512-point sub-transforms inside a wave is 8 (registers) . 4 (waves: the one cross-wave exchange,
# through LDS) . 8 (lane bits 3-5) . 8 (lane bits 0-2), then the LDS exchange into natural order: TWO LDS exchanges and TWO
# in-wave transpositions of three register bits against three lane bits, where the kernel has three LDS exchanges today.  An
# in-wave transposition level swaps, for each of the four register pairs, one register of the lanes whose bit is 0 with the
# other register of their partners: v_permlane32_swap (lane bit 5) and v_permlane16_swap (bit 4) do that in one instruction per
# dword -- 8 per level --; lane bits 0-3 have no swap instruction: select the outgoing register, move it across by DPP
# (row_ror:8 / row_ror:4+12 / quad_perm), select it into place -- 4 instructions per dword, 32 per level.  Six levels per
# transform: 2 x 8 + 4 x 32 = 144 VALU instructions in place of 8 ds_write_b64 + 8 ds_read_b64 + two barriers.
#
# Three builds, same box:
#   base    the library as it is
#   twoex   the third exchange of every single-symbol 2048-point inverse transform simply left out (r05's experiment):
#           the instruction stream with one LDS exchange less and NOTHING in its place -- the absolute upper bound
#   perm    twoex + the 144 instructions of the transposition levels executed on the transform's registers in its place
#           (an upper bound still: no second radix-8 stage's twiddles by lane group, no extra registers for the selects)
# usage: tools/experiments/exp_r06_permlane_bound.sh ; then on the GPU box
#   for l in base twoex perm; do DABGPU_LIB=tools/_variants/libdabgpu_$l.so python tools/experiments/exp_r05.py cfg3power 32768; done
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
"$ROOT/tools/variants.sh" base ""
for v in twoex perm; do
d="$ROOT/tools/_variants/src_$v"
rm -rf "$d"; mkdir -p "$d/odr-dabmod_amd" "$d/include"
cp -r "$ROOT/odr-dabmod_amd/csrc" "$d/odr-dabmod_amd/csrc"; cp "$ROOT/include/"*.h "$d/include/"
rm -f "$d/odr-dabmod_amd/csrc/"*.o "$d/odr-dabmod_amd/csrc/libdabgpu.so"
python3 - "$d/odr-dabmod_amd/csrc/device_common.h" $v <<'PY'
import sys
p, v = sys.argv[1], sys.argv[2]
s = open(p).read()
old = '''            if (NR8 > 3 || RF > 1) exchange<64, DBUF, V>(v, DABGPU_NEXT_BUF, t);'''
skip = '''            if constexpr (LOGN == 11 && S > 0 && std::is_same<V, cf>::value) {   // EXPERIMENT: no third exchange
                %s
            } else if (NR8 > 3 || RF > 1) exchange<64, DBUF, V>(v, DABGPU_NEXT_BUF, t);'''
perm = '''inwave_transpose_cost(v, t);'''
assert s.count(old) == 1
s = s.replace(old, skip % (perm if v == "perm" else ""), 1)
if v == "perm":
    helper = r'''
// EXPERIMENT: the instruction stream of two in-wave 3-bit transpositions (register index against lane bits 3-5 and 0-2)
template <typename V> static DEV void inwave_transpose_cost(V *v, int t)
{
    float *f = reinterpret_cast<float *>(v);          // 16 dwords: (re, im) of 8 points
    // lane bit 5 / 4: one swap instruction per dword of the four register pairs
#pragma unroll
    for (int pr = 0; pr < 4; ++pr)
#pragma unroll
        for (int c = 0; c < 2; ++c)
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(f[2 * (2 * pr) + c]), "+v"(f[2 * (2 * pr + 1) + c]));
#pragma unroll
    for (int pr = 0; pr < 4; ++pr)
#pragma unroll
        for (int c = 0; c < 2; ++c)
            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(f[2 * (pr + (pr & 2)) + c]), "+v"(f[2 * (pr + (pr & 2) + 2) + c]));
    // lane bits 3, 2, 1, 0: select the outgoing dword, move it across by DPP, select it into place
#define DABGPU_LEVEL(BIT, CTRL, PAIR_A, PAIR_B)                                                             \
    {                                                                                                      \
        const bool hi = (t >> BIT) & 1;                                                                    \
        _Pragma("unroll") for (int pr = 0; pr < 4; ++pr) _Pragma("unroll") for (int c = 0; c < 2; ++c) {  \
            float &a = f[2 * (PAIR_A) + c], &b = f[2 * (PAIR_B) + c];                                      \
            float out = hi ? a : b, in;                                                                    \
            asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "=v"(in) : "v"(out)); \
            a = hi ? in : a;                                                                               \
            b = hi ? b : in;                                                                               \
        }                                                                                                  \
    }
    DABGPU_LEVEL(3, "row_ror:8", 2 * pr, 2 * pr + 1)
    DABGPU_LEVEL(2, "row_ror:4", pr + (pr & 2), pr + (pr & 2) + 2)
    DABGPU_LEVEL(1, "quad_perm:[2,3,0,1]", pr, pr + 4)
    DABGPU_LEVEL(0, "quad_perm:[1,0,3,2]", 2 * pr, 2 * pr + 1)
#undef DABGPU_LEVEL
}
'''
    anchor = "template <int LOGN> struct Fft {"
    assert s.count(anchor) == 1
    s = s.replace(anchor, helper + "\n" + anchor, 1)
open(p, "w").write(s)
PY
make -s -C "$d/odr-dabmod_amd/csrc" -j8 > "$ROOT/tools/_variants/$v.log" 2>&1 || { tail -30 "$ROOT/tools/_variants/$v.log"; exit 1; }
cp "$d/odr-dabmod_amd/csrc/libdabgpu.so" "$ROOT/tools/_variants/libdabgpu_$v.so"
cmp -s "$ROOT/tools/_variants/libdabgpu_$v.so" "$ROOT/tools/_variants/libdabgpu_base.so" && { echo "the patch did not change the library" >&2; exit 1; }
echo "built tools/_variants/libdabgpu_$v.so"
done
