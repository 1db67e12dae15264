// tf_layout.h -- LDS layout constants of the frame kernel (tf_kernel.h) that its host-side size computation
// (tf_launch.hip: tf_lds_bytes) has to agree with.
#pragma once
#include "dabgpu_internal.h"

namespace dabgpu {
// LDS of the equalised-boundary variants (see tf_kernel), in complex slots: two windows of the previous filtered symbol and
// their difference against the current one (kEqWLen each), the unfiltered differences d (kEqDLen), the inverse filter
// (kEqTaps + 8 floats), the raised-cosine factors of the windowed form (2 kEqWinMax floats)
constexpr int kEqWinMax = 10;      // widest overlap the equalised-boundary variant windows itself: 2 x 10 + 44 = 64 boundary outputs
constexpr int kEqWLen = 240, kEqDLen = 176;    // (d: 80 used by the windowed form; 2 x 44 differences + 2 x 44 tails by the pipelined boundary)
constexpr int kEqElems = 3 * kEqWLen + kEqDLen + (kEqTaps + 8) / 2 + 16;
constexpr int kWinMax = 128;       // widest raised-cosine overlap the frame kernel applies itself (TF_WINDOW)
constexpr int kBnd = 128;          // LDS slots per boundary buffer; the fused FIR handles ntaps <= kBnd

// ---- ONE place for what a frame-kernel variant is built like --------------------------------------------------------------
// The kernel's __launch_bounds__ and buffer layout (tf_kernel.h) and the host's LDS size (tf_launch.hip: tf_lds_bytes) read
// the same table; nothing else restates these conditions.  Arguments: the kernel's template arguments (nt = the COMPILE-TIME
// tap count of the instantiation, 0 = run-time).
struct TfVariant {
    bool cfr_lean;      // Mode I coded-bits CFR chains with the guard interval (no FIRFilter, or the default-length one): built for
                        // four waves per SIMD -- one exchange buffer, loop invariants re-derived or parked in LDS
    bool cfr_seq;       // ... with FIRFilter: the corrected spectrum's two inverse transforms one after the other, not as a packed pair
    bool nofir_1buf;    // the reference's default chain (Mode I from coded bits, no FIRFilter): one exchange buffer, five waves per SIMD
    bool dbuf;          // two exchange buffers (one barrier per exchange) -- the variants without FIRFilter that are not one of the above
    bool dual;          // unfiltered + filtered transform of a symbol as ONE packed transform: 16-byte exchange elements
    bool halves;        // Mode III, the plain coded-bits chain with the default-length filter (round 6): the two halves of the one wave
                        // work on two frames -- every per-frame LDS buffer twice, blockIdx counts pairs of frames
    bool bwin;          // modes II - IV, the packed dual transform with the default-length filter (round 6): the boundary filter works
                        // from a register window (four outputs x twelve taps per lane); its sample buffers carry four slots of padding
    bool pair;          // Mode I equalised-boundary kernels (round 6): the coded bits are prefetched two symbols' blocks at a time -- four
                        // block slots in LDS instead of two -- so that the wait behind the previous iteration's stores comes every other symbol
    int waves_per_simd; // asked of the register allocator (HIP: the second __launch_bounds__ argument)
};
constexpr TfVariant tf_variant(int logn, bool from_bits, bool gain, bool guard, bool fir, int nt, bool cfr, bool gvar,
                               int ofmt, bool win, bool eq)
{
    TfVariant v{};
    v.cfr_lean = cfr && logn == 11 && from_bits && guard && !win && (!fir || nt == 45);
    v.cfr_seq = v.cfr_lean && fir;
    v.nofir_1buf = logn == 11 && from_bits && guard && !fir && !cfr && (!win || ofmt != 0);
    v.dbuf = !fir && !v.cfr_lean && !v.nofir_1buf;
    v.dual = fir && !eq && !v.cfr_seq;
    v.bwin = logn != 11 && fir && nt == 45 && !cfr && !win && !eq;
    v.halves = logn == 8 && from_bits && guard && fir && nt == 45 && !cfr && !gvar && ofmt == 0 && !win && !eq;
    v.pair = eq && logn == 11;
    // EQ: 4 (128 VGPRs, 29 KB of LDS: four workgroups per CU); modes II - IV: 3 (their one- and two-wave workgroups are limited by
    //   the windows' LDS before that, and at 128 registers they spill)
    // CFR: 4 where it is built lean, 3 on the other coded-bits chains with the guard interval that keep one transform live, else 2
    // no FIRFilter: the default chain 5 (windowed, or the s8 store without GainControl: 4), everything else 2
    // FIRFilter: 3 (<= 168 VGPRs, 42 KB of LDS) -- except the carriers-input variants with time-domain gain statistics: 2 (both
    //   transforms of a symbol stay live: 256 VGPRs instead of spilling at 168)
    v.waves_per_simd = eq ? (logn == 11 ? 4 : 3)
                       : cfr ? (v.cfr_lean ? 4 : ((from_bits && guard && !(win && fir)) ? 3 : 2))
                       : !fir ? (v.nofir_1buf ? ((win || (ofmt == 3 && !gain)) ? 4 : 5) : 2)
                       : (gvar ? 3 : ((gain && !from_bits) ? 2 : 3));
    return v;
}
}  // namespace dabgpu
