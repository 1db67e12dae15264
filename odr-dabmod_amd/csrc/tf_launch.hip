// tf_launch.hip -- host side of the frame kernel: LDS size, which fused variants exist, and the dispatch to the
// per-mode translation units (tf_inst.hip).
#include "tf_layout.h"

#include <algorithm>

namespace dabgpu {

// defined in tf_inst.hip, one per (FFT size, compile-time tap count)
hipError_t launch_tf_8_0(const TfArgs &a, unsigned flags, hipStream_t s);
hipError_t launch_tf_9_0(const TfArgs &a, unsigned flags, hipStream_t s);
hipError_t launch_tf_10_0(const TfArgs &a, unsigned flags, hipStream_t s);
hipError_t launch_tf_11_0(const TfArgs &a, unsigned flags, hipStream_t s);
hipError_t launch_tf_11_45(const TfArgs &a, unsigned flags, hipStream_t s);
hipError_t launch_tf_8_45(const TfArgs &a, unsigned flags, hipStream_t s);       // (modes II - IV: the equalised-boundary variant alone)
hipError_t launch_tf_9_45(const TfArgs &a, unsigned flags, hipStream_t s);
hipError_t launch_tf_10_45(const TfArgs &a, unsigned flags, hipStream_t s);

size_t tf_lds_bytes(int logN, unsigned flags, int nt, int overlap, int ntaps)
{
    const size_t N = (size_t)1 << logN;
    const bool eq = flags & TF_EQ;

    // the buffer scheme of the variant (tf_layout.h: the table the kernel itself reads)
    const TfVariant v = tf_variant(logN, flags & TF_FROM_BITS, flags & TF_GAIN, flags & TF_GUARD, flags & TF_FIR, nt, flags & TF_CFR,
                                   flags & TF_GVAR, tf_ofmt(flags), flags & TF_WINDOW, eq);
    const bool cfr_lean = v.cfr_lean, dbuf = v.dbuf, dual = v.dual;
    const size_t nh = v.halves ? 2 : 1;          // (two frames per workgroup: every per-frame buffer twice)
    size_t b = dual ? (N + N / 8) * 2 * sizeof(float2) : (dbuf ? 2 : 1) * (N + N / 8) * sizeof(float2);
    b += 16 * sizeof(double);
    if (flags & TF_GAIN) b += ((flags & TF_FROM_BITS) ? 1 : 6) * (N / 8) * sizeof(uint32_t);   // phase words / paired bins
    const bool wf = (flags & TF_WINDOW) && (flags & TF_FIR) && !eq;
    if (eq) b += kEqElems * sizeof(float2);                                       // windows, w, d, inverse filter, window factors
    else if ((flags & TF_FIR) && !wf) b += (4 * (nt ? nt - 1 : kBnd) + (v.bwin ? 4 : 0)) * sizeof(float2);  // 2 x [tail | next head] (+ padding)
    if (flags & TF_FROM_BITS) b += (v.pair ? 4 : 2) * ((3 * N / 4) / 16 + 2) * sizeof(uint32_t);  // staged coded bits (kBitStride)
    b *= nh;                                                                      // (everything so far belongs to a frame)
    b += ((v.bwin ? 0 : kMaxTaps) + 160) * sizeof(float) + 64 * sizeof(float2);  // taps (not where they live in registers), |y_s| table, unit vectors (8 rotations)
    b += 56 * sizeof(float2);                                      // twiddles of the stride-8 stage
    if (flags & TF_CFR) b += 6 * ((N / 8 + 63) / 64) * sizeof(float);   // cfr_red
    if (cfr_lean) b += 3 * (N / 8) * 2 * sizeof(uint32_t);               // bit positions of the lanes' carriers
    if (wf) {
        // behind the twiddle table: two stashes, the next symbol's samples, the windowed stream, the window
        const size_t C = (size_t)std::max(ntaps - 1, 0), W = (size_t)std::max(overlap, 0);
        b += (2 * (C + 2 * W) + (2 * W + C) + (2 * W + 2 * C)) * sizeof(float2) + 2 * W * sizeof(float) + 16;
    } else if ((flags & TF_WINDOW) && !eq) {
        b += 7 * kWinMax * sizeof(float2);                                     // seam buffers + window
    }
    return b;
}

// the frame-kernel variants that window the guard interval themselves: coded-bits chain with guard interval and
// without s16 store (with or without FIRFilter, with or without CFR), overlap up to kWinMax (and inside the cyclic prefix)
bool tf_has_window(const TfArgs &a, unsigned flags)
{
    // (which of them store an integer format: tf_has_fmt)
    const unsigned want = TF_FROM_BITS | TF_GUARD;
    if ((flags & want) != want || a.overlap < 1 || a.overlap > kWinMax) return false;
    // with FIR: the filter's look-ahead and the window must both fit into the cyclic prefix
    if (flags & TF_FIR)
        return a.ntaps >= 1 && a.ntaps <= kBnd && a.ntaps <= kMaxTaps &&
               a.overlap + a.ntaps - 1 <= a.g.sym_size - a.g.N;
    return a.overlap <= a.g.sym_size - a.g.N;
}

int tf_max_fused_taps() { return kBnd < kMaxTaps ? kBnd : kMaxTaps; }

// modes II - IV: the chains built with the compile-time tap count (tf_kernel.h: launch_tf_small45) -- coded bits, guard interval,
// the 45-tap filter, complexf output, no CFR, no windowing; with or without GainControl, equalised (TF_EQ) or not
bool tf_small45(const TfArgs &a, unsigned flags)
{
    const unsigned want = TF_FROM_BITS | TF_GUARD | TF_FIR;
    // (Mode III runs two frames per wave there, with the gain statistic per half-wave: not the wave-wide maximum of gain mode max)
    if (a.g.logN == 8 && (flags & TF_GAIN) && a.gain.mode == 1) return false;
    return a.g.logN >= 8 && a.g.logN <= 10 && a.ntaps == 45 && (flags & ~(unsigned)(TF_GAIN | TF_EQ)) == want;
}

// the equalised-boundary variant: the chains of the pruned-dual-transform variant, given the inverse of the taps; with OFDM
// windowing (TF_WINDOW) for overlaps up to kEqWinMax
bool tf_has_eq(const TfArgs &a, unsigned flags)
{
    const unsigned want = TF_FROM_BITS | TF_GUARD | TF_FIR;
    if ((flags & TF_WINDOW) && (a.overlap < 1 || a.overlap > kEqWinMax)) return false;
    // Mode IV (round 6): complexf output, no windowing -- the form tf_inst_10_45.o holds.  Not modes II and III: the inverse
    // filter (44 x 160) and the correction (44 x 45 / 2) are the same work per symbol whatever its length, 16 k multiply-adds,
    // and the second half of a packed transform of 512 or 256 points plus the direct boundary filter is LESS than that
    // (measured, Mode II: 2.70 ms per 16384 frames equalised, 2.58 ms as a packed pair; profiles/r06_small_modes.txt).
    if (a.g.logN != 11 && ((flags & TF_WINDOW) || tf_ofmt(flags))) return false;
    return a.t.eq_g != nullptr && a.g.logN >= 10 && a.g.logN <= 11 && a.ntaps == 45 && (flags & want) == want && !(flags & TF_CFR) &&
           (!(flags & TF_GAIN) || a.gain.mode != 1);
}

// TII inside the frame kernel: the coded-bits chain with guard interval, without windowing and CFR, ending in the equalised
// variant (TF_EQ set by the caller) or without FIRFilter; a workgroup that owns the null symbol must own symbol 1 as well
// (its multiplier is the null symbol's)
bool tf_has_tii(const TfArgs &a, unsigned flags)
{
    const unsigned want = TF_FROM_BITS | TF_GUARD;
    if ((flags & want) != want || a.syms_per_chunk < 2) return false;
    if (flags & TF_WINDOW) return (flags & TF_EQ) && tf_has_eq(a, flags);     // windowed: the equalised-boundary form alone
    // every other form with the guard interval (round 5): with or without FIRFilter (the segment's last ntaps - 1 samples ride on
    // the null symbol's boundary outputs), with or without CFR (the cached segment is the CFR'd null symbol)
    return (flags & TF_EQ) ? tf_has_eq(a, flags) : true;
}

// the frame-kernel variants that store an integer format (flags' TF_OUT_* bit) themselves: the Mode I coded-bits chain with the
// guard interval -- s16 on every form that is one kernel (with or without FIRFilter, CFR, a windowed guard interval without
// FIRFilter), u8 / s8 without FIRFilter and on the equalised-boundary form (TF_EQ set by the caller)
bool tf_has_fmt(const TfArgs &a, unsigned flags)
{
    const int of = tf_ofmt(flags);
    if (!of || a.g.logN != 11 || (flags & (TF_FROM_BITS | TF_GUARD)) != (TF_FROM_BITS | TF_GUARD)) return false;
    if (flags & TF_WINDOW) {
        // windowed: every format on the equalised-boundary form, s16 on the chain without FIRFilter
        if (flags & TF_CFR) return false;
        return (flags & TF_FIR) ? ((flags & TF_EQ) && tf_has_eq(a, flags)) : of == 1;      // (1 = s16)
    }
    if (flags & TF_CFR) return of == 1;       // crest-factor reduction, with or without FIRFilter: s16
    // without FIRFilter (the reference's default): every gain mode
    if (!(flags & TF_FIR)) return true;
    // with FIRFilter: s16 on every form (equalised, pruned, packed dual transform; any tap count the fused filter takes, any gain
    // mode -- round 5), u8 / s8 on the equalised-boundary form
    if (of == 1) return true;
    return (flags & TF_EQ) && a.ntaps == 45 && (!(flags & TF_GAIN) || a.gain.mode != 1);
}

hipError_t launch_tf(const TfArgs &a, unsigned flags, hipStream_t s)
{
    if (flags & TF_FIR) {
        // the fused (spectral) FIR needs its look-ahead to fit in a cyclic prefix
        const int C = a.ntaps - 1;
        if (a.ntaps < 1 || a.ntaps > kMaxTaps || a.ntaps > kBnd || C > a.g.sym_size - a.g.N)
            return hipErrorInvalidValue;
    }
    switch (a.g.logN) {
        // Mode I with the default filter length gets the compile-time tap count
        case 8: return tf_small45(a, flags) ? launch_tf_8_45(a, flags, s) : launch_tf_8_0(a, flags, s);
        case 9: return tf_small45(a, flags) ? launch_tf_9_45(a, flags, s) : launch_tf_9_0(a, flags, s);
        case 10: return tf_small45(a, flags) ? launch_tf_10_45(a, flags, s) : launch_tf_10_0(a, flags, s);
        case 11:
            return ((flags & TF_FIR) && a.ntaps == 45 && !((flags & TF_CFR) && (flags & TF_WINDOW))) ? launch_tf_11_45(a, flags, s)
                                                       : launch_tf_11_0(a, flags, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace dabgpu
