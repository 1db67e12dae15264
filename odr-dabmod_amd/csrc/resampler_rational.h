// resampler_rational.h -- a10 Resampler for every ratio L / M the reference can run on whole frames (M a power of two
// up to the FFT size; up- and down-sampling): resampler_rational_kernel, resampler_lane_kernel and their launchers.
#pragma once
#include "dabgpu_internal.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace dabgpu {
namespace {

// ---------------------------------------------------------------------------
// a10 for rational ratios L / M (M a power of two dividing nin, L > M): nout = S L with S = nin / M.
// Same overlap-add on the spectra as above (G_h = F_h + (-1)^k F_{h-1}, first half of the output only).
// The zero-stuffed nout-point IDFT factors over n = L j + p and stuffed bin k' + S r:
//     y_p[j] = IDFT_S over k' of  W_nout^{k' p} * sum_r Gst[k' + S r] e^{2 pi i r p / L},
// and only M (+1, Nyquist) of the L values of r are occupied: bin k = k' + S rho of the nin-point spectrum
// sits at r = rho (k < nin/2), r = rho + L - M (k > nin/2), or both (k = nin/2).  So a hop is one nin-point
// forward transform, then L "branches": an M-term fold per bin and an S-point IFFT.  The nin / 8 lanes of the
// workgroup split into M groups of S / 8 lanes, one branch per group and round, ceil(L / M) rounds.
// General, not tuned: the integer ratios 2 and 4 keep the kernel above.
// DOWN (L < M, nout < nin; src/Resampler.cpp:165-177): the spectrum is truncated instead of zero-stuffed -- the bins
// nout/2 < k < nin - nout/2 are dropped, and bin nout/2 of the output spectrum is the average of the input bins nout/2
// and nin - nout/2.  Input bin k' + S rho then sits at r = rho (k < nout/2) or r = rho - (M - L) (k > nin - nout/2):
// the same fold with per-bin weights 1, 1/2 or 0, every one of the L values of r occupied.
template <int LOGNIN, int LOGS, bool DOWN = false> __global__ __launch_bounds__((1 << LOGNIN) / 8 < 64 ? 64 : (1 << LOGNIN) / 8)
void resampler_rational_kernel(const ResamplerArgs a, int hops_per_run, int cl_in_lds)
{
    typedef Fft<LOGNIN> F;
    typedef Fft<LOGS> FS;
    constexpr int NIN = F::N, T = F::T, HIN = NIN / 2, S = FS::N, TS = FS::T, M = NIN / S;
    static_assert(M >= 1 && TS >= 1, "group geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf *xbuf = reinterpret_cast<cf *>(smem);                 // exchange buffers: 2 x LDS_ELEMS of the nin-point
                                                             // transform = M groups x 2 x LDS_ELEMS of the S-point one
    cf *gl = xbuf + 2 * F::LDS_ELEMS;                        // G: the nin bins of the hop's spectrum
    cf *fprev = gl + NIN;                                    // F of the previous hop (each lane reads and writes its own bins)
    float *win = reinterpret_cast<float *>(fprev + NIN);     // first half of the symmetric Hann window
    cf *cl_l = reinterpret_cast<cf *>(win + HIN);            // exp(2 pi i m / L), m < L -- when it fits (cl_in_lds)
    const int L = a.L, nout = a.nout, HOUT = nout / 2;
    int fpar = 0, spar = 0;
    const int t = threadIdx.x;
    const bool lane_on = t < T;                              // nin = 256 would leave half a wave idle (not used)
    const int tt = lane_on ? t : 0;
    const long h0 = (long)blockIdx.x * hops_per_run;
    const long h1 = min((long)a.nhops, h0 + hops_per_run);
    if (h0 >= (long)a.nhops) return;

    const int grp = tt / TS, l = tt % TS;                    // branch group and lane inside it
    if (cl_in_lds)
        for (int i = t; i < L; i += blockDim.x) cl_l[i] = a.tw_l[i];
#pragma unroll
    for (int m = 0; m < 4; ++m) win[tt + T * m] = a.window[tt + T * m];
    lds_barrier();
    auto cl = [&](int i) __attribute__((always_inline)) -> cf { return cl_in_lds ? cl_l[i] : a.tw_l[i]; };
    auto wnd = [&](int m) __attribute__((always_inline)) -> float {
        return m < 4 ? win[tt + T * m] : win[T * (7 - m) + (T - 1 - tt)];
    };
    // nin-point forward transform of windowed hop h (conjugate trick: DFT(x) = conj(IDFT(conj(x)))).  The resident
    // twiddles of both transform sizes (up to 42 registers each) are NOT kept across the hop loop -- that is what
    // used to push this kernel into scratch: they are re-read from the (L2-resident) table right before each use,
    // behind an opaque copy of the lane index so that the loads cannot be hoisted back out of the loop.
    auto forward = [&](long h, cf *f) __attribute__((always_inline)) {
        cf v[8];
        const long base = (h + 1) * HIN;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const long i = base + tt + T * m;
            const cf x = i < NIN ? a.halo[i] : a.in[i - NIN];
            const float w = wnd(m);
            v[m] = mk(x.x * w, -x.y * w);
        }
        int to = tt;
        asm volatile("" : "+v"(to));
        cf tw[F::NTW > 0 ? F::NTW : 1];
        F::template load_twiddles<false>(a.tw_in, to, tw);
        F::template run<+1, true, cf, false>(v, xbuf, fpar, tw, tt);
#pragma unroll
        for (int m = 0; m < 8; ++m) f[m] = mk(v[m].x * a.factor, -v[m].y * a.factor);
    };
    const float sgn = (tt & 1) ? -1.0f : 1.0f;
    {
        cf f0[8];
        forward(h0 - 1, f0);
#pragma unroll
        for (int m = 0; m < 8; ++m) fprev[tt + T * m] = f0[m];
    }
    cf *gbuf = xbuf + (size_t)grp * 2 * FS::LDS_ELEMS;       // this group's pair of exchange buffers
    const int rounds = (L + M - 1) / M;

    for (long h = h0; h < h1; ++h) {
        {
            cf Fc[8];
            forward(h, Fc);
            lds_barrier();                                   // the previous hop's folds have read gl
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const cf fp = fprev[tt + T * m];
                if (lane_on) {
                    gl[tt + T * m] = mk(fmaf(sgn, fp.x, Fc[m].x), fmaf(sgn, fp.y, Fc[m].y));
                    fprev[tt + T * m] = Fc[m];
                }
            }
        }
        lds_barrier();
        cf *dst = a.out + (size_t)h * HOUT;
        for (int q = 0; q < rounds; ++q) {
            const int p = q * M + grp;                       // this group's branch (may run past L: computed, not stored)
            const int pe = p < L ? p : 0;
            const int off = (int)(((long)M * pe) % L);       // (r p) mod L for r = rho + L - M is (rho p - M p) mod L
            // the fold: the weights depend on (p, rho) only, so rho is the outer loop and the lane's eight bins
            // k' = l + TS i share them
            cf v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = mk(0.f, 0.f);
            int idx = 0;                                     // (rho p) mod L
            if constexpr (DOWN) {
                const int hout = nout / 2, hi0 = NIN - hout;                 // live bands: k <= hout and k >= hi0
#pragma unroll 1
                for (int rho = 0; rho < M; ++rho) {
                    const bool low = S * rho <= hout, high = S * (rho + 1) - 1 >= hi0;
                    if (low || high) {                                       // (a block is never both: L < M)
                        int in2 = idx - off;
                        in2 += in2 < 0 ? L : 0;
                        const cf c = low ? cl(idx) : cl(in2);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int k = l + TS * i + S * rho;
                            const float w = low ? (k < hout ? 1.0f : (k == hout ? 0.5f : 0.0f))
                                                : (k > hi0 ? 1.0f : (k == hi0 ? 0.5f : 0.0f));
                            v[i] = cadd(v[i], cscale(cmul(gl[k], c), w));
                        }
                    }
                    idx += pe;
                    idx -= idx >= L ? L : 0;
                }
            } else {
                // (kept rolled: M is a compile-time constant, and the fully unrolled fold of M = 8 ... 32 blocks is
                // what drove some instantiations into scratch)
#pragma unroll 1
                for (int rho = 0; rho < M; ++rho) {
                    int in2 = idx - off;
                    in2 += in2 < 0 ? L : 0;
                    const cf cpos = cl(idx), cneg = cl(in2);
                    if (M == 1) {
                        // one term per bin: the half of the spectrum decides, and the Nyquist bin gets both
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int k = l + TS * i;
                            cf c = k < HIN ? cpos : cneg;
                            if (k == HIN) c = cadd(cpos, cneg);
                            v[i] = cmul(gl[k], c);
                        }
                    } else {
                        const cf c = rho < M / 2 ? cpos : cneg;  // k = k' + S rho < nin/2  <=>  rho < M/2
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = cadd(v[i], cmul(gl[l + TS * i + S * rho], c));
                        // the Nyquist bin (k' = 0, rho = M/2) sits at -nin/2 (above) and at +nin/2 as well
                        if (rho == M / 2 && l == 0) v[0] = cadd(v[0], cmul(gl[HIN], cpos));
                    }
                    idx += pe;
                    idx -= idx >= L ? L : 0;
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = cmul(v[i], a.tw_out[(int)(((long)(l + TS * i) * pe) % nout)]);
            {
                int lo = l;
                asm volatile("" : "+v"(lo));
                cf tws[FS::NTW > 0 ? FS::NTW : 1];
                FS::template load_twiddles<false>(a.tw_s, lo, tws);
                FS::template run<+1, true, cf, false>(v, gbuf, spar, tws, l);
            }
            if (p < L && lane_on) {
#pragma unroll
                for (int m = 0; m < 4; ++m) dst[(size_t)L * (l + TS * m) + p] = v[m];   // j = l + TS m < S / 2
            }
        }
        // the groups' exchange buffers and the nin-point transform's share the same memory under different
        // layouts: nobody may still be reading the former when the next hop starts writing the latter
        lds_barrier();
    }
}

// ---------------------------------------------------------------------------
// a10, the remaining ratios: M = nin / 2 or nin / 4 (S = 2 or 4 point branches; e.g. 2 048 000 -> 2 049 000).
// A branch is then too small for a group of lanes: ONE lane owns branch p -- it folds every occupied bin of the
// output spectrum onto the S residues (sum over r of B[k' + S r] e^{2 pi i (k' + S r) p / nout}, the phase read from
// the nout-entry table at an index that advances by S p mod nout), finishes with the S-point IDFT in registers and
// stores the first S / 2 outputs.  Up- and down-sampling share the code: only the map from output bin to input bin
// differs (zero-stuffing with the Nyquist bin on both sides, or truncation with the averaged Nyquist bin).
template <int LOGNIN, int LOGS> __global__ __launch_bounds__((1 << LOGNIN) / 8 < 64 ? 64 : (1 << LOGNIN) / 8)
void resampler_lane_kernel(const ResamplerArgs a, int hops_per_run)
{
    typedef Fft<LOGNIN> F;
    constexpr int NIN = F::N, T = F::T, HIN = NIN / 2, S = 1 << LOGS;
    static_assert(S == 2 || S == 4, "branches of 2 or 4 points");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf *xbuf = reinterpret_cast<cf *>(smem);
    cf *gl = xbuf + 2 * F::LDS_ELEMS;
    cf *fprev = gl + NIN;
    float *win = reinterpret_cast<float *>(fprev + NIN);
    const int L = a.L, nout = a.nout, HOUT = nout / 2;
    const bool down = nout < NIN;
    int fpar = 0;
    const int t = threadIdx.x;
    const bool lane_on = t < T;
    const int tt = lane_on ? t : 0;
    const long h0 = (long)blockIdx.x * hops_per_run;
    const long h1 = min((long)a.nhops, h0 + hops_per_run);
    if (h0 >= (long)a.nhops) return;
#pragma unroll
    for (int m = 0; m < 4; ++m) win[tt + T * m] = a.window[tt + T * m];
    lds_barrier();
    auto wnd = [&](int m) __attribute__((always_inline)) -> float {
        return m < 4 ? win[tt + T * m] : win[T * (7 - m) + (T - 1 - tt)];
    };
    auto forward = [&](long h, cf *f) __attribute__((always_inline)) {
        cf v[8];
        const long base = (h + 1) * HIN;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const long i = base + tt + T * m;
            const cf x = i < NIN ? a.halo[i] : a.in[i - NIN];
            const float w = wnd(m);
            v[m] = mk(x.x * w, -x.y * w);
        }
        int to = tt;
        asm volatile("" : "+v"(to));
        cf tw[F::NTW > 0 ? F::NTW : 1];
        F::template load_twiddles<false>(a.tw_in, to, tw);
        F::template run<+1, true, cf, false>(v, xbuf, fpar, tw, tt);
#pragma unroll
        for (int m = 0; m < 8; ++m) f[m] = mk(v[m].x * a.factor, -v[m].y * a.factor);
    };
    const float sgn = (tt & 1) ? -1.0f : 1.0f;
    {
        cf f0[8];
        forward(h0 - 1, f0);
#pragma unroll
        for (int m = 0; m < 8; ++m) fprev[tt + T * m] = f0[m];
    }
    // the occupied blocks r of the output spectrum (k_out = k' + S r): all of them when down-sampling; the two ends
    // when up-sampling -- [0, (nin/2)/S] and [L - (nin/2)/S, L)
    const int rA1 = down ? L : HIN / S + 1, rB0 = down ? L : L - HIN / S;
    for (long h = h0; h < h1; ++h) {
        {
            cf Fc[8];
            forward(h, Fc);
            lds_barrier();
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const cf fp = fprev[tt + T * m];
                if (lane_on) {
                    gl[tt + T * m] = mk(fmaf(sgn, fp.x, Fc[m].x), fmaf(sgn, fp.y, Fc[m].y));
                    fprev[tt + T * m] = Fc[m];
                }
            }
        }
        lds_barrier();
        cf *dst = a.out + (size_t)h * HOUT;
        for (int p = t; p < L; p += (int)blockDim.x) {
            const int step = (int)(((long)S * p) % nout);
            // (sums of up to nin / S = 2048 terms: accumulated in float64 -- the fp32 running sum alone cost 7.7e-7 of
            // the 1e-6 budget; the products stay fp32)
            cf acc[S];
#pragma unroll
            for (int kk = 0; kk < S; ++kk) {
                double ar = 0.0, ai = 0.0;
                // one output-spectrum bin: its source in G (weight 1), the shared Nyquist bin, or nothing
                auto term = [&](int r, int idx) __attribute__((always_inline)) {
                    const int ko = kk + S * r;
                    cf g;
                    if (!down) {
                        if (ko < HIN) g = gl[ko];
                        else if (ko == HIN || ko == nout - HIN) g = gl[HIN];
                        else if (ko > nout - HIN) g = gl[ko - (nout - NIN)];
                        else return;
                    } else {
                        if (ko < HOUT) g = gl[ko];
                        else if (ko == HOUT) g = cscale(cadd(gl[HOUT], gl[NIN - HOUT]), 0.5f);
                        else g = gl[ko + (NIN - nout)];
                    }
                    const cf pr = cmul(g, a.tw_out[idx]);
                    ar += (double)pr.x;
                    ai += (double)pr.y;
                };
                int idx = (int)(((long)kk * p) % nout);
                for (int r = 0; r < rA1; ++r) {
                    term(r, idx);
                    idx += step;
                    idx -= idx >= nout ? nout : 0;
                }
                if (rB0 < L) {
                    idx = (int)((((long)kk + (long)S * rB0) * p) % nout);
                    for (int r = rB0 < rA1 ? rA1 : rB0; r < L; ++r) {
                        term(r, idx);
                        idx += step;
                        idx -= idx >= nout ? nout : 0;
                    }
                }
                acc[kk] = mk((float)ar, (float)ai);
            }
            // S-point IDFT, outputs j < S / 2 -> samples L j + p
            if (S == 2) {
                dst[p] = cadd(acc[0], acc[1]);
            } else {
                dst[p] = cadd(cadd(acc[0], acc[2]), cadd(acc[1], acc[3 % S]));
                dst[(size_t)L + p] = cadd(csub(acc[0], acc[2]), mul_i<+1>(csub(acc[1], acc[3 % S])));
            }
        }
        lds_barrier();
    }
}

// LDS of the general kernels: two exchange buffers, G, the previous hop's F, half the window, and -- when it fits
// next to them in 160 KiB -- the L-th roots of unity
template <int LOGNIN> size_t rational_lds_bytes(int L, int *cl_in_lds)
{
    constexpr size_t NIN = (size_t)1 << LOGNIN;
    const size_t base = (2 * (NIN + NIN / 8) + 2 * NIN) * sizeof(float2) + (NIN / 2) * sizeof(float);
    const bool fits = base + (size_t)L * sizeof(float2) <= 160 * 1024;
    if (cl_in_lds) *cl_in_lds = fits ? 1 : 0;
    return base + (fits ? (size_t)L * sizeof(float2) : 0);
}

template <typename K> hipError_t allow_lds(K kernel, size_t lds)
{
    // more than 64 KiB of dynamic LDS has to be asked for
    return lds > 64 * 1024 ? hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                           : hipSuccess;
}

template <int LOGNIN, int LOGS> hipError_t launch_resampler_rational(const ResamplerArgs &a, hipStream_t s)
{
    constexpr int NIN = 1 << LOGNIN;
    const int hpr = (int)std::max<size_t>(2, std::min<size_t>(96, a.nhops / 512));
    const dim3 grid((unsigned)((a.nhops + hpr - 1) / hpr)), block(NIN / 8 < 64 ? 64 : NIN / 8);
    int cl_in_lds = 0;
    const size_t lds = rational_lds_bytes<LOGNIN>(a.L, &cl_in_lds);
    hipError_t e;
    if (a.nout < a.nin) {
        if ((e = allow_lds(resampler_rational_kernel<LOGNIN, LOGS, true>, lds)) != hipSuccess) return e;
        DABGPU_LAUNCH((resampler_rational_kernel<LOGNIN, LOGS, true>), grid, block, lds, s, a, hpr, cl_in_lds);
    } else {
        if ((e = allow_lds(resampler_rational_kernel<LOGNIN, LOGS, false>, lds)) != hipSuccess) return e;
        DABGPU_LAUNCH((resampler_rational_kernel<LOGNIN, LOGS, false>), grid, block, lds, s, a, hpr, cl_in_lds);
    }
    return hipGetLastError();
}

template <int LOGNIN, int LOGS> hipError_t launch_resampler_lane(const ResamplerArgs &a, hipStream_t s)
{
    constexpr int NIN = 1 << LOGNIN;
    const int hpr = (int)std::max<size_t>(2, std::min<size_t>(96, a.nhops / 512));
    const dim3 grid((unsigned)((a.nhops + hpr - 1) / hpr)), block(NIN / 8 < 64 ? 64 : NIN / 8);
    const size_t lds = rational_lds_bytes<LOGNIN>(0, nullptr);
    hipError_t e = allow_lds(resampler_lane_kernel<LOGNIN, LOGS>, lds);
    if (e != hipSuccess) return e;
    DABGPU_LAUNCH((resampler_lane_kernel<LOGNIN, LOGS>), grid, block, lds, s, a, hpr);
    return hipGetLastError();
}

// S = nin / M: groups of S / 8 lanes per branch for S >= 8, one lane per branch below
template <int LOGNIN, int LOGS> hipError_t launch_resampler_rational_s(const ResamplerArgs &a, int logs, hipStream_t s)
{
    if constexpr (LOGS < 3) {
        return hipErrorInvalidValue;
    } else {
        if (logs == LOGS) return launch_resampler_rational<LOGNIN, LOGS>(a, s);
        return launch_resampler_rational_s<LOGNIN, LOGS - 1>(a, logs, s);
    }
}

template <int LOGNIN> hipError_t launch_resampler_rational_n(const ResamplerArgs &a, hipStream_t s)
{
    const int S = a.nin / a.M;
    int logs = 0;
    while ((1 << logs) < S) ++logs;
    if ((1 << logs) != S || a.M * S != a.nin) return hipErrorInvalidValue;
    if (logs == 1) return launch_resampler_lane<LOGNIN, 1>(a, s);
    if (logs == 2) return launch_resampler_lane<LOGNIN, 2>(a, s);
    return launch_resampler_rational_s<LOGNIN, LOGNIN>(a, logs, s);
}


}  // namespace
}  // namespace dabgpu
