// resampler.hip -- a10 Resampler + a11 MemlessPoly: the x2 / x4 kernel of BASELINE config 4 (resampler_kernel) and the
// dispatch of every ratio (the general kernels live in resampler_rational.h, one translation unit per FFT size).
#include "device_common.h"

namespace dabgpu {

// defined in resampler_rational_inst.hip, one per input FFT size
hipError_t launch_resampler_rational_9(const ResamplerArgs &a, hipStream_t s);
hipError_t launch_resampler_rational_10(const ResamplerArgs &a, hipStream_t s);
hipError_t launch_resampler_rational_11(const ResamplerArgs &a, hipStream_t s);
hipError_t launch_resampler_rational_12(const ResamplerArgs &a, hipStream_t s);

namespace {


// ===========================================================================
// a10 Resampler (src/Resampler.cpp:131-195), up-sampling by Q = nout/nin.
//
// Stateless restatement: out_h = second_half(Y_{h-1}) + first_half(Y_h),
// Y_h = IDFT_nout( stuff( DFT_nin( w * [c_{h-1} | c_h] ) ) * factor ).
// The zero-stuffed nout-point IDFT is never formed: because only the nin lowest
// |frequencies| are occupied, Y[Q q + p] = IDFT_nin_k( F[k] * W_nout^{kappa(k) p} )
// with kappa the signed frequency of bin k -- Q independent nin-point IFFTs of
// the same spectrum under a per-branch twiddle (the Nyquist bin, which the
// reference places at both +nin/2 and -nin/2, gets the sum of both twiddles).
// A workgroup walks a run of consecutive hops; the overlap-add tail (second
// half of Y) never leaves registers: lane t produces q = t + T m in every hop,
// m < 4 being the first half and m >= 4 the tail.
// MemlessPoly polynomial (reference src/MemlessPoly.cpp:237-276) on one sample; shared by the
// stand-alone kernel and the resampler's fused epilogue.
struct PolyCoef { float a0, a1, a2, a3, a4, p0, p1, p2, p3, p4; };
DEV cf poly_apply(cf x, const PolyCoef &c)
{
    const float m = x.x * x.x + x.y * x.y;
    const float a = c.a0 + m * (c.a1 + m * (c.a2 + m * (c.a3 + m * c.a4)));
    const float p = -1.0f * (c.p0 + m * (c.p1 + m * (c.p2 + m * (c.p3 + m * c.p4))));
    const float q = p * p;
    const float cr = (1.0f - q * (-0.5f + q * (0.486666f + q * (-0.00138888f))));
    const float ci = p * (1.0f + q * (0.166666f + q * (0.00833333f)));
    const float sr = x.x * a, si = x.y * a;
    return mk(sr * cr - si * ci, sr * ci + si * cr);
}
// two samples at a time: every operation is a packed fp32 instruction (v_pk_fma_f32 / v_pk_mul_f32)
DEV void poly_apply2(cf &s0, cf &s1, const PolyCoef &c)
{
    // (every multiply-add spelled as an explicit packed FMA: see pk_fma)
    auto k = [](float v) __attribute__((always_inline)) { return make_float2(v, v); };
    const float2 x = make_float2(s0.x, s1.x), y = make_float2(s0.y, s1.y);
    const float2 m = pk_fma(x, x, y * y);
    const float2 a = pk_fma(m, pk_fma(m, pk_fma(m, pk_fma(m, k(c.a4), k(c.a3)), k(c.a2)), k(c.a1)), k(c.a0));
    const float2 p = pk_neg(pk_fma(m, pk_fma(m, pk_fma(m, pk_fma(m, k(c.p4), k(c.p3)), k(c.p2)), k(c.p1)), k(c.p0)));
    const float2 q = p * p;
    const float2 cr = pk_fma(pk_neg(q), pk_fma(q, pk_fma(q, k(-0.00138888f), k(0.486666f)), k(-0.5f)), k(1.0f));
    const float2 ci = p * pk_fma(q, pk_fma(q, k(0.00833333f), k(0.166666f)), k(1.0f));
    const float2 sr = x * a, si = y * a;
    const float2 re = pk_fma(sr, cr, pk_neg(si * ci)), im = pk_fma(sr, ci, si * cr);
    s0 = mk(re.x, im.x);
    s1 = mk(re.y, im.y);
}

// S16: FormatConverter fused into the store (4-byte s16 pairs, clipped components counted into *a.clipped)
template <int LOGNIN, int Q, bool POLY, bool S16 = false> __global__ __launch_bounds__((1 << LOGNIN) / 8)
void resampler_kernel(const ResamplerArgs a, int hops_per_run)
{
    unsigned nclip = 0;
    typedef Fft<LOGNIN> F;
    constexpr int NIN = F::N, T = F::T, HIN = NIN / 2, HOUT = HIN * Q, NOUT = NIN * Q;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // two exchange buffers of 16-byte elements (packed dual transforms, one barrier per exchange)
    c2 *fbuf2 = reinterpret_cast<c2 *>(smem);
    cf *nyq = reinterpret_cast<cf *>(fbuf2 + 2 * F::LDS_ELEMS);   // [2]: Nyquist bin per hop parity
    cf *tw8_l = nyq + 2;                                           // 7 x 8 twiddles of the stride-8 stage
    // first half of the (symmetric) Hann window; w[i] = w[NIN-1-i] serves the second half
    float *win = reinterpret_cast<float *>(tw8_l + 56);
    int fpar = 0;
    const int t = threadIdx.x;
    const long h0 = (long)blockIdx.x * hops_per_run;
    const long h1 = min((long)a.nhops, h0 + hops_per_run);
    if (h0 >= (long)a.nhops) return;

    cf tw[F::NTW];
    F::template load_twiddles<true>(a.tw_in, t, tw);
    F::fill_tw8(a.tw_in, tw8_l, t);
#pragma unroll
    for (int m = 0; m < 4; ++m) win[t + T * m] = a.window[t + T * m];
    auto wnd = [&](int m) __attribute__((always_inline)) -> float {
        return m < 4 ? win[t + T * m] : win[T * (7 - m) + (T - 1 - t)];
    };
    // per-branch twiddle of bin k = t + T m:  W_nout^{kappa p} = W_nout^{t p} * e^{2 pi i m p / (8Q)}
    // (* (-i)^p for the negative-frequency half, kappa = k - NIN): one table value per branch
    // and lane, the rest are compile-time rotations.
    cf wp[Q];
#pragma unroll
    for (int p = 1; p < Q; ++p) wp[p] = a.tw_out[(t * p) & (NOUT - 1)];
    PolyCoef pc{};
    if (POLY) {
        pc.a0 = a.poly[0]; pc.a1 = a.poly[1]; pc.a2 = a.poly[2]; pc.a3 = a.poly[3]; pc.a4 = a.poly[4];
        pc.p0 = a.poly[8]; pc.p1 = a.poly[9]; pc.p2 = a.poly[10]; pc.p3 = a.poly[11]; pc.p4 = a.poly[12];
    }
    lds_barrier();

    // S = [halo (2 hops) | in]; hop h uses S[(h+1)*HIN .. (h+3)*HIN)
    auto fetch = [&](long h, cf *x) __attribute__((always_inline)) {
        const long base = (h + 1) * HIN;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const long i = base + t + T * m;
            x[m] = i < NIN ? a.halo[i] : a.in[i - NIN];
        }
    };
    // out_h = second_half(Y_{h-1}) + first_half(Y_h).  A shift by half the period is a sign
    // flip of the odd bins, so out_h = first_half(IDFT_nout(stuff(G_h))) with
    //     G_h[k] = F_h[k] + (-1)^k F_{h-1}[k]:
    // the overlap-add happens on the nin-point spectra in registers ((-1)^k = (-1)^t for every
    // bin of lane t) and no time-domain tail is carried from hop to hop.
    // Transforms run two at a time as one packed dual IFFT (struct c2): the Q-1 branch IFFTs
    // of hop h plus the FORWARD transform of hop h+1 (DFT(x) = conj(IDFT(conj(x)))).
    const float sgn = (t & 1) ? -1.0f : 1.0f;
    const float sc = (float)NIN * a.factor;
    cf xn[8], G[8], Fc[8], b0[4];
    {
        // run prologue: F_{h0-1} and F_{h0} as one dual forward transform
        cf xa[8];
        fetch(h0 - 1, xa);
        fetch(h0, xn);
        c2 v2[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const float w = wnd(m);
            v2[m] = c2{make_float2(xa[m].x * w, xn[m].x * w), make_float2(-xa[m].y * w, -xn[m].y * w)};
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) b0[m] = cscale(xn[m], (wnd(m) + wnd(m + 4)) * sc);
        F::template run<+1, true, c2, true>(v2, fbuf2, fpar, tw, t, tw8_l);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            Fc[m] = mk(v2[m].re.y * a.factor, -v2[m].im.y * a.factor);
            G[m] = mk(fmaf(sgn * a.factor, v2[m].re.x, Fc[m].x), fmaf(-sgn * a.factor, v2[m].im.x, Fc[m].y));
        }
    }
    if (h0 + 1 < h1) fetch(h0 + 1, xn);

    // branch twiddle of bin t + T m for branch p (see above); Nyquist bin gets both copies
    auto branch_rot = [](int p, int m) __attribute__((always_inline)) -> cf {
        const double ang = 2.0 * 3.14159265358979323846 * (double)((m * p) % (8 * Q)) / (double)(8 * Q)
                           - (m >= 4 ? 2.0 * 3.14159265358979323846 * (double)p / (double)Q : 0.0);
        return mk((float)__builtin_cos(ang), (float)__builtin_sin(ang));
    };
    auto nyq_scale = [](int p) __attribute__((always_inline)) -> float {
        return 2.0f * (float)__builtin_cos(3.14159265358979323846 * (double)p / (double)Q);
    };
    auto branch_in = [&](int p, int m) __attribute__((always_inline)) -> cf {
        // (G * wp) * rot, in this order: G changes every hop, so nothing is loop-invariant and
        // the products cannot be hoisted into long-lived registers
        cf y = cmul(cmul(G[m], wp[p]), branch_rot(p, m));
        if (m == HIN / T && t == 0) y = cscale(G[m], nyq_scale(p));
        return y;
    };

    for (long h = h0; h < h1; ++h) {
        const int slot = (int)(h & 1);
        // bin HIN lives in lane 0; every lane reads it back after the first transform of the
        // hop (at least one barrier later; the slot is rewritten two hops later)
        if (t == 0) nyq[slot] = G[HIN / T];
        const bool more = h + 1 < h1;

        cf o[4 * Q];                              // all Q branches of the lane's 4 output samples
        // item a of pass i: branch 2i+1; item b: branch 2i+2, or (last pass) the forward transform of the next hop
        auto build = [&](auto passc, c2 *v2) __attribute__((always_inline)) {
            constexpr int pa = 2 * decltype(passc)::value + 1, pb = pa + 1;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (pb < Q) {
                    // two branches: both twiddle products as packed fp32 operations
                    const float2 wr = make_float2(wp[pa].x, wp[pb < Q ? pb : 0].x), wi = make_float2(wp[pa].y, wp[pb < Q ? pb : 0].y);
                    const cf ra = branch_rot(pa, m), rb = branch_rot(pb, m);
                    const float2 rr = make_float2(ra.x, rb.x), ri = make_float2(ra.y, rb.y);
                    const float2 gx = make_float2(G[m].x, G[m].x), gy = make_float2(G[m].y, G[m].y);
                    const float2 yr = pk_fma(gx, wr, pk_neg(gy * wi)), yi = pk_fma(gx, wi, gy * wr);
                    v2[m] = c2{pk_fma(yr, rr, pk_neg(yi * ri)), pk_fma(yr, ri, yi * rr)};
                    if (m == HIN / T && t == 0) {
                        const float2 ny2 = make_float2(nyq_scale(pa), nyq_scale(pb));
                        v2[m] = c2{G[m].x * ny2, G[m].y * ny2};
                    }
                    continue;
                }
                const cf xa = branch_in(pa, m);
                cf xb;
                {
                    // conjugated windowed input of the next hop (zeros past the end of the run)
                    const float w = more ? wnd(m) : 0.0f;
                    xb = mk(xn[m].x * w, -xn[m].y * w);
                }
                v2[m] = c2{make_float2(xa.x, xb.x), make_float2(xa.y, xb.y)};
            }
        };
        auto consume = [&](auto passc, const c2 *v2) __attribute__((always_inline)) {
            constexpr int pa = 2 * decltype(passc)::value + 1, pb = pa + 1;
#pragma unroll
            for (int m = 0; m < 4; ++m) o[m * Q + pa] = mk(v2[m].re.x, v2[m].im.x);
            if (pb < Q) {
#pragma unroll
                for (int m = 0; m < 4; ++m) o[m * Q + (pb < Q ? pb : 0)] = mk(v2[m].re.y, v2[m].im.y);
            } else {
                // branch p = 0 needs no transform: IDFT(DFT(u)) = NIN u, i.e. the input samples
                // under the sum of the two window halves (b0, prepared a hop ahead), plus the
                // second copy of the Nyquist bin, G[NIN/2] e^{i pi q}  (q = t + T m, T even)
                const cf ny = nyq[slot];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    o[m * Q] = mk(fmaf(sgn, ny.x, b0[m].x), fmaf(sgn, ny.y, b0[m].y));
                    b0[m] = cscale(xn[m], (wnd(m) + wnd(m + 4)) * sc);
                }
                // item b = conj(F_{h+1}); the overlap-add with F_h gives the next hop's spectrum
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const cf fn = mk(v2[m].re.y * a.factor, -v2[m].im.y * a.factor);
                    G[m] = mk(fmaf(sgn, Fc[m].x, fn.x), fmaf(sgn, Fc[m].y, fn.y));
                    Fc[m] = fn;
                }
            }
        };
        {
            c2 v2[8];
            build(std::integral_constant<int, 0>{}, v2);
            F::template run<+1, true, c2, true>(v2, fbuf2, fpar, tw, t, tw8_l);
            consume(std::integral_constant<int, 0>{}, v2);
            if constexpr (Q == 4) {
                build(std::integral_constant<int, 1>{}, v2);
                F::template run<+1, true, c2, true>(v2, fbuf2, fpar, tw, t, tw8_l);
                consume(std::integral_constant<int, 1>{}, v2);
            }
        }
        // the input after next is requested before this hop's stores (vmcnt retires in order) ...
        // (only the NEW half: the window of hop h + 2 starts with the second half of hop h + 1's, and sample t + T m of
        // the one is sample t + T (m + 4) of the other -- the same lane.  Every input sample is read once, not twice.)
        if (h + 2 < h1) {
            const long base = (h + 3) * HIN;
#pragma unroll
            for (int m = 0; m < 4; ++m) xn[m] = xn[m + 4];
#pragma unroll
            for (int m = 4; m < 8; ++m) {
                const long i = base + t + T * m;
                xn[m] = i < NIN ? a.halo[i] : a.in[i - NIN];
            }
        }
        // ... and the Q branches of an output sample leave together: 8Q contiguous bytes per lane and
        // slot, so HBM sees whole 32-byte sectors (16-byte pairs stored a transform apart cost 1.5x
        // the write traffic)
        cf *dst = a.out + (size_t)h * HOUT;
        uint32_t *dst16 = reinterpret_cast<uint32_t *>(a.out) + (size_t)h * HOUT;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float4 *d4 = reinterpret_cast<float4 *>(dst + (size_t)Q * (t + T * m));
            uint32_t w16[Q];
#pragma unroll
            for (int p = 0; p < Q; p += 2) {
                cf a0 = o[m * Q + p], a1 = o[m * Q + p + 1];
                if (POLY) poly_apply2(a0, a1, pc);
                if (S16) { w16[p] = s16_pack(a0, nclip); w16[p + 1] = s16_pack(a1, nclip); }
                else d4[p / 2] = make_float4(a0.x, a0.y, a1.x, a1.y);
            }
            if (S16) {
                uint32_t *d = dst16 + (size_t)Q * (t + T * m);
                if (Q == 4) *reinterpret_cast<uint4 *>(d) = make_uint4(w16[0], w16[1], w16[2 % Q], w16[3 % Q]);
                else *reinterpret_cast<uint2 *>(d) = make_uint2(w16[0], w16[1]);
            }
        }
    }
    if (S16) s16_flush_count(nclip, a.clipped);
}

// ===========================================================================
// a10 + a11 at the BASELINE config 4 shape (nin = 4096, x4; also x2): resampler16_kernel.
//
// Two things differ from resampler_kernel above.
// (1) HOPS ARE INDEPENDENT WORK ITEMS.  out_h = second_half(Y_{h-1}) + first_half(Y_h) and the interpolation
//     u -> IDFT_nout(stuff(DFT_nin(u))) is linear and commutes with circular shifts (a shift by nin/2 in goes to a shift
//     by nout/2 out), so the overlap-add moves IN FRONT of the forward transform:
//         out_h = first_half( IDFT_nout( stuff( DFT_nin( g_h ) ) ) ),
//         g_h = [ (w1 + w2) c_{h-1} | w2 c_h + w1 c_{h-2} ]        (w1, w2 = the halves of the window, c = input hops).
//     One forward transform per hop, nothing carried from hop to hop (the kernel above carries the previous spectrum and
//     forms G_h = F_h + (-1)^k F_{h-1}): no run prologue, no hop state in registers.
// (2) 4096 = 16 . 16 . 16 ON 256 LANES: sixteen points per lane, three radix-16 stages, TWO exchanges through LDS per
//     transform where 8 . 8 . 8 . 8 on 512 lanes has three.  A SIMD issues one instruction at a time, VALU or LDS (DESIGN.md
//     section 6), so a transform costs the SUM of its butterfly and its exchange instructions: a third fewer of the latter.
//     Plain single transforms on 256-lane workgroups, TWO independent workgroups per CU (250 VGPRs, 76 KB of LDS with two
//     exchange buffers) that fill each other's barrier waits, where the packed kernel is one 512-lane workgroup per CU in
//     lockstep.  Every input sample is read once (two of a hop's three input hops stay in registers from the hops before,
//     the new one is requested a hop ahead).
//     Measured (same box, cfg 4, 4096 frames): 364 k TF/s against 320 k for resampler_kernel<12, 4>; with three workgroups
//     per CU at <= 168 VGPRs (no room to keep the inputs or to request them ahead) 322 k.  DESIGN.md section 4.3 has the steps.
// The numpy model of every index mapping below: tools/design/resampler16_model.py.
//
// Lane t holds point t + 256 m in slot m, before and after every transform (natural order both sides).
//   stage 1: DFT16 over the slots; exchange 1 keeps element (t, r) at r (256 + 2) + t (one row per output: scatter
//            contiguous across lanes, gather (t & 15) 258 + (t >> 4) + 16 m conflict-free: 2 (t & 15) + (t >> 4) takes every
//            value mod 32 once per 32 lanes)
//   stage 2: twiddle W_256^{m (t mod 16)} (a 16 x 16 LDS table), DFT16; exchange 2 keeps element (t, r) at
//            (t >> 4) 256 + (t & 15) + 16 r, gather t + 256 m (both contiguous across lanes)
//   stage 3: twiddle W_4096^{m t} -- fifteen values per lane, resident in registers (two workgroups per CU leave room for
//            them; formed as products of four resident powers they cost 44 instructions per transform), DFT16.
// 16-point DFT as 4 x 4: DFT4 over slots {i, i+4, i+8, i+12}, the nine twiddles W16^{i k} (one of them +-i, two of them
// 45-degree rotations), DFT4 over i.
// TW: the inputs 1 ... 15 are first multiplied by w[0 ... 14].  The products of the upper two inputs of every first-layer
// DFT4 are folded into its sums -- s = u + v w is four FMAs, d = 2 u - s two -- instead of product, sum and difference:
// four instructions fewer per DFT4, sixteen per twiddled 16-point DFT.
template <int S, bool HALF = false, bool TW = false> DEV void dft16(cf *v, const cf *w = nullptr)
{
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;     // cos, sin (pi / 8)
    if (TW) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const cf x0 = i == 0 ? v[0] : cmul(v[i], w[i - 1]), x1 = cmul(v[i + 4], w[i + 3]);
            const cf s0 = cfma(x0, v[i + 8], w[i + 7]), s2 = cfma(x1, v[i + 12], w[i + 11]);
            const cf s1_ = twice_minus(x0, s0), d3 = twice_minus(x1, s2);
            v[i] = cadd(s0, s2);
            v[i + 8] = csub(s0, s2);
            v[i + 4] = caddi<S>(s1_, d3);
            v[i + 12] = csubi<S>(s1_, d3);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) dft4<S>(v[i], v[i + 4], v[i + 8], v[i + 12]);      // a[i][k] at v[i + 4 k]
    }
    // second layer: DFT4 over i for every k, input i under the twiddle W16^{i k}.  The products are folded into the sums
    // as above (s = u + v w, d = 2 u - s), ten instructions fewer per 16-point DFT; W16^4 = +-i costs nothing.
    constexpr float kh = kSqrtHalf, fS = (float)S;
    const cf W1 = mk(c1, fS * s1), W2 = mk(kh, fS * kh), W3 = mk(s1, fS * c1), W6 = mk(-kh, fS * kh), W9 = mk(-c1, -fS * s1);
    cf y[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const cf b0 = v[4 * k], b1 = v[4 * k + 1], b2 = v[4 * k + 2], b3 = v[4 * k + 3];
        cf s0, s1_, s2, d3;
        if (k == 0) {
            s0 = cadd(b0, b2); s1_ = csub(b0, b2); s2 = cadd(b1, b3); d3 = csub(b1, b3);
        } else {
            const cf wa = k == 1 ? W1 : (k == 2 ? W2 : W3), wb = k == 1 ? W2 : W6, wc = k == 1 ? W3 : (k == 2 ? W6 : W9);
            if (k == 2) { s0 = caddi<S>(b0, b2); s1_ = csubi<S>(b0, b2); }
            else { s0 = cfma(b0, b2, wb); s1_ = twice_minus(b0, s0); }
            const cf x1 = cmul(b1, wa);
            s2 = cfma(x1, b3, wc);
            d3 = twice_minus(x1, s2);
        }
        y[k] = cadd(s0, s2);
        y[4 + k] = caddi<S>(s1_, d3);
        if (!HALF) {                  // (HALF: outputs 0 ... 7 only -- the first half of a branch transform's samples)
            y[8 + k] = csub(s0, s2);
            y[12 + k] = csubi<S>(s1_, d3);
        }
    }
#pragma unroll
    for (int r = 0; r < (HALF ? 8 : 16); ++r) v[r] = y[r];
}

struct Fft16 {
    static constexpr int N = 4096, T = 256, P1 = T + 2;
    static constexpr int LDS_ELEMS = 16 * P1;                 // the row image of exchange 1 (exchange 2 needs N of them)
    template <int S> static DEV cf tw(cf w) { return S > 0 ? w : mk(w.x, -w.y); }
    // pw: W^{m t}, m = 1 ... 15, resident (table holds exp(+2 pi i k / N)); tw2: [16][16] W_256^{m a}
    // Two exchange buffers, used in turn: ONE barrier per exchange (a wave that runs ahead scatters into the buffer its
    // slower siblings are not gathering from; it cannot reach that one again before they have passed the next barrier).
    // w1: optional twiddles on the inputs 1 ... 15 of the FIRST stage (the branch transforms' per-slot rotations: folded
    // into that stage's first butterfly layer like the stage twiddles)
    template <int S, bool HALF, bool TW1 = false> static DEV void run(cf *v, cf *lds0, const cf *tw2, const cf *pw, int t, const cf *w1 = nullptr)
    {
        dft16<S, false, TW1>(v, w1);
        {
            cf *lds = lds0;
            cf *wp = lds + t;
#pragma unroll
            for (int r = 0; r < 16; ++r) wp[r * P1] = v[r];
            xbarrier();
            const cf *rp = lds + ((t & 15) * P1 + (t >> 4));
#pragma unroll
            for (int m = 0; m < 16; ++m) v[m] = rp[16 * m];
        }
        {
            const cf *tp = tw2 + (t & 15);
            cf w[15];
#pragma unroll
            for (int m = 1; m < 16; ++m) w[m - 1] = tw<S>(tp[16 * m]);
            dft16<S, false, true>(v, w);
        }
        {
            cf *lds = lds0 + LDS_ELEMS;
            cf *wp = lds + ((t >> 4) * 256 + (t & 15));
#pragma unroll
            for (int r = 0; r < 16; ++r) wp[16 * r] = v[r];
            xbarrier();
            const cf *rp = lds + t;
#pragma unroll
            for (int m = 0; m < 16; ++m) v[m] = rp[256 * m];
        }
        {
            cf w[15];
#pragma unroll
            for (int m = 1; m < 16; ++m) w[m - 1] = tw<S>(pw[m - 1]);
            dft16<S, HALF, true>(v, w);
        }
    }
};

template <bool POLY, bool S16, int Q = 4> __global__ __launch_bounds__(256, 2)
void resampler16_kernel(const ResamplerArgs a, int hops_per_run)
{
    typedef Fft16 F;
    static_assert(Q == 2 || Q == 4, "x2 and x4");
    constexpr int NIN = F::N, T = F::T, HIN = NIN / 2, HOUT = HIN * Q, NOUT = NIN * Q;
    unsigned nclip = 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf *xbuf = reinterpret_cast<cf *>(smem);                       // two exchange buffers
    cf *tw2 = xbuf + 2 * F::LDS_ELEMS;                             // [16][16]: W_256^{m a}
    cf *nyq = tw2 + 256;                                           // [2]: Nyquist bin per hop parity (+ pad)
    float *win = reinterpret_cast<float *>(nyq + 8);               // first half of the (symmetric) Hann window
    const int t = threadIdx.x;
    const long h0 = (long)blockIdx.x * hops_per_run;
    const long h1 = min((long)a.nhops, h0 + hops_per_run);
    if (h0 >= (long)a.nhops) return;

    tw2[t] = a.tw_in[(16 * (t >> 4) * (t & 15)) & (NIN - 1)];
#pragma unroll
    for (int m = 0; m < 8; ++m) win[t + T * m] = a.window[t + T * m];
    cf pw[15];                                                     // the last stage's twiddles W_nin^{m t}: thirty resident registers
#pragma unroll
    for (int m = 1; m < 16; ++m) pw[m - 1] = a.tw_in[(m * t) & (NIN - 1)];
    const cf wp1 = a.tw_out[t & (NOUT - 1)];                      // W_nout^t; the branches' W_nout^{t p} are its powers
    PolyCoef pc{};
    if (POLY) {
        // wave-uniform: as scalars (read through the pointer they arrive in vector registers, ten loop-invariant ones that
        // the allocator then parks in scratch and reloads every hop)
        auto sc = [&](int i) __attribute__((always_inline)) -> float {
            return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.poly[i])));
        };
        pc.a0 = sc(0); pc.a1 = sc(1); pc.a2 = sc(2); pc.a3 = sc(3); pc.a4 = sc(4);
        pc.p0 = sc(8); pc.p1 = sc(9); pc.p2 = sc(10); pc.p3 = sc(11); pc.p4 = sc(12);
    }
    lds_barrier_vm();

    // S = [halo (2 hops) | in]; sample q of input hop h + k - 2 (k = 0, 1, 2: c_{h-2}, c_{h-1}, c_h)
    auto sample = [&](long h, int k, int q) __attribute__((always_inline)) -> cf {
        const long i = (h + k) * HIN + q;
        return i < NIN ? a.halo[i] : a.in[i - NIN];
    };
    // branch twiddle of bin t + 256 m for branch p:  W_nout^{kappa p} = W_nout^{t p} * e^{2 pi i m p / (16 Q)}
    // (* (-i)^p for the negative-frequency half, kappa = k - nin): one table value per branch and lane, the rest are
    // compile-time rotations; the Nyquist bin (lane 0, slot 8), which the reference places at +nin/2 AND -nin/2
    // (src/Resampler.cpp:153-164), gets the sum of both twiddles
    auto branch_rot = [](int p, int m) __attribute__((always_inline)) -> cf {
        const double ang = 2.0 * 3.14159265358979323846 * (double)((m * p) % (16 * Q)) / (double)(16 * Q)
                           - (m >= 8 ? 2.0 * 3.14159265358979323846 * (double)p / (double)Q : 0.0);
        return mk((float)__builtin_cos(ang), (float)__builtin_sin(ang));
    };
    const float sgn = (t & 1) ? -1.0f : 1.0f;
    const float fN = (float)NIN;

    // Every input sample is read ONCE: a hop needs the input hops c_{h-2}, c_{h-1}, c_h, and the workgroup walks consecutive
    // hops, so two of the three are the previous hop's -- they stay in registers (2 x 16), and the new one (a miss all the way
    // to HBM) is requested a hop AHEAD, before the previous hop's last transform, so that its latency passes behind that
    // transform, the predistorter and the stores.  (With all three loaded at the top of the hop the four waves of the
    // workgroup sat out the HBM latency together: a fifth of the kernel's time.)
    cf in0[8], in1[8], in2[8], nxt[8];                             // c_h, c_{h-1}, c_{h-2}, c_{h+1}: samples t + 256 m
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int q = t + T * m;
        in0[m] = sample(h0, 2, q); in1[m] = sample(h0, 1, q); in2[m] = sample(h0, 0, q);
        nxt[m] = mk(0.f, 0.f);
    }
    for (long h = h0; h < h1; ++h) {
        cf v[16], b0[8];
        // ---- g_h: the overlap-add in front of the forward transform (window halves w1[q] = win[q], w2[q] = win[HIN-1-q]) ----
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int q = t + T * m;
            const float w1 = win[q], w2 = win[HIN - 1 - q];
            const cf c1 = in1[m], c0 = in0[m], c2 = in2[m];
            b0[m] = cscale(c1, (w1 + w2) * a.factor);
            v[m] = b0[m];
            v[m + 8] = mk((w2 * c0.x + w1 * c2.x) * a.factor, (w2 * c0.y + w1 * c2.y) * a.factor);
        }
        F::template run<-1, false>(v, xbuf, tw2, pw, t);           // v = G_h, bin t + 256 m in slot m
        const int slot = (int)(h & 1);
        if (t == 0) nyq[slot] = v[8];                              // (read after the next transform's barriers)
        cf G[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) G[m] = v[m];
        cf o[8 * (Q - 1)];
        cf wpp = wp1;
#pragma unroll
        for (int p = 1; p < Q; ++p) {
            if (p > 1) wpp = cmul(wpp, wp1);
            // v = G W_nout^{t p}; the per-slot rotation goes into the transform's first stage as input twiddles.  The Nyquist
            // bin (lane 0, slot 8) carries 2 cos(pi p / Q) instead: pre-divided by its rotation.
            cf rot[15];
#pragma unroll
            for (int m = 1; m < 16; ++m) rot[m - 1] = branch_rot(p, m);
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                v[m] = cmul(G[m], wpp);
                if (m == 8 && t == 0) {
                    const cf r8 = branch_rot(p, 8);                // (unit modulus: 1 / r8 = conj r8)
                    v[m] = cmul(cscale(G[m], 2.0f * (float)__builtin_cos(3.14159265358979323846 * (double)p / (double)Q)), mk(r8.x, -r8.y));
                }
            }
            if (p == Q - 1 && h + 1 < h1) {
#pragma unroll
                for (int m = 0; m < 8; ++m) nxt[m] = sample(h + 1, 2, t + T * m);
            }
            F::template run<+1, true, true>(v, xbuf, tw2, pw, t, rot);
#pragma unroll
            for (int m = 0; m < 8; ++m) o[(p - 1) * 8 + m] = v[m];
        }
        // ---- branch 0 is the input under the summed window halves (IDFT(DFT(g)) = nin g) plus the second Nyquist copy;
        //      the Q branches of an output sample leave together, 32 contiguous bytes per lane and slot ----
        const cf ny = nyq[slot];
        cf *dst = a.out + (size_t)h * HOUT;
        uint32_t *dst16 = reinterpret_cast<uint32_t *>(a.out) + (size_t)h * HOUT;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            cf a0 = mk(fmaf(fN, b0[m].x, sgn * ny.x), fmaf(fN, b0[m].y, sgn * ny.y)), a1 = o[m];
            cf a2 = Q == 4 ? o[(Q == 4 ? 8 : 0) + m] : a0, a3 = Q == 4 ? o[(Q == 4 ? 16 : 0) + m] : a1;
            if (POLY) { poly_apply2(a0, a1, pc); if (Q == 4) poly_apply2(a2, a3, pc); }
            const size_t at = (size_t)Q * (t + T * m);
            if (S16) {
                if (Q == 4)
                    *reinterpret_cast<uint4 *>(dst16 + at) = make_uint4(s16_pack(a0, nclip), s16_pack(a1, nclip), s16_pack(a2, nclip), s16_pack(a3, nclip));
                else
                    *reinterpret_cast<uint2 *>(dst16 + at) = make_uint2(s16_pack(a0, nclip), s16_pack(a1, nclip));
            } else {
                float4 *d4 = reinterpret_cast<float4 *>(dst + at);
                d4[0] = make_float4(a0.x, a0.y, a1.x, a1.y);
                if (Q == 4) d4[1] = make_float4(a2.x, a2.y, a3.x, a3.y);
            }
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) { in2[m] = in1[m]; in1[m] = in0[m]; in0[m] = nxt[m]; }
    }
    // The stream's state for the next call -- the last two input hops of [halo | in] -- is in the registers of the workgroup
    // that ran the last hop (in2 = c_{last-1}, in1 = c_last after the rotation above): it leaves them in the OTHER halo buffer
    // (the readers of this launch use a.halo), where a device-to-device copy launched behind the kernel used to put them --
    // one launch less per call, 25 -> 18 us for a single frame of cfg 4.
    if (a.halo_out != nullptr && h1 == (long)a.nhops) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            a.halo_out[t + T * m] = in2[m];
            a.halo_out[HIN + t + T * m] = in1[m];
        }
    }
    if (S16) s16_flush_count(nclip, a.clipped);
}

template <int LOGNIN> hipError_t launch_resampler_n(const ResamplerArgs &a, hipStream_t s)
{
    constexpr int NIN = 1 << LOGNIN;
    const int Q = a.nout / a.nin;
    // runs of hops: every run starts with one dual forward transform (half a hop's work).  Long streams
    // get runs of 96 hops (one Mode-I frame); short ones are cut finer so that the launch still covers the
    // chip (>= 512 workgroups when there are that many pairs of hops) -- latency, not efficiency, counts there
    int hpr = (int)std::max<size_t>(2, std::min<size_t>(96, a.nhops / 512));
    const dim3 grid((unsigned)((a.nhops + hpr - 1) / hpr)), block(NIN / 8);
    const size_t lds = 2 * (size_t)(NIN + NIN / 8) * 16 + (2 + 56) * sizeof(float2) + (size_t)(NIN / 2) * sizeof(float);
    const bool poly = a.poly != nullptr;
    switch (Q) {
        case 2:
        case 4:
            if constexpr (LOGNIN == 12) {
                // Mode I (the BASELINE config 4 shape, and x2): hop-independent radix-16 kernel, two 256-lane workgroups per CU.
                // No run prologue, so short streams are cut into single hops; long ones into runs of 24 (four runs per frame)
                const int hpr16 = (int)std::max<size_t>(1, std::min<size_t>(24, a.nhops / 1536));
                const dim3 grid16((unsigned)((a.nhops + hpr16 - 1) / hpr16)), block16(256);
                const size_t lds16 = (size_t)(2 * Fft16::LDS_ELEMS + 256 + 8) * sizeof(float2) + (size_t)(NIN / 2) * sizeof(float);
                // (more than 64 KiB of dynamic LDS has to be asked for)
#define RS16_LAUNCH(P, F, QQ)                                                                                          \
                do {                                                                                                   \
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(resampler16_kernel<P, F, QQ>),   \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);        \
                    if (e != hipSuccess) return e;                                                                     \
                    DABGPU_LAUNCH((resampler16_kernel<P, F, QQ>), grid16, block16, lds16, s, a, hpr16);           \
                } while (0)
                if (Q == 4) {
                    if (a.clipped) { if (poly) RS16_LAUNCH(true, true, 4); else RS16_LAUNCH(false, true, 4); }
                    else           { if (poly) RS16_LAUNCH(true, false, 4); else RS16_LAUNCH(false, false, 4); }
                } else {
                    if (a.clipped) { if (poly) RS16_LAUNCH(true, true, 2); else RS16_LAUNCH(false, true, 2); }
                    else           { if (poly) RS16_LAUNCH(true, false, 2); else RS16_LAUNCH(false, false, 2); }
                }
#undef RS16_LAUNCH
            } else if (Q == 2) {
                if (a.clipped) return hipErrorInvalidValue;
                if (poly) DABGPU_LAUNCH((resampler_kernel<LOGNIN, 2, true>), grid, block, lds, s, a, hpr);
                else DABGPU_LAUNCH((resampler_kernel<LOGNIN, 2, false>), grid, block, lds, s, a, hpr);
            } else {
                if (a.clipped) return hipErrorInvalidValue;
                if (poly) DABGPU_LAUNCH((resampler_kernel<LOGNIN, 4, true>), grid, block, lds, s, a, hpr);
                else DABGPU_LAUNCH((resampler_kernel<LOGNIN, 4, false>), grid, block, lds, s, a, hpr);
            }
            break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace

// the kernel that leaves the next call's halo behind itself (resampler16_kernel): x2 and x4 at nin = 4096 (Mode I)
bool resampler_writes_halo(const ResamplerArgs &a)
{
    return a.nin == 4096 && a.nout % a.nin == 0 && (a.nout / a.nin == 2 || a.nout / a.nin == 4);
}

// the kernels that store s16 themselves: x2 and x4 at nin = 4096 (Mode I)
bool resampler_has_s16(const ResamplerArgs &a)
{
    return a.nin == 4096 && a.nout % a.nin == 0 && (a.nout / a.nin == 2 || a.nout / a.nin == 4);
}

hipError_t launch_resampler(const ResamplerArgs &a, hipStream_t s)
{
    if (a.nhops == 0) return hipSuccess;
    if (a.clipped && !resampler_has_s16(a)) return hipErrorInvalidValue;
    if (a.nout == a.nin || a.nout < 2 || (a.nout & 1)) return hipErrorInvalidValue;
    const bool fast = a.nout % a.nin == 0 && (a.nout / a.nin == 2 || a.nout / a.nin == 4);
    if (!fast && a.poly) return hipErrorInvalidValue;        // the general kernel has no fused predistorter
    switch (a.nin) {
        case 512: return fast ? launch_resampler_n<9>(a, s) : launch_resampler_rational_9(a, s);
        case 1024: return fast ? launch_resampler_n<10>(a, s) : launch_resampler_rational_10(a, s);
        case 2048: return fast ? launch_resampler_n<11>(a, s) : launch_resampler_rational_11(a, s);
        case 4096: return fast ? launch_resampler_n<12>(a, s) : launch_resampler_rational_12(a, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace dabgpu
