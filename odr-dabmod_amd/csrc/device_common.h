// device_common.h -- device-side building blocks shared by every kernel translation unit: complex arithmetic on
// (re, im) registers and on packed pairs of transforms (struct c2), radix-4/8 butterflies, the LDS-resident Stockham FFT
// (struct Fft), workgroup reductions, the register-blocked direct FIR and the FormatConverter (s16) store helpers.
// Everything lives in an anonymous namespace: each .hip file that includes it gets its own copy.
#pragma once
#include "dabgpu_internal.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace dabgpu {
namespace {

typedef float2 cf;
#define DEV __device__ __forceinline__

constexpr float kSqrtHalf = 0.70710678118654752440f;

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a
// workgroup fence over ALL address spaces, which on gfx950 becomes s_waitcnt vmcnt(0):
// with global stores in flight (every symbol ends with ~10 of them per wave) each
// barrier would wait for HBM write acknowledgements.  Nothing in these kernels
// communicates between waves through global memory, so LDS ordering is all we need.
DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// (the variant for data that came through a global load on its way into LDS: the compiler's own wait for the load
// precedes the ds_write, this only orders the write against the other waves)
DEV void lds_barrier_vm() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
DEV void xbarrier() { lds_barrier(); }       // the barriers of the FFT exchanges

// Per-phase cycle stamps of the frame kernel's symbol loop -- a TOOL build only (tools/phase_timing.py compiles a copy of
// the library with -DDABGPU_PHASE_TIMING); in the product build PhaseTimer is an empty type and every stamp() compiles to
// nothing.  A stamp drains the wave's outstanding LDS operations (their latency belongs to the phase that issued them),
// reads s_memtime (shader cycles) and adds the time since the previous stamp to the phase's counter; sched_barrier keeps
// the compiler from moving work across it.  The counters are wave-uniform (SGPRs); flush() adds them to a global array.
// The samples a timing build produces are the product's (timing only, no arithmetic changes); it runs ~10 % slower.
enum Phase { PH_INPUT = 0, PH_BUTTERFLY, PH_EXCHANGE, PH_GAIN_WINDOWS, PH_STORES, PH_BOUNDARY, PH_LOOP, PH_COUNT };
#ifdef DABGPU_PHASE_TIMING
struct PhaseTimer {
    unsigned long long last, acc[PH_COUNT];
    DEV void begin()
    {
#pragma unroll
        for (int i = 0; i < PH_COUNT; ++i) acc[i] = 0;
        last = __builtin_amdgcn_s_memtime();
    }
    DEV void stamp(int phase)
    {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        acc[phase] += now - last;
        last = now;
        __builtin_amdgcn_sched_barrier(0);
    }
    DEV void flush(unsigned long long *out, unsigned iterations)
    {
        if (out && (threadIdx.x & 63) == 0) {
#pragma unroll
            for (int i = 0; i < PH_COUNT; ++i) atomicAdd(out + i, acc[i]);
            atomicAdd(out + 15, (unsigned long long)iterations);      // wave-iterations behind the sums
        }
    }
};
#else
struct PhaseTimer {
    DEV void begin() {}
    DEV void stamp(int) {}
    DEV void flush(unsigned long long *, unsigned) {}
};
#endif

DEV cf mk(float x, float y) { return make_float2(x, y); }
// One-instruction square root / reciprocal (v_sqrt_f32, v_rcp_f32: 1 ulp).  The per-symbol gain is a wave-uniform
// scalar that every lane computes for itself; the correctly rounded sqrtf and division expand to ~17 and ~10
// instructions each, which made this scalar a tenth of the fused kernel's vector instructions.  Arguments are
// zero or far above FLT_MIN, so the denormal pre-scaling of sqrtf is not needed either.
DEV float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
DEV float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
DEV cf cadd(cf a, cf b) { return mk(a.x + b.x, a.y + b.y); }
DEV cf csub(cf a, cf b) { return mk(a.x - b.x, a.y - b.y); }
DEV cf cmul(cf a, cf b) { return mk(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x)); }
DEV cf cscale(cf a, float s) { return mk(a.x * s, a.y * s); }

// Two transforms in lockstep: re = (re_a, re_b), im = (im_a, im_b).  Every complex
// operation is then a pair of packed-fp32 instructions with no lane shuffling at all (a
// multiplication by +-i is a register rename plus a sign modifier), which the interleaved
// (re, im) layout cannot offer.  The fused kernel runs the unfiltered and the filtered IFFT
// of a symbol this way.
struct c2 {
    float2 re, im;
};
DEV c2 cadd(c2 a, c2 b) { return c2{a.re + b.re, a.im + b.im}; }
DEV c2 csub(c2 a, c2 b) { return c2{a.re - b.re, a.im - b.im}; }
typedef float v2f __attribute__((ext_vector_type(2)));
// Twiddle product of the packed pair: four VOP3P instructions that read the twiddle's two halves through
// op_sel.  Written out by hand because the compiler does not use op_sel here: from the plain expression
// below it keeps every twiddle duplicated as (x, x) and (y, y) register pairs -- 28 extra VGPRs in the FIR
// variants of the frame kernel.
DEV c2 cmul(c2 a, cf w)
{
    const v2f are = {a.re.x, a.re.y}, aim = {a.im.x, a.im.y}, ww = {w.x, w.y};
    v2f t0, t1, re, im;
    // (a product and the FMA that consumes it are ONE statement: between two asm statements that depend on each other the
    // compiler puts an s_nop -- 39 of them per hop in the resampler)
    asm("v_pk_mul_f32 %1, %3, %4 op_sel:[0,1] op_sel_hi:[1,1]\n\t"                                                    // t0 = im * w.y
        "v_pk_fma_f32 %0, %2, %4, %1 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]"                   // re * w.x - t0
        : "=v"(re), "=&v"(t0) : "v"(are), "v"(aim), "v"(ww));
    asm("v_pk_mul_f32 %1, %3, %4 op_sel:[0,0] op_sel_hi:[1,0]\n\t"                                                    // t1 = im * w.x
        "v_pk_fma_f32 %0, %2, %4, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]"                                                 // re * w.y + t1
        : "=v"(im), "=&v"(t1) : "v"(are), "v"(aim), "v"(ww));
    return c2{make_float2(re.x, re.y), make_float2(im.x, im.y)};
}

// a * b + c on both halves, as ONE packed FMA whatever the surrounding code looks like.  A sum of two products
// written with operators (x * y - z * w) leaves the choice of the product that is fused to the instruction selector,
// and that choice can differ between two instantiations of the same kernel: the variants of a kernel that differ
// in their output format only must produce the same floats.
typedef float v2f_ __attribute__((ext_vector_type(2)));
DEV float2 pk_fma(float2 a, float2 b, float2 c)
{
    const v2f_ r = __builtin_elementwise_fma(v2f_{a.x, a.y}, v2f_{b.x, b.y}, v2f_{c.x, c.y});
    return make_float2(r.x, r.y);
}
DEV float2 pk_neg(float2 a) { return make_float2(-a.x, -a.y); }

// multiply by (S * i)
template <int S> DEV cf mul_i(cf a) { return S > 0 ? mk(-a.y, a.x) : mk(a.y, -a.x); }
template <int S> DEV c2 mul_i(c2 a) { return S > 0 ? c2{-a.im, a.re} : c2{a.im, -a.re}; }
// multiply by exp(S i pi/4) and by exp(S 3 i pi/4)
template <int S> DEV cf rot1(cf b) { return mk(kSqrtHalf * (b.x - S * b.y), kSqrtHalf * (S * b.x + b.y)); }
template <int S> DEV cf rot3(cf b) { return mk(kSqrtHalf * (-b.x - S * b.y), kSqrtHalf * (S * b.x - b.y)); }
template <int S> DEV c2 rot1(c2 b)
{
    return c2{(b.re - (float)S * b.im) * kSqrtHalf, ((float)S * b.re + b.im) * kSqrtHalf};
}
template <int S> DEV c2 rot3(c2 b)
{
    return c2{(-b.re - (float)S * b.im) * kSqrtHalf, ((float)S * b.re - b.im) * kSqrtHalf};
}

// a + S i b,  a - S i b: for cf and c2 a multiplication by +-i (register renames and sign modifiers) and a sum; the
// packed single-complex type pc (below) does both in ONE instruction through op_sel
template <int S> DEV cf caddi(cf a, cf b) { return cadd(a, mul_i<S>(b)); }
template <int S> DEV cf csubi(cf a, cf b) { return csub(a, mul_i<S>(b)); }
template <int S> DEV c2 caddi(c2 a, c2 b) { return cadd(a, mul_i<S>(b)); }
template <int S> DEV c2 csubi(c2 a, c2 b) { return csub(a, mul_i<S>(b)); }

// 4-point DFT, exp(S 2 pi i nk/4), natural order in place
template <int S, typename V> DEV void dft4(V &x0, V &x1, V &x2, V &x3)
{
    const V s0 = cadd(x0, x2), s1 = csub(x0, x2), s2 = cadd(x1, x3), d3 = csub(x1, x3);
    x0 = cadd(s0, s2);
    x2 = csub(s0, s2);
    x1 = caddi<S>(s1, d3);
    x3 = csubi<S>(s1, d3);
}

// The odd half of an 8-point DFT: b1 and b3 are due a rotation by exp(S i pi/4) resp. exp(S 3 i pi/4), i.e. a
// rotation by +-45 degrees without its factor sqrt(1/2) and then that factor.  The factor moves into the last layer
// of the 4-point DFT, where it is the multiplier of an FMA: the four multiplications of rot1 / rot3 disappear.
DEV cf axpy(cf a, float c, cf b) { return mk(fmaf(c, b.x, a.x), fmaf(c, b.y, a.y)); }          // a + c b
DEV c2 axpy(c2 a, float c, c2 b) { return c2{pk_fma(b.re, make_float2(c, c), a.re), pk_fma(b.im, make_float2(c, c), a.im)}; }
// a + c S i b
template <int S> DEV cf axpyi(cf a, float c, cf b) { return axpy(a, c, mul_i<S>(b)); }
template <int S> DEV c2 axpyi(c2 a, float c, c2 b) { return axpy(a, c, mul_i<S>(b)); }
template <int S> DEV cf urot1(cf b) { return mk(b.x - S * b.y, S * b.x + b.y); }                // sqrt(2) exp(S i pi/4) b
template <int S> DEV cf urot3(cf b) { return mk(-b.x - S * b.y, S * b.x - b.y); }               // sqrt(2) exp(S 3 i pi/4) b
template <int S> DEV c2 urot1(c2 b) { return c2{b.re - (float)S * b.im, (float)S * b.re + b.im}; }
template <int S> DEV c2 urot3(c2 b) { return c2{-b.re - (float)S * b.im, (float)S * b.re - b.im}; }
template <int S, typename V> DEV void dft8_odd(V &b0, V &b1, V &b2, V &b3)
{
    const V p1 = urot1<S>(b1), p3 = urot3<S>(b3);
    const V s0 = caddi<S>(b0, b2), s1 = csubi<S>(b0, b2), s2 = cadd(p1, p3), d3 = csub(p1, p3);
    b0 = axpy(s0, kSqrtHalf, s2);
    b2 = axpy(s0, -kSqrtHalf, s2);
    b1 = axpyi<S>(s1, kSqrtHalf, d3);
    b3 = axpyi<S>(s1, -kSqrtHalf, d3);
}

// 8-point DFT (decimation in frequency), natural order in place
template <int S, typename V> DEV void dft8(V *v)
{
    V a0 = cadd(v[0], v[4]), b0 = csub(v[0], v[4]);
    V a1 = cadd(v[1], v[5]), b1 = csub(v[1], v[5]);
    V a2 = cadd(v[2], v[6]), b2 = csub(v[2], v[6]);
    V a3 = cadd(v[3], v[7]), b3 = csub(v[3], v[7]);
    dft4<S>(a0, a1, a2, a3);
    dft8_odd<S>(b0, b1, b2, b3);
    v[0] = a0; v[2] = a1; v[4] = a2; v[6] = a3;
    v[1] = b0; v[3] = b1; v[5] = b2; v[7] = b3;
}

// The same with v[4] known to be zero (first stage of the frame kernel: bin t + 4T lies in the unoccupied band)
template <int S, typename V> DEV void dft8_v4zero(V *v)
{
    V a0 = v[0], b0 = v[0];
    V a1 = cadd(v[1], v[5]), b1 = csub(v[1], v[5]);
    V a2 = cadd(v[2], v[6]), b2 = csub(v[2], v[6]);
    V a3 = cadd(v[3], v[7]), b3 = csub(v[3], v[7]);
    dft4<S>(a0, a1, a2, a3);
    dft8_odd<S>(b0, b1, b2, b3);
    v[0] = a0; v[2] = a1; v[4] = a2; v[6] = a3;
    v[1] = b0; v[3] = b1; v[5] = b2; v[7] = b3;
}

// Twiddles W^1..W^7 on v[1..7], then the 8-point DFT.  Packed pairs: the products of the upper four inputs are folded
// into the first butterfly layer -- a_i = u_i + v_{i+4} w (four packed FMAs, the same count as the product alone) and
// b_i = 2 u_i - a_i (two) instead of product, sum and difference: 8 packed instructions fewer per stage.
DEV c2 cfma(c2 t, c2 x, cf w)          // t + x * w
{
    const v2f tre = {t.re.x, t.re.y}, tim = {t.im.x, t.im.y}, xre = {x.re.x, x.re.y}, xim = {x.im.x, x.im.y}, ww = {w.x, w.y};
    v2f r0, re, i0, im;
    asm("v_pk_fma_f32 %1, %2, %4, %5 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"                                             // r0 = t.re + x.re w.x
        "v_pk_fma_f32 %0, %3, %4, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"                    // - x.im w.y
        : "=v"(re), "=&v"(r0) : "v"(xre), "v"(xim), "v"(ww), "v"(tre));
    asm("v_pk_fma_f32 %1, %2, %4, %5 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"                                             // i0 = t.im + x.re w.y
        "v_pk_fma_f32 %0, %3, %4, %1 op_sel:[0,0,0] op_sel_hi:[1,0,1]"                                                  // + x.im w.x
        : "=v"(im), "=&v"(i0) : "v"(xre), "v"(xim), "v"(ww), "v"(tim));
    return c2{make_float2(re.x, re.y), make_float2(im.x, im.y)};
}
DEV cf cfma(cf t, cf x, cf w)          // t + x * w: four FMAs, what the product alone costs
{
    return mk(fmaf(-x.y, w.y, fmaf(x.x, w.x, t.x)), fmaf(x.y, w.x, fmaf(x.x, w.y, t.y)));
}
DEV cf twice_minus(cf u, cf s) { return mk(fmaf(2.0f, u.x, -s.x), fmaf(2.0f, u.y, -s.y)); }       // 2 u - s
template <int S> DEV void twiddle_dft8(cf *v, const cf *w)
{
    // (the same folding as for packed pairs below: a_i = u_i + v_{i+4} w, b_i = 2 u_i - a_i -- two instructions fewer per
    // pair than product, sum and difference)
    cf a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const cf u = i == 0 ? v[0] : cmul(v[i], w[i - 1]);
        a[i] = cfma(u, v[i + 4], w[i + 3]);
        b[i] = twice_minus(u, a[i]);
    }
    dft4<S>(a[0], a[1], a[2], a[3]);
    dft8_odd<S>(b[0], b[1], b[2], b[3]);
    v[0] = a[0]; v[2] = a[1]; v[4] = a[2]; v[6] = a[3];
    v[1] = b[0]; v[3] = b[1]; v[5] = b[2]; v[7] = b[3];
}
template <int S> DEV void twiddle_dft8(c2 *v, const cf *w)
{
    c2 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const c2 u = i == 0 ? v[0] : cmul(v[i], w[i - 1]);
        a[i] = cfma(u, v[i + 4], w[i + 3]);
        b[i] = c2{pk_fma(u.re, make_float2(2.0f, 2.0f), pk_neg(a[i].re)), pk_fma(u.im, make_float2(2.0f, 2.0f), pk_neg(a[i].im))};
    }
    dft4<S>(a[0], a[1], a[2], a[3]);
    dft8_odd<S>(b[0], b[1], b[2], b[3]);
    v[0] = a[0]; v[2] = a[1]; v[4] = a[2]; v[6] = a[3];
    v[1] = b[0]; v[3] = b[1]; v[5] = b[2]; v[7] = b[3];
}

// ---------------------------------------------------------------------------
// N-point FFT, N/8 lanes, 8 points per lane.  Lane t holds x[t + T*m], m=0..7,
// before and after (natural order both sides).  Stockham autosort: after the
// stage with stride Ns lane j writes element r to j0 + r*Ns,
// j0 = (j/Ns)*8*Ns + j%Ns, and reads back t + T*m.
template <int LOGN> struct Fft {
    static constexpr int N = 1 << LOGN;
    static constexpr int T = N / 8;
    static constexpr int NR8 = LOGN / 3;          // radix-8 stages
    static constexpr int RF = N >> (3 * NR8);     // final radix 1/2/4
    static constexpr int NB = RF > 1 ? 8 / RF : 0;  // final-stage butterflies per lane
    static constexpr int TW_PER_STAGE = 7;                    // W^1..W^7 of every radix-8 stage stay resident
    static constexpr int TW_FINAL = RF > 1 ? RF - 1 : 0;
    static constexpr int NTW = TW_PER_STAGE * (NR8 - 1) + NB * TW_FINAL;

    // LDS image of the exchange buffer: element i lives at i + (i >> 3) for the
    // two scatters with stride 1 and 8 (pad one slot per 8: ds_write_b64 is then
    // bank-conflict-free, ds_read_b64 2-way) and at i for strides >= 64 (both
    // conflict-free).  Additive padding (unlike an XOR swizzle) keeps every
    // address of a lane at base + compile-time offset, so the 16 accesses of an
    // exchange need 2 address registers instead of 16.
    static constexpr int LDS_ELEMS = N + N / 8;

    // One barrier per exchange: consecutive exchanges alternate between two LDS
    // buffers, so the next scatter can never overtake a lane still gathering from
    // the previous one (that lane is at most one barrier behind).
    // 8-byte elements (cf): padded i + (i >> 3) for strides 1 and 8.  16-byte elements (c2,
    // ds_*_b128): only the stride-1 scatter needs padding, i + (i >> 3) as well -- ds_write_b128 is served 8 lanes at
    // a time over a 128-byte bank window, so the stride-1 scatter (lane t -> elements 8t .. 8t+7) is conflict-free at a
    // 144-byte lane pitch and 2-way conflicted with one pad per 16; strides 8 and 64 are conflict-free unpadded.
    // the two halves of an exchange: scatter after the stage with stride NS, gather in natural order.
    // First exchange (NS = 1): element (lane t, output r) -- position 8t + r of the autosort order -- is kept at
    // r (T + 4) + t, one row per output.  The scatter is then contiguous across lanes, and the gather of lane t'
    // (positions t' + T m, i.e. output t' % 8 of lane t'/8 + (T/8) m) reads (t' % 8)(T + 4) + t'/8 + (T/8) m:
    // with a row pitch of T + 4 elements the 16 lanes that ds_read_b128 serves together ({0-3, 12-15, 20-27}, ...)
    // and the 32 lanes of a ds_read_b64 group fall into distinct bank slots (4 (t' % 8) + t'/8 mod 16 resp. mod 32 takes
    // every value once).  Both sides conflict-free; every address still base + immediate.
    static constexpr int X1_PITCH = T + 4;
    static constexpr bool X1_ROWS = T >= 32;      // (8 (T + 4) elements must fit in LDS_ELEMS)
    // stride-8 exchange of 8-byte elements with one pad block of 8 per 64 elements (see xwrite)
    template <int NS, typename V> static constexpr bool BLOCKPAD8() { return NS == 8 && sizeof(V) == 8 && T % 64 == 0; }
    template <int NS, typename V> static DEV void xwrite(const V *v, V *lds, int t)
    {
        if (NS == 1 && X1_ROWS) {
            V *wp = lds + t;
#pragma unroll
            for (int r = 0; r < 8; ++r) wp[r * X1_PITCH] = v[r];
            return;
        }
        if (BLOCKPAD8<NS, V>()) {
            // 8-byte elements, stride 8: element i at i + 8 (i >> 6) -- the lane's eight elements 64 a + b + 8 r share one
            // block of 64, so the scatter is base + 8 r (bank pairs 16 a + 2 b + 16 r: distinct over the 32 lanes of a pass) and
            // the gather of t + T m is t + 8 (t >> 6) + m (T + T/8): 32 consecutive elements, conflict-free too.  (One pad per
            // 8 elements, i + (i >> 3), made the gather 2-way conflicted: lanes 29 - 31 of a pass wrapped onto lanes 0 - 2.)
            V *wp = lds + ((t >> 3) * 72 + (t & 7));
#pragma unroll
            for (int r = 0; r < 8; ++r) wp[8 * r] = v[r];
            return;
        }
        // (the padded read address base + m (T + T/P) needs T to be a multiple of P: tiny transforms go unpadded)
        constexpr int PS = 3, P = 1 << PS;
        constexpr bool PAD = (T % P == 0) && (sizeof(V) == 8 ? (NS < 64) : (NS == 1));
        const int j0 = (t / NS) * NS * 8 + (t % NS);
        V *wp = lds + (PAD ? j0 + (j0 >> PS) : j0);
#pragma unroll
        for (int r = 0; r < 8; ++r) wp[PAD ? r * NS + (r * NS) / P : r * NS] = v[r];
    }
    template <int NS, typename V> static DEV void xread(V *v, const V *lds, int t)
    {
        if (NS == 1 && X1_ROWS) {
            const V *rp = lds + ((t & 7) * X1_PITCH + (t >> 3));
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = rp[m * (T / 8)];
            return;
        }
        if (BLOCKPAD8<NS, V>()) {
            const V *rp = lds + (t + 8 * (t >> 6));
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = rp[m * (T + T / 8)];
            return;
        }
        constexpr int PS = 3, P = 1 << PS;
        constexpr bool PAD = (T % P == 0) && (sizeof(V) == 8 ? (NS < 64) : (NS == 1));
        const V *rp = lds + (PAD ? t + (t >> PS) : t);
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = rp[PAD ? m * (T + T / P) : m * T];
    }
    template <int NS, bool DBUF, typename V> static DEV void exchange(V *v, V *lds, int t)
    {
        xwrite<NS, V>(v, lds, t);
        xbarrier();
        xread<NS, V>(v, lds, t);
        if (!DBUF) xbarrier();
    }

    // SKIP8: the stride-8 stage reads its twiddles from the LDS table (fill_tw8) instead
    template <bool SKIP8 = false>
    static DEV void load_twiddles(const cf *__restrict__ wtab, int t, cf *tw)
    {
        int n = 0;
        int ns = 8;
#pragma unroll
        for (int st = 1; st < NR8; ++st) {
            const int base = (t % ns) * (N / (ns * 8));
#pragma unroll
            for (int r = 1; r < 8; ++r) {
                if (!(SKIP8 && st == 1)) tw[n] = wtab[(r * base) & (N - 1)];
                ++n;
            }
            ns *= 8;
        }
        if (RF > 1) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int r = 1; r < RF; ++r)
                    tw[n++] = wtab[(r * (t + T * b)) & (N - 1)];
        }
    }

    // fill the 7 x 8 twiddle table of the stride-8 stage (they depend on lane % 8 only: an LDS table instead of 14
    // resident VGPRs); call once, then barrier
    static DEV void fill_tw8(const cf *__restrict__ wtab, cf *tw8, int t)
    {
        if (t < 56) tw8[t] = wtab[(((t >> 3) + 1) * (t & 7) * (N / 64)) & (N - 1)];
    }

    // the seven twiddles W^1..W^7 of a radix-8 stage
    template <int S> static DEV void stage_twiddles(const cf *tw, int &n, cf *w)
    {
#pragma unroll
        for (int r = 0; r < 7; ++r) w[r] = twid<S>(tw[n + r]);
        n += 7;
    }

    // conjugate twiddles when S < 0 (table holds exp(+2 pi i m/N))
    template <int S> static DEV cf twid(cf w) { return S > 0 ? w : mk(w.x, -w.y); }

    // U8: read the stride-8 stage's twiddles from the LDS table tw8 (true) or from the resident set tw (false).  A
    // template argument, not a test of the pointer: a pointer into the dynamic LDS block is never provably non-null,
    // and the run-time test costs a scalar branch per twiddle.
    template <int S, bool DBUF, typename V, bool U8>
    static DEV void run(V *v, V *lds2, int &par, const cf *tw, int t, const cf *tw8 = nullptr, PhaseTimer *pt = nullptr)
    {
#define DABGPU_STAMP(PH) do { if (pt) pt->stamp(PH); } while (0)
#define DABGPU_NEXT_BUF (lds2 + ((DBUF && (par ^= 1)) ? LDS_ELEMS : 0))
        int n = 0;
        cf w[7];
        // table twiddles of the second stage: requested ahead of the first exchange, whose barriers their LDS
        // round trip then hides behind (read after it, the compiler serialises them: four round trips per transform)
        constexpr bool EARLY8 = U8 && NR8 >= 2;
        if (EARLY8) {
#pragma unroll
            for (int r = 0; r < 7; ++r) w[r] = twid<S>(tw8[r * 8 + (t & 7)]);
        }
        dft8<S>(v);
        DABGPU_STAMP(PH_BUTTERFLY);
        exchange<1, DBUF, V>(v, DABGPU_NEXT_BUF, t);
        DABGPU_STAMP(PH_EXCHANGE);
        if (NR8 >= 2) {
            if (EARLY8) n += 7; else stage_twiddles<S>(tw, n, w);
            twiddle_dft8<S>(v, w);
            DABGPU_STAMP(PH_BUTTERFLY);
            if (NR8 > 2 || RF > 1) exchange<8, DBUF, V>(v, DABGPU_NEXT_BUF, t);
            DABGPU_STAMP(PH_EXCHANGE);
        }
        if (NR8 >= 3) {
            stage_twiddles<S>(tw, n, w);
            twiddle_dft8<S>(v, w);
            DABGPU_STAMP(PH_BUTTERFLY);
            if (NR8 > 3 || RF > 1) exchange<64, DBUF, V>(v, DABGPU_NEXT_BUF, t);
            DABGPU_STAMP(PH_EXCHANGE);
        }
        if (NR8 >= 4) {
            stage_twiddles<S>(tw, n, w);
            twiddle_dft8<S>(v, w);
            DABGPU_STAMP(PH_BUTTERFLY);
            if (RF > 1) exchange<512, DBUF, V>(v, DABGPU_NEXT_BUF, t);
            DABGPU_STAMP(PH_EXCHANGE);
        }
        if (RF == 4) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const cf w1 = twid<S>(tw[n]), w2 = twid<S>(tw[n + 1]), w3 = twid<S>(tw[n + 2]);
                n += 3;
                if constexpr (std::is_same<V, cf>::value) {
                    // (products folded into the first layer's sums, as in twiddle_dft8)
                    const cf x0 = v[b], x1 = cmul(v[b + 2], w1);
                    const cf s0 = cfma(x0, v[b + 4], w2), s2 = cfma(x1, v[b + 6], w3);
                    const cf s1 = twice_minus(x0, s0), d3 = twice_minus(x1, s2);
                    v[b] = cadd(s0, s2); v[b + 4] = csub(s0, s2);
                    v[b + 2] = caddi<S>(s1, d3); v[b + 6] = csubi<S>(s1, d3);
                } else {
                    V x0 = v[b], x1 = cmul(v[b + 2], w1), x2 = cmul(v[b + 4], w2), x3 = cmul(v[b + 6], w3);
                    dft4<S>(x0, x1, x2, x3);
                    v[b] = x0; v[b + 2] = x1; v[b + 4] = x2; v[b + 6] = x3;
                }
            }
        } else if (RF == 2) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const V x0 = v[b], x1 = cmul(v[b + 4], twid<S>(tw[n++]));
                v[b] = cadd(x0, x1);
                v[b + 4] = csub(x0, x1);
            }
        }
        DABGPU_STAMP(PH_BUTTERFLY);
#undef DABGPU_STAMP
#undef DABGPU_NEXT_BUF
    }

    // Packed dual transform (N = 2048, one exchange buffer) whose FIRST half is wanted at two short runs of
    // outputs only -- the FIR boundary samples of the frame kernel: sample t + 6T in the first wave (the head of
    // the cyclic prefix starts at N - cp = 6T + 8) and sample t + 7T in the last wave (the symbol's tail).
    // Identical to run<S, false, c2> up to the third butterfly stage.  The last exchange then moves the SECOND
    // half alone, as 8-byte elements (ds_write_b128 costs 13 LDS cycles per wave, ds_write_b64 6; reads 4 vs 2),
    // plus outputs 0 and 7 of the first half's stage -- the only inputs of the wanted samples:
    //   sample t + 6T (slot 6 = butterfly 0, output 3) reads positions t + 512 q       = stage output 0 of lane (q, t)
    //   sample t + 7T (slot 7 = butterfly 1, output 3) reads positions t + 256 + 512 q = stage output 7 of lane (q, t - 192)
    // and the final radix-4 stage runs on the second half (z, natural order) and on that one sample (uedge;
    // meaningful in waves 0 and 3).
    template <int S>
    static DEV void run_dual_zonly(c2 *v, c2 *lds, const cf *tw, int t, const cf *tw8, cf *z, cf &uedge)
    {
        static_assert(LOGN == 11, "geometry of transmission mode I");
        int n = 7;                    // (the stride-8 stage's twiddles come from the LDS table)
        cf w[7];
#pragma unroll
        for (int r = 0; r < 7; ++r) w[r] = twid<S>(tw8[r * 8 + (t & 7)]);
        dft8_v4zero<S>(v);            // (every caller is the frame kernel: input 4 is in the empty band)
        exchange<1, false, c2>(v, lds, t);
        twiddle_dft8<S>(v, w);
        exchange<8, false, c2>(v, lds, t);
        stage_twiddles<S>(tw, n, w);
        twiddle_dft8<S>(v, w);
        // third exchange: second half alone, first half's outputs 0 and 7 beside it.  Real and imaginary parts go to
        // separate planes (N floats each, 32 x 256 bytes apart) with ds_write2st64_b32 / ds_read2st64_b32: those take their
        // two dwords from / deliver them to ANY two registers, so the halves of the (unfiltered, filtered) register pairs
        // need no moves into (re, im) order on the way out, and arrive as (re, im) pairs on the way back.  Written by
        // hand: the backend's own pairing is switched off (Makefile), and it would not pair across planes anyway.
        float *zre = reinterpret_cast<float *>(lds);                 // [N] re, [N] im
        float *ure = zre + 2 * N;                                    // [T] u0.re, [T] u0.im, [T] u7.re, [T] u7.im
        {
            const unsigned waddr = (unsigned)(uintptr_t)(zre + ((t / 64) * 512 + (t % 64)));    // LDS byte address
            const unsigned uaddr = (unsigned)(uintptr_t)(ure + t);
#define DABGPU_ZW(R)                                                                                          \
            asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4"                                  \
                         :: "v"(waddr), "v"(v[R].re.y), "v"(v[R].im.y), "n"(R), "n"(32 + R) : "memory")
            DABGPU_ZW(0); DABGPU_ZW(1); DABGPU_ZW(2); DABGPU_ZW(3); DABGPU_ZW(4); DABGPU_ZW(5); DABGPU_ZW(6); DABGPU_ZW(7);
#undef DABGPU_ZW
            static_assert(T == 256, "plane offsets below are in units of 64 dwords");
            asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:0 offset1:4" :: "v"(uaddr), "v"(v[0].re.x), "v"(v[0].im.x) : "memory");
            asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:8 offset1:12" :: "v"(uaddr), "v"(v[7].re.x), "v"(v[7].im.x) : "memory");
        }
        xbarrier();
        const int wv = __builtin_amdgcn_readfirstlane(t >> 6);     // wave index, as a scalar: the branches below are
                                                                   // then scalar branches, not exec-masked copies
        const bool edge_wave = wv == 0 || wv == 3;
        cf e[4];                                                   // (read only where edge_wave)
        if (edge_wave) {
            const float *up = (wv == 0) ? ure + t : ure + 2 * T + (t - 192);
#pragma unroll
            for (int q = 0; q < 4; ++q) e[q] = mk(up[64 * q], up[T + 64 * q]);
        }
        {
            const unsigned raddr = (unsigned)(uintptr_t)(zre + t);
            v2f zz[8];
            // (the wait is part of the statement: the compiler does not count LDS operations issued from asm)
            asm volatile("ds_read2st64_b32 %0, %8 offset0:0 offset1:32\n\t"
                         "ds_read2st64_b32 %1, %8 offset0:4 offset1:36\n\t"
                         "ds_read2st64_b32 %2, %8 offset0:8 offset1:40\n\t"
                         "ds_read2st64_b32 %3, %8 offset0:12 offset1:44\n\t"
                         "ds_read2st64_b32 %4, %8 offset0:16 offset1:48\n\t"
                         "ds_read2st64_b32 %5, %8 offset0:20 offset1:52\n\t"
                         "ds_read2st64_b32 %6, %8 offset0:24 offset1:56\n\t"
                         "ds_read2st64_b32 %7, %8 offset0:28 offset1:60\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(zz[0]), "=&v"(zz[1]), "=&v"(zz[2]), "=&v"(zz[3]), "=&v"(zz[4]), "=&v"(zz[5]), "=&v"(zz[6]),
                           "=&v"(zz[7])
                         : "v"(raddr) : "memory");
#pragma unroll
            for (int m = 0; m < 8; ++m) z[m] = mk(zz[m].x, zz[m].y);
        }
        xbarrier();
        cf wb[2][3];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int r = 0; r < 3; ++r) wb[b][r] = twid<S>(tw[n + 3 * b + r]);
            cf x0 = z[b], x1 = cmul(z[b + 2], wb[b][0]), x2 = cmul(z[b + 4], wb[b][1]), x3 = cmul(z[b + 6], wb[b][2]);
            dft4<S>(x0, x1, x2, x3);
            z[b] = x0; z[b + 2] = x1; z[b + 4] = x2; z[b + 6] = x3;
        }
        // output 3 of the butterfly: (x0 - x2) - S i (x1 - x3).  The last wave needs the twiddles of butterfly 1,
        // W^{r (t + T)} = W^{r t} exp(S i r pi/4): butterfly 0's, and three fixed rotations on the products -- selecting
        // between two twiddle sets would cost every wave a dozen register copies
        uedge = mk(0.f, 0.f);
        if (edge_wave) {
            cf y1 = cmul(e[1], wb[0][0]), y2 = cmul(e[2], wb[0][1]), y3 = cmul(e[3], wb[0][2]);
            if (wv == 3) {
                y1 = rot1<S>(y1);
                y2 = mul_i<S>(y2);
                y3 = rot3<S>(y3);
            }
            uedge = csub(csub(e[0], y2), mul_i<S>(csub(y1, y3)));
        }
    }

};

// ---------------------------------------------------------------------------
// block reductions (T lanes, T multiple of 32; red = small LDS scratch)
DEV float wave_sum(float x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}
DEV float wave_max(float x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
    return x;
}

DEV double wave_sum_d(double x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// two sums at once, accumulated in float64 (four values per symbol: cheap, and it
// keeps the variance within 1e-8 of the exact population variance)
template <int T> DEV void block_sum2(double &a, double &b, double *red, int t)
{
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    constexpr int NW = (T + 63) / 64;
    if (NW > 1) {
        if ((t & 63) == 0) { red[2 * (t >> 6)] = a; red[2 * (t >> 6) + 1] = b; }
        lds_barrier();
        double sa = 0., sb = 0.;
#pragma unroll
        for (int w = 0; w < NW; ++w) { sa += red[2 * w]; sb += red[2 * w + 1]; }
        a = sa; b = sb;
        lds_barrier();
    }
}

template <int T> DEV float block_max(float a, double *redd, int t)
{
    float *red = reinterpret_cast<float *>(redd);
    a = wave_max(a);
    constexpr int NW = (T + 63) / 64;
    if (NW > 1) {
        if ((t & 63) == 0) red[t >> 6] = a;
        lds_barrier();
        float m = red[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) m = fmaxf(m, red[w]);
        a = m;
        lds_barrier();
    }
    return a;
}

// Gain of one symbol from its N samples held 8 per lane.
// Reference src/GainControl.cpp:196-340 (a per-SSE-lane running mean / running variance;
// here: two-pass mean / population variance, parallel reduction).
template <int T> DEV float symbol_gain(const cf *v, const GainParams &gp, double *red, int t,
                                        bool on = true)
{
    constexpr double invN = 1.0 / (8 * T);
    const float live = on ? 1.0f : 0.0f;  // lanes beyond T (N = 256 only) contribute nothing
    if (gp.mode == 0) return 512.0f;
    if (gp.mode == 1) {
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) m = fmaxf(m, fmaxf(fabsf(v[i].x), fabsf(v[i].y)));
        m = block_max<T>(m * live, red, t);
        return ((int)m != 0) ? 32767.0f / m : 1.0f;
    }
    double sr = 0., si = 0.;
#pragma unroll
    for (int i = 0; i < 8; ++i) { sr += (double)v[i].x; si += (double)v[i].y; }
    sr *= live; si *= live;
    block_sum2<T>(sr, si, red, t);
    const float mr = (float)(sr * invN), mi = (float)(si * invN);
    double qr = 0., qi = 0.;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float dr = v[i].x - mr, di = v[i].y - mi;
        qr += (double)dr * (double)dr;
        qi += (double)di * (double)di;
    }
    qr *= live; qi *= live;
    block_sum2<T>(qr, qi, red, t);
    const float vr = sqrtf((float)(qr * invN)) * gp.var_variance,
                vi = sqrtf((float)(qi * invN)) * gp.var_variance;
    if ((int)vr == 0) return 1.0f;
    return 32767.0f / fmaxf(vr, vi);
}

// ---------------------------------------------------------------------------
// Wave-wide reductions on DPP (VALU data paths, no LDS traffic): two quad
// permutes, two row rotations, then the four row results through SGPRs.
template <int CTRL> DEV float dpp_mov(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL,
                                                                 0xF, 0xF, false));
}
DEV float lane_bcast(float x, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
}
// The steps are v_add_f32 / v_max_f32 with the DPP operand in place, written by hand (through update_dpp the compiler spends
// a v_mov_b32_dpp, a zeroing v_mov_b32 and the addition per step): quad_perm [1,0,3,2], [2,3,0,1], row_ror 4, 8 leave every
// lane with its row's result; row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3 leave the wave's in lane 63.
// A DPP read needs two wait states after the VALU write of its source: values reduced together fill them for each other.
#define DABGPU_RED_STEPS(OP, R)                                                                  \
    OP " " R ", " R ", " R " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"      \
    OP " " R ", " R ", " R " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"      \
    OP " " R ", " R ", " R " row_ror:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                \
    OP " " R ", " R ", " R " row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                \
    OP " " R ", " R ", " R " row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"             \
    OP " " R ", " R ", " R " row_bcast:31 row_mask:0xc bank_mask:0xf"
DEV float wave_sum_dpp(float x)
{
    asm volatile("s_nop 1\n\t" DABGPU_RED_STEPS("v_add_f32_dpp", "%0") : "+v"(x));
    return lane_bcast(x, 63);
}
DEV float wave_max_dpp(float x)
{
    asm volatile("s_nop 1\n\t" DABGPU_RED_STEPS("v_max_f32_dpp", "%0") : "+v"(x));
    return lane_bcast(x, 63);
}
#undef DABGPU_RED_STEPS
// four values at once (each step's four instructions are independent: no wait states to pad); OP0 is the first value's
// operation, the other three are sums
#define DABGPU_RED4_STEP(OP0, CTRL)                                                              \
    OP0 " %0, %0, %0 " CTRL "\n\tv_add_f32_dpp %1, %1, %1 " CTRL "\n\t"                          \
    "v_add_f32_dpp %2, %2, %2 " CTRL "\n\tv_add_f32_dpp %3, %3, %3 " CTRL "\n\t"
#define DABGPU_RED4(OP0)                                                                         \
    "s_nop 1\n\t"                                                                               \
    DABGPU_RED4_STEP(OP0, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")                      \
    DABGPU_RED4_STEP(OP0, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")                      \
    DABGPU_RED4_STEP(OP0, "row_ror:4 row_mask:0xf bank_mask:0xf")                                \
    DABGPU_RED4_STEP(OP0, "row_ror:8 row_mask:0xf bank_mask:0xf")                                \
    DABGPU_RED4_STEP(OP0, "row_bcast:15 row_mask:0xa bank_mask:0xf")                             \
    DABGPU_RED4_STEP(OP0, "row_bcast:31 row_mask:0xc bank_mask:0xf")
DEV void wave_sum4_dpp(float &a, float &b, float &c, float &d)
{
    asm volatile(DABGPU_RED4("v_add_f32_dpp") "s_nop 0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    a = lane_bcast(a, 63); b = lane_bcast(b, 63); c = lane_bcast(c, 63); d = lane_bcast(d, 63);
}
DEV void wave_max_sum3_dpp(float &mx, float &b, float &c, float &d)
{
    asm volatile(DABGPU_RED4("v_max_f32_dpp") "s_nop 0" : "+v"(mx), "+v"(b), "+v"(c), "+v"(d));
    mx = lane_bcast(mx, 63); b = lane_bcast(b, 63); c = lane_bcast(c, 63); d = lane_bcast(d, 63);
}
#undef DABGPU_RED4
#undef DABGPU_RED4_STEP
// x, y summed over the four lanes of every DPP quad
DEV void quad_sum2_dpp(float &x, float &y)
{
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                 : "+v"(x), "+v"(y));
}
// x, y summed over the sixteen lanes of every DPP row (every lane of the row ends up with the sum)
DEV void row16_sum2_dpp(float &x, float &y)
{
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf"
                 : "+v"(x), "+v"(y));
}
// two values (one wait state between the steps left to pad)
#define DABGPU_RED2_STEP(OP0, CTRL)                                                              \
    "s_nop 0\n\t" OP0 " %0, %0, %0 " CTRL "\n\tv_add_f32_dpp %1, %1, %1 " CTRL "\n\t"
#define DABGPU_RED2(OP0)                                                                         \
    "s_nop 0\n\t"                                                                               \
    DABGPU_RED2_STEP(OP0, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")                      \
    DABGPU_RED2_STEP(OP0, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")                      \
    DABGPU_RED2_STEP(OP0, "row_ror:4 row_mask:0xf bank_mask:0xf")                                \
    DABGPU_RED2_STEP(OP0, "row_ror:8 row_mask:0xf bank_mask:0xf")                                \
    DABGPU_RED2_STEP(OP0, "row_bcast:15 row_mask:0xa bank_mask:0xf")                             \
    DABGPU_RED2_STEP(OP0, "row_bcast:31 row_mask:0xc bank_mask:0xf")
DEV void wave_sum2_dpp(float &a, float &b)
{
    asm volatile(DABGPU_RED2("v_add_f32_dpp") "s_nop 1" : "+v"(a), "+v"(b));
    a = lane_bcast(a, 63); b = lane_bcast(b, 63);
}
DEV void wave_max_sum_dpp(float &mx, float &b)
{
    asm volatile(DABGPU_RED2("v_max_f32_dpp") "s_nop 1" : "+v"(mx), "+v"(b));
    mx = lane_bcast(mx, 63); b = lane_bcast(b, 63);
}
#undef DABGPU_RED2
#undef DABGPU_RED2_STEP

// Gain of one OFDM symbol inside the fused kernel.  One pass: the DC bin of every
// symbol is zero by construction (reference src/OfdmGenerator.cpp:209-210), so the
// time-domain mean is rounding noise and var = E[x^2] - mean^2 has no cancellation.
// Per-lane partial sums and the DPP wave reduction in fp32, waves combined in float64.
// `redd` must alternate between two scratch areas from call to call (one barrier only).
template <int T> DEV float symbol_gain_fused(const cf *v, const GainParams &gp, double *redd, int t,
                                              bool on)
{
    constexpr double invN = 1.0 / (8 * T);
    constexpr int NW = (T + 63) / 64;
    float *red = reinterpret_cast<float *>(redd);
    if (gp.mode == 0) return 512.0f;
    if (gp.mode == 1) {
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) m = fmaxf(m, fmaxf(fabsf(v[i].x), fabsf(v[i].y)));
        m = wave_max_dpp(on ? m : 0.f);
        if (NW > 1) {
            if ((t & 63) == 0) red[t >> 6] = m;
            lds_barrier();
            m = red[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) m = fmaxf(m, red[w]);
        }
        return ((int)m != 0) ? 32767.0f * fast_rcp(m) : 1.0f;
    }
    // per-lane partial sums of 8 samples in fp32: their rounding errors are independent
    // across the 256 lanes and average out (~1e-8 on the total); the cross-lane tree is fp32
    // too, the cross-wave combine and the variance formula are float64
    float sr = 0.f, si = 0.f, qr = 0.f, qi = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sr += v[i].x; si += v[i].y;
        qr = fmaf(v[i].x, v[i].x, qr); qi = fmaf(v[i].y, v[i].y, qi);
    }
    float f0 = on ? sr : 0.f, f1 = on ? si : 0.f, f2 = on ? qr : 0.f, f3 = on ? qi : 0.f;
    wave_sum4_dpp(f0, f1, f2, f3);
    if (NW > 1) {
        if ((t & 63) == 0) {
            red[4 * (t >> 6)] = f0; red[4 * (t >> 6) + 1] = f1;
            red[4 * (t >> 6) + 2] = f2; red[4 * (t >> 6) + 3] = f3;
        }
        lds_barrier();
        f0 = f1 = f2 = f3 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            f0 += red[4 * w]; f1 += red[4 * w + 1];
            f2 += red[4 * w + 2]; f3 += red[4 * w + 3];
        }
    }
    const double d0 = f0, d1 = f1, d2 = f2, d3 = f3;
    const double mr = d0 * invN, mi = d1 * invN;
    const float vr = fast_sqrt((float)fmax(d2 * invN - mr * mr, 0.0)) * gp.var_variance,
                vi = fast_sqrt((float)fmax(d3 * invN - mi * mi, 0.0)) * gp.var_variance;
    if ((int)vr == 0) return 1.0f;
    return 32767.0f * fast_rcp(fmaxf(vr, vi));
}

// ---------------------------------------------------------------------------
// FIR over the LDS stream buffer: lane computes R consecutive outputs starting
// at j0; taps are wave-uniform (SGPR operands).  out[j] = sum_k taps[k]*sb[j+k]
// accumulated in tap order (reference src/FIRFilter.cpp:168-184; fused
// multiply-add instead of mul+add: float tolerance class).
// The stream buffer is padded one slot per 8 samples (fir_pad): lane l starts at sample 8 l, and a lane
// stride of 64 bytes would put the 64 lanes of a wave on four banks (16-way conflicts); 72 bytes spreads
// them over all 64.
DEV constexpr int fir_pad(int j) { return j + (j >> 3); }
// taps as a kernel argument by value: they arrive in SGPRs through scalar loads (read through a pointer
// they come as vector loads + v_readlane, with a hazard nop in front of every multiply)
template <int NTP> struct FirTaps { float t[NTP]; };
// lane = &sb[fir_pad(8 l)] = sb + 9 l: every access below is lane + a compile-time offset
template <int NTP, int R> DEV void fir_block(const cf *__restrict__ lane, const FirTaps<NTP> &taps, cf *acc)
{
    static_assert(R == 8, "the padded addressing assumes 8 outputs per lane");
    constexpr int G = 8;  // taps per window refill
    cf w[R + G - 1];
#pragma unroll
    for (int i = 0; i < R + G - 1; ++i) w[i] = lane[fir_pad(i)];
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = mk(0.f, 0.f);
#pragma unroll
    for (int g = 0; g < NTP / G; ++g) {
#pragma unroll
        for (int jj = 0; jj < G; ++jj) {
            const float tp = taps.t[g * G + jj];
#pragma unroll
            for (int i = 0; i < R; ++i) acc[i] = w[i + jj] * tp + acc[i];   // one v_pk_fma_f32 (re, im) per tap
        }
        if (g + 1 < NTP / G) {
#pragma unroll
            for (int i = 0; i < R - 1; ++i) w[i] = w[i + G];
#pragma unroll
            for (int i = R - 1; i < R + G - 1; ++i) w[i] = lane[fir_pad((g + 1) * G + i)];
        }
        // keep the scheduler from hoisting every group's LDS loads to the top of the
        // unrolled block (that is what drove the kernel past 128 VGPRs into scratch)
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---------------------------------------------------------------------------
// f-2 fused into the chain's last store: cf32 -> s16 with FormatConverter's range test, truncation toward zero and
// clipped-component count (reference src/FormatConverter.cpp:111-143), one 4-byte word per complex sample.
// FormatConverter s16 on one sample (src/FormatConverter.cpp:111-139: compare against INT16_MIN / MAX in float, count,
// else truncate toward zero) in 9 instructions instead of 16:
//   value: v_cvt_i32_f32 (toward zero, saturating, NaN -> 0) on both parts, then v_cvt_pk_i16_i32, which saturates to
//          16 bits and packs -- the clipped values are exactly the saturated ones;
//   count: x > 32767 or x < -32768  <=>  |x + 0.5| > 32767.5 (the sum is exact wherever the comparison is close: |x| <
//          2^16 has an ulp of 2^-8 or finer), one addition, one compare with |.|, one add-with-carry per part.
DEV int cvt_i32_sat(float x)
{
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));     // (a C cast of an out-of-range float is undefined; the instruction is not)
    return r;
}
// FormatConverter on one component, any format (src/FormatConverter.cpp:111-170): FMT 1 = s16, 2 = u8 (x + 128), 3 = s8;
// range test in float, count, else truncate toward zero.  (format_kernel; the fused u8 / s8 stores below)
template <int FMT> DEV int format_one(float x, unsigned &clipped)
{
    constexpr float lo = FMT == 1 ? -32768.0f : (FMT == 2 ? 0.0f : -128.0f);
    constexpr float hi = FMT == 1 ? 32767.0f : (FMT == 2 ? 255.0f : 127.0f);
    // (u8: the reference adds 128 to the ROUNDED float sample.  In a kernel's store epilogue the compiler would contract the
    // sum into the multiply that produced x -- 35.999992 + 128 is 164.0 in fp32, 163 after truncation when fused -- so x is made
    // opaque first: an empty asm, no instruction.)
    if (FMT == 2) asm("" : "+v"(x));
    const float v = FMT == 2 ? x + 128.0f : x;
    if (v < lo) { ++clipped; return (int)lo; }
    if (v > hi) { ++clipped; return (int)hi; }
    return (int)v;                       // v_cvt_i32_f32: toward zero, NaN -> 0
}
// u8 / s8 fused into the chain's last store: one 2-byte word per complex sample, the components as format_kernel forms them
template <int FMT> DEV unsigned short b8_pack(cf y, unsigned &clipped)
{
    static_assert(FMT == 2 || FMT == 3, "u8 or s8");
    const int re = format_one<FMT>(y.x, clipped), im = format_one<FMT>(y.y, clipped);
    return (unsigned short)((unsigned)(re & 0xff) | ((unsigned)(im & 0xff) << 8));
}
DEV uint32_t s16_pack(cf y, unsigned &clipped)
{
    clipped += (__builtin_fabsf(y.x + 0.5f) > 32767.5f ? 1u : 0u) + (__builtin_fabsf(y.y + 0.5f) > 32767.5f ? 1u : 0u);
    typedef short s2_ __attribute__((ext_vector_type(2)));
    const s2_ p = __builtin_amdgcn_cvt_pk_i16(cvt_i32_sat(y.x), cvt_i32_sat(y.y));
    return __builtin_bit_cast(uint32_t, p);
}
// per-workgroup epilogue of a kernel that stored s16: the lanes' clip counts, summed per wave, onto the call's counter
DEV void s16_flush_count(unsigned nclip, unsigned long long *total)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nclip += __shfl_xor(nclip, o, 64);
    if ((threadIdx.x & 63) == 0 && nclip) atomicAdd(total, (unsigned long long)nclip);
}


}  // namespace
}  // namespace dabgpu
