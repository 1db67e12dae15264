// stage_kernels.hip -- the stand-alone stage kernels: one per reference plugin (the per-stage drop-ins of include/dabgpu.h)
// and the non-fused fallbacks of the chain.
#include "device_common.h"

namespace dabgpu {

// ===========================================================================
// Stand-alone stage kernels: the per-plugin drop-ins.  These are thin,
// memory-bound, coalesced; the fused kernel above is the production path.
namespace {

// a1 QpskSymbolMapper (src/QpskSymbolMapper.cpp:138-156): one lane per output
// pair of carriers -> one 16-byte store.
__global__ void qpsk_kernel(const uint8_t *__restrict__ in, size_t npairs, int K,
                            float4 *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npairs) return;
    const size_t car = 2 * i;                 // global carrier index
    const size_t blk = car / (size_t)K;
    const int n = (int)(car - blk * (size_t)K);
    const uint8_t *b = in + blk * (size_t)(K / 4);
    const unsigned ib = b[n >> 3], qb = b[(K >> 3) + (n >> 3)];
    const int sh = 6 - (n & 7);               // n even: bits (7-n&7) and (6-n&7)
    const float c = kSqrtHalf;
    float4 o;
    o.x = ((ib >> (sh + 1)) & 1u) ? -c : c;
    o.y = ((qb >> (sh + 1)) & 1u) ? -c : c;
    o.z = ((ib >> sh) & 1u) ? -c : c;
    o.w = ((qb >> sh) & 1u) ? -c : c;
    out[i] = o;
}

// a2 FrequencyInterleaver (src/FrequencyInterleaver.cpp:103-126) as a gather:
// out[s][k] = in[s][src[k]] -> coalesced stores.
__global__ void freq_interleave_kernel(const cf *__restrict__ in, size_t nsamples, int K,
                                       const uint16_t *__restrict__ src, cf *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsamples) return;
    const size_t s = i / (size_t)K;
    const int k = (int)(i - s * (size_t)K);
    out[i] = in[s * (size_t)K + src[k]];
}

// a3 PhaseReference (src/PhaseReference.cpp:126-171)
__global__ void phase_reference_kernel(const uint8_t *__restrict__ q, int K, cf *__restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const unsigned p = q[k] & 3u;
    out[k] = mk(p == 0 ? 1.f : (p == 2 ? -1.f : 0.f), p == 1 ? 1.f : (p == 3 ? -1.f : 0.f));
}

// a4 DifferentialModulator (src/DifferentialModulator.cpp:65-76) for ARBITRARY
// complex input: the serial, non-contracted fp32 product chain of the
// reference, one lane per carrier -> bit-exact.
__global__ void diff_mod_kernel(const cf *__restrict__ phase, const cf *__restrict__ data,
                                size_t nsym, int K, cf *__restrict__ out)
{
#pragma clang fp contract(off)  // round products and sums separately (the HIP *_rn helpers are plain operators)
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    cf y = phase[k];
    out[k] = y;
    for (size_t s = 0; s < nsym; ++s) {
        const cf x = data[s * (size_t)K + k];
        const float rr = y.x * x.x, ii = y.y * x.y;
        const float ri = y.x * x.y, ir = y.y * x.x;
        y = mk(rr - ii, ri + ir);
        out[(s + 1) * (size_t)K + k] = y;
    }
}

// a7 GainControl stand-alone: one workgroup per symbol pair (statistics symbol,
// output symbol); N/8 lanes, 8 samples per lane.
//
// Gain mode var replays the reference's x86 code path operation for operation (src/GainControl.cpp:251-340):
// the symbol is N/2 vectors {re0, im0, re1, im1}; four independent fp32 running means (mean += (x - mean) / count),
// the two means of each part averaged, four running variances against those, averaged, sqrt, times var_variance.
// The recurrence is serial in the sample index, so four lanes -- one per SSE lane -- walk the symbol (staged in LDS)
// while the rest of the workgroup waits: ~2 x N/2 dependent divisions per symbol, microseconds, and the drop-in
// stage then returns the reference's gain BIT FOR BIT instead of the exact population variance the fused chain
// uses (which differs from this recurrence by up to 5e-7 relative).  Products and sums are rounded separately.
// (sym may be global memory: the samples of sixteen steps are requested together, a block ahead of the serial chain that
// consumes them -- nvec is a multiple of 16 for every transmission mode, N / 2 >= 128)
DEV float gain_var_replay(const float *sym, int nvec, float var_variance, int l)
{
#pragma clang fp contract(off)
    constexpr int kAhead = 16;
    float mean = 0.f;
    float cur[kAhead], nxt[kAhead];
#pragma unroll
    for (int i = 0; i < kAhead; ++i) cur[i] = sym[4 * i + l];
    for (int v0 = 0; v0 < nvec; v0 += kAhead) {
        const int vn = v0 + kAhead < nvec ? v0 + kAhead : 0;             // (the last block requests the first again: pass 2 starts there)
#pragma unroll
        for (int i = 0; i < kAhead; ++i) nxt[i] = sym[4 * (vn + i) + l];
#pragma unroll
        for (int i = 0; i < kAhead; ++i) {
            const float d = cur[i] - mean;
            mean = mean + d / (float)(v0 + i + 1);
        }
#pragma unroll
        for (int i = 0; i < kAhead; ++i) cur[i] = nxt[i];
    }
    // lanes {0,2} hold re, {1,3} hold im
    const float other = __shfl_xor(mean, 2, 64);
    const float m2 = (mean + other) * 0.5f;
    float var = 0.f;
    for (int v0 = 0; v0 < nvec; v0 += kAhead) {
        const int vn = v0 + kAhead < nvec ? v0 + kAhead : v0;
#pragma unroll
        for (int i = 0; i < kAhead; ++i) nxt[i] = sym[4 * (vn + i) + l];
#pragma unroll
        for (int i = 0; i < kAhead; ++i) {
            const float diff = cur[i] - m2;
            const float sq = diff * diff;
            const float d = sq - var;
            var = var + d / (float)(v0 + i + 1);
        }
#pragma unroll
        for (int i = 0; i < kAhead; ++i) cur[i] = nxt[i];
    }
    const float merged = (var + __shfl_xor(var, 2, 64)) * 0.5f;       // lanes 0 and 1: re and im
    const float sd = sqrtf(merged) * var_variance;
    const int quad = (int)(threadIdx.x & 63u) & ~3;                       // (every group of four lanes walks a symbol of its own)
    const float sd_re = __shfl(sd, quad, 64), sd_im = __shfl(sd, quad + 1, 64);
    if ((int)sd_re == 0) return 1.0f;
    return 32767.0f / (sd_re > sd_im ? sd_re : sd_im);
}

template <int LOGN> __global__ void gain_kernel(const cf *__restrict__ in, size_t nsym,
                                                GainParams gp, cf *__restrict__ out)
{
    constexpr int N = 1 << LOGN, T = N / 8;
    __shared__ double red[16];
    __shared__ float stat[2 * N];
    const size_t s = blockIdx.x;
    const int t = threadIdx.x;
    const bool on = t < T;
    const int tt = on ? t : 0;
    const size_t src = (s == 0 && nsym > 1) ? 1 : s;  // src/GainControl.cpp:139-144
    cf v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = in[src * N + tt + T * m];
    float g;
    if (gp.mode == 2) {
        if (on) {
#pragma unroll
            for (int m = 0; m < 8; ++m) reinterpret_cast<cf *>(stat)[t + T * m] = v[m];
        }
        __syncthreads();
        if (t < 64) {                                   // the first wave; lanes 0..3 carry the four statistics
            const float gv = gain_var_replay(stat, N / 2, gp.var_variance, t & 3);
            if (t == 0) reinterpret_cast<float *>(red)[0] = gv;
        }
        __syncthreads();
        g = reinterpret_cast<float *>(red)[0];
    } else {
        g = symbol_gain<T>(v, gp, red, tt, on);
    }
    {
#pragma clang fp contract(off)
        g = g * gp.constant;
    }
    if (!on) return;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const cf x = in[s * N + t + T * m];
        out[s * N + t + T * m] = cscale(x, g);
    }
}

// The reference's gain scalars inside a CHAIN call (dabgpu_set_gain_rounding(ctx, 1); src/GainControl.cpp:251-340): the
// recurrence is serial in the sample index, so a chain that wants it bit for bit cannot form it inside the frame kernel
// (2 x N/2 dependent divisions per symbol against a few microseconds per symbol there).  Instead the frame kernel stops
// after the IFFT, this kernel walks SIXTEEN symbols per wave -- four lanes each, one per SSE lane, straight from the
// unscaled symbols in memory (sixteen 16-byte segments per load, every line read twice and served by the cache the
// second time) -- and gain_apply_kernel scales the symbols in place before the guard interval / FIRFilter kernels.
// gains[frame * nsym + s] = scalar(symbol s) * constant, the two roundings of src/GainControl.cpp:146-155.
__global__ void gain_replay_kernel(const cf *__restrict__ x0, size_t total_syms, int nsym, int N, GainParams gp,
                                   float *__restrict__ gains, float *__restrict__ gain1)
{
    const size_t sym = (size_t)blockIdx.x * 16 + (threadIdx.x >> 2);
    const bool on = sym < total_syms;
    const float *p = reinterpret_cast<const float *>(x0 + (on ? sym : 0) * (size_t)N);
    const float gv = gain_var_replay(p, N / 2, gp.var_variance, threadIdx.x & 3);
    float g;
    {
#pragma clang fp contract(off)
        g = gv * gp.constant;
    }
    if (on && (threadIdx.x & 3) == 0) {
        gains[sym] = g;
        // the multiplier of symbol 1 (what the null symbol -- TII -- is scaled by, src/GainControl.cpp:139-144)
        if (gain1 && (int)(sym % (size_t)nsym) == (nsym > 1 ? 1 : 0)) gain1[sym / (size_t)nsym] = g;
    }
}

// x[frame][s][n] *= gains[frame][s], symbol 0 with symbol 1's (src/GainControl.cpp:139-144).  Four samples per lane.  (Chains that
// stop at GainControl; with a guard interval behind it the guard kernels below take the multipliers as they gather.)
__global__ void gain_apply_kernel(cf *__restrict__ x0, size_t n_frames, int nsym, int N, const float *__restrict__ gains)
{
    const size_t per_sym = (size_t)N / 4;                 // lanes per symbol
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_frames * (size_t)nsym * per_sym) return;
    const size_t symi = i / per_sym, f = symi / (size_t)nsym;
    const int s = (int)(symi % (size_t)nsym);
    const int src = (s == 0 && nsym > 1) ? 1 : s;
    const float g = gains[f * (size_t)nsym + (size_t)src];
    float4 *q = reinterpret_cast<float4 *>(x0 + symi * (size_t)N) + 2 * (i % per_sym);
    float4 a = q[0], b = q[1];
    a.x *= g; a.y *= g; a.z *= g; a.w *= g;
    b.x *= g; b.y *= g; b.z *= g; b.w *= g;
    q[0] = a; q[1] = b;
}

// a8 GuardIntervalInserter as a gather: sample p of a frame's output stream from the frame's
// (nb_symbols+1) x N IFFT output x0.
// Overlap 0 (src/GuardIntervalInserter.cpp:301-319): a pure copy.
// (segment s, offset o inside it) of stream position p
DEV void guard_locate(const Geometry &g, int p, int &s, int &o)
{
    if (p < g.null_size) { s = 0; o = p; }
    else { s = 1 + (p - g.null_size) / g.sym_size; o = (p - g.null_size) % g.sym_size; }
}

// Sample n of symbol s, times the symbol's multiplier where the caller has a table of them (gs: one frame's nsym multipliers
// from gain_replay_kernel; symbol 0 takes symbol 1's) -- the product rounded by itself, as GainControl's output is before the
// guard interval sees it.
DEV cf guard_sample(const cf *__restrict__ x0, const Geometry &g, const float *__restrict__ gs, int s, int n)
{
#pragma clang fp contract(off)
    const cf x = x0[(size_t)s * (size_t)g.N + (size_t)n];
    if (!gs) return x;
    const float m = gs[(s == 0 && g.nb_symbols > 0) ? 1 : s];
    return mk(x.x * m, x.y * m);
}

DEV cf guard_copy_at(const cf *__restrict__ x0, const Geometry &g, int s, int o, const float *__restrict__ gs = nullptr)
{
    const int cpl = (s == 0 ? g.null_size : g.sym_size) - g.N;
    const int n = o < cpl ? g.N - cpl + o : o - cpl;
    return guard_sample(x0, g, gs, s, n);
}

// Raised-cosine overlap W > 0 (src/GuardIntervalInserter.cpp:149-300): every output sample is its
// own symbol's sample times a window factor, plus (inside 2W-wide seams) one neighbour term.
// Products and the sum are rounded separately, as in the reference.
DEV cf guard_window_at(const cf *__restrict__ x0, const Geometry &g, int W, const float *__restrict__ win, int s,
                       int o, const float *__restrict__ gs = nullptr)
{
#pragma clang fp contract(off)  // products and sums rounded separately, like the reference
    const int N = g.N, nsym = g.nb_symbols + 1;
    const int seg = s == 0 ? g.null_size : g.sym_size;
    const int cpl = seg - N;
    const bool last = (s == nsym - 1);
    if (s >= 1 && o < W) {
        // overwritten first by the previous symbol's suffix (1/2 -> 0), then += own rising edge
        const float fs = win[W - 1 - o];
        const cf xq = guard_sample(x0, g, gs, s - 1, o);
        cf r = mk(xq.x * fs, xq.y * fs);
        const float fr = win[W + o];
        const cf xr = guard_sample(x0, g, gs, s, N - cpl + o);
        const float pr_ = xr.x * fr, pi_ = xr.y * fr;
        return mk(r.x + pr_, r.y + pi_);
    }
    const int n = o < cpl ? N - cpl + o : o - cpl;
    if (!last && o >= seg - W) {
        // falling half window 1 -> 1/2, then the next symbol's rising edge is added
        const int i2 = o - (seg - W);
        const float ff = win[2 * W - 1 - i2];
        const cf xc = guard_sample(x0, g, gs, s, n);
        const cf r = mk(xc.x * ff, xc.y * ff);
        const int cpn = g.sym_size - N;
        const cf xr = guard_sample(x0, g, gs, s + 1, N - cpn - W + i2);
        const float fr = win[i2];
        const float pr_ = xr.x * fr, pi_ = xr.y * fr;
        return mk(r.x + pr_, r.y + pi_);
    }
    return guard_sample(x0, g, gs, s, n);
}

__global__ void guard_copy_kernel(const cf *__restrict__ in, size_t n_frames, Geometry g,
                                  cf *__restrict__ out, const float *__restrict__ gains)
{
    const size_t tf = (size_t)g.null_size + (size_t)g.nb_symbols * (size_t)g.sym_size;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_frames * tf) return;
    const size_t f = i / tf;
    int s, o;
    guard_locate(g, (int)(i - f * tf), s, o);
    out[i] = guard_copy_at(in + f * (size_t)(g.nb_symbols + 1) * (size_t)g.N, g, s, o,
                           gains ? gains + f * (size_t)(g.nb_symbols + 1) : nullptr);
}

__global__ void guard_window_kernel(const cf *__restrict__ in, size_t n_frames, Geometry g, int W,
                                    const float *__restrict__ win, cf *__restrict__ out, const float *__restrict__ gains)
{
    const size_t tf = (size_t)g.null_size + (size_t)g.nb_symbols * (size_t)g.sym_size;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_frames * tf) return;
    const size_t f = i / tf;
    int s, o;
    guard_locate(g, (int)(i - f * tf), s, o);
    out[i] = guard_window_at(in + f * (size_t)(g.nb_symbols + 1) * (size_t)g.N, g, W, win, s, o,
                             gains ? gains + f * (size_t)(g.nb_symbols + 1) : nullptr);
}

// a9 FIRFilter stand-alone (src/FIRFilter.cpp:162-192): LDS-tiled look-ahead FIR,
// truncated at the end of each frame.
template <int NTP> __global__ __launch_bounds__(256)
void fir_kernel(const cf *__restrict__ in, size_t frame_samples, const FirTaps<NTP> taps,
                cf *__restrict__ out)
{
    constexpr int R = 8, TILE = 256 * R;
    __shared__ cf sb[fir_pad(TILE + NTP + R + 8) + 1];
    const size_t f = blockIdx.y;
    const size_t base = (size_t)blockIdx.x * TILE;
    const cf *fin = in + f * frame_samples;
    constexpr int LIMIT = TILE + NTP + R + 8, KMAX = (LIMIT + 255) / 256;
    cf fetched[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const size_t p = base + threadIdx.x + 256 * (size_t)k;
        fetched[k] = (p < frame_samples && (int)threadIdx.x + 256 * k < LIMIT) ? fin[p] : mk(0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if ((int)threadIdx.x + 256 * k < LIMIT) sb[fir_pad((int)threadIdx.x + 256 * k)] = fetched[k];
    lds_barrier();
    cf acc[R];
    const int j0 = threadIdx.x * R;
    fir_block<NTP, R>(sb + 9 * threadIdx.x, taps, acc);
    // a lane holds 8 consecutive outputs (64 bytes apart from its neighbour's): back through LDS so that
    // every store instruction writes 512 contiguous bytes
    lds_barrier();
#pragma unroll
    for (int i = 0; i < R; ++i) sb[fir_pad(j0 + i)] = acc[i];
    lds_barrier();
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const size_t p = base + threadIdx.x + 256 * (size_t)k;
        if (p < frame_samples) out[f * frame_samples + p] = sb[fir_pad((int)threadIdx.x + 256 * k)];
    }
}

// a8 + a9 in one pass for the chains that cannot use the frame kernel's fused epilogue (windowed guard
// interval, crest-factor reduction, filters longer than the cyclic prefix): the FIR's LDS tile is filled
// straight from the IFFT output through the guard-interval gather, so the guard-extended stream never
// goes to HBM (1.57 MB written + 1.57 MB read per Mode-I frame less).
template <int NTP> __global__ __launch_bounds__(256)
void guard_fir_kernel(const cf *__restrict__ in, Geometry g, int W, const float *__restrict__ win,
                      const FirTaps<NTP> taps, cf *__restrict__ out, const float *__restrict__ gains)
{
    constexpr int R = 8, TILE = 256 * R;
    __shared__ cf sb[fir_pad(TILE + NTP + R + 8) + 1];
    const size_t f = blockIdx.y;
    const int tf = g.null_size + g.nb_symbols * g.sym_size;
    const int base = (int)blockIdx.x * TILE;
    const cf *x0 = in + f * (size_t)(g.nb_symbols + 1) * (size_t)g.N;
    const float *gs = gains ? gains + f * (size_t)(g.nb_symbols + 1) : nullptr;
    // The lane's samples are 256 apart: locate the first one, then step (no division per sample).
    // All gathers are issued before the first LDS store, so the lane waits for memory once, not per sample.
    constexpr int LIMIT = TILE + NTP + R + 8, KMAX = (LIMIT + 255) / 256;
    int sg, og;
    guard_locate(g, min(base + (int)threadIdx.x, tf - 1), sg, og);
    cf fetched[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int p = base + (int)threadIdx.x + 256 * k;
        fetched[k] = mk(0.f, 0.f);
        if (p < tf && (int)threadIdx.x + 256 * k < LIMIT)
            fetched[k] = W > 0 ? guard_window_at(x0, g, W, win, sg, og, gs) : guard_copy_at(x0, g, sg, og, gs);
        og += 256;
        for (int len = sg == 0 ? g.null_size : g.sym_size; og >= len; len = g.sym_size) { og -= len; ++sg; }
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if ((int)threadIdx.x + 256 * k < LIMIT) sb[fir_pad((int)threadIdx.x + 256 * k)] = fetched[k];
    lds_barrier();
    cf acc[R];
    const int j0 = threadIdx.x * R;
    fir_block<NTP, R>(sb + 9 * threadIdx.x, taps, acc);
    // a lane holds 8 consecutive outputs (64 bytes apart from its neighbour's): back through LDS so that
    // every store instruction writes 512 contiguous bytes
    lds_barrier();
#pragma unroll
    for (int i = 0; i < R; ++i) sb[fir_pad(j0 + i)] = acc[i];
    lds_barrier();
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int p = base + (int)threadIdx.x + 256 * k;
        if (p < tf) out[f * (size_t)tf + (size_t)p] = sb[fir_pad((int)threadIdx.x + 256 * k)];
    }
}

// a11 MemlessPoly polynomial (src/MemlessPoly.cpp:237-276), literal constants.  The stand-alone drop-in is bound by
// memory, not arithmetic, so it rounds every product and every sum on its own -- the reference's default x86-64 build has
// no fused multiply-add -- and returns the reference's samples bit for bit (the resampler's fused epilogue keeps FMAs).
__global__ void poly_kernel(const float4 *__restrict__ in, size_t npairs, const float *__restrict__ am,
                            const float *__restrict__ pm, float4 *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npairs) return;
    const float a0 = am[0], a1 = am[1], a2 = am[2], a3 = am[3], a4 = am[4];
    const float p0 = pm[0], p1 = pm[1], p2 = pm[2], p3 = pm[3], p4 = pm[4];
    const float4 x = in[i];
    float4 y;
    auto one = [&](float xr, float xi, float &yr, float &yi) {
#pragma clang fp contract(off)
        const float m = xr * xr + xi * xi;
        const float a = a0 + m * (a1 + m * (a2 + m * (a3 + m * a4)));
        const float p = -1.0f * (p0 + m * (p1 + m * (p2 + m * (p3 + m * p4))));
        const float q = p * p;
        const float cr = (1.0f - q * (-0.5f + q * (0.486666f + q * (-0.00138888f))));
        const float ci = p * (1.0f + q * (0.166666f + q * (0.00833333f)));
        const float sr = xr * a, si = xi * a;
        yr = sr * cr - si * ci;
        yi = sr * ci + si * cr;
    };
    one(x.x, x.y, y.x, y.y);
    one(x.z, x.w, y.z, y.w);
    out[i] = y;
}

// a11 LUT mode (src/MemlessPoly.cpp:278-309)
__global__ void lut_kernel(const cf *__restrict__ in, size_t n, float scale,
                           const float *__restrict__ lut, cf *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const cf x = in[i];
    const float mag = hypotf(x.x, x.y);
    const unsigned scaled = (unsigned)(long long)rintf(mag * scale);
    const float l = lut[(scaled >> 27) & 31u];
    out[i] = mk(x.x * l, x.y * l);
}

inline unsigned blocks_for(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

}  // namespace

hipError_t launch_qpsk(const uint8_t *in, size_t nbytes, int K, float2 *out, hipStream_t s)
{
    const size_t npairs = nbytes * 2;
    if (npairs == 0) return hipSuccess;
    DABGPU_LAUNCH(qpsk_kernel, dim3(blocks_for(npairs, 256)), dim3(256), 0, s, in, npairs, K,
                       reinterpret_cast<float4 *>(out));
    return hipGetLastError();
}

hipError_t launch_freq_interleave(const float2 *in, size_t nsamples, int K,
                                  const uint16_t *src_carrier, float2 *out, hipStream_t s)
{
    if (nsamples == 0) return hipSuccess;
    DABGPU_LAUNCH(freq_interleave_kernel, dim3(blocks_for(nsamples, 256)), dim3(256), 0, s, in,
                       nsamples, K, src_carrier, out);
    return hipGetLastError();
}

hipError_t launch_phase_reference(const uint8_t *phase_q, int K, float2 *out, hipStream_t s)
{
    DABGPU_LAUNCH(phase_reference_kernel, dim3(blocks_for((size_t)K, 256)), dim3(256), 0, s,
                       phase_q, K, out);
    return hipGetLastError();
}

hipError_t launch_diff_mod(const float2 *phase, const float2 *data, size_t nsym_data, int K,
                           float2 *out, hipStream_t s)
{
    DABGPU_LAUNCH(diff_mod_kernel, dim3(blocks_for((size_t)K, 64)), dim3(64), 0, s, phase, data,
                       nsym_data, K, out);
    return hipGetLastError();
}

hipError_t launch_gain(const float2 *in, size_t nsym, int N, GainParams gp, float2 *out,
                       hipStream_t s)
{
    if (nsym == 0) return hipSuccess;
    const dim3 grid((unsigned)nsym);
    switch (N) {
        case 256: DABGPU_LAUNCH(gain_kernel<8>, grid, dim3(64), 0, s, in, nsym, gp, out); break;
        case 512: DABGPU_LAUNCH(gain_kernel<9>, grid, dim3(64), 0, s, in, nsym, gp, out); break;
        case 1024: DABGPU_LAUNCH(gain_kernel<10>, grid, dim3(128), 0, s, in, nsym, gp, out); break;
        case 2048: DABGPU_LAUNCH(gain_kernel<11>, grid, dim3(256), 0, s, in, nsym, gp, out); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_gain_replay(float2 *x0, size_t n_frames, int nsym, int N, GainParams gp, float *gains, float *gain1,
                              bool apply, hipStream_t s)
{
    const size_t total = n_frames * (size_t)nsym;
    if (total == 0) return hipSuccess;
    DABGPU_LAUNCH(gain_replay_kernel, dim3(blocks_for(total, 16)), dim3(64), 0, s, x0, total, nsym, N, gp, gains, gain1);
    if (apply)
        DABGPU_LAUNCH(gain_apply_kernel, dim3(blocks_for(total * (size_t)(N / 4), 256)), dim3(256), 0, s, x0, n_frames, nsym,
                      N, gains);
    return hipGetLastError();
}

hipError_t launch_guard_copy(const float2 *in, size_t n_frames, Geometry g, float2 *out,
                             hipStream_t s, const float *gains)
{
    const size_t n = n_frames * ((size_t)g.null_size + (size_t)g.nb_symbols * (size_t)g.sym_size);
    if (n == 0) return hipSuccess;
    DABGPU_LAUNCH(guard_copy_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, in, n_frames, g,
                       out, gains);
    return hipGetLastError();
}

hipError_t launch_guard_window(const float2 *in, size_t n_frames, Geometry g, int overlap,
                               const float *window, float2 *out, hipStream_t s, const float *gains)
{
    const size_t n = n_frames * ((size_t)g.null_size + (size_t)g.nb_symbols * (size_t)g.sym_size);
    if (n == 0) return hipSuccess;
    DABGPU_LAUNCH(guard_window_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, in, n_frames,
                       g, overlap, window, out, gains);
    return hipGetLastError();
}

hipError_t launch_fir(const float2 *in, size_t frame_samples, size_t n_frames, const float *taps,
                      int ntaps, float2 *out, hipStream_t s)
{
    if (frame_samples == 0 || n_frames == 0) return hipSuccess;
    if (ntaps < 1 || ntaps > kMaxTapsUnfused) return hipErrorInvalidValue;
    const dim3 grid(blocks_for(frame_samples, 256 * 8), (unsigned)n_frames);
    if (ntaps <= 48) {
        FirTaps<48> t{};
        std::copy(taps, taps + ntaps, t.t);
        DABGPU_LAUNCH(fir_kernel<48>, grid, dim3(256), 0, s, in, frame_samples, t, out);
    } else if (ntaps <= 128) {
        FirTaps<128> t{};
        std::copy(taps, taps + ntaps, t.t);
        DABGPU_LAUNCH(fir_kernel<128>, grid, dim3(256), 0, s, in, frame_samples, t, out);
    } else {
        FirTaps<512> t{};
        std::copy(taps, taps + ntaps, t.t);
        DABGPU_LAUNCH(fir_kernel<512>, grid, dim3(256), 0, s, in, frame_samples, t, out);
    }
    return hipGetLastError();
}

hipError_t launch_guard_fir(const float2 *in, size_t n_frames, Geometry g, int overlap, const float *window,
                            const float *taps, int ntaps, float2 *out, hipStream_t s, const float *gains)
{
    if (n_frames == 0) return hipSuccess;
    if (ntaps < 1 || ntaps > kMaxTapsUnfused) return hipErrorInvalidValue;
    const size_t tf = (size_t)g.null_size + (size_t)g.nb_symbols * (size_t)g.sym_size;
    const dim3 grid(blocks_for(tf, 256 * 8), (unsigned)n_frames);
    if (ntaps <= 48) {
        FirTaps<48> t{};
        std::copy(taps, taps + ntaps, t.t);
        DABGPU_LAUNCH(guard_fir_kernel<48>, grid, dim3(256), 0, s, in, g, overlap, window, t, out, gains);
    } else if (ntaps <= 128) {
        FirTaps<128> t{};
        std::copy(taps, taps + ntaps, t.t);
        DABGPU_LAUNCH(guard_fir_kernel<128>, grid, dim3(256), 0, s, in, g, overlap, window, t, out, gains);
    } else {
        FirTaps<512> t{};
        std::copy(taps, taps + ntaps, t.t);
        DABGPU_LAUNCH(guard_fir_kernel<512>, grid, dim3(256), 0, s, in, g, overlap, window, t, out, gains);
    }
    return hipGetLastError();
}

hipError_t launch_poly(const float2 *in, size_t nsamples, const float *am, const float *pm,
                       float2 *out, hipStream_t s)
{
    if (nsamples == 0) return hipSuccess;
    // pairs of samples as float4; an odd tail sample is handled as a second tiny launch
    const size_t npairs = nsamples / 2;
    if (npairs)
        DABGPU_LAUNCH(poly_kernel, dim3(blocks_for(npairs, 256)), dim3(256), 0, s,
                           reinterpret_cast<const float4 *>(in), npairs, am, pm,
                           reinterpret_cast<float4 *>(out));
    if (nsamples & 1) {
        // process the last sample through the LUT-free scalar path: reuse poly on an overlapping pair
        return hipErrorInvalidValue;  // odd lengths never occur (frame sizes are even)
    }
    return hipGetLastError();
}

hipError_t launch_lut(const float2 *in, size_t nsamples, float scale, const float *lut, float2 *out,
                      hipStream_t s)
{
    if (nsamples == 0) return hipSuccess;
    DABGPU_LAUNCH(lut_kernel, dim3(blocks_for(nsamples, 256)), dim3(256), 0, s, in, nsamples,
                       scale, lut, out);
    return hipGetLastError();
}

namespace {

// ===========================================================================
// a12 CicEqualizer (reference src/CicEqualizer.cpp:66-91): every carrier times its real gain.
__global__ void cic_kernel(const cf *__restrict__ in, size_t n, int K, const float *__restrict__ filter,
                           cf *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float f = filter[i % (size_t)K];
    const cf x = in[i];
    out[i] = mk(x.x * f, x.y * f);
}

// ===========================================================================
// f-4 TII (reference src/TII.cpp:172-211): the sparse TII symbol from the phase reference symbol.
// Gather form of the reference's loop "if (Acp[i]) { out[i] = in[i]; out[i+1] = old ? in[i+1] : in[i]; }".
__global__ void tii_kernel(const cf *__restrict__ in, const uint8_t *__restrict__ acp, int K, int old_variant,
                           int insert, cf *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    cf y = mk(0.f, 0.f);
    if (insert) {
        if (acp[i]) y = in[i];
        else if (i > 0 && acp[i - 1]) y = old_variant ? in[i] : in[i - 1];
    }
    out[i] = y;
}

// Everything after the IFFT is linear, and the null symbol takes the gain of symbol 1: on a frame
// that carries TII the stream is the stream with a blank null symbol plus g_1 times a constant
// segment (the TII symbol through IFFT, guard interval and FIR, computed once per setting).
__global__ void tii_add_kernel(cf *__restrict__ out, size_t stride, const cf *__restrict__ seg, int seg_len,
                               const float *__restrict__ gain1, int insert0)
{
    const int f = blockIdx.y;
    if (((f & 1) == 0) != (insert0 != 0)) return;     // TII::m_insert toggles per frame
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= seg_len) return;
    const float g = gain1 ? gain1[f] : 1.0f;
    cf *o = out + (size_t)f * stride + n;
    const cf x = seg[n], y = *o;
    *o = mk(fmaf(g, x.x, y.x), fmaf(g, x.y, y.y));
}

// ===========================================================================
// f-2 FormatConverter, float input (reference src/FormatConverter.cpp:111-178): range test
// against the integer limits (clipped components counted), otherwise float -> integer by
// truncation toward zero; u8 adds 128.0f first.  FMT: 1 = s16, 2 = u8, 3 = s8.
// HBM-bound elementwise: 8 floats per lane and tile (two 16-byte loads, one 16- or 8-byte store), a workgroup walks
// `tiles` consecutive tiles; the clip count is reduced per workgroup and added to a device counter -- ONE atomic per
// workgroup (a badly scaled stream clips everywhere: one atomic per wave and tile then was 6 million additions to one address
// for 8192 frames, twenty times the kernel's own time).
// (format_one: device_common.h)
template <int FMT> __global__ __launch_bounds__(256)
void format_kernel(const float *__restrict__ in, size_t n, void *__restrict__ out,
                   unsigned long long *__restrict__ clipped_total, int tiles)
{
    __shared__ unsigned wave_sum[4];
    unsigned clipped = 0;
    for (int k = 0; k < tiles; ++k) {
    const size_t i0 = (((size_t)blockIdx.x * tiles + k) * 256 + threadIdx.x) * 8;
    if (i0 >= n) break;
    if (i0 + 8 <= n) {
        const float4 a = reinterpret_cast<const float4 *>(in + i0)[0];
        const float4 b = reinterpret_cast<const float4 *>(in + i0)[1];
        const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        int y[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = format_one<FMT>(x[k], clipped);
        if (FMT == 1) {
            uint4 w;
            w.x = (unsigned)(y[0] & 0xffff) | ((unsigned)y[1] << 16);
            w.y = (unsigned)(y[2] & 0xffff) | ((unsigned)y[3] << 16);
            w.z = (unsigned)(y[4] & 0xffff) | ((unsigned)y[5] << 16);
            w.w = (unsigned)(y[6] & 0xffff) | ((unsigned)y[7] << 16);
            reinterpret_cast<uint4 *>(reinterpret_cast<int16_t *>(out) + i0)[0] = w;
        } else {
            uint2 w;
            w.x = (unsigned)(y[0] & 0xff) | ((unsigned)(y[1] & 0xff) << 8) | ((unsigned)(y[2] & 0xff) << 16) |
                  ((unsigned)y[3] << 24);
            w.y = (unsigned)(y[4] & 0xff) | ((unsigned)(y[5] & 0xff) << 8) | ((unsigned)(y[6] & 0xff) << 16) |
                  ((unsigned)y[7] << 24);
            reinterpret_cast<uint2 *>(reinterpret_cast<uint8_t *>(out) + i0)[0] = w;
        }
    } else {
        for (size_t i = i0; i < n; ++i) {
            const int y = format_one<FMT>(in[i], clipped);
            if (FMT == 1) reinterpret_cast<int16_t *>(out)[i] = (int16_t)y;
            else reinterpret_cast<uint8_t *>(out)[i] = (uint8_t)y;
        }
    }
    }
    unsigned tot = clipped;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned all = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
        if (all) atomicAdd(clipped_total, (unsigned long long)all);
    }
}

}  // namespace

hipError_t launch_cic(const float2 *in, size_t nsamples, int K, const float *filter, float2 *out, hipStream_t s)
{
    if (nsamples == 0) return hipSuccess;
    DABGPU_LAUNCH(cic_kernel, dim3(blocks_for(nsamples, 256)), dim3(256), 0, s, in, nsamples, K, filter, out);
    return hipGetLastError();
}

hipError_t launch_tii(const float2 *in, const uint8_t *acp, int K, int old_variant, int insert, float2 *out,
                      hipStream_t s)
{
    DABGPU_LAUNCH(tii_kernel, dim3((K + 255) / 256), dim3(256), 0, s, in, acp, K, old_variant, insert, out);
    return hipGetLastError();
}

hipError_t launch_tii_add(float2 *out, size_t stride, const float2 *seg, int seg_len, const float *gain1,
                          int insert0, size_t n_frames, hipStream_t s)
{
    if (n_frames == 0 || seg_len <= 0) return hipSuccess;
    DABGPU_LAUNCH(tii_add_kernel, dim3((seg_len + 255) / 256, (unsigned)n_frames), dim3(256), 0, s, out,
                       stride, seg, seg_len, gain1, insert0);
    return hipGetLastError();
}

hipError_t launch_format(const float *in, size_t nfloats, int fmt, void *out, unsigned long long *clipped,
                         hipStream_t s)
{
    if (nfloats == 0) return hipSuccess;
    // tiles of 2048 floats; a workgroup takes up to 16 of them once there are enough workgroups to fill the chip
    const size_t n_tiles = blocks_for((nfloats + 7) / 8, 256);
    const int tiles = (int)std::min<size_t>(16, std::max<size_t>(1, n_tiles / 4096));
    const dim3 grid((unsigned)((n_tiles + tiles - 1) / tiles)), block(256);
    switch (fmt) {
        case 1: DABGPU_LAUNCH(format_kernel<1>, grid, block, 0, s, in, nfloats, out, clipped, tiles); break;
        case 2: DABGPU_LAUNCH(format_kernel<2>, grid, block, 0, s, in, nfloats, out, clipped, tiles); break;
        case 3: DABGPU_LAUNCH(format_kernel<3>, grid, block, 0, s, in, nfloats, out, clipped, tiles); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace dabgpu
