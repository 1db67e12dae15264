// api_context.hip -- the context behind the C-ABI (include/dabgpu.h): creation, the settings block the setters write and
// apply_settings turns into device tables at the next frame boundary, diagnostics.  (dabgpu_ctx.h has the map of the library's
// host side.)
#include "dabgpu_ctx.h"

using namespace dabgpu;
using namespace dabgpu_api;

namespace {
thread_local std::string g_create_error;

// src/FIRFilter.cpp:59-71 == doc/fir-filter/filtertaps.txt (configuration data)
}  // namespace
namespace dabgpu_api {
const float kDefaultTaps[45] = {
    -0.00110450468492f, 0.00120703084394f, -0.000840645749122f, -0.000187368263141f,
    0.00184351124335f, -0.00355578539893f, 0.00419321097434f, -0.00254214904271f,
    -0.00183473504148f, 0.00781436730176f, -0.0125957569107f, 0.0126200336963f,
    -0.00537294941023f, -0.00866683479398f, 0.0249746385962f, -0.0356550291181f,
    0.0319730602205f, -0.00795613788068f, -0.0363943465054f, 0.0938014090061f,
    -0.151176810265f, 0.193567320704f, 0.791776955128f, 0.193567320704f,
    -0.151176810265f, 0.0938014090061f, -0.0363943465054f, -0.00795613788068f,
    0.0319730602205f, -0.0356550291181f, 0.0249746385962f, -0.00866683479398f,
    -0.00537294941023f, 0.0126200336963f, -0.0125957569107f, 0.00781436730176f,
    -0.00183473504148f, -0.00254214904271f, 0.00419321097434f, -0.00355578539893f,
    0.00184351124335f, -0.000187368263141f, -0.000840645749122f, 0.00120703084394f,
    -0.00110450468492f};
}  // namespace dabgpu_api
namespace {

// ETSI EN 300 401 table 43 (h_{i,j}) and tables 44-47 ((i, n) per 32-carrier
// block, positive carriers first) -- the data of src/PhaseReference.cpp:35-124.
const char *const kH[4] = {"0200001120002211", "0323013021232330", "0002021322022013",
                           "0121033223212132"};
const char *const kPrBlocks[4] = {
    "033121110232211002322313003221130333231003302111"
    "011220310312223302112233011223330212223101132132",
    "201202312013021322320112",
    "322212021320",
    "003120120031221202312310001121320212203303112332",
};

}  // namespace

// ---- launch trace (dabgpu_internal.h) -------------------------------------------------------------------------------
namespace dabgpu {
namespace {
thread_local std::string *g_trace_sink = nullptr;
}
bool trace_on() { return g_trace_sink != nullptr; }
std::string *&trace_sink() { return g_trace_sink; }
void trace_launch(const char *what)
{
    if (!g_trace_sink) return;
    // (the macro stringifies "(kernel<...>)": drop the outer parentheses)
    std::string w(what);
    if (w.size() > 2 && w.front() == '(' && w.back() == ')') w = w.substr(1, w.size() - 2);
    if (!g_trace_sink->empty()) *g_trace_sink += "; ";
    *g_trace_sink += w;
}
}  // namespace dabgpu


namespace dabgpu { std::string *&trace_sink(); }
namespace dabgpu_api {
std::string *&trace_sink_ref() { return dabgpu::trace_sink(); }

int fail(dabgpu_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

int hip_fail(dabgpu_ctx *c, hipError_t e, const char *what)
{
    return fail(c, e == hipErrorOutOfMemory ? DABGPU_E_NOMEM : DABGPU_E_DEVICE,
                std::string(what) + ": " + hipGetErrorString(e));
}

bool mode_geometry(int mode, Geometry *g)
{
    // src/DabModulator.cpp:84-122
    static const Geometry tab[4] = {
        {1, 76, 1536, 2048, 11, 2656, 2552},
        {2, 76, 384, 512, 9, 664, 638},
        {3, 153, 192, 256, 8, 345, 319},
        {4, 76, 768, 1024, 10, 1328, 1276},
    };
    if (mode == 0) mode = 4;
    if (mode < 1 || mode > 4) return false;
    *g = tab[mode - 1];
    return true;
}

size_t tf_in_bytes(const Geometry &g) { return (size_t)(g.nb_symbols - 1) * (size_t)(g.K / 4); }
size_t tf_samples(const Geometry &g)
{
    return (size_t)g.null_size + (size_t)g.nb_symbols * (size_t)g.sym_size;
}

// Inverse of the FIR filter on the occupied carriers, for the equalised-boundary variant of the frame kernel
// (tf_kernel<..., EQ>, tf_kernel.h): real g[0 .. L) with
//     x[n] = sum_j g[j] z[n - (j - c)]      (z = x filtered cyclically, z[n] = sum_j taps[j] x[n + j]),
// i.e. G[k] H[k] = 1 on the K occupied bins, G[k] = sum_j g[j] exp(-2 pi i k (j - c) / N), and little gain in the empty
// band (stop-band rows weighted sqrt(lambda); the transition bins are free, which is what lets a short g fit to 1e-8).
// Least squares over the bins k = 1 .. K/2 (g and the taps are real: the negative half follows by symmetry), solved by
// Householder QR in float64 -- the normal equations of this problem are numerically singular.  Returns false when the
// taps have no well-conditioned inverse (e.g. a zero of H inside the occupied band): the chain then keeps the packed
// dual transform.  tools/design/inverse_filter_study.py is the numpy study behind the constants.
bool design_inverse_filter(const std::vector<float> &taps, int N, int K, std::vector<float> &g_out, double *fit_out)
{
    const int L = kEqTaps, cen = kEqCentre;
    const int edge = K / 2 + (N - K) / 4;
    const double lambda = 1e-6;
    const int nocc = K / 2, nstop = N / 2 - edge + 1, m = 2 * (nocc + nstop), n = L;
    std::vector<double> A((size_t)m * n), b((size_t)m, 0.0);           // column-major
    std::vector<double> hre(nocc + 1), him(nocc + 1);
    for (int k = 1; k <= nocc; ++k) {
        double re = 0.0, im = 0.0;
        for (size_t j = 0; j < taps.size(); ++j) {
            const double a = 2.0 * M_PI * (double)((j * (size_t)k) % (size_t)N) / (double)N;
            re += (double)taps[j] * std::cos(a);
            im += (double)taps[j] * std::sin(a);
        }
        hre[k] = re; him[k] = im;
        const double d = re * re + im * im;
        if (!(d > 1e-12)) return false;
        b[2 * (k - 1)] = re / d;                                         // 1 / H
        b[2 * (k - 1) + 1] = -im / d;
    }
    const double ws = std::sqrt(lambda);
    for (int j = 0; j < n; ++j) {
        double *col = &A[(size_t)j * m];
        for (int r = 0; r < nocc + nstop; ++r) {
            const int k = r < nocc ? r + 1 : edge + (r - nocc);
            const double w = r < nocc ? 1.0 : ws;
            const long q = ((long)k * (long)(j - cen)) % N;
            const double a = -2.0 * M_PI * (double)q / (double)N;
            col[2 * r] = w * std::cos(a);
            col[2 * r + 1] = w * std::sin(a);
        }
    }
    std::vector<double> v(m);
    for (int k = 0; k < n; ++k) {
        double *ck = &A[(size_t)k * m];
        double nrm = 0.0;
        for (int i = k; i < m; ++i) nrm += ck[i] * ck[i];
        nrm = std::sqrt(nrm);
        if (nrm == 0.0) return false;
        const double alpha = ck[k] > 0.0 ? -nrm : nrm;
        for (int i = k; i < m; ++i) v[i] = ck[i];
        v[k] -= alpha;
        double vv = 0.0;
        for (int i = k; i < m; ++i) vv += v[i] * v[i];
        if (vv == 0.0) return false;
        auto reflect = [&](double *x) {
            double dot = 0.0;
            for (int i = k; i < m; ++i) dot += v[i] * x[i];
            const double f = 2.0 * dot / vv;
            for (int i = k; i < m; ++i) x[i] -= f * v[i];
        };
        for (int j = k + 1; j < n; ++j) reflect(&A[(size_t)j * m]);
        reflect(b.data());
        ck[k] = alpha;
    }
    std::vector<double> g(n);
    for (int k = n - 1; k >= 0; --k) {
        double acc = b[k];
        for (int j = k + 1; j < n; ++j) acc -= A[(size_t)j * m + k] * g[j];
        const double d = A[(size_t)k * m + k];
        if (std::fabs(d) < 1e-300) return false;
        g[k] = acc / d;
    }
    // what the kernel will use: the fp32 taps; fit over the occupied bins and the noise gain
    g_out.assign((size_t)L + 1, 0.0f);
    double norm2 = 0.0;
    for (int j = 0; j < L; ++j) { g_out[j] = (float)g[j]; norm2 += (double)g_out[j] * (double)g_out[j]; }
    double fit = 0.0;
    for (int k = 1; k <= nocc; ++k) {
        double re = 0.0, im = 0.0;
        for (int j = 0; j < L; ++j) {
            const long q = ((long)k * (long)(j - cen)) % N;
            const double a = -2.0 * M_PI * (double)q / (double)N;
            re += (double)g_out[j] * std::cos(a);
            im += (double)g_out[j] * std::sin(a);
        }
        const double pr = re * hre[k] - im * him[k] - 1.0, pi = re * him[k] + im * hre[k];
        fit = std::max(fit, std::sqrt(pr * pr + pi * pi));
    }
    if (fit_out) *fit_out = fit;
    return fit < 1e-7 && norm2 < 4.0;
}

// (the design takes ~0.1 s: one per distinct set of taps and process)
bool cached_inverse_filter(const std::vector<float> &taps, int N, int K, std::vector<float> &g, double *fit)
{
    struct Entry { std::vector<float> taps, g; int N, K; bool ok; double fit; };
    static std::mutex mu;
    static std::vector<Entry> cache;
    std::lock_guard<std::mutex> lk(mu);
    for (const Entry &e : cache)
        if (e.N == N && e.K == K && e.taps == taps) { g = e.g; if (fit) *fit = e.fit; return e.ok; }
    Entry e{taps, {}, N, K, false, 0.0};
    e.ok = design_inverse_filter(taps, N, K, e.g, &e.fit);
    if (cache.size() >= 16) cache.erase(cache.begin());
    cache.push_back(e);
    g = e.g;
    if (fit) *fit = e.fit;
    return e.ok;
}

int build_tables(dabgpu_ctx *c)
{
    const Geometry &g = c->g;
    const int N = g.N, K = g.K;
    std::vector<float2> tw(N);
    for (int m = 0; m < N; ++m) {
        const double a = 2.0 * M_PI * (double)m / (double)N;
        tw[m] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    // frequency interleaver permutation, src/FrequencyInterleaver.cpp:73-92
    std::vector<uint16_t> dst(K), src(K);
    {
        const unsigned lo = (unsigned)(N - K) / 2, hi = (unsigned)N - lo, beta = (unsigned)N / 4 - 1;
        unsigned p = 0, n = 0;
        for (unsigned j = 1; j < (unsigned)N; ++j) {
            p = (13u * p + beta) & (unsigned)(N - 1);
            if (p >= lo && p <= hi && p != (unsigned)N / 2) {
                if (n >= (unsigned)K) return fail(c, DABGPU_E_INVALID, "interleaver table overflow");
                dst[n++] = (uint16_t)(p > (unsigned)N / 2 ? p - ((unsigned)N / 2 + 1)
                                                          : p + (unsigned)(K - N / 2));
            }
        }
        if (n != (unsigned)K) return fail(c, DABGPU_E_INVALID, "interleaver table short");
        for (int i = 0; i < K; ++i) src[dst[i]] = (uint16_t)i;
    }
    // phase reference quarter-turn index, src/PhaseReference.cpp:152-171
    std::vector<uint8_t> phq(K);
    {
        const char *blk = kPrBlocks[g.mode - 1];
        for (int o = 0; o < K / 32; ++o) {
            const int i = blk[2 * o] - '0', n = blk[2 * o + 1] - '0';
            for (int k = 0; k < 32; ++k) phq[32 * o + k] = (uint8_t)(((kH[i][k & 15] - '0') + n) & 3);
        }
    }
    // |y_s| of the fp32 differential recurrence y_{s+1} = y_s * x_s with
    // x = (+-c +-jc), c = (float)sqrt(1/2) (src/DifferentialModulator.cpp:65-76):
    // an axis state (m, 0) goes to (fl(m c), fl(m c)), a diagonal state (a, a) to
    // (fl(a c) + fl(a c), 0) -- independent of the data, so it is a table.
    std::vector<float> mag(g.nb_symbols);
    {
        const volatile float c45 = (float)0.70710678118654752440;
        volatile float m = 1.0f;
        mag[0] = 1.0f;
        for (int s = 1; s < g.nb_symbols; ++s) {
            volatile float p = m * c45;
            m = (s & 1) ? p : (float)(p + p);
            mag[s] = m;
        }
    }
    hipStream_t s = c->stream;
    HIPCHK(c, upload(c->d_twiddle, tw, s));
    HIPCHK(c, upload(c->d_src, src, s));
    HIPCHK(c, upload(c->d_dst, dst, s));
    HIPCHK(c, upload(c->d_phq, phq, s));
    HIPCHK(c, upload(c->d_mag, mag, s));
    return DABGPU_OK;
}

// Take the settings snapshot and (re)upload the tables of the parameter GROUPS that changed: filter taps
// (+ their frequency response), guard window, predistorter coefficients, resampler.  Gain, CFR and TII
// parameters are kernel arguments / cached-segment keys and need no upload at all.  A table is only
// rewritten after the device has drained: the previous call may still be running on a caller's stream.
int apply_settings_groups(dabgpu_ctx *c)
{
    Settings prev;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        if (c->set.epoch == c->applied_epoch) return DABGPU_OK;
        prev = c->cur;
        c->cur = c->set;
        c->set.resampler_reset = false;
        c->applied_epoch = c->set.epoch;
    }
    const bool first = !c->tables_valid;
    const bool taps_changed = first || prev.fir_key() != c->cur.fir_key();
    const bool window_changed = c->cur.overlap && (first || prev.window_key() != c->cur.window_key());
    const bool coef_changed = first || !prev.coef_equal(c->cur);
    const bool rs_changed = first || prev.resampler_key() != c->cur.resampler_key() || c->cur.resampler_reset;
    if (prev.tii_segment_key() != c->cur.tii_segment_key()) c->tii_seg_epoch = 0;   // the cached segment went through the old filter / CFR / window
    if (!(taps_changed || window_changed || coef_changed || rs_changed)) return DABGPU_OK;
    if (!first) HIPCHK(c, hipDeviceSynchronize());
    hipStream_t s = c->stream;
    if (taps_changed) {
        std::vector<float> taps(kMaxTaps, 0.0f);   // the fused kernel's copy (longer filters take the unfused kernels)
        std::copy(c->cur.taps.begin(), c->cur.taps.begin() + std::min<size_t>(c->cur.taps.size(), kMaxTaps), taps.begin());
        HIPCHK(c, upload(c->d_taps, taps, s));
        // frequency response seen by the look-ahead FIR on a cyclically extended symbol:
        // H[k] = sum_j taps[j] exp(+2 pi i j k / N), evaluated in float64
        const int N = c->g.N;
        std::vector<float2> h(N);
        for (int k = 0; k < N; ++k) {
            double re = 0.0, im = 0.0;
            for (size_t j = 0; j < c->cur.taps.size(); ++j) {
                const double a = 2.0 * M_PI * (double)((j * (size_t)k) % (size_t)N) / (double)N;
                re += (double)c->cur.taps[j] * std::cos(a);
                im += (double)c->cur.taps[j] * std::sin(a);
            }
            h[k] = make_float2((float)re, (float)im);
        }
        HIPCHK(c, upload(c->d_firh, h, s));
        // the equalised-boundary variant of the frame kernel (Mode I, up to 45 taps: a shorter filter is the same filter with
        // zero taps behind it -- fused_ntaps): the taps' inverse on the occupied bins
        c->eq_ok = false;
        if (!c->cur.taps.empty() && c->cur.taps.size() <= 45) {
            std::vector<float> g;
            c->eq_ok = cached_inverse_filter(c->cur.taps, N, c->g.K, g, &c->eq_fit);
            if (c->eq_ok) HIPCHK(c, upload(c->d_eqg, g, s));
        }
    }
    if (window_changed) {
        // src/GuardIntervalInserter.cpp:106-111
        const size_t W = c->cur.overlap;
        std::vector<float> w(2 * W);
        for (size_t i = 0; i < 2 * W; ++i)
            w[i] = (float)(0.5 * (1.0 - std::cos(M_PI * (double)i / (double)(2 * W - 1))));
        HIPCHK(c, upload(c->d_window, w, s));
    }
    if (coef_changed) {
        std::vector<float> coef(48, 0.f);
        std::copy(c->cur.am, c->cur.am + 5, coef.begin());
        std::copy(c->cur.pm, c->cur.pm + 5, coef.begin() + 8);
        std::copy(c->cur.lut, c->cur.lut + 32, coef.begin() + 16);
        HIPCHK(c, upload(c->d_coef, coef, s));
    }
    c->tables_valid = true;

    // resampler geometry, src/Resampler.cpp:65-112
    if (rs_changed) {
        size_t a = c->cur.rs_in, b = c->cur.rs_out;
        while (b) { size_t t = a % b; a = b; b = t; }
        const size_t L = c->cur.rs_out / a, M = c->cur.rs_in / a;
        size_t f = (size_t)c->g.N * 2 / M;
        if (f & 1) ++f;
        const size_t nin = f * M, nout = f * L;
        c->rs_L = L; c->rs_M = M;
        const bool changed = (int)nin != c->rs_nin || (int)nout != c->rs_nout;
        c->rs_nin = (int)nin; c->rs_nout = (int)nout;
        const size_t big = std::max(nin, nout);
        c->rs_factor = 1.0f / (float)big * (float)c->cur.rs_out / (float)c->cur.rs_in;
        if (c->cur.rs_in != c->cur.rs_out && (changed || c->cur.resampler_reset)) {
            std::vector<float> w(nin);
            for (size_t i = 0; i < nin; ++i)
                w[i] = (float)(0.5 * (1.0 - std::cos(2.0 * M_PI * (double)i / (double)(nin - 1))));
            HIPCHK(c, upload(c->d_rs_window, w, s));
            std::vector<float2> ti(nin), to(nout);
            for (size_t m = 0; m < nin; ++m) {
                const double x = 2.0 * M_PI * (double)m / (double)nin;
                ti[m] = make_float2((float)std::cos(x), (float)std::sin(x));
            }
            for (size_t m = 0; m < nout; ++m) {
                const double x = 2.0 * M_PI * (double)m / (double)nout;
                to[m] = make_float2((float)std::cos(x), (float)std::sin(x));
            }
            HIPCHK(c, upload(c->d_rs_tw_in, ti, s));
            HIPCHK(c, upload(c->d_rs_tw_out, to, s));
            // the general (rational) kernel: S = nin / M point transforms, and the L-th roots of unity
            const size_t S = nin / M;
            std::vector<float2> tsv(std::max<size_t>(S, 1)), tlv(L);
            for (size_t m = 0; m < tsv.size(); ++m) {
                const double x = 2.0 * M_PI * (double)m / (double)tsv.size();
                tsv[m] = make_float2((float)std::cos(x), (float)std::sin(x));
            }
            for (size_t m = 0; m < L; ++m) {
                const double x = 2.0 * M_PI * (double)m / (double)L;
                tlv[m] = make_float2((float)std::cos(x), (float)std::sin(x));
            }
            HIPCHK(c, upload(c->d_rs_tw_s, tsv, s));
            HIPCHK(c, upload(c->d_rs_tw_l, tlv, s));
            HIPCHK(c, c->d_rs_halo.reserve(2 * nin * sizeof(float2)));          // two buffers: read this call's, write the next's
            HIPCHK(c, hipMemsetAsync(c->d_rs_halo.p, 0, 2 * nin * sizeof(float2), s));
            c->rs_halo_cur = 0;
        }
    }
    // everything above went through the context's own stream; the caller may launch on another one
    HIPCHK(c, hipStreamSynchronize(s));
    return DABGPU_OK;
}

// A failed upload leaves some group half-written: forget what was applied, so that the next call redoes all of them.
int apply_settings(dabgpu_ctx *c)
{
    const int rc = apply_settings_groups(c);
    if (rc != DABGPU_OK) {
        std::lock_guard<std::mutex> lk(c->mu);
        c->applied_epoch = 0;
        c->tables_valid = false;
        c->tii_seg_epoch = 0;
        c->rs_nin = c->rs_nout = 0;           // (the resampler's tables and halo are rebuilt as well)
    }
    return rc;
}

Tables tables_of(dabgpu_ctx *c)
{
    Tables t;
    t.twiddle = (const float2 *)c->d_twiddle.p;
    t.src_carrier = (const uint16_t *)c->d_src.p;
    t.dst_pos = (const uint16_t *)c->d_dst.p;
    t.phase_q = (const uint8_t *)c->d_phq.p;
    t.mag = (const float *)c->d_mag.p;
    t.taps = (const float *)c->d_taps.p;
    t.window = (const float *)c->d_window.p;
    t.fir_h = (const float2 *)c->d_firh.p;
    t.eq_g = c->eq_ok ? (const float *)c->d_eqg.p : nullptr;
    return t;
}

GainParams gain_of(const dabgpu_ctx *c)
{
    GainParams gp;
    gp.mode = c->cur.gain_mode;
    gp.constant = c->cur.normalise * c->cur.digital;  // src/GainControl.cpp:118
    gp.var_variance = c->cur.var_variance;
    // (the reference forms normalise * digital in fp32, src/GainControl.cpp:118: gp.constant)
    const double c1 = c->cur.var_variance != 0.f ? 32767.0 * (double)gp.constant / (double)c->cur.var_variance : 0.0;
    gp.var_c1 = (float)c1;                                       // (var_variance 0: every symbol takes gain 1, var_sq = 0)
    gp.var_c1_lo = (float)(c1 - (double)gp.var_c1);
    gp.var_sq = c->cur.var_variance * c->cur.var_variance;
    return gp;
}

}  // namespace dabgpu_api

extern "C" {

const char *dabgpu_version(void) { return "dabgpu 0.1 (gfx950)"; }

const char *dabgpu_last_error(const dabgpu_ctx *ctx)
{
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int dabgpu_create(const dabgpu_config *cfg, dabgpu_ctx **out)
{
    if (!cfg || !out) return fail(nullptr, DABGPU_E_INVALID, "null argument");
    *out = nullptr;
    Geometry g;
    if (!mode_geometry(cfg->mode, &g))
        return fail(nullptr, DABGPU_E_INVALID, "invalid DAB transmission mode");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(nullptr, DABGPU_E_DEVICE,
                    std::string("no HIP device available (there is no CPU fallback): ") +
                        hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, DABGPU_E_INVALID, "device ordinal out of range");
    e = hipSetDevice(cfg->device);
    if (e != hipSuccess) return hip_fail(nullptr, e, "hipSetDevice");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, cfg->device);
    if (e != hipSuccess) return hip_fail(nullptr, e, "hipGetDeviceProperties");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, DABGPU_E_DEVICE,
                    std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");

    dabgpu_ctx *c = new dabgpu_ctx();
    c->g = g;
    c->device = cfg->device;
    c->max_frames = std::max(1, cfg->max_frames);
    c->chunks_cfg = cfg->chunks_per_frame;
    auto bail = [&](int rc) {
        g_create_error = c->err;
        dabgpu_destroy(c);
        return rc;
    };
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) return bail(hip_fail(c, e, "hipStreamCreate"));
#ifdef DABGPU_PHASE_TIMING
    if (c->d_phase.reserve(16 * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(c->d_phase.p, 0, 16 * sizeof(unsigned long long)) != hipSuccess)
        return bail(fail(c, DABGPU_E_DEVICE, "phase counters"));
#endif
    // the fused kernel uses up to ~40 KiB of dynamic LDS; nothing to opt in on gfx950 (<= 64 KiB)
    int rc = build_tables(c);
    if (rc) return bail(rc);
    c->set.taps.assign(kDefaultTaps, kDefaultTaps + 45);
    rc = apply_settings(c);
    if (rc) return bail(rc);
    *out = c;
    return DABGPU_OK;
}

void dabgpu_destroy(dabgpu_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto &l : c->lane)
        if (l.stream) (void)hipStreamSynchronize(l.stream);
    for (DevBuf *b : {&c->d_twiddle, &c->d_src, &c->d_dst, &c->d_phq, &c->d_mag, &c->d_taps, &c->d_firh, &c->d_eqg,
                      &c->d_window, &c->d_coef, &c->d_rs_window, &c->d_rs_tw_in, &c->d_rs_tw_out,
                      &c->d_rs_halo, &c->d_rs_tw_s, &c->d_rs_tw_l, &c->d_a, &c->d_b, &c->d_c, &c->d_in, &c->d_out, &c->d_count, &c->d_fmt, &c->d_clip, &c->d_phase,
                      &c->d_acp, &c->d_tii_car, &c->d_tii_frame, &c->d_gain1, &c->d_gains, &c->d_cic,
                      &c->d_cfr_counts, &c->d_cfr_mer, &c->d_cfr_papr, &c->d_cfr_tmp})
        b->release();
    for (auto &sl : c->slot) {
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_clip) (void)hipHostFree(sl.h_clip);
        sl.d_in.release();
        sl.d_out.release();
        if (sl.computed) (void)hipEventDestroy(sl.computed);
        if (sl.copied) (void)hipEventDestroy(sl.copied);
    }
    for (void *h : c->h_out)
        if (h) (void)hipHostFree(h);
    for (auto &l : c->lane) {
        if (l.stream) { (void)hipStreamSynchronize(l.stream); (void)hipStreamDestroy(l.stream); }
        if (l.ev) (void)hipEventDestroy(l.ev);
        for (DevBuf *b : {&l.d_a, &l.d_b, &l.d_fmt, &l.d_clip, &l.d_gain1, &l.d_gains, &l.d_cfr_counts, &l.d_cfr_mer, &l.d_cfr_papr, &l.d_cfr_tmp})
            b->release();
    }
    for (hipEvent_t e : {c->ho_prod[0], c->ho_prod[1], c->ho_cons[0], c->ho_cons[1], c->ho_start, c->ho_join, c->own_ev})
        if (e) (void)hipEventDestroy(e);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int dabgpu_get_geometry(const dabgpu_ctx *c, dabgpu_geometry *g)
{
    if (!c || !g) return DABGPU_E_INVALID;
    g->mode = c->g.mode; g->nb_symbols = c->g.nb_symbols; g->carriers = c->g.K;
    g->spacing = c->g.N; g->null_size = c->g.null_size; g->sym_size = c->g.sym_size;
    g->tf_input_bytes = tf_in_bytes(c->g);
    g->tf_samples = tf_samples(c->g);
    return DABGPU_OK;
}

// ---- setters ---------------------------------------------------------------

int dabgpu_set_gain(dabgpu_ctx *c, int gain_mode, float digital, float normalise, float var_variance)
{
    if (!c) return DABGPU_E_INVALID;
    if (gain_mode < 0 || gain_mode > 2) return fail(c, DABGPU_E_INVALID, "invalid gainmode");
    std::lock_guard<std::mutex> lk(c->mu);
    // (the adapters push their parameters on every frame: only a CHANGE makes the processing thread look)
    if (c->set.gain_mode == gain_mode && c->set.digital == digital && c->set.normalise == normalise &&
        c->set.var_variance == var_variance)
        return DABGPU_OK;
    c->set.gain_mode = gain_mode; c->set.digital = digital; c->set.normalise = normalise;
    c->set.var_variance = var_variance;
    ++c->set.epoch;
    return DABGPU_OK;
}

int dabgpu_set_gain_rounding(dabgpu_ctx *c, int rounding)
{
    if (!c) return DABGPU_E_INVALID;
    if (rounding != DABGPU_GAIN_ROUNDING_EXACT && rounding != DABGPU_GAIN_ROUNDING_REFERENCE)
        return fail(c, DABGPU_E_INVALID, "invalid gain rounding");
    std::lock_guard<std::mutex> lk(c->mu);
    const bool ref = rounding == DABGPU_GAIN_ROUNDING_REFERENCE;
    if (c->set.gain_reference_rounding == ref) return DABGPU_OK;
    c->set.gain_reference_rounding = ref;
    ++c->set.epoch;
    return DABGPU_OK;
}

int dabgpu_set_fir_taps(dabgpu_ctx *c, const float *taps, size_t n)
{
    if (!c) return DABGPU_E_INVALID;
    if (!taps || n == 0) return fail(c, DABGPU_E_INVALID, "FIRFilter: taps file has invalid format.");
    if (n > (size_t)kMaxTapsUnfused) return fail(c, DABGPU_E_INVALID, "FIRFilter: more than 512 taps not supported");
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->set.taps.size() == n && std::equal(taps, taps + n, c->set.taps.begin())) return DABGPU_OK;
    c->set.taps.assign(taps, taps + n);
    ++c->set.epoch;
    return DABGPU_OK;
}

int dabgpu_set_fir_default_taps(dabgpu_ctx *c) { return dabgpu_set_fir_taps(c, kDefaultTaps, 45); }

int dabgpu_set_window_overlap(dabgpu_ctx *c, size_t overlap)
{
    if (!c) return DABGPU_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->set.overlap == overlap) return DABGPU_OK;
    c->set.overlap = overlap;
    ++c->set.epoch;
    return DABGPU_OK;
}

int dabgpu_set_cfr(dabgpu_ctx *c, int enable, float clip, float error_clip)
{
    if (!c) return DABGPU_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->set.cfr_enable == (enable != 0) && c->set.cfr_clip == clip && c->set.cfr_errclip == error_clip)
        return DABGPU_OK;
    c->set.cfr_enable = enable != 0;
    c->set.cfr_clip = clip;
    c->set.cfr_errclip = error_clip;
    ++c->set.epoch;
    return DABGPU_OK;
}

int dabgpu_set_tii(dabgpu_ctx *c, int enable, int comb, int pattern, int old_variant)
{
    if (!c) return DABGPU_E_INVALID;
    // src/TII.cpp:119-150
    if (c->g.mode != 1 && c->g.mode != 2)
        return fail(c, DABGPU_E_INVALID, "TII::TII DAB mode " + std::to_string(c->g.mode) + " not valid!");
    if (pattern < 0 || pattern > 69) return fail(c, DABGPU_E_INVALID, "TII::TII pattern not valid!");
    if (comb < 0 || comb > 23) return fail(c, DABGPU_E_INVALID, "TII::TII comb not valid!");
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->set.tii_enable == (enable != 0) && c->set.tii_comb == comb && c->set.tii_pattern == pattern &&
        c->set.tii_old_variant == (old_variant != 0))
        return DABGPU_OK;
    c->set.tii_enable = enable != 0;
    c->set.tii_comb = comb;
    c->set.tii_pattern = pattern;
    c->set.tii_old_variant = old_variant != 0;
    ++c->set.epoch;
    return DABGPU_OK;
}

int dabgpu_set_resampler(dabgpu_ctx *c, size_t in_rate, size_t out_rate)
{
    if (!c) return DABGPU_E_INVALID;
    // an unsupported ratio fails HERE, at configuration time (the drop-in's constructor), not at the first frame
    if (const char *e = resampler_ratio_error(c->g.N, in_rate, out_rate)) return fail(c, DABGPU_E_INVALID, e);
    std::lock_guard<std::mutex> lk(c->mu);
    c->set.rs_in = in_rate; c->set.rs_out = out_rate; c->set.resampler_reset = true;
    ++c->set.epoch;
    return DABGPU_OK;
}

int dabgpu_debug_trace(dabgpu_ctx *c, int enable)
{
    if (!c) return DABGPU_E_INVALID;
    c->trace_enabled = enable != 0;
    if (!enable) c->last_variant.clear();
    return DABGPU_OK;
}

int dabgpu_debug_last_variant(dabgpu_ctx *c, char *buf, size_t cap)
{
    if (!c || !buf || cap == 0) return DABGPU_E_INVALID;
    const std::string &v = c->last_variant;
    if (v.size() + 1 > cap) return fail(c, DABGPU_E_CAPACITY, "buffer too small for the launch trace");
    std::memcpy(buf, v.c_str(), v.size() + 1);
    return DABGPU_OK;
}

int dabgpu_set_fir_boundary_mode(dabgpu_ctx *c, int mode)
{
    if (!c) return DABGPU_E_INVALID;
    if (mode != DABGPU_FIR_BOUNDARY_AUTO && mode != DABGPU_FIR_BOUNDARY_DIRECT)
        return fail(c, DABGPU_E_INVALID, "FIRFilter: unknown boundary mode");
    std::lock_guard<std::mutex> lk(c->mu);
    c->use_eq = mode == DABGPU_FIR_BOUNDARY_AUTO;
    return DABGPU_OK;
}

int dabgpu_set_output_format(dabgpu_ctx *c, int format)
{
    if (!c) return DABGPU_E_INVALID;
    if (format != 0 && !dabgpu_format_size(format)) return fail(c, DABGPU_E_INVALID, "FormatConverter: Invalid format");
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->set.out_format == format) return DABGPU_OK;
    c->set.out_format = format;
    ++c->set.epoch;
    return DABGPU_OK;
}

int dabgpu_get_num_clipped(dabgpu_ctx *c, size_t *num_clipped)
{
    CTXCHK(c);
    if (!num_clipped) return fail(c, DABGPU_E_INVALID, "null argument");
    *num_clipped = 0;
    if (c->clip_from_collect) {
        *num_clipped = c->collected_clipped;
        return DABGPU_OK;
    }
    LaneScope scratch(c, c->clip_lane);
    if (!c->d_clip.p || !c->clip_valid) return DABGPU_OK;
    HIPCHK(c, hipStreamSynchronize(c->clip_stream ? c->clip_stream : c->stream));
    unsigned long long v = 0;
    HIPCHK(c, hipMemcpy(&v, c->d_clip.p, sizeof v, hipMemcpyDeviceToHost));
    *num_clipped = (size_t)v;
    return DABGPU_OK;
}

int dabgpu_set_poly(dabgpu_ctx *c, const float am[5], const float pm[5])
{
    if (!c || !am || !pm) return DABGPU_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->set.poly_is_lut && std::equal(am, am + 5, c->set.am) && std::equal(pm, pm + 5, c->set.pm)) return DABGPU_OK;
    std::copy(am, am + 5, c->set.am);
    std::copy(pm, pm + 5, c->set.pm);
    c->set.poly_is_lut = false;
    ++c->set.epoch;
    return DABGPU_OK;
}

int dabgpu_set_lut(dabgpu_ctx *c, float scalefactor, const float lut[32])
{
    if (!c || !lut) return DABGPU_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->set.poly_is_lut && c->set.lut_scale == scalefactor && std::equal(lut, lut + 32, c->set.lut)) return DABGPU_OK;
    c->set.lut_scale = scalefactor;
    std::copy(lut, lut + 32, c->set.lut);
    c->set.poly_is_lut = true;
    ++c->set.epoch;
    return DABGPU_OK;
}

int dabgpu_fir_inverse_design(const float *taps, size_t ntaps, float *g, double *fit)
{
    if (!taps || !g || ntaps < 1 || ntaps > 45) return DABGPU_E_INVALID;
    std::vector<float> t(taps, taps + ntaps), out;
    double f = 0.0;
    const bool ok = cached_inverse_filter(t, 2048, 1536, out, &f);
    if (fit) *fit = f;
    if (out.size() >= (size_t)kEqTaps) std::copy(out.begin(), out.begin() + kEqTaps, g);
    return ok ? DABGPU_OK : DABGPU_E_INVALID;
}

size_t dabgpu_format_size(int format)
{
    return format == DABGPU_FMT_S16 ? 4 : (format == DABGPU_FMT_U8 || format == DABGPU_FMT_S8) ? 2 : 0;
}

}  // extern "C"
