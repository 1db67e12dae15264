/*
 * dab_oracle.c -- CPU oracle (test infrastructure; see dab_oracle.h).
 *
 * Every stage is a from-scratch scalar restatement of the arithmetic the
 * reference performs; the reference lines followed are cited per function.
 * Float stages keep the reference's operation ORDER and are compiled with
 * -ffp-contract=off so that they round exactly like the reference's default
 * x86-64 (SSE2, no FMA) build.
 */
#include "dab_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ modes */

/* src/DabModulator.cpp:84-122, src/BlockPartitioner.cpp:44-73 */
int dabo_mode_params(int mode, dabo_mode_t *p)
{
    static const dabo_mode_t tab[4] = {
        {1, 76, 1536, 2048, 2656, 2552, 288, 4},
        {2, 76, 384, 512, 664, 638, 288, 1},
        {3, 153, 192, 256, 345, 319, 384, 1},
        {4, 76, 768, 1024, 1328, 1276, 288, 2},
    };
    if (mode == 0) mode = 4;
    if (mode < 1 || mode > 4) return -1;
    *p = tab[mode - 1];
    return 0;
}

size_t dabo_tf_input_bytes(const dabo_mode_t *p)
{
    return (size_t)(p->nb_symbols - 1) * (size_t)(p->carriers / 4);
}

size_t dabo_tf_samples(const dabo_mode_t *p)
{
    return (size_t)p->null_size + (size_t)p->nb_symbols * (size_t)p->sym_size;
}

/* ------------------------------------------------------------------- a1 */

/* src/QpskSymbolMapper.cpp:104-156: a block of K/4 bytes carries K carriers:
 * the first K/8 bytes are the I bits, the next K/8 bytes the Q bits, both
 * MSB first; bit 0 -> +1/sqrt2, bit 1 -> -1/sqrt2 (the 16-entry LUT of the
 * reference enumerates exactly these sign combinations). */
int dabo_qpsk_map(const uint8_t *in, size_t nbytes, int carriers, float *out)
{
    const size_t blk = (size_t)carriers / 4, half = (size_t)carriers / 8;
    const float c = (float)0.70710678118654752440; /* (float)M_SQRT1_2 */
    if (blk == 0 || nbytes % blk != 0) return -1;
    for (size_t b = 0; b < nbytes / blk; ++b) {
        const uint8_t *ib = in + b * blk, *qb = ib + half;
        float *o = out + b * (size_t)carriers * 2;
        for (size_t n = 0; n < (size_t)carriers; ++n) {
            const int ibit = (ib[n >> 3] >> (7 - (n & 7))) & 1;
            const int qbit = (qb[n >> 3] >> (7 - (n & 7))) & 1;
            o[2 * n] = ibit ? -c : c;
            o[2 * n + 1] = qbit ? -c : c;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------- a2 */

/* src/FrequencyInterleaver.cpp:41-92: LCG pi(j) = 13*pi(j-1) + N/4-1 mod N,
 * keep values inside the occupied band except the DC bin; carriers above DC
 * map to 0..K/2-1, carriers below DC to K/2..K-1. */
int dabo_freq_interleave_table(int mode, uint16_t *idx)
{
    dabo_mode_t m;
    if (dabo_mode_params(mode, &m)) return -1;
    const unsigned N = (unsigned)m.spacing, K = (unsigned)m.carriers;
    const unsigned lo = (N - K) / 2, hi = N - lo, beta = N / 4 - 1;
    unsigned p = 0, n = 0;
    for (unsigned j = 1; j < N; ++j) {
        p = (13u * p + beta) & (N - 1);
        if (p >= lo && p <= hi && p != N / 2) {
            if (n >= K) return -1;
            idx[n++] = (uint16_t)(p > N / 2 ? p - (N / 2 + 1) : p + (K - N / 2));
        }
    }
    return n == K ? 0 : -1;
}

/* src/FrequencyInterleaver.cpp:103-126: out[sym][idx[n]] = in[sym][n] */
int dabo_freq_interleave(const float *in, size_t nsamples, int mode, float *out)
{
    dabo_mode_t m;
    uint16_t idx[1536];
    if (dabo_mode_params(mode, &m) || dabo_freq_interleave_table(mode, idx)) return -1;
    const size_t K = (size_t)m.carriers;
    if (nsamples % K != 0) return -1;
    for (size_t s = 0; s < nsamples / K; ++s) {
        const float *i = in + 2 * s * K;
        float *o = out + 2 * s * K;
        for (size_t n = 0; n < K; ++n) {
            o[2 * idx[n]] = i[2 * n];
            o[2 * idx[n] + 1] = i[2 * n + 1];
        }
    }
    return 0;
}

/* ------------------------------------------------------------------- a3 */

/* ETSI EN 300 401 table 43 (h_{i,j}), as held in src/PhaseReference.cpp:35-44. */
static const char *const H_ROWS[4] = {
    "0200001120002211", "0323013021232330", "0002021322022013", "0121033223212132",
};
/* ETSI EN 300 401 tables 44-47, (i, n) per block of 32 carriers, carriers above
 * DC first then carriers below DC, as held in src/PhaseReference.cpp:91-124. */
static const char *const PR_BLOCKS[4] = {
    /* mode I */
    "033121110232211002322313003221130333231003302111"
    "011220310312223302112233011223330212223101132132",
    /* mode II */
    "201202312013" "021322320112",
    /* mode III */
    "322212" "021320",
    /* mode IV */
    "003120120031221202312310" "001121320212203303112332",
};

/* src/PhaseReference.cpp:126-171: value = {1, j, -1, -j}[(h[i][k] + n) mod 4] */
int dabo_phase_reference(int mode, float *out, uint8_t *qidx)
{
    static const float RE[4] = {1.f, 0.f, -1.f, 0.f}, IM[4] = {0.f, 1.f, 0.f, -1.f};
    dabo_mode_t m;
    if (dabo_mode_params(mode, &m)) return -1;
    const char *blk = PR_BLOCKS[m.mode - 1];
    for (int o = 0; o < m.carriers / 32; ++o) {
        const int i = blk[2 * o] - '0', n = blk[2 * o + 1] - '0';
        for (int k = 0; k < 32; ++k) {
            const int q = ((H_ROWS[i][k & 15] - '0') + n) & 3;
            if (out) {
                out[2 * (32 * o + k)] = RE[q];
                out[2 * (32 * o + k) + 1] = IM[q];
            }
            if (qidx) qidx[32 * o + k] = (uint8_t)q;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------- a4 */

/* src/DifferentialModulator.cpp:65-76: y[0] = phase; y[s+1] = y[s] * x[s],
 * std::complex<float> product = (ar*br - ai*bi, ar*bi + ai*br), each product
 * and the sum rounded separately (no FMA). */
int dabo_diff_mod(const float *phase, const float *data, size_t ndata, int carriers, float *out)
{
    const size_t K = (size_t)carriers;
    if (K == 0 || ndata % K != 0) return -1;
    memcpy(out, phase, K * 2 * sizeof(float));
    for (size_t s = 0; s < ndata / K; ++s) {
        const float *y = out + 2 * s * K, *x = data + 2 * s * K;
        float *z = out + 2 * (s + 1) * K;
        for (size_t k = 0; k < K; ++k) {
            const float ar = y[2 * k], ai = y[2 * k + 1], br = x[2 * k], bi = x[2 * k + 1];
            const float rr = ar * br, ii = ai * bi, ri = ar * bi, ir = ai * br;
            z[2 * k] = rr - ii;
            z[2 * k + 1] = ri + ir;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------- a5 */

/* src/SignalMultiplexer.cpp:59-68: plain concatenation (null or TII first). */
void dabo_signal_mux(const float *first, size_t nfirst, const float *rest, size_t nrest, float *out)
{
    memcpy(out, first, nfirst * 2 * sizeof(float));
    memcpy(out + 2 * nfirst, rest, nrest * 2 * sizeof(float));
}

/* ---------------------------------------------------------- float64 DFT */

/* Unnormalised DFT, y[n] = sum_k x[k] exp(sign * 2 pi i k n / N), N a power of
 * two, evaluated in float64 (iterative radix-2, twiddles straight from
 * cos/sin).  This is the published definition of what fftwf_plan_dft_1d
 * computes (FFTW_FORWARD: sign -1, FFTW_BACKWARD: sign +1, no scaling). */
typedef struct {
    size_t n;
    double *tw; /* n/2 pairs, exp(+2 pi i k / n) */
    uint32_t *rev;
} dft_plan;

static dft_plan g_plans[16];

static const dft_plan *dft_get_plan(size_t n)
{
    for (int i = 0; i < 16; ++i)
        if (g_plans[i].n == n) return &g_plans[i];
    for (int i = 0; i < 16; ++i) {
        if (g_plans[i].n != 0) continue;
        dft_plan *p = &g_plans[i];
        p->tw = (double *)malloc(sizeof(double) * n);
        p->rev = (uint32_t *)malloc(sizeof(uint32_t) * n);
        int lg = 0;
        while (((size_t)1 << lg) < n) ++lg;
        for (size_t k = 0; k < n / 2; ++k) {
            p->tw[2 * k] = cos(2.0 * M_PI * (double)k / (double)n);
            p->tw[2 * k + 1] = sin(2.0 * M_PI * (double)k / (double)n);
        }
        for (size_t k = 0; k < n; ++k) {
            uint32_t r = 0;
            for (int b = 0; b < lg; ++b)
                if (k & ((size_t)1 << b)) r |= 1u << (lg - 1 - b);
            p->rev[k] = r;
        }
        p->n = n;
        return p;
    }
    return NULL;
}

static void dft_pow2_f64(const double *in, double *out, size_t n, int sign);

/* Any other length: Bluestein's chirp-z identity, nk = (n^2 + k^2 - (k-n)^2) / 2, turns the DFT into a
 * circular convolution evaluated with the power-of-two transform above -- still float64 throughout, the
 * chirp angles reduced mod 2N as integers (the resampler's output FFT has sizes such as 4800 or 6144). */
static void dft_bluestein_f64(const double *in, double *out, size_t n, int sign)
{
    size_t m = 1;
    while (m < 2 * n - 1) m <<= 1;
    double *c = (double *)malloc(sizeof(double) * 2 * n);      /* chirp exp(sign * pi i j^2 / n) */
    double *a = (double *)calloc(2 * m, sizeof(double)), *b = (double *)calloc(2 * m, sizeof(double));
    double *fa = (double *)malloc(sizeof(double) * 2 * m), *fb = (double *)malloc(sizeof(double) * 2 * m);
    for (size_t j = 0; j < n; ++j) {
        const size_t q = (size_t)(((unsigned long long)j * j) % (2 * n));
        const double ang = (double)sign * M_PI * (double)q / (double)n;
        c[2 * j] = cos(ang);
        c[2 * j + 1] = sin(ang);
        a[2 * j] = in[2 * j] * c[2 * j] - in[2 * j + 1] * c[2 * j + 1];
        a[2 * j + 1] = in[2 * j] * c[2 * j + 1] + in[2 * j + 1] * c[2 * j];
        b[2 * j] = c[2 * j];                       /* conj(chirp), also at the mirrored index */
        b[2 * j + 1] = -c[2 * j + 1];
        if (j) { b[2 * (m - j)] = c[2 * j]; b[2 * (m - j) + 1] = -c[2 * j + 1]; }
    }
    dft_pow2_f64(a, fa, m, -1);
    dft_pow2_f64(b, fb, m, -1);
    for (size_t k = 0; k < m; ++k) {
        const double re = fa[2 * k] * fb[2 * k] - fa[2 * k + 1] * fb[2 * k + 1];
        const double im = fa[2 * k] * fb[2 * k + 1] + fa[2 * k + 1] * fb[2 * k];
        a[2 * k] = re;
        a[2 * k + 1] = im;
    }
    dft_pow2_f64(a, fa, m, +1);
    for (size_t k = 0; k < n; ++k) {
        const double re = fa[2 * k] / (double)m, im = fa[2 * k + 1] / (double)m;
        out[2 * k] = re * c[2 * k] - im * c[2 * k + 1];
        out[2 * k + 1] = re * c[2 * k + 1] + im * c[2 * k];
    }
    free(c); free(a); free(b); free(fa); free(fb);
}

void dabo_dft_f64(const double *in, double *out, size_t n, int sign)
{
    if (n & (n - 1)) dft_bluestein_f64(in, out, n, sign);
    else dft_pow2_f64(in, out, n, sign);
}

static void dft_pow2_f64(const double *in, double *out, size_t n, int sign)
{
    const dft_plan *p = dft_get_plan(n);
    if (!p) return;
    for (size_t k = 0; k < n; ++k) {
        out[2 * p->rev[k]] = in[2 * k];
        out[2 * p->rev[k] + 1] = in[2 * k + 1];
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const size_t half = len / 2, step = n / len;
        for (size_t base = 0; base < n; base += len) {
            for (size_t j = 0; j < half; ++j) {
                const double wr = p->tw[2 * j * step];
                const double wi = sign > 0 ? p->tw[2 * j * step + 1] : -p->tw[2 * j * step + 1];
                double *a = out + 2 * (base + j), *b = out + 2 * (base + j + half);
                const double tr = b[0] * wr - b[1] * wi, ti = b[0] * wi + b[1] * wr;
                b[0] = a[0] - tr;
                b[1] = a[1] - ti;
                a[0] += tr;
                a[1] += ti;
            }
        }
    }
}

#ifdef DABO_FAST
/* CPU-BASELINE BUILD ONLY (liboracle_fast.so, bench.py's cpu_baseline leg; never the parity oracle): the
 * power-of-two transforms in fp32, Stockham autosort radix 4 (one radix-2 step when log2 n is odd) with table
 * twiddles -- the arithmetic class of the reference's FFTW3f plans, so that the timed baseline is not the
 * float64 radix-2 evaluation the oracle proper uses to define the result. */
static float *g_ftw[32];   /* per log2 n: n pairs exp(+2 pi i k / n) */
static const float *fast_twiddles(size_t n)
{
    int lg = 0;
    while (((size_t)1 << lg) < n) ++lg;
    if (!g_ftw[lg]) {
        float *t = (float *)malloc(sizeof(float) * 2 * n);
        for (size_t k = 0; k < n; ++k) {
            t[2 * k] = (float)cos(2.0 * M_PI * (double)k / (double)n);
            t[2 * k + 1] = (float)sin(2.0 * M_PI * (double)k / (double)n);
        }
        g_ftw[lg] = t;                 /* (a racing second writer computes the same table: one copy leaks) */
    }
    return g_ftw[lg];
}

/* FFTW3f engine of the baseline build, when the host has it: the reference's own transform library
 * (src/OfdmGenerator.cpp:106-117, src/Resampler.cpp:94-108: fftwf_plan_dft_1d(..., FFTW_MEASURE), fftwf_execute).  Not a
 * link-time dependency -- libfftw3f.so.3 is looked for at run time (DABO_FFTW=0 in the environment skips the search), and
 * the few prototypes used are declared here.  dabo_fft_engine() says which engine the baseline ran on. */
#include <dlfcn.h>
#include <pthread.h>
typedef float fftwf_cpx_[2];
typedef void *(*fftwf_plan_fn_)(int, fftwf_cpx_ *, fftwf_cpx_ *, int, unsigned);
typedef void (*fftwf_exec_fn_)(void *, fftwf_cpx_ *, fftwf_cpx_ *);
static struct {
    int tried;
    void *lib;
    fftwf_plan_fn_ plan;
    fftwf_exec_fn_ exec;
    void *plans[32][2];
    pthread_mutex_t mu;
} g_fftw = {0, NULL, NULL, NULL, {{NULL}}, PTHREAD_MUTEX_INITIALIZER};

static void fftw_try_load(void)
{
    pthread_mutex_lock(&g_fftw.mu);
    if (!g_fftw.tried) {
        const char *e = getenv("DABO_FFTW");
        if (!e || strcmp(e, "0") != 0) {
            void *h = dlopen("libfftw3f.so.3", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("libfftw3f.so", RTLD_NOW | RTLD_LOCAL);
            if (h) {
                g_fftw.plan = (fftwf_plan_fn_)dlsym(h, "fftwf_plan_dft_1d");
                g_fftw.exec = (fftwf_exec_fn_)dlsym(h, "fftwf_execute_dft");
                if (g_fftw.plan && g_fftw.exec) g_fftw.lib = h;
            }
        }
        g_fftw.tried = 1;
    }
    pthread_mutex_unlock(&g_fftw.mu);
}

const char *dabo_fft_engine(void)
{
    fftw_try_load();
    return g_fftw.lib ? "fftw3f" : "port";
}

/* plan for (n = 2^lg, sign), created once (planning is not thread-safe, executing a plan on new arrays is):
 * FFTW_MEASURE as the reference plans, FFTW_UNALIGNED because the plan runs on the callers' arrays */
static void *fftw_plan_for(int lg, int sign)
{
    const int si = sign > 0 ? 1 : 0;
    void *p = __atomic_load_n(&g_fftw.plans[lg][si], __ATOMIC_ACQUIRE);
    if (p) return p;
    pthread_mutex_lock(&g_fftw.mu);
    p = g_fftw.plans[lg][si];
    if (!p) {
        const size_t n = (size_t)1 << lg;
        fftwf_cpx_ *a = (fftwf_cpx_ *)calloc(n, sizeof(fftwf_cpx_)), *b = (fftwf_cpx_ *)calloc(n, sizeof(fftwf_cpx_));
        if (a && b) p = g_fftw.plan((int)n, a, b, sign > 0 ? +1 : -1, /* FFTW_MEASURE */ 0u | /* FFTW_UNALIGNED */ (1u << 1));
        free(a);
        free(b);
        __atomic_store_n(&g_fftw.plans[lg][si], p, __ATOMIC_RELEASE);
    }
    pthread_mutex_unlock(&g_fftw.mu);
    return p;
}

static void dft_f32_fast(const float *in, float *out, size_t n, int sign, float *work)
{
    fftw_try_load();
    if (g_fftw.lib && in != out) {
        int lgn = 0;
        while (((size_t)1 << lgn) < n) ++lgn;
        void *p = fftw_plan_for(lgn, sign);
        if (p) {
            g_fftw.exec(p, (fftwf_cpx_ *)(uintptr_t)in, (fftwf_cpx_ *)out);     /* (out of place: the input is left alone) */
            return;
        }
    }
    const float *tw = fast_twiddles(n);
    const float sg = sign > 0 ? 1.0f : -1.0f;
    int lg = 0;
    while (((size_t)1 << lg) < n) ++lg;
    /* ping-pong so that the last stage lands in `out` */
    const int nst = lg / 2 + (lg & 1);
    const float *src = in;
    float *dst = (nst & 1) ? out : work;
    size_t ns = 1;
    if (lg & 1) {
        for (size_t j = 0; j < n / 2; ++j) {
            const float ar = src[2 * j], ai = src[2 * j + 1], br = src[2 * (j + n / 2)], bi = src[2 * (j + n / 2) + 1];
            dst[4 * j] = ar + br; dst[4 * j + 1] = ai + bi;
            dst[4 * j + 2] = ar - br; dst[4 * j + 3] = ai - bi;
        }
        ns = 2;
        src = dst;
        dst = (dst == out) ? work : out;
    }
    for (; ns < n; ns *= 4) {
        const size_t q = n / 4, step = n / (4 * ns);
        if (ns < 16) {
            /* few butterflies per block: twiddles outermost, the long loop runs over the blocks */
            for (size_t k = 0; k < ns; ++k) {
                const float w1r = tw[2 * k * step], w1i = sg * tw[2 * k * step + 1];
                const float w2r = tw[4 * k * step], w2i = sg * tw[4 * k * step + 1];
                const float w3r = tw[6 * k * step], w3i = sg * tw[6 * k * step + 1];
                const float *x0 = src + 2 * k, *x1 = x0 + 2 * q, *x2 = x1 + 2 * q, *x3 = x2 + 2 * q;
                float *y = dst + 2 * k;
                const size_t nb = q / ns;
                for (size_t blk = 0; blk < nb; ++blk) {
                    const size_t i = 2 * blk * ns, o = 8 * blk * ns;
                    const float ar = x0[i], ai = x0[i + 1];
                    const float br = x1[i] * w1r - x1[i + 1] * w1i, bi = x1[i] * w1i + x1[i + 1] * w1r;
                    const float cr = x2[i] * w2r - x2[i + 1] * w2i, ci = x2[i] * w2i + x2[i + 1] * w2r;
                    const float dr = x3[i] * w3r - x3[i + 1] * w3i, di = x3[i] * w3i + x3[i + 1] * w3r;
                    const float s0r = ar + cr, s0i = ai + ci, s1r = ar - cr, s1i = ai - ci;
                    const float s2r = br + dr, s2i = bi + di, s3r = -sg * (bi - di), s3i = sg * (br - dr);
                    y[o] = s0r + s2r; y[o + 1] = s0i + s2i;
                    y[o + 2 * ns] = s1r + s3r; y[o + 2 * ns + 1] = s1i + s3i;
                    y[o + 4 * ns] = s0r - s2r; y[o + 4 * ns + 1] = s0i - s2i;
                    y[o + 6 * ns] = s1r - s3r; y[o + 6 * ns + 1] = s1i - s3i;
                }
            }
            src = dst;
            dst = (dst == out) ? work : out;
            continue;
        }
        for (size_t blk = 0; blk < q / ns; ++blk) {
            const float *x0 = src + 2 * (blk * ns), *x1 = x0 + 2 * q, *x2 = x1 + 2 * q, *x3 = x2 + 2 * q;
            float *y = dst + 2 * (blk * 4 * ns);
            for (size_t k = 0; k < ns; ++k) {
                const float w1r = tw[2 * k * step], w1i = sg * tw[2 * k * step + 1];
                const float w2r = tw[4 * k * step], w2i = sg * tw[4 * k * step + 1];
                const float w3r = tw[6 * k * step], w3i = sg * tw[6 * k * step + 1];
                const float ar = x0[2 * k], ai = x0[2 * k + 1];
                const float br = x1[2 * k] * w1r - x1[2 * k + 1] * w1i, bi = x1[2 * k] * w1i + x1[2 * k + 1] * w1r;
                const float cr = x2[2 * k] * w2r - x2[2 * k + 1] * w2i, ci = x2[2 * k] * w2i + x2[2 * k + 1] * w2r;
                const float dr = x3[2 * k] * w3r - x3[2 * k + 1] * w3i, di = x3[2 * k] * w3i + x3[2 * k + 1] * w3r;
                const float s0r = ar + cr, s0i = ai + ci, s1r = ar - cr, s1i = ai - ci;
                const float s2r = br + dr, s2i = bi + di, s3r = -sg * (bi - di), s3i = sg * (br - dr);   /* +-i (b - d) */
                y[2 * k] = s0r + s2r; y[2 * k + 1] = s0i + s2i;
                y[2 * (k + ns)] = s1r + s3r; y[2 * (k + ns) + 1] = s1i + s3i;
                y[2 * (k + 2 * ns)] = s0r - s2r; y[2 * (k + 2 * ns) + 1] = s0i - s2i;
                y[2 * (k + 3 * ns)] = s1r - s3r; y[2 * (k + 3 * ns) + 1] = s1i - s3i;
            }
        }
        src = dst;
        dst = (dst == out) ? work : out;
    }
    if (src != out) memcpy(out, src, sizeof(float) * 2 * n);
}
#endif

static void dft_f32_via_f64(const float *in, float *out, size_t n, int sign, double *w0, double *w1)
{
#ifdef DABO_FAST
    if ((n & (n - 1)) == 0 && n >= 4) {
        dft_f32_fast(in, out, n, sign, (float *)w0);          /* w0 holds 2 n doubles: room for 2 n floats */
        return;
    }
#endif
    for (size_t k = 0; k < 2 * n; ++k) w0[k] = (double)in[k];
    dabo_dft_f64(w0, w1, n, sign);
    for (size_t k = 0; k < 2 * n; ++k) out[k] = (float)w1[k];
}

/* ------------------------------------------------------------------- a6 */

/* src/OfdmGenerator.cpp:77-94 (bin layout) and :207-283 (per-symbol loop,
 * CFR disabled): bins 1..K/2 <- first half of the symbol's carriers, bins
 * N-K/2..N-1 <- second half, everything else (DC, guard band) zero, then the
 * unnormalised backward DFT (fftwf_execute, :228). */
int dabo_ofdm_generate(const float *in, int nsym, int carriers, int spacing, float *out)
{
    const size_t K = (size_t)carriers, N = (size_t)spacing;
    if (K > N || (N & (N - 1)) != 0) return -1;
    const size_t pos_dst = (K & 1) ? 0 : 1, pos_n = (K + 1) / 2, neg_dst = N - K / 2, neg_n = K / 2;
    double *x = (double *)calloc(2 * N, sizeof(double));
    double *y = (double *)malloc(2 * N * sizeof(double));
    if (!x || !y) { free(x); free(y); return -1; }
#ifdef DABO_FAST
    {
        /* (baseline build: fp32 transform, see dft_f32_fast) */
        float *xf = (float *)x, *wf = (float *)y;
        for (int s = 0; s < nsym; ++s) {
            const float *i = in + 2 * (size_t)s * K;
            memset(xf, 0, 2 * N * sizeof(float));
            memcpy(xf + 2 * pos_dst, i, 2 * pos_n * sizeof(float));
            memcpy(xf + 2 * neg_dst, i + 2 * pos_n, 2 * neg_n * sizeof(float));
            dft_f32_fast(xf, out + 2 * (size_t)s * N, N, +1, wf);
        }
        free(x);
        free(y);
        return 0;
    }
#endif
    for (int s = 0; s < nsym; ++s) {
        const float *i = in + 2 * (size_t)s * K;
        memset(x, 0, 2 * N * sizeof(double));
        for (size_t k = 0; k < 2 * pos_n; ++k) x[2 * pos_dst + k] = (double)i[k];
        for (size_t k = 0; k < 2 * neg_n; ++k) x[2 * neg_dst + k] = (double)i[2 * pos_n + k];
        dabo_dft_f64(x, y, N, +1);
        float *o = out + 2 * (size_t)s * N;
        for (size_t k = 0; k < 2 * N; ++k) o[k] = (float)y[k];
    }
    free(x);
    free(y);
    return 0;
}

/* a12 CicEqualizer, src/CicEqualizer.cpp:29-57 (filter) and :66-91 (process) */
void dabo_cic_filter(int carriers, size_t spacing, int R, float *filter)
{
    const int M = 1, N = 4;
    const float pi = 4.0f * atanf(1.0f);
    for (int i = 0; i < carriers; ++i) {
        const int k = i < (carriers + 1) / 2 ? i + ((carriers & 1) ^ 1) : i - carriers;
        const float angle = pi * k / spacing;
        if (k == 0) {
            filter[i] = 1.0f;
        } else {
            float f = sinf(angle / R) / sinf(angle * M);
            f = fabsf(f) * R * M;
            filter[i] = powf(f, N);
        }
    }
}

int dabo_cic_equalize(const float *in, size_t nsamples, int carriers, const float *filter, float *out)
{
    if (nsamples % (size_t)carriers) return -1;
    for (size_t i = 0; i < nsamples; ++i) {
        const float f = filter[i % (size_t)carriers];
        out[2 * i] = in[2 * i] * f;
        out[2 * i + 1] = in[2 * i + 1] * f;
    }
    return 0;
}

/* f-3: the same loop with crest-factor reduction enabled (src/OfdmGenerator.cpp:207-283 with
 * myCfr, cfr_one_iteration :310-373). */
static void papr_block(const float *x, size_t n, double *peak, double *mean)
{
    /* PAPRStats::process_block, src/PAPRStats.cpp:41-60: std::norm in float, accumulated in double */
    double pk = 0, rms2 = 0;
    for (size_t i = 0; i < n; ++i) {
        const float re = x[2 * i], im = x[2 * i + 1];
        const double nrm = (double)(re * re + im * im);
        if (nrm > pk) pk = nrm;
        rms2 += nrm;
    }
    *peak = pk;
    *mean = rms2 / (double)n;
}

int dabo_ofdm_generate_cfr(const float *in, int nsym, int carriers, int spacing, float clip,
                           float error_clip, int mer_index, float *out, dabo_cfr_stats *st, double *papr)
{
    const size_t K = (size_t)carriers, N = (size_t)spacing;
    if (K > N || (N & (N - 1)) != 0) return -1;
    const size_t pos_dst = (K & 1) ? 0 : 1, pos_n = (K + 1) / 2, neg_dst = N - K / 2, neg_n = K / 2;
    float *ref = (float *)malloc(2 * N * sizeof(float));      /* `reference`, :222-226 */
    float *sym = (float *)malloc(2 * N * sizeof(float));      /* myFftOut */
    float *before = (float *)malloc(2 * N * sizeof(float));   /* before_cfr, :234-238 */
    float *post = (float *)malloc(2 * N * sizeof(float));     /* myCfrPostFft */
    float *fin = (float *)malloc(2 * N * sizeof(float));      /* myFftIn, second pass */
    double *w0 = (double *)malloc(2 * N * sizeof(double)), *w1 = (double *)malloc(2 * N * sizeof(double));
    if (!ref || !sym || !before || !post || !fin || !w0 || !w1) return -1;
    dabo_cfr_stats s0 = {0, 0, 0.0, 0.0, NAN};
    const float clip_squared = clip * clip;                    /* :315 */
    const float err_clip_squared = error_clip * error_clip;    /* :339 */
    for (int s = 0; s < nsym; ++s) {
        const float *i = in + 2 * (size_t)s * K;
        memset(ref, 0, 2 * N * sizeof(float));
        memcpy(ref + 2 * pos_dst, i, 2 * pos_n * sizeof(float));
        memcpy(ref + 2 * neg_dst, i + 2 * pos_n, 2 * neg_n * sizeof(float));
        dft_f32_via_f64(ref, sym, N, +1, w0, w1);              /* :228 */
        double *pp = papr ? papr + 4 * (size_t)s : NULL;
        if (pp) { papr_block(sym, N, &pp[0], &pp[1]); pp[2] = pp[3] = 0.0; }   /* :232 */
        if (mer_index == s) memcpy(before, sym, 2 * N * sizeof(float));
        /* clip, :320-330 */
        for (size_t n = 0; n < N; ++n) {
            const float mag_squared = sym[2 * n] * sym[2 * n] + sym[2 * n + 1] * sym[2 * n + 1];
            if (mag_squared > clip_squared) {
                const float f = sqrtf(clip_squared / mag_squared);
                sym[2 * n] *= f;
                sym[2 * n + 1] *= f;
                ++s0.num_clip;
            }
        }
        dft_f32_via_f64(sym, post, N, -1, w0, w1);             /* :333-334 */
        /* error in the frequency domain, clipped, :341-366 */
        for (size_t k = 0; k < N; ++k) {
            const float cr = post[2 * k] / (float)N, ci = post[2 * k + 1] / (float)N;
            float er = ref[2 * k] - cr, ei = ref[2 * k + 1] - ci;
            const float mag_squared = er * er + ei * ei;
            if (mag_squared > err_clip_squared) {
                const float f = sqrtf(err_clip_squared / mag_squared);
                er *= f;
                ei *= f;
                ++s0.num_error_clip;
            }
            fin[2 * k] = cr + er;
            fin[2 * k + 1] = ci + ei;
        }
        dft_f32_via_f64(fin, sym, N, +1, w0, w1);              /* :369-370 */
        if (s > 0 && pp) papr_block(sym, N, &pp[2], &pp[3]);   /* :246-248 */
        if (s > 0 && mer_index == s) {                          /* :250-273 */
            double sum_iq = 0, sum_delta = 0;
            for (size_t n = 0; n < N; ++n) {
                const float br = before[2 * n], bi = before[2 * n + 1];
                const float dr = sym[2 * n] - br, di = sym[2 * n + 1] - bi;
                sum_iq += (double)(br * br + bi * bi);
                sum_delta += (double)(dr * dr + di * di);
            }
            s0.mer_sum_iq = sum_iq;
            s0.mer_sum_delta = sum_delta;
            s0.mer_db = sum_delta > 0 ? 10.0 * log10(sum_iq / sum_delta) : 90.0;
        }
        memcpy(out + 2 * (size_t)s * N, sym, 2 * N * sizeof(float));
    }
    if (st) *st = s0;
    free(ref); free(sym); free(before); free(post); free(fin); free(w0); free(w1);
    return 0;
}

/* PAPRStats::calculate_papr, src/PAPRStats.cpp:74-103 */
double dabo_papr_db(const double *pm, size_t nblocks)
{
    double peak = 0, rms2 = 0;
    for (size_t i = 0; i < nblocks; ++i) {
        if (pm[2 * i] > peak) peak = pm[2 * i];
        rms2 += pm[2 * i + 1];
    }
    rms2 /= (double)nblocks;
    return 10.0 * log10(peak / rms2);
}

/* ------------------------------------------------------------------- a7 */

/* The reference's x86 build runs the SSE code, which views a symbol of N
 * complex samples as N/2 vectors of four floats {re0, im0, re1, im1} and keeps
 * four independent running statistics; the helpers below replay that lane by
 * lane (src/GainControl.cpp:201-340). */

/* src/GainControl.cpp:201-249 */
static float gain_max_sse(const float *f, size_t nvec)
{
    float mn[4], mx[4];
    for (int l = 0; l < 4; ++l) { mn[l] = FLT_MAX; mx[l] = FLT_MIN; }
    for (size_t v = 0; v < nvec; ++v)
        for (int l = 0; l < 4; ++l) {
            const float x = f[4 * v + l];
            /* _mm_min_ps(a,b) = a < b ? a : b ; _mm_max_ps(a,b) = a > b ? a : b */
            mn[l] = x < mn[l] ? x : mn[l];
            mx[l] = x > mx[l] ? x : mx[l];
        }
    float lo = mn[0], hi = mx[0];
    for (int l = 1; l < 4; ++l) { if (mn[l] < lo) lo = mn[l]; if (mx[l] > hi) hi = mx[l]; }
    const float nlo = lo * -1.0f;
    const float m = nlo > hi ? nlo : hi;
    return ((int)m != 0) ? 32767.0f / m : 1.0f;
}

/* src/GainControl.cpp:251-340 */
static float gain_var_sse(const float *f, size_t nvec, float var_variance)
{
    float mean[4] = {0, 0, 0, 0}, var[4] = {0, 0, 0, 0};
    for (size_t v = 0; v < nvec; ++v) {
        const float cnt = (float)(v + 1);
        for (int l = 0; l < 4; ++l) {
            const float d = f[4 * v + l] - mean[l];
            const float q = d / cnt;
            mean[l] = mean[l] + q;
        }
    }
    /* lanes {0,2} hold re, {1,3} hold im; merged as (a + b) * 0.5 */
    float m2[4];
    for (int l = 0; l < 4; ++l) m2[l] = (mean[l] + mean[l ^ 2]) * 0.5f;
    for (size_t v = 0; v < nvec; ++v) {
        const float cnt = (float)(v + 1);
        for (int l = 0; l < 4; ++l) {
            const float diff = f[4 * v + l] - m2[l];
            const float sq = diff * diff;
            const float d = sq - var[l];
            const float q = d / cnt;
            var[l] = var[l] + q;
        }
    }
    float sd[2];
    for (int l = 0; l < 2; ++l) {
        const float merged = (var[l] + var[l + 2]) * 0.5f;
        sd[l] = sqrtf(merged) * var_variance;
    }
    if ((int)sd[0] == 0) return 1.0f;
    return 32767.0f / (sd[0] > sd[1] ? sd[0] : sd[1]);
}

/* src/GainControl.cpp:118-155: symbol 0 (NULL/TII) takes the statistics of
 * symbol 1; gain = mode_gain * (normalise * digital), applied to every float. */
int dabo_gain_control(const float *in, size_t nsamples, int framesize, int gain_mode,
                      float dig_gain, float normalise, float var_variance,
                      float *out, float *gains)
{
    const size_t F = (size_t)framesize;
    if (F == 0 || nsamples % F != 0 || (F & 1)) return -1;
    const float constant_gain = normalise * dig_gain;
    const size_t nsym = nsamples / F;
    for (size_t s = 0; s < nsym; ++s) {
        const size_t src = (s == 0 && nsym > 1) ? 1 : s;
        const float *stat = in + 2 * src * F;
        float g;
        switch (gain_mode) {
            case 0: g = 512.0f; break;
            case 1: g = gain_max_sse(stat, F / 2); break;
            case 2: g = gain_var_sse(stat, F / 2, var_variance); break;
            default: return -1;
        }
        g = g * constant_gain;
        if (gains) gains[s] = g;
        const float *i = in + 2 * s * F;
        float *o = out + 2 * s * F;
        for (size_t k = 0; k < 2 * F; ++k) o[k] = i[k] * g;
    }
    return 0;
}

/* ------------------------------------------------------------------- a8 */

/* src/GuardIntervalInserter.cpp:301-319 (overlap 0: cyclic-prefix copies) and
 * :149-300 (raised-cosine overlap; window from :106-111). */
int dabo_guard_interval(const float *in, int nb_symbols, int spacing, int null_size,
                        int sym_size, int overlap, float *out)
{
    const size_t N = (size_t)spacing, W = (size_t)overlap;
    if (null_size < spacing || sym_size < spacing) return -1;
    const size_t cp0 = (size_t)null_size - N, cp = (size_t)sym_size - N;
    if (W == 0) {
        float *o = out;
        memcpy(o, in + 2 * (N - cp0), cp0 * 2 * sizeof(float));
        memcpy(o + 2 * cp0, in, N * 2 * sizeof(float));
        o += 2 * (size_t)null_size;
        for (int s = 0; s < nb_symbols; ++s) {
            const float *x = in + 2 * N * (size_t)(s + 1);
            memcpy(o, x + 2 * (N - cp), cp * 2 * sizeof(float));
            memcpy(o + 2 * cp, x, N * 2 * sizeof(float));
            o += 2 * (size_t)sym_size;
        }
        return 0;
    }
    if (cp + W > N || W > cp || W > cp0) return -1;
    float *w = (float *)malloc(2 * W * sizeof(float));
    if (!w) return -1;
    for (size_t i = 0; i < 2 * W; ++i)
        w[i] = (float)(0.5 * (1.0 - cos(M_PI * (double)i / (double)(2 * W - 1))));

    /* NULL symbol: prefix, body, falling half-window 1 -> 1/2, suffix 1/2 -> 0 */
    memcpy(out, in + 2 * (N - cp0), cp0 * 2 * sizeof(float));
    memcpy(out + 2 * cp0, in, (N - W) * 2 * sizeof(float));
    for (size_t i = 0; i < W; ++i) {
        const float f = w[2 * W - 1 - i];
        out[2 * (cp0 + N - W + i)] = in[2 * (N - W + i)] * f;
        out[2 * (cp0 + N - W + i) + 1] = in[2 * (N - W + i) + 1] * f;
    }
    for (size_t i = 0; i < W; ++i) {
        const float f = w[W - 1 - i];
        out[2 * (cp0 + N + i)] = in[2 * i] * f;
        out[2 * (cp0 + N + i) + 1] = in[2 * i + 1] * f;
    }
    for (int s = 0; s < nb_symbols; ++s) {
        const float *x = in + 2 * N * (size_t)(s + 1);
        float *o = out + 2 * ((size_t)null_size + (size_t)s * (size_t)sym_size);
        /* rising edge accumulates onto the previous symbol's tail: o[-W .. W) */
        for (size_t i = 0; i < 2 * W; ++i) {
            const float *xi = x + 2 * (N - cp - W + i);
            float *oo = o + 2 * ((ptrdiff_t)i - (ptrdiff_t)W);
            oo[0] += xi[0] * w[i];
            oo[1] += xi[1] * w[i];
        }
        memcpy(o + 2 * W, x + 2 * (N - cp + W), (cp - W) * 2 * sizeof(float));
        if (s + 1 >= nb_symbols) {
            memcpy(o + 2 * cp, x, N * 2 * sizeof(float));
        } else {
            memcpy(o + 2 * cp, x, (N - W) * 2 * sizeof(float));
            for (size_t i = 0; i < W; ++i) {
                const float f = w[2 * W - 1 - i];
                o[2 * ((size_t)sym_size - W + i)] = x[2 * (N - W + i)] * f;
                o[2 * ((size_t)sym_size - W + i) + 1] = x[2 * (N - W + i) + 1] * f;
            }
            for (size_t i = 0; i < W; ++i) {
                const float f = w[W - 1 - i];
                o[2 * ((size_t)sym_size + i)] = x[2 * i] * f;
                o[2 * ((size_t)sym_size + i) + 1] = x[2 * i + 1] * f;
            }
        }
    }
    free(w);
    return 0;
}

/* ------------------------------------------------------------------- a9 */

/* src/FIRFilter.cpp:59-71; identical to doc/fir-filter/filtertaps.txt */
static const float FIR_DEFAULT[45] = {
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

const float *dabo_fir_default_taps(int *ntaps)
{
    if (ntaps) *ntaps = 45;
    return FIR_DEFAULT;
}

/* src/FIRFilter.cpp:162-192: for float index i, out[i] = sum_j in[i+2j]*taps[j]
 * accumulated from 0 in tap order, product rounded then added; past the end of
 * the frame the remaining terms are dropped (the scalar tail loop :186-191). */
void dabo_fir_filter(const float *in, size_t nsamples, const float *taps, int ntaps, float *out)
{
    const size_t nf = 2 * nsamples;
    size_t i0 = 0;
#ifdef DABO_FAST
    /* baseline build: the main loop 16 outputs at a time (the reference's runs 4 at a time in SSE registers,
     * src/FIRFilter.cpp:168-184), same order of accumulation per output */
    if (nf > 2 * (size_t)ntaps + 16) {
        for (; i0 + 16 <= nf - 2 * (size_t)ntaps; i0 += 16) {
            float acc[16];
            for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
            for (int j = 0; j < ntaps; ++j) {
                const float tp = taps[j];
                const float *x = in + i0 + 2 * (size_t)j;
                for (int v = 0; v < 16; ++v) acc[v] += x[v] * tp;
            }
            for (int v = 0; v < 16; ++v) out[i0 + v] = acc[v];
        }
    }
#endif
    for (size_t i = i0; i < nf; ++i) {
        float acc = 0.0f;
        for (int j = 0; j < ntaps && i + 2 * (size_t)j < nf; ++j) {
            const float p = in[i + 2 * (size_t)j] * taps[j];
            acc = acc + p;
        }
        out[i] = acc;
    }
}

/* ------------------------------------------------------------------ a10 */

struct dabo_resampler {
    size_t L, M, nin, nout;
    float factor;
    float *window;      /* nin */
    float *prev_in;     /* nin/2 complex */
    float *tail;        /* nout/2 complex */
    float *fft_in, *front, *back, *fft_out;
    double *w0, *w1;
};

static size_t gcd_sz(size_t a, size_t b)
{
    while (b) { size_t t = a % b; a = b; b = t; }
    return a;
}

/* src/Resampler.cpp:51-112 */
dabo_resampler *dabo_resampler_create(size_t in_rate, size_t out_rate, size_t resolution)
{
    if (!in_rate || !out_rate) return NULL;
    dabo_resampler *r = (dabo_resampler *)calloc(1, sizeof(*r));
    if (!r) return NULL;
    const size_t g = gcd_sz(in_rate, out_rate);
    r->L = out_rate / g;
    r->M = in_rate / g;
    size_t f = resolution * 2 / r->M;
    if (f & 1) ++f;
    r->nin = f * r->M;
    r->nout = f * r->L;
    const size_t big = r->nin > r->nout ? r->nin : r->nout;
    /* 1.0f / size * outputRate / inputRate, evaluated left to right in float */
    r->factor = 1.0f / (float)big * (float)out_rate / (float)in_rate;
    r->window = (float *)malloc(sizeof(float) * r->nin);
    r->prev_in = (float *)calloc(r->nin, sizeof(float));      /* nin/2 complex */
    r->tail = (float *)calloc(r->nout, sizeof(float));        /* nout/2 complex */
    r->fft_in = (float *)malloc(sizeof(float) * 2 * r->nin);
    r->front = (float *)malloc(sizeof(float) * 2 * r->nin);
    r->back = (float *)malloc(sizeof(float) * 2 * r->nout);
    r->fft_out = (float *)malloc(sizeof(float) * 2 * r->nout);
    r->w0 = (double *)malloc(sizeof(double) * 2 * big);
    r->w1 = (double *)malloc(sizeof(double) * 2 * big);
    for (size_t i = 0; i < r->nin; ++i)
        r->window[i] = (float)(0.5 * (1.0 - cos(2.0 * M_PI * (double)i / (double)(r->nin - 1))));
    return r;
}

void dabo_resampler_destroy(dabo_resampler *r)
{
    if (!r) return;
    free(r->window); free(r->prev_in); free(r->tail); free(r->fft_in);
    free(r->front); free(r->back); free(r->fft_out); free(r->w0); free(r->w1);
    free(r);
}

void dabo_resampler_geometry(const dabo_resampler *r, size_t *L, size_t *M,
                             size_t *fft_in, size_t *fft_out, float *factor)
{
    if (L) *L = r->L;
    if (M) *M = r->M;
    if (fft_in) *fft_in = r->nin;
    if (fft_out) *fft_out = r->nout;
    if (factor) *factor = r->factor;
}

/* src/Resampler.cpp:142-192 */
int dabo_resampler_process(dabo_resampler *r, const float *in, size_t nsamples, float *out)
{
    const size_t hin = r->nin / 2, hout = r->nout / 2;
    if (nsamples % hin != 0) return -1;
    for (size_t h = 0; h < nsamples / hin; ++h) {
        const float *cur = in + 2 * h * hin;
        float *o = out + 2 * h * hout;
        memcpy(r->fft_in, r->prev_in, hin * 2 * sizeof(float));
        memcpy(r->fft_in + 2 * hin, cur, hin * 2 * sizeof(float));
        memcpy(r->prev_in, cur, hin * 2 * sizeof(float));
        for (size_t k = 0; k < r->nin; ++k) {
            r->fft_in[2 * k] *= r->window[k];
            r->fft_in[2 * k + 1] *= r->window[k];
        }
        dft_f32_via_f64(r->fft_in, r->front, r->nin, -1, r->w0, r->w1);
        if (r->nout > r->nin) {
            memset(r->back, 0, r->nout * 2 * sizeof(float));
            memcpy(r->back, r->front, hin * 2 * sizeof(float));
            memcpy(r->back + 2 * (r->nout - hin), r->front + 2 * hin, hin * 2 * sizeof(float));
            r->back[2 * hin] = r->front[2 * hin];
            r->back[2 * hin + 1] = r->front[2 * hin + 1];
        } else {
            memcpy(r->back, r->front, hout * 2 * sizeof(float));
            memcpy(r->back + 2 * hout, r->front + 2 * (r->nin - hout), hout * 2 * sizeof(float));
            r->back[2 * hout] += r->front[2 * hout];
            r->back[2 * hout + 1] += r->front[2 * hout + 1];
            r->back[2 * hout] *= 0.5f;
            r->back[2 * hout + 1] *= 0.5f;
        }
        for (size_t k = 0; k < 2 * r->nout; ++k) r->back[k] *= r->factor;
        dft_f32_via_f64(r->back, r->fft_out, r->nout, +1, r->w0, r->w1);
        for (size_t k = 0; k < 2 * hout; ++k) o[k] = r->tail[k] + r->fft_out[k];
        memcpy(r->tail, r->fft_out + 2 * hout, hout * 2 * sizeof(float));
    }
    return 0;
}

/* ------------------------------------------------------------------ a11 */

/* src/MemlessPoly.cpp:237-276. The cos/sin constants are the reference's
 * literals (they are not Taylor coefficients) and are kept verbatim. */
void dabo_memless_poly(const float *in, size_t nsamples, const float am[5], const float pm[5], float *out)
{
    for (size_t i = 0; i < nsamples; ++i) {
        const float xr = in[2 * i], xi = in[2 * i + 1];
        const float m = xr * xr + xi * xi;
        const float a = am[0] + m * (am[1] + m * (am[2] + m * (am[3] + m * am[4])));
        const float p = -1 * (pm[0] + m * (pm[1] + m * (pm[2] + m * (pm[3] + m * pm[4]))));
        const float q = p * p;
        const float cr = (1.0f - q * (-0.5f + q * (0.486666f + q * (-0.00138888f))));
        const float ci = p * (1.0f + q * (0.166666f + q * (0.00833333f)));
        /* (x * a) * complex(cr, ci) */
        const float sr = xr * a, si = xi * a;
        const float rr = sr * cr, ii = si * ci, ri = sr * ci, ir = si * cr;
        out[2 * i] = rr - ii;
        out[2 * i + 1] = ri + ir;
    }
}

/* src/MemlessPoly.cpp:278-309; LUT entries are real (loaded as complexf(a,0),
 * :214-219) so the product is complex * complex with a zero imaginary part. */
void dabo_memless_lut(const float *in, size_t nsamples, float scalefactor,
                      const float lut_re[32], float *out)
{
    for (size_t i = 0; i < nsamples; ++i) {
        const float xr = in[2 * i], xi = in[2 * i + 1];
        const float mag = hypotf(xr, xi);
        const uint32_t scaled = (uint32_t)lrintf(mag * scalefactor);
        const uint8_t ix = (uint8_t)(scaled >> 27);
        const float lr = lut_re[ix], li = 0.0f;
        const float rr = xr * lr, ii = xi * li, ri = xr * li, ir = xi * lr;
        out[2 * i] = rr - ii;
        out[2 * i + 1] = ri + ir;
    }
}

/* ---------------------------------------------------------------- chain */

struct dabo_chain {
    dabo_chain_cfg cfg;
    dabo_mode_t m;
    dabo_resampler *rs;
    float *taps;
    float *phase;           /* K */
    float *a, *b;           /* ping-pong scratch */
    size_t out_per_tf;
    int mer_index;          /* myMERCalcIndex, src/OfdmGenerator.h:109 */
    dabo_cfr_stats *cfr_stats; double *cfr_papr; size_t cfr_frames;   /* last call */
    uint8_t *acp;           /* TII carrier set, K */
    int tii_insert;         /* TII::m_insert, src/TII.h:112 (starts true, toggles per frame) */
};

/* Stage order of src/DabModulator.cpp:385-419 (TII/CIC/CFR/FormatConverter off). */
dabo_chain *dabo_chain_create(const dabo_chain_cfg *cfg)
{
    dabo_chain *c = (dabo_chain *)calloc(1, sizeof(*c));
    if (!c) return NULL;
    c->cfg = *cfg;
    if (dabo_mode_params(cfg->mode, &c->m)) { free(c); return NULL; }
    const size_t tf = dabo_tf_samples(&c->m);
    c->out_per_tf = tf;
    if (cfg->stages & DABO_STAGE_FIR) {
        c->taps = (float *)malloc(sizeof(float) * (size_t)cfg->ntaps);
        memcpy(c->taps, cfg->taps, sizeof(float) * (size_t)cfg->ntaps);
    }
    if (cfg->stages & DABO_STAGE_RESAMPLE) {
        /* src/DabModulator.cpp:265-268: Resampler(2048000, outputRate, m_spacing) */
        c->rs = dabo_resampler_create(cfg->in_rate, cfg->out_rate, (size_t)c->m.spacing);
        if (!c->rs) { dabo_chain_destroy(c); return NULL; }
        c->out_per_tf = tf * c->rs->L / c->rs->M;
    }
    const size_t big = (c->out_per_tf > tf ? c->out_per_tf : tf) + 4 * (size_t)c->m.spacing;
    c->a = (float *)malloc(sizeof(float) * 2 * big);
    c->b = (float *)malloc(sizeof(float) * 2 * big);
    c->phase = (float *)malloc(sizeof(float) * 2 * (size_t)c->m.carriers);
    dabo_phase_reference(c->m.mode, c->phase, NULL);
    c->tii_insert = 1;
    if (cfg->tii_enable) {
        c->acp = (uint8_t *)calloc((size_t)c->m.carriers, 1);
        if (dabo_tii_pattern(cfg->mode, cfg->tii_comb, cfg->tii_pattern, c->acp)) {
            dabo_chain_destroy(c);
            return NULL;
        }
    }
    return c;
}

void dabo_chain_destroy(dabo_chain *c)
{
    if (!c) return;
    dabo_resampler_destroy(c->rs);
    free(c->taps); free(c->phase); free(c->a); free(c->b); free(c->acp); free(c->cfr_stats); free(c->cfr_papr);
    free(c);
}

size_t dabo_chain_out_samples_per_tf(const dabo_chain *c) { return c->out_per_tf; }

const dabo_cfr_stats *dabo_chain_cfr_stats(const dabo_chain *c, size_t f, double *papr)
{
    if (!c->cfg.cfr_enable || !c->cfr_stats || f >= c->cfr_frames) return NULL;
    const size_t nsym = (size_t)c->m.nb_symbols + 1;
    if (papr) memcpy(papr, c->cfr_papr + f * nsym * 4, nsym * 4 * sizeof(double));
    return &c->cfr_stats[f];
}

int dabo_chain_process(dabo_chain *c, const uint8_t *bits, size_t nframes, float *out)
{
    const dabo_mode_t *m = &c->m;
    const size_t K = (size_t)m->carriers, N = (size_t)m->spacing;
    const size_t ndata = (size_t)(m->nb_symbols - 1) * K, nsym = (size_t)m->nb_symbols + 1;
    const size_t tf = dabo_tf_samples(m), inb = dabo_tf_input_bytes(m);
    if (c->cfg.cfr_enable) {
        free(c->cfr_stats); free(c->cfr_papr);
        c->cfr_stats = (dabo_cfr_stats *)calloc(nframes ? nframes : 1, sizeof(dabo_cfr_stats));
        c->cfr_papr = (double *)calloc((nframes ? nframes : 1) * nsym * 4, sizeof(double));
        c->cfr_frames = nframes;
    }
    for (size_t f = 0; f < nframes; ++f) {
        float *a = c->a, *b = c->b, *t;
        int rc = 0;
        rc |= dabo_qpsk_map(bits + f * inb, inb, m->carriers, a);
        rc |= dabo_freq_interleave(a, ndata, m->mode, b);
        /* null symbol (K zeros), or the TII symbol (SignalMultiplexer input 2,
         * src/SignalMultiplexer.cpp:63-66) ++ diff-mod output */
        memset(a, 0, K * 2 * sizeof(float));
        if (c->acp) {
            dabo_tii_process(c->phase, m->carriers, c->acp, c->cfg.tii_old_variant, c->tii_insert, a);
            c->tii_insert = !c->tii_insert;
        }
        rc |= dabo_diff_mod(c->phase, b, ndata, m->carriers, a + 2 * K);
        if (c->cfg.cfr_enable) {
            c->mer_index = (c->mer_index + 1) % (int)nsym;           /* src/OfdmGenerator.cpp:198 */
            rc |= dabo_ofdm_generate_cfr(a, (int)nsym, m->carriers, m->spacing, c->cfg.cfr_clip,
                                         c->cfg.cfr_error_clip, c->mer_index, b, &c->cfr_stats[f],
                                         c->cfr_papr + f * nsym * 4);
        } else {
            rc |= dabo_ofdm_generate(a, (int)nsym, m->carriers, m->spacing, b);
        }
        t = a; a = b; b = t;                       /* a = ofdm out */
        if (c->cfg.stages & DABO_STAGE_GAIN) {
            rc |= dabo_gain_control(a, nsym * N, m->spacing, c->cfg.gain_mode, c->cfg.dig_gain,
                                    c->cfg.normalise, c->cfg.var_variance, b, NULL);
            t = a; a = b; b = t;
        }
        rc |= dabo_guard_interval(a, m->nb_symbols, m->spacing, m->null_size, m->sym_size,
                                  c->cfg.window_overlap, b);
        t = a; a = b; b = t;
        if (c->cfg.stages & DABO_STAGE_FIR) {
            dabo_fir_filter(a, tf, c->taps, c->cfg.ntaps, b);
            t = a; a = b; b = t;
        }
        size_t n = tf;
        if (c->cfg.stages & DABO_STAGE_RESAMPLE) {
            rc |= dabo_resampler_process(c->rs, a, tf, b);
            n = c->out_per_tf;
            t = a; a = b; b = t;
        }
        if (c->cfg.stages & DABO_STAGE_POLY) {
            dabo_memless_poly(a, n, c->cfg.am, c->cfg.pm, b);
            t = a; a = b; b = t;
        }
        if (rc) return -1;
        memcpy(out + 2 * f * c->out_per_tf, a, n * 2 * sizeof(float));
    }
    return 0;
}

#ifdef DABO_FAST
/* ---------------------------------------------------------------------------
 * CPU-BASELINE BUILD ONLY: one stream in the reference's THREADING MODEL (src/ModPlugin.cpp:90-154).  The caller is
 * the modulator thread and runs every plain ModCodec (mapper, interleaver, differential modulator, OfdmGenerator,
 * GuardIntervalInserter, Resampler); GainControl, FIRFilter and MemlessPoly are PipelinedModCodecs: process() hands
 * the frame to the stage's own thread and returns the PREVIOUS frame's result (nothing on the first call, which
 * stops the flowgraph walk for that round, src/Flowgraph.cpp:325-337), and MemlessPoly splits every frame over
 * `poly_threads` workers (src/MemlessPoly.cpp:352-391).  Returns the frames that reached the output. */
#include <pthread.h>

typedef struct pstage {
    pthread_t th;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int busy, has_out, quit;
    float *in, *out;
    size_t n;
    int kind;                 /* 0 gain, 1 fir, 2 poly */
    dabo_chain *c;
    int poly_threads;
} pstage;

typedef struct { const float *in; float *out; size_t n; const dabo_chain *c; } poly_job;
static void *poly_part(void *p)
{
    poly_job *j = (poly_job *)p;
    dabo_memless_poly(j->in, j->n, j->c->cfg.am, j->c->cfg.pm, j->out);
    return NULL;
}

static void pstage_run(pstage *st)
{
    dabo_chain *c = st->c;
    const dabo_mode_t *m = &c->m;
    if (st->kind == 0) {
        dabo_gain_control(st->in, st->n, m->spacing, c->cfg.gain_mode, c->cfg.dig_gain, c->cfg.normalise,
                          c->cfg.var_variance, st->out, NULL);
    } else if (st->kind == 1) {
        dabo_fir_filter(st->in, st->n, c->taps, c->cfg.ntaps, st->out);
    } else {
        /* worker t handles [t step, (t + 1) step), the stage's own thread the remainder */
        const int nt = st->poly_threads > 0 ? st->poly_threads : 0;
        const size_t step = nt ? st->n / (size_t)nt : 0;
        pthread_t th[64];
        poly_job job[64];
        for (int t = 0; t < nt && t < 64; ++t) {
            job[t].in = st->in + 2 * (size_t)t * step; job[t].out = st->out + 2 * (size_t)t * step;
            job[t].n = step; job[t].c = c;
            pthread_create(&th[t], NULL, poly_part, &job[t]);
        }
        const size_t done = (size_t)(nt < 64 ? nt : 64) * step;
        dabo_memless_poly(st->in + 2 * done, st->n - done, c->cfg.am, c->cfg.pm, st->out + 2 * done);
        for (int t = 0; t < nt && t < 64; ++t) pthread_join(th[t], NULL);
    }
}

static void *pstage_main(void *p)
{
    pstage *st = (pstage *)p;
    pthread_mutex_lock(&st->mu);
    for (;;) {
        while (!st->busy && !st->quit) pthread_cond_wait(&st->cv, &st->mu);
        if (st->quit) break;
        pthread_mutex_unlock(&st->mu);
        pstage_run(st);
        pthread_mutex_lock(&st->mu);
        st->busy = 0;
        st->has_out = 1;
        pthread_cond_broadcast(&st->cv);
    }
    pthread_mutex_unlock(&st->mu);
    return NULL;
}

/* hand `in` (n samples) to the stage with `spare` as its next output buffer; returns the previous result (the
 * caller owns it now) or NULL on the first call, and *freed = the input buffer the stage has finished with */
static float *pstage_swap(pstage *st, float *in, size_t n, float *spare, float **freed)
{
    pthread_mutex_lock(&st->mu);
    while (st->busy) pthread_cond_wait(&st->cv, &st->mu);
    float *res = st->has_out ? st->out : NULL;
    *freed = st->has_out ? st->in : NULL;
    st->in = in; st->out = spare; st->n = n; st->has_out = 0; st->busy = 1;
    pthread_cond_broadcast(&st->cv);
    pthread_mutex_unlock(&st->mu);
    return res;
}

int dabo_chain_process_pipelined(dabo_chain *c, const uint8_t *bits, size_t nframes, int poly_threads, float *out,
                                 size_t *frames_out)
{
    const dabo_mode_t *m = &c->m;
    const size_t K = (size_t)m->carriers, N = (size_t)m->spacing;
    const size_t ndata = (size_t)(m->nb_symbols - 1) * K, nsym = (size_t)m->nb_symbols + 1;
    const size_t tf = dabo_tf_samples(m), inb = dabo_tf_input_bytes(m);
    const size_t big = (c->out_per_tf > tf ? c->out_per_tf : tf) + 4 * N;
    float *pool[16];
    int npool = 0;
#define GETBUF() (npool ? pool[--npool] : (float *)malloc(sizeof(float) * 2 * big))
#define PUTBUF(b) do { if (b) pool[npool++] = (b); } while (0)
    pstage st[3];
    const int use[3] = {!!(c->cfg.stages & DABO_STAGE_GAIN), !!(c->cfg.stages & DABO_STAGE_FIR),
                        !!(c->cfg.stages & DABO_STAGE_POLY)};
    for (int i = 0; i < 3; ++i) {
        memset(&st[i], 0, sizeof st[i]);
        if (!use[i]) continue;
        st[i].kind = i; st[i].c = c; st[i].poly_threads = poly_threads;
        pthread_mutex_init(&st[i].mu, NULL);
        pthread_cond_init(&st[i].cv, NULL);
        pthread_create(&st[i].th, NULL, pstage_main, &st[i]);
    }
    size_t nout = 0;
    int rc = 0;
    for (size_t f = 0; f < nframes && !rc; ++f) {
        float *a = GETBUF(), *b = GETBUF(), *freed = NULL, *t;
        rc |= dabo_qpsk_map(bits + f * inb, inb, m->carriers, a);
        rc |= dabo_freq_interleave(a, ndata, m->mode, b);
        memset(a, 0, K * 2 * sizeof(float));
        rc |= dabo_diff_mod(c->phase, b, ndata, m->carriers, a + 2 * K);
        rc |= dabo_ofdm_generate(a, (int)nsym, m->carriers, m->spacing, b);
        PUTBUF(a);
        float *cur = b;                                           /* OfdmGenerator output */
        if (use[0]) {
            t = pstage_swap(&st[0], cur, nsym * N, GETBUF(), &freed);
            PUTBUF(freed);
            if (!(cur = t)) continue;
        }
        a = GETBUF();
        rc |= dabo_guard_interval(cur, m->nb_symbols, m->spacing, m->null_size, m->sym_size, c->cfg.window_overlap, a);
        PUTBUF(cur);
        cur = a;
        if (use[1]) {
            t = pstage_swap(&st[1], cur, tf, GETBUF(), &freed);
            PUTBUF(freed);
            if (!(cur = t)) continue;
        }
        size_t n = tf;
        if (c->cfg.stages & DABO_STAGE_RESAMPLE) {
            a = GETBUF();
            rc |= dabo_resampler_process(c->rs, cur, tf, a);
            PUTBUF(cur);
            cur = a;
            n = c->out_per_tf;
        }
        if (use[2]) {
            t = pstage_swap(&st[2], cur, n, GETBUF(), &freed);
            PUTBUF(freed);
            if (!(cur = t)) continue;
        }
        memcpy(out + 2 * nout * c->out_per_tf, cur, n * 2 * sizeof(float));
        ++nout;
        PUTBUF(cur);
    }
    for (int i = 0; i < 3; ++i) {
        if (!use[i]) continue;
        pthread_mutex_lock(&st[i].mu);
        while (st[i].busy) pthread_cond_wait(&st[i].cv, &st[i].mu);
        st[i].quit = 1;
        pthread_cond_broadcast(&st[i].cv);
        pthread_mutex_unlock(&st[i].mu);
        pthread_join(st[i].th, NULL);
        free(st[i].in);
        free(st[i].out);
    }
    while (npool) free(pool[--npool]);
#undef GETBUF
#undef PUTBUF
    if (frames_out) *frames_out = nout;
    return rc ? -1 : 0;
}
#endif

/* ---------------------------------------------------------------------------
 * f-4 TII (reference src/TII.cpp).  The 70 patterns of EN 300 401 table 64 are the 8-bit
 * words of weight 4 in increasing numerical order (:34-104), bit b = 0 being the leftmost. */
static int tii_pattern_bit(int pattern, int b)
{
    int idx = 0;
    for (int w = 0; w < 256; ++w) {
        int pop = 0;
        for (int i = 0; i < 8; ++i) pop += (w >> i) & 1;
        if (pop != 4) continue;
        if (idx == pattern) return (w >> (7 - b)) & 1;
        ++idx;
    }
    return 0;
}

static int tii_enable_carrier(int carriers, int k, uint8_t *acp)
{
    const int ix = carriers / 2 + k + (k >= 0 ? -1 : 0);      /* :251-256 */
    if (ix < 0 || ix + 1 >= carriers) return -1;               /* :258-260 */
    acp[ix] = 1;
    return 0;
}

int dabo_tii_pattern(int mode, int comb, int pattern, uint8_t *acp)
{
    if (mode != 1 && mode != 2) return -1;                     /* :119-145 */
    if (pattern < 0 || pattern > 69 || comb < 0 || comb > 23) return -1;
    const int carriers = mode == 1 ? 1536 : 384;
    memset(acp, 0, (size_t)carriers);
    int rc = 0;
    if (mode == 1) {
        /* :280-316: k = base + 2 comb + 48 b inside each quarter of the spectrum */
        static const int base[4] = {-768, -384, 1, 385};
        static const int last[4] = {-385, -1, 384, 768};
        for (int q = 0; q < 4; ++q)
            for (int b = 0; b < 8; ++b) {
                const int k = base[q] + 2 * comb + 48 * b;
                if (k <= last[q] && tii_pattern_bit(pattern, b)) rc |= tii_enable_carrier(carriers, k, acp);
            }
    } else {
        /* :317-333: -192 + 2c + 48b for b < 4, -191 + 2c + 48b for b >= 4, k in [-192, 192] */
        for (int b = 0; b < 8; ++b) {
            const int k = (b < 4 ? -192 : -191) + 2 * comb + 48 * b;
            if (k >= -192 && k <= 192 && tii_pattern_bit(pattern, b)) rc |= tii_enable_carrier(carriers, k, acp);
        }
    }
    return rc;
}

void dabo_tii_process(const float *in, int carriers, const uint8_t *acp, int old_variant, int insert,
                      float *out)
{
    memset(out, 0, sizeof(float) * 2 * (size_t)carriers);      /* :223-224 */
    if (!insert) return;
    for (int i = 0; i < carriers; ++i) {                        /* :186-210 */
        if (!acp[i]) continue;
        out[2 * i] = in[2 * i];
        out[2 * i + 1] = in[2 * i + 1];
        const int j = old_variant ? i + 1 : i;
        out[2 * (i + 1)] = in[2 * j];
        out[2 * (i + 1) + 1] = in[2 * j + 1];
    }
}

/* ---------------------------------------------------------------------------
 * f-2 FormatConverter, float input (reference src/FormatConverter.cpp:111-178):
 * range test against the integer limits, clipped components counted, otherwise the
 * C float -> integer conversion (truncation toward zero).  u8 adds 128.0f first. */
size_t dabo_format_convert(const float *in, size_t n, int fmt, void *out)
{
    size_t clipped = 0;
    if (fmt == DABO_FMT_S16) {
        int16_t *o = (int16_t *)out;                 /* :116-131 */
        for (size_t i = 0; i < n; ++i) {
            if (in[i] < INT16_MIN) { o[i] = INT16_MIN; ++clipped; }
            else if (in[i] > INT16_MAX) { o[i] = INT16_MAX; ++clipped; }
            else o[i] = (int16_t)in[i];
        }
    } else if (fmt == DABO_FMT_U8) {
        uint8_t *o = (uint8_t *)out;                 /* :133-151 */
        for (size_t i = 0; i < n; ++i) {
            const float samp = in[i] + 128.0f;
            if (samp < 0) { o[i] = 0; ++clipped; }
            else if (samp > UINT8_MAX) { o[i] = UINT8_MAX; ++clipped; }
            else o[i] = (uint8_t)samp;
        }
    } else if (fmt == DABO_FMT_S8) {
        int8_t *o = (int8_t *)out;                   /* :153-169 */
        for (size_t i = 0; i < n; ++i) {
            if (in[i] < INT8_MIN) { o[i] = INT8_MIN; ++clipped; }
            else if (in[i] > INT8_MAX) { o[i] = INT8_MAX; ++clipped; }
            else o[i] = (int8_t)in[i];
        }
    } else {
        return (size_t)-1;                           /* :171-173 */
    }
    return clipped;
}
