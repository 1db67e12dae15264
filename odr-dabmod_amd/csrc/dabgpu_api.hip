// dabgpu_api.hip -- the C-ABI of include/dabgpu.h: context, device tables,
// host staging, and the mapping from reference plugins to kernel launches.

#include "dabgpu.h"
#include "dabgpu_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

using namespace dabgpu;

namespace {

thread_local std::string g_create_error;

// src/FIRFilter.cpp:59-71 == doc/fir-filter/filtertaps.txt (configuration data)
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

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct Settings {
    int gain_mode = DABGPU_GAIN_VAR;      // src/ConfigParser.h:60-91 defaults
    float digital = 1.0f, normalise = 1.0f, var_variance = 4.0f;
    std::vector<float> taps;
    size_t overlap = 0;
    size_t rs_in = 2048000, rs_out = 2048000;
    bool poly_is_lut = false;
    float am[5] = {1, 0, 0, 0, 0}, pm[5] = {0, 0, 0, 0, 0};
    float lut_scale = 0.f, lut[32] = {0};
    bool cfr_enable = false;               // src/ConfigParser.h: enableCfr / cfrClip / cfrErrorClip
    float cfr_clip = 1.0f, cfr_errclip = 1.0f;
    int out_format = 0;                    // 0 = complexf, else DABGPU_FMT_*: FormatConverter as the chain's last step
    bool tii_enable = false, tii_old_variant = false;   // src/TII.h:42-69 (tii_config_t)
    int tii_comb = 0, tii_pattern = 0;
    unsigned long long epoch = 1;  // bumped by every setter
    bool resampler_reset = true;

    // What each group of device data is a function of.  apply_settings_groups compares these keys, never single fields: a
    // setting that starts to feed a table is added to that table's key HERE, next to its declaration.
    //   the fused FIR's tap table, its frequency response and the inverse filter of the equalised-boundary variant
    auto fir_key() const { return std::tie(taps); }
    //   the raised-cosine window of the guard interval
    auto window_key() const { return std::tie(overlap); }
    //   the predistorter's coefficient block (polynomial and LUT share it; the selector and the LUT scale are kernel arguments)
    bool coef_equal(const Settings &o) const
    {
        return poly_is_lut == o.poly_is_lut && lut_scale == o.lut_scale && !std::memcmp(am, o.am, sizeof am) &&
               !std::memcmp(pm, o.pm, sizeof pm) && !std::memcmp(lut, o.lut, sizeof lut);
    }
    //   the resampler's window, twiddles and geometry
    auto resampler_key() const { return std::tie(rs_in, rs_out); }
    //   the cached unit-gain TII segment (TII symbol -> IFFT -> [CFR] -> guard [window] -> [FIR]); gain scales it at use
    auto tii_segment_key() const
    {
        return std::tie(taps, overlap, tii_comb, tii_pattern, tii_old_variant, cfr_enable, cfr_clip, cfr_errclip);
    }
};

}  // namespace

// ---- launch trace (dabgpu_internal.h) -------------------------------------------------------------------------------
namespace dabgpu {
namespace {
thread_local std::string *g_trace_sink = nullptr;
}
bool trace_on() { return g_trace_sink != nullptr; }
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

namespace {
// names the kernels of one chain call into ctx->last_variant
struct TraceScope {
    std::string *prev;
    // (sink == nullptr: tracing is off for this context -- nothing is installed, a launch costs one pointer test)
    explicit TraceScope(std::string *sink) : prev(g_trace_sink_ref())
    {
        if (sink) sink->clear();
        g_trace_sink_ref() = sink;
    }
    ~TraceScope() { g_trace_sink_ref() = prev; }
    static std::string *&g_trace_sink_ref() { return dabgpu::g_trace_sink; }
};
}  // namespace

struct dabgpu_ctx {
    std::string last_variant;             // dabgpu_debug_last_variant: the kernels the most recent chain call launched
    bool trace_enabled = false;           // dabgpu_debug_trace: off by default (names are formatted per launch when on)
    Geometry g{};
    int device = 0;
    int max_frames = 1;
    int chunks_cfg = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // constant tables
    DevBuf d_twiddle, d_src, d_dst, d_phq, d_mag, d_taps, d_firh, d_window, d_coef, d_eqg;
    bool use_eq = true;                   // dabgpu_set_fir_boundary_mode: false = always the packed dual transform
    bool eq_ok = false;                   // d_eqg holds a well-conditioned inverse of the current taps (TF_EQ may be used)
    double eq_fit = 0.0;                  // max |G H - 1| over the occupied bins
    // resampler
    DevBuf d_rs_window, d_rs_tw_in, d_rs_tw_out, d_rs_halo, d_rs_tw_s, d_rs_tw_l;
    int rs_nin = 0, rs_nout = 0;
    int rs_halo_cur = 0;                  // which of the two halo buffers holds the state the next call reads
    size_t rs_L = 1, rs_M = 1;
    float rs_factor = 1.f;
    // scratch
    DevBuf d_a, d_b, d_c, d_in, d_out, d_count, d_fmt, d_clip;
    DevBuf d_phase;                        // tool builds only (-DDABGPU_PHASE_TIMING): the frame kernel's per-phase cycle counters
    hipStream_t clip_stream = nullptr;     // stream of the most recent chain call that converted its output
    // TII (f-4): carrier set, the one-frame carrier image and its native-rate response, gain of symbol 1
    DevBuf d_acp, d_tii_car, d_tii_frame, d_gain1, d_cic;
    size_t cic_spacing = 0;               // what d_cic was built for (CicEqualizer, a12)
    int cic_R = 0;
    // CFR statistics (f-3) of the most recent chain / OfdmGenerator call, and a scratch set for internal runs
    DevBuf d_cfr_counts, d_cfr_mer, d_cfr_papr, d_cfr_tmp;
    int cfr_mer_index = 0;                // myMERCalcIndex (src/OfdmGenerator.h:109): advances once per frame
    int cfr_last_base = 0;
    size_t cfr_last_frames = 0;
    hipStream_t cfr_last_stream = nullptr;
    bool tii_insert = true;               // TII::m_insert (src/TII.h:112): this frame of the stream carries TII
    bool tables_valid = false;            // apply_settings has uploaded every table group once
    unsigned long long tii_seg_epoch = 0; // 1 while the cached segment matches the settings (apply_settings zeroes it), and its stage mask
    unsigned tii_seg_mask = ~0u;
    int tii_seg_len = 0;

    // Batches in flight inside ONE context (the idiom of PipelinedModCodec, src/ModPlugin.cpp:90-154: the caller hands over
    // batch i + 1 while batch i is still being worked on).  A chain call on the context's own stream (stream argument NULL)
    // goes to one of n_lanes internal HIP streams in turn; every lane has its own per-call scratch, so the kernels of
    // consecutive calls overlap where one launch alone cannot fill the chip.  Lane 0 is `stream` and the scratch members
    // above; LaneScope swaps another lane's buffers in for the duration of a call.  Calls with the Resampler stay on lane 0
    // (its state runs from frame to frame).
    struct Lane {
        hipStream_t stream = nullptr;
        hipEvent_t ev = nullptr;
        DevBuf d_a, d_b, d_fmt, d_clip, d_gain1, d_cfr_counts, d_cfr_mer, d_cfr_papr, d_cfr_tmp;
    };
    enum { kMaxLanes = 4, kLaneMaxFrames = 2048, kLaneScratchBytes = 256 << 20 };
    Lane lane[kMaxLanes];                 // (entry 0: only `ev` is used)
    bool lane_own_queue[kMaxLanes] = {true, false, false, false};   // the probe found the lane a hardware queue of its own
    int n_lanes = 3;
    int call_lanes = 1;                   // lanes the CURRENT chain call rotates over (1: an explicit stream, lane 0 only)
    unsigned long long lane_seq = 0;
    int clip_lane = 0, cfr_last_lane = 0; // whose scratch holds the clip count / the CFR statistics of the most recent call
    // Ordering between the lanes and the context's own stream for the NULL-stream entry points that do NOT rotate
    // (dabgpu_format_process_dev, dabgpu_post_process_dev): they queue on `stream` behind everything the lanes hold
    // (lane_dirty: the lane has work `stream` has not been ordered behind yet), and a later chain call that goes to
    // another lane is ordered behind them (own_epoch / lane_seen_epoch).
    bool lane_dirty[kMaxLanes] = {false, false, false, false};
    unsigned long long own_epoch = 0, lane_seen_epoch[kMaxLanes] = {0, 0, 0, 0};
    hipEvent_t own_ev = nullptr;
    // The native-rate stream between FIRFilter and Resampler (src/DabModulator.cpp:403-406) in pieces of this many frames
    // through a two-piece ring that stays cache-resident, produced on lane 1's stream while the consumer works on the
    // piece before (dabgpu_set_handover_frames; 0 = one piece, the whole batch through memory)
    int handover_frames = 0;
    hipEvent_t ho_prod[2] = {nullptr, nullptr}, ho_cons[2] = {nullptr, nullptr}, ho_start = nullptr, ho_join = nullptr;

    std::mutex mu;
    Settings set;                    // guarded by mu
    Settings cur;                    // snapshot used by the processing thread
    unsigned long long applied_epoch = 0;


    // asynchronous host path (dabgpu_chain_submit / dabgpu_chain_collect): two batches in flight,
    // pinned staging on both sides, device->host copies on their own stream
    struct Slot {
        void *h_in = nullptr;                          // pinned (hipHostMalloc)
        size_t h_in_cap = 0, out_bytes = 0;
        int h_out_index = 0;                           // which of the three pinned output buffers this batch lands in
        unsigned long long *h_clip = nullptr;          // pinned: this batch's clipped-component count (output formats)
        DevBuf d_in, d_out;
        hipEvent_t computed = nullptr, copied = nullptr;
        hipStream_t stream = nullptr;                  // the lane this batch's kernels were queued on
        bool busy = false;
        int out_format = 0;                            // the output format this batch was submitted with
    } slot[2];
    // Pinned output buffers, THREE for two batches in flight: submit n copies into buffer n mod 3, so the
    // buffer handed out by collect() of batch n is next written by submit n + 3 -- after the collect at the
    // latest the second next submit.  (With one buffer per slot the very next submit overwrote it.)
    void *h_out[3] = {nullptr, nullptr, nullptr};
    size_t h_out_cap[3] = {0, 0, 0};
    unsigned long long submit_seq = 0;
    bool clip_from_collect = false;        // dabgpu_get_num_clipped answers for the batch collect() returned last
    bool clip_valid = false;               // the most recent chain call converted its output (d_clip holds ITS count)
    size_t collected_clipped = 0;
    hipStream_t copy_stream = nullptr;
    int slot_head = 0, slot_count = 0;                 // oldest batch in flight, number in flight
};

namespace {

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

#define HIPCHK(ctx, expr)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #expr);                                 \
    } while (0)

// the stream of lane i (created on first use, lanes 1 .. i in order; lane 0 is the context's stream)
int lane_stream(dabgpu_ctx *c, int i, hipStream_t *out)
{
    if (i == 0) { *out = c->stream; return DABGPU_OK; }
    for (int k = 1; k <= i; ++k) {
        if (c->lane[k].stream) continue;
        std::vector<hipStream_t> others{c->stream};
        for (int j = 1; j < k; ++j) others.push_back(c->lane[j].stream);
        if (c->copy_stream) others.push_back(c->copy_stream);
        HIPCHK(c, create_stream_apart(others.data(), (int)others.size(), &c->lane[k].stream, &c->lane_own_queue[k]));
    }
    *out = c->lane[i].stream;
    return DABGPU_OK;
}

// A NULL-stream call that runs on the context's own stream: behind everything the other lanes have been given.
int own_stream_joins_lanes(dabgpu_ctx *c)
{
    for (int i = 1; i < (int)dabgpu_ctx::kMaxLanes; ++i) {
        if (!c->lane[i].stream || !c->lane_dirty[i]) continue;
        if (!c->lane[i].ev) HIPCHK(c, hipEventCreateWithFlags(&c->lane[i].ev, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(c->lane[i].ev, c->lane[i].stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->lane[i].ev, 0));
        c->lane_dirty[i] = false;
    }
    ++c->own_epoch;            // (what follows on `stream` is work the lanes have not been ordered behind)
    return DABGPU_OK;
}

// ... and a chain call that goes to lane i: behind such work of the context's own stream
int lane_joins_own_stream(dabgpu_ctx *c, int i)
{
    if (i == 0) return DABGPU_OK;
    c->lane_dirty[i] = true;
    if (c->lane_seen_epoch[i] == c->own_epoch) return DABGPU_OK;
    if (!c->own_ev) HIPCHK(c, hipEventCreateWithFlags(&c->own_ev, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->own_ev, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->lane[i].stream, c->own_ev, 0));
    c->lane_seen_epoch[i] = c->own_epoch;
    return DABGPU_OK;
}

// every stream of the context idle (before a table that kernels in flight on ANY lane may read is rewritten)
int drain_lanes(dabgpu_ctx *c)
{
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 1; i < (int)dabgpu_ctx::kMaxLanes; ++i)
        if (c->lane[i].stream) HIPCHK(c, hipStreamSynchronize(c->lane[i].stream));
    return DABGPU_OK;
}

// lane i's per-call scratch in place of the context's for the lifetime of the object
struct LaneScope {
    dabgpu_ctx *c;
    int i;
    LaneScope(dabgpu_ctx *ctx, int lane) : c(ctx), i(lane) { swap(); }
    ~LaneScope() { swap(); }
    LaneScope(const LaneScope &) = delete;
    LaneScope &operator=(const LaneScope &) = delete;
    void swap()
    {
        if (i == 0) return;
        dabgpu_ctx::Lane &l = c->lane[i];
        std::swap(c->d_a, l.d_a); std::swap(c->d_b, l.d_b); std::swap(c->d_fmt, l.d_fmt); std::swap(c->d_clip, l.d_clip);
        std::swap(c->d_gain1, l.d_gain1); std::swap(c->d_cfr_counts, l.d_cfr_counts); std::swap(c->d_cfr_mer, l.d_cfr_mer);
        std::swap(c->d_cfr_papr, l.d_cfr_papr); std::swap(c->d_cfr_tmp, l.d_cfr_tmp);
    }
};

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

template <typename T> hipError_t upload(DevBuf &b, const std::vector<T> &v, hipStream_t s)
{
    hipError_t e = b.reserve(std::max<size_t>(v.size() * sizeof(T), 16));
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(s);  // v may be a temporary
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

int auto_chunks(const dabgpu_ctx *c, size_t n_frames)
{
    if (c->chunks_cfg > 0) return c->chunks_cfg;
    // One workgroup per frame once the batch alone fills the chip (1024 workgroups: four per CU); below that frames are
    // split into runs of symbols so that the launch still has about 1024 of them.  Every run pays a prologue (the
    // differential state up to its first symbol: a bit-sliced sum over the blocks before it, a few microseconds whatever
    // the depth) and, with FIR, one look-ahead transform.  Measured optimum, Mode I (tools/sweep_chunks.py, round 3):
    // 1024 / B runs down to B = 32, two symbols per run for 14 ... 31 frames, single symbols below (latency, not
    // efficiency, counts there: 10 us per Mode-I frame).
    // With the call rotating over L lanes (section 4.5 of DESIGN.md), L launches are in flight and the chip is filled by
    // FEWER, LONGER runs per launch -- and every run saved is a prologue and a look-ahead transform saved: measured optimum
    // with three lanes (tools/exp_r05.py chunks, profiles/r05_exp_chunks.jsonl) 26 runs per frame at 16 frames (416 workgroups;
    // +10 % over 624), 6 ... 8 at 64 (+14 % over 1024), 2 at 256 (+4 %): about 1280 / L workgroups per launch.
    const int nsym = c->g.nb_symbols + 1;
    const size_t n = n_frames;
    const size_t target = c->call_lanes > 1 ? std::max<size_t>(256, 1280 / (size_t)c->call_lanes) : 1024;
    const int want = std::max(1, std::min(n >= target ? 1 : (int)((target + n - 1) / n), nsym));
    // no empty runs: the callers give every run ceil(nsym / chunks) symbols, so ask for exactly as many runs as that
    // run length needs (74 wanted -> 2 symbols per run -> 39 runs, not 74 workgroups of which 35 return after the prologue)
    const int per_run = (nsym + want - 1) / want;
    return (nsym + per_run - 1) / per_run;
}

// Symbols per run of a frame cut into `chunks` runs.  A run of a chain with FIRFilter or a windowed guard interval transforms
// one symbol MORE than it stores (the look-ahead symbol its last boundary needs) -- except the frame's last run, which ends
// with the frame.  So the last run takes one symbol more than the others where that evens them out: 77 symbols in four runs
// are 19 + 1, 19 + 1, 19 + 1, 20 transforms, not 20 + 1, 20 + 1, 20 + 1, 17 (the kernel gives the last run whatever is left).
int run_symbols(int nsym, int chunks, bool lookahead)
{
    return std::max(1, (nsym - (lookahead ? 1 : 0) + chunks - 1) / chunks);
}

bool is_pow2(size_t x) { return x && !(x & (x - 1)); }

// ratios with a dedicated kernel (integer 2 and 4: packed dual transforms, fused predistorter)
bool resampler_fast_ratio(const dabgpu_ctx *c)
{
    return c->rs_nout % c->rs_nin == 0 && (c->rs_nout / c->rs_nin == 2 || c->rs_nout / c->rs_nin == 4);
}

// Ratios the kernels cover: L / M (reduced) with M a power of two up to the FFT size N of the transmission mode,
// any L -- up- and down-sampling.  Then nin = 2 N is a power of two and the nout = (nin / M) L point transform
// factors into L branches of nin / M points.  Every other ratio is one the reference itself cannot run on whole
// transmission frames: with M = 2^a 5^b, b > 0 (the input rate is 2 048 000 = 2^14 5^3), half its FFT size does
// not divide the frame length, and its hop loop (src/Resampler.cpp:142) runs past the input buffer; with M > N
// its `factor` is 1 or 0 (src/Resampler.cpp:69-75).
const char *resampler_ratio_error(int N, size_t in_rate, size_t out_rate)
{
    if (!in_rate || !out_rate) return "Resampler: invalid rate";
    if (in_rate == out_rate) return nullptr;
    size_t a = in_rate, b = out_rate;
    while (b) { size_t t = a % b; a = b; b = t; }
    const size_t L = out_rate / a, M = in_rate / a;
    if (!is_pow2(M) || M > (size_t)N)
        return "Resampler: only ratios L/M with M a power of two up to the FFT size are supported "
               "(the reference's hop size does not divide a transmission frame for any other)";
    if ((2 * (size_t)N / M) * L > ((size_t)1 << 20)) return "Resampler: output FFT size beyond 2^20";
    return nullptr;
}

int check_resampler(dabgpu_ctx *c)
{
    const char *e = resampler_ratio_error(c->g.N, c->cur.rs_in, c->cur.rs_out);
    if (e) return fail(c, DABGPU_E_INVALID, e);
    if ((size_t)c->rs_nin != 2 * (size_t)c->g.N || (size_t)c->rs_nout != (size_t)c->rs_nin / c->rs_M * c->rs_L)
        return fail(c, DABGPU_E_INVALID, "Resampler: inconsistent geometry");
    return DABGPU_OK;
}

// stream of `total` samples at d_in -> resampled at d_out (stateful)
int run_resampler(dabgpu_ctx *c, const float2 *d_in, size_t total, float2 *d_out, hipStream_t s,
                  bool fuse_poly = false, unsigned long long *s16_clipped = nullptr)
{
    int rc = check_resampler(c);
    if (rc) return rc;
    const size_t hin = (size_t)c->rs_nin / 2;
    if (total % hin) return fail(c, DABGPU_E_INVALID, "Resampler::process input size not valid!");
    const size_t nhops = total / hin;
    if (nhops == 0) return DABGPU_OK;     // (nothing in, nothing out, the state -- halo buffers included -- as it was)
    ResamplerArgs a{};
    a.nin = c->rs_nin; a.nout = c->rs_nout; a.factor = c->rs_factor;
    a.window = (const float *)c->d_rs_window.p;
    a.tw_in = (const float2 *)c->d_rs_tw_in.p;
    a.tw_out = (const float2 *)c->d_rs_tw_out.p;
    float2 *halo = (float2 *)c->d_rs_halo.p + (size_t)c->rs_halo_cur * (size_t)c->rs_nin;
    float2 *halo_next = (float2 *)c->d_rs_halo.p + (size_t)(c->rs_halo_cur ^ 1) * (size_t)c->rs_nin;
    a.in = d_in; a.halo = halo;
    a.out = d_out; a.nhops = nhops;
    a.poly = (fuse_poly && resampler_fast_ratio(c)) ? (const float *)c->d_coef.p : nullptr;
    a.clipped = s16_clipped;
    a.L = (int)c->rs_L;
    a.M = (int)c->rs_M;
    a.tw_s = (const float2 *)c->d_rs_tw_s.p;
    a.tw_l = (const float2 *)c->d_rs_tw_l.p;
    // new halo = last two hops of the concatenation [halo | in]: the x2 / x4 kernel of Mode I writes it itself, into the
    // other buffer (launches of one stream are in order: the next call reads what this one wrote)
    if (resampler_writes_halo(a)) {
        a.halo_out = halo_next;
        HIPCHK(c, launch_resampler(a, s));
        c->rs_halo_cur ^= 1;
        return DABGPU_OK;
    }
    HIPCHK(c, launch_resampler(a, s));
    if (nhops >= 2) {
        HIPCHK(c, hipMemcpyAsync(halo, d_in + (nhops - 2) * hin, 2 * hin * sizeof(float2),
                                 hipMemcpyDeviceToDevice, s));
    } else {
        HIPCHK(c, hipMemcpyAsync(halo, halo + hin, hin * sizeof(float2), hipMemcpyDeviceToDevice, s));
        HIPCHK(c, hipMemcpyAsync(halo + hin, d_in, hin * sizeof(float2), hipMemcpyDeviceToDevice, s));
    }
    return DABGPU_OK;
}

int run_poly(dabgpu_ctx *c, const float2 *d_in, size_t n, float2 *d_out, hipStream_t s)
{
    const float *coef = (const float *)c->d_coef.p;
    if (c->cur.poly_is_lut)
        HIPCHK(c, launch_lut(d_in, n, c->cur.lut_scale, coef + 16, d_out, s));
    else
        HIPCHK(c, launch_poly(d_in, n, coef, coef + 8, d_out, s));
    return DABGPU_OK;
}

size_t out_samples_per_frame(const dabgpu_ctx *c, unsigned mask, size_t L, size_t M)
{
    size_t n = (mask & DABGPU_STAGE_NOGUARD) ? (size_t)(c->g.nb_symbols + 1) * (size_t)c->g.N
                                             : tf_samples(c->g);
    if (mask & DABGPU_STAGE_RESAMPLE) n = n * L / M;
    return n;
}

size_t bytes_per_sample(int fmt) { return fmt ? dabgpu_format_size(fmt) : sizeof(float2); }

// The chain on device pointers.  from_bits: d_in is coded bits, else carriers.
// The native-rate part of the chain (everything up to and including FIRFilter) for n_frames frames
// into native_out (`native` samples per frame).
// tii_seg / tii_done: the caller's cached TII segment; *tii_done says whether the frame kernel added it itself (else the caller
// runs launch_tii_add on the result)
// Tap count the frame kernel is given.  A filter of fewer than 45 taps runs as a 45-tap filter whose last taps are zero
// (out[n] = sum_j taps[j] in[n + j]: zero taps add nothing; the device table is zero padded) -- the kernels with the compile-time
// tap count, the equalised-boundary variant among them, then serve every filter up to the default length.
int fused_ntaps(const dabgpu_ctx *c)
{
    const size_t n = c->cur.taps.size();
    return (n >= 1 && n < 45) ? 45 : (int)n;
}

int run_native(dabgpu_ctx *c, const void *d_in, bool from_bits, size_t n_frames, unsigned mask, bool windowed,
               float2 *native_out, size_t native, float *gain1, hipStream_t s, bool keep_stats = true,
               unsigned long long *s16_clipped = nullptr, const float2 *tii_seg = nullptr, bool *tii_done = nullptr,
               int fused_fmt = DABGPU_FMT_S16)
{
    if (tii_done) *tii_done = false;
    TfArgs a{};
    a.clipped = s16_clipped;
#ifdef DABGPU_PHASE_TIMING
    a.phase_cycles = (unsigned long long *)c->d_phase.p;
#endif
    a.g = c->g;
    a.t = tables_of(c);
    a.gain = gain_of(c);
    a.ntaps = (int)c->cur.taps.size();
    a.n_frames = (int)n_frames;
    a.bits = from_bits ? (const uint8_t *)d_in : nullptr;
    a.carriers = from_bits ? nullptr : (const float2 *)d_in;
    a.gain1 = from_bits ? gain1 : nullptr;
    unsigned flags = from_bits ? TF_FROM_BITS : 0;
    if (mask & DABGPU_STAGE_GAIN) flags |= TF_GAIN;
    if (c->cur.cfr_enable) {
        // crest-factor reduction inside OfdmGenerator (f-3): statistics per frame, zeroed per call
        const size_t nsym = (size_t)c->g.nb_symbols + 1;
        const size_t b0 = n_frames * 2 * sizeof(unsigned), b1 = n_frames * 2 * sizeof(double),
                     b2 = n_frames * nsym * 4 * sizeof(double);
        flags |= TF_CFR;
        a.cfr_clip = c->cur.cfr_clip;
        a.cfr_errclip = c->cur.cfr_errclip;
        if (keep_stats) {
            HIPCHK(c, c->d_cfr_counts.reserve(b0));
            HIPCHK(c, c->d_cfr_mer.reserve(b1));
            HIPCHK(c, c->d_cfr_papr.reserve(b2));
            a.cfr_counts = (unsigned *)c->d_cfr_counts.p;
            a.cfr_mer = (double *)c->d_cfr_mer.p;
            a.cfr_papr = (double *)c->d_cfr_papr.p;
            a.cfr_mer_base = c->cfr_mer_index + 1;                       // src/OfdmGenerator.cpp:198
            c->cfr_last_base = a.cfr_mer_base;
            c->cfr_last_frames = n_frames;
            c->cfr_last_stream = s;
            c->cfr_mer_index = (int)((c->cfr_mer_index + n_frames) % nsym);
        } else {
            HIPCHK(c, c->d_cfr_tmp.reserve(b0 + b1 + b2 + 16));
            a.cfr_mer = (double *)c->d_cfr_tmp.p;
            a.cfr_papr = a.cfr_mer + n_frames * 2;
            a.cfr_counts = (unsigned *)(a.cfr_papr + n_frames * nsym * 4);
            a.cfr_mer_base = 0;
        }
        HIPCHK(c, hipMemsetAsync(a.cfr_counts, 0, b0, s));
        HIPCHK(c, hipMemsetAsync(a.cfr_mer, 0, b1, s));
        HIPCHK(c, hipMemsetAsync(a.cfr_papr, 0, b2, s));
    }

    a.overlap = (int)c->cur.overlap;
    if (!windowed) {
        if (!(mask & DABGPU_STAGE_NOGUARD)) flags |= TF_GUARD;
        if (mask & DABGPU_STAGE_FIR) flags |= TF_FIR;
        if (s16_clipped) flags |= tf_ofmt_flag(fused_fmt);        // (the frame kernel stores the integers itself)
        if (!(flags & TF_CFR)) a.ntaps = fused_ntaps(c);     // (the CFR variants loop over the run-time tap count)
        // cfg 3 chain: the filtered transform alone with equalised boundaries (dabgpu_set_fir_boundary_mode(ctx, 1): the packed
        // dual transform)
        if (c->use_eq && tf_has_eq(a, flags)) flags |= TF_EQ;
        a.chunks_per_frame = auto_chunks(c, n_frames);
        a.syms_per_chunk = run_symbols(c->g.nb_symbols + 1, a.chunks_per_frame, flags & TF_FIR);
        a.out = native_out;
        a.out_stride = native;
        if (tii_seg && tf_has_tii(a, flags)) {
            a.tii_seg = tii_seg;
            a.tii_insert0 = c->tii_insert ? 1 : 0;
            if (tii_done) *tii_done = true;
        }
        HIPCHK(c, launch_tf(a, flags, s));
    } else if (tf_has_window(a, flags | TF_GUARD | ((mask & DABGPU_STAGE_FIR) ? TF_FIR : 0))) {
        // OFDM windowing on the coded-bits chain, with or without FIRFilter: the frame kernel windows the guard interval
        // itself (and filters across the seams)
        flags |= TF_GUARD | TF_WINDOW | ((mask & DABGPU_STAGE_FIR) ? TF_FIR : 0);
        if (c->use_eq && (flags & TF_FIR) && !(flags & TF_CFR)) {
            // narrow overlaps on the cfg 3 chain: the equalised-boundary variant with the seam inside its boundary outputs
            // (the filter run at the default length, as without windowing)
            TfArgs e = a;
            e.ntaps = fused_ntaps(c);
            if (tf_has_eq(e, flags)) {
                a.ntaps = e.ntaps;
                flags |= TF_EQ;
                if (s16_clipped) flags |= tf_ofmt_flag(fused_fmt);    // (this form stores the integers itself)
            }
        }
        // (the default chain -- no FIRFilter -- with a windowed guard interval: its s16 store; run_chain asked tf_has_fmt)
        if (s16_clipped && !(flags & (TF_FIR | TF_CFR))) flags |= tf_ofmt_flag(fused_fmt);
        a.chunks_per_frame = auto_chunks(c, n_frames);
        a.syms_per_chunk = run_symbols(c->g.nb_symbols + 1, a.chunks_per_frame, true);
        a.out = native_out;
        a.out_stride = native;
        if (tii_seg && tf_has_tii(a, flags)) {
            a.tii_seg = tii_seg;
            a.tii_insert0 = c->tii_insert ? 1 : 0;
            if (tii_done) *tii_done = true;
        }
        HIPCHK(c, launch_tf(a, flags, s));
    } else {
        // OFDM windowing: IFFT(+gain) -> windowed guard -> FIR as separate kernels
        const size_t nsymN = (size_t)(c->g.nb_symbols + 1) * (size_t)c->g.N;
        HIPCHK(c, c->d_b.reserve(n_frames * nsymN * sizeof(float2)));
        a.chunks_per_frame = auto_chunks(c, n_frames);
        a.syms_per_chunk = run_symbols(c->g.nb_symbols + 1, a.chunks_per_frame, false);
        a.out = (float2 *)c->d_b.p;
        a.out_stride = nsymN;
        HIPCHK(c, launch_tf(a, flags, s));
        if (mask & DABGPU_STAGE_FIR)
            HIPCHK(c, launch_guard_fir((const float2 *)c->d_b.p, n_frames, c->g, (int)c->cur.overlap,
                                       (const float *)c->d_window.p, c->cur.taps.data(),
                                       (int)c->cur.taps.size(), native_out, s));
        else if (c->cur.overlap > 0)
            HIPCHK(c, launch_guard_window((const float2 *)c->d_b.p, n_frames, c->g, (int)c->cur.overlap,
                                          (const float *)c->d_window.p, native_out, s));
        else
            HIPCHK(c, launch_guard_copy((const float2 *)c->d_b.p, n_frames, c->g, native_out, s));
    }

    return DABGPU_OK;
}

// TII A_{c,p} in the reference's index convention (src/TII.cpp:247-337)
int tii_carrier_set(int mode, int comb, int pattern, std::vector<uint8_t> &acp)
{
    const int K = mode == 1 ? 1536 : 384;
    acp.assign((size_t)K, 0);
    // the 70 patterns are the 8-bit words of weight 4 in increasing order, leftmost bit = b 0 (:34-104)
    int word = 0;
    for (int w = 0, idx = 0; w < 256; ++w)
        if (__builtin_popcount((unsigned)w) == 4 && idx++ == pattern) word = w;
    auto enable = [&](int k) {
        const int ix = K / 2 + k + (k >= 0 ? -1 : 0);
        if (ix < 0 || ix + 1 >= K) return false;
        acp[(size_t)ix] = 1;
        return true;
    };
    bool ok = true;
    for (int b = 0; b < 8; ++b) {
        if (!((word >> (7 - b)) & 1)) continue;
        if (mode == 1) {
            for (int base : {-768, -384, 1, 385}) ok = enable(base + 2 * comb + 48 * b) && ok;
        } else {
            ok = enable((b < 4 ? -192 : -191) + 2 * comb + 48 * b) && ok;
        }
    }
    return ok ? DABGPU_OK : DABGPU_E_INVALID;
}

// (Re)build the stream contribution of one TII null symbol at unit gain for this stage mask:
// TII symbol -> IFFT -> guard interval (-> FIR) of a frame whose other symbols are blank.
int ensure_tii_segment(dabgpu_ctx *c, unsigned mask, bool windowed, size_t native, hipStream_t s)
{
    const unsigned key = mask & (DABGPU_STAGE_FIR | DABGPU_STAGE_NOGUARD);
    if (c->tii_seg_epoch != 0 && c->tii_seg_mask == key) return DABGPU_OK;
    const size_t K = (size_t)c->g.K, car_bytes = (size_t)(c->g.nb_symbols + 1) * K * sizeof(float2);
    std::vector<uint8_t> acp;
    if (tii_carrier_set(c->g.mode, c->cur.tii_comb, c->cur.tii_pattern, acp))
        return fail(c, DABGPU_E_INVALID, "TII::enable_carrier invalid k!");
    // d_acp / d_tii_car / d_tii_frame are shared by the lanes: batches still in flight on ANOTHER lane read the old
    // segment (in-kernel, or launch_tii_add) -- they finish before it is overwritten.  Once per TII / CFR setting or mask.
    {
        const int rc_drain = drain_lanes(c);
        if (rc_drain) return rc_drain;
    }
    HIPCHK(c, upload(c->d_acp, acp, s));
    HIPCHK(c, c->d_tii_car.reserve(car_bytes + K * sizeof(float2)));
    HIPCHK(c, c->d_tii_frame.reserve(native * sizeof(float2)));
    HIPCHK(c, hipMemsetAsync(c->d_tii_car.p, 0, car_bytes, s));
    float2 *phase = (float2 *)((char *)c->d_tii_car.p + car_bytes);
    HIPCHK(c, launch_phase_reference((const uint8_t *)c->d_phq.p, c->g.K, phase, s));
    HIPCHK(c, launch_tii(phase, (const uint8_t *)c->d_acp.p, c->g.K, c->cur.tii_old_variant ? 1 : 0, 1,
                         (float2 *)c->d_tii_car.p, s));
    // the segment is built from CARRIERS: CFR with the guard interval alone is fused from coded bits only (run_chain's
    // `windowed` is false for it), so here that combination takes the unfused IFFT + CFR -> guard kernels
    const bool seg_windowed = windowed || (c->cur.cfr_enable && !(key & (DABGPU_STAGE_FIR | DABGPU_STAGE_NOGUARD)));
    int rc = run_native(c, c->d_tii_car.p, false, 1, key, seg_windowed, (float2 *)c->d_tii_frame.p, native, nullptr, s,
                        false);
    if (rc) return rc;
    // the response of the null symbol: its own segment plus whatever a windowed guard interval spills
    // into the next one (zeros beyond; adding them is harmless)
    const size_t ext = (mask & DABGPU_STAGE_NOGUARD) ? (size_t)c->g.N
                                                     : (size_t)c->g.null_size + 2 * c->cur.overlap + 8;
    c->tii_seg_len = (int)std::min(native, ext);
    HIPCHK(c, hipStreamSynchronize(s));   // (once per setting: the segment is read by whichever lane runs the next call)
    c->tii_seg_epoch = 1;
    c->tii_seg_mask = key;
    return DABGPU_OK;
}

int run_chain(dabgpu_ctx *c, const void *d_in, bool from_bits, size_t n_frames, unsigned mask,
              void *d_out_v, size_t out_cap, size_t *out_bytes, hipStream_t s, bool apply_format = true, int lane = 0)
{
    int rc = apply_settings(c);
    if (rc) return rc;
    LaneScope scratch(c, lane);
    if (c->cur.cfr_enable) c->cfr_last_lane = lane;   // (also the OfdmGenerator stage wrapper: ITS statistics are the most recent)
    if ((mask & DABGPU_STAGE_NOGUARD) && (mask & (DABGPU_STAGE_FIR | DABGPU_STAGE_RESAMPLE | DABGPU_STAGE_POLY)))
        return fail(c, DABGPU_E_INVALID, "NOGUARD cannot be combined with FIR/RESAMPLE/POLY");
    if ((mask & DABGPU_STAGE_FIR) && c->cur.taps.empty())
        return fail(c, DABGPU_E_INVALID, "FIRFilter: no taps loaded");
    if ((mask & DABGPU_STAGE_RESAMPLE) && c->cur.rs_in == c->cur.rs_out) mask &= ~DABGPU_STAGE_RESAMPLE;
    if (mask & DABGPU_STAGE_RESAMPLE)
        if ((rc = check_resampler(c))) return rc;
    // FormatConverter as the last step of the chain (src/DabModulator.cpp:395-419): the stage-level entry points
    // that borrow the chain (OfdmGenerator, the TII segment) stay complexf
    const int fmt = apply_format ? c->cur.out_format : 0;
    const size_t per = out_samples_per_frame(c, mask, c->rs_L, c->rs_M);
    const size_t need = n_frames * per * bytes_per_sample(fmt);
    if (out_bytes) *out_bytes = need;
    if (need > out_cap) return fail(c, DABGPU_E_CAPACITY, "output buffer too small");
    if (n_frames == 0) return DABGPU_OK;

    const size_t native = (mask & DABGPU_STAGE_NOGUARD) ? per : tf_samples(c->g);
    const bool post = mask & (DABGPU_STAGE_RESAMPLE | DABGPU_STAGE_POLY);
    const bool fir_fits = (int)c->cur.taps.size() - 1 <= c->g.sym_size - c->g.N &&
                          (int)c->cur.taps.size() <= tf_max_fused_taps();
    // one fused kernel, unless the guard interval is windowed or the filter does not fit it
    // (then: IFFT[+CFR][+gain] -> guard kernel -> FIR kernel)
    // (CFR has fused variants with the whole epilogue -- guard + FIR --, with none of it, and, from coded bits, with the
    // guard interval alone)
    const bool windowed = (c->cur.overlap > 0 || ((mask & DABGPU_STAGE_FIR) && !fir_fits) ||
                           (c->cur.cfr_enable && !(mask & DABGPU_STAGE_FIR) && !from_bits)) &&
                          !(mask & DABGPU_STAGE_NOGUARD);
    if (windowed && c->cur.overlap > 0) {
        const size_t W = c->cur.overlap;
        if (W > (size_t)(c->g.sym_size - c->g.N))
            return fail(c, DABGPU_E_INVALID, "window overlap larger than the guard interval");
    }
    const bool tii = from_bits && c->cur.tii_enable;

    // s16 leaves the LAST kernel of the chain directly where that kernel has a variant for it: the frame kernel
    // (Mode I coded-bits chain with guard interval and the default-length filter) or the x2 / x4 resampler (with
    // the polynomial predistorter inside, or none).  Every other combination, and u8 / s8, converts afterwards.
    unsigned long long *clip = nullptr;
    bool fuse_native = false, fuse_post = false;
    if (apply_format) c->clip_valid = fmt != 0;       // (a complexf call leaves no count behind: never the previous call's)
    if (fmt) {
        HIPCHK(c, c->d_clip.reserve(16));
        HIPCHK(c, hipMemsetAsync(c->d_clip.p, 0, 16, s));
        clip = (unsigned long long *)c->d_clip.p;
        c->clip_stream = s;
        c->clip_lane = lane;
        if (from_bits) {
            // ask the kernels' own predicates (the ones their launchers test), so that the separate convert kernel is taken
            // whenever a variant does not exist in this build
            const bool poly_ok = !(mask & DABGPU_STAGE_POLY) || (!c->cur.poly_is_lut && (mask & DABGPU_STAGE_RESAMPLE));
            TfArgs ta{};
            ta.g = c->g;
            ta.t = tables_of(c);
            ta.gain = gain_of(c);
            ta.ntaps = c->cur.cfr_enable ? (int)c->cur.taps.size() : fused_ntaps(c);
            ta.chunks_per_frame = auto_chunks(c, n_frames);
            ta.syms_per_chunk = run_symbols(c->g.nb_symbols + 1, ta.chunks_per_frame, mask & DABGPU_STAGE_FIR);
            unsigned tflags = TF_FROM_BITS | ((mask & DABGPU_STAGE_GAIN) ? TF_GAIN : 0) |
                              ((mask & DABGPU_STAGE_NOGUARD) ? 0 : TF_GUARD) | ((mask & DABGPU_STAGE_FIR) ? TF_FIR : 0) |
                              (c->cur.cfr_enable ? TF_CFR : 0);
            if (c->use_eq && tf_has_eq(ta, tflags)) tflags |= TF_EQ;
            // (a windowed guard interval has variants without the integer store only, and TII is added to the native-rate
            // complexf stream afterwards unless the frame kernel adds it itself: the frame kernel's own store is out then, the
            // resampler's is not.  u8 / s8: the frame kernel's equalised-boundary and no-FIRFilter variants; s16: those, the
            // pruned dual transform and the x2 / x4 resampler.)
            fuse_native = !post && !windowed && (!tii || tf_has_tii(ta, tflags)) && tf_has_fmt(ta, tflags | tf_ofmt_flag(fmt));
            if (!post && windowed && c->cur.overlap > 0 && !(tflags & TF_CFR)) {
                // ... except where a windowed form has the store (the decisions run_native takes): narrow overlaps on the cfg 3
                // chain (the equalised-boundary form, every format, TII inside), the chain without FIRFilter (s16, no TII)
                const unsigned wflags = tflags | TF_WINDOW;
                ta.overlap = (int)c->cur.overlap;
                ta.ntaps = (int)c->cur.taps.size();
                if (tf_has_window(ta, wflags)) {
                    if (tflags & TF_FIR) {
                        ta.ntaps = fused_ntaps(c);
                        fuse_native = c->use_eq && tf_has_eq(ta, wflags) && tf_has_fmt(ta, wflags | TF_EQ | tf_ofmt_flag(fmt)) &&
                                      (!tii || tf_has_tii(ta, wflags | TF_EQ));
                    } else {
                        fuse_native = !tii && tf_has_fmt(ta, wflags | tf_ofmt_flag(fmt));
                    }
                }
            }
            ResamplerArgs ra{};
            ra.nin = c->rs_nin;
            ra.nout = c->rs_nout;
            fuse_post = fmt == DABGPU_FMT_S16 && (mask & DABGPU_STAGE_RESAMPLE) && resampler_fast_ratio(c) &&
                        resampler_has_s16(ra) && poly_ok;
        }
    }
    float2 *d_out = (float2 *)d_out_v;
    if (fmt && !fuse_native && !fuse_post) {
        HIPCHK(c, c->d_fmt.reserve(n_frames * per * sizeof(float2)));
        d_out = (float2 *)c->d_fmt.p;
    }

    // The hand-over FIRFilter -> Resampler in cache-sized pieces (dabgpu_set_handover_frames): x2 / x4 with the predistorter
    // inside the resampler's store or absent; CFR (per-frame statistics) and TII (per-frame gain, frame parity) keep the
    // one-piece path.
    const bool fuse_poly = (mask & DABGPU_STAGE_POLY) && !c->cur.poly_is_lut && resampler_fast_ratio(c);
    const size_t piece = (size_t)c->handover_frames & ~(size_t)1;
    if ((mask & DABGPU_STAGE_RESAMPLE) && resampler_fast_ratio(c) && (fuse_poly || !(mask & DABGPU_STAGE_POLY)) &&
        !c->cur.cfr_enable && !tii && piece >= 2 && n_frames > piece) {
        HIPCHK(c, c->d_a.reserve(2 * piece * native * sizeof(float2)));
        hipStream_t prod;
        if ((rc = lane_stream(c, 1, &prod))) return rc;
        if (!c->ho_start) {
            HIPCHK(c, hipEventCreateWithFlags(&c->ho_start, hipEventDisableTiming));
            for (int i = 0; i < 2; ++i) {
                HIPCHK(c, hipEventCreateWithFlags(&c->ho_prod[i], hipEventDisableTiming));
                HIPCHK(c, hipEventCreateWithFlags(&c->ho_cons[i], hipEventDisableTiming));
            }
        }
        // the producer starts after whatever the caller queued on s (the input; the previous call's use of the ring)
        HIPCHK(c, hipEventRecord(c->ho_start, s));
        HIPCHK(c, hipStreamWaitEvent(prod, c->ho_start, 0));
        const size_t in_per = from_bits ? tf_in_bytes(c->g) : (size_t)(c->g.nb_symbols + 1) * (size_t)c->g.K * sizeof(float2);
        const size_t bps = bytes_per_sample((fuse_post && fmt) ? fmt : 0);
        size_t f0 = 0;
        for (int i = 0; f0 < n_frames; ++i, f0 += piece) {
            const size_t nf = std::min(piece, n_frames - f0);
            const int slot = i & 1;
            float2 *ring = (float2 *)c->d_a.p + (size_t)slot * piece * native;
            if (i >= 2) HIPCHK(c, hipStreamWaitEvent(prod, c->ho_cons[slot], 0));   // the consumer is done with piece i - 2
            if ((rc = run_native(c, (const char *)d_in + f0 * in_per, from_bits, nf, mask, windowed, ring, native, nullptr,
                                 prod)))
                return rc;
            HIPCHK(c, hipEventRecord(c->ho_prod[slot], prod));
            HIPCHK(c, hipStreamWaitEvent(s, c->ho_prod[slot], 0));
            if ((rc = run_resampler(c, ring, nf * native, (float2 *)((char *)d_out + f0 * per * bps), s, fuse_poly,
                                    fuse_post ? clip : nullptr)))
                return rc;
            HIPCHK(c, hipEventRecord(c->ho_cons[slot], s));
        }
        if (from_bits && (n_frames & 1)) c->tii_insert = !c->tii_insert;   // (src/TII.cpp:241-242: toggles with TII off as well)
        if (fmt && !fuse_native && !fuse_post)
            HIPCHK(c, launch_format((const float *)d_out, 2 * n_frames * per, fmt, d_out_v, clip, s));
        return DABGPU_OK;
    }

    // where the native-rate stream goes
    float2 *native_out = d_out;
    if (post) {
        HIPCHK(c, c->d_a.reserve(n_frames * native * sizeof(float2)));
        native_out = (float2 *)c->d_a.p;
    }

    float *gain1 = nullptr;
    if (tii) {
        if ((rc = ensure_tii_segment(c, mask, windowed, native, s))) return rc;
        if (mask & DABGPU_STAGE_GAIN) {
            HIPCHK(c, c->d_gain1.reserve(n_frames * sizeof(float)));
            gain1 = (float *)c->d_gain1.p;
        }
    }
    bool tii_done = false;
    if ((rc = run_native(c, d_in, from_bits, n_frames, mask, windowed, native_out, native, gain1, s, true,
                         fuse_native ? clip : nullptr, tii ? (const float2 *)c->d_tii_frame.p : nullptr, &tii_done, fmt)))
        return rc;
    if (tii && !tii_done) {
        if (fuse_native) return fail(c, DABGPU_E_DEVICE, "s16 stored by the frame kernel, TII still to be added");
        HIPCHK(c, launch_tii_add(native_out, native, (const float2 *)c->d_tii_frame.p, c->tii_seg_len, gain1,
                                 c->tii_insert ? 1 : 0, n_frames, s));
    }
    // the insert flag toggles once per frame of the stream whether or not TII is enabled (src/TII.cpp:241-242)
    if (from_bits && (n_frames & 1)) c->tii_insert = !c->tii_insert;

    if (post) {
        const float2 *cur = native_out;
        size_t n = n_frames * native;
        bool poly_done = false;
        if (mask & DABGPU_STAGE_RESAMPLE) {
            // the polynomial predistorter is an epilogue of the resampler's store (LUT mode is not)
            const bool fuse = (mask & DABGPU_STAGE_POLY) && !c->cur.poly_is_lut && resampler_fast_ratio(c);
            float2 *dst = d_out;
            if ((mask & DABGPU_STAGE_POLY) && !fuse) {
                HIPCHK(c, c->d_b.reserve(n_frames * per * sizeof(float2)));
                dst = (float2 *)c->d_b.p;
            }
            rc = run_resampler(c, cur, n, dst, s, fuse, fuse_post ? clip : nullptr);
            if (rc) return rc;
            cur = dst;
            n = n_frames * per;
            poly_done = fuse;
        }
        if ((mask & DABGPU_STAGE_POLY) && !poly_done) {
            rc = run_poly(c, cur, n, d_out, s);
            if (rc) return rc;
        }
    }
    if (fmt && !fuse_native && !fuse_post)
        HIPCHK(c, launch_format((const float *)d_out, 2 * n_frames * per, fmt, d_out_v, clip, s));
    return DABGPU_OK;
}

// host-pointer stage wrapper: H2D, launch, D2H on the context stream
struct HostIO {
    dabgpu_ctx *c;
    explicit HostIO(dabgpu_ctx *ctx) : c(ctx) {}
    int in(DevBuf &b, const void *h, size_t n)
    {
        HIPCHK(c, b.reserve(std::max<size_t>(n, 16)));
        if (n) HIPCHK(c, hipMemcpyAsync(b.p, h, n, hipMemcpyHostToDevice, c->stream));
        return DABGPU_OK;
    }
    int out(void *h, const void *d, size_t n)
    {
        if (n) HIPCHK(c, hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return DABGPU_OK;
    }
};

int check_out(dabgpu_ctx *c, size_t need, size_t cap, size_t *out_bytes)
{
    if (out_bytes) *out_bytes = need;
    if (need > cap) return fail(c, DABGPU_E_CAPACITY, "output buffer too small");
    return DABGPU_OK;
}

#define CTXCHK(c)                                                                              \
    do {                                                                                       \
        if (!(c)) return DABGPU_E_INVALID;                                                     \
        hipError_t e_ = hipSetDevice((c)->device);                                             \
        if (e_ != hipSuccess) return hip_fail((c), e_, "hipSetDevice");                        \
    } while (0)

}  // namespace

// ===========================================================================
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
                      &c->d_acp, &c->d_tii_car, &c->d_tii_frame, &c->d_gain1, &c->d_cic,
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
        for (DevBuf *b : {&l.d_a, &l.d_b, &l.d_fmt, &l.d_clip, &l.d_gain1, &l.d_cfr_counts, &l.d_cfr_mer, &l.d_cfr_papr, &l.d_cfr_tmp})
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

int dabgpu_get_cfr_stats(dabgpu_ctx *c, size_t frame, dabgpu_cfr_stats *out)
{
    CTXCHK(c);
    if (!out) return fail(c, DABGPU_E_INVALID, "null argument");
    if (frame >= c->cfr_last_frames)
        return fail(c, DABGPU_E_INVALID, "no CFR statistics for this frame (CFR off, or frame index out of range)");
    const size_t nsym = (size_t)c->g.nb_symbols + 1;
    HIPCHK(c, hipStreamSynchronize(c->cfr_last_stream ? c->cfr_last_stream : c->stream));
    LaneScope scratch(c, c->cfr_last_lane);
    unsigned counts[2];
    double mer[2];
    std::vector<double> papr(nsym * 4);
    HIPCHK(c, hipMemcpy(counts, (const unsigned *)c->d_cfr_counts.p + 2 * frame, sizeof counts, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(mer, (const double *)c->d_cfr_mer.p + 2 * frame, sizeof mer, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(papr.data(), (const double *)c->d_cfr_papr.p + frame * nsym * 4, nsym * 4 * sizeof(double),
                        hipMemcpyDeviceToHost));
    std::memset(out, 0, sizeof *out);
    out->num_clip = counts[0];
    out->num_error_clip = counts[1];
    out->num_samples = nsym * (size_t)c->g.N;
    out->mer_symbol = (int)(((size_t)c->cfr_last_base + frame) % nsym);
    out->mer_sum_iq = mer[0];
    out->mer_sum_delta = mer[1];
    out->nb_symbols = (int)nsym;
    for (size_t s = 0; s < nsym; ++s) {
        out->papr_before[s][0] = papr[4 * s];
        out->papr_before[s][1] = papr[4 * s + 1];
        out->papr_after[s][0] = papr[4 * s + 2];
        out->papr_after[s][1] = papr[4 * s + 3];
    }
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

// ---- per-stage, host buffers ------------------------------------------------

int dabgpu_qpsk_process(dabgpu_ctx *c, const void *in, size_t in_bytes, void *out, size_t out_cap,
                        size_t *out_bytes)
{
    CTXCHK(c);
    if (in_bytes % (size_t)(c->g.K / 4) != 0)
        return fail(c, DABGPU_E_INVALID, "QpskSymbolMapper::process input size not valid!");
    const size_t need = in_bytes * 4 * sizeof(float2);
    int rc = check_out(c, need, out_cap, out_bytes);
    if (rc) return rc;
    HostIO io(c);
    if ((rc = io.in(c->d_a, in, in_bytes))) return rc;
    HIPCHK(c, c->d_b.reserve(std::max<size_t>(need, 16)));
    HIPCHK(c, launch_qpsk((const uint8_t *)c->d_a.p, in_bytes, c->g.K, (float2 *)c->d_b.p, c->stream));
    return io.out(out, c->d_b.p, need);
}

int dabgpu_freq_interleave_process(dabgpu_ctx *c, const void *in, size_t in_bytes, void *out,
                                   size_t out_cap, size_t *out_bytes)
{
    CTXCHK(c);
    const size_t ns = in_bytes / sizeof(float2);
    if (in_bytes % sizeof(float2) || ns % (size_t)c->g.K != 0)
        return fail(c, DABGPU_E_INVALID, "FrequencyInterleaver::process input size not valid!");
    int rc = check_out(c, in_bytes, out_cap, out_bytes);
    if (rc) return rc;
    HostIO io(c);
    if ((rc = io.in(c->d_a, in, in_bytes))) return rc;
    HIPCHK(c, c->d_b.reserve(std::max<size_t>(in_bytes, 16)));
    HIPCHK(c, launch_freq_interleave((const float2 *)c->d_a.p, ns, c->g.K,
                                     (const uint16_t *)c->d_src.p, (float2 *)c->d_b.p, c->stream));
    return io.out(out, c->d_b.p, in_bytes);
}

int dabgpu_phase_reference_process(dabgpu_ctx *c, void *out, size_t out_cap, size_t *out_bytes)
{
    CTXCHK(c);
    const size_t need = (size_t)c->g.K * sizeof(float2);
    int rc = check_out(c, need, out_cap, out_bytes);
    if (rc) return rc;
    HostIO io(c);
    HIPCHK(c, c->d_b.reserve(need));
    HIPCHK(c, launch_phase_reference((const uint8_t *)c->d_phq.p, c->g.K, (float2 *)c->d_b.p, c->stream));
    return io.out(out, c->d_b.p, need);
}

int dabgpu_diff_mod_process(dabgpu_ctx *c, const void *phase, size_t phase_bytes, const void *data,
                            size_t data_bytes, void *out, size_t out_cap, size_t *out_bytes)
{
    CTXCHK(c);
    const size_t K = (size_t)c->g.K;
    if (phase_bytes != K * sizeof(float2))
        return fail(c, DABGPU_E_INVALID, "DifferentialModulator::process input phase size not valid!");
    if (data_bytes % (K * sizeof(float2)) != 0)
        return fail(c, DABGPU_E_INVALID, "DifferentialModulator::process input data size not valid!");
    const size_t need = phase_bytes + data_bytes;
    int rc = check_out(c, need, out_cap, out_bytes);
    if (rc) return rc;
    HostIO io(c);
    if ((rc = io.in(c->d_a, phase, phase_bytes))) return rc;
    if ((rc = io.in(c->d_c, data, data_bytes))) return rc;
    HIPCHK(c, c->d_b.reserve(need));
    HIPCHK(c, launch_diff_mod((const float2 *)c->d_a.p, (const float2 *)c->d_c.p,
                              data_bytes / (K * sizeof(float2)), c->g.K, (float2 *)c->d_b.p, c->stream));
    return io.out(out, c->d_b.p, need);
}

int dabgpu_null_symbol_process(dabgpu_ctx *c, void *out, size_t out_cap, size_t *out_bytes)
{
    CTXCHK(c);
    const size_t need = (size_t)c->g.K * sizeof(float2);
    int rc = check_out(c, need, out_cap, out_bytes);
    if (rc) return rc;
    HostIO io(c);
    HIPCHK(c, c->d_b.reserve(need));
    HIPCHK(c, hipMemsetAsync(c->d_b.p, 0, need, c->stream));
    return io.out(out, c->d_b.p, need);
}

int dabgpu_cic_equalizer_process(dabgpu_ctx *c, size_t spacing, int R, const void *in, size_t in_bytes, void *out,
                                 size_t out_cap, size_t *out_bytes)
{
    CTXCHK(c);
    const size_t K = (size_t)c->g.K;
    if (in_bytes % (K * sizeof(float2))) return fail(c, DABGPU_E_INVALID, "CicEqualizer::process input size not valid!");
    if (!spacing || R <= 0) return fail(c, DABGPU_E_INVALID, "CicEqualizer: spacing and R must be positive");
    int rc = check_out(c, in_bytes, out_cap, out_bytes);
    if (rc) return rc;
    if (c->cic_spacing != spacing || c->cic_R != R) {
        // the reference's constructor, src/CicEqualizer.cpp:38-55, in float with the libm float functions
        std::vector<float> filter(K);
        const int M = 1, N = 4;
        const float pi = 4.0f * atanf(1.0f);
        for (size_t i = 0; i < K; ++i) {
            const int k = i < (K + 1) / 2 ? (int)i + (int)((K & 1) ^ 1) : (int)i - (int)K;
            const float angle = pi * k / spacing;
            if (k == 0) {
                filter[i] = 1.0f;
            } else {
                float f = sinf(angle / R) / sinf(angle * M);
                f = fabsf(f) * R * M;
                filter[i] = powf(f, N);
            }
        }
        HIPCHK(c, upload(c->d_cic, filter, c->stream));
        c->cic_spacing = spacing;
        c->cic_R = R;
    }
    HostIO io(c);
    if ((rc = io.in(c->d_a, in, in_bytes))) return rc;
    HIPCHK(c, c->d_b.reserve(std::max<size_t>(in_bytes, 16)));
    HIPCHK(c, launch_cic((const float2 *)c->d_a.p, in_bytes / sizeof(float2), c->g.K, (const float *)c->d_cic.p,
                         (float2 *)c->d_b.p, c->stream));
    return io.out(out, c->d_b.p, in_bytes);
}

int dabgpu_tii_process(dabgpu_ctx *c, const void *in, size_t in_bytes, void *out, size_t out_cap,
                       size_t *out_bytes)
{
    CTXCHK(c);
    int rc = apply_settings(c);
    if (rc) return rc;
    const size_t need = (size_t)c->g.K * sizeof(float2);
    if (c->g.mode != 1 && c->g.mode != 2)
        return fail(c, DABGPU_E_INVALID, "TII::TII DAB mode " + std::to_string(c->g.mode) + " not valid!");
    if (!in || in_bytes != need) return fail(c, DABGPU_E_INVALID, "TII::process input size not valid!");
    if ((rc = check_out(c, need, out_cap, out_bytes))) return rc;
    std::vector<uint8_t> acp;
    if (tii_carrier_set(c->g.mode, c->cur.tii_comb, c->cur.tii_pattern, acp))
        return fail(c, DABGPU_E_INVALID, "TII::enable_carrier invalid k!");
    HostIO io(c);
    if ((rc = io.in(c->d_a, in, in_bytes))) return rc;
    HIPCHK(c, upload(c->d_acp, acp, c->stream));
    c->tii_seg_epoch = 0;                      // d_acp is shared with the chain's cached segment
    HIPCHK(c, c->d_b.reserve(need));
    const int insert = (c->cur.tii_enable && c->tii_insert) ? 1 : 0;       // src/TII.cpp:226
    HIPCHK(c, launch_tii((const float2 *)c->d_a.p, (const uint8_t *)c->d_acp.p, c->g.K,
                         c->cur.tii_old_variant ? 1 : 0, insert, (float2 *)c->d_b.p, c->stream));
    c->tii_insert = !c->tii_insert;                                        // :241-242
    return io.out(out, c->d_b.p, need);
}

int dabgpu_signal_mux_process(dabgpu_ctx *c, const void *first, size_t first_bytes, const void *rest,
                              size_t rest_bytes, void *out, size_t out_cap, size_t *out_bytes)
{
    CTXCHK(c);
    const size_t need = first_bytes + rest_bytes;
    int rc = check_out(c, need, out_cap, out_bytes);
    if (rc) return rc;
    HIPCHK(c, c->d_b.reserve(std::max<size_t>(need, 16)));
    if (first_bytes) HIPCHK(c, hipMemcpyAsync(c->d_b.p, first, first_bytes, hipMemcpyHostToDevice, c->stream));
    if (rest_bytes)
        HIPCHK(c, hipMemcpyAsync((char *)c->d_b.p + first_bytes, rest, rest_bytes, hipMemcpyHostToDevice, c->stream));
    HostIO io(c);
    return io.out(out, c->d_b.p, need);
}

int dabgpu_ofdm_process(dabgpu_ctx *c, const void *in, size_t in_bytes, void *out, size_t out_cap,
                        size_t *out_bytes)
{
    CTXCHK(c);
    const size_t per_in = (size_t)(c->g.nb_symbols + 1) * (size_t)c->g.K * sizeof(float2);
    if (in_bytes != per_in)
        return fail(c, DABGPU_E_INVALID, "OfdmGenerator::process input size not valid!");
    HostIO io(c);
    int rc = io.in(c->d_c, in, in_bytes);
    if (rc) return rc;
    const size_t need = (size_t)(c->g.nb_symbols + 1) * (size_t)c->g.N * sizeof(float2);
    if ((rc = check_out(c, need, out_cap, out_bytes))) return rc;
    HIPCHK(c, c->d_b.reserve(need));
    size_t ob = 0;
    rc = run_chain(c, c->d_c.p, false, 1, DABGPU_STAGE_NOGUARD, (float2 *)c->d_b.p, need, &ob, c->stream, false);
    if (rc) return rc;
    return io.out(out, c->d_b.p, need);
}

int dabgpu_gain_process(dabgpu_ctx *c, const void *in, size_t in_bytes, void *out, size_t out_cap,
                        size_t *out_bytes)
{
    CTXCHK(c);
    int rc = apply_settings(c);
    if (rc) return rc;
    const size_t ns = in_bytes / sizeof(float2);
    if (in_bytes % sizeof(float2) || ns % (size_t)c->g.N != 0)
        return fail(c, DABGPU_E_INVALID, "GainControl::process input size not valid!");
    if ((rc = check_out(c, in_bytes, out_cap, out_bytes))) return rc;
    HostIO io(c);
    if ((rc = io.in(c->d_a, in, in_bytes))) return rc;
    HIPCHK(c, c->d_b.reserve(std::max<size_t>(in_bytes, 16)));
    HIPCHK(c, launch_gain((const float2 *)c->d_a.p, ns / (size_t)c->g.N, c->g.N, gain_of(c),
                          (float2 *)c->d_b.p, c->stream));
    return io.out(out, c->d_b.p, in_bytes);
}

int dabgpu_guard_process(dabgpu_ctx *c, const void *in, size_t in_bytes, void *out, size_t out_cap,
                         size_t *out_bytes)
{
    CTXCHK(c);
    int rc = apply_settings(c);
    if (rc) return rc;
    const size_t per_in = (size_t)(c->g.nb_symbols + 1) * (size_t)c->g.N * sizeof(float2);
    if (in_bytes != per_in)
        return fail(c, DABGPU_E_INVALID, "GuardIntervalInserter::process input size not valid!");
    const size_t need = tf_samples(c->g) * sizeof(float2);
    if ((rc = check_out(c, need, out_cap, out_bytes))) return rc;
    if (c->cur.overlap > (size_t)(c->g.sym_size - c->g.N))
        return fail(c, DABGPU_E_INVALID, "window overlap larger than the guard interval");
    HostIO io(c);
    if ((rc = io.in(c->d_a, in, in_bytes))) return rc;
    HIPCHK(c, c->d_b.reserve(need));
    if (c->cur.overlap == 0)
        HIPCHK(c, launch_guard_copy((const float2 *)c->d_a.p, 1, c->g, (float2 *)c->d_b.p, c->stream));
    else
        HIPCHK(c, launch_guard_window((const float2 *)c->d_a.p, 1, c->g, (int)c->cur.overlap,
                                      (const float *)c->d_window.p, (float2 *)c->d_b.p, c->stream));
    return io.out(out, c->d_b.p, need);
}

int dabgpu_fir_process(dabgpu_ctx *c, const void *in, size_t in_bytes, void *out, size_t out_cap,
                       size_t *out_bytes)
{
    CTXCHK(c);
    int rc = apply_settings(c);
    if (rc) return rc;
    if (in_bytes % sizeof(float2)) return fail(c, DABGPU_E_INVALID, "FIRFilter: input size not valid");
    if ((rc = check_out(c, in_bytes, out_cap, out_bytes))) return rc;
    HostIO io(c);
    if ((rc = io.in(c->d_a, in, in_bytes))) return rc;
    HIPCHK(c, c->d_b.reserve(std::max<size_t>(in_bytes, 16)));
    HIPCHK(c, launch_fir((const float2 *)c->d_a.p, in_bytes / sizeof(float2), 1,
                         c->cur.taps.data(), (int)c->cur.taps.size(), (float2 *)c->d_b.p, c->stream));
    return io.out(out, c->d_b.p, in_bytes);
}

int dabgpu_resampler_process(dabgpu_ctx *c, const void *in, size_t in_bytes, void *out, size_t out_cap,
                             size_t *out_bytes)
{
    CTXCHK(c);
    int rc = apply_settings(c);
    if (rc) return rc;
    if (in_bytes % sizeof(float2)) return fail(c, DABGPU_E_INVALID, "Resampler: input size not valid");
    const size_t ns = in_bytes / sizeof(float2);
    const size_t need = ns * c->rs_L / c->rs_M * sizeof(float2);
    if ((rc = check_out(c, need, out_cap, out_bytes))) return rc;
    HostIO io(c);
    if ((rc = io.in(c->d_a, in, in_bytes))) return rc;
    HIPCHK(c, c->d_b.reserve(std::max<size_t>(need, 16)));
    if (c->cur.rs_in == c->cur.rs_out) {
        HIPCHK(c, hipMemcpyAsync(c->d_b.p, c->d_a.p, in_bytes, hipMemcpyDeviceToDevice, c->stream));
    } else {
        if ((rc = run_resampler(c, (const float2 *)c->d_a.p, ns, (float2 *)c->d_b.p, c->stream))) return rc;
    }
    return io.out(out, c->d_b.p, need);
}

int dabgpu_poly_process(dabgpu_ctx *c, const void *in, size_t in_bytes, void *out, size_t out_cap,
                        size_t *out_bytes)
{
    CTXCHK(c);
    int rc = apply_settings(c);
    if (rc) return rc;
    if (in_bytes % (2 * sizeof(float2)))
        return fail(c, DABGPU_E_INVALID, "MemlessPoly: input size not valid");
    if ((rc = check_out(c, in_bytes, out_cap, out_bytes))) return rc;
    HostIO io(c);
    if ((rc = io.in(c->d_a, in, in_bytes))) return rc;
    HIPCHK(c, c->d_b.reserve(std::max<size_t>(in_bytes, 16)));
    if ((rc = run_poly(c, (const float2 *)c->d_a.p, in_bytes / sizeof(float2), (float2 *)c->d_b.p, c->stream)))
        return rc;
    return io.out(out, c->d_b.p, in_bytes);
}

// ---- f-2 FormatConverter -------------------------------------------------------

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

int dabgpu_format_process_dev(dabgpu_ctx *c, const void *d_in, size_t n_floats, int format, void *d_out,
                              size_t out_cap, size_t *out_bytes, unsigned long long *d_num_clipped,
                              void *stream)
{
    CTXCHK(c);
    const size_t elem = dabgpu_format_size(format) / 2;
    if (!elem) return fail(c, DABGPU_E_INVALID, "FormatConverter: Invalid format");
    int rc = check_out(c, n_floats * elem, out_cap, out_bytes);
    if (rc) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    if (!stream && (rc = own_stream_joins_lanes(c))) return rc;     // (its input may be a chain call's output on any lane)
    if (!d_num_clipped) {
        HIPCHK(c, c->d_count.reserve(16));
        d_num_clipped = (unsigned long long *)c->d_count.p + 1;      // scratch slot, never read
    }
    HIPCHK(c, launch_format((const float *)d_in, n_floats, format, d_out, d_num_clipped, s));
    return DABGPU_OK;
}

int dabgpu_format_process(dabgpu_ctx *c, const void *in, size_t in_bytes, int format, void *out,
                          size_t out_cap, size_t *out_bytes, size_t *num_clipped)
{
    CTXCHK(c);
    const size_t elem = dabgpu_format_size(format) / 2;
    if (!elem) return fail(c, DABGPU_E_INVALID, "FormatConverter: Invalid format");
    const size_t n = in_bytes / sizeof(float);                       // src/FormatConverter.cpp:112
    int rc = check_out(c, n * elem, out_cap, out_bytes);
    if (rc) return rc;
    HostIO io(c);
    if ((rc = io.in(c->d_a, in, n * sizeof(float)))) return rc;
    HIPCHK(c, c->d_b.reserve(std::max<size_t>(n * elem, 16)));
    HIPCHK(c, c->d_count.reserve(16));
    HIPCHK(c, hipMemsetAsync(c->d_count.p, 0, 16, c->stream));
    HIPCHK(c, launch_format((const float *)c->d_a.p, n, format, c->d_b.p, (unsigned long long *)c->d_count.p,
                            c->stream));
    unsigned long long cnt = 0;
    HIPCHK(c, hipMemcpyAsync(&cnt, c->d_count.p, sizeof(cnt), hipMemcpyDeviceToHost, c->stream));
    rc = io.out(out, c->d_b.p, n * elem);
    if (num_clipped) *num_clipped = (size_t)cnt;
    return rc;
}

// ---- chain -------------------------------------------------------------------

size_t dabgpu_chain_out_bytes_per_frame(const dabgpu_ctx *c, unsigned mask)
{
    if (!c) return 0;
    size_t L = 1, M = 1;
    {
        std::lock_guard<std::mutex> lk(const_cast<dabgpu_ctx *>(c)->mu);
        size_t a = c->set.rs_in, b = c->set.rs_out;
        while (b) { size_t t = a % b; a = b; b = t; }
        L = c->set.rs_out / a; M = c->set.rs_in / a;
    }
    int fmt;
    {
        std::lock_guard<std::mutex> lk(const_cast<dabgpu_ctx *>(c)->mu);
        fmt = c->set.out_format;
    }
    return out_samples_per_frame(c, mask, L, M) * bytes_per_sample(fmt);
}

}  // extern "C"

namespace {

// Which lane a call on the context's own stream goes to: the lanes in turn while a launch alone cannot fill the chip many
// times over; lane 0 for everything that carries stream state (Resampler) and for large batches (nothing to gain, and
// the scratch of some chains grows with the batch).
int pick_lane(dabgpu_ctx *c, size_t n_frames, unsigned mask, bool *rotating)
{
    *rotating = false;
    bool resample;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        resample = (mask & DABGPU_STAGE_RESAMPLE) && c->set.rs_in != c->set.rs_out;
    }
    if (c->n_lanes <= 1 || resample || n_frames > (size_t)dabgpu_ctx::kLaneMaxFrames) return 0;
    // Every lane owns a set of per-call scratch buffers that grow to the largest call they have seen and are never trimmed.
    // The one-kernel chains need none; the others (a separate guard / FIRFilter / convert / predistorter kernel: up to two
    // native-rate frames of 8 B per sample per frame) rotate only while that stays within kLaneScratchBytes per lane --
    // the small batches the lanes exist for.
    bool scratch;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        scratch = c->set.overlap > 0 || c->set.out_format != 0 || c->set.cfr_enable || c->set.tii_enable ||
                  (mask & DABGPU_STAGE_POLY) || (int)c->set.taps.size() > tf_max_fused_taps();
    }
    if (scratch && n_frames * 2 * tf_samples(c->g) * sizeof(float2) > (size_t)dabgpu_ctx::kLaneScratchBytes) return 0;
    *rotating = true;
    return (int)(c->lane_seq++ % (unsigned long long)c->n_lanes);
}

int chain_dev(dabgpu_ctx *c, const void *d_in, bool from_bits, size_t n_frames, unsigned mask, void *d_iq, size_t out_cap,
              size_t *out_bytes, void *stream)
{
    int lane = 0;
    hipStream_t s = (hipStream_t)stream;
    c->call_lanes = 1;
    if (!s) {
        bool rotating = false;
        lane = pick_lane(c, n_frames, mask, &rotating);
        int rc = lane_stream(c, lane, &s);
        if (rc) return rc;
        if ((rc = lane_joins_own_stream(c, lane))) return rc;
        if (rotating) c->call_lanes = c->n_lanes;
    }
    c->clip_from_collect = false;
    TraceScope trace(c->trace_enabled ? &c->last_variant : nullptr);
    const int rc = run_chain(c, d_in, from_bits, n_frames, mask, (float2 *)d_iq, out_cap, out_bytes, s, true, lane);
    c->call_lanes = 1;
    return rc;
}

}  // namespace

extern "C" {

int dabgpu_chain_process_dev(dabgpu_ctx *c, const void *d_bits, size_t n_frames, unsigned mask,
                             void *d_iq, size_t out_cap, size_t *out_bytes, void *stream)
{
    CTXCHK(c);
    return chain_dev(c, d_bits, true, n_frames, mask, d_iq, out_cap, out_bytes, stream);
}

int dabgpu_symbols_process_dev(dabgpu_ctx *c, const void *d_car, size_t n_frames, unsigned mask,
                               void *d_iq, size_t out_cap, size_t *out_bytes, void *stream)
{
    CTXCHK(c);
    return chain_dev(c, d_car, false, n_frames, mask, d_iq, out_cap, out_bytes, stream);
}

// cifRes -> cifPoly on a native-rate stream that is already in device memory: the tail of the chain by itself
int dabgpu_post_process_dev(dabgpu_ctx *c, const void *d_native, size_t n_samples, unsigned mask, void *d_iq, size_t out_cap,
                            size_t *out_bytes, void *stream)
{
    CTXCHK(c);
    int rc = apply_settings(c);
    if (rc) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    if (!stream && (rc = own_stream_joins_lanes(c))) return rc;     // (d_native: a chain call's output on any lane)
    if (mask & ~(unsigned)(DABGPU_STAGE_RESAMPLE | DABGPU_STAGE_POLY))
        return fail(c, DABGPU_E_INVALID, "post-processing: DABGPU_STAGE_RESAMPLE and / or DABGPU_STAGE_POLY");
    if ((mask & DABGPU_STAGE_RESAMPLE) && c->cur.rs_in == c->cur.rs_out) mask &= ~(unsigned)DABGPU_STAGE_RESAMPLE;
    size_t n_out = n_samples;
    if (mask & DABGPU_STAGE_RESAMPLE) {
        if ((rc = check_resampler(c))) return rc;
        if (n_samples % ((size_t)c->rs_nin / 2)) return fail(c, DABGPU_E_INVALID, "Resampler::process input size not valid!");
        n_out = n_samples * c->rs_L / c->rs_M;
    }
    if ((rc = check_out(c, n_out * sizeof(float2), out_cap, out_bytes))) return rc;
    if (n_samples == 0) return DABGPU_OK;
    TraceScope trace(c->trace_enabled ? &c->last_variant : nullptr);
    const float2 *cur = (const float2 *)d_native;
    bool poly_done = !(mask & DABGPU_STAGE_POLY);
    if (mask & DABGPU_STAGE_RESAMPLE) {
        const bool fuse = (mask & DABGPU_STAGE_POLY) && !c->cur.poly_is_lut && resampler_fast_ratio(c);
        float2 *dst = (float2 *)d_iq;
        if (!poly_done && !fuse) {
            HIPCHK(c, c->d_b.reserve(n_out * sizeof(float2)));
            dst = (float2 *)c->d_b.p;
        }
        if ((rc = run_resampler(c, cur, n_samples, dst, s, fuse))) return rc;
        cur = dst;
        poly_done = poly_done || fuse;
    }
    if (!poly_done && (rc = run_poly(c, cur, n_out, (float2 *)d_iq, s))) return rc;
    if (cur == (const float2 *)d_native && poly_done)     // (neither stage: the stream passes through)
        HIPCHK(c, hipMemcpyAsync(d_iq, d_native, n_samples * sizeof(float2), hipMemcpyDeviceToDevice, s));
    return DABGPU_OK;
}

int dabgpu_set_lanes(dabgpu_ctx *c, int lanes)
{
    CTXCHK(c);
    if (lanes < 1 || lanes > (int)dabgpu_ctx::kMaxLanes) return fail(c, DABGPU_E_INVALID, "lanes: 1 ... 4");
    const int rc = dabgpu_synchronize(c);
    if (rc) return rc;
    c->n_lanes = lanes;
    c->lane_seq = 0;
    return DABGPU_OK;
}

int dabgpu_debug_lanes(dabgpu_ctx *c, int *own_queue_mask)
{
    CTXCHK(c);
    int n = 1, mask = 1;
    for (int i = 1; i < (int)dabgpu_ctx::kMaxLanes; ++i)
        if (c->lane[i].stream) { ++n; if (c->lane_own_queue[i]) mask |= 1 << i; }
    if (own_queue_mask) *own_queue_mask = mask;
    return n;
}

int dabgpu_set_handover_frames(dabgpu_ctx *c, int frames)
{
    CTXCHK(c);
    if (frames < 0 || (frames & 1)) return fail(c, DABGPU_E_INVALID, "hand-over piece: an even number of frames, or 0");
    const int rc = dabgpu_synchronize(c);
    if (rc) return rc;
    c->handover_frames = frames;
    return DABGPU_OK;
}

// everything the context queues from now on starts after what `stream` holds now
int dabgpu_wait_for_stream(dabgpu_ctx *c, void *stream)
{
    CTXCHK(c);
    if (!c->lane[0].ev) HIPCHK(c, hipEventCreateWithFlags(&c->lane[0].ev, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->lane[0].ev, (hipStream_t)stream));
    for (int i = 0; i < (int)dabgpu_ctx::kMaxLanes; ++i) {
        hipStream_t ls = i ? c->lane[i].stream : c->stream;
        if (ls) HIPCHK(c, hipStreamWaitEvent(ls, c->lane[0].ev, 0));
    }
    // (a lane whose stream does not exist yet is created later, by a call the host makes after this one: it cannot start
    // before the host has seen `stream` reach this point only if the caller relies on stream order alone -- so create them)
    for (int i = 1; i < c->n_lanes; ++i)
        if (!c->lane[i].stream) {
            hipStream_t ls;
            const int rc = lane_stream(c, i, &ls);
            if (rc) return rc;
            HIPCHK(c, hipStreamWaitEvent(ls, c->lane[0].ev, 0));
        }
    return DABGPU_OK;
}

// everything queued on `stream` from now on starts after what the context has queued so far (all lanes)
int dabgpu_stream_wait_for(dabgpu_ctx *c, void *stream)
{
    CTXCHK(c);
    for (int i = 0; i < (int)dabgpu_ctx::kMaxLanes; ++i) {
        hipStream_t ls = i ? c->lane[i].stream : c->stream;
        if (!ls) continue;
        // (lane 0's `ev` is dabgpu_wait_for_stream's: the joins use events 1 ... and one more for lane 0)
        hipEvent_t &ev = i ? c->lane[i].ev : c->ho_join;
        if (!ev) HIPCHK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(ev, ls));
        HIPCHK(c, hipStreamWaitEvent((hipStream_t)stream, ev, 0));
    }
    return DABGPU_OK;
}

int dabgpu_chain_process(dabgpu_ctx *c, const uint8_t *bits, size_t n_frames, unsigned mask,
                         void *iq_out, size_t out_cap, size_t *out_bytes)
{
    CTXCHK(c);
    if (n_frames > (size_t)c->max_frames)
        return fail(c, DABGPU_E_CAPACITY, "n_frames exceeds max_frames of the context");
    c->clip_from_collect = false;
    int rc = apply_settings(c);
    if (rc) return rc;
    unsigned m2 = mask;
    if ((m2 & DABGPU_STAGE_RESAMPLE) && c->cur.rs_in == c->cur.rs_out) m2 &= ~DABGPU_STAGE_RESAMPLE;
    const size_t need = n_frames * out_samples_per_frame(c, m2, c->rs_L, c->rs_M) * bytes_per_sample(c->cur.out_format);
    if ((rc = check_out(c, need, out_cap, out_bytes))) return rc;
    HostIO io(c);
    if ((rc = io.in(c->d_in, bits, n_frames * tf_in_bytes(c->g)))) return rc;
    // final output lives in its own buffer: d_a / d_b / d_c are the chain's scratch
    HIPCHK(c, c->d_out.reserve(std::max<size_t>(need, 16)));
    size_t ob = 0;
    {
        TraceScope trace(c->trace_enabled ? &c->last_variant : nullptr);
        rc = run_chain(c, c->d_in.p, true, n_frames, mask, (float2 *)c->d_out.p, need, &ob, c->stream);
    }
    if (rc) return rc;
    return io.out(iq_out, c->d_out.p, need);
}

// ---- asynchronous host path ------------------------------------------------------

int dabgpu_chain_submit(dabgpu_ctx *c, const uint8_t *bits, size_t n_frames, unsigned mask)
{
    CTXCHK(c);
    if (c->slot_count == 2) return fail(c, DABGPU_E_CAPACITY, "two batches are already in flight: collect one first");
    if (n_frames > (size_t)c->max_frames)
        return fail(c, DABGPU_E_CAPACITY, "n_frames exceeds max_frames of the context");
    int rc = apply_settings(c);
    if (rc) return rc;
    unsigned m2 = mask;
    if ((m2 & DABGPU_STAGE_RESAMPLE) && c->cur.rs_in == c->cur.rs_out) m2 &= ~DABGPU_STAGE_RESAMPLE;
    const size_t in_bytes = n_frames * tf_in_bytes(c->g);
    const size_t need = n_frames * out_samples_per_frame(c, m2, c->rs_L, c->rs_M) * bytes_per_sample(c->cur.out_format);
    const int slot_index = (c->slot_head + c->slot_count) & 1;
    dabgpu_ctx::Slot &sl = c->slot[slot_index];
    // the two batches in flight run on two lanes where the chain carries no stream state: their kernels overlap
    const int lane = (c->n_lanes > 1 && !(m2 & DABGPU_STAGE_RESAMPLE)) ? slot_index : 0;
    hipStream_t ls;
    if ((rc = lane_stream(c, lane, &ls))) return rc;
    if ((rc = lane_joins_own_stream(c, lane))) return rc;
    if (!c->copy_stream) {
        // the copy back must overlap with the kernels of BOTH batches in flight: apart from lanes 0 and 1
        hipStream_t l1 = nullptr;
        bool own = false;
        if (c->n_lanes > 1 && (rc = lane_stream(c, 1, &l1))) return rc;
        std::vector<hipStream_t> others{c->stream};
        if (l1) others.push_back(l1);
        HIPCHK(c, create_stream_apart(others.data(), (int)others.size(), &c->copy_stream, &own));
    }
    if (!sl.computed) {
        HIPCHK(c, hipEventCreateWithFlags(&sl.computed, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming));
    }
    if (sl.h_in_cap < in_bytes) {
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        sl.h_in = nullptr;
        sl.h_in_cap = 0;
        HIPCHK(c, hipHostMalloc(&sl.h_in, std::max<size_t>(in_bytes, 16), hipHostMallocDefault));
        sl.h_in_cap = in_bytes;
    }
    const int ho = (int)(c->submit_seq % 3);
    if (c->h_out_cap[ho] < need) {
        if (c->h_out[ho]) (void)hipHostFree(c->h_out[ho]);
        c->h_out[ho] = nullptr;
        c->h_out_cap[ho] = 0;
        HIPCHK(c, hipHostMalloc(&c->h_out[ho], std::max<size_t>(need, 16), hipHostMallocDefault));
        c->h_out_cap[ho] = need;
    }
    HIPCHK(c, sl.d_in.reserve(std::max<size_t>(in_bytes, 16)));
    HIPCHK(c, sl.d_out.reserve(std::max<size_t>(need, 16)));
    std::memcpy(sl.h_in, bits, in_bytes);                       // 28.8 kB per frame
    if (in_bytes) HIPCHK(c, hipMemcpyAsync(sl.d_in.p, sl.h_in, in_bytes, hipMemcpyHostToDevice, ls));
    size_t ob = 0;
    {
        TraceScope trace(c->trace_enabled ? &c->last_variant : nullptr);
        c->call_lanes = (c->n_lanes > 1 && !(m2 & DABGPU_STAGE_RESAMPLE)) ? 2 : 1;     // (the two batches in flight)
        rc = run_chain(c, sl.d_in.p, true, n_frames, mask, (float2 *)sl.d_out.p, need, &ob, ls, true, lane);
        c->call_lanes = 1;
    }
    if (rc) return rc;
    sl.out_format = c->cur.out_format;
    if (sl.out_format) {
        // the clip counter is one per lane and the next submit on it zeroes it: this batch's count goes to the slot now
        if (!sl.h_clip) HIPCHK(c, hipHostMalloc((void **)&sl.h_clip, 16, hipHostMallocDefault));
        LaneScope scratch(c, lane);
        HIPCHK(c, hipMemcpyAsync(sl.h_clip, c->d_clip.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, ls));
    }
    HIPCHK(c, hipEventRecord(sl.computed, ls));
    // the copy back runs on its own stream: the next batch's kernels overlap it
    HIPCHK(c, hipStreamWaitEvent(c->copy_stream, sl.computed, 0));
    if (need) HIPCHK(c, hipMemcpyAsync(c->h_out[ho], sl.d_out.p, need, hipMemcpyDeviceToHost, c->copy_stream));
    HIPCHK(c, hipEventRecord(sl.copied, c->copy_stream));
    sl.out_bytes = need;
    sl.h_out_index = ho;
    sl.stream = ls;
    sl.busy = true;
    ++c->slot_count;
    ++c->submit_seq;
    return DABGPU_OK;
}

int dabgpu_chain_collect(dabgpu_ctx *c, const void **iq, size_t *out_bytes)
{
    CTXCHK(c);
    if (!iq) return fail(c, DABGPU_E_INVALID, "null argument");
    if (c->slot_count == 0) return fail(c, DABGPU_E_INVALID, "no batch in flight");
    dabgpu_ctx::Slot &sl = c->slot[c->slot_head];
    HIPCHK(c, hipEventSynchronize(sl.copied));
    // Nothing ever synchronises the lanes or the copy stream themselves (the events do the ordering), and the HIP runtime
    // retires the commands of a stream -- their signals, kernel-argument blocks, command objects -- only when somebody asks
    // about that stream: after ~800 one-frame batches it stopped for 48 ms inside one call to catch up
    // (profiles/r06_async_series.txt).  Everything this batch queued is complete here; a query is the asking.
    (void)hipStreamQuery(sl.stream);
    (void)hipStreamQuery(c->copy_stream);
    *iq = c->h_out[sl.h_out_index];
    if (out_bytes) *out_bytes = sl.out_bytes;
    c->clip_from_collect = true;
    c->collected_clipped = (sl.out_format && sl.h_clip) ? (size_t)*sl.h_clip : 0;     // (the format of ITS submit)
    sl.busy = false;
    c->slot_head ^= 1;
    --c->slot_count;
    return DABGPU_OK;
}

int dabgpu_synchronize(dabgpu_ctx *c)
{
    CTXCHK(c);
    const int rc = drain_lanes(c);
    if (rc) return rc;
    for (bool &d : c->lane_dirty) d = false;
    return DABGPU_OK;
}

#ifdef DABGPU_PHASE_TIMING
// tool builds only (tools/phase_timing.py): the frame kernel's per-phase shader-cycle sums since the last call, 16 words
// (Phase order of device_common.h; word 15 = wave-iterations behind the sums); zeroes them
DABGPU_API int dabgpu_debug_phase_cycles(dabgpu_ctx *c, unsigned long long *out16)
{
    CTXCHK(c);
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(out16, c->d_phase.p, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemset(c->d_phase.p, 0, 16 * sizeof(unsigned long long)));
    return DABGPU_OK;
}
#endif

}  // extern "C"
