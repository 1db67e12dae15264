// dabgpu_ctx.h -- the device context behind the C-ABI of include/dabgpu.h and the helpers its translation units share:
//   api_context.hip   context, settings, device tables (apply_settings), setters, diagnostics
//   api_chain.hip     the chain dispatch: which kernels a stage mask runs (run_native / run_chain), chain entry points
//   api_lanes.hip     batches in flight inside one context: lanes, their ordering, submit / collect
//   api_stages.hip    one host-buffer entry point per reference plugin, FormatConverter, CFR statistics
#pragma once
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

namespace dabgpu_api {
using namespace dabgpu;

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
    bool gain_reference_rounding = false;  // dabgpu_set_gain_rounding: chain calls replay the reference's variance recurrence
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


// names the kernels of one chain call into ctx->last_variant (the sink is a thread-local of api_context.hip)
std::string *&trace_sink_ref();
struct TraceScope {
    std::string *prev;
    // (sink == nullptr: tracing is off for this context -- nothing is installed, a launch costs one pointer test)
    explicit TraceScope(std::string *sink) : prev(trace_sink_ref())
    {
        if (sink) sink->clear();
        trace_sink_ref() = sink;
    }
    ~TraceScope() { trace_sink_ref() = prev; }
};
}  // namespace dabgpu_api

struct dabgpu_ctx {
    std::string last_variant;             // dabgpu_debug_last_variant: the kernels the most recent chain call launched
    bool trace_enabled = false;           // dabgpu_debug_trace: off by default (names are formatted per launch when on)
    dabgpu::Geometry g{};
    int device = 0;
    int max_frames = 1;
    int chunks_cfg = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // constant tables
    dabgpu_api::DevBuf d_twiddle, d_src, d_dst, d_phq, d_mag, d_taps, d_firh, d_window, d_coef, d_eqg;
    bool use_eq = true;                   // dabgpu_set_fir_boundary_mode: false = always the packed dual transform
    bool eq_ok = false;                   // d_eqg holds a well-conditioned inverse of the current taps (TF_EQ may be used)
    double eq_fit = 0.0;                  // max |G H - 1| over the occupied bins
    // resampler
    dabgpu_api::DevBuf d_rs_window, d_rs_tw_in, d_rs_tw_out, d_rs_halo, d_rs_tw_s, d_rs_tw_l;
    int rs_nin = 0, rs_nout = 0;
    int rs_halo_cur = 0;                  // which of the two halo buffers holds the state the next call reads
    size_t rs_L = 1, rs_M = 1;
    float rs_factor = 1.f;
    // scratch
    dabgpu_api::DevBuf d_a, d_b, d_c, d_in, d_out, d_count, d_fmt, d_clip;
    dabgpu_api::DevBuf d_phase;                        // tool builds only (-DDABGPU_PHASE_TIMING): the frame kernel's per-phase cycle counters
    hipStream_t clip_stream = nullptr;     // stream of the most recent chain call that converted its output
    // TII (f-4): carrier set, the one-frame carrier image and its native-rate response, gain of symbol 1
    dabgpu_api::DevBuf d_acp, d_tii_car, d_tii_frame, d_gain1, d_cic;
    dabgpu_api::DevBuf d_gains;           // gain rounding REFERENCE: the multipliers of a call's symbols
    size_t cic_spacing = 0;               // what d_cic was built for (CicEqualizer, a12)
    int cic_R = 0;
    // CFR statistics (f-3) of the most recent chain / OfdmGenerator call, and a scratch set for internal runs
    dabgpu_api::DevBuf d_cfr_counts, d_cfr_mer, d_cfr_papr, d_cfr_tmp;
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
        dabgpu_api::DevBuf d_a, d_b, d_fmt, d_clip, d_gain1, d_gains, d_cfr_counts, d_cfr_mer, d_cfr_papr, d_cfr_tmp;
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
    dabgpu_api::Settings set;                    // guarded by mu
    dabgpu_api::Settings cur;                    // snapshot used by the processing thread
    unsigned long long applied_epoch = 0;


    // asynchronous host path (dabgpu_chain_submit / dabgpu_chain_collect): two batches in flight,
    // pinned staging on both sides, device->host copies on their own stream
    struct Slot {
        void *h_in = nullptr;                          // pinned (hipHostMalloc)
        size_t h_in_cap = 0, out_bytes = 0;
        int h_out_index = 0;                           // which of the three pinned output buffers this batch lands in
        unsigned long long *h_clip = nullptr;          // pinned: this batch's clipped-component count (output formats)
        dabgpu_api::DevBuf d_in, d_out;
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

namespace dabgpu_api {

int fail(dabgpu_ctx *c, int code, const std::string &msg);
int hip_fail(dabgpu_ctx *c, hipError_t e, const char *what);

#define HIPCHK(ctx, expr)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #expr);                                 \
    } while (0)


// ---- api_lanes.hip
int lane_stream(dabgpu_ctx *c, int i, hipStream_t *out);          // the stream of lane i (created on first use)
int own_stream_joins_lanes(dabgpu_ctx *c);                         // a NULL-stream call on the context's own stream: behind every lane
int lane_joins_own_stream(dabgpu_ctx *c, int i);                   // a chain call that goes to lane i: behind such work
int drain_lanes(dabgpu_ctx *c);                                    // every stream of the context idle
int chain_dev(dabgpu_ctx *c, const void *d_in, bool from_bits, size_t n_frames, unsigned mask, void *d_iq, size_t out_cap,
              size_t *out_bytes, void *stream);

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
        std::swap(c->d_gain1, l.d_gain1); std::swap(c->d_gains, l.d_gains); std::swap(c->d_cfr_counts, l.d_cfr_counts); std::swap(c->d_cfr_mer, l.d_cfr_mer);
        std::swap(c->d_cfr_papr, l.d_cfr_papr); std::swap(c->d_cfr_tmp, l.d_cfr_tmp);
    }
};


// ---- api_context.hip
extern const float kDefaultTaps[45];
bool mode_geometry(int mode, Geometry *g);
size_t tf_in_bytes(const Geometry &g);
size_t tf_samples(const Geometry &g);
template <typename T> hipError_t upload(DevBuf &b, const std::vector<T> &v, hipStream_t s)
{
    hipError_t e = b.reserve(std::max<size_t>(v.size() * sizeof(T), 16));
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(s);  // v may be a temporary
}

bool design_inverse_filter(const std::vector<float> &taps, int N, int K, std::vector<float> &g_out, double *fit_out);
int apply_settings(dabgpu_ctx *c);
Tables tables_of(dabgpu_ctx *c);
GainParams gain_of(const dabgpu_ctx *c);

// ---- api_chain.hip
int auto_chunks(const dabgpu_ctx *c, size_t n_frames);
int run_symbols(int nsym, int chunks, bool lookahead);
bool is_pow2(size_t x);
bool resampler_fast_ratio(const dabgpu_ctx *c);
const char *resampler_ratio_error(int N, size_t in_rate, size_t out_rate);
int check_resampler(dabgpu_ctx *c);
int run_resampler(dabgpu_ctx *c, const float2 *d_in, size_t total, float2 *d_out, hipStream_t s, bool fuse_poly = false,
                  unsigned long long *s16_clipped = nullptr);
int run_poly(dabgpu_ctx *c, const float2 *d_in, size_t n, float2 *d_out, hipStream_t s);
size_t out_samples_per_frame(const dabgpu_ctx *c, unsigned mask, size_t L, size_t M);
size_t bytes_per_sample(int fmt);
int fused_ntaps(const dabgpu_ctx *c);
int tii_carrier_set(int mode, int comb, int pattern, std::vector<uint8_t> &acp);
int run_native(dabgpu_ctx *c, const void *d_in, bool from_bits, size_t n_frames, unsigned mask, bool windowed,
               float2 *native_out, size_t native, float *gain1, hipStream_t s, bool keep_stats = true,
               unsigned long long *s16_clipped = nullptr, const float2 *tii_seg = nullptr, bool *tii_done = nullptr,
               int fused_fmt = DABGPU_FMT_S16);
int run_chain(dabgpu_ctx *c, const void *d_in, bool from_bits, size_t n_frames, unsigned mask, void *d_out_v, size_t out_cap,
              size_t *out_bytes, hipStream_t s, bool apply_format = true, int lane = 0);

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

inline int check_out(dabgpu_ctx *c, size_t need, size_t cap, size_t *out_bytes)
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

}  // namespace dabgpu_api
