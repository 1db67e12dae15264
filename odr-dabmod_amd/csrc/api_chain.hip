// api_chain.hip -- the chain dispatch: which kernels a stage mask runs for the context's settings (run_native: the frame kernel's
// variant or the unfused sequence; run_chain: + TII, Resampler, MemlessPoly, FormatConverter), and the chain's entry points.
#include "dabgpu_ctx.h"

using namespace dabgpu;
using namespace dabgpu_api;

namespace dabgpu_api {
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
    // with three lanes (tools/experiments/exp_r05.py chunks, profiles/r05_exp_chunks.jsonl) 26 runs per frame at 16 frames (416 workgroups;
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
                  bool fuse_poly, unsigned long long *s16_clipped)
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

// gain mode var by the reference's recurrence (dabgpu_set_gain_rounding): the chain call is split at GainControl
bool gain_replay(const dabgpu_ctx *c, unsigned mask)
{
    return c->cur.gain_reference_rounding && c->cur.gain_mode == DABGPU_GAIN_VAR && (mask & DABGPU_STAGE_GAIN);
}

int run_native(dabgpu_ctx *c, const void *d_in, bool from_bits, size_t n_frames, unsigned mask, bool windowed,
               float2 *native_out, size_t native, float *gain1, hipStream_t s, bool keep_stats,
               unsigned long long *s16_clipped, const float2 *tii_seg, bool *tii_done, int fused_fmt)
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
    const bool replay = gain_replay(c, mask);
    if (replay) {
        // OfdmGenerator (+ CFR) alone, then the reference's recurrence on the unscaled symbols (src/GainControl.cpp:118-155), then
        // guard interval / FIRFilter as kernels of their own, which scale the symbols as they read them
        flags &= ~(unsigned)TF_GAIN;
        a.gain1 = nullptr;
        const int nsym = c->g.nb_symbols + 1;
        const size_t nsymN = (size_t)nsym * (size_t)c->g.N;
        const bool noguard = mask & DABGPU_STAGE_NOGUARD;
        float2 *x0 = native_out;
        if (!noguard) {
            HIPCHK(c, c->d_b.reserve(n_frames * nsymN * sizeof(float2)));
            x0 = (float2 *)c->d_b.p;
        }
        a.chunks_per_frame = auto_chunks(c, n_frames);
        a.syms_per_chunk = run_symbols(nsym, a.chunks_per_frame, false);
        a.out = x0;
        a.out_stride = noguard ? native : nsymN;
        if (noguard && native != nsymN) return fail(c, DABGPU_E_DEVICE, "gain rounding: unexpected frame stride");
        HIPCHK(c, launch_tf(a, flags, s));
        HIPCHK(c, c->d_gains.reserve(n_frames * (size_t)nsym * sizeof(float)));
        // (the multipliers are applied by the guard kernel as it gathers the symbols; a chain that stops here scales in place)
        const float *gains = (const float *)c->d_gains.p;
        HIPCHK(c, launch_gain_replay(x0, n_frames, nsym, c->g.N, a.gain, (float *)c->d_gains.p, from_bits ? gain1 : nullptr,
                                     noguard, s));
        if (noguard) return DABGPU_OK;
        if (mask & DABGPU_STAGE_FIR)
            HIPCHK(c, launch_guard_fir(x0, n_frames, c->g, (int)c->cur.overlap, (const float *)c->d_window.p,
                                       c->cur.taps.data(), (int)c->cur.taps.size(), native_out, s, gains));
        else if (c->cur.overlap > 0)
            HIPCHK(c, launch_guard_window(x0, n_frames, c->g, (int)c->cur.overlap, (const float *)c->d_window.p, native_out, s,
                                          gains));
        else
            HIPCHK(c, launch_guard_copy(x0, n_frames, c->g, native_out, s, gains));
        return DABGPU_OK;
    }
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
              void *d_out_v, size_t out_cap, size_t *out_bytes, hipStream_t s, bool apply_format, int lane)
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
    if (gain_replay(c, mask)) fuse_native = false;     // (the frame kernel is not the chain's last kernel then)
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

}  // namespace dabgpu_api

extern "C" {
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
