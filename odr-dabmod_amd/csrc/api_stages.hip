// api_stages.hip -- one host-buffer entry point per reference plugin (the per-stage drop-ins call these), FormatConverter,
// and the CFR statistics of the most recent call.
#include "dabgpu_ctx.h"

using namespace dabgpu;
using namespace dabgpu_api;

extern "C" {
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

}  // extern "C"
