/*
 * ref_harness.cpp -- C-ABI shim around the REFERENCE's own stage classes.
 *
 * TEST INFRASTRUCTURE.  This file is ours; it #includes the reference headers
 * from where they lie under /root/reference and is linked against objects
 * compiled from the reference's own .cpp files (oracle/Makefile target `ref`,
 * everything lands in oracle/_ref/, which is git-ignored).  No reference
 * source is copied into this repository.  It exists so that tests can check
 * the plain-C oracle (dab_oracle.c) against the real implementation and so
 * that tests/golden/make_golden.py can generate the golden fixtures.
 *
 * Not covered (unbuildable in this image): OfdmGenerator.cpp and
 * Resampler.cpp include <fftw3.h>, which is not installed.
 */
#include "QpskSymbolMapper.h"
#include "FrequencyInterleaver.h"
#include "PhaseReference.h"
#include "DifferentialModulator.h"
#include "NullSymbol.h"
#include "SignalMultiplexer.h"
#include "GainControl.h"
#include "GuardIntervalInserter.h"
#include "FIRFilter.h"
#include "MemlessPoly.h"
#include "FormatConverter.h"
#include "TII.h"
#include "PAPRStats.h"

#include <cstring>
#include <string>
#include <vector>

namespace {

struct GainProbe : public GainControl {
    using GainControl::GainControl;
    using GainControl::internal_process;
};

struct FirProbe : public FIRFilter {
    using FIRFilter::FIRFilter;
    using FIRFilter::internal_process;
};

void fill(Buffer &b, const void *p, size_t bytes) { b.setData(p, bytes); }

int copy_out(const Buffer &b, void *out, size_t expect_bytes)
{
    if (b.getLength() != expect_bytes) return -2;
    memcpy(out, b.getData(), expect_bytes);
    return 0;
}

} // namespace

extern "C" {

int ref_qpsk(const uint8_t *in, size_t nbytes, int carriers, float *out)
{
    try {
        QpskSymbolMapper st((size_t)carriers, false);
        Buffer bi, bo;
        fill(bi, in, nbytes);
        st.process(&bi, &bo);
        return copy_out(bo, out, nbytes * 4 * sizeof(complexf));
    } catch (const std::exception &) { return -1; }
}

int ref_freq_interleave(const float *in, size_t nsamples, int mode, float *out)
{
    try {
        FrequencyInterleaver st((size_t)mode, false);
        Buffer bi, bo;
        fill(bi, in, nsamples * sizeof(complexf));
        st.process(&bi, &bo);
        return copy_out(bo, out, nsamples * sizeof(complexf));
    } catch (const std::exception &) { return -1; }
}

int ref_phase_reference(int mode, int carriers, float *out)
{
    try {
        PhaseReference st((unsigned)mode, false);
        Buffer bo;
        st.process(&bo);
        return copy_out(bo, out, (size_t)carriers * sizeof(complexf));
    } catch (const std::exception &) { return -1; }
}

int ref_diff_mod(const float *phase, const float *data, size_t ndata, int carriers, float *out)
{
    try {
        DifferentialModulator st((size_t)carriers, false);
        Buffer bp, bd, bo;
        fill(bp, phase, (size_t)carriers * sizeof(complexf));
        fill(bd, data, ndata * sizeof(complexf));
        st.process(std::vector<Buffer *>{&bp, &bd}, &bo);
        return copy_out(bo, out, ((size_t)carriers + ndata) * sizeof(complexf));
    } catch (const std::exception &) { return -1; }
}

/* NullSymbol ++ data through SignalMultiplexer (2-input form) */
int ref_null_mux(const float *rest, size_t nrest, int carriers, float *out)
{
    try {
        NullSymbol ns((size_t)carriers, sizeof(complexf));
        SignalMultiplexer mux;
        Buffer bn, br, bo;
        ns.process(&bn);
        fill(br, rest, nrest * sizeof(complexf));
        mux.process(std::vector<Buffer *>{&bn, &br}, &bo);
        return copy_out(bo, out, ((size_t)carriers + nrest) * sizeof(complexf));
    } catch (const std::exception &) { return -1; }
}

int ref_gain_control(const float *in, size_t nsamples, int framesize, int gain_mode,
                     float dig_gain, float normalise, float var_variance, float *out)
{
    try {
        GainMode gm = (GainMode)gain_mode;
        float dg = dig_gain, vv = var_variance;
        GainProbe st((size_t)framesize, gm, dg, normalise, vv);
        Buffer bi, bo;
        fill(bi, in, nsamples * sizeof(complexf));
        st.internal_process(&bi, &bo);
        return copy_out(bo, out, nsamples * sizeof(complexf));
    } catch (const std::exception &) { return -1; }
}

int ref_guard_interval(const float *in, int nb_symbols, int spacing, int null_size,
                       int sym_size, int overlap, float *out)
{
    try {
        size_t ov = (size_t)overlap;
        GuardIntervalInserter st((size_t)nb_symbols, (size_t)spacing, (size_t)null_size,
                                 (size_t)sym_size, ov, FFTEngine::FFTW);
        Buffer bi, bo;
        fill(bi, in, (size_t)(nb_symbols + 1) * (size_t)spacing * sizeof(complexf));
        st.process(&bi, &bo);
        return copy_out(bo, out,
                ((size_t)null_size + (size_t)nb_symbols * (size_t)sym_size) * sizeof(complexf));
    } catch (const std::exception &) { return -1; }
}

/* taps_file: path, or "default" for the built-in taps */
int ref_fir_filter(const float *in, size_t nsamples, const char *taps_file, float *out)
{
    try {
        std::string tf(taps_file);
        FirProbe st(tf);
        Buffer bi, bo;
        fill(bi, in, nsamples * sizeof(complexf));
        bo.setLength(bi.getLength()); /* the pipeline worker does this: src/ModPlugin.cpp:143-144 */
        st.internal_process(&bi, &bo);
        return copy_out(bo, out, nsamples * sizeof(complexf));
    } catch (const std::exception &) { return -1; }
}

/* MemlessPoly::internal_process is private: go through the pipelined
 * process() twice; the second call returns the first frame's result. */
int ref_memless_poly(const float *in, size_t nsamples, const char *coef_file,
                     unsigned num_threads, float *out)
{
    try {
        std::string cf(coef_file);
        MemlessPoly st(cf, num_threads);
        Buffer b1, b2, bo;
        fill(b1, in, nsamples * sizeof(complexf));
        fill(b2, in, nsamples * sizeof(complexf));
        st.process(&b1, &bo);
        st.process(&b2, &bo);
        return copy_out(bo, out, nsamples * sizeof(complexf));
    } catch (const std::exception &) { return -1; }
}

// f-3: PAPRStats fed nblocks blocks of blocklen samples; returns calculate_papr() (0 when fewer
// than `accumulate` blocks have been seen)
double ref_papr(const float *x, size_t nblocks, size_t blocklen, size_t accumulate)
{
    PAPRStats st(accumulate);
    for (size_t b = 0; b < nblocks; ++b)
        st.process_block(reinterpret_cast<const complexf *>(x) + b * blocklen, blocklen);
    return st.calculate_papr();
}

// f-4: TII fed by a PhaseReference, called ncalls times (the insert flag toggles per call).
// out: ncalls x carriers complexf.  -1 on exception (TIIError for invalid mode/comb/pattern).
int ref_tii(int mode, int enable, int comb, int pattern, int old_variant, int ncalls, float *out)
{
    try {
        tii_config_t conf;
        conf.enable = enable != 0;
        conf.comb = comb;
        conf.pattern = pattern;
        conf.old_variant = old_variant != 0;
        TII tii((unsigned)mode, conf, false);
        PhaseReference ref((unsigned)mode, false);
        Buffer bp, bo;
        ref.process(&bp);
        const size_t n = bp.getLength();
        for (int i = 0; i < ncalls; ++i) {
            tii.process(&bp, &bo);
            if (bo.getLength() != n) return -2;
            memcpy(reinterpret_cast<char *>(out) + (size_t)i * n, bo.getData(), n);
        }
        return 0;
    } catch (const std::exception &) { return -1; }
}

// f-2: FormatConverter (float input).  Returns bytes written, -1 on exception; *clipped = its counter.
int ref_format_convert(const float *in, size_t nfloats, const char *fmt, void *out, size_t out_cap,
                       size_t *clipped)
{
    try {
        FormatConverter st(false, std::string(fmt));
        Buffer bi, bo;
        fill(bi, in, nfloats * sizeof(float));
        st.process(&bi, &bo);
        if (clipped) *clipped = st.get_num_clipped_samples();
        if (bo.getLength() > out_cap) return -2;
        memcpy(out, bo.getData(), bo.getLength());
        return (int)bo.getLength();
    } catch (const std::exception &) { return -1; }
}

} // extern "C"
