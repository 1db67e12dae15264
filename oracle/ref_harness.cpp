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
#include "CicEqualizer.h"
#include "EtiReader.h"
#include "FicSource.h"
#include "SubchannelSource.h"
#include "PrbsGenerator.h"
#include "ConvEncoder.h"
#include "PuncturingEncoder.h"
#include "TimeInterleaver.h"
#include "FrameMultiplexer.h"
#include "BlockPartitioner.h"
#include <memory>

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

// f-1: the ETI -> coded-bits front-end built from the reference's own classes and run in the order
// DabModulator wires them (src/DabModulator.cpp:131-139,281-385); frames before the first FP == 0 are
// parsed but not modulated (src/DabMod.cpp:684-693).  out receives one BlockPartitioner block per
// completed transmission frame; returns the number of blocks, or -1 on exception.
int ref_eti_frontend(const uint8_t *eti, size_t nframes, unsigned mode, uint8_t *out, size_t out_cap)
{
    try {
        double tist_offset = 0.0;
        EtiReader reader(tist_offset);
        std::shared_ptr<FicSource> fic;
        std::unique_ptr<PrbsGenerator> cifPrbs, ficPrbs;
        std::unique_ptr<ConvEncoder> ficConv;
        std::unique_ptr<PuncturingEncoder> ficPunc;
        std::unique_ptr<FrameMultiplexer> cifMux;
        std::unique_ptr<BlockPartitioner> cifPart;
        struct Sub {
            std::shared_ptr<SubchannelSource> src;
            std::unique_ptr<PrbsGenerator> prbs;
            std::unique_ptr<ConvEncoder> conv;
            std::unique_ptr<PuncturingEncoder> punc;
            std::unique_ptr<TimeInterleaver> ti;
            Buffer b0, b1, b2, b3, b4;
        };
        std::vector<std::unique_ptr<Sub>> subs;
        bool started = false;
        int nblocks = 0;
        size_t pos = 0;
        Buffer part;      // BlockPartitioner fills ONE output buffer over the frames of a transmission frame
        for (size_t f = 0; f < nframes; ++f) {
            Buffer frame(6144, eti + f * 6144);
            reader.loadEtiData(frame);
            if (!started) {
                if (reader.getFp() != 0) continue;
                started = true;
                cifPrbs.reset(new PrbsGenerator(864 * 8, 0x110));
                cifMux.reset(new FrameMultiplexer(reader));
                cifPart.reset(new BlockPartitioner(mode));
                fic = reader.getFic();
                const size_t n = fic->getFramesize();
                ficPrbs.reset(new PrbsGenerator(n, 0x110));
                ficConv.reset(new ConvEncoder(n));
                ficPunc.reset(new PuncturingEncoder());
                for (const auto &r : fic->get_rules()) ficPunc->append_rule(r);
                ficPunc->append_tail_rule(PuncturingRule(3, 0xcccccc));
                for (const auto &sc : reader.getSubchannels()) {
                    std::unique_ptr<Sub> s(new Sub);
                    s->src = sc;
                    s->prbs.reset(new PrbsGenerator(sc->framesize(), 0x110));
                    s->conv.reset(new ConvEncoder(sc->framesize()));
                    s->punc.reset(new PuncturingEncoder(sc->framesizeCu()));
                    for (const auto &r : sc->get_rules()) s->punc->append_rule(r);
                    s->punc->append_tail_rule(PuncturingRule(3, 0xcccccc));
                    s->ti.reset(new TimeInterleaver(sc->framesizeCu() * 8));
                    subs.push_back(std::move(s));
                }
            }
            Buffer prbs, f0, f1, f2, f3, cif;
            cifPrbs->process({}, {&prbs});
            fic->process(&f0);
            ficPrbs->process({&f0}, {&f1});
            ficConv->process(&f1, &f2);
            ficPunc->process(&f2, &f3);
            std::vector<Buffer *> muxin{&prbs};
            // the subchannel sources are re-created by the reader only when the STC changes
            const auto cur = reader.getSubchannels();
            if (cur.size() != subs.size()) return -3;
            for (size_t i = 0; i < subs.size(); ++i) {
                Sub &s = *subs[i];
                if (cur[i] != s.src) return -3;
                s.src->process(&s.b0);
                s.prbs->process({&s.b0}, {&s.b1});
                s.conv->process(&s.b1, &s.b2);
                s.punc->process(&s.b2, &s.b3);
                s.ti->process(&s.b3, &s.b4);
                muxin.push_back(&s.b4);
            }
            cifMux->process(muxin, &cif);
            if (cifPart->process({&f3, &cif}, &part)) {
                if (pos + part.getLength() > out_cap) return -2;
                memcpy(out + pos, part.getData(), part.getLength());
                pos += part.getLength();
                ++nblocks;
            }
        }
        return nblocks;
    } catch (const std::exception &) { return -1; }
}

// f-1 piece by piece (each against one reference class)
int ref_prbs(size_t framesize, const uint8_t *in /* may be NULL */, uint8_t *out)
{
    try {
        PrbsGenerator g(framesize, 0x110);
        Buffer bi, bo;
        if (in) { fill(bi, in, framesize); g.process({&bi}, {&bo}); } else g.process({}, {&bo});
        memcpy(out, bo.getData(), bo.getLength());
        return (int)bo.getLength();
    } catch (const std::exception &) { return -1; }
}

int ref_conv_encode(const uint8_t *in, size_t framesize, uint8_t *out)
{
    try {
        ConvEncoder e(framesize);
        Buffer bi, bo;
        fill(bi, in, framesize);
        e.process(&bi, &bo);
        memcpy(out, bo.getData(), bo.getLength());
        return (int)bo.getLength();
    } catch (const std::exception &) { return -1; }
}

// sub-channel protection: the puncturing rules and CU count the reference derives from (STL, TPL);
// rules: up to 8 (length, pattern) pairs.  Returns the number of rules, -1 when the reference throws.
int ref_subchannel_profile(unsigned stl, unsigned tpl, uint32_t *rules, size_t *framesize_cu,
                           size_t *bitrate)
{
    try {
        SubchannelSource s(0, (uint16_t)stl, (uint8_t)tpl);
        int n = 0;
        for (const auto &r : s.get_rules()) {
            if (n >= 8) return -2;
            rules[2 * n] = (uint32_t)r.length();
            rules[2 * n + 1] = r.pattern();
            ++n;
        }
        *framesize_cu = s.framesizeCu();
        *bitrate = s.bitrate();
        return n;
    } catch (const std::exception &) { return -1; }
}

// puncture `in` with the rules of (stl, tpl) plus the tail rule, as DabModulator configures it
int ref_puncture(const uint8_t *in, size_t in_len, unsigned stl, unsigned tpl, int is_fic, unsigned mid,
                 uint8_t *out)
{
    try {
        std::unique_ptr<PuncturingEncoder> p;
        if (is_fic) {
            FicSource f(1, mid);
            p.reset(new PuncturingEncoder());
            for (const auto &r : f.get_rules()) p->append_rule(r);
        } else {
            SubchannelSource s(0, (uint16_t)stl, (uint8_t)tpl);
            p.reset(new PuncturingEncoder(s.framesizeCu()));
            for (const auto &r : s.get_rules()) p->append_rule(r);
        }
        p->append_tail_rule(PuncturingRule(3, 0xcccccc));
        Buffer bi, bo;
        fill(bi, in, in_len);
        p->process(&bi, &bo);
        memcpy(out, bo.getData(), bo.getLength());
        return (int)bo.getLength();
    } catch (const std::exception &) { return -1; }
}

// nframes frames of `framesize` bytes through ONE TimeInterleaver
int ref_time_interleave(const uint8_t *in, size_t framesize, size_t nframes, uint8_t *out)
{
    try {
        TimeInterleaver ti(framesize);
        for (size_t f = 0; f < nframes; ++f) {
            Buffer bi, bo;
            fill(bi, in + f * framesize, framesize);
            ti.process(&bi, &bo);
            memcpy(out + f * framesize, bo.getData(), framesize);
        }
        return 0;
    } catch (const std::exception &) { return -1; }
}

// a12: CicEqualizer(nbCarriers, spacing, R)
int ref_cic_equalizer(const float *in, size_t nsamples, size_t carriers, size_t spacing, int R, float *out)
{
    try {
        CicEqualizer st(carriers, spacing, R);
        Buffer bi, bo;
        fill(bi, in, nsamples * sizeof(complexf));
        st.process(&bi, &bo);
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
