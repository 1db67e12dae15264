// frontend_cabi.cpp -- include/dabfrontend.h over the classes of Frontend.h.
#include "dabfrontend.h"

#include "Frontend.h"

#include <cstring>
#include <exception>

namespace {
int copy_out(const Buffer &b, uint8_t *out)
{
    std::memcpy(out, b.getData(), b.getLength());
    return static_cast<int>(b.getLength());
}
}  // namespace

extern "C" {

int dabfe_prbs(size_t framesize, const uint8_t *in, uint8_t *out)
{
    try {
        PrbsGenerator g(framesize, 0x110);
        Buffer bi(in ? framesize : 0, in), bo;
        if (in) g.process({&bi}, {&bo}); else g.process({}, {&bo});
        return copy_out(bo, out);
    } catch (const std::exception &) { return -1; }
}

int dabfe_conv_encode(const uint8_t *in, size_t framesize, uint8_t *out)
{
    try {
        ConvEncoder e(framesize);
        Buffer bi(framesize, in), bo;
        e.process(&bi, &bo);
        return copy_out(bo, out);
    } catch (const std::exception &) { return -1; }
}

int dabfe_subchannel_profile(unsigned stl, unsigned tpl, uint32_t *rules, size_t *framesize_cu, size_t *bitrate)
{
    try {
        SubchannelSource s(0, static_cast<uint16_t>(stl), static_cast<uint8_t>(tpl));
        int n = 0;
        for (const auto &r : s.get_rules()) {
            if (n >= 8) return -2;
            rules[2 * n] = static_cast<uint32_t>(r.length());
            rules[2 * n + 1] = r.pattern();
            ++n;
        }
        *framesize_cu = s.framesizeCu();
        *bitrate = s.bitrate();
        return n;
    } catch (const std::exception &) { return -1; }
}

int dabfe_puncture(const uint8_t *in, size_t in_len, unsigned stl, unsigned tpl, int is_fic, unsigned mid, uint8_t *out)
{
    try {
        std::unique_ptr<PuncturingEncoder> p;
        if (is_fic) {
            FicSource f(1, mid);
            p.reset(new PuncturingEncoder());
            for (const auto &r : f.get_rules()) p->append_rule(r);
        } else {
            SubchannelSource s(0, static_cast<uint16_t>(stl), static_cast<uint8_t>(tpl));
            p.reset(new PuncturingEncoder(s.framesizeCu()));
            for (const auto &r : s.get_rules()) p->append_rule(r);
        }
        p->append_tail_rule(PuncturingRule(3, 0xcccccc));
        Buffer bi(in_len, in), bo;
        p->process(&bi, &bo);
        return copy_out(bo, out);
    } catch (const std::exception &) { return -1; }
}

int dabfe_time_interleave(const uint8_t *in, size_t framesize, size_t nframes, uint8_t *out)
{
    try {
        TimeInterleaver ti(framesize);
        for (size_t f = 0; f < nframes; ++f) {
            Buffer bi(framesize, in + f * framesize), bo;
            ti.process(&bi, &bo);
            std::memcpy(out + f * framesize, bo.getData(), framesize);
        }
        return 0;
    } catch (const std::exception &) { return -1; }
}

int dabfe_eti_frontend(const uint8_t *eti, size_t nframes, unsigned mode, uint8_t *out, size_t out_cap)
{
    try {
        EtiFrontend fe(mode);
        Buffer tf;
        size_t pos = 0;
        int blocks = 0;
        for (size_t f = 0; f < nframes; ++f) {
            if (!fe.push(eti + f * 6144, tf)) continue;
            if (pos + tf.getLength() > out_cap) return -2;
            std::memcpy(out + pos, tf.getData(), tf.getLength());
            pos += tf.getLength();
            ++blocks;
        }
        return blocks;
    } catch (const std::exception &) { return -1; }
}

int dabfe_eti_reader_stream(const uint8_t *bytes, size_t n, size_t piece, unsigned *fct_out, size_t fct_cap,
                            size_t *n_errors, size_t *n_short)
{
    // ONE EtiReader fed `piece` bytes at a time; after every call the frame counter of the header parsed last is noted
    // whenever it changed: the sequence of frames the reader locked onto
    if (!bytes || !piece) return -1;
    double off = 0.0;
    EtiReader rd(off);
    size_t nf = 0, errors = 0, shorts = 0;
    int last = -1;
    for (size_t pos = 0; pos < n;) {
        const size_t len = std::min(piece, n - pos);
        int used;
        try {
            used = rd.loadEtiData(Buffer(len, bytes + pos));
        } catch (const std::exception &) {
            ++errors;
            pos += len;                      // (the exception drops the rest of this buffer)
            continue;
        }
        if ((size_t)used != len) ++shorts;
        pos += used > 0 ? (size_t)used : len;
        try {
            const int fct = (int)rd.getFct();
            if (fct != last) {
                if (nf < fct_cap && fct_out) fct_out[nf] = (unsigned)fct;
                ++nf;
                last = fct;
            }
        } catch (const std::exception &) {}
    }
    if (n_errors) *n_errors = errors;
    if (n_short) *n_short = shorts;
    return (int)nf;
}

}  // extern "C"
