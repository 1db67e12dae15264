// Frontend.cpp -- see Frontend.h.  Every class cites the reference lines whose behaviour it keeps.
#include "Frontend.h"

#include <algorithm>
#include <cstdio>
#include <cstring>

namespace {

#include "protection_tables.inc"

// MSB-first bit sink over a byte buffer
class BitWriter {
public:
    explicit BitWriter(uint8_t *out) : m_out(out) {}
    void put(unsigned bit)
    {
        m_acc = (m_acc << 1) | (bit & 1u);
        if (++m_nbits == 8) flush_byte();
    }
    // zero-fill the byte in progress, if any
    void align()
    {
        while (m_nbits) put(0);
    }
    size_t bytes() const { return m_count; }

private:
    void flush_byte()
    {
        m_out[m_count++] = static_cast<uint8_t>(m_acc);
        m_acc = 0;
        m_nbits = 0;
    }
    uint8_t *m_out;
    unsigned m_acc = 0, m_nbits = 0;
    size_t m_count = 0;
};

inline unsigned parity32(uint32_t v) { return static_cast<unsigned>(__builtin_parity(v)); }

}  // namespace

// ---------------------------------------------------------------- PuncturingRule
size_t PuncturingRule::bit_size() const { return static_cast<size_t>(__builtin_popcount(m_pattern)); }

// ---------------------------------------------------------------- PrbsGenerator
PrbsGenerator::PrbsGenerator(size_t framesize, uint32_t polynomial, uint32_t accum, size_t init)
    : m_framesize(framesize), m_polynomial(polynomial), m_accum_init(accum), m_init(init)
{
}

int PrbsGenerator::process(std::vector<Buffer *> dataIn, std::vector<Buffer *> dataOut)
{
    if (dataIn.size() > 1)
        throw std::runtime_error("Invalid dataIn size for PrbsGenerator " + std::to_string(dataIn.size()));
    if (dataOut.size() != 1)
        throw std::runtime_error("Invalid dataOut size for PrbsGenerator " + std::to_string(dataOut.size()));
    dataOut[0]->setLength(m_framesize);
    uint8_t *out = static_cast<uint8_t *>(dataOut[0]->getData());

    // The register restarts on every call (all ones up to the polynomial's degree, reference :144-153), so
    // the sequence is the same every time: generated once, copied afterwards.
    if (m_sequence.size() != m_framesize) {
        m_sequence.resize(m_framesize);
        uint32_t acc = m_accum_init;
        if (!acc)
            while (acc < m_polynomial) acc = (acc << 1) | 1u;
        size_t i = 0;
        for (; i < m_init && i < m_framesize; ++i) m_sequence[i] = 0xff;
        for (; i < m_framesize; ++i) {
            // eight steps of the Fibonacci register; the byte is its low eight bits (reference :58-73,112-123)
            for (int k = 0; k < 8; ++k) acc = (acc << 1) ^ parity32(acc & m_polynomial);
            // (the DVB variant of the reference blanks every 188th byte, :163-165)
            m_sequence[i] = (m_accum_init == 0xa9 && i % 188 == 0) ? 0 : static_cast<uint8_t>(acc);
        }
    }
    std::memcpy(out, m_sequence.data(), m_framesize);
    if (!dataIn.empty()) {
        if (dataIn[0]->getLength() != m_framesize)
            throw std::runtime_error("PrbsGenerator::process input size is not equal to output size!\n");
        const uint8_t *in = static_cast<const uint8_t *>(dataIn[0]->getData());
        for (size_t j = 0; j < m_framesize; ++j) out[j] ^= in[j];
    }
    return static_cast<int>(m_framesize);
}

// ---------------------------------------------------------------- ConvEncoder
ConvEncoder::ConvEncoder(size_t framesize) : m_framesize(framesize) {}

int ConvEncoder::process(Buffer *const dataIn, Buffer *dataOut)
{
    if (dataIn->getLength() != m_framesize)
        throw std::runtime_error("ConvEncoder::process input size not valid!\n");
    dataOut->setLength(4 * m_framesize + 3);
    const uint8_t *in = static_cast<const uint8_t *>(dataIn->getData());
    uint8_t *out = static_cast<uint8_t *>(dataOut->getData());
    // 7-bit register, new bit enters at bit 6; generators 133, 171, 145, 133 (octal) read as the masks
    // 0x5b, 0x79, 0x65, 0x5b on it (reference :95-113).  One input byte = 32 code bits that depend on the
    // byte and on the six bits before it: a 64 x 256 table of code words, built once.
    struct Table {
        uint32_t code[64][256];
        uint8_t next[256];
        Table()
        {
            static const unsigned gen[4] = {0x5b, 0x79, 0x65, 0x5b};
            for (unsigned st = 0; st < 64; ++st)
                for (unsigned b = 0; b < 256; ++b) {
                    unsigned reg = st << 1;
                    uint32_t w = 0;
                    for (int i = 7; i >= 0; --i) {
                        reg = (reg >> 1) | (((b >> i) & 1u) << 6);
                        for (unsigned g : gen) w = (w << 1) | parity32(reg & g);
                    }
                    code[st][b] = w;
                    next[b] = static_cast<uint8_t>(reg >> 1);      // the six newest bits: a function of the byte alone
                }
        }
    };
    static const Table tab;
    unsigned st = 0;
    for (size_t i = 0; i < m_framesize; ++i) {
        const uint32_t w = tab.code[st][in[i]];
        out[4 * i] = static_cast<uint8_t>(w >> 24);
        out[4 * i + 1] = static_cast<uint8_t>(w >> 16);
        out[4 * i + 2] = static_cast<uint8_t>(w >> 8);
        out[4 * i + 3] = static_cast<uint8_t>(w);
        st = tab.next[in[i]];
    }
    // flush: six zero bits = 24 code bits (reference :120-139): the top 24 bits of the word for a zero byte
    const uint32_t tail = tab.code[st][0];
    out[4 * m_framesize] = static_cast<uint8_t>(tail >> 24);
    out[4 * m_framesize + 1] = static_cast<uint8_t>(tail >> 16);
    out[4 * m_framesize + 2] = static_cast<uint8_t>(tail >> 8);
    return static_cast<int>(4 * m_framesize + 3);
}

// ---------------------------------------------------------------- PuncturingEncoder
void PuncturingEncoder::adjust_item_size()
{
    size_t in_size = 0, out_bits = 0;
    for (const auto &r : m_rules) {
        const size_t groups = (r.length() + 3) / 4;      // reference :60-65 counts down by 4
        in_size += 4 * groups;
        out_bits += groups * r.bit_size();
    }
    if (m_tail_rule) {
        in_size += m_tail_rule->length();
        out_bits += m_tail_rule->bit_size();
    }
    m_in_block_size = in_size;
    m_out_block_size = (out_bits + 7) / 8;
}

void PuncturingEncoder::append_rule(const PuncturingRule &rule)
{
    m_rules.push_back(rule);
    adjust_item_size();
}

void PuncturingEncoder::append_tail_rule(const PuncturingRule &rule)
{
    m_tail_rule.reset(new PuncturingRule(rule));
    adjust_item_size();
}

int PuncturingEncoder::process(Buffer *const dataIn, Buffer *dataOut)
{
    if (m_num_cu > 0) {
        // EN 300 401 table 31: some UEP profiles carry one padding byte (reference :125-139)
        if (m_num_cu * 8 == m_out_block_size + 1) m_out_block_size = m_num_cu * 8;
        if (m_num_cu * 8 != m_out_block_size)
            throw std::runtime_error("PuncturingEncoder encoder initialisation failed.  CU: " +
                                     std::to_string(m_num_cu) + " block_size: " + std::to_string(m_out_block_size));
    }
    dataOut->setLength(m_out_block_size);
    if (dataIn->getLength() != m_in_block_size) throw std::runtime_error("PuncturingEncoder::process wrong input size");
    const uint8_t *in = static_cast<const uint8_t *>(dataIn->getData());
    uint8_t *out = static_cast<uint8_t *>(dataOut->getData());
    BitWriter w(out);

    const size_t body = m_in_block_size - (m_tail_rule ? m_tail_rule->length() : 0);
    size_t pos = 0;
    // the rules apply in turn to successive 4-byte groups, length/4 groups each, cycling (reference :152-176)
    for (size_t r = 0; pos < body; r = (r + 1) % m_rules.size()) {
        if (m_rules.empty()) throw std::runtime_error("PuncturingEncoder::process no rules");
        const uint32_t pattern = m_rules[r].pattern();
        for (size_t g = (m_rules[r].length() + 3) / 4; g > 0; --g) {
            const uint32_t word = (uint32_t(in[pos]) << 24) | (uint32_t(in[pos + 1]) << 16) |
                                  (uint32_t(in[pos + 2]) << 8) | uint32_t(in[pos + 3]);
            pos += 4;
            for (int b = 31; b >= 0; --b)
                if ((pattern >> b) & 1u) w.put((word >> b) & 1u);
        }
    }
    if (m_tail_rule) {
        // 24-bit vector on the last bytes (reference :177-196)
        const uint32_t pattern = m_tail_rule->pattern();
        uint32_t mask = 0x800000;
        for (size_t i = 0; i < m_tail_rule->length(); ++i) {
            const uint8_t d = in[pos++];
            for (int b = 7; b >= 0; --b, mask >>= 1)
                if (pattern & mask) w.put((d >> b) & 1u);
        }
    }
    w.align();
    size_t n = w.bytes();
    if (n > m_out_block_size) throw std::runtime_error("PuncturingEncoder::process output size does not correspond!");
    while (n < m_out_block_size) out[n++] = 0;
    return static_cast<int>(m_out_block_size);
}

// ---------------------------------------------------------------- TimeInterleaver
TimeInterleaver::TimeInterleaver(size_t framesize) : m_framesize(framesize)
{
    if (framesize & 1) throw std::invalid_argument("framesize must be 16 bits multiple");
    for (auto &h : m_history) h.assign(framesize, 0);
}

int TimeInterleaver::process(Buffer *const dataIn, Buffer *dataOut)
{
    if (dataIn->getLength() != m_framesize)
        throw std::invalid_argument("Interleaver buffer input size " + std::to_string(dataIn->getLength()) +
                                    " expected " + std::to_string(m_framesize));
    dataOut->setLength(m_framesize);
    const uint8_t *in = static_cast<const uint8_t *>(dataIn->getData());
    uint8_t *out = static_cast<uint8_t *>(dataOut->getData());
    // the newest frame replaces the oldest slot of the ring; slot(k) = the frame k calls ago
    m_head = (m_head + 15) & 15;
    std::memcpy(m_history[m_head].data(), in, m_framesize);
    // bit b (MSB first) of byte j comes from the frame delayed by bitrev4(2b + (j & 1)) ... spelled
    // out: even bytes 0,8,4,12,2,10,6,14, odd bytes 1,9,5,13,3,11,7,15 (reference :66-93)
    static const unsigned delay[2][8] = {{0, 8, 4, 12, 2, 10, 6, 14}, {1, 9, 5, 13, 3, 11, 7, 15}};
    // one pass per bit plane over 16-bit words (even byte | odd byte): plain loops the compiler vectorises
    const size_t nw = m_framesize / 2;
    for (int b = 0; b < 8; ++b) {
        const uint8_t *he = m_history[(m_head + delay[0][b]) & 15].data();
        const uint8_t *ho = m_history[(m_head + delay[1][b]) & 15].data();
        const uint8_t mask = static_cast<uint8_t>(0x80u >> b);
        uint8_t *o = out;
        if (b == 0) {
            for (size_t w = 0; w < nw; ++w) { o[2 * w] = he[2 * w] & mask; o[2 * w + 1] = ho[2 * w + 1] & mask; }
        } else {
            for (size_t w = 0; w < nw; ++w) { o[2 * w] |= he[2 * w] & mask; o[2 * w + 1] |= ho[2 * w + 1] & mask; }
        }
    }
    return static_cast<int>(m_framesize);
}

// ---------------------------------------------------------------- FicSource
FicSource::FicSource(unsigned ficf, unsigned mid)
{
    if (ficf == 0) return;
    // 3 or 4 FIBs of 32 bytes per 24 ms; code rate 1/3: 21 (29) blocks of PI 16, 3 of PI 15
    // (EN 300 401 clause 11.2; reference :51-59)
    const size_t fibs = (mid == 3) ? 4 : 3;
    m_framesize = 32 * fibs;
    m_rules.emplace_back((8 * fibs - 3) * 16, kPuncturingVector[16]);
    m_rules.emplace_back(3 * 16, kPuncturingVector[15]);
    m_buffer.setLength(m_framesize);
}

int FicSource::process(Buffer *outputData)
{
    if (m_buffer.getLength() != m_framesize)
        throw std::runtime_error("ERROR: FicSource::process.outputSize != m_framesize: " +
                                 std::to_string(m_buffer.getLength()) + " != " + std::to_string(m_framesize));
    *outputData = m_buffer;
    return static_cast<int>(outputData->getLength());
}

// ---------------------------------------------------------------- SubchannelSource
SubchannelSource::SubchannelSource(uint16_t sad, uint16_t stl, uint8_t tpl)
    : m_start_address(sad), m_framesize(static_cast<size_t>(stl) * 8), m_protection(tpl)
{
    auto rule = [&](size_t blocks, int pi) { m_rules.emplace_back(blocks * 16, kPuncturingVector[pi]); };
    const size_t level = protectionLevel();
    if (protectionForm()) {
        // equal error protection, EN 300 401 clause 11.3.2 (reference :84-163 rules, :657-688 size)
        // (integer arithmetic in the reference's order -- multiply, divide, then the offset -- so that
        // bit rates off the 8 / 32 kbit/s grid resolve to the same rules as well)
        const size_t br = bitrate();
        if (protectionOption() == 0) {                // set A: multiples of 8 kbit/s
            switch (level) {
                case 1: rule(6 * br / 8 - 3, 24); rule(3, 23); m_framesize_cu = (br / 8) * 12; break;
                case 2:
                    if (br == 8) { rule(5, 13); rule(1, 12); } else { rule(2 * br / 8 - 3, 14); rule(4 * br / 8 + 3, 13); }
                    m_framesize_cu = (br / 8) * 8;
                    break;
                case 3: rule(6 * br / 8 - 3, 8); rule(3, 7); m_framesize_cu = (br / 8) * 6; break;
                case 4: rule(4 * br / 8 - 3, 3); rule(2 * br / 8 + 3, 2); m_framesize_cu = (br / 8) * 4; break;
                default: throw std::runtime_error("SubchannelSource::SubchannelSource unknown protection level!");
            }
        } else if (protectionOption() == 1) {         // set B: multiples of 32 kbit/s
            static const int pi_body[4] = {10, 6, 4, 2}, pi_end[4] = {9, 5, 3, 1};
            static const size_t cu[4] = {27, 21, 18, 15};
            if (level < 1 || level > 4)
                throw std::runtime_error("SubchannelSource::SubchannelSource unknown protection level!");
            rule(24 * br / 32 - 3, pi_body[level - 1]);
            rule(3, pi_end[level - 1]);
            m_framesize_cu = (br / 32) * cu[level - 1];
        } else {
            throw std::runtime_error("SubchannelSource::SubchannelSource unknown protection option!");
        }
    } else {
        // unequal error protection: table 31 profile for (bit rate, protection level)
        const UepProfile *p = nullptr;
        for (const UepProfile &q : kUepProfiles)
            if (q.bitrate == bitrate() && q.level == level) p = &q;
        if (!p) throw std::runtime_error("SubchannelSource UEP puncturing rules do not exist!");
        for (int i = 0; i < p->nrules; ++i) rule(p->l[i], p->pi[i]);
        m_framesize_cu = p->cu;
    }
}

size_t SubchannelSource::framesizeCu() const
{
    // reference :997-1008
    if (m_framesize_cu == 0) throw std::runtime_error("SubchannelSource::framesizeCu protection not yet coded!");
    if (m_framesize_cu == 0xffff) throw std::runtime_error("SubchannelSource::framesizeCu invalid protection!");
    return m_framesize_cu;
}

int SubchannelSource::process(Buffer *outputData)
{
    if (m_buffer.getLength() != m_framesize)
        throw std::runtime_error("ERROR: Subchannel::process: d_buffer != d_framesize: " +
                                 std::to_string(m_buffer.getLength()) + " != " + std::to_string(m_framesize));
    *outputData = m_buffer;
    return static_cast<int>(outputData->getLength());
}

// ---------------------------------------------------------------- EtiReader
EtiReader::EtiReader(double &) {}

unsigned EtiReader::getMode()
{
    if (!m_fc_valid) throw std::runtime_error("Trying to access Mode before it is ready!");
    return m_mid;
}

unsigned EtiReader::getFp()
{
    if (!m_fc_valid) throw std::runtime_error("Trying to access FP before it is ready!");
    return m_fp;
}

unsigned EtiReader::getFct()
{
    if (!m_fc_valid) throw std::runtime_error("Trying to access FCT before it is ready!");
    return m_fct;
}

int EtiReader::loadEtiData(const Buffer &dataIn)
{
    const uint8_t *in = static_cast<const uint8_t *>(dataIn.getData());
    size_t left = dataIn.getLength();
    auto take = [&](size_t n) { in += n; left -= n; m_remaining -= n; };
    while (left > 0) {
        switch (m_state) {
            case State::Sync:                                  // ERR + FSYNC
                if (m_resync) {
                    // After a refused frame whose end could not be located (below), the stream position is not known to
                    // be a frame start: look for one of the two FSYNC words (src/Eti.h:50-61, doc/README-Fileinput)
                    // instead of taking the next four bytes on trust.  The ERR byte in front of it is NOT tested -- a
                    // multiplexer may mark its frames with an error level other than 0xFF, and such a stream must lock
                    // again --, and a candidate is confirmed, whenever this buffer reaches that far, by the OTHER FSYNC
                    // word 6144 bytes later (the two alternate): four payload bytes that happen to spell FSYNC do not
                    // lock the reader onto payload.  Bytes that cannot start a frame are consumed, including a tail
                    // shorter than a sync word: a caller that treats "consumed < length" as a read error sees none.
                    auto fsync = [](const uint8_t *p) {
                        return (p[1] == 0x07 && p[2] == 0x3A && p[3] == 0xB6) ? 1 : (p[1] == 0xF8 && p[2] == 0xC5 && p[3] == 0x49) ? 2 : 0;
                    };
                    if (left < 4) { in += left; left = 0; break; }
                    const int k = fsync(in);
                    bool lock = k != 0;
                    if (lock && left >= 6144 + 4) {
                        const int k2 = fsync(in + 6144);
                        lock = k2 != 0 && k2 != k;
                    }
                    if (!lock) {
                        ++in;
                        --left;
                        break;
                    }
                    m_resync = false;
                }
                if (left < 4) return static_cast<int>(dataIn.getLength() - left);
                m_remaining = 6144;
                take(4);
                m_state = State::Fc;
                break;
            case State::Fc:                                    // FCT | FICF,NST | FP,MID,FL[10:8] | FL[7:0]
                if (left < 4) return static_cast<int>(dataIn.getLength() - left);
                m_fct = in[0];
                m_ficf = in[1] >> 7;
                m_nst = in[1] & 0x7f;
                m_fp = in[2] >> 5;
                m_mid = (in[2] >> 3) & 3;
                m_fc_valid = true;
                take(4);
                m_state = State::Nst;
                if (!m_ficf) throw std::runtime_error("FIC must be present to modulate!");
                if (!myFicSource) myFicSource = std::make_shared<FicSource>(m_ficf, m_mid);
                break;
            case State::Nst: {                                 // SCID,SAD[9:8] | SAD[7:0] | TPL,STL[9:8] | STL[7:0]
                const size_t n = 4 * static_cast<size_t>(m_nst);
                if (left < n) return static_cast<int>(dataIn.getLength() - left);
                if (m_stc.size() != n || std::memcmp(m_stc.data(), in, n) != 0) {
                    // a new multiplex layout: new sources (the modulator rebuilds around them)
                    m_stc.assign(in, in + n);
                    mySources.clear();
                    for (unsigned i = 0; i < m_nst; ++i) {
                        const uint8_t *s = &m_stc[4 * i];
                        const uint16_t sad = static_cast<uint16_t>(((s[0] & 3u) << 8) | s[1]);
                        const uint16_t stl = static_cast<uint16_t>(((s[2] & 3u) << 8) | s[3]);
                        mySources.push_back(std::make_shared<SubchannelSource>(sad, stl, static_cast<uint8_t>(s[2] >> 2)));
                    }
                }
                take(n);
                m_state = State::Eoh;
                {
                    // Consistency against the 6144-byte frame (the reference does not check: a header whose
                    // sub-channels overrun the frame would make m_remaining wrap and the reader swallow the
                    // frames that follow as payload).  Refuse it and resynchronise on the next frame.
                    size_t need = 4 /* EOH */ + myFicSource->getFramesize() + 4 /* EOF */ + 4 /* TIST */;
                    for (const auto &src : mySources) need += src->framesize();
                    if (need > m_remaining) {
                        // The exception drops the rest of THIS input buffer; whatever part of the offending frame lies
                        // beyond it (a caller feeding a byte stream in pieces) is skipped as padding, so that the
                        // next frame start is a real one and its payload is never parsed as FC / STC.
                        const size_t beyond = m_remaining > left ? m_remaining - left : 0;
                        // Is the NEXT call's first byte a frame start?  Yes when the offending frame ends inside this buffer
                        // and what follows it here is a whole number of frames -- the frame-aligned caller (one frame, or n
                        // frames, per call: the reference's own loop, src/DabMod.cpp:605-724), who keeps the reference's
                        // behaviour of taking frame starts on trust.  Otherwise alignment is lost (the tail of this buffer
                        // is dropped in mid-frame) and the next frame start is searched for.
                        const bool aligned = beyond == 0 && (left - m_remaining) % 6144 == 0;
                        m_remaining = beyond;
                        m_state = beyond ? State::Pad : State::Sync;
                        m_resync = !aligned;
                        m_stc.clear();
                        mySources.clear();
                        throw std::runtime_error("EtiReader: stream characterisation exceeds the 6144-byte ETI frame");
                    }
                }
                break;
            }
            case State::Eoh:                                   // MNSC + header CRC (not checked, as in the reference)
                if (left < 4) return static_cast<int>(dataIn.getLength() - left);
                take(4);
                m_state = State::Fic;
                break;
            case State::Fic: {
                const size_t n = myFicSource->getFramesize();
                if (left < n) return static_cast<int>(dataIn.getLength() - left);
                myFicSource->loadFicData(Buffer(n, in));
                take(n);
                m_state = State::Subch;
                break;
            }
            case State::Subch: {
                size_t total = 0;
                for (const auto &s : mySources) total += s->framesize();
                if (left < total) return static_cast<int>(dataIn.getLength() - left);
                for (const auto &s : mySources) {
                    s->loadSubchannelData(Buffer(s->framesize(), in));
                    take(s->framesize());
                }
                m_state = State::Eof;
                break;
            }
            case State::Eof:                                   // CRC + RFU
                if (left < 4) return static_cast<int>(dataIn.getLength() - left);
                take(4);
                m_state = State::Tist;
                break;
            case State::Tist:
                if (left < 4) return static_cast<int>(dataIn.getLength() - left);
                take(4);
                m_state = State::Pad;
                break;
            case State::Pad: {
                const size_t n = std::min(left, m_remaining);
                take(n);
                if (m_remaining == 0) m_state = State::Sync;
                break;
            }
        }
    }
    return static_cast<int>(dataIn.getLength() - left);
}

// ---------------------------------------------------------------- InputFileReader
namespace {
bool is_sync(const uint8_t *p)
{
    // ERR = 0xFF followed by FSYNC 0x073AB6 or its complement (reference :84: the little-endian
    // words 0x49c5f8ff / 0xb63a07ff)
    return p[0] == 0xFF && ((p[1] == 0x07 && p[2] == 0x3A && p[3] == 0xB6) || (p[1] == 0xF8 && p[2] == 0xC5 && p[3] == 0x49));
}
}  // namespace

int InputFileReader::Open(const std::string &filename, bool loop)
{
    m_filename = filename;
    m_loop = loop;
    FILE *f = std::fopen(filename.c_str(), "rb");
    if (!f) return -1;
    m_data.clear();
    uint8_t chunk[65536];
    size_t n;
    while ((n = std::fread(chunk, 1, sizeof chunk, f)) > 0) m_data.insert(m_data.end(), chunk, chunk + n);
    std::fclose(f);
    return identify();
}

int InputFileReader::identify()
{
    m_type = EtiStreamType::None;
    const size_t n = m_data.size();
    const uint8_t *d = m_data.data();
    if (n >= 4 && is_sync(d)) {                       // raw
        m_type = EtiStreamType::Raw;
        m_start = 0;
    } else if (n >= 6 && is_sync(d + 2)) {            // streamed: u16 frame length first
        m_type = EtiStreamType::Streamed;
        m_start = 0;
    } else if (n >= 10 && is_sync(d + 6)) {           // framed: u32 number of frames, then as streamed
        m_type = EtiStreamType::Framed;
        m_start = 4;
    } else {
        // raw with leading garbage: look for the first sync word in the next 6144 bytes (reference :161-190)
        for (size_t i = 7; i + 4 <= n && i < 6144 + 10; ++i)
            if (is_sync(d + i)) {
                m_type = EtiStreamType::Raw;
                m_start = i;
                break;
            }
    }
    m_pos = m_start;
    return m_type == EtiStreamType::None ? -1 : 0;
}

std::string InputFileReader::GetPrintableInfo() const
{
    static const char *names[] = {"unknown!", "raw", "streamed", "framed"};
    std::string info = std::string("Input file format: ") + names[static_cast<int>(m_type)] +
                       ", length: " + std::to_string(m_data.size());
    if (m_type == EtiStreamType::Raw) info += ", nb frames: " + std::to_string((m_data.size() - m_start) / 6144);
    return info;
}

int InputFileReader::GetNextFrame(void *buffer)
{
    if (m_type == EtiStreamType::None) return -1;
    for (int attempt = 0; attempt < 2; ++attempt) {
        size_t len = 6144;
        size_t p = m_pos;
        if (m_type != EtiStreamType::Raw) {
            if (p + 2 > m_data.size()) { len = 0; }
            else {
                len = static_cast<size_t>(m_data[p]) | (static_cast<size_t>(m_data[p + 1]) << 8);
                p += 2;
                if (len > 6144) return -1;             // "Wrong frame size" (reference :241-244)
            }
        }
        if (len && p + len <= m_data.size()) {
            std::memcpy(buffer, m_data.data() + p, len);
            std::memset(static_cast<uint8_t *>(buffer) + len, 0x55, 6144 - len);
            m_pos = p + len;
            return 6144;
        }
        if (len && p < m_data.size()) return -1;       // a truncated last frame
        if (!m_loop) return 0;                         // end of file
        m_pos = m_start;                               // rewind once and try again
    }
    return -1;
}

// ---------------------------------------------------------------- FrameMultiplexer
int FrameMultiplexer::process(std::vector<Buffer *> dataIn, Buffer *dataOut)
{
    if (dataIn.empty() || dataIn[0]->getLength() != 864 * 8)
        throw FrameMultiplexerError("FrameMultiplexer: input 0 must be one CIF of padding");
    *dataOut = *dataIn[0];
    uint8_t *out = static_cast<uint8_t *>(dataOut->getData());
    const auto subchannels = m_etiSource.getSubchannels();
    if (subchannels.size() != dataIn.size() - 1)
        throw FrameMultiplexerError("FrameMultiplexer detected subchannel size change from " +
                                    std::to_string(dataIn.size() - 1) + " to " + std::to_string(subchannels.size()));
    for (size_t i = 0; i < subchannels.size(); ++i) {
        const Buffer *in = dataIn[i + 1];
        if (subchannels[i]->framesizeCu() * 8 != in->getLength())
            throw FrameMultiplexerError("FrameMultiplexer detected invalid subchannel size! " +
                                        std::to_string(subchannels[i]->framesizeCu() * 8) + " != " +
                                        std::to_string(in->getLength()));
        const size_t offset = subchannels[i]->startAddress() * 8;     // capacity units of 64 bits
        if (offset + in->getLength() > dataOut->getLength())
            throw FrameMultiplexerError("FrameMultiplexer: sub-channel beyond the end of the CIF");
        std::memcpy(out + offset, in->getData(), in->getLength());
    }
    return static_cast<int>(dataOut->getLength());
}

// ---------------------------------------------------------------- BlockPartitioner
BlockPartitioner::BlockPartitioner(unsigned mode)
{
    switch (mode) {                                   // reference :43-73
        case 1: m_ficSize = 2304 / 8; m_cifCount = 4; break;
        case 2: m_ficSize = 2304 / 8; m_cifCount = 1; break;
        case 3: m_ficSize = 3072 / 8; m_cifCount = 1; break;
        case 4: m_ficSize = 2304 / 8; m_cifCount = 2; break;
        default: throw std::runtime_error("BlockPartitioner::BlockPartitioner invalid mode");
    }
}

int BlockPartitioner::process(std::vector<Buffer *> dataIn, Buffer *dataOut)
{
    if (dataIn.size() != 2) throw std::runtime_error("BlockPartitioner::process needs FIC and CIF");
    dataOut->setLength(m_cifCount * (m_ficSize + m_cifSize));
    if (dataIn[0]->getLength() != m_ficSize) throw std::runtime_error("BlockPartitioner::process input 0 size not valid!");
    if (dataIn[1]->getLength() != m_cifSize) throw std::runtime_error("BlockPartitioner::process input 1 size not valid!");
    uint8_t *out = static_cast<uint8_t *>(dataOut->getData());
    // [FIC_0 .. FIC_{n-1} | CIF_0 .. CIF_{n-1}] (reference :111-117)
    std::memcpy(out + m_cifNb * m_ficSize, dataIn[0]->getData(), m_ficSize);
    std::memcpy(out + m_cifCount * m_ficSize + m_cifNb * m_cifSize, dataIn[1]->getData(), m_cifSize);
    m_cifNb = (m_cifNb + 1) % m_cifCount;
    return m_cifNb == 0;
}

meta_vec_t BlockPartitioner::process_metadata(const meta_vec_t &metadataIn)
{
    if (m_cifNb == 1) m_meta.clear();                 // reference :126-140
    m_meta.insert(m_meta.end(), metadataIn.begin(), metadataIn.end());
    return m_cifNb == 0 ? m_meta : meta_vec_t{};
}

// ---------------------------------------------------------------- EtiFrontend
struct EtiFrontend::Sub {
    std::shared_ptr<SubchannelSource> src;
    std::unique_ptr<PrbsGenerator> prbs;
    std::unique_ptr<ConvEncoder> conv;
    std::unique_ptr<PuncturingEncoder> punc;
    std::unique_ptr<TimeInterleaver> interleaver;
    Buffer b0, b1, b2, b3, b4;
};

// Mode 0 ("take it from the ETI stream") is only a provisional Mode I in the reference's constructor; the
// flowgraph is built by DabModulator::process with setMode(m_settings.dabMode), which rejects it
// (src/DabModulator.cpp:75-80, :131-133, :119-121).  The stage classes below that level (PhaseReference,
// FrequencyInterleaver: src/PhaseReference.cpp:72-76) read 0 as Mode IV, and so does dabgpu_create.
EtiFrontend::EtiFrontend(unsigned mode) : m_mode(mode), m_reader(m_tist_offset)
{
    if (mode < 1 || mode > 4) throw std::runtime_error("DabModulator::setMode invalid mode size");
}

void EtiFrontend::build()
{
    // src/DabModulator.cpp:138-140 (CIF), :286-321 (FIC), :326-383 (sub-channels)
    m_cifPrbs.reset(new PrbsGenerator(864 * 8, 0x110));
    m_cifMux.reset(new FrameMultiplexer(m_reader));
    m_cifPart.reset(new BlockPartitioner(m_mode));
    m_fic = m_reader.getFic();
    const size_t n = m_fic->getFramesize();
    m_ficPrbs.reset(new PrbsGenerator(n, 0x110));
    m_ficConv.reset(new ConvEncoder(n));
    m_ficPunc.reset(new PuncturingEncoder());
    for (const auto &r : m_fic->get_rules()) m_ficPunc->append_rule(r);
    m_ficPunc->append_tail_rule(PuncturingRule(3, 0xcccccc));
    m_subs.clear();
    for (const auto &sc : m_reader.getSubchannels()) {
        auto s = std::make_shared<Sub>();
        s->src = sc;
        s->prbs.reset(new PrbsGenerator(sc->framesize(), 0x110));
        s->conv.reset(new ConvEncoder(sc->framesize()));
        s->punc.reset(new PuncturingEncoder(sc->framesizeCu()));
        for (const auto &r : sc->get_rules()) s->punc->append_rule(r);
        s->punc->append_tail_rule(PuncturingRule(3, 0xcccccc));
        s->interleaver.reset(new TimeInterleaver(sc->framesizeCu() * 8));
        m_subs.push_back(std::move(s));
    }
}

bool EtiFrontend::push(const uint8_t *frame6144, Buffer &tf)
{
    m_reader.loadEtiData(Buffer(6144, frame6144));
    if (!m_started) {
        if (m_reader.getFp() != 0) return false;     // align the 4-frame groups (src/DabMod.cpp:684-693)
        m_started = true;
        build();
    }
    m_cifPrbs->process({}, {&m_prbs});
    m_fic->process(&m_f0);
    m_ficPrbs->process({&m_f0}, {&m_f1});
    m_ficConv->process(&m_f1, &m_f2);
    m_ficPunc->process(&m_f2, &m_f3);
    const auto current = m_reader.getSubchannels();
    if (current.size() != m_subs.size())
        throw FrameMultiplexerError("FrameMultiplexer detected subchannel size change from " +
                                    std::to_string(m_subs.size()) + " to " + std::to_string(current.size()));
    std::vector<Buffer *> mux{&m_prbs};
    for (size_t i = 0; i < m_subs.size(); ++i) {
        Sub &s = *m_subs[i];
        if (current[i] != s.src) throw FrameMultiplexerError("FrameMultiplexer detected a multiplex reconfiguration");
        s.src->process(&s.b0);
        s.prbs->process({&s.b0}, {&s.b1});
        s.conv->process(&s.b1, &s.b2);
        s.punc->process(&s.b2, &s.b3);
        s.interleaver->process(&s.b3, &s.b4);
        mux.push_back(&s.b4);
    }
    m_cifMux->process(mux, &m_cif);
    // BlockPartitioner fills one buffer over the 4 / 1 / 1 / 2 frames of a transmission frame
    if (m_cifPart->process({&m_f3, &m_cif}, &m_part) == 0) return false;
    tf = m_part;
    return true;
}
