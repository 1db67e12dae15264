// Frontend.h -- the CPU front-end that produces the hot path's input:
// ETI(NI) frame -> FIC / sub-channel sources -> energy dispersal -> convolutional code ->
// puncturing -> time interleaving -> CIF assembly -> BlockPartitioner.
//
// Serial, bit-level, ~60 kB per transmission frame: it stays on the CPU (there is nothing for a
// GPU to win), written from scratch with the reference's class names and constructor signatures
// (file:line at each class) so that src/DabModulator.cpp:131-139,281-385 wires it unchanged.
// Pure integer work: bit-exact against the reference's classes (tests/test_frontend.py and the
// goldens it reads).  Not restated: timestamp decoding (MNSC/TIST -> metadata), EDI
// input, FIC decoding for the remote control -- metadata and I/O, SURVEY 2 rows 19-20.
#pragma once

#include "ModPlugin.h"

#include <array>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

// reference src/PuncturingRule.h, .cpp:25-42
class PuncturingRule {
public:
    PuncturingRule(size_t length, uint32_t pattern) : m_length(length), m_pattern(pattern) {}
    size_t length() const { return m_length; }      // bytes of mother-code output the rule covers
    uint32_t pattern() const { return m_pattern; }  // 32-bit puncturing vector, MSB first
    size_t bit_size() const;                        // bits kept per 4-byte group

private:
    size_t m_length;
    uint32_t m_pattern;
};

// reference src/PrbsGenerator.h:60-90, .cpp:31-188: energy-dispersal sequence (polynomial given as
// a tap mask, 0x110 = x^9 + x^5 + 1), XORed onto the input when there is one; without input it is
// the padding source of the CIF.
class PrbsGenerator : public ModPlugin {
public:
    PrbsGenerator(size_t framesize, uint32_t polynomial, uint32_t accum = 0, size_t init = 0);
    int process(std::vector<Buffer *> dataIn, std::vector<Buffer *> dataOut) override;
    const char *name() override { return "PrbsGenerator"; }

private:
    size_t m_framesize;
    uint32_t m_polynomial, m_accum_init;
    size_t m_init;
    std::vector<uint8_t> m_sequence;     // the (input-independent) sequence, generated on first use
};

// reference src/ConvEncoder.h, .cpp:59-150: K = 7 mother code of rate 1/4, six tail bits
class ConvEncoder : public ModCodec {
public:
    explicit ConvEncoder(size_t framesize);
    int process(Buffer *const dataIn, Buffer *dataOut) override;
    const char *name() override { return "ConvEncoder"; }

private:
    size_t m_framesize;
};

// reference src/PuncturingEncoder.h, .cpp:36-210
class PuncturingEncoder : public ModCodec {
public:
    PuncturingEncoder() = default;
    explicit PuncturingEncoder(size_t num_cu) : m_num_cu(num_cu) {}
    void append_rule(const PuncturingRule &rule);
    void append_tail_rule(const PuncturingRule &rule);
    int process(Buffer *const dataIn, Buffer *dataOut) override;
    const char *name() override { return "PuncturingEncoder"; }

private:
    void adjust_item_size();
    size_t m_num_cu = 0, m_in_block_size = 0, m_out_block_size = 0;
    std::vector<PuncturingRule> m_rules;
    std::unique_ptr<PuncturingRule> m_tail_rule;
};

// reference src/TimeInterleaver.h, .cpp:30-96: 16-frame convolutional interleaver
class TimeInterleaver : public ModCodec {
public:
    explicit TimeInterleaver(size_t framesize);
    int process(Buffer *const dataIn, Buffer *dataOut) override;
    const char *name() override { return "TimeInterleaver"; }

private:
    size_t m_framesize;
    unsigned m_head = 0;                              // slot of the newest frame in the ring
    std::array<std::vector<uint8_t>, 16> m_history;   // ring of the last 16 frames, zero at start
};

// reference src/FicSource.h, .cpp:36-111
class FicSource : public ModInput {
public:
    FicSource(unsigned ficf, unsigned mid);
    size_t getFramesize() const { return m_framesize; }
    const std::vector<PuncturingRule> &get_rules() const { return m_rules; }
    void loadFicData(const Buffer &fic) { m_buffer = fic; }
    int process(Buffer *outputData) override;
    const char *name() override { return "FicSource"; }

private:
    size_t m_framesize = 0;
    Buffer m_buffer;
    std::vector<PuncturingRule> m_rules;
};

// reference src/SubchannelSource.h, .cpp:70-1068: one MST sub-channel; protection profile -> rules
class SubchannelSource : public ModInput {
public:
    SubchannelSource(uint16_t sad, uint16_t stl, uint8_t tpl);
    size_t startAddress() const { return m_start_address; }
    size_t framesize() const { return m_framesize; }
    size_t framesizeCu() const;   // capacity units of 64 bits; throws for a profile without a size
    size_t bitrate() const { return m_framesize / 3; }
    size_t protection() const { return m_protection; }
    size_t protectionForm() const { return (m_protection >> 5) & 1; }
    size_t protectionLevel() const { return protectionForm() ? (m_protection & 3) + 1 : (m_protection & 7) + 1; }
    size_t protectionOption() const { return protectionForm() ? (m_protection >> 2) & 7 : 0; }
    const std::vector<PuncturingRule> &get_rules() const { return m_rules; }
    void loadSubchannelData(Buffer &&data) { m_buffer = std::move(data); }
    int process(Buffer *outputData) override;
    const char *name() override { return "SubchannelSource"; }

private:
    size_t m_start_address, m_framesize, m_protection, m_framesize_cu = 0xffff;
    Buffer m_buffer;
    std::vector<PuncturingRule> m_rules;
};

// reference src/EtiReader.h:51-74
class EtiSource {
public:
    virtual ~EtiSource() = default;
    virtual unsigned getMode() = 0;
    virtual unsigned getFp() = 0;
    virtual unsigned getFct() = 0;
    virtual std::shared_ptr<FicSource> &getFic() { return myFicSource; }
    virtual const std::vector<std::shared_ptr<SubchannelSource>> getSubchannels() const = 0;

protected:
    std::shared_ptr<FicSource> myFicSource;
};

// reference src/EtiReader.h:95-150, .cpp:93-284: the raw ETI(NI) byte-stream state machine
// (SYNC, FC, STC x NST, EOH, FIC, MST, EOF, TIST, padding to 6144 bytes).  Input may be cut anywhere.
class EtiReader : public EtiSource {
public:
    explicit EtiReader(double &tist_offset_s);
    // consumes as much of dataIn as forms complete fields; returns the number of bytes consumed
    int loadEtiData(const Buffer &dataIn);
    unsigned getMode() override;
    unsigned getFp() override;
    unsigned getFct() override;
    const std::vector<std::shared_ptr<SubchannelSource>> getSubchannels() const override { return mySources; }

private:
    enum class State { Sync, Fc, Nst, Eoh, Fic, Subch, Eof, Tist, Pad };
    State m_state = State::Sync;
    size_t m_remaining = 0;                    // bytes of the 6144-byte frame not yet consumed
    bool m_resync = false;                     // after a refused frame that lost the alignment: State::Sync searches for FSYNC
    bool m_fc_valid = false;
    unsigned m_fct = 0, m_ficf = 0, m_nst = 0, m_fp = 0, m_mid = 0;
    std::vector<uint8_t> m_stc;                // raw STC words of the current layout
    std::vector<std::shared_ptr<SubchannelSource>> mySources;
};

// reference src/InputFileReader.h, .cpp:40-288: an ETI(NI) file in one of the three layouts of
// doc/README-Fileinput -- raw (6144-byte frames), streamed (u16 length + frame) or framed (u32 frame
// count, then u16 length + frame) -- delivered as 6144-byte frames padded with 0x55.
class InputFileReader {
public:
    enum class EtiStreamType { None, Raw, Streamed, Framed };
    int Open(const std::string &filename, bool loop);      // 0, or -1 (cannot open / format not recognised)
    int GetNextFrame(void *buffer);                         // 6144, 0 at the end of the file, -1 on error
    std::string GetPrintableInfo() const;
    EtiStreamType type() const { return m_type; }

private:
    int identify();
    std::vector<uint8_t> m_data;       // the file (ETI files are small: 256 kB per second)
    size_t m_pos = 0, m_start = 0;     // read position; offset of the first frame (length field)
    bool m_loop = false;
    EtiStreamType m_type = EtiStreamType::None;
    std::string m_filename;
};

class FrameMultiplexerError : public std::runtime_error {
public:
    explicit FrameMultiplexerError(const std::string &m) : std::runtime_error(m) {}
};

// reference src/FrameMultiplexer.h, .cpp:36-92: one CIF = padding PRBS with the sub-channels on top
class FrameMultiplexer : public ModMux {
public:
    explicit FrameMultiplexer(const EtiSource &etiSource) : m_etiSource(etiSource) {}
    int process(std::vector<Buffer *> dataIn, Buffer *dataOut) override;
    const char *name() override { return "FrameMultiplexer"; }

private:
    const EtiSource &m_etiSource;
};

// reference src/BlockPartitioner.h, .cpp:36-140: collects the FIC and CIF of 4 / 1 / 1 / 2 ETI frames
// (modes I..IV) into one transmission frame of hot-path input; non-zero return on the last one only
class BlockPartitioner : public ModMux, public ModMetadata {
public:
    explicit BlockPartitioner(unsigned mode);
    int process(std::vector<Buffer *> dataIn, Buffer *dataOut) override;
    const char *name() override { return "BlockPartitioner"; }
    meta_vec_t process_metadata(const meta_vec_t &metadataIn) override;

private:
    size_t m_ficSize = 0, m_cifCount = 0, m_cifNb = 0;
    static constexpr size_t m_cifSize = 864 * 8;
    meta_vec_t m_meta;
};

// The sub-graph of src/DabModulator.cpp:131-139,281-385 as one object: push ETI frames, get the
// hot-path input of every completed transmission frame.  Modulation starts at the first frame with
// FP == 0 (src/DabMod.cpp:684-693).
class EtiFrontend {
public:
    explicit EtiFrontend(unsigned mode);
    // one 6144-byte ETI(NI) frame; returns true when `tf` now holds a transmission frame's input
    bool push(const uint8_t *frame6144, Buffer &tf);
    EtiReader &reader() { return m_reader; }

private:
    struct Sub;
    void build();
    unsigned m_mode;
    double m_tist_offset = 0.0;
    EtiReader m_reader;
    bool m_started = false;
    std::shared_ptr<FicSource> m_fic;
    std::unique_ptr<PrbsGenerator> m_cifPrbs, m_ficPrbs;
    std::unique_ptr<ConvEncoder> m_ficConv;
    std::unique_ptr<PuncturingEncoder> m_ficPunc;
    std::unique_ptr<FrameMultiplexer> m_cifMux;
    std::unique_ptr<BlockPartitioner> m_cifPart;
    std::vector<std::shared_ptr<Sub>> m_subs;
    Buffer m_prbs, m_f0, m_f1, m_f2, m_f3, m_cif, m_part;
};
