// host_selftest.cpp -- driver for the C++ host mirror.
//
//   host_selftest cpu
//       plumbing semantics without a GPU: Buffer, arity assertions, the
//       0-return convention, PipelinedModCodec latency and metadata delay.
//   host_selftest gpu <mode> <bits.bin> <nframes> <graph_out.iq> <chain_out.iq> [normalise]
//       builds the inner flowgraph the way src/DabModulator.cpp:385-419 wires it
//       (cifMap -> cifFreq -> cifDiff(+cifRef) -> cifSig(+cifNull) -> cifOfdm ->
//       cifGain -> cifGuard -> cifFilter -> output) from the drop-in stages, feeds
//       <nframes> blocks of hot-path input, and writes what reaches the sink; then
//       runs the same frames through the single DabGpuChain plugin.
//   host_selftest cfg4 <bits.bin> <nframes> <graph_out.iq> <poly.coef> <output rate>
//       the same graph continued the way src/DabModulator.cpp:403-406 continues it for an SDR sink
//       with a resampler and a predistorter: ... cifFilter -> cifRes -> cifPoly -> output (Mode I).
//   host_selftest memlesspoly <frame.iq> <out prefix> <format-1 file> <format-2 file> <identity file>
//       the MemlessPoly drop-in fed from coefficient FILES and through its remote-control
//       parameters ncoefs / coefs / coeffile (reference src/MemlessPoly.cpp:145-232, :413-470); writes
//       <prefix>.poly.iq, .lut.iq, .rc.iq, .identity.iq, .invalid.iq
#include "Flowgraph.h"
#include "Frontend.h"
#include "GpuStages.h"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <variant>

namespace {

int g_checks = 0;
#define CHECK(cond)                                                                            \
    do {                                                                                       \
        ++g_checks;                                                                            \
        if (!(cond)) {                                                                         \
            std::fprintf(stderr, "CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);       \
            std::exit(1);                                                                      \
        }                                                                                      \
    } while (0)

// feeds successive blocks of a byte vector (the role of cifPart / InputMemory)
class BlockSource : public ModInput {
public:
    BlockSource(std::vector<uint8_t> data, size_t block) : m_data(std::move(data)), m_block(block) {}
    int process(Buffer *out) override
    {
        if (m_pos + m_block > m_data.size()) return 0;
        out->setData(m_data.data() + m_pos, m_block);
        m_pos += m_block;
        return static_cast<int>(m_block);
    }
    const char *name() override { return "BlockSource"; }

private:
    std::vector<uint8_t> m_data;
    size_t m_block, m_pos = 0;
};

// appends everything it receives to a file (the role of OutputMemory + OutputFile)
class FileSink : public ModOutput {
public:
    explicit FileSink(const std::string &path) : m_f(path, std::ios::binary) {}
    int process(Buffer *in) override
    {
        m_f.write(static_cast<const char *>(in->getData()), static_cast<std::streamsize>(in->getLength()));
        ++frames;
        return static_cast<int>(in->getLength());
    }
    const char *name() override { return "FileSink"; }
    int frames = 0;

private:
    std::ofstream m_f;
};

class AddOne : public ModCodec {
public:
    int process(Buffer *const in, Buffer *out) override
    {
        out->setLength(in->getLength());
        for (size_t i = 0; i < in->getLength(); ++i)
            static_cast<uint8_t *>(out->getData())[i] = static_cast<uint8_t>((*in)[i] + 1);
        return 1;
    }
    const char *name() override { return "AddOne"; }
};

class EveryOther : public ModCodec {  // returns 0 on odd calls, like BlockPartitioner in modes with >1 frame/TF
public:
    int process(Buffer *const in, Buffer *out) override
    {
        *out = *in;
        return (m_n++ & 1) ? static_cast<int>(out->getLength()) : 0;
    }
    const char *name() override { return "EveryOther"; }

private:
    int m_n = 0;
};

class PipeDouble : public PipelinedModCodec {
public:
    PipeDouble() { start_pipeline_thread(); }
    ~PipeDouble() override { stop_pipeline_thread(); }
    const char *name() override { return "PipeDouble"; }

protected:
    int internal_process(Buffer *const in, Buffer *out) override
    {
        out->setLength(in->getLength());
        for (size_t i = 0; i < in->getLength(); ++i)
            static_cast<uint8_t *>(out->getData())[i] = static_cast<uint8_t>((*in)[i] * 2);
        return 1;
    }
};

class CountSink : public ModOutput {
public:
    int process(Buffer *in) override
    {
        ++frames;
        last = in->getLength() ? (*in)[0] : -1;
        return 1;
    }
    const char *name() override { return "CountSink"; }
    int frames = 0, last = -1;
};

int run_cpu()
{
    // Buffer: alignment, growth keeps contents, shrink keeps the allocation
    Buffer b(5, "abcde");
    CHECK(b.getLength() == 5 && reinterpret_cast<uintptr_t>(b.getData()) % 32 == 0);
    void *p0 = b.getData();
    b.setLength(3);
    CHECK(b.getData() == p0 && b.getLength() == 3);
    b.setLength(4096);
    CHECK(std::memcmp(b.getData(), "abc", 3) == 0 && reinterpret_cast<uintptr_t>(b.getData()) % 32 == 0);
    Buffer c;
    c = b;
    c += Buffer(2, "xy");
    CHECK(c.getLength() == 4098 && c[4096] == 'x');
    Buffer d(std::move(c));
    CHECK(c.getLength() == 0 && c.getData() == nullptr && d.getLength() == 4098);
    bool threw = false;
    try { (void)d[5000]; } catch (const std::out_of_range &) { threw = true; }
    CHECK(threw);

    // arity assertions throw std::runtime_error naming the plugin
    AddOne a1;
    Buffer x(1, "a"), y;
    threw = false;
    ModPlugin *as_plugin = &a1;
    try { as_plugin->process({&x, &x}, {&y}); } catch (const std::runtime_error &e) {
        threw = std::string(e.what()).find("AddOne") != std::string::npos;
    }
    CHECK(threw);

    // flowgraph: a node returning 0 stops the walk for that round
    {
        auto src = std::make_shared<BlockSource>(std::vector<uint8_t>{1, 2, 3, 4}, 1);
        auto eo = std::make_shared<EveryOther>();
        auto add = std::make_shared<AddOne>();
        auto sink = std::make_shared<CountSink>();
        Flowgraph fg;
        fg.connect(src, eo);
        fg.connect(eo, add);
        fg.connect(add, sink);
        CHECK(fg.run() == false && sink->frames == 0);
        CHECK(fg.run() == true && sink->frames == 1 && sink->last == 3);
        CHECK(fg.run() == false);
        CHECK(fg.run() == true && sink->frames == 2 && sink->last == 5);
        CHECK(fg.run() == false);  // source exhausted
    }
    // a producer wired after its consumer (tii -> cifSig, src/DabModulator.cpp:392-395): the consumer
    // moves behind it in the run order (reference src/Flowgraph.cpp:299-308)
    {
        class Concat : public ModMux {
        public:
            int process(std::vector<Buffer *> in, Buffer *out) override
            {
                out->setLength(0);
                for (Buffer *b : in) *out += *b;
                return static_cast<int>(out->getLength());
            }
            const char *name() override { return "Concat"; }
        };
        auto s1 = std::make_shared<BlockSource>(std::vector<uint8_t>{1, 2}, 1);
        auto s2 = std::make_shared<BlockSource>(std::vector<uint8_t>{7, 8}, 1);
        auto late = std::make_shared<AddOne>();
        auto mux = std::make_shared<Concat>();
        class LastByteSink : public ModOutput {
        public:
            int process(Buffer *in) override
            {
                len = in->getLength();
                last = len ? (*in)[len - 1] : -1;
                return 1;
            }
            const char *name() override { return "LastByteSink"; }
            size_t len = 0;
            int last = -1;
        };
        auto sink = std::make_shared<LastByteSink>();
        Flowgraph fg;
        fg.connect(s1, mux);
        fg.connect(s2, late);
        fg.connect(late, mux);     // `late` is listed after `mux`: mux must move to the end
        fg.connect(mux, sink);
        CHECK(fg.run() == true && sink->len == 2 && sink->last == 8);    // {1, 7+1}
        CHECK(fg.run() == true && sink->len == 2 && sink->last == 9);    // {2, 8+1}
    }
    // pipelined stage: call i returns frame i-1, call 0 returns nothing, input buffer is stolen
    {
        PipeDouble pd;
        Buffer in(1, "\x03"), out;
        CHECK(pd.process(&in, &out) == 0 && in.getLength() == 0 && in.getData() == nullptr);
        in.setData("\x05", 1);
        CHECK(pd.process(&in, &out) == 1 && out[0] == 6);
        in.setData("\x07", 1);
        CHECK(pd.process(&in, &out) == 1 && out[0] == 10);
        flowgraph_metadata m0, m1;
        m0.ts.fct = 10;
        m1.ts.fct = 11;
        CHECK(pd.process_metadata({m0}).empty());
        meta_vec_t r = pd.process_metadata({m1});
        CHECK(r.size() == 1 && r[0].ts.fct == 10);
    }
    // EtiReader: the byte stream may be cut anywhere (reference src/EtiReader.cpp:93-270 returns the
    // number of bytes it could use); two frames fed in pieces of 7 bytes parse like two whole frames
    {
        std::vector<uint8_t> eti(2 * 6144, 0x55);
        for (int f = 0; f < 2; ++f) {
            uint8_t *p = eti.data() + f * 6144;
            const uint8_t head[] = {0xFF, 0x07, 0x3A, 0xB6, static_cast<uint8_t>(8 + f), 0x81,
                                    static_cast<uint8_t>((f << 5) | (1 << 3)), 0x40,
                                    0x00, 0x00, static_cast<uint8_t>(0x22 << 2), 48,     // SAD 0, TPL 0x22, STL 48
                                    0, 0, 0, 0};
            std::memcpy(p, head, sizeof head);
            for (int i = 0; i < 96 + 384; ++i) p[16 + i] = static_cast<uint8_t>(i * 7 + f);
        }
        double off = 0;
        EtiReader whole(off), pieces(off);
        Buffer a(6144, eti.data());
        CHECK(whole.loadEtiData(a) == 6144 && whole.getFct() == 8 && whole.getFp() == 0 && whole.getMode() == 1);
        CHECK(whole.getSubchannels().size() == 1 && whole.getSubchannels()[0]->framesize() == 384 &&
              whole.getSubchannels()[0]->framesizeCu() == 96);
        size_t fed = 0;
        std::vector<uint8_t> pending;
        while (fed < eti.size()) {
            const size_t n = std::min<size_t>(7, eti.size() - fed);
            pending.insert(pending.end(), eti.begin() + fed, eti.begin() + fed + n);
            fed += n;
            Buffer b(pending.size(), pending.data());
            const int used = pieces.loadEtiData(b);
            pending.erase(pending.begin(), pending.begin() + used);
        }
        Buffer f0, f1;
        CHECK(pieces.getFct() == 9 && pieces.getFp() == 1 && pending.empty());
        pieces.getFic()->process(&f0);
        CHECK(f0.getLength() == 96 && f0[0] == 1 && f0[95] == static_cast<uint8_t>(95 * 7 + 1));
        pieces.getSubchannels()[0]->process(&f1);
        CHECK(f1.getLength() == 384 && f1[0] == static_cast<uint8_t>(96 * 7 + 1));
    }
    // pipelined stage inside a graph: N rounds in, N-1 frames out
    {
        auto src = std::make_shared<BlockSource>(std::vector<uint8_t>{1, 2, 3, 4, 5}, 1);
        auto pd = std::make_shared<PipeDouble>();
        auto sink = std::make_shared<CountSink>();
        Flowgraph fg;
        fg.connect(src, pd);
        fg.connect(pd, sink);
        int ok = 0;
        for (int i = 0; i < 5; ++i) ok += fg.run() ? 1 : 0;
        CHECK(ok == 4 && sink->frames == 4 && sink->last == 8);
    }
    // EtiReader fed a byte stream in pieces: a frame whose sub-channel list overruns the 6144-byte frame is refused,
    // and the part of it that lies beyond the offending buffer is skipped -- the next frame start is a real one
    {
        auto frame = [](unsigned nst, unsigned stl) {
            std::vector<uint8_t> f(6144, 0x55);
            const uint8_t sync[4] = {0xFF, 0x07, 0x3A, 0xB6};
            std::memcpy(f.data(), sync, 4);
            f[4] = 0; f[5] = static_cast<uint8_t>(0x80 | nst); f[6] = static_cast<uint8_t>(1u << 3); f[7] = 0;   // FCT, FICF|NST, FP|MID=1, FL
            for (unsigned i = 0; i < nst; ++i) {
                uint8_t *st = &f[8 + 4 * i];
                st[0] = static_cast<uint8_t>(i << 2); st[1] = static_cast<uint8_t>(96 * i);
                st[2] = static_cast<uint8_t>((0x22u << 2) | (stl >> 8)); st[3] = static_cast<uint8_t>(stl & 0xff);
            }
            return f;
        };
        const std::vector<uint8_t> bad = frame(2, 1000), good = frame(1, 48);     // 2 x 8000 bytes of payload: overrun
        double tist_offset = 0.0;
        EtiReader rd(tist_offset);
        bool refused = false;
        try { rd.loadEtiData(Buffer(1000, bad.data())); } catch (const std::runtime_error &) { refused = true; }
        CHECK(refused);
        // the remaining 5144 bytes of the bad frame arrive next, together with a good frame: no exception, and the good
        // frame's header is the one in force afterwards
        std::vector<uint8_t> rest(bad.begin() + 1000, bad.end());
        rest.insert(rest.end(), good.begin(), good.end());
        bool threw2 = false;
        int used = 0;
        try { used = rd.loadEtiData(Buffer(rest.size(), rest.data())); } catch (const std::exception &) { threw2 = true; }
        CHECK(!threw2 && used == static_cast<int>(rest.size()));
        CHECK(rd.getSubchannels().size() == 1 && rd.getSubchannels()[0]->framesize() == 48 * 8);
        // the other case: the offending frame ENDS inside the buffer that is refused (the exception drops the rest of that
        // buffer, the first 100 bytes of the following frame with it), so the next call starts in the middle of a frame:
        // the reader looks for ERR + FSYNC and picks up the first whole frame
        EtiReader rd2(tist_offset);
        std::vector<uint8_t> first(bad);
        first.insert(first.end(), good.begin(), good.begin() + 100);
        refused = false;
        try { rd2.loadEtiData(Buffer(first.size(), first.data())); } catch (const std::runtime_error &) { refused = true; }
        CHECK(refused);
        std::vector<uint8_t> next(good.begin() + 100, good.end());           // 6044 bytes without a frame start
        std::vector<uint8_t> good2 = frame(1, 48);
        const uint8_t sync2[4] = {0xFF, 0xF8, 0xC5, 0x49};
        std::memcpy(good2.data(), sync2, 4);
        next.insert(next.end(), good2.begin(), good2.end());
        threw2 = false;
        try { used = rd2.loadEtiData(Buffer(next.size(), next.data())); } catch (const std::exception &) { threw2 = true; }
        CHECK(!threw2 && used == static_cast<int>(next.size()));
        CHECK(rd2.getSubchannels().size() == 1 && rd2.getSubchannels()[0]->framesize() == 48 * 8);
    }
    std::printf("host_selftest cpu: OK (%d checks)\n", g_checks);
    return 0;
}

std::vector<uint8_t> read_all(const std::string &path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}


// one frame through a pipelined stage: the frame comes back on the following call (reference
// src/ModPlugin.cpp:90-154), so every frame is pushed twice
void through_pipeline(ModCodec &stage, const std::vector<uint8_t> &frame, const std::string &path)
{
    Buffer out;
    for (int k = 0; k < 2; ++k) {
        Buffer in(frame.size(), frame.data());
        stage.process(&in, &out);
    }
    std::ofstream f(path, std::ios::binary);
    f.write(static_cast<const char *>(out.getData()), static_cast<std::streamsize>(out.getLength()));
}

std::string slurp(const std::string &path)
{
    std::ifstream f(path);
    return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int run_memlesspoly(int argc, char **argv)
{
    if (argc < 7) {
        std::fprintf(stderr, "usage: host_selftest memlesspoly <frame.iq> <out prefix> <fmt1 file> <fmt2 file> <identity file>\n");
        return 2;
    }
    const std::vector<uint8_t> frame = read_all(argv[2]);
    const std::string prefix = argv[3], f1 = argv[4], f2 = argv[5], fid = argv[6];
    std::string coefs_file = f1;
    MemlessPoly poly(coefs_file, 4);
    CHECK(std::string(poly.name()) == "MemlessPoly" && poly.get_rc_name() == "memlesspoly");
    CHECK(poly.get_parameter("coeffile") == f1 && poly.get_parameter("ncoefs") == "5");
    // format 1: "1 / 5 / 5 AM values / 5 PM values", one per line -- what python/dpd/Adapt.py:142-155 writes
    {
        std::stringstream ss(poly.get_parameter("coefs"));
        int fmt = 0, n = 0;
        ss >> fmt >> n;
        CHECK(fmt == 1 && n == 5);
        std::stringstream file(slurp(f1));
        int ffmt, fn;
        file >> ffmt >> fn;
        for (int i = 0; i < 10; ++i) {
            float a, b;
            ss >> a;
            file >> b;
            CHECK(a == b);
        }
    }
    through_pipeline(poly, frame, prefix + ".poly.iq");
    // coeffile: load another file (format 2, look-up table) while running
    poly.set_parameter("coeffile", f2);
    CHECK(poly.get_parameter("coeffile") == f2);
    CHECK(poly.get_parameter("coefs").substr(0, 5) == "2\n32\n");
    CHECK(std::get<uint64_t>(poly.get_all_values().at("ncoefs").v) == 5);     // the AM vector keeps its size
    through_pipeline(poly, frame, prefix + ".lut.iq");
    // coefs: coefficients as a string in the file's own format; written back to the current coefficient file
    const std::string rc = "1\n5\n0.9\n0.1\n-0.02\n0.004\n0.0005\n0.01\n-0.03\n0.002\n0.001\n-0.0002\n";
    poly.set_parameter("coefs", rc);
    CHECK(slurp(f2) == rc);
    CHECK(poly.get_parameter("coeffile") == f2);
    through_pipeline(poly, frame, prefix + ".rc.iq");
    // the identity file the reference ships (python/poly.coef)
    poly.set_parameter("coeffile", fid);
    through_pipeline(poly, frame, prefix + ".identity.iq");
    // errors: read-only and unknown parameters, wrong coefficient count, missing file -- ParameterError, settings kept
    auto throws = [&](const std::string &p, const std::string &v) {
        try { poly.set_parameter(p, v); } catch (const ParameterError &) { return true; }
        return false;
    };
    CHECK(throws("ncoefs", "5"));
    CHECK(throws("nonsense", "1"));
    CHECK(throws("coefs", "1\n4\n1\n0\n0\n0\n0\n0\n0\n0\n"));
    CHECK(throws("coefs", "1\n5\n1\n0\n0\n"));                                    // EOF before ten values
    CHECK(throws("coeffile", prefix + ".does-not-exist"));
    bool threw = false;
    try { (void)poly.get_parameter("nonsense"); } catch (const ParameterError &) { threw = true; }
    CHECK(threw);
    CHECK(poly.get_parameter("coeffile") == fid);
    // unknown format id: no exception, settings invalid, frames pass through (src/MemlessPoly.cpp:227-231, :404-408)
    poly.set_parameter("coefs", "3\n1\n1.0\n");
    CHECK(poly.get_parameter("coefs").empty());
    through_pipeline(poly, frame, prefix + ".invalid.iq");
    // a constructor given a missing file throws like the reference's
    threw = false;
    try {
        std::string missing = prefix + ".does-not-exist";
        MemlessPoly bad(missing, 1);
    } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
    std::printf("host_selftest memlesspoly: OK (%d checks)\n", g_checks);
    return 0;
}

int run_cfg4(int argc, char **argv)
{
    if (argc < 7) {
        std::fprintf(stderr, "usage: host_selftest cfg4 <bits.bin> <nframes> <graph.iq> <poly.coef> <output rate>\n");
        return 2;
    }
    const std::vector<uint8_t> bits = read_all(argv[2]);
    const size_t nframes = static_cast<size_t>(std::atoi(argv[3]));
    std::string coefFile = argv[5];
    const size_t outputRate = static_cast<size_t>(std::atol(argv[6]));
    const unsigned mode = 1;
    const size_t nbSymbols = 76, nbCarriers = 1536, spacing = 2048, nullSize = 2656, symSize = 2552;
    const size_t block = (nbSymbols - 1) * nbCarriers / 4;
    if (bits.size() < nframes * block) throw std::runtime_error("bits file too short");
    GainMode gainMode = GainMode::GAIN_VAR;
    float digitalGain = 1.0f, gainmodeVariance = 4.0f, cfrClip = 1.0f, cfrErrorClip = 1.0f;
    const float normalise = 1.0f / 50000.0f;                  // what every SDR sink sets (src/DabMod.cpp:293,304)
    bool enableCfr = false;
    size_t windowOverlap = 0;
    std::string tapsFile = "default";

    auto cifPart = std::make_shared<BlockSource>(bits, block);
    auto cifMap = std::make_shared<QpskSymbolMapper>(nbCarriers, false);
    auto cifRef = std::make_shared<PhaseReference>(mode, false);
    auto cifFreq = std::make_shared<FrequencyInterleaver>(mode, false);
    auto cifDiff = std::make_shared<DifferentialModulator>(nbCarriers, false);
    auto cifNull = std::make_shared<NullSymbol>(nbCarriers, sizeof(complexf));
    auto cifSig = std::make_shared<SignalMultiplexer>();
    auto cifOfdm = std::make_shared<OfdmGeneratorCF32>(1 + nbSymbols, nbCarriers, spacing, enableCfr, cfrClip,
                                                       cfrErrorClip);
    auto cifGain = std::make_shared<GainControl>(spacing, gainMode, digitalGain, normalise, gainmodeVariance);
    auto cifGuard = std::make_shared<GuardIntervalInserter>(nbSymbols, spacing, nullSize, symSize, windowOverlap,
                                                            FFTEngine::FFTW);
    auto cifFilter = std::make_shared<FIRFilter>(tapsFile);
    // src/DabModulator.cpp:265-268: resolution = m_spacing;  :256-262: MemlessPoly(polyCoefFilename, polyNumThreads)
    auto cifRes = std::make_shared<Resampler>(2048000, outputRate, spacing);
    {
        // a ratio no kernel covers (M = 5: the reference's own hop size would not divide a frame) is refused by the
        // constructor, not by the first process() call
        bool threw = false;
        try { Resampler bad(2048000, 2457600, spacing); } catch (const std::runtime_error &) { threw = true; }
        CHECK(threw);
        Resampler down(2048000, 1024000, spacing);            // down-sampling constructs
    }
    auto cifPoly = std::make_shared<MemlessPoly>(coefFile, 4);
    auto output = std::make_shared<FileSink>(argv[4]);

    Flowgraph fg(true);
    fg.connect(cifPart, cifMap);
    fg.connect(cifMap, cifFreq);
    fg.connect(cifRef, cifDiff);
    fg.connect(cifFreq, cifDiff);
    fg.connect(cifNull, cifSig);
    fg.connect(cifDiff, cifSig);
    fg.connect(cifSig, cifOfdm);
    fg.connect(cifOfdm, cifGain);
    fg.connect(cifGain, cifGuard);
    fg.connect(cifGuard, cifFilter);
    fg.connect(cifFilter, cifRes);                            // :403-406
    fg.connect(cifRes, cifPoly);
    fg.connect(cifPoly, output);
    int rounds_ok = 0;
    for (size_t i = 0; i < nframes; ++i) rounds_ok += fg.run() ? 1 : 0;
    std::printf("cfg4 stage graph: %zu rounds, %d reached the sink (%d frames written)\n", nframes, rounds_ok,
                output->frames);
    return 0;
}

int run_gpu(int argc, char **argv)
{
    if (argc < 7) {
        std::fprintf(stderr, "usage: host_selftest gpu <mode> <bits.bin> <nframes> <graph.iq> <chain.iq> [normalise] [chain.s16] [tii comb,pattern | -] [cfr clip,errorclip]\n");
        return 2;
    }
    const unsigned mode = static_cast<unsigned>(std::atoi(argv[2]));
    const std::vector<uint8_t> bits = read_all(argv[3]);
    const size_t nframes = static_cast<size_t>(std::atoi(argv[4]));
    const float normalise = argc > 7 ? static_cast<float>(std::atof(argv[7])) : 1.0f / 50000.0f;
    // geometry, as DabModulator::setMode (src/DabModulator.cpp:84-122)
    size_t nbSymbols = 76, nbCarriers = 1536, spacing = 2048, nullSize = 2656, symSize = 2552;
    if (mode == 2) { nbCarriers = 384; spacing = 512; nullSize = 664; symSize = 638; }
    if (mode == 3) { nbSymbols = 153; nbCarriers = 192; spacing = 256; nullSize = 345; symSize = 319; }
    if (mode == 4) { nbCarriers = 768; spacing = 1024; nullSize = 1328; symSize = 1276; }
    const size_t block = (nbSymbols - 1) * nbCarriers / 4;
    if (bits.size() < nframes * block) throw std::runtime_error("bits file too short");

    // the settings the stages hold references into (mod_settings_t in the reference)
    GainMode gainMode = GainMode::GAIN_VAR;
    float digitalGain = 1.0f, gainmodeVariance = 4.0f, cfrClip = 1.0f, cfrErrorClip = 1.0f;
    bool enableCfr = false;
    size_t windowOverlap = 0;
    std::string tapsFile = "default";
    // optional: TII "comb,pattern" (SURVEY 8 f-4), wired like src/DabModulator.cpp:178-190,392-395
    tii_config_t tiiConfig;
    if (argc > 9 && std::string(argv[9]) != "-") {
        if (std::sscanf(argv[9], "%d,%d", &tiiConfig.comb, &tiiConfig.pattern) != 2)
            throw std::runtime_error("tii argument: comb,pattern");
        tiiConfig.enable = true;
    }
    // optional: crest-factor reduction "clip,errorclip" 
    if (argc > 10) {
        if (std::sscanf(argv[10], "%f,%f", &cfrClip, &cfrErrorClip) != 2)
            throw std::runtime_error("cfr argument: clip,errorclip");
        enableCfr = true;
    }

    {
        auto cifPart = std::make_shared<BlockSource>(bits, block);
        auto cifMap = std::make_shared<QpskSymbolMapper>(nbCarriers, false);
        auto cifRef = std::make_shared<PhaseReference>(mode, false);
        auto cifFreq = std::make_shared<FrequencyInterleaver>(mode, false);
        auto cifDiff = std::make_shared<DifferentialModulator>(nbCarriers, false);
        auto cifNull = std::make_shared<NullSymbol>(nbCarriers, sizeof(complexf));
        auto cifSig = std::make_shared<SignalMultiplexer>();
        auto cifOfdm = std::make_shared<OfdmGeneratorCF32>(1 + nbSymbols, nbCarriers, spacing, enableCfr,
                                                           cfrClip, cfrErrorClip);
        auto cifGain = std::make_shared<GainControl>(spacing, gainMode, digitalGain, normalise, gainmodeVariance);
        auto cifGuard = std::make_shared<GuardIntervalInserter>(nbSymbols, spacing, nullSize, symSize,
                                                                windowOverlap, FFTEngine::FFTW);
        auto cifFilter = std::make_shared<FIRFilter>(tapsFile);
        auto output = std::make_shared<FileSink>(argv[5]);

        Flowgraph fg(true);
        fg.connect(cifPart, cifMap);
        fg.connect(cifMap, cifFreq);
        fg.connect(cifRef, cifDiff);
        fg.connect(cifFreq, cifDiff);
        fg.connect(cifNull, cifSig);
        fg.connect(cifDiff, cifSig);
        std::shared_ptr<TII> tii;
        if (tiiConfig.enable) {
            tii = std::make_shared<TII>(mode, tiiConfig, false);
            auto tiiRef = std::make_shared<PhaseReference>(mode, false);
            fg.connect(tiiRef, tii);
            fg.connect(tii, cifSig);
        }
        fg.connect(cifSig, cifOfdm);
        fg.connect(cifOfdm, cifGain);
        fg.connect(cifGain, cifGuard);
        fg.connect(cifGuard, cifFilter);
        fg.connect(cifFilter, output);
        int rounds_ok = 0;
        for (size_t i = 0; i < nframes; ++i) rounds_ok += fg.run() ? 1 : 0;
        // remote control through the stage interface
        cifGain->set_parameter("mode", "VAR");
        if (cifGain->get_parameter("mode") != "var") throw std::runtime_error("RC mode round trip failed");
        if (enableCfr) {
            // RC statistics of the OfdmGenerator drop-in (reference src/OfdmGenerator.cpp:419-451)
            auto *ofdm = dynamic_cast<OfdmGeneratorCF32 *>(cifOfdm.get());
            std::printf("ofdm clip_stats: %s\nofdm papr: %s\n", ofdm->get_parameter("clip_stats").c_str(),
                        ofdm->get_parameter("papr").c_str());
            bool threw = false;
            try { ofdm->set_parameter("papr", "1"); } catch (const ParameterError &) { threw = true; }
            if (!threw) throw std::runtime_error("papr must be read-only");
            if (ofdm->get_parameter("cfr") != "1") throw std::runtime_error("RC cfr read-back failed");
        }
        if (tii) {
            const int comb0 = tiiConfig.comb;      // the stage holds a reference into tiiConfig
            tii->set_parameter("comb", "7");
            if (tii->get_parameter("comb") != "7" || std::string(tii->name()).find("c:7") == std::string::npos)
                throw std::runtime_error("RC tii comb round trip failed");
            bool threw = false;
            try { tii->set_parameter("pattern", "70"); } catch (const TIIError &) { threw = true; }
            if (!threw) throw std::runtime_error("TII pattern 70 must be rejected");
            tii->set_parameter("comb", std::to_string(comb0));
        }
        std::printf("stage graph: %zu rounds, %d reached the sink (%d frames written)\n", nframes, rounds_ok,
                    output->frames);
    }
    {
        DabGpuChain::Settings s;
        s.dabMode = mode;
        s.normalise = normalise;
        s.filterTapsFilename = "default";
        s.tiiConfig = tiiConfig;
        s.enableCfr = enableCfr;
        s.cfrClip = cfrClip;
        s.cfrErrorClip = cfrErrorClip;
        auto cifPart = std::make_shared<BlockSource>(bits, block);
        auto chain = std::make_shared<DabGpuChain>(s);
        auto output = std::make_shared<FileSink>(argv[6]);
        Flowgraph fg;
        fg.connect(cifPart, chain);
        fg.connect(chain, output);
        for (size_t i = 0; i < nframes; ++i) fg.run();
        std::printf("chain plugin: %d frames written\n", output->frames);
    }
    if (argc > 8) {
        // file-output shape of the reference (src/DabModulator.cpp:395-419, normalise 1.0):
        // chain -> FormatConverter("s16") -> sink
        DabGpuChain::Settings s;
        s.dabMode = mode;
        s.normalise = 1.0f;
        s.filterTapsFilename = "default";
        auto cifPart = std::make_shared<BlockSource>(bits, block);
        auto chain = std::make_shared<DabGpuChain>(s);
        auto conv = std::make_shared<FormatConverter>(false, "s16");
        auto output = std::make_shared<FileSink>(argv[8]);
        Flowgraph fg;
        fg.connect(cifPart, chain);
        fg.connect(chain, conv);
        fg.connect(conv, output);
        for (size_t i = 0; i < nframes; ++i) fg.run();
        std::printf("s16 chain: %d frames written, last frame clipped=%zu\n", output->frames,
                    conv->get_num_clipped_samples());
        if (FormatConverter::get_format_size("u8") != 2) throw std::runtime_error("get_format_size");
    }
    return 0;
}

// ---- the production plugin's remote-control surface and metadata semantics
//   host_selftest chainrc <bits file> <nframes> <out.iq> <drops> [NAME,PARAM,VALUE@FRAME]...
// One DabGpuChain (gain var, "default" FIRFilter taps, predistorter from <dir of out.iq>/poly.coef when that file exists)
// between a source that attaches the frame number as metadata and a sink that reports what arrives with each frame.
// Each action is applied through the RemoteControllable of that NAME just before frame FRAME enters the graph.
class NumberedSource : public ModInput, public ModMetadata {
public:
    NumberedSource(std::vector<uint8_t> data, size_t block) : m_data(std::move(data)), m_block(block) {}
    int process(Buffer *out) override
    {
        if (m_pos + m_block > m_data.size()) return 0;
        out->setData(m_data.data() + m_pos, m_block);
        m_pos += m_block;
        return static_cast<int>(m_block);
    }
    meta_vec_t process_metadata(const meta_vec_t &) override
    {
        flowgraph_metadata md;
        md.ts.fct = m_next++;
        return {md};
    }
    const char *name() override { return "NumberedSource"; }

private:
    std::vector<uint8_t> m_data;
    size_t m_block, m_pos = 0;
    int32_t m_next = 0;
};

class ReportingSink : public ModOutput, public ModMetadata {
public:
    explicit ReportingSink(const std::string &path) : m_f(path, std::ios::binary) {}
    int process(Buffer *in) override
    {
        m_f.write(static_cast<const char *>(in->getData()), static_cast<std::streamsize>(in->getLength()));
        return static_cast<int>(in->getLength());
    }
    meta_vec_t process_metadata(const meta_vec_t &in) override
    {
        std::printf("frame %d carries metadata of", frames++);
        for (const auto &md : in) std::printf(" %d", static_cast<int>(md.ts.fct));
        std::printf("\n");
        return {};
    }
    const char *name() override { return "ReportingSink"; }
    int frames = 0;

private:
    std::ofstream m_f;
};

int run_chainrc(int argc, char **argv)
{
    if (argc < 6) {
        std::fprintf(stderr, "usage: host_selftest chainrc <bits file> <nframes> <out.iq> <drops> [NAME,PARAM,VALUE@FRAME]...\n");
        return 2;
    }
    const std::vector<uint8_t> bits = read_all(argv[2]);
    const size_t nframes = std::strtoul(argv[3], nullptr, 10);
    const std::string out = argv[4];
    DabGpuChain::Settings s;
    s.dabMode = 1;
    s.normalise = 1.0f / 50000.0f;
    s.filterTapsFilename = "default";
    s.emulatePipelineDrops = static_cast<unsigned>(std::strtoul(argv[5], nullptr, 10));
    const std::string coef = out.substr(0, out.find_last_of('/') + 1) + "poly.coef";
    if (std::ifstream(coef)) s.polyCoefFilename = coef;
    // the RC-mutable values live OUTSIDE the plugin, as mod_settings_t does in the reference
    GainMode gainMode = GainMode::GAIN_VAR;
    float digital = 1.0f;
    std::string taps = "default";
    DabGpuChain::LiveSettings live;
    live.gainMode = &gainMode;
    live.digitalGain = &digital;
    live.filterTapsFilename = &taps;
    auto chain = std::make_shared<DabGpuChain>(s, live);
    std::map<std::string, RemoteControllable *> rcs;
    for (RemoteControllable *c : chain->remote_controllables()) rcs[c->get_rc_name()] = c;
    std::string names;
    for (const auto &kv : rcs) names += kv.first + " ";
    std::printf("controllables: %s\n", names.c_str());
    struct Action { size_t at; std::string name, param, value; };
    std::vector<Action> actions;
    for (int i = 6; i < argc; ++i) {
        const std::string a = argv[i];
        const size_t at = a.find_last_of('@'), c1 = a.find(','), c2 = a.find(',', c1 + 1);
        if (at == std::string::npos || c1 == std::string::npos || c2 == std::string::npos) throw std::runtime_error("bad action " + a);
        std::string value = a.substr(c2 + 1, at - c2 - 1);
        if (!value.empty() && value[0] == '<') value = slurp(value.substr(1));         // "<file": the value is the file's text
        actions.push_back({std::strtoul(a.c_str() + at + 1, nullptr, 10), a.substr(0, c1), a.substr(c1 + 1, c2 - c1 - 1), value});
    }
    auto source = std::make_shared<NumberedSource>(bits, chain->input_bytes_per_frame());
    auto sink = std::make_shared<ReportingSink>(out);
    Flowgraph fg;
    fg.connect(source, chain);
    fg.connect(chain, sink);
    for (size_t i = 0; i < nframes; ++i) {
        for (const auto &a : actions)
            if (a.at == i) {
                auto it = rcs.find(a.name);
                if (it == rcs.end()) throw std::runtime_error("no controllable " + a.name);
                it->second->set_parameter(a.param, a.value);
            }
        fg.run();
    }
    for (const auto &a : actions) {
        std::string v = rcs[a.name]->get_parameter(a.param);
        for (auto &c : v) if (c == '\n') c = ' ';
        std::printf("rc %s %s = %s\n", a.name.c_str(), a.param.c_str(), v.c_str());
    }
    std::printf("settings now: digital=%g mode=%d taps=%s\n", digital, static_cast<int>(gainMode), taps.c_str());
    // error conventions of the reference's controllables
    bool threw = false;
    try { rcs["gain"]->set_parameter("nonsense", "1"); } catch (const ParameterError &) { threw = true; }
    CHECK(threw);
    threw = false;
    try { rcs["firfilter"]->set_parameter("ntaps", "3"); } catch (const ParameterError &) { threw = true; }
    CHECK(threw);
    CHECK(rcs.count("ofdm") && rcs.count("guardinterval") && rcs.count("tii"));
    std::printf("chainrc: %d frames written\n", sink->frames);
    return 0;
}

}  // namespace

int main(int argc, char **argv)
{
    try {
        if (argc >= 2 && std::string(argv[1]) == "cpu") return run_cpu();
        if (argc >= 2 && std::string(argv[1]) == "gpu") return run_gpu(argc, argv);
        if (argc >= 2 && std::string(argv[1]) == "cfg4") return run_cfg4(argc, argv);
        if (argc >= 2 && std::string(argv[1]) == "memlesspoly") return run_memlesspoly(argc, argv);
        if (argc >= 2 && std::string(argv[1]) == "chainrc") return run_chainrc(argc, argv);
        std::fprintf(stderr, "usage: host_selftest cpu | gpu ...\n");
        return 2;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "host_selftest: %s\n", e.what());
        return 1;
    }
}
