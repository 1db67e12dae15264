// GpuStages.h -- MI355X drop-ins for ODR-DabMod's hot-path plugins.
//
// Same class names, constructor signatures, name() strings, remote-control
// names/parameters and error behaviour as the reference classes they replace
// (file:line at each class), so that src/DabModulator.cpp:131-419 builds its
// inner flowgraph from them unchanged.  Each process() forwards the host
// Buffers to one dabgpu_*_process entry point of include/dabgpu.h and turns a
// negative status into the std::runtime_error the reference would have thrown.
//
// Per-stage drop-ins pay PCIe both ways per stage; the production shape is
// DabGpuChain, ONE plugin that replaces the sub-graph cifMap..cifPoly and moves
// 28.8 kB in / one IQ frame out per transmission frame.
#pragma once

#include "ModPlugin.h"
#include "RemoteControl.h"

#include <atomic>
#include <deque>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

struct dabgpu_ctx;

enum class GainMode { GAIN_FIX = 0, GAIN_MAX = 1, GAIN_VAR = 2 };  // reference src/GainControl.h:45
// FFTEngine (reference src/ConfigParser.h:39-43).  Inside a reference tree ConfigParser.h defines it -- and includes this
// header, through GainControl.h / TII.h, BEFORE it does: the adapters therefore need the opaque declaration only (a scoped
// enumeration's underlying type is fixed, so it is a complete type), and the definition below exists only where no
// ConfigParser.h is on the include path (the host mirror of this repository).
enum class FFTEngine;
#if !__has_include("ConfigParser.h")
enum class FFTEngine { FFTW, KISS, DEXTER };
#endif

namespace dabgpu_host {
// one device context per stage object; mode derived from the stage's geometry
class Context {
public:
    explicit Context(int mode, int max_frames = 1);
    ~Context();
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    dabgpu_ctx *get() const { return m_ctx; }
    // status -> exception, out_bytes -> Buffer::setLength protocol
    void check(int rc) const;

private:
    dabgpu_ctx *m_ctx = nullptr;
};
int mode_from_carriers(size_t carriers);
int mode_from_spacing(size_t spacing);
}  // namespace dabgpu_host

// reference src/QpskSymbolMapper.h:34, .cpp:39-213
class QpskSymbolMapper : public ModCodec {
public:
    QpskSymbolMapper(size_t carriers, bool fixedPoint);
    int process(Buffer *const dataIn, Buffer *dataOut) override;
    const char *name() override { return "QpskSymbolMapper"; }

private:
    dabgpu_host::Context m_ctx;
};

// reference src/FrequencyInterleaver.h, .cpp:31-145
class FrequencyInterleaver : public ModCodec {
public:
    FrequencyInterleaver(size_t mode, bool fixedPoint);
    int process(Buffer *const dataIn, Buffer *dataOut) override;
    const char *name() override { return "FrequencyInterleaver"; }

private:
    dabgpu_host::Context m_ctx;
};

// reference src/PhaseReference.h:51, .cpp:61-190
class PhaseReference : public ModInput {
public:
    PhaseReference(unsigned int dabmode, bool fixedPoint);
    int process(Buffer *dataOut) override;
    const char *name() override { return "PhaseReference"; }

private:
    dabgpu_host::Context m_ctx;
};

// reference src/DifferentialModulator.h:38, .cpp:45-108
class DifferentialModulator : public ModMux {
public:
    DifferentialModulator(size_t carriers, bool fixedPoint);
    int process(std::vector<Buffer *> dataIn, Buffer *dataOut) override;
    const char *name() override { return "DifferentialModulator"; }

private:
    dabgpu_host::Context m_ctx;
};

// reference src/NullSymbol.cpp:33-57
class NullSymbol : public ModInput {
public:
    NullSymbol(size_t numCarriers, size_t typeSize);
    int process(Buffer *dataOut) override;
    const char *name() override { return "NullSymbol"; }

private:
    size_t m_bytes;
};

// reference src/SignalMultiplexer.cpp:45-71 (pure layout: host concatenation, as in the reference)
class SignalMultiplexer : public ModMux {
public:
    SignalMultiplexer() = default;
    int process(std::vector<Buffer *> dataIn, Buffer *dataOut) override;
    const char *name() override { return "SignalMultiplexer"; }
};

// reference src/OfdmGenerator.h:50-56, .cpp:42-308; crest-factor reduction (f-3) and its RC statistics
// clip_stats / papr (:376-451) included
class OfdmGeneratorCF32 : public ModCodec, public RemoteControllable {
public:
    OfdmGeneratorCF32(size_t nbSymbols, size_t nbCarriers, size_t spacing, bool &enableCfr,
                      float &cfrClip, float &cfrErrorClip, bool inverse = true);
    int process(Buffer *const dataIn, Buffer *dataOut) override;
    const char *name() override { return "OfdmGenerator"; }
    void set_parameter(const std::string &parameter, const std::string &value) override;
    const std::string get_parameter(const std::string &parameter) const override;
    const json::map_t get_all_values() const override;

private:
    double papr_db(const std::deque<double> &pairs) const;
    dabgpu_host::Context m_ctx;
    size_t m_nbSymbols, m_nbCarriers, m_spacing;
    bool &m_cfr;
    float &m_cfrClip, &m_cfrErrorClip;
    mutable std::mutex m_mutex;
    std::atomic<bool> m_paprClearRequest{false};
    size_t m_paprBlocks;                                   // PAPRStats(nbSymbols * 50), reference :60-61
    std::deque<double> m_clipRatios, m_errorClipRatios, m_mers, m_paprBefore, m_paprAfter;
};

// reference src/OfdmGenerator.h:114-146: the fixed-point (KISS FFT) engine, FFTEngine::KISS in src/DabModulator.cpp:208-213.
// Not offloaded (out of scope); the class exists so that the graph builder compiles unchanged, and
// its constructor throws: a configuration with fft_engine=kiss fails at the first frame with a message that says why,
// exactly like the reference's own FFTEngine::DEXTER branch without --enable-dexter (:222).
class OfdmGeneratorFixed : public ModCodec {
public:
    OfdmGeneratorFixed(size_t nbSymbols, size_t nbCarriers, size_t spacing, bool inverse = true);
    int process(Buffer *const dataIn, Buffer *dataOut) override;
    const char *name() override { return "OfdmGenerator"; }
};

// reference src/GainControl.h:47-91, .cpp:48-192, RC :505-572
class GainControl : public PipelinedModCodec, public RemoteControllable {
public:
    GainControl(size_t framesize, GainMode &gainMode, float &digGain, float normalise,
                float &varVariance);
    ~GainControl() override;
    const char *name() override { return "GainControl"; }
    void set_parameter(const std::string &parameter, const std::string &value) override;
    const std::string get_parameter(const std::string &parameter) const override;
    const json::map_t get_all_values() const override;

protected:
    int internal_process(Buffer *const dataIn, Buffer *dataOut) override;

private:
    dabgpu_host::Context m_ctx;
    size_t m_frameSize;
    float &m_digGain;
    float m_normalise;
    float &m_var_variance_rc;
    GainMode &m_gainmode;
    mutable std::mutex m_mutex;
};

// reference src/GuardIntervalInserter.h:45-98, .cpp:47-336, RC :338-375
class GuardIntervalInserter : public ModCodec, public RemoteControllable {
public:
    GuardIntervalInserter(size_t nbSymbols, size_t spacing, size_t nullSize, size_t symSize,
                          size_t &windowOverlap, FFTEngine fftEngine);
    int process(Buffer *const dataIn, Buffer *dataOut) override;
    const char *name() override { return "GuardIntervalInserter"; }
    void set_parameter(const std::string &parameter, const std::string &value) override;
    const std::string get_parameter(const std::string &parameter) const override;
    const json::map_t get_all_values() const override;

private:
    dabgpu_host::Context m_ctx;
    size_t &m_windowOverlap;
    mutable std::mutex m_mutex;
};

// reference src/FIRFilter.h:45-72, .cpp:73-141 (taps file), :144-309, RC :311-352
class FIRFilter : public PipelinedModCodec, public RemoteControllable {
public:
    explicit FIRFilter(std::string &taps_file);
    ~FIRFilter() override;
    const char *name() override { return "FIRFilter"; }
    void set_parameter(const std::string &parameter, const std::string &value) override;
    const std::string get_parameter(const std::string &parameter) const override;
    const json::map_t get_all_values() const override;

protected:
    int internal_process(Buffer *const dataIn, Buffer *dataOut) override;
    void load_filter_taps(const std::string &tapsFile);

private:
    dabgpu_host::Context m_ctx;
    std::string &m_taps_file;
    mutable std::mutex m_taps_mutex;
    std::vector<float> m_taps;
};

// reference src/Resampler.h:44, .cpp:51-195
class Resampler : public ModCodec {
public:
    Resampler(size_t inputRate, size_t outputRate, size_t resolution = 512);
    int process(Buffer *const dataIn, Buffer *dataOut) override;
    const char *name() override { return "Resampler"; }

private:
    dabgpu_host::Context m_ctx;
    size_t m_L, m_M;
};

// reference src/MemlessPoly.h:56, .cpp:59-232 (coefficient file), :342-411, RC :413-470
class MemlessPoly : public PipelinedModCodec, public RemoteControllable {
public:
    MemlessPoly(std::string &coefs_file, unsigned int num_threads);
    ~MemlessPoly() override;
    const char *name() override { return "MemlessPoly"; }
    void set_parameter(const std::string &parameter, const std::string &value) override;
    const std::string get_parameter(const std::string &parameter) const override;
    const json::map_t get_all_values() const override;

protected:
    int internal_process(Buffer *const dataIn, Buffer *dataOut) override;

private:
    void load_coefficients(std::istream &coefData);
    std::string serialise_coefficients() const;
    dabgpu_host::Context m_ctx;
    std::string &m_coefs_file;
    mutable std::mutex m_coefs_mutex;
    bool m_valid = false, m_is_lut = false;
    std::vector<float> m_am, m_pm, m_lut;
    float m_lut_scale = 0.f;
};

// reference src/CicEqualizer.h:37-52, .cpp:29-91
class CicEqualizer : public ModCodec {
public:
    CicEqualizer(size_t nbCarriers, size_t spacing, int R);
    int process(Buffer *const dataIn, Buffer *dataOut) override;
    const char *name() override { return "CicEqualizer"; }

private:
    dabgpu_host::Context m_ctx;
    size_t m_spacing;
    int m_R;
};

// reference src/TII.h:42-69 (settings) and :79-130, src/TII.cpp:106-245, RC :339-410
struct tii_config_t {
    bool enable = false;
    int comb = 0;
    int pattern = 0;
    bool old_variant = false;
};

class TIIError : public std::runtime_error {
public:
    explicit TIIError(const std::string &msg) : std::runtime_error(msg) {}
};

class TII : public ModCodec, public RemoteControllable {
public:
    TII(unsigned int dabmode, tii_config_t &tii_config, bool fixedPoint);
    int process(Buffer *dataIn, Buffer *dataOut) override;
    const char *name() override;
    void set_parameter(const std::string &parameter, const std::string &value) override;
    const std::string get_parameter(const std::string &parameter) const override;
    const json::map_t get_all_values() const override;

private:
    void push_settings();
    dabgpu_host::Context m_ctx;
    tii_config_t &m_conf;
    std::string m_name;
    mutable std::mutex m_mutex;
};

// reference src/FormatConverter.h:42-66, .cpp:41-209 (float input only: the
// fixed-point engine is not offloaded)
class FormatConverter : public ModCodec {
public:
    static size_t get_format_size(const std::string &format);
    FormatConverter(bool input_is_complexfix_wide, const std::string &format_out);
    int process(Buffer *const dataIn, Buffer *dataOut) override;
    const char *name() override { return "FormatConverter"; }
    size_t get_num_clipped_samples() const { return m_num_clipped_samples.load(); }

private:
    dabgpu_host::Context m_ctx;
    std::string m_format_out;
    std::atomic<size_t> m_num_clipped_samples{0};
};

// The production plugin: cifPart output (28 800 B per Mode-I transmission frame)
// in, finished IQ out -- replaces cifMap .. cifGuard/cifFilter/cifRes/cifPoly of
// src/DabModulator.cpp:385-419 with one node.  Not pipelined: no frame is lost at
// start-up (each PipelinedModCodec of the reference drops one).
class DabGpuChain : public ModCodec {
public:
    struct Settings {
        unsigned dabMode = 1;
        GainMode gainMode = GainMode::GAIN_VAR;
        float digitalGain = 1.0f, normalise = 1.0f, gainmodeVariance = 4.0f;
        bool enableGain = true;
        std::string filterTapsFilename;   // "" = no FIR, "default" = built-in taps
        size_t outputRate = 2048000;
        std::string polyCoefFilename;     // "" = no predistortion
        size_t ofdmWindowOverlap = 0;
        tii_config_t tiiConfig;           // TII on every other frame of the stream (modes I and II)
        bool enableCfr = false;           // crest-factor reduction inside OfdmGenerator
        float cfrClip = 1.0f, cfrErrorClip = 1.0f;
        // "complexf" (default), "s16", "u8" or "s8": FormatConverter as the chain's last step
        // (src/DabModulator.cpp:270-276, :407) -- for s16 the last kernel stores the integers itself
        std::string outputFormat = "complexf";
        // frames per call of the streaming interface below (process() always takes one)
        size_t maxBatchFrames = 1;
        // Compatibility with the reference's start-up behaviour (0 = off, the default: every frame comes out, with
        // no latency).  Each PipelinedModCodec of the reference -- GainControl, FIRFilter, MemlessPoly -- returns the
        // PREVIOUS frame and nothing on its first call (src/ModPlugin.cpp:90-115), so a reference flowgraph with k of
        // them emits frame i - k on call i and N - k frames for N calls.  With emulatePipelineDrops = k, process() does
        // the same: the first k calls return 0 (Flowgraph::run stops the walk, src/Flowgraph.cpp:334-336), call i
        // returns frame i - k, the last k frames never leave.  referencePipelineDepth() is the k of the equivalent
        // reference graph.  process() ONLY: submit() throws when the option is set (a streaming caller holds the frames
        // back itself, see dabmod_file --batch --reference-latency).
        unsigned emulatePipelineDrops = 0;
        unsigned referencePipelineDepth() const
        {
            return (enableGain ? 1u : 0u) + (filterTapsFilename.empty() ? 0u : 1u) + (polyCoefFilename.empty() ? 0u : 1u);
        }
    };
    explicit DabGpuChain(const Settings &s);
    // Streaming interface for a caller that can look ahead (a file, a buffered network input): submit() queues
    // n_frames transmission frames (n_frames x the hot-path input), at most two batches in flight; collect() waits
    // for the oldest and returns its IQ in a pinned buffer of the context, valid until the second next submit().
    // Same stream state (resampler halo, TII parity) and the same samples as process() frame by frame.
    void submit(const void *bits, size_t n_frames);
    size_t collect(const void **iq);
    size_t input_bytes_per_frame() const { return m_in_bytes; }
    size_t output_bytes_per_frame() const;
    int process(Buffer *const dataIn, Buffer *dataOut) override;
    const char *name() override { return "DabGpuChain"; }
    // FormatConverter::get_num_clipped_samples of the most recent frame (src/FormatConverter.cpp:56-59)
    size_t get_num_clipped_samples() const;

private:
    dabgpu_host::Context m_ctx;
    unsigned m_mask = 0;
    size_t m_in_bytes = 0;
    unsigned m_drops = 0;                 // Settings::emulatePipelineDrops
    std::deque<Buffer> m_delayed;         // the frames "inside the reference's pipeline"
};
