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
#include <memory>
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
// first base of an adapter that owns its context: constructed before the control facet that is bound to it
struct OwnContext {
    explicit OwnContext(int mode, int max_frames = 1) : m_ctx(mode, max_frames) {}
    Context m_ctx;
};

// ---- The remote-control surface of each stage, bound to a device context SOMEONE ELSE owns.
// The reference's runtime parameters ("RC-mutable parameters the device context must accept at runtime")
// live in the stage objects; here each stage's set of them is a RemoteControllable of its own -- same RC name, same
// parameters, same strings and error texts as the reference class -- that forwards to the dabgpu_set_* entry points of
// include/dabgpu.h.  The per-stage adapters below ARE one of these (on their own context); DabGpuChain OWNS one per
// stage its chain contains (all on the chain's one context) and hands them to the remote-control registry.
// Values live where the reference keeps them: in the mod_settings_t fields the constructors take by reference, so a
// modulator restart (src/DabMod.cpp:593-724) comes back with what the remote control set.

// reference src/OfdmGenerator.cpp:376-458 (cfr, clip, errorclip; read-only clip_stats, papr)
class OfdmControl : public RemoteControllable {
public:
    OfdmControl(dabgpu_ctx *dev, size_t nbSymbols, bool &enableCfr, float &cfrClip, float &cfrErrorClip);
    void set_parameter(const std::string &parameter, const std::string &value) override;
    const std::string get_parameter(const std::string &parameter) const override;
    const json::map_t get_all_values() const override;
    bool push_settings();        // before a frame: the live values go to the context; returns "CFR is on"
    void collect_statistics();   // after a frame that ran with CFR: the reference's running statistics (:232,246-306)

private:
    double papr_db(const std::deque<double> &pairs) const;
    dabgpu_ctx *m_dev;
    bool &m_cfr;
    float &m_cfrClip, &m_cfrErrorClip;
    mutable std::mutex m_mutex;
    std::atomic<bool> m_paprClearRequest{false};
    size_t m_paprBlocks;                                   // PAPRStats(nbSymbols * 50), reference :60-61
    std::deque<double> m_clipRatios, m_errorClipRatios, m_mers, m_paprBefore, m_paprAfter;
};

// reference src/GainControl.cpp:505-572 (digital, mode, var)
class GainParameters : public RemoteControllable {
public:
    GainParameters(dabgpu_ctx *dev, GainMode &gainMode, float &digGain, float normalise, float &varVariance);
    void set_parameter(const std::string &parameter, const std::string &value) override;
    const std::string get_parameter(const std::string &parameter) const override;
    const json::map_t get_all_values() const override;
    void push_settings();

private:
    dabgpu_ctx *m_dev;
    float &m_digGain;
    float m_normalise;
    float &m_var_variance_rc;
    GainMode &m_gainmode;
    mutable std::mutex m_mutex;
};

// reference src/GuardIntervalInserter.cpp:338-375 (windowlen)
class GuardParameters : public RemoteControllable {
public:
    GuardParameters(dabgpu_ctx *dev, size_t &windowOverlap);
    void set_parameter(const std::string &parameter, const std::string &value) override;
    const std::string get_parameter(const std::string &parameter) const override;
    const json::map_t get_all_values() const override;

private:
    dabgpu_ctx *m_dev;
    size_t &m_windowOverlap;
    mutable std::mutex m_mutex;
};

// reference src/FIRFilter.cpp:73-141 (taps file), :311-352 (ntaps, tapsfile)
class FirParameters : public RemoteControllable {
public:
    FirParameters(dabgpu_ctx *dev, std::string &taps_file);
    void set_parameter(const std::string &parameter, const std::string &value) override;
    const std::string get_parameter(const std::string &parameter) const override;
    const json::map_t get_all_values() const override;

protected:
    void load_filter_taps(const std::string &tapsFile);

private:
    dabgpu_ctx *m_dev;
    std::string &m_taps_file;
    mutable std::mutex m_taps_mutex;
    std::vector<float> m_taps;
};

// reference src/MemlessPoly.cpp:59-232 (coefficient file), :413-470 (ncoefs, coefs, coeffile)
class PolyParameters : public RemoteControllable {
public:
    PolyParameters(dabgpu_ctx *dev, std::string &coefs_file);
    void set_parameter(const std::string &parameter, const std::string &value) override;
    const std::string get_parameter(const std::string &parameter) const override;
    const json::map_t get_all_values() const override;
    bool settings_valid() const;   // false: frames pass through unchanged (reference :397-409)

private:
    void load_coefficients(std::istream &coefData);
    std::string serialise_coefficients() const;
    dabgpu_ctx *m_dev;
    std::string &m_coefs_file;
    mutable std::mutex m_coefs_mutex;
    bool m_valid = false, m_is_lut = false;
    std::vector<float> m_am, m_pm, m_lut;
    float m_lut_scale = 0.f;
};
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
class OfdmGeneratorCF32 : private dabgpu_host::OwnContext, public ModCodec, public dabgpu_host::OfdmControl {
public:
    OfdmGeneratorCF32(size_t nbSymbols, size_t nbCarriers, size_t spacing, bool &enableCfr,
                      float &cfrClip, float &cfrErrorClip, bool inverse = true);
    int process(Buffer *const dataIn, Buffer *dataOut) override;
    const char *name() override { return "OfdmGenerator"; }

private:
    size_t m_nbSymbols, m_nbCarriers, m_spacing;
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
class GainControl : private dabgpu_host::OwnContext, public PipelinedModCodec, public dabgpu_host::GainParameters {
public:
    GainControl(size_t framesize, GainMode &gainMode, float &digGain, float normalise,
                float &varVariance);
    ~GainControl() override;
    const char *name() override { return "GainControl"; }

protected:
    int internal_process(Buffer *const dataIn, Buffer *dataOut) override;
};

// reference src/GuardIntervalInserter.h:45-98, .cpp:47-336, RC :338-375
class GuardIntervalInserter : private dabgpu_host::OwnContext, public ModCodec, public dabgpu_host::GuardParameters {
public:
    GuardIntervalInserter(size_t nbSymbols, size_t spacing, size_t nullSize, size_t symSize,
                          size_t &windowOverlap, FFTEngine fftEngine);
    int process(Buffer *const dataIn, Buffer *dataOut) override;
    const char *name() override { return "GuardIntervalInserter"; }
};

// reference src/FIRFilter.h:45-72, .cpp:73-141 (taps file), :144-309, RC :311-352
class FIRFilter : private dabgpu_host::OwnContext, public PipelinedModCodec, public dabgpu_host::FirParameters {
public:
    explicit FIRFilter(std::string &taps_file);
    ~FIRFilter() override;
    const char *name() override { return "FIRFilter"; }

protected:
    int internal_process(Buffer *const dataIn, Buffer *dataOut) override;
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
class MemlessPoly : private dabgpu_host::OwnContext, public PipelinedModCodec, public dabgpu_host::PolyParameters {
public:
    MemlessPoly(std::string &coefs_file, unsigned int num_threads);
    ~MemlessPoly() override;
    const char *name() override { return "MemlessPoly"; }

protected:
    int internal_process(Buffer *const dataIn, Buffer *dataOut) override;
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

namespace dabgpu_host {
// reference src/TII.cpp:339-410 (enable, comb, pattern, old_variant)
class TiiParameters : public RemoteControllable {
public:
    TiiParameters(dabgpu_ctx *dev, tii_config_t &tii_config);
    void set_parameter(const std::string &parameter, const std::string &value) override;
    const std::string get_parameter(const std::string &parameter) const override;
    const json::map_t get_all_values() const override;

protected:
    void push_settings();
    dabgpu_ctx *m_dev;
    tii_config_t &m_conf;
    mutable std::mutex m_mutex;
};
}  // namespace dabgpu_host

class TII : private dabgpu_host::OwnContext, public ModCodec, public dabgpu_host::TiiParameters {
public:
    TII(unsigned int dabmode, tii_config_t &tii_config, bool fixedPoint);
    int process(Buffer *dataIn, Buffer *dataOut) override;
    const char *name() override;

private:
    std::string m_name;
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
//
// Remote control: the stages inside keep the reference's runtime parameters.  remote_controllables() lists one
// RemoteControllable per stage the chain contains, under the reference's names -- "ofdm", "gain", "guardinterval",
// always; "tii" in modes I and II; "firfilter" / "memlesspoly" when the chain was built with a taps / coefficient file,
// exactly the objects src/DabModulator.cpp:186,203,238,245,252,260 enrols -- all forwarding to the chain's ONE context;
// a parameter set between two process() calls shapes the next frame.
// Metadata: a ModMetadata like the reference's pipelined stages (src/ModPlugin.cpp:117-128).  With
// emulatePipelineDrops = k the frame AND its metadata leave k calls later; with 0 both pass straight through.
class DabGpuChain : public ModCodec, public ModMetadata {
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
        // Gain mode var: form the multiplier by the reference's running fp32 recurrence (src/GainControl.cpp:251-340)
        // instead of the exact variance -- the reference's scalars bit for bit on the same symbols, at about 30 % of
        // the chain's rate (dabgpu_set_gain_rounding).  Off by default.
        bool referenceGainRounding = false;
        unsigned referencePipelineDepth() const
        {
            return (enableGain ? 1u : 0u) + (filterTapsFilename.empty() ? 0u : 1u) + (polyCoefFilename.empty() ? 0u : 1u);
        }
    };
    // Where the RC-mutable values LIVE when the chain sits inside the reference's DabModulator: its mod_settings_t
    // (src/ConfigParser.h:45-98), whose fields the reference's stage constructors take by reference
    // (src/DabModulator.cpp:195-260).  Every pointer may be null: that parameter then lives in the chain's own copy
    // of Settings.  Settings still supplies everything that is fixed at construction.
    struct LiveSettings {
        GainMode *gainMode = nullptr;
        float *digitalGain = nullptr, *gainmodeVariance = nullptr;
        std::string *filterTapsFilename = nullptr, *polyCoefFilename = nullptr;
        size_t *ofdmWindowOverlap = nullptr;
        tii_config_t *tiiConfig = nullptr;
        bool *enableCfr = nullptr;
        float *cfrClip = nullptr, *cfrErrorClip = nullptr;
    };
    explicit DabGpuChain(const Settings &s);
    DabGpuChain(const Settings &s, const LiveSettings &live);
    ~DabGpuChain() override;
    // the RemoteControllables to enrol (rcs.enrol(p), lib/RemoteControl.h:141); owned by the chain
    std::vector<RemoteControllable *> remote_controllables() const;
    // Streaming interface for a caller that can look ahead (a file, a buffered network input): submit() queues
    // n_frames transmission frames (n_frames x the hot-path input), at most two batches in flight; collect() waits
    // for the oldest and returns its IQ in a pinned buffer of the context, valid until the second next submit().
    // Same stream state (resampler halo, TII parity) and the same samples as process() frame by frame.
    void submit(const void *bits, size_t n_frames);
    size_t collect(const void **iq);
    size_t input_bytes_per_frame() const { return m_in_bytes; }
    size_t output_bytes_per_frame() const;
    int process(Buffer *const dataIn, Buffer *dataOut) override;
    meta_vec_t process_metadata(const meta_vec_t &metadataIn) override;
    const char *name() override { return "DabGpuChain"; }
    // FormatConverter::get_num_clipped_samples of the most recent frame (src/FormatConverter.cpp:56-59)
    size_t get_num_clipped_samples() const;

private:
    unsigned stage_mask();                // the live mask: predistortion leaves it while its settings are invalid
    void before_frames();                 // live parameters -> context
    void after_frames();                  // CFR statistics
    dabgpu_host::Context m_ctx;
    Settings m_own;                       // the values no LiveSettings pointer claims
    unsigned m_mask = 0;
    size_t m_in_bytes = 0;
    unsigned m_drops = 0;                 // Settings::emulatePipelineDrops
    bool m_cfr_on = false;                // the frame in hand runs with crest-factor reduction
    std::deque<Buffer> m_delayed;         // the frames "inside the reference's pipeline"
    std::deque<meta_vec_t> m_delayed_meta;   // ... and their metadata
    std::unique_ptr<dabgpu_host::OfdmControl> m_rc_ofdm;
    std::unique_ptr<dabgpu_host::GainParameters> m_rc_gain;
    std::unique_ptr<dabgpu_host::GuardParameters> m_rc_guard;
    std::unique_ptr<dabgpu_host::FirParameters> m_rc_fir;
    std::unique_ptr<dabgpu_host::PolyParameters> m_rc_poly;
    std::unique_ptr<dabgpu_host::TiiParameters> m_rc_tii;
};
