// GpuStages.cpp -- see GpuStages.h.  Everything numerical happens behind the
// C-ABI of include/dabgpu.h; this file is argument plumbing, parameter parsing
// and the reference's error conventions.
#include "GpuStages.h"

#include "dabgpu.h"

#include <algorithm>
#include <cctype>
#include <cstring>
#include <complex>
#include <fstream>
#include <deque>
#include <numeric>
#include <cmath>
#include <iomanip>
#include <sstream>
#include <stdexcept>

namespace dabgpu_host {

Context::Context(int mode, int max_frames)
{
    dabgpu_config cfg{};
    cfg.mode = mode;
    cfg.device = 0;
    cfg.max_frames = max_frames;
    cfg.chunks_per_frame = 0;
    if (const char *d = std::getenv("DABGPU_DEVICE")) cfg.device = std::atoi(d);
    const int rc = dabgpu_create(&cfg, &m_ctx);
    if (rc != DABGPU_OK)
        throw std::runtime_error(std::string("dabgpu_create: ") + dabgpu_last_error(nullptr));
}

Context::~Context() { dabgpu_destroy(m_ctx); }

void Context::check(int rc) const
{
    if (rc != DABGPU_OK) throw std::runtime_error(dabgpu_last_error(m_ctx));
}

int mode_from_carriers(size_t carriers)
{
    switch (carriers) {
        case 1536: return 1;
        case 384: return 2;
        case 192: return 3;
        case 768: return 4;
    }
    throw std::runtime_error("unsupported number of carriers: " + std::to_string(carriers));
}

int mode_from_spacing(size_t spacing)
{
    switch (spacing) {
        case 2048: return 1;
        case 512: return 2;
        case 256: return 3;
        case 1024: return 4;
    }
    throw std::runtime_error("unsupported carrier spacing: " + std::to_string(spacing));
}

namespace {
// run a (ctx, in, in_bytes, out, cap, &n) entry point with the producer-sizes-its-output protocol
template <typename Fn> int run_codec(const Context &c, Fn fn, Buffer *const in, Buffer *out, size_t out_bytes)
{
    out->setLength(out_bytes);
    size_t n = 0;
    c.check(fn(c.get(), in->getData(), in->getLength(), out->getData(), out->getLength(), &n));
    out->setLength(n);
    return static_cast<int>(n);
}

void check_dev(dabgpu_ctx *dev, int rc)
{
    if (rc != DABGPU_OK) throw std::runtime_error(dabgpu_last_error(dev));
}

[[noreturn]] void not_exported(const std::string &parameter, const std::string &rc_name)
{
    throw ParameterError("Parameter '" + parameter + "' is not exported by controllable " + rc_name);
}
}  // namespace
}  // namespace dabgpu_host

using dabgpu_host::run_codec;

// ---------------------------------------------------------------- QpskSymbolMapper
QpskSymbolMapper::QpskSymbolMapper(size_t carriers, bool fixedPoint)
    : m_ctx(dabgpu_host::mode_from_carriers(carriers))
{
    if (fixedPoint) throw std::runtime_error("QpskSymbolMapper: the fixed-point engine is not offloaded");
}

int QpskSymbolMapper::process(Buffer *const dataIn, Buffer *dataOut)
{
    run_codec(m_ctx, dabgpu_qpsk_process, dataIn, dataOut, dataIn->getLength() * 4 * sizeof(complexf));
    return 1;
}

// ---------------------------------------------------------------- FrequencyInterleaver
FrequencyInterleaver::FrequencyInterleaver(size_t mode, bool fixedPoint) : m_ctx(static_cast<int>(mode))
{
    if (fixedPoint) throw std::runtime_error("FrequencyInterleaver: the fixed-point engine is not offloaded");
}

int FrequencyInterleaver::process(Buffer *const dataIn, Buffer *dataOut)
{
    run_codec(m_ctx, dabgpu_freq_interleave_process, dataIn, dataOut, dataIn->getLength());
    return 1;
}

// ---------------------------------------------------------------- PhaseReference
PhaseReference::PhaseReference(unsigned int dabmode, bool fixedPoint) : m_ctx(static_cast<int>(dabmode))
{
    if (fixedPoint) throw std::runtime_error("PhaseReference: the fixed-point engine is not offloaded");
}

int PhaseReference::process(Buffer *dataOut)
{
    dabgpu_geometry g;
    m_ctx.check(dabgpu_get_geometry(m_ctx.get(), &g));
    dataOut->setLength(static_cast<size_t>(g.carriers) * sizeof(complexf));
    size_t n = 0;
    m_ctx.check(dabgpu_phase_reference_process(m_ctx.get(), dataOut->getData(), dataOut->getLength(), &n));
    return 1;
}

// ---------------------------------------------------------------- DifferentialModulator
DifferentialModulator::DifferentialModulator(size_t carriers, bool fixedPoint)
    : m_ctx(dabgpu_host::mode_from_carriers(carriers))
{
    if (fixedPoint) throw std::runtime_error("DifferentialModulator: the fixed-point engine is not offloaded");
}

int DifferentialModulator::process(std::vector<Buffer *> dataIn, Buffer *dataOut)
{
    if (dataIn.size() != 2)
        throw std::runtime_error("DifferentialModulator::process nb of input streams not 2!");
    dataOut->setLength(dataIn[0]->getLength() + dataIn[1]->getLength());
    size_t n = 0;
    m_ctx.check(dabgpu_diff_mod_process(m_ctx.get(), dataIn[0]->getData(), dataIn[0]->getLength(),
                                        dataIn[1]->getData(), dataIn[1]->getLength(),
                                        dataOut->getData(), dataOut->getLength(), &n));
    return static_cast<int>(dataOut->getLength());
}

// ---------------------------------------------------------------- NullSymbol / SignalMultiplexer
NullSymbol::NullSymbol(size_t numCarriers, size_t typeSize) : m_bytes(numCarriers * typeSize) {}

int NullSymbol::process(Buffer *dataOut)
{
    dataOut->setLength(m_bytes);
    std::memset(dataOut->getData(), 0, m_bytes);
    return static_cast<int>(m_bytes);
}

int SignalMultiplexer::process(std::vector<Buffer *> dataIn, Buffer *dataOut)
{
    if (dataIn.size() != 2 && dataIn.size() != 3)
        throw std::runtime_error("SignalMultiplexer::process needs 2 or 3 inputs");
    *dataOut = *dataIn[dataIn.size() == 3 ? 2 : 0];  // TII symbol when present, else NULL symbol
    *dataOut += *dataIn[1];
    return static_cast<int>(dataOut->getLength());
}

// ---------------------------------------------------------------- OfdmGenerator
// The fixed-point engine (reference src/OfdmGenerator.cpp:467-579) is not offloaded: refuse at construction, which
// src/DabModulator.cpp:208-213 reaches on the first frame -- run_modulator reports it (src/DabMod.cpp:732-735).
OfdmGeneratorFixed::OfdmGeneratorFixed(size_t, size_t, size_t, bool)
{
    throw std::runtime_error("OfdmGenerator: the fixed-point engine (fft_engine=kiss) is not offloaded to the GPU; "
                             "set fft_engine=fftw");
}

int OfdmGeneratorFixed::process(Buffer *const, Buffer *) { return 0; }

OfdmGeneratorCF32::OfdmGeneratorCF32(size_t nbSymbols, size_t nbCarriers, size_t spacing,
                                     bool &enableCfr, float &cfrClip, float &cfrErrorClip, bool inverse)
    : OwnContext(dabgpu_host::mode_from_spacing(spacing)),
      OfdmControl(m_ctx.get(), nbSymbols, enableCfr, cfrClip, cfrErrorClip),
      m_nbSymbols(nbSymbols), m_nbCarriers(nbCarriers), m_spacing(spacing)
{
    if (nbCarriers > spacing) throw std::runtime_error("OfdmGenerator nbCarriers > spacing!");
    if (!inverse) throw std::runtime_error("OfdmGenerator: forward transform is not offloaded");
}

int OfdmGeneratorCF32::process(Buffer *const dataIn, Buffer *dataOut)
{
    const bool cfr = push_settings();
    const int n = run_codec(m_ctx, dabgpu_ofdm_process, dataIn, dataOut, m_nbSymbols * m_spacing * sizeof(complexf)) /
                  static_cast<int>(sizeof(complexf));
    if (cfr) collect_statistics();
    return n;
}

namespace dabgpu_host {
OfdmControl::OfdmControl(dabgpu_ctx *dev, size_t nbSymbols, bool &enableCfr, float &cfrClip, float &cfrErrorClip)
    : RemoteControllable("ofdm"), m_dev(dev), m_cfr(enableCfr), m_cfrClip(cfrClip), m_cfrErrorClip(cfrErrorClip),
      m_paprBlocks(nbSymbols * 50)
{
    // reference src/OfdmGenerator.cpp:67-74
    RC_ADD_PARAMETER(cfr, "Enable crest factor reduction");
    RC_ADD_PARAMETER(clip, "CFR: Clip to amplitude");
    RC_ADD_PARAMETER(errorclip, "CFR: Limit error");
    RC_ADD_PARAMETER(clip_stats, "CFR: statistics (clip ratio, errorclip ratio)");
    RC_ADD_PARAMETER(papr, "PAPR measurements (before CFR, after CFR)");
}

bool OfdmControl::push_settings()
{
    std::lock_guard<std::mutex> lock(m_mutex);
    check_dev(m_dev, dabgpu_set_cfr(m_dev, m_cfr ? 1 : 0, m_cfrClip, m_cfrErrorClip));
    if (m_paprClearRequest.exchange(false)) {            // reference :202-205
        m_paprBefore.clear();
        m_paprAfter.clear();
    }
    return m_cfr;
}

void OfdmControl::collect_statistics()
{
    {
        // the reference's running statistics (src/OfdmGenerator.cpp:232,246-306) from the raw per-frame figures
        dabgpu_cfr_stats st;
        check_dev(m_dev, dabgpu_get_cfr_stats(m_dev, 0, &st));
        std::lock_guard<std::mutex> lock(m_mutex);
        auto push = [](std::deque<double> &d, double v, size_t cap) {
            d.push_back(v);
            while (d.size() > cap) d.pop_front();
        };
        for (int s = 0; s < st.nb_symbols; ++s) {
            push(m_paprBefore, st.papr_before[s][0], 2 * m_paprBlocks);   // stored as (peak, mean) pairs
            push(m_paprBefore, st.papr_before[s][1], 2 * m_paprBlocks);
            if (s > 0) {
                push(m_paprAfter, st.papr_after[s][0], 2 * m_paprBlocks);
                push(m_paprAfter, st.papr_after[s][1], 2 * m_paprBlocks);
            }
        }
        constexpr size_t MAX_CLIP_STATS = 10;                               // :38
        push(m_clipRatios, static_cast<double>(st.num_clip) / static_cast<double>(st.num_samples), MAX_CLIP_STATS);
        push(m_errorClipRatios, static_cast<double>(st.num_error_clip) / static_cast<double>(st.num_samples),
             MAX_CLIP_STATS);
        if (st.mer_symbol > 0)
            push(m_mers, st.mer_sum_delta > 0 ? 10.0 * std::log10(st.mer_sum_iq / st.mer_sum_delta) : 90.0,
                 MAX_CLIP_STATS);
    }
}

// PAPRStats::calculate_papr (reference src/PAPRStats.cpp:74-103) over stored (peak, mean) pairs
double OfdmControl::papr_db(const std::deque<double> &pairs) const
{
    if (pairs.size() / 2 < m_paprBlocks) return 0;
    double peak = 0, rms2 = 0;
    for (size_t i = 0; i + 1 < pairs.size(); i += 2) {
        peak = std::max(peak, pairs[i]);
        rms2 += pairs[i + 1];
    }
    rms2 /= static_cast<double>(pairs.size() / 2);
    return 10.0 * std::log10(peak / rms2);
}

void OfdmControl::set_parameter(const std::string &parameter, const std::string &value)
{
    std::stringstream ss(value);
    ss.exceptions(std::stringstream::failbit | std::stringstream::badbit);
    std::lock_guard<std::mutex> lock(m_mutex);
    if (parameter == "cfr") {
        ss >> m_cfr;
    } else if (parameter == "clip") {
        ss >> m_cfrClip;
    } else if (parameter == "errorclip") {
        ss >> m_cfrErrorClip;
    } else if (parameter == "clip_stats" || parameter == "papr") {
        throw ParameterError("Parameter '" + parameter + "' is read-only");
    } else {
        not_exported(parameter, get_rc_name());
    }
    m_paprClearRequest.store(true);
}

const std::string OfdmControl::get_parameter(const std::string &parameter) const
{
    std::stringstream ss;
    std::lock_guard<std::mutex> lock(m_mutex);
    if (parameter == "cfr") {
        ss << m_cfr;
    } else if (parameter == "clip") {
        ss << std::fixed << m_cfrClip;
    } else if (parameter == "errorclip") {
        ss << std::fixed << m_cfrErrorClip;
    } else if (parameter == "clip_stats") {
        if (m_clipRatios.empty() || m_errorClipRatios.empty() || m_mers.empty()) {
            ss << "No stats available";
        } else {
            auto avg = [](const std::deque<double> &d) {
                return std::accumulate(d.begin(), d.end(), 0.0) / static_cast<double>(d.size());
            };
            ss << "Statistics : " << std::fixed << avg(m_clipRatios) * 100 << "% samples clipped, "
               << avg(m_errorClipRatios) * 100 << "% errors clipped. MER after CFR: " << avg(m_mers) << " dB";
        }
    } else if (parameter == "papr") {
        const double before = papr_db(m_paprBefore), after = papr_db(m_paprAfter);
        ss << "PAPR [dB]: " << std::fixed << (before == 0 ? std::string("N/A") : std::to_string(before)) << ", "
           << (after == 0 ? std::string("N/A") : std::to_string(after));
    } else {
        not_exported(parameter, get_rc_name());
    }
    return ss.str();
}

const json::map_t OfdmControl::get_all_values() const
{
    json::map_t m;   // (empty in the reference too: "TODO needs rework of the values", :453-458)
    return m;
}
}  // namespace dabgpu_host

// ---------------------------------------------------------------- GainControl
GainControl::GainControl(size_t framesize, GainMode &gainMode, float &digGain, float normalise,
                         float &varVariance)
    : OwnContext(dabgpu_host::mode_from_spacing(framesize)),
      GainParameters(m_ctx.get(), gainMode, digGain, normalise, varVariance)
{
    start_pipeline_thread();
}

GainControl::~GainControl() { stop_pipeline_thread(); }

int GainControl::internal_process(Buffer *const dataIn, Buffer *dataOut)
{
    push_settings();
    return run_codec(m_ctx, dabgpu_gain_process, dataIn, dataOut, dataIn->getLength()) /
           static_cast<int>(sizeof(complexf));
}

namespace dabgpu_host {
GainParameters::GainParameters(dabgpu_ctx *dev, GainMode &gainMode, float &digGain, float normalise, float &varVariance)
    : RemoteControllable("gain"), m_dev(dev), m_digGain(digGain), m_normalise(normalise),
      m_var_variance_rc(varVariance), m_gainmode(gainMode)
{
    RC_ADD_PARAMETER(digital, "Digital Gain");
    RC_ADD_PARAMETER(mode, "Gainmode (fix|max|var)");
    RC_ADD_PARAMETER(var, "Variance setting for gainmode var (default: 4)");
}

void GainParameters::push_settings()
{
    int mode;
    float dig, var;
    {
        std::lock_guard<std::mutex> lock(m_mutex);
        mode = static_cast<int>(m_gainmode);
        dig = m_digGain;
        var = m_var_variance_rc;
    }
    check_dev(m_dev, dabgpu_set_gain(m_dev, mode, dig, m_normalise, var));
}

void GainParameters::set_parameter(const std::string &parameter, const std::string &value)
{
    std::stringstream ss(value);
    ss.exceptions(std::stringstream::failbit | std::stringstream::badbit);
    if (parameter == "digital") {
        float f;
        ss >> f;
        std::lock_guard<std::mutex> lock(m_mutex);
        m_digGain = f;
    } else if (parameter == "mode") {
        std::string m;
        ss >> m;
        std::transform(m.begin(), m.end(), m.begin(), [](char c) { return std::tolower(c); });
        GainMode g;
        if (m == "fix") g = GainMode::GAIN_FIX;
        else if (m == "max") g = GainMode::GAIN_MAX;
        else if (m == "var") g = GainMode::GAIN_VAR;
        else throw ParameterError("Gainmode " + m + " unknown");
        std::lock_guard<std::mutex> lock(m_mutex);
        m_gainmode = g;
    } else if (parameter == "var") {
        float f = 0;
        ss >> f;
        std::lock_guard<std::mutex> lock(m_mutex);
        m_var_variance_rc = f;
    } else {
        not_exported(parameter, get_rc_name());
    }
}

const std::string GainParameters::get_parameter(const std::string &parameter) const
{
    std::stringstream ss;
    std::lock_guard<std::mutex> lock(m_mutex);
    if (parameter == "digital") ss << std::fixed << m_digGain;
    else if (parameter == "mode")
        ss << (m_gainmode == GainMode::GAIN_FIX ? "fix" : m_gainmode == GainMode::GAIN_MAX ? "max" : "var");
    else if (parameter == "var") ss << std::fixed << m_var_variance_rc;
    else not_exported(parameter, get_rc_name());
    return ss.str();
}

const json::map_t GainParameters::get_all_values() const
{
    json::map_t m;
    std::lock_guard<std::mutex> lock(m_mutex);
    m["digital"].v = static_cast<double>(m_digGain);
    m["mode"].v = std::string(m_gainmode == GainMode::GAIN_FIX ? "fix"
                              : m_gainmode == GainMode::GAIN_MAX ? "max" : "var");
    m["var"].v = static_cast<double>(m_var_variance_rc);
    return m;
}
}  // namespace dabgpu_host

// ---------------------------------------------------------------- GuardIntervalInserter
GuardIntervalInserter::GuardIntervalInserter(size_t nbSymbols, size_t spacing, size_t nullSize,
                                             size_t symSize, size_t &windowOverlap, FFTEngine fftEngine)
    : OwnContext(dabgpu_host::mode_from_spacing(spacing)), GuardParameters(m_ctx.get(), windowOverlap)
{
    if (nullSize == 0) throw std::logic_error("NULL symbol must be present");
    if (static_cast<int>(fftEngine) != 0 /* FFTEngine::FFTW: the enumeration may be opaque here, GpuStages.h */) throw std::runtime_error("GuardIntervalInserter: only the float engine is offloaded");
    dabgpu_geometry g;
    m_ctx.check(dabgpu_get_geometry(m_ctx.get(), &g));
    if ((size_t)g.nb_symbols != nbSymbols || (size_t)g.null_size != nullSize || (size_t)g.sym_size != symSize)
        throw std::runtime_error("GuardIntervalInserter: geometry does not match the transmission mode");
}

int GuardIntervalInserter::process(Buffer *const dataIn, Buffer *dataOut)
{
    dabgpu_geometry g;
    m_ctx.check(dabgpu_get_geometry(m_ctx.get(), &g));
    return run_codec(m_ctx, dabgpu_guard_process, dataIn, dataOut, g.tf_samples * sizeof(complexf));
}

namespace dabgpu_host {
GuardParameters::GuardParameters(dabgpu_ctx *dev, size_t &windowOverlap)
    : RemoteControllable("guardinterval"), m_dev(dev), m_windowOverlap(windowOverlap)
{
    RC_ADD_PARAMETER(windowlen, "Window length for OFDM windowng [0 to disable]");
    check_dev(m_dev, dabgpu_set_window_overlap(m_dev, windowOverlap));
}

void GuardParameters::set_parameter(const std::string &parameter, const std::string &value)
{
    if (parameter != "windowlen") not_exported(parameter, get_rc_name());
    std::stringstream ss(value);
    ss.exceptions(std::stringstream::failbit | std::stringstream::badbit);
    size_t w = 0;
    ss >> w;
    std::lock_guard<std::mutex> lock(m_mutex);
    m_windowOverlap = w;
    check_dev(m_dev, dabgpu_set_window_overlap(m_dev, w));
}

const std::string GuardParameters::get_parameter(const std::string &parameter) const
{
    if (parameter != "windowlen") not_exported(parameter, get_rc_name());
    std::lock_guard<std::mutex> lock(m_mutex);
    return std::to_string(m_windowOverlap);
}

const json::map_t GuardParameters::get_all_values() const
{
    json::map_t m;
    std::lock_guard<std::mutex> lock(m_mutex);
    m["windowlen"].v = static_cast<uint64_t>(m_windowOverlap);
    return m;
}
}  // namespace dabgpu_host

// ---------------------------------------------------------------- FIRFilter
FIRFilter::FIRFilter(std::string &taps_file) : OwnContext(1), FirParameters(m_ctx.get(), taps_file)
{
    start_pipeline_thread();
}

FIRFilter::~FIRFilter() { stop_pipeline_thread(); }

int FIRFilter::internal_process(Buffer *const dataIn, Buffer *dataOut)
{
    return run_codec(m_ctx, dabgpu_fir_process, dataIn, dataOut, dataIn->getLength());
}

namespace dabgpu_host {
FirParameters::FirParameters(dabgpu_ctx *dev, std::string &taps_file)
    : RemoteControllable("firfilter"), m_dev(dev), m_taps_file(taps_file)
{
    RC_ADD_PARAMETER(ntaps, "(Read-only) number of filter taps.");
    RC_ADD_PARAMETER(tapsfile, "Filename containing filter taps. When written to, the new file gets automatically loaded.");
    load_filter_taps(m_taps_file);
}

// taps file: number of taps, then one tap per line (reference src/FIRFilter.cpp:103-133)
void FirParameters::load_filter_taps(const std::string &tapsFile)
{
    std::vector<float> taps;
    if (tapsFile == "default") {
        check_dev(m_dev, dabgpu_set_fir_default_taps(m_dev));
        std::lock_guard<std::mutex> lock(m_taps_mutex);
        m_taps.assign(45, 0.f);
        return;
    }
    std::ifstream f(tapsFile.c_str());
    if (!f) throw std::runtime_error("FIRFilter: Could not open taps file " + tapsFile);
    int n = 0;
    f >> n;
    if (n <= 0) throw std::runtime_error("FIRFilter: taps file has invalid format.");
    taps.resize(n);
    for (int i = 0; i < n; ++i) {
        f >> taps[i];
        if (f.eof())
            throw std::runtime_error("FIRFilter: file " + tapsFile + " should contain " + std::to_string(n) +
                                     " taps, but EOF reached after " + std::to_string(i) + " taps!");
    }
    check_dev(m_dev, dabgpu_set_fir_taps(m_dev, taps.data(), taps.size()));
    std::lock_guard<std::mutex> lock(m_taps_mutex);
    m_taps = taps;
}

void FirParameters::set_parameter(const std::string &parameter, const std::string &value)
{
    if (parameter == "ntaps") throw ParameterError("Parameter 'ntaps' is read-only");
    if (parameter != "tapsfile") not_exported(parameter, get_rc_name());
    try {
        load_filter_taps(value);
        m_taps_file = value;
    } catch (const std::runtime_error &e) {
        throw ParameterError(e.what());
    }
}

const std::string FirParameters::get_parameter(const std::string &parameter) const
{
    std::lock_guard<std::mutex> lock(m_taps_mutex);
    if (parameter == "ntaps") return std::to_string(m_taps.size());
    if (parameter == "tapsfile") return m_taps_file;
    not_exported(parameter, get_rc_name());
}

const json::map_t FirParameters::get_all_values() const
{
    json::map_t m;
    std::lock_guard<std::mutex> lock(m_taps_mutex);
    m["ntaps"].v = static_cast<uint64_t>(m_taps.size());
    m["tapsfile"].v = m_taps_file;
    return m;
}
}  // namespace dabgpu_host

// ---------------------------------------------------------------- Resampler
Resampler::Resampler(size_t inputRate, size_t outputRate, size_t resolution)
    : m_ctx(dabgpu_host::mode_from_spacing(resolution))
{
    size_t a = inputRate, b = outputRate;
    while (b) { const size_t t = a % b; a = b; b = t; }
    m_L = outputRate / a;
    m_M = inputRate / a;
    m_ctx.check(dabgpu_set_resampler(m_ctx.get(), inputRate, outputRate));
}

int Resampler::process(Buffer *const dataIn, Buffer *dataOut)
{
    run_codec(m_ctx, dabgpu_resampler_process, dataIn, dataOut, dataIn->getLength() * m_L / m_M);
    return 1;
}

// ---------------------------------------------------------------- CicEqualizer
CicEqualizer::CicEqualizer(size_t nbCarriers, size_t spacing, int R)
    : m_ctx(dabgpu_host::mode_from_carriers(nbCarriers)), m_spacing(spacing), m_R(R)
{
}

int CicEqualizer::process(Buffer *const dataIn, Buffer *dataOut)
{
    dataOut->setLength(dataIn->getLength());
    size_t n = 0;
    m_ctx.check(dabgpu_cic_equalizer_process(m_ctx.get(), m_spacing, m_R, dataIn->getData(), dataIn->getLength(),
                                             dataOut->getData(), dataOut->getLength(), &n));
    return static_cast<int>(n / sizeof(complexf));
}

// ---------------------------------------------------------------- TII
namespace {
// the TIIError the reference constructor throws for a mode without TII (src/TII.cpp:144-149),
// before a device context is created
int tii_mode(unsigned int dabmode)
{
    if (dabmode != 1 && dabmode != 2)
        throw TIIError("TII::TII DAB mode " + std::to_string(dabmode) + " not valid!");
    return static_cast<int>(dabmode);
}
}  // namespace

TII::TII(unsigned int dabmode, tii_config_t &tii_config, bool fixedPoint)
    : OwnContext(tii_mode(dabmode)), TiiParameters(m_ctx.get(), tii_config)
{
    if (fixedPoint) throw std::runtime_error("TII: the fixed-point engine is not offloaded");
}

namespace dabgpu_host {
TiiParameters::TiiParameters(dabgpu_ctx *dev, tii_config_t &tii_config)
    : RemoteControllable("tii"), m_dev(dev), m_conf(tii_config)
{
    RC_ADD_PARAMETER(enable, "enable TII [0-1]");
    RC_ADD_PARAMETER(comb, "TII comb number [0-23]");
    RC_ADD_PARAMETER(pattern, "TII pattern number [0-69]");
    RC_ADD_PARAMETER(old_variant, "select old TII variant for old (buggy) receivers [0-1]");
    push_settings();
}

void TiiParameters::push_settings()
{
    if (dabgpu_set_tii(m_dev, m_conf.enable, m_conf.comb, m_conf.pattern, m_conf.old_variant) != DABGPU_OK)
        throw TIIError(dabgpu_last_error(m_dev));
}
}  // namespace dabgpu_host

const char *TII::name()
{
    // computed on demand: comb and pattern are RC-mutable (reference src/TII.cpp:159-170)
    std::lock_guard<std::mutex> lock(m_mutex);
    m_name = "TII(c:" + std::to_string(m_conf.comb) + " p:" + std::to_string(m_conf.pattern) +
             " vrnt:" + (m_conf.old_variant ? "old" : "new") + ")";
    return m_name.c_str();
}

int TII::process(Buffer *dataIn, Buffer *dataOut)
{
    if (dataIn == nullptr) throw TIIError("TII::process input size not valid!");
    dataOut->setLength(dataIn->getLength());
    size_t n = 0;
    if (dabgpu_tii_process(m_ctx.get(), dataIn->getData(), dataIn->getLength(), dataOut->getData(),
                           dataOut->getLength(), &n) != DABGPU_OK)
        throw TIIError(dabgpu_last_error(m_ctx.get()));
    dataOut->setLength(n);
    return 1;
}

namespace dabgpu_host {
void TiiParameters::set_parameter(const std::string &parameter, const std::string &value)
{
    std::stringstream ss(value);
    ss.exceptions(std::stringstream::failbit | std::stringstream::badbit);
    std::lock_guard<std::mutex> lock(m_mutex);
    if (parameter == "enable") {
        ss >> m_conf.enable;
    } else if (parameter == "pattern") {
        int v = 0;
        ss >> v;
        if (v < 0 || v > 69) throw TIIError("TII pattern not valid!");
        m_conf.pattern = v;
    } else if (parameter == "comb") {
        int v = 0;
        ss >> v;
        if (v < 0 || v > 23) throw TIIError("TII comb not valid!");
        m_conf.comb = v;
    } else if (parameter == "old_variant") {
        ss >> m_conf.old_variant;
    } else {
        not_exported(parameter, get_rc_name());
    }
    push_settings();
}

const std::string TiiParameters::get_parameter(const std::string &parameter) const
{
    std::lock_guard<std::mutex> lock(m_mutex);
    if (parameter == "enable") return m_conf.enable ? "1" : "0";
    if (parameter == "pattern") return std::to_string(m_conf.pattern);
    if (parameter == "comb") return std::to_string(m_conf.comb);
    if (parameter == "old_variant") return m_conf.old_variant ? "1" : "0";
    not_exported(parameter, get_rc_name());
    return "";
}

const json::map_t TiiParameters::get_all_values() const
{
    json::map_t m;
    std::lock_guard<std::mutex> lock(m_mutex);
    m["enable"].v = m_conf.enable;
    m["pattern"].v = static_cast<int64_t>(m_conf.pattern);
    m["comb"].v = static_cast<int64_t>(m_conf.comb);
    m["old_variant"].v = m_conf.old_variant;
    return m;
}
}  // namespace dabgpu_host

// ---------------------------------------------------------------- FormatConverter
namespace {
int format_code(const std::string &f)
{
    return f == "s16" ? DABGPU_FMT_S16 : f == "u8" ? DABGPU_FMT_U8 : f == "s8" ? DABGPU_FMT_S8 : 0;
}
}  // namespace

size_t FormatConverter::get_format_size(const std::string &format)
{
    const size_t n = dabgpu_format_size(format_code(format));
    if (!n) throw std::runtime_error("FormatConverter: Invalid format " + format);  // reference :203-205
    return n;
}

FormatConverter::FormatConverter(bool input_is_complexfix_wide, const std::string &format_out)
    : m_ctx(1), m_format_out(format_out)
{
    if (input_is_complexfix_wide)
        throw std::runtime_error("FormatConverter: the fixed-point engine is not offloaded");
}

int FormatConverter::process(Buffer *const dataIn, Buffer *dataOut)
{
    const int code = format_code(m_format_out);
    if (!code) throw std::runtime_error("FormatConverter: Invalid format " + m_format_out);
    const size_t n = dataIn->getLength() / sizeof(float);
    dataOut->setLength(n * dabgpu_format_size(code) / 2);
    size_t nb = 0, clipped = 0;
    m_ctx.check(dabgpu_format_process(m_ctx.get(), dataIn->getData(), dataIn->getLength(), code,
                                      dataOut->getData(), dataOut->getLength(), &nb, &clipped));
    m_num_clipped_samples.store(clipped);
    return static_cast<int>(dataOut->getLength());
}

// ---------------------------------------------------------------- MemlessPoly
MemlessPoly::MemlessPoly(std::string &coefs_file, unsigned int) : OwnContext(1), PolyParameters(m_ctx.get(), coefs_file)
{
    start_pipeline_thread();
}

MemlessPoly::~MemlessPoly() { stop_pipeline_thread(); }

int MemlessPoly::internal_process(Buffer *const dataIn, Buffer *dataOut)
{
    if (!settings_valid()) {
        // the reference passes the frame through when no valid settings are loaded
        *dataOut = *dataIn;
        return static_cast<int>(dataOut->getLength());
    }
    return run_codec(m_ctx, dabgpu_poly_process, dataIn, dataOut, dataIn->getLength());
}

namespace dabgpu_host {
PolyParameters::PolyParameters(dabgpu_ctx *dev, std::string &coefs_file)
    : RemoteControllable("memlesspoly"), m_dev(dev), m_coefs_file(coefs_file)
{
    RC_ADD_PARAMETER(ncoefs, "(Read-only) number of coefficients.");
    RC_ADD_PARAMETER(coefs, "Predistortion coefficients, same format as file.");
    RC_ADD_PARAMETER(coeffile, "Filename containing coefficients. When set, the file gets loaded.");
    std::ifstream f(m_coefs_file);
    load_coefficients(f);
}

bool PolyParameters::settings_valid() const
{
    std::lock_guard<std::mutex> lock(m_coefs_mutex);
    return m_valid;
}

// coefficient stream: format 1 = "1, 5, 5 AM values, 5 PM values"; format 2 = "2, scalefactor,
// 32 LUT values" (reference src/MemlessPoly.cpp:145-232)
void PolyParameters::load_coefficients(std::istream &in)
{
    if (!in) throw std::runtime_error("MemlessPoly: Could not open file with coefs!");
    uint32_t fmt = 0;
    in >> fmt;
    if (fmt == 1) {
        int n = 0;
        in >> n;
        if (n <= 0) throw std::runtime_error("MemlessPoly: coefs file has invalid format.");
        if (n != 5)
            throw std::runtime_error("MemlessPoly: invalid number of coefs: " + std::to_string(n) + " expected 5");
        std::vector<float> am(5), pm(5);
        for (int i = 0; i < 10; ++i) {
            float a;
            in >> a;
            (i < 5 ? am[i] : pm[i - 5]) = a;
            if (in.eof()) throw std::runtime_error("MemlessPoly: coefs file invalid !");
        }
        check_dev(m_dev, dabgpu_set_poly(m_dev, am.data(), pm.data()));
        std::lock_guard<std::mutex> lock(m_coefs_mutex);
        m_am = am; m_pm = pm; m_is_lut = false; m_valid = true;
    } else if (fmt == 2) {
        float scale = 0;
        in >> scale;
        std::vector<float> lut(32);
        for (auto &v : lut) in >> v;
        check_dev(m_dev, dabgpu_set_lut(m_dev, scale, lut.data()));
        std::lock_guard<std::mutex> lock(m_coefs_mutex);
        m_lut = lut; m_lut_scale = scale; m_is_lut = true; m_valid = true;
    } else {
        std::lock_guard<std::mutex> lock(m_coefs_mutex);
        m_valid = false;
    }
}

std::string PolyParameters::serialise_coefficients() const
{
    std::stringstream ss;
    std::lock_guard<std::mutex> lock(m_coefs_mutex);
    if (!m_valid) return ss.str();
    if (!m_is_lut) {
        ss << 1 << std::endl << m_am.size() << std::endl;
        for (float c : m_am) ss << c << std::endl;
        for (float c : m_pm) ss << c << std::endl;
    } else {
        // (the reference's table is std::array<complexf, 32>: operator<< prints "(re,im)", src/MemlessPoly.cpp:131-138)
        ss << 2 << std::endl << m_lut.size() << std::endl << m_lut_scale << std::endl;
        for (float c : m_lut) ss << std::complex<float>(c) << std::endl;
    }
    return ss.str();
}

void PolyParameters::set_parameter(const std::string &parameter, const std::string &value)
{
    if (parameter == "ncoefs") throw ParameterError("Parameter 'ncoefs' is read-only");
    if (parameter == "coefs") {
        std::stringstream ss(value);
        try {
            load_coefficients(ss);
            // "Write back to the file to ensure we will start up with the same settings next time": the
            // value as received (src/MemlessPoly.cpp:431-437)
            std::ofstream f(m_coefs_file);
            f << value;
        } catch (const std::runtime_error &e) {
            throw ParameterError(e.what());
        }
    } else if (parameter == "coeffile") {
        try {
            std::ifstream f(value);
            load_coefficients(f);
            m_coefs_file = value;
        } catch (const std::runtime_error &e) {
            throw ParameterError(e.what());
        }
    } else {
        not_exported(parameter, get_rc_name());
    }
}

const std::string PolyParameters::get_parameter(const std::string &parameter) const
{
    if (parameter == "ncoefs") {
        std::lock_guard<std::mutex> lock(m_coefs_mutex);
        return std::to_string(m_am.size());                  // the AM vector's size whatever the type (:453-454)
    }
    if (parameter == "coefs") return serialise_coefficients();
    if (parameter == "coeffile") return m_coefs_file;
    not_exported(parameter, get_rc_name());
}

const json::map_t PolyParameters::get_all_values() const
{
    json::map_t m;
    {
        std::lock_guard<std::mutex> lock(m_coefs_mutex);
        m["ncoefs"].v = m_am.size();
    }
    m["coefs"].v = serialise_coefficients();
    m["coeffile"].v = m_coefs_file;
    return m;
}
}  // namespace dabgpu_host

// ---------------------------------------------------------------- DabGpuChain
DabGpuChain::DabGpuChain(const Settings &s) : DabGpuChain(s, LiveSettings()) {}

DabGpuChain::~DabGpuChain() = default;

DabGpuChain::DabGpuChain(const Settings &s, const LiveSettings &live)
    : m_ctx(static_cast<int>(s.dabMode), static_cast<int>(std::max<size_t>(1, s.maxBatchFrames))), m_own(s)
{
    using namespace dabgpu_host;
    dabgpu_ctx *dev = m_ctx.get();
    dabgpu_geometry g;
    m_ctx.check(dabgpu_get_geometry(dev, &g));
    m_in_bytes = g.tf_input_bytes;
    if (s.emulatePipelineDrops > 3) throw std::runtime_error("DabGpuChain: emulatePipelineDrops is 0 ... 3");
    m_drops = s.emulatePipelineDrops;
    // each RC-mutable value lives in the caller's settings where LiveSettings points there, else in m_own -- the stage
    // parameter objects hold references, like the reference's stage classes (src/DabModulator.cpp:195-260)
    GainMode &gainMode = live.gainMode ? *live.gainMode : m_own.gainMode;
    float &digitalGain = live.digitalGain ? *live.digitalGain : m_own.digitalGain;
    float &variance = live.gainmodeVariance ? *live.gainmodeVariance : m_own.gainmodeVariance;
    std::string &tapsFile = live.filterTapsFilename ? *live.filterTapsFilename : m_own.filterTapsFilename;
    std::string &coefFile = live.polyCoefFilename ? *live.polyCoefFilename : m_own.polyCoefFilename;
    size_t &overlap = live.ofdmWindowOverlap ? *live.ofdmWindowOverlap : m_own.ofdmWindowOverlap;
    tii_config_t &tii = live.tiiConfig ? *live.tiiConfig : m_own.tiiConfig;
    bool &cfr = live.enableCfr ? *live.enableCfr : m_own.enableCfr;
    float &cfrClip = live.cfrClip ? *live.cfrClip : m_own.cfrClip;
    float &cfrErrorClip = live.cfrErrorClip ? *live.cfrErrorClip : m_own.cfrErrorClip;

    // the stages, in the reference's order of construction (src/DabModulator.cpp:178-268)
    if (s.dabMode == 1 || s.dabMode == 2)       // TII::TII throws TIIError in the other modes (src/TII.cpp:144-149): no "tii"
        m_rc_tii.reset(new TiiParameters(dev, tii));
    else if (tii.enable)
        throw TIIError("TII::TII DAB mode " + std::to_string(s.dabMode) + " not valid!");
    m_rc_ofdm.reset(new OfdmControl(dev, static_cast<size_t>(g.nb_symbols) + 1, cfr, cfrClip, cfrErrorClip));
    m_rc_ofdm->push_settings();
    if (s.enableGain) {
        m_mask |= DABGPU_STAGE_GAIN;
        m_rc_gain.reset(new GainParameters(dev, gainMode, digitalGain, s.normalise, variance));
        m_rc_gain->push_settings();
    }
    if (s.referenceGainRounding) m_ctx.check(dabgpu_set_gain_rounding(dev, DABGPU_GAIN_ROUNDING_REFERENCE));
    m_rc_guard.reset(new GuardParameters(dev, overlap));
    if (!tapsFile.empty()) {
        m_mask |= DABGPU_STAGE_FIR;
        m_rc_fir.reset(new FirParameters(dev, tapsFile));
    }
    if (s.outputFormat != "complexf") {
        const int code = format_code(s.outputFormat);
        if (!code) throw std::runtime_error("FormatConverter: Invalid format " + s.outputFormat);
        m_ctx.check(dabgpu_set_output_format(dev, code));
    }
    if (s.outputRate != 2048000) {
        m_mask |= DABGPU_STAGE_RESAMPLE;
        m_ctx.check(dabgpu_set_resampler(dev, 2048000, s.outputRate));
    }
    if (!coefFile.empty()) m_rc_poly.reset(new PolyParameters(dev, coefFile));   // (in the mask while its settings are valid)
}

std::vector<RemoteControllable *> DabGpuChain::remote_controllables() const
{
    std::vector<RemoteControllable *> v;
    for (RemoteControllable *p : {static_cast<RemoteControllable *>(m_rc_tii.get()),
                                  static_cast<RemoteControllable *>(m_rc_ofdm.get()),
                                  static_cast<RemoteControllable *>(m_rc_gain.get()),
                                  static_cast<RemoteControllable *>(m_rc_guard.get()),
                                  static_cast<RemoteControllable *>(m_rc_fir.get()),
                                  static_cast<RemoteControllable *>(m_rc_poly.get())})
        if (p) v.push_back(p);
    return v;
}

unsigned DabGpuChain::stage_mask()
{
    // MemlessPoly passes frames through while no valid coefficients are loaded (src/MemlessPoly.cpp:397-409)
    return m_mask | (m_rc_poly && m_rc_poly->settings_valid() ? static_cast<unsigned>(DABGPU_STAGE_POLY) : 0u);
}

void DabGpuChain::before_frames()
{
    // parameters whose home is a mod_settings_t field the remote control may also reach through another object are
    // pushed every time (a setter that repeats the current value is free)
    m_cfr_on = m_rc_ofdm->push_settings();
    if (m_rc_gain) m_rc_gain->push_settings();
}

void DabGpuChain::after_frames()
{
    if (m_cfr_on) m_rc_ofdm->collect_statistics();
}

void DabGpuChain::submit(const void *bits, size_t n_frames)
{
    // (the start-up emulation lives in process(): a streaming caller that wants the reference's frame count holds the
    // frames back itself, as dabmod_file --batch --reference-latency does -- never silently N frames where N - k were asked)
    if (m_drops)
        throw std::runtime_error("DabGpuChain::submit: Settings::emulatePipelineDrops applies to process() only");
    before_frames();
    m_ctx.check(dabgpu_chain_submit(m_ctx.get(), static_cast<const uint8_t *>(bits), n_frames, stage_mask()));
}

size_t DabGpuChain::collect(const void **iq)
{
    size_t n = 0;
    m_ctx.check(dabgpu_chain_collect(m_ctx.get(), iq, &n));
    return n;
}

size_t DabGpuChain::output_bytes_per_frame() const
{
    return dabgpu_chain_out_bytes_per_frame(m_ctx.get(), const_cast<DabGpuChain *>(this)->stage_mask());
}

size_t DabGpuChain::get_num_clipped_samples() const
{
    size_t n = 0;
    m_ctx.check(dabgpu_get_num_clipped(m_ctx.get(), &n));
    return n;
}

// The reference's pipelined stages delay the metadata with the frame, one call each (src/ModPlugin.cpp:117-128; a stage
// behind one that returned 0 is not reached in that round, src/Flowgraph.cpp:334-336): k of them hand the metadata of
// call i - k to the sink together with frame i - k.  Same here, as one FIFO k deep.
meta_vec_t DabGpuChain::process_metadata(const meta_vec_t &metadataIn)
{
    if (!m_drops) return metadataIn;
    m_delayed_meta.push_back(metadataIn);
    if (m_delayed_meta.size() <= m_drops) return {};
    meta_vec_t r = std::move(m_delayed_meta.front());
    m_delayed_meta.pop_front();
    return r;
}

int DabGpuChain::process(Buffer *const dataIn, Buffer *dataOut)
{
    if (dataIn->getLength() != m_in_bytes)
        throw std::runtime_error("DabGpuChain::process input size not valid!");
    const Buffer *in = dataIn;
    Buffer oldest;
    if (m_drops) {
        // the reference's pipelined stages: this call's frame goes in, the frame of m_drops calls ago comes out
        m_delayed.emplace_back(dataIn->getLength(), dataIn->getData());
        if (m_delayed.size() <= m_drops) {
            dataOut->setLength(0);
            return 0;
        }
        oldest = std::move(m_delayed.front());
        m_delayed.pop_front();
        in = &oldest;
    }
    before_frames();
    const unsigned mask = stage_mask();
    dataOut->setLength(dabgpu_chain_out_bytes_per_frame(m_ctx.get(), mask));
    size_t n = 0;
    m_ctx.check(dabgpu_chain_process(m_ctx.get(), static_cast<const uint8_t *>(in->getData()), 1, mask,
                                     dataOut->getData(), dataOut->getLength(), &n));
    dataOut->setLength(n);
    after_frames();
    return static_cast<int>(n);
}
