// dabmod_file -- ETI file in, IQ file out: BASELINE config 1 ("Mode I, native 2.048 Msps, ETI file ->
// IQ file") as one command, the file-to-file shape of the reference's odr-dabmod
// (src/DabMod.cpp:365-520 for the wiring, doc/example.ini for the options).
//
//   InputFileReader -> EtiFrontend (CPU: ETI -> coded bits)        odr-dabmod_amd/host/Frontend.h
//                   -> DabGpuChain (MI355X: coded bits -> IQ)       odr-dabmod_amd/host/GpuStages.h
//                   -> [FormatConverter] -> file
//
// usage: dabmod_file <in.eti> <out> [options]
//   --mode N             transmission mode 1..4 (default: from the ETI header, 0 -> 4)
//   --format F           complexf (default) | s16 | u8 | s8
//   --gainmode M         var (default) | fix | max        --digital G   --normalise X   --var V
//   --fir none|default|<tapsfile>      (default: none, as config 1)
//   --rate R             output sample rate (default 2048000)
//   --poly <coeffile>    MemlessPoly coefficient file
//   --ofdmwindowing W    raised-cosine overlap in samples
//   --tii comb,pattern   --cfr clip,errorclip
//   --loop N             read the file N times
//   --bits-only          stop after the front-end: write the hot path's input blocks (no GPU needed)
//   --reference-latency  emit exactly the frames the reference emits: one transmission frame fewer per pipelined stage
//                        of the equivalent reference graph (GainControl, FIRFilter, MemlessPoly: src/ModPlugin.cpp:90-115)
//   --reference-gain     gain mode var: the reference's running fp32 recurrence (src/GainControl.cpp:251-340) instead of the
//                        exact variance (DabGpuChain::Settings::referenceGainRounding)
#include "Frontend.h"
#include "GpuStages.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

namespace {
[[noreturn]] void usage()
{
    std::fprintf(stderr, "usage: dabmod_file <in.eti> <out> [--mode N] [--format complexf|s16|u8|s8] [--gainmode var|fix|max]\n"
                         "       [--digital G] [--normalise X] [--var V] [--fir none|default|file] [--rate R] [--poly file]\n"
                         "       [--ofdmwindowing W] [--tii comb,pattern] [--cfr clip,errorclip] [--loop N] [--bits-only]\n"
                         "       [--batch N]   N transmission frames per GPU call, two calls in flight (default 1: frame by frame)\n"
                         "       [--reference-latency]   drop the frames the reference's pipelined stages never emit\n"
                         "       [--reference-gain]      gain mode var by the reference's running recurrence (bit-equal scalars, slower)\n");
    std::exit(2);
}
}  // namespace

int main(int argc, char **argv)
{
    if (argc < 3) usage();
    const std::string in_path = argv[1], out_path = argv[2];
    DabGpuChain::Settings gs;
    gs.dabMode = 0;
    std::string format = "complexf";
    bool separate_converter = false;
    int loops = 1;
    bool bits_only = false;
    size_t batch = 1;
    bool reference_latency = false;
    try {
        for (int i = 3; i < argc; ++i) {
            const std::string a = argv[i];
            auto val = [&]() -> std::string {
                if (i + 1 >= argc) usage();
                return argv[++i];
            };
            if (a == "--mode") gs.dabMode = static_cast<unsigned>(std::stoul(val()));
            else if (a == "--format") format = val();
            else if (a == "--separate-converter") separate_converter = true;
            else if (a == "--gainmode") {
                const std::string m = val();
                gs.gainMode = m == "fix" ? GainMode::GAIN_FIX : m == "max" ? GainMode::GAIN_MAX : GainMode::GAIN_VAR;
            }
            else if (a == "--digital") gs.digitalGain = std::stof(val());
            else if (a == "--normalise") gs.normalise = std::stof(val());
            else if (a == "--var") gs.gainmodeVariance = std::stof(val());
            else if (a == "--fir") { const std::string f = val(); gs.filterTapsFilename = f == "none" ? "" : f; }
            else if (a == "--rate") gs.outputRate = std::stoul(val());
            else if (a == "--poly") gs.polyCoefFilename = val();
            else if (a == "--ofdmwindowing") gs.ofdmWindowOverlap = std::stoul(val());
            else if (a == "--tii") {
                if (std::sscanf(val().c_str(), "%d,%d", &gs.tiiConfig.comb, &gs.tiiConfig.pattern) != 2) usage();
                gs.tiiConfig.enable = true;
            }
            else if (a == "--cfr") {
                if (std::sscanf(val().c_str(), "%f,%f", &gs.cfrClip, &gs.cfrErrorClip) != 2) usage();
                gs.enableCfr = true;
            }
            else if (a == "--loop") loops = std::atoi(val().c_str());
            else if (a == "--bits-only") bits_only = true;
            else if (a == "--batch") batch = std::max<size_t>(1, std::stoul(val()));
            else if (a == "--reference-latency") reference_latency = true;
            else if (a == "--reference-gain") gs.referenceGainRounding = true;
            else usage();
        }

        InputFileReader reader;
        if (reader.Open(in_path, false) != 0) {
            std::fprintf(stderr, "dabmod_file: cannot read %s as an ETI file\n", in_path.c_str());
            return 1;
        }
        std::fprintf(stderr, "%s\n", reader.GetPrintableInfo().c_str());
        std::ofstream out(out_path, std::ios::binary);
        if (!out) {
            std::fprintf(stderr, "dabmod_file: cannot write %s\n", out_path.c_str());
            return 1;
        }

        std::unique_ptr<EtiFrontend> frontend;
        std::unique_ptr<DabGpuChain> chain;
        std::unique_ptr<FormatConverter> converter;
        Buffer bits, iq, converted;
        uint8_t frame[6144];
        size_t n_eti = 0, n_tf = 0, clipped = 0;
        std::vector<uint8_t> pending;             // --batch: hot-path input of the batch being filled
        std::deque<std::vector<uint8_t>> held;    // --batch with --reference-latency: the frames "inside the pipeline"
        size_t n_out = 0;                         // transmission frames written
        int in_flight = 0;
        auto drain_one = [&]() {
            const void *p = nullptr;
            const size_t n = chain->collect(&p);
            n_out += n / chain->output_bytes_per_frame();
            if (format != "complexf") clipped += chain->get_num_clipped_samples();
            out.write(static_cast<const char *>(p), static_cast<std::streamsize>(n));
            --in_flight;
        };
        for (int l = 0; l < loops; ++l) {
            if (l && reader.Open(in_path, false) != 0) return 1;
            int got;
            while ((got = reader.GetNextFrame(frame)) == 6144) {
                ++n_eti;
                if (!frontend) {
                    if (gs.dabMode == 0) {
                        // MID of the first frame; 0 means mode IV (EN 300 799 5.3.2)
                        const unsigned mid = (frame[6] >> 3) & 3;
                        gs.dabMode = mid ? mid : 4;
                    }
                    frontend.reset(new EtiFrontend(gs.dabMode));
                }
                if (!frontend->push(frame, bits)) continue;
                ++n_tf;
                if (bits_only) {
                    out.write(static_cast<const char *>(bits.getData()), static_cast<std::streamsize>(bits.getLength()));
                    continue;
                }
                if (batch > 1 && !separate_converter) {
                    // streaming shape: the front-end fills a batch while the GPU works on the previous two
                    if (!chain) {
                        gs.outputFormat = format;
                        gs.maxBatchFrames = batch;
                        chain.reset(new DabGpuChain(gs));
                        pending.reserve(batch * bits.getLength());
                    }
                    const uint8_t *b = static_cast<const uint8_t *>(bits.getData());
                    if (reference_latency) {
                        // frame i is modulated when frame i + k has arrived; the last k frames never are (see --reference-latency)
                        held.emplace_back(b, b + bits.getLength());
                        if (held.size() <= gs.referencePipelineDepth()) continue;
                        pending.insert(pending.end(), held.front().begin(), held.front().end());
                        held.pop_front();
                    } else
                    pending.insert(pending.end(), b, b + bits.getLength());
                    if (pending.size() == batch * chain->input_bytes_per_frame()) {
                        if (in_flight == 2) { drain_one(); }
                        chain->submit(pending.data(), batch);
                        ++in_flight;
                        pending.clear();
                    }
                    continue;
                }
                if (!chain) {
                    // the output format is the chain's own last step (stored by its last kernel for s16); the
                    // stand-alone FormatConverter plugin stays available with --separate-converter
                    if (!separate_converter) gs.outputFormat = format;
                    if (reference_latency) gs.emulatePipelineDrops = gs.referencePipelineDepth();
                    chain.reset(new DabGpuChain(gs));
                    if (separate_converter && format != "complexf") converter.reset(new FormatConverter(false, format));
                }
                if (chain->process(&bits, &iq) == 0) continue;       // (a frame inside the emulated pipeline: nothing yet)
                ++n_out;
                const Buffer *o = &iq;
                if (converter) {
                    converter->process(&iq, &converted);
                    clipped += converter->get_num_clipped_samples();
                    o = &converted;
                } else if (format != "complexf") {
                    clipped += chain->get_num_clipped_samples();
                }
                out.write(static_cast<const char *>(o->getData()), static_cast<std::streamsize>(o->getLength()));
            }
            if (got < 0) {
                std::fprintf(stderr, "dabmod_file: error while reading %s\n", in_path.c_str());
                return 1;
            }
        }
        if (chain && batch > 1 && !separate_converter) {
            // the tail: a last, shorter batch, then whatever is still in flight, in order
            const size_t rest = pending.size() / chain->input_bytes_per_frame();
            if (rest) {
                if (in_flight == 2) drain_one();
                chain->submit(pending.data(), rest);
                ++in_flight;
            }
            while (in_flight) drain_one();
        }
        if (bits_only) n_out = n_tf;
        std::fprintf(stderr, "dabmod_file: %zu ETI frames -> %zu transmission frames in, %zu out (mode %u)", n_eti, n_tf, n_out, gs.dabMode);
        if (format != "complexf") std::fprintf(stderr, ", %zu clipped components", clipped);
        std::fprintf(stderr, "\n");
        std::printf("%zu %zu %zu\n", n_eti, n_tf, n_out);
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "dabmod_file: %s\n", e.what());
        return 1;
    }
}
