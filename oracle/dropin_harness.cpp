/*
 * dropin_harness.cpp -- the REFERENCE's own graph builder on the MI355X drop-ins.
 *
 * TEST INFRASTRUCTURE (ours).  Built by `make -C oracle dropin` into oracle/_ref/dabmod_dropin, in this container only:
 * the reference's UNMODIFIED DabModulator.cpp, Flowgraph.cpp, EtiReader.cpp, InputFileReader.cpp, OutputFile.cpp,
 * OutputMemory.cpp and its whole channel-coding front end are compiled from where they lie (through a directory of
 * symbolic links in which the fifteen stage headers are the forwarding headers of
 * odr-dabmod_amd/host/install_dropins.sh), and linked with GpuStages.cpp and libdabgpu.so.  Nothing FFTW-dependent is
 * left in that build: OfdmGenerator.cpp and Resampler.cpp are exactly the files the drop-ins replace.
 *
 * What it proves ("Construction signatures the drop-ins must keep so DabModulator.cpp compiles
 * unchanged"): DabModulator::process -- src/DabModulator.cpp:125-424 as written -- builds its inner flowgraph from the
 * GPU classes and modulates an ETI file.  main() below is the loop of run_modulator (src/DabMod.cpp:593-724) reduced to
 * its file-input branch: GetNextFrame -> loadEtiData -> FP == 0 start gate -> Flowgraph::run.
 *
 *   dabmod_dropin in.eti out.iq [--fir default|FILE] [--rate HZ] [--poly FILE] [--gainmode fix|max|var] [--normalise X]
 *                               [--window N] [--format s16|u8|s8] [--engine fftw|kiss] [--mode 1..4] [--digital X] [--var X]
 *                               [--tii COMB,PATTERN] [--cfr CLIP,ERRORCLIP] [--loop N]
 *                               [--rc N,NAME,PARAM,VALUE]...  [--show-metadata 1] [--reference-latency 1]
 *
 * The same main() built with -DDABGPU_FUSED_BUILD over the tree install_fused.sh has edited is oracle/_ref/dabmod_fused
 * (`make -C oracle fused`): INTEGRATION.md section A inside the reference's own DabModulator.cpp / Flowgraph.cpp /
 * OutputMemory.cpp -- ONE DabGpuChain node, its remote-controllables enrolled in the reference's registry `rcs`
 * (lib/RemoteControl.cpp, compiled from where it lies).
 *   --rc N,NAME,PARAM,VALUE   rcs.set_param(NAME, PARAM, VALUE) -- what the telnet / ZMQ remote control does
 *                             (lib/RemoteControl.cpp:150-160) -- just before the N-th ETI frame (from 0, counted behind
 *                             the start gate) enters the flowgraph; VALUE "@file" takes the value from a file.  After the
 *                             run, "rc NAME PARAM = ..." lines on stderr read every set parameter back (rcs.get_param).
 *   --show-metadata 1         one line "meta K: FCT FCT FCT FCT" on stdout per frame the sink writes: the frame counts of the
 *                             metadata that ARRIVED WITH output frame K (src/Flowgraph.cpp:146-175 moves it along the edges)
 *   --reference-latency 1     (fused build) mod_settings_t::gpuReferenceLatency: the reference's start-up frame count
 *   --reference-gain 1        (fused build) mod_settings_t::gpuReferenceGain: gain mode var by the reference's recurrence
 */
#include "DabModulator.h"
#include "EtiReader.h"
#include "Flowgraph.h"
#include "InputReader.h"
#include "Log.h"
#include "OutputFile.h"
#include "RemoteControl.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

namespace {
struct RcAction {
    unsigned long at;
    std::string name, param, value;
};

// the reference's file sink, reporting the metadata each written frame came with
class ReportingOutputFile : public OutputFile {
public:
    ReportingOutputFile(const std::string &filename, bool report) : OutputFile(filename, false), m_report(report) {}
    meta_vec_t process_metadata(const meta_vec_t &metadataIn) override
    {
        if (m_report) {
            printf("meta %lu:", m_frames++);
            for (const auto &md : metadataIn) printf(" %d", (int)md.ts.fct);
            printf("\n");
        }
        return OutputFile::process_metadata(metadataIn);
    }

private:
    bool m_report;
    unsigned long m_frames = 0;
};
}  // namespace

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: %s in.eti out.iq [options]\n", argv[0]);
        return 2;
    }
    mod_settings_t s;
    s.inputName = argv[1];
    s.outputName = argv[2];
    s.useFileOutput = true;
    s.showProcessTime = false;
    std::string format;
    std::vector<RcAction> rc_actions;
    bool show_metadata = false;
    int loops = 1;
    for (int i = 3; i + 1 < argc; i += 2) {
        const std::string k = argv[i], v = argv[i + 1];
        if (k == "--fir") s.filterTapsFilename = v;
        else if (k == "--rate") s.outputRate = strtoul(v.c_str(), nullptr, 10);
        else if (k == "--poly") s.polyCoefFilename = v;
        else if (k == "--normalise") s.normalise = strtof(v.c_str(), nullptr);
        else if (k == "--window") s.ofdmWindowOverlap = strtoul(v.c_str(), nullptr, 10);
        else if (k == "--format") format = v;
        else if (k == "--gainmode") s.gainMode = v == "fix" ? GainMode::GAIN_FIX : v == "max" ? GainMode::GAIN_MAX : GainMode::GAIN_VAR;
        else if (k == "--engine") s.fftEngine = v == "kiss" ? FFTEngine::KISS : FFTEngine::FFTW;
        else if (k == "--digital") s.digitalgain = strtof(v.c_str(), nullptr);
        else if (k == "--var") s.gainmodeVariance = strtof(v.c_str(), nullptr);
        else if (k == "--mode") s.dabMode = strtoul(v.c_str(), nullptr, 10);
        else if (k == "--show-metadata") show_metadata = v != "0";
        else if (k == "--loop") loops = atoi(v.c_str());        // the file again and again (src/DabMod.cpp:695-706, `loop`)
        else if (k == "--tii") { s.tiiConfig.enable = true; sscanf(v.c_str(), "%d,%d", &s.tiiConfig.comb, &s.tiiConfig.pattern); }
        else if (k == "--cfr") { s.enableCfr = true; sscanf(v.c_str(), "%f,%f", &s.cfrClip, &s.cfrErrorClip); }
#ifdef DABGPU_FUSED_BUILD
        else if (k == "--reference-latency") s.gpuReferenceLatency = v != "0";
        else if (k == "--reference-gain") s.gpuReferenceGain = v != "0";
#endif
        else if (k == "--rc") {
            RcAction a;
            std::stringstream ss(v);
            std::string at;
            std::getline(ss, at, ',');
            std::getline(ss, a.name, ',');
            std::getline(ss, a.param, ',');
            std::getline(ss, a.value);
            a.at = strtoul(at.c_str(), nullptr, 10);
            if (!a.value.empty() && a.value[0] == '@') {
                std::ifstream f(a.value.substr(1));
                std::stringstream body;
                body << f.rdbuf();
                a.value = body.str();
            }
            rc_actions.push_back(a);
        }
        else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    try {
        InputFileReader reader;
        if (reader.Open(s.inputName, false) != 0) throw std::runtime_error("cannot open " + s.inputName);
        EtiReader eti(s.tist_offset_s);
        auto output = std::make_shared<ReportingOutputFile>(s.outputName, show_metadata);
        Flowgraph flowgraph(s.showProcessTime);
        auto modulator = std::make_shared<DabModulator>(eti, s, format);   // src/DabMod.cpp:532
        flowgraph.connect(modulator, output);                              // :536
        Buffer data;
        data.setLength(6144);
        int last_fct = -1;
        unsigned long frames = 0;
        for (int pass = 0;;) {
            const int framesize = reader.GetNextFrame(data.getData());
            if (framesize <= 0) {
                if (++pass >= loops || reader.Open(s.inputName, false) != 0) break;
                continue;
            }
            if ((size_t)eti.loadEtiData(data) != data.getLength()) throw std::runtime_error("ETI read error");
            if (last_fct == -1 && eti.getFp() != 0) continue;              // :684-693
            last_fct = (int)eti.getFct();
            for (const auto &a : rc_actions)
                if (a.at == frames) rcs.set_param(a.name, a.param, a.value);
            frames++;
            flowgraph.run();                                                // :711
        }
        for (const auto &a : rc_actions) {
            std::string v = rcs.get_param(a.name, a.param);
            for (auto &c : v) if (c == '\n') c = ' ';
            fprintf(stderr, "rc %s %s = %s\n", a.name.c_str(), a.param.c_str(), v.c_str());
        }
        fprintf(stderr, "dabmod_dropin: %lu ETI frames modulated\n", frames);
    }
    catch (const std::exception &e) {
        fprintf(stderr, "dabmod_dropin: %s\n", e.what());
        return 1;
    }
    return 0;
}
