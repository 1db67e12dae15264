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
 *                               [--window N] [--format s16|u8|s8] [--engine fftw|kiss]
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
#include <memory>
#include <string>

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
        else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    try {
        InputFileReader reader;
        if (reader.Open(s.inputName, false) != 0) throw std::runtime_error("cannot open " + s.inputName);
        EtiReader eti(s.tist_offset_s);
        auto output = std::make_shared<OutputFile>(s.outputName, false);
        Flowgraph flowgraph(s.showProcessTime);
        auto modulator = std::make_shared<DabModulator>(eti, s, format);   // src/DabMod.cpp:532
        flowgraph.connect(modulator, output);                              // :536
        Buffer data;
        data.setLength(6144);
        int last_fct = -1;
        unsigned long frames = 0;
        for (;;) {
            const int framesize = reader.GetNextFrame(data.getData());
            if (framesize <= 0) break;
            if ((size_t)eti.loadEtiData(data) != data.getLength()) throw std::runtime_error("ETI read error");
            if (last_fct == -1 && eti.getFp() != 0) continue;              // :684-693
            last_fct = (int)eti.getFct();
            frames++;
            flowgraph.run();                                                // :711
        }
        fprintf(stderr, "dabmod_dropin: %lu ETI frames modulated\n", frames);
    }
    catch (const std::exception &e) {
        fprintf(stderr, "dabmod_dropin: %s\n", e.what());
        return 1;
    }
    return 0;
}
