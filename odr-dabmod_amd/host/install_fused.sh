#!/bin/sh
# install_fused.sh <dir holding a copy of ODR-DabMod's src/ on which install_dropins.sh has run>
#
# INTEGRATION.md section A as a scripted edit: the production shape, ONE DabGpuChain node in place of the sub-graph
# cifMap .. cifPoly (+ FormatConverter) of DabModulator::process.  Three files of the copy are rewritten FROM THE
# ORIGINALS THEY ARE (this script holds no text of theirs beyond the anchor lines it looks for, and stops if an anchor is
# not found exactly once -- a different upstream version needs a look, not a guess):
#
#   DabModulator.cpp  the block that creates the stage objects (from "const bool fixedPoint = ..." to the line before
#                     "m_output = make_shared<OutputMemory>(dataOut);") becomes the construction of the chain -- the
#                     RC-mutable values stay in mod_settings_t, by pointer, as the reference's stages take them by reference
#                     -- and the enrolment of its remote-controllables; the wiring from "connect(cifPart, cifMap)" to the
#                     end of the plugin loop becomes two connects; "num_clipped_samples" reads the chain.
#   DabModulator.h    one member more: std::shared_ptr<DabGpuChain> m_gpuChain.
#   ConfigParser.h    two settings more in mod_settings_t: gpuReferenceLatency (the reference's start-up frame count,
#                     DabGpuChain::Settings::emulatePipelineDrops) and gpuReferenceGain (gain mode var by the reference's
#                     running recurrence, DabGpuChain::Settings::referenceGainRounding); both default off.
set -e
dst=${1:?usage: install_fused.sh <copy of the reference src/ with the drop-in headers installed>}
[ -f "$dst/GpuStages.h" ] || { echo "run install_dropins.sh on $dst first" >&2; exit 1; }

once() {        # once <file> <fixed string>: the anchor must be there exactly once
    n=$(grep -cF -- "$2" "$1" || true)
    [ "$n" = 1 ] || { echo "install_fused.sh: anchor '$2' found $n times in $1" >&2; exit 1; }
}

# work on real copies (the directory may hold symbolic links into the original tree: never write through them)
for f in DabModulator.cpp DabModulator.h ConfigParser.h; do
    cp -L "$dst/$f" "$dst/$f.orig.$$"
    rm -f "$dst/$f"
done
cpp="$dst/DabModulator.cpp.orig.$$"; hdr="$dst/DabModulator.h.orig.$$"; cfg="$dst/ConfigParser.h.orig.$$"

A1='const bool fixedPoint = m_settings.fftEngine != FFTEngine::FFTW;'
A2='m_output = make_shared<OutputMemory>(dataOut);'
A3='m_flowgraph->connect(cifPart, cifMap);'
A4='etiLog.level(debug) << "DabModulator set up.";'
A5='if (m_formatConverter) {'
A6='ss << m_formatConverter->get_num_clipped_samples();'
A7='m_formatConverter ? m_formatConverter->get_num_clipped_samples() : 0;'
for a in "$A1" "$A2" "$A3" "$A4" "$A5" "$A6" "$A7"; do once "$cpp" "$a"; done
once "$hdr" 'std::shared_ptr<FormatConverter> m_formatConverter;'
once "$cfg" 'bool showProcessTime = true;'

awk -v a1="$A1" -v a2="$A2" -v a3="$A3" -v a4="$A4" -v a5="$A5" -v a6="$A6" -v a7="$A7" '
function construct() {
    print "        // ---- MI355X: one fused plugin for cifMap .. cifPoly + FormatConverter (INTEGRATION.md section A; install_fused.sh)"
    print "        if (m_settings.fftEngine != FFTEngine::FFTW)"
    print "            throw std::runtime_error(\"OfdmGenerator: the fixed-point engine (fft_engine=kiss) is not offloaded to the GPU; \""
    print "                                     \"set fft_engine=fftw\");"
    print "        if (m_settings.clockRate)"
    print "            throw std::runtime_error(\"DabGpuChain: the CIC equaliser is not part of the fused chain (use the per-stage drop-ins)\");"
    print "        DabGpuChain::Settings gs;"
    print "        gs.dabMode = mode;"
    print "        gs.gainMode = m_settings.gainMode;"
    print "        gs.digitalGain = m_settings.digitalgain;"
    print "        gs.normalise = m_settings.normalise;"
    print "        gs.gainmodeVariance = m_settings.gainmodeVariance;"
    print "        gs.filterTapsFilename = m_settings.filterTapsFilename;"
    print "        gs.outputRate = m_settings.outputRate;"
    print "        gs.polyCoefFilename = m_settings.polyCoefFilename;"
    print "        gs.ofdmWindowOverlap = m_settings.ofdmWindowOverlap;"
    print "        gs.tiiConfig = m_settings.tiiConfig;"
    print "        gs.enableCfr = m_settings.enableCfr;"
    print "        gs.cfrClip = m_settings.cfrClip;"
    print "        gs.cfrErrorClip = m_settings.cfrErrorClip;"
    print "        gs.outputFormat = m_format.empty() ? \"complexf\" : m_format;"
    print "        if (m_settings.gpuReferenceLatency) gs.emulatePipelineDrops = gs.referencePipelineDepth();"
    print "        gs.referenceGainRounding = m_settings.gpuReferenceGain;"
    print "        DabGpuChain::LiveSettings live;     // the remote control writes where the reference keeps these values"
    print "        live.gainMode = &m_settings.gainMode;"
    print "        live.digitalGain = &m_settings.digitalgain;"
    print "        live.gainmodeVariance = &m_settings.gainmodeVariance;"
    print "        live.filterTapsFilename = &m_settings.filterTapsFilename;"
    print "        live.polyCoefFilename = &m_settings.polyCoefFilename;"
    print "        live.ofdmWindowOverlap = &m_settings.ofdmWindowOverlap;"
    print "        live.tiiConfig = &m_settings.tiiConfig;"
    print "        live.enableCfr = &m_settings.enableCfr;"
    print "        live.cfrClip = &m_settings.cfrClip;"
    print "        live.cfrErrorClip = &m_settings.cfrErrorClip;"
    print "        m_gpuChain = make_shared<DabGpuChain>(gs, live);"
    print "        for (auto *controllable : m_gpuChain->remote_controllables()) rcs.enrol(controllable);"
    print ""
}
{
    if (index($0, a1)) { construct(); skip = 1 }
    if (index($0, a2)) skip = 0
    if (index($0, a3)) {
        print "        m_flowgraph->connect(cifPart, m_gpuChain);"
        print "        m_flowgraph->connect(m_gpuChain, m_output);"
        skip = 1
    }
    if (index($0, a4)) skip = 0
    if (skip) next
    if (index($0, a5)) { sub(/m_formatConverter/, "m_gpuChain and not m_format.empty()") }
    else if (index($0, a6)) { sub(/m_formatConverter/, "m_gpuChain") }
    else if (index($0, a7)) { gsub(/m_formatConverter/, "m_gpuChain") }
    print
}' "$cpp" > "$dst/DabModulator.cpp"

awk '{ print } index($0, "std::shared_ptr<FormatConverter> m_formatConverter;") { print "    std::shared_ptr<DabGpuChain> m_gpuChain;   // install_fused.sh" }' \
    "$hdr" > "$dst/DabModulator.h"
awk '{ print } index($0, "bool showProcessTime = true;") { print "    bool gpuReferenceLatency = false;   // install_fused.sh: DabGpuChain::Settings::emulatePipelineDrops"; print "    bool gpuReferenceGain = false;      // install_fused.sh: DabGpuChain::Settings::referenceGainRounding" }' \
    "$cfg" > "$dst/ConfigParser.h"
rm -f "$cpp" "$hdr" "$cfg"
