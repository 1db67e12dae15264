#!/bin/sh
# install_dropins.sh <dir holding a copy of ODR-DabMod's src/>
#
# Turns a COPY of the reference's src/ into the per-stage drop-in build of INTEGRATION.md section B: the fifteen stage
# headers become forwarding headers to GpuStages.h, their .cpp files go away (GpuStages.cpp defines the classes), and
# GpuStages.{h,cpp} are put beside them.  DabModulator.cpp is NOT touched: it compiles as it is
# (tests/test_host_mirror.py::test_unmodified_dabmodulator_compiles_against_the_drop_ins).  Add GpuStages.cpp to
# odr_dabmod_SOURCES, -I<this repo>/include to the compiler flags and -ldabgpu to the link.
set -e
dst=${1:?usage: install_dropins.sh <copy of the reference src/>}
here=$(cd "$(dirname "$0")" && pwd)
[ -f "$dst/DabModulator.cpp" ] || { echo "$dst does not look like ODR-DabMod's src/" >&2; exit 1; }
for c in QpskSymbolMapper FrequencyInterleaver PhaseReference DifferentialModulator NullSymbol SignalMultiplexer \
         OfdmGenerator GainControl GuardIntervalInserter FIRFilter Resampler MemlessPoly CicEqualizer TII FormatConverter
do
    rm -f "$dst/$c.h" "$dst/$c.cpp"       # (never write through a symbolic link into the original tree)
    printf '// forwarded to the MI355X drop-in of the same name\n#pragma once\n#include "GpuStages.h"\n' > "$dst/$c.h"
done
rm -f "$dst/GpuStages.h" "$dst/GpuStages.cpp"
cp "$here/GpuStages.h" "$here/GpuStages.cpp" "$dst/"
