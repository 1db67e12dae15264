#!/bin/bash
# TIMING EXPERIMENT (same samples, same instruction stream): what a FUSED frame-kernel + resampler workgroup would pay for its
# LDS footprint.  The fused workgroup needs 123 kB (resampler 76 kB + a 37 kB ring between 2552-sample symbols and 2048-sample
# hops + the frame kernel's 10 kB): ONE workgroup per CU where today's resampler runs two.  Here the
# product resampler simply asks for 100 kB of dynamic LDS, so that one workgroup fits a CU.
# Builds tools/_variants/libdabgpu_{base,rs1wg}.so; time with
#   DABGPU_LIB=tools/_variants/libdabgpu_x.so python tools/experiments/exp_r05.py parts
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
"$ROOT/tools/variants.sh" base ""
d="$ROOT/tools/_variants/src_rs1wg"
rm -rf "$d"; mkdir -p "$d/odr-dabmod_amd" "$d/include"
cp -r "$ROOT/odr-dabmod_amd/csrc" "$d/odr-dabmod_amd/csrc"; cp "$ROOT/include/"*.h "$d/include/"
rm -f "$d/odr-dabmod_amd/csrc/"*.o "$d/odr-dabmod_amd/csrc/libdabgpu.so"
python3 - "$d/odr-dabmod_amd/csrc/resampler.hip" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
old = "                const size_t lds16 = (size_t)(2 * Fft16::LDS_ELEMS + 256 + 8) * sizeof(float2) + (size_t)(NIN / 2) * sizeof(float);"
new = "                const size_t lds16 = 100 * 1024;   // EXPERIMENT: one workgroup per CU (the kernel uses the first 76 kB)"
assert s.count(old) == 1
open(p, "w").write(s.replace(old, new, 1))
PY
make -s -C "$d/odr-dabmod_amd/csrc" -j8 > "$ROOT/tools/_variants/rs1wg.log" 2>&1
cp "$d/odr-dabmod_amd/csrc/libdabgpu.so" "$ROOT/tools/_variants/libdabgpu_rs1wg.so"
cmp -s "$ROOT/tools/_variants/libdabgpu_rs1wg.so" "$ROOT/tools/_variants/libdabgpu_base.so" && { echo "the patch did not change the library" >&2; exit 1; }
echo "built tools/_variants/libdabgpu_rs1wg.so"
