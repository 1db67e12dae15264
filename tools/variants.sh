#!/bin/bash
# Build A/B variants of libdabgpu.so into tools/_variants/ (git-ignored): a copy of the csrc tree per variant, compiled with
# extra flags (or after a patch applied by hand to the copy).  The timing tools pick a library through DABGPU_LIB.
# usage: tools/variants.sh name "<extra hipcc flags>" [name flags]...
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$ROOT/tools/_variants"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  d="$ROOT/tools/_variants/src_$name"
  rm -rf "$d"; mkdir -p "$d/odr-dabmod_amd" "$d/include"
  cp -r "$ROOT/odr-dabmod_amd/csrc" "$d/odr-dabmod_amd/csrc"; cp "$ROOT/include/"*.h "$d/include/"
  rm -f "$d/odr-dabmod_amd/csrc/"*.o "$d/odr-dabmod_amd/csrc/libdabgpu.so"
  make -s -C "$d/odr-dabmod_amd/csrc" -j8 NOSLP="-fno-slp-vectorize $flags" > "$ROOT/tools/_variants/$name.log" 2>&1
  cp "$d/odr-dabmod_amd/csrc/libdabgpu.so" "$ROOT/tools/_variants/libdabgpu_$name.so"
  echo "built tools/_variants/libdabgpu_$name.so"
done
