#!/bin/bash
# A/B: cache-policy bits on the frame kernel's output stores (raw_buffer_store aux: 1 = sc0, 2 = nt, 16 = sc1).  Under the board's
# power limit the cheapest store in ENERGY wins, which need not be the fastest one in a bandwidth test.
# Time with tools/time_cfg3_variants.py 32768.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$ROOT/tools/_variants"
for aux in 0 2 1 16 17 3 18; do
  d="$ROOT/tools/_variants/src_aux$aux"
  rm -rf "$d"; mkdir -p "$d/odr-dabmod_amd" "$d/include"
  cp -r "$ROOT/odr-dabmod_amd/csrc" "$d/odr-dabmod_amd/csrc"; cp "$ROOT/include/"*.h "$d/include/"
  rm -f "$d/odr-dabmod_amd/csrc/"*.o "$d/odr-dabmod_amd/csrc/libdabgpu.so"
  sed -i "s/__builtin_amdgcn_raw_buffer_store_b64(d, orsrc, voff \* 8, soff \* 8, 0);/__builtin_amdgcn_raw_buffer_store_b64(d, orsrc, voff * 8, soff * 8, $aux);/" "$d/odr-dabmod_amd/csrc/tf_kernel.h"
  grep -c "soff \* 8, $aux);" "$d/odr-dabmod_amd/csrc/tf_kernel.h" > /dev/null
  make -s -C "$d/odr-dabmod_amd/csrc" -j8 > "$ROOT/tools/_variants/aux$aux.log" 2>&1
  cp "$d/odr-dabmod_amd/csrc/libdabgpu.so" "$ROOT/tools/_variants/libdabgpu_aux$(printf %02d $aux).so"
  echo "built aux$aux"
done
