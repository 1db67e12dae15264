#!/bin/bash
# A/B: cache-policy bits on the frame kernel's output stores (raw_buffer_store aux: 1 = sc0, 2 = nt, 16 = sc1).  Under the board's
# power limit the cheapest store in ENERGY wins, which need not be the fastest one in a bandwidth test.
# Builds tools/_variants/libdabgpu_auxNN.so through the tool-only flag -DDABGPU_STORE_AUX=n (tf_kernel.h: kStoreAux) and checks
# that the flag reached the kernel (a variant that is byte-identical to the product build would be a silent no-op).
# Time with tools/time_cfg3_variants.py 32768 (cfg 3) or DABGPU_LIB=... python bench.py --no-cpu-baseline --counters off.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
grep -q "DABGPU_STORE_AUX" "$ROOT/odr-dabmod_amd/csrc/tf_kernel.h" || { echo "tf_kernel.h no longer reads DABGPU_STORE_AUX" >&2; exit 1; }
for aux in ${@:-0 2 1 16 17 3 18}; do
  name="aux$(printf %02d $aux)"
  "$ROOT/tools/variants.sh" "$name" "-DDABGPU_STORE_AUX=$aux"
  # the product build stores non-temporally (aux 2) on the coded-bits chain: every other value must change the code object
  if [ "$aux" != 2 ] && cmp -s "$ROOT/tools/_variants/libdabgpu_$name.so" "$ROOT/odr-dabmod_amd/csrc/libdabgpu.so"; then
    echo "variant $name is identical to the product library: the flag did not take" >&2; exit 1
  fi
done
