#!/bin/bash
# cfg 4 A/B builds (scratch copies, git-ignored): wave priority around the resampler's exchanges, run lengths.
# Time with: python tools/time_cfg4.py 4096
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
mk() {   # name, python patch body operating on resampler.hip text `s`
  name=$1; d="$ROOT/tools/_variants/src_$name"
  rm -rf "$d"; mkdir -p "$d/odr-dabmod_amd" "$d/include"
  cp -r "$ROOT/odr-dabmod_amd/csrc" "$d/odr-dabmod_amd/csrc"; cp "$ROOT/include/"*.h "$d/include/"
  rm -f "$d/odr-dabmod_amd/csrc/"*.o "$d/odr-dabmod_amd/csrc/libdabgpu.so"
  python3 - "$d/odr-dabmod_amd/csrc/resampler.hip" <<PY
import sys
p = sys.argv[1]
s = open(p).read()
$2
open(p, "w").write(s)
PY
  make -s -C "$d/odr-dabmod_amd/csrc" -j8 > "$ROOT/tools/_variants/$name.log" 2>&1
  cp "$d/odr-dabmod_amd/csrc/libdabgpu.so" "$ROOT/tools/_variants/libdabgpu_$name.so"
  echo "built $name"
}
mkdir -p "$ROOT/tools/_variants"
mk base "pass"
# exchanges at raised priority: the waves that are between a scatter and a gather hold up three others at the barrier
mk prio_x "
a = '''#pragma unroll
            for (int r = 0; r < 16; ++r) wp[r * P1] = v[r];
            xbarrier();'''
assert a in s
s = s.replace(a, '''            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int r = 0; r < 16; ++r) wp[r * P1] = v[r];
            xbarrier();''')
a = '''            for (int m = 0; m < 16; ++m) v[m] = rp[16 * m];
        }'''
assert a in s
s = s.replace(a, '''            for (int m = 0; m < 16; ++m) v[m] = rp[16 * m];
            __builtin_amdgcn_s_setprio(0);
        }''')
a = '''#pragma unroll
            for (int r = 0; r < 16; ++r) wp[16 * r] = v[r];
            xbarrier();'''
assert a in s
s = s.replace(a, '''            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int r = 0; r < 16; ++r) wp[16 * r] = v[r];
            xbarrier();''')
a = '''            for (int m = 0; m < 16; ++m) v[m] = rp[256 * m];
        }'''
assert a in s
s = s.replace(a, '''            for (int m = 0; m < 16; ++m) v[m] = rp[256 * m];
            __builtin_amdgcn_s_setprio(0);
        }''')
"
# the other way round: butterflies at raised priority (a wave in its VALU phase is never held up by the other's LDS traffic)
mk prio_v "
a = '''#pragma unroll
            for (int r = 0; r < 16; ++r) wp[r * P1] = v[r];
            xbarrier();'''
assert a in s
s = s.replace(a, '''            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) wp[r * P1] = v[r];
            xbarrier();''')
a = '''            for (int m = 0; m < 16; ++m) v[m] = rp[16 * m];
        }'''
assert a in s
s = s.replace(a, '''            for (int m = 0; m < 16; ++m) v[m] = rp[16 * m];
            __builtin_amdgcn_s_setprio(2);
        }''')
a = '''#pragma unroll
            for (int r = 0; r < 16; ++r) wp[16 * r] = v[r];
            xbarrier();'''
assert a in s
s = s.replace(a, '''            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) wp[16 * r] = v[r];
            xbarrier();''')
a = '''            for (int m = 0; m < 16; ++m) v[m] = rp[256 * m];
        }'''
assert a in s
s = s.replace(a, '''            for (int m = 0; m < 16; ++m) v[m] = rp[256 * m];
            __builtin_amdgcn_s_setprio(2);
        }''')
"
mk run96 "
a = 'std::min<size_t>(24, a.nhops / 1536)'
assert a in s
s = s.replace(a, 'std::min<size_t>(96, a.nhops / 1536)')
"
mk run8 "
a = 'std::min<size_t>(24, a.nhops / 1536)'
assert a in s
s = s.replace(a, 'std::min<size_t>(8, a.nhops / 1536)')
"
