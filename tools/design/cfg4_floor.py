#!/usr/bin/env python3
"""cfg 4 (Resampler x4 + MemlessPoly, resampler16_kernel<true, false, 4>): the least arithmetic one hop can take, counted on
paper, against what the compiler emits for the hot loop ("move it or close it").

One hop = 2048 samples in, 8192 predistorted samples out = one forward and three branch 4096-point transforms
(16 . 16 . 16 on 256 lanes, 16 points per lane), the window / overlap-add in front (/root/reference/src/Resampler.cpp:142-192)
and the polynomial on 32 output samples per lane (src/MemlessPoly.cpp:237-276).  Counted per LANE and hop in fp32
operations that each need an issue slot of their own (an FMA is one; a packed v_pk_* instruction does two and occupies
the SIMD-32 for two passes, so it counts as two):

  16-point DFT                      144  (split radix: 144 additions + 24 multiplications, every multiplication fused into an addition)
  ... with 15 twiddled inputs      +30   (a complex product is 4 FMAs of which 2 replace additions of the butterfly)
  ... half of the outputs wanted   -32   (last stage of a branch transform: 4 of the 8 second-layer additions per DFT4)
  forward transform                 144 + 174 + 174 = 492
  branch transform                  144 + 174 + 142 = 460   (x3)
  g_h = [(w1 + w2) c_{h-1} | w2 c_h + w1 c_{h-2}]   8 x 2 + 8 x 4 = 48
  branch twiddles W^{kappa p}       16 bins x 3 branches x 2 = 96   (4 FMAs each, 2 of them absorbed by the first butterflies)
  branch p = 0 (scaled input)       8 x 2 = 16
  polynomial, 32 samples            32 x 23 = 736   (|x|^2 2, two quartic Horner chains 8, p^2 1, cos 3, sin 3, x a 2, rotation 4)
usage: tools/design/cfg4_floor.py  (reads profiles/isa_mix.json and recompiles the kernel for the opcode histogram)"""
import collections, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_mix

FLOOR = collections.OrderedDict([
    ("forward transform (16 . 16 . 16)", 144 + 174 + 174),
    ("three branch transforms, last stage half-pruned", 3 * (144 + 174 + 142)),
    ("window / overlap-add in front of the forward transform", 48),
    ("branch twiddles", 96),
    ("branch 0 (no transform)", 16),
    ("MemlessPoly polynomial on 32 samples", 32 * 23),
])


def main():
    floor = sum(FLOOR.values())
    src, extra, pat, unit = isa_mix.KERNELS["cfg4"]
    out = tempfile.mktemp(suffix=".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + isa_mix.FLAGS + extra + ["-o", out, os.path.join(isa_mix.CSRC, src)],
                          stderr=subprocess.DEVNULL)
    bodies = isa_mix.kernel_bodies(out)
    names = subprocess.run(["c++filt"], input="\n".join(bodies), capture_output=True, text=True).stdout.splitlines()
    body = [bodies[k] for k, n in zip(bodies, names) if pat in n][0]
    os.unlink(out)
    loop = isa_mix.hot_loop(body)
    ops = collections.Counter()
    for l in loop:
        t = re.sub(r";.*$", "", l).strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        ops[t.split()[0]] += 1
    arith = lambda o: re.match(r"v_(pk_)?(fma|fmamk|fmaak|mul|add|sub|mac|fmac|subrev)_f32", o)
    valu = {o: n for o, n in ops.items() if o.startswith("v_")}
    n_valu = sum(valu.values())
    n_pk = sum(n for o, n in valu.items() if o.startswith("v_pk_"))
    n_arith = sum(n * (2 if o.startswith("v_pk_") else 1) for o, n in valu.items() if arith(o))
    n_other = sum(n * (2 if o.startswith("v_pk_") else 1) for o, n in valu.items() if not arith(o))
    print("cfg 4, one hop of resampler16_kernel<true, false, 4> per lane (hot loop of the committed sources)")
    print()
    print("paper floor, fp32 operations with an issue slot of their own:")
    for k, v in FLOOR.items():
        print("  %-58s %5d" % (k, v))
    print("  %-58s %5d" % ("total", floor))
    print()
    print("emitted: %d VALU instructions, %d of them packed = %d slot-equivalents" % (n_valu, n_pk, n_valu + n_pk))
    print("  fp32 arithmetic (fma / mul / add / sub, packed counted twice)   %5d   = %.3f x the floor" % (n_arith, n_arith / floor))
    print("  everything else (moves, selects, integer address work, conversions) %5d" % n_other)
    print("  total                                                              %5d   = %.3f x the floor" % (n_arith + n_other, (n_arith + n_other) / floor))
    print()
    print("largest non-arithmetic opcodes: " + ", ".join("%s x%d" % (o, n) for o, n in
          sorted(((o, n) for o, n in valu.items() if not arith(o)), key=lambda x: -x[1])[:10]))
    lds = {o: n for o, n in ops.items() if o.startswith("ds_")}
    print("LDS instructions: %d (%s)" % (sum(lds.values()), ", ".join("%s x%d" % kv for kv in sorted(lds.items(), key=lambda x: -x[1]))))
    print("  floor: 4 transforms x 2 exchanges x (16 writes + 16 reads of 8 bytes) = 256, + the twiddle table reads of stage 2")
    try:
        mix = json.load(open(os.path.join(ROOT, "profiles", "isa_mix.json")))["cfg4"]
        print("profiles/isa_mix.json: %d VALU, packed fraction %.4f" % (mix["valu_instructions"], mix["packed_fraction_of_valu"]))
    except Exception as ex:
        print("profiles/isa_mix.json: %s" % ex)


if __name__ == "__main__":
    main()
