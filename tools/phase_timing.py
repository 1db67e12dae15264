#!/usr/bin/env python3
"""Where a symbol iteration of the frame kernel spends its cycles, measured: s_memtime stamps between the phases of the
symbol loop (device_common.h: PhaseTimer), summed per wave, from a TOOL build of the library (-DDABGPU_PHASE_TIMING; the
product build compiles the stamps away).  Same samples as the product, ~10 % slower (every stamp drains the wave's LDS
queue).  Build here, run on the GPU box:
    tools/variants.sh phase "-DDABGPU_PHASE_TIMING"          # -> tools/_variants/libdabgpu_phase.so
    gpurun -- python tools/phase_timing.py [frames] > profiles/rNN_cfg3_phase_cycles.txt
Prints, for cfg 3 (and the product's time for the same launch next to it): shader cycles per wave and symbol by phase,
their share, and -- from the launch's duration and clock -- what that is in ms per launch."""
import ctypes as C, importlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHASES = ["input: barrier, bit gather, unit-vector table, H multiply, placement", "butterflies (3 x radix 8 + radix 4)",
          "exchanges (3 x scatter, barrier, gather, barrier)", "gain, scaling, boundary windows to LDS + barrier",
          "stores (body + cyclic prefix)", "boundary: inverse filter, correction, 44 stores", "loop: branch, counters"]

CHILD = r'''
import ctypes as C, importlib, json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
P = importlib.import_module("odr-dabmod_amd")
B = %(B)d
MODE = %(mode)d
md = P.Modulator(mode=MODE, max_frames=B)
md.set_gain(2, 1.0, 1 / 50000., 4.0)
torch.manual_seed(1234)            # (the same input in the product run and the timing run: their samples are compared)
d_in = torch.randint(0, 256, (B, md.geometry["tf_input_bytes"]), dtype=torch.uint8, device="cuda")
out = torch.empty((B, md.geometry["tf_samples"]), dtype=torch.complex64, device="cuda")
st = torch.cuda.Stream()
lib = P.load_library()
timed = hasattr(lib, "dabgpu_debug_phase_cycles")
buf = (C.c_ulonglong * 16)()
with torch.cuda.stream(st):
    for _ in range(3): md.chain_dev(d_in, B, 3, out, stream=st.cuda_stream)
    st.synchronize()
    if timed:
        lib.dabgpu_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
        lib.dabgpu_debug_phase_cycles(md._h, buf)            # (zeroes the counters)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(%(iters)d): md.chain_dev(d_in, B, 3, out, stream=st.cuda_stream)
    e1.record(st); st.synchronize()
    ms = e0.elapsed_time(e1) / %(iters)d
    if timed: lib.dabgpu_debug_phase_cycles(md._h, buf)
import hashlib
print(json.dumps({"ms_per_launch": ms, "counters": list(buf), "timed": timed,
                  "sha": hashlib.sha256(out[:4].cpu().numpy().tobytes()).hexdigest()[:16]}))
'''


MODE = 1


def run(lib, B, iters):
    env = dict(os.environ)
    if lib:
        env["DABGPU_LIB"] = lib
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "B": B, "iters": iters, "mode": MODE}], env=env, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-3000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def main():
    global MODE
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    MODE = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # (modes II - IV: the generic packed-dual-transform kernel)
    iters = 5
    wpg = {1: 4, 2: 1, 3: 1, 4: 2}[MODE]                           # waves per workgroup
    nsym = 154 if MODE == 3 else 77
    variant = os.path.join(ROOT, "tools", "_variants", "libdabgpu_phase.so")
    if not os.path.exists(variant):
        sys.exit("build the tool library first: tools/variants.sh phase \"-DDABGPU_PHASE_TIMING\"")
    prod = run(None, B, iters)
    tim = run(variant, B, iters)
    assert tim["timed"] and not prod["timed"]
    c = tim["counters"]
    n_iter = c[15]
    tot = sum(c[:len(PHASES)])
    print("cfg 3 (coded bits -> gain var -> guard -> 45-tap FIR%s), mode %d, %d frames per launch"
          % (", equalised boundaries" if MODE == 1 else ", packed dual transform", MODE, B))
    print("product library: %.3f ms per launch;  timing build: %.3f ms (+%.1f %%);  same samples: %s"
          % (prod["ms_per_launch"], tim["ms_per_launch"], 100 * (tim["ms_per_launch"] / prod["ms_per_launch"] - 1),
             "yes" if prod["sha"] == tim["sha"] else "NO"))
    print("wave-iterations timed: %d (launches x frames x %d symbols x %d waves = %d, plus one look-ahead symbol per run)"
          % (n_iter, nsym, wpg, iters * B * nsym * wpg))
    print("%-78s %14s %8s %12s" % ("phase of one symbol iteration", "cycles/wave", "share", "ms of launch"))
    for name, v in zip(PHASES, c):
        print("%-78s %14.1f %7.1f%% %12.3f" % (name, v / n_iter, 100.0 * v / tot, prod["ms_per_launch"] * v / tot))
    print("%-78s %14.1f %7.1f%% %12.3f" % ("sum (a wave shares its SIMD with three others: wall = sum / 4 per symbol)",
                                          tot / n_iter, 100.0, prod["ms_per_launch"]))


if __name__ == "__main__":
    main()
