#!/usr/bin/env python3
"""Time cfg 4 (and the resampler alone on a resident native-rate stream) for several library builds.
usage (on the GPU box): python tools/time_cfg4.py [B]"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
CODE = r'''
import sys, importlib
sys.path.insert(0, %r)
import numpy as np, torch
P = importlib.import_module("odr-dabmod_amd")
B = %d
md = P.Modulator(mode=1, max_frames=B)
md.set_gain(2, 1.0, 1/50000., 4.0)
md.set_resampler(2048000, 8192000)
md.set_poly([1.0, 0.05, -0.01, 0.002, 0.0], [0.0, 0.02, 0.003, 0.0, 0.0])
bits = torch.randint(0, 256, (B, 28800), dtype=torch.uint8, device="cuda")
st = torch.cuda.Stream()
res = {}
with torch.cuda.stream(st):
    for name, mask in (("cfg3", 3), ("cfg3+res", 7), ("cfg4", 15)):
        out = torch.empty((B, md.out_samples_per_frame(mask)), dtype=torch.complex64, device="cuda")
        for _ in range(2): md.chain_dev(bits, B, mask, out, stream=st.cuda_stream)
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5): md.chain_dev(bits, B, mask, out, stream=st.cuda_stream)
        e1.record(st); st.synchronize()
        res[name] = round(B * 5 / (e0.elapsed_time(e1) * 1e-3))
        del out
print(res)
''' % (ROOT, B)
libs = sorted(glob.glob(os.path.join(ROOT, "tools", "_variants", "*.so"))) or [""]
for lib in libs:
    env = dict(os.environ)
    if lib: env["DABGPU_LIB"] = lib
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(os.path.basename(lib) or "default", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
