#!/usr/bin/env python3
"""cfg 3 at small batches (B = 1, 4, 16, 64, 256, 1024) for several library builds (tuning aid, GPU box)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, importlib
sys.path.insert(0, %r)
import torch
P = importlib.import_module("odr-dabmod_amd")
res = {}
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for B in (1, 4, 16, 64, 256, 1024):
        md = P.Modulator(mode=1, max_frames=B)
        md.set_gain(2, 1.0, 1/50000., 4.0)
        bits = torch.randint(0, 256, (B, 28800), dtype=torch.uint8, device="cuda")
        out = torch.empty((B, 196608), dtype=torch.complex64, device="cuda")
        for _ in range(5): md.chain_dev(bits, B, 3, out, stream=st.cuda_stream)
        st.synchronize()
        n = 200 if B <= 64 else 40
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(n): md.chain_dev(bits, B, 3, out, stream=st.cuda_stream)
        e1.record(st); st.synchronize()
        res[B] = "%%.1f us/launch, %%d frames/s" %% (e0.elapsed_time(e1) * 1e3 / n, round(B * n / (e0.elapsed_time(e1) * 1e-3)))
        md.close()
print(res)
''' % ROOT
libs = sorted(glob.glob(os.path.join(ROOT, "tools", "_variants", "*.so"))) or [""]
for lib in libs:
    env = dict(os.environ)
    if lib: env["DABGPU_LIB"] = lib
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(os.path.basename(lib) or "default", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-600:], flush=True)
