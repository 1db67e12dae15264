#!/usr/bin/env python3
"""cfg 2 (IFFT + guard from carriers) and the IFFT + FIR stage for every library in tools/_variants (tuning aid, GPU box)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, importlib
sys.path.insert(0, %r)
import numpy as np, torch
P = importlib.import_module("odr-dabmod_amd")
B = 16384
md = P.Modulator(mode=1, max_frames=B)
md.set_gain(2, 1.0, 1/50000., 4.0)
d_in = torch.zeros((B, 77 * 1536), dtype=torch.complex64, device="cuda")
for f0 in range(0, B, 2048):
    ang = (torch.randint(0, 4, (2048, 76 * 1536), device="cuda").float() * 2 + 1) * (np.pi / 4)
    d_in[f0:f0 + 2048, 1536:] = torch.polar(torch.ones_like(ang), ang)
    del ang
out = torch.empty((B, 196608), dtype=torch.complex64, device="cuda")
st = torch.cuda.Stream()
res = {}
with torch.cuda.stream(st):
    for name, mask in (("cfg2", 0), ("ifft_fir", 3)):
        for _ in range(3): md.symbols_dev(d_in, B, mask, out, stream=st.cuda_stream)
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(8): md.symbols_dev(d_in, B, mask, out, stream=st.cuda_stream)
        e1.record(st); st.synchronize()
        res[name] = round(B * 8 / (e0.elapsed_time(e1) * 1e-3))
print(res)
''' % ROOT
for lib in sorted(glob.glob(os.path.join(ROOT, "tools", "_variants", "*.so"))) or [""]:
    env = dict(os.environ)
    if lib: env["DABGPU_LIB"] = lib
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(os.path.basename(lib) or "default", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
