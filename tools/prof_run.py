#!/usr/bin/env python3
"""Run one workload a few times (target of rocprofv3).  usage: prof_run.py [workload] [B] [iters] [mode]
workload: cfg2 | cfg3 | cfg4 | ifft_fir_stage (the bench workloads), or a stage mask of the chain from coded bits
(0 = IFFT+guard, 3 = cfg3, 15 = cfg4)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
P = importlib.import_module("odr-dabmod_amd")
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
named = {"cfg2": (0, False), "cfg3": (3, True), "cfg4": (15, True), "ifft_fir_stage": (3, False)}
mask, from_bits = named[wl] if wl in named else (int(wl), True)
MODE = int(sys.argv[4]) if len(sys.argv) > 4 else 1            # (modes II - IV: the coded-bits chains, native rate)
md = P.Modulator(mode=MODE, max_frames=B)
md.set_gain(2, 1.0, 1 / 50000., 4.0)
if mask & 4:
    md.set_resampler(2048000, 8192000)
    md.set_poly([1.0, 0.05, -0.01, 0.002, 0.0], [0.0, 0.02, 0.003, 0.0, 0.0])
if from_bits:
    d_in = torch.randint(0, 256, (B, md.geometry["tf_input_bytes"]), dtype=torch.uint8, device="cuda")
else:
    # SignalMultiplexer output: unit-modulus constellation points, blank null symbol (as bench.py builds it)
    d_in = torch.zeros((B, 77 * 1536), dtype=torch.complex64, device="cuda")
    for f0 in range(0, B, 2048):
        f1 = min(B, f0 + 2048)
        ang = (torch.randint(0, 4, (f1 - f0, 76 * 1536), device="cuda").float() * 2 + 1) * (np.pi / 4)
        d_in[f0:f1, 1536:] = torch.polar(torch.ones_like(ang), ang)
        del ang
out = torch.empty((B, md.out_samples_per_frame(mask)), dtype=torch.complex64, device="cuda")
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(iters):
        if from_bits:
            md.chain_dev(d_in, B, mask, out, stream=st.cuda_stream)
        else:
            md.symbols_dev(d_in, B, mask, out, stream=st.cuda_stream)
    st.synchronize()
print("done", wl, B, iters)
