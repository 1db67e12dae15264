#!/usr/bin/env python3
"""Run one workload a few times (target of rocprofv3).  usage: prof_run.py [mask] [B] [iters]
mask: stage mask of the chain from coded bits (0 = IFFT+guard, 3 = cfg3, 15 = cfg4)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
P = importlib.import_module("odr-dabmod_amd")
mask = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
md = P.Modulator(mode=1, max_frames=B)
md.set_gain(2, 1.0, 1 / 50000., 4.0)
if mask & 4:
    md.set_resampler(2048000, 8192000)
    md.set_poly([1.0, 0.05, -0.01, 0.002, 0.0], [0.0, 0.02, 0.003, 0.0, 0.0])
bits = torch.randint(0, 256, (B, 28800), dtype=torch.uint8, device="cuda")
out = torch.empty((B, md.out_samples_per_frame(mask)), dtype=torch.complex64, device="cuda")
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(iters):
        md.chain_dev(bits, B, mask, out, stream=st.cuda_stream)
    st.synchronize()
print("done", mask, B, iters)
