#!/usr/bin/env python3
"""Throughput of the carriers ("symbols") entry for several stage masks (tuning aid, GPU box).
usage: python tools/time_symbols.py [B]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
P = importlib.import_module("odr-dabmod_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
md = P.Modulator(mode=1, max_frames=B)
md.set_gain(2, 1.0, 1 / 50000., 4.0)
st = torch.cuda.Stream()
res = {}
with torch.cuda.stream(st):
    d_in = torch.zeros((B, 77 * 1536), dtype=torch.complex64, device="cuda")
    q = torch.randint(0, 4, (B, 76 * 1536), device="cuda")
    ang = (q.float() * 2 + 1) * (np.pi / 4)
    d_in[:, 1536:] = torch.polar(torch.ones_like(ang), ang)
    del q, ang
    out = torch.empty((B, 196608), dtype=torch.complex64, device="cuda")
    for name, mask in (("ifft+guard", 0), ("ifft+gain+guard", 1), ("ifft+guard+fir", 2), ("ifft+gain+guard+fir", 3)):
        for _ in range(2): md.symbols_dev(d_in, B, mask, out, stream=st.cuda_stream)
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(8): md.symbols_dev(d_in, B, mask, out, stream=st.cuda_stream)
        e1.record(st); st.synchronize()
        fps = B * 8 / (e0.elapsed_time(e1) * 1e-3)
        res[name] = (round(fps), round(fps * 2519040 / 1e9))
print(res)
