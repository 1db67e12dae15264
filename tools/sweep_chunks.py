#!/usr/bin/env python3
"""cfg 3: workgroups per frame (chunks) against batch size (tuning aid for auto_chunks, GPU box)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
P = importlib.import_module("odr-dabmod_amd")
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for B in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024):
        res = {}
        for ch in (0, 1, 2, 3, 4, 6, 8, 11, 13, 16, 20, 26, 39, 77):
            md = P.Modulator(mode=1, max_frames=B, chunks_per_frame=ch)
            md.set_gain(2, 1.0, 1/50000., 4.0)
            bits = torch.randint(0, 256, (B, 28800), dtype=torch.uint8, device="cuda")
            out = torch.empty((B, 196608), dtype=torch.complex64, device="cuda")
            for _ in range(3): md.chain_dev(bits, B, 3, out, stream=st.cuda_stream)
            st.synchronize()
            n = 100 if B <= 64 else 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(n): md.chain_dev(bits, B, 3, out, stream=st.cuda_stream)
            e1.record(st); st.synchronize()
            res[ch] = round(e0.elapsed_time(e1) * 1e3 / n, 1)
            md.close()
        best = min((v, k) for k, v in res.items() if k)
        print("B=%4d  auto %.1f us | best %.1f us at %d chunks | %s" % (B, res[0], best[0], best[1], res), flush=True)
