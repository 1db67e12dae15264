#!/usr/bin/env python3
"""Throughput vs batch size / chunks per frame (tuning aid, run on the GPU box).
usage: sweep_b.py [mask] [B,chunks ...]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
P = importlib.import_module("odr-dabmod_amd")
st = torch.cuda.Stream()
def run(B, chunks, mask, iters=8):
    md = P.Modulator(mode=1, max_frames=B, chunks_per_frame=chunks)
    md.set_gain(2, 1.0, 1 / 50000., 4.0)
    if os.environ.get("CFR"):
        md.set_cfr(True, 50.0, 0.1)
    if os.environ.get("WIN"):
        md.set_window_overlap(int(os.environ["WIN"]))
    if os.environ.get("TII"):
        md.set_tii(True, 3, 5)
    if mask & 4:
        md.set_resampler(2048000, int(os.environ.get("RATE", "8192000")))
        md.set_poly([1.0, 0.05, -0.01, 0.002, 0.0], [0.0, 0.02, 0.003, 0.0, 0.0])
    with torch.cuda.stream(st):
        bits = torch.randint(0, 256, (B, 28800), dtype=torch.uint8, device="cuda")
        out = torch.empty((B, md.out_samples_per_frame(mask)), dtype=torch.complex64, device="cuda")
        for _ in range(2): md.chain_dev(bits, B, mask, out, stream=st.cuda_stream)
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(iters): md.chain_dev(bits, B, mask, out, stream=st.cuda_stream)
        e1.record(st); st.synchronize()
    md.close()
    del out, bits
    torch.cuda.empty_cache()
    return B * iters / (e0.elapsed_time(e1) * 1e-3)
mask = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cases = [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:]] or \
    [(256, 1), (512, 1), (768, 1), (1024, 1), (1536, 1), (2048, 1), (3072, 1), (3840, 1), (4096, 1), (4608, 1),
     (6144, 1), (8192, 1), (64, 11), (1, 11)]
print("mask", mask)
for B, ch in cases:
    print("  B=%5d chunks=%2d  %10.0f frames/s" % (B, ch, run(B, ch, mask)), flush=True)
