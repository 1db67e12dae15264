#!/usr/bin/env python3
"""cfg 3 with the two roundings of gain mode var (dabgpu_set_gain_rounding): frames/s of each, same box, same batch.
usage: python tools/time_gain_rounding.py [frames per call]"""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
P = importlib.import_module("odr-dabmod_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
md = P.Modulator(mode=1, max_frames=B)
md.set_gain(2, 1.0, 1.0 / 50000.0, 4.0)
stages = P.STAGE_GAIN | P.STAGE_FIR
bits = torch.from_numpy(np.random.default_rng(42).integers(0, 256, (B, 28800), dtype=np.uint8)).cuda()
out = torch.empty(B * 196608, dtype=torch.complex64, device="cuda")
s = torch.cuda.Stream()
md.trace(True)
for ref in (False, True):
    md.set_gain_rounding(ref)
    with torch.cuda.stream(s):
        for _ in range(2):
            md.chain_dev(bits, B, stages, out, stream=s.cuda_stream)
        s.synchronize()
        kernels = md.last_variant()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            md.chain_dev(bits, B, stages, out, stream=s.cuda_stream)
        s.synchronize()
        dt = (time.perf_counter() - t0) / n
    print(json.dumps({"gain_rounding": "reference" if ref else "exact", "frames_per_call": B, "ms_per_call": round(dt * 1e3, 3),
                      "frames_per_s": round(B / dt, 1), "roofline_frac": round(B * 1601664 / dt / 8e12, 4),
                      "kernels": kernels.split(";")[:6] if ";" in kernels else kernels}))
md.close()
