#!/usr/bin/env python3
"""cfg 3 (coded bits -> gain var -> guard -> 45-tap FIR) in every transmission mode: frames/s, share of the HBM roofline,
and which kernel ran.  usage (GPU box): [DABGPU_LIB=...] python tools/time_modes.py [modes, e.g. 234] [frames of Mode I]"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
P = importlib.import_module("odr-dabmod_amd")
modes = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "1234")]
base = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
st = torch.cuda.Stream()
for mode in modes:
    B = base * {1: 1, 2: 4, 3: 4, 4: 2}[mode]
    md = P.Modulator(mode=mode, max_frames=B)
    md.set_gain(2, 1.0, 1 / 50000., 4.0)
    if os.environ.get("DABGPU_DIRECT_BOUNDARY") == "1":
        md.set_fir_boundary_mode(True)          # (the packed dual transform in place of the equalised-boundary variant)
    md.trace(True)
    g = md.geometry
    with torch.cuda.stream(st):
        bits = torch.randint(0, 256, (B, g["tf_input_bytes"]), dtype=torch.uint8, device="cuda")
        out = torch.empty((B, g["tf_samples"]), dtype=torch.complex64, device="cuda")
        for _ in range(3):
            md.chain_dev(bits, B, 3, out, stream=st.cuda_stream)
        st.synchronize()
        name = md.last_variant()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(5):
                md.chain_dev(bits, B, 3, out, stream=st.cuda_stream)
            e1.record(st); st.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
    fps = B / (best * 1e-3)
    algo = g["tf_input_bytes"] + g["tf_samples"] * 8
    print(json.dumps({"mode": mode, "frames_per_call": B, "ms_per_call": round(best, 4), "frames_per_s": round(fps, 1),
                      "roofline_frac": round(fps * algo / 8e12, 4), "kernel": name,
                      "lib": os.path.basename(os.environ.get("DABGPU_LIB", "product"))}), flush=True)
    md.close()
    del bits, out
    torch.cuda.empty_cache()
