#!/usr/bin/env python3
"""Board power and shader clock under the cfg 3 frame kernel of every transmission mode, and under cfg 3 with crest-factor
reduction / OFDM windowing (Mode I): which of them sit at the board's power limit.  usage (GPU box): python tools/power_modes.py"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from power_probe import PowerProbe, sample_load
P = importlib.import_module("odr-dabmod_amd")
probe = PowerProbe(0)
st = torch.cuda.Stream()
cases = [(1, None), (2, None), (3, None), (4, None), (1, "cfr"), (1, "window"), (1, "nofir")]
for mode, opt in cases:
    B = {1: 8192, 2: 32768, 3: 32768, 4: 16384}[mode]
    md = P.Modulator(mode=mode, max_frames=B)
    md.set_gain(2, 1.0, 1 / 50000., 4.0)
    stages = 1 if opt == "nofir" else 3
    if opt == "cfr":
        md.set_cfr(True, 50.0, 0.1)
    if opt == "window":
        md.set_window_overlap(10)
    g = md.geometry
    with torch.cuda.stream(st):
        d_in = torch.randint(0, 256, (B, g["tf_input_bytes"]), dtype=torch.uint8, device="cuda")
        out = torch.empty((B, g["tf_samples"]), dtype=torch.complex64, device="cuda")
        step = lambda: md.chain_dev(d_in, B, stages, out, stream=st.cuda_stream)
        for _ in range(3): step()
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5): step()
        e1.record(st); st.synchronize()
        ms = e0.elapsed_time(e1) / 5
        pw = sample_load(step, 3.0, ms, st, probe=probe)
    algo = (g["tf_input_bytes"] + 8 * g["tf_samples"]) * B
    print(json.dumps({"mode": mode, "option": opt, "frames_per_call": B, "ms_per_call": round(ms, 3),
                      "roofline_frac": round(algo / (ms * 1e-3) / 8e12, 4),
                      "watts_avg": pw.get("watts_avg"), "watts_cap": pw.get("watts_cap"), "sclk_MHz_avg": pw.get("sclk_MHz_avg"),
                      "mJ_per_frame": round(pw.get("watts_avg", 0) * ms * 1e-3 / B * 1e3, 4) if pw.get("watts_avg") else None}))
    md.close()
    del d_in, out
    torch.cuda.empty_cache()
