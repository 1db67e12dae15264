#!/usr/bin/env python3
"""Throughput of the chain's configurations beside the bench workloads (tuning aid, run on the GPU box): transmission
modes, gain modes, custom / long filters, windowing, TII, CFR, output formats, resampler ratios.
usage: survey_variants.py [B]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
P = importlib.import_module("odr-dabmod_amd")
st = torch.cuda.Stream()
B0 = int(sys.argv[1]) if len(sys.argv) > 1 else 4096


def run(name, mode=1, mask=3, setup=None, B=None, iters=6, fmt=None):
    B = B or B0
    md = P.Modulator(mode=mode, max_frames=B)
    md.set_gain(2, 1.0, 1 / 50000., 4.0)
    if setup: setup(md)
    if fmt: md.set_output_format(fmt)
    with torch.cuda.stream(st):
        nb = md.geometry["tf_input_bytes"]
        bits = torch.randint(0, 256, (B, nb), dtype=torch.uint8, device="cuda")
        ob = md.out_bytes_per_frame(mask)
        out = torch.empty((B, ob), dtype=torch.uint8, device="cuda")
        for _ in range(2): md.chain_dev(bits, B, mask, out, stream=st.cuda_stream)
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(iters): md.chain_dev(bits, B, mask, out, stream=st.cuda_stream)
        e1.record(st); st.synchronize()
    dt = e0.elapsed_time(e1) * 1e-3 / iters
    md.close()
    del out, bits
    torch.cuda.empty_cache()
    print("%-46s mode %d B=%5d  %9.0f TF/s  %7.1f GB/s out" % (name, mode, B, B / dt, B * ob / dt / 1e9), flush=True)


rng = np.random.default_rng(1)
lp = lambda n: (np.sinc((np.arange(n) - (n - 1) / 2) * 0.8) * np.hamming(n)).astype(np.float32)
for mode in (1, 2, 3, 4):
    run("cfg3 default", mode)
run("gain fix", setup=lambda m: m.set_gain(0, 1.0, 1 / 50000., 4.0))
run("gain max", setup=lambda m: m.set_gain(1, 1.0, 1 / 50000., 4.0))
run("no FIR (mask 1)", mask=1)
run("custom 45 taps", setup=lambda m: m.set_fir_taps(lp(45)))
run("custom 31 taps", setup=lambda m: m.set_fir_taps(lp(31)))
run("custom 101 taps", setup=lambda m: m.set_fir_taps(lp(101)))
run("custom 255 taps (unfused)", setup=lambda m: m.set_fir_taps(lp(255)), B=1024)
run("window 10 (equalised boundaries)", setup=lambda m: m.set_window_overlap(10))
run("window 10 -> s16", setup=lambda m: m.set_window_overlap(10), fmt="s16")
run("window 10 + TII", setup=lambda m: (m.set_window_overlap(10), m.set_tii(True, 3, 5)))
run("window 100", setup=lambda m: m.set_window_overlap(100))
run("window 100, no FIR", mask=1, setup=lambda m: m.set_window_overlap(100))
run("TII", setup=lambda m: m.set_tii(True, 3, 5))
run("CFR", setup=lambda m: m.set_cfr(True, 50.0, 0.1))
run("CFR, no FIR", mask=1, setup=lambda m: m.set_cfr(True, 50.0, 0.1))
run("CFR + TII", setup=lambda m: (m.set_cfr(True, 50.0, 0.1), m.set_tii(True, 3, 5)))
run("CFR -> s16", setup=lambda m: m.set_cfr(True, 50.0, 0.1), fmt="s16")
run("CFR + TII -> s16", setup=lambda m: (m.set_cfr(True, 50.0, 0.1), m.set_tii(True, 3, 5)), fmt="s16")
run("CFR, no FIR -> s16", mask=1, setup=lambda m: m.set_cfr(True, 50.0, 0.1), fmt="s16")
run("window 10, no FIR -> s16", mask=1, setup=lambda m: m.set_window_overlap(10), fmt="s16")
run("gain max -> s16", setup=lambda m: m.set_gain(1, 1.0, 0.9, 4.0), fmt="s16")
run("custom 101 taps -> s16", setup=lambda m: m.set_fir_taps(lp(101)), fmt="s16")
for f in ("s16", "u8", "s8"):
    run("cfg3 -> " + f, fmt=f)
poly = lambda m: m.set_poly([1.0, 0.05, -0.01, 0.002, 0.0], [0.0, 0.02, 0.003, 0.0, 0.0])
for rate in (4096000, 8192000, 3072000, 1024000, 2500000):
    run("resample %d + poly" % rate, mask=7, setup=lambda m, r=rate: (m.set_resampler(2048000, r), poly(m)), B=1024)
run("resample 8192000, no poly", mask=7, setup=lambda m: m.set_resampler(2048000, 8192000), B=1024)
run("poly only (native rate)", mask=7, setup=poly, B=1024)
run("LUT only (native rate)", mask=7, setup=lambda m: m.set_lut(1.0 / 32768, np.linspace(1.0, 1.2, 32).astype(np.float32)), B=1024)
run("no FIR -> s16", mask=1, fmt="s16")
run("no FIR -> u8", mask=1, fmt="u8")
run("no FIR -> s8", mask=1, fmt="s8")
run("CFR + window 10", setup=lambda m: (m.set_cfr(True, 50.0, 0.1), m.set_window_overlap(10)))
run("CFR + window 10, no FIR", mask=1, setup=lambda m: (m.set_cfr(True, 50.0, 0.1), m.set_window_overlap(10)))
run("window 10", setup=lambda m: m.set_window_overlap(10))
run("window 10, no FIR", mask=1, setup=lambda m: m.set_window_overlap(10))
run("TII + window 10", setup=lambda m: (m.set_tii(True, 3, 5), m.set_window_overlap(10)))
run("TII -> s16", setup=lambda m: m.set_tii(True, 3, 5), fmt="s16")
run("gain max -> s16", setup=lambda m: m.set_gain(1, 1.0, 0.9, 4.0), fmt="s16")
run("gain max, no FIR", mask=1, setup=lambda m: m.set_gain(1, 1.0, 1 / 50000., 4.0))
run("TII, no FIR", mask=1, setup=lambda m: m.set_tii(True, 3, 5))
run("TII, no FIR -> s16", mask=1, setup=lambda m: m.set_tii(True, 3, 5), fmt="s16")
run("custom 5 taps (simplefiltertaps.txt)", setup=lambda m: m.set_fir_taps(np.array([0, 0, 1, 0, 0], np.float32)))
