#!/usr/bin/env python3
"""Time series of submit / collect (profiles/r06_async_modes.txt: the first repetition of a measurement is slow at 32 frames
per batch, the SECOND one at 1 frame per batch): per-block means of the time spent inside submit and inside collect.
usage (GPU box): python tools/experiments/exp_r06_async_series.py FMT B N BLOCK [idle_ms]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
P = importlib.import_module("odr-dabmod_amd")
fmt, B, N, BLOCK = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
idle_ms = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
fmt = None if fmt == "complexf" else fmt
md = P.Modulator(mode=1, max_frames=B)
md.set_gain(2, 1.0, 1 / 50000. if fmt is None else 1.0, 4.0)
md.set_output_format(fmt)
bits = np.frombuffer(np.random.RandomState(1).bytes(B * 28800), np.uint8).reshape(B, 28800)
if idle_ms:
    time.sleep(idle_ms / 1e3)
ts = np.zeros((N, 2))
md.submit(bits, 3)
pc = time.perf_counter
for i in range(N):
    t0 = pc(); md.submit(bits, 3); t1 = pc(); md.collect(copy=False); t2 = pc()
    ts[i] = (t1 - t0, t2 - t1)
md.collect(copy=False)
md.close()
print("%s B=%d: per block of %d batches: us in submit, us in collect, frames/s" % (fmt or "complexf", B, BLOCK))
for b in range(0, N, BLOCK):
    s, c = ts[b:b + BLOCK, 0].mean() * 1e6, ts[b:b + BLOCK, 1].mean() * 1e6
    print("  %5d  %8.1f %8.1f  %8.0f   (max submit %.0f us, max collect %.0f us)"
          % (b, s, c, B / ((s + c) * 1e-6), ts[b:b + BLOCK, 0].max() * 1e6, ts[b:b + BLOCK, 1].max() * 1e6))
