#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in entry point dabgpu_chain_process (host buffers in, host buffers out).
usage (GPU box): python tools/time_host_path.py"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
P = importlib.import_module("odr-dabmod_amd")
FMT = sys.argv[1] if len(sys.argv) > 1 else None       # e.g. s16: FormatConverter fused into the chain's last store
BPS = 8 if FMT is None else 4
print("output format:", FMT or "complexf")
print("synchronous (dabgpu_chain_process; the caller's output buffer is allocated once and reused, like a flowgraph edge's Buffer):")
for B in (1, 8, 64, 256):
    md = P.Modulator(mode=1, max_frames=B)
    md.set_gain(2, 1.0, 1 / 50000. if FMT is None else 1.0, 4.0)
    md.set_output_format(FMT)
    bits = np.frombuffer(np.random.RandomState(1).bytes(B * 28800), np.uint8).reshape(B, 28800)
    out = md.chain(bits, 3)
    for _ in range(3): md.chain(bits, 3, out=out)
    n = max(3, 256 // B)
    t0 = time.perf_counter()
    for _ in range(n): md.chain(bits, 3, out=out)
    dt = time.perf_counter() - t0
    print("B=%3d  %8.0f frames/s  (%.2f ms per call, %.2f GB/s of IQ to the host)"
          % (B, B * n / dt, dt / n * 1e3, B * n * 196608 * BPS / dt / 1e9), flush=True)
    if B == 64:
        # what round 3 measured as a cliff: a fresh (untouched) 100 MB output array per call
        t0 = time.perf_counter()
        for _ in range(n): md.chain(bits, 3)
        dt = time.perf_counter() - t0
        print("B=%3d  %8.0f frames/s  with a NEW output array per call (first-touch page faults: the harness, not the library)"
              % (B, B * n / dt), flush=True)
    md.close()
print("asynchronous (submit / collect, two batches in flight, pinned output handed out without a copy):")
for B in (1, 8, 32):
    md = P.Modulator(mode=1, max_frames=B)
    md.set_gain(2, 1.0, 1 / 50000. if FMT is None else 1.0, 4.0)
    md.set_output_format(FMT)
    bits = np.frombuffer(np.random.RandomState(1).bytes(B * 28800), np.uint8).reshape(B, 28800)
    md.submit(bits, 3)
    for _ in range(4):
        md.submit(bits, 3); md.collect(copy=False)
    n = max(8, 512 // B)
    t0 = time.perf_counter()
    for _ in range(n):
        md.submit(bits, 3)
        md.collect(copy=False)
    dt = time.perf_counter() - t0
    md.collect(copy=False)
    print("B=%3d  %8.0f frames/s  (%.2f ms per batch, %.2f GB/s of IQ to the host)"
          % (B, B * n / dt, dt / n * 1e3, B * n * 196608 * BPS / dt / 1e9), flush=True)
    md.close()
