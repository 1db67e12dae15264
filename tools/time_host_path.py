#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in entry points (host buffers in, host buffers out): the synchronous call
dabgpu_chain_process and the streaming pair dabgpu_chain_submit / dabgpu_chain_collect.

How it measures (round 6; profiles/r06_hostpath_bisect.txt, r06_async_modes.txt, r06_async_series*.txt, r06_stall*.txt):
a fresh process has a slow start -- pinned buffers are allocated inside the first calls, and ONE call somewhere in the
first second may stall for 25 ... 50 ms inside the HIP runtime (plain torch copy + event loops show the same stall) --
so a single short repetition right behind a five-call warm-up, which is what this tool did until round 5, reports
anything between the steady rate and less than half of it.  Now: >= 0.3 s of warm-up, then five repetitions of >= 0.2 s
each; the line carries the MEDIAN rate, the spread, and the longest single call of the measured repetitions.

usage (GPU box): python tools/time_host_path.py [s16|u8|s8] [--json FILE]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
P = importlib.import_module("odr-dabmod_amd")
argv = sys.argv[1:]
JSON = None
if "--json" in argv:
    i = argv.index("--json")
    JSON = argv[i + 1]
    del argv[i:i + 2]
FMT = argv[0] if argv else None
BPS = 8 if FMT is None else 4
pc = time.perf_counter


def measure(step, B, warm_s=0.3, rep_s=0.2, reps=5):
    t0 = pc()
    n = 0
    while pc() - t0 < warm_s or n < 8:
        step(); n += 1
    per_call = (pc() - t0) / n
    calls = max(8, int(rep_s / per_call))
    rates, worst = [], 0.0
    for _ in range(reps):
        t0 = pc()
        for _ in range(calls):
            t1 = pc(); step(); worst = max(worst, pc() - t1)
        rates.append(B * calls / (pc() - t0))
    rates.sort()
    return {"frames_per_s": rates[len(rates) // 2], "min": rates[0], "max": rates[-1], "worst_call_ms": worst * 1e3,
            "calls_per_rep": calls}


def line(tag, B, r):
    print("%s B=%3d  %8.0f frames/s  (median of 5 x %d calls: %.0f ... %.0f; %.2f ms per call, %.2f GB/s of IQ to the host; "
          "longest call %.2f ms)" % (tag, B, r["frames_per_s"], r["calls_per_rep"], r["min"], r["max"],
                                     B / r["frames_per_s"] * 1e3, r["frames_per_s"] * 196608 * BPS / 1e9, r["worst_call_ms"]),
          flush=True)


def modulator(B):
    md = P.Modulator(mode=1, max_frames=B)
    md.set_gain(2, 1.0, 1 / 50000. if FMT is None else 1.0, 4.0)
    md.set_output_format(FMT)
    bits = np.frombuffer(np.random.RandomState(1).bytes(B * 28800), np.uint8).reshape(B, 28800)
    return md, bits


result = {"format": FMT or "complexf", "sync": {}, "async": {}}
print("output format:", FMT or "complexf")
print("synchronous (dabgpu_chain_process; the caller's output buffer is allocated once and reused, like a flowgraph edge's Buffer):")
for B in (1, 8, 32, 64, 256):
    md, bits = modulator(B)
    out = md.chain(bits, 3)
    r = measure(lambda: md.chain(bits, 3, out=out), B)
    line("sync ", B, r)
    result["sync"][B] = r
    md.close()
print("asynchronous (submit / collect, two batches in flight, pinned output handed out without a copy):")
for B in (1, 8, 32):
    md, bits = modulator(B)
    md.submit(bits, 3)
    def step():
        md.submit(bits, 3)
        md.collect(copy=False)
    r = measure(step, B)
    md.collect(copy=False)
    line("async", B, r)
    result["async"][B] = r
    md.close()
if JSON:
    json.dump(result, open(JSON, "w"), indent=1)
