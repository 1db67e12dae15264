#!/usr/bin/env python3
"""Where does the ONE ~40 ms stall of a fresh process come from (profiles/r06_async_series_ab.txt: one collect() between
the 512th and the 1024th one-frame batch, never again)?  Same loop on the synchronous entry point, with one lane, with the
copy back on the lane's own stream removed (device-resident call + dabgpu_synchronize), and after an idle second.
usage (GPU box): python tools/experiments/exp_r06_stall.py"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
P = importlib.import_module("odr-dabmod_amd")
pc = time.perf_counter


def series(name, step, n=3072, block=512):
    ts = np.zeros(n)
    for i in range(n):
        t0 = pc(); step(); ts[i] = pc() - t0
    worst = int(ts.argmax())
    print("%-34s mean %6.1f us  worst call %8.0f us at call %4d   per block max: %s"
          % (name, ts.mean() * 1e6, ts[worst] * 1e6, worst,
             " ".join("%.0f" % (ts[b:b + block].max() * 1e6) for b in range(0, n, block))), flush=True)


bits = np.frombuffer(np.random.RandomState(1).bytes(28800), np.uint8).reshape(1, 28800)
for lanes in (3, 1):
    md = P.Modulator(mode=1, max_frames=1)
    md.set_gain(2, 1.0, 1 / 50000., 4.0)
    md.set_lanes(lanes)
    out = md.chain(bits, 3)
    series("sync host call, lanes=%d" % lanes, lambda: md.chain(bits, 3, out=out))
    md.close()
md = P.Modulator(mode=1, max_frames=1)
md.set_gain(2, 1.0, 1 / 50000., 4.0)
md.submit(bits, 3)
def step():
    md.submit(bits, 3); md.collect(copy=False)
series("submit/collect", step)
series("submit/collect, same context again", step)
md.collect(copy=False)
md.close()
md = P.Modulator(mode=1, max_frames=1)
md.set_gain(2, 1.0, 1 / 50000., 4.0)
d_bits = torch.from_numpy(bits).cuda()
d_out = torch.empty((1, 196608), dtype=torch.complex64, device="cuda")
st = torch.cuda.Stream()
def step_dev():
    md.chain_dev(d_bits, 1, 3, d_out, stream=st.cuda_stream); st.synchronize()
series("device call + stream sync", step_dev)
md.close()
# plain HIP through torch: a 1.5 MB D2H copy into pinned memory + event sync, no library at all
h = torch.empty(196608, dtype=torch.complex64).pin_memory()
ev = torch.cuda.Event()
def step_torch():
    with torch.cuda.stream(st):
        h.copy_(d_out[0], non_blocking=True); ev.record(st)
    ev.synchronize()
series("torch: D2H 1.5 MB + event sync", step_torch)
