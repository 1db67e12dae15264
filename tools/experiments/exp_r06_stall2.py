#!/usr/bin/env python3
"""profiles/r06_stall.txt: plain torch (D2H copy + event record + event synchronize) shows the same one-off stall as
submit/collect; hipStreamSynchronize loops do not.  Which half is it -- the record or the host-side wait?
usage (GPU box): python tools/experiments/exp_r06_stall2.py"""
import time
import numpy as np
import torch
pc = time.perf_counter


def series(name, step, n=3072, block=512):
    ts = np.zeros(n)
    for i in range(n):
        t0 = pc(); step(); ts[i] = pc() - t0
    worst = int(ts.argmax())
    print("%-46s mean %6.1f us  worst %8.0f us at call %4d   per block max: %s"
          % (name, ts.mean() * 1e6, ts[worst] * 1e6, worst,
             " ".join("%.0f" % (ts[b:b + block].max() * 1e6) for b in range(0, n, block))), flush=True)


d = torch.zeros(196608, dtype=torch.complex64, device="cuda")
h = torch.empty(196608, dtype=torch.complex64).pin_memory()
torch.cuda.synchronize()


def mk(wait, timing=False, fresh_stream=True):
    st = torch.cuda.Stream()
    ev = torch.cuda.Event(enable_timing=timing)
    def step():
        with torch.cuda.stream(st):
            h.copy_(d, non_blocking=True)
            if wait != "stream-noevent":
                ev.record(st)
        if wait == "event":
            ev.synchronize()
        elif wait == "query":
            while not ev.query():
                pass
        else:
            st.synchronize()
    return step


series("copy + record + hipEventSynchronize", mk("event"))
series("copy + record + hipEventQuery spin", mk("query"))
series("copy + record + hipStreamSynchronize", mk("stream"))
series("copy + hipStreamSynchronize (no event)", mk("stream-noevent"))
series("copy + record(timing) + hipEventSynchronize", mk("event", timing=True))
k = torch.zeros(64, device="cuda")
st = torch.cuda.Stream(); ev = torch.cuda.Event()
def kstep():
    with torch.cuda.stream(st):
        k.add_(1); ev.record(st)
    ev.synchronize()
series("kernel + record + hipEventSynchronize", kstep)
