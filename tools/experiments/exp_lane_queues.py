#!/usr/bin/env python3
"""Do a context's lanes overlap whatever the process did before?  cfg 3, 16 frames per call on three lanes, microseconds per
call (four regions of 3000 calls) in five situations: a fresh process, a second context, after a large workload, after the
power probe, after six contexts with lanes have come and gone.  Before the lanes were probed onto hardware queues of their
own (stream_probe.hip) the answer depended on the history: profiles/r05_lane_queues.txt.  Run on the GPU box, from the
repository root; GPU_MAX_HW_QUEUES=2 / 8 in the environment shows the runtime's side of it."""
import importlib, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
P = importlib.import_module("odr-dabmod_amd")
dev = torch.device("cuda", 0)
def measure(tag, B=16, calls=3000, inside=False, pre=None):
    if pre: pre()
    st = torch.cuda.Stream(device=dev)
    md = P.Modulator(mode=1, device=0, max_frames=B)
    md.set_gain(P.GAIN_VAR, 1.0, 1.0 / 50000.0, 4.0)
    md.set_lanes(3)
    with torch.cuda.stream(st):
        bits = [torch.randint(0, 256, (B, 28800), dtype=torch.uint8, device=dev) for _ in range(4)]
        outs = [torch.empty((B, 196608), dtype=torch.complex64, device=dev) for _ in range(4)]
    st.synchronize()
    k = [0]
    def body():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        md.wait_for_stream(st.cuda_stream)
        for _ in range(calls):
            i = k[0] & 3; k[0] += 1
            md.chain_dev_queued(bits[i], B, 3, outs[i])
        md.stream_wait_for(st.cuda_stream)
        e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / calls * 1e3
    res = []
    if inside:
        with torch.cuda.stream(st):
            for _ in range(4): res.append(body())
    else:
        for _ in range(4): res.append(body())
    md.close()
    print(tag, ["%.2f" % r for r in res], flush=True)
measure("fresh, outside stream ctx")
measure("fresh, inside stream ctx", inside=True)
def big():
    md = P.Modulator(mode=1, device=0, max_frames=8192)
    md.set_gain(P.GAIN_VAR, 1.0, 1.0 / 50000.0, 4.0)
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        b = torch.randint(0, 256, (8192, 28800), dtype=torch.uint8, device=dev)
        o = torch.empty((8192, 196608), dtype=torch.complex64, device=dev)
        for _ in range(3): md.chain_dev(b, 8192, 3, o, stream=st.cuda_stream)
        st.synchronize()
    md.close(); del b, o; torch.cuda.empty_cache()
measure("after a big workload", pre=big)
import threading
sys.path.insert(0, os.path.join(ROOT, "tools"))
from power_probe import PowerProbe
p = PowerProbe(0)
measure("after PowerProbe()")
# many modulators created and closed
for _ in range(6):
    m = P.Modulator(mode=1, device=0, max_frames=16); m.set_lanes(3)
    o = torch.empty((16, 196608), dtype=torch.complex64, device=dev); b = torch.randint(0, 256, (16, 28800), dtype=torch.uint8, device=dev)
    for _ in range(6): m.chain_dev_queued(b, 16, 3, o)
    m.synchronize(); m.close()
measure("after six contexts with lanes came and went")
