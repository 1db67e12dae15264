#!/usr/bin/env python3
"""Round-5 measurements on one MI355X (python tools/experiments/exp_r05.py <what> ...; results as JSON lines on stdout):

  lanes      cfg 3 at B = 1 ... 1024 frames per call on ONE context, 1 ... 4 lanes, HIP-event timed through the two fences
             (dabgpu_wait_for_stream / dabgpu_stream_wait_for) -- what dabgpu_set_lanes buys a ModPlugin-sized caller
  handover   cfg 4 at 4096 frames per call with the FIRFilter -> Resampler hand-over in pieces of P frames (0 = one piece):
             frames/s, board power, shader clock, joules per frame
  rsonly     the x4 resampler + predistorter ALONE (dabgpu_post_process_dev) in calls of 128 frames whose input comes
             from a ring of R frames: R = 128 (201 MB, fits the 256 MiB last-level cache) against R = 4096 (6.4 GB):
             the same launches, only the addresses differ -- the energy of reading the native-rate stream from memory
  tfonly     the frame kernel alone, 128 frames per call, output to a ring of R frames: the WRITE side of the same question
  parts      cfg 4's two kernels one at a time and the chain: time, power, joules per frame
  cfg3power  cfg 3 (or + cfr / nofir / window) at B frames with board power: one arm of an A/B over DABGPU_LIB builds
  chunks     cfg 3 on three lanes: runs of symbols per frame against the batch size
  cfg4small  cfg 4 at small batches on one stream: the chain, the frame kernel alone, the resampler alone
  cfg4lanes  cfg 4 on the context's own stream, 1 ... 4 lanes (the chain pipelined across calls)
"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np
import torch

from power_probe import PowerProbe, sample_load

P = importlib.import_module("odr-dabmod_amd")
dev = torch.device("cuda", 0)
ALGO3, ALGO4 = 28800 + 1572864, 28800 + 6291456


def emit(**kw):
    print(json.dumps(kw), flush=True)


def event_time(st, body, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        body()
    e1.record(st)
    st.synchronize()
    return e0.elapsed_time(e1) / reps


def lanes(argv):
    batches = [int(x) for x in argv] or [1, 4, 16, 64, 256, 1024]
    stages = P.STAGE_GAIN | P.STAGE_FIR
    st = torch.cuda.Stream(device=dev)
    for B in batches:
        nbuf = 4
        with torch.cuda.stream(st):
            bits = [torch.randint(0, 256, (B, 28800), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
            outs = [torch.empty((B, 196608), dtype=torch.complex64, device=dev) for _ in range(nbuf)]
        st.synchronize()
        for n in (1, 2, 3, 4):
            md = P.Modulator(mode=1, device=0, max_frames=B)
            md.set_gain(P.GAIN_VAR, 1.0, 1.0 / 50000.0, 4.0)
            md.set_lanes(n)
            calls = max(40, min(400, 8192 // B))
            k = [0]

            def body():
                md.wait_for_stream(st.cuda_stream)
                for _ in range(calls):
                    i = k[0] % nbuf
                    k[0] += 1
                    md.chain_dev_queued(bits[i], B, stages, outs[i])
                md.stream_wait_for(st.cuda_stream)
            body()
            st.synchronize()
            best = min(event_time(st, body, 3) for _ in range(3)) / calls
            fps = B / (best * 1e-3)
            emit(exp="lanes", frames_per_call=B, lanes=n, us_per_call=round(best * 1e3, 2), frames_per_s=round(fps, 1),
                 roofline_frac=round(ALGO3 * fps / 8e12, 4), calls_per_timing=calls)
            md.close()
        del bits, outs
        torch.cuda.empty_cache()


def _cfg4(B):
    md = P.Modulator(mode=1, device=0, max_frames=B)
    md.set_gain(P.GAIN_VAR, 1.0, 1.0 / 50000.0, 4.0)
    md.set_resampler(2048000, 8192000)
    md.set_poly([1.0, 0.05, -0.01, 0.002, 0.0], [0.0, 0.02, 0.003, 0.0, 0.0])
    return md


def handover(argv):
    B = 4096
    pieces = [int(x) for x in argv] or [0, 32, 64, 128, 256, 512, 0]
    stages = P.STAGE_GAIN | P.STAGE_FIR | P.STAGE_RESAMPLE | P.STAGE_POLY
    st = torch.cuda.Stream(device=dev)
    probe = PowerProbe(0)
    with torch.cuda.stream(st):
        bits = torch.randint(0, 256, (B, 28800), dtype=torch.uint8, device=dev)
        out = torch.empty((B, 4 * 196608), dtype=torch.complex64, device=dev)
    for p in pieces:
        md = _cfg4(B)
        md.set_handover_frames(p)
        # on the context's OWN stream: the consumer is lane 0, the producer lane 1, which the library has probed onto
        # different hardware queues (on a caller's stream the two may share one and run in order: profiles/r05_lane_queues.txt)
        def step():
            md.wait_for_stream(st.cuda_stream)
            md.chain_dev_queued(bits, B, stages, out)
            md.stream_wait_for(st.cuda_stream)
        for _ in range(3):
            step()
        st.synchronize()
        ms = min(event_time(st, step, 5) for _ in range(2))
        fps = B / (ms * 1e-3)
        pw = sample_load(step, 3.0, ms, st, probe=probe)
        rec = dict(exp="handover", piece_frames=p, lanes=md.lanes_info(), frames_per_call=B, ms_per_call=round(ms, 3), frames_per_s=round(fps, 1),
                   roofline_frac=round(ALGO4 * fps / 8e12, 4), power=pw)
        if pw and "watts_avg" in pw:
            rec["mJ_per_frame"] = round(1e3 * pw["watts_avg"] / fps, 4)
        emit(**rec)
        md.close()
        time.sleep(1.0)


def rsonly(argv):
    PIECE = 128
    rings = [int(x) for x in argv] or [128, 4096, 128, 4096]
    NOUT = 1024                                  # output ring: 1024 frames x 6.29 MB = 6.4 GB, written round robin in both arms
    st = torch.cuda.Stream(device=dev)
    probe = PowerProbe(0)
    with torch.cuda.stream(st):
        src = (torch.randn((4096, 196608, 2), device=dev) * 0.2).view(torch.float32)
        native = torch.view_as_complex(src)
        out = torch.empty((NOUT, 4 * 196608), dtype=torch.complex64, device=dev)
    st.synchronize()
    for R in rings:
        md = _cfg4(PIECE)
        k = [0]

        def step():
            # one "step" = 32 calls of 128 frames = 4096 frames
            for _ in range(32):
                i = (k[0] * PIECE) % R
                o = (k[0] * PIECE) % NOUT
                k[0] += 1
                md.post_process_dev(native[i:i + PIECE], P.STAGE_RESAMPLE | P.STAGE_POLY, out[o:o + PIECE], stream=st.cuda_stream)
        step()
        st.synchronize()
        ms = min(event_time(st, step, 3) for _ in range(2))
        fps = 32 * PIECE / (ms * 1e-3)
        pw = sample_load(step, 3.0, ms, st, probe=probe)
        rec = dict(exp="rsonly", ring_frames=R, ring_MB=round(R * 196608 * 8 / 1e6), frames_per_call=PIECE,
                   frames_per_s=round(fps, 1), power=pw)
        if pw and "watts_avg" in pw:
            rec["mJ_per_frame"] = round(1e3 * pw["watts_avg"] / fps, 4)
        emit(**rec)
        md.close()
        time.sleep(1.0)


def tfonly(argv):
    """The frame kernel ALONE (cfg 3) in calls of 128 frames whose output goes to a ring of R frames: R = 128 (201 MB:
    the lines are rewritten while they are still in the last-level cache) against R = 4096 (6.4 GB) -- the WRITE side of
    the hand-over."""
    PIECE = 128
    rings = [int(x) for x in argv] or [128, 4096, 128, 4096]
    st = torch.cuda.Stream(device=dev)
    probe = PowerProbe(0)
    with torch.cuda.stream(st):
        bits = torch.randint(0, 256, (PIECE, 28800), dtype=torch.uint8, device=dev)
        out = torch.empty((4096, 196608), dtype=torch.complex64, device=dev)
    st.synchronize()
    stages = P.STAGE_GAIN | P.STAGE_FIR
    for R in rings:
        md = P.Modulator(mode=1, device=0, max_frames=PIECE)
        md.set_gain(P.GAIN_VAR, 1.0, 1.0 / 50000.0, 4.0)
        k = [0]

        def step():
            for _ in range(32):
                o = (k[0] * PIECE) % R
                k[0] += 1
                md.chain_dev(bits, PIECE, stages, out[o:o + PIECE], stream=st.cuda_stream)
        step()
        st.synchronize()
        ms = min(event_time(st, step, 3) for _ in range(2))
        fps = 32 * PIECE / (ms * 1e-3)
        pw = sample_load(step, 3.0, ms, st, probe=probe)
        rec = dict(exp="tfonly", ring_frames=R, ring_MB=round(R * 196608 * 8 / 1e6), frames_per_call=PIECE,
                   frames_per_s=round(fps, 1), power=pw)
        if pw and "watts_avg" in pw:
            rec["mJ_per_frame"] = round(1e3 * pw["watts_avg"] / fps, 4)
        emit(**rec)
        md.close()
        time.sleep(1.0)


def parts(argv):
    """cfg 4 at 4096 frames per call, its two kernels one at a time: the frame kernel alone (cfg 3 into a 6.4 GB buffer),
    the resampler + predistorter alone (dabgpu_post_process_dev on that buffer), and the chain."""
    B = 4096
    st = torch.cuda.Stream(device=dev)
    probe = PowerProbe(0)
    with torch.cuda.stream(st):
        bits = torch.randint(0, 256, (B, 28800), dtype=torch.uint8, device=dev)
        native = torch.empty((B, 196608), dtype=torch.complex64, device=dev)
        out = torch.empty((B, 4 * 196608), dtype=torch.complex64, device=dev)
    md = _cfg4(B)
    h = st.cuda_stream
    steps = {"frame_kernel": lambda: md.chain_dev(bits, B, P.STAGE_GAIN | P.STAGE_FIR, native, stream=h),
             "resampler_poly": lambda: md.post_process_dev(native, P.STAGE_RESAMPLE | P.STAGE_POLY, out, stream=h),
             "chain": lambda: md.chain_dev(bits, B, 15, out, stream=h)}
    for name in ("frame_kernel", "resampler_poly", "chain", "frame_kernel", "resampler_poly", "chain"):
        step = steps[name]
        for _ in range(3):
            step()
        st.synchronize()
        ms = min(event_time(st, step, 5) for _ in range(2))
        pw = sample_load(step, 3.0, ms, st, probe=probe)
        rec = dict(exp="parts", part=name, frames_per_call=B, ms_per_call=round(ms, 3), frames_per_s=round(B / (ms * 1e-3), 1), power=pw)
        if pw and "watts_avg" in pw:
            rec["mJ_per_frame"] = round(pw["watts_avg"] * ms / B, 4)
        emit(**rec)
        time.sleep(1.0)
    md.close()


def chunks(argv):
    """cfg 3 on three lanes: workgroups (runs of symbols) per frame against the batch size -- with three launches in flight
    the chip is filled by FEWER, LONGER runs per launch than auto_chunks' 1024 workgroups, and every run costs a prologue and
    one look-ahead transform."""
    batches = [int(x) for x in argv] or [4, 16, 64, 256]
    stages = P.STAGE_GAIN | P.STAGE_FIR
    st = torch.cuda.Stream(device=dev)
    for B in batches:
        with torch.cuda.stream(st):
            bits = [torch.randint(0, 256, (B, 28800), dtype=torch.uint8, device=dev) for _ in range(4)]
            outs = [torch.empty((B, 196608), dtype=torch.complex64, device=dev) for _ in range(4)]
        st.synchronize()
        for ch in (0, 77, 39, 26, 20, 16, 13, 11, 8, 6, 4, 3, 2, 1):
            if ch and B * ch > 4096:
                continue
            md = P.Modulator(mode=1, device=0, max_frames=B, chunks_per_frame=ch)
            md.set_gain(P.GAIN_VAR, 1.0, 1.0 / 50000.0, 4.0)
            calls = max(40, min(400, 8192 // B))
            k = [0]

            def body():
                md.wait_for_stream(st.cuda_stream)
                for _ in range(calls):
                    i = k[0] & 3
                    k[0] += 1
                    md.chain_dev_queued(bits[i], B, stages, outs[i])
                md.stream_wait_for(st.cuda_stream)
            body()
            st.synchronize()
            best = min(event_time(st, body, 3) for _ in range(3)) / calls
            fps = B / (best * 1e-3)
            emit(exp="chunks", frames_per_call=B, chunks_per_frame=ch, workgroups_per_call=(B * ch if ch else None),
                 us_per_call=round(best * 1e3, 2), frames_per_s=round(fps, 1), roofline_frac=round(ALGO3 * fps / 8e12, 4))
            md.close()
        del bits, outs
        torch.cuda.empty_cache()


def cfg4small(argv):
    """cfg 4 (frame kernel -> x4 resampler + predistorter) at small batches, one context: microseconds per call of the chain,
    of the frame kernel alone and of the resampler alone -- what pipelining the two ACROSS calls could hide."""
    st = torch.cuda.Stream(device=dev)
    h = st.cuda_stream
    for B in [int(x) for x in argv] or [1, 4, 16, 64, 256]:
        md = _cfg4(B)
        with torch.cuda.stream(st):
            bits = torch.randint(0, 256, (B, 28800), dtype=torch.uint8, device=dev)
            native = torch.empty((B, 196608), dtype=torch.complex64, device=dev)
            out = torch.empty((B, 4 * 196608), dtype=torch.complex64, device=dev)
        steps = {"chain": lambda: md.chain_dev(bits, B, 15, out, stream=h),
                 "frame_kernel": lambda: md.chain_dev(bits, B, 3, native, stream=h),
                 "resampler_poly": lambda: md.post_process_dev(native, P.STAGE_RESAMPLE | P.STAGE_POLY, out, stream=h)}
        rec = dict(exp="cfg4small", frames_per_call=B)
        calls = max(20, min(300, 4096 // B))
        for name, step in steps.items():
            def body():
                for _ in range(calls):
                    step()
            body()
            st.synchronize()
            rec[name + "_us"] = round(min(event_time(st, body, 2) for _ in range(3)) / calls * 1e3, 2)
        rec["frames_per_s"] = round(B / (rec["chain_us"] * 1e-6), 1)
        rec["roofline_frac"] = round(ALGO4 * rec["frames_per_s"] / 8e12, 4)
        emit(**rec)
        md.close()


def cfg4lanes(argv):
    """cfg 4 on the context's own stream, 1 ... 4 lanes: the native-rate part of call i + 1 on a lane of its own while lane 0
    runs the resampler of call i (the chain pipelined across calls)."""
    st = torch.cuda.Stream(device=dev)
    stages = 15
    for B in [int(x) for x in argv] or [1, 4, 16, 64, 256]:
        with torch.cuda.stream(st):
            bits = [torch.randint(0, 256, (B, 28800), dtype=torch.uint8, device=dev) for _ in range(4)]
            outs = [torch.empty((B, 4 * 196608), dtype=torch.complex64, device=dev) for _ in range(4)]
        st.synchronize()
        for n in (1, 2, 3, 4):
            md = _cfg4(B)
            md.set_lanes(n)
            calls = max(20, min(300, 4096 // B))
            k = [0]

            def body():
                md.wait_for_stream(st.cuda_stream)
                for _ in range(calls):
                    i = k[0] & 3
                    k[0] += 1
                    md.chain_dev_queued(bits[i], B, stages, outs[i])
                md.stream_wait_for(st.cuda_stream)
            body()
            st.synchronize()
            best = min(event_time(st, body, 2) for _ in range(3)) / calls
            fps = B / (best * 1e-3)
            emit(exp="cfg4lanes", frames_per_call=B, lanes=n, us_per_call=round(best * 1e3, 2), frames_per_s=round(fps, 1),
                 roofline_frac=round(ALGO4 * fps / 8e12, 4))
            md.close()
        del bits, outs
        torch.cuda.empty_cache()


def cfg3power(argv):
    """cfg 3 (or cfg3 + option) at B frames per launch with board power: one arm of an A/B over libraries
    (DABGPU_LIB=tools/_variants/libdabgpu_x.so python tools/experiments/exp_r05.py cfg3power [B] [cfr|nofir|window] [tag])."""
    B = int(argv[0]) if argv else 32768
    option = argv[1] if len(argv) > 1 and argv[1] != "-" else None
    tag = argv[2] if len(argv) > 2 else os.path.basename(os.environ.get("DABGPU_LIB", "product"))
    st = torch.cuda.Stream(device=dev)
    probe = PowerProbe(0)
    md = P.Modulator(mode=1, device=0, max_frames=B)
    md.set_gain(P.GAIN_VAR, 1.0, 1.0 / 50000.0, 4.0)
    if option == "cfr":
        md.set_cfr(True, 50.0, 0.1)
    elif option == "window":
        md.set_window_overlap(10)
    stages = P.STAGE_GAIN | (0 if option == "nofir" else P.STAGE_FIR)
    with torch.cuda.stream(st):
        bits = torch.randint(0, 256, (B, 28800), dtype=torch.uint8, device=dev)
        out = torch.empty((B, 196608), dtype=torch.complex64, device=dev)
    step = lambda: md.chain_dev(bits, B, stages, out, stream=st.cuda_stream)
    for _ in range(5):
        step()
    st.synchronize()
    for rep in range(2):
        ms = min(event_time(st, step, 10) for _ in range(2))
        fps = B / (ms * 1e-3)
        pw = sample_load(step, 3.0, ms, st, probe=probe)
        rec = dict(exp="cfg3power", lib=tag, option=option, frames_per_call=B, ms_per_call=round(ms, 4), frames_per_s=round(fps, 1),
                   roofline_frac=round(ALGO3 * fps / 8e12, 4))
        if pw and "watts_avg" in pw:
            rec.update(watts=pw["watts_avg"], sclk_MHz=pw["sclk_MHz_avg"], mJ_per_frame=round(1e3 * pw["watts_avg"] / fps, 4))
        emit(**rec)
    md.close()


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "lanes"
    {"lanes": lanes, "handover": handover, "rsonly": rsonly, "tfonly": tfonly, "parts": parts, "cfg3power": cfg3power, "chunks": chunks, "cfg4small": cfg4small, "cfg4lanes": cfg4lanes}[what](sys.argv[2:])
