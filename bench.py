#!/usr/bin/env python3
"""bench.py -- Mode-I TX frames/s of the MI355X-native DAB hot path.

    python bench.py --gpus N --steps K --warmup W [--workload cfg3] [--frames B]

A "step" is one pass of the hot path over one batch of B synthetic transmission
frames (inputs resident in HBM before the timed region).  For N > 1 the driver
launches one rank per GPU through torch.distributed.run; frames are independent
units, so every rank runs its own stream of batches (weak scaling, no data-path
collective); the only collectives are the barrier around the timed region and
the MAX-reduce of the elapsed time.

Workloads (BASELINE.json configs):
  cfg3 (default)  coded bits -> QPSK/interleave/diff-mod -> 77x IFFT -> GainControl(var)
                  -> guard interval -> FIRFilter(45 default taps), native 2.048 Msps
  cfg2            SignalMultiplexer output (cf32) -> OfdmGenerator + GuardIntervalInserter
  cfg4            cfg3 + Resampler 2.048->8.192 Msps + MemlessPoly
  ifft_fir_stage  SignalMultiplexer output (cf32) -> OfdmGenerator + GainControl(var) + guard + FIRFilter: the
                  "IFFT+FIR stage" of the north-star target, SURVEY 8(d) (2 519 040 algorithmic B/TF)

One JSON line on stdout (rank 0).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PREWARM = 3            # untimed steps run before the W warm-up steps of the contract
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
SMALL_BATCH_LANES = 3   # the library's default (dabgpu_set_lanes): batches in flight inside one context, B = 1 / 16 / 256

# ALGORITHMIC (compulsory) bytes per transmission frame, SURVEY 8(d) / BASELINE.md section 3
ALGO_BYTES = {
    "cfg2": 946176 + 1572864,        # 77x1536 cf32 read + 196608 cf32 written
    "cfg3": 28800 + 1572864,         # coded bits read + native-rate IQ written
    "ifft_fir_stage": 946176 + 1572864,  # SURVEY 8(d) "IFFT+Gain+Guard+FIR (IFFT+FIR stage)": carriers in, IQ out
    "cfg4": 28800 + 6291456,         # coded bits read + 8.192 Msps IQ written
}


# Nominal fp32 operations per transmission frame of what the kernels EXECUTE: 5 N log2 N per
# N-point transform, 2 per real-by-complex multiply-add, 40 per predistorted sample.  cfg 3: 77 transforms of 2048 points +
# the spectral multiply (77 x 1536 x 6) + per boundary a 160-tap inverse filter over 44 outputs and a 990-term triangular
# correction ((7040 + 990) x 4).  cfg 4: cfg 3 + per hop four 4096-point transforms and 8192 predistorted samples (96 hops).
VALU_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector peak
_CFG3_FLOPS = 77 * 5 * 2048 * 11 + 77 * 1536 * 6 + 77 * (7040 + 990) * 4
EXEC_FLOPS = {"cfg3": _CFG3_FLOPS, "cfg4": _CFG3_FLOPS + 96 * (4 * 5 * 4096 * 12 + 8192 * 40)}


sys.path.insert(0, os.path.join(ROOT, "tools"))
from power_probe import PowerProbe, sample_load  # noqa: E402  (board power / shader clock of the HIP device's own card, by PCI address)


def pkg():
    mod = importlib.import_module("odr-dabmod_amd")
    sys.modules["odr_dabmod_amd"] = mod
    return mod


def cpu_model():
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(workload, seconds_budget=16.0):
    """The oracle's chain (kind "port": the -O3 -march=native build of oracle/dab_oracle.c, liboracle_fast.so, with fp32
    transforms -- FFTW3f when the host has libfftw3f.so.3, else the file's own radix-4 Stockham; `fft` says which) timed on
    a bounded sample of the same workload.  `value` is the host's best: C independent single-thread streams side by side
    (C = the cores this process may use), or -- if that ever wins -- cores/4 streams in the reference's threading model.
    Next to it, as SURVEY 8(d) asks: ONE stream in the reference's threading model (modulator thread + one thread per
    PipelinedModCodec -- GainControl, FIRFilter, MemlessPoly --, MemlessPoly split over further workers,
    src/ModPlugin.cpp:90-154), cores/4 such streams, and one plain thread.  Only frames that REACH the output are counted
    (a pipelined stage holds one frame back at the start of every call)."""
    import threading
    import numpy as np
    import oracle as O
    from tests.golden.synth import synth_bits
    kw = dict(mode=1, fast=True)
    if workload == "cfg2":
        kw.update(stages=0)
    elif workload in ("cfg3", "ifft_fir_stage"):
        kw.update(stages=3, gain_mode=2, normalise=1.0 / 50000.0)
    else:
        kw.update(stages=15, gain_mode=2, normalise=1.0 / 50000.0, out_rate=8192000,
                  am=(1.0, 0.05, -0.01, 0.002, 0.0), pm=(0.0, 0.02, 0.003, 0.0, 0.0))
    nb = 32                    # frames per call: the pipelined model's start-up (threads, held-back frames) amortised
    bits = np.stack([synth_bits(28800, seed=500 + i) for i in range(nb)])
    cores = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    try:                                   # a container's CPU quota, when there is one (cgroup v2)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    nstreams = max(1, cores // 4)
    poly_threads = 0 if workload != "cfg4" else max(1, min(8, cores // nstreams - 3))

    def timed_run(nthreads, seconds, pipelined):
        """nthreads independent streams, each looping nb-frame calls until the deadline; frames OUT / elapsed."""
        chains = [O.Chain(**kw) for _ in range(nthreads)]
        outs = [np.empty((nb, c.out_samples_per_tf), np.complex64) for c in chains]
        for c, o in zip(chains, outs):
            c.process(bits[:1], o[:1])                    # touch the pages before the clock starts
        done = [0] * nthreads
        start = threading.Barrier(nthreads + 1)
        deadline = [0.0]

        def worker(i):
            start.wait()
            while time.perf_counter() < deadline[0]:
                if pipelined:
                    done[i] += len(chains[i].process_pipelined(bits, poly_threads, outs[i]))
                else:
                    chains[i].process(bits, outs[i])
                    done[i] += nb

        threads = [threading.Thread(target=worker, args=(i,)) for i in range(nthreads)]
        for t in threads:
            t.start()
        deadline[0] = time.perf_counter() + seconds + 0.05
        start.wait()
        t0 = time.perf_counter()
        for t in threads:
            t.join()
        dt = time.perf_counter() - t0
        return sum(done), dt

    n1, dt1 = timed_run(1, 0.15 * seconds_budget, False)
    nc, dtc = timed_run(cores, 0.35 * seconds_budget, False)
    np1, dtp1 = timed_run(1, 0.2 * seconds_budget, True)
    nn, dtn = timed_run(nstreams, 0.3 * seconds_budget, True)
    all_cores, ref_model = nc / dtc, nn / dtn
    return {"value": round(max(all_cores, ref_model), 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "fft": O.fft_engine(),
            "value_is": "independent single-thread streams on every core" if all_cores >= ref_model
                        else "cores/4 streams in the reference threading model",
            "independent_streams_all_cores": round(all_cores, 3),
            "reference_threading_streams": nstreams, "reference_threading_value": round(ref_model, 3),
            "one_stream_reference_threading": round(np1 / dtp1, 3),
            "single_thread_value": round(n1 / dt1, 3), "cpu_model": cpu_model(),
            "sample": "%s chain, Mode I, %d frames per call: %d single-thread streams side by side %d frames in %.1f s; "
                      "%d streams in the reference's threading model (modulator thread + one thread per pipelined "
                      "stage%s) %d frames out in %.1f s; one such stream %d frames out in %.1f s; one plain thread %d "
                      "frames in %.1f s (oracle/dab_oracle.c, -O3 -march=native, fp32 transforms: %s)"
                      % (workload, nb, cores, nc, dtc, nstreams,
                         ", %d MemlessPoly workers" % poly_threads if poly_threads else "", nn, dtn, np1, dtp1, n1, dt1,
                         O.fft_engine())}


def host_tools(device_index=0, budget_s=45.0):
    """The DELIVERABLE's own throughput (north_star: "stages drop into DabModulator.cpp") -- ETI file in,
    IQ out to /dev/null, PCIe both ways inside the timed region -- of the four host programs, where they have been built:
      cfg1_end_to_end            odr-dabmod_amd/host/dabmod_file (this repository's front-end + DabGpuChain): BASELINE config 1,
                                 frame by frame (DabGpuChain::process) and --batch 32 (submit / collect), complexf and u8;
      fused_in_reference_graph   oracle/_ref/dabmod_fused: the reference's own DabModulator.cpp / Flowgraph.cpp / front end
                                 with ONE DabGpuChain node, config 1 and config 3;
      dropin_per_stage           oracle/_ref/dabmod_dropin: the reference's UNMODIFIED graph builder on the per-stage
                                 drop-ins (PCIe both ways per stage), config 3.
    Rates exclude process start-up: every case runs the same 2000-frame ETI file with --loop 1 and --loop L and reports
    (L - 1) x 500 transmission frames over the difference of the two wall times.  The comparable CPU figure is
    cpu_baseline.one_stream_reference_threading (one stream of the port in the reference's threading model, hot path
    only).  A program that is not there (oracle/_ref exists only where the reference tree was at hand) is skipped."""
    import subprocess
    import tempfile
    from tests.golden.synth import synth_eti
    host = os.path.join(ROOT, "odr-dabmod_amd", "host")
    ref = os.path.join(ROOT, "oracle", "_ref")
    norm = repr(1.0 / 50000.0)
    cases = [
        ("cfg1_end_to_end", "frame_by_frame_complexf", os.path.join(host, "dabmod_file"), []),
        ("cfg1_end_to_end", "batch32_complexf", os.path.join(host, "dabmod_file"), ["--batch", "32"]),
        ("cfg1_end_to_end", "frame_by_frame_u8", os.path.join(host, "dabmod_file"), ["--format", "u8"]),
        ("cfg1_end_to_end", "batch32_u8", os.path.join(host, "dabmod_file"), ["--format", "u8", "--batch", "32"]),
        ("cfg1_end_to_end", "front_end_alone_no_gpu", os.path.join(host, "dabmod_file"), ["--bits-only"]),
        ("fused_in_reference_graph", "cfg1_complexf", os.path.join(ref, "dabmod_fused"), []),
        ("fused_in_reference_graph", "cfg3_complexf", os.path.join(ref, "dabmod_fused"), ["--fir", "default", "--normalise", norm]),
        ("fused_in_reference_graph", "cfg3_s16", os.path.join(ref, "dabmod_fused"),
         ["--fir", "default", "--normalise", "1.0", "--format", "s16"]),
        ("dropin_per_stage", "cfg3_complexf", os.path.join(ref, "dabmod_dropin"), ["--fir", "default", "--normalise", norm]),
    ]
    out = {"timing": "(L - 1) x 500 transmission frames / (wall(--loop L) - wall(--loop 1)); output to /dev/null; one process, "
                     "one context, the GPU this bench runs on", "unit": "transmission frames/s"}
    env = dict(os.environ, DABGPU_DEVICE=str(device_index))
    t_start = time.perf_counter()
    with tempfile.TemporaryDirectory() as tmp:
        fin = os.path.join(tmp, "in.eti")
        synth_eti(2000).tofile(fin)              # a multiple of 1000 frames: looping keeps FCT / FP aligned

        def wall(tool, opts, loops):
            t0 = time.perf_counter()
            r = subprocess.run([tool, fin, "/dev/null"] + opts + ["--loop", str(loops)], env=env, capture_output=True,
                               text=True, timeout=120)
            if r.returncode != 0:
                raise RuntimeError((r.stderr or r.stdout)[-160:])
            return time.perf_counter() - t0

        for group, name, tool, opts in cases:
            grp = out.setdefault(group, {})
            if not os.path.exists(tool):
                grp[name] = {"skipped": os.path.relpath(tool, ROOT) + " is not built here"}
                continue
            if time.perf_counter() - t_start > budget_s:
                grp[name] = {"skipped": "time budget of this leg spent"}
                continue
            try:
                wall(tool, opts, 1)                              # (page the binary and the libraries in)
                t1 = wall(tool, opts, 1)
                t3 = wall(tool, opts, 3)
                per_loop = max((t3 - t1) / 2, 1e-3)
                L = int(max(3, min(40, 1 + 2.5 / per_loop)))
                tl = wall(tool, opts, L) if L > 3 else t3
                grp[name] = {"frames_per_s": round((L - 1) * 500 / max(tl - t1, 1e-6), 1), "loops": L,
                             "startup_s": round(t1 - per_loop, 2), "options": " ".join(opts)}
            except Exception as ex:
                grp[name] = {"error": str(ex)[:200]}
    return out


def free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def rank_devices(n):
    """Device ordinal of every local rank.  Default: rank r -> GPU r.  DABGPU_BENCH_DEVICES="0,0" maps ranks onto
    explicit ordinals (tests put two ranks on the one leased GPU)."""
    spec = os.environ.get("DABGPU_BENCH_DEVICES")
    if spec:
        devs = [int(x) for x in spec.split(",")]
        if len(devs) < n:
            raise SystemExit("bench.py: DABGPU_BENCH_DEVICES names %d devices for %d ranks" % (len(devs), n))
        return devs[:n]
    return list(range(n))


def launch_ranks(n, argv, dry_run):
    """`python bench.py --gpus N` outside torch.distributed.run: start N ranks of this script, one per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, as the driver's launcher sets them), rank 0's
    stdout on ours, the others' on stderr.  Returns the exit status of the job."""
    import subprocess
    devs = rank_devices(n)
    if not dry_run:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have <= max(devs):
            raise SystemExit("bench.py: --gpus %d needs devices %s, this node has %d GPU(s)" % (n, devs, have))
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DABGPU_BENCH_CHILD="1",
                   DABGPU_BENCH_PARENT=str(os.getpid()))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                st = p.poll()
                if st is None:
                    continue
                pending.remove(p)
                if st != 0 and rc == 0:
                    rc = st
                    for q in pending:            # one rank failed: the others would wait in a barrier for ever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def dry_run(args, streams):
    """--dry-run: the launcher, the process group (gloo when there is no GPU), the barrier-bracketed timing and the
    one-line contract with a sleep standing in for the kernels -- what the CPU tests exercise at N = 2."""
    if os.environ.get("DABGPU_BENCH_FAIL_RANK") is not None and os.environ.get("DABGPU_BENCH_FAIL_RANK") == os.environ.get("RANK"):
        raise SystemExit(3)        # (tests: a rank that dies before the rendezvous -- the launcher must end the others)
    grp = streams.StreamGroup(backend=os.environ.get("DABGPU_DIST_BACKEND", "gloo"))
    B = args.frames
    wall = grp.timed(lambda: time.sleep(0.01 * (1 + grp.rank)), args.steps, lambda: None)
    line = {"metric": "Mode-I TX frames/sec (196608 IQ/frame)", "value": round(grp.job_frames_per_second(B, args.steps, wall), 2),
            "unit": "frames/s", "n_gpus": grp.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "dry-run (no kernels)",
            "config": {"workload": "dry run of the %d-rank harness" % grp.world, "frames_per_step_per_gpu": B,
                       "devices": rank_devices(grp.world), "stream_seed": grp.stream_seed(42)}}
    if grp.rank == 0:
        grp.emit(json.dumps(line))
    grp.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--dry-run", action="store_true",
                    help="run the N-rank harness (launcher, process group, timing, JSON line) without any kernel")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=sorted(ALGO_BYTES))
    ap.add_argument("--frames", type=int, default=32768,
                    help="transmission frames per step per GPU (32768 = 51.5 GB of IQ per step: sized for 288 GB of HBM)")
    ap.add_argument("--chunks", type=int, default=0, help="workgroups per frame (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--counters", choices=["live", "replay", "off"], default="live",
                    help="profiler counters of the headline workload: collected by this run through rocprofv3 --pmc (N = 1; falls "
                         "back to replay), replayed from profiles/traffic.json, or none")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--gather", type=int, default=0, metavar="FRAMES",
                    help="after the timed region, gather FRAMES frames of IQ from every rank on rank 0 (the optional "
                         "final IQ gather over RCCL / xGMI) and report its time separately; 0 = off")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None:
        if args.gpus > 1:
            # not under torch.distributed.run: be the launcher (one rank per GPU, rank 0's line on our stdout)
            sys.exit(launch_ranks(args.gpus, sys.argv[1:], args.dry_run))
    elif int(env_world) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s: launch one rank per GPU (python -m torch.distributed.run "
                         "--nproc-per-node %d bench.py --gpus %d ..., or plain `python bench.py --gpus %d`)"
                         % (args.gpus, env_world, args.gpus, args.gpus, args.gpus))

    streams = importlib.import_module("odr-dabmod_amd.streams")
    if args.dry_run:
        return dry_run(args, streams)

    import numpy as np
    import torch

    P = pkg()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    devs = rank_devices(args.gpus)
    device_index = devs[int(os.environ.get("LOCAL_RANK", "0"))]
    if device_index >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %s wants GPU %d, this node has %d" % (os.environ.get("RANK", "0"), device_index,
                                                                              torch.cuda.device_count()))
    torch.cuda.set_device(device_index)
    # torch.distributed over RCCL when N > 1 (DABGPU_DIST_BACKEND=gloo: ranks that share a GPU in the tests)
    grp = streams.StreamGroup(backend=os.environ.get("DABGPU_DIST_BACKEND", "nccl"), device_index=device_index)
    rank, world = grp.rank, grp.world
    local_rank = device_index
    dev = torch.device("cuda", device_index)

    power_of = {}
    probe0 = PowerProbe(device_index)      # (created before any load: its idle reading is the card's idle power)

    def run_workload(workload, B, steps, warmup, fmt=None, option=None, power_seconds=0.0, lanes=0, repeats=1):
        """lanes > 0: the calls go to the context's OWN stream (dabgpu_chain_process_dev with stream NULL) and rotate over
        that many internal lanes -- batches in flight inside ONE context, include/dabgpu.h -- ordered against the timing
        stream by the two fences, so that the HIP events on it bracket all of them.  lanes = 0: every call on the timing
        stream itself, one after the other."""
        md = P.Modulator(mode=1, device=local_rank, max_frames=B, chunks_per_frame=args.chunks)
        if lanes:
            md.set_lanes(lanes)
        # rows f-3 / f-4 on the cfg 3 chain, with the values doc/example.ini of the reference suggests
        if option == "cfr":
            md.set_cfr(True, 50.0, 0.1)
        elif option == "window":
            md.set_window_overlap(10)
        # (s16: file-style normalisation for the native-rate chain, 30000/50000 where the polynomial needs |x| < 1)
        md.set_gain(P.GAIN_VAR, 1.0, (1.0 if workload != "cfg4" else 0.6) if fmt else 1.0 / 50000.0, 4.0)
        md.set_output_format(fmt)
        if workload == "cfg2":
            stages, from_bits = 0, False
        elif workload == "ifft_fir_stage":
            stages, from_bits = P.STAGE_GAIN | P.STAGE_FIR, False
        elif workload == "cfg3":
            # (option "nofir": the reference's default configuration, firfilter.enabled = 0, src/ConfigParser.cpp:198)
            stages, from_bits = P.STAGE_GAIN | (0 if option == "nofir" else P.STAGE_FIR), True
        else:
            stages, from_bits = P.STAGE_GAIN | P.STAGE_FIR | P.STAGE_RESAMPLE | P.STAGE_POLY, True
            md.set_resampler(2048000, 8192000)
            md.set_poly([1.0, 0.05, -0.01, 0.002, 0.0], [0.0, 0.02, 0.003, 0.0, 0.0])
        ns = md.out_samples_per_frame(stages)
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            # synthetic hot-path input: uniform random bytes, a different stream per rank, generated ON the
            # device -- eight ranks of 32768 frames would otherwise each draw 0.94 GB through numpy and stage it through
            # pageable host memory (2 GB of host RAM and ~3 s per rank before the first launch)
            gen = torch.Generator(device=dev)
            gen.manual_seed(grp.stream_seed(42))
            d_bits = torch.randint(0, 256, (B, 28800), dtype=torch.uint8, device=dev, generator=gen)
            if from_bits:
                d_in = d_bits
            else:
                # cfg2 input = SignalMultiplexer output: produce it on the device once
                # (QPSK constellation points of the same bits, unit modulus, blank NULL symbol)
                d_in = torch.zeros((B, 77 * 1536), dtype=torch.complex64, device=dev)
                for f0 in range(0, B, 2048):          # (in slices: the torch temporaries are 5x the slice)
                    f1 = min(B, f0 + 2048)
                    q = torch.randint(0, 4, (f1 - f0, 76 * 1536), device=dev)
                    ang = (q.float() * 2 + 1) * (np.pi / 4)
                    d_in[f0:f1, 1536:] = torch.polar(torch.ones_like(ang), ang)
                    del q, ang
            d_out = torch.empty((B, ns), dtype=torch.complex64 if fmt is None else torch.int32, device=dev)
            h = stream.cuda_stream
            # batches in flight have their own buffers, as the edges of a pipeline do: four, rotating
            ring = [(d_in, d_out)] + ([(d_in.clone(), torch.empty_like(d_out)) for _ in range(3)] if lanes else [])
            turn = [0]

            def step():
                if lanes:
                    bi, bo = ring[turn[0] & 3]
                    turn[0] += 1
                    md.chain_dev_queued(bi, B, stages, bo, from_bits=from_bits)
                elif from_bits:
                    md.chain_dev(d_in, B, stages, d_out, stream=h)
                else:
                    md.symbols_dev(d_in, B, stages, d_out, stream=h)

            for _ in range(warmup + PREWARM):     # W warm-up steps as asked, after a fixed pre-warm (clock ramp-up,
                step()                            # first-touch of the output pages); none of them is timed
            md.synchronize()
            stream.synchronize()
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)

            def timed_steps():
                # HIP events on the stream the kernels are launched on: per-launch kernel time
                e0.record(stream)
                if lanes:
                    md.wait_for_stream(h)         # the lanes start after e0 ...
                for _ in range(steps):
                    step()
                if lanes:
                    md.stream_wait_for(h)         # ... and e1 after every lane
                e1.record(stream)

            # barrier + synchronize | K steps | synchronize + barrier, MAX over ranks
            # (repeats > 1, secondary small-batch workloads only: the timed region is a few milliseconds long there, so it is
            #  run `repeats` times and the MEDIAN region is reported; the headline workload is timed once, exactly K steps)
            runs = []
            for _ in range(repeats):
                w_ = grp.timed(timed_steps, 1, torch.cuda.synchronize)
                runs.append((e0.elapsed_time(e1), w_))
            runs.sort()
            ev_ms, wall = runs[len(runs) // 2]
            if power_seconds > 0:
                # OUTSIDE the timed region: about power_seconds of the same launches queued at once, the host samples the
                # board's power and clock while the device works through them (verdict round 3: "the part is power-limited:
                # report W and J / frame beside the clock")
                try:
                    pw = sample_load(step, power_seconds, ev_ms / steps, stream, probe=probe0)
                    if pw and "error" in pw:
                        power_of[(workload, option, fmt, B)] = pw
                    elif pw:
                        fps = B / (ev_ms / steps * 1e-3)
                        pw["joules_per_frame"] = round(pw["watts_avg"] / fps, 6)
                        power_of[(workload, option, fmt, B)] = pw
                except Exception as ex:                     # (no hwmon on the box: the line simply has no power figure)
                    power_of[(workload, option, fmt, B)] = {"error": str(ex)[:120]}
            if args.gather > 0 and workload == args.workload:
                # the optional final IQ gather, OUTSIDE the timed region and reported on its own
                ng = min(B, args.gather)
                piece = d_out[:ng].contiguous()
                stream.synchronize()
                grp.gather_to_root(piece)                         # communicator warm-up
                torch.cuda.synchronize()
                grp.barrier()
                t0 = time.perf_counter()
                got = grp.gather_to_root(piece)
                torch.cuda.synchronize()
                grp.barrier()
                dt = grp.max_over_ranks(time.perf_counter() - t0)
                nbytes = piece.numel() * piece.element_size()
                gather_info.update({"frames_per_rank": ng, "bytes_per_rank": nbytes, "ms": round(dt * 1e3, 3),
                                    "GBps_into_rank0": round((world - 1) * nbytes / dt / 1e9, 2) if world > 1 else None,
                                    "ranks": world})
                del got, piece
        md.close()
        del d_out, d_in, d_bits, ring
        torch.cuda.empty_cache()
        return wall, ev_ms / steps

    def cfg4_parts(b):
        """The two kernels of cfg 4 one at a time -- the frame kernel into a native-rate buffer, the x4 resampler + predistorter
        out of it (dabgpu_post_process_dev) -- each with its time, board power and joules per frame: the chain is their sum, and
        the native-rate hand-over between them (src/DabModulator.cpp:403-406) is the frame kernel's stores plus the resampler's
        loads (profiles/r05_cfg4_energy.txt has the cache-resident A/B of both sides)."""
        md = P.Modulator(mode=1, device=local_rank, max_frames=b)
        md.set_gain(P.GAIN_VAR, 1.0, 1.0 / 50000.0, 4.0)
        md.set_resampler(2048000, 8192000)
        md.set_poly([1.0, 0.05, -0.01, 0.002, 0.0], [0.0, 0.02, 0.003, 0.0, 0.0])
        st = torch.cuda.Stream(device=dev)
        res = {}
        with torch.cuda.stream(st):
            bits = torch.randint(0, 256, (b, 28800), dtype=torch.uint8, device=dev)
            native = torch.empty((b, 196608), dtype=torch.complex64, device=dev)
            out = torch.empty((b, 4 * 196608), dtype=torch.complex64, device=dev)
            h = st.cuda_stream
            for name, step in (("frame_kernel", lambda: md.chain_dev(bits, b, P.STAGE_GAIN | P.STAGE_FIR, native, stream=h)),
                               ("resampler_poly", lambda: md.post_process_dev(native, P.STAGE_RESAMPLE | P.STAGE_POLY, out, stream=h))):
                for _ in range(3):
                    step()
                st.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(5):
                    step()
                e1.record(st)
                st.synchronize()
                ms = e0.elapsed_time(e1) / 5
                pw = sample_load(step, 2.0, ms, st, probe=probe0)
                res[name] = {"ms_per_launch": round(ms, 4), "frames_per_launch": b}
                if pw and "watts_avg" in pw:
                    res[name].update({"watts_avg": pw["watts_avg"], "sclk_MHz_avg": pw["sclk_MHz_avg"],
                                      "joules_per_frame": round(pw["watts_avg"] * ms * 1e-3 / b, 6)})
        md.close()
        del bits, native, out
        torch.cuda.empty_cache()
        return res

    def other_mode(mode, b, steps):
        """The cfg 3 chain (coded bits -> ... -> GainControl(var) -> guard -> FIRFilter) in transmission mode II / III / IV
        (src/DabModulator.cpp:84-122; N = 512 / 256 / 1024).  Round 6: kernels with the compile-time tap count -- Mode IV the
        equalised-boundary variant, modes II and III the packed dual transform with the boundary filter from a register window,
        Mode III two frames per wave; `kernel` is what the launch trace says ran.  Frames per second of THAT mode's frames
        (24 / 24 / 48 ms of air time), and its own algorithmic bytes per frame."""
        md = P.Modulator(mode=mode, device=local_rank, max_frames=b, chunks_per_frame=args.chunks)
        md.set_gain(P.GAIN_VAR, 1.0, 1.0 / 50000.0, 4.0)
        md.trace(True)
        stages = P.STAGE_GAIN | P.STAGE_FIR
        nin, ns = md.geometry["tf_input_bytes"], md.out_samples_per_frame(stages)
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            d_bits = torch.randint(0, 256, (b, nin), dtype=torch.uint8, device=dev)
            d_out = torch.empty((b, ns), dtype=torch.complex64, device=dev)
            for _ in range(3):
                md.chain_dev(d_bits, b, stages, d_out, stream=st.cuda_stream)
            st.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(steps):
                md.chain_dev(d_bits, b, stages, d_out, stream=st.cuda_stream)
            e1.record(st)
            st.synchronize()
        ms = e0.elapsed_time(e1) / steps
        ran = "; ".join(md.last_variant())
        md.close()
        del d_bits, d_out
        torch.cuda.empty_cache()
        algo_b = nin + 8 * ns
        fps = b / (ms * 1e-3)
        return {"frames_per_s": round(fps, 2), "frames_per_step": b, "algorithmic_bytes_per_frame": algo_b,
                "achieved_GBps": round(algo_b * fps / 1e9, 2), "roofline_frac": round(algo_b * fps / 1e9 / HBM_PEAK_GBPS, 4),
                "frame_ms_of_air_time": {2: 24, 3: 24, 4: 48}[mode], "kernel": ran}

    # The batch every rank can hold.  32768 frames are 52.5 GB of input + output per GPU: a rank that cannot get them
    # (another tenant on the GPU, a smaller part) halves its batch until the buffers fit, the ranks agree on the SMALLEST
    # such batch (weak scaling: the same work on every GPU), and the line says so -- rather than one rank dying of
    # hipErrorOutOfMemory and taking the job with it.
    def frames_that_fit(b):
        per_frame = 28800 + (4 if args.workload == "cfg4" else 1) * 196608 * 8 + \
            (77 * 1536 * 8 if args.workload in ("cfg2", "ifft_fir_stage") else 0) + \
            (196608 * 8 if args.workload == "cfg4" else 0)                       # (cfg 4: the native-rate scratch)
        while b > 64:
            try:
                probe = torch.empty(int(b * per_frame * 1.02) + (256 << 20), dtype=torch.uint8, device=dev)
                del probe
                break
            except RuntimeError:                   # torch.cuda.OutOfMemoryError is a RuntimeError
                b //= 2
            finally:
                torch.cuda.empty_cache()
        return grp.min_over_ranks(b)

    B = frames_that_fit(args.frames)
    frames_note = None
    if B != args.frames:
        frames_note = "--frames %d did not fit this GPU's free memory on every rank: %d frames per step per GPU" % (args.frames, B)
        if rank == 0:
            print("bench.py: " + frames_note, file=sys.stderr)
    gather_info = {}
    wall, kern_ms = run_workload(args.workload, B, args.steps, args.warmup,
                                 power_seconds=3.0 if (os.environ.get("WORLD_SIZE") is None and not args.no_extra) else 0.0)
    value = grp.job_frames_per_second(B, args.steps, wall)
    algo = ALGO_BYTES[args.workload]
    achieved = algo * B / (kern_ms * 1e-3) / 1e9  # GB/s per GPU, per-launch HIP-event time

    # Profiler counters.  At N = 1 the bench collects them ITSELF for the headline workload (--counters live, the default):
    # three rocprofv3 --pmc passes of the same launch (tools/prof_run.py: same workload, same batch) right after the timed
    # region -- SQ + GRBM counters, FETCH_SIZE, WRITE_SIZE in separate passes, no tracing beside --pmc -- each under a
    # timeout.  If that is not possible (no rocprofv3, a pass fails or times out, N > 1) the figures are REPLAYED from
    # profiles/traffic.json -- the same passes collected by tools/profile_all.sh -- and only while the device sources are
    # the ones those were collected on (source hash); `counters_source` says which.  Formulas: tools/counter_math.py.
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import counter_math
    traffic, busy, replay = None, {}, None
    packed_fraction = 0.0
    try:
        mj = json.load(open(os.path.join(ROOT, "profiles", "isa_mix.json")))
        if mj.get("source_hash") == P.source_hash():
            packed_fraction = mj.get(args.workload, {}).get("packed_fraction_of_valu", 0.0)
    except Exception:
        mj = {}
    KEEP = ("valu_busy", "lds_busy", "hbm_frac", "packed_fraction_of_valu", "wave_active_frac", "wave_issue_stall_frac",
            "wave_issue_stall_lds_frac", "wave_parked_frac", "lds_bank_conflict_share", "traffic_over_algorithmic")

    def live_counters(workload=None, frames=None):
        workload = workload or args.workload
        frames = frames or B
        packed = packed_fraction if workload == args.workload else mj.get(workload, {}).get("packed_fraction_of_valu", 0.0) \
            if mj.get("source_hash") == P.source_hash() else 0.0
        import glob
        import shutil
        import subprocess
        import tempfile
        if shutil.which("rocprofv3") is None:
            return None, "rocprofv3 not found"
        top = tempfile.mkdtemp(prefix="dabgpu_pmc_")
        passes = ["SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY "
                  "SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT FETCH_SIZE", "WRITE_SIZE"]
        try:
            dbs = []
            for i, pmc in enumerate(passes):
                out = os.path.join(top, "p%d" % i)
                cmd = ["rocprofv3", "--pmc"] + pmc.split() + ["-d", out, "-o", "pmc", "--", sys.executable,
                       os.path.join(ROOT, "tools", "prof_run.py"), workload, str(frames), "3"]
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=150)
                found = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
                if r.returncode != 0 or not found:
                    return None, "rocprofv3 pass %d failed (rc %d)" % (i, r.returncode)
                dbs += found
            blocks = counter_math.read_rocpd(dbs)
            if not blocks:
                return None, "no kernel of the workload in the rocprofv3 output"
            return counter_math.figures(blocks, ALGO_BYTES[workload] * frames, packed, "resampler" if "resampler" in blocks else "tf_kernel"), None
        except Exception as ex:                               # (timeout, unreadable output, ...)
            return None, "%s: %s" % (type(ex).__name__, str(ex)[:120])
        finally:
            shutil.rmtree(top, ignore_errors=True)

    t = None
    if args.counters == "live" and rank == 0 and world == 1:
        torch.cuda.empty_cache()
        t, why = live_counters()
        replay = ("collected in this run: rocprofv3 --pmc, 3 passes of tools/prof_run.py %s %d" % (args.workload, B)) if t \
            else "live collection failed (%s); " % why
    if t is None and args.counters != "off":
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        try:
            tj = json.load(open(tp))
            tt = tj.get(args.workload)
            if tj.get("source_hash") != P.source_hash():
                replay = (replay or "") + "replay dropped: profiles/traffic.json was collected on other device sources (%s)" % tj.get("source_hash")
            elif tt and tt.get("frames") == B:
                t = tt
                replay = (replay or "") + "replayed from profiles/traffic.json (%s, source hash %s)" % (tj.get("profile_dir"), tj.get("source_hash"))
            else:
                replay = (replay or "") + "replay dropped: no counters for %d frames per launch" % B
        except Exception:
            replay = (replay or "") + "replay dropped: unreadable profiles/traffic.json"
    if t:
        traffic = t.get("hbm_bytes_per_launch")
        busy = {k: t[k] for k in KEEP if k in t}
        if t.get("gpu_cycles_per_launch_profiled"):
            # the part is power-limited under this kernel: GRBM_GUI_ACTIVE / 8 XCDs over the launch time is the clock it
            # actually ran at (2.4 GHz nominal)
            busy["effective_clock_GHz"] = t.get("effective_clock_GHz_profiled") or round(t["gpu_cycles_per_launch_profiled"] / (kern_ms * 1e6), 2)

    # What limits the kernel, from those numbers (never a fixed string): a unit at >= 80 % is the limiter; otherwise no
    # unit is saturated and the line says where the waves' time goes (issuing, stalled at issue, parked) -- measured shares.
    def limiter_of(b, pw=None):
        if pw and pw.get("watts_cap") and pw.get("watts_avg", 0.0) >= 0.95 * pw["watts_cap"]:
            # measured by this run: the board sits at its power limit, the clock is what the limit leaves
            return ("board power: %.0f W of the %.0f W limit, shader clock %.2f of 2.40 GHz (%.3g mJ per frame); inside that "
                    "budget -- %s" % (pw["watts_avg"], pw["watts_cap"], pw.get("sclk_MHz_avg", 0.0) / 1e3,
                                      1e3 * pw.get("joules_per_frame", 0.0), limiter_of(b)))
        if not b:
            return "not profiled for this build"
        units = {"valu": b.get("valu_busy", 0.0), "lds": b.get("lds_busy", 0.0), "hbm": b.get("hbm_frac", 0.0)}
        top_unit = max(units, key=units.get)
        if units[top_unit] >= 0.8:
            return "%s (%.0f %% busy)" % (top_unit, 100 * units[top_unit])
        return ("no unit saturated: valu %.0f %% busy, lds %.0f %%, hbm %.0f %%; the waves spend %.0f %% of their life issuing, %.0f %% "
                "stalled at issue (%.0f %% on the LDS path), %.0f %% parked at s_waitcnt / s_barrier -- latency the resident waves do "
                "not cover (measured counters only)"
                % (100 * units["valu"], 100 * units["lds"], 100 * units["hbm"], 100 * b.get("wave_active_frac", 0.0),
                   100 * b.get("wave_issue_stall_frac", 0.0), 100 * b.get("wave_issue_stall_lds_frac", 0.0),
                   100 * b.get("wave_parked_frac", 0.0)))

    # (Rounds 2 - 5 carried a static issue-time model of the hot loop here -- the sum of per-instruction issue costs from a
    # microbenchmark against the measured SIMD cycles per symbol.  It over-predicted by 1.63x (it adds VALU and LDS issue
    # times that in fact overlap across the waves of a SIMD), so it explained nothing and is retired: `limiter` below is a
    # statement of the measured counters only.  profiles/isa_mix.json keeps the static instruction counts.)
    measured_cycles = None
    try:
        if t and t.get("dominant_kernel_cycles_profiled"):
            per_frame = 96 * 4 if args.workload == "cfg4" else 77 * 4      # wave-iterations per frame (hops / symbols x 4 waves)
            measured_cycles = round(t["dominant_kernel_cycles_profiled"] * 1024 / (B * per_frame), 1)
    except Exception:
        measured_cycles = None

    # this box's own ceilings next to the nominal 8 TB/s: a fill (write only) and a copy (read + write) over 4 GiB
    def measured_peaks():
        n = 1 << 30
        a = torch.empty(n, dtype=torch.float32, device=dev)
        b = torch.empty(n, dtype=torch.float32, device=dev)
        res = {}
        for name, fn, nbytes in (("fill_GBps", lambda: a.fill_(1.0), 4 * n), ("copy_GBps", lambda: b.copy_(a), 8 * n)):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[name] = round(nbytes * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        del a, b
        torch.cuda.empty_cache()
        return res

    line = {
        "metric": "Mode-I TX frames/sec (196608 IQ/frame)",
        "value": round(value, 2),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": {"cfg2": "Mode I OfdmGenerator+GuardIntervalInserter (BASELINE config 2)",
                                "ifft_fir_stage": "Mode I OfdmGenerator + GainControl(var) + GuardIntervalInserter + "
                                                  "FIRFilter(45 default taps) from SignalMultiplexer output",
                                "cfg3": "Mode I full chain from coded bits + GainControl(var) + "
                                        "FIRFilter(45 default taps), native 2.048 Msps (BASELINE config 3)",
                                "cfg4": "Mode I cfg3 + Resampler 2.048->8.192 Msps + MemlessPoly "
                                        "(BASELINE config 4)"}[args.workload],
                   "frames_per_step_per_gpu": B, "mode": 1, "prewarm_steps": PREWARM, "parallelism": "%d independent streams" % world,
                   "realtime_multiple": round(value / 10.4167, 1), **({"frames_note": frames_note} if frames_note else {}),
                   "residency": "input and output device-resident (no PCIe in the timed region); the host entry points "
                                "are PCIe-bound, see DESIGN.md section 6"},
        # "bound": the roofline the fraction is priced against (HBM).  "limiter": what the counters say
        # actually limits the kernel (limiter_of above).
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                     "kernel": "tf_kernel" if args.workload != "cfg4" else "tf_kernel+resampler_kernel",
                     "algorithmic_bytes_per_frame": algo, "kernel_ms_per_launch": round(kern_ms, 4), **busy,
                     "valu_frac_of_peak": round(EXEC_FLOPS[args.workload] * value / world / 1e12 / VALU_PEAK_TFLOPS, 4)
                     if args.workload in EXEC_FLOPS else None,
                     "power": power_of.get((args.workload, None, None, B)),
                     "limiter": limiter_of(busy, power_of.get((args.workload, None, None, B))),
                     "simd_cycles_per_wave_iteration": measured_cycles,
                     "counters_source": replay},
    }
    if rank == 0 and world == 1 and not args.no_extra:
        try:
            line["roofline"]["peak_measured_GBps"] = measured_peaks()
        except Exception as ex:
            line["roofline"]["peak_measured_GBps"] = {"error": str(ex)[:120]}

    if rank == 0 and world == 1:
        if not args.no_extra:
            extra = {}
            # the other BASELINE configs, and the headline workload one frame at a time (B = 1:
            # what a single real-time stream sees; the frame is split over 11 workgroups)
            bs = min(B, 16384)     # (carrier inputs: 0.95 MB per frame on top of the 1.57 MB of output)
            # (batch B in {1, 16, 256} next to the best; "_s16": FormatConverter fused into the last store)
            for wl, b2 in (("cfg2", bs), ("ifft_fir_stage", bs), ("cfg4", max(64, bs // 4)),
                           (args.workload + "_B1", 1), (args.workload + "_B16", 16), (args.workload + "_B256", 256),
                           (args.workload + "_B1_one_lane", 1), (args.workload + "_B16_one_lane", 16),
                           (args.workload + "_B256_one_lane", 256),
                           ("cfg3_s16", B), ("cfg4_s16", max(64, bs // 4)),
                           ("cfg3_cfr", max(64, bs // 2)), ("cfg3_cfr_s16", max(64, bs // 2)),
                           ("cfg3_window", max(64, bs // 2)), ("cfg3_nofir", B)):
                if wl == args.workload:
                    continue
                try:
                    parts = wl.split("_B")[0].split("_")           # e.g. cfg3_cfr_s16 / ifft_fir_stage / cfg3_B16_one_lane
                    option = next((p for p in parts[1:] if p in ("cfr", "window", "nofir")), None)
                    base = "_".join(p for p in parts if p not in ("cfr", "window", "nofir", "s16"))
                    # (small batches: regions of 30 ... 80 ms -- a region of a few milliseconds right after a synchronisation
                    #  is spent in the clock's ramp from idle and under-reports by a third)
                    k = max(3, args.steps // 4) if b2 > 256 else (3000 if b2 <= 16 else 800)
                    # B = 1 / 16 / 256: ONE context, its own stream, the library's lanes (SMALL_BATCH_LANES); "_one_lane":
                    # the same calls in order on one stream (what rounds 1-4 reported under these names)
                    small = "_B" in wl
                    nl = 0 if not small else (1 if wl.endswith("_one_lane") else SMALL_BATCH_LANES)
                    w2, k2 = run_workload(base, b2, k, k if small else 1, fmt="s16" if wl.endswith("_s16") else None,
                                          option=option, power_seconds=3.0 if wl in ("cfg4", "cfg2", "cfg3_nofir") else 0.0,
                                          lanes=nl, repeats=5 if small else 1)
                    algo2 = ALGO_BYTES[base] if not wl.endswith("_s16") else \
                        28800 + (ALGO_BYTES[base] - 28800) // 2                 # 4 bytes per sample written
                    gbps = algo2 * b2 / (k2 * 1e-3) / 1e9
                    extra[wl] = {"frames_per_s": round(b2 * k / w2, 2), "frames_per_step": b2,
                                 "achieved_GBps": round(gbps, 2), "roofline_frac": round(gbps / HBM_PEAK_GBPS, 4)}
                    if small:
                        extra[wl].update({"contexts": 1, "lanes": nl, "us_per_call": round(k2 * 1e3, 2),
                                          "timing": "HIP events on the caller's stream around %d calls on the context's own "
                                                    "stream, ordered by dabgpu_wait_for_stream / dabgpu_stream_wait_for; "
                                                    "median of 5 such regions after %d warm-up calls" % (k, k + PREWARM)})
                    pw2 = power_of.get((base, option, "s16" if wl.endswith("_s16") else None, b2)) \
                        if wl in ("cfg4", "cfg2", "cfg3_nofir") else None
                    if pw2:
                        extra[wl]["power"] = pw2
                    if wl == "cfg4":
                        # "report fp32 VALU utilisation alongside" -- cfg 4 is bound by instruction issue, not by
                        # HBM: the executed fp32 operations against the vector peak, and the same counters as the headline
                        # kernel's (three more rocprofv3 --pmc passes, same launch shape)
                        tf_s = b2 * k / w2
                        extra[wl].update({"exec_flops_per_frame": EXEC_FLOPS["cfg4"],
                                          "valu_TFLOPs": round(EXEC_FLOPS["cfg4"] * tf_s / 1e12, 2),
                                          "valu_frac_of_peak": round(EXEC_FLOPS["cfg4"] * tf_s / 1e12 / VALU_PEAK_TFLOPS, 4),
                                          "valu_peak_TFLOPs": VALU_PEAK_TFLOPS})
                        try:
                            extra[wl]["parts"] = cfg4_parts(b2)
                        except Exception as ex:
                            extra[wl]["parts"] = {"error": str(ex)[:200]}
                        if args.counters == "live":
                            torch.cuda.empty_cache()
                            t4, why4 = live_counters("cfg4", b2)
                            if t4:
                                extra[wl].update({k4: t4[k4] for k4 in KEEP if k4 in t4})
                                extra[wl]["traffic"] = t4.get("hbm_bytes_per_launch")
                                extra[wl]["effective_clock_GHz"] = t4.get("effective_clock_GHz_profiled")
                                extra[wl]["limiter"] = limiter_of({k4: t4[k4] for k4 in KEEP if k4 in t4}, pw2)
                                extra[wl]["counters_source"] = "collected in this run: rocprofv3 --pmc, 3 passes of tools/prof_run.py cfg4 %d" % b2
                            else:
                                extra[wl]["counters_source"] = "live collection failed (%s)" % why4
                except Exception as ex:  # secondary numbers must never break the contract line
                    extra[wl] = {"error": str(ex)[:200]}
            if args.workload == "cfg3":
                for mode in (2, 3, 4):
                    try:
                        # (the same 51.5 GB of IQ per step as the Mode I headline: frames are a quarter / a half as long)
                        extra["cfg3_mode%d" % mode] = other_mode(mode, min(B, 16384) * (2 if mode == 4 else 4), max(3, args.steps // 4))
                    except Exception as ex:
                        extra["cfg3_mode%d" % mode] = {"error": str(ex)[:200]}
            try:
                extra["host_tools"] = host_tools(device_index)
            except Exception as ex:
                extra["host_tools"] = {"error": str(ex)[:200]}
            line["other_workloads"] = extra
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload)
    if gather_info:
        line["iq_gather"] = gather_info
    if rank == 0:
        grp.emit(json.dumps(line))
    grp.close()


if __name__ == "__main__":
    main()
