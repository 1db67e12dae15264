"""The N > 1 path on CPU: two processes over gloo exercise the harness bench.py
uses on N GPUs (barrier-bracketed timing, MAX over ranks, per-rank streams,
whole-job throughput).  There is no data-path collective to test: frames are
independent units."""
import os
import socket
import subprocess
import sys
import textwrap

from tests.conftest import ROOT

WORKER = textwrap.dedent("""
    import importlib, json, os, sys, time
    sys.path.insert(0, %r)
    streams = importlib.import_module("odr-dabmod_amd.streams")
    g = streams.StreamGroup(backend="gloo")
    assert g.world == 2 and g.backend == "gloo"
    delay = 0.05 if g.rank == 0 else 0.15          # rank 1 is the slow one
    calls = []
    el = g.timed(lambda: (calls.append(1), time.sleep(delay)), steps=3, sync=lambda: None)
    fps = g.job_frames_per_second(frames_per_step_per_gpu=100, steps=3, seconds=el)
    fit = g.min_over_ranks(32768 if g.rank == 0 else 8192)     # rank 1's GPU holds a quarter of the batch: everybody takes 8192
    g.emit(json.dumps({"rank": g.rank, "elapsed": el, "fps": fps, "seed": g.stream_seed(), "calls": len(calls), "fit": fit}))
    g.close()
""") % ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_harness_over_gloo(tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    # both ranks report the SLOWEST rank's time (3 x 0.15 s), not their own
    assert abs(outs[0]["elapsed"] - outs[1]["elapsed"]) < 1e-9
    assert 0.44 < outs[0]["elapsed"] < 1.5
    # whole-job value: 2 ranks x 100 frames x 3 steps / elapsed
    assert abs(outs[0]["fps"] - 600 / outs[0]["elapsed"]) < 1e-6
    assert outs[0]["seed"] != outs[1]["seed"]
    assert outs[0]["calls"] == outs[1]["calls"] == 3
    assert outs[0]["fit"] == outs[1]["fit"] == 8192


WORKER2 = textwrap.dedent("""
    import importlib, json, os, sys, hashlib
    sys.path.insert(0, %r)
    import numpy as np, torch
    streams = importlib.import_module("odr-dabmod_amd.streams")
    fe = importlib.import_module("odr-dabmod_amd.frontend")
    from tests.golden.synth import synth_eti
    g = streams.StreamGroup(backend="gloo")
    # one modulator-shaped worker per rank, CPU half: its OWN ETI stream (seeded by the rank) through its own
    # front-end instance -> the hot path's coded-bits input (on a GPU rank this is what chain_dev consumes)
    eti = synth_eti(16, seed=g.stream_seed(1234))
    bits = fe.Frontend().eti_to_bits(eti, 1)
    assert bits.shape == (4, 28800)
    mine = hashlib.sha256(bits.tobytes()).hexdigest()
    # the optional final gather (here: of the coded bits, standing in for the IQ) lands on rank 0 in rank order
    got = g.gather_to_root(torch.from_numpy(bits.copy()))
    res = {"rank": g.rank, "sha": mine, "seed": g.stream_seed(1234)}
    if g.rank == 0:
        assert len(got) == 2
        res["gathered"] = [hashlib.sha256(t.numpy().tobytes()).hexdigest() for t in got]
    else:
        assert got is None
    el = g.timed(lambda: fe.Frontend().eti_to_bits(eti, 1), steps=2, sync=lambda: None)
    res["fps"] = g.job_frames_per_second(4, 2, el)
    g.emit(json.dumps(res))
    g.close()
""") % ROOT


def test_two_ranks_modulate_their_own_streams_and_gather(tmp_path):
    """World size 2 over gloo with real per-rank workers: each rank builds its own front-end on its own seeded ETI
    stream (distinct input, as on N GPUs), the harness times them, and the optional gather delivers both ranks'
    frames to rank 0 in order."""
    import json
    script = tmp_path / "worker2.py"
    script.write_text(WORKER2)
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert outs[0]["seed"] != outs[1]["seed"] and outs[0]["sha"] != outs[1]["sha"]      # two different streams
    assert outs[0]["gathered"] == [outs[0]["sha"], outs[1]["sha"]]                      # rank order on the root
    assert abs(outs[0]["fps"] - outs[1]["fps"]) < 1e-6 and outs[0]["fps"] > 0


def test_bench_launches_n_ranks_itself_dry_run():
    """`python bench.py --gpus 2` outside torch.distributed.run is its own launcher: two ranks over gloo, ONE JSON
    line (rank 0's) with n_gpus = 2 and the whole-job value (--dry-run: a sleep stands in for the kernels)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3",
                        "--warmup", "1", "--frames", "100"], env=env, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    # rank 1 sleeps 20 ms per step, rank 0 10 ms: the job's time is the slow rank's, the value both ranks' frames
    assert 0.02 * 3 <= d["ms_per_step"] * 3e-3 < 0.5
    assert abs(d["value"] - 2 * 100 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-3
    assert d["config"]["devices"] == [0, 1]


def _ranks_of_launcher(pid):
    """Live processes that a bench.py launcher with this pid started (it marks its ranks' environment)."""
    import psutil
    found = []
    for q in psutil.process_iter(["pid"]):
        try:
            if q.environ().get("DABGPU_BENCH_PARENT") == str(pid) and q.status() != psutil.STATUS_ZOMBIE:
                found.append(q.pid)
        except (psutil.Error, OSError):
            pass
    return found


def test_bench_launches_eight_ranks_dry_run():
    """The shape of the driver's largest scan point: `python bench.py --gpus 8` as its own launcher -- eight children, one
    rendezvous port, ONE line (n_gpus = 8, value = eight ranks' frames over the slowest rank's time), every child gone
    afterwards.  (--dry-run: the launcher, the gloo process group and the timing harness without a kernel; rank r sleeps
    10 (1 + r) ms per step.)"""
    import json
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    t0 = time.time()
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "2",
                          "--warmup", "1", "--frames", "32768"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    out, err = p.communicate(timeout=600)
    assert p.returncode == 0, err[-2000:]
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["config"]["devices"] == list(range(8)) and d["config"]["frames_per_step_per_gpu"] == 32768
    # the slowest rank (rank 7) sleeps 80 ms per step
    assert 0.08 * 2 <= d["ms_per_step"] * 2e-3 < 2.0
    assert abs(d["value"] - 8 * 32768 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-3
    # teardown: no child of ours is left (the launcher reaps all eight before it returns)
    assert not _ranks_of_launcher(p.pid)
    assert time.time() - t0 < 300


def test_bench_launcher_ends_the_other_ranks_when_one_fails():
    """A rank that dies must not leave seven others waiting in a barrier: the launcher terminates them and returns the
    failing status (DABGPU_BENCH_FAIL_RANK makes one dry-run rank exit before the rendezvous)."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["DABGPU_BENCH_FAIL_RANK"] = "3"
    t0 = time.time()
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run", "--steps", "2"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    out, err = p.communicate(timeout=300)
    assert p.returncode != 0 and not out.strip()
    assert time.time() - t0 < 120
    assert not _ranks_of_launcher(p.pid)


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr and not r.stdout.strip()


def test_bench_one_rank_dry_run_needs_no_process_group():
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "2"], env=env,
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip())["n_gpus"] == 1


def test_bench_under_torch_distributed_run_as_the_driver_launches_it():
    """The driver's N > 1 command: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ... -- every process is one rank (no second launcher inside), rank 0 prints the one line."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run",
                        "--steps", "2", "--warmup", "1", "--frames", "50"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["config"]["frames_per_step_per_gpu"] == 50
