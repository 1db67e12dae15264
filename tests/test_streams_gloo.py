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
    print(json.dumps({"rank": g.rank, "elapsed": el, "fps": fps, "seed": g.stream_seed(), "calls": len(calls)}))
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
