"""bench.py on the GPU box: the contract line, and the distributed path with a real RCCL communicator on one rank
(DABGPU_FORCE_DIST=1) -- what the driver's N = 2, 4, 8 launches initialise, exercised where only one GPU is leased."""
import json
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(extra_args, extra_env, timeout=600):
    env = dict(os.environ, **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--frames", "64", "--no-extra", "--no-cpu-baseline"] + extra_args,
                       env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                       # ONE JSON line on stdout, nothing else (RCCL's banner goes to stderr)
    return json.loads(lines[0])


def test_bench_line_single_process():
    d = run_bench([], {})
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "frames/s" and d["value"] > 0
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    assert d["config"]["frames_per_step_per_gpu"] == 64


def test_bench_with_a_real_rccl_communicator_on_one_rank():
    """torch.distributed over RCCL initialised for WORLD_SIZE = 1: barrier, all_reduce(MAX) and the optional IQ
    gather run through the same code as on N ranks."""
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = {"DABGPU_FORCE_DIST": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
           "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    d = run_bench(["--gather", "8"], env)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "weak"
    g = d["iq_gather"]
    assert g["frames_per_rank"] == 8 and g["bytes_per_rank"] == 8 * 196608 * 8 and g["ranks"] == 1 and g["ms"] > 0
