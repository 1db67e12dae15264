#!/usr/bin/env python3
"""Why is submit/collect bimodal from process to process (s16, 32 frames per batch: 0.46 or 0.86 ms per batch on one box,
profiles/r06_hostpath_bisect.txt)?  Runs the loop in fresh processes, then under rocprofv3 --memory-copy-trace.
usage (GPU box): python tools/experiments/exp_r06_async_modes.py [child FMT B]"""
import importlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(fmt, B):
    import numpy as np
    P = importlib.import_module("odr-dabmod_amd")
    fmt = None if fmt == "complexf" else fmt
    md = P.Modulator(mode=1, max_frames=B)
    md.set_gain(2, 1.0, 1 / 50000. if fmt is None else 1.0, 4.0)
    md.set_output_format(fmt)
    bits = np.frombuffer(np.random.RandomState(1).bytes(B * 28800), np.uint8).reshape(B, 28800)
    md.submit(bits, 3)
    for _ in range(4):
        md.submit(bits, 3); md.collect(copy=False)
    n = max(8, 512 // B)
    rates = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            md.submit(bits, 3)
            md.collect(copy=False)
        dt = time.perf_counter() - t0
        rates.append(B * n / dt)
    md.collect(copy=False)
    md.close()
    print("%s B=%d  %s frames/s" % (fmt or "complexf", B, " ".join("%.0f" % r for r in rates)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    me = os.path.abspath(__file__)
    for fmt, B in (("s16", 32), ("complexf", 32), ("s16", 8), ("complexf", 1)):
        for i in range(6):
            subprocess.run([sys.executable, me, "child", fmt, str(B)])
    for env_name, env_val in (("HSA_ENABLE_SDMA", "0"),):
        e = dict(os.environ); e[env_name] = env_val
        print("with %s=%s:" % (env_name, env_val), flush=True)
        for i in range(3):
            subprocess.run([sys.executable, me, "child", "s16", "32"], env=e)
            subprocess.run([sys.executable, me, "child", "complexf", "32"], env=e)
    out = os.path.join(ROOT, "gpurun_out", "r06_copytrace")
    os.makedirs(out, exist_ok=True)
    for i in range(4):
        d = os.path.join(out, "run%d" % i)
        r = subprocess.run(["rocprofv3", "--memory-copy-trace", "--kernel-trace", "-d", d, "-o", "t", "--output-format", "csv", "--",
                            sys.executable, me, "child", "s16", "32"], capture_output=True, text=True)
        print("traced run %d: %s" % (i, (r.stdout.strip().splitlines() or ["?"])[-1]), flush=True)
