#!/usr/bin/env python3
"""End-to-end rate of the file-to-file tool (ETI file -> CPU front-end -> fused chain -> output), frame by frame and
batched / pipelined.  The output goes to /dev/null: this times the path, not a disk.  usage (GPU box): python tools/time_dabmod_file.py"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.golden.synth import synth_eti
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "odr-dabmod_amd", "host"), "-j4"])
tool = os.path.join(ROOT, "odr-dabmod_amd", "host", "dabmod_file")
fin = "/tmp/time_dabmod.eti"
synth_eti(8000).tofile(fin)                                   # 2000 Mode-I transmission frames, looped 10 times
for args in (["--bits-only"], [], ["--batch", "8"], ["--batch", "32"], ["--format", "s16"], ["--format", "s16", "--batch", "8"],
             ["--format", "s16", "--batch", "32"], ["--format", "s16", "--batch", "32", "--rate", "8192000"]):
    base = [] if args == ["--bits-only"] else ["--fir", "default"]
    t0 = time.perf_counter()
    r = subprocess.run([tool, fin, "/dev/null", "--loop", "10"] + base + args, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    print("%-50s %7.0f TF/s (%.2f s)%s" % (" ".join(base + args) or "(frame by frame, complexf)", 20000 / dt, dt,
                                           "" if r.returncode == 0 else "  FAILED: " + r.stderr[-200:]), flush=True)
