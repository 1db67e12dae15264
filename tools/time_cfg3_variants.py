#!/usr/bin/env python3
"""cfg 3 at the bench's batch for every library in tools/_variants (tuning aid, run on the GPU box).
usage: time_cfg3_variants.py [B]"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = sys.argv[1] if len(sys.argv) > 1 else "16384"
for lib in sorted(glob.glob(os.path.join(ROOT, "tools", "_variants", "*.so"))):
    env = dict(os.environ, DABGPU_LIB=lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_b.py"), "3", B + ",1"], env=env, capture_output=True, text=True)
    print(os.path.basename(lib), (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
