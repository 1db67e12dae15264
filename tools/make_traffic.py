#!/usr/bin/env python3
"""profiles/traffic.json from a tools/profile_all.sh output directory: per bench workload the HBM bytes per launch
(rocprofv3 PMC passes) and the utilisation of the same runs, stamped with the hash of the device sources the counters were
collected on (bench.py, when it cannot collect counters itself, replays these -- and drops them when the library has
changed since).  The formulas, units and gfx950 corrections: tools/counter_math.py.
usage: make_traffic.py <gpurun_out/prof_TAG> [workload=frames ...]"""
import importlib, json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
P = importlib.import_module("odr-dabmod_amd")
ALGO = {"cfg2": 946176 + 1572864, "cfg3": 28800 + 1572864, "ifft_fir_stage": 946176 + 1572864, "cfg4": 28800 + 6291456}
sys.path.insert(0, os.path.join(root, "tools"))
import counter_math
top = sys.argv[1]
frames = dict(a.split("=") for a in sys.argv[2:])
path = os.path.join(root, "profiles", "traffic.json")
mix_path = os.path.join(root, "profiles", "isa_mix.json")
mix = json.load(open(mix_path)) if os.path.exists(mix_path) else {}
if mix.get("source_hash") != P.source_hash():
    sys.exit("profiles/isa_mix.json belongs to other device sources: run tools/isa_mix.py --json profiles/isa_mix.json first")
d = {"_about": counter_math.__doc__.split("Units and corrections")[1].strip(),
     "source_hash": P.source_hash(), "profile_dir": os.path.basename(top.rstrip("/"))}
for wl in ("cfg3", "cfg2", "ifft_fir_stage", "cfg4"):
    f = os.path.join(top, wl, "summary.txt")
    if not os.path.exists(f):
        continue
    txt = open(f).read()
    # per kernel block of the PMC part of the summary: {counter: value}
    blocks = {}
    cur = None
    for line in txt.split("== PMC")[-1].splitlines():
        m = re.match(r"^  (\S.*?)\s+vgpr=", line)
        if m:
            name = "resampler" if "resampler" in m.group(1) else ("tf_kernel" if "tf_kernel" in m.group(1) else m.group(1)[:40])
            cur = blocks.setdefault(name, {})
            continue
        m = re.match(r"^\s+([A-Z_a-z0-9]+)\s+([0-9.e+]+)", line)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(2))
    n = int(frames.get(wl, 0))
    e = {"frames": n}
    e.update(counter_math.figures(blocks, ALGO[wl] * n, mix.get(wl, {}).get("packed_fraction_of_valu", 0.0),
                                  "resampler" if "resampler" in blocks else "tf_kernel"))
    d[wl] = e
json.dump(d, open(path, "w"), indent=1)
print(json.dumps(d, indent=1))
