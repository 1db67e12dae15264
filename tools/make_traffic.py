#!/usr/bin/env python3
"""profiles/traffic.json from a tools/profile_all.sh output directory: per bench workload the HBM bytes per launch
(rocprofv3 PMC passes) and the VALU / LDS utilisation of the same runs, stamped with the hash of the device sources
the counters were collected on (bench.py drops the replayed fields when the library has changed since).
usage: make_traffic.py <gpurun_out/prof_TAG> [workload=frames ...]
WRITE_SIZE / FETCH_SIZE are in KiB per dispatch; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950
(MI355X_MICROARCH.md, HBM section), hence the factor 2.  With two kernels in a launch (cfg 4) their counters add."""
import importlib, json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
P = importlib.import_module("odr-dabmod_amd")
ALGO = {"cfg2": 946176 + 1572864, "cfg3": 28800 + 1572864, "ifft_fir_stage": 946176 + 1572864, "cfg4": 28800 + 6291456}
top = sys.argv[1]
frames = dict(a.split("=") for a in sys.argv[2:])
path = os.path.join(root, "profiles", "traffic.json")
d = {"_about": ("HBM bytes per launch from rocprofv3 PMC passes (tools/profile_all.sh): WRITE_SIZE*1024 + 2*FETCH_SIZE*1024 "
                "(FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950, MI355X_MICROARCH.md 'HBM'); valu_busy = "
                "SQ_INSTS_VALU * 4 cycles / (1024 SIMDs * GRBM_GUI_ACTIVE / 8), lds_busy = SQ_LDS_IDX_ACTIVE / (256 CUs * "
                "GRBM_GUI_ACTIVE / 8); summed over the kernels of a launch"),
     "source_hash": P.source_hash(), "profile_dir": os.path.basename(top.rstrip("/"))}
for wl in ("cfg3", "cfg2", "ifft_fir_stage", "cfg4"):
    f = os.path.join(top, wl, "summary.txt")
    if not os.path.exists(f):
        continue
    txt = open(f).read()
    def tot(name):
        # one value per kernel block of the summary: sum over the chain's kernels
        return sum(float(v) for v in re.findall(r"^\s+" + name + r"\s+([0-9.e+]+)", txt, re.M))
    w, fe = tot("WRITE_SIZE"), tot("FETCH_SIZE")
    n = int(frames.get(wl, 0))
    e = {"frames": n, "hbm_bytes_per_launch": int(w * 1024 + 2 * fe * 1024), "write_size_kb": w, "fetch_size_kb": fe,
         "algorithmic_bytes_per_launch": ALGO[wl] * n}
    if n:
        e["traffic_over_algorithmic"] = round(e["hbm_bytes_per_launch"] / e["algorithmic_bytes_per_launch"], 4)
    gui, valu, lds = tot("GRBM_GUI_ACTIVE"), tot("SQ_INSTS_VALU"), tot("SQ_LDS_IDX_ACTIVE")
    if gui and valu:
        cyc = gui / 8
        e.update({"valu_busy": round(valu * 4 / (1024 * cyc), 3), "lds_busy": round(lds / (256 * cyc), 3),
                  "valu_insts_per_launch": valu, "gpu_cycles_per_launch_profiled": cyc})
    d[wl] = e
json.dump(d, open(path, "w"), indent=1)
print(json.dumps(d, indent=1))
