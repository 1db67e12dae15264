#!/usr/bin/env python3
"""profiles/traffic.json from a tools/profile_all.sh output directory: per bench workload the HBM bytes per launch
(rocprofv3 PMC passes) and the utilisation of the same runs, stamped with the hash of the device sources the counters were
collected on (bench.py drops the replayed fields when the library has changed since).
usage: make_traffic.py <gpurun_out/prof_TAG> [workload=frames ...]

Units and corrections (MI355X_MICROARCH.md):
  * WRITE_SIZE / FETCH_SIZE are KiB per dispatch; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 ('HBM'
    section), hence the factor 2.  With two kernels in a launch (cfg 4) their counters add.
  * A CU has four SIMD-32s: a plain wave64 VALU instruction occupies its SIMD for 2 cycles, a packed-fp32 one
    (v_pk_*_f32: two passes) for 4.  The packed share of a kernel's VALU instructions comes from the static mix of its
    hot loop (tools/isa_mix.py -> profiles/isa_mix.json):
        valu_busy = SQ_INSTS_VALU * (2 + 2 * packed_fraction) / (1024 SIMDs * cycles),  cycles = GRBM_GUI_ACTIVE / 8 XCDs
  * lds_busy = SQ_LDS_IDX_ACTIVE / (256 CUs * cycles) (LDS-array cycles, bank conflicts included).
  * SQ_WAVE_CYCLES = SQ_ACTIVE_INST_ANY + SQ_WAIT_INST_ANY + SQ_WAIT_ANY (disjoint, quad-cycles): the share of its life a
    wave spends issuing, stalled at issue (pipe busy / dependency; SQ_WAIT_INST_LDS = the LDS part of it) and parked
    (s_waitcnt, s_barrier)."""
import importlib, json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
P = importlib.import_module("odr-dabmod_amd")
ALGO = {"cfg2": 946176 + 1572864, "cfg3": 28800 + 1572864, "ifft_fir_stage": 946176 + 1572864, "cfg4": 28800 + 6291456}
HBM_PEAK = 8.0e12
top = sys.argv[1]
frames = dict(a.split("=") for a in sys.argv[2:])
path = os.path.join(root, "profiles", "traffic.json")
mix_path = os.path.join(root, "profiles", "isa_mix.json")
mix = json.load(open(mix_path)) if os.path.exists(mix_path) else {}
if mix.get("source_hash") != P.source_hash():
    sys.exit("profiles/isa_mix.json belongs to other device sources: run tools/isa_mix.py --json profiles/isa_mix.json first")
d = {"_about": __doc__.split("Units and corrections")[1].strip(),
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
    def tot(name):
        return sum(b.get(name, 0.0) for b in blocks.values())
    w, fe = tot("WRITE_SIZE"), tot("FETCH_SIZE")
    n = int(frames.get(wl, 0))
    e = {"frames": n, "hbm_bytes_per_launch": int(w * 1024 + 2 * fe * 1024), "write_size_kb": w, "fetch_size_kb": fe,
         "algorithmic_bytes_per_launch": ALGO[wl] * n}
    if n:
        e["traffic_over_algorithmic"] = round(e["hbm_bytes_per_launch"] / e["algorithmic_bytes_per_launch"], 4)
    # the dominant kernel of the launch (cfg 4: the resampler) carries the utilisation figures
    dom = "resampler" if "resampler" in blocks else "tf_kernel"
    b = blocks.get(dom, {})
    gui = b.get("GRBM_GUI_ACTIVE", 0.0)
    if gui and b.get("SQ_INSTS_VALU"):
        cyc = gui / 8
        fpk = mix.get(wl, {}).get("packed_fraction_of_valu", 0.0)
        dur_s = b.get("_duration_ns", 0.0) * 1e-9
        wave = b.get("SQ_WAVE_CYCLES", 0.0)
        e.update({"dominant_kernel": dom,
                  "packed_fraction_of_valu": fpk,
                  "valu_busy": round(b["SQ_INSTS_VALU"] * (2 + 2 * fpk) / (1024 * cyc), 3),
                  "lds_busy": round(b.get("SQ_LDS_IDX_ACTIVE", 0.0) / (256 * cyc), 3),
                  "lds_bank_conflict_share": round(b.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(b.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0), 3),
                  "valu_insts_per_launch": b["SQ_INSTS_VALU"], "lds_insts_per_launch": b.get("SQ_INSTS_LDS", 0.0),
                  "gpu_cycles_per_launch_profiled": sum(bb.get("GRBM_GUI_ACTIVE", 0.0) for bb in blocks.values()) / 8})
        if dur_s:
            e["hbm_frac"] = round(e["hbm_bytes_per_launch"] / (sum(bb.get("_duration_ns", 0.0) for bb in blocks.values()) * 1e-9) / HBM_PEAK, 3)
            e["effective_clock_GHz_profiled"] = round(cyc / dur_s / 1e9, 3)
        if wave:
            e.update({"wave_active_frac": round(b.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 3),
                      "wave_issue_stall_frac": round(b.get("SQ_WAIT_INST_ANY", 0.0) / wave, 3),
                      "wave_issue_stall_lds_frac": round(b.get("SQ_WAIT_INST_LDS", 0.0) / wave, 3),
                      "wave_parked_frac": round(b.get("SQ_WAIT_ANY", 0.0) / wave, 3)})
    d[wl] = e
json.dump(d, open(path, "w"), indent=1)
print(json.dumps(d, indent=1))
