"""Utilisation figures from rocprofv3 PMC counters, shared by tools/make_traffic.py (the committed profiles) and bench.py
(counters collected live in the bench run).  Units and corrections (MI355X_MICROARCH.md):
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
HBM_PEAK = 8.0e12


def figures(blocks, algo_bytes_per_launch, packed_fraction, dominant=None):
    """blocks: {kernel label: {counter: average per dispatch, '_duration_ns': ...}} for the kernels of ONE launch of the
    workload.  Returns the entry profiles/traffic.json / bench.py carry for it."""
    def tot(name):
        return sum(b.get(name, 0.0) for b in blocks.values())
    w, fe = tot("WRITE_SIZE"), tot("FETCH_SIZE")
    e = {"hbm_bytes_per_launch": int(w * 1024 + 2 * fe * 1024), "write_size_kb": w, "fetch_size_kb": fe,
         "algorithmic_bytes_per_launch": algo_bytes_per_launch}
    if algo_bytes_per_launch:
        e["traffic_over_algorithmic"] = round(e["hbm_bytes_per_launch"] / algo_bytes_per_launch, 4)
    if dominant is None:
        dominant = max(blocks, key=lambda k: blocks[k].get("_duration_ns", 0.0)) if blocks else None
    b = blocks.get(dominant, {})
    gui = b.get("GRBM_GUI_ACTIVE", 0.0)
    if gui and b.get("SQ_INSTS_VALU"):
        cyc = gui / 8
        dur_s = b.get("_duration_ns", 0.0) * 1e-9
        wave = b.get("SQ_WAVE_CYCLES", 0.0)
        e.update({"dominant_kernel": dominant, "packed_fraction_of_valu": packed_fraction,
                  "valu_busy": round(b["SQ_INSTS_VALU"] * (2 + 2 * packed_fraction) / (1024 * cyc), 3),
                  "lds_busy": round(b.get("SQ_LDS_IDX_ACTIVE", 0.0) / (256 * cyc), 3),
                  "lds_bank_conflict_share": round(b.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(b.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0), 3),
                  "valu_insts_per_launch": b["SQ_INSTS_VALU"], "lds_insts_per_launch": b.get("SQ_INSTS_LDS", 0.0),
                  "gpu_cycles_per_launch_profiled": sum(bb.get("GRBM_GUI_ACTIVE", 0.0) for bb in blocks.values()) / 8,
                  "dominant_kernel_cycles_profiled": cyc})
        if dur_s:
            e["hbm_frac"] = round(e["hbm_bytes_per_launch"] / (sum(bb.get("_duration_ns", 0.0) for bb in blocks.values()) * 1e-9) / HBM_PEAK, 3)
            e["effective_clock_GHz_profiled"] = round(cyc / dur_s / 1e9, 3)
        if wave:
            e.update({"wave_active_frac": round(b.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 3),
                      "wave_issue_stall_frac": round(b.get("SQ_WAIT_INST_ANY", 0.0) / wave, 3),
                      "wave_issue_stall_lds_frac": round(b.get("SQ_WAIT_INST_LDS", 0.0) / wave, 3),
                      "wave_parked_frac": round(b.get("SQ_WAIT_ANY", 0.0) / wave, 3)})
    return e


def label(kernel_name):
    if "resampler" in kernel_name:
        return "resampler"
    if "tf_kernel" in kernel_name:
        return "tf_kernel"
    return None


def read_rocpd(paths):
    """{label: {counter: average per dispatch}} from rocprofv3's sqlite output files (one per PMC pass)"""
    import collections
    import sqlite3
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in paths:
        c = sqlite3.connect(f)
        try:
            rows = c.execute("select kernel_name, counter_name, value, duration from counters_collection")
        except sqlite3.Error:
            continue
        dur_seen = set()
        for k, cn, v, dur in rows:
            lb = label(k)
            if lb is None:
                continue
            acc[lb][cn].append(v)
            acc[lb]["_duration_ns"].append(dur)
        c.close()
    return {lb: {cn: sum(v) / len(v) for cn, v in cs.items()} for lb, cs in acc.items()}
