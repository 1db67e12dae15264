#!/usr/bin/env python3
"""DESIGN.md section 6, "Round 5: the line as the driver runs it": the table's rows written from profiles/r05_bench_line.json
(and the rocprofv3 run of the same command beside it), so that the document and the committed line cannot drift apart.
usage: python tools/design_table.py      (rewrites the rows between the table's first and last row in DESIGN.md)"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
d = json.load(open(os.path.join(P, "r05_bench_line.json")))
u = json.load(open(os.path.join(P, "r05_cfg3_bench_under_rocprof.json")))
tr = re.search(r"last_half_avg_us=([0-9.]+)", open(os.path.join(P, "r05_cfg3_bench_kernel_trace_stats.txt")).readline()).group(1)
r, o = d["roofline"], d["other_workloads"]
p = r["power"]
M = lambda k: o[k]["frames_per_s"] / 1e6
F = lambda k: 100 * o[k]["roofline_frac"]
pw = lambda k: o[k]["power"]
c4 = o["cfg4"]["parts"]
others = []
for name in ("second", "third", "fourth", "fifth", "sixth"):
    f = os.path.join(P, "r05_bench_line_%s_box.json" % name)
    if os.path.exists(f):
        others.append("`..._%s_box.json` %.1f %%" % (name, 100 * json.load(open(f))["roofline"]["frac"]))
rows = f"""| **cfg 3** (`value`), 32768 frames per step | **{d['value']/1e6:.2f} M** ({d['ms_per_step']:.3f} ms per step; HIP events {r['kernel_ms_per_launch']:.3f} ms; the short run under the rocprofv3 kernel trace, `profiles/r05_cfg3_bench_under_rocprof.json`: {u['roofline']['kernel_ms_per_launch']:.3f} ms by its own events, {float(tr)/1e3:.3f} ms in the trace over the last half of the launches, `profiles/r05_cfg3_bench_kernel_trace_stats.txt`) | **{100*r['frac']:.1f} %** (2.71-2.87 M = 54.3-57.4 % over the boxes of the round, same frame kernel: `profiles/r05_bench_line{', '.join(others)[4:] if others else ''}; earlier builds on faster boxes 56.8 / 57.2 %) | {p['watts_avg']:.0f} W of 1400, {p['sclk_MHz_avg']/1e3:.2f} GHz, {p['joules_per_frame']*1e3:.2f} mJ per frame | traffic {r['traffic_over_algorithmic']:.3f} x algorithmic; VALU {100*r['valu_busy']:.0f} % + LDS {100*r['lds_busy']:.0f} % of the SIMD time; LDS bank conflicts {100*r['lds_bank_conflict_share']:.1f} % of LDS cycles (11.3 % before the bit gather was re-laid) |
| cfg 3, 16 frames per call, ONE context, three lanes | **{M('cfg3_B16'):.2f} M** ({o['cfg3_B16']['us_per_call']:.1f} µs per call) | **{F('cfg3_B16'):.1f} %** | | one lane: {M('cfg3_B16_one_lane'):.2f} M = {F('cfg3_B16_one_lane'):.1f} % (round 4: 15.8 %; its 25 % needed two contexts) |
| cfg 3, 256 per call, one context, three lanes | **{M('cfg3_B256'):.2f} M** | **{F('cfg3_B256'):.1f} %** | | one lane: {M('cfg3_B256_one_lane'):.2f} M = {F('cfg3_B256_one_lane'):.1f} % |
| cfg 3, one frame per call, three lanes | {M('cfg3_B1')*1e3:.0f} k ({o['cfg3_B1']['us_per_call']:.1f} µs per call) | {F('cfg3_B1'):.1f} % | | one lane: {M('cfg3_B1_one_lane')*1e3:.0f} k ({o['cfg3_B1_one_lane']['us_per_call']:.1f} µs) |
| default chain (cfg 3 without FIRFilter) | {M('cfg3_nofir'):.2f} M | {F('cfg3_nofir'):.1f} % | {pw('cfg3_nofir')['watts_avg']:.0f} W, {pw('cfg3_nofir')['sclk_MHz_avg']/1e3:.2f} GHz, {pw('cfg3_nofir')['joules_per_frame']*1e3:.2f} mJ | |
| cfg 2 / IFFT + FIR stage (from carriers) | {M('cfg2'):.2f} M / {M('ifft_fir_stage'):.2f} M | {F('cfg2'):.1f} % / {F('ifft_fir_stage'):.1f} % | {pw('cfg2')['watts_avg']:.0f} W, {pw('cfg2')['sclk_MHz_avg']/1e3:.2f} GHz (nominal) | the box's copy rate |
| **cfg 4** | **{M('cfg4')*1e3:.0f} k** | **{F('cfg4'):.1f} %**; {100*o['cfg4']['valu_frac_of_peak']:.1f} % of the fp32 vector peak | {pw('cfg4')['watts_avg']:.0f} W, {pw('cfg4')['sclk_MHz_avg']/1e3:.2f} GHz, {pw('cfg4')['joules_per_frame']*1e3:.2f} mJ per frame | its two kernels alone: frame kernel {c4['frame_kernel']['ms_per_launch']:.2f} ms / {c4['frame_kernel']['joules_per_frame']*1e3:.2f} mJ, resampler + predistorter {c4['resampler_poly']['ms_per_launch']:.2f} ms / {c4['resampler_poly']['joules_per_frame']*1e3:.2f} mJ per frame (`other_workloads.cfg4.parts`) |
| cfg 3 + CFR; the same with s16 output | {M('cfg3_cfr'):.2f} M; **{M('cfg3_cfr_s16'):.2f} M** | {F('cfg3_cfr'):.1f} % | | four waves per SIMD since round 5 (section 9.6; 1.19 M on the A/B's box); s16: ONE kernel (the CFR kernel stores the integers and adds TII itself; before: CFR kernel -> `format_kernel`) |
| cfg 3 + OFDM windowing (overlap 10) | **{M('cfg3_window'):.2f} M** | **{F('cfg3_window'):.1f} %** | | the equalised-boundary kernel with the seam in its boundary outputs (section 4.4); before: 1.97 M = 39.5 % on the packed dual transform |
| cfg 3 in modes II / III / IV (generic kernels) | {M('cfg3_mode2'):.2f} M / {M('cfg3_mode3'):.2f} M / {M('cfg3_mode4'):.2f} M of their frames | {F('cfg3_mode2'):.1f} % / {F('cfg3_mode3'):.1f} % / {F('cfg3_mode4'):.1f} % | | new in the line |
| s16 output: cfg 3 / cfg 4 | {M('cfg3_s16'):.2f} M / {M('cfg4_s16')*1e3:.0f} k | | | u8 / s8 from the frame kernel's own store: see `profiles/r05_variant_survey.txt` |"""
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
i = s.find("| **cfg 3** (`value`), 32768 frames per step |")
j = s.find("\n", s.find("| s16 output: cfg 3 / cfg 4 |", i))
assert i > 0 and j > i
s = s[:i] + rows + s[j:]
s = re.sub(r"of the same run: \d+ frames/s on 16 host cores", "of the same run: %d frames/s on 16 host cores" % round(d["cpu_baseline"]["value"]), s)
open(path, "w").write(s)
print("DESIGN.md section 6 table: cfg 3 %.2f M = %.1f %%" % (d["value"] / 1e6, 100 * r["frac"]))
