#!/usr/bin/env python3
"""Static instruction mix of the hot loop of the bench kernels, from the device assembly hipcc emits for the committed
sources (no GPU needed): plain / packed VALU, LDS reads and writes by width, VMEM, SALU, waits and barriers per loop
iteration (= one OFDM symbol of tf_kernel, one hop of resampler_kernel), and an ISSUE-TIME MODEL of the iteration priced
with the per-instruction costs measured on MI355X (profiles/r02_issue_cost_microbench.txt: SIMD ticks per wave-instruction
at three or more waves per SIMD).  tools/make_traffic.py takes the packed share of the VALU instructions from here
(a v_pk_*_f32 occupies the SIMD-32 for two passes = 4 cycles, a plain wave64 VALU instruction for 2: MI355X_MICROARCH.md).
usage: tools/isa_mix.py [--json profiles/isa_mix.json]"""
import collections, importlib, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "odr-dabmod_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fvisibility=hidden", "-Xclang", "-target-feature", "-Xclang",
         "-load-store-opt", "-fno-slp-vectorize", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include")]
KERNELS = {
    # bench workload -> (source, extra flags, demangled-name pattern of the kernel, what one loop iteration is)
    "cfg3": ("tf_inst.hip", ["-DTF_LOGN=11", "-DTF_NT=45"],
             r"tf_kernel<11, true, true, true, true, 45, false, false, false, 0, false, true>", "OFDM symbol"),
    "cfg2": ("tf_inst.hip", ["-DTF_LOGN=11", "-DTF_NT=0"],
             r"tf_kernel<11, false, false, true, false, 0, false, false, false, 0, false, false>", "OFDM symbol"),
    "ifft_fir_stage": ("tf_inst.hip", ["-DTF_LOGN=11", "-DTF_NT=45"],
                       r"tf_kernel<11, false, true, true, true, 45, false, true, true, 0, false, false>", "OFDM symbol"),
    "cfg4": ("resampler.hip", [], r"resampler16_kernel<true, false, 4>", "hop (2048 samples in, 8192 out)"),
}
# SIMD ticks per wave-instruction with >= 3 waves per SIMD (profiles/r02_issue_cost_microbench.txt); the packed VALU cost
# is the architectural one (two passes of the SIMD-32), which the clock-throttled microbenchmark understates
COST = {"valu": 2.04, "valu_pk": 4.0, "valu_dpp": 3.0, "valu_trans": 4.0, "ds_read_b32": 6.0, "ds_read_b64": 6.2,
        "ds_read_b128": 11.4, "ds_read2": 11.6, "ds_write_b32": 11.2, "ds_write_b64": 16.5, "ds_write_b128": 35.5,
        "ds_write2": 16.6, "ds_other": 6.0, "salu": 0.0, "vmem": 4.0,
        # global / buffer STORES: 28 cycles per 512 bytes.  Measured on the cfg 3 kernel itself: a build
        # without its ten 8-byte stores per wave and symbol runs 280 SIMD cycles per wave and symbol shorter -- and at exactly
        # the VALU + LDS sum of this model (2238 measured, 2266 modelled).  tools/microbench/issue_cost.cpp: 8- and 16-byte
        # stores cost the same per byte (40 - 47 cycles per 512 bytes with every wave of the CU storing at once).
        "vmem_store_b32": 14.0, "vmem_store_b64": 28.0, "vmem_store_b128": 56.0}


def classify(op):
    if op.startswith("v_"):
        if op.startswith("v_pk_"): return "valu_pk"
        if op.endswith("_dpp") or "_dpp" in op: return "valu_dpp"
        if re.match(r"v_(rcp|sqrt|rsq|exp|log|sin|cos)_", op): return "valu_trans"
        return "valu"
    if op.startswith("ds_"):
        m = re.match(r"ds_(read|write)(2st64|2)?_(b\d+|u8|i8|u16|i16)", op)
        if m:
            if m.group(2): return "ds_%s2" % m.group(1)
            w = m.group(3)
            if w in ("u8", "i8", "u16", "i16"): w = "b32"
            return "ds_%s_%s" % (m.group(1), w)
        return "ds_other"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        m = re.match(r"(global|buffer|flat|scratch)_store_(dwordx4|dwordx2|dwordx3|dword|short|byte)", op)
        if m: return "vmem_store_" + {"dwordx4": "b128", "dwordx3": "b128", "dwordx2": "b64"}.get(m.group(2), "b32")
        return "vmem"
    if op in ("s_waitcnt", "s_barrier", "s_nop") or op.startswith(("s_cbranch", "s_branch")): return op if op in ("s_waitcnt", "s_barrier", "s_nop") else "branch"
    if op.startswith("s_"): return "salu"
    return "other"


def kernel_bodies(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1); out[cur] = []
            continue
        if cur is None: continue
        t = line.strip()
        if t.startswith("s_endpgm"):
            out[cur].append(t); cur = None
            continue
        out[cur].append(line.rstrip())
    return out


def hot_loop(lines):
    """the largest loop: from the label the longest-reaching backward branch targets to that branch"""
    labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    best = (0, 0, 0)
    for i, l in enumerate(lines):
        m = re.match(r"\s+s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
            best = (i - labels[m.group(1)], labels[m.group(1)], i)
    return lines[best[1]:best[2] + 1]


def mix_of(lines):
    c = collections.Counter()
    for l in lines:
        t = re.sub(r";.*$", "", l).strip()
        if not t or t.startswith(".") or t.endswith(":"): continue
        c[classify(t.split()[0])] += 1
    return c


def main():
    res = {"_about": __doc__.split("usage:")[0].strip(), "cost_simd_ticks_per_wave_instruction": COST}
    sys.path.insert(0, ROOT)
    res["source_hash"] = importlib.import_module("odr-dabmod_amd").source_hash()
    cache = {}
    for wl, (src, extra, pat, unit) in KERNELS.items():
        key = (src, tuple(extra))
        if key not in cache:
            out = tempfile.mktemp(suffix=".s")
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-o", out, os.path.join(CSRC, src)],
                                  stderr=subprocess.DEVNULL)
            bodies = kernel_bodies(out)
            names = subprocess.run(["c++filt"], input="\n".join(bodies), capture_output=True, text=True).stdout.splitlines()
            cache[key] = {n: bodies[k] for k, n in zip(bodies, names)}
            os.unlink(out)
        hit = [n for n in cache[key] if pat in n]
        assert len(hit) == 1, (wl, pat, hit[:3])
        body = cache[key][hit[0]]
        loop = hot_loop(body)
        m = mix_of(loop)
        nv = m["valu"] + m["valu_pk"] + m["valu_dpp"] + m["valu_trans"]
        ticks = {k: round(m[k] * COST[k], 1) for k in m if k in COST and m[k]}
        valu_t = sum(v for k, v in ticks.items() if k.startswith("valu"))
        lds_t = sum(v for k, v in ticks.items() if k.startswith("ds_"))
        vmem_t = sum(v for k, v in ticks.items() if k.startswith("vmem"))
        res[wl] = {"kernel": hit[0], "loop_iteration": unit, "loop_instructions": sum(m.values()),
                   "mix": dict(sorted(m.items())), "valu_instructions": nv,
                   "packed_fraction_of_valu": round(m["valu_pk"] / max(nv, 1), 4),
                   "issue_model_simd_ticks": {"valu": round(valu_t, 1), "lds": round(lds_t, 1), "vmem": round(vmem_t, 1),
                                              "total": round(valu_t + lds_t + vmem_t, 1)},
                   "kernel_instructions_total": sum(mix_of(body).values())}
    txt = json.dumps(res, indent=1)
    if "--json" in sys.argv:
        open(sys.argv[sys.argv.index("--json") + 1], "w").write(txt + "\n")
    print(txt)




def phase_table():
    """--phases: the cfg 3 hot loop of the -DDABGPU_PHASE_TIMING build split at its s_memtime stamps (sched_barrier keeps
    the compiler from moving work across them): instruction mix and modelled issue ticks per phase of a symbol iteration"""
    order = ["loop", "input", "butterfly", "exchange", "butterfly", "exchange", "butterfly", "exchange", "butterfly",
             "gain+windows", "stores", "boundary"]
    src, extra, pat, _ = KERNELS["cfg3"]
    out = tempfile.mktemp(suffix=".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-DDABGPU_PHASE_TIMING", "-o", out, os.path.join(CSRC, src)],
                          stderr=subprocess.DEVNULL)
    bodies = kernel_bodies(out)
    names = subprocess.run(["c++filt"], input="\n".join(bodies), capture_output=True, text=True).stdout.splitlines()
    body = [bodies[k] for k, n in zip(bodies, names) if pat in n][0]
    os.unlink(out)
    loop = hot_loop(body)
    segs, cur = [], []
    for l in loop:
        t = re.sub(r";.*$", "", l).strip()
        if t.startswith("s_memtime"):
            segs.append(cur); cur = []
        else:
            cur.append(l)
    tail = cur                                    # after the last stamp: back edge, belongs to "loop" with the head
    segs[0] = tail + segs[0]
    assert len(segs) == len(order), (len(segs), len(order))
    agg = collections.OrderedDict()
    for name, seg in zip(order, segs):
        agg.setdefault(name, collections.Counter()).update(mix_of(seg))
    rows = []
    for name, m in agg.items():
        valu = sum(m[k] * COST[k] for k in m if k.startswith("valu"))
        lds = sum(m[k] * COST[k] for k in m if k.startswith("ds_"))
        vm = sum(m[k] * COST[k] for k in m if k.startswith("vmem"))
        rows.append((name, sum(m[k] for k in m if k.startswith("valu")), sum(m[k] for k in m if k.startswith("ds_read")),
                     sum(m[k] for k in m if k.startswith("ds_write")), sum(m[k] for k in m if k.startswith("vmem")), m["salu"],
                     m["s_barrier"], valu, lds, vm))
    tot = sum(r[7] + r[8] + r[9] for r in rows)
    print("cfg 3 symbol iteration by phase (static, timing build; the stamps' own s_waitcnt / s_memtime / s_sub / s_add not counted as SALU work)")
    print("%-14s %6s %8s %9s %5s %5s %8s %11s %10s %11s %7s" % ("phase", "VALU", "LDS rd", "LDS wr", "VMEM", "SALU", "barriers", "VALU ticks", "LDS ticks", "VMEM ticks", "share"))
    for r in rows:
        print("%-14s %6d %8d %9d %5d %5d %8d %11.0f %10.0f %11.0f %6.1f%%" % (r + (100 * (r[7] + r[8] + r[9]) / tot,)))
    print("%-14s %6d %8d %9d %5d %5d %8d %11.0f %10.0f %11.0f %6.1f%%" % (("total",) + tuple(sum(r[i] for r in rows) for i in range(1, 10)) + (100.0,)))


if __name__ == "__main__":
    if "--phases" in sys.argv:
        phase_table()
    else:
        main()
