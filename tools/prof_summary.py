#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory (rocprofv3 rocpd sqlite) into a short summary."""
import glob, os, sqlite3, sys, collections
d = sys.argv[1]
KEEP = ("tf_kernel", "resampler", "poly_kernel", "fir_kernel", "gain_kernel", "guard_")
print("== kernel stats (rocprofv3 --kernel-trace --stats)")
for f in glob.glob(os.path.join(d, "stats", "**", "*.db"), recursive=True):
    c = sqlite3.connect(f)
    for name, calls, total, avg, pct in c.execute(
            "select name,total_calls,total_duration,average,percentage from top_kernels"):
        # the first launches run on a cold device (clock ramp, first touch of the output pages): also give the
        # mean of the LAST half of the calls, which is what a timed region after warm-up sees
        dur = [r[0] / 1e3 for r in c.execute("select duration from kernels where name = ? order by start", (name,))]
        tail = dur[len(dur) // 2:] or [avg]
        print("  %-78s calls=%d avg_us=%.2f last_half_avg_us=%.2f total_us=%.1f pct=%.1f"
              % (name[:78], calls, avg, sum(tail) / len(tail), total, pct))
print("== PMC (average per dispatch)")
for f in sorted(glob.glob(os.path.join(d, "pmc*", "**", "*.db"), recursive=True)):
    c = sqlite3.connect(f)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    info = {}
    for k, cn, v, dur, vg, sc, lds in c.execute(
            "select kernel_name,counter_name,value,duration,vgpr_count,scratch_size,lds_block_size from counters_collection"):
        if not any(x in k for x in KEEP):
            continue
        acc[k][cn].append(v)
        info[k] = (vg, sc, lds)
        acc[k]["_duration_ns"].append(dur)
    for k, cs in acc.items():
        print("  %s  vgpr=%s scratch=%s lds=%s" % ((k[:90],) + info[k]))
        for cn, v in sorted(cs.items()):
            print("     %-24s %.6g   (n=%d)" % (cn, sum(v) / len(v), len(v)))
