#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for B in 4096 512 128 64; do
  for pmc in FETCH_SIZE WRITE_SIZE; do
    out=$R/gpurun_out/fetch_probe/$B/$pmc; mkdir -p $out
    rocprofv3 --pmc $pmc -d $out -o pmc -- python $R/tools/prof_run.py cfg4 $B 6 > $out/log.txt 2>&1
  done
  python3 - <<PY
import glob, sqlite3, collections
for pmc in ("FETCH_SIZE","WRITE_SIZE"):
    acc=collections.defaultdict(list)
    for f in glob.glob("$R/gpurun_out/fetch_probe/$B/%s/**/*.db"%pmc, recursive=True):
        c=sqlite3.connect(f)
        for k,cn,v in c.execute("select kernel_name,counter_name,value from counters_collection"):
            if "resampler" in k or "tf_kernel" in k: acc[("rs" if "resampler" in k else "tf", cn)].append(v)
    for k,v in sorted(acc.items()): print("B=$B", k, "KB per frame: %.1f (last of %d: %.1f)" % (sum(v)/len(v)/$B, len(v), v[-1]/$B))
PY
done
find $R/gpurun_out/fetch_probe -name "*.db" -delete
