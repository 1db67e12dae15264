cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for mode in 1 2 3 4; do
  B=$((16384 * (mode == 1 ? 1 : (mode == 4 ? 2 : 4))))
  for pass in "WRITE_SIZE" "FETCH_SIZE"; do
    out=/tmp/pm_$mode
    rm -rf $out
    (cd /tmp && rocprofv3 --pmc $pass -d $out -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/prof_run.py 3 $B 3 $mode > /dev/null 2>&1)
    f=$(find $out -name "*counter_collection.csv" | head -1)
    python3 - "$f" $mode $B <<'PY'
import csv, sys, collections
f, mode, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f)):
    if "tf_kernel" not in r["Kernel_Name"]: continue
    acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
out_bytes = {1: 196608, 2: 49152, 3: 49152, 4: 98304}[mode] * 8 * B
for k in sorted(acc):
    v = acc[k] / max(n[k], 1)
    print("mode %d  %-12s %.4g KB per launch = %.3f x the output bytes (%d launches)" % (mode, k, v, v * 1024 / out_bytes, n[k]))
PY
  done
done
