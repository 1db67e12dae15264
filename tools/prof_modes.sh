#!/bin/bash
# rocprofv3 counters of the cfg 3 frame kernel in transmission modes II - IV (the passes of tools/profile_all.sh, one workload)
# usage (GPU box): bash tools/prof_modes.sh > gpurun_out/r06_modes_counters.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for mode in 2 3 4; do
  B=$((16384 * (mode == 4 ? 2 : 4)))
  for pass in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_BUSY_CYCLES"; do
    out=/tmp/pm_$mode
    rm -rf $out
    (cd /tmp && rocprofv3 --pmc $pass -d $out -o pmc --output-format csv -- python $OLDPWD/tools/prof_run.py 3 $B 3 $mode > /dev/null 2>&1)
    f=$(find $out -name "*counter_collection.csv" | head -1)
    python3 - "$f" $mode <<'PY'
import csv, sys, collections
f, mode = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f)):
    if "tf_kernel" not in r["Kernel_Name"]: continue
    acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc): print("mode %s  %-22s %.4g per launch (%d launches)" % (mode, k, acc[k] / max(n[k], 1), n[k]))
PY
  done
done
