#!/bin/bash
# rocprofv3 evidence for every bench workload (run on the GPU box): kernel-trace stats of the bench command itself
# for the headline workload, then per workload a stats run and the PMC passes (separate runs, no tracing beside
# --pmc) that profiles/traffic.json is made from.
# usage: tools/profile_all.sh <tag> [frames cfg3] [frames others] [frames cfg4]
set -u
tag=$1; B3=${2:-32768}; BO=${3:-16384}; B4=${4:-4096}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
top=$R/gpurun_out/prof_$tag
mkdir -p $top
rocprofv3 --kernel-trace --stats -d $top/bench_stats -o stats -- python $R/bench.py --frames $B3 --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $top/bench_under_rocprof.json 2> $top/bench_stats.log
python3 - <<PYEOF > $top/bench_stats_summary.txt 2>&1
import glob, sqlite3
for f in glob.glob("$top/bench_stats/**/*.db", recursive=True):
    c = sqlite3.connect(f)
    for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        dur = [r[0] / 1e3 for r in c.execute("select duration from kernels where name = ? order by start", (name,))]
        tail = dur[len(dur) // 2:] or [avg]
        print("%-100s calls=%d avg_us=%.2f last_half_avg_us=%.2f total_us=%.1f pct=%.1f" % (name[:100], calls, avg, sum(tail) / len(tail), total, pct))
PYEOF
find $top/bench_stats -name "*.db" -delete
for wl in cfg3 cfg2 ifft_fir_stage cfg4; do
  B=$BO; [ $wl = cfg3 ] && B=$B3; [ $wl = cfg4 ] && B=$B4
  out=$top/$wl
  mkdir -p $out
  rocprofv3 --kernel-trace --stats -d $out/stats -o stats -- python $R/tools/prof_run.py $wl $B 5 > $out/stats.log 2>&1
  i=0
  for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
             "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --pmc $pmc -d $out/pmc$i -o pmc -- python $R/tools/prof_run.py $wl $B 3 > $out/pmc$i.log 2>&1
  done
  python3 $R/tools/prof_summary.py $out > $out/summary.txt 2>&1
  find $out -name "*.db" -delete          # (the sqlite traces are tens of MB: only the summaries travel back)
  echo "== $wl (B = $B)"; grep -E "tf_kernel|resampler" $out/summary.txt | head -4 | cut -c1-170
done
