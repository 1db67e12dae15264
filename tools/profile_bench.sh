#!/bin/bash
# rocprofv3 of the bench command itself (kernel trace + stats), then PMC passes on the same workload.
# usage (GPU box): tools/profile_bench.sh <tag> [frames]
set -u
tag=$1; B=${2:-4096}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/prof_$tag
mkdir -p $out/stats
rocprofv3 --kernel-trace --stats -d $out/stats -o stats -- python $R/bench.py --frames $B --steps 5 --warmup 1 --no-extra --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/stats.log
i=0
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc -d $out/pmc$i -o pmc -- python $R/tools/prof_run.py 3 $B 3 > $out/pmc$i.log 2>&1
done
python3 $R/tools/prof_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt | cut -c1-150
