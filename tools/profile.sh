#!/bin/bash
# rocprofv3 recipe (run on the GPU box): kernel-trace stats + separate PMC passes.
# usage: tools/profile.sh <tag> <mask> <B>
set -u
tag=$1; mask=$2; B=$3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/stats -o stats -- python $R/tools/prof_run.py $mask $B 5 > $out/stats.log 2>&1
i=0
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT SQ_INSTS_SCRATCH"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc -d $out/pmc$i -o pmc -- python $R/tools/prof_run.py $mask $B 3 > $out/pmc$i.log 2>&1
done
python3 $R/tools/prof_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
