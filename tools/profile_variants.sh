#!/bin/bash
# rocprofv3 evidence for the chains beside the bench workloads (run on the GPU box): the reference's default chain (no
# FIRFilter), cfg 3 with crest-factor reduction (f-3) and with OFDM windowing (f-4).  Per chain a kernel-trace stats run and the
# PMC passes of tools/profile_all.sh (separate runs, nothing beside --pmc), condensed by tools/prof_summary.py.
# usage: tools/profile_variants.sh <tag> [frames]
set -u
tag=$1; B=${2:-8192}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
top=$R/gpurun_out/prof_$tag
mkdir -p $top
for wl in nofir cfr cfr_nofir window; do
  unset CFR WIN; mask=3
  case $wl in nofir) mask=1;; cfr) export CFR=1;; cfr_nofir) export CFR=1; mask=1;; window) export WIN=10;; esac
  out=$top/$wl
  mkdir -p $out
  rocprofv3 --kernel-trace --stats -d $out/stats -o stats -- python $R/tools/sweep_b.py $mask $B,1 > $out/stats.log 2>&1
  i=0
  for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
             "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $pmc -d $out/pmc$i -o pmc -- python $R/tools/sweep_b.py $mask $B,1 > $out/pmc$i.log 2>&1
  done
  python3 $R/tools/prof_summary.py $out > $out/summary.txt 2>&1
  find $out -name "*.db" -delete
  echo "== $wl (B = $B)"; grep -E "tf_kernel" $out/summary.txt | head -2 | cut -c1-170
done
