#!/bin/bash
# kernel-level timing + a few counters of the unfused chain (windowed guard / CFR)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_unfused
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_unfused/stats -o stats -- env WIN=${WIN:-10} python $R/tools/sweep_b.py 3 2048,0 > $R/gpurun_out/prof_unfused.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $R/gpurun_out/prof_unfused/pmc1 -o pmc -- env WIN=${WIN:-10} python $R/tools/sweep_b.py 3 2048,0 >> $R/gpurun_out/prof_unfused.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/prof_unfused/pmc2 -o pmc -- env WIN=${WIN:-10} python $R/tools/sweep_b.py 3 2048,0 >> $R/gpurun_out/prof_unfused.log 2>&1
python3 $R/tools/prof_summary.py $R/gpurun_out/prof_unfused 2>&1 | grep -v "at::native\|rocclr" | head -40
