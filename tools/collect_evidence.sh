#!/bin/bash
# Copy what a GPU run of tools/profile_all.sh <tag> + tools/profile_variants.sh <tag>v + the timing tools left under gpurun_out/
# into profiles/ (tracked), named per round, and rebuild profiles/traffic.json for the current device sources.
# usage: tools/collect_evidence.sh r04 [frames cfg3] [frames others] [frames cfg4]
set -e
tag=$1; B3=${2:-32768}; BO=${3:-16384}; B4=${4:-4096}
R="$(cd "$(dirname "$0")/.." && pwd)"
G=$R/gpurun_out; P=$R/profiles
for wl in cfg2 cfg3 cfg4 ifft_fir_stage; do cp $G/prof_$tag/$wl/summary.txt $P/${tag}_${wl}_rocprofv3_summary.txt; done
cp $G/prof_$tag/bench_stats_summary.txt $P/${tag}_cfg3_bench_kernel_trace_stats.txt
cp $G/prof_$tag/bench_under_rocprof.json $P/${tag}_cfg3_bench_under_rocprof.json
for v in nofir cfr cfr_nofir window; do
  [ -f $G/prof_${tag}v/$v/summary.txt ] && cp $G/prof_${tag}v/$v/summary.txt $P/${tag}_cfg3_${v}_rocprofv3_summary.txt
done
[ -f $G/host_path.txt ] && { grep -v amdgpu.ids $G/host_path.txt > $P/${tag}_host_path.txt; [ -f $G/host_path_s16.txt ] && grep -v amdgpu.ids $G/host_path_s16.txt >> $P/${tag}_host_path.txt; }
[ -f $G/time_small.txt ] && grep -v amdgpu.ids $G/time_small.txt > $P/${tag}_small_batches.txt
[ -f $G/measured_bounds.jsonl ] && cp $G/measured_bounds.jsonl $P/${tag}_measured_bounds.jsonl
[ -f $G/dispatch_matrix.txt ] && cp $G/dispatch_matrix.txt $P/${tag}_dispatch_matrix.txt
[ -f $G/d2h.txt ] && cp $G/d2h.txt $P/${tag}_d2h_pageable.txt
# (tools/evidence_run.sh: the bench line, the modes' rates and counters, the gain roundings' rates, the -m gpu suite's log)
[ -s $G/${tag}_bench_line.json ] && cp $G/${tag}_bench_line.json $P/${tag}_bench_line.json
[ -s $G/${tag}_small_modes_final.txt ] && cp $G/${tag}_small_modes_final.txt $P/${tag}_small_modes_final.txt
[ -s $G/${tag}_modes_counters.txt ] && cp $G/${tag}_modes_counters.txt $P/${tag}_modes_counters.txt
[ -s $G/${tag}_gain_rounding_rates.txt ] && cp $G/${tag}_gain_rounding_rates.txt $P/${tag}_gain_rounding_rates.txt
[ -s $G/final_tests.log ] && tail -40 $G/final_tests.log > $P/${tag}_gpu_tests.txt
python3 $R/tools/isa_mix.py --json $P/isa_mix.json > /dev/null
python3 $R/tools/make_traffic.py $G/prof_$tag cfg3=$B3 cfg2=$BO ifft_fir_stage=$BO cfg4=$B4 > /dev/null
python3 $R/tools/readme_dispatch.py > /dev/null
echo "profiles/${tag}_* written; traffic.json source hash: $(python3 -c "import json;print(json.load(open('$P/traffic.json'))['source_hash'])")"
