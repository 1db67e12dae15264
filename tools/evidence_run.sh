#!/bin/bash
# One GPU-box call that produces everything tools/collect_evidence.sh <tag> copies into profiles/ (about 25 minutes):
#   gpurun --timeout 3000 -- 'bash tools/evidence_run.sh r06'      then, in the build container:   bash tools/collect_evidence.sh r06
# Order: the profiler passes first (they define traffic.json), the bench line, the timing tools, the full -m gpu suite last (it
# is the last writer of gpurun_out/measured_bounds.jsonl and dispatch_matrix.txt).
tag=${1:?usage: evidence_run.sh <tag>}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
G=$R/gpurun_out
mkdir -p $G
noids() { grep -v amdgpu.ids; }
bash tools/profile_all.sh $tag > $G/profile_all_$tag.log 2>&1
bash tools/profile_variants.sh ${tag}v > $G/profile_variants_$tag.log 2>&1
bash tools/prof_modes.sh 2>&1 | noids > $G/${tag}_modes_counters.txt
cd $R
t0=$SECONDS
python bench.py > $G/${tag}_bench_line.json 2> $G/${tag}_bench_line.err
echo "python bench.py (default flags): $((SECONDS - t0)) s wall" > $G/${tag}_bench_wall.txt
python tools/time_host_path.py 2>&1 | noids > $G/host_path.txt
python tools/time_host_path.py s16 2>&1 | noids > $G/host_path_s16.txt
python tools/time_small.py 2>&1 | noids > $G/time_small.txt
python tools/time_modes.py 1234 16384 2>&1 | noids > $G/${tag}_small_modes_final.txt
python tools/time_gain_rounding.py 4096 2>&1 | noids > $G/${tag}_gain_rounding_rates.txt
python -m pytest tests -m gpu -q > $G/final_tests.log 2>&1
tail -3 $G/final_tests.log
tail -c 600 $G/${tag}_bench_line.json
cat $G/${tag}_bench_wall.txt
