python -m pytest tests/test_gain_rounding_gpu.py tests/test_host_mirror.py -m gpu -q -x -k "gain_rounding or reference_gain or config1" 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/gr -o gr -- python $GRAFT_REPO_ROOT/tools/time_gain_rounding.py 4096 > /tmp/gr.log 2>&1
python3 - <<'PY'
import glob, sqlite3
for f in glob.glob("/tmp/gr/**/*.db", recursive=True):
    c = sqlite3.connect(f)
    for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-90s calls=%d avg_us=%.1f pct=%.1f" % (name[:90], calls, avg/1e3, pct))
PY
