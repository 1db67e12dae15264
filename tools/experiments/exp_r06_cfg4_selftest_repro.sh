cd $GRAFT_REPO_ROOT
python - <<'PY'
import numpy as np, oracle as O, subprocess, os, sys, collections
from tests.golden.synth import synth_bits
sys.path.insert(0, ".")
n, mode = 6, 1
per = O.tf_input_bytes(mode)
bits = np.stack([synth_bits(per, seed=600 + i) for i in range(n)])
os.makedirs("/tmp/rp", exist_ok=True)
bits.tofile("/tmp/rp/bits.bin")
O.write_poly_file("/tmp/rp/poly.coef", (1.0, 0.05, -0.01, 0.002, 0.0), (0.0, 0.02, 0.003, 0.0, 0.0))
BIN = "odr-dabmod_amd/host/host_selftest"
rc = collections.Counter()
for i in range(int(os.environ.get("RUNS", "60"))):
    r = subprocess.run([BIN, "cfg4", "/tmp/rp/bits.bin", str(n), "/tmp/rp/graph.iq", "/tmp/rp/poly.coef", "8192000"], capture_output=True, text=True, timeout=600)
    rc[r.returncode] += 1
    if r.returncode != 0 and rc[r.returncode] <= 2:
        print("run", i, "rc", r.returncode, "stdout tail:", r.stdout[-200:].replace("\n", " | "), "stderr tail:", r.stderr[-600:].replace("\n", " | "))
print("return codes:", dict(rc))
PY
which gdb catchsegv 2>&1 | head -2
