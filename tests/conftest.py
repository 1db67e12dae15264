import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """Import the product package (its directory name has a hyphen)."""
    import importlib
    if "odr_dabmod_amd" in sys.modules:
        return sys.modules["odr_dabmod_amd"]
    mod = importlib.import_module("odr-dabmod_amd")
    sys.modules["odr_dabmod_amd"] = mod
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


# Integer outputs of the fused chain (FormatConverter inside the last kernel) against the reference's: truncations of two
# float streams that agree to rel-RMS 2.4e-7 (bar 1e-6).  They are NEVER more than one step apart, and they differ exactly
# where an integer lies between the two floats:
#   (1) a share of the components equal to the mean absolute difference in steps, about 0.8 x 2.4e-7 x the RMS of the output in
#       steps -- 2e-3 for a file output at normalise 1.0 (10 000 steps RMS), 1e-3 in the s16 cells of the dispatch matrix;
#   (2) whatever the amplitude, the samples that are integers in EXACT arithmetic: without GainControl (or in mode fix at a
#       normalise that keeps them) every second symbol's carriers are +-1 / +-i, so samples 0, N/4, N/2, 3N/4 of it (and their
#       copies in the cyclic prefix) are sums of integers -- 12.9999995 here, 13.0000005 there.  Four samples in N: measured
#       1.2e-4 ... 2.6e-4 of the components at N = 2048 (profiles/r05_dispatch_matrix.txt), eight times that at N = 256;
#   (3) u8 alone: the offset of 128 makes ZERO a truncation boundary (127.99999 -> 127, 128.00001 -> 128), and behind the
#       Resampler the null symbol is not exact zeros but rounding residue of either sign: measured 5.8e-4.
# The limit is 2.5 x expectation (1) plus floors for (2) and (3); a kernel that gets 1 % of the samples wrong by a step fails at
# every amplitude a test uses (the largest limit, Mode III u8 at 10 000 steps RMS, would be 9.9e-3; the tests' largest is 6e-3).
def int_off_by_one_limit(want, n_fft=2048, fmt="s16"):
    import numpy as np
    rms = float(np.sqrt(np.mean(np.asarray(want, dtype=np.float64) ** 2)))
    if fmt == "u8":
        rms = float(np.sqrt(np.mean((np.asarray(want, dtype=np.float64) - 128.0) ** 2)))
    return 5e-7 * rms + 1.0 / n_fft + (1.5e-3 if fmt == "u8" else 0.0)


def record_bound(name, measured, limit, warn_at=None):
    """Log a measured worst case next to the bound it is held to (gpurun_out/measured_bounds.jsonl, one JSON
    object per line; copied to profiles/ when a bound is (re)derived from it) and return measured <= limit.
    warn_at: a tighter, warning-level bound -- a measured value beyond it still passes but is reported
    (warnings.warn -> pytest's warnings summary, and "warned" in the log), so that a drift towards `limit` is seen
    before it is reached."""
    import json
    import warnings
    warned = warn_at is not None and float(measured) > float(warn_at)
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        rec = {"name": name, "measured": float(measured), "limit": float(limit)}
        if warn_at is not None:
            rec.update(warn_at=float(warn_at), warned=bool(warned))
        with open(os.path.join(d, "measured_bounds.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    if warned and float(measured) <= float(limit):
        warnings.warn("%s: measured %.3g is beyond the warning level %.3g (limit %.3g)" % (name, measured, warn_at, limit))
    return float(measured) <= float(limit)
