#!/usr/bin/env python3
"""Signed per-symbol deviation of the fused chain's gain scalar from the exact value (float64 statistics of the oracle's own
symbols), and the per-symbol scale of the chain WITHOUT GainControl against the oracle (a systematic scale error of the
transform would show there).  usage (GPU box): python tools/gain_scalar_probe.py"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle as O
from tests.golden.synth import synth_bits
P = importlib.import_module("odr-dabmod_amd")
mode = 1
g = O.mode_params(mode)
ns, ss, nsym, K, N = g["null_size"], g["sym_size"], g["nb_symbols"], g["carriers"], g["spacing"]
nf = 3
bits = np.stack([synth_bits(O.tf_input_bytes(mode), seed=1000 + i) for i in range(nf)])
norm = 1.0 / 50000.0
pr, _ = O.phase_reference(mode)


def alphas(y, ref):
    out = np.zeros((nf, nsym + 1))
    for f in range(nf):
        for s in range(nsym + 1):
            lo = 0 if s == 0 else ns + (s - 1) * ss
            hi = ns if s == 0 else lo + ss
            r, d = ref[f, lo:hi].astype(np.complex128), y[f, lo:hi].astype(np.complex128)
            e = np.vdot(r, r).real
            out[f, s] = np.vdot(r, d).real / e if e else 1.0
    return out


ratio = np.ones((nf, nsym + 1))
for f in range(nf):
    z = O.signal_mux(np.zeros(K, np.complex64), O.diff_mod(pr, O.freq_interleave(O.qpsk_map(bits[f], K), mode), K))
    x = O.ofdm_generate(z, nsym + 1, K, N).reshape(nsym + 1, N)
    yg = O.gain_control(x.reshape(-1), N, 2, 1.0, norm, 4.0).reshape(nsym + 1, N)
    for s in range(1, nsym + 1):
        xs, ys = x[s].astype(np.complex128), yg[s].astype(np.complex128)
        ratio[f, s] = (np.vdot(xs, ys).real / np.vdot(xs, xs).real) / (32767.0 / (4.0 * max(xs.real.std(), xs.imag.std())) * float(np.float32(norm)))
    ratio[f, 0] = ratio[f, 1]

for name, stages, gain in (("no gain, no FIR", 0, None), ("no gain, FIR", P.STAGE_FIR, None), ("gain var, no FIR", P.STAGE_GAIN, 2),
                           ("gain var, FIR (cfg 3)", P.STAGE_GAIN | P.STAGE_FIR, 2)):
    md = P.Modulator(mode=mode, max_frames=nf)
    kw = {}
    if gain is not None:
        md.set_gain(gain, 1.0, norm, 4.0)
        kw = dict(gain_mode=gain, normalise=norm)
    y = md.chain(bits, stages)
    ref = O.Chain(mode=mode, stages=stages & 0xF, **kw).process(bits)
    a = alphas(y, ref)[:, 1:]
    dev = a - 1.0 if gain is None else a * ratio[:, 1:] - 1.0
    print("%-24s %s: mean %+.3e  std %.3e  min %+.3e  max %+.3e" % (name, "scale vs oracle " if gain is None else "g_dev / g_exact - 1",
                                                                dev.mean(), dev.std(), dev.min(), dev.max()))
    if gain is not None:
        print("%-24s reference / exact - 1 : mean %+.3e  std %.3e  min %+.3e  max %+.3e" % ("", (ratio[:, 1:] - 1).mean(), (ratio[:, 1:] - 1).std(),
                                                                                         (ratio[:, 1:] - 1).min(), (ratio[:, 1:] - 1).max()))
    md.close()
