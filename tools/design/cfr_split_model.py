#!/usr/bin/env python3
"""Numpy model for the round-4 review's CFR proposal (item 6): the second IFFT of crest-factor reduction
(/root/reference/src/OfdmGenerator.cpp:344-370) takes X' = c + clip(ref - c) with c = FFT(clipped symbol) / N, and
IFFT(c) is the clipped symbol that is already in registers -- so transform only the clipped ERROR and add the result to the
held clipped symbol:  t' = t_c + IFFT(e_c)  instead of  t' = IFFT(c + e_c).

(1) accuracy: both forms in float32 against the float64 evaluation of the reference's order of operations;
(2) what each form keeps live across the last transform (per lane, 8 samples per lane): the register accounting that decides
    whether the split buys a fifth wave.
"""
import numpy as np

N, K = 2048, 1536
CLIP, ERRCLIP = 50.0, 0.1          # doc/example.ini of the reference


def symbol(rs):
    q = rs.randint(0, 4, K)
    x = np.exp(1j * (2 * q + 1) * np.pi / 4)
    X = np.zeros(N, np.complex128)
    X[1:K // 2 + 1] = x[:K // 2]
    X[N - K // 2:] = x[K // 2:]
    return X


def cfr(X, dt, split):
    """dt: complex64 (fp32 transforms, numpy >= 2 computes them in single precision) or complex128."""
    f = np.float32 if dt == np.complex64 else np.float64
    X = X.astype(dt)
    t = (np.fft.ifft(X) * N).astype(dt)
    m2 = (t.real * t.real + t.imag * t.imag).astype(f)
    tc = np.where(m2 > f(CLIP) ** 2, t * (f(CLIP) / np.sqrt(m2)).astype(f), t).astype(dt)
    c = (np.fft.fft(tc) / N).astype(dt)
    e = (X - c).astype(dt)
    e2 = (e.real * e.real + e.imag * e.imag).astype(f)
    ec = np.where(e2 > f(ERRCLIP) ** 2, e * (f(ERRCLIP) / np.sqrt(e2)).astype(f), e).astype(dt)
    if split:
        return (tc + (np.fft.ifft(ec) * N).astype(dt)).astype(dt)
    return (np.fft.ifft((c + ec).astype(dt)) * N).astype(dt)


def main():
    rs = np.random.RandomState(3)
    worst = {False: 0.0, True: 0.0}
    for _ in range(200):
        X = symbol(rs)
        ref = cfr(X, np.complex128, False)
        for split in (False, True):
            y = cfr(X, np.complex64, split)
            worst[split] = max(worst[split], np.linalg.norm(y - ref) / np.linalg.norm(ref))
    print("rel-RMS against the float64 evaluation, worst of 200 symbols:")
    print("  t' = IFFT(c + e_c)      (the reference's form, the kernel's today): %.3g" % worst[False])
    print("  t' = t_c + IFFT(e_c)    (the proposal)                            : %.3g" % worst[True])
    print()
    print("live complex values per lane across the LAST transform (8 samples per lane; statistics on = always, the reference")
    print("computes PAPR before / after and the MER of one symbol per frame, src/OfdmGenerator.cpp:232-306):")
    print("  today   : X' (8, being transformed) + the unclipped symbol for the MER (8)              = 16")
    print("  proposal: e_c (8, being transformed) + t_c to add afterwards (8) + unclipped symbol (8)   = 24")
    print("  with FIRFilter the last transform is the packed pair (X', X' H): the proposal would also need IFFT(c H), the")
    print("  FILTERED clipped symbol, which nobody holds -- a fourth transform.")


if __name__ == "__main__":
    main()
