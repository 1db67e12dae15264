#!/usr/bin/env python3
"""numpy model of resampler16_kernel (odr-dabmod_amd/csrc/resampler16.hip): the x4 resampler of BASELINE config 4 with
hop-independent work items -- time-domain overlap-add in front of ONE forward transform per hop -- and 4096-point
transforms as three radix-16 Stockham stages on 256 lanes x 16 points (two LDS exchanges).  Checks every index mapping
of the kernel (exchange layouts, stage twiddles, branch twiddles, Nyquist copies, output order) against the oracle's
Resampler (src/Resampler.cpp:142-192 restated) in float64, then the fp32 error of the scheme.
usage: python tools/design/resampler16_model.py"""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import oracle as O

N, T, R, Q = 4096, 256, 16, 4
NOUT, HIN, HOUT = N * Q, N // 2, N // 2 * Q
P1 = T + 2                                    # row pitch of the first exchange


def dft16(v, S, dt):
    """v: (lanes, 16) -> natural-order 16-point DFT with sign S, as 4 x 4 (steps A, B, C of the kernel)"""
    w16 = np.exp(S * 2j * np.pi * np.arange(16) / 16).astype(dt)
    w4 = np.array([1, S * 1j, -1, -S * 1j], dtype=dt)
    a = np.zeros_like(v)
    for m1 in range(4):
        for r2 in range(4):
            a[:, m1 * 4 + r2] = sum(v[:, m1 + 4 * m2] * w4[(m2 * r2) % 4] for m2 in range(4)) * w16[(m1 * r2) % 16]
    y = np.zeros_like(v)
    for r2 in range(4):
        for r1 in range(4):
            y[:, 4 * r1 + r2] = sum(a[:, m1 * 4 + r2] * w4[(m1 * r1) % 4] for m1 in range(4))
    return y


def fft4096(x, S, dt=np.complex128):
    """x: (256, 16) with x[t, m] = sample t + 256 m -> same layout of the DFT with sign S; LDS images as in the kernel"""
    t = np.arange(T)
    wN = lambda e: np.exp(S * 2j * np.pi * (e % N) / N).astype(dt)
    v = dft16(x.astype(dt), S, dt)
    # exchange 1: element (t, r) at r * P1 + t; lane t gathers (t & 15) * P1 + (t >> 4) + 16 m
    lds = np.zeros(16 * P1, dt)
    for r in range(16):
        lds[r * P1 + t] = v[:, r]
    v = np.stack([lds[(t & 15) * P1 + (t >> 4) + 16 * m] for m in range(16)], axis=1)
    # stage 2: twiddle W_256^(m (t % 16)), DFT16, element r to (t / 16) * 256 + t % 16 + 16 r; gather t + 256 m
    v = dft16(v * np.stack([wN(16 * m * (t % 16)) for m in range(16)], axis=1), S, dt)
    lds = np.zeros(N, dt)
    for r in range(16):
        lds[(t // 16) * 256 + t % 16 + 16 * r] = v[:, r]
    v = np.stack([lds[t + 256 * m] for m in range(16)], axis=1)
    # stage 3: twiddle W_4096^(m t) -- in the kernel products of the resident W^t, W^2t, W^4t, W^8t
    tw = np.ones((T, 16), dt)
    p = {1: wN(t), 2: wN(2 * t), 4: wN(4 * t), 8: wN(8 * t)}
    for m in range(1, 16):
        acc = None
        for b in (8, 4, 2, 1):
            if m & b:
                acc = p[b] if acc is None else (acc * p[b]).astype(dt)
        tw[:, m] = acc
    return dft16(v * tw, S, dt)


def resample_x4(stream, halo, dt=np.complex128, ft=np.float64):
    """stream: whole hops of 2048 samples; halo: the 4096 samples before it.  out: 4 x as many samples."""
    i = np.arange(N)
    w = (0.5 * (1.0 - np.cos(2.0 * np.pi * i / (N - 1)))).astype(np.float32).astype(ft)
    factor = ft(2.0 ** -12)                                       # 1 / max(nin, nout) * out / in, src/Resampler.cpp:79-83
    S = np.concatenate([halo, stream]).astype(dt)
    nh = len(stream) // HIN
    t = np.arange(T)
    out = np.zeros(nh * HOUT, dt)
    wp = [np.exp(2j * np.pi * ((t * p) % NOUT) / NOUT).astype(dt) for p in range(Q)]
    for h in range(nh):
        cm2, cm1, c0 = (S[(h + k) * HIN:(h + k + 1) * HIN] for k in range(3))
        # g_h = [ (w1 + w2) c_{h-1} | w2 c_h + w1 c_{h-2} ] * factor, lane layout [t, m] = index t + 256 m
        g = np.concatenate([(w[:HIN] + w[HIN:]) * cm1, w[HIN:] * c0 + w[:HIN] * cm2]) * factor
        x = g.reshape(16, T).T.astype(dt)
        G = fft4096(x, -1, dt)
        nyq = G[0, 8]                                              # bin 2048 = lane 0, slot 8
        o = np.zeros((T, 8, Q), dt)                                # outputs q = t + 256 m (m < 8), branch p
        o[:, :, 0] = N * x[:, :8] + nyq * ((-1.0) ** t)[:, None]
        for p in range(1, Q):
            rot = np.array([np.exp(2j * np.pi * ((m * p) % 64) / 64) * ((-1j) ** p if m >= 8 else 1) for m in range(16)])
            v = (G * wp[p][:, None]).astype(dt) * rot.astype(dt)[None, :]
            v[0, 8] = G[0, 8] * 2.0 * np.cos(np.pi * p / Q)        # the Nyquist bin sits at +nin/2 AND -nin/2
            y = fft4096(v.astype(dt), +1, dt)
            o[:, :, p] = y[:, :8]
        blk = np.zeros((HIN, Q), dt)
        for m in range(8):
            blk[t + 256 * m, :] = o[:, m, :]
        out[h * HOUT:(h + 1) * HOUT] = blk.reshape(-1)
    return out


if __name__ == "__main__":
    rs = np.random.RandomState(3)
    nh = 6
    x = ((rs.randn(nh * HIN) + 1j * rs.randn(nh * HIN)) * 0.2).astype(np.complex64)
    ref = O.Resampler(2048000, 8192000, 2048)
    want = np.concatenate([ref.process(x[:3 * HIN]), ref.process(x[3 * HIN:])])
    got = resample_x4(x.astype(np.complex128), np.zeros(N, np.complex128))
    e64 = np.linalg.norm(got - want) / np.linalg.norm(want)
    got32 = resample_x4(x, np.zeros(N, np.complex64), np.complex64, np.float32)
    e32 = np.linalg.norm(got32 - want) / np.linalg.norm(want)
    print("float64 model vs oracle Resampler: rel-RMS %.3g;  complex64 model (fp32 twiddle products): rel-RMS %.3g" % (e64, e32))
    assert e64 < 3e-7 and e32 < 1e-6
