#!/usr/bin/env python3
"""Feasibility study (numpy): the FIR boundary samples of the frame kernel from the FILTERED symbols alone.
d[m] = x_cur[N-cp+m] - x_prev[m] = sum_j g[j] w[m-j],  w[q] = z_cur[N-cp+q] - z_prev[q mod N],  G H = 1 on the occupied bins."""
import sys, numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import oracle as O
N, K, cp, C = 2048, 1536, 504, 44
taps = O.fir_default_taps().astype(np.float64)
kk = np.arange(N)
H = (taps[None, :] * np.exp(2j * np.pi * np.outer(kk, np.arange(45)) / N)).sum(1)
occ = np.r_[1:K // 2 + 1, N - K // 2:N]

def design(L, c, edge=900, lam=1e-6):
    j = np.arange(L) - c
    stop = np.r_[edge:N - edge + 1]
    A = np.exp(-2j * np.pi * np.outer(occ, j) / N)
    S = np.sqrt(lam) * np.exp(-2j * np.pi * np.outer(stop, j) / N)
    M = np.vstack([A.real, A.imag, S.real, S.imag])
    b = np.r_[(1 / H[occ]).real, (1 / H[occ]).imag, np.zeros(2 * len(stop))]
    g = np.linalg.lstsq(M, b, rcond=None)[0]
    return g, np.abs(A @ g - 1 / H[occ]).max()

rs = np.random.RandomState(1)
nsym = 12
X = np.zeros((nsym, N), complex)
ph = rs.randint(0, 4, (nsym, K))
X[:, occ] = np.exp(1j * (np.pi / 4 + np.pi / 2 * ph)) * 39.0 / np.sqrt(K) * 50  # arbitrary level
x = np.fft.ifft(X, axis=1) * N
z = np.fft.ifft(X * H[None, :], axis=1) * N
noise = lambda a: a * (1 + 1.2e-7 * (rs.randn(*a.shape) + 1j * rs.randn(*a.shape)) / np.sqrt(2))
x32 = noise(x).astype(np.complex64)
z32 = noise(z).astype(np.complex64)
# true boundary outputs between symbol s-1 and s: positions n in [N-44, N) of symbol s-1
def truth(s):
    u = np.concatenate([x[s - 1], x[s][N - cp:N - cp + C + 1]])
    return np.array([np.dot(taps, u[n:n + 45]) for n in range(N - C, N)])
ymax = np.abs(z).max()
for L, c in ((96, 26), (128, 40), (160, 56), (192, 72)):
    g, fit = design(L, c)
    g32 = g.astype(np.float32)
    errs, errs_now = [], []
    for s in range(1, nsym):
        q = np.arange(-(L - 1 - c), C + c)
        w = (z32[s][(N - cp + q) % N] - z32[s - 1][q % N]).astype(np.complex64)
        d = np.zeros(C, np.complex64)
        for m in range(C):
            idx = (m - (np.arange(L) - c)) - q[0]
            d[m] = np.sum((g32 * w[idx]).astype(np.complex64), dtype=np.complex64)
        y = np.zeros(C, np.complex64)
        for i in range(C):
            n = N - C + i
            acc = z32[s - 1][n]
            for j in range(N - n, 45):
                acc = np.complex64(acc + np.float32(taps[j]) * d[n + j - N])
            y[i] = acc
        errs.append(np.abs(y - truth(s)).max())
        # today's method: FIR over the fp32 unfiltered samples
        u = np.concatenate([x32[s - 1][N - C:], x32[s][N - cp:N - cp + C + 1]]).astype(np.complex64)
        yn = np.array([np.sum((taps.astype(np.float32) * u[i:i + 45]).astype(np.complex64), dtype=np.complex64) for i in range(C)])
        errs_now.append(np.abs(yn - truth(s)).max())
    print("L=%3d c=%3d  fit %.1e  |g|2 %.2f  sum|g| %.2f  max err / max|z|: new %.2e  today %.2e"
          % (L, c, fit, np.sqrt((g ** 2).sum()), np.abs(g).sum(), max(errs) / ymax, max(errs_now) / ymax))

# ---- design through the (Toeplitz) normal equations, as the host code would do it ----
def design_ne(L, c, edge=900, lam=1e-6, mu=1e-9, dtype=np.float64):
    stop = np.r_[edge:N - edge + 1]
    th_o, th_s = 2 * np.pi * occ / N, 2 * np.pi * stop / N
    d = np.arange(L)
    r = np.cos(np.outer(d, th_o)).sum(1) + lam * np.cos(np.outer(d, th_s)).sum(1)
    R = r[np.abs(d[:, None] - d[None, :])].astype(dtype) + mu * N * np.eye(L)
    jp = d - c
    T = 1 / H[occ]
    p = (np.cos(np.outer(jp, th_o)) * T.real[None, :] - np.sin(np.outer(jp, th_o)) * T.imag[None, :]).sum(1).astype(dtype)
    Lc = np.linalg.cholesky(R)
    g = np.linalg.solve(Lc.T, np.linalg.solve(Lc, p))
    A = np.exp(-2j * np.pi * np.outer(occ, jp) / N)
    return g, np.abs(A @ g - T).max(), np.linalg.cond(R.astype(np.float64))
print("normal equations:")
for L, c in ((128, 40), (160, 56), (192, 72)):
    for lam, mu in ((1e-6, 1e-10), (1e-6, 1e-9), (1e-6, 1e-8), (1e-4, 1e-9)):
        g, fit, cond = design_ne(L, c, lam=lam, mu=mu)
        g2, fit2 = design(L, c, lam=lam)
        print("L=%d mu=%.0e lam=%.0e: fit %.2e (lstsq %.2e) cond %.1e |g|2 %.3f max|g-g2| %.1e"
              % (L, mu, lam, fit, fit2, cond, np.sqrt((g ** 2).sum()), np.abs(g - g2).max()))
