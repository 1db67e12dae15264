"""An OFDM receiver for Mode-I frames, written from ETSI EN 300 401 and independent of the oracle: the modulators
(device chain and oracle alike) are checked by decoding their output back to the coded bits."""
import numpy as np


MODES = {1: (2048, 1536, 76, 2656, 2552), 2: (512, 384, 76, 664, 638), 3: (256, 192, 153, 345, 319),
         4: (1024, 768, 76, 1328, 1276)}          # FFT size, carriers, symbols, null length, symbol length


def dab_demodulate(y, mode, early=0):
    """Receiver for one frame of the guard-interval output: strip the cyclic prefix (FFT window `early` samples before
    the end of the symbol: any window inside the cyclic extension only rotates all symbols alike), FFT, undo the
    differential modulation, the frequency interleaver (ETSI EN 300 401 14.6: pi(j) = 13 pi(j-1) + N/4 - 1 mod N) and
    the QPSK mapping -> the coded bytes ((symbols - 1) x carriers/4).  Written from the standard, independent of the oracle."""
    N, K, nsym, null, sym = MODES[mode]
    z = np.empty((nsym, K), np.complex128)
    for s in range(nsym):
        seg = y[null + s * sym: null + (s + 1) * sym]
        X = np.fft.fft(seg[sym - N - early: sym - early].astype(np.complex128))
        z[s, :K // 2] = X[1:K // 2 + 1]
        z[s, K // 2:] = X[N - K // 2:]
    d = z[1:] * np.conj(z[:-1])                                 # the data symbols
    idx, pi = [], 0
    for _ in range(1, N):
        pi = (13 * pi + N // 4 - 1) % N
        if (N - K) // 2 <= pi <= N - (N - K) // 2 and pi != N // 2:
            idx.append(pi - (1 + N // 2) if pi > N // 2 else pi + (K - N // 2))
    idx = np.array(idx)
    q = d[:, idx]                                               # carrier n of the mapper sits at position idx[n]
    ibits = (q.real < 0).astype(np.uint8).reshape(nsym - 1, K // 8, 8)
    qbits = (q.imag < 0).astype(np.uint8).reshape(nsym - 1, K // 8, 8)
    w = (1 << np.arange(7, -1, -1)).astype(np.uint16)
    blocks = np.concatenate([(ibits * w).sum(-1), (qbits * w).sum(-1)], axis=1).astype(np.uint8)
    return blocks.reshape(-1)


def dab_demodulate_mode1(y, early=0):
    return dab_demodulate(y, 1, early)
