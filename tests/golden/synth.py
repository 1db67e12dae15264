"""Deterministic, library-independent synthetic inputs shared by the golden
generator and the tests (pure integer arithmetic: splitmix64 on a counter)."""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(idx, seed):
    with np.errstate(over="ignore"):
        z = (idx.astype(np.uint64) + np.uint64(seed) * np.uint64(0x632BE59BD9B4E019)) \
            * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def synth_bits(nbytes, seed):
    """nbytes pseudo-random bytes (hot-path input: BlockPartitioner output)."""
    z = _splitmix64(np.arange(nbytes, dtype=np.uint64), seed)
    return (z >> np.uint64(56)).astype(np.uint8)


def synth_signal(nsamples, seed):
    """nsamples complex64 with re/im = int16 / 512: exactly representable,
    uniform in [-64, 64), RMS ~ 37 (an OFDM symbol's is sqrt(1536) ~ 39)."""
    z = _splitmix64(np.arange(2 * nsamples, dtype=np.uint64), seed)
    v = (z >> np.uint64(48)).astype(np.uint16).view(np.int16).astype(np.float32) / np.float32(512)
    return v.view(np.complex64).copy()


# MemlessPoly settings used for parity (a non-trivial coefficient set)
POLY_AM = [1.0, 0.05, -0.01, 0.002, 0.0]
POLY_PM = [0.0, 0.02, 0.003, 0.0, 0.0]
LUT_SCALE = np.float32(2.0 ** 32 / 1.25)


def lut_table():
    return (np.float32(1.0) + np.float32(0.0078125) * np.arange(32, dtype=np.float32)).astype(np.float32)


def format_input(nsamples, seed, fmt):
    """FormatConverter input: int16/512-grid samples scaled so that about 2 % of the
    components fall outside the target range (s16: x640 -> |x| < 40960; 8-bit: x2.5)."""
    scale = np.float32(640.0) if fmt == "s16" else np.float32(2.5)
    return (synth_signal(nsamples, seed).view(np.float32) * scale).view(np.complex64)


def format_edges(fmt):
    """Values around every range edge and the truncation-toward-zero cases."""
    if fmt == "s16":
        e = [-40000.0, -32769.0, -32768.5, -32768.0, -32767.99, -1.5, -0.99, -0.0, 0.0, 0.99, 1.5,
             32766.99, 32767.0, 32767.5, 32768.0, 40000.0]
    elif fmt == "u8":
        e = [-200.0, -129.0, -128.5, -128.0, -127.99, -127.01, -1.5, -0.99, 0.0, 0.99, 1.5, 126.99,
             127.0, 127.5, 128.0, 200.0]
    else:
        e = [-200.0, -129.0, -128.5, -128.0, -127.99, -1.5, -0.99, -0.0, 0.0, 0.99, 1.5, 126.99, 127.0,
             127.5, 128.0, 200.0]
    return np.asarray(e, np.float32)


def synth_eti(nframes, subchannels=((0, 48, 0x22),), mid=1, seed=1234, first_fct=0):
    """Raw ETI(NI) frames (src/Eti.h:50-97): nframes x 6144 bytes.
    subchannels: (SAD, STL in 64-bit words, TPL); FIC and MST payloads are pseudo-random bytes.
    Default = BASELINE config 1: one 128 kbit/s sub-channel, EEP 3-A (96 CU)."""
    fic_len = 128 if mid == 3 else 96
    out = np.full((nframes, 6144), 0x55, np.uint8)
    for n in range(nframes):
        f = out[n]
        fct = (first_fct + n) % 250
        f[0] = 0xFF
        f[1:4] = (0x07, 0x3A, 0xB6) if fct % 2 == 0 else (0xF8, 0xC5, 0x49)
        mst = sum(stl * 8 for _, stl, _ in subchannels)
        fl = (len(subchannels) * 4 + 4 + fic_len + mst) // 4          # words after FC up to EOF (not checked)
        f[4] = fct
        f[5] = 0x80 | len(subchannels)                                # FICF | NST
        f[6] = ((fct % 8) << 5) | ((mid & 3) << 3) | ((fl >> 8) & 7)  # FP | MID | FL[10:8]
        f[7] = fl & 0xFF
        p = 8
        for i, (sad, stl, tpl) in enumerate(subchannels):
            f[p] = (i << 2) | ((sad >> 8) & 3)
            f[p + 1] = sad & 0xFF
            f[p + 2] = ((tpl & 0x3F) << 2) | ((stl >> 8) & 3)
            f[p + 3] = stl & 0xFF
            p += 4
        f[p:p + 4] = (0, 0, 0, 0)                                      # EOH: MNSC, CRC (not checked)
        p += 4
        payload = synth_bits(fic_len + mst, seed=seed * 1000003 + n)
        f[p:p + fic_len + mst] = payload
        p += fic_len + mst
        f[p:p + 4] = (0, 0, 0xFF, 0xFF)                                # EOF: CRC, RFU
        f[p + 4:p + 8] = 0xFF                                          # TIST: not set
    return out
