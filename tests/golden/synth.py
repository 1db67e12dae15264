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
