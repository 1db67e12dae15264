"""The dispatch matrix of the fused chain, cell by cell.

The chain from coded bits picks its kernels from the settings: transmission mode x GainControl mode x FIRFilter length class x
guard interval (copy / raised-cosine window) x crest-factor reduction x TII x output format.  `expected_kernels` below is
the RULE BOOK -- written from the documentation, not from the launcher -- and every cell of the
matrix (i) asks the library which kernels it actually launched (dabgpu_debug_last_variant) and compares, (ii) checks the
frames against the oracle.  The table of all cells goes to gpurun_out/dispatch_matrix.txt (committed as
profiles/r05_dispatch_matrix.txt); README's "which kernels run for which configuration" is generated from it
(tools/readme_dispatch.py).

Throughput shape: one workgroup per frame (chunks_per_frame = 1), two frames per call, the second call of a configuration is
the one examined (a TII configuration builds its cached null-symbol segment in the first)."""
import itertools
import os

import numpy as np
import pytest

import oracle as O
from tests.conftest import int_off_by_one_limit, ROOT, record_bound
from tests.golden.synth import synth_bits

pytestmark = pytest.mark.gpu

LOGN = {1: 11, 2: 9, 3: 8, 4: 10}
CP = {1: 504, 2: 126, 3: 63, 4: 252}           # cyclic prefix = sym_size - spacing
FMT_CODE = {"s16": 1, "u8": 2, "s8": 3}


def tf(logn, gain, guard, fir, nt, cfr=0, gvar=0, zonly=0, ofmt=0, win=0, eq=0, bits=1):
    return ("tf_kernel<logn=%d bits=%d gain=%d guard=%d fir=%d nt=%d cfr=%d gvar=%d zonly=%d ofmt=%d win=%d eq=%d>"
            % (logn, bits, gain, guard, fir, nt, cfr, gvar, zonly, ofmt, win, eq))


def expected_kernels(mode, gain, fir, overlap, cfr, tii, fmt):
    """The rule book.  gain: None | 0 fix | 1 max | 2 var.  fir: None, a tap count, or "notch" (45 taps without an inverse on
    the occupied band).  overlap: 0 = plain cyclic prefix.  Returns the kernel names in launch order."""
    logn, cp = LOGN[mode], CP[mode]
    G = int(gain is not None)
    T = 45 if fir == "notch" else fir
    F = int(fir is not None)
    tii = tii and mode in (1, 2)                         # TII exists in modes I and II only
    fits = not F or (T - 1 <= cp and T <= 128)           # the spectral FIR needs its look-ahead inside a cyclic prefix
    out = []
    if overlap == 0 and fits:
        # ---- ONE frame kernel: bits -> ... -> guard interval (-> FIRFilter)
        # every filter of up to 45 taps runs as a 45-tap filter; CFR variants keep the run-time count
        ntaps = T if (not F or cfr) else (45 if T < 45 else T)
        default_len = F and ntaps == 45 and not cfr and gain != 1      # (gain max needs the unfiltered samples)
        # boundary outputs through the taps' inverse: Mode I, and (round 6) Mode IV -- at 512 and 256 points the inverse costs
        # more than the second half of the packed transform it saves
        eq = default_len and fir != "notch" and mode in (1, 4)
        # the compile-time tap count: Mode I every form with 45 taps; modes II - IV (round 6) the plain chains without CFR
        nt = 45 if (F and ntaps == 45 and (mode == 1 or not cfr) and not (mode == 3 and gain == 1)) else 0
        default_len = default_len and (mode == 1 or eq)
        tii_inside = tii                                  # (round 5: every form of the one frame kernel adds the TII null symbol itself)
        # integer output stored by the frame kernel itself (Mode I): s16 on every form (round 5: also CFR, gain mode max, other tap
        # counts); u8 / s8 (round 5) on the no-FIRFilter and the equalised-boundary variants
        fmt_inside = fmt is not None and mode == 1 and (fmt == "s16" or (not cfr and (not F or eq)))
        s16_inside = fmt_inside
        of = FMT_CODE[fmt] if fmt_inside else 0
        if cfr:
            out.append(tf(logn, G, 1, F, nt if F else 0, cfr=1, ofmt=of))
        elif default_len:
            out.append(tf(logn, G, 1, 1, 45, ofmt=of, eq=1) if eq else tf(11, G, 1, 1, 45, zonly=1, ofmt=of))
        else:
            out.append(tf(logn, G, 1, F, nt if F else 0, ofmt=of))
        if tii and not tii_inside:
            out.append("tii_add_kernel")
        if fmt and not s16_inside:
            out.append("format_kernel<%d>" % FMT_CODE[fmt])
        return out
    # ---- a windowed guard interval, or a filter the spectral form cannot take
    fused_window = 1 <= overlap <= 128 and (overlap + (T - 1 if F else 0) <= cp) and (not F or T <= 128)
    if fused_window:
        # Mode I, a filter of up to 45 taps with an inverse, gain none / fix / var, overlap <= 10 (round 5): the equalised-boundary
        # variant carries the seam in its boundary outputs -- one transform per symbol, as without windowing
        eq_win = mode == 1 and F and fir != "notch" and T <= 45 and not cfr and gain != 1 and overlap <= 10
        if eq_win:
            # ... adds the TII segment and stores the integer formats itself, as it does without windowing
            out.append(tf(11, G, 1, 1, 45, ofmt=FMT_CODE[fmt] if fmt is not None else 0, win=1, eq=1))
            return out
        nt = 45 if (mode == 1 and F and T == 45 and not cfr) else 0
        # (the windowed default chain -- no FIRFilter -- stores s16 itself; TII is added to the complexf stream: then it cannot)
        of = 1 if (fmt == "s16" and mode == 1 and not F and not cfr and not tii) else 0
        out.append(tf(logn, G, 1, F, nt, cfr=int(cfr), ofmt=of, win=1))
        if tii:
            out.append("tii_add_kernel")
        if fmt and not of:
            out.append("format_kernel<%d>" % FMT_CODE[fmt])
        return out
    else:
        out.append(tf(logn, G, 0, 0, 0, cfr=int(cfr)))
        if F:
            out.append("guard_fir_kernel<%d>" % (48 if T <= 48 else 128 if T <= 128 else 512))
        elif overlap:
            out.append("guard_window_kernel")
        else:
            out.append("guard_copy_kernel")
    if tii:
        out.append("tii_add_kernel")
    if fmt:
        out.append("format_kernel<%d>" % FMT_CODE[fmt])
    return out


def taps_of(fir):
    """Tap sets of the FIR-length classes: up to 45 (31), the default 45, 46 ... 128 (101), beyond 128 (300), no inverse."""
    if fir is None:
        return None
    d = O.fir_default_taps().astype(np.float64)
    if fir == "notch":
        return np.convolve(d[:43], [1, -2 * np.cos(2 * np.pi * 300 / 2048), 1]).astype(np.float32)
    if fir == 45:
        return d.astype(np.float32)
    n = np.arange(fir) - (fir - 1) / 2.0
    h = 0.79 * np.sinc(0.79 * n) * np.hamming(fir)          # low-pass at 0.79 of Nyquist (810 kHz at 2.048 Msps)
    return (h / h.sum()).astype(np.float32)


def normalise_of(gain, fmt):
    if fmt is None:
        return 1.0 / 50000.0 if gain == 2 else 1.0
    # (integers worth comparing: an RMS of a few thousand for s16, of ~25 for u8; gain fix multiplies the raw symbols -- RMS 39
    # in Mode I -- by 512)
    full = 32767.0 if fmt == "s16" else 127.0
    return {None: 1.0, 0: full / 2.0e5, 1: full / 50000.0, 2: full / 50000.0}[gain]


def cells():
    out = []
    for mode in (1, 2, 3, 4):
        if mode == 1:
            gains, firs, fmts = (None, 0, 1, 2), (None, 31, 45, 101, 300, "notch"), (None, "s16", "u8")
        else:
            gains, firs, fmts = (None, 2), (None, 45, 101), (None, "s16")
        for gain, fir, overlap, cfr, tii, fmt in itertools.product(gains, firs, (0, 10), (False, True),
                                                                    (False, True) if mode in (1, 2) else (False,), fmts):
            if fir == 101 and 100 > CP[mode]:
                continue                                  # (the oracle takes any filter; the cell is the > 128 class there)
            out.append((mode, gain, fir, overlap, cfr, tii, fmt))
    return out


CELLS = cells()
_rows = []


@pytest.fixture(scope="module")
def mods(pkg):
    m = {mode: pkg.Modulator(mode=mode, max_frames=2, chunks_per_frame=1) for mode in (1, 2, 3, 4)}
    for md in m.values():
        md.trace(True)
    yield m
    for md in m.values():
        md.close()
    if _rows:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "dispatch_matrix.txt"), "w") as f:
            f.write("# mode gain fir overlap cfr tii fmt | rel-RMS vs oracle (worst frame) | kernels launched (= the rule book's)\n")
            for r in _rows:
                f.write(r + "\n")


def _cell_id(c):
    mode, gain, fir, overlap, cfr, tii, fmt = c
    return "m%d-g%s-f%s-w%d-c%d-t%d-%s" % (mode, gain, fir, overlap, cfr, tii, fmt or "cf32")


@pytest.mark.parametrize("cell", CELLS, ids=_cell_id)
def test_dispatch_cell(pkg, mods, cell):
    mode, gain, fir, overlap, cfr, tii, fmt = cell
    md = mods[mode]
    K = md.geometry["carriers"]
    norm = normalise_of(gain, fmt)
    clip = float(np.float32(50.0 * np.sqrt(K / 1536.0)))
    # ---- settings (every one, every time: the contexts are shared by the cells of a mode)
    md.set_gain(gain if gain is not None else 2, 1.0, norm, 4.0)
    taps = taps_of(fir)
    if taps is not None:
        md.set_fir_taps(taps)
    md.set_window_overlap(overlap)
    md.set_cfr(cfr, clip, 0.1)
    tii_on = tii and mode in (1, 2)
    md.set_tii(tii_on, 3, 5, False) if mode in (1, 2) else None
    md.set_output_format(fmt)
    stages = (pkg.STAGE_GAIN if gain is not None else 0) | (pkg.STAGE_FIR if fir is not None else 0)
    per = md.geometry["tf_input_bytes"]
    bits = np.stack([synth_bits(per, seed=7000 + i) for i in range(2)])
    md.chain(bits, stages)                                   # (builds a TII segment where there is one; frame parity back to even)
    y = md.chain(bits, stages)
    got = md.last_variant()
    want = expected_kernels(mode, gain, fir, overlap, cfr, tii, fmt)
    # ---- the frames against the oracle
    kw = dict(mode=mode, stages=stages, window_overlap=overlap)
    if gain is not None:
        kw.update(gain_mode=gain, normalise=norm)
    if taps is not None:
        kw.update(taps=taps)
    if cfr:
        kw.update(cfr=(clip, 0.1))
    if tii_on:
        kw.update(tii=(3, 5, False))
    ref = O.Chain(**kw).process(bits)
    if fmt is None:
        err = max(np.linalg.norm(y[f] - ref[f]) / np.linalg.norm(ref[f]) for f in range(2))
        ok = err < 1e-6
        shown = "%.2e" % err
    else:
        wantq, _ = O.format_convert(ref, fmt)
        d = np.abs(y.reshape(-1).astype(np.int32) - wantq.reshape(-1).astype(np.int32))
        err = float((d != 0).mean())
        # the integers of two fp32 chains: at most one step apart, and as rarely as tests/conftest.py::int_off_by_one_limit says
        ok = d.max() <= 1 and record_bound("dispatch matrix: integer components one step from the reference's, %s" %
                                           "mode %d %s %s ov %d cfr %d tii %d %s" % (mode, gain, fir, overlap, cfr, tii_on, fmt),
                                           err, int_off_by_one_limit(wantq, 1 << LOGN[mode], fmt))
        shown = "%.1e of the components one step off" % err
    _rows.append("%d %-4s %-5s %3d %d %d %-4s | %s | %s" % (mode, gain, fir, overlap, cfr, tii_on, fmt or "cf32", shown, "; ".join(got)))
    assert got == want, "kernels launched %s, the rule book says %s" % (got, want)
    assert ok, shown
