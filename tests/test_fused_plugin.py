"""The PRODUCTION plugin as a drop-in: DabGpuChain with the reference's
remote-control surface and metadata semantics, (1) inside this repository's host mirror (host_selftest chainrc) and
(2) inside the reference's OWN graph builder: oracle/_ref/dabmod_fused is the reference's DabModulator.cpp after
odr-dabmod_amd/host/install_fused.sh's scripted edit, its Flowgraph.cpp, OutputMemory.cpp, lib/RemoteControl.cpp ...
compiled from where they lie (oracle/Makefile `fused`), linked with GpuStages.cpp and libdabgpu.so."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from tests.conftest import ROOT, int_off_by_one_limit

HOST = os.path.join(ROOT, "odr-dabmod_amd", "host")
BIN = os.path.join(HOST, "host_selftest")
FUSED = os.path.join(ROOT, "oracle", "_ref", "dabmod_fused")
DROPIN = os.path.join(ROOT, "oracle", "_ref", "dabmod_dropin")
POLY_AM = (1.0, 0.05, -0.01, 0.002, 0.0)
POLY_PM = (0.0, 0.02, 0.003, 0.0, 0.0)
POLY2_AM = (0.9, 0.08, -0.02, 0.001, 0.0005)
POLY2_PM = (0.01, 0.03, -0.004, 0.0, 0.0)
NORM = 1.0 / 50000.0
have_ref = pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference tree is not on this machine")
have_fused = pytest.mark.skipif(not (os.path.exists(FUSED) and os.path.exists(DROPIN)),
                                reason="oracle/_ref/dabmod_fused is built where the reference tree is")


def build_host():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "odr-dabmod_amd", "csrc"), "-j2"])
    subprocess.check_call(["make", "-s", "-C", HOST, "-j2"])


def rel_rms(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


# ------------------------------------------------------------------ CPU: the scripted edit
@have_ref
def test_install_fused_edit_passes_the_reference_compiler(tmp_path):
    """A copy of the reference's src/, install_dropins.sh, then install_fused.sh: the edited DabModulator.cpp compiles
    against the reference's own headers; nothing but DabModulator.cpp/.h and ConfigParser.h differs from the original
    tree (besides the fifteen forwarding headers); the edit adds only (no line of the original is altered outside the
    two replaced blocks and the three m_formatConverter reads)."""
    src = tmp_path / "src"
    shutil.copytree("/root/reference/src", str(src))
    for script in ("install_dropins.sh", "install_fused.sh"):
        r = subprocess.run(["sh", os.path.join(HOST, script), str(src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    text = (src / "DabModulator.cpp").read_text()
    assert "make_shared<DabGpuChain>(gs, live)" in text and "rcs.enrol(controllable)" in text
    for gone in ("make_shared<QpskSymbolMapper>", "make_shared<GuardIntervalInserter>", "make_shared<FormatConverter>",
                 "for (auto& p : plugins)"):
        assert gone not in text, gone
    assert "cifPart, m_gpuChain" in text and "m_gpuChain, m_output" in text
    # the front half of the graph builder is untouched
    orig = open("/root/reference/src/DabModulator.cpp").read()
    a = orig.index("auto cifPrbs"); b = orig.index("const bool fixedPoint")
    assert orig[a:b] in text
    a = orig.index("m_flowgraph->connect(cifPrbs, cifMux);"); b = orig.index("m_flowgraph->connect(cifPart, cifMap);")
    assert orig[a:b] in text
    for f in ("DabModulator.cpp",):
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-DPACKAGE_NAME=\"odr-dabmod\"",
                            "-DPACKAGE_VERSION=\"3.0.1\"", "-DVERSION=\"3.0.1\"", "-I.", "-I/root/reference/lib",
                            "-I/root/reference", "-I/root/reference/kiss", "-I" + os.path.join(ROOT, "include"), f],
                           cwd=str(src), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, f + ":\n" + r.stderr[-3000:]


@have_ref
def test_install_fused_refuses_a_tree_it_does_not_recognise(tmp_path):
    """An anchor that is missing (another upstream version) stops the script; no half-edited file is left behind."""
    src = tmp_path / "src"
    shutil.copytree("/root/reference/src", str(src))
    subprocess.check_call(["sh", os.path.join(HOST, "install_dropins.sh"), str(src)])
    p = src / "DabModulator.cpp"
    p.write_text(p.read_text().replace("m_flowgraph->connect(cifPart, cifMap);", "m_flowgraph->connect(cifPart, cifMapper);"))
    before = p.read_text()
    r = subprocess.run(["sh", os.path.join(HOST, "install_fused.sh"), str(src)], capture_output=True, text=True)
    assert r.returncode != 0 and "anchor" in r.stderr
    assert "DabGpuChain" not in before


# ------------------------------------------------------------------ GPU: host mirror (no reference needed)
def _chainrc(tmp_path, bits, drops, actions, poly=None):
    build_host()
    fbits, fout = str(tmp_path / "bits.bin"), str(tmp_path / "out.iq")
    bits.tofile(fbits)
    coef = tmp_path / "poly.coef"
    if poly is not None:
        import oracle as O
        O.write_poly_file(str(coef), *poly)
    elif coef.exists():
        coef.unlink()
    r = subprocess.run([BIN, "chainrc", fbits, str(bits.shape[0]), fout, str(drops)] + actions,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout, np.fromfile(fout, dtype=np.complex64)


@pytest.mark.gpu
def test_chain_remote_control_changes_the_next_frame_like_a_fresh_context(tmp_path):
    """`set gain digital`, `set gain mode`, `set firfilter tapsfile`, `set memlesspoly coefs` between two frames: every
    frame equals the oracle chain configured with the values in force when the frame went in (the reference applies a
    parameter at the next frame its stage processes: src/GainControl.cpp:96-116, src/FIRFilter.cpp:166,
    src/MemlessPoly.cpp:360-372) -- and the values land in the caller's settings (the mod_settings_t fields of the reference)."""
    import oracle as O
    from tests.golden.synth import synth_bits
    n = 8
    bits = np.stack([synth_bits(28800, seed=900 + i) for i in range(n)])
    taps2 = np.hamming(31).astype(np.float32)
    taps2 /= taps2.sum()
    ftaps = tmp_path / "taps31.txt"
    ftaps.write_text("31\n" + "".join("%r\n" % float(t) for t in taps2))
    fcoef2 = tmp_path / "poly2.txt"
    O.write_poly_file(str(fcoef2), POLY2_AM, POLY2_PM)
    actions = ["gain,digital,0.5@2", "firfilter,tapsfile,%s@4" % ftaps, "memlesspoly,coefs,<%s@5" % fcoef2, "gain,mode,max@6"]
    out, got = _chainrc(tmp_path, bits, 0, actions, poly=(POLY_AM, POLY_PM))
    assert "controllables: firfilter gain guardinterval memlesspoly ofdm tii" in out
    assert "rc gain digital = 0.500000" in out and "rc gain mode = max" in out and "rc firfilter tapsfile = %s" % ftaps in out
    assert "settings now: digital=0.5 mode=1 taps=%s" % ftaps in out
    assert "chainrc: %d frames written" % n in out
    tf = 196608
    got = got.reshape(n, tf)
    stages = O.STAGE_GAIN | O.STAGE_FIR | O.STAGE_POLY
    segs = [(0, 2, dict(dig_gain=1.0, gain_mode=2, taps=None, am=POLY_AM, pm=POLY_PM)),
            (2, 4, dict(dig_gain=0.5, gain_mode=2, taps=None, am=POLY_AM, pm=POLY_PM)),
            (4, 5, dict(dig_gain=0.5, gain_mode=2, taps=taps2, am=POLY_AM, pm=POLY_PM)),
            (5, 6, dict(dig_gain=0.5, gain_mode=2, taps=taps2, am=POLY2_AM, pm=POLY2_PM)),
            (6, 8, dict(dig_gain=0.5, gain_mode=1, taps=taps2, am=POLY2_AM, pm=POLY2_PM))]
    for a, b, kw in segs:
        ref = O.Chain(mode=1, stages=stages, normalise=NORM, **kw).process(bits[a:b])
        for f in range(a, b):
            assert rel_rms(got[f], ref[f - a]) < 1e-6, (f, kw)
    # the written-back coefficient file is the value as received (src/MemlessPoly.cpp:431-437)
    assert (tmp_path / "poly.coef").read_text() == fcoef2.read_text()


@pytest.mark.gpu
@pytest.mark.parametrize("drops", [0, 1, 3])
def test_chain_metadata_leaves_with_its_frame(tmp_path, drops):
    """DabGpuChain is a ModMetadata: with emulatePipelineDrops = k the frame of call i - k AND its metadata leave on call
    i (the reference's k pipelined stages delay both, src/ModPlugin.cpp:90-128); Flowgraph would otherwise hand the
    sink call i's metadata with frame i - k (src/Flowgraph.cpp:157-175)."""
    import oracle as O
    from tests.golden.synth import synth_bits
    n = 7
    bits = np.stack([synth_bits(28800, seed=950 + i) for i in range(n)])
    out, got = _chainrc(tmp_path, bits, drops, [])
    lines = re.findall(r"frame (\d+) carries metadata of (\d+)", out)
    assert lines == [(str(k), str(k)) for k in range(n - drops)], out
    ref = O.Chain(mode=1, stages=O.STAGE_GAIN | O.STAGE_FIR, normalise=NORM).process(bits)
    got = got.reshape(n - drops, -1)
    for f in range(n - drops):
        assert rel_rms(got[f], ref[f]) < 1e-6


# ------------------------------------------------------------------ GPU: inside the reference's own graph builder
def _eti(tmp_path, n_eti=32):
    import importlib
    from tests.golden.synth import synth_eti
    fe_mod = importlib.import_module("odr-dabmod_amd.frontend")
    eti = synth_eti(n_eti)
    fin = str(tmp_path / "in.eti")
    eti.tofile(fin)
    return fin, fe_mod.Frontend().eti_to_bits(eti, 1)


def _run(tool, fin, fout, args):
    r = subprocess.run([tool, fin, fout] + args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    return r


def _cfg(tmp_path, cfg):
    import oracle as O
    if cfg == "cfg1":
        return [], 1, None, O.Chain(mode=1, stages=O.STAGE_GAIN, gain_mode=2, normalise=1.0)
    if cfg == "cfg3":
        return (["--fir", "default", "--normalise", repr(NORM)], 2, None,
                O.Chain(mode=1, stages=O.STAGE_GAIN | O.STAGE_FIR, gain_mode=2, normalise=NORM))
    if cfg == "cfg4":
        coef = tmp_path / "poly.coef"
        O.write_poly_file(str(coef), POLY_AM, POLY_PM)
        return (["--fir", "default", "--normalise", repr(NORM), "--rate", "8192000", "--poly", str(coef)], 3, None,
                O.Chain(mode=1, stages=O.STAGE_GAIN | O.STAGE_FIR | O.STAGE_RESAMPLE | O.STAGE_POLY, gain_mode=2,
                        normalise=NORM, out_rate=8192000, am=POLY_AM, pm=POLY_PM))
    return (["--window", "10", "--gainmode", "max", "--normalise", repr(32767.0 / 50000.0), "--format", "s16"], 1, "s16",
            O.Chain(mode=1, stages=O.STAGE_GAIN, gain_mode=1, normalise=32767.0 / 50000.0, window_overlap=10))


@pytest.mark.gpu
@have_fused
@pytest.mark.parametrize("cfg", ["cfg1", "cfg3", "cfg4", "window_s16"])
def test_fused_plugin_inside_the_reference_graph_builder(tmp_path, cfg):
    """dabmod_fused -- the reference's DabModulator::process with ONE DabGpuChain node -- modulates an ETI file on the
    MI355X: every frame within the chain bars of the oracle and of what the per-stage drop-ins (dabmod_dropin, the
    reference's unmodified graph builder) write; with gpuReferenceLatency the frame COUNT is the reference's too
    (N/4 - 1, - 2, - 3), and those frames are byte for byte the first frames of the run without it."""
    import oracle as O
    fin, bits = _eti(tmp_path)
    n = bits.shape[0]
    args, drops, fmt, chain = _cfg(tmp_path, cfg)
    f_all, f_lat, f_drop = (str(tmp_path / x) for x in ("all.iq", "lat.iq", "dropin.iq"))
    _run(FUSED, fin, f_all, args)
    _run(FUSED, fin, f_lat, args + ["--reference-latency", "1"])
    _run(DROPIN, fin, f_drop, args)
    ref = chain.process(bits)
    if fmt is None:
        got = np.fromfile(f_all, dtype=np.complex64).reshape(-1, ref.shape[1])
        lat = np.fromfile(f_lat, dtype=np.complex64).reshape(-1, ref.shape[1])
        per_stage = np.fromfile(f_drop, dtype=np.complex64).reshape(-1, ref.shape[1])
        assert got.shape[0] == n and lat.shape[0] == n - drops == per_stage.shape[0]
        assert np.array_equal(lat, got[:n - drops])
        for f in range(n):
            assert rel_rms(got[f], ref[f]) < 1e-6, f
        for f in range(n - drops):
            assert rel_rms(got[f], per_stage[f]) < 1e-6, f
    else:
        want, _ = O.format_convert(ref, fmt)
        got = np.fromfile(f_all, dtype=np.int16)
        lat = np.fromfile(f_lat, dtype=np.int16)
        per_stage = np.fromfile(f_drop, dtype=np.int16)
        per = want.size // n
        assert got.size == want.size and lat.size == (n - drops) * per == per_stage.size
        assert np.array_equal(lat, got[:lat.size])
        for other in (want.reshape(-1), ):
            d = np.abs(got.astype(np.int32) - other.astype(np.int32))
            assert d.max() <= 1 and (d != 0).mean() < int_off_by_one_limit(want)
        d = np.abs(got[:per_stage.size].astype(np.int32) - per_stage.astype(np.int32))
        assert d.max() <= 1 and (d != 0).mean() < int_off_by_one_limit(want)


@pytest.mark.gpu
@have_fused
@pytest.mark.parametrize("cfg", ["cfg1", "cfg3"])
def test_fused_plugin_with_the_reference_gain_rounding(tmp_path, cfg):
    """mod_settings_t::gpuReferenceGain -> DabGpuChain::Settings::referenceGainRounding -> dabgpu_set_gain_rounding: the one
    fused node then forms the var-gain multipliers by the reference's recurrence (src/GainControl.cpp:251-340), as the
    per-stage GainControl drop-in inside dabmod_dropin does -- same transform, same recurrence, same filter arithmetic: the
    two graphs write the same bytes, and both are closer to the oracle than the fused default (exact variance)."""
    from tests.conftest import record_bound
    fin, bits = _eti(tmp_path)
    n = bits.shape[0]
    args, drops, _, chain = _cfg(tmp_path, cfg)
    f_ref, f_def, f_drop = (str(tmp_path / x) for x in ("refgain.iq", "default.iq", "dropin.iq"))
    _run(FUSED, fin, f_ref, args + ["--reference-gain", "1"])
    _run(FUSED, fin, f_def, args)
    _run(DROPIN, fin, f_drop, args)
    ref = chain.process(bits)
    a = np.fromfile(f_ref, dtype=np.complex64).reshape(-1, ref.shape[1])
    b = np.fromfile(f_def, dtype=np.complex64).reshape(-1, ref.shape[1])
    per_stage = np.fromfile(f_drop, dtype=np.complex64).reshape(-1, ref.shape[1])
    assert a.shape[0] == n == b.shape[0] and per_stage.shape[0] == n - drops
    assert np.array_equal(a[:n - drops].view(np.uint32), per_stage.view(np.uint32))
    ea, eb = (np.abs(y - ref).max() / np.abs(ref).max() for y in (a, b))
    assert record_bound("dabmod_fused %s, gain rounding REFERENCE, max-abs / |out|_inf against the oracle" % cfg, ea, 4e-7)
    assert record_bound("dabmod_fused %s, exact variance, max-abs / |out|_inf against the oracle" % cfg, eb, 8e-7)
    assert ea < eb


@pytest.mark.gpu
@have_fused
def test_fused_plugin_remote_control_through_the_reference_registry(tmp_path):
    """rcs.set_param("gain", "digital", ...) etc. -- the reference's own RemoteControllers (lib/RemoteControl.cpp) on
    the controllables DabGpuChain enrolled -- between two transmission frames: the frames before the change are byte
    for byte those of a run that never changed anything, the frames from the change on byte for byte those of a run
    STARTED with the new values (a fresh context; with the Resampler in the chain, from the third output hop on).  gain digital / mode / var, firfilter tapsfile, memlesspoly coefs and
    coeffile, guardinterval windowlen, ofdm cfr / clip / errorclip, tii enable / comb / pattern."""
    import oracle as O
    fin, bits = _eti(tmp_path, 40)
    n = bits.shape[0]                                        # 10 transmission frames; changes before ETI frame 16 = TF 4
    coef1, coef2 = str(tmp_path / "p1.coef"), str(tmp_path / "p2.coef")
    O.write_poly_file(coef1, POLY_AM, POLY_PM)
    O.write_poly_file(coef2, POLY2_AM, POLY2_PM)
    taps2 = np.hamming(31).astype(np.float32)
    taps2 /= taps2.sum()
    ftaps = str(tmp_path / "taps31.txt")
    open(ftaps, "w").write("31\n" + "".join("%r\n" % float(t) for t in taps2))
    base = ["--fir", "default", "--normalise", repr(NORM), "--rate", "8192000"]
    cases = [
        # (name, options of the unchanged run, remote-control actions, options of the run started with the new values)
        ("gain", base + ["--poly", coef1], [("gain", "digital", "0.5"), ("gain", "mode", "max")],
         base + ["--poly", coef1, "--digital", "0.5", "--gainmode", "max"]),
        ("gainvar", base, [("gain", "var", "3.0")], base + ["--var", "3.0"]),
        ("fir", base + ["--poly", coef1], [("firfilter", "tapsfile", ftaps)], ["--fir", ftaps] + base[2:] + ["--poly", coef1]),
        ("coefs", base + ["--poly", coef1], [("memlesspoly", "coefs", "@" + coef2)], base + ["--poly", coef2]),
        ("coeffile", base + ["--poly", coef1], [("memlesspoly", "coeffile", coef2)], base + ["--poly", coef2]),
        ("window", ["--normalise", repr(NORM)], [("guardinterval", "windowlen", "12")], ["--normalise", repr(NORM), "--window", "12"]),
        ("cfr", ["--normalise", repr(NORM)], [("ofdm", "clip", "40"), ("ofdm", "errorclip", "0.2"), ("ofdm", "cfr", "1")],
         ["--normalise", repr(NORM), "--cfr", "40,0.2"]),
        ("tii", ["--fir", "default", "--normalise", repr(NORM)], [("tii", "comb", "5"), ("tii", "pattern", "11"), ("tii", "enable", "1")],
         ["--fir", "default", "--normalise", repr(NORM), "--tii", "5,11"]),
        # with the reference's gain recurrence inside the chain (gpuReferenceGain): the split path follows the remote control as well
        ("refgain_var", ["--fir", "default", "--normalise", repr(NORM), "--reference-gain", "1"], [("gain", "var", "3.0"), ("gain", "digital", "0.7")],
         ["--fir", "default", "--normalise", repr(NORM), "--reference-gain", "1", "--var", "3.0", "--digital", "0.7"]),
        ("refgain_mode", ["--fir", "default", "--normalise", repr(NORM), "--reference-gain", "1", "--gainmode", "fix"], [("gain", "mode", "var")],
         ["--fir", "default", "--normalise", repr(NORM), "--reference-gain", "1"]),
    ]
    at = 16
    for name, opts0, actions, opts1 in cases:
        if name in ("coefs", "coeffile"):
            O.write_poly_file(coef1, POLY_AM, POLY_PM)       # (an earlier `coefs` wrote the new set back into the file)
        f0, f1, fc = (str(tmp_path / ("%s_%s.iq" % (name, x))) for x in ("old", "new", "changed"))
        _run(FUSED, fin, f0, opts0)
        _run(FUSED, fin, f1, opts1)
        rc = []
        for a in actions:
            rc += ["--rc", "%d,%s,%s,%s" % (at, a[0], a[1], a[2])]
        r = _run(FUSED, fin, fc, opts0 + rc)
        for a in actions:
            assert re.search(r"^rc %s %s = \S" % (a[0], a[1]), r.stderr, re.M), (name, r.stderr[-600:])
        old, new, changed = (np.fromfile(f, dtype=np.uint8) for f in (f0, f1, fc))
        per = old.size // n
        assert old.size == new.size == changed.size == n * per, name
        k = at // 4
        assert np.array_equal(changed[:k * per], old[:k * per]), name
        assert not np.array_equal(old[k * per:], new[k * per:]), name            # (the change is a change)
        # The Resampler carries two hops of its INPUT from frame to frame (src/Resampler.cpp:142-192: prevIn and the overlap
        # tail), here as in the reference: the first two output hops of the first frame after the change still see the old
        # stream.  Everything behind them is the fresh run's, byte for byte.
        skip = 3 * 8192 * 8 if "--rate" in opts0 else 0
        assert np.array_equal(changed[k * per + skip:], new[k * per + skip:]), name
        if skip:
            a = changed[k * per:k * per + skip].view(np.complex64)
            b = new[k * per:k * per + skip].view(np.complex64)
            assert np.array_equal(a[2 * 8192:], b[2 * 8192:]), name


@pytest.mark.gpu
@have_fused
@pytest.mark.parametrize("latency", [0, 1])
def test_fused_plugin_metadata_inside_the_reference_flowgraph(tmp_path, latency):
    """The reference's Flowgraph (src/Flowgraph.cpp:146-175) moves BlockPartitioner's four timestamps per transmission
    frame along the edges; the sink must see frame K together with the frame counts 4K .. 4K + 3 -- what the per-stage
    graph (three PipelinedModCodecs, src/ModPlugin.cpp:117-128) delivers -- also when the fused plugin holds frames back."""
    fin, bits = _eti(tmp_path, 40)
    args = ["--fir", "default", "--normalise", repr(NORM), "--show-metadata", "1"]
    r = _run(FUSED, fin, str(tmp_path / "f.iq"), args + (["--reference-latency", "1"] if latency else []))
    n = bits.shape[0] - (2 if latency else 0)
    want = ["meta %d: %d %d %d %d" % (k, 4 * k, 4 * k + 1, 4 * k + 2, 4 * k + 3) for k in range(n)]
    assert [l for l in r.stdout.splitlines() if l.startswith("meta")] == want
    if latency:
        d = _run(DROPIN, fin, str(tmp_path / "d.iq"), args)
        assert [l for l in d.stdout.splitlines() if l.startswith("meta")] == want
