"""the CPU front-end (odr-dabmod_amd/host/Frontend.*) that turns ETI(NI) frames into the
hot path's coded-bits input.  Pure integer work -> bit-exact against the reference's own classes: the
goldens under "frontend" in tests/golden/golden.json were produced by running the same cases
(tests/golden/frontend_cases.py) through oracle/_ref.  CPU only; runs on the GPU box as well."""
import importlib
import json
import os

import numpy as np
import pytest

import oracle as O
from tests.conftest import ROOT
from tests.golden.frontend_cases import ETI_CASES, run_cases
from tests.golden.synth import synth_eti

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))["frontend"]


@pytest.fixture(scope="module")
def fe_mod():
    m = importlib.import_module("odr-dabmod_amd.frontend")
    m.build()
    return m


@pytest.fixture(scope="module")
def product_digests(fe_mod):
    return run_cases(fe_mod.Frontend())


@pytest.mark.parametrize("key", sorted(GOLD))
def test_frontend_bit_exact_vs_reference_golden(product_digests, key):
    assert product_digests[key] == GOLD[key]


def test_known_answer_vectors(fe_mod):
    """The two short vectors the survey checked against the reference build (appendix B)."""
    fe = fe_mod.Frontend()
    assert bytes(fe.prbs(8)).hex() == "07be2e64129da3cf"
    assert bytes(fe.conv_encode(np.array([0x80, 0x01], np.uint8))).hex() == "f6dd29f00000000f6dd29f"


def test_library_exports_every_declared_symbol(fe_mod):
    import re
    hdr = open(os.path.join(ROOT, "include", "dabfrontend.h")).read()
    declared = set(re.findall(r"\b(dabfe_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(fe_mod.EXPORTS)
    lib = fe_mod.lib()
    for name in declared:
        assert hasattr(lib, name), name


def test_protection_profiles(fe_mod):
    fe = fe_mod.Frontend()
    # BASELINE config 1: 128 kbit/s, EEP 3-A -> ((6*128/8 - 3)*16, PI 8), (3*16, PI 7), 96 CU
    rules, cu, br = fe.subchannel_profile(48, 0x22)
    assert (rules, cu, br) == ([(93 * 16, 0xcccccccc), (3 * 16, 0xccccccc8)], 96, 128)
    # UEP 128 kbit/s protection level 3 (EN 300 401 table 31 / table 7): L = 11, 22, 60, 3 blocks, 96 CU;
    # the blocks cover the 24 ms frame: 4 * (11 + 22 + 60 + 3) = 384 bytes
    rules, cu, _ = fe.subchannel_profile(48, 2)
    assert [r[0] // 16 for r in rules] == [11, 22, 60, 3] and cu == 96
    assert fe.subchannel_profile(21, 0) is None            # 56 kbit/s has no level-1 profile
    assert fe.subchannel_profile(48, 0x28) is None         # unknown protection option
    assert GOLD["profiles_valid"] > 1500          # of 197 x 64 (STL, TPL) pairs


def test_start_gate_and_block_counts(fe_mod):
    """Modulation starts at FP == 0 (src/DabMod.cpp:684-693); a Mode-I block needs 4 ETI frames."""
    fe = fe_mod.Frontend()
    for first_fct, nframes, want in ((0, 8, 2), (1, 8, 0), (1, 15, 2), (245, 43, 10), (0, 3, 0)):
        blocks = fe.eti_to_bits(synth_eti(nframes, first_fct=first_fct), 1)
        assert blocks.shape[0] == want, (first_fct, nframes)
    assert fe.eti_to_bits(synth_eti(8), 1).shape[1] == O.tf_input_bytes(1)


def test_time_interleaver_is_a_16_frame_convolutional_delay(fe_mod):
    fe = fe_mod.Frontend()
    n = 64
    frames = np.zeros((40, n), np.uint8)
    frames[5] = 0xFF                                        # one frame of ones
    out = fe.time_interleave(frames)
    delays = {0: [0, 8, 4, 12, 2, 10, 6, 14], 1: [1, 9, 5, 13, 3, 11, 7, 15]}
    for j in (0, 1, 10, 63):
        for b in range(8):
            hit = [f for f in range(40) if out[f, j] & (0x80 >> b)]
            assert hit == [5 + delays[j & 1][b]]


def test_frontend_rejects_what_the_reference_throws_on(fe_mod):
    fe = fe_mod.Frontend()
    eti = synth_eti(8)
    eti[:, 5] &= 0x7F                                       # FICF = 0: "FIC must be present to modulate!"
    with pytest.raises(ValueError):
        fe.eti_to_bits(eti, 1)
    with pytest.raises(ValueError):
        fe.eti_to_bits(synth_eti(8), 5)                     # BlockPartitioner invalid mode
    with pytest.raises(ValueError):
        fe.eti_to_bits(synth_eti(8, subchannels=((0, 21, 0),)), 1)   # no UEP profile 56 kbit/s level 1


def test_frontend_refuses_a_header_whose_subchannels_overrun_the_frame(fe_mod):
    """STC entries that add up to more than the 6144-byte frame holds (2 x 1280 kbit/s here): the reader must not
    consume the following frames as payload -- it throws and resynchronises (the reference does not check)."""
    fe = fe_mod.Frontend()
    eti = synth_eti(8, subchannels=((0, 48, 0x22), (100, 48, 0x22)))
    for f in eti:
        for i in range(2):                                  # STL = 480 words = 3840 bytes each
            f[8 + 4 * i + 2] = (0x22 << 2) | (480 >> 8)
            f[8 + 4 * i + 3] = 480 & 0xFF
    with pytest.raises(ValueError):
        fe.eti_to_bits(eti, 1)
    # the same reader geometry with a layout that fits is accepted
    assert fe.eti_to_bits(synth_eti(8, subchannels=((0, 48, 0x22), (100, 48, 0x22))), 1).shape[0] == 2


def _bad_header(eti, which):
    """Make frame `which` claim 2 x 3840 bytes of sub-channel data (more than a 6144-byte frame holds)."""
    f = eti[which]
    for i in range(2):
        f[8 + 4 * i + 2] = (0x22 << 2) | (480 >> 8)
        f[8 + 4 * i + 3] = 480 & 0xFF


def test_eti_reader_relocks_after_a_refused_frame(fe_mod):
    """What happens AFTER the reader refuses a header (advisor, round 4).  (i) A frame-aligned caller -- one frame per
    call, the reference's own loop -- loses exactly the bad frame, also when the stream's ERR byte is not 0xFF (frame
    starts are taken on trust, like the reference).  (ii) A caller that feeds pieces which do not line up with frames
    loses its alignment with the dropped buffer: the reader then searches for FSYNC whatever ERR says, is not fooled by four
    payload bytes that spell a sync word (the candidate is confirmed by the OTHER sync word 6144 bytes on), consumes
    every byte it is given while it searches, and is back on the stream within two frames."""
    fe = fe_mod.Frontend()
    sub = ((0, 48, 0x22), (100, 48, 0x22))
    for err_byte in (0xFF, 0x00):
        eti = synth_eti(12, subchannels=sub)
        eti[:, 0] = err_byte                                         # ERR: error level marking of the multiplexer
        _bad_header(eti, 4)
        # (i) frame-aligned, one and three frames per call
        for piece in (6144, 3 * 6144):
            fct, errors, short = fe.eti_reader_stream(eti, piece)
            assert errors == 1 and short == 0
            # (the view notes the header parsed LAST in every call that returned; the call that threw is dropped whole)
            assert fct == ([0, 1, 2, 3, 5, 6, 7, 8, 9, 10, 11] if piece == 6144 else [2, 8, 11]), (piece, fct)
        # (ii) pieces of 20000 bytes: the piece [20000, 40000) holds the whole bad frame, the exception drops its tail, and
        # the next call starts in the middle of frame 6 -- with a false sync word planted in what is left of frame 6 (its
        # partner 6144 bytes on, in frame 7, is payload: rejected).  The reader finds frame 7 and reads on: the last headers
        # of the four calls are frames 3, (dropped), 9, 11.
        eti[6, 4000:4004] = (err_byte, 0x07, 0x3A, 0xB6)
        fct, errors, short = fe.eti_reader_stream(eti, 20000)
        assert errors == 1 and fct == [3, 9, 11], fct
        # pieces of 1000 bytes: the bad frame's end is known (it lies beyond the piece that threw), nothing but frame 4 is lost
        fct, errors, short = fe.eti_reader_stream(eti, 1000)
        assert errors == 1 and fct == list(range(12)), fct           # (frame 4's header is parsed before its layout is refused)
    # a clean stream in awkward pieces: nothing is lost, nothing is left unconsumed except across field boundaries
    clean = synth_eti(10, subchannels=sub)
    for piece in (1000, 3000, 6143):          # (at least the largest field, 768 bytes; under a frame, so every header is seen)
        fct, errors, short = fe.eti_reader_stream(clean, piece)
        assert fct == list(range(10)) and errors == 0


def test_frontend_mode_zero_is_rejected_like_dabmodulator_setmode(fe_mod):
    """DabModulator::process builds the flowgraph with setMode(dabMode), which throws for 0
    (src/DabModulator.cpp:131-133, :119-121)."""
    with pytest.raises(ValueError):
        fe_mod.Frontend().eti_to_bits(synth_eti(8), 0)


@pytest.mark.skipif(not O.have_ref(), reason="reference build (oracle/_ref) not present")
def test_frontend_vs_live_reference_on_fresh_streams(fe_mod):
    mine = fe_mod.Frontend()
    ref = fe_mod.Frontend(fe_mod.bind(O.ref(), "ref_"), "ref_")
    for seed, sub in ((91, ((0, 96, 0x21), (200, 48, 3))), (92, ((10, 18, 1), (60, 96, 0x24), (400, 60, 0x23)))):
        eti = synth_eti(36, subchannels=sub, seed=seed, first_fct=seed)
        assert np.array_equal(mine.eti_to_bits(eti, 1), ref.eti_to_bits(eti, 1))
    assert set(ETI_CASES) == {"cfg1", "multi", "mode2", "mode3", "mode4"}


# --------------------------------------------------------------------------- file layouts, the tool
HOST = os.path.join(ROOT, "odr-dabmod_amd", "host")
TOOL = os.path.join(HOST, "dabmod_file")


def _write_eti(path, eti, layout):
    """doc/README-Fileinput: raw = frames back to back; streamed = u16 length + frame (padding
    stripped); framed = u32 frame count, then as streamed."""
    with open(path, "wb") as f:
        if layout == "raw":
            f.write(eti.tobytes())
            return
        if layout == "garbage+raw":
            f.write(bytes(range(1, 201)))
            f.write(eti.tobytes())
            return
        if layout == "framed":
            f.write(np.uint32(eti.shape[0]).tobytes())
        for fr in eti:
            n = 6144
            while n > 0 and fr[n - 1] == 0x55:
                n -= 1
            f.write(np.uint16(n).tobytes())
            f.write(fr[:n].tobytes())


@pytest.mark.parametrize("layout", ["raw", "streamed", "framed", "garbage+raw"])
def test_dabmod_file_front_end_only(tmp_path, fe_mod, layout):
    """ETI file (every layout InputFileReader knows) -> hot-path input blocks, no GPU involved."""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "odr-dabmod_amd", "csrc"), "-j2"])
    subprocess.check_call(["make", "-s", "-C", HOST, "-j2"])
    c = ETI_CASES["cfg1"]
    eti = synth_eti(c["nframes"], **c["kw"])
    fin, fout = str(tmp_path / "in.eti"), str(tmp_path / "out.bits")
    _write_eti(fin, eti, layout)
    r = subprocess.run([TOOL, fin, fout, "--bits-only"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == [str(c["nframes"])] + [str(GOLD["eti_cfg1"]["blocks"])] * 2
    assert ("Input file format: " + layout.split("+")[-1]) in r.stderr
    import hashlib
    assert hashlib.sha256(open(fout, "rb").read()).hexdigest() == GOLD["eti_cfg1"]["sha256"]


def test_dabmod_file_rejects_a_file_that_is_not_eti(tmp_path):
    import subprocess
    subprocess.check_call(["make", "-s", "-C", HOST, "-j2"])
    fin = str(tmp_path / "junk.bin")
    open(fin, "wb").write(bytes(20000))
    r = subprocess.run([TOOL, fin, str(tmp_path / "o"), "--bits-only"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "cannot read" in r.stderr
