"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and
exports every symbol include/dabgpu.h declares, and refuses to run without a
GPU (no CPU fallback).  No compute calls here."""
import os
import re

import pytest

from tests.conftest import ROOT, load_pkg


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dabgpu.h")).read()
    return sorted(set(re.findall(r"DABGPU_API[^;]*?\b(dabgpu_[a-z_0-9]+)\s*\(", text, re.S)))


def test_header_declares_the_expected_surface():
    names = declared_symbols()
    for must in ("dabgpu_create", "dabgpu_chain_process", "dabgpu_ofdm_process", "dabgpu_fir_process",
                 "dabgpu_resampler_process", "dabgpu_poly_process", "dabgpu_set_gain"):
        assert must in names
    assert len(names) >= 30


def test_library_builds_loads_and_exports_every_declared_symbol():
    pkg = load_pkg()
    pkg.build()
    lib = pkg.load_library()
    for name in declared_symbols():
        assert hasattr(lib, name), "libdabgpu.so does not export %s" % name
    assert sorted(pkg.EXPORTS) == declared_symbols()
    assert b"gfx950" in lib.dabgpu_version()


def test_every_stage_entry_point_cites_a_reference_location():
    text = open(os.path.join(ROOT, "include", "dabgpu.h")).read()
    for name in declared_symbols():
        if not name.endswith("_process"):
            continue
        i = text.index(name + "(")
        comment = text[text.rfind("/*", 0, i):i]
        assert re.search(r"src/[A-Za-z]+\.cpp:\d+", comment), name


def test_no_cpu_fallback_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    pkg = load_pkg()
    with pytest.raises(pkg.DabGpuError) as e:
        pkg.Modulator(mode=1)
    assert "no HIP device" in str(e.value) or "hip" in str(e.value).lower()


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under odr-dabmod_amd/ may import,
    link or execute it."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "odr-dabmod_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"\boracle\b|dab_oracle|liboracle|dabo_", txt):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_fir_inverse_design_is_host_logic_and_inverts_the_filter_on_the_occupied_carriers():
    """The inverse filter behind the frame kernel's equalised boundary (host code, no device): for the reference's
    default taps G H = 1 on the 1536 occupied bins to < 1e-7, the noise gain |g|_2 is about one, and a symbol that
    lives on those bins comes back from its cyclically filtered self; taps with a notch inside the band, or another
    tap count, are refused (the chain then keeps the packed dual transform)."""
    import numpy as np
    import oracle as O
    pkg = load_pkg()
    taps = O.fir_default_taps()
    ok, g, fit = pkg.fir_inverse_design(taps)
    assert ok and fit < 1e-7 and 0.9 < float(np.sqrt((g.astype(np.float64) ** 2).sum())) < 1.1
    N, K = 2048, 1536
    occ = np.r_[1:K // 2 + 1, N - K // 2:N]
    H = (taps.astype(np.float64)[None, :] * np.exp(2j * np.pi * np.outer(np.arange(N), np.arange(45)) / N)).sum(1)
    G = (g.astype(np.float64)[None, :] * np.exp(-2j * np.pi * np.outer(np.arange(N), np.arange(160) - 56) / N)).sum(1)
    assert np.abs(G[occ] * H[occ] - 1).max() < 1e-7
    rs = np.random.RandomState(5)
    X = np.zeros(N, complex)
    X[occ] = np.exp(1j * np.pi / 4 * rs.randint(0, 8, K))
    x, z = np.fft.ifft(X) * N, np.fft.ifft(X * H) * N            # z[n] = sum_j taps[j] x[n + j], cyclically
    back = sum(float(g[j]) * np.roll(z, j - 56) for j in range(160))
    assert np.abs(back - x).max() < 1e-7 * np.abs(x).max()
    notch = np.convolve(taps[:43].astype(np.float64), [1, -2 * np.cos(2 * np.pi * 300 / N), 1]).astype(np.float32)
    assert notch.size == 45 and not pkg.fir_inverse_design(notch)[0]
    # filters longer than the default are not for this kernel; shorter ones run as 45 taps with zeros behind them
    assert not pkg.fir_inverse_design(np.concatenate([taps, np.zeros(1, np.float32)]))[0]
    short = np.array([0.0, 0.0, 1.0, 0.0, 0.0], np.float32)      # the reference's doc/fir-filter/simplefiltertaps.txt
    ok5, g5, fit5 = pkg.fir_inverse_design(short)
    assert ok5 and fit5 < 1e-7


def test_counter_math_and_the_static_mix_belong_to_the_committed_sources():
    """The utilisation figures of the bench line: (i) the formulas (tools/counter_math.py) on a synthetic set of counters --
    VALU priced at 2 cycles per plain and 4 per packed wave64 instruction, FETCH_SIZE doubled, shares of a wave's life;
    (ii) profiles/isa_mix.json (the packed share and the issue-time model, tools/isa_mix.py) carries the hash of the device
    sources in the tree, i.e. it was regenerated after the last kernel change."""
    import importlib
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import counter_math as cm
    cyc = 1.0e6
    blocks = {"tf_kernel": {"GRBM_GUI_ACTIVE": 8 * cyc, "SQ_INSTS_VALU": 1024 * cyc * 0.1, "SQ_LDS_IDX_ACTIVE": 256 * cyc * 0.5,
                            "SQ_LDS_BANK_CONFLICT": 256 * cyc * 0.05, "SQ_WAVE_CYCLES": 1000.0, "SQ_ACTIVE_INST_ANY": 400.0,
                            "SQ_WAIT_INST_ANY": 350.0, "SQ_WAIT_INST_LDS": 100.0, "SQ_WAIT_ANY": 250.0,
                            "WRITE_SIZE": 1000.0, "FETCH_SIZE": 100.0, "_duration_ns": 1.0e6}}
    e = cm.figures(blocks, 1200 * 1024, 0.5)
    assert e["hbm_bytes_per_launch"] == (1000 + 2 * 100) * 1024 and e["traffic_over_algorithmic"] == 1.0
    assert abs(e["valu_busy"] - 0.1 * (2 + 2 * 0.5)) < 1e-9           # half of the instructions packed: 3 cycles on average
    assert abs(cm.figures(blocks, 0, 0.0)["valu_busy"] - 0.2) < 1e-9   # all plain: 2 cycles
    assert e["lds_busy"] == 0.5 and e["lds_bank_conflict_share"] == 0.1
    assert (e["wave_active_frac"], e["wave_issue_stall_frac"], e["wave_issue_stall_lds_frac"], e["wave_parked_frac"]) == (0.4, 0.35, 0.1, 0.25)
    assert e["effective_clock_GHz_profiled"] == 1.0 and e["dominant_kernel"] == "tf_kernel"
    mix = json.load(open(os.path.join(ROOT, "profiles", "isa_mix.json")))
    pkg = importlib.import_module("odr-dabmod_amd")
    assert mix["source_hash"] == pkg.source_hash(), "run tools/isa_mix.py --json profiles/isa_mix.json after changing a kernel"
    for wl in ("cfg2", "cfg3", "cfg4", "ifft_fir_stage"):
        m = mix[wl]
        assert m["valu_instructions"] > 100 and 0.0 <= m["packed_fraction_of_valu"] <= 1.0
        t = m["issue_model_simd_ticks"]
        assert abs(t["valu"] + t["lds"] + t["vmem"] - t["total"]) < 0.5


def test_power_probe_picks_the_devices_own_card_or_refuses(tmp_path, monkeypatch):
    """tools/power_probe.py (advisor, round 4): without a PCI mapping the probe falls back to "the card whose power rises" --
    idle reading taken when the probe is created -- and REFUSES when a second card rises by more than half as much (another
    tenant's load), instead of reporting that card's watts as the kernel's."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import power_probe as pp
    cards = []
    for i in range(3):
        d = tmp_path / ("card%d" % i)
        d.mkdir()
        (d / "power1_input").write_text("250000000")
        (d / "freq1_input").write_text("2400000000")
        (d / "power1_cap").write_text("1400000000")
        cards.append(str(d))
    monkeypatch.setattr(pp, "_hwmon_of_device", lambda i: None)
    monkeypatch.setattr(pp.glob, "glob", lambda pat: list(cards))
    probe = pp.PowerProbe(0)
    assert probe.by == "largest rise" and len(probe.cards) == 3
    state = {"n": 0}

    def busy(load):
        def f():
            state["n"] += 1
            for d, w in zip(cards, load):
                open(os.path.join(d, "power1_input"), "w").write(str(int(w * 1e6)))
            return state["n"] % 13 != 0
        return f
    r = probe.measure(busy((1350, 260, 255)), interval=0.0)
    assert r["watts_avg"] == 1350.0 and r["watts_before"] == 250.0 and r["watts_cap"] == 1400.0 and "largest rise" in r["card"]
    r = probe.measure(busy((1350, 900, 255)), interval=0.0)
    assert "error" in r and "ambiguous" in r["error"]
    # with the PCI mapping there is exactly one card and nothing to guess
    monkeypatch.setattr(pp, "_hwmon_of_device", lambda i: cards[1])
    probe = pp.PowerProbe(0)
    assert probe.cards == [cards[1]] and probe.by == "pci"
    r = probe.measure(busy((1350, 900, 255)), interval=0.0)
    assert r["watts_avg"] == 900.0


def test_hand_written_prefetch_of_the_pair_kernels_is_not_touched_before_its_wait():
    """The Mode I equalised kernels fetch the coded bits of two symbols with an inline `global_load_dword` and wait for it with an
    inline `s_waitcnt vmcnt(0)` tied to the loaded register (tf_kernel.h: fetch_pair / park_pair) -- the compiler's own wait-count
    bookkeeping does not see that load, so nothing it emits may read or move the register in between.  Checked on the device
    assembly of every instantiation (tools/check_pair_prefetch_asm.py; hipcc cross-compiles here)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_pair_prefetch_asm.py")], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "prefetch loads checked" in r.stdout and " ok" in r.stdout
