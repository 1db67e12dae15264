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
