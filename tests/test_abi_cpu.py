"""CPU tests of the boundary: the C-ABI library builds, loads, exports every symbol that
include/sdr_hip.h declares, and its host-only entry points (filter design, error paths that
need no device) behave.  No compute call is made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from sdr_amd import capi
    capi.build()
    return capi


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sdr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sdr_[a-z0-9_]+)\s*\(", text)))


def test_exports_every_declared_symbol(capi):
    L = capi.lib()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"{n} declared in sdr_hip.h but not exported"


def test_strerror(capi):
    L = capi.lib()
    assert L.sdr_strerror(0) == b"ok"
    assert b"invalid" in L.sdr_strerror(capi.EINVAL)


def test_designer_matches_oracle_definition(capi, oracle):
    # host code, no device.  PARITY UNPINNED vs the reference (DSP::create_fir_filter is absent);
    # product and oracle state the same documented definition.
    for name in capi.WINDOWS:
        for n in (1, 2, 64, 101):
            assert np.allclose(capi.window(name, n), oracle.window(name, n), rtol=0, atol=1e-15)
    for n, fc in ((64, 0.1), (101, 0.1), (200, 20800 / 62400.0), (195, 2400 / 62400.0), (511, 0.05)):
        assert np.allclose(capi.create_fir_filter(n, fc, "hann"),
                           oracle.create_fir_filter(n, fc, "hann"), rtol=0, atol=1e-16)


def test_bad_arguments_without_device(capi):
    L = capi.lib()
    assert L.sdr_window(C.c_int(99), C.c_size_t(8), (C.c_double * 8)()) == capi.EINVAL
    assert L.sdr_create_fir_filter(C.c_size_t(0), C.c_double(0.1), C.c_int(1), None) == capi.EINVAL
    # the reference throws std::invalid_argument for D <= 0 (bladerf_apt_demod.cpp:254-255)
    taps = np.ones(4)
    d = capi.FirDesc(0, 1, 0, 4, taps.ctypes.data_as(C.POINTER(C.c_double)), 1.0, 0.0, 0, 0, 0)
    h = C.c_void_p()
    assert L.sdr_fir_plan_create(C.byref(d), C.byref(h)) == capi.EINVAL
    assert b"positive" in L.sdr_last_error()
    assert L.sdr_fir_plan_destroy(None) == capi.OK


def test_no_cpu_fallback(capi):
    """Without a GPU the product path must fail loudly, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.SdrError) as e:
        capi.FirPlan(np.ones(8), 2)
    assert e.value.status in (capi.ENODEV, capi.EHIP)


def test_product_sources_never_reference_the_oracle():
    """the product path (sdr_amd/, include/) must not import, load or link the CPU checker: no mention of
    oracle/, its libraries or dlopen in any shipped source (only tests/, smoke() and bench.py's cpu legs may)"""
    import re
    bad = []
    pat = re.compile(r"oracle|liboracle|libsdrref|dlopen|LoadLibrary", re.I)
    for top in ("sdr_amd", "include"):
        for dirpath, dirnames, files in os.walk(os.path.join(ROOT, top)):
            dirnames[:] = [d for d in dirnames if d not in ("build", "__pycache__")]
            for f in files:
                if not f.endswith((".py", ".h", ".hpp", ".hip", ".cpp", ".c")):
                    continue
                path = os.path.join(dirpath, f)
                for i, line in enumerate(open(path, errors="replace"), 1):
                    if pat.search(line):
                        bad.append(f"{os.path.relpath(path, ROOT)}:{i}: {line.strip()[:100]}")
    assert not bad, "product sources mention the oracle:\n" + "\n".join(bad)


def test_product_build_is_not_a_measurement_build(capi):
    """kernel ablation switches exist only behind -DSDR_MEASUREMENT_BUILD, which capi.build() never sets"""
    assert capi.build_info().startswith("libsdrhip product build")
    src = open(os.path.join(ROOT, "sdr_amd", "capi.py")).read()
    assert "-DSDR_MEASUREMENT_BUILD" not in src and "-DSDR_I8_ABLATE" not in src and "-DSDR_FFT_ABLATE" not in src


def test_apt_image_state_stays_bounded_on_constant_input(capi):
    """a long run of constant envelope blocks (squelched input) finds no line and must not
    grow the pending backlog without bound (host code, no device needed)"""
    img = capi.AptImage(width=2080, rows=8)
    blk = np.full(8320, 0.25, np.float32)            # 2 s of APT envelope at 4160 Hz
    pend = []
    for _ in range(40):
        assert img.push(blk) == 0
        pend.append(img.get()[2])
    assert max(pend) <= 2080 + 500, max(pend)
    # and a real signal afterwards still synchronises: lines of 2080 samples with a sync pulse train
    line = np.full(2080, 0.3, np.float32)
    sync = np.array([0, 0, 0, 0] + [1, 1, 0, 0] * 7 + [0] * 7, np.float32)
    line[:39] = 0.1 + 0.8 * sync
    added = 0
    for _ in range(6):
        added += img.push(np.tile(line, 4))
    assert added >= 8, added
