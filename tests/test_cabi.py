"""CPU-side checks of the drop-in boundary: libblsmi.so builds for gfx950, loads, and exports every
symbol include/blsmi.h declares; the host mirror fails loudly (no fallback) without a device."""
import ctypes
import os

import pytest

from bls_amd import _native


@pytest.fixture(scope="module")
def lib():
    _native.build()
    return _native.load()


def test_exports_every_declared_symbol(lib):
    syms = _native.declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_library_is_gfx950_code_object():
    data = open(_native.SO_PATH, "rb").read()
    assert b"gfx950" in data


def test_no_device_is_a_loud_error(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from bls_amd import engine, g2pubs
    assert lib.blsmi_init(0) == -1          # BLSMI_E_NODEVICE
    with pytest.raises(engine.BlsmiError):
        engine.pairing_batch(bytes(96), bytes(192), 1)
    with pytest.raises(engine.BlsmiError):
        g2pubs.Verify(b"m", g2pubs.NewPublicKeyFromG2(bytes(192)), g2pubs.NewSignatureFromG1(bytes(96)))


def test_product_never_touches_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "bls_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".inc", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "refcpu" not in txt and "pyref" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_duplicate_argument_checks(lib):
    # argument validation happens before any device work
    assert lib.blsmi_pairing_batch(None, None, None, ctypes.c_size_t(1)) == -3
    ok = ctypes.c_int(7)
    assert lib.blsmi_g2pubs_verify_aggregate(None, None, None, None, ctypes.c_size_t(0), ctypes.byref(ok)) == -3


def test_header_is_plain_c():
    """The boundary is a C ABI: include/blsmi.h must compile as C99 (cgo feeds it to a C compiler) and as C++."""
    import os
    import shutil
    import subprocess
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "blsmi.h")
    gcc = shutil.which("gcc")
    if gcc is None:
        import pytest
        pytest.skip("no gcc")
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    gpp = shutil.which("g++")
    if gpp:
        subprocess.check_call([gpp, "-std=c++17", "-fsyntax-only", "-x", "c++", hdr])
