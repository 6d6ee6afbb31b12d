"""CPU (hipcc cross-compiles): guards on the GENERATED code of the kernels.

Round 6 found that `v_subrev_u32_dpp d, x, y` computes dpp(x) - y on gfx950 -- what v_sub_u32_dpp computes -- not y - dpp(x)
(tools/dpp_sub_probe.hip, profiles/r06_dpp_sub_probe.txt: 64 of 64 lanes), and that the compiler's DPP combine produces exactly that
instruction when a DPP move with a single use is the SUBTRAHEND of a subtraction (`P - rrot<7>(P)` in the row layout's addition step:
wrong Y3 and c0).  The sources avoid the pattern; this test makes sure no rebuild brings a reversed-operand DPP instruction back."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bls_amd", "csrc")
REV_DPP = re.compile(r"^\s+v_(sub|subb|lshl|lshr|ashr)rev[a-z0-9_]*_dpp\b", re.M)


def _compile(unit, out):
    from bls_amd import _native
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc] + [f for f in _native._FLAGS if f != "-fPIC"] + _native._unit_flags(unit) + ["-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, unit)], stderr=subprocess.DEVNULL)
    return open(out).read()


def test_no_reversed_operand_dpp_instruction_in_any_unit(tmp_path):
    """Every kernel unit, compiled with the library's own flags (concurrently: about a minute).  Late in round 6 the guard caught what the parity tests could not:
    quad_g1.inc's homogeneous addition had `b - quad_perm(a)` folded into v_subrev_u32_dpp, the result was right on lane 0 of every quad only -- and every consumer
    happened to read lane 0 (tools/dpp_sub_probe2.hip, profiles/r06_dpp_sub_probe2.txt: right on the lanes that are their own source)."""
    import sys
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, ROOT)
    from bls_amd import _native
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc) and shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    units = [u for u in _native._UNITS if u != "blsmi.hip"]                # (the host side has no kernels of its own)
    with ThreadPoolExecutor(8) as ex:
        asms = list(ex.map(lambda u: _compile(u, str(tmp_path / (u + ".s"))), units))
    for unit, asm in zip(units, asms):
        if unit in ("k_pairing_row.hip", "k_hash_quad.hip", "k_pairing_quad.hip"):
            assert asm.count("_dpp") > 1000, unit                          # these units are DPP code all over
        bad = REV_DPP.findall(asm)
        assert not bad, "%s: %d reversed-operand DPP instructions (v_*rev*_dpp do not compute S1 op dpp(S0) on gfx950)" % (unit, len(bad))
