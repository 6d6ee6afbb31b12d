"""CPU (hipcc cross-compiles): guards on the GENERATED code of the lane-row kernels.

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


@pytest.mark.parametrize("unit", ["k_pairing_row.hip", "k_hash_quad.hip"])
def test_no_reversed_operand_dpp_instruction(unit, tmp_path):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc) and shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    out = str(tmp_path / (unit + ".s"))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-DBLSMI_LIMBS28", "-S", "--cuda-device-only",
                           "-o", out, os.path.join(CSRC, unit)], stderr=subprocess.DEVNULL)
    asm = open(out).read()
    assert asm.count("_dpp") > 1000                                        # the unit is DPP code all over
    bad = REV_DPP.findall(asm)
    assert not bad, "%s: %d reversed-operand DPP instructions (v_*rev*_dpp compute the un-reversed result on gfx950)" % (unit, len(bad))
