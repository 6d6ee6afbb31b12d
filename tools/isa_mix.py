#!/usr/bin/env python3
"""Instruction mix of the pairing kernels' device code, by ENCODING -- the input of bench.py's `valu.frac_mix` ("make the
two roofline numbers honest").

Why: the flat VALU ceiling charges every vector instruction 4 cycles.  tools/ubench_rate.hip measured otherwise on this chip
(profiles/r03_ubench_rate.log, two waves per SIMD, the pairing kernels' occupancy): a VOP3 / DPP encoding (v_mad_i64_i32, v_mul_lo_u32,
v_mov_b32_dpp, v_bfi_b32, v_ashrrev_i64, ...) retires one wave-instruction per ~4.5 cycles per SIMD, a 4-byte VOP1 / VOP2 encoding
(v_add_u32, v_and_b32, v_lshrrev_b32, v_mov_b32) per ~2.1, a VOP2 with a 32-bit literal per ~2.6.  The minimum time of a kernel is therefore
sum_i count_i x cycles_i, not count x 4.

What this tool produces (no GPU needed; hipcc + llvm-objdump):
  * per function of the lane-pair pairing units (k_pairing_pair.hip, k_fe_pair.hip, compiled exactly as bls_amd/_native.py compiles them): the
    number of VALU instructions by class -- `w64` (64-bit integer ops: v_mad_[iu]64_*, *_[iu]64, *_b64; what SQ_INSTS_VALU_INT64 counts),
    `vop3` (other 8-byte VOP3 / DPP / SDWA encodings), `lit` (4-byte op + literal), `e32` (4-byte) -- and the MAC count (v_mad_[iu]64);
  * per unit: the STATIC split of the non-w64 instructions into vop3 / lit / e32.  bench.py combines it with the DYNAMIC counts of a rocprofv3
    PMC pass of the same kernels (SQ_INSTS_VALU = all VALU wave-instructions executed, SQ_INSTS_VALU_INT64 = the w64 ones; tools/profile_round.sh):
        executed w64 instructions            -> taken from the counter (exact);
        executed other instructions          -> SQ_INSTS_VALU - SQ_INSTS_VALU_INT64, split by the unit's static ratio (the approximation: the
                                                glue's loops are assumed to execute its encodings in their static proportion; the straight-line
                                                multiply cores, where 78 % of the instructions are, are exact either way because they are w64-dominated);
  * the cycle table, parsed from profiles/r03_ubench_rate.log (medians of the two-waves-per-SIMD rows).
usage: python tools/isa_mix.py [profiles/r05_isa_mix.json]"""
import collections
import json
import os
import re
import statistics
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bls_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
UNITS = {"k_pairing_pair.hip": ["k_miller1h_pair", "k_miller1_pair", "k_miller2_pair", "k_miller1x2_pair"], "k_fe_pair.hip": ["k_final_exp_pair", "k_final_exp_is_one_pair"]}
W64 = re.compile(r"^v_\w*(_[iu]64|_b64)(_|$)|^v_mad_[iu]64")
MAC = re.compile(r"^v_mad_[iu]64")


def disassemble(unit):
    sys.path.insert(0, ROOT)
    from bls_amd import _native
    flags = _native._FLAGS + _native._unit_flags(unit)
    with tempfile.TemporaryDirectory() as d:
        obj, elf = os.path.join(d, "u.o"), os.path.join(d, "u.elf")
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ["--cuda-device-only", "-c", "-o", obj, os.path.join(CSRC, unit)])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + obj, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf])
        return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", elf], text=True)


def classify(mnemonic, nbytes):
    if W64.search(mnemonic):
        return "w64"
    if mnemonic.endswith("_e32"):
        return "lit" if nbytes > 4 else "e32"
    return "vop3"                                            # _e64, _dpp, _sdwa and the VOP3-only opcodes (v_bfi_b32, v_add3_u32, v_mul_lo_u32, ...)


def functions_of(dis):
    fns, cur = collections.OrderedDict(), None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = fns.setdefault(m.group(1), collections.Counter())
            continue
        m = re.match(r"^\s+(v_\w+)\s.*//\s*[0-9A-F]+:((?:\s+[0-9A-F]{8})+)", line)
        if m and cur is not None:
            nbytes = 4 * len(m.group(2).split())
            cur[classify(m.group(1), nbytes)] += 1
            cur["valu"] += 1
            cur["mac"] += 1 if MAC.search(m.group(1)) else 0
            cur["bytes8plus"] += 1 if nbytes >= 8 else 0
    return fns


def cycle_table(path=os.path.join(ROOT, "profiles", "r03_ubench_rate.log")):
    """cycles per wave-instruction per SIMD issue slot at two waves per SIMD, by class (median over the measured opcodes of the class)"""
    cls = {"v_mad_u64_u32": "w64", "v_mad_i64_i32": "w64", "v_ashrrev_i64": "w64", "v_add3_u32": "vop3", "v_mul_lo_u32": "vop3", "v_mov_b32_dpp quad_perm": "vop3",
           "v_bfi_b32": "vop3", "v_and_or_b32": "vop3", "v_alignbit_b32": "vop3", "v_lshl_add_u32": "vop3", "v_add_u32 literal": "lit",
           "v_add_u32": "e32", "v_and_b32": "e32", "v_lshrrev_b32": "e32", "v_sub_u32": "e32", "v_mov_b32 pair": "e32"}
    got = collections.defaultdict(list)
    for line in open(path):
        m = re.match(r"^(.*?)\s+(\d) chains?\s+waves/SIMD=2\s+[\d.]+ ms\s+[\d.]+ cycles per instruction per wave,\s+([\d.]+) per SIMD issue slot", line)
        if m and m.group(1).strip() in cls:
            got[cls[m.group(1).strip()]].append(float(m.group(3)))
    return {k: round(statistics.median(v), 3) for k, v in got.items()}, os.path.relpath(path, ROOT)


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_isa_mix.json")
    cycles, src = cycle_table()
    res = {"what": "static instruction mix by encoding of the lane-pair pairing units (tools/isa_mix.py); bench.py: valu.frac_mix", "cycles_per_wave_instruction_per_simd": cycles,
           "cycles_source": src + " (two waves per SIMD; clock taken as 2.4 GHz there and here)", "units": {}}
    for unit, kernels in UNITS.items():
        fns = functions_of(disassemble(unit))
        tot = collections.Counter()
        for name, c in fns.items():
            kern = re.match(r"^_Z\d+(k_\w+?)P", name)                     # a kernel of this unit: only the ones the pairing path launches count
            if "debug" in name or (kern and kern.group(1) not in kernels):   # (the unit also holds the parity tests' debug kernels and table builders)
                continue
            tot.update(c)
        rest = max(1, tot["valu"] - tot["w64"])
        short = {}
        for name, c in fns.items():
            if c["valu"] >= 200:
                key = re.sub(r"^_ZN5blsmi5pairl\d+|^_ZN5blsmi\d+|^_Z\d+", "", name)[:40]
                short[key] = {k: c[k] for k in ("valu", "w64", "mac", "vop3", "lit", "e32")}
        res["units"][unit] = {"kernels": kernels, "static": {k: tot[k] for k in ("valu", "w64", "mac", "vop3", "lit", "e32", "bytes8plus")},
                              "non_w64_split": {k: round(tot[k] / rest, 4) for k in ("vop3", "lit", "e32")},
                              "static_share_w64": round(tot["w64"] / max(1, tot["valu"]), 4), "functions": short}
    core = res["units"]["k_fe_pair.hip"]["functions"]
    res["cores"] = {k: v for k, v in core.items() if k.startswith("blsmi_core")}
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({u: (v["static"], v["non_w64_split"]) for u, v in res["units"].items()}, indent=1))
    print("cycles", cycles, "->", os.path.relpath(out_path, ROOT))


if __name__ == "__main__":
    main()
