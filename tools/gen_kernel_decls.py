#!/usr/bin/env python3
"""Regenerates bls_amd/csrc/kernels.h (the host side's declarations of every kernel) from the kernel definitions."""
import os
import re

D = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bls_amd", "csrc")
FILES = ["k_pairing_single.hip", "k_fe_single.hip", "k_fq12_single.hip", "k_pairing_pair.hip", "pair_kernels.inc", "k_pairing_quad.hip", "k_prepared_pair.hip", "k_lat.hip", "k_hash.hip", "k_wire.hip", "k_hash_pair.hip", "k_curve.hip", "k_msm_pair.hip", "msm.inc"]
PAT = re.compile(r"^(KERNEL2|KERNEL_PAIR|KERNEL_QUAD|KERNEL_LAT|KERNEL|__global__ void __launch_bounds__\([^)]*\))\s+(k_\w+)\(([^)]*)\)\s*\{", re.M)

out = """// kernels.h -- declarations of the kernels defined in the k_*.hip translation units, for the host side (blsmi.hip).
// Generated from the definitions by tools/gen_kernel_decls.py; the launch bounds live with the definitions.
#pragma once
#include "fp.cuh"
using namespace blsmi;
#define WG 64
constexpr int PT = WG / 2;          // tuples per workgroup of the lane-pair kernels
constexpr int QT = WG / 4;          // tuples per workgroup of the lane-quad kernels
"""
seen = set()
for f in FILES:
    p = os.path.join(D, f)
    if not os.path.exists(p):
        continue
    out += "// %s\n" % f
    for m in PAT.finditer(open(p).read()):
        assert m.group(2) not in seen, m.group(2)
        seen.add(m.group(2))
        out += "__global__ void %s(%s);\n" % (m.group(2), m.group(3))
open(os.path.join(D, "kernels.h"), "w").write(out)
print("%d kernels" % len(seen))
