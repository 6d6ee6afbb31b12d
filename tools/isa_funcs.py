#!/usr/bin/env python3
"""Per-function static instruction counts of one kernel unit's device code (hipcc -S): total, 64-bit multiply-adds, DPP moves, scratch
accesses, calls.  usage: python tools/isa_funcs.py k_pairing_row.hip [-DFLAG ...] [name-filter]"""
import os, re, subprocess, sys, collections
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
unit = sys.argv[1]
flags = [a for a in sys.argv[2:] if a.startswith("-")]
filt = [a for a in sys.argv[2:] if not a.startswith("-")]
out = "/tmp/isa_%s.s" % os.path.basename(unit)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-DBLSMI_LIMBS28", "-S", "--cuda-device-only", "-o", out] + flags + [os.path.join(root, "bls_amd", "csrc", unit)])
cur, stats = None, collections.OrderedDict()
for line in open(out):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1); stats[cur] = collections.Counter(); continue
    if cur is None or not re.match(r"^\s+[a-z]", line) or line.strip().startswith("."):
        continue
    op = line.split()[0]
    c = stats[cur]
    c["total"] += 1
    if op.startswith("v_mad_") and "64" in op: c["mad64"] += 1
    if "dpp" in line: c["dpp"] += 1
    if op.startswith("scratch_") or op.startswith("buffer_"): c["scratch"] += 1
    if op.startswith("s_swappc"): c["calls"] += 1
    if op.startswith("v_accvgpr"): c["acc"] += 1
    if op.startswith("s_waitcnt"): c["wait"] += 1
def dem(n):
    try: return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()[:110]
    except Exception: return n
for n, c in stats.items():
    d = dem(n)
    if filt and not any(f in d for f in filt): continue
    print("%6d  mad64=%-5d dpp=%-5d scratch=%-4d acc=%-4d calls=%-3d wait=%-4d %s" % (c["total"], c["mad64"], c["dpp"], c["scratch"], c["acc"], c["calls"], c["wait"], d))
