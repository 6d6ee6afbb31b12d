#!/usr/bin/env python3
"""bench.py -- BLS12-381 pairings/sec (batch verify path) on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  A step is one pass of the hot path over one batch of
synthetic input resident in HBM: BASELINE.json configs[1] -- 65 536 independent pairings (Miller loop + final
exponentiation, the reference's bls.Pairing) per GPU.  W untimed steps, then exactly K timed steps bracketed by barrier +
synchronize on both sides, MAX over ranks, one JSON line from rank 0.  Units shard across GPUs with no data-path collective
(weak scaling: 64k pairings per GPU).  After the timed region the output is compared with the oracle and the run FAILS on a
mismatch.

Two launch styles, same numbers at N = 1:
  * under torch.distributed.run (RANK / WORLD_SIZE in the environment): one process per GPU, RCCL through torch.distributed;
  * plain `python bench.py --gpus N`: ONE process drives N devices through the library (blsmi_init_devices(N)) -- the
    deployment the Go API implies (one process, one call, g2pubs/bls.go:159, 240).  The headline step runs one resident
    65 536-pairing batch per device concurrently (blsmi_pairing_batch_dev routes by buffer ownership); `inlibrary_bench`
    times the split HOST entry points -- N x 65 536 pairings, N x 65 536 verifies with the library's own ncclAllReduce of
    the bitmap inside every timed call, and the 2^20-signature VerifyAggregate with its ncclAllGather of partial products.

Every BASELINE config has an entry under `configs` (value, roofline of its dominant kernel, CPU baseline of the same shape):
  configs[0]  1 000 g2pubs.Verify tuples on the CPU restatement of the reference (plumbing), the GPU verdicts beside it
  configs[1]  the headline (65 536 pairings per GPU)
  configs[2]  2^20-point G1 and G2 scalar multiplication and MSM, inputs resident (blsmi_g*_mul_batch_dev / _msm_dev)
  configs[3]  one 2^20-signature g2pubs VerifyAggregate, sharded over the N GPUs
  configs[4]  one 262 144-message g1pubs VerifyAggregate on one GPU
Rooflines: algorithmic bytes per unit x units per launch / the dominant kernel's duration measured with HIP events
on the launch stream by the library itself (blsmi_set_profiling / blsmi_last_profile); `traffic` and the VALU counters come
from the committed rocprofv3 PMC passes of this same command (profiles/rNN_counters.json) -- the line says which file, and
"stale": true when any kernel source changed since (source digest recorded at profile time).
"""
import argparse
import ctypes
import glob
import hashlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PAIRINGS_PER_GPU = 65536
HBM_PEAK_GBS = 8000.0              # MI355X spec (MI355X_MICROARCH.md)
# algorithmic bytes per unit
BYTES = {"pairing": 864,           # 96 B G1 + 192 B G2 in, 576 B Fq12 out
         "g1_mul": 224, "g2_mul": 416,          # point + 32-byte scalar in, point out
         "g1_msm": 128, "g2_msm": 224,          # point + scalar in (one point out per launch)
         "verify": 320,                         # 96 + 192 + 32-byte message in, 1 bit out (either package)
         "g2pubs_aggregate": 224, "g1pubs_aggregate": 128}   # 32-byte message + key in
# integer-VALU issue ceiling: 1 024 SIMDs x 64 lanes per wave-instruction / 4 cycles x clock (the int32 / v_mad_*64 rate, tools/ubench*)
SIMDS, LANES, CYCLES_PER_VALU, CLOCK_GHZ = 1024, 64, 4.0, 2.4
# measured ceiling of the 15x27-limb Montgomery multiplication core (tools/ubench2.hip, profiles/r01_ubench2_fmul_15x27.log), carried to the
# 14x28-limb core the pairing kernels use since round 4 by the MEASURED time ratio of the two cores (tools/ubench_core28.hip)
CORE28_OVER_CORE27 = 3236.0 / 3738.2      # measured: lane-pair Fq2 product, two waves per SIMD, dependent chain -- profiles/r04_ubench_core28.log (196 / 225 = 0.871 by MAC count)
VALU_PEAK_GMULS = 61.2 / CORE28_OVER_CORE27
VALU_PEAK_2WAVE_GMULS = 56.7 / CORE28_OVER_CORE27
FQ_MULS_PER_PAIRING = 14600        # SURVEY 8d optimised estimate: the nominal work unit of `valu.nominal`
R_ORDER = 52435875175126190479447740508185965837690552500527637822603658699938581184513


# ---------------------------------------------------------------------------------------------------------------------
# committed rocprofv3 counters (bench.py cannot read PMCs itself)
# ---------------------------------------------------------------------------------------------------------------------
# host-side translation units: no kernel in them, a change there cannot make a counter or a kernel duration stale
HOST_ONLY = ("blsmi.hip", "verify_host.inc", "kernels.h")


def source_digest():
    """sha256 over the kernel sources: recorded by tools/rocpd_summary.py at profile time, recomputed here -> `stale`"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "bls_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".cuh", ".inc", ".h", ".py")) and not f.startswith("lat_programs") and f not in HOST_ONLY:
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def profile_counters():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_counters.json")))
    if not files:
        return {"file": None, "stale": None, "kernels": {}}
    try:
        j = json.load(open(files[-1]))
        dig = j.get("source_digest")
        return {"file": os.path.relpath(files[-1], ROOT), "commit": j.get("commit"), "source_digest": dig,
                "stale": (dig != source_digest()) if dig else True, "kernels": j["kernels"]}
    except (OSError, ValueError, KeyError):
        return {"file": None, "stale": None, "kernels": {}}


def isa_mix():
    """the committed static instruction mix by encoding of the pairing units + the measured cycles per encoding class (tools/isa_mix.py)"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_isa_mix.json")))
    if not files:
        return None
    try:
        j = json.load(open(files[-1])); j["file"] = os.path.relpath(files[-1], ROOT)
        return j
    except (OSError, ValueError):
        return None


def valu_mix_floor(ctr, mix, kernel_of_unit, grid, n):
    """Mix-weighted minimum time of the pairing kernels: sum over encoding classes of executed wave-instructions x measured cycles per
    instruction of that class (two waves per SIMD), spread over the chip's SIMDs.  Executed counts: SQ_INSTS_VALU (all) and
    SQ_INSTS_VALU_INT64 (the 64-bit integer ops: the multiply-adds and column shifts) of the committed PMC pass; the remainder is split by
    the unit's static encoding ratio (tools/isa_mix.py says what that assumes).  Returns {kernel: {...}}, total minimum ms -- or None."""
    if not mix:
        return None
    cyc = mix["cycles_per_wave_instruction_per_simd"]
    out, total = {}, 0.0
    for unit, kernel in kernel_of_unit.items():
        c = counter_of(ctr, kernel, grid)
        u = mix["units"].get(unit)
        if "SQ_INSTS_VALU" not in c or not u:
            return None
        tot = c["SQ_INSTS_VALU"] * (n / 65536.0)
        dyn = "SQ_INSTS_VALU_INT64" in c
        w64 = (c["SQ_INSTS_VALU_INT64"] * (n / 65536.0)) if dyn else tot * u["static_share_w64"]
        rest = max(0.0, tot - w64)
        sp = u["non_w64_split"]
        cycles = w64 * cyc["w64"] + rest * (sp["vop3"] * cyc["vop3"] + sp["lit"] * cyc["lit"] + sp["e32"] * cyc["e32"])
        ms = cycles / SIMDS / (CLOCK_GHZ * 1e9) * 1e3
        out[kernel] = {"wave_instructions": round(tot), "w64": round(w64), "w64_source": "SQ_INSTS_VALU_INT64" if dyn else "static share (no INT64 counter in the PMC file)",
                       "w64_share": round(w64 / tot, 4) if tot else None, "mean_cycles_per_instruction": round(cycles / tot, 3) if tot else None, "min_ms": round(ms, 3)}
        total += ms
    return out, total


def counter_of(ctr, kernel, grid=None):
    """counters of `kernel` at launch grid `grid` (lanes); the profile keeps one record per (kernel, grid)"""
    k = ctr["kernels"].get(kernel)
    if not k:
        return {}
    if grid is not None and "by_grid" in k and str(grid) in k["by_grid"]:
        return k["by_grid"][str(grid)]
    if grid is not None and k.get("grid") not in (None, grid) and k.get("pmc_grid") not in (None, grid):
        return {}
    return k


# ---------------------------------------------------------------------------------------------------------------------
# synthetic inputs
# ---------------------------------------------------------------------------------------------------------------------
def _gens():
    g1gen = bytes.fromhex(
        "17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
        "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1")
    g2gen = bytes.fromhex(
        "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8"
        "13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
        "0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801"
        "0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be")
    return g1gen, g2gen


def synth_inputs(engine, n, seed):
    """n (P_i, Q_i) pairs: P = a_j G1, Q = b_j G2 for 512 seeded scalars, tiled with a row rotation so that all
    n combinations are distinct.  Generated on the device by the library's own scalar multiplication."""
    base = 512
    sc = [hashlib.sha256(b"blsmi-bench-%d-%d" % (seed, i)).digest() for i in range(2 * base)]
    sc = [(int.from_bytes(s, "big") % (R_ORDER - 1) + 1).to_bytes(32, "big") for s in sc]
    g1b, _ = engine.g1_mul_generator_batch(b"".join(sc[:base]), base)
    g2b, _ = engine.g2_mul_generator_batch(b"".join(sc[base:]), base)
    reps = (n + base - 1) // base
    g1 = np.tile(g1b, (reps, 1))[:n]
    g2 = np.concatenate([np.roll(g2b, -r, axis=0) for r in range(reps)])[:n]
    return np.ascontiguousarray(g1), np.ascontiguousarray(g2)


def _verify_tuples(engine, group, n, tag=0, nk=256):
    """n valid (message, public key, signature) tuples of one package as host arrays (signed on the device)"""
    sk = b"".join(hashlib.sha256(b"bench-sk-%d" % i).digest()[:31].rjust(32, b"\0") for i in range(nk))
    msgs = [b"Hello world! 16 characters %d" % (i + tag * n) for i in range(n)]
    buf = np.frombuffer(b"".join(msgs), dtype=np.uint8)
    off = np.zeros(n + 1, dtype=np.uint64); off[1:] = np.cumsum([len(m) for m in msgs])
    packed = engine.PackedMsgs.__new__(engine.PackedMsgs); packed.buf, packed.off, packed.n = buf, off, n
    if group == "g2pubs":
        pks, _ = engine.g2_mul_generator_batch(sk, nk)
        h = engine.hash_g1_batch(packed)
        sigs, _ = engine.g1_mul_batch(h.reshape(-1), sk * (n // nk), n)
    else:
        pks, _ = engine.g1_mul_generator_batch(sk, nk)
        h = engine.hash_g2_batch(packed)
        sigs, _ = engine.g2_mul_batch(h.reshape(-1), sk * (n // nk), n)
    allpk = np.ascontiguousarray(np.tile(pks, (n // nk, 1)))
    return packed, allpk, np.ascontiguousarray(sigs)


def _aggregate_inputs(engine, group, lo, n, nk=256):
    """n (32-byte message, key) pairs with indices lo .. lo+n and this block's share of the aggregate signature"""
    sk = b"".join(hashlib.sha256(b"agg-sk-%d" % i).digest()[:31].rjust(32, b"\0") for i in range(nk))
    msgs = [hashlib.sha256(int(i).to_bytes(8, "little")).digest() for i in range(lo, lo + n)]
    packed = engine.PackedMsgs(msgs)
    if group == "g2pubs":
        pks, _ = engine.g2_mul_generator_batch(sk, nk)
        h = engine.hash_g1_batch(packed)
        sigs, _ = engine.g1_mul_batch(h.reshape(-1), sk * (n // nk), n)
        part = engine.g1_sum(sigs.reshape(-1), n)
    else:
        pks, _ = engine.g1_mul_generator_batch(sk, nk)
        h = engine.hash_g2_batch(packed)
        sigs, _ = engine.g2_mul_batch(h.reshape(-1), sk * (n // nk), n)
        part = engine.g2_sum(sigs.reshape(-1), n)
    allpk = np.ascontiguousarray(np.tile(pks, (n // nk, 1))).reshape(-1)
    return packed, allpk, part, pks


# ---------------------------------------------------------------------------------------------------------------------
# per-kernel timing through the library (HIP events on the launch stream) and the roofline object of a leg
# ---------------------------------------------------------------------------------------------------------------------
def read_profile(lib):
    buf = ctypes.create_string_buffer(1 << 16)
    lib.blsmi_last_profile(buf, ctypes.c_size_t(len(buf)))
    out = {}
    for item in buf.value.decode().split(";"):
        if "=" in item:
            k, v = item.split("=")
            e = out.setdefault(k, [0.0, 0]); e[0] += float(v); e[1] += 1
    return out          # kernel -> [total ms, launches]


def profiled(lib, fn):
    """run fn() once with the library's per-kernel event timing on; returns {kernel: [ms, launches]}"""
    read_profile(lib)
    lib.blsmi_set_profiling(1)
    try:
        fn()
    finally:
        lib.blsmi_set_profiling(0)
    return read_profile(lib)


def hbm_traffic(c, scale=1.0):
    """HBM bytes per launch from one kernel's PMC record: (2 x FETCH_SIZE + WRITE_SIZE) x 1024.  On gfx950 FETCH_SIZE (derived from
    TCC_EA0_RDREQ, which counts a 128-byte request as 64) reports HALF of the bytes read; WRITE_SIZE is exact.  Calibrated on this chip
    with a scratch pattern of known size (tools/profile_tcc.sh, profiles/r04_tcc_summary.txt) and prescribed by MI355X_MICROARCH.md.
    Returns (corrected, raw FETCH_SIZE + WRITE_SIZE as rocprofv3 prints them), or (None, None)."""
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        return None, None
    return (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 * scale, (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 * scale


def roofline_of(prof, units, bytes_per_unit, ctr, grid_of=None):
    """roofline object of one leg: dominant kernel of `prof`, algorithmic bytes / its duration vs the HBM peak"""
    if not prof:
        return None
    prof = {k: v for k, v in prof.items() if not k.startswith("(")} or prof     # "(between)": gaps between the marked kernels
    dom = max(prof, key=lambda k: prof[k][0])
    ms, launches = prof[dom]
    per_launch_ms = ms / max(1, launches)
    algo = bytes_per_unit * units
    achieved = algo / (ms * 1e-3) / 1e9
    c = counter_of(ctr, dom, grid_of(dom) if grid_of else None)
    if not grid_of and "by_grid" in ctr["kernels"].get(dom, {}):          # launch grid not stated: the profile's record of this kernel that ran closest to this duration
        recs = [r for r in ctr["kernels"][dom]["by_grid"].values() if "avg_ns" in r]
        if recs:
            c = min(recs, key=lambda r: abs(r["avg_ns"] / 1e6 - per_launch_ms))
    traffic, traffic_raw = hbm_traffic(c)
    return {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 8),
            "traffic": traffic, "traffic_raw": traffic_raw, "algorithmic_bytes_per_launch": algo, "bytes_per_unit": bytes_per_unit, "units_per_launch": units,
            "kernel_ms": round(per_launch_ms, 4), "kernel_launches": launches,
            "all_kernels_ms": {k: round(v[0], 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:10]},
            "rocprof_avg_ms": round(c["avg_ns"] / 1e6, 4) if "avg_ns" in c else None}


# ---------------------------------------------------------------------------------------------------------------------
# CPU baselines (the oracle = C restatement of the reference algorithm, kind "port"), bounded samples
# ---------------------------------------------------------------------------------------------------------------------
def usable_cores():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (cpu.max / cfs_quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_timed(work, per_thread, unit, what, budget_s):
    """work(k, m): process m units as thread k.  Calibrates on 2 units, then runs per-core chunks for ~budget_s."""
    from concurrent.futures import ThreadPoolExecutor
    cores = usable_cores()
    t0 = time.time(); work(0, 2); per = (time.time() - t0) / 2
    m = int(max(2, min(per_thread, budget_s / max(per, 1e-6))))
    t0 = time.time()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(lambda k: work(k, m), range(cores)))
    dt = time.time() - t0
    return {"value": round(cores * m / dt, 2), "unit": unit, "cores": cores, "kind": "port",
            "sample": "%d %s (%d per thread x %d threads = usable cores: affinity mask capped by the cgroup CPU quota; host reports %d logical CPUs; %.1f s wall); "
                      "oracle/refcpu.c = C restatement of the Go reference (no Go toolchain on this image)" % (cores * m, what, m, cores, os.cpu_count() or 0, dt),
            "single_core_per_s": round(1.0 / per, 2)}


def cpu_pairing(g1, g2, budget_s=8.0):
    from oracle import refcpu as RC
    n = g1.shape[0]

    def work(k, m):
        lo = (k * m) % max(1, n - m)
        RC.pairing_batch(g1[lo:lo + m].tobytes(), g2[lo:lo + m].tobytes(), m)
    r = cpu_timed(work, n // usable_cores(), "pairings/s", "reference-algorithm Pairing() calls of the same workload", budget_s)
    r["single_core_pairings_per_s"] = r["single_core_per_s"]
    return r


def cpu_mul(group, pts, ks, budget_s=2.5):
    from oracle import refcpu as RC
    fn = RC.g1_mul if group == "g1" else RC.g2_mul

    def work(k, m):
        for i in range(m):
            j = (k * m + i) % pts.shape[0]
            fn(pts[j].tobytes(), ks[j].tobytes())
    return cpu_timed(work, 4096, "scalar multiplications/s", "reference-algorithm %s MulFR (bit-serial double-and-add, %s.go) of the same points and scalars" % (group.upper(), group), budget_s)


def cpu_verify(group, packed, pks, sigs, budget_s=3.0):
    from oracle import refcpu as RC
    o = RC.g2pubs if group == "g2pubs" else RC.g1pubs
    msgs = [bytes(packed.buf[int(packed.off[i]):int(packed.off[i + 1])]) for i in range(min(packed.n, 2048))]

    def work(k, m):
        idx = [(k * m + i) % len(msgs) for i in range(m)]
        o.verify_batch([msgs[i] for i in idx], [pks[i].tobytes() for i in idx], [sigs[i].tobytes() for i in idx])
    return cpu_timed(work, 1024, "verifies/s", "reference-algorithm %s.Verify calls (hash-to-curve + CompareTwoPairings) of the same tuples" % group, budget_s)


def cpu_aggregate(engine, group, m=1024):
    """VerifyAggregate-shaped CPU baseline: every usable core runs ONE reference-algorithm VerifyAggregate of the same m distinct-message
    signers (n + 1 full pairings + n hashes, g2pubs/bls.go:240-270), all cores concurrently; verdict True is asserted."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import refcpu as RC
    o = RC.g2pubs if group == "g2pubs" else RC.g1pubs
    pkb = 192 if group == "g2pubs" else 96
    packed, allpk, agg, _ = _aggregate_inputs(engine, group, 0, m)
    msgs = [bytes(packed.buf[int(packed.off[i]):int(packed.off[i + 1])]) for i in range(m)]
    pks = [allpk[i * pkb:(i + 1) * pkb].tobytes() for i in range(m)]
    sig = bytes(agg)
    cores = usable_cores()
    # one core: the MARGINAL cost of a signer -- the slope between a 4-signer and a 20-signer prefix (4 + 1 and 20 + 1 pairings: the signature side's
    # pairing, which every call pays once, drops out; ADVICE r05: dividing one 16-signer call's time by 16 was ~6 % pessimistic, by 17 the other way).
    # Their verdicts are False: the signature is the aggregate of all m.
    t0 = time.time(); ok_a = o.verify_aggregate(sig, pks[:4], msgs[:4]); ta = time.time() - t0
    t0 = time.time(); ok_b = o.verify_aggregate(sig, pks[:20], msgs[:20]); tb = time.time() - t0
    ok1 = ok_a or ok_b
    per = max((tb - ta) / 16, 1e-9)
    t0 = time.time()
    with ThreadPoolExecutor(cores) as ex:
        oks = list(ex.map(lambda k: o.verify_aggregate(sig, pks, msgs), range(cores)))
    dt = time.time() - t0
    assert all(oks) and not ok1, "CPU VerifyAggregate: the m-signer aggregate must verify (and its prefixes must not)"
    return {"value": round(cores * m / dt, 2), "unit": "signatures/s", "cores": cores, "kind": "port", "single_core_per_s": round(1.0 / per, 2), "single_core_basis": "slope between a 4-signer and a 20-signer call (marginal cost of a signer)",
            "sample": "%d concurrent reference-algorithm %s VerifyAggregate calls of %d signers each (%.1f s wall); oracle/refcpu.c" % (cores, group, m, dt)}


# ---------------------------------------------------------------------------------------------------------------------
# the stdout line: ONE compact JSON object (< 4 KB) the driver can keep whole; everything else -> bench_detail.json + stderr
# ---------------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 4096


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _roof_small(r, full=False):
    """the roofline object of a leg, cut to what the contract names (bound, achieved, peak, unit, frac, traffic + the kernel it is about)"""
    if not isinstance(r, dict):
        return None
    keys = ("bound", "kernel", "kernel_ms", "achieved", "peak", "unit", "frac", "traffic", "traffic_raw", "algorithmic_bytes_per_launch") if full else ("kernel", "kernel_ms", "frac", "traffic")
    o = _pick(r, keys)
    if isinstance(o.get("kernel_ms"), dict):                                # headline: both pairing kernels -> the dominant one's duration
        o["kernel_ms"] = r["kernel_ms"].get(r.get("kernel"))
    for k in ("traffic", "traffic_raw"):
        if isinstance(o.get(k), float):
            o[k] = int(o[k])
    return o


def _cpu_small(c, full=False):
    if not isinstance(c, dict):
        return None
    o = _pick(c, ("value", "unit", "cores", "kind", "single_core_per_s") if full else ("value", "cores"))
    if full and "sample" in c:
        o["sample"] = c["sample"][:150]
    return o


def _entry(value, unit, ms, roof, cpu, **more):
    e = {"value": value, "unit": unit, "ms": ms, "roofline": _roof_small(roof), "cpu_baseline": _cpu_small(cpu)}
    e.update({k: v for k, v in more.items() if v is not None})
    return e


def compact_configs(d):
    """five entries, one per BASELINE config, from the legs of the detail object `d` (absent legs -> absent entries)"""
    def ok(name):
        return isinstance(d.get(name), dict) and "error" not in d[name]
    cfg = {}
    if ok("config0"):
        c0 = d["config0"]
        cfg["0"] = _entry(c0["gpu"]["value"], "verifies/s", c0["gpu"]["ms_one_call"], c0.get("roofline"), c0["cpu"], verdicts_identical=c0.get("verdicts_identical"))
    cfg["1"] = _entry(d["value"], d["unit"], d["ms_per_step"], d.get("roofline"), d.get("cpu_baseline"),
                      prepared_g2_per_s=d["pairing_prepared"]["pairings_per_s"] if ok("pairing_prepared") else None)
    if ok("msm_bench"):
        cfg["2"] = {k: _entry(v["value"], v["unit"], v["ms_per_step"], v.get("roofline"), v.get("cpu_baseline"))
                    for k, v in d["msm_bench"].items() if k in ("g1_mul", "g1_msm", "g2_mul", "g2_msm")}
    a_dev, a_host = (d["g2pubs_aggregate_dev_bench"] if ok("g2pubs_aggregate_dev_bench") else None), (d["aggregate_bench"] if ok("aggregate_bench") else None)
    if a_dev or a_host:
        a = a_dev or a_host
        cfg["3"] = _entry(a["signatures_per_s"], "signatures/s", a["ms"], a.get("roofline"), a.get("cpu_baseline"), n=a.get("signatures"),
                          host_buffers_ms=a_host["ms"] if a_host and a_dev else None,
                          prepared_keys_ms=a_dev["prepared_keys"]["ms"] if a_dev and "prepared_keys" in a_dev else None)
    if ok("g1pubs_aggregate_bench"):
        a = d["g1pubs_aggregate_bench"]
        cfg["4"] = _entry(a["signatures_per_s"], "signatures/s", a["ms"], a.get("roofline"), a.get("cpu_baseline"), n=a.get("signatures"))
    return cfg


def compact_line(d, detail_file=None):
    """The ONE stdout line: the contract's keys + headline roofline / cpu_baseline / valu.frac / counters + five config entries.
    Guaranteed < LINE_LIMIT bytes: optional parts are dropped in a fixed order if a run ever produced more (they stay in the detail file)."""
    line = _pick(d, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                     "launch", "devices", "rccl_ranks", "library"))
    if d.get("aliased_devices"):
        line["aliased_devices"] = True
        if d.get("scaling_measured") is False:
            line["scaling_measured"] = False
    line["config"] = _pick(d.get("config", {}), ("workload", "pairings_per_gpu", "parallelism", "layout"))
    if len(line["config"].get("workload", "")) > 200:
        line["config"]["workload"] = line["config"]["workload"][:200]
    line["roofline"] = _roof_small(d.get("roofline"), full=True)
    if "cpu_baseline" in d:
        line["cpu_baseline"] = _cpu_small(d["cpu_baseline"], full=True)
    v = d.get("valu") or {}
    line["valu"] = _pick(v, ("frac", "frac_mix", "lane_instructions_per_pairing", "achieved", "peak", "unit"))
    line["counters"] = _pick(d.get("counters", {}), ("file", "stale"))
    line["self_check"] = _pick(d.get("self_check", {}), ("rows_per_device", "passed"))
    line["configs"] = compact_configs(d)
    vb = d.get("verify_bench")
    if isinstance(vb, dict) and "error" not in vb:
        line["verifies_per_s"] = {"g2pubs": vb.get("g2pubs_verifies_per_s"), "g1pubs": vb.get("g1pubs_verifies_per_s"),
                                  "g1pubs_with_domain": (vb.get("g1pubs_with_domain") or {}).get("verifies_per_s"),
                                  "g2pubs_prepared_keys": (vb.get("g2pubs_prepared_keys") or {}).get("verifies_per_s"),
                                  "g2pubs_in_memory_points": (vb.get("g2pubs_in_memory_points") or {}).get("verifies_per_s")}
    e2e = (d.get("reference_shapes") or {}).get("end_to_end_host")
    if isinstance(e2e, dict):                                                       # what a Go caller reaches, marshalling included (host buffers)
        line["end_to_end_host"] = {"jac_entry_verifies_per_s": e2e.get("jac_entry_verifies_per_s"),
                                   "affine_entry_one_marshalling_core": e2e.get("affine_entry_verifies_per_s_one_marshalling_core")}
    mid = d.get("mid_batches")
    if isinstance(mid, dict) and "error" not in mid:
        line["mid_batches"] = {"pairings_per_s": mid.get("pairings_per_s"), "verifies_per_s": mid.get("verifies_per_s")}
    il = d.get("inlibrary_bench")
    if isinstance(il, dict) and "error" not in il and il.get("devices", 1) > 1:
        line["inlibrary"] = _pick(il, ("devices", "rccl_ranks", "aliased_devices", "tuples_per_call", "pairings_per_s", "g2pubs_verifies_per_s", "g1pubs_verifies_per_s"))
        ab = d.get("aggregate_bench")
        if isinstance(ab, dict) and "error" not in ab:
            line["inlibrary"]["sharded_aggregate"] = _pick(ab, ("signatures", "ms", "signatures_per_s"))
    errs = [k for k, x in d.items() if isinstance(x, dict) and "error" in x]
    if errs:
        line["leg_errors"] = errs
    if detail_file:
        line["detail"] = detail_file
    for drop in ("mid_batches", "end_to_end_host", "verifies_per_s", "self_check", "library", "launch"):
        if len(json.dumps(line, separators=(",", ":"))) < LINE_LIMIT:
            break
        line.pop(drop, None)
    if len(json.dumps(line, separators=(",", ":"))) >= LINE_LIMIT:                  # last resort: configs without their sub-objects
        for e in line["configs"].values():
            for sub in (e.values() if "value" not in e else [e]):
                sub.pop("cpu_baseline", None)
    return line


def write_detail(d):
    """the full record of a run: bench_detail.json at the repo root, and under gpurun_out/ (what a gpurun call brings back)"""
    rel = "bench_detail.json"
    for path in (os.path.join(ROOT, rel), os.path.join(ROOT, "gpurun_out", rel)):
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                json.dump(d, f, indent=1)
        except OSError:
            pass
    return rel


# ---------------------------------------------------------------------------------------------------------------------
# legs
# ---------------------------------------------------------------------------------------------------------------------
class Env:
    """what a leg needs: engine, lib, torch device(s), rank/world, collectives"""


def fence(E):
    import torch
    if E.use_dist:
        E.dist.barrier()
    for d in E.devs:
        torch.cuda.synchronize(d)


def max_over_ranks(E, dt):
    import torch
    if not E.use_dist:
        return dt
    t = torch.tensor([dt], dtype=torch.float64, device=E.dev)
    E.dist.all_reduce(t, op=E.dist.ReduceOp.MAX)
    return float(t.item())


def timed_steps(E, step, steps, warmup):
    # two warm-up calls by default in the legs: the first grows the call context's arena of temporaries (overflow blocks), the
    # second's lease merges them into one block (hipFree + hipMalloc, sporadically tens of ms) -- neither belongs in a timed step
    for _ in range(warmup):
        step()
    fence(E); t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence(E)
    return max_over_ranks(E, time.perf_counter() - t0)


def verify_bench(E, steps=5, warmup=2, n=65536):
    """The batch-verify workload under bench.py's own timing discipline, both packages, inputs resident.  With a process
    group the bitmap all-reduce (SUM over disjoint bit ownership == OR, RCCL) runs inside every timed step."""
    import torch
    engine, dev, rank, world, dist = E.engine, E.dev, E.rank, E.world, E.dist
    out = {"tuples_per_gpu": n, "steps": steps, "warmup": warmup, "rccl_ranks": world if E.use_dist else 0,
           "collective": ("one all_reduce(SUM, int32 lanes, disjoint bit ownership) of the %d-byte bitmap over RCCL inside every timed step" % (world * n // 8)) if E.use_dist else "none (one GPU)"}
    weights = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=dev)
    for group in ("g2pubs", "g1pubs"):
        packed, pks, sigs = _verify_tuples(engine, group, n, tag=rank)
        d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (packed.buf.copy(), packed.off.view(np.int64), pks, sigs)]
        d_ok = torch.zeros(n, dtype=torch.uint8, device=dev)
        full = torch.zeros(world * n // 8, dtype=torch.int32, device=dev)

        def local_step():
            engine.verify_batch_dev(group, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 0, d_ok.data_ptr(), n)

        def step():
            local_step()
            if E.use_dist:
                full.zero_()
                full[rank * n // 8:(rank + 1) * n // 8] = (d_ok.view(-1, 8).to(torch.int32) * weights).sum(dim=1, dtype=torch.int32)
                dist.all_reduce(full, op=dist.ReduceOp.SUM)
        dt = timed_steps(E, step, steps, warmup)
        if E.use_dist:
            assert bool((full == 255).all().item()), "every rank's tuples must verify and land in the shared bitmap"
        assert bool(d_ok.all().item()), "synthetic tuples must all verify"
        out[group + "_verifies_per_s"] = round(world * n * steps / dt, 1)
        out[group + "_ms_per_step"] = round(dt / steps * 1e3, 3)
        if rank == 0:
            prof = profiled(E.lib, local_step)                            # rank 0 alone: no collective in here
            out[group + "_roofline"] = roofline_of(prof, n, BYTES["verify"], E.ctr, lambda k: 2 * n if k.endswith("_pair") else n)
            if world == 1 and E.cpu:
                out[group + "_cpu_baseline"] = cpu_verify(group, packed, pks, sigs)
        if group == "g2pubs":
            # the same tuples with the public keys PREPARED (blsmi 0.4): one 24 KB table per tuple -- every key distinct as far as the
            # memory system is concerned (n tables = 1.6 GB read per step); the timed step has the same collective as above
            tab = torch.empty(n * engine.G2_PREPARED_BYTES, dtype=torch.uint8, device=dev)
            t0 = time.perf_counter()
            engine.g2_prepare_batch_dev(d[2].data_ptr(), n, tab.data_ptr())
            t_prep = time.perf_counter() - t0

            def local_step_prepared():
                engine.g2pubs_verify_batch_prepared_dev(d[0].data_ptr(), d[1].data_ptr(), tab.data_ptr(), 0, d[3].data_ptr(), 0, d_ok.data_ptr(), n)

            def step_prepared():
                d_ok.zero_()
                local_step_prepared()
                if E.use_dist:
                    full.zero_()
                    full[rank * n // 8:(rank + 1) * n // 8] = (d_ok.view(-1, 8).to(torch.int32) * weights).sum(dim=1, dtype=torch.int32)
                    dist.all_reduce(full, op=dist.ReduceOp.SUM)
            dtp = timed_steps(E, step_prepared, steps, warmup)
            assert bool(d_ok.all().item()), "synthetic tuples must all verify with prepared keys"
            pk = {"verifies_per_s": round(world * n * steps / dtp, 1), "ms_per_step": round(dtp / steps * 1e3, 3),
                  "prepare_ms_once": round(t_prep * 1e3, 3), "table_bytes_per_key": engine.G2_PREPARED_BYTES,
                  "note": "G2AffineToPrepared once per key into HBM (blsmi_g2_prepare_batch_dev), then Miller loops that read a key's 68 line triples instead of "
                          "recomputing them; here every tuple has its own table (no reuse in cache); verdicts identical to the unprepared path (tests/test_gpu_prepared.py)"}
            if rank == 0:
                pk["roofline"] = roofline_of(profiled(E.lib, local_step_prepared), n, BYTES["verify"], E.ctr, lambda k: 2 * n if k.endswith("_pair") else n)
            out["g2pubs_prepared_keys"] = pk
            del tab
            if rank == 0 and world == 1:
                # the same tuples as the Go side holds them (blsmi 0.6): Jacobian records with z != 1 resident in HBM, ToAffine on the device inside the call
                from tools import reference_shapes as RS
                zs = RS.jac_zs()
                nkk = 256
                pkj = np.frombuffer(b"".join(RS.to_jac2(pks[i].tobytes(), (zs[i], zs[i + 1])) for i in range(nkk)) * (n // nkk), dtype=np.uint8)
                sgj = np.frombuffer(b"".join(RS.to_jac1(sigs[i].tobytes(), zs[i % 257]) for i in range(n)), dtype=np.uint8)
                assert np.array_equal(pks[:nkk], pks[nkk:2 * nkk]), "bench keys repeat every 256 tuples"
                d_pj = torch.from_numpy(pkj.copy()).to(dev); d_sj = torch.from_numpy(sgj.copy()).to(dev)

                def step_jac():
                    d_ok.zero_()
                    engine.verify_batch_jac_dev("g2pubs", d[0].data_ptr(), d[1].data_ptr(), d_pj.data_ptr(), d_sj.data_ptr(), d_ok.data_ptr(), n)
                dtj = timed_steps(E, step_jac, steps, warmup)
                assert bool(d_ok.all().item()), "synthetic tuples must all verify from their in-memory form"
                pj = profiled(E.lib, step_jac)
                out["g2pubs_in_memory_points"] = {"verifies_per_s": round(n * steps / dtj, 1), "ms_per_step": round(dtj / steps * 1e3, 3),
                                                  "to_affine_ms": {k: round(v[0], 4) for k, v in pj.items() if "jac_to_affine" in k},
                                                  "note": "keys and signatures resident as bls.G2Projective / bls.G1Projective records (z != 1): ToAffine + the wire form on the device, then the same kernels"}
                del d_pj, d_sj
        if group == "g1pubs":
            # VerifyWithDomain (g1pubs/bls.go:171-174): the same keys, 32-byte messages hashed by HashG2WithDomain (try-and-increment + ScaleByCofactor)
            nk = 256
            sk = b"".join(hashlib.sha256(b"bench-sk-%d" % i).digest()[:31].rjust(32, b"\0") for i in range(nk))
            m32 = [hashlib.sha256(b"domain-leg-%d-%d" % (rank, i)).digest() for i in range(n)]
            dom = bytes(range(8))
            hd = engine.hash_g2_with_domain_batch(m32, dom)
            sd, _ = engine.g2_mul_batch(hd.reshape(-1), sk * (n // nk), n)
            d_m = torch.from_numpy(np.frombuffer(b"".join(m32), dtype=np.uint8).copy()).to(dev)
            d_dom = torch.from_numpy(np.frombuffer(dom, dtype=np.uint8).copy()).to(dev)
            d_sd = torch.from_numpy(np.ascontiguousarray(sd)).to(dev)

            def local_step_domain():
                engine.g1pubs_verify_with_domain_batch_dev(d_m.data_ptr(), d_dom.data_ptr(), d[2].data_ptr(), d_sd.data_ptr(), 0, d_ok.data_ptr(), n)

            def step_domain():
                d_ok.zero_()
                local_step_domain()
                if E.use_dist:
                    full.zero_()
                    full[rank * n // 8:(rank + 1) * n // 8] = (d_ok.view(-1, 8).to(torch.int32) * weights).sum(dim=1, dtype=torch.int32)
                    dist.all_reduce(full, op=dist.ReduceOp.SUM)
            dtd = timed_steps(E, step_domain, steps, warmup)
            assert bool(d_ok.all().item()), "synthetic tuples must all verify (VerifyWithDomain)"
            wd = {"verifies_per_s": round(world * n * steps / dtd, 1), "ms_per_step": round(dtd / steps * 1e3, 3),
                  "note": "g1pubs.VerifyWithDomain on 32-byte messages, inputs resident (blsmi_g1pubs_verify_with_domain_batch_dev)"}
            if rank == 0:
                wd["roofline"] = roofline_of(profiled(E.lib, local_step_domain), n, BYTES["verify"], E.ctr, lambda k: 2 * n if k.endswith("_pair") else n)
            out["g1pubs_with_domain"] = wd
    out["note"] = "all tuples valid; inputs resident in HBM; hash-to-curve on the GPU included; 1 Verify = 2 Miller-loop pairs + 1 final exponentiation + 1 hash"
    return out


def mid_bench(E, steps=5, warmup=2):
    """Batches between the latency hand-over and a full chip (the reference's API is one tuple per call and its aggregate benchmarks use
    128 signers, g1pubs/verify_benchmark_test.go:33-85: real batches are not 65 536 tuples): pairings at 2 048 / 4 096 / 8 192 / 16 384 /
    32 768 and g2pubs / g1pubs verifies at 4 096 and 16 384, inputs resident, on the library's own choice of layout (one tuple per wave below
    2 048 tuples, per lane ROW -- round 6 -- up to 8 192, per lane QUAD up to 16 384, per lane pair beyond); every size's first 16 rows re-checked
    against the oracle."""
    import torch
    from oracle import refcpu as RC
    engine, dev = E.engine, E.dev
    nmax = 32768
    g1, g2 = synth_inputs(engine, nmax, seed=77)
    d1 = torch.from_numpy(g1).to(dev); d2 = torch.from_numpy(g2).to(dev); do = torch.zeros((nmax, 72), dtype=torch.int64, device=dev)
    want = RC.pairing_batch(g1[:16].tobytes(), g2[:16].tobytes(), 16)
    out = {"steps": steps, "pairings_per_s": {}, "pairing_ms": {}, "pairing_kernels": {}, "verifies_per_s": {}, "verify_ms": {}}
    for n in (2048, 4096, 8192, 16384, 32768):
        def step():
            engine.pairing_batch_dev(d1.data_ptr(), d2.data_ptr(), do.data_ptr(), n)
        dt = timed_steps(E, step, steps, warmup)
        assert np.array_equal(do[:16].cpu().numpy().view(np.uint64), want), "mid-size pairings differ from the oracle at n = %d" % n
        out["pairings_per_s"][str(n)] = round(n * steps / dt, 1); out["pairing_ms"][str(n)] = round(dt / steps * 1e3, 3)
        kern = {k: v[0] for k, v in profiled(E.lib, step).items() if not k.startswith("(")}
        out["pairing_kernels"][str(n)] = {k: round(v, 3) for k, v in kern.items()}
        # integer-issue fraction of this size's kernels: executed VALU wave-instructions of the committed PMC pass at THIS launch grid x 4 cycles over the
        # SIMDs that have a wave, against the measured kernel time (the bound that limits these kernels; the HBM roofline object is the headline's)
        issue = {}
        for k, ms in kern.items():
            c = counter_of(E.ctr, k, {"k_miller1h_row": 16 * n, "k_final_exp_row": 16 * n, "k_miller1h_quad": 4 * n, "k_final_exp_quad": 4 * n, "k_miller1h_pair": 2 * n, "k_final_exp_pair": 2 * n}.get(k))
            if c.get("SQ_INSTS_VALU") and c.get("grid") and ms > 0:
                waves = c["grid"] / 64.0
                busy_simds = min(float(SIMDS), waves)
                issue[k] = round(c["SQ_INSTS_VALU"] * CYCLES_PER_VALU / busy_simds / (CLOCK_GHZ * 1e9) * 1e3 / ms, 4)
        if issue:
            out.setdefault("valu_issue_frac", {})[str(n)] = issue
    nv = 16384
    out["verify_kernels"] = {}
    for group in ("g2pubs", "g1pubs"):
        packed, pks, sigs = _verify_tuples(engine, group, nv, tag=11)
        d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (packed.buf.copy(), packed.off.view(np.int64), pks, sigs)]
        d_ok = torch.zeros(nv, dtype=torch.uint8, device=dev)
        for n in (4096, nv):                                                # (the first n tuples of the same buffers: offsets are prefix sums)
            def vstep():
                engine.verify_batch_dev(group, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 0, d_ok.data_ptr(), n)
            d_ok.zero_()
            dt = timed_steps(E, vstep, steps, warmup)
            assert bool(d_ok[:n].all().item())
            out["verifies_per_s"]["%s@%d" % (group, n)] = round(n * steps / dt, 1); out["verify_ms"]["%s@%d" % (group, n)] = round(dt / steps * 1e3, 3)
            out["verify_kernels"]["%s@%d" % (group, n)] = {k: round(v[0], 3) for k, v in profiled(E.lib, vstep).items() if not k.startswith("(")}
    out["note"] = "round 3 (lane-pair kernels only): 16 384 pairings 11.6 ms (1.41 M/s), 32 768 12.6 ms; g2pubs verifies at 16 384 16.9 ms.  round 5 (no lane-row layout): 4 096 pairings 4.18 ms, 2 048: 2.49 ms, 4 096 g2pubs verifies 6.76 ms"
    return out


def msm_bench(E, n=1 << 20, steps=3, warmup=2):
    """BASELINE configs[2]: 2^20-point G1 and G2 scalar multiplication and MSM, inputs resident in HBM (rank 0 / device 0)."""
    import torch
    from oracle import refcpu as RC
    engine, dev = E.engine, E.dev
    rng = np.random.default_rng(3)
    k = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); k[:, 0] &= 0x3f
    base = 4096
    out = {"points": n, "steps": steps, "warmup": warmup, "note": "inputs and outputs resident in HBM (device-pointer entry points, blsmi 0.3); scalars uniform below 2^254; "
           "points = 4096 distinct multiples of the generator, tiled"}
    d_k = torch.from_numpy(k.reshape(-1)).to(dev)
    for grp, pb in (("g1", 96), ("g2", 192)):
        bk = rng.integers(0, 256, size=(base, 32), dtype=np.uint8); bk[:, 0] &= 0x3f
        bpts, _ = (engine.g1_mul_generator_batch if grp == "g1" else engine.g2_mul_generator_batch)(bk.reshape(-1), base)
        pts = np.ascontiguousarray(np.tile(bpts, (n // base, 1)))
        d_p = torch.from_numpy(pts.reshape(-1)).to(dev)
        d_out = torch.empty(n * pb, dtype=torch.uint8, device=dev); d_inf = torch.empty(n, dtype=torch.uint8, device=dev)
        d_one = torch.zeros(pb, dtype=torch.uint8, device=dev)

        def mul_step():
            engine.mul_batch_dev(grp, d_p.data_ptr(), d_k.data_ptr(), d_out.data_ptr(), d_inf.data_ptr(), n)

        def msm_step():
            engine.msm_dev(grp, d_p.data_ptr(), d_k.data_ptr(), n, d_one.data_ptr())
        for name, step, units_bytes in ((grp + "_mul", mul_step, BYTES[grp + "_mul"]), (grp + "_msm", msm_step, BYTES[grp + "_msm"])):
            dt = timed_steps(E, step, steps, warmup)
            prof = profiled(E.lib, step)
            out[name] = {"value": round(n * steps / dt, 1), "unit": "scalar multiplications/s" if name.endswith("mul") else "points/s",
                         "ms_per_step": round(dt / steps * 1e3, 3), "roofline": roofline_of(prof, n, units_bytes, E.ctr)}
        # parity gate of the leg: sampled multiples against the oracle, and MSM == sum of the multiples (device tree sum)
        got = d_out.view(n, pb)
        ref_mul = RC.g1_mul if grp == "g1" else RC.g2_mul
        for i in (0, 1, 4097, n // 2, n - 1):
            assert got[i].cpu().numpy().tobytes() == ref_mul(pts[i].tobytes(), k[i].tobytes()), "%s_mul row %d differs from the oracle" % (grp, i)
        d_sum = torch.zeros(pb, dtype=torch.uint8, device=dev)
        assert engine.sum_dev(grp, d_out.data_ptr(), 0, n, d_sum.data_ptr()) is False
        assert torch.equal(d_sum, d_one), "%s MSM differs from the sum of the per-point multiples" % grp
        if E.cpu:
            out[grp + "_mul"]["cpu_baseline"] = cpu_mul(grp, pts[:4096], k[:4096])
            out[grp + "_msm"]["cpu_baseline"] = dict(out[grp + "_mul"]["cpu_baseline"], note="the reference has no MSM: a caller loops MulFR and AddAssign; one MulFR per point is the CPU cost (the addition is <1 % of it)")
        del d_p, d_out
    return out


def aggregate_dev_bench(E, group, n, reps=3):
    """One n-message VerifyAggregate with messages and keys resident in HBM (blsmi_g*pubs_verify_aggregate_dev), rank 0 / device 0.
    Duplicate rejection on the device included.  Verdict True; a one-wrong-key run must say False (outside the timed region)."""
    import torch
    engine, dev = E.engine, E.dev
    packed, allpk, agg, pks = _aggregate_inputs(engine, group, 0, n)
    d_m = torch.from_numpy(packed.buf.copy()).to(dev); d_o = torch.from_numpy(packed.off.view(np.int64).copy()).to(dev); d_k = torch.from_numpy(allpk).to(dev)
    res = {}

    def step():
        res["ok"] = engine.verify_aggregate_dev(group, d_m.data_ptr(), d_o.data_ptr(), d_k.data_ptr(), agg, n)
    dt = timed_steps(E, step, reps, 2)
    assert res["ok"] is True, "the synthetic aggregate must verify"
    prof = profiled(E.lib, step)
    pkb = 192 if group == "g2pubs" else 96
    bad = d_k.clone(); bad[pkb * (n // 3):pkb * (n // 3 + 1)] = torch.from_numpy(pks[(n // 3 + 1) % pks.shape[0]].copy()).to(dev)
    assert engine.verify_aggregate_dev(group, d_m.data_ptr(), d_o.data_ptr(), bad.data_ptr(), agg, n) is False, "one wrong key must fail the aggregate"
    out = {"signatures": n, "ms": round(dt / reps * 1e3, 2), "signatures_per_s": round(n * reps / dt, 1),
           "roofline": roofline_of(prof, n, BYTES[group + "_aggregate"], E.ctr),
           "note": "messages (32 bytes each), offsets and keys resident in HBM; duplicate-message rejection on the device (keyed open-addressing table, exact comparisons) included; "
                   "n Miller loops (two tuples per loop) + Fq12 product tree + ONE final exponentiation; verdict True, and False with one key replaced"}
    if group == "g2pubs":
        # the same aggregate over PREPARED keys, one table per signer: n x 24 704 bytes resident (25.9 GB at n = 2^20 -- what 288 GB of HBM are for)
        del bad
        tab = torch.empty(n * engine.G2_PREPARED_BYTES, dtype=torch.uint8, device=dev)
        engine.g2_prepare_batch_dev(d_k.data_ptr(), n, tab.data_ptr())

        def step_prepared():
            res["okp"] = engine.g2pubs_verify_aggregate_prepared_dev(d_m.data_ptr(), d_o.data_ptr(), tab.data_ptr(), 0, agg, n)
        dtp = timed_steps(E, step_prepared, reps, 2)
        assert res["okp"] is True, "the synthetic aggregate must verify with prepared keys"
        profp = profiled(E.lib, step_prepared)
        idx = torch.arange(n, dtype=torch.int32, device=dev); idx[n // 3] = n // 3 + 1
        assert engine.g2pubs_verify_aggregate_prepared_dev(d_m.data_ptr(), d_o.data_ptr(), tab.data_ptr(), idx.data_ptr(), agg, n) is False, "one wrong key must fail the aggregate"
        out["prepared_keys"] = {"ms": round(dtp / reps * 1e3, 2), "signatures_per_s": round(n * reps / dtp, 1), "tables_GB": round(n * engine.G2_PREPARED_BYTES / 1e9, 2),
                                "roofline": roofline_of(profp, n, BYTES[group + "_aggregate"], E.ctr),
                                "note": "one prepared table per signer resident in HBM; the Miller loops read the lines (k_miller1x2_prep_pair)"}
        del tab
    if E.cpu:
        out["cpu_baseline"] = cpu_aggregate(engine, group)
    return out, (packed, allpk, agg)


def aggregate_bench(E, n_total=1 << 20, reps=2):
    """BASELINE configs[3]: one n_total-signature g2pubs VerifyAggregate over distinct 32-byte messages, block-sharded over the
    ranks (torchrun) -- per rank: duplicate screening, hash-to-curve, n/N Miller loops and the Fq12 product tree on its GPU; then
    the digests and the 576-byte partial products are all-gathered (bls_amd/dist.py) and every rank finishes with one final
    exponentiation -- or, in one process, through the library's split host entry point (blsmi_g2pubs_verify_aggregate).
    Host buffers (what a caller of the Go API holds), so PCIe is included."""
    import torch
    from bls_amd import dist as bdist
    engine, dev, rank, world, dist = E.engine, E.dev, E.rank, E.world, E.dist
    n = n_total // world
    packed, allpk, part, _ = _aggregate_inputs(engine, "g2pubs", rank * n, n)
    if E.use_dist:
        gather = bdist.torch_all_gather_bytes(dev)
        agg = engine.g1_sum(b"".join(gather(part)), world)
    else:
        gather, agg = None, part
    best, ok = 1e9, None
    for _ in range(reps + 1):                                              # first repetition warms the pools
        fence(E); t0 = time.perf_counter()
        if E.use_dist:
            ok = bdist.sharded_verify_aggregate("g2pubs", packed, allpk, agg, rank, world, gather)
        else:
            ok = engine.g2pubs_verify_aggregate(packed, allpk, agg)
        fence(E)
        best = min(best, max_over_ranks(E, time.perf_counter() - t0))
    assert ok is True, "the synthetic aggregate must verify"
    return {"signatures": n * world, "signatures_per_gpu": n, "ms": round(best * 1e3, 2), "signatures_per_s": round(n * world / best, 1),
            "exchange": ("all-gather of 8-byte keyed message fingerprints (global duplicate rejection: each rank screens 1/world of them, full keys only on suspicion) + all-gather of %d x 576-byte Fq12 partial products over RCCL" % world) if E.use_dist else "none (one GPU)",
            "note": "host buffers: PCIe included; min over %d repetitions; verdict True (false-verdict cases: tests/test_gpu_fullsize.py)" % reps}


def config0(E):
    """BASELINE configs[0]: 1 000 g2pubs tuples, every 16th corrupted, through g2pubs.Verify on the CPU restatement of the
    reference (the plumbing case) and, beside it, through the library: the verdict tables must be identical."""
    from oracle import refcpu as RC
    engine = E.engine
    n = 1000
    packed, pks, sigs = _verify_tuples(engine, "g2pubs", 1024, tag=7, nk=64)
    msgs = [bytes(packed.buf[int(packed.off[i]):int(packed.off[i + 1])]) for i in range(n)]
    pks = pks[:n].copy(); sigs = sigs[:n].copy()
    expect = np.ones(n, dtype=bool)
    for i in range(15, n, 16):
        kind = (i // 16) % 3
        expect[i] = False
        if kind == 0:
            msgs[i] = msgs[i] + b"!"
        elif kind == 1:
            pks[i] = pks[(i + 1) % 64]
        else:
            y = int.from_bytes(sigs[i, 48:].tobytes(), "big")
            sigs[i, 48:] = np.frombuffer(((RC_Q() - y) % RC_Q()).to_bytes(48, "big"), dtype=np.uint8)
    gpu_s = 1e9
    pm = engine.PackedMsgs(msgs)                                           # the form the C ABI takes (one buffer + offsets); packing 1 000 Python objects is the harness's work, not the call's
    a, b = np.ascontiguousarray(pks.reshape(-1)), np.ascontiguousarray(sigs.reshape(-1))
    for _ in range(5):                                                     # the call is blocking; best of five (a context's first calls size its temporaries)
        t0 = time.perf_counter()
        ok, _ = engine.g2pubs_verify_batch(pm, a, b)
        gpu_s = min(gpu_s, time.perf_counter() - t0)
    assert np.array_equal(ok, expect), "configs[0]: library verdicts differ from the corruption schedule"
    # the call's dominant kernel (HIP events of the library) against the algorithmic bytes of 1 000 verify tuples, like every other leg
    prof = profiled(E.lib, lambda: engine.g2pubs_verify_batch(pm, a, b))
    roof = roofline_of(prof, n, BYTES["verify"], E.ctr)
    from concurrent.futures import ThreadPoolExecutor
    cores = usable_cores()
    per = (n + cores - 1) // cores
    chunks = [(lo, min(n, lo + per)) for lo in range(0, n, per)]
    t0 = time.time()
    with ThreadPoolExecutor(cores) as ex:
        parts = list(ex.map(lambda c: RC.g2pubs.verify_batch(msgs[c[0]:c[1]], [pks[i].tobytes() for i in range(*c)], [sigs[i].tobytes() for i in range(*c)]), chunks))
    cpu_s = time.time() - t0
    cpu_ok = np.concatenate(parts)
    assert np.array_equal(cpu_ok, expect), "configs[0]: oracle verdicts differ from the corruption schedule"
    return {"workload": "1 000 (msg, G2 pubkey, G1 sig) tuples through g2pubs.Verify, every 16th corrupted (wrong message / wrong key / negated signature)",
            "cpu": {"value": round(n / cpu_s, 1), "unit": "verifies/s", "cores": cores, "kind": "port", "wall_s": round(cpu_s, 2),
                    "sample": "all 1 000 tuples on the C restatement of the reference (oracle/refcpu.c), %d threads" % cores},
            "gpu": {"value": round(n / gpu_s, 1), "unit": "verifies/s", "ms_one_call": round(gpu_s * 1e3, 2), "path": "host buffers (messages packed as the C ABI takes them), one call of 1 000 tuples (latency path: one tuple per wave), best of five calls"},
            "roofline": roof, "verdicts_identical": True, "rejected": int((~expect).sum())}


def RC_Q():
    from oracle import pyref as P
    return P.Q


def inlibrary_bench(E, ndev, n_per_dev=65536, steps=3):
    """One process, `ndev` devices behind the C ABI: the split HOST entry points.  Every call carries ndev x n_per_dev tuples in
    host memory; the library cuts it into one block per device, each block on its own host thread and stream, and -- for the
    verify batch with a bitmap -- completes the packed verdicts with ONE ncclAllReduce inside the call (the north-star's only
    collective).  PCIe-inclusive by construction."""
    engine = E.engine
    n = ndev * n_per_dev
    aliased = getattr(E, "alias", False)
    out = {"devices": engine.device_count(), "shards": engine.shard_count(), "tuples_per_call": n, "steps": steps, "rccl_ranks": ndev if ndev > 1 else 0,
           "collective": ("host-staged stand-in of ncclAllReduce under the BLSMI_DEVICE_ALIAS test hook (RCCL refuses two ranks on one GPU): same buffers, same result; rccl_ranks counts the stand-in's ranks" if aliased
                          else "ncclAllReduce(uint8 SUM, disjoint bit ownership) of the %d-byte bitmap inside every verify call" % (n // 8)) if ndev > 1 else "none (one device: the call is not split)"}
    if aliased:
        out["aliased_devices"] = True
    g1, g2 = synth_inputs(engine, n, seed=99)

    def best(fn):
        fn(); b = 1e9
        for _ in range(steps):
            t0 = time.perf_counter(); fn(); b = min(b, time.perf_counter() - t0)
        return b
    t = best(lambda: engine.pairing_batch(g1.reshape(-1), g2.reshape(-1), n))
    out["pairings_per_s"] = round(n / t, 1); out["pairing_ms_per_call"] = round(t * 1e3, 2)
    # the same call from page-locked buffers (blsmi_host_alloc): inputs written there once, the 576-byte results land there
    hb = [engine.HostBuffer(96 * n), engine.HostBuffer(192 * n), engine.HostBuffer(576 * n)]
    hb[0].a[:] = g1.reshape(-1); hb[1].a[:] = g2.reshape(-1)
    pin_out = hb[2].a.view(np.uint64).reshape(n, 72)
    ref = engine.pairing_batch(g1.reshape(-1), g2.reshape(-1), n)
    t = best(lambda: engine.pairing_batch(hb[0].a, hb[1].a, n, out=pin_out))
    assert np.array_equal(pin_out, ref), "page-locked buffers: different pairing values"
    out["pairings_per_s_pinned_host"] = round(n / t, 1); out["pairing_ms_per_call_pinned_host"] = round(t * 1e3, 2)
    del pin_out
    for b in hb:
        b.free()
    for group in ("g2pubs", "g1pubs"):
        packed, pks, sigs = _verify_tuples(engine, group, n, tag=5)
        fn = engine.g2pubs_verify_batch if group == "g2pubs" else engine.g1pubs_verify_batch
        res = {}

        def call():
            res["ok"], res["bm"] = fn(packed, pks.reshape(-1), sigs.reshape(-1))
        t = best(call)
        assert res["ok"].all() and (res["bm"] == 255).all(), "in-library split verify: every tuple must verify and every bitmap byte must be complete"
        out[group + "_verifies_per_s"] = round(n / t, 1); out[group + "_ms_per_call"] = round(t * 1e3, 2)
    return out


# ---------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairings", type=int, default=PAIRINGS_PER_GPU, help="pairings per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify-extra", action="store_true", help="headline only: skip every other leg")
    ap.add_argument("--no-aggregate", action="store_true")
    ap.add_argument("--no-ref-shapes", action="store_true")
    ap.add_argument("--no-msm", action="store_true")
    args = ap.parse_args()

    # The contract is ONE line on stdout.  Libraries print banners there (RCCL announces its version when the first
    # communicator is built), so stdout is pointed at stderr until the JSON line is ready.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    torchrun = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    rank = int(os.environ.get("RANK", "0")) if torchrun else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if torchrun else 0
    world = int(os.environ.get("WORLD_SIZE", "1")) if torchrun else 1
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X (no CPU fallback: the HIP path is the product)", file=sys.stderr)
        sys.exit(2)
    if torchrun and world != args.gpus:
        print("bench.py: launched with WORLD_SIZE=%d but --gpus %d" % (world, args.gpus), file=sys.stderr)
        sys.exit(2)
    single_process_devices = args.gpus if not torchrun else 1            # one process driving N devices through the library
    # TEST HOOK (include/blsmi.h: BLSMI_DEVICE_ALIAS=0,0[,...]): N LOGICAL devices on the listed physical GPUs, so that a one-GPU box runs the
    # N-device plumbing of this script and of the library end to end (tests/test_bench_line.py).  Not a measurement of scaling: the line says so.
    alias = [int(x) for x in os.environ.get("BLSMI_DEVICE_ALIAS", "").split(",") if x.strip() != ""] if not torchrun else []
    if alias and single_process_devices > len(alias):
        print("bench.py: --gpus %d but BLSMI_DEVICE_ALIAS names %d logical device(s)" % (args.gpus, len(alias)), file=sys.stderr)
        sys.exit(2)
    if not alias and single_process_devices > torch.cuda.device_count():
        print("bench.py: --gpus %d but only %d device(s) visible" % (args.gpus, torch.cuda.device_count()), file=sys.stderr)
        sys.exit(2)

    E = Env()
    E.rank, E.world, E.dist = rank, world, dist
    E.use_dist = torchrun and (world > 1 or bool(os.environ.get("BLSMI_BENCH_FORCE_DIST")))
    if not torchrun and os.environ.get("BLSMI_BENCH_FORCE_DIST"):          # exercise the RCCL path of the rank style on a 1-GPU box
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        E.use_dist = True
    torch.cuda.set_device(local_rank)
    E.dev = torch.device("cuda", local_rank)
    E.devs = [torch.device("cuda", alias[i] if alias else i) for i in range(single_process_devices)] if not torchrun else [E.dev]
    E.alias = bool(alias) and single_process_devices > 1
    if E.use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=E.dev)

    from bls_amd import engine, _native
    if torchrun:
        engine.init(local_rank)
    else:
        engine.init_devices(single_process_devices)
    lib = _native.load()
    E.engine, E.lib = engine, lib
    E.ctr = profile_counters()
    E.cpu = (world == 1 and single_process_devices == 1 and not args.no_cpu_baseline)
    n = args.pairings
    ndev = len(E.devs)

    # ---- headline: configs[1], one resident batch per device ----------------------------------------------------------
    g1, g2 = synth_inputs(engine, n, seed=rank)
    bufs = []
    for d in E.devs:
        bufs.append((torch.from_numpy(g1).to(d), torch.from_numpy(g2).to(d), torch.zeros((n, 72), dtype=torch.int64, device=d)))
    if E.alias:                                                            # aliased devices share one HIP ordinal: tell the library which logical device owns which buffer
        for i, tensors in enumerate(bufs):
            for t in tensors:
                engine.debug_alias_own(t.data_ptr(), t.numel() * t.element_size(), i)
    fence(E)

    def step_dev(i):
        a, b, o = bufs[i]
        engine.pairing_batch_dev(a.data_ptr(), b.data_ptr(), o.data_ptr(), n)

    def step():
        if ndev == 1:
            step_dev(0)
        else:                                                              # one host thread per device; the C call releases the interpreter lock
            th = [threading.Thread(target=step_dev, args=(i,)) for i in range(1, ndev)]
            for t in th:
                t.start()
            step_dev(0)
            for t in th:
                t.join()

    for _ in range(2):                                                     # set-up, not a step: the library sizes its per-context temporaries on a context's first two calls
        step()
    for _ in range(args.warmup):
        step()
    lib.blsmi_set_profiling(1)
    read_profile(lib)
    fence(E)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence(E)
    dt = time.perf_counter() - t0
    lib.blsmi_set_profiling(0)
    prof = read_profile(lib)                                               # the calling thread's device (device 0 / this rank), all timed steps
    dt = max_over_ranks(E, dt)
    total_gpus = world * ndev

    # parity gate, outside the timed region, on every rank and device: a fast kernel with different results is not a result
    from oracle import refcpu as RC
    # one whole workgroup's rows (the first 64 tuples) + 8 rows spread over the batch; the full 65 536 rows: tests/test_gpu_fullsize.py
    idx = sorted((set(range(64)) | {64, n // 3, n // 2, n // 2 + 1, n - 130, n - 65, n - 2, n - 1}) & set(range(n)))
    want = RC.pairing_batch(g1[idx].tobytes(), g2[idx].tobytes(), len(idx))
    good, bad_at = True, None
    for di, (_, _, o) in enumerate(bufs):
        got = o[idx].cpu().numpy().view(np.uint64)
        if not np.array_equal(got, want):
            good, bad_at = False, (di, [idx[k] for k in range(len(idx)) if not np.array_equal(got[k], want[k])][:4])
    flag = torch.tensor([0 if good else 1], dtype=torch.int32, device=E.dev)
    if E.use_dist:
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()):
        print("bench.py: SELF-CHECK FAILED: device output differs from the oracle's Pairing() (rank %d, device/row %s)" % (rank, bad_at), file=sys.stderr)
        sys.exit(3)
    checksum = int(bufs[0][2][::997].sum().item()) & 0xffffffff

    extras = {}

    def leg(name, fn):
        try:
            extras[name] = fn()
        except Exception as e:  # noqa: BLE001 -- an extra must never cost the headline line
            extras[name] = {"error": repr(e)[:400]}
    agg_dev_inputs = {}

    def pairing_prepared_leg():
        """the headline workload with the G2 arguments PREPARED (bls.MillerLoop's own calling convention: MillerLoopItem{P, *G2Prepared}): one
        table per tuple resident in HBM; output must equal the headline run's bit for bit"""
        a, b, o = bufs[0]
        tab = torch.empty(n * engine.G2_PREPARED_BYTES, dtype=torch.uint8, device=E.dev)
        t0 = time.perf_counter()
        engine.g2_prepare_batch_dev(b.data_ptr(), n, tab.data_ptr())
        t_prep = time.perf_counter() - t0
        o2 = torch.zeros_like(o)

        def stp():
            engine.pairing_batch_prepared_dev(a.data_ptr(), tab.data_ptr(), 0, o2.data_ptr(), n)
        dtp = timed_steps(E, stp, args.steps, 1)
        assert torch.equal(o, o2), "pairings over prepared keys differ from the headline run"
        return {"pairings_per_s": round(n * args.steps / dtp, 1), "ms_per_step": round(dtp / args.steps * 1e3, 3), "prepare_ms_once": round(t_prep * 1e3, 3),
                "prepare_points_per_s": round(n / t_prep, 1), "roofline": roofline_of(profiled(lib, stp), n, BYTES["pairing"], E.ctr, lambda k: 2 * n),
                "note": "G2AffineToPrepared once (blsmi_g2_prepare_batch_dev, the reference's BenchmarkG2Prepare shape), then Pairing over the resident tables; same Fq12 bits as the headline"}
    if rank == 0 and ndev == 1 and world == 1 and not args.no_verify_extra:
        leg("pairing_prepared", pairing_prepared_leg)
    if not args.no_verify_extra:
        if ndev == 1:                                                      # rank style (or one device): all ranks take part in the collectives
            leg("verify_bench", lambda: verify_bench(E))
            if not args.no_aggregate:
                leg("aggregate_bench", lambda: aggregate_bench(E))
        if rank == 0 and ndev == 1 and world == 1:
            if not args.no_msm:
                leg("msm_bench", lambda: msm_bench(E))
            if not args.no_aggregate:
                def _agg(group, m):
                    r, inp = aggregate_dev_bench(E, group, m)
                    agg_dev_inputs[group] = inp
                    return r
                leg("g2pubs_aggregate_dev_bench", lambda: _agg("g2pubs", 1 << 20))
                leg("g1pubs_aggregate_bench", lambda: _agg("g1pubs", 1 << 18))
            leg("config0", lambda: config0(E))
            leg("mid_batches", lambda: mid_bench(E))
        if not torchrun:                                                   # one process behind the C ABI: the split host entry points
            leg("inlibrary_bench", lambda: inlibrary_bench(E, ndev))
            if ndev > 1 and not args.no_aggregate:
                leg("aggregate_bench", lambda: aggregate_bench(E))

    if rank == 0:
        suffix = "" if os.environ.get("BLSMI_LAYOUT") == "single" else "_pair"
        kname = {"ml": "k_miller1h" + suffix, "fe": "k_final_exp" + suffix}
        ml = prof.get(kname["ml"], [0.0, 1]); fe = prof.get(kname["fe"], [0.0, 1])
        ml_ms, fe_ms = ml[0] / max(1, ml[1]), fe[0] / max(1, fe[1])
        grid = 2 * n if suffix else n
        value = total_gpus * n * args.steps / dt
        per_gpu = value / total_gpus
        ctr = E.ctr
        roof = roofline_of({k: [v[0] / max(1, v[1]), 1] for k, v in prof.items()}, n, BYTES["pairing"], ctr, lambda k: grid)
        if roof:
            roof["traffic_unit"] = ("bytes per launch: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 from the separate rocprofv3 --pmc passes of %s -- FETCH_SIZE counts half of the bytes read on "
                                    "gfx950 (calibration: profiles/r04_tcc_summary.txt; MI355X_MICROARCH.md); traffic_raw = FETCH_SIZE + WRITE_SIZE as printed" % ctr["file"])
            roof["kernel_ms"] = {kname["ml"]: round(ml_ms, 3), kname["fe"]: round(fe_ms, 3)}
            tr = {}
            for k in kname.values():
                c = counter_of(ctr, k, grid)
                tr[k] = hbm_traffic(c, n / 65536.0)[0]
            roof["traffic_all"] = tr
            roof["note"] = "864 algorithmic bytes per pairing: compute-bound by construction (SURVEY 8d); see valu"
        valu_insts = [counter_of(ctr, kname[k], grid).get("SQ_INSTS_VALU") for k in ("ml", "fe")]
        lane_instr = (sum(valu_insts) * LANES / 65536.0) if all(v is not None for v in valu_insts) else None
        issue_peak = SIMDS * LANES / CYCLES_PER_VALU * CLOCK_GHZ * 1e9       # lane-instructions per second per GPU
        mix = isa_mix()
        unit_of = {"k_pairing_pair.hip": kname["ml"], "k_fe_pair.hip": kname["fe"]}
        mixed = valu_mix_floor(ctr, mix, unit_of, grid, n) if suffix else None
        if mixed:
            per_k, floor_ms = mixed
            meas = {kname["ml"]: ml_ms, kname["fe"]: fe_ms}
            for k in per_k:
                per_k[k]["measured_ms"] = round(meas[k], 3)
                per_k[k]["frac"] = round(per_k[k]["min_ms"] / meas[k], 4) if meas[k] else None
            mix_obj = {"frac_mix": round(floor_ms / (ml_ms + fe_ms), 4) if (ml_ms + fe_ms) else None, "min_ms": round(floor_ms, 3), "measured_kernel_ms": round(ml_ms + fe_ms, 3),
                       "kernels": per_k, "cycles_per_class": mix["cycles_per_wave_instruction_per_simd"], "static_mix_file": mix["file"],
                       "note": "minimum time = sum over encoding classes of executed wave-instructions x measured cycles per class (%s) / %d SIMDs / %.1f GHz; "
                               "the flat `frac` charges every instruction %.0f cycles" % (mix["cycles_source"], SIMDS, CLOCK_GHZ, CYCLES_PER_VALU)}
        else:
            mix_obj = None
        line = {
            "metric": "BLS12-381 pairings/sec (batch verify)", "value": round(value, 1), "unit": "pairings/s",
            "n_gpus": total_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32 (14 x 28-bit limbs in the pairing / hash / curve kernels, 15 x 27 in the latency programs; int64 accumulate)",
            "data": "synthetic",
            "launch": "torchrun: one process per GPU" if torchrun else ("single process: %d device(s) behind the C ABI (blsmi_init_devices)" % ndev) +
                      (" -- LOGICAL devices on physical GPU(s) %s (BLSMI_DEVICE_ALIAS test hook, host-staged collectives): plumbing check, NO scaling curve was measured" % sorted(set(alias[:ndev])) if E.alias else ""),
            "devices": total_gpus, "rccl_ranks": world if E.use_dist else 0, "aliased_devices": True if E.alias else None,
            "scaling_measured": False if (E.alias and total_gpus > 1) else None,   # logical devices on one GPU: the N-device code ran, no scaling was measured
            "library": engine.version(),
            "config": {"workload": "configs[1]: %d independent pairings (Miller loop + final exponentiation = bls.Pairing) per GPU per step, inputs resident in HBM, output bit-exact Fq12" % n,
                       "pairings_per_gpu": n, "parallelism": "shard%d" % total_gpus,
                       "layout": "one tuple per lane" if suffix == "" else "lane pair per tuple (one Fq2 coefficient per lane), 2 waves/SIMD"},
            "self_check": {"rows_per_device": len(idx), "against": "oracle Pairing() (oracle/refcpu.c), bit-exact 576-byte Fq12", "passed": True},
            "roofline": roof,
            "valu": {"bound": "int32 VALU issue (v_mad_i64_i32): %d SIMDs x %d lanes / %.0f cycles x %.1f GHz" % (SIMDS, LANES, CYCLES_PER_VALU, CLOCK_GHZ),
                     "lane_instructions_per_pairing": None if lane_instr is None else round(lane_instr),
                     "lane_instructions_source": "SQ_INSTS_VALU (wave-instructions per 65 536-pairing launch) x 64 / 65 536, both pairing kernels, %s" % ctr["file"],
                     "achieved": None if lane_instr is None else round(per_gpu * lane_instr / 1e12, 3),
                     "peak": round(issue_peak / 1e12, 3), "unit": "T lane-instructions/s per GPU",
                     "frac": None if lane_instr is None else round(per_gpu * lane_instr / issue_peak, 4),
                     "frac_mix": mix_obj["frac_mix"] if mix_obj else None, "mix": mix_obj,
                     "nominal": {"achieved": round(per_gpu * FQ_MULS_PER_PAIRING / 1e9, 2), "peak": VALU_PEAK_GMULS, "unit": "G Fq-mul/s",
                                 "frac": round(per_gpu * FQ_MULS_PER_PAIRING / 1e9 / VALU_PEAK_GMULS, 4),
                                 "frac_of_2wave_per_simd_peak": round(per_gpu * FQ_MULS_PER_PAIRING / 1e9 / VALU_PEAK_2WAVE_GMULS, 4),
                                 "note": "pairings/s x 14.6k nominal Fq multiplications (SURVEY 8d) against the measured ceiling of the multiply core (profiles/r01_ubench2_fmul_15x27.log)"},
                     "note": "frac = measured VALU lane-instructions per second / issue peak: the instruction COUNT is a committed measurement (same command, PMC pass), the RATE is this run's"},
            "counters": {"file": ctr["file"], "commit": ctr.get("commit"), "source_digest": ctr.get("source_digest"), "stale": ctr["stale"],
                         "note": "traffic / VALU counts are read from this committed rocprofv3 PMC round; stale = a kernel source changed since it was taken"},
            "checksum": checksum,
        }
        line.update(extras)
        if world == 1 and ndev == 1 and not args.no_verify_extra and not args.no_ref_shapes:
            try:
                from tools import reference_shapes
                line["reference_shapes"] = reference_shapes.run(engine)
            except Exception as e:  # noqa: BLE001
                line["reference_shapes"] = {"error": repr(e)[:300]}
        if E.cpu:
            line["cpu_baseline"] = cpu_pairing(g1, g2)
        # ---- output: everything into bench_detail.json and onto stderr; the compact line is the LAST thing the run writes ----
        line["configs"] = compact_configs(line)
        detail_rel = write_detail(line)
        text = json.dumps(compact_line(line, detail_rel), separators=(",", ":"))
        assert len(text) < LINE_LIMIT, "bench.py: the stdout line is %d bytes (limit %d)" % (len(text), LINE_LIMIT)
        print("bench.py detail (also in %s):\n%s" % (detail_rel, json.dumps(line)), file=sys.stderr, flush=True)
    if E.use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush(); sys.stderr.flush()
        os.dup2(saved_stdout, 1)
        print(text, flush=True)
        os.dup2(2, 1)

if __name__ == "__main__":
    main()
