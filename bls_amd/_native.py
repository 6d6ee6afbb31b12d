"""Loader / builder for the HIP extension libblsmi.so (the product path; no CPU fallback).

build() compiles bls_amd/csrc/blsmi.hip for gfx950 with hipcc (cross-compiles without a GPU).
load() dlopens the in-tree library and fails loudly when it is missing or has no device.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(_HERE, "libblsmi.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "blsmi.h")
# translation units of libblsmi.so: the host side + one unit per kernel family, compiled in parallel
_UNITS = ["blsmi.hip", "k_pairing_pair.hip", "k_fe_pair.hip", "k_pairing_quad.hip", "k_pairing_row.hip", "k_prepared_pair.hip", "k_pairing_single.hip", "k_fe_single.hip", "k_fq12_single.hip", "k_hash.hip", "k_wire.hip", "k_hash_pair.hip", "k_hash_quad.hip", "k_curve.hip", "k_msm_pair.hip", "k_lat.hip", "k_util.hip"]
LAT_BIN = os.path.join(CSRC, "lat_programs.z")            # level programs of the latency path (gen_lat.py), zlib-compressed, embedded into blsmi.hip.o
BUILD_DIR = os.path.join(CSRC, "build")
# -Werror=pass-failed: a kernel that misses its declared waves-per-SIMD (a shared device function that outgrew the register budget)
# stops the build instead of running at half occupancy
_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Werror=pass-failed"]
# The lane-pair / lane-quad pairing kernels compute in 14 x 28-bit limbs (fp.cuh: BLSMI_LIMBS28; 196 instead of 225 multiply-adds per
# product), every other unit in 15 x 27; buffers that cross between kernels keep the 27-bit form.  BLSMI_BUILD_LIMBS27=1 builds those
# units in 15 x 27 as well (A/B).
_LIMBS28_UNITS = () if os.environ.get("BLSMI_BUILD_LIMBS27") else ("k_pairing_pair.hip", "k_fe_pair.hip", "k_pairing_quad.hip", "k_pairing_row.hip", "k_prepared_pair.hip", "k_hash_pair.hip", "k_hash_quad.hip", "k_hash.hip", "k_wire.hip", "k_curve.hip", "k_msm_pair.hip")


# rough compile cost in seconds (scheduling order only)
_COST = {"k_fe_single.hip": 45, "k_hash.hip": 40, "k_curve.hip": 50, "k_pairing_quad.hip": 45, "k_pairing_row.hip": 15, "k_pairing_single.hip": 40, "k_pairing_pair.hip": 32, "k_wire.hip": 30,
         "k_fq12_single.hip": 30, "k_fe_pair.hip": 17, "k_hash_pair.hip": 17, "k_prepared_pair.hip": 16, "blsmi.hip": 5, "k_msm_pair.hip": 13, "k_lat.hip": 8, "k_util.hip": 20}


def _unit_flags(u):
    return ["-DBLSMI_LIMBS28"] if u in _LIMBS28_UNITS else []


def _deps(unit):
    """Prerequisites of one object file, from the depfile the previous compilation left (else: everything in csrc/)."""
    dfile = os.path.join(BUILD_DIR, unit + ".d")
    if os.path.exists(dfile):
        txt = open(dfile).read().replace("\\\n", " ")
        deps = [t for t in txt.split(":", 1)[1].split() if t]
        if all(os.path.exists(t) for t in deps):
            return deps
        return None                                       # a prerequisite vanished: rebuild
    return None


def _unit_stale(unit):
    obj = os.path.join(BUILD_DIR, unit + ".o")
    if not os.path.exists(obj):
        return True
    try:                                                  # built with other flags (the unit changed representation, an A/B build): rebuild
        if open(obj + ".flags").read() != " ".join(_FLAGS + _unit_flags(unit)):
            return True
    except OSError:
        return True
    deps = _deps(unit)
    if deps is None:
        return True
    if unit == "blsmi.hip":
        deps = deps + [LAT_BIN]                           # .incbin: not in the compiler's depfile
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(x) > t for x in deps)


def _stale():
    if not os.path.exists(SO_PATH):
        return True
    if not os.path.isdir(BUILD_DIR):                      # a shipped .so without its objects (the GPU box): trust mtimes of the sources
        t = os.path.getmtime(SO_PATH)
        srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cuh", ".inc", ".h"))] + [HEADER]
        return any(os.path.getmtime(x) > t for x in srcs)
    t = os.path.getmtime(SO_PATH)
    return any(_unit_stale(u) or os.path.getmtime(os.path.join(BUILD_DIR, u + ".o")) > t for u in _UNITS)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> bls_amd/libblsmi.so (in-tree so that it travels to the GPU box).  The units are
    compiled concurrently (one hipcc process each) and only when a prerequisite changed."""
    consts = os.path.join(CSRC, "consts.cuh")
    gen = os.path.join(CSRC, "gen_consts.py")
    if not os.path.exists(consts) or os.path.getmtime(gen) > os.path.getmtime(consts):
        subprocess.check_call(["python3", gen])
    consts28 = os.path.join(CSRC, "consts28.cuh")
    if not os.path.exists(consts28) or os.path.getmtime(gen) > os.path.getmtime(consts28):
        subprocess.check_call(["python3", gen, "--limbs28"])
    genlat = os.path.join(CSRC, "gen_lat.py")
    latgen = None                                          # the level programs take ~7 s to generate and only blsmi.hip needs them: generated BESIDE the other units' compilation
    if not os.path.exists(LAT_BIN) or os.path.getmtime(genlat) > os.path.getmtime(LAT_BIN):
        latgen = subprocess.Popen(["python3", genlat])
    genmul = os.path.join(CSRC, "gen_lat_mul.py")
    mulinc = os.path.join(CSRC, "lat_mul.inc")
    if not os.path.exists(mulinc) or os.path.getmtime(genmul) > os.path.getmtime(mulinc):
        subprocess.check_call(["python3", genmul])
    if latgen is None and not force and not _stale():
        return SO_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(BUILD_DIR, exist_ok=True)
    # longest units first, as many at a time as there are cores (one hipcc process each): ~55 s from scratch on 8 cores
    todo = [u for u in sorted(_UNITS, key=lambda x: -_COST.get(x, 20)) if force or u == "blsmi.hip" and latgen is not None or _unit_stale(u)]
    slots = max(2, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 4))
    procs, running, failed = [], [], []

    def reap(block):
        import time
        while True:
            for item in list(running):
                if item[1].poll() is not None:
                    running.remove(item)
                    if item[1].returncode != 0:
                        failed.append(item[0])
            if not block or len(running) < slots:
                return
            time.sleep(0.05)
    for u in todo:
        if u == "blsmi.hip" and latgen is not None:          # (its .incbin and lat_programs.h come out of the generator)
            if latgen.wait() != 0:
                raise subprocess.CalledProcessError(1, "python3 gen_lat.py")
        reap(block=True)
        obj = os.path.join(BUILD_DIR, u + ".o")
        cmd = [hipcc] + _FLAGS + _unit_flags(u) + ["-c", "-MD", "-MF", os.path.join(BUILD_DIR, u + ".d"), "-o", obj, os.path.join(CSRC, u)]
        if u == "blsmi.hip":
            cmd.insert(1, '-DBLSMI_LAT_BIN="%s"' % LAT_BIN)
        if verbose:
            print(" ".join(cmd))
        p = subprocess.Popen(cmd)
        procs.append((u, p)); running.append((u, p))
    for u, p in procs:
        p.wait()
    reap(block=False)
    if failed:
        raise subprocess.CalledProcessError(1, "hipcc -c " + " ".join(failed))
    for u, _ in procs:
        with open(os.path.join(BUILD_DIR, u + ".o.flags"), "w") as f:
            f.write(" ".join(_FLAGS + _unit_flags(u)))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden", "-o", SO_PATH] + [os.path.join(BUILD_DIR, u + ".o") for u in _UNITS] + ["-lz"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO_PATH


_lib = None


class NativeError(RuntimeError):
    pass


def _prefer_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so.7 (and HSA runtime) under
    torch/lib; libblsmi.so names the same SONAME with /opt/rocm on its runpath.  Whichever copy is mapped first
    serves both, and a process that maps /opt/rocm's first and imports torch afterwards ends up with torch unable
    to see the GPU ("ProcessGroupNCCL ... no GPUs found").  When torch is installed, map its copy first -- without
    importing torch -- so the load order of `bls_amd` and `torch` no longer matters.  Without torch (the Go/cgo
    deployment) the runpath copy is used."""
    import sys
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:  # noqa: BLE001 -- best effort: fall back to the runpath copy
        pass


def load():
    """Return the ctypes handle of libblsmi.so; never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    alt = os.environ.get("BLSMI_LIB")                      # development only: an A/B build of the same library (tools/build_variant.sh)
    if alt:
        _prefer_torch_hip_runtime()
        _lib = C.CDLL(os.path.abspath(alt))
        _lib.blsmi_version.restype = C.c_char_p
        return _lib
    if not os.path.exists(SO_PATH):
        raise NativeError("bls_amd/libblsmi.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    _prefer_torch_hip_runtime()
    _lib = C.CDLL(SO_PATH)
    _lib.blsmi_version.restype = C.c_char_p
    return _lib


def declared_symbols():
    """Entry points declared in include/blsmi.h."""
    import re
    txt = open(HEADER).read()
    return sorted(set(re.findall(r"\b(blsmi_[a-z0-9_]+)\s*\(", txt)))
