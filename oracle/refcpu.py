"""ctypes binding of the C oracle (oracle/librefcpu.so).  TEST INFRASTRUCTURE ONLY -- imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by bls_amd/."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librefcpu.so")


def build():
    """Compile the oracle with gcc (no GPU, no reference sources needed)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "librefcpu.so"])


def _stale():
    if not os.path.exists(_SO):
        return True
    t = os.path.getmtime(_SO)
    return any(os.path.getmtime(os.path.join(_HERE, f)) > t for f in ("refcpu.c", "refcpu_curve.inc") if os.path.exists(os.path.join(_HERE, f)))


def _load():
    if _stale():
        build()
    return C.CDLL(_SO)


lib = _load()
_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)


def _a64(x, n=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.uint64).reshape(-1))
    if n is not None:
        assert a.size == n, (a.size, n)
    return a


def _p64(a):
    return a.ctypes.data_as(_u64p)


def _b(x):
    """bytes-like -> contiguous uint8 array"""
    if isinstance(x, np.ndarray):
        return np.ascontiguousarray(x.astype(np.uint8, copy=False).reshape(-1))
    return np.frombuffer(bytes(x), dtype=np.uint8).copy() if len(x) else np.zeros(1, dtype=np.uint8)


def _p8(a):
    return a.ctypes.data_as(_u8p)


for _name in dir():
    pass
lib.rc_fq_inverse.restype = C.c_int
lib.rc_fq_sqrt.restype = C.c_int
for _n in ["rc_fq2_inverse", "rc_fq2_sqrt", "rc_fq6_inverse", "rc_fq12_inverse", "rc_final_exponentiation", "rc_g1_jac_to_affine_bytes",
           "rc_g2_jac_to_affine_bytes", "rc_g1_mul", "rc_g2_mul", "rc_g1_sum", "rc_g2_sum", "rc_g1_decompress", "rc_g2_decompress", "rc_hash_g2",
           "rc_g2_prepare", "rc_pairing_batch", "rc_g2pubs_verify", "rc_g2pubs_verify_aggregate", "rc_g2pubs_verify_aggregate_common",
           "rc_g1pubs_verify", "rc_g1pubs_verify_with_domain", "rc_g1pubs_verify_aggregate", "rc_g1pubs_verify_aggregate_common",
           "rc_g1pubs_verify_aggregate_common_with_domain", "rc_g1pubs_verify_aggregate_with_domain",
           "rc_fq_cmp", "rc_fq_parity", "rc_fq2_cmp", "rc_fq2_parity"]:
    getattr(lib, _n).restype = C.c_int


# ---- limb-level -----------------------------------------------------------------------------
def multiply_fqrepr(a, b):
    a, b = _a64(a, 6), _a64(b, 6)
    hi, lo = np.zeros(6, np.uint64), np.zeros(6, np.uint64)
    lib.rc_multiply_fqrepr(_p64(a), _p64(b), _p64(hi), _p64(lo))
    return hi, lo


def mont_reduce(hi, lo):
    hi, lo = _a64(hi, 6), _a64(lo, 6)
    out = np.zeros(6, np.uint64)
    lib.rc_mont_reduce(_p64(hi), _p64(lo), _p64(out))
    return out


def _prim(fn, *args):
    o, c = C.c_uint64(), C.c_uint64()
    fn(*[C.c_uint64(int(a)) for a in args], C.byref(o), C.byref(c))
    return o.value, c.value


def mac_with_carry(a, b, c, carry):
    return _prim(lib.rc_mac_with_carry, a, b, c, carry)


def add_with_carry(a, b, carry):
    return _prim(lib.rc_add_with_carry, a, b, carry)


def sub_with_borrow(a, b, borrow):
    return _prim(lib.rc_sub_with_borrow, a, b, borrow)


def _op(fn, nout, *ins, ret=False, extra=()):
    arrs = [_a64(x) for x in ins]
    out = np.zeros(nout, np.uint64)
    r = fn(*[_p64(a) for a in arrs], *extra, _p64(out))
    return (r, out) if ret else out


def fq_from_repr(r): return _op(lib.rc_fq_from_repr, 6, r)
def fq_to_repr(a): return _op(lib.rc_fq_to_repr, 6, a)
def fq_add(a, b): return _op(lib.rc_fq_add, 6, a, b)
def fq_sub(a, b): return _op(lib.rc_fq_sub, 6, a, b)
def fq_mul(a, b): return _op(lib.rc_fq_mul, 6, a, b)
def fq_sqr(a): return _op(lib.rc_fq_sqr, 6, a)
def fq_neg(a): return _op(lib.rc_fq_neg, 6, a)
def fq_dbl(a): return _op(lib.rc_fq_dbl, 6, a)
def fq_inverse(a): return _op(lib.rc_fq_inverse, 6, a, ret=True)
def fq_sqrt(a): return _op(lib.rc_fq_sqrt, 6, a, ret=True)
def fq2_add(a, b): return _op(lib.rc_fq2_add, 12, a, b)
def fq2_sub(a, b): return _op(lib.rc_fq2_sub, 12, a, b)
def fq2_mul(a, b): return _op(lib.rc_fq2_mul, 12, a, b)
def fq2_sqr(a): return _op(lib.rc_fq2_sqr, 12, a)
def fq2_neg(a): return _op(lib.rc_fq2_neg, 12, a)
def fq2_dbl(a): return _op(lib.rc_fq2_dbl, 12, a)
def fq2_mul_nr(a): return _op(lib.rc_fq2_mul_nr, 12, a)
def fq2_inverse(a): return _op(lib.rc_fq2_inverse, 12, a, ret=True)
def fq2_sqrt(a): return _op(lib.rc_fq2_sqrt, 12, a, ret=True)
def fq2_frobenius(a, p): return _op(lib.rc_fq2_frobenius, 12, a, extra=(C.c_uint(p),))
def fq6_mul(a, b): return _op(lib.rc_fq6_mul, 36, a, b)
def fq6_sqr(a): return _op(lib.rc_fq6_sqr, 36, a)
def fq6_inverse(a): return _op(lib.rc_fq6_inverse, 36, a, ret=True)
def fq6_frobenius(a, p): return _op(lib.rc_fq6_frobenius, 36, a, extra=(C.c_uint(p),))
def fq12_mul(a, b): return _op(lib.rc_fq12_mul, 72, a, b)
def fq12_sqr(a): return _op(lib.rc_fq12_sqr, 72, a)
def fq12_inverse(a): return _op(lib.rc_fq12_inverse, 72, a, ret=True)
def fq12_frobenius(a, p): return _op(lib.rc_fq12_frobenius, 72, a, extra=(C.c_uint(p),))
def fq12_mul_by_014(a, c0, c1, c4): return _op(lib.rc_fq12_mul_by_014, 72, a, c0, c1, c4)
def fq6_mul_by_1(a, c1): return _op(lib.rc_fq6_mul_by_1, 36, a, c1)
def fq6_mul_by_01(a, c0, c1): return _op(lib.rc_fq6_mul_by_01, 36, a, c0, c1)
def fq_cmp(a, b): return int(lib.rc_fq_cmp(_p64(_a64(a, 6)), _p64(_a64(b, 6))))
def fq_parity(a): return bool(lib.rc_fq_parity(_p64(_a64(a, 6))))
def fq2_cmp(a, b): return int(lib.rc_fq2_cmp(_p64(_a64(a, 12)), _p64(_a64(b, 12))))
def fq2_parity(a): return bool(lib.rc_fq2_parity(_p64(_a64(a, 12))))
def fq12_exp_u64(a, e): return _op(lib.rc_fq12_exp_u64, 72, a, extra=(C.c_uint64(e),))
def final_exponentiation(a): return _op(lib.rc_final_exponentiation, 72, a, ret=True)
def g1_double(p): return _op(lib.rc_g1_double, 18, p)
def g1_add(p, q): return _op(lib.rc_g1_add, 18, p, q)
def g2_double(p): return _op(lib.rc_g2_double, 36, p)
def g2_add(p, q): return _op(lib.rc_g2_add, 36, p, q)


def g1_jac_to_affine_bytes(p):
    a = _a64(p, 18); out = np.zeros(96, np.uint8)
    inf = lib.rc_g1_jac_to_affine_bytes(_p64(a), _p8(out))
    return None if inf else out.tobytes()


def g2_jac_to_affine_bytes(p):
    a = _a64(p, 36); out = np.zeros(192, np.uint8)
    inf = lib.rc_g2_jac_to_affine_bytes(_p64(a), _p8(out))
    return None if inf else out.tobytes()


def jac_to_affine_bytes_batch(group, recs, n):
    """n in-memory Jacobian records -> n wire records, one C call (the per-point host cost of the affine boundary, bench.py `marshal`)"""
    w, pb = (18, 96) if group == 1 else (36, 192)
    a = _a64(recs, w * n); out = np.zeros(pb * n, np.uint8)
    (lib.rc_g1_jac_to_affine_bytes_batch if group == 1 else lib.rc_g2_jac_to_affine_bytes_batch)(_p64(a), _p8(out), C.c_size_t(n))
    return out


# ---- byte-level -----------------------------------------------------------------------------
def sha256(msg):
    m = _b(msg); out = np.zeros(32, np.uint8)
    lib.rc_sha256(_p8(m), C.c_size_t(len(msg)), _p8(out))
    return out.tobytes()


def g1_generator():
    out = np.zeros(96, np.uint8); lib.rc_g1_generator(_p8(out)); return out.tobytes()


def g2_generator():
    out = np.zeros(192, np.uint8); lib.rc_g2_generator(_p8(out)); return out.tobytes()


def g1_mul(p, k):
    a, s, out = _b(p), _b(k), np.zeros(96, np.uint8)
    inf = lib.rc_g1_mul(_p8(a), _p8(s), _p8(out))
    return None if inf else out.tobytes()


def g2_mul(p, k):
    a, s, out = _b(p), _b(k), np.zeros(192, np.uint8)
    inf = lib.rc_g2_mul(_p8(a), _p8(s), _p8(out))
    return None if inf else out.tobytes()


def g1_sum(pts, n, inf_flags=None):
    a, out = _b(pts), np.zeros(96, np.uint8)
    f = _b(inf_flags) if inf_flags is not None else None
    inf = lib.rc_g1_sum(_p8(a), _p8(f) if f is not None else None, C.c_size_t(n), _p8(out))
    return None if inf else out.tobytes()


def g2_sum(pts, n, inf_flags=None):
    a, out = _b(pts), np.zeros(192, np.uint8)
    f = _b(inf_flags) if inf_flags is not None else None
    inf = lib.rc_g2_sum(_p8(a), _p8(f) if f is not None else None, C.c_size_t(n), _p8(out))
    return None if inf else out.tobytes()


def g1_compress(p, inf=False):
    a, out = _b(p if p is not None else bytes(96)), np.zeros(48, np.uint8)
    lib.rc_g1_compress(_p8(a), C.c_int(int(inf or p is None)), _p8(out)); return out.tobytes()


def g2_compress(p, inf=False):
    a, out = _b(p if p is not None else bytes(192)), np.zeros(96, np.uint8)
    lib.rc_g2_compress(_p8(a), C.c_int(int(inf or p is None)), _p8(out)); return out.tobytes()


def g1_decompress(c, checked=True):
    """-> (err, point bytes | None for infinity)"""
    a, out, inf = _b(c), np.zeros(96, np.uint8), C.c_int(0)
    e = lib.rc_g1_decompress(_p8(a), C.c_int(int(checked)), _p8(out), C.byref(inf))
    return e, (None if (e or inf.value) else out.tobytes())


def g2_decompress(c, checked=True):
    a, out, inf = _b(c), np.zeros(192, np.uint8), C.c_int(0)
    e = lib.rc_g2_decompress(_p8(a), C.c_int(int(checked)), _p8(out), C.byref(inf))
    return e, (None if (e or inf.value) else out.tobytes())


def hash_g1(msg):
    m, out = _b(msg), np.zeros(96, np.uint8)
    lib.rc_hash_g1(_p8(m), C.c_size_t(len(msg)), _p8(out)); return out.tobytes()


def hash_g2(msg):
    m, out = _b(msg), np.zeros(192, np.uint8)
    ok = lib.rc_hash_g2(_p8(m), C.c_size_t(len(msg)), _p8(out)); assert ok; return out.tobytes()


def hash_g2_with_domain(msg32, domain8):
    m, d, out = _b(msg32), _b(domain8), np.zeros(192, np.uint8)
    lib.rc_hash_g2_with_domain(_p8(m), _p8(d), _p8(out)); return out.tobytes()


def hash_secret_key(b32):
    m, out = _b(b32), np.zeros(32, np.uint8)
    lib.rc_hash_secret_key(_p8(m), _p8(out)); return out.tobytes()


def g2_prepare(q):
    a, out = _b(q), np.zeros(68 * 3 * 12, np.uint64)
    n = lib.rc_g2_prepare(_p8(a), _p64(out)); assert n == 68
    return out.reshape(68, 3, 12)


def miller_loop(g1s, g2s, n):
    a, b, out = _b(g1s), _b(g2s), np.zeros(72, np.uint64)
    lib.rc_miller_loop(_p8(a), _p8(b), C.c_size_t(n), _p64(out)); return out


def pairing_batch(g1s, g2s, n):
    a, b, out = _b(g1s), _b(g2s), np.zeros(72 * n, np.uint64)
    rc = lib.rc_pairing_batch(_p8(a), _p8(b), _p64(out), C.c_size_t(n)); assert rc == 0
    return out.reshape(n, 72)


def _msgs(msgs):
    off = np.zeros(len(msgs) + 1, np.uint64)
    off[1:] = np.cumsum([len(m) for m in msgs])
    return _b(b"".join(msgs) or b"\0"), off


class g2pubs:
    @staticmethod
    def priv_to_pub(sk):
        s, out = _b(sk), np.zeros(192, np.uint8); lib.rc_g2pubs_priv_to_pub(_p8(s), _p8(out)); return out.tobytes()

    @staticmethod
    def sign(msg, sk):
        m, s, out = _b(msg), _b(sk), np.zeros(96, np.uint8); lib.rc_g2pubs_sign(_p8(m), C.c_size_t(len(msg)), _p8(s), _p8(out)); return out.tobytes()

    @staticmethod
    def verify(msg, pk, sig):
        m, p, s = _b(msg), _b(pk or bytes(192)), _b(sig or bytes(96))
        return bool(lib.rc_g2pubs_verify(_p8(m), C.c_size_t(len(msg)), _p8(p), C.c_int(pk is None), _p8(s), C.c_int(sig is None)))

    @staticmethod
    def verify_aggregate(sig, pks, msgs):
        if len(pks) != len(msgs):
            return False
        mb, off = _msgs(msgs); p, s = _b(b"".join(pks) or b"\0"), _b(sig)
        return bool(lib.rc_g2pubs_verify_aggregate(_p8(mb), _p64(off), _p8(p), _p8(s), C.c_size_t(len(msgs))))

    @staticmethod
    def verify_aggregate_common(sig, pks, msg):
        m, p, s = _b(msg), _b(b"".join(pks) or b"\0"), _b(sig)
        return bool(lib.rc_g2pubs_verify_aggregate_common(_p8(m), C.c_size_t(len(msg)), _p8(p), _p8(s), C.c_size_t(len(pks))))

    @staticmethod
    def verify_batch(msgs, pks, sigs, inf_flags=None):
        n = len(msgs); mb, off = _msgs(msgs); p, s = _b(b"".join(pks)), _b(b"".join(sigs)); ok = np.zeros(n, np.uint8)
        f = _b(inf_flags) if inf_flags is not None else None
        lib.rc_g2pubs_verify_batch(_p8(mb), _p64(off), _p8(p), _p8(s), _p8(f) if f is not None else None, _p8(ok), C.c_size_t(n))
        return ok.astype(bool)


class g1pubs:
    @staticmethod
    def priv_to_pub(sk):
        s, out = _b(sk), np.zeros(96, np.uint8); lib.rc_g1pubs_priv_to_pub(_p8(s), _p8(out)); return out.tobytes()

    @staticmethod
    def sign(msg, sk):
        m, s, out = _b(msg), _b(sk), np.zeros(192, np.uint8); lib.rc_g1pubs_sign(_p8(m), C.c_size_t(len(msg)), _p8(s), _p8(out)); return out.tobytes()

    @staticmethod
    def sign_with_domain(msg32, sk, domain8):
        m, s, d, out = _b(msg32), _b(sk), _b(domain8), np.zeros(192, np.uint8)
        lib.rc_g1pubs_sign_with_domain(_p8(m), _p8(s), _p8(d), _p8(out)); return out.tobytes()

    @staticmethod
    def verify(msg, pk, sig):
        m, p, s = _b(msg), _b(pk or bytes(96)), _b(sig or bytes(192))
        return bool(lib.rc_g1pubs_verify(_p8(m), C.c_size_t(len(msg)), _p8(p), C.c_int(pk is None), _p8(s), C.c_int(sig is None)))

    @staticmethod
    def verify_with_domain(msg32, pk, sig, domain8):
        m, p, s, d = _b(msg32), _b(pk), _b(sig), _b(domain8)
        return bool(lib.rc_g1pubs_verify_with_domain(_p8(m), _p8(p), C.c_int(0), _p8(s), C.c_int(0), _p8(d)))

    @staticmethod
    def verify_aggregate(sig, pks, msgs):
        if len(pks) != len(msgs):
            return False
        mb, off = _msgs(msgs); p, s = _b(b"".join(pks) or b"\0"), _b(sig)
        return bool(lib.rc_g1pubs_verify_aggregate(_p8(mb), _p64(off), _p8(p), _p8(s), C.c_size_t(len(msgs))))

    @staticmethod
    def verify_aggregate_common(sig, pks, msg):
        m, p, s = _b(msg), _b(b"".join(pks) or b"\0"), _b(sig)
        return bool(lib.rc_g1pubs_verify_aggregate_common(_p8(m), C.c_size_t(len(msg)), _p8(p), _p8(s), C.c_size_t(len(pks))))

    @staticmethod
    def verify_aggregate_common_with_domain(sig, pks, msg32, domain8):
        m, p, s, d = _b(msg32), _b(b"".join(pks)), _b(sig), _b(domain8)
        return bool(lib.rc_g1pubs_verify_aggregate_common_with_domain(_p8(m), _p8(p), _p8(s), C.c_size_t(len(pks)), _p8(d)))

    @staticmethod
    def verify_aggregate_with_domain(sig, pks, msgs32, domain8):
        if len(pks) != len(msgs32):
            return False
        m, p, s, d = _b(b"".join(msgs32)), _b(b"".join(pks)), _b(sig), _b(domain8)
        return bool(lib.rc_g1pubs_verify_aggregate_with_domain(_p8(m), _p8(p), _p8(s), C.c_size_t(len(pks)), _p8(d)))

    @staticmethod
    def verify_batch(msgs, pks, sigs, inf_flags=None):
        n = len(msgs); mb, off = _msgs(msgs); p, s = _b(b"".join(pks)), _b(b"".join(sigs)); ok = np.zeros(n, np.uint8)
        f = _b(inf_flags) if inf_flags is not None else None
        lib.rc_g1pubs_verify_batch(_p8(mb), _p64(off), _p8(p), _p8(s), _p8(f) if f is not None else None, _p8(ok), C.c_size_t(n))
        return ok.astype(bool)
