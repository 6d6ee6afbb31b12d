import sys, time, ctypes as C, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bls_amd import _native
from oracle import pyref as P, refcpu as RC
lib = _native.load()
print("init", lib.blsmi_init(0), lib.blsmi_version())
u64p = C.POINTER(C.c_uint64); u8p = C.POINTER(C.c_uint8)
def mont(v): return np.array(P.limbs64(P.to_mont(v)), dtype=np.uint64)
def unmont(l): return P.from_mont(P.from_limbs64(l))
xs = P.XORShift(11)
def dbg(op, a, b, width, n):
    a = np.ascontiguousarray(a, dtype=np.uint64); out = np.zeros_like(a); flag = np.zeros(n, np.uint8)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, dtype=np.uint64); bp = b.ctypes.data_as(u64p)
    rc = lib.blsmi_debug_op(op, a.ctypes.data_as(u64p), bp, out.ctypes.data_as(u64p), flag.ctypes.data_as(u8p), C.c_size_t(n))
    assert rc == 0, rc
    return out, flag
n = 70
A = [P.rand_int(xs, P.Q) for _ in range(n)]; B = [P.rand_int(xs, P.Q) for _ in range(n)]
A[0] = 0; A[1] = 1; A[2] = P.Q - 1; B[2] = P.Q - 1
a = np.concatenate([mont(v) for v in A]); b = np.concatenate([mont(v) for v in B])
for name, op, f in [("mul", 1, lambda x, y: x * y % P.Q), ("sqr", 2, lambda x, y: x * x % P.Q), ("add", 3, lambda x, y: (x + y) % P.Q), ("sub", 4, lambda x, y: (x - y) % P.Q), ("neg", 5, lambda x, y: -x % P.Q)]:
    out, _ = dbg(op, a, b, 1, n)
    got = [unmont(out[6 * i:6 * i + 6]) for i in range(n)]
    exp = [f(x, y) for x, y in zip(A, B)]
    print("fq", name, got == exp)
    if got != exp:
        for i in range(n):
            if got[i] != exp[i]: print(i, hex(A[i]), hex(B[i]), hex(got[i]), hex(exp[i])); break
t = time.time(); out, fl = dbg(6, a, None, 1, n); print("fq inv", [unmont(out[6 * i:6 * i + 6]) for i in range(n)] == [pow(x, P.Q - 2, P.Q) for x in A], time.time() - t)
# fq12 mul
def r12(): return [P.rand_int(xs, P.Q) for _ in range(12)]
def to6(v): return tuple((v[2 * j], v[2 * j + 1]) for j in range(3))
def to12(v): return (to6(v[:6]), to6(v[6:]))
X = [r12() for _ in range(n)]; Y = [r12() for _ in range(n)]
xa = np.concatenate([mont(v) for r in X for v in r]); ya = np.concatenate([mont(v) for r in Y for v in r])
out, _ = dbg(48, xa, ya, 12, n)
got = [[unmont(out[72 * i + 6 * e:72 * i + 6 * e + 6]) for e in range(12)] for i in range(n)]
print("fq12 mul", got == [P.fq12_flat(P.fq12_mul(to12(x), to12(y))) for x, y in zip(X, Y)])
out, _ = dbg(49, xa, None, 12, n); got = [[unmont(out[72 * i + 6 * e:72 * i + 6 * e + 6]) for e in range(12)] for i in range(n)]
print("fq12 sqr", got == [P.fq12_flat(P.fq12_sqr(to12(x))) for x in X])
out, _ = dbg(50, xa, None, 12, n); got = [[unmont(out[72 * i + 6 * e:72 * i + 6 * e + 6]) for e in range(12)] for i in range(n)]
print("fq12 inv", got == [P.fq12_flat(P.fq12_inv(to12(x))) for x in X])
for op, pw in [(51, 1), (52, 2), (53, 3)]:
    out, _ = dbg(op, xa, None, 12, n); got = [[unmont(out[72 * i + 6 * e:72 * i + 6 * e + 6]) for e in range(12)] for i in range(n)]
    print("fq12 frob", pw, got == [P.fq12_flat(P.fq12_frob(to12(x), pw)) for x in X])
# pairing
m = 66
g1 = b""; g2 = b""; exp = []
for i in range(m):
    aa, bb = P.rand_fr(xs), P.rand_fr(xs)
    if i == 0: aa, bb = 1, 1
    pa = RC.g1_mul(RC.g1_generator(), aa.to_bytes(32, "big")); qb = RC.g2_mul(RC.g2_generator(), bb.to_bytes(32, "big"))
    g1 += pa; g2 += qb
g1a = np.frombuffer(g1, np.uint8).copy(); g2a = np.frombuffer(g2, np.uint8).copy()
out = np.zeros(72 * m, np.uint64)
t = time.time(); rc = lib.blsmi_miller_loop_batch(g1a.ctypes.data_as(u8p), g2a.ctypes.data_as(u8p), out.ctypes.data_as(u64p), C.c_size_t(m)); print("miller rc", rc, time.time() - t)
ref = np.stack([RC.miller_loop(g1[96 * i:96 * i + 96], g2[192 * i:192 * i + 192], 1) for i in range(m)])
print("miller loop bit-exact:", np.array_equal(out.reshape(m, 72), ref))
t = time.time(); rc = lib.blsmi_pairing_batch(g1a.ctypes.data_as(u8p), g2a.ctypes.data_as(u8p), out.ctypes.data_as(u64p), C.c_size_t(m)); print("pairing rc", rc, time.time() - t)
ref = RC.pairing_batch(g1, g2, m)
print("pairing bit-exact:", np.array_equal(out.reshape(m, 72), ref))
for n2 in [65536]:
    reps = (n2 + m - 1) // m
    G1 = np.tile(g1a, reps)[:96 * n2].copy(); G2 = np.tile(g2a, reps)[:192 * n2].copy(); O = np.zeros(72 * n2, np.uint64)
    t = time.time(); rc = lib.blsmi_pairing_batch(G1.ctypes.data_as(u8p), G2.ctypes.data_as(u8p), O.ctypes.data_as(u64p), C.c_size_t(n2)); dt = time.time() - t
    print("pairing n=%d rc=%d %.3fs -> %.0f pairings/s (host buffers)" % (n2, rc, dt, n2 / dt), np.array_equal(O.reshape(n2, 72)[:m], ref))
