"""Per-kernel HIP-event times of the reference's benchmark call shapes (one tuple per call, 128-signer aggregates), host buffers:
where a small call's milliseconds go.  python tools/small_call_profile.py  (GPU box)"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bls_amd import engine as E, _native
from oracle import refcpu as RC

E.init(0)
lib = _native.load()
sk = [hashlib.sha256(b"refshape-%d" % i).digest()[:31].rjust(32, b"\0") for i in range(256)]
dom = bytes(range(8))
g1gen, g2gen = RC.g1_generator(), RC.g2_generator()
p1 = RC.g1_mul(g1gen, sk[0]); q1 = RC.g2_mul(g2gen, sk[1])
m = b"Hello world! 16 characters 0"
pk2 = RC.g2pubs.priv_to_pub(sk[2]); sg2 = RC.g2pubs.sign(m, sk[2])
m32 = hashlib.sha256(b"x").digest()
pk1 = RC.g1pubs.priv_to_pub(sk[3]); sg1 = RC.g1pubs.sign_with_domain(m32, sk[3], dom)
nsig = 128
mc = b"Some message".ljust(32, b"\0")
pk128 = b"".join(RC.g1pubs.priv_to_pub(s) for s in sk[:nsig])
sig_c = RC.g2_sum(b"".join(RC.g1pubs.sign_with_domain(mc, s, dom) for s in sk[:nsig]), nsig)
mm = [(b"Some message %d" % i).ljust(32, b"\0") for i in range(nsig)]
sig_m = RC.g2_sum(b"".join(RC.g1pubs.sign_with_domain(x, s, dom) for x, s in zip(mm, sk[:nsig])), nsig)
msgs2 = [b"Hello world! 16 characters %d" % i for i in range(nsig)]
pk2s = b"".join(RC.g2pubs.priv_to_pub(s) for s in sk[:nsig])
sig2 = RC.g1_sum(b"".join(RC.g2pubs.sign(x, s) for x, s in zip(msgs2, sk[:nsig])), nsig)
shapes = {
    "Pairing": lambda: E.pairing_batch(p1, q1, 1),
    "g2pubs.Verify": lambda: E.g2pubs_verify_batch([m], pk2, sg2),
    "g1pubs.VerifyWithDomain": lambda: E.g1pubs_verify_with_domain_batch([m32], dom, pk1, sg1),
    "g1pubs.VerifyAggregateCommonWithDomain(128)": lambda: E.g1pubs_verify_aggregate_common_with_domain(mc, dom, pk128, sig_c, nsig),
    "g1pubs.VerifyAggregateWithDomain(128)": lambda: E.g1pubs_verify_aggregate_with_domain(mm, dom, pk128, sig_m),
    "g2pubs.VerifyAggregate(128)": lambda: E.g2pubs_verify_aggregate(msgs2, pk2s, sig2),
}
for name, fn in shapes.items():
    r = fn(); fn()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
    prof = bench.profiled(lib, fn)
    ks = {k: round(v[0], 3) for k, v in prof.items()}
    print("%-46s %.3f ms wall; kernels %.3f ms: %s" % (name, best * 1e3, sum(v for k, v in ks.items() if not k.startswith("(")), ks), flush=True)
