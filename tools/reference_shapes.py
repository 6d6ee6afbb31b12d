"""The reference's own benchmark shapes, measured on the GPU library and -- beside it -- on the CPU restatement of the
reference algorithm (oracle/refcpu.c, ONE core: the reference's Go benchmarks are single-threaded).

    pairing_test.go:60-152                    BenchmarkG2Prepare / MillerLoop / FinalExponentiation / Pairing
    g2pubs/bls_test.go:215-256                BenchmarkBLSAggregateSignature / BLSSign / BLSVerify
    g1pubs/verify_benchmark_test.go:15-85     BenchmarkVerifyWithDomain / VerifyAggregateCommonWithDomain (128 signers) /
                                              VerifyAggregateMultipleWithDomain (128 signers, distinct messages)

For every shape: `cpu_ms_per_op` (what `go test -bench` would report per op, restated in C), `gpu_ms_single_call` (the
latency of ONE call through the C ABI with host buffers, the shape of the Go API) and, where the op batches,
`gpu_batch_ops_per_s` (one call carrying 65 536 of them, host buffers, PCIe included).  Called by bench.py (N = 1) and
runnable alone on the GPU box: python tools/reference_shapes.py
"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _best(fn, reps):
    fn()
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); b = min(b, time.perf_counter() - t0)
    return b


def RC_Q():
    return 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


_Q = RC_Q()
_R384 = (1 << 384) % _Q


def _m(v):
    return (v * _R384 % _Q).to_bytes(48, "little")


def to_jac1(w, z):
    """96-byte affine wire record -> the 144 bytes of a bls.G1Projective holding the same point with the given z (6 LE u64 Montgomery limbs per FQ)"""
    x, y = int.from_bytes(w[:48], "big"), int.from_bytes(w[48:96], "big")
    return _m(x * z * z % _Q) + _m(y * z * z * z % _Q) + _m(z)


def _f2(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % _Q, (a[0] * b[1] + a[1] * b[0]) % _Q)


def to_jac2(w, z):
    """192-byte affine wire record -> the 288 bytes of a bls.G2Projective"""
    c = [int.from_bytes(w[48 * i:48 * i + 48], "big") for i in range(4)]
    z2 = _f2(z, z); z3 = _f2(z2, z)
    X, Y = _f2((c[0], c[1]), z2), _f2((c[2], c[3]), z3)
    return _m(X[0]) + _m(X[1]) + _m(Y[0]) + _m(Y[1]) + _m(z[0]) + _m(z[1])


def jac_zs(k=257):
    return [int.from_bytes(hashlib.sha256(b"z%d" % i).digest() * 2, "big") % (_Q - 1) + 1 for i in range(k)]


def run(engine, batch=65536):
    from oracle import refcpu as RC                     # the CPU baseline leg (kind "port"): checker code, timed here as the reference's stand-in
    from bls_amd import g1pubs as G1P
    g1gen, g2gen = RC.g1_generator(), RC.g2_generator()
    sk = [hashlib.sha256(b"refshape-%d" % i).digest()[:31].rjust(32, b"\0") for i in range(256)]
    p1 = RC.g1_mul(g1gen, sk[0]); q1 = RC.g2_mul(g2gen, sk[1])
    out = {"cores": 1, "cpu_kind": "port (oracle/refcpu.c, gcc -O2, one thread)", "batch": batch, "shapes": {}}
    S = out["shapes"]
    # ---- pairing_test.go:60-152
    t_prep = _best(lambda: RC.g2_prepare(q1), 5)
    t_mlp = _best(lambda: RC.miller_loop(p1, q1, 1), 5)
    ml1 = RC.miller_loop(p1, q1, 1)
    t_fe = _best(lambda: RC.final_exponentiation(ml1), 3)
    t_pair = _best(lambda: RC.pairing_batch(p1, q1, 1), 3)
    nb = batch
    g1b, _ = engine.g1_mul_batch(g1gen * 256, b"".join(sk), 256); g2b, _ = engine.g2_mul_batch(g2gen * 256, b"".join(sk[::-1]), 256)
    G1 = np.ascontiguousarray(np.tile(g1b, (nb // 256, 1))).reshape(-1); G2 = np.ascontiguousarray(np.tile(g2b, (nb // 256, 1))).reshape(-1)
    assert np.array_equal(engine.g2_prepare_batch(q1, 1)[0], RC.g2_prepare(q1))
    S["G2Prepare"] = {"ref": "pairing_test.go:60-81", "cpu_ms_per_op": round(t_prep * 1e3, 4),
                      "gpu_ms_single_call": round(_best(lambda: engine.g2_prepare_batch(q1, 1), 3) * 1e3, 3),
                      "gpu_batch_ops_per_s": round(8192 / _best(lambda: engine.g2_prepare_batch(G2[:192 * 8192], 8192), 2), 1),
                      "gpu_note": "blsmi_g2_prepare_batch: the reference's 68 coefficient triples come back to the host (19.6 KB per point: the copy dominates the batch figure); "
                                  "blsmi_g2_prepare_batch_dev keeps them in HBM for the *_prepared entry points; the unprepared Miller-loop kernels fuse the preparation and never write it"}
    S["MillerLoop"] = {"ref": "pairing_test.go:83-108", "cpu_ms_per_op": round(max(t_mlp - t_prep, 0) * 1e3, 4), "cpu_ms_per_op_incl_prepare": round(t_mlp * 1e3, 4),
                       "gpu_ms_single_call": round(_best(lambda: engine.miller_loop_batch(p1, q1, 1), 3) * 1e3, 3),
                       "gpu_batch_ops_per_s": round(nb / _best(lambda: engine.miller_loop_batch(G1, G2, nb), 2), 1), "gpu_note": "includes the G2 preparation (fused)"}
    mlb = engine.miller_loop_batch(G1[:96 * 4096], G2[:192 * 4096], 4096)
    mlb = np.ascontiguousarray(np.tile(mlb, (nb // 4096, 1)))
    S["FinalExponentiation"] = {"ref": "pairing_test.go:110-131", "cpu_ms_per_op": round(t_fe * 1e3, 4),
                                "gpu_ms_single_call": round(_best(lambda: engine.final_exponentiation_batch(mlb[:1]), 3) * 1e3, 3),
                                "gpu_batch_ops_per_s": round(nb / _best(lambda: engine.final_exponentiation_batch(mlb), 2), 1)}
    S["Pairing"] = {"ref": "pairing_test.go:133-152", "cpu_ms_per_op": round(t_pair * 1e3, 4),
                    "gpu_ms_single_call": round(_best(lambda: engine.pairing_batch(p1, q1, 1), 3) * 1e3, 3),
                    "gpu_batch_ops_per_s": round(nb / _best(lambda: engine.pairing_batch(G1, G2, nb), 2), 1)}
    # ---- g2pubs/bls_test.go:215-256
    msg = b">16 character identical message"
    pk = RC.g2pubs.priv_to_pub(sk[5]); sig = RC.g2pubs.sign(msg, sk[5])
    t_ver = _best(lambda: RC.g2pubs.verify(msg, pk, sig), 3)
    t_sign = _best(lambda: RC.g2pubs.sign(b"Hello world! 16 characters 7", sk[6]), 3)
    t_add = _best(lambda: RC.g1_sum(sig + sig, 2), 5)
    msgs = [b"Hello world! 16 characters %d" % i for i in range(nb)]
    pm = engine.PackedMsgs(msgs)                          # (buffer + offsets, what the C ABI takes; packing Python objects is not what is measured)
    h = engine.hash_g1_batch(pm)
    sigs, _ = engine.g1_mul_batch(h.reshape(-1), b"".join(sk) * (nb // 256), nb)
    pks = np.ascontiguousarray(np.tile(engine.g2_mul_batch(g2gen * 256, b"".join(sk), 256)[0], (nb // 256, 1))).reshape(-1)
    assert engine.g2pubs_verify_batch([msg], pk, sig)[0][0] and engine.g2pubs_verify_batch(msgs[:4], pks[:4 * 192], sigs[:4].reshape(-1))[0].all()
    S["BLSVerify"] = {"ref": "g2pubs/bls_test.go:244-256", "cpu_ms_per_op": round(t_ver * 1e3, 4),
                      "gpu_ms_single_call": round(_best(lambda: engine.g2pubs_verify_batch([msg], pk, sig), 5) * 1e3, 3),
                      "gpu_batch_ops_per_s": round(nb / _best(lambda: engine.g2pubs_verify_batch(pm, pks, sigs.reshape(-1)), 2), 1)}

    # ---- the boundary itself: what handing its points over costs a Go caller, per point, on ONE host core ------------
    # The Go values hold Jacobian / Montgomery points (g2pubs/bls.go:13-15, 53-55).  Affine entry points: the shim runs ToAffine() +
    # SerializeBytes() per point (g1.go:322-340 + 157-167, g2.go:365-386 + 172-186) -- timed on the C restatement, n points in ONE C call (no
    # ctypes overhead per point).  In-memory (*_jac) entry points, blsmi 0.6: the shim copies the struct (144 / 288 bytes) and the library runs
    # ToAffine on the device inside the call.
    Q = RC_Q()
    zs = jac_zs()
    sg_bytes = sigs.reshape(nb, 96)
    sgj = np.frombuffer(b"".join(to_jac1(sg_bytes[i].tobytes(), zs[i % 257]) for i in range(nb)), dtype=np.uint8)
    pk256 = pks.reshape(nb, 192)[:256]
    pkj256 = b"".join(to_jac2(pk256[i].tobytes(), (zs[i], zs[i + 1])) for i in range(256))
    pkj = np.frombuffer(pkj256 * (nb // 256), dtype=np.uint8)
    m = 4096
    a1 = np.frombuffer(sgj[:144 * m].tobytes(), dtype=np.uint64); a2 = np.frombuffer(pkj[:288 * m].tobytes(), dtype=np.uint64)
    t_g1 = _best(lambda: RC.jac_to_affine_bytes_batch(1, a1, m), 2) / m
    t_g2 = _best(lambda: RC.jac_to_affine_bytes_batch(2, a2, m), 2) / m
    assert RC.jac_to_affine_bytes_batch(1, np.frombuffer(sgj[:144 * 8].tobytes(), dtype=np.uint64), 8).tobytes() == sg_bytes[:8].tobytes()
    dst1, dst2 = np.empty_like(sgj), np.empty_like(pkj)

    def struct_copies():                                  # what packSigs / packKeys of the shims do: one fixed-size copy per point
        a = sgj.reshape(nb, 144); b = pkj.reshape(nb, 288)
        d1 = dst1.reshape(nb, 144); d2 = dst2.reshape(nb, 288)
        for lo in range(0, nb, 4096):                      # (numpy row-block copies: an upper bound on what a Go copy loop costs per point)
            d1[lo:lo + 4096] = a[lo:lo + 4096]; d2[lo:lo + 4096] = b[lo:lo + 4096]
    t_copy = _best(struct_copies, 3) / nb
    # the messages cross as one buffer + offsets (the shims' packMsgs: one append per message).  Packing 65 536 Python bytes objects costs this
    # harness 10-15 ms -- its own cost, not the library's and not a Go caller's -- so the calls are timed on the packed form and the packing is
    # charged like the struct copies: as row-block copies of the same bytes
    t_pack_py = _best(lambda: engine.PackedMsgs(msgs), 2)
    mrows = np.zeros((nb, 32), dtype=np.uint8); mdst = np.empty_like(mrows)

    def msg_copies():
        for lo in range(0, nb, 4096):
            mdst[lo:lo + 4096] = mrows[lo:lo + 4096]
    t_copy += _best(msg_copies, 3) / nb
    got_j, _ = engine.g2pubs_verify_batch_jac(pm, pkj, sgj)
    assert bool(np.all(got_j))
    t_jac = _best(lambda: engine.g2pubs_verify_batch_jac(pm, pkj, sgj), 3)
    t_aff = _best(lambda: engine.g2pubs_verify_batch(pm, pks, sigs.reshape(-1)), 3)
    t_dev1 = _best(lambda: engine.g1_jac_to_affine_batch(sgj, nb), 2)
    t_dev2 = _best(lambda: engine.g2_jac_to_affine_batch(pkj, nb), 2)
    old_per_tuple = t_aff / nb + t_g1 + t_g2
    S["marshal"] = {"ref": "what crossing the C ABI costs the caller per point: G?Projective.ToAffine() + SerializeBytes() (g1.go:322-340, 157-167; g2.go:365-386, 172-186) "
                           "for the affine entry points against one struct copy for the *_jac entry points (blsmi 0.6)",
                    "cpu_us_per_point_to_affine_bytes": {"g1": round(t_g1 * 1e6, 3), "g2": round(t_g2 * 1e6, 3)},
                    "jac_host_us_per_tuple_struct_copies": round(t_copy * 1e6, 4),
                    "jac_to_affine_batch_host_call_points_per_s": {"g1": round(nb / t_dev1, 1), "g2": round(nb / t_dev2, 1)},
                    "note": "cpu: oracle/refcpu.c on one core, n points per C call; struct copies: 144 + 288 bytes per g2pubs tuple; "
                            "jac_to_affine_batch: blsmi_g?_jac_to_affine_batch from host buffers (copies in and out included)"}
    out["end_to_end_host"] = {
        "what": "g2pubs.VerifyBatch of %d tuples as a Go caller reaches it: host buffers, marshalling included, one call" % nb,
        "jac_entry_verifies_per_s": round(nb / (t_jac + t_copy * nb), 1),
        "jac_entry_ms": round(t_jac * 1e3, 3), "jac_marshal_ms_one_core": round(t_copy * nb * 1e3, 3),
        "affine_entry_verifies_per_s_one_marshalling_core": round(1.0 / old_per_tuple, 1),
        "affine_entry_ms": round(t_aff * 1e3, 3), "affine_marshal_ms_one_core": round((t_g1 + t_g2) * nb * 1e3, 1),
        "affine_entry_verifies_per_s_without_marshalling": round(nb / t_aff, 1),
        "python_harness_message_packing_ms_not_counted": round(t_pack_py * 1e3, 2),
        "note": "the affine figure adds the measured per-point ToAffine + SerializeBytes of one host core (the shims' packKeys / packSigs before blsmi 0.6 were one "
                "goroutine) to the measured call; the jac figure adds the measured struct copies and message appends (as block copies) to the measured blsmi_g2pubs_verify_batch_jac call; "
                "both calls take the messages already packed (buffer + offsets): packing 65 536 Python bytes objects is the cost of this harness, listed and not counted"}

    def sign_batch(ms, sks):                              # g2pubs.Sign = sk * HashG1(m) (g2pubs/bls.go:132-135): one call, both steps on the device
        return engine.g2pubs_sign_batch(ms, sks)
    S["BLSSign"] = {"ref": "g2pubs/bls_test.go:227-242", "cpu_ms_per_op": round(t_sign * 1e3, 4),
                    "gpu_ms_single_call": round(_best(lambda: sign_batch(msgs[:1], sk[6]), 3) * 1e3, 3),
                    "gpu_batch_ops_per_s": round(nb / _best(lambda: sign_batch(pm, b"".join(sk) * (nb // 256)), 2), 1)}
    S["BLSAggregateSignature"] = {"ref": "g2pubs/bls_test.go:215-225", "cpu_ms_per_op": round(t_add / 2 * 1e3, 5), "cpu_note": "one Jacobian addition (two per timed call incl. the affine conversion)",
                                  "gpu_batch_ops_per_s": round(nb / _best(lambda: engine.g1_sum(sigs.reshape(-1), nb), 2), 1), "gpu_note": "tree sum of 65 536 signatures in one call"}
    # ---- g1pubs/verify_benchmark_test.go:15-85
    dom = bytes([42, 0, 0, 0, 0, 0, 0, 0])
    m32 = b"Some msg".ljust(32, b"\0")
    pk1 = RC.g1pubs.priv_to_pub(sk[9]); sg1 = RC.g1pubs.sign_with_domain(m32, sk[9], dom)
    t_vwd = _best(lambda: RC.g1pubs.verify_with_domain(m32, pk1, sg1, dom), 3)
    assert engine.g1pubs_verify_with_domain_batch([m32], dom, pk1, sg1)[0]
    S["VerifyWithDomain"] = {"ref": "g1pubs/verify_benchmark_test.go:15-31", "cpu_ms_per_op": round(t_vwd * 1e3, 4),
                             "gpu_ms_single_call": round(_best(lambda: engine.g1pubs_verify_with_domain_batch([m32], dom, pk1, sg1), 5) * 1e3, 3)}
    # the same shape as one call of 65 536 distinct 32-byte messages (signatures made on the device: sk * HashG2WithDomain(m))
    m32s = [hashlib.sha256(i.to_bytes(4, "little")).digest() for i in range(nb)]
    sk_all = b"".join(sk) * (nb // 256)
    pk_b, _ = engine.g1_mul_generator_batch(sk_all, nb)
    hd = engine.hash_g2_with_domain_batch(m32s, dom)
    sg_b, _ = engine.g2_mul_batch(hd.reshape(-1), sk_all, nb)
    assert sg_b[9].tobytes() == RC.g1pubs.sign_with_domain(m32s[9], sk[9], dom)
    okb = engine.g1pubs_verify_with_domain_batch(m32s, dom, pk_b.reshape(-1), sg_b.reshape(-1))
    assert bool(np.all(okb))
    S["VerifyWithDomain"]["gpu_batch_ops_per_s"] = round(nb / _best(lambda: engine.g1pubs_verify_with_domain_batch(m32s, dom, pk_b.reshape(-1), sg_b.reshape(-1)), 2), 1)
    nsig = 128
    mc = b"Some message".ljust(32, b"\0")
    pk128 = [RC.g1pubs.priv_to_pub(s) for s in sk[:nsig]]
    sig_c = RC.g2_sum(b"".join(RC.g1pubs.sign_with_domain(mc, s, dom) for s in sk[:nsig]), nsig)
    t_vac = _best(lambda: RC.g1pubs.verify_aggregate_common_with_domain(sig_c, pk128, mc, dom), 2)
    assert engine.g1pubs_verify_aggregate_common_with_domain(mc, dom, b"".join(pk128), sig_c, nsig) is True
    S["VerifyAggregateCommonWithDomain_128"] = {"ref": "g1pubs/verify_benchmark_test.go:33-56", "cpu_ms_per_op": round(t_vac * 1e3, 3),
                                                "gpu_ms_single_call": round(_best(lambda: engine.g1pubs_verify_aggregate_common_with_domain(mc, dom, b"".join(pk128), sig_c, nsig), 3) * 1e3, 3)}
    mm = [(b"Some message %d" % i).ljust(32, b"\0") for i in range(nsig)]
    sig_m = RC.g2_sum(b"".join(RC.g1pubs.sign_with_domain(m, s, dom) for m, s in zip(mm, sk[:nsig])), nsig)
    t_vam = _best(lambda: RC.g1pubs.verify_aggregate_with_domain(sig_m, pk128, mm, dom), 1)
    assert engine.g1pubs_verify_aggregate_with_domain(mm, dom, b"".join(pk128), sig_m) is True
    S["VerifyAggregateMultipleWithDomain_128"] = {"ref": "g1pubs/verify_benchmark_test.go:58-85", "cpu_ms_per_op": round(t_vam * 1e3, 3),
                                                  "gpu_ms_single_call": round(_best(lambda: engine.g1pubs_verify_aggregate_with_domain(mm, dom, b"".join(pk128), sig_m), 3) * 1e3, 3)}
    return out


if __name__ == "__main__":
    import json
    from bls_amd import engine
    engine.init(0)
    print(json.dumps(run(engine), indent=1))
