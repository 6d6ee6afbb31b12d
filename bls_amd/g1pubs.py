"""Host-side mirror of the reference package g1pubs (g1pubs/bls.go): PublicKey in G1, Signature in
G2, messages hashed to G2.  Same names, argument meaning and results as the Go API; every group,
pairing and hash operation runs in the HIP kernels of libblsmi.so (no CPU fallback).

    Verify(m, pub, sig)                         g1pubs/bls.go:165-168
    sig.VerifyAggregate(pubKeys, msgs)          g1pubs/bls.go:252-282
    sig.VerifyAggregateCommon(pubKeys, msg)     g1pubs/bls.go:287-290
    AggregateSignatures / AggregatePublicKeys   g1pubs/bls.go:177-204
    DeserializeSignature / DeserializePublicKey g1pubs/bls.go:33-40, 89-96
plus VerifyBatch, the batch form the one-tuple-per-call Go API lacks.
"""
from . import engine
from ._groups import DeserializeError, Point, all_in_memory, point_sum  # noqa: F401

SIG_GROUP, PK_GROUP = 2, 1


class Signature:
    def __init__(self, point):
        self.s = point

    def Serialize(self):                      # g1pubs/bls.go:18-20
        return self.s.serialize()

    def Copy(self):
        return Signature(self.s.copy())

    def Aggregate(self, other):               # g1pubs/bls.go:174-177
        self.s = point_sum([self.s, other.s], SIG_GROUP)

    def VerifyAggregate(self, pubKeys, msgs):
        if len(pubKeys) != len(msgs):          # g1pubs/bls.go:241-243
            return False
        if self.s.infinity or any(p.p.infinity for p in pubKeys):
            return False                       # the reference panics in MillerLoop on infinity; defined as false here
        if all_in_memory([p.p for p in pubKeys] + [self.s]):   # the points as the Go values hold them: ToAffine on the device
            return engine.g1pubs_verify_aggregate_jac(msgs, b"".join(p.p.jac for p in pubKeys), self.s.jac)
        return engine.g1pubs_verify_aggregate(msgs, b"".join(p.p.raw for p in pubKeys), self.s.raw)

    def VerifyAggregateCommon(self, pubKeys, msg):
        if all_in_memory([p.p for p in pubKeys] + [self.s]):   # key sum and Verify in one library call, Jacobian points summed as they are
            return engine.g1pubs_verify_aggregate_common_jac(msg, b"".join(p.p.jac for p in pubKeys), self.s.jac, len(pubKeys))
        return Verify(msg, AggregatePublicKeys(pubKeys), self)


class PublicKey:
    def __init__(self, point):
        self.p = point

    def Serialize(self):                      # g1pubs/bls.go:67-69
        return self.p.serialize()

    def Copy(self):
        return PublicKey(self.p.copy())

    def Equals(self, other):
        return self.p == other.p

    def Aggregate(self, other):               # g1pubs/bls.go:189-192
        self.p = point_sum([self.p, other.p], PK_GROUP)


def NewSignatureFromG2(raw192):
    return Signature(Point(raw192, SIG_GROUP))


def NewPublicKeyFromG1(raw96):
    return PublicKey(Point(raw96, PK_GROUP))


def NewSignatureFromG2Projective(jac288):
    """a Signature holding its point the way the reference's does (g1pubs/bls.go:13-15): the 288 bytes of a *bls.G2Projective"""
    return Signature(Point(None, SIG_GROUP, jac=jac288))


def NewPublicKeyFromG1Projective(jac144):
    """g1pubs/bls.go:53-55: the 144 bytes of a *bls.G1Projective"""
    return PublicKey(Point(None, PK_GROUP, jac=jac144))


def DeserializeSignature(b96):
    return Signature(Point.deserialize(b96, SIG_GROUP))


def DeserializePublicKey(b48):
    return PublicKey(Point.deserialize(b48, PK_GROUP))


def NewAggregateSignature():
    return Signature(Point(None, SIG_GROUP))


def NewAggregatePubkey():
    return PublicKey(Point(None, PK_GROUP))


def AggregateSignatures(sigs):
    return Signature(point_sum([s.s for s in sigs], SIG_GROUP))


def AggregatePublicKeys(pubs):
    return PublicKey(point_sum([p.p for p in pubs], PK_GROUP))


def VerifyBatch(msgs, pubs, sigs):
    """[Verify(msgs[i], pubs[i], sigs[i]) for i] in one launch sequence."""
    n = len(msgs)
    if not (len(pubs) == len(sigs) == n):
        raise ValueError("length mismatch")
    if n == 0:
        return []
    if all_in_memory([p.p for p in pubs] + [s.s for s in sigs]):
        ok, _ = engine.g1pubs_verify_batch_jac(msgs, b"".join(p.p.jac for p in pubs), b"".join(s.s.jac for s in sigs))
        return [bool(x) for x in ok]
    flags = [(1 if p.p.infinity else 0) | (2 if s.s.infinity else 0) for p, s in zip(pubs, sigs)]
    ok, _ = engine.g1pubs_verify_batch(msgs, b"".join(p.p.bytes_or_zero() for p in pubs), b"".join(s.s.bytes_or_zero() for s in sigs), flags)
    return [bool(x) for x in ok]


def VerifySerializedBatch(msgs, pub_bytes, sig_bytes):
    """DeserializePublicKey + DeserializeSignature + Verify per tuple, in one device pass over the 48-byte keys and
    96-byte signatures of the wire format.  A tuple whose key or signature does not deserialise (the reference returns
    an error there and Verify is never reached) or is the point at infinity yields False."""
    ok, _, _ = engine.verify_serialized_batch(__name__.rsplit(".", 1)[-1], msgs, b"".join(pub_bytes), b"".join(sig_bytes), True)
    return [bool(x) for x in ok]


def Verify(m, pub, sig):
    return VerifyBatch([m], [pub], [sig])[0]


def Sign(message, key):
    """Sign(message, key) (g1pubs/bls.go:132-135): key = the secret scalar as 32 big-endian bytes (SecretKey.Serialize()).  One call is
    one hash-to-curve plus one windowed multiplication on the latency path; not side-channel hardened (include/blsmi.h)."""
    return SignBatch([message], [key])[0]


def PrivToPub(k):
    """PrivToPub(k) (g1pubs/bls.go:144-146): k = the secret scalar as 32 big-endian bytes."""
    return PrivToPubBatch([k])[0]


def PrivToPubBatch(secret_scalars):
    """pk_i = sk_i * generator (PrivToPub, g1pubs/bls.go:144-146); scalars are 32-byte big-endian."""
    n = len(secret_scalars)
    out, inf = engine.g1_mul_generator_batch(b"".join(secret_scalars), n)
    return [PublicKey(Point(None if inf[i] else out[i].tobytes(), PK_GROUP)) for i in range(n)]


def SignBatch(msgs, secret_scalars):
    """sigma_i = sk_i * HashG2(m_i) (Sign, g1pubs/bls.go:132-135); scalars are 32-byte big-endian."""
    n = len(msgs)
    out, inf = engine.g1pubs_sign_batch(msgs, b"".join(secret_scalars))    # one call: hash, then multiply, on the device
    return [Signature(Point(None if inf[i] else out[i].tobytes(), SIG_GROUP)) for i in range(n)]


# ---- the *WithDomain family (g1pubs/bls.go:138-141, 171-174, 294-311) ---------------------------------
def VerifyWithDomainBatch(msgs32, pubs, sigs, domain8):
    n = len(msgs32)
    if n == 0:
        return []
    if all_in_memory([p.p for p in pubs] + [s.s for s in sigs]):
        return [bool(x) for x in engine.g1pubs_verify_with_domain_batch_jac(msgs32, domain8, b"".join(p.p.jac for p in pubs), b"".join(s.s.jac for s in sigs))]
    flags = [(1 if p.p.infinity else 0) | (2 if s.s.infinity else 0) for p, s in zip(pubs, sigs)]
    ok = engine.g1pubs_verify_with_domain_batch(msgs32, domain8, b"".join(p.p.bytes_or_zero() for p in pubs), b"".join(s.s.bytes_or_zero() for s in sigs), flags)
    return [bool(x) for x in ok]


def VerifyWithDomain(m32, pub, sig, domain8):
    return VerifyWithDomainBatch([m32], [pub], [sig], domain8)[0]


def VerifyAggregateCommonWithDomain(sig, pubKeys, msg32, domain8):
    if all_in_memory([p.p for p in pubKeys] + [sig.s]):
        return engine.g1pubs_verify_aggregate_common_with_domain_jac(msg32, domain8, b"".join(p.p.jac for p in pubKeys), sig.s.jac, len(pubKeys))
    return VerifyWithDomain(msg32, AggregatePublicKeys(pubKeys), sig, domain8)


def VerifyAggregateWithDomain(sig, pubKeys, msgs32, domain8):
    if len(pubKeys) != len(msgs32):
        return False
    if sig.s.infinity or any(p.p.infinity for p in pubKeys):
        return False
    if all_in_memory([p.p for p in pubKeys] + [sig.s]):
        return engine.g1pubs_verify_aggregate_with_domain_jac(msgs32, domain8, b"".join(p.p.jac for p in pubKeys), sig.s.jac)
    return engine.g1pubs_verify_aggregate_with_domain(msgs32, domain8, b"".join(p.p.raw for p in pubKeys), sig.s.raw)


def SignWithDomainBatch(msgs32, secret_scalars, domain8):
    n = len(msgs32)
    out, inf = engine.g1pubs_sign_with_domain_batch(msgs32, domain8, b"".join(secret_scalars))
    return [Signature(Point(None if inf[i] else out[i].tobytes(), SIG_GROUP)) for i in range(n)]
