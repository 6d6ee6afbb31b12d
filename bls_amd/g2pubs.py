"""Host-side mirror of the reference package g2pubs (g2pubs/bls.go): PublicKey in G2, Signature in
G1, messages hashed to G1.  Same names, argument meaning and results as the Go API; every group,
pairing and hash operation runs in the HIP kernels of libblsmi.so (no CPU fallback).

    Verify(m, pub, sig)                         g2pubs/bls.go:159-162
    sig.VerifyAggregate(pubKeys, msgs)          g2pubs/bls.go:240-270
    sig.VerifyAggregateCommon(pubKeys, msg)     g2pubs/bls.go:275-278
    AggregateSignatures / AggregatePublicKeys   g2pubs/bls.go:165-192
    DeserializeSignature / DeserializePublicKey g2pubs/bls.go:33-40, 89-96
plus VerifyBatch, the batch form the one-tuple-per-call Go API lacks, and PrepareKeys / VerifyBatchPrepared: public keys run
through G2AffineToPrepared (g2.go:639-801) ONCE into tables resident on the device instead of on every Verify (pairing.go:140-147).
"""
from . import engine
from ._groups import DeserializeError, Point, all_in_memory, point_sum  # noqa: F401

SIG_GROUP, PK_GROUP = 1, 2


class Signature:
    def __init__(self, point):
        self.s = point

    def Serialize(self):                      # g2pubs/bls.go:18-20
        return self.s.serialize()

    def Copy(self):
        return Signature(self.s.copy())

    def Aggregate(self, other):               # g2pubs/bls.go:174-177
        self.s = point_sum([self.s, other.s], SIG_GROUP)

    def VerifyAggregate(self, pubKeys, msgs):
        if len(pubKeys) != len(msgs):          # g2pubs/bls.go:241-243
            return False
        if self.s.infinity or any(p.p.infinity for p in pubKeys):
            return False                       # the reference panics in MillerLoop on infinity; defined as false here
        if all_in_memory([p.p for p in pubKeys] + [self.s]):   # the points as the Go values hold them: ToAffine on the device
            return engine.g2pubs_verify_aggregate_jac(msgs, b"".join(p.p.jac for p in pubKeys), self.s.jac)
        return engine.g2pubs_verify_aggregate(msgs, b"".join(p.p.raw for p in pubKeys), self.s.raw)

    def VerifyAggregateCommon(self, pubKeys, msg):
        if all_in_memory([p.p for p in pubKeys] + [self.s]):   # key sum and Verify in one library call, Jacobian points summed as they are
            return engine.g2pubs_verify_aggregate_common_jac(msg, b"".join(p.p.jac for p in pubKeys), self.s.jac, len(pubKeys))
        return Verify(msg, AggregatePublicKeys(pubKeys), self)


class PublicKey:
    def __init__(self, point):
        self.p = point

    def Serialize(self):                      # g2pubs/bls.go:67-69
        return self.p.serialize()

    def Copy(self):
        return PublicKey(self.p.copy())

    def Equals(self, other):
        return self.p == other.p

    def Aggregate(self, other):               # g2pubs/bls.go:189-192
        self.p = point_sum([self.p, other.p], PK_GROUP)


def NewSignatureFromG1(raw96):
    return Signature(Point(raw96, SIG_GROUP))


def NewPublicKeyFromG2(raw192):
    return PublicKey(Point(raw192, PK_GROUP))


def NewSignatureFromG1Projective(jac144):
    """a Signature holding its point the way the reference's does (g2pubs/bls.go:13-15): the 144 bytes of a *bls.G1Projective"""
    return Signature(Point(None, SIG_GROUP, jac=jac144))


def NewPublicKeyFromG2Projective(jac288):
    """g2pubs/bls.go:53-55: the 288 bytes of a *bls.G2Projective"""
    return PublicKey(Point(None, PK_GROUP, jac=jac288))


def DeserializeSignature(b48):
    return Signature(Point.deserialize(b48, SIG_GROUP))


def DeserializePublicKey(b96):
    return PublicKey(Point.deserialize(b96, PK_GROUP))


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
        ok, _ = engine.g2pubs_verify_batch_jac(msgs, b"".join(p.p.jac for p in pubs), b"".join(s.s.jac for s in sigs))
        return [bool(x) for x in ok]
    flags = [(1 if p.p.infinity else 0) | (2 if s.s.infinity else 0) for p, s in zip(pubs, sigs)]
    ok, _ = engine.g2pubs_verify_batch(msgs, b"".join(p.p.bytes_or_zero() for p in pubs), b"".join(s.s.bytes_or_zero() for s in sigs), flags)
    return [bool(x) for x in ok]


class PreparedKeys:
    """The reference's G2Prepared of n public keys, kept in device memory the library owns."""

    def __init__(self, pubs):
        self.n = len(pubs)
        if all_in_memory([p.p for p in pubs]):
            self._h = engine.PreparedKeysJac(b"".join(p.p.jac for p in pubs), self.n)         # z == 0 = infinity: verdict False
        else:
            self._h = engine.PreparedKeys(b"".join(p.p.bytes_or_zero() for p in pubs), self.n)   # all-zero record = infinity: verdict False

    def Close(self):
        self._h.close()


def PrepareKeys(pubs):
    return PreparedKeys(pubs)


def VerifyBatchPrepared(msgs, keys, key_idx, sigs):
    """[Verify(msgs[i], pubs[key_idx[i]], sigs[i]) for i] with pubs = the keys given to PrepareKeys; same verdicts as VerifyBatch."""
    n = len(msgs)
    if not (len(key_idx) == len(sigs) == n):
        raise ValueError("length mismatch")
    if n == 0:
        return []
    if any(not 0 <= int(k) < keys.n for k in key_idx):
        raise IndexError("key index out of range")
    if all_in_memory([s.s for s in sigs]):
        return [bool(x) for x in engine.g2pubs_verify_batch_prepared_jac(msgs, keys._h, key_idx, b"".join(s.s.jac for s in sigs))]
    flags = [2 if s.s.infinity else 0 for s in sigs]
    ok, _ = engine.g2pubs_verify_batch_prepared(msgs, keys._h, key_idx, b"".join(s.s.bytes_or_zero() for s in sigs), flags)
    return [bool(x) for x in ok]


def VerifySerializedBatch(msgs, pub_bytes, sig_bytes):
    """DeserializePublicKey + DeserializeSignature + Verify per tuple, in one device pass over the 96-byte keys and
    48-byte signatures of the wire format.  A tuple whose key or signature does not deserialise (the reference returns
    an error there and Verify is never reached) or is the point at infinity yields False."""
    ok, _, _ = engine.verify_serialized_batch(__name__.rsplit(".", 1)[-1], msgs, b"".join(pub_bytes), b"".join(sig_bytes), True)
    return [bool(x) for x in ok]


def Verify(m, pub, sig):
    return VerifyBatch([m], [pub], [sig])[0]


def Sign(message, key):
    """Sign(message, key) (g2pubs/bls.go:132-135): key = the secret scalar as 32 big-endian bytes (SecretKey.Serialize()).  One call is
    one hash-to-curve plus one windowed multiplication on the latency path; not side-channel hardened (include/blsmi.h)."""
    return SignBatch([message], [key])[0]


def PrivToPub(k):
    """PrivToPub(k) (g2pubs/bls.go:138-140): k = the secret scalar as 32 big-endian bytes."""
    return PrivToPubBatch([k])[0]


def PrivToPubBatch(secret_scalars):
    """pk_i = sk_i * generator (PrivToPub, g2pubs/bls.go:138-140); scalars are 32-byte big-endian."""
    n = len(secret_scalars)
    out, inf = engine.g2_mul_generator_batch(b"".join(secret_scalars), n)
    return [PublicKey(Point(None if inf[i] else out[i].tobytes(), PK_GROUP)) for i in range(n)]


def SignBatch(msgs, secret_scalars):
    """sigma_i = sk_i * HashG1(m_i) (Sign, g2pubs/bls.go:132-135); scalars are 32-byte big-endian."""
    n = len(msgs)
    out, inf = engine.g2pubs_sign_batch(msgs, b"".join(secret_scalars))    # one call: hash, then multiply, on the device
    return [Signature(Point(None if inf[i] else out[i].tobytes(), SIG_GROUP)) for i in range(n)]
