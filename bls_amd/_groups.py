"""Shared implementation of the g1pubs / g2pubs host mirrors (see g1pubs.py, g2pubs.py)."""
from . import engine


class DeserializeError(ValueError):
    """Mirrors the (nil, error) returns of DeserializePublicKey / DeserializeSignature."""


_DECODE_ERRORS = {1: "unexpected compression mode", 2: "unexpected information in compressed infinity",
                  3: "point not on curve", 4: "not in correct subgroup"}


_Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


def _jac_is_infinity(jac, group):
    """z.IsZero() of an in-memory record (g1.go:287-289, g2.go:325-327); a coordinate not below q reads as 0, as in the library"""
    nc = 1 if group == 1 else 2
    z = jac[48 * 2 * nc:]
    vals = [int.from_bytes(z[48 * e:48 * e + 48], "little") for e in range(nc)]
    return not any(v for v in vals if v < _Q)


class Point:
    """A group element as the library takes it: the affine wire form (96 bytes for G1, 192 for G2; None = infinity) or -- what the
    reference's own types hold, g2pubs/bls.go:13-15, 53-55 -- the in-memory Jacobian record `jac` (bls.G?Projective: 144 / 288 bytes of
    little-endian Montgomery limbs), converted to the wire form by the library only when somebody asks for .raw."""
    __slots__ = ("_raw", "group", "jac", "_inf")

    def __init__(self, raw, group, jac=None):
        self.group = group
        self.jac = None if jac is None else bytes(jac)
        if self.jac is not None:
            if len(self.jac) != (144 if group == 1 else 288):
                raise ValueError("in-memory G%d point: %d bytes" % (group, len(self.jac)))
            self._inf = _jac_is_infinity(self.jac, group)
            self._raw = None
        else:
            self._raw = None if raw is None else bytes(raw)
            self._inf = raw is None

    @property
    def raw(self):
        """ToAffine().SerializeBytes() (None for infinity); for a point held in memory form: one call into the library, cached"""
        if self._raw is None and self.jac is not None and not self._inf:
            fn = engine.g1_jac_to_affine_batch if self.group == 1 else engine.g2_jac_to_affine_batch
            self._raw = fn(self.jac, 1)[0]
        return self._raw

    @property
    def infinity(self):
        return self._inf

    def copy(self):
        return Point(self._raw, self.group, jac=self.jac)

    def bytes_or_zero(self):
        return self.raw if self.raw is not None else bytes(96 if self.group == 1 else 192)

    def __eq__(self, other):
        return isinstance(other, Point) and self.group == other.group and self.raw == other.raw

    def serialize(self):
        """CompressG1 / CompressG2 (g1.go:230-249, g2.go:269-289)."""
        fn = engine.g1_compress_batch if self.group == 1 else engine.g2_compress_batch
        return fn(self.bytes_or_zero(), 1, [1 if self.infinity else 0])[0].tobytes()

    @staticmethod
    def deserialize(data, group):
        """DecompressG1 / DecompressG2 with the subgroup check (g1.go:185-198, g2.go:219-230)."""
        fn = engine.g1_decompress_batch if group == 1 else engine.g2_decompress_batch
        out, inf, err = fn(bytes(data), 1, True)
        if err[0]:
            raise DeserializeError(_DECODE_ERRORS.get(int(err[0]), "decode error"))
        return Point(None if inf[0] else out[0].tobytes(), group)


def point_sum(points, group):
    """Sequential Add from the zero point in the reference (g2pubs/bls.go:165-192); a tree on the device.  Points that are all held in
    memory form are added as the Jacobian points they are and the sum comes back the same way (blsmi_g?_sum_jac)."""
    n = len(points)
    if n == 0:
        return Point(None, group)
    if all(p.jac is not None for p in points):
        out, inf = (engine.g1_sum_jac if group == 1 else engine.g2_sum_jac)(b"".join(p.jac for p in points), n)
        return Point(None, group, jac=out)
    buf = b"".join(p.bytes_or_zero() for p in points)
    inf = [1 if p.infinity else 0 for p in points]
    fn = engine.g1_sum if group == 1 else engine.g2_sum
    return Point(fn(buf, n, inf), group)


def all_in_memory(points):
    """True when every point of a call is held as an in-memory Jacobian record: the call takes the *_jac entry point"""
    return len(points) > 0 and all(p.jac is not None for p in points)
