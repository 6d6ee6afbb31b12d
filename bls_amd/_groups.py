"""Shared implementation of the g1pubs / g2pubs host mirrors (see g1pubs.py, g2pubs.py)."""
from . import engine


class DeserializeError(ValueError):
    """Mirrors the (nil, error) returns of DeserializePublicKey / DeserializeSignature."""


_DECODE_ERRORS = {1: "unexpected compression mode", 2: "unexpected information in compressed infinity",
                  3: "point not on curve", 4: "not in correct subgroup"}


class Point:
    """An affine group element as the library's wire form: bytes (96 for G1, 192 for G2) or infinity."""
    __slots__ = ("raw", "group")

    def __init__(self, raw, group):
        self.raw = None if raw is None else bytes(raw)
        self.group = group

    @property
    def infinity(self):
        return self.raw is None

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
    """Sequential Add from the zero point in the reference (g2pubs/bls.go:165-192); a tree on the device."""
    n = len(points)
    if n == 0:
        return Point(None, group)
    buf = b"".join(p.bytes_or_zero() for p in points)
    inf = [1 if p.infinity else 0 for p in points]
    fn = engine.g1_sum if group == 1 else engine.g2_sum
    return Point(fn(buf, n, inf), group)
