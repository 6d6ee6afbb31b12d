"""bls_amd -- MI355X-native batch BLS12-381 engine behind the g1pubs / g2pubs verify surface of
phoreproject/bls.  `bls_amd.g2pubs` / `bls_amd.g1pubs` mirror the reference packages; `bls_amd.engine`
is the raw batch interface over the C ABI (include/blsmi.h)."""
from . import _native  # noqa: F401

__all__ = ["engine", "g1pubs", "g2pubs"]
