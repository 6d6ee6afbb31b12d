"""Thin ctypes layer over the C ABI of libblsmi.so (include/blsmi.h).  Host code only: every
computation happens in the HIP kernels; a missing library or device raises, nothing falls back."""
import ctypes as C

import numpy as np

from . import _native

_u8p = C.POINTER(C.c_uint8)
_u64p = C.POINTER(C.c_uint64)


class BlsmiError(RuntimeError):
    pass


_ERR = {-1: "no usable HIP device", -2: "HIP runtime call failed", -3: "bad argument", -4: "out of memory", -5: "RCCL unavailable or collective failed"}


def _check(rc, what):
    if rc != 0:
        raise BlsmiError("%s failed: %s (%d)" % (what, _ERR.get(rc, "unknown"), rc))


def _lib():
    return _native.load()


def _u8(x, nbytes=None):
    a = np.frombuffer(x, dtype=np.uint8) if isinstance(x, (bytes, bytearray, memoryview)) else np.ascontiguousarray(x, dtype=np.uint8).reshape(-1)
    if nbytes is not None and a.size != nbytes:
        raise ValueError("expected %d bytes, got %d" % (nbytes, a.size))
    return a


def _p8(a):
    return a.ctypes.data_as(_u8p) if a is not None and a.size else None


def init(device=0):
    """Bind this process to one device (one rank per GPU under torchrun)."""
    _check(_lib().blsmi_init(int(device)), "blsmi_init")


def init_devices(ndev=0):
    """Drive the first ndev devices (0 = all visible) from this one process: large verify / pairing batches are split
    over them inside the library (RCCL bitmap all-reduce / partial-product all-gather), see include/blsmi.h."""
    _check(_lib().blsmi_init_devices(int(ndev)), "blsmi_init_devices")


def debug_alias_own(ptr, nbytes, device_index):
    """TEST HOOK (BLSMI_DEVICE_ALIAS): the device-pointer range belongs to logical device `device_index` (include/blsmi.h)."""
    _check(_lib().blsmi_debug_alias_own(C.c_void_p(int(ptr)), C.c_size_t(int(nbytes)), int(device_index)), "blsmi_debug_alias_own")


def trim(keep_bytes_per_context=0):
    """Give the temporaries idle call contexts hold beyond keep_bytes_per_context (and the pool cache) back to the driver; bytes freed."""
    freed = C.c_size_t(0)
    _check(_lib().blsmi_trim(C.c_size_t(int(keep_bytes_per_context)), C.byref(freed)), "blsmi_trim")
    return int(freed.value)


def held_bytes():
    """device memory the call contexts currently hold for temporaries (all devices)"""
    _lib().blsmi_held_bytes.restype = C.c_size_t
    return int(_lib().blsmi_held_bytes())


SHAPE_PAIRING, SHAPE_MILLER_LOOP, SHAPE_FINAL_EXP, SHAPE_G2_PREPARE, SHAPE_VERIFY, SHAPE_SIGN, SHAPE_VERIFY_DOMAIN, SHAPE_POINT_ADD = range(8)


def prefer_cpu(shape, n):
    """True when n operations of `shape` in one call finish sooner on one core of the upstream CPU path (include/blsmi.h)"""
    return bool(_lib().blsmi_prefer_cpu(int(shape), C.c_size_t(int(n))))


def device_leases(device_index):
    """context leases device `device_index` has served since initialisation (-1: no such device)"""
    _lib().blsmi_debug_device_leases.restype = C.c_longlong
    return int(_lib().blsmi_debug_device_leases(int(device_index)))


def device_count():
    return int(_lib().blsmi_device_count())


def shard_count():
    return int(_lib().blsmi_shard_count())


def set_latency_threshold(max_tuples):
    """Batches of at most max_tuples tuples take the latency path (one tuple per wave); 0 = always the lane-pair kernels."""
    _check(_lib().blsmi_set_latency_threshold(C.c_size_t(int(max_tuples))), "blsmi_set_latency_threshold")


def set_quad_threshold(max_tuples):
    """Batches above the latency threshold and of at most max_tuples tuples take the lane-quad kernels (four lanes per tuple); 0 = off."""
    _check(_lib().blsmi_set_quad_threshold(C.c_size_t(int(max_tuples))), "blsmi_set_quad_threshold")


def set_row_threshold(min_tuples, max_tuples):
    """A lone pairing / verify call of min_tuples .. max_tuples tuples takes the lane-row kernels (sixteen lanes per tuple); max 0 = off."""
    _check(_lib().blsmi_set_row_threshold(C.c_size_t(int(min_tuples)), C.c_size_t(int(max_tuples))), "blsmi_set_row_threshold")


ROW_DEFAULT = (2048, 8192)


def set_option(name, value):
    """run-time switch between code paths with identical results: "agg_cofactor_pow", "msm_sort", "dup_force_sort" (include/blsmi.h)"""
    _check(_lib().blsmi_set_option(name.encode(), C.c_longlong(int(value))), "blsmi_set_option(%s)" % name)


def set_mul_assume_subgroup(on=True):
    """scalar multiplications through the curve endomorphisms (multiplicands in the prime-order subgroup; the default) or,
    with on=False, the plain windowed ladder that serves every curve point"""
    _check(_lib().blsmi_set_mul_assume_subgroup(C.c_int(1 if on else 0)), "blsmi_set_mul_assume_subgroup")


def shutdown():
    _lib().blsmi_shutdown.restype = None
    _lib().blsmi_shutdown()


def version():
    return _lib().blsmi_version().decode()


class PackedMsgs:
    """n messages already laid out as the C ABI wants them (concatenated bytes + n+1 offsets): lets a caller that issues
    the same batch repeatedly -- or holds a million messages -- pay the Python-side packing once."""

    def __init__(self, msgs):
        self.buf, self.off = _msgs(msgs)
        self.n = len(msgs)

    def __len__(self):
        return self.n


def _msgs(msgs):
    if isinstance(msgs, PackedMsgs):
        return msgs.buf, msgs.off
    off = np.zeros(len(msgs) + 1, dtype=np.uint64)
    if len(msgs):
        off[1:] = np.cumsum([len(m) for m in msgs])
    buf = np.frombuffer(b"".join(bytes(m) for m in msgs) or b"\0", dtype=np.uint8)
    return buf, off


# ---- pairing ---------------------------------------------------------------------------------------
class HostBuffer:
    """Page-locked host memory from the library (blsmi_host_alloc) as a numpy uint8 array: `.a`.  Arrays handed to the host entry
    points from such memory are copied by DMA instead of being staged.  The block is returned to the runtime when the LAST numpy view of
    it is gone: `.a` and every slice taken from it keep the allocation alive (they hold the ctypes block whose finaliser frees it), so
    .free() -- or dropping the HostBuffer -- can never pull memory from under a view that is still in use."""

    def __init__(self, nbytes):
        import weakref
        p = C.c_void_p()
        lib = _lib()
        lib.blsmi_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
        lib.blsmi_host_free.argtypes = [C.c_void_p]
        _check(lib.blsmi_host_alloc(C.c_size_t(int(nbytes)), C.byref(p)), "blsmi_host_alloc")
        self.nbytes = int(nbytes)
        if nbytes:
            block = (C.c_uint8 * self.nbytes).from_address(p.value)         # every numpy view below references this object
            self._finalizer = weakref.finalize(block, lib.blsmi_host_free, C.c_void_p(p.value))
            self.a = np.frombuffer(block, dtype=np.uint8)
        else:
            self._finalizer = None
            self.a = np.zeros(0, np.uint8)

    @property
    def alive(self):
        """False once the block has gone back to the runtime (no view of it is left)"""
        return self._finalizer is not None and self._finalizer.alive

    def free(self):
        """drop this object's reference; the memory is released as soon as no view of `.a` remains (at once if there is none)"""
        self.a = None


def pairing_batch(g1_aff, g2_aff, n, out=None):
    """n x 96 B G1 affine, n x 192 B G2 affine -> (n, 72) uint64: the reference's in-memory FQ12."""
    a, b = _u8(g1_aff, 96 * n), _u8(g2_aff, 192 * n)
    if out is None:
        out = np.zeros((n, 72), dtype=np.uint64)
    _check(_lib().blsmi_pairing_batch(_p8(a), _p8(b), out.ctypes.data_as(_u64p), C.c_size_t(n)), "blsmi_pairing_batch")
    return out


def miller_loop_batch(g1_aff, g2_aff, n):
    a, b = _u8(g1_aff, 96 * n), _u8(g2_aff, 192 * n)
    out = np.zeros((n, 72), dtype=np.uint64)
    _check(_lib().blsmi_miller_loop_batch(_p8(a), _p8(b), out.ctypes.data_as(_u64p), C.c_size_t(n)), "blsmi_miller_loop_batch")
    return out


def final_exponentiation_batch(fq12):
    x = np.ascontiguousarray(fq12, dtype=np.uint64).reshape(-1, 72)
    out = np.zeros_like(x)
    _check(_lib().blsmi_final_exponentiation_batch(x.ctypes.data_as(_u64p), out.ctypes.data_as(_u64p), C.c_size_t(x.shape[0])), "blsmi_final_exponentiation_batch")
    return out


def pairing_batch_dev(d_g1, d_g2, d_out, n, stream=0):
    """Device-pointer form (ints, e.g. torch.Tensor.data_ptr()); enqueues and synchronises `stream`."""
    _check(_lib().blsmi_pairing_batch_dev(C.c_void_p(d_g1), C.c_void_p(d_g2), C.c_void_p(d_out), C.c_size_t(n), C.c_void_p(stream)), "blsmi_pairing_batch_dev")


def verify_batch_dev(group, d_msgs, d_off, d_pks, d_sigs, d_inf, d_ok, n, stream=0):
    fn = _lib().blsmi_g2pubs_verify_batch_dev if group == "g2pubs" else _lib().blsmi_g1pubs_verify_batch_dev
    _check(fn(C.c_void_p(d_msgs), C.c_void_p(d_off), C.c_void_p(d_pks), C.c_void_p(d_sigs), C.c_void_p(d_inf or 0), C.c_void_p(d_ok), C.c_size_t(n), C.c_void_p(stream)), "verify_batch_dev")


def g1pubs_verify_with_domain_batch_dev(d_msgs32, d_domain, d_pks, d_sigs, d_inf, d_ok, n, stream=0):
    """VerifyWithDomain (g1pubs/bls.go:171-174) with everything resident on the device (ints = device pointers)"""
    _check(_lib().blsmi_g1pubs_verify_with_domain_batch_dev(C.c_void_p(d_msgs32), C.c_void_p(d_domain), C.c_void_p(d_pks), C.c_void_p(d_sigs), C.c_void_p(d_inf or 0), C.c_void_p(d_ok), C.c_size_t(n), C.c_void_p(stream)), "verify_with_domain_batch_dev")


def mul_batch_dev(group, d_pts, d_scalars, d_out, d_out_inf, n, stream=0, any_point=False):
    """k_i * P_i with everything resident on the device (ints = device pointers; d_pts = 0: the group generator).
    any_point: multiplicands outside the prime-order subgroup are allowed for this call (see g1_mul_batch)."""
    if any_point:
        fn = _lib().blsmi_g1_mul_batch_dev_ex if group == "g1" else _lib().blsmi_g2_mul_batch_dev_ex
        _check(fn(C.c_void_p(d_pts or 0), C.c_void_p(d_scalars), C.c_void_p(d_out), C.c_void_p(d_out_inf), C.c_size_t(n), C.c_void_p(stream), C.c_uint(MUL_ANY_POINT)), "mul_batch_dev_ex")
        return
    fn = _lib().blsmi_g1_mul_batch_dev if group == "g1" else _lib().blsmi_g2_mul_batch_dev
    _check(fn(C.c_void_p(d_pts or 0), C.c_void_p(d_scalars), C.c_void_p(d_out), C.c_void_p(d_out_inf), C.c_size_t(n), C.c_void_p(stream)), "mul_batch_dev")


def sum_dev(group, d_pts, d_in_inf, n, d_out, stream=0):
    """sum of n resident points -> affine bytes at d_out (device); returns True when the sum is the point at infinity."""
    fn = _lib().blsmi_g1_sum_dev if group == "g1" else _lib().blsmi_g2_sum_dev
    oinf = C.c_int(0)
    _check(fn(C.c_void_p(d_pts), C.c_void_p(d_in_inf or 0), C.c_size_t(n), C.c_void_p(d_out), C.byref(oinf), C.c_void_p(stream)), "sum_dev")
    return bool(oinf.value)


def msm_dev(group, d_pts, d_scalars, n, d_out, stream=0, any_point=False):
    """sum_i k_i P_i over resident points and scalars -> affine bytes at d_out (device); True when it is the point at infinity."""
    oinf = C.c_int(0)
    if any_point:
        fn = _lib().blsmi_g1_msm_dev_ex if group == "g1" else _lib().blsmi_g2_msm_dev_ex
        _check(fn(C.c_void_p(d_pts), C.c_void_p(d_scalars), C.c_size_t(n), C.c_void_p(d_out), C.byref(oinf), C.c_void_p(stream), C.c_uint(MUL_ANY_POINT)), "msm_dev_ex")
        return bool(oinf.value)
    fn = _lib().blsmi_g1_msm_dev if group == "g1" else _lib().blsmi_g2_msm_dev
    _check(fn(C.c_void_p(d_pts), C.c_void_p(d_scalars), C.c_size_t(n), C.c_void_p(d_out), C.byref(oinf), C.c_void_p(stream)), "msm_dev")
    return bool(oinf.value)


def verify_aggregate_common_dev(group, d_pks, n, msg, sig, domain=None, stream=0):
    """VerifyAggregateCommon (g2pubs/bls.go:275-278, g1pubs/bls.go:287-297) with the n public keys resident on the device (int =
    device pointer); msg / sig = host bytes; domain (8 bytes, g1pubs only): the *WithDomain form over a 32-byte message."""
    ok = C.c_int(0)
    m = _u8(msg) if len(msg) else None
    if domain is not None:
        assert group == "g1pubs"
        _check(_lib().blsmi_g1pubs_verify_aggregate_common_with_domain_dev(C.c_void_p(d_pks or 0), C.c_size_t(n), _p8(_u8(msg, 32)), _p8(_u8(domain, 8)), _p8(_u8(sig, 192)), C.byref(ok), C.c_void_p(stream)), "verify_aggregate_common_with_domain_dev")
        return bool(ok.value)
    fn, sgb = (_lib().blsmi_g2pubs_verify_aggregate_common_dev, 96) if group == "g2pubs" else (_lib().blsmi_g1pubs_verify_aggregate_common_dev, 192)
    _check(fn(C.c_void_p(d_pks or 0), C.c_size_t(n), _p8(m) if m is not None else None, C.c_size_t(len(msg)), _p8(_u8(sig, sgb)), C.byref(ok), C.c_void_p(stream)), "verify_aggregate_common_dev")
    return bool(ok.value)


def verify_aggregate_dev(group, d_msgs, d_off, d_pks, sig, n, stream=0):
    """VerifyAggregate with messages / offsets / keys resident on the device; sig = host bytes of the aggregate signature."""
    fn, sgb = (_lib().blsmi_g2pubs_verify_aggregate_dev, 96) if group == "g2pubs" else (_lib().blsmi_g1pubs_verify_aggregate_dev, 192)
    s = _u8(sig, sgb)
    ok = C.c_int(0)
    _check(fn(C.c_void_p(d_msgs), C.c_void_p(d_off), C.c_void_p(d_pks), _p8(s), C.c_size_t(n), C.byref(ok), C.c_void_p(stream)), "verify_aggregate_dev")
    return bool(ok.value)


def verify_aggregate_with_domain_dev(d_msgs32, d_domain, d_pks, sig, n, stream=0):
    s = _u8(sig, 192)
    ok = C.c_int(0)
    _check(_lib().blsmi_g1pubs_verify_aggregate_with_domain_dev(C.c_void_p(d_msgs32), C.c_void_p(d_domain), C.c_void_p(d_pks), _p8(s), C.c_size_t(n), C.byref(ok), C.c_void_p(stream)), "verify_aggregate_with_domain_dev")
    return bool(ok.value)


# ---- prepared public keys (g2pubs): G2AffineToPrepared once, Miller loops that read the lines (blsmi 0.4) ----
G2_PREPARED_BYTES = 24704


def g2_prepare_batch_dev(d_g2_aff, n, d_prepared, stream=0):
    """n resident G2 points -> n tables of G2_PREPARED_BYTES at d_prepared (device)."""
    _check(_lib().blsmi_g2_prepare_batch_dev(C.c_void_p(d_g2_aff), C.c_size_t(n), C.c_void_p(d_prepared), C.c_void_p(stream)), "blsmi_g2_prepare_batch_dev")


def g2_prepared_export_dev(d_prepared, n, d_out, stream=0):
    """the reference's G2Prepared.coeffs of n tables: n x 68 x 3 x 12 uint64 at d_out (device)."""
    _check(_lib().blsmi_g2_prepared_export_dev(C.c_void_p(d_prepared), C.c_size_t(n), C.c_void_p(d_out), C.c_void_p(stream)), "blsmi_g2_prepared_export_dev")


def g2_prepare_batch(g2_aff, n):
    """G2AffineToPrepared (g2.go:639-801) of n host points: (n, 68, 3, 12) uint64 -- coefficient triples of FQ2 as c0 | c1 limbs."""
    q = _u8(g2_aff, 192 * n)
    out = np.empty((n, 68, 3, 12), dtype=np.uint64)
    _check(_lib().blsmi_g2_prepare_batch(_p8(q), C.c_size_t(n), out.ctypes.data_as(_u64p)), "blsmi_g2_prepare_batch")
    return out


class PreparedKeys:
    """n g2pubs public keys prepared into tables the library owns (blsmi_g2_prepared_create); .ptr is the device pointer."""

    def __init__(self, g2_aff, n):
        q = _u8(g2_aff, 192 * n)
        h = C.c_void_p(0)
        _check(_lib().blsmi_g2_prepared_create(_p8(q), C.c_size_t(n), C.byref(h)), "blsmi_g2_prepared_create")
        self.ptr, self.n = h.value, n

    def close(self):
        if self.ptr:
            _check(_lib().blsmi_g2_prepared_destroy(C.c_void_p(self.ptr)), "blsmi_g2_prepared_destroy")
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def g2pubs_verify_batch_prepared(msgs, prepared, key_idx, sigs, inf_flags=None):
    """g2pubs.Verify x n, host buffers, keys prepared (a PreparedKeys or a device pointer): (ok bool array, packed bitmap)."""
    n = len(msgs)
    buf, off = _msgs(msgs)
    ptr = prepared.ptr if isinstance(prepared, PreparedKeys) else prepared
    idx = None if key_idx is None else np.ascontiguousarray(key_idx, dtype=np.uint32)
    s = _u8(sigs, 96 * n)
    fl = None if inf_flags is None else _u8(inf_flags, n)
    ok = np.empty(n, dtype=np.uint8)
    bm = np.empty((n + 7) // 8, dtype=np.uint8)
    _check(_lib().blsmi_g2pubs_verify_batch_prepared(_p8(buf), off.ctypes.data_as(_u64p), C.c_void_p(ptr), None if idx is None else idx.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                    _p8(s), _p8(fl), _p8(ok), _p8(bm), C.c_size_t(n)), "blsmi_g2pubs_verify_batch_prepared")
    return ok.astype(bool), bm


def pairing_batch_prepared_dev(d_g1, d_prepared, d_key_idx, d_out, n, stream=0):
    _check(_lib().blsmi_pairing_batch_prepared_dev(C.c_void_p(d_g1), C.c_void_p(d_prepared), C.c_void_p(d_key_idx or 0), C.c_void_p(d_out), C.c_size_t(n), C.c_void_p(stream)), "blsmi_pairing_batch_prepared_dev")


def g2pubs_verify_batch_prepared_dev(d_msgs, d_off, d_prepared, d_key_idx, d_sigs, d_inf, d_ok, n, stream=0):
    _check(_lib().blsmi_g2pubs_verify_batch_prepared_dev(C.c_void_p(d_msgs), C.c_void_p(d_off), C.c_void_p(d_prepared), C.c_void_p(d_key_idx or 0), C.c_void_p(d_sigs),
                                                        C.c_void_p(d_inf or 0), C.c_void_p(d_ok), C.c_size_t(n), C.c_void_p(stream)), "blsmi_g2pubs_verify_batch_prepared_dev")


def g2pubs_verify_aggregate_prepared(msgs, prepared, key_idx, sig):
    """Signature.VerifyAggregate over prepared keys, host buffers (a PreparedKeys or a device pointer)."""
    n = len(msgs)
    buf, off = _msgs(msgs)
    ptr = prepared.ptr if isinstance(prepared, PreparedKeys) else prepared
    idx = None if key_idx is None else np.ascontiguousarray(key_idx, dtype=np.uint32)
    s = _u8(sig, 96)
    ok = C.c_int(0)
    _check(_lib().blsmi_g2pubs_verify_aggregate_prepared(_p8(buf), off.ctypes.data_as(_u64p), C.c_void_p(ptr), None if idx is None else idx.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                        _p8(s), C.c_size_t(n), C.byref(ok)), "blsmi_g2pubs_verify_aggregate_prepared")
    return bool(ok.value)


def g2pubs_verify_aggregate_prepared_dev(d_msgs, d_off, d_prepared, d_key_idx, sig, n, stream=0):
    s = _u8(sig, 96)
    ok = C.c_int(0)
    _check(_lib().blsmi_g2pubs_verify_aggregate_prepared_dev(C.c_void_p(d_msgs), C.c_void_p(d_off), C.c_void_p(d_prepared), C.c_void_p(d_key_idx or 0), _p8(s), C.c_size_t(n),
                                                            C.byref(ok), C.c_void_p(stream)), "blsmi_g2pubs_verify_aggregate_prepared_dev")
    return bool(ok.value)


# ---- groups ----------------------------------------------------------------------------------------
MUL_ANY_POINT = 1          # BLSMI_MUL_ANY_POINT (include/blsmi.h)


def _mul(fn, pb, pts, scalars, n, flags=None):
    p, s = _u8(pts, pb * n), _u8(scalars, 32 * n)
    out = np.empty(pb * n, dtype=np.uint8)              # every byte is written by the call
    inf = np.empty(n, dtype=np.uint8)
    extra = () if flags is None else (C.c_uint(flags),)
    _check(fn(_p8(p), _p8(s), _p8(out), _p8(inf), C.c_size_t(n), *extra), "mul_batch")
    return out.reshape(n, pb), inf.astype(bool)


def g1_mul_batch(pts, scalars, n, any_point=False):
    """k_i * P_i (G1Affine.MulFR, g1.go:80-90).  The default ladder runs through the curve endomorphism and REQUIRES multiplicands
    in the prime-order subgroup (hash points, generators, keys / signatures that passed Deserialize's subgroup check): an on-curve
    point outside it gives a different point, silently.  any_point=True selects, for this call only, the plain windowed ladder
    that serves every curve point like the reference does (e.g. after g1_decompress_batch(check_subgroup=False))."""
    if any_point:
        return _mul(_lib().blsmi_g1_mul_batch_ex, 96, pts, scalars, n, MUL_ANY_POINT)
    return _mul(_lib().blsmi_g1_mul_batch, 96, pts, scalars, n)


def g2_mul_batch(pts, scalars, n, any_point=False):
    """k_i * P_i (G2Affine.MulFR, g2.go:92-102); subgroup points unless any_point=True -- see g1_mul_batch."""
    if any_point:
        return _mul(_lib().blsmi_g2_mul_batch_ex, 192, pts, scalars, n, MUL_ANY_POINT)
    return _mul(_lib().blsmi_g2_mul_batch, 192, pts, scalars, n)


def _mul_gen(fn, pb, scalars, n):
    s = _u8(scalars, 32 * n)
    out = np.zeros(pb * n, dtype=np.uint8)
    inf = np.zeros(n, dtype=np.uint8)
    _check(fn(_p8(s), _p8(out), _p8(inf), C.c_size_t(n)), "mul_generator_batch")
    return out.reshape(n, pb), inf.astype(bool)


def g1_mul_generator_batch(scalars, n):
    """k_i * G1 generator (PrivToPub of g1pubs)."""
    return _mul_gen(_lib().blsmi_g1_mul_generator_batch, 96, scalars, n)


def g2_mul_generator_batch(scalars, n):
    """k_i * G2 generator (PrivToPub of g2pubs)."""
    return _mul_gen(_lib().blsmi_g2_mul_generator_batch, 192, scalars, n)


def _msm(fn, pb, pts, scalars, n, flags=None):
    p = _u8(pts, pb * n) if n else np.zeros(1, np.uint8)
    s = _u8(scalars, 32 * n) if n else np.zeros(1, np.uint8)
    out = np.zeros(pb, dtype=np.uint8)
    oinf = C.c_int(0)
    extra = () if flags is None else (C.c_uint(flags),)
    _check(fn(_p8(p), _p8(s), C.c_size_t(n), _p8(out), C.byref(oinf), *extra), "msm")
    return None if oinf.value else out.tobytes()


def g1_msm(pts, scalars, n, any_point=False):
    """sum_i k_i * P_i -> 96 affine bytes, or None for the point at infinity (subgroup points unless any_point=True, see g1_mul_batch)."""
    if any_point:
        return _msm(_lib().blsmi_g1_msm_ex, 96, pts, scalars, n, MUL_ANY_POINT)
    return _msm(_lib().blsmi_g1_msm, 96, pts, scalars, n)


def g2_msm(pts, scalars, n, any_point=False):
    if any_point:
        return _msm(_lib().blsmi_g2_msm_ex, 192, pts, scalars, n, MUL_ANY_POINT)
    return _msm(_lib().blsmi_g2_msm, 192, pts, scalars, n)


def _sum(fn, pb, pts, n, in_inf):
    p = _u8(pts, pb * n) if n else np.zeros(1, np.uint8)
    f = _u8(in_inf, n) if in_inf is not None else None
    out = np.zeros(pb, dtype=np.uint8)
    oinf = C.c_int(0)
    _check(fn(_p8(p), _p8(f), C.c_size_t(n), _p8(out), C.byref(oinf)), "sum")
    return (None if oinf.value else out.tobytes())


def g1_sum(pts, n, in_inf=None):
    """Sum of n affine points -> 96 bytes, or None for the point at infinity."""
    return _sum(_lib().blsmi_g1_sum, 96, pts, n, in_inf)


def g2_sum(pts, n, in_inf=None):
    return _sum(_lib().blsmi_g2_sum, 192, pts, n, in_inf)


# ---- hash to curve ---------------------------------------------------------------------------------
def hash_g1_batch(msgs):
    buf, off = _msgs(msgs)
    out = np.zeros((len(msgs), 96), dtype=np.uint8)
    _check(_lib().blsmi_hash_g1_batch(_p8(buf), off.ctypes.data_as(_u64p), _p8(out.reshape(-1)), C.c_size_t(len(msgs))), "blsmi_hash_g1_batch")
    return out


def hash_g2_batch(msgs):
    buf, off = _msgs(msgs)
    out = np.zeros((len(msgs), 192), dtype=np.uint8)
    _check(_lib().blsmi_hash_g2_batch(_p8(buf), off.ctypes.data_as(_u64p), _p8(out.reshape(-1)), C.c_size_t(len(msgs))), "blsmi_hash_g2_batch")
    return out


def hash_g2_with_domain_batch(msgs32, domain8):
    n = len(msgs32)
    buf = _u8(b"".join(bytes(m) for m in msgs32), 32 * n)
    d = _u8(domain8, 8)
    out = np.zeros((n, 192), dtype=np.uint8)
    _check(_lib().blsmi_hash_g2_with_domain_batch(_p8(buf), _p8(d), _p8(out.reshape(-1)), C.c_size_t(n)), "blsmi_hash_g2_with_domain_batch")
    return out


# ---- sign: sk_i * H(m_i) in one call (the hash points stay on the device) ---------------------------
def _sign(fn, name, ob, n, head, sks):
    k = _u8(sks, 32 * n)
    out = np.zeros((n, ob), dtype=np.uint8); inf = np.zeros(n, dtype=np.uint8)
    _check(fn(*head, _p8(k), _p8(out.reshape(-1)), _p8(inf), C.c_size_t(n)), name)
    return out, inf


def g2pubs_sign_batch(msgs, sks):
    """blsmi_g2pubs_sign_batch: n x 96-byte signatures sk_i * HashG1(m_i) (g2pubs/bls.go:132-135) and their infinity flags"""
    buf, off = _msgs(msgs)
    return _sign(_lib().blsmi_g2pubs_sign_batch, "blsmi_g2pubs_sign_batch", 96, len(msgs), (_p8(buf), off.ctypes.data_as(_u64p)), sks)


def g1pubs_sign_batch(msgs, sks):
    """blsmi_g1pubs_sign_batch: n x 192-byte signatures sk_i * HashG2(m_i) (g1pubs/bls.go:132-135)"""
    buf, off = _msgs(msgs)
    return _sign(_lib().blsmi_g1pubs_sign_batch, "blsmi_g1pubs_sign_batch", 192, len(msgs), (_p8(buf), off.ctypes.data_as(_u64p)), sks)


def g1pubs_sign_with_domain_batch(msgs32, domain8, sks):
    """blsmi_g1pubs_sign_with_domain_batch: sk_i * HashG2WithDomain(m_i, domain) (g1pubs/bls.go:138-141)"""
    n = len(msgs32)
    buf = _u8(b"".join(bytes(m) for m in msgs32), 32 * n)
    d = _u8(domain8, 8)
    return _sign(_lib().blsmi_g1pubs_sign_with_domain_batch, "blsmi_g1pubs_sign_with_domain_batch", 192, n, (_p8(buf), _p8(d)), sks)


# ---- verify ----------------------------------------------------------------------------------------
def _verify_batch(fn, pkb, sgb, msgs, pks, sigs, inf_flags):
    n = len(msgs)
    buf, off = _msgs(msgs)
    p, s = _u8(pks, pkb * n), _u8(sigs, sgb * n)
    f = _u8(inf_flags, n) if inf_flags is not None else None
    ok = np.zeros(n, dtype=np.uint8)
    bitmap = np.zeros((n + 7) // 8, dtype=np.uint8)
    _check(fn(_p8(buf), off.ctypes.data_as(_u64p), _p8(p), _p8(s), _p8(f), _p8(ok), _p8(bitmap), C.c_size_t(n)), "verify_batch")
    return ok.astype(bool), bitmap


def g2pubs_verify_batch(msgs, pks, sigs, inf_flags=None):
    return _verify_batch(_lib().blsmi_g2pubs_verify_batch, 192, 96, msgs, pks, sigs, inf_flags)


def g1pubs_verify_batch(msgs, pks, sigs, inf_flags=None):
    return _verify_batch(_lib().blsmi_g1pubs_verify_batch, 96, 192, msgs, pks, sigs, inf_flags)


def verify_serialized_batch(group, msgs, pks, sigs, check_subgroup=True):
    """Deserialize + Verify in one device pass: compressed keys / signatures as Serialize() emits them.
    Returns (ok, err_pk, err_sig); err_* are the per-element deserialisation error codes (0 = fine)."""
    n = len(msgs)
    buf, off = _msgs(msgs)
    pkc, sgc = (96, 48) if group == "g2pubs" else (48, 96)
    p = _u8(pks, pkc * n) if n else np.zeros(1, np.uint8)
    s = _u8(sigs, sgc * n) if n else np.zeros(1, np.uint8)
    ok = np.zeros(n, dtype=np.uint8); ep = np.zeros(n, dtype=np.uint8); es = np.zeros(n, dtype=np.uint8)
    fn = _lib().blsmi_g2pubs_verify_serialized_batch if group == "g2pubs" else _lib().blsmi_g1pubs_verify_serialized_batch
    _check(fn(_p8(buf), off.ctypes.data_as(_u64p), _p8(p), _p8(s), C.c_int(1 if check_subgroup else 0), _p8(ok), _p8(ep), _p8(es), C.c_size_t(n)), "verify_serialized_batch")
    return ok.astype(bool), ep, es


def g1pubs_verify_with_domain_batch(msgs32, domain8, pks, sigs, inf_flags=None):
    n = len(msgs32)
    buf = _u8(b"".join(bytes(m) for m in msgs32), 32 * n)
    d, p, s = _u8(domain8, 8), _u8(pks, 96 * n), _u8(sigs, 192 * n)
    f = _u8(inf_flags, n) if inf_flags is not None else None
    ok = np.zeros(n, dtype=np.uint8)
    _check(_lib().blsmi_g1pubs_verify_with_domain_batch(_p8(buf), _p8(d), _p8(p), _p8(s), _p8(f), _p8(ok), None, C.c_size_t(n)), "verify_with_domain_batch")
    return ok.astype(bool)


def _verify_aggregate(fn, pkb, sgb, msgs, pks, sig):
    n = len(msgs)
    buf, off = _msgs(msgs)
    p = _u8(pks, pkb * n) if n else np.zeros(1, np.uint8)
    s = _u8(sig, sgb)
    ok = C.c_int(0)
    _check(fn(_p8(buf), off.ctypes.data_as(_u64p), _p8(p), _p8(s), C.c_size_t(n), C.byref(ok)), "verify_aggregate")
    return bool(ok.value)


def g2pubs_verify_aggregate(msgs, pks, sig):
    return _verify_aggregate(_lib().blsmi_g2pubs_verify_aggregate, 192, 96, msgs, pks, sig)


def g1pubs_verify_aggregate(msgs, pks, sig):
    return _verify_aggregate(_lib().blsmi_g1pubs_verify_aggregate, 96, 192, msgs, pks, sig)


def g1pubs_verify_aggregate_with_domain(msgs32, domain8, pks, sig):
    n = len(msgs32)
    buf = _u8(b"".join(bytes(m) for m in msgs32), 32 * n) if n else np.zeros(1, np.uint8)
    d, p, s = _u8(domain8, 8), (_u8(pks, 96 * n) if n else np.zeros(1, np.uint8)), _u8(sig, 192)
    ok = C.c_int(0)
    _check(_lib().blsmi_g1pubs_verify_aggregate_with_domain(_p8(buf), _p8(d), _p8(p), _p8(s), C.c_size_t(n), C.byref(ok)), "verify_aggregate_with_domain")
    return bool(ok.value)


def _verify_aggregate_common(fn, pkb, sgb, msg, pks, sig, n):
    m = _u8(bytes(msg) or b"\0")
    p = _u8(pks, pkb * n) if n else np.zeros(1, np.uint8)
    s = _u8(sig, sgb)
    ok = C.c_int(0)
    _check(fn(_p8(m), C.c_size_t(len(msg)), _p8(p), _p8(s), C.c_size_t(n), C.byref(ok)), "verify_aggregate_common")
    return bool(ok.value)


def g2pubs_verify_aggregate_common(msg, pks, sig, n):
    return _verify_aggregate_common(_lib().blsmi_g2pubs_verify_aggregate_common, 192, 96, msg, pks, sig, n)


def g1pubs_verify_aggregate_common(msg, pks, sig, n):
    return _verify_aggregate_common(_lib().blsmi_g1pubs_verify_aggregate_common, 96, 192, msg, pks, sig, n)


def g1pubs_verify_aggregate_common_with_domain(msg32, domain8, pks, sig, n):
    m, d = _u8(msg32, 32), _u8(domain8, 8)
    p = _u8(pks, 96 * n) if n else np.zeros(1, np.uint8)
    s = _u8(sig, 192)
    ok = C.c_int(0)
    _check(_lib().blsmi_g1pubs_verify_aggregate_common_with_domain(_p8(m), _p8(d), _p8(p), _p8(s), C.c_size_t(n), C.byref(ok)), "verify_aggregate_common_with_domain")
    return bool(ok.value)


def aggregate_partial(group, msgs, pks):
    """Shard-level half of VerifyAggregate: (prod_i MillerLoop(H(m_i), pk_i) as (72,) uint64 in the wire format,
    True if some key was the point at infinity)."""
    n = len(msgs)
    buf, off = _msgs(msgs)
    pkb = 192 if group == "g2pubs" else 96
    p = _u8(pks, pkb * n) if n else np.zeros(1, np.uint8)
    out = np.zeros(72, dtype=np.uint64)
    bad = C.c_int(0)
    fn = _lib().blsmi_g2pubs_aggregate_partial if group == "g2pubs" else _lib().blsmi_g1pubs_aggregate_partial
    _check(fn(_p8(buf), off.ctypes.data_as(_u64p), _p8(p), C.c_size_t(n), out.ctypes.data_as(_u64p), C.byref(bad)), "aggregate_partial")
    return out, bool(bad.value)


def fq12_product(vals):
    x = np.ascontiguousarray(vals, dtype=np.uint64).reshape(-1, 72)
    out = np.zeros(72, dtype=np.uint64)
    _check(_lib().blsmi_fq12_product(x.ctypes.data_as(_u64p), C.c_size_t(x.shape[0]), out.ctypes.data_as(_u64p)), "fq12_product")
    return out


# ---- wire format -----------------------------------------------------------------------------------
def _decompress(fn, ib, ob, data, n, check):
    a = _u8(data, ib * n)
    out = np.zeros((n, ob), dtype=np.uint8)
    inf = np.zeros(n, dtype=np.uint8)
    err = np.zeros(n, dtype=np.uint8)
    _check(fn(_p8(a), C.c_int(int(check)), _p8(out.reshape(-1)), _p8(inf), _p8(err), C.c_size_t(n)), "decompress")
    return out, inf.astype(bool), err


def g1_decompress_batch(data, n, check_subgroup=True):
    return _decompress(_lib().blsmi_g1_decompress_batch, 48, 96, data, n, check_subgroup)


def g2_decompress_batch(data, n, check_subgroup=True):
    return _decompress(_lib().blsmi_g2_decompress_batch, 96, 192, data, n, check_subgroup)


def _compress(fn, ib, ob, pts, n, in_inf):
    a = _u8(pts, ib * n)
    f = _u8(in_inf, n) if in_inf is not None else None
    out = np.zeros((n, ob), dtype=np.uint8)
    _check(fn(_p8(a), _p8(f), _p8(out.reshape(-1)), C.c_size_t(n)), "compress")
    return out


def g1_compress_batch(pts, n, in_inf=None):
    return _compress(_lib().blsmi_g1_compress_batch, 96, 48, pts, n, in_inf)


def g2_compress_batch(pts, n, in_inf=None):
    return _compress(_lib().blsmi_g2_compress_batch, 192, 96, pts, n, in_inf)


# ---- unit-level device ops (parity tests) -------------------------------------------------------------
OPS = dict(FQ_MUL=1, FQ_SQR=2, FQ_ADD=3, FQ_SUB=4, FQ_NEG=5, FQ_INV=6, FQ_SQRT=7, FQ_DBL=8, FQ_CMP=9, FQ_PARITY=10,
           FQ2_MUL=16, FQ2_SQR=17, FQ2_INV=18, FQ2_MUL_NR=19, FQ2_SQRT=20, FQ2_SQRT_ANY=21, FQ2_PARITY=22,
           FQ6_MUL=32, FQ6_SQR=33, FQ6_INV=34, FQ6_FROB1=35, FQ6_MUL_BY_1=36, FQ6_MUL_BY_01=37,
           FQ12_MUL=48, FQ12_SQR=49, FQ12_INV=50, FQ12_FROB1=51, FQ12_FROB2=52, FQ12_FROB3=53, FQ12_CYCLO_SQR=54, FQ12_CYCLO_RUN16=55,
           FQ12_MUL_BY_014=56, FQ12_MUL_BY_LINE_PAIR=57,
           G1_DOUBLE=64, G1_ADD=65, G2_DOUBLE=66, G2_ADD=67, SWU_G1=68, SWU_G2=69,
           ROW_DBL_STEP=80, ROW_DBL_STEP_REF=81, ROW_ADD_STEP=82, ROW_ADD_STEP_REF=83, ROW_G2_DOUBLE=84, ROW_G2_ADD=85, ROW_CLEAR_H2=86)


LANE_PAIR = 0x100


def debug_g2_prepare(g2_aff=None, mode=0):
    """G2AffineToPrepared of one point -> (68, 3, 12) uint64.  mode 0 / 1: computed by the one-tuple-per-lane / lane-pair
    steps; mode 2: the generator table the library prepared at start-up."""
    out = np.zeros(68 * 3 * 12, dtype=np.uint64)
    q = _u8(g2_aff, 192) if g2_aff is not None else None
    _check(_lib().blsmi_debug_g2_prepare(_p8(q), C.c_int(mode), out.ctypes.data_as(_u64p)), "blsmi_debug_g2_prepare")
    return out.reshape(68, 3, 12)


def debug_hash_tail(kind, pts, n):
    """the level program behind a small-batch hash: mapped points -> (hash points (n, 96|192) uint8, good (n,) uint8)"""
    rec, hb = (384 if kind == 1 else 192), (96 if kind == 0 else 192)
    p = _u8(pts, rec * n)
    out = np.zeros((n, hb), dtype=np.uint8); good = np.zeros(n, dtype=np.uint8)
    _check(_lib().blsmi_debug_hash_tail(C.c_int(kind), _p8(p), _p8(out.reshape(-1)), _p8(good), C.c_size_t(n)), "blsmi_debug_hash_tail")
    return out, good


def debug_hash_redo(kind, msgs, good, out, domain8=None):
    """re-hash the messages whose good byte is 0 into `out` (n, 96|192) uint8; returns the updated array"""
    n = len(msgs)
    hb = 96 if kind == 0 else 192
    o = np.ascontiguousarray(out, dtype=np.uint8).reshape(n, hb).copy()
    g = _u8(good, n)
    if kind == 2:
        buf = _u8(b"".join(bytes(m) for m in msgs), 32 * n); off = _u8(domain8, 8)
        offp = off.ctypes.data_as(_u64p)
    else:
        buf, off = _msgs(msgs)
        offp = off.ctypes.data_as(_u64p)
    _check(_lib().blsmi_debug_hash_redo(C.c_int(kind), _p8(buf), offp, _p8(g), _p8(o.reshape(-1)), C.c_size_t(n)), "blsmi_debug_hash_redo")
    return o


LANE_QUAD = 0x200          # BLSMI_OP_LANE_QUAD
LANE_ROW = 0x400           # BLSMI_OP_LANE_ROW


def debug_op(name, a, b=None, lane_pair=False, raw_flag=False, lane_quad=False, lane_row=False):
    op = OPS[name]
    width = 1 if op < 16 else 2 if op < 32 else 6 if op < 48 else 12 if op < 64 or op >= 80 else (3 if op in (64, 65, 68) else 6)
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 6 * width)
    n = a.shape[0]
    out = np.zeros_like(a)
    flag = np.zeros(n, dtype=np.uint8)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 6 * width)
        assert b.shape == a.shape
        bp = b.ctypes.data_as(_u64p)
    _check(_lib().blsmi_debug_op(op | (LANE_PAIR if lane_pair else 0) | (LANE_QUAD if lane_quad else 0) | (LANE_ROW if lane_row else 0), a.ctypes.data_as(_u64p), bp, out.ctypes.data_as(_u64p), _p8(flag), C.c_size_t(n)), "blsmi_debug_op")
    return out, (flag.copy() if raw_flag else flag.astype(bool))


# ---- the reference's in-memory points at the boundary (blsmi 0.6, include/blsmi.h "in-memory points") ------------------------
# bls.G1Projective = 18 x u64, bls.G2Projective = 36 x u64 (x, y, z; each FQ 6 little-endian u64 Montgomery limbs): 144 / 288 bytes.
G1_JAC_BYTES, G2_JAC_BYTES = 144, 288


def _j64(x, nbytes):
    a = _u8(x, nbytes) if nbytes else np.zeros(8, np.uint8)
    return a, a.ctypes.data_as(_u64p)


def _jac_to_affine(fn, jb, wb, jac, n):
    a, pa = _j64(jac, jb * n)
    out = np.zeros(max(1, wb * n), dtype=np.uint8)
    inf = np.zeros(max(1, n), dtype=np.uint8)
    _check(fn(pa, _p8(out), _p8(inf), C.c_size_t(n)), "jac_to_affine_batch")
    return out[:wb * n].tobytes(), inf[:n].astype(bool)


def g1_jac_to_affine_batch(jac, n):
    """G1Projective.ToAffine().SerializeBytes() for n in-memory points -> (n*96 wire bytes, infinity flags)"""
    return _jac_to_affine(_lib().blsmi_g1_jac_to_affine_batch, 144, 96, jac, n)


def g2_jac_to_affine_batch(jac, n):
    return _jac_to_affine(_lib().blsmi_g2_jac_to_affine_batch, 288, 192, jac, n)


def pairing_batch_jac(g1_jac, g2_jac, n):
    """bls.Pairing on n (G1Projective, G2Projective) pairs as the Go heap holds them -> (n, 72) uint64"""
    a, pa = _j64(g1_jac, 144 * n)
    b, pb = _j64(g2_jac, 288 * n)
    out = np.zeros((n, 72), dtype=np.uint64)
    _check(_lib().blsmi_pairing_batch_jac(pa, pb, out.ctypes.data_as(_u64p), C.c_size_t(n)), "blsmi_pairing_batch_jac")
    return out


def _sum_jac(fn, jb, pts, n):
    a, pa = _j64(pts, jb * n)
    out = np.zeros(jb // 8, dtype=np.uint64)
    oinf = C.c_int(0)
    _check(fn(pa, C.c_size_t(n), out.ctypes.data_as(_u64p), C.byref(oinf)), "sum_jac")
    return out.tobytes(), bool(oinf.value)


def g1_sum_jac(pts, n):
    """sum of n in-memory G1 points -> (the sum as an in-memory point with z = 1, is_infinity)"""
    return _sum_jac(_lib().blsmi_g1_sum_jac, 144, pts, n)


def g2_sum_jac(pts, n):
    return _sum_jac(_lib().blsmi_g2_sum_jac, 288, pts, n)


def _verify_batch_jac(fn, pkb, sgb, msgs, pks, sigs):
    n = len(msgs)
    buf, off = _msgs(msgs)
    p, pp = _j64(pks, pkb * n)
    s, ps = _j64(sigs, sgb * n)
    ok = np.zeros(n, dtype=np.uint8)
    bitmap = np.zeros((n + 7) // 8, dtype=np.uint8)
    _check(fn(_p8(buf), off.ctypes.data_as(_u64p), pp, ps, _p8(ok), _p8(bitmap), C.c_size_t(n)), "verify_batch_jac")
    return ok.astype(bool), bitmap


def g2pubs_verify_batch_jac(msgs, pks, sigs):
    return _verify_batch_jac(_lib().blsmi_g2pubs_verify_batch_jac, 288, 144, msgs, pks, sigs)


def g1pubs_verify_batch_jac(msgs, pks, sigs):
    return _verify_batch_jac(_lib().blsmi_g1pubs_verify_batch_jac, 144, 288, msgs, pks, sigs)


def g1pubs_verify_with_domain_batch_jac(msgs32, domain8, pks, sigs):
    n = len(msgs32)
    buf = _u8(b"".join(bytes(m) for m in msgs32), 32 * n)
    d = _u8(domain8, 8)
    p, pp = _j64(pks, 144 * n)
    s, ps = _j64(sigs, 288 * n)
    ok = np.zeros(n, dtype=np.uint8)
    _check(_lib().blsmi_g1pubs_verify_with_domain_batch_jac(_p8(buf), _p8(d), pp, ps, _p8(ok), None, C.c_size_t(n)), "verify_with_domain_batch_jac")
    return ok.astype(bool)


def _verify_aggregate_jac(fn, pkb, sgb, msgs, pks, sig):
    n = len(msgs)
    buf, off = _msgs(msgs)
    p, pp = _j64(pks, pkb * n)
    s, ps = _j64(sig, sgb)
    ok = C.c_int(0)
    _check(fn(_p8(buf), off.ctypes.data_as(_u64p), pp, ps, C.c_size_t(n), C.byref(ok)), "verify_aggregate_jac")
    return bool(ok.value)


def g2pubs_verify_aggregate_jac(msgs, pks, sig):
    return _verify_aggregate_jac(_lib().blsmi_g2pubs_verify_aggregate_jac, 288, 144, msgs, pks, sig)


def g1pubs_verify_aggregate_jac(msgs, pks, sig):
    return _verify_aggregate_jac(_lib().blsmi_g1pubs_verify_aggregate_jac, 144, 288, msgs, pks, sig)


def g1pubs_verify_aggregate_with_domain_jac(msgs32, domain8, pks, sig):
    n = len(msgs32)
    buf = _u8(b"".join(bytes(m) for m in msgs32), 32 * n) if n else np.zeros(1, np.uint8)
    d = _u8(domain8, 8)
    p, pp = _j64(pks, 144 * n)
    s, ps = _j64(sig, 288)
    ok = C.c_int(0)
    _check(_lib().blsmi_g1pubs_verify_aggregate_with_domain_jac(_p8(buf), _p8(d), pp, ps, C.c_size_t(n), C.byref(ok)), "verify_aggregate_with_domain_jac")
    return bool(ok.value)


def _verify_aggregate_common_jac(fn, pkb, sgb, msg, pks, sig, n):
    m = _u8(bytes(msg) or b"\0")
    p, pp = _j64(pks, pkb * n)
    s, ps = _j64(sig, sgb)
    ok = C.c_int(0)
    _check(fn(_p8(m), C.c_size_t(len(msg)), pp, ps, C.c_size_t(n), C.byref(ok)), "verify_aggregate_common_jac")
    return bool(ok.value)


def g2pubs_verify_aggregate_common_jac(msg, pks, sig, n):
    return _verify_aggregate_common_jac(_lib().blsmi_g2pubs_verify_aggregate_common_jac, 288, 144, msg, pks, sig, n)


def g1pubs_verify_aggregate_common_jac(msg, pks, sig, n):
    return _verify_aggregate_common_jac(_lib().blsmi_g1pubs_verify_aggregate_common_jac, 144, 288, msg, pks, sig, n)


def g1pubs_verify_aggregate_common_with_domain_jac(msg32, domain8, pks, sig, n):
    m, d = _u8(msg32, 32), _u8(domain8, 8)
    p, pp = _j64(pks, 144 * n)
    s, ps = _j64(sig, 288)
    ok = C.c_int(0)
    _check(_lib().blsmi_g1pubs_verify_aggregate_common_with_domain_jac(_p8(m), _p8(d), pp, ps, C.c_size_t(n), C.byref(ok)), "verify_aggregate_common_with_domain_jac")
    return bool(ok.value)


class PreparedKeysJac(PreparedKeys):
    """PreparedKeys made from n in-memory G2 points (blsmi_g2_prepared_create_jac)"""

    def __init__(self, g2_jac, n):  # (does not call the affine constructor)
        a, pa = _j64(g2_jac, 288 * n)
        h = C.c_void_p(0)
        _check(_lib().blsmi_g2_prepared_create_jac(pa, C.c_size_t(n), C.byref(h)), "blsmi_g2_prepared_create_jac")
        self.ptr, self.n = h.value, n


def g2pubs_verify_batch_prepared_jac(msgs, prepared, key_idx, sigs):
    """g2pubs.Verify x n over prepared keys, the signatures as in-memory G1 points"""
    n = len(msgs)
    buf, off = _msgs(msgs)
    ptr = prepared.ptr if isinstance(prepared, PreparedKeys) else prepared
    s, ps = _j64(sigs, 144 * n)
    idx = None if key_idx is None else np.ascontiguousarray(key_idx, dtype=np.uint32)
    ok = np.zeros(n, dtype=np.uint8)
    _check(_lib().blsmi_g2pubs_verify_batch_prepared_jac(_p8(buf), off.ctypes.data_as(_u64p), C.c_void_p(ptr), None if idx is None else idx.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                        ps, _p8(ok), None, C.c_size_t(n)), "verify_batch_prepared_jac")
    return ok.astype(bool)


def g2pubs_verify_aggregate_prepared_jac(msgs, prepared, key_idx, sig):
    n = len(msgs)
    buf, off = _msgs(msgs)
    ptr = prepared.ptr if isinstance(prepared, PreparedKeys) else prepared
    s, ps = _j64(sig, 144)
    idx = None if key_idx is None else np.ascontiguousarray(key_idx, dtype=np.uint32)
    ok = C.c_int(0)
    _check(_lib().blsmi_g2pubs_verify_aggregate_prepared_jac(_p8(buf), off.ctypes.data_as(_u64p), C.c_void_p(ptr), None if idx is None else idx.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                            ps, C.c_size_t(n), C.byref(ok)), "verify_aggregate_prepared_jac")
    return bool(ok.value)


def pairing_batch_jac_dev(d_g1_jac, d_g2_jac, d_out, n, stream=0):
    _check(_lib().blsmi_pairing_batch_jac_dev(C.c_void_p(d_g1_jac), C.c_void_p(d_g2_jac), C.c_void_p(d_out), C.c_size_t(n), C.c_void_p(stream)), "blsmi_pairing_batch_jac_dev")


def verify_batch_jac_dev(group, d_msgs, d_off, d_pks_jac, d_sigs_jac, d_ok, n, stream=0):
    """group: "g2pubs" / "g1pubs" / "g1pubs_with_domain" (d_off = the 8-byte domain on the device)"""
    fn = {"g2pubs": _lib().blsmi_g2pubs_verify_batch_jac_dev, "g1pubs": _lib().blsmi_g1pubs_verify_batch_jac_dev,
          "g1pubs_with_domain": _lib().blsmi_g1pubs_verify_with_domain_batch_jac_dev}[group]
    _check(fn(C.c_void_p(d_msgs), C.c_void_p(d_off), C.c_void_p(d_pks_jac), C.c_void_p(d_sigs_jac), C.c_void_p(d_ok), C.c_size_t(n), C.c_void_p(stream)), "verify_batch_jac_dev")


def _mul_gen_jac(fn, jb, scalars, n):
    sc = _u8(scalars, 32 * n)
    out = np.zeros(max(1, jb // 8 * n), dtype=np.uint64)
    _check(fn(_p8(sc), out.ctypes.data_as(_u64p), C.c_size_t(n)), "mul_generator_batch_jac")
    return out[:jb // 8 * n].view(np.uint8).reshape(n, jb)


def g1_mul_generator_batch_jac(scalars, n):
    """PrivToPub for g1pubs: k_i * G1 as (n, 144) bytes of bls.G1Projective records (z = 1; (0, 1, 0) for k = 0 mod r)"""
    return _mul_gen_jac(_lib().blsmi_g1_mul_generator_batch_jac, 144, scalars, n)


def g2_mul_generator_batch_jac(scalars, n):
    return _mul_gen_jac(_lib().blsmi_g2_mul_generator_batch_jac, 288, scalars, n)


def _sign_jac(fn, jb, n, head, sks):
    sk = _u8(sks, 32 * n)
    out = np.zeros(max(1, jb // 8 * n), dtype=np.uint64)
    _check(fn(*head, _p8(sk), out.ctypes.data_as(_u64p), C.c_size_t(n)), "sign_batch_jac")
    return out[:jb // 8 * n].view(np.uint8).reshape(n, jb)


def g2pubs_sign_batch_jac(msgs, sks):
    buf, off = _msgs(msgs)
    return _sign_jac(_lib().blsmi_g2pubs_sign_batch_jac, 144, len(msgs), (_p8(buf), off.ctypes.data_as(_u64p)), sks)


def g1pubs_sign_batch_jac(msgs, sks):
    buf, off = _msgs(msgs)
    return _sign_jac(_lib().blsmi_g1pubs_sign_batch_jac, 288, len(msgs), (_p8(buf), off.ctypes.data_as(_u64p)), sks)


def g1pubs_sign_with_domain_batch_jac(msgs32, domain8, sks):
    n = len(msgs32)
    buf = _u8(b"".join(bytes(m) for m in msgs32), 32 * n)
    d = _u8(domain8, 8)
    return _sign_jac(_lib().blsmi_g1pubs_sign_with_domain_batch_jac, 288, n, (_p8(buf), _p8(d)), sks)
