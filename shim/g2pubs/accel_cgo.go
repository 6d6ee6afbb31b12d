// +build cgo,blsmi

// accel_cgo.go: the verify surface of package g2pubs (PublicKey in G2, Signature in G1) on libblsmi.so.
//
// Drop this file into github.com/phoreproject/bls/g2pubs and build with `-tags blsmi`; the upstream
// implementations of Verify, VerifyAggregate and VerifyAggregateCommon (g2pubs/bls.go:159-162, 240-270,
// 275-278) move behind `// +build !blsmi`.  Everything else in bls.go (types, Sign, PrivToPub,
// serialisation, the aggregation helpers) stays as it is; SignBatch below is an addition for bulk signing.  The C prototypes are include/blsmi.h;
// tests/test_shim.py checks every C.blsmi_* call below against it (name, arity and the C type of every argument, in order).
// Go version: the language of the reference's go.mod (`go 1.13`, /root/reference/go.mod:15) -- nothing newer is used (no unsafe.Slice /
// unsafe.Add, no generics, no `any`, old-style build tags); cgo as shipped with go >= 1.13.
package g2pubs

/*
#cgo CFLAGS: -I${SRCDIR}/../../blsmi/include
#cgo LDFLAGS: -L${SRCDIR}/../../blsmi/bls_amd -lblsmi -Wl,-rpath,${SRCDIR}/../../blsmi/bls_amd
#include "blsmi.h"
*/
import "C"

import (
	"runtime"
	"strconv"
	"unsafe"

	"github.com/phoreproject/bls"
)

// must: ONE error policy for every entry point.  A non-zero return code is a device / runtime failure (BLSMI_E_HIP, _NOMEM, _RCCL, _ARG:
// include/blsmi.h), never a verdict -- `false` from a Verify* function only ever means that the pairing check failed, as upstream
// (g2pubs/bls.go).  A consensus caller must not mistake a failed hipMalloc for an invalid signature: panic, like upstream does on its own
// internal errors (a deployment that prefers to degrade wraps the call sites in recover() and re-runs them on the upstream CPU path).
func must(rc C.int, what string) {
	if rc != 0 {
		panic("blsmi: " + what + " failed (" + strconv.Itoa(int(rc)) + ")")
	}
}

// Compile-time layout guards: the *_jac entry points read the structs below as 36 / 18 contiguous uint64 (x, y, z; FQ2 = two FQ; FQ =
// FQRepr = [6]uint64: g2.go:298-302, g1.go:252-256, fq2.go:14-17, fq.go:11-13, fqrepr.go:14).  If upstream ever changes a layout the
// index below is no longer the constant 0 and the package stops compiling (constant index out of range / constant overflow) instead of
// handing the device garbage.
var _ = [1]struct{}{}[unsafe.Sizeof(bls.G2Projective{})-288]
var _ = [1]struct{}{}[unsafe.Sizeof(bls.G1Projective{})-144]
var _ = [1]struct{}{}[unsafe.Sizeof(bls.FQRepr{})-48]

func init() {
	// Every visible device (<= 0), or the first n.  Must precede any other blsmi call.  With more than one
	// device, batches of BLSMI_SHARD_MIN tuples or more are split over the GPUs inside the library.
	must(C.blsmi_init_devices(0), "init_devices (no usable MI355X, or RCCL missing on a multi-GPU node)")
}

func u8(b []byte) *C.uint8_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&b[0]))
}

// packMsgs concatenates the messages and records n+1 offsets (message i = m[off[i]:off[i+1]]).
func packMsgs(msgs [][]byte) (m []byte, off []C.uint64_t) {
	off = make([]C.uint64_t, len(msgs)+1)
	for i, x := range msgs {
		m = append(m, x...)
		off[i+1] = C.uint64_t(len(m))
	}
	if len(m) == 0 {
		m = []byte{0}
	}
	return
}

// The points cross the boundary AS THE GO HEAP HOLDS THEM (blsmi 0.6, the *_jac entry points): a *bls.G2Projective is 36 contiguous
// uint64 -- x, y, z, each FQ2 two FQ, each FQ 6 little-endian Montgomery limbs (g2.go:298-302, fq2.go:14-17, fq.go:11-13, fqrepr.go:14)
// -- and a *bls.G1Projective 18 (g1.go:252-256).  One 288 / 144-byte copy per point: no ToAffine (an Fq inversion, g2.go:365-386), no
// SerializeBytes (MontReduce + byte swap, g2.go:172-186) on a host core; the library runs ToAffine on the device.  z == 0 is the point at
// infinity there (G2Projective.IsZero), so no flag bytes travel either.

// packKeys: n*36 uint64, the public keys' G2Projective structs.
func packKeys(pubs []*PublicKey) []C.uint64_t {
	pk := make([]C.uint64_t, 36*len(pubs))
	for i := range pubs {
		copy(pk[36*i:36*i+36], (*[36]C.uint64_t)(unsafe.Pointer(pubs[i].p))[:])
	}
	return pk
}

// packSigs: n*18 uint64, the signatures' G1Projective structs.
func packSigs(sigs []*Signature) []C.uint64_t {
	sg := make([]C.uint64_t, 18*len(sigs))
	for i := range sigs {
		copy(sg[18*i:18*i+18], (*[18]C.uint64_t)(unsafe.Pointer(sigs[i].s))[:])
	}
	return sg
}

func u64(b []C.uint64_t) *C.uint64_t {
	if len(b) == 0 {
		return nil
	}
	return &b[0]
}

// sigWords: the one signature of an aggregate call, in place (the struct is plain memory without Go pointers: cgo may read it directly).
func sigWords(s *Signature) *C.uint64_t {
	return (*C.uint64_t)(unsafe.Pointer(s.s))
}

// VerifyBatch is the batch form the one-tuple-per-call API lacks: out[i] = Verify(msgs[i], pubs[i], sigs[i]).
func VerifyBatch(msgs [][]byte, pubs []*PublicKey, sigs []*Signature) []bool {
	n := len(msgs)
	out := make([]bool, n)
	if n == 0 {
		return out
	}
	m, off := packMsgs(msgs)
	pk := packKeys(pubs)
	sg := packSigs(sigs)
	ok := make([]byte, n)
	must(C.blsmi_g2pubs_verify_batch_jac(u8(m), &off[0], u64(pk), u64(sg), u8(ok), nil, C.size_t(n)), "g2pubs_verify_batch_jac")
	for i := range ok {
		out[i] = ok[i] != 0
	}
	return out
}

// Verify keeps the upstream signature (g2pubs/bls.go:159).  A lone Verify is faster on the device than on
// one CPU core (2.0 ms against 3.7 ms): blsmi_prefer_cpu(BLSMI_SHAPE_VERIFY, 1) is 0, so there is no CPU branch.
func Verify(m []byte, pub *PublicKey, sig *Signature) bool {
	return VerifyBatch([][]byte{m}, []*PublicKey{pub}, []*Signature{sig})[0]
}

// VerifyAggregate keeps the upstream signature (g2pubs/bls.go:240): length check here, duplicate-message
// rejection (bls.go:245-261) inside the library.  A key or the signature at infinity: false (upstream panics in MillerLoop).
func (s *Signature) VerifyAggregate(pubKeys []*PublicKey, msgs [][]byte) bool {
	if len(pubKeys) != len(msgs) {
		return false
	}
	m, off := packMsgs(msgs)
	pk := packKeys(pubKeys)
	var ok C.int
	must(C.blsmi_g2pubs_verify_aggregate_jac(u8(m), &off[0], u64(pk), sigWords(s), C.size_t(len(msgs)), &ok), "g2pubs_verify_aggregate_jac")
	return ok != 0
}

// VerifyAggregateCommon keeps the upstream signature (g2pubs/bls.go:275): the key sum stays on the upstream
// path for small sets (one Jacobian addition is 6.5 us on a CPU core) and goes to the device for large ones,
// where the keys are added as the Jacobian points they are.
func (s *Signature) VerifyAggregateCommon(pubKeys []*PublicKey, msg []byte) bool {
	if C.blsmi_prefer_cpu(C.BLSMI_SHAPE_POINT_ADD, C.size_t(len(pubKeys))) != 0 {
		return Verify(msg, AggregatePublicKeys(pubKeys), s)
	}
	pk := packKeys(pubKeys)
	one := []byte{0}
	mp := u8(msg)
	if len(msg) == 0 {
		mp = u8(one)
	}
	var ok C.int
	must(C.blsmi_g2pubs_verify_aggregate_common_jac(mp, C.size_t(len(msg)), u64(pk), sigWords(s), C.size_t(len(pubKeys)), &ok), "g2pubs_verify_aggregate_common_jac")
	return ok != 0
}

// SumPublicKeys is AggregatePublicKeys (g2pubs/bls.go:180-192) for large sets: the points are summed on the device as they are and the sum
// comes back as a G2Projective (z = 1; the reference's G2ProjectiveZero for the empty or cancelling sum).
func SumPublicKeys(pubKeys []*PublicKey) *PublicKey {
	if C.blsmi_prefer_cpu(C.BLSMI_SHAPE_POINT_ADD, C.size_t(len(pubKeys))) != 0 {
		return AggregatePublicKeys(pubKeys)
	}
	pk := packKeys(pubKeys)
	out := new(bls.G2Projective)
	var inf C.int
	must(C.blsmi_g2_sum_jac(u64(pk), C.size_t(len(pubKeys)), (*C.uint64_t)(unsafe.Pointer(out)), &inf), "g2_sum_jac")
	return &PublicKey{p: out}
}

// SumSignatures is AggregateSignatures (g2pubs/bls.go:165-177) the same way.
func SumSignatures(sigs []*Signature) *Signature {
	if C.blsmi_prefer_cpu(C.BLSMI_SHAPE_POINT_ADD, C.size_t(len(sigs))) != 0 {
		return AggregateSignatures(sigs)
	}
	sg := packSigs(sigs)
	out := new(bls.G1Projective)
	var inf C.int
	must(C.blsmi_g1_sum_jac(u64(sg), C.size_t(len(sigs)), (*C.uint64_t)(unsafe.Pointer(out)), &inf), "g1_sum_jac")
	return &Signature{s: out}
}

// VerifySerializedBatch: DeserializePublicKey + DeserializeSignature + Verify (g2pubs/bls.go:91-98, 33-40,
// 159-162) for n tuples in one device pass, straight from the 96 / 48-byte Serialize() forms, subgroup
// checks included.
func VerifySerializedBatch(msgs [][]byte, pubs [][96]byte, sigs [][48]byte) []bool {
	n := len(msgs)
	out := make([]bool, n)
	if n == 0 {
		return out
	}
	m, off := packMsgs(msgs)
	ok := make([]byte, n)
	must(C.blsmi_g2pubs_verify_serialized_batch(u8(m), &off[0],
		(*C.uint8_t)(unsafe.Pointer(&pubs[0])), (*C.uint8_t)(unsafe.Pointer(&sigs[0])), 1,
		u8(ok), nil, nil, C.size_t(n)), "g2pubs_verify_serialized_batch")
	for i := range ok {
		out[i] = ok[i] != 0
	}
	return out
}

// PreparedKeys: n public keys run through G2AffineToPrepared (g2.go:639-801) once, resident in device
// memory the library owns (24 704 bytes per key).  Verify runs the preparation on every call upstream
// (pairing.go:140-147); a validator set prepares once and passes key INDICES afterwards.
type PreparedKeys struct {
	h unsafe.Pointer
	n int
}

func PrepareKeys(pubs []*PublicKey) *PreparedKeys {
	pk := packKeys(pubs) // a key at infinity (z == 0) keeps that mark in its table: every verdict over it is false
	var h unsafe.Pointer
	must(C.blsmi_g2_prepared_create_jac(u64(pk), C.size_t(len(pubs)), &h), "g2_prepared_create_jac")
	k := &PreparedKeys{h, len(pubs)}
	runtime.SetFinalizer(k, func(k *PreparedKeys) { k.Close() })
	return k
}

func (k *PreparedKeys) Close() {
	if k.h != nil {
		C.blsmi_g2_prepared_destroy(k.h)
		k.h = nil
	}
}

// VerifyBatchPrepared: out[i] = Verify(msgs[i], keys[keyIdx[i]], sigs[i]).
func VerifyBatchPrepared(msgs [][]byte, keys *PreparedKeys, keyIdx []uint32, sigs []*Signature) []bool {
	n := len(msgs)
	out := make([]bool, n)
	if n == 0 {
		return out
	}
	m, off := packMsgs(msgs)
	sg := packSigs(sigs)
	ok := make([]byte, n)
	must(C.blsmi_g2pubs_verify_batch_prepared_jac(u8(m), &off[0], keys.h, (*C.uint32_t)(unsafe.Pointer(&keyIdx[0])),
		u64(sg), u8(ok), nil, C.size_t(n)), "g2pubs_verify_batch_prepared_jac")
	for i := range ok {
		out[i] = ok[i] != 0
	}
	return out
}

// Trim hands the temporaries the library keeps for future calls back to the driver (after a burst of very
// large calls, or before another library in the process needs the HBM).
func Trim() uint64 {
	var freed C.size_t
	C.blsmi_trim(0, &freed)
	return uint64(freed)
}

// SignBatch is the batch form of Sign (g2pubs/bls.go:132-135): out[i] = Sign(msgs[i], keys[i]) = keys[i] * HashG1(msgs[i]), one library
// call for the n signatures (hash and multiplication both on the device).  The secret scalars cross the PCIe bus; a lone Sign is faster on
// the upstream CPU path (blsmi_prefer_cpu), which is what this function takes for small n.
func SignBatch(msgs [][]byte, keys []*SecretKey) []*Signature {
	n := len(msgs)
	out := make([]*Signature, n)
	if n == 0 {
		return out
	}
	if C.blsmi_prefer_cpu(C.BLSMI_SHAPE_SIGN, C.size_t(n)) != 0 {
		for i := range msgs {
			out[i] = Sign(msgs[i], keys[i])
		}
		return out
	}
	m, off := packMsgs(msgs)
	sk := make([]byte, 0, 32*n)
	for i := range keys {
		kb := keys[i].Serialize() // g2pubs/bls.go:115-117: 32 bytes big-endian
		sk = append(sk, kb[:]...)
	}
	// the signatures come back as G1Projective records (z = 1; the reference's zero point for sk = 0 mod r): one copy each
	sg := make([]C.uint64_t, 18*n)
	must(C.blsmi_g2pubs_sign_batch_jac(u8(m), &off[0], u8(sk), &sg[0], C.size_t(n)), "g2pubs_sign_batch_jac")
	for i := range out {
		p := new(bls.G1Projective)
		copy((*[18]C.uint64_t)(unsafe.Pointer(p))[:], sg[18*i:18*i+18])
		out[i] = &Signature{s: p}
	}
	return out
}
