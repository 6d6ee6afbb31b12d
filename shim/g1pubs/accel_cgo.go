// +build cgo,blsmi

// accel_cgo.go: the verify surface of package g1pubs (PublicKey in G1, 96 B; Signature in G2, 192 B) on
// libblsmi.so.
//
// Drop this file into github.com/phoreproject/bls/g1pubs and build with `-tags blsmi`; the upstream
// implementations of Verify, VerifyWithDomain, VerifyAggregate, VerifyAggregateCommon,
// VerifyAggregateCommonWithDomain and VerifyAggregateWithDomain (g1pubs/bls.go:165-174, 252-311) move
// behind `// +build !blsmi`.  bls.go keeps types, Sign, SignWithDomain, PrivToPub, (de)serialisation and
// the aggregation helpers.  tests/test_shim.py checks every C.blsmi_* call below against include/blsmi.h (name, arity and the
// C type of every argument, in order).  Go version: the language of the reference's go.mod (`go 1.13`, /root/reference/go.mod:15)
// -- nothing newer is used (no unsafe.Slice / unsafe.Add, no generics, no `any`, old-style build tags).
package g1pubs

/*
#cgo CFLAGS: -I${SRCDIR}/../../blsmi/include
#cgo LDFLAGS: -L${SRCDIR}/../../blsmi/bls_amd -lblsmi -Wl,-rpath,${SRCDIR}/../../blsmi/bls_amd
#include "blsmi.h"
*/
import "C"

import (
	"strconv"
	"unsafe"

	"github.com/phoreproject/bls"
)

// must: ONE error policy for every entry point.  A non-zero return code is a device / runtime failure (BLSMI_E_HIP, _NOMEM, _RCCL, _ARG:
// include/blsmi.h), never a verdict -- `false` from a Verify* function only ever means that the pairing check failed, as upstream
// (g1pubs/bls.go).  A consensus caller must not mistake a failed hipMalloc for an invalid signature: panic, like upstream does on its own
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
	must(C.blsmi_init_devices(0), "init_devices (no usable MI355X, or RCCL missing on a multi-GPU node)")
}

func u8(b []byte) *C.uint8_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&b[0]))
}

// The points cross the boundary AS THE GO HEAP HOLDS THEM (blsmi 0.6, the *_jac entry points): a *bls.G1Projective is 18 contiguous
// uint64 -- x, y, z, each FQ 6 little-endian Montgomery limbs (g1.go:252-256, fq.go:11-13, fqrepr.go:14) -- and a *bls.G2Projective 36
// (g2.go:298-302).  One 144 / 288-byte copy per point: no ToAffine (an Fq inversion, g1.go:322-340 -- run even at z = 1), no
// SerializeBytes (MontReduce + byte swap, g1.go:157-167) on a host core; the library runs ToAffine on the device.  z == 0 is the point at
// infinity there (G1Projective.IsZero, g1.go:287-289), so no flag bytes travel either.

// packKeys: n*18 uint64, the public keys' G1Projective structs.
func packKeys(pubs []*PublicKey) []C.uint64_t {
	pk := make([]C.uint64_t, 18*len(pubs))
	for i := range pubs {
		copy(pk[18*i:18*i+18], (*[18]C.uint64_t)(unsafe.Pointer(pubs[i].p))[:])
	}
	return pk
}

// packSigs: n*36 uint64, the signatures' G2Projective structs.
func packSigs(sigs []*Signature) []C.uint64_t {
	sg := make([]C.uint64_t, 36*len(sigs))
	for i := range sigs {
		copy(sg[36*i:36*i+36], (*[36]C.uint64_t)(unsafe.Pointer(sigs[i].s))[:])
	}
	return sg
}

func u64(b []C.uint64_t) *C.uint64_t {
	if len(b) == 0 {
		return nil
	}
	return &b[0]
}

// sigWords: the one signature of an aggregate call, in place (plain memory without Go pointers: cgo may read it directly).
func sigWords(s *Signature) *C.uint64_t {
	return (*C.uint64_t)(unsafe.Pointer(s.s))
}

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

// VerifyBatch: n independent Verify() in one call (g1pubs/bls.go:165-168 per tuple).
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
	must(C.blsmi_g1pubs_verify_batch_jac(u8(m), &off[0], u64(pk), u64(sg), u8(ok), nil, C.size_t(n)), "g1pubs_verify_batch_jac")
	for i := range ok {
		out[i] = ok[i] != 0
	}
	return out
}

// Verify keeps the upstream signature (g1pubs/bls.go:165).
func Verify(m []byte, pub *PublicKey, sig *Signature) bool {
	return VerifyBatch([][]byte{m}, []*PublicKey{pub}, []*Signature{sig})[0]
}

// VerifyWithDomain keeps the upstream signature (g1pubs/bls.go:171).
func VerifyWithDomain(m [32]byte, pub *PublicKey, sig *Signature, domain [8]byte) bool {
	return VerifyWithDomainBatch([][32]byte{m}, []*PublicKey{pub}, []*Signature{sig}, domain)[0]
}

// VerifyWithDomainBatch: out[i] = VerifyWithDomain(msgs[i], pubs[i], sigs[i], domain).
func VerifyWithDomainBatch(msgs [][32]byte, pubs []*PublicKey, sigs []*Signature, domain [8]byte) []bool {
	n := len(msgs)
	out := make([]bool, n)
	if n == 0 {
		return out
	}
	pk := packKeys(pubs)
	sg := packSigs(sigs)
	ok := make([]byte, n)
	must(C.blsmi_g1pubs_verify_with_domain_batch_jac((*C.uint8_t)(unsafe.Pointer(&msgs[0])), (*C.uint8_t)(unsafe.Pointer(&domain[0])),
		u64(pk), u64(sg), u8(ok), nil, C.size_t(n)), "g1pubs_verify_with_domain_batch_jac")
	for i := range ok {
		out[i] = ok[i] != 0
	}
	return out
}

// VerifyAggregate keeps the upstream signature (g1pubs/bls.go:252): length check here, duplicate
// rejection in the library.
func (s *Signature) VerifyAggregate(pubKeys []*PublicKey, msgs [][]byte) bool {
	if len(pubKeys) != len(msgs) {
		return false
	}
	m, off := packMsgs(msgs)
	pk := packKeys(pubKeys) // a key or the signature at infinity (z == 0): verdict false (upstream panics in MillerLoop)
	var ok C.int
	must(C.blsmi_g1pubs_verify_aggregate_jac(u8(m), &off[0], u64(pk), sigWords(s), C.size_t(len(msgs)), &ok), "g1pubs_verify_aggregate_jac")
	return ok != 0
}

// VerifyAggregateCommon keeps the upstream signature (g1pubs/bls.go:287).
func (s *Signature) VerifyAggregateCommon(pubKeys []*PublicKey, msg []byte) bool {
	if C.blsmi_prefer_cpu(C.BLSMI_SHAPE_POINT_ADD, C.size_t(len(pubKeys))) != 0 {
		return Verify(msg, AggregatePublicKeys(pubKeys), s) // a handful of keys: sum them on the upstream path
	}
	pk := packKeys(pubKeys)
	one := []byte{0}
	mp := u8(msg)
	if len(msg) == 0 {
		mp = u8(one)
	}
	var ok C.int
	must(C.blsmi_g1pubs_verify_aggregate_common_jac(mp, C.size_t(len(msg)), u64(pk), sigWords(s), C.size_t(len(pubKeys)), &ok), "g1pubs_verify_aggregate_common_jac")
	return ok != 0
}

// VerifyAggregateCommonWithDomain keeps the upstream signature (g1pubs/bls.go:294).
func (s *Signature) VerifyAggregateCommonWithDomain(pubKeys []*PublicKey, msg [32]byte, domain [8]byte) bool {
	pk := packKeys(pubKeys)
	var ok C.int
	must(C.blsmi_g1pubs_verify_aggregate_common_with_domain_jac((*C.uint8_t)(unsafe.Pointer(&msg[0])), (*C.uint8_t)(unsafe.Pointer(&domain[0])),
		u64(pk), sigWords(s), C.size_t(len(pubKeys)), &ok), "g1pubs_verify_aggregate_common_with_domain_jac")
	return ok != 0
}

// VerifyAggregateWithDomain keeps the upstream signature (g1pubs/bls.go:300); no duplicate check upstream
// either.  (Upstream on zero messages compares Pairing(G1One, sig) with 1 -- true only for the infinity
// signature, where it panics first; the shim returns false.)
func (s *Signature) VerifyAggregateWithDomain(pubKeys []*PublicKey, msgs [][32]byte, domain [8]byte) bool {
	if len(pubKeys) != len(msgs) {
		return false
	}
	if len(msgs) == 0 {
		return false
	}
	pk := packKeys(pubKeys)
	var ok C.int
	must(C.blsmi_g1pubs_verify_aggregate_with_domain_jac((*C.uint8_t)(unsafe.Pointer(&msgs[0])), (*C.uint8_t)(unsafe.Pointer(&domain[0])),
		u64(pk), sigWords(s), C.size_t(len(msgs)), &ok), "g1pubs_verify_aggregate_with_domain_jac")
	return ok != 0
}

// SumPublicKeys is AggregatePublicKeys (g1pubs/bls.go:192-204) for large sets: the points are summed on the device as they are and the sum
// comes back as a G1Projective (z = 1; the reference's G1ProjectiveZero for the empty or cancelling sum).
func SumPublicKeys(pubKeys []*PublicKey) *PublicKey {
	if C.blsmi_prefer_cpu(C.BLSMI_SHAPE_POINT_ADD, C.size_t(len(pubKeys))) != 0 {
		return AggregatePublicKeys(pubKeys)
	}
	pk := packKeys(pubKeys)
	out := new(bls.G1Projective)
	var inf C.int
	must(C.blsmi_g1_sum_jac(u64(pk), C.size_t(len(pubKeys)), (*C.uint64_t)(unsafe.Pointer(out)), &inf), "g1_sum_jac")
	return &PublicKey{p: out}
}

// SumSignatures is AggregateSignatures (g1pubs/bls.go:177-189) the same way.
func SumSignatures(sigs []*Signature) *Signature {
	if C.blsmi_prefer_cpu(C.BLSMI_SHAPE_POINT_ADD, C.size_t(len(sigs))) != 0 {
		return AggregateSignatures(sigs)
	}
	sg := packSigs(sigs)
	out := new(bls.G2Projective)
	var inf C.int
	must(C.blsmi_g2_sum_jac(u64(sg), C.size_t(len(sigs)), (*C.uint64_t)(unsafe.Pointer(out)), &inf), "g2_sum_jac")
	return &Signature{s: out}
}

// VerifySerializedBatch: DeserializePublicKey + DeserializeSignature + Verify for n tuples in one device
// pass, from the 48 / 96-byte Serialize() forms (g1pubs/bls.go:18-20, 67-69), subgroup checks included.
func VerifySerializedBatch(msgs [][]byte, pubs [][48]byte, sigs [][96]byte) []bool {
	n := len(msgs)
	out := make([]bool, n)
	if n == 0 {
		return out
	}
	m, off := packMsgs(msgs)
	ok := make([]byte, n)
	must(C.blsmi_g1pubs_verify_serialized_batch(u8(m), &off[0],
		(*C.uint8_t)(unsafe.Pointer(&pubs[0])), (*C.uint8_t)(unsafe.Pointer(&sigs[0])), 1,
		u8(ok), nil, nil, C.size_t(n)), "g1pubs_verify_serialized_batch")
	for i := range ok {
		out[i] = ok[i] != 0
	}
	return out
}

func secretBytes(keys []*SecretKey) []byte {
	sk := make([]byte, 0, 32*len(keys))
	for i := range keys {
		kb := keys[i].Serialize() // 32 bytes big-endian
		sk = append(sk, kb[:]...)
	}
	return sk
}

// sigsFromWords: the library hands signatures back as G2Projective records (z = 1; the reference's zero point for sk = 0 mod r): one copy each.
func sigsFromWords(sg []C.uint64_t, n int) []*Signature {
	out := make([]*Signature, n)
	for i := range out {
		p := new(bls.G2Projective)
		copy((*[36]C.uint64_t)(unsafe.Pointer(p))[:], sg[36*i:36*i+36])
		out[i] = &Signature{s: p}
	}
	return out
}

// SignBatch is the batch form of Sign (g1pubs/bls.go:132-135): out[i] = keys[i] * HashG2(msgs[i]), one library call.  Small batches stay
// on the upstream CPU path (blsmi_prefer_cpu); the secret scalars cross the PCIe bus otherwise.
func SignBatch(msgs [][]byte, keys []*SecretKey) []*Signature {
	n := len(msgs)
	if n == 0 {
		return nil
	}
	if C.blsmi_prefer_cpu(C.BLSMI_SHAPE_SIGN, C.size_t(n)) != 0 {
		out := make([]*Signature, n)
		for i := range msgs {
			out[i] = Sign(msgs[i], keys[i])
		}
		return out
	}
	m, off := packMsgs(msgs)
	sg := make([]C.uint64_t, 36*n)
	must(C.blsmi_g1pubs_sign_batch_jac(u8(m), &off[0], u8(secretBytes(keys)), &sg[0], C.size_t(n)), "g1pubs_sign_batch_jac")
	return sigsFromWords(sg, n)
}

// SignWithDomainBatch: out[i] = SignWithDomain(msgs[i], keys[i], domain) (g1pubs/bls.go:138-141).
func SignWithDomainBatch(msgs [][32]byte, keys []*SecretKey, domain [8]byte) []*Signature {
	n := len(msgs)
	if n == 0 {
		return nil
	}
	sg := make([]C.uint64_t, 36*n)
	must(C.blsmi_g1pubs_sign_with_domain_batch_jac((*C.uint8_t)(unsafe.Pointer(&msgs[0])), (*C.uint8_t)(unsafe.Pointer(&domain[0])),
		u8(secretBytes(keys)), &sg[0], C.size_t(n)), "g1pubs_sign_with_domain_batch_jac")
	return sigsFromWords(sg, n)
}
