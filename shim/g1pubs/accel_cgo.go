// +build cgo,blsmi

// accel_cgo.go: the verify surface of package g1pubs (PublicKey in G1, 96 B; Signature in G2, 192 B) on
// libblsmi.so.
//
// Drop this file into github.com/phoreproject/bls/g1pubs and build with `-tags blsmi`; the upstream
// implementations of Verify, VerifyWithDomain, VerifyAggregate, VerifyAggregateCommon,
// VerifyAggregateCommonWithDomain and VerifyAggregateWithDomain (g1pubs/bls.go:165-174, 252-311) move
// behind `// +build !blsmi`.  bls.go keeps types, Sign, SignWithDomain, PrivToPub, (de)serialisation and
// the aggregation helpers.  tests/test_shim.py checks every C.blsmi_* call below against include/blsmi.h.
package g1pubs

/*
#cgo CFLAGS: -I${SRCDIR}/../../blsmi/include
#cgo LDFLAGS: -L${SRCDIR}/../../blsmi/bls_amd -lblsmi -Wl,-rpath,${SRCDIR}/../../blsmi/bls_amd
#include "blsmi.h"
*/
import "C"

import (
	"unsafe"

	"github.com/phoreproject/bls"
)

func init() {
	if rc := C.blsmi_init_devices(0); rc != 0 {
		panic("blsmi: no usable MI355X (or RCCL missing on a multi-GPU node)")
	}
}

func u8(b []byte) *C.uint8_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&b[0]))
}

// packKeys: n*96 B affine public keys (g1.go:157-167) + one flag byte per key (bit 0 = infinity).
func packKeys(pubs []*PublicKey) (pk, inf []byte) {
	pk = make([]byte, 0, 96*len(pubs))
	inf = make([]byte, len(pubs))
	for i := range pubs {
		pa := pubs[i].p.ToAffine() // g1.go:322-340
		if pa.IsZero() {
			inf[i] |= 1
		}
		pb := pa.SerializeBytes() // all zero for infinity: the library reads that as infinity too
		pk = append(pk, pb[:]...)
	}
	return
}

// packSigs: n*192 B affine signatures (g2.go:172-186); bit 1 of inf[i] marks infinity.
func packSigs(sigs []*Signature, inf []byte) (sg []byte) {
	sg = make([]byte, 0, 192*len(sigs))
	for i := range sigs {
		sa := sigs[i].s.ToAffine() // g2.go:365-386
		if sa.IsZero() {
			inf[i] |= 2
		}
		sb := sa.SerializeBytes()
		sg = append(sg, sb[:]...)
	}
	return
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
	pk, inf := packKeys(pubs)
	sg := packSigs(sigs, inf)
	ok := make([]byte, n)
	if rc := C.blsmi_g1pubs_verify_batch(u8(m), &off[0], u8(pk), u8(sg), u8(inf), u8(ok), nil, C.size_t(n)); rc != 0 {
		panic("blsmi: g1pubs verify_batch failed")
	}
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
	pk, inf := packKeys(pubs)
	sg := packSigs(sigs, inf)
	ok := make([]byte, n)
	rc := C.blsmi_g1pubs_verify_with_domain_batch((*C.uint8_t)(unsafe.Pointer(&msgs[0])), (*C.uint8_t)(unsafe.Pointer(&domain[0])),
		u8(pk), u8(sg), u8(inf), u8(ok), nil, C.size_t(n))
	if rc != 0 {
		panic("blsmi: g1pubs verify_with_domain_batch failed")
	}
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
	pk, _ := packKeys(pubKeys)            // a key at infinity travels as the all-zero record: verdict false
	sb := s.s.ToAffine().SerializeBytes() // likewise for the signature
	var ok C.int
	rc := C.blsmi_g1pubs_verify_aggregate(u8(m), &off[0], u8(pk), (*C.uint8_t)(unsafe.Pointer(&sb[0])), C.size_t(len(msgs)), &ok)
	return rc == 0 && ok != 0
}

// VerifyAggregateCommon keeps the upstream signature (g1pubs/bls.go:287).
func (s *Signature) VerifyAggregateCommon(pubKeys []*PublicKey, msg []byte) bool {
	if C.blsmi_prefer_cpu(C.BLSMI_SHAPE_POINT_ADD, C.size_t(len(pubKeys))) != 0 {
		return Verify(msg, AggregatePublicKeys(pubKeys), s) // a handful of keys: sum them on the upstream path
	}
	pk, _ := packKeys(pubKeys)
	sb := s.s.ToAffine().SerializeBytes()
	one := []byte{0}
	mp := u8(msg)
	if len(msg) == 0 {
		mp = u8(one)
	}
	var ok C.int
	rc := C.blsmi_g1pubs_verify_aggregate_common(mp, C.size_t(len(msg)), u8(pk), (*C.uint8_t)(unsafe.Pointer(&sb[0])), C.size_t(len(pubKeys)), &ok)
	return rc == 0 && ok != 0
}

// VerifyAggregateCommonWithDomain keeps the upstream signature (g1pubs/bls.go:294).
func (s *Signature) VerifyAggregateCommonWithDomain(pubKeys []*PublicKey, msg [32]byte, domain [8]byte) bool {
	pk, _ := packKeys(pubKeys)
	sb := s.s.ToAffine().SerializeBytes()
	var ok C.int
	rc := C.blsmi_g1pubs_verify_aggregate_common_with_domain((*C.uint8_t)(unsafe.Pointer(&msg[0])), (*C.uint8_t)(unsafe.Pointer(&domain[0])),
		u8(pk), (*C.uint8_t)(unsafe.Pointer(&sb[0])), C.size_t(len(pubKeys)), &ok)
	return rc == 0 && ok != 0
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
	pk, _ := packKeys(pubKeys)
	sb := s.s.ToAffine().SerializeBytes()
	var ok C.int
	rc := C.blsmi_g1pubs_verify_aggregate_with_domain((*C.uint8_t)(unsafe.Pointer(&msgs[0])), (*C.uint8_t)(unsafe.Pointer(&domain[0])),
		u8(pk), (*C.uint8_t)(unsafe.Pointer(&sb[0])), C.size_t(len(msgs)), &ok)
	return rc == 0 && ok != 0
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
	rc := C.blsmi_g1pubs_verify_serialized_batch(u8(m), &off[0],
		(*C.uint8_t)(unsafe.Pointer(&pubs[0])), (*C.uint8_t)(unsafe.Pointer(&sigs[0])), 1,
		u8(ok), nil, nil, C.size_t(n))
	if rc != 0 {
		panic("blsmi: verify_serialized_batch failed")
	}
	for i := range ok {
		out[i] = ok[i] != 0
	}
	return out
}

// g2FromBytes rebuilds a G2 point from the 192 affine bytes the library returns (x.c0, x.c1, y.c0, y.c1, 48 bytes big-endian each:
// G2Affine.SerializeBytes, g2.go:172-186).
func g2FromBytes(b []byte) *bls.G2Affine {
	var c [4][48]byte
	for j := range c {
		copy(c[j][:], b[48*j:48*j+48])
	}
	fq := func(x [48]byte) bls.FQ { return bls.FQReprToFQ(bls.FQReprFromBytes(x)) }
	return bls.NewG2Affine(bls.NewFQ2(fq(c[0]), fq(c[1])), bls.NewFQ2(fq(c[2]), fq(c[3])))
}

func secretBytes(keys []*SecretKey) []byte {
	sk := make([]byte, 0, 32*len(keys))
	for i := range keys {
		kb := keys[i].Serialize() // 32 bytes big-endian
		sk = append(sk, kb[:]...)
	}
	return sk
}

func sigsFromBytes(sg, inf []byte) []*Signature {
	out := make([]*Signature, len(inf))
	for i := range out {
		if inf[i] != 0 {
			out[i] = NewSignatureFromG2(bls.G2AffineZero.Copy())
		} else {
			out[i] = NewSignatureFromG2(g2FromBytes(sg[192*i : 192*i+192]))
		}
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
	sg := make([]byte, 192*n)
	inf := make([]byte, n)
	if rc := C.blsmi_g1pubs_sign_batch(u8(m), &off[0], u8(secretBytes(keys)), u8(sg), u8(inf), C.size_t(n)); rc != 0 {
		panic("blsmi: g1pubs sign_batch failed")
	}
	return sigsFromBytes(sg, inf)
}

// SignWithDomainBatch: out[i] = SignWithDomain(msgs[i], keys[i], domain) (g1pubs/bls.go:138-141).
func SignWithDomainBatch(msgs [][32]byte, keys []*SecretKey, domain [8]byte) []*Signature {
	n := len(msgs)
	if n == 0 {
		return nil
	}
	sg := make([]byte, 192*n)
	inf := make([]byte, n)
	rc := C.blsmi_g1pubs_sign_with_domain_batch((*C.uint8_t)(unsafe.Pointer(&msgs[0])), (*C.uint8_t)(unsafe.Pointer(&domain[0])),
		u8(secretBytes(keys)), u8(sg), u8(inf), C.size_t(n))
	if rc != 0 {
		panic("blsmi: g1pubs sign_with_domain_batch failed")
	}
	return sigsFromBytes(sg, inf)
}
