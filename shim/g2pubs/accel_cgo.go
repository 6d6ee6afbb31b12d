// +build cgo,blsmi

// accel_cgo.go: the verify surface of package g2pubs (PublicKey in G2, Signature in G1) on libblsmi.so.
//
// Drop this file into github.com/phoreproject/bls/g2pubs and build with `-tags blsmi`; the upstream
// implementations of Verify, VerifyAggregate and VerifyAggregateCommon (g2pubs/bls.go:159-162, 240-270,
// 275-278) move behind `// +build !blsmi`.  Everything else in bls.go (types, Sign, PrivToPub,
// serialisation, the aggregation helpers) stays as it is; SignBatch below is an addition for bulk signing.  The C prototypes are include/blsmi.h;
// tests/test_shim.py checks every C.blsmi_* call below against it (name and arity).
package g2pubs

/*
#cgo CFLAGS: -I${SRCDIR}/../../blsmi/include
#cgo LDFLAGS: -L${SRCDIR}/../../blsmi/bls_amd -lblsmi -Wl,-rpath,${SRCDIR}/../../blsmi/bls_amd
#include "blsmi.h"
*/
import "C"

import (
	"runtime"
	"unsafe"

	"github.com/phoreproject/bls"
)

func init() {
	// Every visible device (<= 0), or the first n.  Must precede any other blsmi call.  With more than one
	// device, batches of BLSMI_SHARD_MIN tuples or more are split over the GPUs inside the library.
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

// packKeys: n*192 B affine public keys (G2Affine.SerializeBytes, g2.go:172-186); bit 0 of inf[i] marks infinity.
func packKeys(pubs []*PublicKey, inf []byte) (pk []byte) {
	pk = make([]byte, 0, 192*len(pubs))
	for i := range pubs {
		pa := pubs[i].p.ToAffine() // g2.go:365-386
		if pa.IsZero() {
			inf[i] |= 1
		}
		pb := pa.SerializeBytes() // all zero for infinity: the library reads that as infinity too
		pk = append(pk, pb[:]...)
	}
	return
}

// packSigs: n*96 B affine signatures (G1Affine.SerializeBytes, g1.go:157-167); bit 1 of inf[i] marks infinity.
func packSigs(sigs []*Signature, inf []byte) (sg []byte) {
	sg = make([]byte, 0, 96*len(sigs))
	for i := range sigs {
		sa := sigs[i].s.ToAffine() // g1.go:322-340
		if sa.IsZero() {
			inf[i] |= 2
		}
		sb := sa.SerializeBytes()
		sg = append(sg, sb[:]...)
	}
	return
}

// VerifyBatch is the batch form the one-tuple-per-call API lacks: out[i] = Verify(msgs[i], pubs[i], sigs[i]).
func VerifyBatch(msgs [][]byte, pubs []*PublicKey, sigs []*Signature) []bool {
	n := len(msgs)
	out := make([]bool, n)
	if n == 0 {
		return out
	}
	m, off := packMsgs(msgs)
	inf := make([]byte, n)
	pk := packKeys(pubs, inf)
	sg := packSigs(sigs, inf)
	ok := make([]byte, n)
	rc := C.blsmi_g2pubs_verify_batch(u8(m), &off[0], u8(pk), u8(sg), u8(inf), u8(ok), nil, C.size_t(n))
	if rc != 0 {
		panic("blsmi: g2pubs verify_batch failed")
	}
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
// rejection (bls.go:245-261) inside the library.
func (s *Signature) VerifyAggregate(pubKeys []*PublicKey, msgs [][]byte) bool {
	if len(pubKeys) != len(msgs) {
		return false
	}
	m, off := packMsgs(msgs)
	inf := make([]byte, len(pubKeys))
	pk := packKeys(pubKeys, inf) // a key at infinity travels as the all-zero record: verdict false
	sa := s.s.ToAffine()
	if sa.IsZero() {
		return false
	}
	sb := sa.SerializeBytes()
	var ok C.int
	rc := C.blsmi_g2pubs_verify_aggregate(u8(m), &off[0], u8(pk), (*C.uint8_t)(unsafe.Pointer(&sb[0])), C.size_t(len(msgs)), &ok)
	return rc == 0 && ok != 0
}

// VerifyAggregateCommon keeps the upstream signature (g2pubs/bls.go:275): the key sum stays on the upstream
// path for small sets (one Jacobian addition is 6.5 us on a CPU core) and goes to the device for large ones.
func (s *Signature) VerifyAggregateCommon(pubKeys []*PublicKey, msg []byte) bool {
	if C.blsmi_prefer_cpu(C.BLSMI_SHAPE_POINT_ADD, C.size_t(len(pubKeys))) != 0 {
		return Verify(msg, AggregatePublicKeys(pubKeys), s)
	}
	inf := make([]byte, len(pubKeys))
	pk := packKeys(pubKeys, inf)
	sb := s.s.ToAffine().SerializeBytes()
	one := []byte{0}
	mp := u8(msg)
	if len(msg) == 0 {
		mp = u8(one)
	}
	var ok C.int
	rc := C.blsmi_g2pubs_verify_aggregate_common(mp, C.size_t(len(msg)), u8(pk), (*C.uint8_t)(unsafe.Pointer(&sb[0])), C.size_t(len(pubKeys)), &ok)
	return rc == 0 && ok != 0
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
	rc := C.blsmi_g2pubs_verify_serialized_batch(u8(m), &off[0],
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

// PreparedKeys: n public keys run through G2AffineToPrepared (g2.go:639-801) once, resident in device
// memory the library owns (24 704 bytes per key).  Verify runs the preparation on every call upstream
// (pairing.go:140-147); a validator set prepares once and passes key INDICES afterwards.
type PreparedKeys struct {
	h unsafe.Pointer
	n int
}

func PrepareKeys(pubs []*PublicKey) *PreparedKeys {
	inf := make([]byte, len(pubs))
	pk := packKeys(pubs, inf) // all-zero record = infinity: every verdict over that key is false
	var h unsafe.Pointer
	if rc := C.blsmi_g2_prepared_create(u8(pk), C.size_t(len(pubs)), &h); rc != 0 {
		panic("blsmi: prepare failed")
	}
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
	inf := make([]byte, n) // any non-zero flag byte makes verdict i false (bit 1: signature at infinity)
	sg := packSigs(sigs, inf)
	ok := make([]byte, n)
	rc := C.blsmi_g2pubs_verify_batch_prepared(u8(m), &off[0], keys.h, (*C.uint32_t)(unsafe.Pointer(&keyIdx[0])),
		u8(sg), u8(inf), u8(ok), nil, C.size_t(n))
	if rc != 0 {
		panic("blsmi: verify_batch_prepared failed")
	}
	for i := range ok {
		out[i] = ok[i] != 0
	}
	return out
}

// staging is a reusable page-locked buffer (blsmi_host_alloc): copies from it are single DMAs instead of
// being staged by the HIP runtime.  b aliases C memory (no Go pointers inside: safe to hand to cgo as-is);
// keep one set per goroutine that calls into the library and serialise points straight into b[:0].
type staging struct {
	p unsafe.Pointer
	b []byte
}

func newStaging(n int) *staging {
	var p unsafe.Pointer
	if rc := C.blsmi_host_alloc(C.size_t(n), &p); rc != 0 {
		panic("blsmi_host_alloc")
	}
	return &staging{p: p, b: unsafe.Slice((*byte)(p), n)}
}

func (s *staging) free() {
	C.blsmi_host_free(s.p)
	s.p, s.b = nil, nil
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
	sg := make([]byte, 96*n)
	inf := make([]byte, n)
	if rc := C.blsmi_g2pubs_sign_batch(u8(m), &off[0], u8(sk), u8(sg), u8(inf), C.size_t(n)); rc != 0 {
		panic("blsmi: g2pubs sign_batch failed")
	}
	for i := range out {
		if inf[i] != 0 {
			out[i] = NewSignatureFromG1(bls.G1AffineZero.Copy())
			continue
		}
		var xb, yb [48]byte
		copy(xb[:], sg[96*i:96*i+48])
		copy(yb[:], sg[96*i+48:96*i+96])
		out[i] = NewSignatureFromG1(bls.NewG1Affine(bls.FQReprToFQ(bls.FQReprFromBytes(xb)), bls.FQReprToFQ(bls.FQReprFromBytes(yb))))
	}
	return out
}
