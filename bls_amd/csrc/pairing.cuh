// pairing.cuh -- optimal-ate Miller loop and final exponentiation; the body (pairing_body.inc) is instantiated
// twice: one (P, Q) tuple per lane (namespace blsmi) and one per lane pair (namespace blsmi::pairl).
//
// Reference path: G2AffineToPrepared (g2.go:650-801) stores 68 line-coefficient triples per Q
// (19.6 KB each), MillerLoop (pairing.go:16-75) replays them, FinalExponentiation
// (pairing.go:79-129) raises to 3(q^12-1)/r with a generic square-and-multiply.
// Here the line computation is FUSED into the Miller loop (no coefficient spill: the 68 triples
// never exist in memory), with the reference's exact doubling/addition formulas so that the
// Miller-loop output -- not only the final result -- is the same field element; and the hard part
// of the final exponentiation uses Granger-Scott cyclotomic squarings along the reference's
// addition chain, which yields the same power.
#pragma once
#include "curve.cuh"

namespace blsmi {

#define BLSMI_NOINLINE __device__ __noinline__

#include "pairing_body.inc"

}  // namespace blsmi

// The same tower / Miller loop / final exponentiation over the lane-pair Fq2 layer
#include "pair_field.cuh"
namespace blsmi {
namespace pairl {
#include "tower_body.inc"
#include "pairing_body.inc"
}  // namespace pairl

}  // namespace blsmi
