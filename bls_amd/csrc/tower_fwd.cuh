// tower_fwd.cuh -- element types of the extension tower (fq2.go:13-17, fq6.go:9-14, fq12.go:9-13).
// All components of one element share one <L,V> bound (see fp.cuh).
#pragma once
namespace blsmi {
template <int L_, int V_> struct Fp2 { static constexpr int L = L_, V = V_; Fp<L_, V_> c0, c1; };
template <int L_, int V_> struct Fp6 { static constexpr int L = L_, V = V_; Fp2<L_, V_> c0, c1, c2; };
template <int L_, int V_> struct Fp12 { static constexpr int L = L_, V = V_; Fp6<L_, V_> c0, c1; };
using Fp2S = Fp2<FpS::L, FpS::V>;
using Fp6S = Fp6<FpS::L, FpS::V>;
using Fp12S = Fp12<FpS::L, FpS::V>;
}  // namespace blsmi
