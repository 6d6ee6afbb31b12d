// msm_raw.cuh -- the raw-limb point records of the endomorphism MSM (msm.inc), shared by the translation units that read them
// (k_curve.hip, k_msm_pair.hip): a point in the device's own representation together with its endomorphism images.
#pragma once
#include "fp.cuh"
namespace blsmi {
constexpr int RAW1_WORDS = 48;                                             // x | beta x | y (45 words), word 45 = infinity flag
constexpr int RAW2_WORDS = 244;                                            // 4 variants x (x.c0, x.c1, y.c0, y.c1) (240 words), word 240 = infinity flag
BLSMI_DEV FpS raw_load(const i32* p) { FpS x; for (int j = 0; j < NL; j++) x.v[j] = p[j]; return x; }
BLSMI_DEV void raw_store(i32* p, const FpS& x) { for (int j = 0; j < NL; j++) p[j] = x.v[j]; }
}  // namespace blsmi
