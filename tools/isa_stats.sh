#!/bin/bash
# per-function instruction counts of the device code (glue vs v_mad), for tracking kernel optimisation
cd "$(dirname "$0")/../bls_amd/csrc" || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -S --cuda-device-only -o /tmp/blsmi.s blsmi.hip 2>&1 | grep error
for f in fp_mul_core fp_sqr_core nf_fp12_cyc_sqr nf_fp12_mul nf_fp12_sqr doubling_step addition_step 3ell 8exp_by_x; do
  body=$(awk "/^_ZN5blsmi[0-9]*${f#[0-9]}.*:/,/s_setpc_b64/" /tmp/blsmi.s)
  printf "%-18s total=%-6s mad=%-5s scratch=%-5s accvgpr=%-5s mov=%-5s swappc=%s\n" "$f" "$(echo "$body" | grep -cE '^\s+[a-z]')" "$(echo "$body" | grep -cE 'v_mad_[iu]64')" "$(echo "$body" | grep -cE 'scratch_|flat_')" "$(echo "$body" | grep -c v_accvgpr)" "$(echo "$body" | grep -cE 'v_mov_b')" "$(echo "$body" | grep -c s_swappc)"
done
