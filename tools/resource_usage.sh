#!/bin/bash
# Compiler view of every kernel's resources (VGPRs, AGPRs, scratch, occupancy): hipcc -Rpass-analysis=kernel-resource-usage
# over the kernel translation units.  usage: tools/resource_usage.sh profiles/rNN_kernel_resource_usage.txt
OUT=${1:-/tmp/kernel_resource_usage.txt}
cd "$(dirname "$0")/../bls_amd/csrc" || exit 1
: > "$OUT.tmp"
UNITS="k_pairing_pair k_fe_pair k_pairing_quad k_pairing_row k_prepared_pair k_pairing_single k_fe_single k_fq12_single k_hash k_wire k_hash_pair k_hash_quad k_curve k_msm_pair k_lat k_util"
L28=" k_pairing_pair k_fe_pair k_pairing_quad k_pairing_row k_prepared_pair k_hash k_wire k_hash_pair k_hash_quad k_curve k_msm_pair "    # the 14 x 28-bit units (bls_amd/_native.py: _LIMBS28_UNITS)
for u in $UNITS; do
  [ -f $u.hip ] || continue
  F=""; case "$L28" in *" $u "*) F="-DBLSMI_LIMBS28";; esac
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $F --cuda-device-only -c -o /dev/null -Rpass-analysis=kernel-resource-usage $u.hip 2>&1 \
      | grep -E "remark:" | sed -E 's/^.*remark: [^ ]+ //; s/^ +//' > "$OUT.$u" ) &
done
wait
for u in $UNITS; do
  [ -f "$OUT.$u" ] || continue
  echo "== $u.hip" >> "$OUT.tmp"; cat "$OUT.$u" >> "$OUT.tmp"; rm -f "$OUT.$u"
done
mv "$OUT.tmp" "$OUT"
grep -c "Function Name" "$OUT"
