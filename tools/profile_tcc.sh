#!/bin/bash
# does the memory side bind the scalar-multiplication kernels?  L2 (TCC) hit / miss and the memory-side request
# counters of k_g1_mul_glv, k_g2_mul_glv_pair and k_miller1x2_pair (tools/scalar_mul_profile.py), with FETCH_SIZE / WRITE_SIZE calibrated
# on a known scratch pattern (tools/ubench_scratch.hip).  Counter passes only (--pmc with no trace domains), one pass per counter group.
# usage (GPU box, repo root): tools/profile_tcc.sh <tag>      -> gpurun_out/<tag>_tcc_summary.txt, <tag>_tcc_available.txt
set -u
TAG=${1:-r04_tcc}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\(HIT\|MISS\|RDREQ\|WRREQ\|MALL\|REQ\)[A-Za-z0-9_]*" | sort -u > $R/gpurun_out/${TAG}_available.txt
have() { grep -qx "$1" $R/gpurun_out/${TAG}_available.txt; }
G1="FETCH_SIZE"; G2="WRITE_SIZE"
G3=""; for c in TCC_HIT_sum TCC_MISS_sum; do have $c && G3="$G3 $c"; done
G4=""; for c in TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum; do have $c && G4="$G4 $c"; done
DBS=""
i=0
for G in "$G1" "$G2" "$G3" "$G4"; do
  i=$((i+1))
  [ -z "$G" ] && continue
  rocprofv3 --pmc $G -d $R/gpurun_out/tcc_${TAG}_cal_$i -o pmc -- $R/tools/ubench_scratch 4 > $R/gpurun_out/${TAG}_cal_$i.log 2>&1
  rocprofv3 --pmc $G -d $R/gpurun_out/tcc_${TAG}_mul_$i -o pmc -- python $R/tools/scalar_mul_profile.py > $R/gpurun_out/${TAG}_mul_$i.log 2>&1
done
cd $R
python tools/rocpd_summary.py gpurun_out/${TAG}_summary.txt $(find gpurun_out/tcc_${TAG}_* -name "*.db" | sort) > /dev/null
grep -h "k_scratch_pattern\|mul 2^20\|VerifyAggregate" gpurun_out/${TAG}_cal_1.log gpurun_out/${TAG}_mul_1.log >> gpurun_out/${TAG}_summary.txt
rm -rf gpurun_out/tcc_${TAG}_*
echo tcc profile $TAG done
