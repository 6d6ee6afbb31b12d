#!/bin/bash
# Reproduces the rocprofv3 evidence under profiles/: per-kernel durations of the default bench.py run
# (--kernel-trace --stats) and, in separate passes as the MI355X guide prescribes, the HBM traffic counters.
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
set -u
TAG=${1:-r01x}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_stdout.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_$TAG -o pmc -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-verify-extra > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_$TAG -o pmc -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-verify-extra > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $R/gpurun_out/pmc_sq_$TAG -o pmc -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-verify-extra > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/${TAG}_rocprof_summary.txt $(find gpurun_out/prof_$TAG gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG gpurun_out/pmc_sq_$TAG -name "*.db" | sort)
grep metric gpurun_out/${TAG}_bench_stdout.log > gpurun_out/${TAG}_bench.json
echo profile $TAG done
