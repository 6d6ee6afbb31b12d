#!/bin/bash
# Reproduces the rocprofv3 evidence under profiles/: per-kernel durations of the default bench.py run -- every leg, i.e. every
# BASELINE config -- (--kernel-trace --stats) and, in separate passes as the MI355X guide prescribes, the HBM traffic and SQ
# counters of the same command.
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag> [commit]      e.g. r03a 3271d4e
# outputs: gpurun_out/<tag>_rocprof_summary.txt, gpurun_out/<tag>_counters.json (copy both to profiles/), <tag>_bench.json
set -u
TAG=${1:-r03zh}
export BLSMI_COMMIT=${2:-unknown}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-ref-shapes"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o kt -- $B --steps 3 --warmup 1 > $R/gpurun_out/${TAG}_bench_stdout.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_$TAG -o pmc -- $B --steps 2 --warmup 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_$TAG -o pmc -- $B --steps 2 --warmup 0 > /dev/null 2>&1
# (SQ_ACTIVE_INST_VALU reads the same as SQ_INSTS_VALU on gfx950 -- profiles/r04m_counters.json -- and gave way to the INT64 / INT32 split: the executed
#  64-bit integer VALU instructions are what bench.py's valu.frac_mix weighs at the VOP3 rate, tools/isa_mix.py)
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $R/gpurun_out/pmc_sq_$TAG -o pmc -- $B --steps 2 --warmup 0 > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/${TAG}_rocprof_summary.txt $(find gpurun_out/prof_$TAG gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG gpurun_out/pmc_sq_$TAG -name "*.db" | sort) > /dev/null
grep metric gpurun_out/${TAG}_bench_stdout.log > gpurun_out/${TAG}_bench.json
rm -rf gpurun_out/prof_$TAG gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG gpurun_out/pmc_sq_$TAG     # the .db files are large; the summaries are what is kept
echo profile $TAG done
