#!/bin/bash
# throughput of tools/soak9.py (every verdict checked) over GPU_MAX_HW_QUEUES x caller threads x call contexts (BLSMI_STREAMS),
# one fresh process per cell, each under its own timeout.   usage: tools/queue_matrix.sh [seconds per cell] > gpurun_out/queue_matrix.log
S=${1:-25}
echo "# soak9 cells of $S s: GPU_MAX_HW_QUEUES x threads x BLSMI_STREAMS -> verifies/s (rc of the process; 124 = killed by timeout)"
for Q in 4 6 8; do for T in 4 8 16; do for C in 4 6 8; do
  out=$(GPU_MAX_HW_QUEUES=$Q BLSMI_STREAMS=$C timeout $((S + 90)) python tools/soak9.py $S $T 2>&1 | tail -1)
  rc=$?
  echo "queues=$Q threads=$T contexts=$C rc=$rc :: $out"
done; done; done
