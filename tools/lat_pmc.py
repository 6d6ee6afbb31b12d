"""workload for a rocprofv3 PMC pass over the latency path's kernel: 1 024 and 4 096 pairings, one tuple per wave (k_lat, program pairing1).
usage (GPU box): rocprofv3 --pmc ... -- python tools/lat_pmc.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bls_amd import engine  # noqa: E402
engine.init(0)
g1, g2 = bench.synth_inputs(engine, 4096, seed=3)
for n in (1, 1024, 2048, 4096):
    for _ in range(3):
        engine.pairing_batch(g1[:n].reshape(-1), g2[:n].reshape(-1), n)
