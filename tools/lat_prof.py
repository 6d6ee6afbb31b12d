import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bls_amd import engine
from oracle import refcpu as RC
engine.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
p1 = RC.g1_mul(RC.g1_generator(), (12345).to_bytes(32, "big")); q1 = RC.g2_mul(RC.g2_generator(), (987).to_bytes(32, "big"))
for _ in range(4):
    engine.pairing_batch(p1 * n, q1 * n, n)
