import time, sys
sys.path.insert(0, "/root/repo")
from bls_amd import engine as eng
eng.init(0)
def best(f,r=4):
    f(); b=1e9
    for _ in range(r):
        t=time.perf_counter(); f(); b=min(b,time.perf_counter()-t)
    return b*1e3
for n in (256,512,1024,2048,4096):
    m=[b"m%d"%i for i in range(n)]
    pm=eng.PackedMsgs(m)
    print(n, "hash_g1 %.3f ms  hash_g2 %.3f ms"%(best(lambda: eng.hash_g1_batch(pm)), best(lambda: eng.hash_g2_batch(pm))))
