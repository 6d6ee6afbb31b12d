"""Which hardware queues do concurrent callers' kernels run on?  T threads x pairing calls of n tuples under `rocprofv3 --kernel-trace`, then the
kernel trace's start / duration / queue per pairing kernel (the call contexts' streams and the HIP runtime's hardware queues).
  run (GPU box, from the repo root):
    R=$(pwd); cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/qt -- python $R/tools/queue_trace.py run 4096 4
    python $R/tools/queue_trace.py show $R/gpurun_out/qt"""
import csv
import glob
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(n, T):
    import numpy as np
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_common import P, RC
    from bls_amd import engine as E
    E.init(0)
    xs = P.XORShift(4096)
    sks = [P.rand_fr(xs).to_bytes(32, "big") for _ in range(16)]
    g1a = [RC.g2pubs.sign(b"x%d" % i, s) for i, s in enumerate(sks)]
    g2a = [RC.g2pubs.priv_to_pub(s) for s in sks]
    a1 = np.frombuffer((b"".join(g1a) * (n // 16 + 1))[:96 * n], dtype=np.uint8); a2 = np.frombuffer((b"".join(g2a) * (n // 16 + 1))[:192 * n], dtype=np.uint8)
    bar = threading.Barrier(T + 1)

    def work():
        bar.wait()
        for _ in range(6):
            E.pairing_batch(a1, a2, n)
        bar.wait()
    ts = [threading.Thread(target=work) for _ in range(T)]
    for t in ts:
        t.start()
    bar.wait(); t0 = time.perf_counter(); bar.wait()
    print("%d callers x %d pairings: %.2f ms per call (under the profiler)" % (T, n, (time.perf_counter() - t0) / 6 * 1e3))
    for t in ts:
        t.join()


def show(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "quad" in r["Kernel_Name"] or "k_lat" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    queues = sorted({r.get("Queue_Id", "?") for r in rows})
    print("hardware queues in use by the pairing kernels: %d (%s)" % (len(queues), ", ".join("q" + q for q in queues)))
    for r in rows[:32]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print("%-24s q%-3s start %8.3f ms  duration %6.3f ms" % (r["Kernel_Name"].split("(")[0][:24], r.get("Queue_Id", "?"), s / 1e6, (e - s) / 1e6))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]))
    else:
        show(sys.argv[2])
