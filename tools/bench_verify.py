#!/usr/bin/env python3
"""bench.py's contract applied to the batch-VERIFY workload (BASELINE configs 1 / 5 at GPU sizes) instead of bare pairings:
a step = every rank verifies its own block of T (message, public key, signature) tuples resident in HBM -- hash-to-curve,
2-pair Miller loop, final exponentiation, compare -- packs the verdicts into its slice of the world*T-bit map, and ONE
RCCL all-reduce completes the map on every rank (the north-star's only collective).  One JSON line on stdout.

    python tools/bench_verify.py --group g2pubs --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 tools/bench_verify.py --gpus 8
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tuples", type=int, default=65536, help="tuples per GPU per step")
    ap.add_argument("--group", choices=["g2pubs", "g1pubs"], default="g2pubs")
    args = ap.parse_args()
    sys.stdout.flush(); saved = os.dup(1); os.dup2(2, 1)                 # one line on stdout, banners to stderr
    import torch
    import torch.distributed as dist
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("nccl", device_id=dev)
    from bls_amd import engine
    engine.init(local_rank)
    n = args.tuples
    d = bench._verify_inputs(engine, dev, args.group, n)
    ok = torch.zeros(n, dtype=torch.uint8, device=dev)
    weights = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=dev)
    full = torch.zeros(world * n // 8, dtype=torch.int32, device=dev)

    def step():
        full.zero_()
        engine.verify_batch_dev(args.group, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 0, ok.data_ptr(), n)
        full[rank * n // 8:(rank + 1) * n // 8] = (ok.view(-1, 8).to(torch.int32) * weights).sum(dim=1, dtype=torch.int32)
        dist.all_reduce(full, op=dist.ReduceOp.SUM)

    def fence():
        dist.barrier(); torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence(); t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence(); dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    assert bool((full == 255).all().item()), "every tuple of every rank must verify"
    if rank == 0:
        value = world * n * args.steps / dt
        bytes_per = 320.0                                                  # 96 + 192 + 32-byte message in, 1 bit out
        line = {"metric": "BLS12-381 %s.Verify per second (batch verify, hash-to-curve included)" % args.group, "value": round(value, 1),
                "unit": "verifies/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32 (15 x 27-bit limbs, int64 accumulate)", "data": "synthetic",
                "config": {"workload": "%d %s tuples per GPU per step, inputs resident in HBM, verdict bitmap all-reduced over RCCL" % (n, args.group),
                           "tuples_per_gpu": n, "parallelism": "shard%d" % world, "bitmap_bytes": world * n // 8},
                "pairing_equivalents_per_s": round(2 * value, 1),
                "roofline": {"bound": "hbm", "achieved": round(value / world * bytes_per / 1e9, 3), "peak": bench.HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(value / world * bytes_per / 1e9 / bench.HBM_PEAK_GBS, 8), "traffic": None,
                             "note": "320 algorithmic bytes per verify; compute-bound (integer VALU), see DESIGN.md 3"}}
        sys.stdout.flush(); os.dup2(saved, 1); print(json.dumps(line), flush=True); os.dup2(2, 1)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
