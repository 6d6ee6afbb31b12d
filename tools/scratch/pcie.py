import torch, time
for mb in (2, 28, 128):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8).random_(0, 255)
    hp = h.pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for name, src in (("pageable", h), ("pinned", hp)):
        for _ in range(3): d.copy_(src); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10): d.copy_(src, non_blocking=True); torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 10
        print(mb, "MB H2D", name, "%.2f ms %.1f GB/s" % (dt * 1e3, n / dt / 1e9))
    t = time.perf_counter()
    for _ in range(10): hp.copy_(h)
    dt = (time.perf_counter() - t) / 10
    print(mb, "MB host memcpy %.2f ms %.1f GB/s" % (dt * 1e3, n / dt / 1e9))
    t = time.perf_counter(); hp2 = torch.empty(n, dtype=torch.uint8).pin_memory(); print("pin alloc %.2f ms" % ((time.perf_counter() - t) * 1e3))
    t = time.perf_counter(); torch.cuda.cudart().cudaHostRegister(h.data_ptr(), n, 0); print("register %.2f ms" % ((time.perf_counter() - t) * 1e3))
    for _ in range(3): d.copy_(h, non_blocking=True); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10): d.copy_(h, non_blocking=True); torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    print(mb, "MB H2D registered %.2f ms %.1f GB/s" % (dt * 1e3, n / dt / 1e9))
    t = time.perf_counter(); torch.cuda.cudart().cudaHostUnregister(h.data_ptr()); print("unregister %.2f ms" % ((time.perf_counter() - t) * 1e3))
