// Concurrent single-tuple callers of the C ABI without an interpreter in the way: T OS threads (what cgo gives a
// goroutine for the duration of a call), each verifying ONE (message, public key, signature) tuple per call.
// Measures what the request combining of verify_host.inc delivers.  Input: tuples dumped by tools/latency.py.
//   g++ -O2 -std=c++17 -I include tools/native/combine_bench.cpp -o tools/native/combine_bench -L bls_amd -lblsmi -Wl,-rpath,'$ORIGIN/../../bls_amd' -lpthread
//   tools/native/combine_bench gpurun_out/tuples.bin 256 8
#include "blsmi.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>

struct Tuple { std::vector<uint8_t> msg; uint8_t pk[192]; uint8_t sig[96]; };

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s tuples.bin nthreads reps\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    uint64_t n = 0;
    if (fread(&n, 8, 1, f) != 1) return 2;
    std::vector<Tuple> ts(n);
    for (auto& t : ts) {
        uint32_t len = 0;
        if (fread(&len, 4, 1, f) != 1) return 2;
        t.msg.resize(len);
        if (len && fread(t.msg.data(), 1, len, f) != len) return 2;
        if (fread(t.pk, 1, 192, f) != 192 || fread(t.sig, 1, 96, f) != 96) return 2;
    }
    fclose(f);
    const int nthreads = atoi(argv[2]), reps = atoi(argv[3]);
    if (blsmi_init(0) != 0) { fprintf(stderr, "blsmi_init failed\n"); return 1; }
    {   // warm-up (module load, generator tables)
        const Tuple& t = ts[0]; uint64_t off[2] = {0, t.msg.size()}; uint8_t ok = 0;
        blsmi_g2pubs_verify_batch(t.msg.data(), off, t.pk, t.sig, nullptr, &ok, nullptr, 1);
    }
    std::atomic<long> good{0}, bad{0};
    auto work = [&](int id) {
        const Tuple& t = ts[id % n];
        uint64_t off[2] = {0, t.msg.size()};
        for (int r = 0; r < reps; r++) {
            uint8_t ok = 0;
            const int rc = blsmi_g2pubs_verify_batch(t.msg.data(), off, t.pk, t.sig, nullptr, &ok, nullptr, 1);
            (rc == 0 && ok) ? good++ : bad++;
        }
    };
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int i = 0; i < nthreads; i++) th.emplace_back(work, i);
    for (auto& x : th) x.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("threads=%d reps=%d: %ld verified, %ld failed, %.3f s -> %.0f single-tuple verifies/s (%.1f ms per call)\n",
           nthreads, reps, good.load(), bad.load(), dt, (double)nthreads * reps / dt, dt / reps * 1e3);
    return bad.load() ? 1 : 0;
}
